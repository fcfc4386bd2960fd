#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -q -m gpu -x "$@" 2>&1 | tail -40
exit ${PIPESTATUS[0]}
