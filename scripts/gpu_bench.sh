#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3 || { echo "smoke failed: not benchmarking"; exit 1; }
python bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-2} "$@" 2>&1 | tail -5 | tee gpurun_out/bench_tail.log
exit ${PIPESTATUS[0]}
