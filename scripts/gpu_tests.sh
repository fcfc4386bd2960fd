#!/bin/bash
# helper for gpurun: run a pytest selection on the GPU box, keep the log under gpurun_out/
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0))"
timeout ${T:-900} python -m pytest ${@:-tests -m gpu} -q --no-header -p no:cacheprovider 2>&1 | tail -${TAIL:-60} | tee gpurun_out/pytest_tail.log
