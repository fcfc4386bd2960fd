#!/bin/bash
# how much of a step the GPU is idle (union of kernel intervals of the default schedule): rocprofv3 --kernel-trace of a short bench run
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rm -rf /tmp/idle
rocprofv3 --kernel-trace --output-format csv -d /tmp/idle -o t -- python3 bench.py --steps 6 --warmup 3 --settle 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 scripts/timeline.py $(find /tmp/idle -name "*kernel_trace.csv" | head -1) 4
