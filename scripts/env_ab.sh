#!/bin/bash
# same-box A / B of the step under one environment switch: usage VAR=PCUDA_W3R_UPMAXC A=0 B=512 bash scripts/env_ab.sh [workloads...]
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
WLS="${@:-full_uda}"
for rep in 1 2 3; do
  for wl in $WLS; do
    for v in "$A" "$B"; do
      env $VAR=$v python bench.py --workload $wl --steps 60 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$wl $VAR=$v', d['value'], d['ms_per_step'], d['clock_ghz_under_load'])"
    done
  done
done
