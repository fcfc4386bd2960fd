#!/bin/bash
# same-box A / B of the anti-phase kernel in the step: with the discriminators on their streams (the default schedule) and
# with one stream (PCUDA_DSTREAMS=0)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for rep in 1 2; do
  for ds in 1 0; do
    for ap in 0 1; do
      PCUDA_DSTREAMS=$ds PCUDA_AP=$ap python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('dstreams $ds ap $ap', d['value'], d['ms_per_step'], d['clock_ghz_under_load'])"
    done
  done
done 2>&1 | tee gpurun_out/ap_ab.txt
