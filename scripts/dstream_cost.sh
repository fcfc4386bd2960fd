#!/bin/bash
# what do the discriminators' update passes cost the step, and what do their streams hide?
# same box, alternating: default schedule | one stream | update passes left out (streams) | update passes left out (one stream)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | \
  python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-52s' % '$*', d['value'], 'img/s', d['ms_per_step'], 'ms', d['clock_ghz_under_load'], 'GHz')"; }
for rep in 1 2; do
  run PCUDA_DSTREAMS=1
  run PCUDA_DSTREAMS=0
  run PCUDA_DSTREAMS=1 PCUDA_EXP_SKIP_DUPDATE=1
  run PCUDA_DSTREAMS=0 PCUDA_EXP_SKIP_DUPDATE=1
done 2>&1 | tee gpurun_out/dstream_cost.txt
