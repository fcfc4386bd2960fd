#!/bin/bash
# re-tune sweep of the plan rules' environment switches at step level (round 6: the rules of rounds 3-5 were measured on a library
# built WITH SLP vectorisation): baseline, knob, baseline, knob ... on one box.   usage: bash scripts/knob_sweep.sh > gpurun_out/knob_sweep.txt
cd "${GRAFT_REPO_ROOT:-.}"
run() { env "$@" python bench.py --steps 50 --warmup 15 --settle 20 --no-cpu-baseline --no-roofline 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-32s %7.2f img/s %7.3f ms  %.3f GHz' % ('$*', d['value'], d['ms_per_step'], d['clock_ghz_under_load']))"; }
KNOBS="${KNOBS:-PCUDA_NO_WGRAD3=1 PCUDA_W3_MINCOUT=64 PCUDA_W8=0 PCUDA_NO_UNDERFILL=1 PCUDA_DGRAD_PAIR=0 PCUDA_WG_NO8=1 PCUDA_WG_BLOCKS1=512 PCUDA_WG_BLOCKS32=1024 PCUDA_WG_BLOCKS64=768 PCUDA_WG3_BLOCKS=768 PCUDA_W3R_BLOCKS=1024 PCUDA_W3R_BLOCKS=256 PCUDA_WG1_BLOCKS=512 PCUDA_FUSE_LRELU_DGRAD=0 PCUDA_FUSE_POOL_BWD=0 PCUDA_NO_WGRAD1=1 PCUDA_NOXQ=1 PCUDA_AP_MIN_ITEMS=128 PCUDA_AP_MIN_ITEMS=256}"
for k in $KNOBS; do
  run PCUDA_NOOP=1
  run $k
done
run PCUDA_NOOP=1
