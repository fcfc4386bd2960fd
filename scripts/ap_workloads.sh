#!/bin/bash
# the anti-phase kernel on the other workloads (same box, PCUDA_AP=0 / 1): config 2 and config 4's per-rank shape (B = 16),
# the 512x512 stand-in, the reference's 224x224 shape
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_conv_ap_gpu.py tests/test_conv_gpu.py -x -q 2>&1 | tail -3
for wl in unet_d2 mmwhs_uda uda_512 mscmrseg_224; do
  for ap in 0 1; do
    PCUDA_AP=$ap python bench.py --workload $wl --steps 40 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$wl ap $ap', d['value'], d['ms_per_step'], d['clock_ghz_under_load'], d['fallback_launches'])"
  done
done 2>&1 | tee gpurun_out/ap_workloads.txt
