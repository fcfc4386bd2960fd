#!/bin/bash
# round 6: are the convolution kernels clock- (power-) limited?  The production kernels and the anti-phase micro on random
# operands against the same instruction streams on zeros; the bare matrix-pipe rates by operand data and MFMA shape.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out/power_cap.txt
{
echo "# production kernels (scripts/conv_micro.py, n = 32, bf16x3), 50 launches each"
for dm in random zerow zeros; do
  echo "## operand data: $dm"
  MICRO_DATA=$dm MICRO_REPS=50 python3 scripts/conv_micro.py g64 g256 g128 g32 b512 d2 d4 2>&1 | grep -v Warning
done
echo "# anti-phase micro (scripts/micro/conv3ap_micro.hip), forward + statistics, 200 launches"
for dm in 0 2 1; do
  echo "## AP_DATA=$dm (0 random, 2 zero weights, 1 all zero)"
  AP_DATA=$dm scripts/micro/bin/conv3ap_micro 32 64 64 128 128 200 | grep time
  AP_DATA=$dm scripts/micro/bin/conv3ap_micro 32 256 256 32 32 200 | grep time
  AP_DATA=$dm scripts/micro/bin/conv3ap_micro 32 128 128 64 64 200 | grep time
done
echo "# bare matrix pipe (scripts/micro/mfma_power.hip)"
scripts/micro/bin/mfma_power
} 2>&1 | tee $O
