#!/bin/bash
# what the epilogue's time is made of (DBG build, mode 1): dbg 16 = no output stores, 32 = no partial sums, 48 = neither
B=scripts/micro/bin/conv3ap_micro
mkdir -p gpurun_out
O=gpurun_out/conv3ap_exp.txt
: > $O
for d in 0 16 32 48 2; do
  echo "== dbg $d" | tee -a $O
  $B 32 64 64 128 128 50 $d 1 2>&1 | grep "time\|group A" | tee -a $O
done
