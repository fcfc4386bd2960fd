#!/bin/bash
# is the kernel power- (clock-) limited?  the same instruction stream on random data, on zero weights, on all-zero operands
B=scripts/micro/bin/conv3ap_micro
mkdir -p gpurun_out
O=gpurun_out/conv3ap_exp.txt
: > $O
for rep in 1 2; do
for shape in "32 256 256 32 32" "32 64 64 128 128"; do
  for dm in 0 2 1; do
    echo "== $shape data $dm" | tee -a $O
    AP_DATA=$dm $B $shape 200 0 0 2>&1 | grep "time" | tee -a $O
    AP_DATA=$dm $B $shape 200 0 1 2>&1 | grep "time\|group A" | tee -a $O
  done
done
done
