#!/bin/bash
# stage-complete micro of the anti-phase 3x3 kernel: correctness on small shapes first (the script stops at the first
# failed check or runtime fault), then timings on the go / no-go shapes
B=scripts/micro/bin/conv3ap_micro
mkdir -p gpurun_out
O=gpurun_out/conv3ap_micro.txt
: > $O
run() { "$@" 2>&1 | tee -a $O | tee /tmp/ap_last.txt; if grep -q "FAILED\|fault\|error" /tmp/ap_last.txt; then echo "STOP: $*" | tee -a $O; exit 1; fi; }
run $B 2 64 64 32 32 5
run $B 2 32 128 16 64 5
run $B 4 128 64 24 32 5
run $B 6 64 128 40 96 5
run $B 32 64 64 128 128 30
run $B 32 256 256 32 32 30
run $B 32 128 128 64 64 30
run $B 32 32 64 256 256 30
for d in 0 2 4; do run $B 32 64 64 128 128 30 $d 1; done
for d in 0 2 4; do run $B 32 256 256 32 32 30 $d 1; done
