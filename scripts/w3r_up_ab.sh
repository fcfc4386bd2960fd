#!/bin/bash
# A / B of the up-convolutions' weight gradient: generic wgrad_kernel (PCUDA_WGRAD3R=0) against the register-window kernel reading
# the input at its stored resolution (conv_wgrad3r.hip, UP).  usage: bash scripts/w3r_up_ab.sh > gpurun_out/w3r_up_ab.txt
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do
  echo "== generic (PCUDA_WGRAD3R=0)"; PCUDA_WGRAD3R=0 python3 scripts/conv_micro.py u6432 u12864 u256128 u512256 2>&1 | grep -v amdgpu.ids
  echo "== register window, up to 512 channels (PCUDA_W3R_UPMAXC=512)"; PCUDA_W3R_UPMAXC=512 python3 scripts/conv_micro.py u6432 u12864 u256128 u512256 2>&1 | grep -v amdgpu.ids
done
