cd "${GRAFT_REPO_ROOT:-.}"
for d in 0 16 32 64 48 112; do echo "PCUDA_DBG=$d"; PCUDA_DBG=$d python scripts/conv_micro.py g32 g64 g128 g256 d4 2>&1 | grep -v amdgpu | sed 's/fwd.*| wgrad/wgrad/'; done
