cd "${GRAFT_REPO_ROOT:-.}"
for d in 0 1 2 4 8 3 7 15; do echo "PCUDA_DBG=$d"; PCUDA_DBG=$d python scripts/conv_micro.py g32 g128 d4 2>&1 | grep -v amdgpu; done
