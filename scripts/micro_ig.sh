cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -3
for f in 1 0; do echo "PCUDA_FAT=$f"; PCUDA_FAT=$f python scripts/conv_micro.py g32 g64 g128 g256 b512 d2 d4 2>&1 | grep -v amdgpu; done
