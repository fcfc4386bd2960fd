cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -3
echo "W8=1"; PCUDA_W8=1 python scripts/conv_micro.py g32 g64 g128 g256 b512 d2 d4 2>&1 | grep -v amdgpu
echo "W8=0"; PCUDA_W8=0 python scripts/conv_micro.py g32 g64 g128 g256 b512 d2 d4 2>&1 | grep -v amdgpu
echo "FAT=1"; PCUDA_FAT=1 python scripts/conv_micro.py g32 g64 g128 g256 b512 d2 d4 2>&1 | grep -v amdgpu
