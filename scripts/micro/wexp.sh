# upper bounds of weight delivery / staging in igemm_pipe_kernel (library built with XFLAGS=-DPCUDA_WEXP; DESIGN section 4 "Round 5")
cd "${GRAFT_REPO_ROOT:-.}"
L=pointcloududa_amd/lib/libpcuda_wexp.so
CASES="g64 g12864 g128 g256 g6432 d2 d3 d4"
echo "== production"; python scripts/conv_micro.py $CASES 2>&1 | grep -v amdgpu
for d in 0 8 24 32 56; do
  echo "== wexp PCUDA_DBG=$d (8 no weight traffic, 16 no group barriers, 32 no input commit)"
  PCUDA_LIB=$L PCUDA_DBG=$d python scripts/conv_micro.py $CASES 2>&1 | grep -v amdgpu
done
