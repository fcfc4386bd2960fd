# upper bounds of weight delivery / staging / epilogue in igemm_pipe_kernel (library built with XFLAGS=-DPCUDA_WEXP BUILD=build_wexp
# OUT=../lib/libpcuda_wexp.so; DESIGN section 4 "Round 5").  Results of the switched-off runs are WRONG by construction: timings only.
cd "${GRAFT_REPO_ROOT:-.}"
L=pointcloududa_amd/lib/libpcuda_wexp.so
CASES="${CASES:-g32 g6432 g64 g128 g256 d2 d4}"
echo "== production"; python scripts/conv_micro.py $CASES 2>&1 | grep -v amdgpu
for d in ${DBGS:-0 24 32 64 56 120}; do
  echo "== wexp PCUDA_DBG=$d (8 no weight traffic, 16 no group barriers, 32 no input commit, 64 no epilogue)"
  PCUDA_LIB=$L PCUDA_DBG=$d python scripts/conv_micro.py $CASES 2>&1 | grep -v amdgpu
done
