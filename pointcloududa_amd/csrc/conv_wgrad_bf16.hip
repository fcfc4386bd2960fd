// single-bf16 (throughput mode) instantiations of the weight-gradient kernels.
#include "conv_wgrad_impl.h"

int wgrad_dispatch_bf16(const WgradParams& p, int co_blks, bool clamp, int taps_max, int pf, int x_cap, size_t lds,
                        float* dbp, dim3 grid, hipStream_t s) {
  return wgrad_dispatch<false>(p, co_blks, clamp, taps_max, pf, x_cap, lds, dbp, grid, s);
}
