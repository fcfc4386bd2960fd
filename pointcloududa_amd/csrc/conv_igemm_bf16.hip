// single-bf16 (throughput mode) instantiations of the forward / dgrad kernels.
#include "conv_igemm_impl.h"

int igemm_dispatch_bf16(const IgemmParams& p, const IgemmPlan& pl, int co_blks, int pf, bool pipe, hipStream_t s) {
  return igemm_dispatch<false>(p, pl, co_blks, pf, pipe, s);
}
