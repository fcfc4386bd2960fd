// Fused loss / entropy kernels of the adversarial step (HBM-bound, one pass each way).
//   entropy map          train_mscmrseg.py:222,265 ; train_mmwhs.py:224,242
//   BCE + Jaccard        train_mscmrseg.py:202-203 ; utils/loss.py:5-37
//   double-softmax CE    train_mmwhs.py:212-214
//   domain BCE(const)    train_mscmrseg.py:224-226
//   batch_NN_loss        utils/loss.py:40-76
//   Dice metric          utils/utils.py:32-40 + utils/metric.py:5-36
#include "common.h"

#define MAXC 16
#define SMOOTHF 1e-7f

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------- entropy
// one thread per pixel, channels strided by hw (coalesced per channel)
__global__ __launch_bounds__(256) void entropy_fwd_kernel(const float* __restrict__ logits, int mode, float norm,
                                                          float* __restrict__ ent, float* __restrict__ prob, int c,
                                                          long long hw, long long npix) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += 256ll * gridDim.x) {
    const long long n = i / hw, px = i - n * hw;
    const float* p = logits + n * c * hw + px;
    float* e = ent + n * c * hw + px;
    float* q = prob ? prob + n * c * hw + px : nullptr;
    if (mode == PCUDA_ACT_SIGMOID) {
      for (int k = 0; k < c; ++k) {
        const float pr = sigmoidf_(p[k * hw]);
        e[k * hw] = -1.0f * pr * logf(pr + SMOOTHF) * norm;
        if (q) q[k * hw] = pr;
      }
    } else {
      float v[MAXC];
      float m = -INFINITY;
      for (int k = 0; k < c; ++k) { v[k] = p[k * hw]; m = fmaxf(m, v[k]); }
      float ssum = 0.f;
      for (int k = 0; k < c; ++k) { v[k] = expf(v[k] - m); ssum += v[k]; }
      for (int k = 0; k < c; ++k) {
        const float pr = v[k] / ssum;
        e[k * hw] = -1.0f * pr * logf(pr + SMOOTHF) * norm;
        if (q) q[k * hw] = pr;
      }
    }
  }
}

__global__ __launch_bounds__(256) void entropy_bwd_kernel(const float* __restrict__ logits, int mode, float norm,
                                                          const float* __restrict__ dent,
                                                          const float* __restrict__ dprob, float* __restrict__ dlogits,
                                                          int accumulate, int c, long long hw, long long npix,
                                                          const float* __restrict__ dmean, float mscale) {
  // dmean: gradient of mean_{n,pixels} sum_c ent (train_mmwhs.py:225,243), a device scalar: every element of the map
  // receives dmean * mscale (mscale = 1 / (n * hw)) on top of its own dent
  const float dm = dmean ? dmean[0] * mscale : 0.f;
  const bool has_e = dent != nullptr || dmean != nullptr;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += 256ll * gridDim.x) {
    const long long n = i / hw, px = i - n * hw;
    const long long base = n * c * hw + px;
    if (mode == PCUDA_ACT_SIGMOID) {
      for (int k = 0; k < c; ++k) {
        const float pr = sigmoidf_(logits[base + k * hw]);
        float gp = 0.f;
        if (has_e) gp += ((dent ? dent[base + k * hw] : 0.f) + dm) * norm * (-logf(pr + SMOOTHF) - pr / (pr + SMOOTHF));
        if (dprob) gp += dprob[base + k * hw];
        const float g = gp * pr * (1.f - pr);
        dlogits[base + k * hw] = accumulate ? dlogits[base + k * hw] + g : g;
      }
    } else {
      float v[MAXC], gp[MAXC];
      float m = -INFINITY;
      for (int k = 0; k < c; ++k) { v[k] = logits[base + k * hw]; m = fmaxf(m, v[k]); }
      float ssum = 0.f;
      for (int k = 0; k < c; ++k) { v[k] = expf(v[k] - m); ssum += v[k]; }
      float dot = 0.f;
      for (int k = 0; k < c; ++k) {
        const float pr = v[k] / ssum;
        v[k] = pr;
        float g = 0.f;
        if (has_e) g += ((dent ? dent[base + k * hw] : 0.f) + dm) * norm * (-logf(pr + SMOOTHF) - pr / (pr + SMOOTHF));
        if (dprob) g += dprob[base + k * hw];
        gp[k] = g;
        dot += g * pr;
      }
      for (int k = 0; k < c; ++k) {
        const float g = v[k] * (gp[k] - dot);
        dlogits[base + k * hw] = accumulate ? dlogits[base + k * hw] + g : g;
      }
    }
  }
}

// ---------------------------------------------------------------------------- segmentation loss
// workspace layout (doubles): [0] main sum, [1..C] I_c, [1+C..2C] S_c, then float partials
// partial row per block: [main, I_0..I_{C-1}, S_0..S_{C-1}]
#define SEG_BLOCKS 1024

template <int C>
__global__ __launch_bounds__(256) void seg_loss_partial_kernel(const float* __restrict__ logits,
                                                               const uint8_t* __restrict__ onehot, int mode,
                                                               long long hw, long long npix,
                                                               float* __restrict__ part) {
  constexpr int NACC = 2 * C + 1;
  __shared__ float sh[4][NACC];
  float acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += 256ll * gridDim.x) {
    const long long n = i / hw, px = i - n * hw;
    const long long base = n * C * hw + px;
    if (mode == PCUDA_ACT_SIGMOID) {
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const float pr = sigmoidf_(logits[base + k * hw]);
        const float y = (float)onehot[base + k * hw];
        // torch BCELoss clamps each log at -100
        const float l1 = fmaxf(logf(pr), -100.f), l0 = fmaxf(logf(1.f - pr), -100.f);
        acc[0] += -(y * l1 + (1.f - y) * l0);
        acc[1 + k] += pr * y;
        acc[1 + C + k] += pr + y;
      }
    } else {
      float v[C];
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < C; ++k) { v[k] = logits[base + k * hw]; m = fmaxf(m, v[k]); }
      float ssum = 0.f;
#pragma unroll
      for (int k = 0; k < C; ++k) { v[k] = expf(v[k] - m); ssum += v[k]; }
      uint8_t best = 0;
      float pm = -INFINITY, plabel = 0.f;
      bool first = true;
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const float pr = v[k] / ssum;
        v[k] = pr;
        const uint8_t yk = onehot[base + k * hw];
        if (first || yk > best) { best = yk; plabel = pr; first = false; }   // first maximum = np.argmax
        pm = fmaxf(pm, pr);
        acc[1 + k] += pr * (float)yk;
        acc[1 + C + k] += pr + (float)yk;
      }
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < C; ++k) s2 += expf(v[k] - pm);
      acc[0] += -(plabel - pm - logf(s2));   // -log_softmax(p)[label]
    }
  }
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    const float s = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NACC)
    part[(long long)blockIdx.x * NACC + threadIdx.x] =
        (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

__global__ void seg_loss_final_kernel(const float* __restrict__ part, int nblocks, int c, double numel_main,
                                      double* __restrict__ sums, float* __restrict__ out2) {
  __shared__ double sh[2 * MAXC + 1];
  __shared__ double red[256];
  const int nacc = 2 * c + 1;
  // fixed-order tree per accumulator (a single lane per accumulator walking thousands of partials took 110 us
  // between the segmenter's forward and backward passes)
  for (int a = 0; a < nacc; ++a) {
    double s = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)part[(long long)b * nacc + a];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) { sums[a] = red[0]; sh[a] = red[0]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = (float)(sh[0] / numel_main);
    double j = 0;
    for (int k = 0; k < c; ++k) {
      const double I = sh[1 + k], S = sh[1 + c + k];
      j += I / (S - I + 1e-7);
    }
    out2[1] = (float)(1.0 - j / c);
  }
}

__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ logits,
                                                           const uint8_t* __restrict__ onehot, int mode, int c,
                                                           long long hw, long long npix, double numel_main,
                                                           const double* __restrict__ sums,
                                                           const float* __restrict__ g_main,
                                                           const float* __restrict__ g_jac,
                                                           float* __restrict__ dlogits) {
  __shared__ float ju[MAXC], jiu[MAXC];   // 1/U_c and I_c/U_c^2
  if (threadIdx.x < c) {
    const double I = sums[1 + threadIdx.x], S = sums[1 + c + threadIdx.x];
    const double U = S - I + 1e-7;
    ju[threadIdx.x] = (float)(1.0 / U);
    jiu[threadIdx.x] = (float)(I / (U * U));
  }
  __syncthreads();
  const float gm = (g_main ? *g_main : 1.f) / (float)numel_main;
  const float gj = (g_jac ? *g_jac : 1.f) / (float)c;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += 256ll * gridDim.x) {
    const long long n = i / hw, px = i - n * hw;
    const long long base = n * c * hw + px;
    if (mode == PCUDA_ACT_SIGMOID) {
      for (int k = 0; k < c; ++k) {
        const float pr = sigmoidf_(logits[base + k * hw]);
        const float y = (float)onehot[base + k * hw];
        const float pq = pr * (1.f - pr);
        // torch: BCE backward divides by max(p(1-p), 1e-12), sigmoid backward multiplies by p(1-p)
        const float gb = gm * (pr - y) / fmaxf(pq, 1e-12f);
        const float gjac = -gj * (y * ju[k] - jiu[k] * (1.f - y));
        dlogits[base + k * hw] = (gb + gjac) * pq;
      }
    } else {
      float v[MAXC], gp[MAXC];
      float m = -INFINITY;
      for (int k = 0; k < c; ++k) { v[k] = logits[base + k * hw]; m = fmaxf(m, v[k]); }
      float ssum = 0.f;
      for (int k = 0; k < c; ++k) { v[k] = expf(v[k] - m); ssum += v[k]; }
      int label = 0;
      uint8_t best = 0;
      float pm = -INFINITY;
      for (int k = 0; k < c; ++k) {
        v[k] = v[k] / ssum;
        const uint8_t yk = onehot[base + k * hw];
        if (yk > best) { best = yk; label = k; }
        pm = fmaxf(pm, v[k]);
      }
      float s2 = 0.f;
      for (int k = 0; k < c; ++k) s2 += expf(v[k] - pm);
      float dot = 0.f;
      for (int k = 0; k < c; ++k) {
        const float y = (float)onehot[base + k * hw];
        const float sm2 = expf(v[k] - pm) / s2;
        const float g = gm * (sm2 - (k == label ? 1.f : 0.f)) - gj * (y * ju[k] - jiu[k] * (1.f - y));
        gp[k] = g;
        dot += g * v[k];
      }
      for (int k = 0; k < c; ++k) dlogits[base + k * hw] = v[k] * (gp[k] - dot);
    }
  }
}

// ---------------------------------------------------------------------------- domain BCE vs a constant
__global__ __launch_bounds__(256) void bce_const_fwd_kernel(const float* __restrict__ x, long long numel, float label,
                                                            float* loss, float* acc) {
  __shared__ float sh[2][4];
  float s = 0.f, hit = 0.f;
  for (long long i = threadIdx.x; i < numel; i += 256) {
    const float v = x[i];
    const float mx = fmaxf(-v, 0.f);   // ATen: (1-t)*x + max(-x,0) + log(exp(-max) + exp(-x-max))
    s += (1.f - label) * v + mx + logf(expf(-mx) + expf(-v - mx));
    hit += (sigmoidf_(v) >= 0.5f) ? 1.f : 0.f;
  }
  s = wave_sum(s);
  hit = wave_sum(hit);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = hit; }
  __syncthreads();
  if (threadIdx.x == 0) {
    *loss = ((sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3])) / (float)numel;
    if (acc) *acc = ((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3])) / (float)numel;
  }
}

__global__ __launch_bounds__(256) void bce_const_bwd_kernel(const float* __restrict__ x, long long numel, float label,
                                                            const float* gout, float gscale, float* __restrict__ dx) {
  const float g = (gout ? *gout : 1.f) * gscale / (float)numel;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < numel; i += 256ll * gridDim.x)
    dx[i] = (sigmoidf_(x[i]) - label) * g;
}

// ---------------------------------------------------------------------------- nearest-neighbour point loss
// One workgroup per (batch item, direction, block of 64 points); both clouds LDS-resident.  Four lanes share a point and
// search a quarter of the other cloud each (round 4: one workgroup per (item, direction) walked 2 x 300 candidates per lane
// in series, 82 us for 5.8 M distance evaluations); the quarters' minima combine with the serial loop's tie rule (the FIRST
// minimal candidate wins), so indices and values are the ones the serial search found.
#define NN_MAXP 1024
#define NN_PTS 64      // points per workgroup
__global__ __launch_bounds__(256) void nn_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          int npts, int* __restrict__ idx_ws,
                                                          float* __restrict__ val_ws, float* __restrict__ part) {
  __shared__ float sx[NN_MAXP * 3], sy[NN_MAXP * 3], rx[NN_MAXP], ry[NN_MAXP];
  const int b = blockIdx.x, nb = gridDim.x, dir = blockIdx.y, blk = blockIdx.z;
  const float* px = x + (long long)b * npts * 3;
  const float* py = y + (long long)b * npts * 3;
  for (int i = threadIdx.x; i < npts * 3; i += 256) { sx[i] = px[i]; sy[i] = py[i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < npts; i += 256) {
    rx[i] = sx[3 * i] * sx[3 * i] + sx[3 * i + 1] * sx[3 * i + 1] + sx[3 * i + 2] * sx[3 * i + 2];
    ry[i] = sy[3 * i] * sy[3 * i] + sy[3 * i + 1] * sy[3 * i + 1] + sy[3 * i + 2] * sy[3 * i + 2];
  }
  __syncthreads();
  // direction 1: for every x_i the nearest y_j ; direction 2: for every y_j the nearest x_i
  const float* A = dir == 0 ? sx : sy;
  const float* Bm = dir == 0 ? sy : sx;
  const float* ra = dir == 0 ? rx : ry;
  const float* rb = dir == 0 ? ry : rx;
  const int sub = threadIdx.x & 3, i = blk * NN_PTS + (threadIdx.x >> 2);
  const int ic = min(i, npts - 1);
  const int q = (npts + 3) >> 2, j0 = sub * q, j1 = min(npts, j0 + q);
  const float a0 = A[3 * ic], a1 = A[3 * ic + 1], a2 = A[3 * ic + 2], rai = ra[ic];
  float best = INFINITY;
  int bj = 0x7fffffff;
  for (int j = j0; j < j1; ++j) {
    const float zz = a0 * Bm[3 * j] + a1 * Bm[3 * j + 1] + a2 * Bm[3 * j + 2];
    const float P = rai + rb[j] - 2.f * zz;
    const float d = sqrtf(P + 0.00001f);
    if (d < best) { best = d; bj = j; }
  }
#pragma unroll
  for (int o = 1; o < 4; o <<= 1) {      // the four quarters: smaller distance, then smaller index
    const float ob = __shfl_xor(best, o, 64);
    const int oj = __shfl_xor(bj, o, 64);
    if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
  }
  // (every distance NaN -- vertices of a diverged step: no comparison above was true.  The reference's torch.min returns NaN
  // there; an index that pcuda_nn_loss_bwd can read with, and NaN as the minimum, so the loss is NaN like the reference's)
  if (bj == 0x7fffffff) { bj = 0; best = __builtin_nanf(""); }
  const bool own = sub == 0 && i < npts;
  if (own) {
    idx_ws[((long long)dir * nb + b) * npts + i] = bj;
    val_ws[((long long)dir * nb + b) * npts + i] = best;
  }
  // this block's sum of minima: the 64 points in index order within a wave (16 per wave), waves in order
  __shared__ float sh[4];
  float tot = own ? best : 0.f;
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) tot += __shfl_xor(tot, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) part[((long long)dir * nb + b) * gridDim.z + blk] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// loss = mean over items of (mean_i min_j + mean_j min_i); the blocks' partial sums in block order
__global__ void nn_loss_final_kernel(const float* __restrict__ part, int b, int nblk, int npts, float* loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < b; ++i) {
      float s0 = 0.f, s1 = 0.f;
      for (int k = 0; k < nblk; ++k) { s0 += part[(long long)i * nblk + k]; s1 += part[((long long)b + i) * nblk + k]; }
      s += s0 / (float)npts + s1 / (float)npts;
    }
    *loss = s / (float)b;
  }
}

__global__ __launch_bounds__(256) void nn_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          int npts, const int* __restrict__ idx_ws,
                                                          const float* __restrict__ val_ws, const float* gout,
                                                          float* __restrict__ dx) {
  const int b = blockIdx.x, nb = gridDim.x;
  const float* px = x + (long long)b * npts * 3;
  const float* py = y + (long long)b * npts * 3;
  const int* i1 = idx_ws + ((long long)0 * nb + b) * npts;
  const int* i2 = idx_ws + ((long long)1 * nb + b) * npts;
  const float* v1 = val_ws + ((long long)0 * nb + b) * npts;
  const float* v2 = val_ws + ((long long)1 * nb + b) * npts;
  const float g = (gout ? *gout : 1.f) / ((float)nb * (float)npts);
  for (int i = threadIdx.x; i < npts; i += 256) {
    const float x0 = px[3 * i], x1 = px[3 * i + 1], x2 = px[3 * i + 2];
    // own nearest neighbour in y
    const int j = i1[i];
    float s = g / v1[i];
    float d0 = s * (x0 - py[3 * j]), d1 = s * (x1 - py[3 * j + 1]), d2 = s * (x2 - py[3 * j + 2]);
    // every y_j whose nearest x is this point (fixed scan order: deterministic)
    for (int jj = 0; jj < npts; ++jj) {
      if (i2[jj] == i) {
        s = g / v2[jj];
        d0 += s * (x0 - py[3 * jj]); d1 += s * (x1 - py[3 * jj + 1]); d2 += s * (x2 - py[3 * jj + 2]);
      }
    }
    float* o = dx + ((long long)b * npts + i) * 3;
    o[0] = d0; o[1] = d1; o[2] = d2;
  }
}

// ---------------------------------------------------------------------------- Dice metric on device
__global__ __launch_bounds__(256) void dice_count_kernel(const float* __restrict__ logits,
                                                         const uint8_t* __restrict__ onehot, int c, long long hw,
                                                         long long npix, unsigned long long* __restrict__ cnt) {
  __shared__ unsigned int sh[3 * MAXC];
  for (int k = threadIdx.x; k < 3 * c; k += 256) sh[k] = 0;
  __syncthreads();
  unsigned int li[MAXC], ly[MAXC], lh[MAXC];
  for (int k = 0; k < c; ++k) li[k] = ly[k] = lh[k] = 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += 256ll * gridDim.x) {
    const long long n = i / hw, px = i - n * hw;
    const long long base = n * c * hw + px;
    float m = -INFINITY;
    for (int k = 0; k < c; ++k) m = fmaxf(m, logits[base + k * hw]);
    for (int k = 1; k < c; ++k) {
      const unsigned int hard = logits[base + k * hw] == m ? 1u : 0u;
      const unsigned int yk = onehot[base + k * hw];
      li[k] += hard * yk; ly[k] += yk; lh[k] += hard;
    }
  }
  for (int k = 1; k < c; ++k) {
    atomicAdd(&sh[k], li[k]); atomicAdd(&sh[c + k], ly[k]); atomicAdd(&sh[2 * c + k], lh[k]);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 3 * c; k += 256)
    if (sh[k]) atomicAdd(&cnt[k], (unsigned long long)sh[k]);
}

__global__ void dice_final_kernel(const unsigned long long* __restrict__ cnt, int c, float* dice) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double d = 0;
    for (int k = 1; k < c; ++k)
      d += (2.0 * (double)cnt[k] + 1.0) / ((double)cnt[c + k] + (double)cnt[2 * c + k] + 1.0);
    *dice = (float)(d / (c - 1));
  }
}

// ============================================================================ host wrappers
namespace {
inline int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }
}

extern "C" int pcuda_entropy_fwd(const float* logits, int mode, float norm, float* ent, float* prob, int n, int c,
                                 long long hw, pcuda_stream_t s) {
  if (!logits || !ent || n <= 0 || c <= 0 || c > MAXC || hw <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "entropy_fwd: bad arguments");
  const long long npix = (long long)n * hw;
  ProfScope prof(PCUDA_FAM_POINTWISE, (prob ? 12.0 : 8.0) * npix * c, (hipStream_t)s);
  hipLaunchKernelGGL(entropy_fwd_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)s, logits, mode, norm, ent,
                     prob, c, hw, npix);
  PCUDA_CHECK_LAUNCH("entropy_fwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_entropy_bwd2(const float* logits, int mode, float norm, const float* dent, const float* dprob,
                                  const float* dmean, float* dlogits, int accumulate, int n, int c, long long hw,
                                  pcuda_stream_t s) {
  if (!logits || !dlogits || (!dent && !dprob && !dmean) || n <= 0 || c <= 0 || c > MAXC || hw <= 0)
    PCUDA_FAIL(PCUDA_E_BADARG, "entropy_bwd: bad arguments");
  const long long npix = (long long)n * hw;
  ProfScope prof(PCUDA_FAM_POINTWISE, 12.0 * npix * c, (hipStream_t)s);
  hipLaunchKernelGGL(entropy_bwd_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)s, logits, mode, norm, dent,
                     dprob, dlogits, accumulate, c, hw, npix, dmean, (float)(1.0 / (double)npix));
  PCUDA_CHECK_LAUNCH("entropy_bwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_entropy_bwd(const float* logits, int mode, float norm, const float* dent, const float* dprob,
                                 float* dlogits, int accumulate, int n, int c, long long hw, pcuda_stream_t s) {
  return pcuda_entropy_bwd2(logits, mode, norm, dent, dprob, nullptr, dlogits, accumulate, n, c, hw, s);
}

// out = scale * sum(x): two fixed-order stages (SUM_BLOCKS block partials in fp32, their sum in fp64): deterministic
#define SUM_BLOCKS 512
__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ x, long long numel,
                                                          float* __restrict__ part) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < numel; i += 256ll * gridDim.x) acc += x[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(64) void sum_final_kernel(const float* __restrict__ part, int nparts, double scale,
                                                       float* __restrict__ out) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) acc += (double)part[i];
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) *out = (float)(acc * scale);
}

extern "C" size_t pcuda_sum_all_workspace_size(void) { return SUM_BLOCKS * sizeof(float); }

extern "C" int pcuda_sum_all(const float* x, long long numel, double scale, float* out, void* workspace,
                             size_t workspace_bytes, pcuda_stream_t s) {
  if (!x || !out || numel <= 0 || !workspace || workspace_bytes < pcuda_sum_all_workspace_size())
    PCUDA_FAIL(PCUDA_E_BADARG, "sum_all: bad arguments");
  long long nb = (numel + 1023) / 1024;
  if (nb > SUM_BLOCKS) nb = SUM_BLOCKS;
  ProfScope prof(PCUDA_FAM_POINTWISE, 4.0 * numel, (hipStream_t)s);
  hipLaunchKernelGGL(sum_partial_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)s, x, numel, (float*)workspace);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (const float*)workspace, (int)nb, scale, out);
  PCUDA_CHECK_LAUNCH("sum_all");
  return PCUDA_OK;
}

extern "C" size_t pcuda_seg_loss_workspace_size(int n, int c, long long hw) {
  (void)n; (void)hw;
  return 64 * sizeof(double) + (size_t)SEG_BLOCKS * (2 * c + 1) * sizeof(float);
}

extern "C" int pcuda_seg_loss_fwd(const float* logits, const uint8_t* onehot, int mode, int n, int c, long long hw,
                                  float* out2, void* workspace, size_t workspace_bytes, pcuda_stream_t s) {
  if (!logits || !onehot || !out2 || n <= 0 || c <= 0 || c > MAXC || hw <= 0)
    PCUDA_FAIL(PCUDA_E_BADARG, "seg_loss_fwd: bad arguments");
  if (!workspace || workspace_bytes < pcuda_seg_loss_workspace_size(n, c, hw))
    PCUDA_FAIL(PCUDA_E_WORKSPACE, "seg_loss_fwd: workspace too small");
  const long long npix = (long long)n * hw;
  int blocks = grid_for(npix);
  if (blocks > SEG_BLOCKS) blocks = SEG_BLOCKS;
  double* sums = (double*)workspace;
  float* part = (float*)((char*)workspace + 64 * sizeof(double));
  const double numel_main = mode == PCUDA_ACT_SIGMOID ? (double)npix * c : (double)npix;
  ProfScope prof(PCUDA_FAM_POINTWISE, 5.0 * npix * c, (hipStream_t)s);
#define SEG_CASE(CC) case CC: hipLaunchKernelGGL(seg_loss_partial_kernel<CC>, dim3(blocks), dim3(256), 0, (hipStream_t)s, logits, onehot, mode, hw, npix, part); break;
  switch (c) {
    SEG_CASE(1) SEG_CASE(2) SEG_CASE(3) SEG_CASE(4) SEG_CASE(5) SEG_CASE(6) SEG_CASE(7) SEG_CASE(8)
    default: PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "seg_loss: at most 8 classes (got %d)", c);
  }
#undef SEG_CASE
  PCUDA_CHECK_LAUNCH("seg_loss_partial_kernel");
  hipLaunchKernelGGL(seg_loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, (const float*)part, blocks, c,
                     numel_main, sums, out2);
  PCUDA_CHECK_LAUNCH("seg_loss_final_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_seg_loss_bwd(const float* logits, const uint8_t* onehot, int mode, int n, int c, long long hw,
                                  const float* g_main, const float* g_jac, float* dlogits, const void* workspace,
                                  pcuda_stream_t s) {
  if (!logits || !onehot || !dlogits || !workspace || n <= 0 || c <= 0 || c > MAXC || hw <= 0)
    PCUDA_FAIL(PCUDA_E_BADARG, "seg_loss_bwd: bad arguments");
  const long long npix = (long long)n * hw;
  const double numel_main = mode == PCUDA_ACT_SIGMOID ? (double)npix * c : (double)npix;
  ProfScope prof(PCUDA_FAM_POINTWISE, 9.0 * npix * c, (hipStream_t)s);
  hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)s, logits, onehot, mode, c,
                     hw, npix, numel_main, (const double*)workspace, g_main, g_jac, dlogits);
  PCUDA_CHECK_LAUNCH("seg_loss_bwd_kernel");
  return PCUDA_OK;
}

// ---------------------------------------------------------------------------- Jaccard on given probabilities
// utils/loss.py:5-37 as a free-standing function (the fused seg_loss above is what the train step uses): per class
// I_c = sum p*y, S_c = sum (p + y) over (batch, pixels); loss = 1 - mean_c I_c / (S_c - I_c + eps).
// ws (doubles): [c] I, [c] S, then float partials [JAC_BLOCKS][c][2]
#define JAC_BLOCKS 256
__global__ __launch_bounds__(256) void jaccard_partial_kernel(const float* __restrict__ p, const float* __restrict__ tf,
                                                              const uint8_t* __restrict__ tu, int c, long long hw,
                                                              long long npix, float* __restrict__ part) {
  __shared__ float sh[4][2];
  const int k = blockIdx.y;
  float a0 = 0.f, a1 = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += 256ll * gridDim.x) {
    const long long n = i / hw, px = i - n * hw;
    const long long o = (n * c + k) * hw + px;
    const float pr = p[o], y = tf ? tf[o] : (float)tu[o];
    a0 += pr * y;
    a1 += pr + y;
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = a0; sh[threadIdx.x >> 6][1] = a1; }
  __syncthreads();
  if (threadIdx.x < 2)
    part[((long long)blockIdx.x * c + k) * 2 + threadIdx.x] =
        (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

__global__ void jaccard_final_kernel(const float* __restrict__ part, int nblocks, int c, float eps,
                                     double* __restrict__ sums, float* __restrict__ loss) {
  __shared__ double tot[2 * MAXC];
  if ((int)threadIdx.x < 2 * c) {
    const int k = threadIdx.x >> 1, which = threadIdx.x & 1;
    double s = 0;
    for (int b = 0; b < nblocks; ++b) s += (double)part[((long long)b * c + k) * 2 + which];   // fixed order
    tot[which * c + k] = s;
    sums[which * c + k] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double j = 0;
    for (int k = 0; k < c; ++k) j += tot[k] / (tot[c + k] - tot[k] + (double)eps);
    loss[0] = (float)(1.0 - j / c);
  }
}

// d loss / d p = -(1/C) * (y * (U + eps) - I * (1 - y)) / (U + eps)^2,   U = S - I
__global__ __launch_bounds__(256) void jaccard_bwd_kernel(const float* __restrict__ tf, const uint8_t* __restrict__ tu,
                                                          int c, long long hw, long long npix, float eps,
                                                          const double* __restrict__ sums,
                                                          const float* __restrict__ gout, float* __restrict__ dp) {
  const int k = blockIdx.y;
  const double I = sums[k], U = sums[c + k] - I + (double)eps;
  const float g = (gout ? *gout : 1.f) / (float)c;
  const float a = (float)(1.0 / U), b = (float)(I / (U * U));
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += 256ll * gridDim.x) {
    const long long n = i / hw, px = i - n * hw;
    const long long o = (n * c + k) * hw + px;
    const float y = tf ? tf[o] : (float)tu[o];
    dp[o] = -g * (y * a - (1.f - y) * b);
  }
}

extern "C" size_t pcuda_jaccard_workspace_size(int c) {
  return (size_t)2 * MAXC * sizeof(double) + (size_t)JAC_BLOCKS * (c > 0 ? c : 1) * 2 * sizeof(float);
}

extern "C" int pcuda_jaccard_fwd(const float* probs, const void* truth, int truth_is_u8, int n, int c, long long hw,
                                 float eps, float* loss, void* workspace, size_t workspace_bytes, pcuda_stream_t s) {
  if (!probs || !truth || !loss || n <= 0 || c <= 0 || c > MAXC || hw <= 0)
    PCUDA_FAIL(PCUDA_E_BADARG, "jaccard_fwd: bad arguments");
  if (!workspace || workspace_bytes < pcuda_jaccard_workspace_size(c))
    PCUDA_FAIL(PCUDA_E_WORKSPACE, "jaccard_fwd: workspace too small");
  const long long npix = (long long)n * hw;
  int blocks = grid_for(npix);
  if (blocks > JAC_BLOCKS) blocks = JAC_BLOCKS;
  double* sums = (double*)workspace;
  float* part = (float*)((char*)workspace + 2 * MAXC * sizeof(double));
  ProfScope prof(PCUDA_FAM_POINTWISE, 8.0 * npix * c, (hipStream_t)s);
  hipLaunchKernelGGL(jaccard_partial_kernel, dim3(blocks, c), dim3(256), 0, (hipStream_t)s, probs,
                     truth_is_u8 ? nullptr : (const float*)truth, truth_is_u8 ? (const uint8_t*)truth : nullptr, c, hw,
                     npix, part);
  PCUDA_CHECK_LAUNCH("jaccard_partial_kernel");
  hipLaunchKernelGGL(jaccard_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (const float*)part, blocks, c, eps, sums,
                     loss);
  PCUDA_CHECK_LAUNCH("jaccard_final_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_jaccard_bwd(const void* truth, int truth_is_u8, int n, int c, long long hw, float eps,
                                 const float* gout, float* dprobs, const void* workspace, pcuda_stream_t s) {
  if (!truth || !dprobs || !workspace || n <= 0 || c <= 0 || c > MAXC || hw <= 0)
    PCUDA_FAIL(PCUDA_E_BADARG, "jaccard_bwd: bad arguments");
  const long long npix = (long long)n * hw;
  int blocks = grid_for(npix);
  ProfScope prof(PCUDA_FAM_POINTWISE, 8.0 * npix * c, (hipStream_t)s);
  hipLaunchKernelGGL(jaccard_bwd_kernel, dim3(blocks, c), dim3(256), 0, (hipStream_t)s,
                     truth_is_u8 ? nullptr : (const float*)truth, truth_is_u8 ? (const uint8_t*)truth : nullptr, c, hw,
                     npix, eps, (const double*)workspace, gout, dprobs);
  PCUDA_CHECK_LAUNCH("jaccard_bwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_bce_const_fwd(const float* x, long long numel, float label, float* loss, float* acc,
                                   pcuda_stream_t s) {
  if (!x || !loss || numel <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "bce_const_fwd: bad arguments");
  hipLaunchKernelGGL(bce_const_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, x, numel, label, loss, acc);
  PCUDA_CHECK_LAUNCH("bce_const_fwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_bce_const_bwd(const float* x, long long numel, float label, const float* gout, float gscale,
                                   float* dx, pcuda_stream_t s) {
  if (!x || !dx || numel <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "bce_const_bwd: bad arguments");
  hipLaunchKernelGGL(bce_const_bwd_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)s, x, numel, label, gout,
                     gscale, dx);
  PCUDA_CHECK_LAUNCH("bce_const_bwd_kernel");
  return PCUDA_OK;
}

extern "C" size_t pcuda_nn_loss_workspace_floats(int b, int npts) {
  if (b <= 0 || npts <= 0) return 0;
  return (size_t)2 * b * npts + (size_t)2 * b * ((npts + NN_PTS - 1) / NN_PTS);
}

extern "C" int pcuda_nn_loss_fwd(const float* x, const float* y, int b, int npts, float* loss, int* idx_ws,
                                 float* val_ws, pcuda_stream_t s) {
  if (!x || !y || !loss || !idx_ws || !val_ws || b <= 0 || npts <= 0 || npts > NN_MAXP)
    PCUDA_FAIL(PCUDA_E_BADARG, "nn_loss_fwd: bad arguments (npts <= %d)", NN_MAXP);
  // the blocks' partial sums live behind the 2*b*npts value slots (caller sizes val_ws with pcuda_nn_loss_workspace_floats)
  float* part = val_ws + (size_t)2 * b * npts;
  const int nblk = (npts + NN_PTS - 1) / NN_PTS;
  hipLaunchKernelGGL(nn_loss_fwd_kernel, dim3(b, 2, nblk), dim3(256), 0, (hipStream_t)s, x, y, npts, idx_ws, val_ws, part);
  PCUDA_CHECK_LAUNCH("nn_loss_fwd_kernel");
  hipLaunchKernelGGL(nn_loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (const float*)part, b, nblk, npts, loss);
  PCUDA_CHECK_LAUNCH("nn_loss_final_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_nn_loss_bwd(const float* x, const float* y, int b, int npts, const int* idx_ws,
                                 const float* val_ws, const float* gout, float* dx, pcuda_stream_t s) {
  if (!x || !y || !dx || !idx_ws || !val_ws || b <= 0 || npts <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "nn_loss_bwd: bad arguments");
  hipLaunchKernelGGL(nn_loss_bwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)s, x, y, npts, idx_ws, val_ws, gout, dx);
  PCUDA_CHECK_LAUNCH("nn_loss_bwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_dice_metric(const float* logits, const uint8_t* onehot, int n, int c, long long hw, float* dice,
                                 void* workspace, size_t workspace_bytes, pcuda_stream_t s) {
  if (!logits || !onehot || !dice || n <= 0 || c < 2 || c > MAXC || hw <= 0)
    PCUDA_FAIL(PCUDA_E_BADARG, "dice_metric: bad arguments");
  if (!workspace || workspace_bytes < (size_t)3 * c * sizeof(unsigned long long))
    PCUDA_FAIL(PCUDA_E_WORKSPACE, "dice_metric: workspace too small");
  if (hipMemsetAsync(workspace, 0, (size_t)3 * c * sizeof(unsigned long long), (hipStream_t)s) != hipSuccess)
    PCUDA_FAIL(PCUDA_E_LAUNCH, "dice_metric: memset failed");
  const long long npix = (long long)n * hw;
  int blocks = grid_for(npix);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(dice_count_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, logits, onehot, c, hw, npix,
                     (unsigned long long*)workspace);
  PCUDA_CHECK_LAUNCH("dice_count_kernel");
  hipLaunchKernelGGL(dice_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (const unsigned long long*)workspace, c,
                     dice);
  PCUDA_CHECK_LAUNCH("dice_final_kernel");
  return PCUDA_OK;
}

// ------------------------------------------------------------------------------------------
// validation metrics (train_mscmrseg.py:85-92; metric.py:39-82): label maps and per-class Dice
// ------------------------------------------------------------------------------------------
// labels[n][i] = FIRST channel holding the per-pixel maximum (= np.argmax(soft_to_hard_pred(x), axis=-1))
template <typename T>
__global__ __launch_bounds__(256) void argmax_labels_kernel(const T* __restrict__ x, long long sn, long long sc, int c,
                                                            long long hw, long long npix, uint8_t* __restrict__ lab) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += 256ll * gridDim.x) {
    const long long n = i / hw, q = i - n * hw;
    const T* px = x + n * sn + q;
    float best = (float)px[0];
    int bi = 0;
    for (int k = 1; k < c; ++k) {
      const float v = (float)px[(long long)k * sc];
      if (v > best) { best = v; bi = k; }
    }
    lab[i] = (uint8_t)bi;
  }
}

// cnt[k][0..2] += (|pred==k & gt==k|, |pred==k|, |gt==k|), integer counts (order independent)
__global__ __launch_bounds__(256) void label_overlap_kernel(const uint8_t* __restrict__ pred,
                                                            const uint8_t* __restrict__ gt, long long numel, int c,
                                                            unsigned long long* __restrict__ cnt) {
  __shared__ unsigned int h[MAXC * 3];
  for (int i = threadIdx.x; i < c * 3; i += 256) h[i] = 0;
  __syncthreads();
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < numel; i += 256ll * gridDim.x) {
    const int a = pred[i], b = gt[i];
    if (a < c) atomicAdd(&h[a * 3 + 1], 1u);
    if (b < c) atomicAdd(&h[b * 3 + 2], 1u);
    if (a == b && a < c) atomicAdd(&h[a * 3 + 0], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c * 3; i += 256)
    if (h[i]) atomicAdd(&cnt[i], (unsigned long long)h[i]);
}
// medpy.metric.binary.dc: 2|A.B| / (|A| + |B|), 0 when both are empty
__global__ void label_dice_final_kernel(const unsigned long long* __restrict__ cnt, int c, float* __restrict__ dice) {
  const int k = threadIdx.x;
  if (k < c) {
    const double den = (double)cnt[k * 3 + 1] + (double)cnt[k * 3 + 2];
    dice[k] = den > 0.0 ? (float)(2.0 * (double)cnt[k * 3 + 0] / den) : 0.f;
  }
}

extern "C" int pcuda_argmax_labels(const void* x, int x_is_u8, long long sn, long long sc, int n, int c, long long hw,
                                   uint8_t* labels, pcuda_stream_t s) {
  if (!x || !labels || n <= 0 || c < 1 || c > 255 || hw <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "argmax_labels: bad arguments");
  const long long npix = (long long)n * hw;
  const int blocks = (int)(cdiv(npix, 256) > 4096 ? 4096 : cdiv(npix, 256));
  ProfScope prof(PCUDA_FAM_POINTWISE, (double)npix * ((x_is_u8 ? 1.0 : 4.0) * c + 1.0), (hipStream_t)s);
  if (x_is_u8)
    hipLaunchKernelGGL(argmax_labels_kernel<uint8_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const uint8_t*)x, sn,
                       sc, c, hw, npix, labels);
  else
    hipLaunchKernelGGL(argmax_labels_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)x, sn, sc,
                       c, hw, npix, labels);
  PCUDA_CHECK_LAUNCH("argmax_labels_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_label_dice(const uint8_t* pred, const uint8_t* gt, long long numel, int c, float* dice,
                                void* workspace, size_t workspace_bytes, pcuda_stream_t s) {
  if (!pred || !gt || !dice || numel <= 0 || c < 1 || c > MAXC) PCUDA_FAIL(PCUDA_E_BADARG, "label_dice: bad arguments");
  if (!workspace || workspace_bytes < (size_t)c * 3 * sizeof(unsigned long long))
    PCUDA_FAIL(PCUDA_E_WORKSPACE, "label_dice: workspace too small");
  if (hipMemsetAsync(workspace, 0, (size_t)c * 3 * sizeof(unsigned long long), (hipStream_t)s) != hipSuccess)
    PCUDA_FAIL(PCUDA_E_LAUNCH, "label_dice: memset failed");
  const int blocks = (int)(cdiv(numel, 256) > 2048 ? 2048 : cdiv(numel, 256));
  ProfScope prof(PCUDA_FAM_POINTWISE, 2.0 * (double)numel, (hipStream_t)s);
  hipLaunchKernelGGL(label_overlap_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, pred, gt, numel, c,
                     (unsigned long long*)workspace);
  PCUDA_CHECK_LAUNCH("label_overlap_kernel");
  hipLaunchKernelGGL(label_dice_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)s,
                     (const unsigned long long*)workspace, c, dice);
  PCUDA_CHECK_LAUNCH("label_dice_final_kernel");
  return PCUDA_OK;
}
