// Direct (vector-ALU) convolution kernels for the degenerate layers of the path, where a 32-deep MFMA chunk is
// mostly padding and the layer is bound by its one large tensor:
//   * the segmenter's first convolution, 1 -> 32 channels, 3x3 (unet.py:23; forward + weight gradient):
//     the MFMA kernels ran it at 6.5 TFLOP/s, 185 us, against ~45 us of HBM time for its 268 MB output;
//   * the 1x1 classifier, 32 -> 4 / 5 channels (unet.py:178; forward + data gradient).
// Dispatched inside pcuda_conv2d_forward / _dgrad / _wgrad (same ABI, same packed weights: the fp32 weight is
// rebuilt as hi + lo from the packed bf16 planes, i.e. exactly the value the MFMA path multiplies by);
// PCUDA_NODIRECT=1 sends these layers back to the implicit-GEMM kernels.  fp32 FMA throughout, fixed reduction
// order (deterministic).  Every thread owns 4 consecutive pixels of a row: 16-byte loads and stores.
#include <stdlib.h>

#include "conv_host.h"
#include "conv_igemm.h"

namespace {

__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

// total over the 64 lanes, in lane 63 (DPP only: four row steps, row_bcast:15, row_bcast:31)
__device__ __forceinline__ float wave_sum_l63(float v) {
  v = row_sum<16>(v);
  v += dpp_mov0<0x142, 0xA>(v);
  v += dpp_mov0<0x143, 0xC>(v);
  return v;
}

// PCUDA_NODIRECT=1 switches every direct kernel off; PCUDA_DIRECT_MASK selects them one by one (A/B and bisection):
// 1 first-layer forward, 2 first-layer wgrad, 4 classifier forward, 8 classifier dgrad, 16 / 32 discriminator first
// layer dgrad / wgrad, 64 discriminator last layer forward
bool direct_enabled(int bit) {
  static int mask = -1;
  if (mask < 0) {
    const char* e = getenv("PCUDA_NODIRECT");
    const char* m = getenv("PCUDA_DIRECT_MASK");
    mask = (e && atoi(e)) ? 0 : (m ? atoi(m) : 0x7f);
  }
  return (mask & bit) != 0;
}

bool aligned16(const void* p, long long sn, long long sc) { return (((uintptr_t)p) & 15) == 0 && (sn & 3) == 0 && (sc & 3) == 0; }

// ------------------------------------------------------------------------------------------
// 1 -> cout channels, 3x3, stride 1, pad 1
// ------------------------------------------------------------------------------------------
struct C1Params {
  const float* x; long long x_sn;
  float* y; long long y_sn, y_sc;
  const uint16_t* wpack; int rec;              // [co-tile][1 chunk][9 taps][CO_TILE][rec]; rec = 72: lo plane at +32 (bf16x3)
  int co_tile;
  const float* bias;
  float slope;
  float* stats;                                // [tile][cout][2] or NULL
  int n, h, w, cout;
};

template <bool STATS>
__global__ __launch_bounds__(256) void c1_fwd_kernel(const C1Params p) {
  __shared__ __attribute__((aligned(16))) float sw[64 * 12];   // [co][9 weights, bias, 2 pad]
  __shared__ float sred[4][64][2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < p.cout * 12; i += 256) {
    const int co = i / 12, k = i - co * 12;
    float v = 0.f;
    const uint16_t* wq = p.wpack;
    if (k < 9) {
      const long long idx = (((long long)(co / p.co_tile) * 9 + k) * p.co_tile + co % p.co_tile) * p.rec;
      wq = p.wpack + idx;
      v = bf16_bits_to_float(wq[0]);
      if (p.rec > IG_REC) v += bf16_bits_to_float(wq[32]);
    } else if (k == 9 && p.bias) {
      wq = (const uint16_t*)(p.bias + co);
      v = *(const float*)wq;
    }
    sw[i] = v;
    PCUDA_KEEP(wq);      // (VMEM address rule, common.h)
  }
  __syncthreads();
  const int n = blockIdx.y, nq = (p.h * p.w) >> 2;
  const int q = blockIdx.x * 256 + tid;
  const bool valid = q < nq;
  const int qq = min(q, nq - 1);
  const int y = (4 * qq) / p.w, x = 4 * qq - y * p.w;
  const float* xp = p.x + (long long)n * p.x_sn;
  float in[3][6];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int yy = y + r - 1;
    const bool rok = (unsigned)yy < (unsigned)p.h;
    const float* row = xp + (long long)min(max(yy, 0), p.h - 1) * p.w;
    const float* pm = row + x;
    const float* pl = row + max(x - 1, 0);
    const float* pr = row + min(x + 4, p.w - 1);
    const f32x4 m = *(const f32x4*)pm;
    const float l = *pl, rr = *pr;
    in[r][0] = (rok && x > 0) ? l : 0.f;
    in[r][1] = rok ? m[0] : 0.f; in[r][2] = rok ? m[1] : 0.f; in[r][3] = rok ? m[2] : 0.f; in[r][4] = rok ? m[3] : 0.f;
    in[r][5] = (rok && x + 4 < p.w) ? rr : 0.f;
    PCUDA_KEEP(pm); PCUDA_KEEP(pl); PCUDA_KEEP(pr);      // (VMEM address rule, common.h)
  }
  float* yp = p.y + (long long)n * p.y_sn + 4 * qq;
  for (int co = 0; co < p.cout; ++co) {
    const f32x4 w0 = *(const f32x4*)(sw + co * 12), w1 = *(const f32x4*)(sw + co * 12 + 4), w2 = *(const f32x4*)(sw + co * 12 + 8);
    const float wk[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0]};
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) a = fmaf(wk[k], in[k / 3][e + k % 3], a);
      a += w2[1];
      o[e] = a > 0.f ? a : a * p.slope;
    }
    if (valid) *(f32x4*)(yp + (long long)co * p.y_sc) = o;
    if (STATS) {
      float s1 = (o[0] + o[1]) + (o[2] + o[3]);
      float s2 = (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
      s1 = wave_sum_l63(valid ? s1 : 0.f);
      s2 = wave_sum_l63(valid ? s2 : 0.f);
      if (lane == 63) { sred[wv][co][0] = s1; sred[wv][co][1] = s2; }
    }
  }
  if (STATS) {
    __syncthreads();
    if (tid < p.cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) { s1 += sred[ww][tid][0]; s2 += sred[ww][tid][1]; }   // fixed order
      const long long tile = (long long)n * gridDim.x + blockIdx.x;
      p.stats[(tile * p.cout + tid) * 2 + 0] = s1;
      p.stats[(tile * p.cout + tid) * 2 + 1] = s2;
    }
  }
}

// weight gradient: every thread owns a 4x4 pixel patch (its 6x6 input patch stays in registers for all output
// channels); per channel 144 FMAs into the nine tap sums, one DPP reduction each; block partials + fixed-order reduce
struct C1WgParams {
  const float* x; long long x_sn;
  const float* dz; long long dz_sn, dz_sc;
  float* partial;                               // [blocks][cout * 9] tap sums, then [blocks][cout] bias sums
  long long nblocks;
  int n, h, w, cout;
};

__global__ __launch_bounds__(256) void c1_wgrad_kernel(const C1WgParams p) {
  __shared__ float sred[4][64 * 10];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = blockIdx.y, qw = p.w >> 2, items = qw * (p.h >> 2);
  const int item = blockIdx.x * 256 + tid;
  const bool valid = item < items;
  const int it = min(item, items - 1);
  const int rg = it / qw, x = 4 * (it - rg * qw), y0 = 4 * rg;
  const float* xp = p.x + (long long)n * p.x_sn;
  float in[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int yy = y0 + r - 1;
    const bool rok = valid && (unsigned)yy < (unsigned)p.h;
    const float* row = xp + (long long)min(max(yy, 0), p.h - 1) * p.w;
    const float* pm = row + x;
    const float* pl = row + max(x - 1, 0);
    const float* pr = row + min(x + 4, p.w - 1);
    const f32x4 m = *(const f32x4*)pm;
    const float l = *pl, rr = *pr;
    in[r][0] = (rok && x > 0) ? l : 0.f;
    in[r][1] = rok ? m[0] : 0.f; in[r][2] = rok ? m[1] : 0.f; in[r][3] = rok ? m[2] : 0.f; in[r][4] = rok ? m[3] : 0.f;
    in[r][5] = (rok && x + 4 < p.w) ? rr : 0.f;
    PCUDA_KEEP(pm); PCUDA_KEEP(pl); PCUDA_KEEP(pr);      // (VMEM address rule, common.h)
  }
  const float* zp = p.dz + (long long)n * p.dz_sn + (long long)y0 * p.w + x;
  // output channels are split over blockIdx.z (twice the workgroups: 512 of them left the chip at two per CU)
  const int co_lo = (int)((long long)p.cout * blockIdx.z / gridDim.z), co_hi = (int)((long long)p.cout * (blockIdx.z + 1) / gridDim.z);
  // (the next channel's four rows are requested before this channel's FMAs and reductions: loaded at the top of each
  // iteration, the loop ran at one memory round trip per channel)
  f32x4 zn[4];
  const float* zq[4];      // addresses of the rows in flight (VMEM address rule, common.h)
#pragma unroll
  for (int r = 0; r < 4; ++r) { zq[r] = zp + (long long)co_lo * p.dz_sc + (long long)r * p.w; zn[r] = *(const f32x4*)zq[r]; }
  for (int co = co_lo; co < co_hi; ++co) {
    f32x4 z[4];
    const float* zc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { z[r] = zn[r]; zc[r] = zq[r]; }
    const int con = min(co + 1, co_hi - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) { zq[r] = zp + (long long)con * p.dz_sc + (long long)r * p.w; zn[r] = *(const f32x4*)zq[r]; }
    float part[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) part[k] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float zz = z[r][e];   // (rows outside the item range carry in = 0 for every tap; db masks below)
#pragma unroll
        for (int k = 0; k < 9; ++k) part[k] = fmaf(zz, in[r + k / 3][e + k % 3], part[k]);
        part[9] += valid ? zz : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const float s = wave_sum_l63(part[k]);
      if (lane == 63) sred[wv][co * 10 + k] = s;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) PCUDA_KEEP(zc[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) PCUDA_KEEP(zq[r]);
  __syncthreads();
  const long long blk = (long long)n * gridDim.x + blockIdx.x;
  for (int i = co_lo * 10 + tid; i < co_hi * 10; i += 256) {
    const float s = ((sred[0][i] + sred[1][i]) + sred[2][i]) + sred[3][i];   // fixed order
    const int co = i / 10, k = i - co * 10;
    if (k < 9) p.partial[blk * p.cout * 9 + co * 9 + k] = s;
    else p.partial[p.nblocks * p.cout * 9 + blk * p.cout + co] = s;
  }
}

// ------------------------------------------------------------------------------------------
// 1x1, cin <= 64 -> cout <= 8 (the classifier)
// ------------------------------------------------------------------------------------------
struct PwParams {
  pcuda_src x;            // forward: input (lazy BatchNorm affine applied on load); dgrad: dz
  pcuda_dst y;            // forward: output; dgrad: dx
  const uint16_t* wpack; int rec;   // rec = 72: lo plane at +32 inside the record (bf16x3), 40: bf16 mode
  int row_tile;           // CO_TILE of the packed layout's row dimension
  const float* bias;
  float slope;
  int accumulate;
  int n, hw, cin, cout;
  // dgrad with the BatchNorm-backward reduce of the layer that consumes dx (see pcuda_conv2d_dgrad_bnred)
  const float* red_a; long long red_sn, red_sc;
  const float* red_mean; const float* red_invstd;
  float* red;              // [n * blocks per image][cin][2]
};

__device__ __forceinline__ const float* pw_src_plane(const pcuda_src& x, int n, int c) {
  return c < x.c1 ? x.p1 + (long long)n * x.sn1 + (long long)c * x.sc1 : x.p2 + (long long)n * x.sn2 + (long long)(c - x.c1) * x.sc2;
}
__device__ __forceinline__ float* pw_dst_plane(const pcuda_dst& y, int n, int c) {
  return c < y.c1 ? y.p1 + (long long)n * y.sn1 + (long long)c * y.sc1 : y.p2 + (long long)n * y.sn2 + (long long)(c - y.c1) * y.sc2;
}

// forward packed layout: rows = cout (one 32-row tile), reduction = cin: w[co][ci] at ((ci/32) * 32 + co) * rec + ci%32
template <int CO>
__global__ __launch_bounds__(256) void pw_fwd_kernel(const PwParams p) {
  __shared__ float sw[64 * CO], ssc[64], ssh[64], sb[CO];
  const int tid = threadIdx.x;
  for (int i = tid; i < p.cin * CO; i += 256) {
    const int ci = i / CO, co = i - ci * CO;
    float v = 0.f;
    const uint16_t* wq = p.wpack;
    if (co < p.cout) {
      const long long idx = ((long long)(ci >> 5) * p.row_tile + co) * p.rec + (ci & 31);
      wq = p.wpack + idx;
      v = bf16_bits_to_float(wq[0]);
      if (p.rec > IG_REC) v += bf16_bits_to_float(wq[32]);
    }
    sw[i] = v;
    PCUDA_KEEP(wq);      // (VMEM address rule, common.h)
  }
  for (int i = tid; i < p.cin; i += 256) {
    const bool first = i < p.x.c1;
    const float* scp = first ? p.x.scale1 : p.x.scale2;
    const float* shp = first ? p.x.shift1 : p.x.shift2;
    const int cc = first ? i : i - p.x.c1;
    ssc[i] = scp ? scp[cc] : 1.f;
    ssh[i] = scp ? shp[cc] : 0.f;
  }
  if (tid < CO) {
    const float* bq = p.bias ? p.bias + min(tid, p.cout - 1) : nullptr;
    sb[tid] = (bq && tid < p.cout) ? *bq : 0.f;
    PCUDA_KEEP(bq);
  }
  __syncthreads();
  const int n = blockIdx.y, nq = p.hw >> 2;
  const int q = blockIdx.x * 256 + tid;
  if (q >= nq) return;
  f32x4 acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int ci = 0; ci < p.cin; ++ci) {
    const float* src = pw_src_plane(p.x, n, ci) + 4 * q;
    f32x4 a = *(const f32x4*)src;
    const float sc = ssc[ci], sh = ssh[ci];
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = fmaf(a[e], sc, sh);
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      const float w = sw[ci * CO + co];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[co][e] = fmaf(w, a[e], acc[co][e]);
    }
    PCUDA_KEEP(src);      // (VMEM address rule, common.h)
  }
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    if (co < p.cout) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = acc[co][e] + sb[co];
        o[e] = t > 0.f ? t : t * p.slope;
      }
      *(f32x4*)(pw_dst_plane(p.y, n, co) + 4 * q) = o;
    }
  }
}

// dgrad packed layout: rows = cin (tiles of row_tile), reduction = cout: w[co][ci] at ((ci/row_tile) * row_tile + ci%row_tile) * rec + co
template <int CO, bool RED>
__global__ __launch_bounds__(256) void pw_dgrad_kernel(const PwParams p) {
  __shared__ float sw[64 * CO];
  __shared__ float sred[4][64][2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < p.cin * CO; i += 256) {
    const int ci = i / CO, co = i - ci * CO;
    float v = 0.f;
    const uint16_t* wq = p.wpack;
    if (co < p.cout) {
      const long long idx = (long long)ci * p.rec + co;
      wq = p.wpack + idx;
      v = bf16_bits_to_float(wq[0]);
      if (p.rec > IG_REC) v += bf16_bits_to_float(wq[32]);
    }
    sw[i] = v;
    PCUDA_KEEP(wq);      // (VMEM address rule, common.h)
  }
  __syncthreads();
  const int n = blockIdx.y, nq = p.hw >> 2;
  const int q0 = blockIdx.x * 256 + tid;
  const bool valid = q0 < nq;
  if (!RED && !valid) return;
  const int q = min(q0, nq - 1);
  f32x4 z[CO];
  const float* zsrc[CO];      // (VMEM address rule, common.h: kept until the loop below has used the data)
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    z[co] = f32x4{0.f, 0.f, 0.f, 0.f};
    zsrc[co] = pw_src_plane(p.x, n, __builtin_amdgcn_readfirstlane(min(co, p.cout - 1))) + 4 * q;
    if (co < p.cout) z[co] = *(const f32x4*)zsrc[co];
  }
#pragma unroll 4
  for (int ci = 0; ci < p.cin; ++ci) {
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      const float w = sw[ci * CO + co];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaf(w, z[co][e], o[e]);
    }
    float* d = pw_dst_plane(p.y, n, ci) + 4 * q;
    if (p.accumulate) {
      const f32x4 old = *(const f32x4*)d;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += old[e];
    }
    if (valid) *(f32x4*)d = o;      // (d stays live up to its store: the accumulate load's address is never reused early)
    if (RED) {
      const float* pav = p.red_a + (long long)n * p.red_sn + (long long)ci * p.red_sc + 4 * q;
      const f32x4 av = *(const f32x4*)pav;
      const float* pmean = p.red_mean + ci;
      const float* pinv = p.red_invstd + ci;
      const float m = *pmean, is = *pinv;
      float s1 = (o[0] + o[1]) + (o[2] + o[3]);
      float s2 = (o[0] * ((av[0] - m) * is) + o[1] * ((av[1] - m) * is)) + (o[2] * ((av[2] - m) * is) + o[3] * ((av[3] - m) * is));
      s1 = wave_sum_l63(valid ? s1 : 0.f);
      s2 = wave_sum_l63(valid ? s2 : 0.f);
      if (lane == 63) { sred[wv][ci][0] = s1; sred[wv][ci][1] = s2; }
      PCUDA_KEEP(pav); PCUDA_KEEP(pmean); PCUDA_KEEP(pinv);
    }
  }
#pragma unroll
  for (int co = 0; co < CO; ++co) PCUDA_KEEP(zsrc[co]);
  if (RED) {
    __syncthreads();
    if (tid < p.cin) {
      const long long tile = (long long)n * gridDim.x + blockIdx.x;
      p.red[(tile * p.cin + tid) * 2 + 0] = ((sred[0][tid][0] + sred[1][tid][0]) + sred[2][tid][0]) + sred[3][tid][0];
      p.red[(tile * p.cin + tid) * 2 + 1] = ((sred[0][tid][1] + sred[1][tid][1]) + sred[2][tid][1]) + sred[3][tid][1];
    }
  }
}

bool c1_geom(const pcuda_conv_geom* g) {
  return g->cin == 1 && g->k == 3 && g->stride == 1 && g->pad == 1 && g->dil == 1 && !g->in_up && g->cout <= 64 &&
         (g->in_w & 3) == 0 && g->in_w >= 8;
}
bool pw_geom(const pcuda_conv_geom* g) {
  return g->k == 1 && g->stride == 1 && g->pad == 0 && !g->in_up && g->cout <= 8 && g->cin <= 64 &&
         ((g->in_h * g->in_w) & 3) == 0;
}

}  // namespace

// ---- entry points used by conv_igemm.hip / conv_wgrad.hip (return 1 when the direct kernel took the launch) ----
int direct_fwd_tiles(const pcuda_conv_geom* g) {
  if (!direct_enabled(1) || !c1_geom(g)) return 0;
  return g->n * cdiv((long long)g->in_h * g->in_w / 4, 256);
}

int direct_forward(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const void* packed_w, long long w_lo_off,
                   const float* bias, float slope, const pcuda_dst* y, float* bn_partials, hipStream_t s, int* rc) {
  *rc = PCUDA_OK;
  const double flops = 2.0 * g->n * (double)g->out_h * g->out_w * g->cout * (double)g->cin * g->k * g->k;
  if (direct_enabled(1) && c1_geom(g)) {
    // (the BatchNorm partials are sized by pcuda_conv2d_fwd_tiles, which answers for this kernel whenever the
    // geometry qualifies: a tensor that fails the checks below must not fall back silently to another tile count)
    if (x->scale1 || y->c1 < g->cout || !aligned16(x->p1, x->sn1, 4) || !aligned16(y->p1, y->sn1, y->sc1)) {
      if (bn_partials) { pcuda_set_error("conv2d_forward: 1-channel 3x3 layer needs 16-byte aligned, unsplit tensors"); *rc = PCUDA_E_UNSUPPORTED; return 1; }
      return 0;
    }
    C1Params p;
    p.x = x->p1; p.x_sn = x->sn1;
    p.y = y->p1; p.y_sn = y->sn1; p.y_sc = y->sc1;
    p.wpack = (const uint16_t*)packed_w; p.rec = ig_rec_bytes(prec == PCUDA_PREC_BF16X3) / 2;
    p.co_tile = 32 * ig_co_blks(g->cout);
    p.bias = bias; p.slope = slope; p.stats = bn_partials;
    p.n = g->n; p.h = g->in_h; p.w = g->in_w; p.cout = g->cout;
    char tag[96];
    snprintf(tag, sizeof(tag), "direct c1 fwd n%d cout%d %dx%d", g->n, g->cout, g->in_h, g->in_w);
    ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
    const dim3 grid(cdiv((long long)g->in_h * g->in_w / 4, 256), g->n);
    if (bn_partials) hipLaunchKernelGGL(c1_fwd_kernel<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(c1_fwd_kernel<false>, grid, dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pcuda_set_error("c1_fwd_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; }
    return 1;
  }
  if (direct_enabled(4) && pw_geom(g) && !bn_partials) {
    const bool two = x->c1 < g->cin, twoy = y->c1 < g->cout;
    if (!aligned16(x->p1, x->sn1, x->sc1) || (two && !aligned16(x->p2, x->sn2, x->sc2)) ||
        !aligned16(y->p1, y->sn1, y->sc1) || (twoy && !aligned16(y->p2, y->sn2, y->sc2)))
      return 0;
    PwParams p;
    p.x = *x; p.y = *y;
    p.wpack = (const uint16_t*)packed_w; p.rec = ig_rec_bytes(prec == PCUDA_PREC_BF16X3) / 2;
    p.row_tile = 32 * ig_co_blks(g->cout);
    p.bias = bias; p.slope = slope; p.accumulate = 0;
    p.n = g->n; p.hw = g->in_h * g->in_w; p.cin = g->cin; p.cout = g->cout;
    char tag[96];
    snprintf(tag, sizeof(tag), "direct 1x1 fwd n%d cin%d cout%d %dx%d", g->n, g->cin, g->cout, g->in_h, g->in_w);
    ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
    const dim3 grid(cdiv(p.hw / 4, 256), g->n);
    if (g->cout <= 4) hipLaunchKernelGGL(pw_fwd_kernel<4>, grid, dim3(256), 0, s, p);
    else if (g->cout <= 5) hipLaunchKernelGGL(pw_fwd_kernel<5>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(pw_fwd_kernel<8>, grid, dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pcuda_set_error("pw_fwd_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; }
    return 1;
  }
  return 0;
}

int direct_dgrad_tiles(const pcuda_conv_geom* g) {
  if (!direct_enabled(8) || !pw_geom(g)) return 0;
  return g->n * cdiv((long long)g->in_h * g->in_w / 4, 256);
}

// red != NULL: the BatchNorm-backward reduce of the layer that consumes dx rides along (a, mean, invstd of that layer)
int direct_dgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad, const pcuda_dst* dx,
                 int accumulate, hipStream_t s, int* rc, const float* red_a, long long red_sn, long long red_sc,
                 const float* red_mean, const float* red_invstd, float* red) {
  *rc = PCUDA_OK;
  if (!direct_enabled(8) || !pw_geom(g) || dy->scale1 || dy->c1 < g->cout) return 0;
  const bool twoy = dx->c1 < g->cin;
  if (!aligned16(dy->p1, dy->sn1, dy->sc1) || !aligned16(dx->p1, dx->sn1, dx->sc1) || (twoy && !aligned16(dx->p2, dx->sn2, dx->sc2)))
    return 0;
  if (red && !aligned16(red_a, red_sn, red_sc)) return 0;
  PwParams p;
  p.x = *dy; p.y = *dx;
  p.wpack = (const uint16_t*)packed_w_dgrad;
  // dgrad layout of a 1x1 stride-1 layer: one parity class, one tap; rows = cin in tiles of 32 / 64, reduction = cout <= 8
  const int rt = 32 * ig_co_blks(g->cin);
  p.rec = ig_rec_bytes(prec == PCUDA_PREC_BF16X3) / 2;
  p.row_tile = rt;
  p.bias = nullptr; p.slope = 1.f; p.accumulate = accumulate;
  p.n = g->n; p.hw = g->in_h * g->in_w; p.cin = g->cin; p.cout = g->cout;
  p.red_a = red_a; p.red_sn = red_sn; p.red_sc = red_sc; p.red_mean = red_mean; p.red_invstd = red_invstd; p.red = red;
  const double flops = 2.0 * g->n * (double)g->out_h * g->out_w * g->cout * (double)g->cin;
  char tag[96];
  snprintf(tag, sizeof(tag), "direct 1x1 dgrad%s n%d cin%d cout%d %dx%d", red ? "+bnred" : "", g->n, g->cin, g->cout, g->in_h, g->in_w);
  ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
  const dim3 grid(cdiv(p.hw / 4, 256), g->n);
#define PW_DG(CO_)                                                                         \
  do {                                                                                     \
    if (red) hipLaunchKernelGGL((pw_dgrad_kernel<CO_, true>), grid, dim3(256), 0, s, p);   \
    else hipLaunchKernelGGL((pw_dgrad_kernel<CO_, false>), grid, dim3(256), 0, s, p);      \
  } while (0)
  if (g->cout <= 4) PW_DG(4);
  else if (g->cout <= 5) PW_DG(5);
  else PW_DG(8);
#undef PW_DG
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { pcuda_set_error("pw_dgrad_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; }
  return 1;
}

size_t direct_wgrad_workspace(const pcuda_conv_geom* g) {
  if (!c1_geom(g) || (g->in_h & 3)) return 0;
  const long long blocks = (long long)g->n * cdiv((long long)(g->in_w / 4) * (g->in_h / 4), 256);
  return (size_t)blocks * g->cout * 10 * sizeof(float) + 256;
}

int direct_wgrad(const pcuda_conv_geom* g, const pcuda_src* x, const float* dz, long long dz_sn, long long dz_sc, float* dw,
                 float* db, int accumulate, void* workspace, hipStream_t s, int* rc) {
  *rc = PCUDA_OK;
  if (!direct_enabled(2) || !c1_geom(g) || (g->in_h & 3) || x->scale1) return 0;
  if (!aligned16(x->p1, x->sn1, 4) || !aligned16(dz, dz_sn, dz_sc)) return 0;
  C1WgParams p;
  p.x = x->p1; p.x_sn = x->sn1;
  p.dz = dz; p.dz_sn = dz_sn; p.dz_sc = dz_sc;
  p.partial = (float*)workspace;
  p.n = g->n; p.h = g->in_h; p.w = g->in_w; p.cout = g->cout;
  const int bpi = cdiv((long long)(g->in_w / 4) * (g->in_h / 4), 256);
  const double flops = 2.0 * g->n * (double)g->out_h * g->out_w * g->cout * 9.0;
  char tag[96];
  snprintf(tag, sizeof(tag), "direct c1 wgrad n%d cout%d %dx%d", g->n, g->cout, g->in_h, g->in_w);
  ProfScope prof(PCUDA_FAM_CONV_WGRAD, flops, s, tag);
  p.nblocks = (long long)bpi * g->n;
  hipLaunchKernelGGL(c1_wgrad_kernel, dim3(bpi, g->n, g->cout >= 8 ? 2 : 1), dim3(256), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { pcuda_set_error("c1_wgrad_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; return 1; }
  // block partials through the split-K reduce of the MFMA weight-gradient kernels (16 waves per 64 outputs, fixed order)
  const float* pw = (const float*)workspace;
  *rc = launch_wgrad_reduce(pw, (long long)g->cout * 9, (int)p.nblocks, dw, accumulate, pw + p.nblocks * g->cout * 9, g->cout, db, s);
  return 1;
}
