// HBM-bound kernels around the convolutions: BatchNorm (train) statistics / apply / backward fused
// with the LeakyReLU that precedes it in the reference (unet.py:23-30: conv -> LeakyReLU -> BN),
// 2x2 max pooling (unet.py:48), nearest-upsample backward (unet.py:111), running sums.
// Tensors are [n][c][hw] with dense planes and explicit batch / channel strides.
#include "common.h"

#ifndef PCH
#define PCH 2048   // plane elements per workgroup (256 threads x 8)
#endif

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];   // blockDim = 256
}


// 16-B vector path: used when hw % 4 == 0 and every plane base / stride is 16-B aligned
template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<4> { typedef float4 T; };
template <int VEC> __device__ __forceinline__ void vload(const float* p, float (&v)[VEC]);
template <> __device__ __forceinline__ void vload<1>(const float* p, float (&v)[1]) { v[0] = *p; }
template <> __device__ __forceinline__ void vload<4>(const float* p, float (&v)[4]) {
  const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <int VEC> __device__ __forceinline__ void vstore(float* p, const float (&v)[VEC]);
template <> __device__ __forceinline__ void vstore<1>(float* p, const float (&v)[1]) { *p = v[0]; }
template <> __device__ __forceinline__ void vstore<4>(float* p, const float (&v)[4]) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
}
static inline bool vec_ok(const void* p, long long sn, long long sc, long long hw) {
  return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (sn & 3) == 0 && (sc & 3) == 0 && (hw & 3) == 0);
}

// ---------------------------------------------------------------------------- statistics
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ a, long long sn, long long sc,
                                                       long long hw, int c, float* __restrict__ partials) {
  __shared__ float sh[4];
  const int ch = blockIdx.y, n = blockIdx.z;
  const float* pl = a + n * sn + ch * sc;
  const long long i0 = (long long)blockIdx.x * PCH, i1 = min(hw, i0 + PCH);
  float s1 = 0.f, s2 = 0.f;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float v = pl[i];
    s1 += v;
    s2 += v * v;
  }
  s1 = block_sum(s1, sh);
  s2 = block_sum(s2, sh);
  if (threadIdx.x == 0) {
    const long long tile = (long long)n * gridDim.x + blockIdx.x;
    partials[(tile * c + ch) * 2 + 0] = s1;
    partials[(tile * c + ch) * 2 + 1] = s2;
  }
}

// One thread's share of a channel's per-tile partial pairs [tile][c][2], tiles t, t + 256, ... in that order (the order
// is part of the result).  Eight 8-byte loads in flight per lane: one load per loop trip made the 8192-tile layers'
// finalize kernels a chain of memory round trips (33 us for 2 MB).
__device__ __forceinline__ void tile_pair_sum(const float* __restrict__ part, int ntiles, int c, int ch, double& s1, double& s2) {
  const float2* p2 = (const float2*)part;
  s1 = 0; s2 = 0;
  int t = threadIdx.x;
  for (; t + 7 * 256 < ntiles; t += 8 * 256) {
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p2[(long long)(t + j * 256) * c + ch];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1 += (double)v[j].x; s2 += (double)v[j].y; }
  }
  for (; t < ntiles; t += 256) {
    const float2 v = p2[(long long)t * c + ch];
    s1 += (double)v.x; s2 += (double)v.y;
  }
}

// one workgroup per channel; fp64 finalisation in a fixed order (deterministic)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials, int ntiles, int c,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum,
                                                          float* running_mean, float* running_var, float* mean,
                                                          float* invstd, float* scale, float* shift) {
  __shared__ double sh[2][256];
  const int ch = blockIdx.x;
  double s1, s2;
  tile_pair_sum(partials, ntiles, c, ch, s1, s2);
  sh[0][threadIdx.x] = s1;
  sh[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double m = sh[0][0] / count;
    double var = sh[1][0] / count - m * m;
    if (var < 0) var = 0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
    mean[ch] = (float)m;
    invstd[ch] = is;
    const float sc = g * is;
    scale[ch] = sc;
    shift[ch] = b - (float)m * sc;
    if (running_mean) {
      const double unb = count > 1 ? var * count / (count - 1.0) : var;
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unb;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ a, long long a_sn, long long a_sc,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int relu, float* __restrict__ y, long long y_sn, long long y_sc,
                                                       long long hw) {
  const int ch = blockIdx.y, n = blockIdx.z;
  const float* pa = a + n * a_sn + ch * a_sc;
  float* py = y + n * y_sn + ch * y_sc;
  const float sc = scale[ch], sf = shift[ch];
  const long long i0 = (long long)blockIdx.x * PCH, i1 = min(hw, i0 + PCH);
  for (long long i = i0 + threadIdx.x * VEC; i < i1; i += 256 * VEC) {
    float v[VEC];
    vload<VEC>(pa + i, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      v[e] = v[e] * sc + sf;
      if (relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    vstore<VEC>(py + i, v);
  }
}

// ---------------------------------------------------------------------------- backward
// "Pooled" gradient source: the incoming gradient (or a share of it) is the scatter of a 2x2 max-pool's backward -- g (+ g2)
// at the pooled resolution and the argmax index the forward kept (unet.py:48) -- read IN PLACE: element (y, x) of the plane
// gets g[y/2][x/2] if idx[y/2][x/2] == 2 (y & 1) + (x & 1), else 0.  The kernels below then never see the 4x larger, three
// quarters zero tensor that maxpool2_bwd_kernel wrote and they read back (round 4).
struct PoolSrc {
  const float* g; long long sn, sc;
  const float* g2; long long sn2, sc2;
  const uint8_t* idx;
  int w, c;      // full-resolution row length (even; a multiple of 4 on the vector path); channels
  long long ohw; // pooled plane size (idx is dense [n][c][h/2][w/2])
};
template <int VEC>
__device__ __forceinline__ void pool_load(const PoolSrc& ps, int n, int ch, long long i, float (&out)[VEC]) {
  const int w = ps.w, ow = w >> 1;
  const int y = (int)(i / w), x = (int)(i - (long long)y * w);
  const long long o = (long long)(y >> 1) * ow + (x >> 1);
  const float* pg = ps.g + n * ps.sn + ch * ps.sc + o;
  const float* pg2 = ps.g2 ? ps.g2 + n * ps.sn2 + ch * ps.sc2 + o : nullptr;
  const uint8_t* pi = ps.idx + ((long long)n * ps.c + ch) * ps.ohw + o;
  const int krow = (y & 1) * 2;
  if (VEC == 4) {
    float2 gv = *(const float2*)pg;
    if (pg2) { const float2 t = *(const float2*)pg2; gv.x += t.x; gv.y += t.y; }
    const unsigned short kk = *(const unsigned short*)pi;
    const int k0 = kk & 0xff, k1 = kk >> 8;
    out[0] = k0 == krow ? gv.x : 0.f;
    out[1] = k0 == krow + 1 ? gv.x : 0.f;
    out[2] = k1 == krow ? gv.y : 0.f;
    out[3] = k1 == krow + 1 ? gv.y : 0.f;
  } else {
    float gv = *pg;
    if (pg2) gv += *pg2;
    out[0] = (int)*pi == krow + (x & 1) ? gv : 0.f;
  }
  PCUDA_KEEP(pg); PCUDA_KEEP(pg2); PCUDA_KEEP(pi);      // (VMEM address rule, common.h)
}

template <int VEC, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ dy, long long dy_sn, long long dy_sc, const float* __restrict__ dy2, long long dy2_sn,
    long long dy2_sc, const float* __restrict__ a, long long a_sn, long long a_sc, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift, int post_relu,
    long long hw, int c, float* __restrict__ red, PoolSrc ps) {
  __shared__ float sh[4];
  const int ch = blockIdx.y, n = blockIdx.z;
  const float* pd = dy ? dy + n * dy_sn + ch * dy_sc : nullptr;     // (POOL: the full-resolution share is optional)
  const float* pd2 = dy2 ? dy2 + n * dy2_sn + ch * dy2_sc : nullptr;
  const float* pa = a + n * a_sn + ch * a_sc;
  const float m = mean[ch], is = invstd[ch];
  const float sc = post_relu ? scale[ch] : 0.f, sf = post_relu ? shift[ch] : 0.f;
  const long long i0 = (long long)blockIdx.x * PCH, i1 = min(hw, i0 + PCH);
  float s1 = 0.f, s2 = 0.f;
  for (long long i = i0 + threadIdx.x * VEC; i < i1; i += 256 * VEC) {
    float av[VEC], g[VEC], g2[VEC];
    vload<VEC>(pa + i, av);
    if (POOL) {
      pool_load<VEC>(ps, n, ch, i, g);
      if (pd) {
        vload<VEC>(pd + i, g2);
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] += g2[e];
      }
    } else {
      vload<VEC>(pd + i, g);
    }
    if (pd2) vload<VEC>(pd2 + i, g2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float gg = pd2 ? g[e] + g2[e] : g[e];
      if (post_relu && !(av[e] * sc + sf > 0.f)) gg = 0.f;
      s1 += gg;
      s2 += gg * ((av[e] - m) * is);
    }
    if (POOL) { PCUDA_KEEP(pa + i); PCUDA_KEEP(pd2 + i); }
  }
  s1 = block_sum(s1, sh);
  s2 = block_sum(s2, sh);
  if (threadIdx.x == 0) {
    const long long tile = (long long)n * gridDim.x + blockIdx.x;
    red[(tile * c + ch) * 2 + 0] = s1;
    red[(tile * c + ch) * 2 + 1] = s2;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ red, int ntiles, int c,
                                                              double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ mean, float* dgamma,
                                                              float* dbeta, int accumulate, float* coef) {
  __shared__ double sh[2][256];
  const int ch = blockIdx.x;
  double s1, s2;
  tile_pair_sum(red, ntiles, c, ch, s1, s2);
  sh[0][threadIdx.x] = s1;
  sh[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double S1 = sh[0][0], S2 = sh[1][0];
    if (dgamma) dgamma[ch] = accumulate ? dgamma[ch] + (float)S2 : (float)S2;
    if (dbeta) dbeta[ch] = accumulate ? dbeta[ch] + (float)S1 : (float)S1;
    const double g = gamma ? (double)gamma[ch] : 1.0, is = invstd[ch], m = mean[ch];
    const double sc = g * is;
    coef[ch * 3 + 0] = (float)sc;
    // count < 0: FROZEN statistics (eval-mode BatchNorm, mean / invstd from the running buffers): the layer is a fixed
    // per-channel affine, the batch-mean terms of the training formula vanish; dgamma / dbeta are the same sums
    coef[ch * 3 + 1] = count < 0 ? 0.f : (float)(-sc * is * S2 / count);
    coef[ch * 3 + 2] = count < 0 ? 0.f : (float)(-sc * S1 / count + sc * is * (S2 / count) * m);
  }
}

template <int VEC, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, long long dy_sn, long long dy_sc, const float* __restrict__ dy2, long long dy2_sn,
    long long dy2_sc, const float* __restrict__ a, long long a_sn, long long a_sc, const float* __restrict__ coef,
    const float* __restrict__ scale, const float* __restrict__ shift, int post_relu, float act_slope,
    float* __restrict__ dz, long long dz_sn, long long dz_sc, long long hw, PoolSrc ps) {
  const int ch = blockIdx.y, n = blockIdx.z;
  const float* pd = dy ? dy + n * dy_sn + ch * dy_sc : nullptr;
  const float* pd2 = dy2 ? dy2 + n * dy2_sn + ch * dy2_sc : nullptr;
  const float* pa = a + n * a_sn + ch * a_sc;
  float* pz = dz + n * dz_sn + ch * dz_sc;
  const float c0 = coef[ch * 3 + 0], c1 = coef[ch * 3 + 1], c2 = coef[ch * 3 + 2];
  const float sc = post_relu ? scale[ch] : 0.f, sf = post_relu ? shift[ch] : 0.f;
  const long long i0 = (long long)blockIdx.x * PCH, i1 = min(hw, i0 + PCH);
  for (long long i = i0 + threadIdx.x * VEC; i < i1; i += 256 * VEC) {
    float av[VEC], g[VEC], g2[VEC], out[VEC];
    vload<VEC>(pa + i, av);
    if (POOL) {
      pool_load<VEC>(ps, n, ch, i, g);
      if (pd) {
        vload<VEC>(pd + i, g2);
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] += g2[e];
      }
    } else {
      vload<VEC>(pd + i, g);
    }
    if (pd2) vload<VEC>(pd2 + i, g2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float gg = pd2 ? g[e] + g2[e] : g[e];
      if (post_relu) {
        if (!(av[e] * sc + sf > 0.f)) gg = 0.f;
        out[e] = c0 * gg + c1 * av[e] + c2;
      } else {
        out[e] = (c0 * gg + c1 * av[e] + c2) * (av[e] > 0.f ? 1.f : act_slope);
      }
    }
    vstore<VEC>(pz + i, out);
    if (POOL) { PCUDA_KEEP(pa + i); PCUDA_KEEP(pd2 + i); }
  }
}

template <int VEC, bool POOL = false>
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ dy, long long dy_sn, long long dy_sc,
                                                        const float* __restrict__ dy2, long long dy2_sn,
                                                        long long dy2_sc, const float* __restrict__ a, long long a_sn,
                                                        long long a_sc, float slope, float* __restrict__ dz,
                                                        long long dz_sn, long long dz_sc, long long hw, PoolSrc ps) {
  const int ch = blockIdx.y, n = blockIdx.z;
  const float* pd = dy ? dy + n * dy_sn + ch * dy_sc : nullptr;
  const float* pd2 = dy2 ? dy2 + n * dy2_sn + ch * dy2_sc : nullptr;
  const float* pa = a + n * a_sn + ch * a_sc;
  float* pz = dz + n * dz_sn + ch * dz_sc;
  const long long i0 = (long long)blockIdx.x * PCH, i1 = min(hw, i0 + PCH);
  for (long long i = i0 + threadIdx.x * VEC; i < i1; i += 256 * VEC) {
    float av[VEC], g[VEC], g2[VEC], out[VEC];
    vload<VEC>(pa + i, av);
    if (POOL) {
      pool_load<VEC>(ps, n, ch, i, g);
      if (pd) {
        vload<VEC>(pd + i, g2);
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] += g2[e];
      }
    } else {
      vload<VEC>(pd + i, g);
    }
    if (pd2) vload<VEC>(pd2 + i, g2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float gg = pd2 ? g[e] + g2[e] : g[e];
      out[e] = av[e] > 0.f ? gg : gg * slope;
    }
    vstore<VEC>(pz + i, out);
    if (POOL) { PCUDA_KEEP(pa + i); PCUDA_KEEP(pd2 + i); }
  }
}

__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ dz, long long sn,
                                                                  long long sc, long long hw, int c,
                                                                  float* __restrict__ part) {
  __shared__ float sh[4];
  const int ch = blockIdx.y, n = blockIdx.z;
  const float* pl = dz + n * sn + ch * sc;
  const long long i0 = (long long)blockIdx.x * PCH, i1 = min(hw, i0 + PCH);
  float s = 0.f;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) s += pl[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[((long long)n * gridDim.x + blockIdx.x) * c + ch] = s;
}

__global__ __launch_bounds__(256) void channel_sum_final_kernel(const float* __restrict__ part, int ntiles, int c,
                                                                float* db, int accumulate) {
  __shared__ double sh[256];
  const int ch = blockIdx.x;
  double s = 0;
  for (int t = threadIdx.x; t < ntiles; t += 256) s += (double)part[(long long)t * c + ch];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[ch] = accumulate ? db[ch] + (float)sh[0] : (float)sh[0];
}

// ---------------------------------------------------------------------------- pooling / resampling
// one thread per pooled element; first maximum in row-major window order wins (ATen semantics)
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, long long x_sn, long long x_sc,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ y,
                                                           long long y_sn, long long y_sc, uint8_t* __restrict__ idx,
                                                           int c, int h, int w) {
  const int ch = blockIdx.y, n = blockIdx.z;
  const int oh = h >> 1, ow = w >> 1;
  const float sc = scale ? scale[ch] : 1.f, sf = shift ? shift[ch] : 0.f;
  const float* px = x + n * x_sn + ch * x_sc;
  float* py = y + n * y_sn + ch * y_sc;
  uint8_t* pi = idx + ((long long)n * c + ch) * oh * ow;
  const int total = oh * ow;
  for (int o = blockIdx.x * PCH + threadIdx.x; o < min(total, (int)(blockIdx.x + 1) * PCH); o += 256) {
    const int oy = o / ow, ox = o - oy * ow;
    const float* r0 = px + (long long)(2 * oy) * w + 2 * ox;
    const float2 t0 = *(const float2*)r0;
    const float2 t1 = *(const float2*)(r0 + w);
    const float v0 = t0.x * sc + sf, v1 = t0.y * sc + sf, v2 = t1.x * sc + sf, v3 = t1.y * sc + sf;
    float m = v0;
    int k = 0;
    if (v1 > m) { m = v1; k = 1; }
    if (v2 > m) { m = v2; k = 2; }
    if (v3 > m) { m = v3; k = 3; }
    py[o] = m;
    pi[o] = (uint8_t)k;
  }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, long long dy_sn,
                                                           long long dy_sc, const float* __restrict__ dy2,
                                                           long long dy2_sn, long long dy2_sc,
                                                           const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                           long long dx_sn, long long dx_sc, int accumulate, int c,
                                                           int h, int w) {
  const int ch = blockIdx.y, n = blockIdx.z;
  const int oh = h >> 1, ow = w >> 1;
  const float* pd = dy + n * dy_sn + ch * dy_sc;
  const float* pd2 = dy2 ? dy2 + n * dy2_sn + ch * dy2_sc : nullptr;
  const uint8_t* pi = idx + ((long long)n * c + ch) * oh * ow;
  float* px = dx + n * dx_sn + ch * dx_sc;
  const int total = oh * ow;
  for (int o = blockIdx.x * PCH + threadIdx.x; o < min(total, (int)(blockIdx.x + 1) * PCH); o += 256) {
    const int oy = o / ow, ox = o - oy * ow;
    float g = pd[o];
    if (pd2) g += pd2[o];
    const int k = pi[o];
    float* r0 = px + (long long)(2 * oy) * w + 2 * ox;
    float2 t0 = make_float2(k == 0 ? g : 0.f, k == 1 ? g : 0.f);
    float2 t1 = make_float2(k == 2 ? g : 0.f, k == 3 ? g : 0.f);
    if (accumulate) {
      const float2 a0 = *(float2*)r0, a1 = *(float2*)(r0 + w);
      t0.x += a0.x; t0.y += a0.y; t1.x += a1.x; t1.y += a1.y;
    }
    *(float2*)r0 = t0;
    *(float2*)(r0 + w) = t1;
  }
}

// RED: the BatchNorm-backward reduce of the layer that consumes dx rides along: per workgroup and channel
// (sum g, sum g * (a - mean) * invstd) over the stored gradient g -- bn_bwd_reduce's partials, same tile indexing
template <bool RED>
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dy, long long dy_sn,
                                                            long long dy_sc, float* __restrict__ dx, long long dx_sn,
                                                            long long dx_sc, int accumulate, int h, int w,
                                                            const float* __restrict__ a, long long a_sn, long long a_sc,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int c,
                                                            float* __restrict__ red) {
  __shared__ float sh[4];
  const int ch = blockIdx.y, n = blockIdx.z;
  const float* pd = dy + n * dy_sn + ch * dy_sc;
  float* px = dx + n * dx_sn + ch * dx_sc;
  const float* pa = RED ? a + n * a_sn + ch * a_sc : nullptr;
  const float m = RED ? mean[ch] : 0.f, is = RED ? invstd[ch] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  const int total = h * w;
  for (int o = blockIdx.x * PCH + threadIdx.x; o < min(total, (int)(blockIdx.x + 1) * PCH); o += 256) {
    const int oy = o / w, ox = o - oy * w;
    const float* r0 = pd + (long long)(2 * oy) * (2 * w) + 2 * ox;
    const float2 t0 = *(const float2*)r0;
    const float2 t1 = *(const float2*)(r0 + 2 * w);
    float s = (t0.x + t0.y) + (t1.x + t1.y);
    if (accumulate) s += px[o];
    px[o] = s;
    if (RED) {
      s1 += s;
      s2 += s * ((pa[o] - m) * is);
    }
  }
  if (RED) {
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
      const long long tile = (long long)n * gridDim.x + blockIdx.x;
      red[(tile * c + ch) * 2 + 0] = s1;
      red[(tile * c + ch) * 2 + 1] = s2;
    }
  }
}

// taps unfolded into channels: u[n][c*k*k + ky*k + kx][oy][ox] = x[n][c][oy*stride - pad + ky*dil][ox*stride - pad + kx*dil]
// (0 outside).  For layers with a handful of input channels (the discriminators' first 4x4 layer: 4 or 5) the
// convolution over u is a 1x1 layer with a 32-deep reduction chunk that is FULL, instead of 16 taps that each use
// 4 of a chunk's 32 channels.
__global__ __launch_bounds__(256) void unfold_taps_kernel(const float* __restrict__ x, long long x_sn, long long x_sc,
                                                          int h, int w, int k, int stride, int pad, int dil,
                                                          float* __restrict__ u, int oh, int ow) {
  const int kc = blockIdx.y, n = blockIdx.z;        // kc = c*k*k + ky*k + kx
  const int c = kc / (k * k), t = kc - c * k * k, ky = t / k, kx = t - ky * k;
  const float* px = x + n * x_sn + c * x_sc;
  float* pu = u + ((long long)n * gridDim.y + kc) * oh * ow;
  const int total = oh * ow;
  for (int o = blockIdx.x * PCH + threadIdx.x; o < min(total, (int)(blockIdx.x + 1) * PCH); o += 256) {
    const int oy = o / ow, ox = o - oy * ow;
    const int iy = oy * stride - pad + ky * dil, ix = ox * stride - pad + kx * dil;
    pu[o] = ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? px[(long long)iy * w + ix] : 0.f;
  }
}

// bilinear resize with align_corners=True (nn.UpsamplingBilinear2d, GAN.py:57,78): ATen's upsample_bilinear2d
// arithmetic -- src = dst * (in - 1) / (out - 1); i0 = (int)src; i1 = i0 + (i0 < in - 1); l1 = src - i0; l0 = 1 - l1
__device__ __forceinline__ void bil_src(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  const float src = scale * (float)o;
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, long long x_sn, long long x_sc,
                                                           int h, int w, float* __restrict__ y, int oh, int ow,
                                                           float sy, float sx) {
  const int ch = blockIdx.y, n = blockIdx.z;
  const float* px = x + n * x_sn + ch * x_sc;
  float* py = y + ((long long)n * gridDim.y + ch) * oh * ow;
  const int total = oh * ow;
  for (int o = blockIdx.x * PCH + threadIdx.x; o < min(total, (int)(blockIdx.x + 1) * PCH); o += 256) {
    const int oy = o / ow, ox = o - oy * ow;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bil_src(oy, sy, h, y0, y1, ly0, ly1);
    bil_src(ox, sx, w, x0, x1, lx0, lx1);
    py[o] = ly0 * (lx0 * px[(long long)y0 * w + x0] + lx1 * px[(long long)y0 * w + x1]) +
            ly1 * (lx0 * px[(long long)y1 * w + x0] + lx1 * px[(long long)y1 * w + x1]);
  }
}
// gather form of the backward pass (deterministic, no atomics): input pixel (iy, ix) collects every output pixel whose
// (y0 | y1, x0 | x1) names it; candidates come from a conservative window around iy / scale
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, int oh, int ow,
                                                           float* __restrict__ dx, long long dx_sn, long long dx_sc,
                                                           int h, int w, float sy, float sx) {
  const int ch = blockIdx.y, n = blockIdx.z;
  const float* pd = dy + ((long long)n * gridDim.y + ch) * oh * ow;
  float* px = dx + n * dx_sn + ch * dx_sc;
  const int total = h * w;
  const float isy = sy > 0.f ? 1.f / sy : 0.f, isx = sx > 0.f ? 1.f / sx : 0.f;
  for (int o = blockIdx.x * PCH + threadIdx.x; o < min(total, (int)(blockIdx.x + 1) * PCH); o += 256) {
    const int iy = o / w, ix = o - iy * w;
    const int oy_lo = sy > 0.f ? max(0, (int)((iy - 1) * isy) - 1) : 0, oy_hi = sy > 0.f ? min(oh - 1, (int)((iy + 1) * isy) + 1) : oh - 1;
    const int ox_lo = sx > 0.f ? max(0, (int)((ix - 1) * isx) - 1) : 0, ox_hi = sx > 0.f ? min(ow - 1, (int)((ix + 1) * isx) + 1) : ow - 1;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly0, ly1;
      bil_src(oy, sy, h, y0, y1, ly0, ly1);
      const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
      if (wy == 0.f && y0 != iy && y1 != iy) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        bil_src(ox, sx, w, x0, x1, lx0, lx1);
        const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
        acc += wy * wx * pd[(long long)oy * ow + ox];
      }
    }
    px[o] = acc;
  }
}

__global__ __launch_bounds__(256) void add4_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ c, const float* __restrict__ d,
                                                   float* __restrict__ y, long long numel) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < numel; i += 256ll * gridDim.x) {
    float v = a[i] + b[i];
    if (c) v += c[i];
    if (d) v += d[i];
    y[i] = v;
  }
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long long numel) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < numel; i += 256ll * gridDim.x) y[i] = a[i] * b[i];
}

// ============================================================================ host wrappers
namespace {
inline bool dims_ok(int n, int c, long long hw) { return n > 0 && c > 0 && hw > 0 && n <= 65535 && c <= 65535; }
inline dim3 plane_grid(int n, int c, long long hw) { return dim3((unsigned)cdiv(hw, PCH), (unsigned)c, (unsigned)n); }
}  // namespace

extern "C" int pcuda_bn_stats(const float* a, long long sn, long long sc, int n, int c, long long hw,
                              float* partials, int* ntiles, pcuda_stream_t s) {
  if (!dims_ok(n, c, hw)) PCUDA_FAIL(PCUDA_E_BADARG, "bn_stats: bad dims");
  const int nt = n * cdiv(hw, PCH);
  if (ntiles) *ntiles = nt;
  if (!partials) return PCUDA_OK;
  if (!a) PCUDA_FAIL(PCUDA_E_BADARG, "bn_stats: null input");
  ProfScope prof(PCUDA_FAM_POINTWISE, 4.0 * n * c * (double)hw, (hipStream_t)s);
  hipLaunchKernelGGL(bn_stats_kernel, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, a, sn, sc, hw, c, partials);
  PCUDA_CHECK_LAUNCH("bn_stats_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_bn_finalize(const float* partials, int ntiles, int c, long long count, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean,
                                 float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                 pcuda_stream_t s) {
  if (!partials || ntiles <= 0 || c <= 0 || count <= 0 || !mean || !invstd || !scale || !shift)
    PCUDA_FAIL(PCUDA_E_BADARG, "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)s, partials, ntiles, c, (double)count,
                     gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  PCUDA_CHECK_LAUNCH("bn_finalize_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_bn_apply(const float* a, long long a_sn, long long a_sc, const float* scale, const float* shift,
                              int relu, float* y, long long y_sn, long long y_sc, int n, int c, long long hw,
                              pcuda_stream_t s) {
  if (!dims_ok(n, c, hw) || !a || !y || !scale || !shift) PCUDA_FAIL(PCUDA_E_BADARG, "bn_apply: bad arguments");
  ProfScope prof(PCUDA_FAM_POINTWISE, 8.0 * n * c * (double)hw, (hipStream_t)s);
  if (vec_ok(a, a_sn, a_sc, hw) && vec_ok(y, y_sn, y_sc, hw))
    hipLaunchKernelGGL(bn_apply_kernel<4>, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, a, a_sn, a_sc, scale,
                       shift, relu, y, y_sn, y_sc, hw);
  else
    hipLaunchKernelGGL(bn_apply_kernel<1>, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, a, a_sn, a_sc, scale,
                       shift, relu, y, y_sn, y_sc, hw);
  PCUDA_CHECK_LAUNCH("bn_apply_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_bn_bwd_reduce(const float* dy, long long dy_sn, long long dy_sc, const float* dy2,
                                   long long dy2_sn, long long dy2_sc, const float* a, long long a_sn, long long a_sc,
                                   const float* mean, const float* invstd, const float* scale, const float* shift,
                                   int post_relu, int n, int c, long long hw, float* red, int* ntiles,
                                   pcuda_stream_t s) {
  if (!dims_ok(n, c, hw)) PCUDA_FAIL(PCUDA_E_BADARG, "bn_bwd_reduce: bad dims");
  const int nt = n * cdiv(hw, PCH);
  if (ntiles) *ntiles = nt;
  if (!red) return PCUDA_OK;
  if (!dy || !a || !mean || !invstd || (post_relu && (!scale || !shift)))
    PCUDA_FAIL(PCUDA_E_BADARG, "bn_bwd_reduce: null pointer");
  ProfScope prof(PCUDA_FAM_POINTWISE, (dy2 ? 12.0 : 8.0) * n * c * (double)hw, (hipStream_t)s);
  if (vec_ok(dy, dy_sn, dy_sc, hw) && vec_ok(dy2, dy2_sn, dy2_sc, hw) && vec_ok(a, a_sn, a_sc, hw))
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, dy, dy_sn, dy_sc,
                       dy2, dy2_sn, dy2_sc, a, a_sn, a_sc, mean, invstd, scale, shift, post_relu, hw, c, red, PoolSrc{});
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, dy, dy_sn, dy_sc,
                       dy2, dy2_sn, dy2_sc, a, a_sn, a_sc, mean, invstd, scale, shift, post_relu, hw, c, red, PoolSrc{});
  PCUDA_CHECK_LAUNCH("bn_bwd_reduce_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_bn_bwd_finalize(const float* red, int ntiles, int c, long long count, const float* gamma,
                                     const float* invstd, const float* mean, float* dgamma, float* dbeta,
                                     int accumulate, float* coef, pcuda_stream_t s) {
  if (!red || ntiles <= 0 || c <= 0 || count == 0 || !invstd || !mean || !coef)
    PCUDA_FAIL(PCUDA_E_BADARG, "bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)s, red, ntiles, c, (double)count,
                     gamma, invstd, mean, dgamma, dbeta, accumulate, coef);
  PCUDA_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_bn_bwd_apply(const float* dy, long long dy_sn, long long dy_sc, const float* dy2,
                                  long long dy2_sn, long long dy2_sc, const float* a, long long a_sn, long long a_sc,
                                  const float* coef, const float* scale, const float* shift, int post_relu,
                                  float act_slope, float* dz, long long dz_sn, long long dz_sc, int n, int c,
                                  long long hw, pcuda_stream_t s) {
  if (!dims_ok(n, c, hw) || !dy || !a || !coef || !dz || (post_relu && (!scale || !shift)))
    PCUDA_FAIL(PCUDA_E_BADARG, "bn_bwd_apply: bad arguments");
  ProfScope prof(PCUDA_FAM_POINTWISE, (dy2 ? 16.0 : 12.0) * n * c * (double)hw, (hipStream_t)s);
  if (vec_ok(dy, dy_sn, dy_sc, hw) && vec_ok(dy2, dy2_sn, dy2_sc, hw) && vec_ok(a, a_sn, a_sc, hw) &&
      vec_ok(dz, dz_sn, dz_sc, hw))
    hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, dy, dy_sn, dy_sc,
                       dy2, dy2_sn, dy2_sc, a, a_sn, a_sc, coef, scale, shift, post_relu, act_slope, dz, dz_sn, dz_sc, hw, PoolSrc{});
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, dy, dy_sn, dy_sc,
                       dy2, dy2_sn, dy2_sc, a, a_sn, a_sc, coef, scale, shift, post_relu, act_slope, dz, dz_sn, dz_sc, hw, PoolSrc{});
  PCUDA_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_lrelu_bwd(const float* dy, long long dy_sn, long long dy_sc, const float* dy2, long long dy2_sn,
                               long long dy2_sc, const float* a, long long a_sn, long long a_sc, float slope,
                               float* dz, long long dz_sn, long long dz_sc, int n, int c, long long hw,
                               pcuda_stream_t s) {
  if (!dims_ok(n, c, hw) || !dy || !a || !dz) PCUDA_FAIL(PCUDA_E_BADARG, "lrelu_bwd: bad arguments");
  ProfScope prof(PCUDA_FAM_POINTWISE, (dy2 ? 16.0 : 12.0) * n * c * (double)hw, (hipStream_t)s);
  // dense tensors are ONE run of n*c*hw elements for this element-wise op: the discriminators' odd planes (129x129,
  // 65x65 ...) fail the per-plane float4 test, and the scalar form ran at a quarter of the vector rate
  auto dense = [&](const void* q, long long sn_, long long sc_) {
    return q == nullptr || ((((uintptr_t)q) & 15) == 0 && sc_ == hw && sn_ == (long long)c * hw);
  };
  const long long total = (long long)n * c * hw;
  if ((hw & 3) && (total & 3) == 0 && dense(dy, dy_sn, dy_sc) && dense(dy2, dy2_sn, dy2_sc) && dense(a, a_sn, a_sc) &&
      dense(dz, dz_sn, dz_sc)) {
    hipLaunchKernelGGL(lrelu_bwd_kernel<4>, plane_grid(1, 1, total), dim3(256), 0, (hipStream_t)s, dy, total, total, dy2,
                       total, total, a, total, total, slope, dz, total, total, total, PoolSrc{});
    PCUDA_CHECK_LAUNCH("lrelu_bwd_kernel");
    return PCUDA_OK;
  }
  if (vec_ok(dy, dy_sn, dy_sc, hw) && vec_ok(dy2, dy2_sn, dy2_sc, hw) && vec_ok(a, a_sn, a_sc, hw) &&
      vec_ok(dz, dz_sn, dz_sc, hw))
    hipLaunchKernelGGL(lrelu_bwd_kernel<4>, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, dy, dy_sn, dy_sc, dy2,
                       dy2_sn, dy2_sc, a, a_sn, a_sc, slope, dz, dz_sn, dz_sc, hw, PoolSrc{});
  else
    hipLaunchKernelGGL(lrelu_bwd_kernel<1>, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, dy, dy_sn, dy_sc, dy2,
                       dy2_sn, dy2_sc, a, a_sn, a_sc, slope, dz, dz_sn, dz_sc, hw, PoolSrc{});
  PCUDA_CHECK_LAUNCH("lrelu_bwd_kernel");
  return PCUDA_OK;
}

// ---- the same three kernels with the gradient arriving through a 2x2 max-pool (PoolSrc above): `dy` here is the OPTIONAL
// full-resolution share added to the scattered one (the encoder's skip gradient)
static int pool_src(const pcuda_pooled* p, int n, int c, PoolSrc& ps, const char* who) {
  if (!p || !p->g || !p->idx || p->h <= 0 || p->w <= 0 || (p->h & 1) || (p->w & 1) || !dims_ok(n, c, (long long)p->h * p->w))
    PCUDA_FAIL(PCUDA_E_BADARG, "%s: bad pooled source (even plane, g and idx required)", who);
  ps.g = p->g; ps.sn = p->g_sn; ps.sc = p->g_sc;
  ps.g2 = p->g2; ps.sn2 = p->g2_sn; ps.sc2 = p->g2_sc;
  ps.idx = p->idx; ps.w = p->w; ps.c = c; ps.ohw = (long long)(p->h >> 1) * (p->w >> 1);
  return PCUDA_OK;
}
// float2 reads of g / g2 and 2-byte reads of idx at even pooled offsets: rows a multiple of 4 wide, even strides, aligned bases
static bool pool_vec_ok(const pcuda_pooled* p) {
  auto ok = [](const void* q, long long sn, long long sc) { return q == nullptr || (((uintptr_t)q & 7) == 0 && ((sn | sc) & 1) == 0); };
  return (p->w & 3) == 0 && ok(p->g, p->g_sn, p->g_sc) && ok(p->g2, p->g2_sn, p->g2_sc) && ((uintptr_t)p->idx & 1) == 0 &&
         ((((long long)(p->h >> 1) * (p->w >> 1)) & 1) == 0);
}

extern "C" int pcuda_bn_bwd_reduce_pooled(const pcuda_pooled* pool, const float* dy, long long dy_sn, long long dy_sc,
                                          const float* a, long long a_sn, long long a_sc, const float* mean,
                                          const float* invstd, int n, int c, float* red, int* ntiles, pcuda_stream_t s) {
  if (!pool) PCUDA_FAIL(PCUDA_E_BADARG, "bn_bwd_reduce_pooled: null pooled source");
  const long long hw = (long long)pool->h * pool->w;
  if (!dims_ok(n, c, hw)) PCUDA_FAIL(PCUDA_E_BADARG, "bn_bwd_reduce_pooled: bad dims");
  const int nt = n * cdiv(hw, PCH);
  if (ntiles) *ntiles = nt;
  if (!red) return PCUDA_OK;
  PoolSrc ps{};
  if (int rc = pool_src(pool, n, c, ps, "bn_bwd_reduce_pooled")) return rc;
  if (!a || !mean || !invstd) PCUDA_FAIL(PCUDA_E_BADARG, "bn_bwd_reduce_pooled: null pointer");
  ProfScope prof(PCUDA_FAM_POINTWISE, ((dy ? 8.0 : 4.0) + (pool->g2 ? 2.25 : 1.25)) * n * c * (double)hw, (hipStream_t)s);
  if (pool_vec_ok(pool) && vec_ok(dy, dy_sn, dy_sc, hw) && vec_ok(a, a_sn, a_sc, hw))
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, true>), plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, nullptr, 0, 0,
                       dy, dy_sn, dy_sc, a, a_sn, a_sc, mean, invstd, nullptr, nullptr, 0, hw, c, red, ps);
  else
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<1, true>), plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, nullptr, 0, 0,
                       dy, dy_sn, dy_sc, a, a_sn, a_sc, mean, invstd, nullptr, nullptr, 0, hw, c, red, ps);
  PCUDA_CHECK_LAUNCH("bn_bwd_reduce_kernel<pooled>");
  return PCUDA_OK;
}

extern "C" int pcuda_bn_bwd_apply_pooled(const pcuda_pooled* pool, const float* dy, long long dy_sn, long long dy_sc,
                                         const float* a, long long a_sn, long long a_sc, const float* coef,
                                         float act_slope, float* dz, long long dz_sn, long long dz_sc, int n, int c,
                                         pcuda_stream_t s) {
  PoolSrc ps{};
  if (int rc = pool_src(pool, n, c, ps, "bn_bwd_apply_pooled")) return rc;
  if (!a || !coef || !dz) PCUDA_FAIL(PCUDA_E_BADARG, "bn_bwd_apply_pooled: null pointer");
  const long long hw = (long long)pool->h * pool->w;
  ProfScope prof(PCUDA_FAM_POINTWISE, ((dy ? 12.0 : 8.0) + (pool->g2 ? 2.25 : 1.25)) * n * c * (double)hw, (hipStream_t)s);
  if (pool_vec_ok(pool) && vec_ok(dy, dy_sn, dy_sc, hw) && vec_ok(a, a_sn, a_sc, hw) && vec_ok(dz, dz_sn, dz_sc, hw))
    hipLaunchKernelGGL((bn_bwd_apply_kernel<4, true>), plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, nullptr, 0, 0, dy,
                       dy_sn, dy_sc, a, a_sn, a_sc, coef, nullptr, nullptr, 0, act_slope, dz, dz_sn, dz_sc, hw, ps);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<1, true>), plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, nullptr, 0, 0, dy,
                       dy_sn, dy_sc, a, a_sn, a_sc, coef, nullptr, nullptr, 0, act_slope, dz, dz_sn, dz_sc, hw, ps);
  PCUDA_CHECK_LAUNCH("bn_bwd_apply_kernel<pooled>");
  return PCUDA_OK;
}

extern "C" int pcuda_lrelu_bwd_pooled(const pcuda_pooled* pool, const float* dy, long long dy_sn, long long dy_sc,
                                      const float* a, long long a_sn, long long a_sc, float slope, float* dz,
                                      long long dz_sn, long long dz_sc, int n, int c, pcuda_stream_t s) {
  PoolSrc ps{};
  if (int rc = pool_src(pool, n, c, ps, "lrelu_bwd_pooled")) return rc;
  if (!a || !dz) PCUDA_FAIL(PCUDA_E_BADARG, "lrelu_bwd_pooled: null pointer");
  const long long hw = (long long)pool->h * pool->w;
  ProfScope prof(PCUDA_FAM_POINTWISE, ((dy ? 12.0 : 8.0) + (pool->g2 ? 2.25 : 1.25)) * n * c * (double)hw, (hipStream_t)s);
  if (pool_vec_ok(pool) && vec_ok(dy, dy_sn, dy_sc, hw) && vec_ok(a, a_sn, a_sc, hw) && vec_ok(dz, dz_sn, dz_sc, hw))
    hipLaunchKernelGGL((lrelu_bwd_kernel<4, true>), plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, nullptr, 0, 0, dy,
                       dy_sn, dy_sc, a, a_sn, a_sc, slope, dz, dz_sn, dz_sc, hw, ps);
  else
    hipLaunchKernelGGL((lrelu_bwd_kernel<1, true>), plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, nullptr, 0, 0, dy,
                       dy_sn, dy_sc, a, a_sn, a_sc, slope, dz, dz_sn, dz_sc, hw, ps);
  PCUDA_CHECK_LAUNCH("lrelu_bwd_kernel<pooled>");
  return PCUDA_OK;
}

extern "C" int pcuda_channel_sum(const float* dz, long long sn, long long sc, int n, int c, long long hw, float* db,
                                 int accumulate, float* workspace, size_t workspace_bytes, pcuda_stream_t s) {
  if (!dims_ok(n, c, hw) || !dz || !db) PCUDA_FAIL(PCUDA_E_BADARG, "channel_sum: bad arguments");
  const int nt = n * cdiv(hw, PCH);
  if (!workspace || workspace_bytes < (size_t)nt * c * sizeof(float))
    PCUDA_FAIL(PCUDA_E_WORKSPACE, "channel_sum: workspace too small (need %zu)", (size_t)nt * c * sizeof(float));
  hipLaunchKernelGGL(channel_sum_partial_kernel, plane_grid(n, c, hw), dim3(256), 0, (hipStream_t)s, dz, sn, sc, hw, c,
                     workspace);
  PCUDA_CHECK_LAUNCH("channel_sum_partial_kernel");
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3(c), dim3(256), 0, (hipStream_t)s, (const float*)workspace, nt, c,
                     db, accumulate);
  PCUDA_CHECK_LAUNCH("channel_sum_final_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_maxpool2_fwd(const float* x, long long x_sn, long long x_sc, const float* scale,
                                  const float* shift, float* y, long long y_sn, long long y_sc, uint8_t* idx, int n,
                                  int c, int h, int w, pcuda_stream_t s) {
  if (!dims_ok(n, c, (long long)h * w) || (h & 1) || (w & 1) || !x || !y || !idx || ((x_sn | x_sc) & 1) ||
      (((uintptr_t)x) & 7))
    PCUDA_FAIL(PCUDA_E_BADARG, "maxpool2_fwd: bad arguments (even dims, 8-byte aligned planes required)");
  const long long ohw = (long long)(h / 2) * (w / 2);
  ProfScope prof(PCUDA_FAM_POINTWISE, 21.0 * n * c * (double)ohw, (hipStream_t)s);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, plane_grid(n, c, ohw), dim3(256), 0, (hipStream_t)s, x, x_sn, x_sc, scale,
                     shift, y, y_sn, y_sc, idx, c, h, w);
  PCUDA_CHECK_LAUNCH("maxpool2_fwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_maxpool2_bwd(const float* dy, long long dy_sn, long long dy_sc, const float* dy2,
                                  long long dy2_sn, long long dy2_sc, const uint8_t* idx, float* dx, long long dx_sn,
                                  long long dx_sc, int accumulate, int n, int c, int h, int w, pcuda_stream_t s) {
  if (!dims_ok(n, c, (long long)h * w) || (h & 1) || (w & 1) || !dy || !dx || !idx || ((dx_sn | dx_sc) & 1) ||
      (((uintptr_t)dx) & 7))
    PCUDA_FAIL(PCUDA_E_BADARG, "maxpool2_bwd: bad arguments");
  const long long ohw = (long long)(h / 2) * (w / 2);
  ProfScope prof(PCUDA_FAM_POINTWISE, 21.0 * n * c * (double)ohw, (hipStream_t)s);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, plane_grid(n, c, ohw), dim3(256), 0, (hipStream_t)s, dy, dy_sn, dy_sc, dy2,
                     dy2_sn, dy2_sc, idx, dx, dx_sn, dx_sc, accumulate, c, h, w);
  PCUDA_CHECK_LAUNCH("maxpool2_bwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_upsample2_bwd(const float* dy, long long dy_sn, long long dy_sc, float* dx, long long dx_sn,
                                   long long dx_sc, int accumulate, int n, int c, int h, int w, pcuda_stream_t s) {
  if (!dims_ok(n, c, (long long)h * w) || !dy || !dx || ((dy_sn | dy_sc) & 1) || (((uintptr_t)dy) & 7))
    PCUDA_FAIL(PCUDA_E_BADARG, "upsample2_bwd: bad arguments");
  ProfScope prof(PCUDA_FAM_POINTWISE, 20.0 * n * c * (double)h * w, (hipStream_t)s);
  hipLaunchKernelGGL(upsample2_bwd_kernel<false>, plane_grid(n, c, (long long)h * w), dim3(256), 0, (hipStream_t)s, dy, dy_sn,
                     dy_sc, dx, dx_sn, dx_sc, accumulate, h, w, nullptr, 0, 0, nullptr, nullptr, c, nullptr);
  PCUDA_CHECK_LAUNCH("upsample2_bwd_kernel");
  return PCUDA_OK;
}

/* the same with the BatchNorm-backward reduce of the layer that consumes dx fused in (unet.py:128-136: the decoder's
 * up-convolution feeds conv -> LeakyReLU -> BN blocks; going back, the 2x2 fold's output IS that BatchNorm's incoming
 * gradient): red[ntiles][c][2], ntiles as pcuda_bn_bwd_reduce reports them (query with red == NULL) */
extern "C" int pcuda_upsample2_bwd_bnred(const float* dy, long long dy_sn, long long dy_sc, float* dx, long long dx_sn,
                                         long long dx_sc, int accumulate, const float* a, long long a_sn, long long a_sc,
                                         const float* mean, const float* invstd, float* red, int* ntiles, int n, int c,
                                         int h, int w, pcuda_stream_t s) {
  if (!dims_ok(n, c, (long long)h * w)) PCUDA_FAIL(PCUDA_E_BADARG, "upsample2_bwd_bnred: bad dims");
  if (ntiles) *ntiles = n * cdiv((long long)h * w, PCH);
  if (!red) return PCUDA_OK;
  if (!dy || !dx || !a || !mean || !invstd || ((dy_sn | dy_sc) & 1) || (((uintptr_t)dy) & 7))
    PCUDA_FAIL(PCUDA_E_BADARG, "upsample2_bwd_bnred: bad arguments");
  ProfScope prof(PCUDA_FAM_POINTWISE, 24.0 * n * c * (double)h * w, (hipStream_t)s);
  hipLaunchKernelGGL(upsample2_bwd_kernel<true>, plane_grid(n, c, (long long)h * w), dim3(256), 0, (hipStream_t)s, dy, dy_sn,
                     dy_sc, dx, dx_sn, dx_sc, accumulate, h, w, a, a_sn, a_sc, mean, invstd, c, red);
  PCUDA_CHECK_LAUNCH("upsample2_bwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_unfold_taps(const float* x, long long x_sn, long long x_sc, int n, int c, int h, int w, int k,
                                 int stride, int pad, int dil, float* u, int oh, int ow, pcuda_stream_t s) {
  if (!x || !u || k < 1 || stride < 1 || dil < 1 || pad < 0 || h < 1 || w < 1 ||
      oh != (h + 2 * pad - dil * (k - 1) - 1) / stride + 1 || ow != (w + 2 * pad - dil * (k - 1) - 1) / stride + 1 ||
      !dims_ok(n, c * k * k, (long long)oh * ow))
    PCUDA_FAIL(PCUDA_E_BADARG, "unfold_taps: bad arguments");
  ProfScope prof(PCUDA_FAM_POINTWISE, 8.0 * n * c * k * k * (double)oh * ow, (hipStream_t)s);
  hipLaunchKernelGGL(unfold_taps_kernel, plane_grid(n, c * k * k, (long long)oh * ow), dim3(256), 0, (hipStream_t)s, x,
                     x_sn, x_sc, h, w, k, stride, pad, dil, u, oh, ow);
  PCUDA_CHECK_LAUNCH("unfold_taps_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_bilinear_fwd(const float* x, long long x_sn, long long x_sc, int n, int c, int h, int w, float* y,
                                  int oh, int ow, pcuda_stream_t s) {
  if (!x || !y || h < 1 || w < 1 || oh < 1 || ow < 1 || !dims_ok(n, c, (long long)oh * ow))
    PCUDA_FAIL(PCUDA_E_BADARG, "bilinear_fwd: bad arguments");
  const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
  ProfScope prof(PCUDA_FAM_POINTWISE, 20.0 * n * c * (double)oh * ow, (hipStream_t)s);
  hipLaunchKernelGGL(bilinear_fwd_kernel, plane_grid(n, c, (long long)oh * ow), dim3(256), 0, (hipStream_t)s, x, x_sn,
                     x_sc, h, w, y, oh, ow, sy, sx);
  PCUDA_CHECK_LAUNCH("bilinear_fwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_bilinear_bwd(const float* dy, int n, int c, int oh, int ow, float* dx, long long dx_sn,
                                  long long dx_sc, int h, int w, pcuda_stream_t s) {
  if (!dy || !dx || h < 1 || w < 1 || oh < 1 || ow < 1 || !dims_ok(n, c, (long long)h * w))
    PCUDA_FAIL(PCUDA_E_BADARG, "bilinear_bwd: bad arguments");
  const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
  ProfScope prof(PCUDA_FAM_POINTWISE, 20.0 * n * c * (double)h * w, (hipStream_t)s);
  hipLaunchKernelGGL(bilinear_bwd_kernel, plane_grid(n, c, (long long)h * w), dim3(256), 0, (hipStream_t)s, dy, oh, ow,
                     dx, dx_sn, dx_sc, h, w, sy, sx);
  PCUDA_CHECK_LAUNCH("bilinear_bwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_add4(const float* a, const float* b, const float* c, const float* d, float* y, long long numel,
                          pcuda_stream_t s) {
  if (!a || !b || !y || numel <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "add4: bad arguments");
  const int blocks = (int)(cdiv(numel, 256) > 4096 ? 4096 : cdiv(numel, 256));
  hipLaunchKernelGGL(add4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, c, d, y, numel);
  PCUDA_CHECK_LAUNCH("add4_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_mul(const float* a, const float* b, float* y, long long numel, pcuda_stream_t s) {
  if (!a || !b || !y || numel <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "mul: bad arguments");
  const int blocks = (int)(cdiv(numel, 256) > 4096 ? 4096 : cdiv(numel, 256));
  hipLaunchKernelGGL(mul_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, y, numel);
  PCUDA_CHECK_LAUNCH("mul_kernel");
  return PCUDA_OK;
}

// ------------------------------------------------------------------------------------------
// loader-side batch assembly (data_generator_mmwhs.py:265-272, utils.py:7-29): centre crop, channel-last ->
// channel-first, labels -> one-hot uint8, in one pass over the batch
// ------------------------------------------------------------------------------------------
// img_out[b][c][y][x] = img_in[b][y0 + y][x0 + x][c];  onehot[b][k][y][x] = (mask[b][y0 + y][x0 + x] == k)
__global__ __launch_bounds__(256) void assemble_batch_kernel(const float* __restrict__ img_in, const int* __restrict__ mask,
                                                             int b, int h, int w, int c, int y0, int x0, int oh, int ow,
                                                             int k, float* __restrict__ img_out,
                                                             uint8_t* __restrict__ onehot) {
  const long long npix = (long long)b * oh * ow;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += 256ll * gridDim.x) {
    const int x = (int)(i % ow);
    const long long t = i / ow;
    const int y = (int)(t % oh), n = (int)(t / oh);
    const long long src = ((long long)n * h + (y0 + y)) * w + (x0 + x);
    const long long plane = (long long)oh * ow, dst = (long long)y * ow + x;
    for (int ch = 0; ch < c; ++ch) img_out[((long long)n * c + ch) * plane + dst] = img_in[src * c + ch];
    if (onehot) {
      const int lab = mask[src];
      for (int kk = 0; kk < k; ++kk) onehot[((long long)n * k + kk) * plane + dst] = (uint8_t)(lab == kk);
    }
  }
}

extern "C" int pcuda_assemble_batch(const float* images_hwc, const int* mask_labels, int b, int h, int w, int c,
                                    int crop, int num_classes, float* images_chw, uint8_t* onehot, pcuda_stream_t s) {
  if (!images_hwc || !images_chw || b <= 0 || h <= 0 || w <= 0 || c <= 0 || (onehot && (!mask_labels || num_classes < 2)))
    PCUDA_FAIL(PCUDA_E_BADARG, "assemble_batch: bad arguments");
  // ImageProcessor.crop_volume(vol, crop_size = crop // 2): [H/2 - crop//2, H/2 + crop//2)
  int y0 = 0, x0 = 0, oh = h, ow = w;
  if (crop > 0) {
    const int hc = crop / 2;
    y0 = h / 2 - hc; x0 = w / 2 - hc; oh = 2 * hc; ow = 2 * hc;
    if (y0 < 0 || x0 < 0 || y0 + oh > h || x0 + ow > w) PCUDA_FAIL(PCUDA_E_BADARG, "assemble_batch: crop larger than the image");
  }
  const long long npix = (long long)b * oh * ow;
  const int blocks = (int)(cdiv(npix, 256) > 4096 ? 4096 : cdiv(npix, 256));
  ProfScope prof(PCUDA_FAM_POINTWISE, (double)npix * (8.0 * c + (onehot ? 4.0 + num_classes : 0.0)), (hipStream_t)s);
  hipLaunchKernelGGL(assemble_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, images_hwc, mask_labels, b, h, w,
                     c, y0, x0, oh, ow, num_classes, images_chw, onehot);
  PCUDA_CHECK_LAUNCH("assemble_batch_kernel");
  return PCUDA_OK;
}
