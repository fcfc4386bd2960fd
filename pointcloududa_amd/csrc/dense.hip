// Small dense ops of the point-cloud discriminator and the point head (PointNetCls.py:38-63,
// 135-168, 204-214; unet.py:86,94-95): nn.Linear forward/backward, the 3x3 / 64x64 bmm's and the
// max over points.  These are launch-latency bound (<= 0.6 GFLOP per sample in total), so one
// generic fp32 LDS-tiled strided GEMM serves all of them; the k=1 Conv1d layers go through the
// MFMA convolution kernels instead (a [B,C,L] tensor is an NCHW image with H = 1).
#include "common.h"

// C[b][i][j] (+)= sum_l A[b](i,l) * B[b](l,j) + bias[j]   with arbitrary element strides
struct GemmParams {
  const float* a; long long a_sb, a_si, a_sl;
  const float* b; long long b_sb, b_sl, b_sj;
  float* c; long long c_sb, c_si, c_sj;
  const float* bias;   // per j, may be NULL
  int m, n, k, accumulate;
};

#define GT 32
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  __shared__ float As[GT][GT + 1], Bs[GT][GT + 1];
  const int bz = blockIdx.z;
  const float* A = p.a + bz * p.a_sb;
  const float* B = p.b + bz * p.b_sb;
  float* C = p.c + bz * p.c_sb;
  const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
  const int tj = threadIdx.x & 31, ti = threadIdx.x >> 5;   // 32 x 8 threads, 4 rows each
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int l0 = 0; l0 < p.k; l0 += GT) {
    for (int e = threadIdx.x; e < GT * GT; e += 256) {
      const int r = e >> 5, cc = e & 31;
      // A tile [i][l]; pick the faster-varying index along the unit-stride axis of A
      int ia, la;
      if (p.a_sl == 1) { ia = r; la = cc; } else { ia = cc; la = r; }
      As[ia][la] = (i0 + ia < p.m && l0 + la < p.k) ? A[(i0 + ia) * p.a_si + (l0 + la) * p.a_sl] : 0.f;
      int lb, jb;
      if (p.b_sj == 1) { lb = r; jb = cc; } else { lb = cc; jb = r; }
      Bs[lb][jb] = (l0 + lb < p.k && j0 + jb < p.n) ? B[(l0 + lb) * p.b_sl + (j0 + jb) * p.b_sj] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int l = 0; l < GT; ++l) {
      const float bv = Bs[l][tj];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += As[ti * 4 + q][l] * bv;
    }
    __syncthreads();
  }
  const int j = j0 + tj;
  if (j < p.n) {
    const float bj = p.bias ? p.bias[j] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + ti * 4 + q;
      if (i < p.m) {
        float* dst = C + i * p.c_si + j * p.c_sj;
        const float v = acc[q] + bj;
        *dst = p.accumulate ? *dst + v : v;
      }
    }
  }
}

// Skinny GEMM (m <= 64 rows: the PointNet classifier's Linear layers at batch 32).  The 32x32-tile kernel
// above ran those on 8-16 workgroups, each walking all of k through two barriers per 32-step: 65-150 us,
// pure latency.  Here a workgroup owns 8 output columns for ALL rows; lane = row, the sixteen waves split k and
// are combined through LDS in a fixed order (deterministic): n/8 workgroups of 16 waves, no barrier inside the k loop.
#define SK_COLS 8
#define SK_WAVES 16
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_skinny_kernel(const GemmParams p) {
  __shared__ float red[SK_WAVES][64][SK_COLS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j0 = blockIdx.x * SK_COLS;
  const int i = min(lane, p.m - 1);
  const float* arow = p.a + (long long)i * p.a_si;
  const int kq = (p.k + SK_WAVES - 1) / SK_WAVES, l_lo = wv * kq, l_hi = min(p.k, l_lo + kq);
  float acc[SK_COLS];
#pragma unroll
  for (int c = 0; c < SK_COLS; ++c) acc[c] = 0.f;
  const float* bcol[SK_COLS];
#pragma unroll
  for (int c = 0; c < SK_COLS; ++c) bcol[c] = p.b + (long long)min(j0 + c, p.n - 1) * p.b_sj;   // wave-uniform
#pragma unroll 4
  for (int l = l_lo; l < l_hi; ++l) {
    const float av = arow[(long long)l * p.a_sl];
#pragma unroll
    for (int c = 0; c < SK_COLS; ++c) acc[c] = fmaf(av, bcol[c][(long long)l * p.b_sl], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < SK_COLS; ++c) red[wv][lane][c] = acc[c];
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * SK_COLS; e += 64 * SK_WAVES) {
    const int r = e / SK_COLS, c = e - r * SK_COLS, j = j0 + c;
    if (r < p.m && j < p.n) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < SK_WAVES; ++q) v += red[q][r][c];
      v += p.bias ? p.bias[j] : 0.f;
      float* dst = p.c + (long long)r * p.c_si + (long long)j * p.c_sj;
      *dst = p.accumulate ? *dst + v : v;
    }
  }
}


// column sums: db[j] (+)= sum_i dy[i][j]; one workgroup per CP (<= 64, power of two) columns, its 256 / CP row
// parts walk the rows strided and are combined through LDS in a fixed order (deterministic).  (With a fixed 64
// columns per workgroup the point head's db -- 3 columns, 9600 rows -- ran on 12 lanes: 300 us.)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, int m, int n, float* db,
                                                     int accumulate, int cp) {
  __shared__ float sh[256];
  const int rparts = 256 / cp;
  const int c = threadIdx.x % cp, rp = threadIdx.x / cp;
  const int j = blockIdx.x * cp + c;
  float s = 0.f;
  if (j < n)
    for (int i = rp; i < m; i += rparts) s += dy[(long long)i * n + j];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int half = rparts >> 1; half > 0; half >>= 1) {      // rparts is a power of two
    if (rp < half) sh[threadIdx.x] += sh[threadIdx.x + half * cp];
    __syncthreads();
  }
  if (rp == 0 && j < n) db[j] = accumulate ? db[j] + sh[c] : sh[c];
}

// out[i] (+)= sum_k part[k][i]  (fixed order)
__global__ void splitk_reduce_kernel(const float* __restrict__ part, long long numel, int ksplit,
                                     float* __restrict__ out, int accumulate) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < numel;
       i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += part[(long long)k * numel + i];
    out[i] = accumulate ? out[i] + s : s;
  }
}

// one wave per (b, c) row of x[b][c][l]
__global__ __launch_bounds__(256) void max_points_fwd_kernel(const float* __restrict__ x, int rows, int l,
                                                             float* __restrict__ y, int* __restrict__ idx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = x + (long long)row * l;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < l; i += 64) {
    const float v = p[i];
    if (v > best) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { y[row] = best; idx[row] = bi; }
}

__global__ __launch_bounds__(256) void max_points_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                             int rows, int l, float* __restrict__ dx) {
  const long long total = (long long)rows * l;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += 256ll * gridDim.x) {
    const int row = (int)(e / l), i = (int)(e - (long long)row * l);
    dx[e] = (idx[row] == i) ? dy[row] : 0.f;
  }
}

namespace {
int launch_gemm(const GemmParams& p, int batch, hipStream_t s) {
  if (p.m <= 0 || p.n <= 0 || p.k <= 0 || batch <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "gemm: bad dims");
  if (batch == 1 && p.m <= 64 && p.k >= 64) {
    hipLaunchKernelGGL(gemm_skinny_kernel, dim3(cdiv(p.n, SK_COLS)), dim3(64 * SK_WAVES), 0, s, p);
    PCUDA_CHECK_LAUNCH("gemm_skinny_kernel");
    return PCUDA_OK;
  }
  dim3 grid(cdiv(p.n, GT), cdiv(p.m, GT), batch);
  hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, s, p);
  PCUDA_CHECK_LAUNCH("gemm_kernel");
  return PCUDA_OK;
}
}  // namespace

extern "C" int pcuda_linear_fwd(const float* x, const float* w, const float* b, float* y, int m, int k, int n,
                                pcuda_stream_t s) {
  if (!x || !w || !y) PCUDA_FAIL(PCUDA_E_BADARG, "linear_fwd: null pointer");
  GemmParams p = {x, 0, k, 1, w, 0, 1, k, y, 0, n, 1, b, m, n, k, 0};   // y[i][j] = sum_l x[i][l] w[j][l]
  return launch_gemm(p, 1, (hipStream_t)s);
}

extern "C" int pcuda_linear_bwd_x(const float* dy, const float* w, float* dx, int m, int k, int n, int accumulate,
                                  pcuda_stream_t s) {
  if (!dy || !w || !dx) PCUDA_FAIL(PCUDA_E_BADARG, "linear_bwd_x: null pointer");
  GemmParams p = {dy, 0, n, 1, w, 0, k, 1, dx, 0, k, 1, nullptr, m, k, n, accumulate};   // dx[i][c] = sum_j dy[i][j] w[j][c]
  return launch_gemm(p, 1, (hipStream_t)s);
}

extern "C" size_t pcuda_linear_bwd_w_workspace_size(int m, int k, int n) {
  const long long out_elems = (long long)n * k;
  if (m >= 2048 && out_elems <= 65536) {
    int ks = m / 256;
    if (ks > 64) ks = 64;
    return (size_t)(ks + 1) * out_elems * sizeof(float);
  }
  return 0;
}

extern "C" int pcuda_linear_bwd_w(const float* dy, const float* x, float* dw, float* db, int m, int k, int n,
                                  int accumulate, void* workspace, size_t workspace_bytes, pcuda_stream_t s) {
  if (!dy || !x || !dw) PCUDA_FAIL(PCUDA_E_BADARG, "linear_bwd_w: null pointer");
  // dw[j][c] = sum_i dy[i][j] x[i][c]: the reduction runs over the batch rows.  A long reduction into
  // a small output (the point head's Linear(121,3) sees 300*B rows) is split into K-slices that run as
  // "batches" of the strided GEMM into the caller's workspace, then summed in a fixed order.
  const long long out_elems = (long long)n * k;
  const size_t need = pcuda_linear_bwd_w_workspace_size(m, k, n);
  int rc;
  if (need > 0 && workspace && workspace_bytes >= need) {
    float* slab = (float*)workspace;
    int ks = m / 256;
    if (ks > 64) ks = 64;
    const int mk = (m + ks - 1) / ks;       // rows per slice
    const int full = m / mk;                // complete slices run batched, the ragged tail as one more launch
    GemmParams pb = {dy, (long long)mk * n, 1, n, x, (long long)mk * k, k, 1, slab, out_elems, k, 1, nullptr, n, k, mk, 0};
    rc = launch_gemm(pb, full, (hipStream_t)s);
    if (rc) return rc;
    int nsl = full;
    if (m - full * mk > 0) {
      GemmParams pt = {dy + (long long)full * mk * n, 0, 1, n, x + (long long)full * mk * k, 0, k, 1,
                       slab + (long long)full * out_elems, 0, k, 1, nullptr, n, k, m - full * mk, 0};
      rc = launch_gemm(pt, 1, (hipStream_t)s);
      if (rc) return rc;
      ++nsl;
    }
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(out_elems, 256)), dim3(256), 0, (hipStream_t)s,
                       (const float*)slab, out_elems, nsl, dw, accumulate);
    PCUDA_CHECK_LAUNCH("splitk_reduce_kernel");
  } else {
    GemmParams p = {dy, 0, 1, n, x, 0, k, 1, dw, 0, k, 1, nullptr, n, k, m, accumulate};
    rc = launch_gemm(p, 1, (hipStream_t)s);
    if (rc) return rc;
  }
  if (db) {
    int cp = 1;
    while (cp < n && cp < 64) cp <<= 1;
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(n, cp)), dim3(256), 0, (hipStream_t)s, dy, m, n, db, accumulate, cp);
    PCUDA_CHECK_LAUNCH("colsum_kernel");
  }
  return PCUDA_OK;
}

extern "C" int pcuda_bmm(const float* a, const float* bmat, float* c, int batch, int m, int k, int n, int ta, int tb,
                         int accumulate, pcuda_stream_t s) {
  if (!a || !bmat || !c) PCUDA_FAIL(PCUDA_E_BADARG, "bmm: null pointer");
  GemmParams p;
  p.a = a; p.a_sb = (long long)m * k; p.a_si = ta ? 1 : k; p.a_sl = ta ? m : 1;
  p.b = bmat; p.b_sb = (long long)k * n; p.b_sl = tb ? 1 : n; p.b_sj = tb ? k : 1;
  p.c = c; p.c_sb = (long long)m * n; p.c_si = n; p.c_sj = 1;
  p.bias = nullptr; p.m = m; p.n = n; p.k = k; p.accumulate = accumulate;
  return launch_gemm(p, batch, (hipStream_t)s);
}

extern "C" int pcuda_max_points_fwd(const float* x, int b, int c, int l, float* y, int* idx, pcuda_stream_t s) {
  if (!x || !y || !idx || b <= 0 || c <= 0 || l <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "max_points_fwd: bad arguments");
  const int rows = b * c;
  hipLaunchKernelGGL(max_points_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)s, x, rows, l, y, idx);
  PCUDA_CHECK_LAUNCH("max_points_fwd_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_max_points_bwd(const float* dy, const int* idx, int b, int c, int l, float* dx, pcuda_stream_t s) {
  if (!dy || !idx || !dx || b <= 0 || c <= 0 || l <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "max_points_bwd: bad arguments");
  const long long total = (long long)b * c * l;
  const int blocks = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
  hipLaunchKernelGGL(max_points_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, dy, idx, b * c, l, dx);
  PCUDA_CHECK_LAUNCH("max_points_bwd_kernel");
  return PCUDA_OK;
}
