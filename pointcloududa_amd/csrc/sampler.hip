// Mask -> surface point cloud sampler (utils/npy2point.py:7-18 graipher; :101-125
// npy2point_datagenerator).  Integer / index work: results are bit-exact with the numpy oracle.
//   surface_vertices: canonical (build-defined, see oracle/sampler.py) vertex list
//   fps:              farthest point sampling, float64, no FMA contraction, first-occurrence argmax
#include "common.h"

// one workgroup per mask; thread t owns row t (+256, ...): count, LDS prefix, ordered write
__global__ __launch_bounds__(256) void surface_vertices_kernel(const uint8_t* __restrict__ mask, int h, int w,
                                                               int* __restrict__ verts, int max_verts,
                                                               int* __restrict__ counts) {
  extern __shared__ int rowcnt[];   // h + 1
  const int b = blockIdx.x;
  const uint8_t* m = mask + (long long)b * h * w;
  auto fg = [&](int y, int x) -> bool { return y >= 0 && y < h && x >= 0 && x < w && m[(long long)y * w + x] > 0; };
  for (int y = threadIdx.x; y < h; y += 256) {
    int c = 0;
    for (int x = 0; x < w; ++x)
      if (!fg(y, x) && (fg(y - 1, x) || fg(y + 1, x) || fg(y, x - 1) || fg(y, x + 1))) ++c;
    rowcnt[y] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // h <= a few hundred: serial exclusive scan is negligible
    int run = 0;
    for (int y = 0; y < h; ++y) { const int c = rowcnt[y]; rowcnt[y] = run; run += c; }
    rowcnt[h] = run;
  }
  __syncthreads();
  const int total = rowcnt[h];
  int* out = verts + (long long)b * max_verts * 3;
  for (int y = threadIdx.x; y < h; y += 256) {
    int pos = rowcnt[y];
    for (int x = 0; x < w; ++x)
      if (!fg(y, x) && (fg(y - 1, x) || fg(y + 1, x) || fg(y, x - 1) || fg(y, x + 1))) {
        for (int z = 0; z < 3; ++z) {
          const int o = z * total + pos;
          if (o < max_verts) { out[o * 3 + 0] = z; out[o * 3 + 1] = y; out[o * 3 + 2] = x; }
        }
        ++pos;
      }
  }
  if (threadIdx.x == 0) counts[b] = 3 * total;
}

// Marching-cubes-order mode (oracle/sampler.py: surface_vertices_mc; parity unpinned): cells of the 3-slice stack in
// (slice, row, column)-nested order, per cell the edges 6, 5, 10, 0, 1, 2, 3, 4, 7, 8, 9, 11 (Bourke numbering), an edge
// is emitted by the first cell that contains it, one vertex per crossing edge at its background end.
// One workgroup per mask; thread t owns the cell rows (i, j) = t, t + 256, ...: count, LDS prefix, ordered write.
__constant__ signed char MC_CORNER[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
__constant__ signed char MC_EDGE[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
__constant__ signed char MC_ORDER[12] = {6, 5, 10, 0, 1, 2, 3, 4, 7, 8, 9, 11};
// bit 0 / 1 / 2: the edge is new only if i / j / k == 0 (all named indices); 6, 5, 10: always
__constant__ signed char MC_NEED0[12] = {6, 4, 4, 5, 2, 0, 0, 1, 3, 2, 0, 1};

__global__ __launch_bounds__(256) void surface_vertices_mc_kernel(const uint8_t* __restrict__ mask, int h, int w,
                                                                  int* __restrict__ verts, int max_verts,
                                                                  int* __restrict__ counts) {
  extern __shared__ int rowcnt[];   // 2 * (h - 1) + 1
  const int b = blockIdx.x;
  const uint8_t* m = mask + (long long)b * h * w;
  const int nj = h - 1, nk = w - 1, nrows = 2 * nj;
  int* out = verts + (long long)b * max_verts * 3;
  // pass 0 counts, pass 1 writes
  for (int pass = 0; pass < 2; ++pass) {
    for (int r = threadIdx.x; r < nrows; r += 256) {
      const int i = r / nj, j = r - i * nj;
      int pos = pass ? rowcnt[r] : 0;
      for (int k = 0; k < nk; ++k) {
        const bool f00 = m[(long long)j * w + k] > 0, f01 = m[(long long)j * w + k + 1] > 0;
        const bool f10 = m[(long long)(j + 1) * w + k] > 0, f11 = m[(long long)(j + 1) * w + k + 1] > 0;
        if (f00 == f01 && f00 == f10 && f00 == f11) continue;      // (the slices are copies: no crossing edge in this cell)
        const int at0 = (i == 0 ? 1 : 0) | (j == 0 ? 2 : 0) | (k == 0 ? 4 : 0);
        for (int q = 0; q < 12; ++q) {
          const int e = MC_ORDER[q];
          const signed char* ca = MC_CORNER[MC_EDGE[e][0]];
          const signed char* cb = MC_CORNER[MC_EDGE[e][1]];
          const bool fa = ca[1] ? (ca[2] ? f11 : f10) : (ca[2] ? f01 : f00);
          const bool fb = cb[1] ? (cb[2] ? f11 : f10) : (cb[2] ? f01 : f00);
          if (fa == fb || (MC_NEED0[e] & ~at0)) continue;
          if (pass && pos < max_verts) {
            const signed char* c = fa ? cb : ca;                   // the background end
            out[pos * 3 + 0] = i + c[0]; out[pos * 3 + 1] = j + c[1]; out[pos * 3 + 2] = k + c[2];
          }
          ++pos;
        }
      }
      if (!pass) rowcnt[r] = pos;
    }
    __syncthreads();
    if (!pass) {
      if (threadIdx.x == 0) {
        int run = 0;
        for (int r = 0; r < nrows; ++r) { const int c = rowcnt[r]; rowcnt[r] = run; run += c; }
        rowcnt[nrows] = run;
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) counts[b] = rowcnt[nrows];
}

// one workgroup per cloud.  dist lives in LDS (npts_max doubles).
__global__ __launch_bounds__(256) void fps_kernel(const double* __restrict__ pts, const int* __restrict__ counts,
                                                  const int* __restrict__ first, int npts_max, int k,
                                                  int* __restrict__ idx) {
  extern __shared__ double dist[];   // npts_max
  __shared__ double rv[4];
  __shared__ int ri[4];
  __shared__ int cur_s;
  const int b = blockIdx.x;
  const int n = counts[b];
  const double* P = pts + (long long)b * npts_max * 3;
  int* out = idx + (long long)b * k;
  if (n <= 0) {
    for (int i = threadIdx.x; i < k; i += 256) out[i] = -1;
    return;
  }
  int cur = first[b] % n;
  if (cur < 0) cur += n;
  if (threadIdx.x == 0) out[0] = cur;
  for (int i = threadIdx.x; i < n; i += 256) dist[i] = INFINITY;
  __syncthreads();
  for (int it = 1; it < k; ++it) {
    const double c0 = P[cur * 3 + 0], c1 = P[cur * 3 + 1], c2 = P[cur * 3 + 2];
    double best = -1.0;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) {
      const double d0 = c0 - P[i * 3 + 0], d1 = c1 - P[i * 3 + 1], d2 = c2 - P[i * 3 + 2];
      // ((d0^2 + d1^2) + d2^2) with separate roundings, as numpy's (..)**2 .sum(axis=1)
      const double s = __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
      const double nd = fmin(dist[i], s);
      dist[i] = nd;
      if (nd > best) { best = nd; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = best; ri[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double bv = rv[0];
      int bj = ri[0];
      for (int q = 1; q < 4; ++q)
        if (rv[q] > bv || (rv[q] == bv && ri[q] < bj)) { bv = rv[q]; bj = ri[q]; }
      cur_s = bj;
      out[it] = bj;
    }
    __syncthreads();
    cur = cur_s;
  }
}

extern "C" int pcuda_surface_vertices(const uint8_t* mask, int b, int h, int w, int* verts, int max_verts,
                                      int* counts, void* workspace, size_t workspace_bytes, pcuda_stream_t s) {
  (void)workspace; (void)workspace_bytes;
  if (!mask || !verts || !counts || b <= 0 || h <= 0 || w <= 0 || max_verts <= 0 || h > 8192)
    PCUDA_FAIL(PCUDA_E_BADARG, "surface_vertices: bad arguments");
  hipLaunchKernelGGL(surface_vertices_kernel, dim3(b), dim3(256), (h + 1) * sizeof(int), (hipStream_t)s, mask, h, w,
                     verts, max_verts, counts);
  PCUDA_CHECK_LAUNCH("surface_vertices_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_surface_vertices_mc(const uint8_t* mask, int b, int h, int w, int* verts, int max_verts,
                                         int* counts, pcuda_stream_t s) {
  if (!mask || !verts || !counts || b <= 0 || h <= 1 || w <= 1 || max_verts <= 0 || h > 8192)
    PCUDA_FAIL(PCUDA_E_BADARG, "surface_vertices_mc: bad arguments");
  hipLaunchKernelGGL(surface_vertices_mc_kernel, dim3(b), dim3(256), (2 * (h - 1) + 1) * sizeof(int), (hipStream_t)s, mask,
                     h, w, verts, max_verts, counts);
  PCUDA_CHECK_LAUNCH("surface_vertices_mc_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_fps(const double* pts, const int* counts, const int* first, int b, int npts_max, int k, int* idx,
                         pcuda_stream_t s) {
  if (!pts || !counts || !first || !idx || b <= 0 || npts_max <= 0 || k <= 0)
    PCUDA_FAIL(PCUDA_E_BADARG, "fps: bad arguments");
  const size_t lds = (size_t)npts_max * sizeof(double);
  if (lds > 150 * 1024) PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "fps: at most %d points per cloud", (int)(150 * 1024 / 8));
  static DeviceOnce lds_opt;
  if (const unsigned long long devbit = lds > 32 * 1024 ? lds_opt.pending() : 0ull) {
    if (hipFuncSetAttribute((const void*)fps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      PCUDA_FAIL(PCUDA_E_LAUNCH, "fps: cannot raise dynamic LDS");
    lds_opt.mark(devbit);
  }
  hipLaunchKernelGGL(fps_kernel, dim3(b), dim3(256), lds, (hipStream_t)s, pts, counts, first, npts_max, k, idx);
  PCUDA_CHECK_LAUNCH("fps_kernel");
  return PCUDA_OK;
}
