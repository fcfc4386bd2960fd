// k = 1 Conv1d of the point-cloud discriminator (PointNetCls.py:26-28,76-78,116-131: torch.nn.Conv1d(cin, cout, 1)
// on [B][C][L] point clouds) in EXACT fp32 on the matrix cores: v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fma
// chain, bit for bit (no operand rounding), at the fp32 vector peak.  Rounds 1-2 ran these layers through the
// bf16x3 image convolution (32-deep chunks, 2^-17 per product); the network behind them -- a max over 300 points
// followed by BatchNorm1d over a batch of 4-16 clouds -- turns that noise into percents of its output (the "spread"
// the goldens carry), so the parity bar of 1e-3 was out of reach there.  The layers are 0.17-0.59 GFLOP per cloud:
// fp32 MFMA costs nothing against the 32-deep bf16 chunks they ran on.
//
//   forward  Y[b][co][l] = sum_ci W[co][ci] X[b][ci][l] + bias[co]      (+ BatchNorm partial sums per column tile)
//   dgrad    dX[b][ci][l] = sum_co W[co][ci] dY[b][co][l]
//   wgrad    dW[co][ci] (+)= sum_{b,l} dY[b][co][l] X[b][ci][l],  db[co] (+)= sum_{b,l} dY[b][co][l]
//
// One GEMM shape serves forward and dgrad (C[M x N] = A[M x K] B[K x N], N = B*L flattened, A = W or W^T), a second
// one the weight gradient (the reduction runs over the B*L columns: split-K slabs + the fixed-order reduce of
// conv_wgrad.hip, deterministic).
#include "common.h"
#include "conv_host.h"

namespace {

constexpr int BN = 64;          // columns (points) per workgroup tile
constexpr int BK = 16;          // reduction depth per stage (forward / dgrad)
constexpr int WK = 32;          // columns per stage of the weight gradient (its reduction axis)

struct GemmP {
  const float* w;   // [cout][cin]
  const float* x;   // B operand: [b][K][L]
  float* y;         // [b][M][L]
  const float* bias;
  float* partials;  // [ntiles][M][2] or NULL
  int M, K, L, ncols, cin;
};

// C[i][j] = sum_k A[i][k] B[k][j];  TA = false: A[i][k] = w[i*cin + k] (forward), true: A[i][k] = w[k*cin + i] (dgrad).
// VEC: 16-byte global loads (L, K and M multiples of 4, 16-byte aligned tensors: every PointNetCls layer but the 3- and
// 8-channel ones) and a 32-deep stage; otherwise dword loads with bounds checks and a 16-deep stage.
template <int MB, bool TA, bool VEC>
__global__ __launch_bounds__(256) void c1d_gemm_kernel(const GemmP p) {
  constexpr int BM = 64 * MB;       // rows per workgroup: wave (w >> 1) owns MB row blocks of 32, wave (w & 1) a column half
  constexpr int KS = VEC ? 32 : BK; // reduction depth per stage
  constexpr int PA = (VEC && TA) ? BM + 4 : BM + 1, PB = VEC ? BN + 4 : BN + 1;
  __shared__ __attribute__((aligned(16))) float As[KS][PA];
  __shared__ __attribute__((aligned(16))) float Bs[KS][PB];
  __shared__ float red[2][BM][2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j0 = blockIdx.x * BN, i0 = blockIdx.y * BM;
  const int ch = w & 1, rp = w >> 1;
  const int r = lane & 31, h = lane >> 5;

  // ---- scalar loaders (VEC = false): B lane column jj, k rows kb + 4q; A see below
  const int jj = tid & 63, kb = tid >> 6;
  const int jcol = j0 + jj;
  const bool jok = jcol < p.ncols;
  const int jb = jok ? jcol / p.L : 0, jl = jok ? jcol - jb * p.L : 0;
  // ---- vector loaders (VEC = true): B lane = 4 columns (tid & 15), k rows (tid >> 4) + 16q;
  //      A, TA = false: 4 k's (tid & 7), rows (tid >> 3) + 32q (transposed scalar LDS stores);
  //      A, TA = true : 4 rows (tid % (BM / 4)), k rows tid / (BM / 4) + (1024 / BM) q (16-byte LDS stores)
  const int vj = (tid & 15) * 4, vk = tid >> 4;
  const int vcol = j0 + vj;
  const bool vok = vcol < p.ncols;          // (ncols % 4 == 0: a quad is inside or outside as a whole, and inside one batch row)
  const int vb = vok ? vcol / p.L : 0, vl = vok ? vcol - vb * p.L : 0;
  constexpr int AL = VEC ? BM * 32 / 1024 : BM * BK / 256;      // A loads per thread and stage (float4 / float)
  constexpr int BL = VEC ? 2 : 4;
  float av[VEC ? 1 : AL], bv[VEC ? 1 : BL];
  f32x4 av4[VEC ? AL : 1], bv4[VEC ? BL : 1];
  // VMEM address rule (common.h): every load goes through a wave-uniform base + a 32-bit byte offset of its own that
  // stays alive (PCUDA_KEEP behind the commit that consumes the data); lanes outside the problem read offset 0 and are
  // zeroed afterwards -- no load sits under a per-lane branch
  unsigned aoff[AL], boff[BL];
  bool aok[AL], bok[BL];
  const char* const wb = (const char*)p.w;
  const char* const xb = (const char*)p.x;
  const unsigned xq_off = (unsigned)(((long long)vb * p.K * p.L + vl) * 4), xc_off = (unsigned)(((long long)jb * p.K * p.L + jl) * 4);
  auto load = [&](int k0) {
    if (VEC) {
#pragma unroll
      for (int q = 0; q < BL; ++q) {
        const int k = k0 + vk + 16 * q;
        bok[q] = vok && k < p.K;
        boff[q] = bok[q] ? xq_off + (unsigned)(k * p.L) * 4u : 0u;
        bv4[q] = *(const f32x4*)(xb + boff[q]);
      }
#pragma unroll
      for (int q = 0; q < AL; ++q) {
        if (!TA) {
          const int k = k0 + (tid & 7) * 4, gi = i0 + (tid >> 3) + 32 * q;
          aok[q] = gi < p.M && k < p.K;
          aoff[q] = aok[q] ? (unsigned)(gi * p.cin + k) * 4u : 0u;
        } else {
          const int gi = i0 + (tid % (BM / 4)) * 4, k = k0 + tid / (BM / 4) + (1024 / BM) * q;
          aok[q] = gi < p.M && k < p.K;
          aoff[q] = aok[q] ? (unsigned)(k * p.cin + gi) * 4u : 0u;
        }
        av4[q] = *(const f32x4*)(wb + aoff[q]);
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < BL; ++q) {
      const int k = k0 + kb + 4 * q;
      bok[q] = jok && k < p.K;
      boff[q] = bok[q] ? xc_off + (unsigned)(k * p.L) * 4u : 0u;
      bv[q] = *(const float*)(xb + boff[q]);
    }
#pragma unroll
    for (int q = 0; q < AL; ++q) {
      int i, k;
      if (!TA) { k = tid & 15; i = (tid >> 4) + 16 * q; }
      else { i = (tid & 63) + 64 * (q / 4); k = (tid >> 6) + 4 * (q & 3); }
      const int gi = i0 + i, gk = k0 + k;
      aok[q] = gi < p.M && gk < p.K;
      aoff[q] = aok[q] ? (unsigned)(TA ? gk * p.cin + gi : gi * p.cin + gk) * 4u : 0u;
      av[q] = *(const float*)(wb + aoff[q]);
    }
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto commit = [&]() {
    if (VEC) {
#pragma unroll
      for (int q = 0; q < BL; ++q) *(f32x4*)&Bs[vk + 16 * q][vj] = bok[q] ? bv4[q] : zero4;
#pragma unroll
      for (int q = 0; q < AL; ++q) {
        const f32x4 a4 = aok[q] ? av4[q] : zero4;
        if (!TA) {
          const int k = (tid & 7) * 4, i = (tid >> 3) + 32 * q;
#pragma unroll
          for (int e = 0; e < 4; ++e) As[k + e][i] = a4[e];
        } else {
          *(f32x4*)&As[tid / (BM / 4) + (1024 / BM) * q][(tid % (BM / 4)) * 4] = a4;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < BL; ++q) Bs[kb + 4 * q][jj] = bok[q] ? bv[q] : 0.f;
#pragma unroll
      for (int q = 0; q < AL; ++q) {
        int i, k;
        if (!TA) { k = tid & 15; i = (tid >> 4) + 16 * q; }
        else { i = (tid & 63) + 64 * (q / 4); k = (tid >> 6) + 4 * (q & 3); }
        As[k][i] = aok[q] ? av[q] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < BL; ++q) PCUDA_KEEP(boff[q]);
#pragma unroll
    for (int q = 0; q < AL; ++q) PCUDA_KEEP(aoff[q]);
  };

  f32x16 acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

  load(0);
  for (int k0 = 0; k0 < p.K; k0 += KS) {
    __syncthreads();          // the previous stage's MFMAs are done with the tiles
    commit();
    __syncthreads();
    if (k0 + KS < p.K) load(k0 + KS);      // in flight during this stage's MFMAs
    const int npair = min(KS, p.K - k0 + 1) >> 1;      // k pairs that hold data (the rest of the tile is zero)
    for (int s = 0; s < npair; ++s) {
      const float b = Bs[2 * s + h][ch * 32 + r];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const float a = As[2 * s + h][(rp * MB + m) * 32 + r];
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
      }
    }
  }

  // epilogue: bias, store, BatchNorm partial sums (sum, sum of squares over the tile's valid columns, per row)
  const int oc = j0 + ch * 32 + r;
  const bool ocok = oc < p.ncols;
  const int ob = ocok ? oc / p.L : 0, ol = ocok ? oc - ob * p.L : 0;
  float* ycol = p.y + ((long long)ob * p.M * p.L + ol);
#pragma unroll
  for (int m = 0; m < MB; ++m) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int il = (rp * MB + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const int gi = i0 + il;
      const bool iok = gi < p.M;
      float v = acc[m][e] + ((p.bias && iok) ? p.bias[gi] : 0.f);
      if (iok && ocok) ycol[(long long)gi * p.L] = v;
      if (p.partials) {
        v = ocok ? v : 0.f;
        const float s1 = half_wave_sum(v), s2 = half_wave_sum(v * v);
        if (r == 0) { red[ch][il][0] = s1; red[ch][il][1] = s2; }
      }
    }
  }
  if (p.partials) {
    __syncthreads();
    if (tid < BM && i0 + tid < p.M) {
      float* dst = p.partials + ((long long)blockIdx.x * p.M + i0 + tid) * 2;
      dst[0] = red[0][tid][0] + red[1][tid][0];
      dst[1] = red[0][tid][1] + red[1][tid][1];
    }
  }
}

struct WgradP {
  const float* dy;  // [b][M][L]
  const float* x;   // [b][N][L]
  float* partial;   // [ksplit][M*N]
  float* db_partial;   // [ksplit][M] or NULL
  int M, N, L, ncols, steps_per_slice;
};

// partial[ks][i][n] = sum over the slice's columns j of dy[i][j] x[n][j]
template <int MB>
__global__ __launch_bounds__(256) void c1d_wgrad_kernel(const WgradP p) {
  constexpr int BM = 64 * MB;
  constexpr int PA = BM + 1, PB = BN + 1;
  __shared__ float As[WK][PA];
  __shared__ float Bs[WK][PB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * BN, i0 = blockIdx.y * BM, ks = blockIdx.z;
  const int ch = w & 1, rp = w >> 1;
  const int r = lane & 31, h = lane >> 5;
  const int kk = tid & 31, r0 = tid >> 5;
  const bool do_db = p.db_partial != nullptr && blockIdx.x == 0;
  constexpr int AL = BM / 8;
  float av[AL], bv[8], dbacc[AL];
#pragma unroll
  for (int q = 0; q < AL; ++q) dbacc[q] = 0.f;
  const int step_lo = ks * p.steps_per_slice;
  const int step_hi = min(step_lo + p.steps_per_slice, (p.ncols + WK - 1) / WK);
  // (VMEM address rule, common.h: uniform base + a kept 32-bit offset per load, unconditional loads, zeros selected after)
  unsigned aoff[AL], boff[8];
  bool aok[AL], bok[8];
  const char* const ab = (const char*)p.dy;
  const char* const bb = (const char*)p.x;
  auto load = [&](int step) {
    const int j = step * WK + kk;
    const bool jok = j < p.ncols;
    const int b = jok ? j / p.L : 0, l = jok ? j - b * p.L : 0;
    const unsigned ac = (unsigned)(((long long)b * p.M * p.L + l) * 4), bc = (unsigned)(((long long)b * p.N * p.L + l) * 4);
#pragma unroll
    for (int q = 0; q < AL; ++q) {
      const int gi = i0 + r0 + 8 * q;
      aok[q] = jok && gi < p.M;
      aoff[q] = aok[q] ? ac + (unsigned)(gi * p.L) * 4u : 0u;
      av[q] = *(const float*)(ab + aoff[q]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int gn = n0 + r0 + 8 * q;
      bok[q] = jok && gn < p.N;
      boff[q] = bok[q] ? bc + (unsigned)(gn * p.L) * 4u : 0u;
      bv[q] = *(const float*)(bb + boff[q]);
    }
  };
  f32x16 acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
  if (step_lo < step_hi) load(step_lo);
  for (int step = step_lo; step < step_hi; ++step) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < AL; ++q) {
      const float a = aok[q] ? av[q] : 0.f;
      As[kk][r0 + 8 * q] = a;
      if (do_db) dbacc[q] += a;
      PCUDA_KEEP(aoff[q]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { Bs[kk][r0 + 8 * q] = bok[q] ? bv[q] : 0.f; PCUDA_KEEP(boff[q]); }
    __syncthreads();
    if (step + 1 < step_hi) load(step + 1);
#pragma unroll 4
    for (int s = 0; s < WK / 2; ++s) {
      const float b = Bs[2 * s + h][ch * 32 + r];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const float a = As[2 * s + h][(rp * MB + m) * 32 + r];
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
      }
    }
  }
  float* slab = p.partial + (long long)ks * p.M * p.N;
  const int gn = n0 + ch * 32 + r;
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int gi = i0 + (rp * MB + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (gi < p.M && gn < p.N) slab[(long long)gi * p.N + gn] = acc[m][e];
    }
  if (do_db) {
#pragma unroll
    for (int q = 0; q < AL; ++q) {
      const float s = half_wave_sum(dbacc[q]);      // over the 32 column lanes that share a row
      const int gi = i0 + r0 + 8 * q;
      if (kk == 0 && gi < p.M) p.db_partial[(long long)ks * p.M + gi] = s;
    }
  }
}

// 16-byte loads: K (reduction rows of the B operand), M (output rows) and L multiples of 4, weight rows of 4k floats,
// 16-byte aligned tensors
bool vec_ok(const float* xb, const float* w, const float* y, int K, int M, int l, int wrow) {
  return (K & 3) == 0 && (M & 3) == 0 && (l & 3) == 0 && (wrow & 3) == 0 && (((uintptr_t)xb | (uintptr_t)w | (uintptr_t)y) & 15) == 0;
}

bool dims_ok(int b, int cin, int cout, int l) {
  // (32-bit byte offsets inside every tensor)
  return b > 0 && cin > 0 && cout > 0 && l > 0 && (long long)b * l < (1ll << 30) &&
         (long long)b * (cin > cout ? cin : cout) * l < (1ll << 29) && (long long)cin * cout < (1ll << 29);
}

int wgrad_ksplit(int b, int cin, int cout, int l, int mb) {
  const int steps = cdiv((long long)b * l, WK);
  const int tiles = cdiv(cout, 64 * mb) * cdiv(cin, BN);
  int ks = 768 / tiles;
  if (ks > steps / 2) ks = steps / 2;
  if (ks < 1) ks = 1;
  while (ks > 1 && (long long)ks * cout * cin * 4 > (48ll << 20)) ks >>= 1;
  return ks;
}
int wgrad_mb(int cout) { return cout > 64 ? 2 : 1; }

}  // namespace

extern "C" int pcuda_conv1d_k1_fwd_tiles(int b, int l) {
  if (b <= 0 || l <= 0) return 0;
  return cdiv((long long)b * l, BN);
}

extern "C" int pcuda_conv1d_k1_fwd(const float* x, const float* w, const float* bias, float* y, int b, int cin, int cout,
                                   int l, float* bn_partials, pcuda_stream_t s_) {
  if (!x || !w || !y || !dims_ok(b, cin, cout, l)) PCUDA_FAIL(PCUDA_E_BADARG, "conv1d_k1_fwd: bad arguments");
  hipStream_t s = (hipStream_t)s_;
  GemmP p = {w, x, y, bias, bn_partials, cout, cin, l, b * l, cin};
  char tag[96];
  snprintf(tag, sizeof(tag), "conv1d f32 fwd n%d cin%d cout%d l%d", b, cin, cout, l);
  ProfScope prof(PCUDA_FAM_DENSE_F32, 2.0 * b * l * (double)cin * cout, s, tag);
  const bool vec = vec_ok(x, w, y, cin, cout, l, cin);
  const dim3 g2(cdiv(p.ncols, BN), cdiv(cout, 128)), g1(cdiv(p.ncols, BN), cdiv(cout, 64));
  if (cout > 64 && (long long)g2.x * g2.y >= 512) {
    if (vec) hipLaunchKernelGGL((c1d_gemm_kernel<2, false, true>), g2, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((c1d_gemm_kernel<2, false, false>), g2, dim3(256), 0, s, p);
  } else {
    if (vec) hipLaunchKernelGGL((c1d_gemm_kernel<1, false, true>), g1, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((c1d_gemm_kernel<1, false, false>), g1, dim3(256), 0, s, p);
  }
  PCUDA_CHECK_LAUNCH("c1d_gemm_kernel(fwd)");
  return PCUDA_OK;
}

extern "C" int pcuda_conv1d_k1_dgrad(const float* dy, const float* w, float* dx, int b, int cin, int cout, int l,
                                     pcuda_stream_t s_) {
  if (!dy || !w || !dx || !dims_ok(b, cin, cout, l)) PCUDA_FAIL(PCUDA_E_BADARG, "conv1d_k1_dgrad: bad arguments");
  hipStream_t s = (hipStream_t)s_;
  GemmP p = {w, dy, dx, nullptr, nullptr, cin, cout, l, b * l, cin};
  char tag[96];
  snprintf(tag, sizeof(tag), "conv1d f32 dgrad n%d cin%d cout%d l%d", b, cin, cout, l);
  ProfScope prof(PCUDA_FAM_DENSE_F32, 2.0 * b * l * (double)cin * cout, s, tag);
  const bool vec = vec_ok(dy, w, dx, cout, cin, l, cin);
  const dim3 g2(cdiv(p.ncols, BN), cdiv(cin, 128)), g1(cdiv(p.ncols, BN), cdiv(cin, 64));
  // (128-row tiles only where they still fill the chip twice: 1024 -> 128 channels on 9600 points is 150 of them)
  if (cin > 64 && (long long)g2.x * g2.y >= 512) {
    if (vec) hipLaunchKernelGGL((c1d_gemm_kernel<2, true, true>), g2, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((c1d_gemm_kernel<2, true, false>), g2, dim3(256), 0, s, p);
  } else {
    if (vec) hipLaunchKernelGGL((c1d_gemm_kernel<1, true, true>), g1, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((c1d_gemm_kernel<1, true, false>), g1, dim3(256), 0, s, p);
  }
  PCUDA_CHECK_LAUNCH("c1d_gemm_kernel(dgrad)");
  return PCUDA_OK;
}

extern "C" size_t pcuda_conv1d_k1_wgrad_workspace_size(int b, int cin, int cout, int l) {
  if (!dims_ok(b, cin, cout, l)) return 0;
  const int ks = wgrad_ksplit(b, cin, cout, l, wgrad_mb(cout));
  return ((size_t)ks * cout * cin + (size_t)ks * cout) * sizeof(float) + 256;
}

extern "C" int pcuda_conv1d_k1_wgrad(const float* x, const float* dy, float* dw, float* db, int b, int cin, int cout, int l,
                                     int accumulate, void* workspace, size_t workspace_bytes, pcuda_stream_t s_) {
  if (!x || !dy || !dw || !dims_ok(b, cin, cout, l)) PCUDA_FAIL(PCUDA_E_BADARG, "conv1d_k1_wgrad: bad arguments");
  if (!workspace || workspace_bytes < pcuda_conv1d_k1_wgrad_workspace_size(b, cin, cout, l))
    PCUDA_FAIL(PCUDA_E_WORKSPACE, "conv1d_k1_wgrad: workspace too small");
  hipStream_t s = (hipStream_t)s_;
  const int mb = wgrad_mb(cout);
  const int ks = wgrad_ksplit(b, cin, cout, l, mb);
  const int steps = cdiv((long long)b * l, WK);
  WgradP p;
  p.dy = dy; p.x = x; p.partial = (float*)workspace;
  p.db_partial = db ? (float*)workspace + (size_t)ks * cout * cin : nullptr;
  p.M = cout; p.N = cin; p.L = l; p.ncols = b * l; p.steps_per_slice = cdiv(steps, ks);
  {
    char tag[96];
    snprintf(tag, sizeof(tag), "conv1d f32 wgrad n%d cin%d cout%d l%d ksplit%d", b, cin, cout, l, ks);
    ProfScope prof(PCUDA_FAM_DENSE_F32, 2.0 * b * l * (double)cin * cout, s, tag);
    const dim3 grid(cdiv(cin, BN), cdiv(cout, 64 * mb), ks);
    if (mb == 2) hipLaunchKernelGGL((c1d_wgrad_kernel<2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((c1d_wgrad_kernel<1>), grid, dim3(256), 0, s, p);
    PCUDA_CHECK_LAUNCH("c1d_wgrad_kernel");
  }
  return launch_wgrad_reduce(p.partial, (long long)cout * cin, ks, dw, accumulate, p.db_partial, cout, db, s);
}
