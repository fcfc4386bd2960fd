// Device side (and the templated launchers) of the forward / dgrad kernels; instantiated once per
// precision (conv_igemm_x3.hip, conv_igemm_bf16.hip) so the two halves compile in parallel.
#pragma once
#include "conv_device.h"
#include "conv_host.h"
#include "variants.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// forward / dgrad kernel.  256 threads = 4 waves; tile = CO_TILE rows x (128*NPB) logical pixels;
// wave w owns pixels [32*NPB*w, +32*NPB) (NPB 32-pixel MFMA column blocks) for all CO_BLKS row blocks.
// ------------------------------------------------------------------------------------------
template <bool X3, int CO_BLKS, bool CLAMP, int NPB>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p, const int x_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CO_TILE = 32 * CO_BLKS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;

  // XCD-aware block order: blocks that share an XCD (bid % 8) walk adjacent tiles, and the
  // co-tiles of one pixel tile are adjacent, so the haloed input tile is fetched once per L2.
  const unsigned bid = blockIdx.x, nwg = gridDim.x;
  const unsigned xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
  const unsigned L = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
  const int cot = L % p.n_co_tiles;
  const int pt = L / p.n_co_tiles;
  const int txi = pt % p.tiles_x;
  const int tmp = pt / p.tiles_x;
  const int tyi = tmp % p.tiles_y;
  const int n = tmp / p.tiles_y;
  const int TW = p.tw, TH = p.th, TPIX = TW * TH;
  const int y0 = tyi * TH, x0 = txi * TW;

  int oy0 = y0 * p.in_step + p.dy_min, ox0 = x0 * p.in_step + p.dx_min;
  int th = p.ih_t, tw = p.iw_t;
  if (CLAMP) {
    const int y1 = min(oy0 + th, p.in_h), x1 = min(ox0 + tw, p.in_w);
    oy0 = max(oy0, 0); ox0 = max(ox0, 0);
    th = max(y1 - oy0, 0); tw = max(x1 - ox0, 0);
  }
  const int npix = th * tw;

  constexpr int REC = IgRec<X3>::BYTES, RECV = REC / 16;
  unsigned char* Xhi = smem;
  unsigned char* Xlo = Xhi + IG_LO_OFF;
  unsigned char* Whi = smem + (size_t)x_cap * REC;
  unsigned char* Wlo = Whi + IG_LO_OFF;

  // per-lane pixel of each MFMA column block
  int pty[NPB], ptx[NPB], bbase[NPB];
  bool pvalid[NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    const int plr = w * (32 * NPB) + pb * 32 + r;
    pvalid[pb] = plr < TPIX;
    const int pl = min(plr, TPIX - 1);          // idle slots of a non-power-of-two tile read a valid pixel
    pty[pb] = IG_TY(pl, p.tmagic);
    ptx[pb] = pl - pty[pb] * TW;
    bbase[pb] = ((pty[pb] * p.in_step) * tw + ptx[pb] * p.in_step) * REC + h * 16;
  }

  f32x16 acc[CO_BLKS][NPB];
#pragma unroll
  for (int cb = 0; cb < CO_BLKS; ++cb)
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][pb][i] = 0.f;

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int cvalid = min(32, p.cin - chunk * 32);
    const int nks = cvalid > 16 ? 2 : 1;
    __syncthreads();   // every wave is done reading the previous chunk's X / W
    if (!(p.dbg & 1))
      stage_x_chunk<X3, 3, REC>(Xhi, Xlo, p.x, n, p.cin, chunk, p.in_h, p.in_w, p.in_shift, p.in_row, oy0, ox0, th, tw,
                        (cvalid + 7) >> 3, nks * 2, tid);
    if (CLAMP && tid < RECV)   // the all-zero record that out-of-image taps read
      *(uint4*)(Xhi + (size_t)npix * REC + tid * 16) = make_uint4(0, 0, 0, 0);
    for (int t0 = 0; t0 < p.ntaps; t0 += p.tg) {
      if (t0 > 0) __syncthreads();
      const int tgc = min(p.tg, p.ntaps - t0);
      {
        const long long slab = (long long)CO_TILE * (REC / 2);   // bf16 elements per tap
        const uint16_t* src = p.wpack + (((long long)cot * p.nchunks + chunk) * p.ntaps + t0) * slab;
        const int nvec = tgc * CO_TILE * RECV;               // 16-B vectors
        if (!(p.dbg & 8)) {
          wcopy<false, 4>(Whi, nullptr, (const uint4*)src, nullptr, nvec, 0, tid);
        }
      }
      __syncthreads();
      for (int tl = 0; tl < ((p.dbg & 2) ? 0 : tgc); ++tl) {
        const int t = t0 + tl;
        int baddr[NPB];
        if (CLAMP) {
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) {
            const int gy = (y0 + pty[pb]) * p.in_step + p.dy[t];
            const int gx = (x0 + ptx[pb]) * p.in_step + p.dx[t];
            const bool ok = ((unsigned)gy < (unsigned)p.in_h) & ((unsigned)gx < (unsigned)p.in_w);
            const int idx = ok ? (gy - oy0) * tw + (gx - ox0) : npix;
            baddr[pb] = idx * REC + h * 16;
          }
        } else {
          const int toff = ((p.dy[t] - p.dy_min) * tw + (p.dx[t] - p.dx_min)) * REC;
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) baddr[pb] = bbase[pb] + toff;
        }
        const int abase = (tl * CO_TILE + r) * REC + h * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (ks < nks) {
            bf16x8 ah[CO_BLKS], al[CO_BLKS], bh[NPB], bl[NPB];
#pragma unroll
            for (int cb = 0; cb < CO_BLKS; ++cb) {
              ah[cb] = lds_frag(Whi + abase + cb * 32 * REC + ks * 32);
              if (X3) al[cb] = lds_frag(Wlo + abase + cb * 32 * REC + ks * 32);
            }
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
              bh[pb] = lds_frag(Xhi + baddr[pb] + ks * 32);
              if (X3) bl[pb] = lds_frag(Xlo + baddr[pb] + ks * 32);
            }
#pragma unroll
            for (int cb = 0; cb < CO_BLKS; ++cb)
#pragma unroll
              for (int pb = 0; pb < NPB; ++pb) {
                if (X3) {
                  acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb], bh[pb], acc[cb][pb], 0, 0, 0);
                  acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bl[pb], acc[cb][pb], 0, 0, 0);
                }
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bh[pb], acc[cb][pb], 0, 0, 0);
              }
          }
        }
      }
    }
  }

  // ---- epilogue: bias + LeakyReLU, NCHW store (register i = one output channel, the 32 lanes of
  // a half-wave = 32 consecutive pixels), optional per-tile BatchNorm partial sums.
  __syncthreads();
  if (p.dbg & 4) {
    if (acc[0][0][0] == 123.456f) p.y.p1[0] = 1.f;   // keep the accumulators live
    return;
  }
  float* sred = (float*)smem;   // [4 waves][CO_TILE][2]
  // per-lane output pixel offsets (elements within a plane), computed once
  int poff[NPB];
  bool pok[NPB], pok1[NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    const int ly = y0 + pty[pb], lx = x0 + ptx[pb];
    pok[pb] = pvalid[pb] & (ly < p.lh) & (lx < p.lw);
    pok1[pb] = pok[pb] & (lx < p.lw2);   // (paired column classes: see igemm_pipe_kernel)
    poff[pb] = (ly * p.oy_mul + p.oy_off) * p.out_w + (lx * p.ox_mul + p.ox_off);
  }
  const int co0 = cot * CO_TILE;
  // fast path: the two half-waves (rows r0 and r0+4) of every register land in the same destination
  // tensor -> the plane base of row r0 is wave-uniform (SGPR) and lanes add a 32-bit offset
  const bool uni = !p.pair & ((p.y.c1 >= p.cout) | ((p.y.c1 & 7) == 0));
  float* const yb1 = p.y.p1 + (long long)n * p.y.sn1;
  float* const yb2 = p.y.p2 + (long long)n * p.y.sn2;
#pragma unroll
  for (int cb = 0; cb < CO_BLKS; ++cb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row0 = cb * 32 + (i & 3) + 8 * (i >> 2);   // wave-uniform
      const int row = row0 + 4 * h;
      const int co = co0 + row;
      const bool cok = co < p.cout;
      const float b = (cok && p.bias) ? p.bias[co] : 0.f;
      float* plane;
      if (uni) {
        const int cu = min(co0 + row0, p.cout - 1);
        float* base = (cu < p.y.c1) ? yb1 + (long long)cu * p.y.sc1 : yb2 + (long long)(cu - p.y.c1) * p.y.sc2;
        const long long hs = (cu < p.y.c1) ? p.y.sc1 : p.y.sc2;
        plane = base + (h ? 4 * hs : 0);
      } else {
        const int cc = min(co, p.cout - 1) >> p.pair;
        plane = (cc < p.y.c1) ? yb1 + (long long)cc * p.y.sc1 : yb2 + (long long)(cc - p.y.c1) * p.y.sc2;
      }
      const bool odd = (i & 1) & p.pair;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        float v = acc[cb][pb][i] + b;
        v = v > 0.f ? v : v * p.slope;
        float* dst = plane + poff[pb] + (odd ? 1 : 0);
        if (cok & (odd ? pok1[pb] : pok[pb])) {
          if (p.accumulate) v += *dst;
          if (p.mask_a) {   // (uniform) LeakyReLU backward of the layer in front: same element of the saved activation
            const float av = *(p.mask_a + (long long)n * p.mask_sn + (dst - yb1));
            v = av > 0.f ? v : v * p.mask_slope;
          }
          *dst = v;
          s1 += v;
          s2 += v * v;
        }
      }
      if (p.stats) {
        s1 = half_wave_sum(s1);
        s2 = half_wave_sum(s2);
        if (r == 0) {
          sred[(w * CO_TILE + row) * 2 + 0] = s1;
          sred[(w * CO_TILE + row) * 2 + 1] = s2;
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    if (tid < CO_TILE) {
      const int co = cot * CO_TILE + tid;
      if (co < p.cout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {   // fixed order: deterministic
          s1 += sred[(ww * CO_TILE + tid) * 2 + 0];
          s2 += sred[(ww * CO_TILE + tid) * 2 + 1];
        }
        p.stats[((long long)pt * p.cout + co) * 2 + 0] = s1;
        p.stats[((long long)pt * p.cout + co) * 2 + 1] = s2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Persistent, software-pipelined forward / dgrad kernel (the default).  Same tiling as igemm_kernel,
// but a workgroup walks a strided list of (tile, 32-channel chunk) stages and keeps the NEXT stage's
// input loads in flight (32*PF dwords per lane, in registers) while the current stage copies its
// weights, runs its MFMAs and stores its outputs.  With only 1-2 workgroups per CU (LDS-limited) the
// unpipelined kernel left HBM idle during compute and the matrix cores idle during staging.
// ------------------------------------------------------------------------------------------
struct TileGeom {
  int cot, pt, n, y0, x0, oy0, ox0, th, tw, npix;
};

__device__ __forceinline__ int ig_fdiv(int v, unsigned m) { return m ? (int)__umulhi((unsigned)v, m) : v; }

template <bool CLAMP, int NPB>
__device__ __forceinline__ TileGeom tile_decode(const IgemmParams& p, int L) {
  TileGeom g;
  g.pt = ig_fdiv(L, p.m_cot);
  g.cot = L - g.pt * p.n_co_tiles;
  const int tmp = ig_fdiv(g.pt, p.m_tx);
  const int txi = g.pt - tmp * p.tiles_x;
  g.n = ig_fdiv(tmp, p.m_ty);
  const int tyi = tmp - g.n * p.tiles_y;
  const int TW = p.tw, TH = p.th;
  g.y0 = tyi * TH; g.x0 = txi * TW;
  g.oy0 = g.y0 * p.in_step + p.dy_min; g.ox0 = g.x0 * p.in_step + p.dx_min;
  g.th = p.ih_t; g.tw = p.iw_t;
  if (CLAMP) {
    const int y1 = min(g.oy0 + g.th, p.in_h), x1 = min(g.ox0 + g.tw, p.in_w);
    g.oy0 = max(g.oy0, 0); g.ox0 = max(g.ox0, 0);
    g.th = max(y1 - g.oy0, 0); g.tw = max(x1 - g.ox0, 0);
  }
  g.npix = g.th * g.tw;
  return g;
}

// PCUDA_DBG bit 128: per-phase cycle sums of wave 0 of every workgroup (timing experiments only).  Compiled in only
// with -DPCUDA_CLK_DEBUG (make CLK=1): the eight 64-bit counters and the stamps cost scalar registers and pin the
// schedule around every phase boundary.
#ifdef PCUDA_CLK_DEBUG
#define DBG_CLK_DECL unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define DBG_CLK(i)                                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                \
  if (p.dbg & 128) {                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              \
    const unsigned long long now_ = __builtin_readcyclecounter();                   \
    clk[i] += now_ - tlast; tlast = now_;                                           \
  }                                                                                 \
  __builtin_amdgcn_sched_barrier(0);
// same, after forcing the accumulators (the MFMAs issued so far) to complete
#define DBG_CLK_ACC(i)                                                              \
  __builtin_amdgcn_sched_barrier(0);                                                \
  if (p.dbg & 128) {                                                                \
    asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[CO_BLKS - 1][NPB - 1][15]));  \
    const unsigned long long now_ = __builtin_readcyclecounter();                   \
    clk[i] += now_ - tlast; tlast = now_;                                           \
  }                                                                                 \
  __builtin_amdgcn_sched_barrier(0);
// (eight-wave kernel: one accumulator array index)
#define DBG_CLK_ACC8(i)                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                \
  if (p.dbg & 128) {                                                                \
    asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[CBW - 1][15]));                  \
    const unsigned long long now_ = __builtin_readcyclecounter();                   \
    clk[i] += now_ - tlast; tlast = now_;                                           \
  }                                                                                 \
  __builtin_amdgcn_sched_barrier(0);
#define DBG_CLK_FLUSH                                                               \
  if ((p.dbg & 128) && p.dbg_clk && tid == 0) {                                     \
    for (int i = 0; i < 8; ++i) atomicAdd(&p.dbg_clk[i], clk[i]);                   \
  }
#else
#define DBG_CLK_DECL
#define DBG_CLK(i) PCUDA_CLK_FENCE
#define DBG_CLK_ACC(i) PCUDA_CLK_FENCE
#define DBG_CLK_ACC8(i) PCUDA_CLK_FENCE
#define DBG_CLK_FLUSH
#ifdef PCUDA_CLK_KEEP_FENCES
#define PCUDA_CLK_FENCE __builtin_amdgcn_sched_barrier(0);
#else
#define PCUDA_CLK_FENCE
#endif
#endif

// -DPCUDA_WEXP (timing experiments, results are wrong): PCUDA_DBG bit 8 = no weight loads / LDS writes after a workgroup's
// first stage, bit 16 = also no barriers between weight groups (as if every tap were resident), bit 32 = no input commit
// after the first stage, bit 64 = no epilogue (no output stores, no statistics) after the first stage: upper bounds of what weight delivery / staging can be worth in this kernel's structure.
#ifdef PCUDA_WEXP
#define WEXP_DECL bool wexp_first = true;
#define WEXP_W (!(p.dbg & 8) || wexp_first)
#define WEXP_B (!(p.dbg & 16) || wexp_first)
#define WEXP_X (!(p.dbg & 32) || wexp_first)
#define WEXP_E (!(p.dbg & 64) || wexp_first)
#define WEXP_END wexp_first = false;
#else
#define WEXP_DECL
#define WEXP_W true
#define WEXP_B true
#define WEXP_X true
#define WEXP_E true
#define WEXP_END
#endif

// WV = weight-copy register slots per lane and plane.  WV = 2: weight groups of <= 512 vectors, two
// workgroups per CU.  WV = 6 / 12 (CO_TILE 32 / 64): ONE workgroup per CU with up to nine taps of weights
// resident (512 VGPRs per lane: the whole group sits in registers between its loads and its LDS
// write) -- a stage then has one weight round trip, issued in front of the X prefetch, and no barrier
// inside its MFMA phase.
// STATS: 0 none, 1 BatchNorm partial sums of the stored values (forward), 2 BatchNorm-BACKWARD reduce partials of the
// stored gradient against the saved activation (dgrad of a block's second convolution; transposed epilogue only)
template <bool X3, int CO_BLKS, bool CLAMP, int NPB, int PF, int WV, bool XQ, int STATS, bool TE, bool FOLD = false>
__global__ __launch_bounds__(256, (WV > 4 ? 1 : 2)) void igemm_pipe_kernel(const IgemmParams p, const int x_cap, const int total) {
  static_assert(!FOLD || (TE && NPB == 2 && STATS != 1), "2x2 fold: transposed-epilogue plans with two pixel blocks (tile rows 2 w, 2 w + 1)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CO_TILE = 32 * CO_BLKS;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int TW = p.tw, TPIX = p.tw * p.th;

  // XCD-aware persistent schedule: the 8 XCDs own contiguous eighths of the (pixel tile, co tile) list;
  // the workgroups of one XCD (blockIdx % 8) interleave over it, so concurrent workgroups touch
  // adjacent tiles (shared halo rows and weights hit that XCD's L2).
  const int nx = min(8, (int)gridDim.x);                          // XCD groups that actually have workgroups
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int gx = ((int)gridDim.x - xcd + nx - 1) / nx;            // workgroups in this group
  const int lo = (int)((long long)total * xcd / nx), hi = (int)((long long)total * (xcd + 1) / nx);

  constexpr int REC = IgRec<X3>::BYTES, RECV = REC / 16;
  // weight-copy register slots per lane (16-B vectors): a bf16x3 record carries both planes
  constexpr int WS = X3 ? 2 * WV + (CO_BLKS == 2 ? 1 : 0) : WV;
  unsigned char* Xhi = smem;
  unsigned char* Xlo = Xhi + IG_LO_OFF;
  unsigned char* Whi = smem + (size_t)x_cap * REC;
  unsigned char* Wlo = Whi + IG_LO_OFF;
  int* taptab = (int*)(Whi + (size_t)p.tg * CO_TILE * REC);   // [ntaps], behind the weight slab
  float* sbias = (float*)(taptab + 64);   // [CO_TILE] bias of the tile being finished
  float* smean = sbias + 64;              // [CO_TILE] STATS == 2: mean / invstd of the rows' BatchNorm (host adds 1 KiB in all)
  float* sinv = smean + 64;
  float* sred = (float*)smem;   // [4 waves][CO_TILE][2], reused between a tile's last MFMA and the next commit

  // per-tap LDS offset (or packed dy/dx in clamp mode), read back with one broadcast ds_read per tap:
  // indexing the signed-char tables of the kernel argument compiled to two global loads per tap, and
  // waiting for those (vmcnt is in order) drained the whole X prefetch at the first tap of every stage
  for (int t = 0; t < p.ntaps; ++t) {   // uniform index: a per-lane index would copy the tables to scratch
    const int dy = p.dy[t], dx = p.dx[t];
    if (tid == 0)
      taptab[t] = CLAMP ? ((dy & 0xffff) | (dx << 16)) : ((dy - p.dy_min) * p.iw_t + (dx - p.dx_min)) * REC;
  }

  int pty[NPB], ptx[NPB];
  bool pvalid[NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    const int plr = w * (32 * NPB) + pb * 32 + r;
    pvalid[pb] = plr < TPIX;
    const int pl = min(plr, TPIX - 1);
    pty[pb] = IG_TY(pl, p.tmagic);
    ptx[pb] = pl - pty[pb] * TW;
  }

  f32x16 acc[CO_BLKS][NPB];
  XFast<PF> pre;
  DBG_CLK_DECL
  WEXP_DECL

  int L = lo + slot, chunk = 0;
  bool have = L < hi;
  TileGeom g;
  if (have) {
    g = tile_decode<CLAMP, NPB>(p, L);
    if constexpr (XQ) xq_issue<PF>(pre, p.x, g.n, p.cin, 0, p.in_h, p.in_w, g.oy0, g.ox0, g.th, g.tw, tid);
    else xfast_issue<PF>(pre, p.x, g.n, p.cin, 0, p.in_h, p.in_w, p.in_shift, p.in_row, g.oy0, g.ox0, g.tw, g.npix,
                         (min(32, p.cin) + 7) >> 3, tid);
  }
  while (have) {
    const int cvalid = min(32, p.cin - chunk * 32);
    const int nks = cvalid > 16 ? 2 : 1;
    DBG_CLK(7)
    __syncthreads();   // every wave is done with the previous stage's X / W / reduction scratch
    DBG_CLK(0)
    if (WEXP_X) {
    if constexpr (XQ) xq_commit<X3, PF, 256, REC>(pre, Xhi, Xlo, p.x, p.cin, chunk, g.ox0, g.th, g.tw, nks * 2, tid);
    else xfast_commit<X3, PF, 256, REC>(pre, Xhi, Xlo, p.x, p.cin, chunk, g.npix, (cvalid + 7) >> 3, nks * 2, tid);
    }
    DBG_CLK(1)
    if (CLAMP && tid < RECV)   // the all-zero record that out-of-image taps read
      *(uint4*)(Xhi + (size_t)g.npix * REC + tid * 16) = make_uint4(0, 0, 0, 0);
    // bias of this tile's rows (consumed by the epilogue after the last chunk): fetched here, in front of the
    // stage's other loads, so its wait never drains them
    float bias_r = 0.f;
    if (chunk == p.nchunks - 1 && p.bias && tid < CO_TILE) bias_r = p.bias[min(g.cot * CO_TILE + tid, p.cout - 1)];
    float mean_r = 0.f, inv_r = 0.f;
    if (STATS == 2 && chunk == p.nchunks - 1 && tid < CO_TILE) {
      mean_r = p.red_mean[min(g.cot * CO_TILE + tid, p.cout - 1)];
      inv_r = p.red_invstd[min(g.cot * CO_TILE + tid, p.cout - 1)];
    }
    // (pinned behind the commit: hoisted above it into one memory clause, the two parked weight groups were live next
    // to the 64 prefetch registers and the kernel spilled)
    __builtin_amdgcn_sched_barrier(0);
    // first weight group: its loads go out BEFORE the next stage's input prefetch (vmcnt retires in order:
    // behind the prefetch they would not be usable until all of it has landed)
    const long long slab = (long long)CO_TILE * (REC / 2);
    const uint16_t* wsrc = p.wpack + ((long long)g.cot * p.nchunks + chunk) * p.ntaps * slab;
    // (an opaque zero in the lane index of the weight copies: their per-lane vector indices and LDS addresses are
    // loop-invariant, were hoisted out of the stage loop -- three groups' worth -- and spilled; every reload sat in
    // front of a load or an LDS write with an s_waitcnt vmcnt(0) that drained the prefetch)
    const int tidw = tid + opaque_zero();
    WPass<false, WS> wp0;
    const int nvec0 = min(p.tg, p.ntaps) * CO_TILE * RECV;
    if (WEXP_W) wcopy_issue<false, WS>(wp0, (const uint4*)wsrc, nullptr, nvec0, 0, tidw);
    // second weight group (more taps than one LDS slab holds): requested here too, IN FRONT of the input prefetch, and
    // parked in registers until the first group's MFMAs are done.  Requested behind the prefetch (vmcnt retires in
    // order) its wait was a wait for the whole next input tile to arrive from HBM, in the middle of every stage.
    // Unconditional (a single-group plan re-reads one vector of group 0): a branch around loads would turn the
    // counted waits below into vmcnt(0).
    // Weight groups (more taps than one LDS slab holds) are requested IN FRONT of the next stage's input prefetch and
    // parked in registers until the slab is free: requested behind the prefetch (vmcnt retires in order) a group's wait
    // was a wait for the whole next input tile to arrive from HBM, in the middle of every stage.
    //  * 32-row tiles (two groups: 7 + 2 taps of a 3x3 layer): both groups go out here, the prefetch follows;
    //  * 64-row tiles (three groups of 3 taps; no registers for two parked groups next to the 64 prefetch registers):
    //    groups 0 and 1 go out here, group 2 and THEN the prefetch right after group 0's MFMAs, when group 0's
    //    registers are free again -- the prefetch is still in flight for two thirds of the stage.
    // The loads are unconditional (a plan with fewer groups re-reads one vector): a branch around loads would turn the
    // counted waits below into vmcnt(0).  Groups past the third are copied in place (latency exposed; 16-tap layers).
    constexpr bool DEFER = CO_BLKS == 2;
    const int ngrp = (p.ntaps + p.tg - 1) / p.tg;
    WPass<false, WS> wp1;
    const int nvec1 = ngrp > 1 ? min(p.tg, p.ntaps - p.tg) * CO_TILE * RECV : 1;
    {
      const uint16_t* src1 = wsrc + (ngrp > 1 ? (long long)p.tg * slab : 0);
      if (WEXP_W) wcopy_issue<false, WS>(wp1, (const uint4*)src1, nullptr, nvec1, 0, tidw);
    }
    __builtin_amdgcn_sched_barrier(0);

    // next stage: its loads stay in flight through everything below
    int nL = L, nchunk = chunk + 1;
    if (nchunk == p.nchunks) { nchunk = 0; nL = L + gx; }
    const bool nhave = nL < hi;
    TileGeom ng = g;
    if (nhave && nL != L) ng = tile_decode<CLAMP, NPB>(p, nL);
    // unconditional (no stage left: zero pixels, every lane out of range -> no memory traffic)
    auto issue_next = [&]() {
      if constexpr (XQ) xq_issue<PF>(pre, p.x, ng.n, p.cin, nhave ? nchunk : 0, p.in_h, p.in_w, ng.oy0, ng.ox0, nhave ? ng.th : 0,
                             ng.tw, tid);
      else xfast_issue<PF>(pre, p.x, ng.n, p.cin, nhave ? nchunk : 0, p.in_h, p.in_w, p.in_shift, p.in_row, ng.oy0,
                           ng.ox0, ng.tw, nhave ? ng.npix : 0, 4, tid);
    };
    if (!DEFER) issue_next();
    DBG_CLK(2)

    if (chunk == 0) {
#pragma unroll
      for (int cb = 0; cb < CO_BLKS; ++cb)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[cb][pb][i] = 0.f;
    }
    int bbase[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
      bbase[pb] = ((pty[pb] * p.in_step) * g.tw + ptx[pb] * p.in_step) * REC + h * 16;

    // MFMA phase of one weight group (taps t0 .. t0 + tgc - 1, resident in the slab).
    // (Operand prefetch across k-steps was tried twice and reverted: two full fragment sets spilled 41-109 registers
    // next to the 64 prefetch registers; a rotating form -- weights one step ahead in a second set, each pixel block's
    // input fragments re-read under the other block's MFMAs -- fit, and measured 4 % SLOWER on the 3x3 layers: with two
    // waves per SIMD the partner wave already covers the LDS round trip, and the schedule fences cost more.)
    auto mfma_group = [&](int t0, int tgc) {
      int tv = taptab[t0];
      for (int tl = 0; tl < tgc; ++tl) {
        const int tcur = tv;
        tv = taptab[min(t0 + tl + 1, p.ntaps - 1)];   // next tap's entry: its LDS latency hides behind this tap
        int baddr[NPB];
        if (CLAMP) {
          const int dy = (tcur << 16) >> 16, dx = tcur >> 16;
          bool any = false;
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) {
            const int gy = (g.y0 + pty[pb]) * p.in_step + dy;
            const int gxx = (g.x0 + ptx[pb]) * p.in_step + dx;
            const bool ok = ((unsigned)gy < (unsigned)p.in_h) & ((unsigned)gxx < (unsigned)p.in_w);
            any |= ok;
            const int idx = ok ? (gy - g.oy0) * g.tw + (gxx - g.ox0) : g.npix;
            baddr[pb] = idx * REC + h * 16;
          }
          // (a tap whose reads all fall outside the image for this wave's pixel blocks multiplies the zero record: skipped)
          if (__builtin_amdgcn_ballot_w64(any) == 0) continue;
        } else {
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) baddr[pb] = bbase[pb] + tcur;
        }
        const int abase = (tl * CO_TILE + r) * REC + h * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (ks < nks) {
            bf16x8 ah[CO_BLKS], al[CO_BLKS], bh[NPB], bl[NPB];
#pragma unroll
            for (int cb = 0; cb < CO_BLKS; ++cb) {
              ah[cb] = lds_frag(Whi + abase + cb * 32 * REC + ks * 32);
              if (X3) al[cb] = lds_frag(Wlo + abase + cb * 32 * REC + ks * 32);
            }
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
              bh[pb] = lds_frag(Xhi + baddr[pb] + ks * 32);
              if (X3) bl[pb] = lds_frag(Xlo + baddr[pb] + ks * 32);
            }
#pragma unroll
            for (int cb = 0; cb < CO_BLKS; ++cb)
#pragma unroll
              for (int pb = 0; pb < NPB; ++pb) {
                if (X3) {
                  acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb], bh[pb], acc[cb][pb], 0, 0, 0);
                  acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bl[pb], acc[cb][pb], 0, 0, 0);
                }
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bh[pb], acc[cb][pb], 0, 0, 0);
              }
          }
        }
      }
    };

    if (WEXP_W) wcopy_commit<false, WS>(wp0, Whi, nullptr, nvec0, 0, tidw);
    DBG_CLK(3)
    __syncthreads();
    DBG_CLK(4)
    mfma_group(0, min(p.tg, p.ntaps));
    DBG_CLK_ACC(5)
    const int nvec2 = ngrp > 2 ? min(p.tg, p.ntaps - 2 * p.tg) * CO_TILE * RECV : 1;
    if (ngrp > 1) {   // uniform
      if (WEXP_B) __syncthreads();
      if (WEXP_W) wcopy_commit<false, WS>(wp1, Whi, nullptr, nvec1, 0, tidw);   // parked since the top of the stage
      if (DEFER) {
        __builtin_amdgcn_sched_barrier(0);   // (group 2's loads stay behind group 1's LDS writes: its registers are free then)
        const uint16_t* src2 = wsrc + (ngrp > 2 ? 2ll * p.tg * slab : 0);
        if (WEXP_W) wcopy_issue<false, WS>(wp0, (const uint4*)src2, nullptr, nvec2, 0, tidw);
        __builtin_amdgcn_sched_barrier(0);
        issue_next();
      }
      DBG_CLK(3)
      if (WEXP_B) __syncthreads();
      DBG_CLK(4)
      mfma_group(p.tg, min(p.tg, p.ntaps - p.tg));
      DBG_CLK_ACC(5)
      for (int t0 = 2 * p.tg, gi = 2; t0 < p.ntaps; t0 += p.tg, ++gi) {
        const int tgc = min(p.tg, p.ntaps - t0);
        if (WEXP_B) __syncthreads();
        if (DEFER && gi == 2) {
          if (WEXP_W) wcopy_commit<false, WS>(wp0, Whi, nullptr, nvec2, 0, tidw);   // requested in front of the prefetch
        } else if (WEXP_W) {
          const uint16_t* src = wsrc + (long long)t0 * slab;
          wcopy<false, WS>(Whi, nullptr, (const uint4*)src, nullptr, tgc * CO_TILE * RECV, 0, tidw);
        }
        DBG_CLK(3)
        if (WEXP_B) __syncthreads();
        DBG_CLK(4)
        mfma_group(t0, tgc);
        DBG_CLK_ACC(5)
      }
    } else if (DEFER) {
      issue_next();   // single group: the prefetch goes out behind the stage's MFMAs
    }

    DBG_CLK(6)
    if (chunk == p.nchunks - 1 && WEXP_E) {
      // ---- epilogue of this tile.  No flat global access in here: a lane-indexed bias load in front of every
      // store made each store wait (vmcnt(0), in order) for the previous one AND for the whole X prefetch.
      // Bias comes through LDS (fetched at the top of the stage), stores go through buffer resources:
      // 32-bit offsets, rows past cout / pixels outside the output dropped by the range check.
      const int co0 = g.cot * CO_TILE;
      if (tid < CO_TILE) {
        sbias[tid] = bias_r;
        if (STATS == 2) { smean[tid] = mean_r; sinv[tid] = inv_r; }
      }
      __syncthreads();   // sbias visible; sred aliases the X tile: every wave's last fragment reads are done
      if constexpr (FOLD) {
        // 2x2 fold (the data gradient of a layer whose input was read through nearest x2, unet.py:111-112: the gradient of the
        // STORED half-resolution tensor is the sum over each 2x2 block of the logical one).  Tiles are 32 pixels wide, so a
        // wave's two pixel blocks are the tile rows 2 w and 2 w + 1: the vertical pair is one add per accumulator register;
        // the row of sums goes through the wave's LDS scratch ([32 channels][32 pixels]) so that a lane holds 4 consecutive
        // pixels of one channel, whose two horizontal pairs it stores as 8 bytes.  The 4x larger logical gradient never
        // reaches HBM (it was written by this kernel and read back by upsample2_bwd_kernel: 1 GB per pass at 256x256), and
        // with STATS == 2 the BatchNorm-backward reduce of the layer in front rides along as in the unfolded epilogue.
        float* const tsc = (float*)smem + w * (32 * 32);
        float* const sred2 = (float*)smem + 4 * 32 * 32;
        const int q = lane & 7, cs = lane >> 3;
        const int ly = g.y0 + 2 * w, lx = g.x0 + 4 * q;
        const bool pokq = (2 * w < p.th) & (ly + 1 < p.lh) & (lx + 3 < p.lw);
        const unsigned pixq = (unsigned)((ly >> 1) * p.out_w + (lx >> 1)) * 4u;
        const int c1 = min(p.y.c1, p.cout);
        char* const yb1 = (char*)(p.y.p1 + (long long)g.n * p.y.sn1);
        char* const yb2 = (char*)(p.y.p2 + (long long)g.n * p.y.sn2);
        const unsigned pl1 = (unsigned)p.y.sc1 * 4u, pl2 = (unsigned)p.y.sc2 * 4u;
#pragma unroll
        for (int cb = 0; cb < CO_BLKS; ++cb) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            tsc[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[cb][0][i] + acc[cb][1][i];
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const f32x4 v = *(const f32x4*)(tsc + (it * 8 + cs) * 32 + 4 * q);
            const float o0 = v[0] + v[1], o1 = v[2] + v[3];
            const int cu = co0 + cb * 32 + it * 8;                     // uniform; the 8 channels lie in one destination
            const bool first = cu < c1;
            char* const base = first ? yb1 + (long long)cu * pl1 : yb2 + (long long)(cu - c1) * pl2;
            char* const dptr = base + (size_t)((unsigned)cs * (first ? pl1 : pl2) + pixq);
            const bool ok = pokq & (cu + cs < p.cout);
            float a0 = 0.f, a1 = 0.f;
            if (STATS == 2 && ok) {
              const float2 av = *(const float2*)((const char*)(p.red_a + (long long)g.n * p.red_sn) + ((long long)(cu + cs) * p.red_sc) * 4 + pixq);
              a0 = av.x; a1 = av.y;
            }
            if (ok) *(float2*)dptr = make_float2(o0, o1);
            if (STATS == 2) {
              const int rr = cb * 32 + it * 8 + cs;
              const float m = smean[rr], is = sinv[rr];
              float s1 = ok ? o0 + o1 : 0.f;
              float s2 = ok ? o0 * ((a0 - m) * is) + o1 * ((a1 - m) * is) : 0.f;
              s1 = row_sum<8>(s1);
              s2 = row_sum<8>(s2);
              if (q == 0) {
                sred2[(w * CO_TILE + rr) * 2 + 0] = s1;
                sred2[(w * CO_TILE + rr) * 2 + 1] = s2;
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        if (STATS == 2) {
          __syncthreads();
          if (tid < CO_TILE) {
            const int co = g.cot * CO_TILE + tid;
            if (co < p.cout) {
              float s1 = 0.f, s2 = 0.f;
#pragma unroll
              for (int ww = 0; ww < 4; ++ww) {
                s1 += sred2[(ww * CO_TILE + tid) * 2 + 0];
                s2 += sred2[(ww * CO_TILE + tid) * 2 + 1];
              }
              p.stats[((long long)g.pt * p.cout + co) * 2 + 0] = s1;
              p.stats[((long long)g.pt * p.cout + co) * 2 + 1] = s2;
            }
          }
        }
      } else if constexpr (TE) {
        // Transposed epilogue (rows of 4k pixels, unit x stride): every wave turns its [32 rows][WPIX pixels]
        // accumulator block through its own LDS scratch so that a lane holds 4 consecutive pixels of one channel:
        // one 16-byte store per lane, a wave instruction writes whole 128-B row segments of CPI channels
        // (8 / 4 stores per row block instead of 32 / 16 one-dword stores), and the BatchNorm partial sums of a
        // channel come out of ONE 16- (8-) lane DPP reduction.  Wave-local: no workgroup barrier inside.
        constexpr int WPIX = 32 * NPB, LPC = WPIX / 4, CPI = 64 / LPC, NIT = 32 / CPI;
        float* const tsc = (float*)smem + w * (32 * WPIX);
        float* const sred2 = (float*)smem + 4 * 32 * WPIX;   // [4 waves][CO_TILE][2], behind the scratch
        const int q = lane & (LPC - 1), cs = lane / LPC;
        const int slot = w * WPIX + 4 * q;
        const int sl = min(slot, TPIX - 1);
        const int qy = IG_TY(sl, p.tmagic), qx = sl - qy * TW;
        const int ly = g.y0 + qy, lx = g.x0 + qx;
        const bool pokq = (slot < TPIX) & (ly < p.lh) & (lx < p.lw);
        const unsigned pixq = (unsigned)((ly * p.oy_mul + p.oy_off) * p.out_w + lx) * 4u;
        const int c1 = min(p.y.c1, p.cout);
        char* const yb1 = (char*)(p.y.p1 + (long long)g.n * p.y.sn1);
        char* const yb2 = (char*)(p.y.p2 + (long long)g.n * p.y.sn2);
        const unsigned pl1 = (unsigned)p.y.sc1 * 4u, pl2 = (unsigned)p.y.sc2 * 4u;
#pragma unroll
        for (int cb = 0; cb < CO_BLKS; ++cb) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
              tsc[((i & 3) + 8 * (i >> 2) + 4 * h) * WPIX + pb * 32 + r] = acc[cb][pb][i];
          __builtin_amdgcn_wave_barrier();
          // (two batches of HN store instructions: a whole row block at once kept 80 more registers live next to
          // the next stage's prefetch and spilled)
          constexpr int HN = NIT > 4 ? 4 : NIT;
#pragma unroll
          for (int ib = 0; ib < NIT; ib += HN) {
            f32x4 v[HN];
            float bia[HN];
            char* dptr[HN];
            bool ok[HN];
#pragma unroll
            for (int k = 0; k < HN; ++k) {
              const int it = ib + k;
              v[k] = *(const f32x4*)(tsc + (it * CPI + cs) * WPIX + 4 * q);
              bia[k] = sbias[cb * 32 + it * CPI + cs];
              const int cu = co0 + cb * 32 + it * CPI;               // uniform; the CPI channels lie in one destination
              const bool first = cu < c1;
              char* const base = first ? yb1 + (long long)cu * pl1 : yb2 + (long long)(cu - c1) * pl2;
              dptr[k] = base + (size_t)((unsigned)cs * (first ? pl1 : pl2) + pixq);
              ok[k] = pokq & (cu + cs < p.cout);
            }
            f32x4 av[HN];
            if (STATS == 2) {   // the saved activation of the rows' BatchNorm, same element as the stored gradient
              char* const ab = (char*)(p.red_a + (long long)g.n * p.red_sn);
#pragma unroll
              for (int k = 0; k < HN; ++k) {
                av[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int cu = co0 + cb * 32 + (ib + k) * CPI + cs;
                if (ok[k]) av[k] = *(const f32x4*)(ab + ((long long)cu * p.red_sc) * 4 + pixq);
              }
            }
            if (p.accumulate) {   // uniform: dgrad into a gradient that already holds another consumer's share
              f32x4 o[HN];
#pragma unroll
              for (int k = 0; k < HN; ++k) {
                o[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok[k]) o[k] = *(const f32x4*)dptr[k];
              }
#pragma unroll
              for (int k = 0; k < HN; ++k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float t = v[k][e] + bia[k];
                  t = t > 0.f ? t : t * p.slope;
                  v[k][e] = t + o[k][e];
                }
              }
            } else {
#pragma unroll
              for (int k = 0; k < HN; ++k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float t = v[k][e] + bia[k];
                  v[k][e] = t > 0.f ? t : t * p.slope;
                }
              }
            }
#pragma unroll
            for (int k = 0; k < HN; ++k) {
              if (ok[k]) *(f32x4*)dptr[k] = v[k];
              if (STATS) {
                float s1 = (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
                float s2;
                if (STATS == 2) {
                  const int rr = cb * 32 + (ib + k) * CPI + cs;
                  const float m = smean[rr], is = sinv[rr];
                  s2 = (v[k][0] * ((av[k][0] - m) * is) + v[k][1] * ((av[k][1] - m) * is)) +
                       (v[k][2] * ((av[k][2] - m) * is) + v[k][3] * ((av[k][3] - m) * is));
                } else {
                  s2 = (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
                }
                s1 = ok[k] ? s1 : 0.f;
                s2 = ok[k] ? s2 : 0.f;
                s1 = row_sum<LPC>(s1);
                s2 = row_sum<LPC>(s2);
                if (q == 0) {
                  const int row = cb * 32 + (ib + k) * CPI + cs;
                  sred2[(w * CO_TILE + row) * 2 + 0] = s1;
                  sred2[(w * CO_TILE + row) * 2 + 1] = s2;
                }
              }
            }
          }
          __builtin_amdgcn_wave_barrier();   // the next row block's scratch writes stay behind these reads
        }
        if (STATS) {
          __syncthreads();
          if (tid < CO_TILE) {
            const int co = g.cot * CO_TILE + tid;
            if (co < p.cout) {
              float s1 = 0.f, s2 = 0.f;
#pragma unroll
              for (int ww = 0; ww < 4; ++ww) {
                s1 += sred2[(ww * CO_TILE + tid) * 2 + 0];
                s2 += sred2[(ww * CO_TILE + tid) * 2 + 1];
              }
              p.stats[((long long)g.pt * p.cout + co) * 2 + 0] = s1;
              p.stats[((long long)g.pt * p.cout + co) * 2 + 1] = s2;
            }
          }
        }
      } else {
      // (paired column classes, IgemmParams::pair: row 2c + rx is channel c, pixel 2 lx + rx -- registers i and i + 1 of
      // a lane are the two halves of 8 consecutive bytes; pair = 0 leaves every index below as it was)
      unsigned pixo[NPB], pixo1[NPB];
      bool pok[NPB];
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        const int ly = g.y0 + pty[pb], lx = g.x0 + ptx[pb];
        pok[pb] = pvalid[pb] & (ly < p.lh) & (lx < p.lw);
        pixo[pb] = pok[pb] ? (unsigned)((ly * p.oy_mul + p.oy_off) * p.out_w + (lx * p.ox_mul + p.ox_off)) * 4u : IG_OOB;
        pixo1[pb] = (pok[pb] & (lx < p.lw2)) ? pixo[pb] + 4u : IG_OOB;
      }
      const int nch = p.cout >> p.pair;                         // channels of the destination(s)
      const int c1 = min(p.y.c1, nch);
      float* const yb1 = p.y.p1 + (long long)g.n * p.y.sn1;
      float* const yb2 = p.y.p2 + (long long)g.n * p.y.sn2;
      const unsigned pl1 = (unsigned)p.y.sc1 * 4u, pl2 = (unsigned)p.y.sc2 * 4u;
      if (STATS == 0 && p.mask_a) {   // (uniform) its own loop: the plain one below keeps its code and registers
        // LeakyReLU backward of the layer in front (GAN.py:97-108 going back): the saved activation has the destination's
        // layout, so one buffer resource per image and the store's own offsets read it.  A row block's 16 x NPB values are
        // requested together, in front of its stores (vmcnt retires in order: a load behind a store waits for that store).
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.mask_a + (long long)g.n * p.mask_sn), 0, (int)(nch * pl1), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)yb1, 0, (int)(nch * pl1), 0x00020000);
#pragma unroll
        for (int cb = 0; cb < CO_BLKS; ++cb) {
          float av[16][NPB];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const unsigned soff = (unsigned)((co0 + cb * 32 + (i & 3) + 8 * (i >> 2)) >> p.pair) * pl1;
            const unsigned hoff = h ? (4u >> p.pair) * pl1 : 0u;
            const bool odd = (i & 1) & p.pair;
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
              av[i][pb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (odd ? pixo1[pb] : pixo[pb]) + hoff, soff, 0));
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const unsigned soff = (unsigned)((co0 + cb * 32 + (i & 3) + 8 * (i >> 2)) >> p.pair) * pl1;
            const unsigned hoff = h ? (4u >> p.pair) * pl1 : 0u;
            const bool odd = (i & 1) & p.pair;
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
              float v = acc[cb][pb][i];
              v = av[i][pb] > 0.f ? v : v * p.mask_slope;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (odd ? pixo1[pb] : pixo[pb]) + hoff, soff, 0);
            }
          }
        }
      } else {
#pragma unroll
      for (int cb = 0; cb < CO_BLKS; ++cb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row0 = cb * 32 + (i & 3) + 8 * (i >> 2);   // this register's row for h = 0; h = 1 is 4 rows on
          const int row = row0 + 4 * h;
          const int cu = (co0 + row0) >> p.pair;                // uniform; rows cu and cu + 4 lie in one destination
          const bool first = cu < c1;
          const unsigned plane = first ? pl1 : pl2;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(first ? yb1 : yb2), 0, (int)((first ? c1 : nch - c1) * plane), 0x00020000);
          const unsigned soff = (unsigned)(first ? cu : cu - c1) * plane;
          const unsigned hoff = h ? (4u >> p.pair) * plane : 0u;
          const bool odd = (i & 1) & p.pair;                    // uniform: the odd column class
          const float b = sbias[row];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) {
            float v = acc[cb][pb][i] + b;
            v = v > 0.f ? v : v * p.slope;
            const unsigned vo = (odd ? pixo1[pb] : pixo[pb]) + hoff;
            if (p.accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, soff, 0));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vo, soff, 0);
            if (STATS) {
              const float vs = (pok[pb] & (co0 + row < p.cout)) ? v : 0.f;
              s1 += vs;
              s2 += vs * vs;
            }
          }
          if (STATS) {
            s1 = half_wave_sum_hi16(s1);
            s2 = half_wave_sum_hi16(s2);
            if (r == 31) {
              sred[(w * CO_TILE + row) * 2 + 0] = s1;
              sred[(w * CO_TILE + row) * 2 + 1] = s2;
            }
          }
        }
      }
      }
      if (STATS) {
        __syncthreads();
        if (tid < CO_TILE) {
          const int co = g.cot * CO_TILE + tid;
          if (co < p.cout) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
              s1 += sred[(ww * CO_TILE + tid) * 2 + 0];
              s2 += sred[(ww * CO_TILE + tid) * 2 + 1];
            }
            p.stats[((long long)g.pt * p.cout + co) * 2 + 0] = s1;
            p.stats[((long long)g.pt * p.cout + co) * 2 + 1] = s2;
          }
        }
      }
          }
    }
    DBG_CLK(6)
    WEXP_END
    L = nL; chunk = nchunk; g = ng; have = nhave;
  }
  DBG_CLK_FLUSH
}




// ------------------------------------------------------------------------------------------
// Eight-wave variant: ONE 512-thread workgroup per CU owning all 160 KiB of LDS, so that (for 3x3
// layers) every tap of a chunk's weights is resident -- one weight round trip per stage, requested
// ahead of the stage's X commit, and no barrier inside the MFMA phase -- while each SIMD still holds two
// waves.  Per lane the staging work and its registers halve (512 threads share a tile).
// Wave roles: NPBT = 32-pixel blocks per tile (8: 256-pixel tiles, every wave owns one pixel block and
// all CO_BLKS row blocks; 4: 128-pixel tiles with CO_BLKS = 2, waves 0-3 take row block 0, 4-7 block 1).
// ------------------------------------------------------------------------------------------
#define IG8_WV 6    // weight-copy slots per lane (bf16 mode, 80-B records): 6 * 512 vectors >= 9 taps x 64 rows x 5
#define IG8_WV3 11  // bf16x3 (144-B records): 11 * 512 vectors >= 9 taps x 64 rows x 9
template <bool X3, int CO_BLKS, bool CLAMP, int NPBT, int PF, bool XQ, bool STATS>
__global__ __launch_bounds__(512, 2) void igemm8_kernel(const IgemmParams p, const int x_cap, const int total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CO_TILE = 32 * CO_BLKS, NT = 512, WV = IG8_WV;
  constexpr int CBW = CO_BLKS * NPBT / 8;   // row blocks per wave
  static_assert(CBW >= 1 && CO_BLKS * NPBT % 8 == 0, "8 waves need 8 or 16 MFMA tiles per stage");
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pbw = w % NPBT, cb0 = (w / NPBT) * CBW;
  const int TW = p.tw, TPIX = p.tw * p.th;

  // one workgroup per CU; XCD-aware persistent schedule as in igemm_pipe_kernel
  const int nx = min(8, (int)gridDim.x);
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int gx = ((int)gridDim.x - xcd + nx - 1) / nx;
  const int lo = (int)((long long)total * xcd / nx), hi = (int)((long long)total * (xcd + 1) / nx);

  constexpr int REC = IgRec<X3>::BYTES, RECV = REC / 16;
  constexpr int WS = X3 ? IG8_WV3 : WV;
  unsigned char* Xhi = smem;
  unsigned char* Xlo = Xhi + IG_LO_OFF;
  unsigned char* Whi = smem + (size_t)x_cap * REC;
  unsigned char* Wlo = Whi + IG_LO_OFF;
  int* taptab = (int*)(Whi + (size_t)p.tg * CO_TILE * REC);
  float* sbias = (float*)(taptab + 64);
  float* sred = (float*)smem;   // [8 waves][CO_TILE][2]

  for (int t = 0; t < p.ntaps; ++t) {
    const int dy = p.dy[t], dx = p.dx[t];
    if (tid == 0)
      taptab[t] = CLAMP ? ((dy & 0xffff) | (dx << 16)) : ((dy - p.dy_min) * p.iw_t + (dx - p.dx_min)) * REC;
  }

  const int plr = pbw * 32 + r;
  const bool pvalid = plr < TPIX;
  const int pl = min(plr, TPIX - 1);
  const int pty = IG_TY(pl, p.tmagic), ptx = pl - pty * TW;

  f32x16 acc[CBW];
  XFast<PF> pre;
  DBG_CLK_DECL
  WEXP_DECL

  auto issue_x = [&](const TileGeom& t, int ch, bool live) {
    if (XQ) xq_issue<PF, NT>(pre, p.x, t.n, p.cin, live ? ch : 0, p.in_h, p.in_w, t.oy0, t.ox0, live ? t.th : 0, t.tw, tid);
    else xfast_issue<PF, NT>(pre, p.x, t.n, p.cin, live ? ch : 0, p.in_h, p.in_w, p.in_shift, p.in_row, t.oy0, t.ox0, t.tw,
                             live ? t.npix : 0, 4, tid);
  };

  int L = lo + slot, chunk = 0;
  bool have = L < hi;
  TileGeom g;
  if (have) {
    g = tile_decode<CLAMP, 1>(p, L);
    issue_x(g, 0, true);
  }
  while (have) {
    const int cvalid = min(32, p.cin - chunk * 32);
    const int nks = cvalid > 16 ? 2 : 1;
    DBG_CLK(7)
    __syncthreads();   // every wave is done with the previous stage's X / W / reduction scratch
    DBG_CLK(0)
    float bias_r = 0.f;
    if (chunk == p.nchunks - 1 && p.bias && tid < CO_TILE) bias_r = p.bias[min(g.cot * CO_TILE + tid, p.cout - 1)];
    if (WEXP_X) {
    if (XQ) xq_commit<X3, PF, NT, REC>(pre, Xhi, Xlo, p.x, p.cin, chunk, g.ox0, g.th, g.tw, nks * 2, tid);
    else xfast_commit<X3, PF, NT, REC>(pre, Xhi, Xlo, p.x, p.cin, chunk, g.npix, (cvalid + 7) >> 3, nks * 2, tid);
    }
    DBG_CLK(1)
    if (CLAMP && tid < RECV) *(uint4*)(Xhi + (size_t)g.npix * REC + tid * 16) = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // the stage's (first) weight group, with nothing else in the load queue: one L2 round trip.  (Requested
    // before the X commit its 48 registers were spilled to scratch one load at a time.  Parking the second group in
    // registers in front of the prefetch, as igemm_pipe_kernel does, measured neutral to 9 % slower here: the 16-tap
    // layers' groups are 44 registers each, and the one-group 3x3 layers only pay for the extra loads.)
    const long long slab = (long long)CO_TILE * (REC / 2);
    const uint16_t* wsrc = p.wpack + ((long long)g.cot * p.nchunks + chunk) * p.ntaps * slab;
    const int nvec0 = min(p.tg, p.ntaps) * CO_TILE * RECV;
    if (WEXP_W) {
      WPass<false, WS> wp0;
      wcopy_issue<false, WS, NT>(wp0, (const uint4*)wsrc, nullptr, nvec0, 0, tid);
      __builtin_amdgcn_sched_barrier(0);
      wcopy_commit<false, WS, NT>(wp0, Whi, nullptr, nvec0, 0, tid);
      __builtin_amdgcn_sched_barrier(0);
    }
    DBG_CLK(3)

    // next stage: its loads stay in flight through the MFMA phase
    int nL = L, nchunk = chunk + 1;
    if (nchunk == p.nchunks) { nchunk = 0; nL = L + gx; }
    const bool nhave = nL < hi;
    TileGeom ng = g;
    if (nhave && nL != L) ng = tile_decode<CLAMP, 1>(p, nL);
    issue_x(ng, nchunk, nhave);
    DBG_CLK(2)

    if (chunk == 0) {
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
    }
    const int bbase = ((pty * p.in_step) * g.tw + ptx * p.in_step) * REC + h * 16;
    __syncthreads();
    DBG_CLK(4)

    for (int t0 = 0; t0 < p.ntaps; t0 += p.tg) {
      const int tgc = min(p.tg, p.ntaps - t0);
      if (t0 > 0) {   // more taps than fit (4x4 / 6x6 kernels): later groups load behind the prefetch
        DBG_CLK_ACC8(5)
        if (WEXP_B) __syncthreads();
        const uint16_t* src = wsrc + (long long)t0 * slab;
        if (WEXP_W) wcopy<false, WS, NT>(Whi, nullptr, (const uint4*)src, nullptr, tgc * CO_TILE * RECV, 0, tid);
        DBG_CLK(3)
        if (WEXP_B) __syncthreads();
        DBG_CLK(4)
      }
      int tv = taptab[t0];
      for (int tl = 0; tl < tgc; ++tl) {
        const int tcur = tv;
        tv = taptab[min(t0 + tl + 1, p.ntaps - 1)];
        int baddr;
        if (CLAMP) {
          const int dy = (tcur << 16) >> 16, dx = tcur >> 16;
          const int gy = (g.y0 + pty) * p.in_step + dy;
          const int gxx = (g.x0 + ptx) * p.in_step + dx;
          const bool ok = ((unsigned)gy < (unsigned)p.in_h) & ((unsigned)gxx < (unsigned)p.in_w);
          // (dilation 8 on a 16x16 map: a wave's two tile rows read outside the image for a third of the taps -- products
          // with the zero record, skipped whole: same sums, the zeros contributed nothing)
          if (__builtin_amdgcn_ballot_w64(ok) == 0) continue;
          const int idx = ok ? (gy - g.oy0) * g.tw + (gxx - g.ox0) : g.npix;
          baddr = idx * REC + h * 16;
        } else {
          baddr = bbase + tcur;
        }
        const int abase = (tl * CO_TILE + cb0 * 32 + r) * REC + h * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (ks < nks) {
            bf16x8 ah[CBW], al[CBW];
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
              ah[cb] = lds_frag(Whi + abase + cb * 32 * REC + ks * 32);
              al[cb] = ah[cb];
              if (X3) al[cb] = lds_frag(Wlo + abase + cb * 32 * REC + ks * 32);
            }
            const bf16x8 bh = lds_frag(Xhi + baddr + ks * 32);
            bf16x8 bl = bh;
            if (X3) bl = lds_frag(Xlo + baddr + ks * 32);
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
              if (X3) {
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb], bh, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bl, acc[cb], 0, 0, 0);
              }
              acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bh, acc[cb], 0, 0, 0);
            }
          }
        }
      }
    }

    DBG_CLK_ACC8(5)
    if (chunk == p.nchunks - 1 && WEXP_E) {
      // ---- epilogue (see igemm_pipe_kernel): bias through LDS, buffer stores, DPP partial sums
      const int co0 = g.cot * CO_TILE;
      if (tid < CO_TILE) sbias[tid] = bias_r;
      __syncthreads();
      const int ly = g.y0 + pty, lx = g.x0 + ptx;
      const bool pok = pvalid & (ly < p.lh) & (lx < p.lw);
      const unsigned pixo = pok ? (unsigned)((ly * p.oy_mul + p.oy_off) * p.out_w + (lx * p.ox_mul + p.ox_off)) * 4u : IG_OOB;
      const unsigned pixo1 = (pok & (lx < p.lw2)) ? pixo + 4u : IG_OOB;   // (paired column classes: see igemm_pipe_kernel)
      const int nch = p.cout >> p.pair;
      const int c1 = min(p.y.c1, nch);
      float* const yb1 = p.y.p1 + (long long)g.n * p.y.sn1;
      float* const yb2 = p.y.p2 + (long long)g.n * p.y.sn2;
      const unsigned pl1 = (unsigned)p.y.sc1 * 4u, pl2 = (unsigned)p.y.sc2 * 4u;
      if (STATS == 0 && p.mask_a) {   // (uniform) LeakyReLU backward of the layer in front: see igemm_pipe_kernel
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.mask_a + (long long)g.n * p.mask_sn), 0, (int)(nch * pl1), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)yb1, 0, (int)(nch * pl1), 0x00020000);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
          float av[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const unsigned soff = (unsigned)((co0 + (cb0 + cb) * 32 + (i & 3) + 8 * (i >> 2)) >> p.pair) * pl1;
            const unsigned vo = (((i & 1) & p.pair) ? pixo1 : pixo) + (h ? (4u >> p.pair) * pl1 : 0u);
            av[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, vo, soff, 0));
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const unsigned soff = (unsigned)((co0 + (cb0 + cb) * 32 + (i & 3) + 8 * (i >> 2)) >> p.pair) * pl1;
            const unsigned vo = (((i & 1) & p.pair) ? pixo1 : pixo) + (h ? (4u >> p.pair) * pl1 : 0u);
            float v = acc[cb][i];
            v = av[i] > 0.f ? v : v * p.mask_slope;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vo, soff, 0);
          }
        }
      } else {
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row0 = (cb0 + cb) * 32 + (i & 3) + 8 * (i >> 2);
          const int row = row0 + 4 * h;
          const int cu = (co0 + row0) >> p.pair;
          const bool first = cu < c1;
          const unsigned plane = first ? pl1 : pl2;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(first ? yb1 : yb2), 0, (int)((first ? c1 : nch - c1) * plane), 0x00020000);
          const unsigned soff = (unsigned)(first ? cu : cu - c1) * plane;
          const unsigned vo = (((i & 1) & p.pair) ? pixo1 : pixo) + (h ? (4u >> p.pair) * plane : 0u);
          float v = acc[cb][i] + sbias[row];
          v = v > 0.f ? v : v * p.slope;
          if (p.accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, soff, 0));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vo, soff, 0);
          if (STATS) {
            const float vs = (pok & (co0 + row < p.cout)) ? v : 0.f;
            const float s1 = half_wave_sum_hi16(vs), s2 = half_wave_sum_hi16(vs * vs);
            if (r == 31) {
              sred[(w * CO_TILE + row) * 2 + 0] = s1;
              sred[(w * CO_TILE + row) * 2 + 1] = s2;
            }
          }
        }
      }
      }
      if (STATS) {
        __syncthreads();
        if (tid < CO_TILE) {
          const int co = g.cot * CO_TILE + tid;
          if (co < p.cout) {
            // the waves that own row block (tid >> 5): NPBT consecutive ones
            const int w0 = ((tid >> 5) / CBW) * NPBT;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NPBT; ++ww) {
              s1 += sred[((w0 + ww) * CO_TILE + tid) * 2 + 0];
              s2 += sred[((w0 + ww) * CO_TILE + tid) * 2 + 1];
            }
            p.stats[((long long)g.pt * p.cout + co) * 2 + 0] = s1;
            p.stats[((long long)g.pt * p.cout + co) * 2 + 1] = s2;
          }
        }
      }
    }
    DBG_CLK(6)
    WEXP_END
    L = nL; chunk = nchunk; g = ng; have = nhave;
  }
  DBG_CLK_FLUSH
}

template <bool X3, int CO_BLKS, bool CLAMP, int NPBT, int PF, bool XQ, bool STATS>
static int launch_igemm8_s(const IgemmParams& p, const IgemmPlan& pl, hipStream_t s) {
  constexpr unsigned vkey = ig8_key(X3, CO_BLKS, CLAMP, NPBT, PF, XQ, STATS);
  variant_log("ig8", vkey);
  if constexpr (!ig8_built(vkey)) {
    variant_fallback_note("igemm8_kernel", vkey);
    return PCUDA_E_NOTBUILT;
  } else {
  auto kern = igemm8_kernel<X3, CO_BLKS, CLAMP, NPBT, PF, XQ, STATS>;
  static DeviceOnce lds_opt;
  if (const unsigned long long devbit = lds_opt.pending()) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD);
    if (e != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "igemm8: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    lds_opt.mark(devbit);
  }
  const int total = p.n_co_tiles * p.n * p.tiles_x * p.tiles_y;
  int grid = 256;
  if (grid > total) grid = total;
  note_kernel("igemm8");
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), pl.lds, s, p, pl.x_cap, total);
  PCUDA_CHECK_LAUNCH("igemm8_kernel");
  return PCUDA_OK;
  }
}

// (BatchNorm partial sums are a template parameter too: only the segmenter's forward convolutions produce them)
template <bool X3, int CO_BLKS, bool CLAMP, int NPBT, int PF, bool XQ>
static int launch_igemm8_t(const IgemmParams& p, const IgemmPlan& pl, hipStream_t s) {
  return p.stats ? launch_igemm8_s<X3, CO_BLKS, CLAMP, NPBT, PF, XQ, true>(p, pl, s)
                 : launch_igemm8_s<X3, CO_BLKS, CLAMP, NPBT, PF, XQ, false>(p, pl, s);
}

template <bool X3>
static int igemm8_dispatch(const IgemmParams& p, const IgemmPlan& pl, int co_blks, int pf, hipStream_t s) {
// (the staging path is a template parameter: with both paths behind a runtime flag every kernel carried the code and
// the register pressure of the one it does not use)
#define IG8_PF(CB_, CL_, NB_)                                                                                        \
  (p.xq ? (pf == 1 ? launch_igemm8_t<X3, CB_, CL_, NB_, 1, true>(p, pl, s) : launch_igemm8_t<X3, CB_, CL_, NB_, 2, true>(p, pl, s)) \
        : (pf == 1 ? launch_igemm8_t<X3, CB_, CL_, NB_, 1, false>(p, pl, s) : launch_igemm8_t<X3, CB_, CL_, NB_, 2, false>(p, pl, s)))
#define IG8_CL(CB_, NB_) (pl.clamp ? IG8_PF(CB_, true, NB_) : IG8_PF(CB_, false, NB_))
  if (pl.npb == 2) return co_blks == 2 ? IG8_CL(2, 8) : IG8_CL(1, 8);
  return IG8_CL(2, 4);   // 128-pixel tiles: only with two row blocks
#undef IG8_CL
#undef IG8_PF
}

template <bool X3, int CO_BLKS, bool CLAMP, int NPB>
static int launch_igemm_t(const IgemmParams& p, int x_cap, size_t lds, hipStream_t s) {
  auto kern = igemm_kernel<X3, CO_BLKS, CLAMP, NPB>;
  static DeviceOnce lds_opt;
  if (const unsigned long long devbit = lds > 32 * 1024 ? lds_opt.pending() : 0ull) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD);
    if (e != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "igemm: cannot raise dynamic LDS to %d: %s", LDS_HARD, hipGetErrorString(e));
    lds_opt.mark(devbit);
  }
  const int grid = p.n_co_tiles * p.n * p.tiles_x * p.tiles_y;
  note_kernel("igemm_generic");
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p, x_cap);
  PCUDA_CHECK_LAUNCH("igemm_kernel");
  return PCUDA_OK;
}

template <bool X3, int CO_BLKS>
static int launch_igemm_c(const IgemmParams& p, const IgemmPlan& pl, hipStream_t s) {
  if (pl.clamp) return pl.npb == 2 ? launch_igemm_t<X3, CO_BLKS, true, 2>(p, pl.x_cap, pl.lds, s)
                                   : launch_igemm_t<X3, CO_BLKS, true, 1>(p, pl.x_cap, pl.lds, s);
  return pl.npb == 2 ? launch_igemm_t<X3, CO_BLKS, false, 2>(p, pl.x_cap, pl.lds, s)
                     : launch_igemm_t<X3, CO_BLKS, false, 1>(p, pl.x_cap, pl.lds, s);
}

template <bool X3, int CO_BLKS, bool CLAMP, int NPB, int PF, int WV, bool XQ, int STATS, bool TE, bool FOLD = false>
static int launch_pipe_s(const IgemmParams& p, const IgemmPlan& pl, hipStream_t s) {
  constexpr unsigned vkey = pipe_key(X3, CO_BLKS, CLAMP, NPB, PF, XQ, STATS, TE) | (FOLD ? 1u << 11 : 0u);
  variant_log("pipe", vkey);
  if constexpr (!pipe_built(vkey)) {
    variant_fallback_note("igemm_pipe_kernel", vkey);
    return PCUDA_E_NOTBUILT;
  } else {
  auto kern = igemm_pipe_kernel<X3, CO_BLKS, CLAMP, NPB, PF, WV, XQ, STATS, TE, FOLD>;
  static DeviceOnce lds_opt;
  if (const unsigned long long devbit = pl.lds > 32 * 1024 ? lds_opt.pending() : 0ull) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD);
    if (e != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "igemm_pipe: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    lds_opt.mark(devbit);
  }
  const int total = p.n_co_tiles * p.n * p.tiles_x * p.tiles_y;
  // persistent grid = what is resident at once (registers and LDS both limit it)
  static int occ_cache[4] = {0, 0, 0, 0};   // by LDS class: <=53K, <=80K, <=160K
  const int cls = pl.lds <= 54528 ? 0 : (pl.lds <= 81920 ? 1 : 2);
  if (occ_cache[cls] == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, pl.lds) != hipSuccess || nb < 1) nb = 1;
    const int lds_lim = (int)((size_t)LDS_HARD / pl.lds);
    if (nb > lds_lim) nb = lds_lim;
    if (nb > 4) nb = 4;
    if (nb < 1) nb = 1;
    occ_cache[cls] = nb;
  }
  int grid = occ_cache[cls] * 256;
  {   // PCUDA_PIPE_GRIDMUL=k, PCUDA_PIPE_GRIDMUL_MINCHUNKS=c (experiment): k times the resident workgroups for layers with >= c
      // 32-channel chunks -- shorter static tile lists per workgroup, the hardware dispatcher balances the rest
    static int mul = -1, minch = -1;
    if (mul < 0) { const char* e = getenv("PCUDA_PIPE_GRIDMUL"); mul = e ? atoi(e) : 1; if (mul < 1) mul = 1; }
    if (minch < 0) { const char* e = getenv("PCUDA_PIPE_GRIDMUL_MINCHUNKS"); minch = e ? atoi(e) : 1; }
    if (p.nchunks >= minch) grid *= mul;
  }
  if (grid > total) grid = total;
  note_kernel(FOLD ? "igemm_pipe+fold" : (STATS == 2 ? "igemm_pipe+bnred" : "igemm_pipe"));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), pl.lds, s, p, pl.x_cap, total);
  PCUDA_CHECK_LAUNCH("igemm_pipe_kernel");
  return PCUDA_OK;
  }
}

template <bool X3, int CO_BLKS, bool CLAMP, int NPB, int PF, int WV, bool XQ>
static int launch_pipe_t(const IgemmParams& p, const IgemmPlan& pl, hipStream_t s) {
  if constexpr (NPB == 2 && !CLAMP) {
    if (p.fold) {   // 2x2-folding epilogue (pcuda_conv2d_dgrad_fold): the caller checked the plan (transposed epilogue, 32 x 8 tiles)
      return (p.stats && p.red_a) ? launch_pipe_s<X3, CO_BLKS, CLAMP, NPB, PF, WV, XQ, 2, true, true>(p, pl, s)
                                  : launch_pipe_s<X3, CO_BLKS, CLAMP, NPB, PF, WV, XQ, 0, true, true>(p, pl, s);
    }
  }
  if (p.fold) PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "igemm: no 2x2-fold instantiation for this plan");
  if (pl.te) {
    if (p.stats && p.red_a) return launch_pipe_s<X3, CO_BLKS, CLAMP, NPB, PF, WV, XQ, 2, true>(p, pl, s);
    return p.stats ? launch_pipe_s<X3, CO_BLKS, CLAMP, NPB, PF, WV, XQ, 1, true>(p, pl, s)
                   : launch_pipe_s<X3, CO_BLKS, CLAMP, NPB, PF, WV, XQ, 0, true>(p, pl, s);
  }
  if (p.red_a) PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "igemm: the fused BatchNorm-backward reduce needs the transposed epilogue");
  return p.stats ? launch_pipe_s<X3, CO_BLKS, CLAMP, NPB, PF, WV, XQ, 1, false>(p, pl, s)
                 : launch_pipe_s<X3, CO_BLKS, CLAMP, NPB, PF, WV, XQ, 0, false>(p, pl, s);
}

template <bool X3, int CO_BLKS, bool CLAMP, int NPB>
static int launch_pipe_pf(const IgemmParams& p, const IgemmPlan& pl, int pf, hipStream_t s) {
  // (the one-workgroup-per-CU "fat" variant, WV = 5, measured slower everywhere and is no longer instantiated)
  if (pl.fat) PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "igemm: the fat plan is not built");
  // weight-copy slots per lane: four for the 32-row tiles (five taps per group where LDS allows: a 3x3 layer's
  // weights in two round trips per stage instead of three), three for the 64-row tiles (LDS holds two taps anyway)
  constexpr int WVN = CO_BLKS == 1 ? 4 : 3;
  if (p.xq) {
    if (pf == 1) return launch_pipe_t<X3, CO_BLKS, CLAMP, NPB, 1, WVN, true>(p, pl, s);
    if (pf == 2) return launch_pipe_t<X3, CO_BLKS, CLAMP, NPB, 2, WVN, true>(p, pl, s);
    return launch_pipe_t<X3, CO_BLKS, CLAMP, NPB, 3, WVN, true>(p, pl, s);
  }
  if (pf == 1) return launch_pipe_t<X3, CO_BLKS, CLAMP, NPB, 1, WVN, false>(p, pl, s);
  if (pf == 2) return launch_pipe_t<X3, CO_BLKS, CLAMP, NPB, 2, WVN, false>(p, pl, s);
  return launch_pipe_t<X3, CO_BLKS, CLAMP, NPB, 3, WVN, false>(p, pl, s);
}

template <bool X3, int CO_BLKS>
static int launch_pipe_c(const IgemmParams& p, const IgemmPlan& pl, int pf, hipStream_t s) {
  if (pl.clamp) return pl.npb == 2 ? launch_pipe_pf<X3, CO_BLKS, true, 2>(p, pl, pf, s)
                                   : launch_pipe_pf<X3, CO_BLKS, true, 1>(p, pl, pf, s);
  return pl.npb == 2 ? launch_pipe_pf<X3, CO_BLKS, false, 2>(p, pl, pf, s)
                     : launch_pipe_pf<X3, CO_BLKS, false, 1>(p, pl, pf, s);
}


template <bool X3>
static int igemm_dispatch(const IgemmParams& p, const IgemmPlan& pl, int co_blks, int pf, bool pipe, hipStream_t s) {
  if (pipe) {
    const int rc = pl.w8 ? igemm8_dispatch<X3>(p, pl, co_blks, pf, s)
                         : (co_blks == 2 ? launch_pipe_c<X3, 2>(p, pl, pf, s) : launch_pipe_c<X3, 1>(p, pl, pf, s));
    if (rc != PCUDA_E_NOTBUILT) return rc;
    // the selected instantiation is not in this build (variants.h): the unpipelined kernel runs any plan
    if (p.red_a) PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_dgrad_bnred: the transposed-epilogue variant of this geometry is not in this build");
  }
  return co_blks == 2 ? launch_igemm_c<X3, 2>(p, pl, s) : launch_igemm_c<X3, 1>(p, pl, s);
}
