// Device side of the weight-gradient kernels; instantiated once per precision (conv_wgrad_x3.hip,
// conv_wgrad_bf16.hip) so the two halves compile in parallel.
#pragma once
#include "conv_device.h"
#include "conv_host.h"
#include "variants.h"

// ------------------------------------------------------------------------------------------
// wgrad kernel.  Block = (co-tile, 32-channel chunk, tap group) x split-K slice; loops over its
// 128-pixel tiles, keeping dW tiles [32 rows][32 ci] per tap in the accumulators.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ bf16x8 lds_tr_frag(const unsigned char* p0, const unsigned char* p1) {
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)p0);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)p1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// geometry of one 128-slot pixel tile and of its (possibly clipped) LDS input tile
struct WTile {
  int n, y0, x0, oy0, ox0, th, tw, npix;
};

template <bool CLAMP>
__device__ __forceinline__ WTile wtile_decode(const WgradParams& p, int tile) {
  WTile t;
  const int txi = tile % p.tiles_x;
  const int tmp = tile / p.tiles_x;
  const int tyi = tmp % p.tiles_y;
  t.n = tmp / p.tiles_y;
  t.y0 = tyi * p.th; t.x0 = txi * p.tw;
  t.oy0 = t.y0 * p.stride + p.dy_min; t.ox0 = t.x0 * p.stride + p.dx_min;
  t.th = p.ih_t; t.tw = p.iw_t;
  if (CLAMP) {   // LDS tile = halo tile clipped to the image (+ one zero record), as in igemm_kernel
    const int y1 = min(t.oy0 + t.th, p.in_h), x1 = min(t.ox0 + t.tw, p.in_w);
    t.oy0 = max(t.oy0, 0); t.ox0 = max(t.ox0, 0);
    t.th = max(y1 - t.oy0, 0); t.tw = max(x1 - t.ox0, 0);
  }
  t.npix = t.th * t.tw;
  return t;
}

// dZ tile [CO_TILE][128 px] in natural (pixel-contiguous) order: every thread owns CO_TILE/16 octets
// (row = item >> 4, 8 consecutive tile slots).  issue = the global loads only (values stay in
// registers), commit = mask, db partial sums, bf16 hi/lo split, LDS write.
template <int CO_TILE, int NT = 256>
__device__ __forceinline__ void zpre_issue(float (&zv)[CO_TILE * 16 / NT][8], const WgradParams& p, const WTile& t,
                                           int cot, int tid0) {
  // (opaque zero: keeps the per-element tile coordinates of the generic path from being hoisted out of the
  // tile loop -- 64 loop-invariant registers that were spilled to scratch and reloaded every tile)
  const int tid = tid0 + opaque_zero();
  const int TW = p.tw, TPIX = p.tw * p.th;
#pragma unroll
  for (int j = 0; j < CO_TILE * 16 / NT; ++j) {
    const int item = tid + NT * j;
    const int row = item >> 4, oct = item & 15;
    const int pl = oct * 8;
    const int cco = min(cot * CO_TILE + row, p.cout - 1);
    const float* planep = p.dz + (long long)t.n * p.dz_sn + (long long)cco * p.dz_sc;
    if (p.aligned4) {   // wave-uniform: TW % 8 == 0 and out_w % 8 == 0 -> an octet is inside or outside a row as a whole
      const int ty = IG_TY(pl, p.tmagic), tx = pl - ty * TW;
      const int oy = t.y0 + ty, ox = t.x0 + tx;
      const float* rowp = planep + (long long)min(oy, p.out_h - 1) * p.out_w + min(ox, p.out_w - 8);
      const float4 a = *(const float4*)rowp, b = *(const float4*)(rowp + 4);
      zv[j][0] = a.x; zv[j][1] = a.y; zv[j][2] = a.z; zv[j][3] = a.w;
      zv[j][4] = b.x; zv[j][5] = b.y; zv[j][6] = b.z; zv[j][7] = b.w;
    } else {            // any tile width: the 8 pixels of an octet may wrap to the next tile row
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ple = min(pl + e, TPIX - 1);
        const int ty = IG_TY(ple, p.tmagic), tx = ple - ty * TW;
        const int oy = t.y0 + ty, ox = t.x0 + tx;
        zv[j][e] = planep[(long long)min(oy, p.out_h - 1) * p.out_w + min(ox, p.out_w - 1)];
      }
    }
  }
}

// dZ fast path (p.aligned4: every octet is inside or outside a row as a whole): the per-lane element
// offsets inside the image are computed once per kernel, a tile adds its (uniform) origin and, on edge
// tiles, one clamp correction shared by all of the lane's octets.  (The wide raw_buffer_load builtins
// of this hipcc compile to single-dword loads, so these stay flat float4 loads + a mask at commit.)
template <int CO_TILE, int NT = 256>
struct ZConst {
  int off[CO_TILE * 16 / NT];
  int ty, tx;
};
template <int CO_TILE, int NT = 256>
__device__ __forceinline__ void zfast_init(ZConst<CO_TILE, NT>& zc, const WgradParams& p, int cot, int tid) {
  const int pl = (tid & 15) * 8;
  zc.ty = IG_TY(pl, p.tmagic); zc.tx = pl - zc.ty * p.tw;
#pragma unroll
  for (int j = 0; j < CO_TILE * 16 / NT; ++j) {
    const int co = min(cot * CO_TILE + ((tid + NT * j) >> 4), p.cout - 1);
    zc.off[j] = co * (int)p.dz_sc + zc.ty * p.out_w + zc.tx;
  }
}
template <int CO_TILE, int NT = 256>
__device__ __forceinline__ void zfast_issue(float (&zv)[CO_TILE * 16 / NT][8], const ZConst<CO_TILE, NT>& zc,
                                            const WgradParams& p, const WTile& t) {
  const float* tb = p.dz + (long long)t.n * p.dz_sn + (long long)(t.y0 * p.out_w + t.x0);   // uniform
  const int cy = min(t.y0 + zc.ty, p.out_h - 1) - (t.y0 + zc.ty);   // <= 0 on edge tiles only
  const int cx = min(t.x0 + zc.tx, p.out_w - 8) - (t.x0 + zc.tx);
  const int dl = cy * p.out_w + cx;
#pragma unroll
  for (int j = 0; j < CO_TILE * 16 / NT; ++j) {
    const float* rp = tb + (zc.off[j] + dl);
    const float4 a = *(const float4*)rp, b = *(const float4*)(rp + 4);
    zv[j][0] = a.x; zv[j][1] = a.y; zv[j][2] = a.z; zv[j][3] = a.w;
    zv[j][4] = b.x; zv[j][5] = b.y; zv[j][6] = b.z; zv[j][7] = b.w;
  }
}

template <bool X3, int CO_TILE, int NT = 256>
__device__ __forceinline__ void zpre_commit(float (&zv)[CO_TILE * 16 / NT][8], const WgradParams& p, const WTile& t,
                                            int cot, int tid, unsigned char* __restrict__ Zhi,
                                            unsigned char* __restrict__ Zlo, bool do_db,
                                            float (&dbacc)[CO_TILE * 16 / NT], bool premasked) {
  tid += opaque_zero();   // see zpre_issue
  const int TW = p.tw, TPIX = p.tw * p.th;
#pragma unroll
  for (int j = 0; j < CO_TILE * 16 / NT; ++j) {
    const int item = tid + NT * j;
    const int row = item >> 4, oct = item & 15;
    const int pl = oct * 8;
    const bool cok = cot * CO_TILE + row < p.cout;
    float(&v)[8] = zv[j];
    if (premasked) {
    } else if (p.aligned4) {
      const int ty = IG_TY(pl, p.tmagic), tx = pl - ty * TW;
      const bool ok = cok & (pl < TPIX) & (t.y0 + ty < p.out_h) & (t.x0 + tx + 8 <= p.out_w);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ok ? v[e] : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ple = min(pl + e, TPIX - 1);
        const int ty = IG_TY(ple, p.tmagic), tx = ple - ty * TW;
        const bool ok = cok & (pl + e < TPIX) & (t.y0 + ty < p.out_h) & (t.x0 + tx < p.out_w);
        v[e] = ok ? v[e] : 0.f;
      }
    }
    if (do_db) dbacc[j] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    uint4 hi, lo;
    if (X3) {
      split2(v[0], v[1], hi.x, lo.x); split2(v[2], v[3], hi.y, lo.y);
      split2(v[4], v[5], hi.z, lo.z); split2(v[6], v[7], hi.w, lo.w);
      *(uint4*)(Zlo + row * WG_ZROW + oct * 16) = lo;
    } else {
      hi.x = pack_bf16x2(v[0], v[1]); hi.y = pack_bf16x2(v[2], v[3]);
      hi.z = pack_bf16x2(v[4], v[5]); hi.w = pack_bf16x2(v[6], v[7]);
    }
    *(uint4*)(Zhi + row * WG_ZROW + oct * 16) = hi;
  }
}

// MFMA phase of one tile: acc[ti] += dZ[32 rows][128 px] * X[128 px (shifted by tap ti)][32 ci].
// Every wave runs MAXT taps unconditionally (tap tables are clamped, surplus accumulators are never
// stored): one straight-line body in which the LDS reads of the NEXT (k-step, tap) are issued before
// the three MFMAs of the current one (pinned with sched_barrier; LDS returns in order, so the wait
// in front of the MFMAs leaves the four new reads outstanding).
// MODE 0: TW % 16 == 0 (a k-step's 16 pixel slots share one tile row: the k-step part of the address
//         is scalar), 1: any tile width, 2: clamped LDS tile (per-tap bounds test).
template <bool X3, int MAXT, int MODE>
__device__ __forceinline__ void wgrad_mfma_phase(const WgradParams& p, f32x16 (&acc)[MAXT],
                                                 const unsigned char* __restrict__ Xhi,
                                                 const unsigned char* __restrict__ Xlo,
                                                 const unsigned char* __restrict__ Zhi,
                                                 const unsigned char* __restrict__ Zlo, int cb, int lane,
                                                 const int (&tap_off)[MAXT], const int (&tap_dy)[MAXT],
                                                 const int (&tap_dx)[MAXT], const WTile& t) {
  const int r = lane & 31, h = lane >> 5;
  // transposed-read lane roles: 16-lane group g -> k half (g>>1), column block (g&1)
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int TW = p.tw, TPIX = p.tw * p.th;
  const int colb = ((g & 1) * 16 + 4 * tp) * 2;
  int rb0[2];
#pragma unroll
  for (int sel = 0; sel < 2; ++sel) rb0[sel] = ((8 * (g >> 1) + 4 * sel + tq) * p.stride) * IG_REC_BYTES + colb;
  auto xaddr = [&](int ks, int ti, int& r0, int& r1) {
    if (MODE == 0) {
      const int s0 = ks * 16;
      const int ry = IG_TY(s0, p.tmagic), rx = s0 - ry * TW;
      const int soff = ((ry * p.stride) * p.iw_t + rx * p.stride) * IG_REC_BYTES + tap_off[ti];   // scalar
      r0 = rb0[0] + soff; r1 = rb0[1] + soff;
      return;
    }
    int rty[2], rtx[2];
#pragma unroll
    for (int sel = 0; sel < 2; ++sel) {
      const int pl = min(ks * 16 + 8 * (g >> 1) + 4 * sel + tq, TPIX - 1);   // idle slots carry dZ = 0
      rty[sel] = IG_TY(pl, p.tmagic); rtx[sel] = pl - rty[sel] * TW;
    }
    int ra[2];
#pragma unroll
    for (int sel = 0; sel < 2; ++sel) {
      if (MODE == 2) {
        const int gy = (t.y0 + rty[sel]) * p.stride + tap_dy[ti], gx = (t.x0 + rtx[sel]) * p.stride + tap_dx[ti];
        const bool ok = ((unsigned)gy < (unsigned)p.in_h) & ((unsigned)gx < (unsigned)p.in_w);
        ra[sel] = (ok ? (gy - t.oy0) * t.tw + (gx - t.ox0) : t.npix) * IG_REC_BYTES + colb;
      } else {
        ra[sel] = ((rty[sel] * p.stride) * p.iw_t + rtx[sel] * p.stride) * IG_REC_BYTES + colb + tap_off[ti];
      }
    }
    r0 = ra[0]; r1 = ra[1];
  };
  const int abase = (cb * 32 + r) * WG_ZROW + h * 16;
  bf16x8 ah = lds_frag(Zhi + abase), al = ah;
  if (X3) al = lds_frag(Zlo + abase);
  bf16x8 bh, bl;
  {
    int r0, r1;
    xaddr(0, 0, r0, r1);
    bh = lds_tr_frag(Xhi + r0, Xhi + r1);
    bl = bh;
    if (X3) bl = lds_tr_frag(Xlo + r0, Xlo + r1);
  }
#pragma unroll 2
  for (int ks = 0; ks < 8; ++ks) {
    bf16x8 ahn = ah, aln = al;
#pragma unroll
    for (int ti = 0; ti < MAXT; ++ti) {
      bf16x8 bhn, bln = bh;
      {   // operands of the next (k-step, tap); the last k-step re-reads its own (harmless)
        const int nks = (ti + 1 < MAXT) ? ks : min(ks + 1, 7);
        const int nti = (ti + 1 < MAXT) ? ti + 1 : 0;
        int r0, r1;
        xaddr(nks, nti, r0, r1);
        bhn = lds_tr_frag(Xhi + r0, Xhi + r1);
        if (X3) bln = lds_tr_frag(Xlo + r0, Xlo + r1);
        if (ti + 1 == MAXT) {
          ahn = lds_frag(Zhi + abase + nks * 32);
          if (X3) aln = lds_frag(Zlo + abase + nks * 32);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (X3) {
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ti], 0, 0, 0);
        acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ti], 0, 0, 0);
      }
      acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ti], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bh = bhn; bl = bln;
    }
    ah = ahn; al = aln;
  }
}

// PF = 0: every tile is loaded, converted and consumed in turn (any input-tile size).
// PF >= 1: software pipeline -- the loads of tile i+1 (32*PF dwords of X and 8*CO_TILE/16 of dZ per lane)
// are issued right after tile i has been committed to LDS and stay in flight, in registers, during
// tile i's MFMA phase; needs input tiles of <= 256*PF pixels.  (Unpipelined, the load, convert and
// MFMA phases of the 1-2 resident workgroups simply added up: 5-10x off both the HBM and the MFMA
// bound on every 3x3 layer.)
// MODE: see wgrad_mfma_phase (2 = clamped LDS tile)
// NW = 4 waves, or 8 (one 512-thread workgroup per CU: two waves per SIMD for the 16-tap stride-2 layers, whose
// LDS tiles allow only one workgroup per CU; per lane the staging work, its registers and the accumulators halve)
template <bool X3, int CO_BLKS, int MODE, int TAPS_MAX, int PF, int NW = 4, bool XQ = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 || (TAPS_MAX <= 9 && PF <= 1)) ? 2 : 1) void wgrad_kernel(const WgradParams p, const int x_cap, float* db_partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool CLAMP = MODE == 2;
  constexpr int CO_TILE = 32 * CO_BLKS;
  constexpr int NT = 64 * NW, ZJ = CO_TILE * 16 / NT;
  static_assert(NW == 4 || PF > 0, "the eight-wave variant is pipelined");
  constexpr int NWT = NW / CO_BLKS;            // waves sharing one row block
  constexpr int MAXT = (TAPS_MAX + NWT - 1) / NWT;   // taps per wave (block handles <= TAPS_MAX taps)
  constexpr int XPF = PF > 0 ? PF : 1;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tap ranges and tap offsets stay in SGPRs

  // XCD-aware block -> work mapping (round 3).  Workgroups are dealt round-robin over the 8 XCDs (ids b and b + 8 share
  // one; each XCD has its own L2).  The workgroups of one split-K slice walk the SAME pixel tiles -- X is re-read once per
  // co-tile, dZ once per chunk -- so the work items, ordered slice-major, are cut into 8 contiguous ranges, one per XCD:
  // whole slices (or a contiguous range of one slice's co-tiles) then share an L2 instead of each XCD fetching all of it.
  int bx = blockIdx.x, kslice = blockIdx.y;
  if (p.xcd_items) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int item = (lin & 7) * p.xcd_items + (lin >> 3);
    kslice = item / (int)gridDim.x;
    bx = item - kslice * (int)gridDim.x;
  }
  const int tgidx = bx % p.tap_groups;
  const int chunk = (bx / p.tap_groups) % p.n_chunks;
  const int cot = bx / (p.tap_groups * p.n_chunks);
  const int t_begin = tgidx * p.ntaps;
  const int tcount = min(p.ntaps, p.ntaps_total - t_begin);
  const int cb = w % CO_BLKS, wsub = w / CO_BLKS;
  const int share = (tcount + NWT - 1) / NWT;
  const int my_t0 = t_begin + wsub * share;
  const int my_cnt = max(0, min(share, tcount - wsub * share));

  unsigned char* Xhi = smem;
  unsigned char* Xlo = smem + (size_t)x_cap * IG_REC_BYTES;
  unsigned char* Zhi = smem + (size_t)(X3 ? 2 : 1) * x_cap * IG_REC_BYTES;
  unsigned char* Zlo = Zhi + (size_t)CO_TILE * WG_ZROW;

  const int ntiles = p.n * p.tiles_y * p.tiles_x;
  // Which tiles a slice sums is free (split-K partial sums).  Plain: a contiguous range.  With the XCD-aware mapping and
  // ksplit a multiple of 8 (round 3): XCD x owns the contiguous eighth [x T / 8, (x + 1) T / 8) of the tiles and its
  // S = ksplit / 8 slices take every S-th tile of it, so that the workgroups resident on the XCD at any moment work
  // on ADJACENT tiles: the halo rows and the neighbouring 128-byte lines of a tile's input rows -- 3 lines touched per
  // 32-pixel row segment, 6 rows per 4 -- are then in that XCD's L2 when the neighbour asks for them.  (Walking a
  // contiguous range per workgroup, 64 workgroups x 63 KB per tile step turned the 4 MB L2 over between two
  // consecutive tiles of one workgroup: the weight gradient of 32->32 at 256x256 fetched 2.5x its algorithmic bytes.)
  int tile_lo = (int)((long long)kslice * ntiles / p.ksplit);
  int tile_hi = (int)((long long)(kslice + 1) * ntiles / p.ksplit);
  int tile_step = 1;
  if (p.xcd_items && p.xcd_slices) {
    const int xcd = (blockIdx.y * gridDim.x + blockIdx.x) & 7, S = p.xcd_slices;
    const int T_lo = (int)((long long)xcd * ntiles / 8), T_hi = (int)((long long)(xcd + 1) * ntiles / 8);
    tile_lo = T_lo + (kslice - xcd * S);
    tile_step = S;
    tile_hi = tile_lo < T_hi ? tile_lo + ((T_hi - tile_lo + S - 1) / S) * S : tile_lo;    // tile_lo + count * S
  }
  const bool do_db = (db_partial != nullptr) && chunk == 0 && tgidx == 0;

  f32x16 acc[MAXT];
#pragma unroll
  for (int ti = 0; ti < MAXT; ++ti)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ti][i] = 0.f;
  float dbacc[ZJ];
#pragma unroll
  for (int j = 0; j < ZJ; ++j) dbacc[j] = 0.f;

  const int cvalid = min(32, p.cin - chunk * 32);
  const int ngroups = (cvalid + 7) >> 3;

  // per-tap LDS offsets / image offsets, fetched once (indexing p.dy[] inside the MFMA loop cost two
  // dependent global loads per tap and k-step)
  int tap_off[MAXT], tap_dy[MAXT], tap_dx[MAXT];
#pragma unroll
  for (int ti = 0; ti < MAXT; ++ti) {
    const int t = min(my_t0 + min(ti, max(my_cnt - 1, 0)), p.ntaps_total - 1);   // surplus slots repeat a valid tap
    tap_dy[ti] = p.dy[t]; tap_dx[ti] = p.dx[t];
    tap_off[ti] = ((tap_dy[ti] - p.dy_min) * p.iw_t + (tap_dx[ti] - p.dx_min)) * IG_REC_BYTES;
  }

  // dZ is prefetched like X except where that does not fit the 256 registers of two waves per SIMD (the
  // 64-row, 9-tap variant spilled 52 dwords inside the tile loop): there dZ is loaded at commit time
  constexpr bool ZPRE = PF > 0 && !(CO_BLKS == 2 && TAPS_MAX == 9 && PF == 1 && MODE != 0);
  XFast<XPF> xpre;
  float zv[ZJ][8];
  ZConst<CO_TILE, NT> zc;
  // (uniform; MODE 0 is only dispatched for float4-readable dZ rows: the scalar dZ path is compiled out of it)
  const bool zfast = PF > 0 && (MODE == 0 || p.aligned4);
  if (zfast) zfast_init<CO_TILE, NT>(zc, p, cot, tid);
  WTile cur = wtile_decode<CLAMP>(p, min(tile_lo, ntiles - 1));
  if (PF > 0 && tile_lo < tile_hi) {
    // (XQ: float4 staging of rows of 4k pixels, as in the forward kernels -- a quarter of the load instructions)
    if (XQ) xq_issue<XPF, NT>(xpre, p.x, cur.n, p.cin, chunk, p.in_h, p.in_w, cur.oy0, cur.ox0, cur.th, cur.tw, tid);
    else xfast_issue<XPF, NT>(xpre, p.x, cur.n, p.cin, chunk, p.in_h, p.in_w, p.in_shift, p.in_row, cur.oy0, cur.ox0, cur.tw,
                              cur.npix, ngroups, tid);
    if (ZPRE) {
      if (zfast) zfast_issue<CO_TILE, NT>(zv, zc, p, cur);
      else zpre_issue<CO_TILE, NT>(zv, p, cur, cot, tid);
    }
  }

  for (int tile = tile_lo; tile < tile_hi; tile += tile_step) {
    if (PF == 0) cur = wtile_decode<CLAMP>(p, tile);
    const int oy0 = cur.oy0, ox0 = cur.ox0, tw = cur.tw, npix = cur.npix;
    __syncthreads();   // the previous tile's MFMA phase is done with the LDS tiles
    if (PF > 0) {
      if (XQ) xq_commit<X3, XPF, NT>(xpre, Xhi, Xlo, p.x, p.cin, chunk, ox0, cur.th, tw, 4, tid);
      else xfast_commit<X3, XPF, NT>(xpre, Xhi, Xlo, p.x, p.cin, chunk, npix, ngroups, 4, tid);
      if (!ZPRE) {
        if (zfast) zfast_issue<CO_TILE, NT>(zv, zc, p, cur);
        else zpre_issue<CO_TILE, NT>(zv, p, cur, cot, tid);
      }
    } else {
      stage_x_chunk<X3, 1>(Xhi, Xlo, p.x, cur.n, p.cin, chunk, p.in_h, p.in_w, p.in_shift, p.in_row, oy0, ox0, cur.th,
                           tw, ngroups, 4, tid);
      zpre_issue<CO_TILE, NT>(zv, p, cur, cot, tid);
    }
    if (CLAMP && tid < 5) {
      *(uint4*)(Xhi + (size_t)npix * IG_REC_BYTES + tid * 16) = make_uint4(0, 0, 0, 0);
      if (X3) *(uint4*)(Xlo + (size_t)npix * IG_REC_BYTES + tid * 16) = make_uint4(0, 0, 0, 0);
    }
    zpre_commit<X3, CO_TILE, NT>(zv, p, cur, cot, tid, Zhi, Zlo, do_db, dbacc, false);
    __syncthreads();
    WTile nxt = cur;
    if (PF > 0 && tile + tile_step < tile_hi) {   // (the block's last tile skips the loads altogether)
      nxt = wtile_decode<CLAMP>(p, tile + tile_step);
      if (XQ) xq_issue<XPF, NT>(xpre, p.x, nxt.n, p.cin, chunk, p.in_h, p.in_w, nxt.oy0, nxt.ox0, nxt.th, nxt.tw, tid);
      else xfast_issue<XPF, NT>(xpre, p.x, nxt.n, p.cin, chunk, p.in_h, p.in_w, p.in_shift, p.in_row, nxt.oy0, nxt.ox0, nxt.tw,
                                nxt.npix, ngroups, tid);
      if (ZPRE) {
        if (zfast) zfast_issue<CO_TILE, NT>(zv, zc, p, nxt);
        else zpre_issue<CO_TILE, NT>(zv, p, nxt, cot, tid);
      }
    }

    // ---- MFMA phase
#ifdef PCUDA_CLK_DEBUG
    if (!(p.dbg & 64))      // (timing experiments: skip the MFMA phase)
#endif
      wgrad_mfma_phase<X3, MAXT, MODE>(p, acc, Xhi, Xlo, Zhi, Zlo, cb, lane, tap_off, tap_dy, tap_dx, cur);
    cur = nxt;
  }

  // partial slabs: partial[kslice][tap][co][ci] -- ci (the lane index) innermost, so every accumulator
  // register stores two 128-B segments; the OIHW transpose happens once, in the reduce kernel
#pragma unroll
  for (int ti = 0; ti < MAXT; ++ti) {
    if (ti < my_cnt) {
      const int t = my_t0 + ti;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = cot * CO_TILE + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int ci = chunk * 32 + r;
        if (co < p.cout && ci < p.cin)
          p.partial[(((long long)kslice * p.ntaps_total + t) * p.cout + co) * p.cin + ci] = acc[ti][i];
      }
    }
  }
  if (do_db) {
#pragma unroll
    for (int j = 0; j < ZJ; ++j) {
      float s = dbacc[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const int row = (tid + NT * j) >> 4;
      const int co = cot * CO_TILE + row;
      if ((tid & 15) == 0 && co < p.cout) db_partial[(long long)kslice * p.cout + co] = s;
    }
  }
}


template <bool X3, int CO_BLKS, int MODE, int TAPS_MAX, int PF, int NW, bool XQ>
static int launch_wgrad_q(const WgradParams& p, int x_cap, size_t lds, float* dbp, dim3 grid, hipStream_t s) {
  constexpr unsigned vkey = wgrad_key(X3, CO_BLKS, MODE, TAPS_MAX, PF, NW, XQ);
  variant_log("wgrad", vkey);
  if constexpr (!wgrad_built(vkey)) {
    // not in this build (variants.h): the same plan on the four-wave kernel without register prefetch (always built)
    variant_fallback_note("wgrad_kernel", vkey);
    return launch_wgrad_q<X3, CO_BLKS, MODE, TAPS_MAX, 0, NW, false>(p, x_cap, lds, dbp, grid, s);
  } else {
  auto kern = wgrad_kernel<X3, CO_BLKS, MODE, TAPS_MAX, PF, NW, XQ>;
  static DeviceOnce lds_opt;
  if (const unsigned long long devbit = lds > 32 * 1024 ? lds_opt.pending() : 0ull) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD);
    if (e != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "wgrad: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    lds_opt.mark(devbit);
  }
  note_kernel(PF == 0 ? "wgrad_generic" : (NW == 8 ? "wgrad8" : "wgrad"));
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, s, p, x_cap, dbp);
  PCUDA_CHECK_LAUNCH("wgrad_kernel");
  return PCUDA_OK;
  }
}

template <bool X3, int CO_BLKS, int MODE, int TAPS_MAX, int PF, int NW = 4>
static int launch_wgrad_t(const WgradParams& p, int x_cap, size_t lds, float* dbp, dim3 grid, hipStream_t s) {
  if (PF > 0 && p.xq) return launch_wgrad_q<X3, CO_BLKS, MODE, TAPS_MAX, PF, NW, (PF > 0)>(p, x_cap, lds, dbp, grid, s);
  return launch_wgrad_q<X3, CO_BLKS, MODE, TAPS_MAX, PF, NW, false>(p, x_cap, lds, dbp, grid, s);
}


template <bool X3>
static int wgrad_dispatch(const WgradParams& p, int co_blks, bool clamp, int taps_max, int pf, int x_cap, size_t lds,
                          float* dbp, dim3 grid, hipStream_t s) {
  const int mode = clamp ? 2 : ((p.tw16 && p.aligned4) ? 0 : 1);
  if (pf >= 100) {   // eight waves (16-tap groups): pf - 100 = staging slots per lane at 512 lanes
#define WG8_PF(CB_, MD_) (pf == 101 ? launch_wgrad_t<X3, CB_, MD_, 16, 1, 8>(p, x_cap, lds, dbp, grid, s) : launch_wgrad_t<X3, CB_, MD_, 16, 2, 8>(p, x_cap, lds, dbp, grid, s))
#define WG8_MD(CB_) (mode == 2 ? WG8_PF(CB_, 2) : mode == 0 ? WG8_PF(CB_, 0) : WG8_PF(CB_, 1))
    return co_blks == 2 ? WG8_MD(2) : WG8_MD(1);
#undef WG8_MD
#undef WG8_PF
  }
#define WG_PF(CB_, MD_, TM_)                                                                     \
  (pf == 1 ? launch_wgrad_t<X3, CB_, MD_, TM_, 1>(p, x_cap, lds, dbp, grid, s)                  \
   : pf == 3 ? launch_wgrad_t<X3, CB_, MD_, TM_, 3>(p, x_cap, lds, dbp, grid, s)                \
             : launch_wgrad_t<X3, CB_, MD_, TM_, 0>(p, x_cap, lds, dbp, grid, s))
#define WG_TM(CB_, MD_) (taps_max == 1 ? WG_PF(CB_, MD_, 1) : taps_max == 9 ? WG_PF(CB_, MD_, 9) : WG_PF(CB_, MD_, 16))
#define WG_MD(CB_) (mode == 2 ? WG_TM(CB_, 2) : mode == 0 ? WG_TM(CB_, 0) : WG_TM(CB_, 1))
  return co_blks == 2 ? WG_MD(2) : WG_MD(1);
#undef WG_MD
#undef WG_TM
#undef WG_PF
}
