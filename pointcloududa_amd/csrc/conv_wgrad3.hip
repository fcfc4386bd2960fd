// Weight gradient of the segmenter's 3x3 / stride-1 / pad-1 layers on aligned maps (rows of 32k pixels, 4k rows,
// channel counts in multiples of 32): the layers that hold most of the weight-gradient time (unet.py:23,27 encoder
// blocks, :116,122 decoder blocks at 256x256 ... 32x32).  Same algorithm, slab format and reduce as the generic
// wgrad_kernel (conv_wgrad_impl.h) -- dW[co][ci][tap] = sum over pixel tiles of dZ[co][128 px] . X[128 px + tap][ci],
// bf16x3 MFMA, X through pixel records and ds_read_b64_tr_b16, split-K slabs summed in a fixed order -- but with the
// geometry fixed at compile time, because the generic kernel spent ~10 vector instructions per MFMA on it
// (profiles/r02_mfma_counters.csv: valu_per_mfma 8.1-12.7; its ISA: 940 VALU + 600 SALU of staging per 120 MFMA):
//   * tile = 32 x 4 output pixels, the staged input tile = 6 rows x 10 ALIGNED quads (40 px): every lane stores all
//     four pixels of its quad, no per-pixel predicates; lanes outside the image are zeroed by one lane mask built from
//     four precomputed ballots and the tile's (uniform) border flags, on border tiles only;
//   * every LDS address of the MFMA phase is a per-kernel VGPR + an instruction immediate (k-step, plane and row
//     offsets are constants): no address arithmetic inside the phase;
//   * tiles are walked incrementally (no divisions), plane pointers advance by scalar adds;
//   * dZ rows are read with float4 loads at per-kernel lane offsets, never masked (the maps are whole tiles).
#include "conv_device.h"
#include "conv_host.h"

namespace {

constexpr int W3_TW = 32, W3_TH = 4;
constexpr int W3_XW = 40;                       // staged pixels per input row: quads ox0-4 .. ox0+35
constexpr int W3_XROWS = W3_TH + 2;
constexpr int W3_XREC = W3_XW * W3_XROWS;       // 240 pixel records
constexpr int W3_XPLANE = W3_XREC * IG_REC_BYTES;   // 19200 bytes per bf16 plane
constexpr int W3_ITEMS = W3_XROWS * (W3_XW / 4);    // 60 (row, quad) items per channel group

struct W3Params {
  pcuda_src x;
  int cin, H, W;
  const float* dz; long long dz_sn, dz_sc;
  int cout;
  int tiles_x, tiles_y, n;
  int ksplit, n_chunks;
  float* partial;       // [ksplit * KH][9][cout][cin]
  float* db_partial;    // [ksplit * KH][cout] or NULL
  int base, xcd;        // workgroups per split-K slice (co-tiles x chunks); 1: XCD-aware block mapping
  int dbg;              // PCUDA_W3DBG (timing experiments): 1 no X loads, 2 no dZ loads, 4 no MFMA phase, 8 no commit
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p) {
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)p);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(p + 4 * IG_REC_BYTES));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// k-step ks covers tile row ks >> 1, pixels 16 * (ks & 1) .. + 15: byte offset of its first record
__device__ __forceinline__ constexpr int w3_ks_off(int ks) { return ((ks >> 1) * W3_XW + (ks & 1) * 16) * IG_REC_BYTES; }

// MFMA phase over k-steps [KS0, KS0 + NKS) and NT taps: acc[ti] += dZ[32 rows][16 px] . X[16 px (+ tap ti)][32 ci].
// The operands of the next (k-step, tap) are requested before the MFMAs of the current one.
template <bool X3, int NT, int KS0, int NKS, int CO_T>
__device__ __forceinline__ void w3_mfma(f32x16 (&acc)[5], const unsigned char* __restrict__ Xs,
                                        const unsigned char* __restrict__ Zs, const int (&xa)[5], int za) {
  constexpr int ZLO = CO_T * WG_ZROW;
  bf16x8 ah = lds_frag(Zs + za + KS0 * 32), al = ah;
  if (X3) al = lds_frag(Zs + za + KS0 * 32 + ZLO);
  bf16x8 bh = tr_frag(Xs + xa[0] + w3_ks_off(KS0)), bl = bh;
  if (X3) bl = tr_frag(Xs + xa[0] + w3_ks_off(KS0) + W3_XPLANE);
#pragma unroll
  for (int k = 0; k < NKS; ++k) {
    const int ks = KS0 + k;
    bf16x8 ahn = ah, aln = al;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      bf16x8 bhn = bh, bln = bl;
      const bool last = (ti + 1 == NT) && (k + 1 == NKS);
      if (!last) {
        const int nks = (ti + 1 < NT) ? ks : ks + 1;
        const int nti = (ti + 1 < NT) ? ti + 1 : 0;
        bhn = tr_frag(Xs + xa[nti] + w3_ks_off(nks));
        if (X3) bln = tr_frag(Xs + xa[nti] + w3_ks_off(nks) + W3_XPLANE);
        if (ti + 1 == NT) {
          ahn = lds_frag(Zs + za + nks * 32);
          if (X3) aln = lds_frag(Zs + za + nks * 32 + ZLO);
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

// RB row blocks of 32 output channels per workgroup; KH = 1: waves = (row block, tap half); RB = 1 uses KH = 2: waves =
// (tap half, k-step half), the two k-step halves writing separate split-K slices.  4 waves, 2 workgroups per CU.
template <bool X3, int RB, int KH>
__global__ __launch_bounds__(64 * RB * 2 * KH, 2) void wgrad3_kernel(const W3Params p) {
  constexpr int NW = RB * 2 * KH, NT = 64 * NW;  // 4 waves (two workgroups per CU) or 8 (RB = 4: one per CU)
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CO_T = 32 * RB;
  constexpr int ZJ = CO_T * 16 / NT;             // dZ octets per thread
  unsigned char* Xs = smem;                                            // hi plane, lo plane at + W3_XPLANE
  unsigned char* Zs = smem + (X3 ? 2 : 1) * W3_XPLANE;                 // hi rows, lo rows at + CO_T * WG_ZROW

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = w % RB, tsub = (w / RB) & 1, khalf = w / (RB * 2);
  // XCD-aware block -> work mapping.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each
  // XCD has its own L2).  XCD x owns the contiguous eighth [x T / 8, (x + 1) T / 8) of the pixel tiles; its S = ksplit / 8
  // split-K slices take every S-th tile of it (which tiles a slice sums is free), in block order -- so the workgroups
  // resident on an XCD at any moment work on ADJACENT tiles and on every (co-tile, chunk) of them: the re-reads of X per
  // co-tile and of dZ per chunk, the halo rows and the neighbouring 128-byte lines of a tile's input rows (3 lines touched
  // per 32-pixel row segment: scripts/micro/fetch_calib.hip) hit that XCD's L2 instead of going out to the fabric again.
  int bwork, kslice, xcd = 0, jloc = 0;
  if (p.xcd) {
    xcd = blockIdx.x & 7;
    const int i = blockIdx.x >> 3;
    jloc = i / p.base;                       // slice index inside the XCD
    kslice = xcd * (p.ksplit >> 3) + jloc;
    bwork = i % p.base;
  } else {
    kslice = blockIdx.x / p.base;
    bwork = blockIdx.x % p.base;
  }
  const int chunk = bwork % p.n_chunks, cot = bwork / p.n_chunks;

  // ---- X staging constants: wave w stages channel group w (8 channels); lane = (row, quad) item
  const int c0 = chunk * 32;
  const bool first = c0 < p.x.c1;
  const float* xsrc = first ? p.x.p1 : p.x.p2;
  const long long x_sn = first ? p.x.sn1 : p.x.sn2, x_sc = first ? p.x.sc1 : p.x.sc2;
  const float* scp = first ? p.x.scale1 : p.x.scale2;
  const float* shp = first ? p.x.shift1 : p.x.shift2;
  // (eight waves: waves w and w + 4 share channel group w & 3 and split its items)
  const int wg = w & 3;
  const int cl0 = (first ? c0 : c0 - p.x.c1) + wg * 8;
  constexpr int IPW = NW == 8 ? W3_ITEMS / 2 : W3_ITEMS;      // items per wave
  const bool xact = lane < IPW;
  const int item = xact ? lane + (NW == 8 ? (w >> 2) * IPW : 0) : 0;
  const int iy = (item * 6554) >> 16, q = item - iy * 10;            // item / 10 (exact below 16384)
  const int x_lane = ((iy - 1) * p.W + 4 * q - 4) * 4;                // byte offset of the quad relative to the tile origin
  const int xw = (iy * W3_XW + 4 * q) * IG_REC_BYTES + wg * 16;       // LDS byte address of the quad's first record
  // lanes whose quad lies outside the image when the tile touches the top / bottom / left / right border
  const unsigned long long m_top = __builtin_amdgcn_ballot_w64(iy == 0), m_bot = __builtin_amdgcn_ballot_w64(iy == W3_XROWS - 1);
  const unsigned long long m_lft = __builtin_amdgcn_ballot_w64(q == 0), m_rgt = __builtin_amdgcn_ballot_w64(q == 9);
  const unsigned long long m_idle = __builtin_amdgcn_ballot_w64(!xact);
  float sc[8], sh[8];
  const bool affine = scp != nullptr;
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = affine ? scp[cl0 + j] : 1.f; sh[j] = affine ? shp[cl0 + j] : 0.f; }

  // ---- dZ staging constants: octet item = tid + NT j -> row (tid >> 4) + (NT / 16) j, octet tid & 15
  const int zrow0 = tid >> 4, oct = tid & 15;
  const int co_base = cot * CO_T;
  const unsigned z_lane = (unsigned)(((long long)(co_base + zrow0) * p.dz_sc + (oct >> 2) * p.W + (oct & 3) * 8) * 4);
  const unsigned z_step = (unsigned)((long long)(NT / 16) * p.dz_sc * 4);
  const int zw = zrow0 * WG_ZROW + oct * 16;
  const bool do_db = p.db_partial != nullptr && chunk == 0;

  // ---- MFMA phase addresses
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int xr = (8 * (g >> 1) + tq) * IG_REC_BYTES + ((g & 1) * 16 + 4 * tp) * 2;
  int xa[5];
#pragma unroll
  for (int ti = 0; ti < 5; ++ti) {
    const int t = min(tsub * 5 + ti, 8);                                // (the second half's fifth slot is never used)
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    xa[ti] = xr + ((dy + 1) * W3_XW + dx + 4) * IG_REC_BYTES;
  }
  const int za = (cb * 32 + r) * WG_ZROW + h * 16;

  f32x16 acc[5];
#pragma unroll
  for (int ti = 0; ti < 5; ++ti)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ti][i] = 0.f;
  float dbacc[ZJ];
#pragma unroll
  for (int j = 0; j < ZJ; ++j) dbacc[j] = 0.f;

  // ---- tiles of this split-K slice: first, step, count; (image, tile row, tile column) advanced without divisions
  const int tiles_img = p.tiles_x * p.tiles_y, ntiles = p.n * tiles_img;
  int tile_first, tile_step, tile_count;
  if (p.xcd) {
    const int S = p.ksplit >> 3;
    const int T_lo = (int)((long long)xcd * ntiles / 8), T_hi = (int)((long long)(xcd + 1) * ntiles / 8);
    tile_first = T_lo + jloc;
    tile_step = S;
    tile_count = tile_first < T_hi ? (T_hi - tile_first + S - 1) / S : 0;
  } else {
    tile_first = (int)((long long)kslice * ntiles / p.ksplit);
    tile_step = 1;
    tile_count = (int)((long long)(kslice + 1) * ntiles / p.ksplit) - tile_first;
  }
  int tn = tile_first / tiles_img, trem = tile_first - tn * tiles_img;
  int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
  const int step_n = tile_step / tiles_img, step_rem = tile_step - step_n * tiles_img;
  const int step_y = step_rem / p.tiles_x, step_x = step_rem - step_y * p.tiles_x;

  f32x4 xv[8];
  f32x4 zv[ZJ][2];
  unsigned long long out_mask = 0;
  auto issue = [&]() {
    const int y0 = tyi * W3_TH, x0 = txi * W3_TW;
    unsigned long long m = m_idle;
    if (tyi == 0) m |= m_top;
    if (tyi == p.tiles_y - 1) m |= m_bot;
    if (txi == 0) m |= m_lft;
    if (txi == p.tiles_x - 1) m |= m_rgt;
    out_mask = m;
    const bool outside = (m >> lane) & 1;
    const unsigned voff = outside ? 0u : (unsigned)(x_lane + (y0 * p.W + x0) * 4);
    const char* plane = (const char*)(xsrc + (long long)tn * x_sn + (long long)cl0 * x_sc);
    if (!(p.dbg & 1)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xv[j] = *(const f32x4*)(plane + voff);
        plane += x_sc * 4;
      }
    }
    const char* zb = (const char*)(p.dz + (long long)tn * p.dz_sn + (long long)(y0 * p.W + x0));
    unsigned zo = z_lane;
    if (!(p.dbg & 2)) {
#pragma unroll
      for (int j = 0; j < ZJ; ++j) {
        zv[j][0] = *(const f32x4*)(zb + zo);
        zv[j][1] = *(const f32x4*)(zb + zo + 16);
        zo += z_step;
      }
    }
  };
  auto advance = [&]() {      // (n, y, x) += (step_n, step_y, step_x) with carries
    txi += step_x; tyi += step_y; tn += step_n;
    if (txi >= p.tiles_x) { txi -= p.tiles_x; ++tyi; }
    if (tyi >= p.tiles_y) { tyi -= p.tiles_y; ++tn; }
  };

  if (tile_count > 0) issue();
  for (int it = 0; it < tile_count; ++it) {
    __syncthreads();        // the previous tile's MFMA phase is done with the LDS tiles
    // ---- commit X: affine, zero padding, bf16 hi / lo split, one 16-byte record slice per pixel and plane
    if (!(p.dbg & 8)) {
      if (affine) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[j][e] = fmaf(xv[j][e], sc[j], sh[j]);
      }
      if (out_mask & ~m_idle) {        // (uniform) a border tile: zero padding is applied AFTER the affine
        const bool outside = (out_mask >> lane) & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[j][e] = outside ? 0.f : xv[j][e];
      }
      if (xact) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint4 hi, lo;
          if (X3) {
            split2(xv[0][e], xv[1][e], hi.x, lo.x); split2(xv[2][e], xv[3][e], hi.y, lo.y);
            split2(xv[4][e], xv[5][e], hi.z, lo.z); split2(xv[6][e], xv[7][e], hi.w, lo.w);
            *(uint4*)(Xs + xw + e * IG_REC_BYTES + W3_XPLANE) = lo;
          } else {
            hi.x = pack_bf16x2(xv[0][e], xv[1][e]); hi.y = pack_bf16x2(xv[2][e], xv[3][e]);
            hi.z = pack_bf16x2(xv[4][e], xv[5][e]); hi.w = pack_bf16x2(xv[6][e], xv[7][e]);
          }
          *(uint4*)(Xs + xw + e * IG_REC_BYTES) = hi;
        }
      }
    }
    // ---- commit dZ: split, natural (pixel-contiguous) rows
    if (!(p.dbg & 8))
#pragma unroll
    for (int j = 0; j < ZJ; ++j) {
      const f32x4 a = zv[j][0], b = zv[j][1];
      if (do_db) dbacc[j] += ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
      uint4 hi, lo;
      if (X3) {
        split2(a[0], a[1], hi.x, lo.x); split2(a[2], a[3], hi.y, lo.y);
        split2(b[0], b[1], hi.z, lo.z); split2(b[2], b[3], hi.w, lo.w);
        *(uint4*)(Zs + zw + j * (NT / 16) * WG_ZROW + CO_T * WG_ZROW) = lo;
      } else {
        hi.x = pack_bf16x2(a[0], a[1]); hi.y = pack_bf16x2(a[2], a[3]);
        hi.z = pack_bf16x2(b[0], b[1]); hi.w = pack_bf16x2(b[2], b[3]);
      }
      *(uint4*)(Zs + zw + j * (NT / 16) * WG_ZROW) = hi;
    }
    __syncthreads();
    if (it + 1 < tile_count) { advance(); issue(); }      // the next tile's loads stay in flight during the MFMA phase
    if (p.dbg & 4) continue;
    if (KH == 1) {
      if (tsub == 0) w3_mfma<X3, 5, 0, 8, CO_T>(acc, Xs, Zs, xa, za);
      else w3_mfma<X3, 4, 0, 8, CO_T>(acc, Xs, Zs, xa, za);
    } else {
      if (khalf == 0) {
        if (tsub == 0) w3_mfma<X3, 5, 0, 4, CO_T>(acc, Xs, Zs, xa, za);
        else w3_mfma<X3, 4, 0, 4, CO_T>(acc, Xs, Zs, xa, za);
      } else {
        if (tsub == 0) w3_mfma<X3, 5, 4, 4, CO_T>(acc, Xs, Zs, xa, za);
        else w3_mfma<X3, 4, 4, 4, CO_T>(acc, Xs, Zs, xa, za);
      }
    }
  }

  // ---- partial slabs [slice][tap][co][ci] (ci on the lanes: 128-byte segments), as wgrad_kernel writes them
  const int slice = kslice * KH + khalf;
  const int ntap = tsub ? 4 : 5;
#pragma unroll
  for (int ti = 0; ti < 5; ++ti) {
    if (ti < ntap) {
      const int t = tsub * 5 + ti;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = co_base + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int ci = c0 + r;
        p.partial[(((long long)slice * 9 + t) * p.cout + co) * p.cin + ci] = acc[ti][i];
      }
    }
  }
  if (do_db) {
#pragma unroll
    for (int j = 0; j < ZJ; ++j) {
      float s = dbacc[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (oct == 0) {
        const int co = co_base + zrow0 + (NT / 16) * j;
        p.db_partial[(long long)(kslice * KH) * p.cout + co] = s;
        if (KH == 2) p.db_partial[(long long)(kslice * KH + 1) * p.cout + co] = 0.f;
      }
    }
  }
}

struct W3Plan {
  int rb, kh, n_co_tiles, n_chunks, tiles_x, tiles_y, ksplit;
};

bool w3_eligible(const pcuda_conv_geom* g, const pcuda_src* x, const float* dy, long long dy_sn, long long dy_sc) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("PCUDA_NO_WGRAD3"); off = (e && atoi(e)) ? 1 : 0; }
  if (off) return false;
  if (g->k != 3 || g->stride != 1 || g->dil != 1 || g->pad != 1 || g->in_up) return false;
  static int mincout = -1;
  if (mincout < 0) { const char* e = getenv("PCUDA_W3_MINCOUT"); mincout = e ? atoi(e) : 128; }
  if ((g->out_w % W3_TW) || (g->out_h % W3_TH) || (g->cout & 63) || g->cout < mincout) return false;
  const int c1 = x->c1 < g->cin ? x->c1 : g->cin;
  if ((c1 & 31) || ((g->cin - c1) & 31)) return false;
  if (!fast_src_ok(x, g->cin)) return false;
  if (((uintptr_t)x->p1 & 15) || (x->sc1 & 3) || (x->sn1 & 3)) return false;
  if (c1 < g->cin && (((uintptr_t)x->p2 & 15) || (x->sc2 & 3) || (x->sn2 & 3))) return false;
  if (((uintptr_t)dy & 15) || (dy_sc & 3) || (dy_sn & 3)) return false;
  if (((long long)g->cout * dy_sc + (long long)g->out_h * g->out_w) * 4 >= (1ll << 31)) return false;
  if ((long long)g->in_h * g->in_w * 4 >= (1ll << 30)) return false;
  return true;
}

W3Plan w3_plan(const pcuda_conv_geom* g) {
  W3Plan w;
  // Row blocks per workgroup: 2 (four waves, two workgroups per CU).  Measured and not dispatched (the template keeps
  // them): RB = 4 with eight waves (one workgroup per CU) halves the X traffic -- loads alone 0.123 -> 0.077 ms on
  // 128->128 at 64x64 -- but its single workgroup serialises commit and MFMA phase behind one barrier pair (nothing
  // computes while it stages): 0.176 vs 0.162 ms; RB = 1 (cout = 32, k-step halves on the waves) 0.268 vs 0.235 ms of
  // the generic kernel.  Layers with cout < 128 stay on the generic kernel (64->64 at 128x128: 0.186 vs 0.176 ms).
  w.rb = 2;
  w.kh = 1;
  w.n_co_tiles = g->cout / (32 * w.rb);
  w.n_chunks = g->cin / 32;
  w.tiles_x = g->out_w / W3_TW;
  w.tiles_y = g->out_h / W3_TH;
  const long long ntiles = (long long)g->n * w.tiles_x * w.tiles_y;
  const int base = w.n_co_tiles * w.n_chunks;
  static int tgt = -1;
  if (tgt < 0) { const char* e = getenv("PCUDA_WG3_BLOCKS"); tgt = e ? atoi(e) : 1024; }
  long long ks = tgt / base;
  if (ks < 1) ks = 1;
  if (ks > ntiles) ks = ntiles;
  const long long welems = (long long)g->cout * g->cin * 9;
  while (ks > 1 && ks * w.kh * welems * 4 > (64ll << 20)) ks >>= 1;
  if (ks >= 8) ks &= ~7ll;       // multiples of 8: the XCD-aware block mapping
  w.ksplit = (int)ks;
  return w;
}

}  // namespace

size_t wgrad3_workspace(const pcuda_conv_geom* g) {
  if (g->k != 3 || g->stride != 1 || g->dil != 1 || g->pad != 1 || g->in_up || (g->out_w % W3_TW) || (g->out_h % W3_TH) ||
      (g->cout & 63) || (g->cin & 31))
    return 0;
  const W3Plan w = w3_plan(g);
  return ((size_t)w.ksplit * w.kh * g->cout * g->cin * 9 + (size_t)w.ksplit * w.kh * g->cout) * sizeof(float) + 256;
}

// returns 1 when it took the launch (*rc = status)
int wgrad3_try(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy, long long dy_sn, long long dy_sc,
               float* dw, float* db, int accumulate, void* workspace, hipStream_t s, pcuda_reduce_job* defer, int* rc) {
  if (!w3_eligible(g, x, dy, dy_sn, dy_sc)) return 0;
  const bool x3 = prec == PCUDA_PREC_BF16X3;
  const W3Plan w = w3_plan(g);
  W3Params p;
  memset(&p, 0, sizeof(p));
  p.x = *x; p.cin = g->cin; p.H = g->in_h; p.W = g->in_w;
  p.dz = dy; p.dz_sn = dy_sn; p.dz_sc = dy_sc; p.cout = g->cout;
  p.tiles_x = w.tiles_x; p.tiles_y = w.tiles_y; p.n = g->n;
  p.ksplit = w.ksplit; p.n_chunks = w.n_chunks;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("PCUDA_W3DBG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
  }
  const int slices = w.ksplit * w.kh;
  const long long welems = (long long)g->cout * g->cin * 9;
  p.partial = (float*)workspace;
  p.db_partial = db ? (float*)workspace + (size_t)slices * welems : nullptr;
  const size_t lds = (size_t)(x3 ? 2 : 1) * (W3_XPLANE + 32 * w.rb * WG_ZROW);
  p.base = w.n_co_tiles * w.n_chunks;
  {
    static int noxcd = -1;
    if (noxcd < 0) { const char* e = getenv("PCUDA_W3_NOXCD"); noxcd = (e && atoi(e)) ? 1 : 0; }
    p.xcd = (!noxcd && (w.ksplit % 8) == 0) ? 1 : 0;
  }
  const dim3 grid(p.base * w.ksplit);
  {
    char tag[160];
    snprintf(tag, sizeof(tag), "wgrad3 n%d cin%d cout%d %dx%d k3 s1 d1 ksplit%d rb%d lds%zu", g->n, g->cin, g->cout, g->out_h,
             g->out_w, w.ksplit, w.rb, lds);
    ProfScope prof(PCUDA_FAM_CONV_WGRAD, 2.0 * g->n * (double)g->out_h * g->out_w * g->cout * (double)g->cin * 9, s, tag);
#define W3_LAUNCH(X3_, RB_, KH_)                                                                                       \
  do {                                                                                                                  \
    auto kern = wgrad3_kernel<X3_, RB_, KH_>;                                                                          \
    if (lds > 48 * 1024) {   /* (the attribute is per device: set on every launch, the call is cheap) */               \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD); \
      if (e != hipSuccess) { pcuda_set_error("wgrad3: cannot raise dynamic LDS: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; return 1; } \
    }                                                                                                                   \
    hipLaunchKernelGGL(kern, grid, dim3(64 * RB_ * 2 * KH_), lds, s, p);                                                \
  } while (0)
    if (x3) W3_LAUNCH(true, 2, 1);
    else W3_LAUNCH(false, 2, 1);
#undef W3_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pcuda_set_error("wgrad3_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; return 1; }
  }
  int nkg = 1;
  while (nkg < 16 && nkg * 2 <= slices) nkg <<= 1;
  if (defer) {
    defer->partial = (const float*)workspace; defer->numel = welems; defer->ksplit = slices; defer->nkg = nkg;
    defer->dw = dw; defer->accumulate = accumulate; defer->ntaps = 9;
    defer->db_partial = (const float*)p.db_partial; defer->nb = db ? g->cout : 0; defer->db = db;
    *rc = PCUDA_OK;
    return 1;
  }
  *rc = launch_wgrad_reduce_taps((const float*)workspace, welems, slices, dw, accumulate, 9, (const float*)p.db_partial,
                                 db ? g->cout : 0, db, s);
  return 1;
}
