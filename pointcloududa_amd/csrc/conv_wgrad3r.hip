// Weight gradient of 3x3 / stride-1 / pad-1 layers with few channels on large maps (the segmenter's 32- / 64-channel levels
// at 256x256 ... 112x112: unet.py:23,27,116,122 going back) WITHOUT LDS staging: an experiment in the structure of
// conv_wgrad1.hip.  dW[co][ci][tap] = sum over pixels of dZ[co][p] X[ci][p + tap]: both operands are natural rows with the
// reduction index (pixels of an image row) contiguous, so a lane's MFMA fragment -- row = lane & 31, 8 consecutive pixels --
// is two float4 loads from global memory.  The nine taps are nine SHIFTED views of the same three input rows:
//   * a wave owns one 32 x 32 block of dW for all nine taps (144 accumulator registers) and walks DOWN a 16-pixel-wide
//     column strip, one image row per step: the three input rows of a step are the previous step's last two plus one new
//     row, held in registers as bf16 hi / lo pairs -- every input element is loaded, affine-transformed and split ONCE;
//   * the +-1-pixel shifts are built in the packed-bf16 domain: v_alignbit over neighbouring register pairs, the pixel
//     that crosses the half-wave boundary comes from the partner lane (one cross-lane exchange per plane and row), the
//     pixel outside the strip from one extra dword per lane and row;
//   * no LDS tiles, no barriers inside the loop, no transposing reads; 27 MFMAs (9 taps x 3 bf16x3 products) per step and
//     wave against ~100 vector instructions; loads run six rows ahead in a register ring;
//   * one workgroup = four waves (one per SIMD: 512 registers each) on the two 16-pixel halves x the two row halves of the
//     same 32-pixel strips, their accumulators added in a fixed order through LDS at the end; slabs go through the
//     common fixed-order reduce.
// Eligible: k = 3, stride 1, pad 1, no dilation, rows of 32 k pixels, 16-byte aligned planes; two sources and the lazy-BatchNorm
// affine are per-row constants (zero padding is applied AFTER the affine).  The up-convolutions (unet.py:85 nn.Upsample(x2) in
// front of the 3x3 convolution; UP = true): the input is read at its STORED resolution -- a lane's eight pixels are four stored
// ones (one float4 instead of two, row y of the convolution's input is stored row y >> 1), split once and doubled in the
// packed-bf16 domain (v_perm_b32): X traffic is a quarter of the upsampled tensor's.
#include <type_traits>

#include "conv_device.h"
#include "conv_host.h"

namespace {

struct W3RParams {
  pcuda_src x;
  int cin, cout, n, H, W;
  const float* dz; long long dz_sn, dz_sc;
  float* partial;         // [slices][9][cout][cin]
  float* db_partial;      // [slices][cout] or NULL
  int n_ci_tiles, tiles;  // 32 x 32 blocks of dW
  int strips;             // W / 32
  int rsplit;             // row blocks per strip (small batches: more workgroups)
  int npairs;             // n * strips * rsplit: (image, 32-pixel strip, row block) items
  int pairs_per_slice;
  int xcd;
  int dbg;                // PCUDA_W3RDBG (timing experiments): 1 loads only once per pair, 2 no MFMAs, 4 no conversion
};

struct RowF {             // one input row of the strip as MFMA B fragments: [shift dx + 1][plane hi / lo]
  uint4 f[3][2];
};
struct RawX {
  f32x4 v0, v1;           // (UP: v1 is never loaded)
  float halo;
};
struct RawZ {
  f32x4 v0, v1;
};

__device__ __forceinline__ uint4 shift_plus(const uint4 a, unsigned edge) {      // pixel p -> p + 1 (edge: low half = pixel 8)
  uint4 o;
  o.x = __builtin_amdgcn_alignbit(a.y, a.x, 16);
  o.y = __builtin_amdgcn_alignbit(a.z, a.y, 16);
  o.z = __builtin_amdgcn_alignbit(a.w, a.z, 16);
  o.w = __builtin_amdgcn_alignbit(edge, a.w, 16);
  return o;
}
__device__ __forceinline__ uint4 shift_minus(const uint4 a, unsigned edge) {     // pixel p -> p - 1 (edge: high half = pixel -1)
  uint4 o;
  o.x = __builtin_amdgcn_alignbit(a.x, edge, 16);
  o.y = __builtin_amdgcn_alignbit(a.y, a.x, 16);
  o.z = __builtin_amdgcn_alignbit(a.z, a.y, 16);
  o.w = __builtin_amdgcn_alignbit(a.w, a.z, 16);
  return o;
}

template <bool X3, bool UP = false>
__global__ __launch_bounds__(256, 1) void wgrad3r_kernel(const W3RParams p) {
  __shared__ float red[9 * 16 * 64];
  __shared__ float dbs[4][32];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

  int item = blockIdx.x;
  if (p.xcd) item = (blockIdx.x & 7) * ((int)gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = item % p.tiles, slice = item / p.tiles;
  const int cit = tile % p.n_ci_tiles, cot = tile / p.n_ci_tiles;

  // ---- per-lane row constants
  const int co = min(cot * 32 + r, p.cout - 1);
  const int ci = min(cit * 32 + r, p.cin - 1);
  const bool first = ci < p.x.c1;
  const float* xbase = first ? p.x.p1 : p.x.p2;
  const long long x_sn = first ? p.x.sn1 : p.x.sn2, x_sc = first ? p.x.sc1 : p.x.sc2;
  const int cl = first ? ci : ci - p.x.c1;
  const float* scp = first ? p.x.scale1 : p.x.scale2;
  const float* shp = first ? p.x.shift1 : p.x.shift2;
  const float sc = scp ? scp[cl] : 1.f, sh = scp ? shp[cl] : 0.f;
  const bool do_db = p.db_partial != nullptr && cit == 0;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float dbacc = 0.f;

  // a workgroup's four waves: the two 16-pixel halves of a 32-pixel strip (one 128-byte line of every row: both halves are
  // requested at about the same time, the line is fetched once) x the upper / lower half of the rows
  const int wx = w & 1, wy = w >> 1;

  const int pair0 = slice * p.pairs_per_slice, pair1 = min(p.npairs, pair0 + p.pairs_per_slice);
  for (int pair = pair0; pair < pair1; ++pair) {
    const int per = p.strips * p.rsplit;
    const int img = pair / per, prem = pair - img * per, strip = prem / p.rsplit, rb = prem - strip * p.rsplit;
    const int x0 = strip * 32 + 16 * wx;
    // this wave's rows: the upper / lower half of the item's row block
    const int y_lo = (int)((long long)p.H * rb / p.rsplit), y_hi = (int)((long long)p.H * (rb + 1) / p.rsplit);
    const int ya = y_lo + (y_hi - y_lo) * wy / 2, yb = y_lo + (y_hi - y_lo) * (wy + 1) / 2;
    const int nrows = yb - ya;
    const float* zrow = p.dz + (long long)img * p.dz_sn + (long long)co * p.dz_sc + x0 + 8 * h;
    const float* xrow = xbase + (long long)img * x_sn + (long long)cl * x_sc + ((x0 + 8 * h) >> (UP ? 1 : 0));
    const int xpitch = p.W >> (UP ? 1 : 0);                             // stored row pitch
    // the pixel beside the strip: left of it for the lower half-wave, right of it for the upper one
    const int hx = h ? x0 + 16 : x0 - 1;
    const bool hvalid = (unsigned)hx < (unsigned)p.W;
    const float* hrow = xbase + (long long)img * x_sn + (long long)cl * x_sc + (min(max(hx, 0), p.W - 1) >> (UP ? 1 : 0));

    // lanes whose halo pixel lies outside the image read zero: folded into that load's affine
    const float sch = hvalid ? sc : 0.f, shh = hvalid ? sh : 0.f;

    RowF rows[4];
    RawX rx[4];
    RawZ rz[4];
    bf16x8 ah[2], al[2];

    // X row slot j is image row ya - 1 + j.  Loads are unconditional (clamped rows: a branch around loads would turn the
    // counted waits into full ones); rows outside the image become zero through their affine (x * 0 + 0).  Every lambda
    // below is branch-free, so that a step is ONE scheduling region: its 27 MFMAs with the conversion of the NEXT step's
    // operands placed in their shadows (sched_group_barrier at the end of ``step``).
    auto issue_x = [&](int j, auto SLOT) {
      constexpr int S = decltype(SLOT)::value;
#ifdef PCUDA_W3R_DEBUG
      if ((p.dbg & 1) && j > 3) return;
#endif
      const int yy = min(max(ya - 1 + j, 0), p.H - 1) >> (UP ? 1 : 0);
      const float* q = xrow + (long long)yy * xpitch;
      rx[S].v0 = *(const f32x4*)q;
      if (!UP) rx[S].v1 = *(const f32x4*)(q + 4);
      rx[S].halo = hrow[(long long)yy * xpitch];
    };
    auto issue_z = [&](int i, auto SLOT) {
      constexpr int S = decltype(SLOT)::value;
#ifdef PCUDA_W3R_DEBUG
      if ((p.dbg & 1) && i > 3) return;
#endif
      const int yy = min(ya + i, p.H - 1);
      const float* q = zrow + (long long)yy * p.W;
      rz[S].v0 = *(const f32x4*)q;
      rz[S].v1 = *(const f32x4*)(q + 4);
    };
    auto convert_x = [&](int j, auto SLOT, auto ROW) {
      constexpr int S = decltype(SLOT)::value, R = decltype(ROW)::value;
      const bool rvalid = (unsigned)(ya - 1 + j) < (unsigned)p.H;      // (uniform)
      const float s1 = rvalid ? sc : 0.f, s0 = rvalid ? sh : 0.f, h1 = rvalid ? sch : 0.f, h0 = rvalid ? shh : 0.f;
      f32x4 a = rx[S].v0, b = UP ? rx[S].v0 : rx[S].v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = fmaf(a[e], s1, s0); if (!UP) b[e] = fmaf(b[e], s1, s0); }
      const float hv = fmaf(rx[S].halo, h1, h0);
      uint4 H, L = make_uint4(0, 0, 0, 0);
      unsigned hh, hl = 0;
      if (UP) {            // four stored pixels: split as two pairs, every half-word doubled
        unsigned h01, l01 = 0, h23, l23 = 0;
        if (X3) { split2(a[0], a[1], h01, l01); split2(a[2], a[3], h23, l23); split2(hv, 0.f, hh, hl); }
        else { h01 = pack_bf16x2(a[0], a[1]); h23 = pack_bf16x2(a[2], a[3]); hh = pack_bf16x2(hv, 0.f); }
        H.x = __builtin_amdgcn_perm(h01, h01, 0x01000100u); H.y = __builtin_amdgcn_perm(h01, h01, 0x03020302u);
        H.z = __builtin_amdgcn_perm(h23, h23, 0x01000100u); H.w = __builtin_amdgcn_perm(h23, h23, 0x03020302u);
        if (X3) {
          L.x = __builtin_amdgcn_perm(l01, l01, 0x01000100u); L.y = __builtin_amdgcn_perm(l01, l01, 0x03020302u);
          L.z = __builtin_amdgcn_perm(l23, l23, 0x01000100u); L.w = __builtin_amdgcn_perm(l23, l23, 0x03020302u);
        }
      } else if (X3) {
        split2(a[0], a[1], H.x, L.x); split2(a[2], a[3], H.y, L.y);
        split2(b[0], b[1], H.z, L.z); split2(b[2], b[3], H.w, L.w);
        split2(hv, 0.f, hh, hl);
      } else {
        H.x = pack_bf16x2(a[0], a[1]); H.y = pack_bf16x2(a[2], a[3]);
        H.z = pack_bf16x2(b[0], b[1]); H.w = pack_bf16x2(b[2], b[3]);
        hh = pack_bf16x2(hv, 0.f);
      }
      // the partner lane's neighbouring pixel pair: the lower half-wave needs the upper one's first pair (pixel 8), the
      // upper half-wave the lower one's last pair (pixel 7)
      const unsigned gotH = (unsigned)__shfl_xor((int)(h ? H.x : H.w), 32, 64);
      const unsigned eplusH = h ? hh : gotH;                  // low half = the pixel right of this lane's eight
      const unsigned eminusH = h ? gotH : (hh << 16);         // high half = the pixel left of them
      rows[R].f[1][0] = H;
      rows[R].f[2][0] = shift_plus(H, eplusH);
      rows[R].f[0][0] = shift_minus(H, eminusH);
      if (X3) {
        const unsigned gotL = (unsigned)__shfl_xor((int)(h ? L.x : L.w), 32, 64);
        const unsigned eplusL = h ? hl : gotL;
        const unsigned eminusL = h ? gotL : (hl << 16);
        rows[R].f[1][1] = L;
        rows[R].f[2][1] = shift_plus(L, eplusL);
        rows[R].f[0][1] = shift_minus(L, eminusL);
      }
    };
    auto convert_z = [&](auto SLOT, auto AB, float counts) {
      constexpr int S = decltype(SLOT)::value, Q = decltype(AB)::value;
      const f32x4 a = rz[S].v0, b = rz[S].v1;
      // bias gradient (stored by the ci-block-0 waves only); counts = 0 for the row converted BEHIND the wave's range
      dbacc = fmaf(counts, ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3])), dbacc);
      uint4 hi, lo = make_uint4(0, 0, 0, 0);
      if (X3) {
        split2(a[0], a[1], hi.x, lo.x); split2(a[2], a[3], hi.y, lo.y);
        split2(b[0], b[1], hi.z, lo.z); split2(b[2], b[3], hi.w, lo.w);
      } else {
        hi.x = pack_bf16x2(a[0], a[1]); hi.y = pack_bf16x2(a[2], a[3]);
        hi.z = pack_bf16x2(b[0], b[1]); hi.w = pack_bf16x2(b[2], b[3]);
      }
      ah[Q] = __builtin_bit_cast(bf16x8, hi);
      al[Q] = __builtin_bit_cast(bf16x8, lo);
    };
    // step i (phase PH = i % 4): the MFMAs of dZ row ya + i against X row slots i, i + 1, i + 2 -- all converted by earlier
    // steps -- while row slot i + 3 and dZ row i + 1 are converted for the next step and the loads four steps ahead go out
    auto step = [&](int i, auto PHASE) {
      constexpr int PH = decltype(PHASE)::value;
#ifdef PCUDA_W3R_DEBUG
      if (!(p.dbg & 4)) {
#endif
      convert_x(i + 3, std::integral_constant<int, (PH + 3) % 4>{}, std::integral_constant<int, (PH + 3) % 4>{});
      convert_z(std::integral_constant<int, (PH + 1) % 4>{}, std::integral_constant<int, (PH + 1) % 2>{}, i + 1 < nrows ? 1.f : 0.f);
#ifdef PCUDA_W3R_DEBUG
      }
#endif
      issue_x(i + 7, std::integral_constant<int, (PH + 3) % 4>{});
      issue_z(i + 5, std::integral_constant<int, (PH + 1) % 4>{});
      const bf16x8 a_h = ah[PH % 2], a_l = al[PH % 2];
#ifdef PCUDA_W3R_DEBUG
      if (!(p.dbg & 2))
#endif
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const RowF& rw = rows[(PH + dy) % 4];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, rw.f[dx][0]);
          if (X3) {
            const bf16x8 bl = __builtin_bit_cast(bf16x8, rw.f[dx][1]);
            acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, bh, acc[dy * 3 + dx], 0, 0, 0);
            acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bl, acc[dy * 3 + dx], 0, 0, 0);
          }
          acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bh, acc[dy * 3 + dx], 0, 0, 0);
        }
      }
      // one wave per SIMD: nothing else fills the matrix pipe's shadow -- ~5 vector instructions behind every MFMA
#pragma unroll
      for (int k = 0; k < (X3 ? 27 : 9); ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, X3 ? 5 : 9, 0);
        if ((k % 5) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    };

    // ---- prologue: X row slots 0 .. 3 and dZ rows 0 .. 3 requested, slots 0 .. 2 and dZ row 0 converted, their ring
    // entries requested again (slots 4 .. 6, dZ row 4)
    issue_x(0, std::integral_constant<int, 0>{}); issue_x(1, std::integral_constant<int, 1>{});
    issue_x(2, std::integral_constant<int, 2>{}); issue_x(3, std::integral_constant<int, 3>{});
    issue_z(0, std::integral_constant<int, 0>{}); issue_z(1, std::integral_constant<int, 1>{});
    issue_z(2, std::integral_constant<int, 2>{}); issue_z(3, std::integral_constant<int, 3>{});
    convert_x(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    convert_x(1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    convert_x(2, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
    convert_z(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 1.f);
    issue_x(4, std::integral_constant<int, 0>{}); issue_x(5, std::integral_constant<int, 1>{});
    issue_x(6, std::integral_constant<int, 2>{});
    issue_z(4, std::integral_constant<int, 0>{});

    int i = 0;
    for (; i + 4 <= nrows; i += 4) {
      step(i, std::integral_constant<int, 0>{}); step(i + 1, std::integral_constant<int, 1>{});
      step(i + 2, std::integral_constant<int, 2>{}); step(i + 3, std::integral_constant<int, 3>{});
    }
    const int rem = nrows - i;      // (uniform) i is a multiple of 4 here: the phases continue from 0
    if (rem > 0) step(i, std::integral_constant<int, 0>{});
    if (rem > 1) step(i + 1, std::integral_constant<int, 1>{});
    if (rem > 2) step(i + 2, std::integral_constant<int, 2>{});
  }

  // ---- the four waves' sums in a fixed order: ((w0 + w1) + w2) + w3
  for (int ww = 1; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(t * 16 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] += red[(t * 16 + i) * 64 + lane];
    }
    __syncthreads();
  }
  if (do_db) {
    const float s = dbacc + __shfl_xor(dbacc, 32, 64);
    if (h == 0) dbs[w][r] = s;
  }
  __syncthreads();
  if (w != 0) return;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int cio = cit * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int coo = cot * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (coo < p.cout && cio < p.cin)
        p.partial[(((long long)slice * 9 + t) * p.cout + coo) * p.cin + cio] = acc[t][i];
    }
  }
  if (do_db && h == 0) {
    const int coo = cot * 32 + r;
    if (coo < p.cout) p.db_partial[(long long)slice * p.cout + coo] = ((dbs[0][r] + dbs[1][r]) + dbs[2][r]) + dbs[3][r];
  }
}

struct W3RPlan {
  int n_co_tiles, n_ci_tiles, strips, rsplit, npairs, pairs_per_slice, slices;
};

// PCUDA_WGRAD3R: 0 off, 1 (default) the layers it was measured faster on, 2 every eligible layer (tests, micro-benchmarks).
// Measured (B = 32, same box, against conv_wgrad_impl.h / conv_wgrad3.hip): 64->32 at 256x256 0.335 against 0.394 ms, 32->64 at
// 128x128 0.075 against 0.089; at parity where cin = cout (32->32 at 256x256 0.215 / 0.211, 64->64 at 128x128 0.158 / 0.163,
// 128->128 at 64x64 0.145 / 0.141) and for 128->64 at 128x128 (0.289 / 0.296).  Its time is its loads: with the MFMAs or the
// conversion switched off (PCUDA_W3R_DEBUG builds, PCUDA_W3RDBG) 32->32 at 256x256 stays at 0.19-0.22 ms, without the loads it
// runs in 0.13 ms (MFMA-bound: 1060 cycles per 27-MFMA step) -- a lane per row, 32 bytes per lane and step, streams 3.8 TB/s.
bool w3r_geom(const pcuda_conv_geom* g) {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("PCUDA_WGRAD3R"); mode = e ? atoi(e) : 1; }
  if (!mode) return false;
  static int maxc = -1;
  if (maxc < 0) { const char* e = getenv("PCUDA_W3R_MAXC"); maxc = e ? atoi(e) : 128; }
  static int upmaxc = -1;     // PCUDA_W3R_UPMAXC: the up-convolutions (input read at its stored resolution) it takes by default
  if (upmaxc < 0) { const char* e = getenv("PCUDA_W3R_UPMAXC"); upmaxc = e ? atoi(e) : 512; }
  const int mc = (g->in_up && upmaxc > maxc) ? upmaxc : maxc;
  const bool shape = g->k == 3 && g->stride == 1 && g->pad == 1 && g->dil == 1 && (g->in_w & 31) == 0 && g->in_h >= 4 &&
                     (!g->in_up || (g->in_h & 1) == 0) &&
                     g->cin >= 16 && g->cin <= mc && g->cout <= mc && (long long)g->in_h * g->in_w < (1ll << 28);
  if (!shape || mode >= 2) return shape;
  if (g->in_up) return g->cin <= upmaxc && g->cout <= upmaxc;
  // default: unequal channel counts up to 64 (one operand's rows are then re-read by fewer blocks than the pixel-record
  // kernel stages them for) on maps of at least 64 rows
  // (round 6, re-measured on the library without SLP vectorisation: every layer on maps of at least 128 x 128 -- 32->32 at 256x256
  //  0.214 -> 0.197 ms, 64->64 at 128x128 0.165 -> 0.152, 128->64 at 128x128 0.290 -> 0.265; still at parity or behind on 64x64 and
  //  32x32 maps: 128->128 0.137 / 0.140, 256->256 0.144 / 0.146)
  if (g->in_h >= 128 && g->in_w >= 128) return true;
  return g->cin != g->cout && g->cin <= 64 && g->cout <= 64 && g->in_h >= 64;
}

W3RPlan w3r_plan(const pcuda_conv_geom* g) {
  W3RPlan w;
  w.n_co_tiles = cdiv(g->cout, 32);
  w.n_ci_tiles = cdiv(g->cin, 32);
  const int tiles = w.n_co_tiles * w.n_ci_tiles;
  w.strips = g->in_w / 32;
  // (small batches: row blocks until there is a workgroup per compute unit, at least 16 rows per wave)
  w.rsplit = 1;
  while ((long long)g->n * w.strips * w.rsplit * tiles < 256 && g->in_h / (w.rsplit * 2) >= 32) w.rsplit *= 2;
  w.npairs = g->n * w.strips * w.rsplit;
  // one workgroup per CU (four waves with 512 registers each): ~512 workgroups = two rounds; slabs below 32 MB
  static int tgt = -1;
  if (tgt < 0) { const char* e = getenv("PCUDA_W3R_BLOCKS"); tgt = e ? atoi(e) : 512; }
  long long want = tgt / tiles;
  const long long welems = (long long)g->cout * g->cin * 9;
  if (want * welems * 4 > (32ll << 20)) want = (32ll << 20) / (welems * 4);
  if (want < 1) want = 1;
  if (want > w.npairs) want = w.npairs;
  w.pairs_per_slice = cdiv(w.npairs, (int)want);
  w.slices = cdiv(w.npairs, w.pairs_per_slice);
  return w;
}

}  // namespace

size_t wgrad3r_workspace(const pcuda_conv_geom* g) {
  if (!w3r_geom(g)) return 0;
  const W3RPlan w = w3r_plan(g);
  return ((size_t)w.slices * g->cout * g->cin * 9 + (size_t)w.slices * g->cout) * sizeof(float) + 256;
}

// returns 1 when it took the launch (*rc = status)
int wgrad3r_try(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy, long long dy_sn, long long dy_sc,
                float* dw, float* db, int accumulate, void* workspace, hipStream_t s, pcuda_reduce_job* defer, int* rc) {
  if (!w3r_geom(g)) return 0;
  auto al = [](const void* q, long long sn, long long sc) { return q == nullptr || ((((uintptr_t)q) & 15) == 0 && (sn & 3) == 0 && (sc & 3) == 0); };
  const int c1 = x->c1 < g->cin ? x->c1 : g->cin;
  if (!al(x->p1, x->sn1, x->sc1) || (c1 < g->cin && !al(x->p2, x->sn2, x->sc2)) || !al(dy, dy_sn, dy_sc)) return 0;
  const bool x3 = prec == PCUDA_PREC_BF16X3;
  const W3RPlan w = w3r_plan(g);
  W3RParams p;
  memset(&p, 0, sizeof(p));
  p.x = *x; p.x.c1 = c1;
  p.cin = g->cin; p.cout = g->cout; p.n = g->n; p.H = g->in_h; p.W = g->in_w;
  p.dz = dy; p.dz_sn = dy_sn; p.dz_sc = dy_sc;
  const long long welems = (long long)g->cout * g->cin * 9;
  p.partial = (float*)workspace;
  p.db_partial = db ? (float*)workspace + (size_t)w.slices * welems : nullptr;
  p.n_ci_tiles = w.n_ci_tiles; p.tiles = w.n_co_tiles * w.n_ci_tiles;
  p.strips = w.strips; p.rsplit = w.rsplit; p.npairs = w.npairs; p.pairs_per_slice = w.pairs_per_slice;
  const long long nwg = (long long)p.tiles * w.slices;
  p.xcd = (nwg >= 16 && (nwg & 7) == 0) ? 1 : 0;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("PCUDA_W3RDBG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
  }
  {
    char tag[160];
    snprintf(tag, sizeof(tag), "wgrad3r n%d cin%d cout%d %dx%d k3 s1 d1 up%d slices%d pairs%d", g->n, g->cin, g->cout, g->in_h, g->in_w,
             g->in_up ? 1 : 0, w.slices, w.pairs_per_slice);
    ProfScope prof(PCUDA_FAM_CONV_WGRAD, 2.0 * g->n * (double)g->in_h * g->in_w * g->cout * (double)g->cin * 9, s, tag);
    const dim3 grid((unsigned)nwg);
    if (g->in_up) {
      if (x3) hipLaunchKernelGGL((wgrad3r_kernel<true, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((wgrad3r_kernel<false, true>), grid, dim3(256), 0, s, p);
    } else {
      if (x3) hipLaunchKernelGGL(wgrad3r_kernel<true>, grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL(wgrad3r_kernel<false>, grid, dim3(256), 0, s, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pcuda_set_error("wgrad3r_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; return 1; }
  }
  int nkg = 1;
  while (nkg < 16 && nkg * 2 <= w.slices) nkg <<= 1;
  if (defer) {
    defer->partial = (const float*)workspace; defer->numel = welems; defer->ksplit = w.slices; defer->nkg = nkg;
    defer->dw = dw; defer->accumulate = accumulate; defer->ntaps = 9;
    defer->db_partial = (const float*)p.db_partial; defer->nb = db ? g->cout : 0; defer->db = db;
    *rc = PCUDA_OK;
    return 1;
  }
  *rc = launch_wgrad_reduce_taps((const float*)workspace, welems, w.slices, dw, accumulate, 9, (const float*)p.db_partial,
                                 db ? g->cout : 0, db, s);
  return 1;
}
