// Row-streaming 3x3 convolution for the segmenter's 32 -> 32-channel layers on the full-resolution maps (unet.py:23,27 encoder
// block 1, :116,122 decoder block 1 at 256x256 / 224x224: forward and data gradient, eight launches per training step).
//
// Those launches are the one 3x3 class that is bound by neither the matrix pipe nor the clock (DESIGN section 4 "Round 6"): 32
// output rows give a staged input tile only 9 x 32 x 32 MACs per pixel, and igemm_pipe_kernel spends its time staging 32 x 8-pixel
// tiles with their halo rows (10 rows staged per 8 computed), behind workgroup barriers: 243 us per launch against an HBM floor of
// ~105 us.  Here every WAVE is its own pipeline (the structure of conv_wgrad3r.hip turned to the forward problem):
//   * a wave owns a 32-pixel-wide column strip of one image (and a segment of its rows) and walks DOWN it one output row per step;
//     the three input rows of a step are the previous step's last two plus ONE new row: every input element is loaded,
//     affine-transformed and split into bf16 hi / lo once (plus two halo pixels per row and channel, one dword per lane);
//   * the rows live in a wave-private LDS ring of four row slots ([34 pixel records][32 hi | 32 lo | pad], the records of the
//     ordinary kernels): the nine taps are nine (slot, pixel offset) immediates on ds_read_b128 -- no barrier anywhere in the
//     loop, LDS ordering of a single wave is program order;
//   * the layer's whole weight tensor (9 taps x 32 rows x 144 B = 41 KB: the ORDINARY packed layout, no image of its own) is
//     copied to LDS once per workgroup; weight fragments are immediates as well;
//   * D[row = pixel][col = output channel] (A = the pixel records, B = the weights): an accumulator register is four consecutive
//     pixels of ONE channel per lane, so the stores, the accumulate loads and the BatchNorm-backward reduce's loads of the saved
//     activation are float4 (as are the input loads: a lane loads four channels x four pixels of a row); bias, mean, 1 / std are
//     per-lane scalars and the BatchNorm partial sums two registers per lane for the whole strip (one cross-lane add per item).
//     Bias, LeakyReLU, accumulate, partial sums (forward) and the BatchNorm-backward reduce (data gradient:
//     pcuda_conv2d_dgrad_bnred) run on the previous row's accumulators while the matrix pipe works on the current one;
//   * one wave per SIMD: the source order IS the schedule -- behind each of a step's 54 MFMAs one piece of that work, pinned by a
//     scheduling fence (scripts/rs_isa_check.py checks the assembly); loads run four rows ahead in a register ring.
// Work item = (image, strip, row segment); items = a multiple of the chip's wave slots where the map allows.  One workgroup =
// four waves on four neighbouring strips (their halo pixels are each other's lines), one workgroup per CU (120 KB of LDS).
// Bound (profiles/r06_experiment_row_streaming.txt): the mixed read / write stream at ~3.3 TB/s (loads alone 4.3, stores alone
// 4.5 TB/s; the instruction stream alone 88 us of 173-178).
#include <stdlib.h>
#include <type_traits>

#include "conv_igemm.h"
#include "conv_device.h"
#include "conv_host.h"
#include "variants.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// cache policy of the streaming loads / stores (timing experiments: -DRS_LD_AUX=2 / -DRS_ST_AUX=2 = non-temporal)
#ifndef RS_LD_AUX
#define RS_LD_AUX 0
#endif
#ifndef RS_ST_AUX
#define RS_ST_AUX 0
#endif

constexpr int RS_REC = 144;                        // record bytes: 32 bf16 hi | 32 bf16 lo | 16 pad (ig_rec_bytes(true))
constexpr int RS_WTAP = 32 * RS_REC;               // one tap of weights
constexpr int RS_WBYTES = 9 * RS_WTAP;             // 41472
constexpr int RS_ROW = 34 * RS_REC;                // one input row of a strip: pixels x0 - 1 .. x0 + 32
constexpr int RS_RING = 4 * RS_ROW;                // 19584
constexpr int RS_LDS = RS_WBYTES + 4 * RS_RING;    // 119808

struct RsParams {
  pcuda_src x;               // 32 channels, all in source 1
  pcuda_dst y;               // 32 rows, all in destination 1
  int H, W, n;
  const unsigned char* wimg; // ordinary packed layout of the launch: [tap][32 rows][144 B]
  int tap_pos[9];            // LDS position (dy + 1) * 3 + dx + 1 of the image's tap i
  const float* bias; float slope;
  float* stats;              // [item][32][2]
  const float* red_a; long long red_sn, red_sc; const float* red_mean; const float* red_invstd;
  int strips, rsplit, items, xcd;
};

// STATS: 0 none, 1 (sum, sum of squares) of the stored values, 2 the BatchNorm-backward reduce (sum g, sum g * xhat(a)).
// ACC: y += result.  AFF: the input has a lazy-BatchNorm affine (scale / shift per channel; padding stays zero).
// DBG (timing experiments, -DPCUDA_RS_DEBUG builds + PCUDA_RSDBG): 1 no MFMAs, 2 no stores, 4 no input loads, 8 no conversion
template <int STATS, bool ACC, bool AFF, int DBG = 0>
__global__ __launch_bounds__(256, 1) void conv3rs_kernel(const RsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nl = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- the weights: tap i of the packed layout to its (dy, dx) position
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    const unsigned char* src = p.wimg + (size_t)t * RS_WTAP;
    unsigned char* dst = smem + p.tap_pos[t] * RS_WTAP;
    for (int i = tid; i < RS_WTAP / 16; i += 256) *(uint4*)(dst + i * 16) = *(const uint4*)(src + i * 16);
  }
  __syncthreads();

  int wg = blockIdx.x;
  if (p.xcd) wg = (blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  const int item = wg * 4 + w;
  if (item >= p.items) return;                    // (no barrier below)
  const int strip = item % p.strips, t2 = item / p.strips, rs = t2 % p.rsplit, img = t2 / p.rsplit;
  const int ya = (int)((long long)p.H * rs / p.rsplit), yb = (int)((long long)p.H * (rs + 1) / p.rsplit);
  const int nrows = yb - ya;
  const int x0 = strip * 32;
  const int W = p.W, H = p.H;

  // ---- input (one source): lane = (quad = lane & 7, channel group g = (lane >> 3) & 3, half = lane >> 5): four float4 loads per
  // row, channels 8 g + 4 half + e at pixels x0 + 4 quad .. + 3 -- 1 KB per wave instruction (a wave is alone on its SIMD: the bytes
  // it keeps in flight are its instructions in flight x their size).  Buffer loads: voffset per lane, soffset = (channel, row)
  const int lq = lane & 7, lg = (lane >> 3) & 3;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x.p1 + (long long)img * p.x.sn1), 0, (int)(32 * p.x.sc1 * 4), 0x00020000);
  const unsigned xpl = (unsigned)(p.x.sc1 * 4);
  const unsigned xo = (unsigned)(((8 * lg + 4 * h) * p.x.sc1 + x0 + 4 * lq) * 4);
  float sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc[e] = (AFF && p.x.scale1) ? p.x.scale1[8 * lg + 4 * h + e] : 1.f;
    sh[e] = (AFF && p.x.scale1) ? p.x.shift1[8 * lg + 4 * h + e] : 0.f;
  }
  // halo: lane = (channel nl, side h): the pixel left of the strip (h = 0) or right of it (h = 1); outside the image: zero
  const int hx = h ? x0 + 32 : x0 - 1;
  const unsigned hmask = (unsigned)hx < (unsigned)W ? 0xffffffffu : 0u;
  const float* const hp = p.x.p1 + (long long)img * p.x.sn1 + (long long)nl * p.x.sc1 + min(max(hx, 0), W - 1);
  const float hsc = (AFF && p.x.scale1) ? p.x.scale1[nl] : 1.f, hsh = (AFF && p.x.scale1) ? p.x.shift1[nl] : 0.f;

  // ---- LDS addresses.  MFMA: D[pixel][output channel] = X[pixel][k] W[k][channel]: A fragments from the ring, B from the weights
  const unsigned char* const wl = smem + nl * RS_REC + h * 16;                            // B fragments: + tap, plane, k-step
  unsigned char* const ring = smem + RS_WBYTES + w * RS_RING;
  const unsigned char* const xr = ring + nl * RS_REC + h * 16;                            // A fragments: + slot, dx record, plane, k-step
  // this lane's converted channels of pixel 4 quad + e: record 4 quad + e + 1, eight bytes of the hi plane (lo at + 64)
  unsigned char* const xw = ring + (4 * lq + 1) * RS_REC + (lg >> 1) * 32 + (lg & 1) * 16 + h * 8;
  unsigned char* const hw = ring + (h ? 33 : 0) * RS_REC + nl * 2;                        // this lane's halo value (hi plane; lo at + 64)

  // ---- output (one destination): accumulator register k = pixel (k & 3) + 8 (k >> 2) + 4 h of output channel nl: four float4
  // stores per row and lane
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y.p1 + (long long)img * p.y.sn1), 0, (int)(32 * p.y.sc1 * 4), 0x00020000);
  const unsigned yo = (unsigned)((nl * p.y.sc1 + x0 + 4 * h) * 4);
  const float bias_l = p.bias ? p.bias[nl] : 0.f;
  const float slope = p.slope;
  __amdgpu_buffer_rsrc_t ars = yrs;
  unsigned ao = 0;
  float mean_l = 0.f, is_l = 0.f;
  if (STATS == 2) {
    ars = __builtin_amdgcn_make_buffer_rsrc((void*)(p.red_a + (long long)img * p.red_sn), 0, (int)(32 * p.red_sc * 4), 0x00020000);
    ao = (unsigned)((nl * p.red_sc + x0 + 4 * h) * 4);
    mean_l = p.red_mean[nl]; is_l = p.red_invstd[nl];
  }
  float s1 = 0.f, s2 = 0.f;

  // ---- register ring of the loads: raw row slot S holds input row ya - 1 + j, j = S (mod 4)
  f32x4 raw[4][4];
  float rawh[4];
  f32x4 aux[4], old[4];        // the epilogue's own loads (saved activation / previous y) of the row computed in this step
  f32x16 acc[2][2];            // [row parity][k-step chain]: a row's result is the sum of its two chains
  bf16x8 fr[2][8];             // fragments of one tap, double-buffered: [ks][w hi, w lo, x hi, x lo]

  auto x_row_off = [&](int j) -> int { return min(max(ya - 1 + j, 0), H - 1) * W; };
  auto load_x = [&](int e, int rw, auto SLOT) {   // load e of 5 of an input row: e < 4 channel e of this lane's four, 4 the halo
    constexpr int S = decltype(SLOT)::value;
    if (DBG & 4) { if (e < 4) raw[S][e] = f32x4{0.f, 0.f, 0.f, 0.f}; else rawh[S] = 0.f; return; }
    if (e < 4) raw[S][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, e * xpl + (unsigned)rw * 4u, RS_LD_AUX));
    else rawh[S] = hp[rw];
  };
  auto issue_x = [&](int j, auto SLOT) {           // unconditional, clamped: rows outside the image are zeroed at conversion
    const int rw = x_row_off(j);
#pragma unroll
    for (int e = 0; e < 5; ++e) load_x(e, rw, SLOT);
  };
  // conversion of raw slot S into LDS row slot S in 10 pieces: pixel e = q >> 1 of the quad (affine + first pair / second pair +
  // the two 8-byte writes), then the halo value and its writes
  uint2 cH, cL;
  auto conv_piece = [&](int q, int j, auto SLOT) {
    constexpr int S = decltype(SLOT)::value;
    const unsigned vm = (unsigned)(ya - 1 + j) < (unsigned)H ? 0xffffffffu : 0u;      // (uniform) rows outside the image: zero
    if (q < 8) {
      const int e = q >> 1, pr = q & 1;
      float a0 = raw[S][2 * pr][e], a1 = raw[S][2 * pr + 1][e];
      if (AFF) { a0 = fmaf(a0, sc[2 * pr], sh[2 * pr]); a1 = fmaf(a1, sc[2 * pr + 1], sh[2 * pr + 1]); }
      unsigned hi, lo;
      split2(a0, a1, hi, lo);
      hi &= vm; lo &= vm;
      if (pr == 0) { cH.x = hi; cL.x = lo; }
      else {
        cH.y = hi; cL.y = lo;
        *(uint2*)(xw + S * RS_ROW + e * RS_REC) = cH;
        *(uint2*)(xw + S * RS_ROW + e * RS_REC + 64) = cL;
      }
    } else if (q == 8) {                           // the halo value
      const float hv = AFF ? fmaf(rawh[S], hsc, hsh) : rawh[S];
      unsigned hh, hl;
      split2(hv, 0.f, hh, hl);
      cH.x = hh & vm & hmask; cL.x = hl & vm & hmask;
    } else {
      *(unsigned short*)(hw + S * RS_ROW) = (unsigned short)cH.x;
      *(unsigned short*)(hw + S * RS_ROW + 64) = (unsigned short)cL.x;
    }
  };
  auto convert_x = [&](int j, auto SLOT) {
#pragma unroll
    for (int q = 0; q < 10; ++q) conv_piece(q, j, SLOT);
  };
  auto load_aux = [&](int g, unsigned ro) {        // epilogue loads of register group g (pixels 8 g + 4 h .. + 3) of row offset ro
    if (STATS == 2) aux[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, ao, 32 * g + ro, 0));
    if (ACC) old[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, yo, 32 * g + ro, 0));
  };
  // fragments of tap t for rows in slots ph .. ph + 2 (piece e of 8)
  auto frag_read = [&](int e, int t, int ph, int buf) {
    const int ks = e >> 2, what = e & 3;
    const int dy = t / 3, dx = t - dy * 3;
    const int so = ((ph + dy) & 3) * RS_ROW + dx * RS_REC + ks * 32;
    if (what == 0) fr[buf][e] = lds_frag(wl + t * RS_WTAP + ks * 32);
    if (what == 1) fr[buf][e] = lds_frag(wl + t * RS_WTAP + 64 + ks * 32);
    if (what == 2) fr[buf][e] = lds_frag(xr + so);
    if (what == 3) fr[buf][e] = lds_frag(xr + so + 64);
  };
  // epilogue of register k of row offset ro from the accumulators of row parity Q (its aux / old loads went out one step earlier)
  f32x4 ev;
  auto epi_elem = [&](int k, unsigned ro, auto QQ, const f32x4 (&av)[4], const f32x4 (&ov)[4]) {
    constexpr int Q = decltype(QQ)::value;
    const int g = k >> 2, e = k & 3;
    const float t = (acc[Q][0][k] + acc[Q][1][k]) + bias_l;
    float a = fmaxf(t, t * slope);                // LeakyReLU for 0 <= slope <= 1 (the launcher checks)
    if (ACC) a += ov[g][e];
    ev[e] = a;
    if (STATS == 1) { s1 += a; s2 = fmaf(a, a, s2); }
    if (STATS == 2) { s1 += a; s2 = fmaf(a, (av[g][e] - mean_l) * is_l, s2); }
    if (e == 3 && !(DBG & 2)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ev), yrs, yo, 32 * g + ro, RS_ST_AUX);
  };
  // step i (phase PH = i % 4): the 54 MFMAs of row i, two chains alternating; behind MFMA q one piece of the other work, pinned
  // there by a scheduling fence (the solver of sched_group_barrier left all of it outside the MFMA sequence):
  //   q 0 .. 15   the previous row's epilogue, one register each (EPI); this row's epilogue loads behind the first four
  //   q 18 .. 27  input row slot j = i + 3 converted for the next step (10 pieces)
  //   q 30 .. 34  input row j = i + 7 requested (4 float4 loads + the halo)
  // and in front of every tap's six MFMAs the next tap's eight fragment reads (the first tap of the NEXT step behind the last).
  auto step = [&](int i, auto PHASE, auto EPI) {
    constexpr int PH = decltype(PHASE)::value;
    constexpr bool E = decltype(EPI)::value;
    constexpr int Q = PH & 1;
    f32x4 av[4], ov[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { av[g] = aux[g]; ov[g] = old[g]; }
    const unsigned ro_prev = (unsigned)((ya + i - 1) * W) * 4u;
    const unsigned ro_aux = (unsigned)(min(ya + i, H - 1) * W) * 4u;
    const int rw_next = x_row_off(i + 7);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int buf = (t + PH) & 1;           // (nine taps: the parity flips from step to step)
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        const int q = t * 6 + m;
        const int ks = m & 1, prod = m >> 1;      // chains alternate; products: lo x hi, hi x lo, hi x hi
        // the next tap's fragments: eight reads over the first four slots of this tap
        if (m < 4) {
          const int tn = t == 8 ? 0 : t + 1, phn = t == 8 ? PH + 1 : PH;
          frag_read(2 * m, tn, phn, buf ^ 1);
          frag_read(2 * m + 1, tn, phn, buf ^ 1);
        }
        const bf16x8 wh = fr[buf][ks * 4 + 0], wlo = fr[buf][ks * 4 + 1], xh = fr[buf][ks * 4 + 2], xl = fr[buf][ks * 4 + 3];
        f32x16 c = acc[Q][ks];
        if (t == 0 && prod == 0) {
#pragma unroll
          for (int k = 0; k < 16; ++k) c[k] = 0.f;
        }
        if (!(DBG & 1)) {
          if (prod == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wlo, c, 0, 0, 0);
          if (prod == 1) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, c, 0, 0, 0);
          if (prod == 2) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, c, 0, 0, 0);
        } else if (prod == 0) {
          c[0] += (float)wlo[0] + (float)xh[0];       // (keeps the fragment reads alive)
        }
        acc[Q][ks] = c;
        if (q < 16) {
          if (E) epi_elem(q, ro_prev, std::integral_constant<int, Q ^ 1>{}, av, ov);
          if (q < 4) load_aux(q, ro_aux);
        } else if (q >= 18 && q < 28) {
          if (!(DBG & 8)) conv_piece(q - 18, i + 3, std::integral_constant<int, (PH + 3) & 3>{});
        } else if (q >= 30 && q < 35) {
          load_x(q - 30, rw_next, std::integral_constant<int, (PH + 3) & 3>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- prologue: rows j = 0 .. 3 requested, 0 .. 2 converted, their ring entries requested again (j = 4 .. 6)
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
#pragma unroll
  for (int g = 0; g < 4; ++g) { aux[g] = f32x4{0.f, 0.f, 0.f, 0.f}; old[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  issue_x(0, I0{}); issue_x(1, I1{}); issue_x(2, I2{}); issue_x(3, I3{});
  convert_x(0, I0{}); convert_x(1, I1{}); convert_x(2, I2{});
  issue_x(4, I0{}); issue_x(5, I1{}); issue_x(6, I2{});
#pragma unroll
  for (int e = 0; e < 8; ++e) frag_read(e, 0, 0, 0);

  step(0, I0{}, std::false_type{});
  int i = 1;
  for (; i + 4 <= nrows; i += 4) {
    step(i, I1{}, std::true_type{}); step(i + 1, I2{}, std::true_type{});
    step(i + 2, I3{}, std::true_type{}); step(i + 3, I0{}, std::true_type{});
  }
  const int rem = nrows - i;      // (uniform) i = 1 (mod 4) here
  if (rem > 0) step(i, I1{}, std::true_type{});
  if (rem > 1) step(i + 1, I2{}, std::true_type{});
  if (rem > 2) step(i + 2, I3{}, std::true_type{});
  {
    f32x4 av[4], ov[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { av[g] = aux[g]; ov[g] = old[g]; }
    const unsigned ro = (unsigned)((ya + nrows - 1) * W) * 4u;
    if ((nrows - 1) & 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) epi_elem(k, ro, I1{}, av, ov);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) epi_elem(k, ro, I0{}, av, ov);
    }
  }

  if (STATS) {      // channel nl: the two half-waves hold the sums over their pixels
    const float a = s1 + __shfl_xor(s1, 32, 64), b = s2 + __shfl_xor(s2, 32, 64);
    if (h == 0) *(float2*)(p.stats + ((long long)item * 32 + nl) * 2) = make_float2(a, b);
  }
}

int rs_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("PCUDA_RS"); on = (e && !atoi(e)) ? 0 : 1; }
  return on;
}

// row segments per strip: items = n x strips x rsplit fill the chip's wave slots (4 per CU) while a segment keeps >= 16 rows
int rs_rsplit(int n, int h, int w) {
  static int slots = 0;
  if (!slots) {
    int d = 0; hipDeviceProp_t prop;
    slots = 4 * ((hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&prop, d) == hipSuccess) ? prop.multiProcessorCount : 256);
  }
  int rsplit = 1;
  while ((long long)n * (w / 32) * rsplit < slots && h / (rsplit * 2) >= 16) rsplit *= 2;
  return rsplit;
}

bool rs_taps(const TapSet& taps, int* pos9) {       // the launch's tap i at offset (dy, dx) -> position (dy + 1) * 3 + dx + 1
  if (taps.n != 9) return false;
  bool seen[9] = {false, false, false, false, false, false, false, false, false};
  for (int i = 0; i < 9; ++i) {
    const int dy = taps.dy[i], dx = taps.dx[i];
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
    const int q = (dy + 1) * 3 + dx + 1;
    if (seen[q]) return false;
    seen[q] = true;
    pos9[i] = q;
  }
  return true;
}

template <int STATS, bool ACC, bool AFF>
int rs_launch_t(const RsParams& rp, int grid, hipStream_t s) {
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    if (hipFuncSetAttribute((const void*)conv3rs_kernel<STATS, ACC, AFF>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS) != hipSuccess)
      PCUDA_FAIL(PCUDA_E_LAUNCH, "conv3rs_kernel: cannot opt in to %d bytes of LDS", RS_LDS);
    once.mark(bit);
  }
  hipLaunchKernelGGL((conv3rs_kernel<STATS, ACC, AFF>), dim3(grid), dim3(256), RS_LDS, s, rp);
  PCUDA_CHECK_LAUNCH("conv3rs_kernel");
  return PCUDA_OK;
}

}  // namespace

// A property of the LAYER and the precision: 3x3 / stride 1 / pad 1, 32 rows over 32 reduction channels, bf16x3
bool rs_layer_ok(const pcuda_conv_geom* g, int rows, int red, int prec) {
  return rs_enabled() && prec == PCUDA_PREC_BF16X3 && g->k == 3 && g->stride == 1 && g->pad == 1 && g->dil == 1 && !g->in_up &&
         rows == 32 && red == 32;
}
// Maps the kernel takes; with rs_layer_ok this decides the TILE COUNT a caller sizes its BatchNorm partial sums by
bool rs_map_ok(int n, int h, int w) {
  const char* e = getenv("PCUDA_RS_MIN_ROWS");       // (read per call: tests flip it inside one process)
  const int min_rows = e ? atoi(e) : 64;
  if (!(rs_enabled() && w >= 32 && (w & 31) == 0 && h >= min_rows && h >= 2 && n >= 1)) return false;
  // enough items -- (image, strip, row segment), one per wave -- for the chip: below three quarters of its wave slots (small
  // batches) the ordinary kernel's 256-pixel tiles fill it better.  PCUDA_RS_MIN_ITEMS overrides the threshold (the kernel-level
  // tests run small maps on the kernel).
  const char* m = getenv("PCUDA_RS_MIN_ITEMS");
  static int cus = 0;
  if (!cus) {
    int d = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&prop, d) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const long long min_items = m ? atoi(m) : 3ll * cus;
  return (long long)n * (w / 32) * rs_rsplit(n, h, w) >= min_items;
}
int rs_tiles(int n, int h, int w) { return n * (w / 32) * rs_rsplit(n, h, w); }

// returns 1 when the launch was taken (*rc = its status), 0 when it is not this kernel's
int rs_try_launch(const IgemmParams& p, int prec, const TapSet& taps, hipStream_t s, int* rc) {
  RsParams rp;
  memset(&rp, 0, sizeof(rp));
  if (!rs_enabled() || prec != PCUDA_PREC_BF16X3 || p.cin != 32 || p.cout != 32 || !rs_taps(taps, rp.tap_pos)) return 0;
  if (p.in_step != 1 || p.in_shift || p.pair || p.mask_a || p.fold || p.oy_mul != 1 || p.ox_mul != 1 || p.oy_off || p.ox_off) return 0;
  if (p.lh != p.in_h || p.lw != p.in_w || p.out_w != p.in_w || p.in_row != p.in_w || !rs_map_ok(p.n, p.in_h, p.in_w)) return 0;
  if (!(p.slope >= 0.f && p.slope <= 1.f)) return 0;
  // one source, one destination, float4 rows (16-byte aligned planes and row segments), an image's 32 planes inside 31 bits
  if (p.x.c1 < 32 || p.y.c1 < 32) return 0;
  auto al = [](const void* q, long long sn, long long sc) { return (((uintptr_t)q) & 15) == 0 && (sn & 3) == 0 && (sc & 3) == 0 && sc < (1ll << 24); };
  if (!al(p.x.p1, p.x.sn1, p.x.sc1) || !al(p.y.p1, p.y.sn1, p.y.sc1)) return 0;
  if (p.red_a && (!p.stats || !p.red_mean || !p.red_invstd || !al(p.red_a, p.red_sn, p.red_sc))) return 0;
  rp.x = p.x; rp.y = p.y;
  rp.H = p.in_h; rp.W = p.in_w; rp.n = p.n;
  rp.wimg = (const unsigned char*)p.wpack;
  rp.bias = p.bias; rp.slope = p.slope;
  rp.stats = p.stats;
  rp.red_a = p.red_a; rp.red_sn = p.red_sn; rp.red_sc = p.red_sc; rp.red_mean = p.red_mean; rp.red_invstd = p.red_invstd;
  rp.strips = p.in_w / 32; rp.rsplit = rs_rsplit(p.n, p.in_h, p.in_w);
  rp.items = p.n * rp.strips * rp.rsplit;
  const int grid = (rp.items + 3) / 4;
  rp.xcd = (grid >= 16 && (grid & 7) == 0) ? 1 : 0;
  const bool aff = p.x.scale1 != nullptr;
  const int st = p.red_a ? 2 : (p.stats ? 1 : 0);
  const bool acc = p.accumulate != 0;
  const double flops = 2.0 * p.n * (double)p.in_h * p.in_w * 32 * 32.0 * 9;
  char tag[160];
  snprintf(tag, sizeof(tag), "conv3rs n%d red32 rows32 %dx%d taps9 stats%d acc%d aff%d items%d", p.n, p.in_h, p.in_w, st, acc ? 1 : 0,
           aff ? 1 : 0, rp.items);
  ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
#ifdef PCUDA_RS_DEBUG
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("PCUDA_RSDBG"); dbg = e ? atoi(e) : 0; }
    if (dbg && st == 1 && !acc && !aff) {
      auto go = [&](auto D) {
        constexpr int DV = decltype(D)::value;
        (void)hipFuncSetAttribute((const void*)conv3rs_kernel<1, false, false, DV>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
        hipLaunchKernelGGL((conv3rs_kernel<1, false, false, DV>), dim3(grid), dim3(256), RS_LDS, s, rp);
      };
      switch (dbg) {
        case 1: go(std::integral_constant<int, 1>{}); break;   case 2: go(std::integral_constant<int, 2>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;   case 8: go(std::integral_constant<int, 8>{}); break;
        case 6: go(std::integral_constant<int, 6>{}); break;   case 7: go(std::integral_constant<int, 7>{}); break;
        case 14: go(std::integral_constant<int, 14>{}); break; case 15: go(std::integral_constant<int, 15>{}); break;
        case 9: go(std::integral_constant<int, 9>{}); break;   case 3: go(std::integral_constant<int, 3>{}); break;
        default: go(std::integral_constant<int, 0>{});
      }
      *rc = PCUDA_OK;
      note_kernel("conv3rs");
      return 1;
    }
  }
#endif
#define RS_GO(ST, AC, AF) *rc = rs_launch_t<ST, AC, AF>(rp, grid, s)
  if (st == 0) { if (acc) { if (aff) RS_GO(0, true, true); else RS_GO(0, true, false); } else { if (aff) RS_GO(0, false, true); else RS_GO(0, false, false); } }
  else if (st == 1) { if (acc) { if (aff) RS_GO(1, true, true); else RS_GO(1, true, false); } else { if (aff) RS_GO(1, false, true); else RS_GO(1, false, false); } }
  else { if (acc) { if (aff) RS_GO(2, true, true); else RS_GO(2, true, false); } else { if (aff) RS_GO(2, false, true); else RS_GO(2, false, false); } }
#undef RS_GO
  note_kernel(st == 2 ? "conv3rs+bnred" : "conv3rs");
  return 1;
}
