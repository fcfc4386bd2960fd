// Implicit-GEMM (im2col-free) NCHW convolution on the CDNA4 matrix cores.
//
// Replaces nn.Conv2d forward / backward of the reference (unet.py:23,27,32,61,85,112,116,122,178;
// GAN.py:97-107).  Activations stay fp32 NCHW in HBM; a block stages a haloed input tile of
// 32 channels into LDS pixel-major ([pixel][32 ch + 8 pad] bf16, 80-B records -> conflict-free
// ds_read_b128), splitting fp32 into bf16 hi (+ lo) on the way, and the four waves issue
// v_mfma_f32_32x32x16_bf16 with D[row = output channel][col = pixel] so that the accumulator
// registers store straight into NCHW rows (32 consecutive pixels = 128 B per register).
//   forward / dgrad : igemm_kernel  (dgrad = the same kernel over role-swapped packed weights,
//                     one launch per stride-parity class = transposed convolution)
//   wgrad           : wgrad_kernel  (reduction over pixels; X read back through
//                     ds_read_b64_tr_b16, split-K partial slabs + deterministic reduce)
#include <algorithm>
#include <stdlib.h>

#include "conv_igemm.h"
#include "conv_host.h"

// ------------------------------------------------------------------------------------------
// weight repack: fp32 [rows][red][taps] (any strides) -> bf16 [co_tile][chunk][tap][CO_TILE][40]
// ------------------------------------------------------------------------------------------
__global__ void pack_kernel(const PackParams p) {
  const long long total = (long long)p.n_co_tiles * p.nchunks * p.ntaps * p.co_tile * p.rec;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    // record = 40 bf16 (bf16 mode: 32 values + 8 pad) or 72 (bf16x3: 32 hi | 32 lo | 8 pad)
    int col = (int)(idx % p.rec);
    long long rest = idx / p.rec;
    int row = (int)(rest % p.co_tile); rest /= p.co_tile;
    int t = (int)(rest % p.ntaps); rest /= p.ntaps;
    int ch = (int)(rest % p.nchunks);
    int cot = (int)(rest / p.nchunks);
    const bool is_lo = p.rec > IG_REC && col >= 32;
    const int cc = is_lo ? col - 32 : col;
    int r = cot * p.co_tile + row, c = ch * 32 + cc;
    float v = 0.f;
    if (cc < 32 && r < p.rows && c < p.red)
      v = p.w[(p.pair ? r >> 1 : r) * p.s_row + c * p.s_red + p.tap_src[p.pair ? (r & 1) * p.ntaps + t : t]];
    __bf16 hi = (__bf16)v;
    if (is_lo) hi = (__bf16)(v - (float)hi);
    p.out[idx] = __builtin_bit_cast(uint16_t, hi);
  }
}


// several repacks of one weight tensor in one launch (forward layout + every dgrad parity class): blockIdx.y = job
#define PACK_MAX_JOBS 5
struct PackJobs {
  int n;
  PackParams p[PACK_MAX_JOBS];
};
__global__ void pack_multi_kernel(const PackJobs jobs) {
  const PackParams& p = jobs.p[blockIdx.y];
  const long long total = (long long)p.n_co_tiles * p.nchunks * p.ntaps * p.co_tile * p.rec;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    // record = 40 bf16 (bf16 mode: 32 values + 8 pad) or 72 (bf16x3: 32 hi | 32 lo | 8 pad)
    int col = (int)(idx % p.rec);
    long long rest = idx / p.rec;
    int row = (int)(rest % p.co_tile); rest /= p.co_tile;
    int t = (int)(rest % p.ntaps); rest /= p.ntaps;
    int ch = (int)(rest % p.nchunks);
    int cot = (int)(rest / p.nchunks);
    const bool is_lo = p.rec > IG_REC && col >= 32;
    const int cc = is_lo ? col - 32 : col;
    int r = cot * p.co_tile + row, c = ch * 32 + cc;
    float v = 0.f;
    if (cc < 32 && r < p.rows && c < p.red)
      v = p.w[(p.pair ? r >> 1 : r) * p.s_row + c * p.s_red + p.tap_src[p.pair ? (r & 1) * p.ntaps + t : t]];
    __bf16 hi = (__bf16)v;
    if (is_lo) hi = (__bf16)(v - (float)hi);
    p.out[idx] = __builtin_bit_cast(uint16_t, hi);
  }
}

// every repack of a network in ONE launch: the jobs (one per layout: forward + each dgrad image of every layer) live in
// device memory; a workgroup finds its job in the table of first-block indices behind the jobs.  (Packed lazily, layer by
// layer, the 43 small launches of the segmenter sat in the dependent chain of the next forward pass.)
// A workgroup owns PACK_ROWS rows of one (co-tile, chunk) for EVERY tap: the fp32 source of those rows x 32 reduction
// indices x all taps is a few whole lines, read once by the workgroup that needs all of it (round 3: with one tap per
// workgroup the nine workgroups sharing a source line sat on different XCDs and FETCH_SIZE was 9x the weights, 673 MB per
// launch for 76 MB), and each tap's PACK_ROWS records are whole lines of the packed image (8 x 144 B = 9 lines, 8 x 80 = 5).
#define PACK_ROWS 8
__global__ __launch_bounds__(256) void pack_table_kernel(const PackParams* __restrict__ jobs, const int* __restrict__ first_block,
                                                         int njobs) {
  int lo = 0, hi = njobs - 1;          // last job whose first block is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackParams& p = jobs[lo];
  const int blk = (int)blockIdx.x - first_block[lo];
  const int rgs = p.co_tile / PACK_ROWS;                      // row groups per co-tile (4 or 8)
  const int rg = blk % rgs, ch = (blk / rgs) % p.nchunks, cot = blk / (rgs * p.nchunks);
  if (cot >= p.n_co_tiles) return;
  // the source of this group: (source row, reduction index) -> kk = k x k contiguous taps; a source row's reduction
  // indices are contiguous for the forward image (s_red = kk), a reduction index's rows for the data gradient's
  // (s_row = kk).  Copied run by run, in memory order, into LDS ([row][c][tap] resp. [c][row][tap]) and packed from there
  // 8 values = one 16-byte vector per item: per ELEMENT index arithmetic (two divisions for two bytes) made the first
  // form of this kernel instruction-bound at 0.2 TB/s.
  __shared__ float sw[PACK_ROWS * 32 * IG_MAX_TAPS];
  const int kk = (int)(p.s_row < p.s_red ? p.s_row : p.s_red);
  const int nsr = PACK_ROWS >> p.pair, r0s = (cot * p.co_tile + rg * PACK_ROWS) >> p.pair, rows_src = p.rows >> p.pair;
  const int c0 = ch * 32;
  const int nr = min(nsr, rows_src - r0s), nc = min(32, p.red - c0);     // valid source rows / reduction indices
  const bool by_row = p.s_red <= p.s_row;
  for (int f = threadIdx.x; f < PACK_ROWS * 32 * kk; f += 256) sw[f] = 0.f;
  __syncthreads();
  {   // eight loads in flight per lane (one load per loop trip waited for each one: the kernel was latency-bound at 0.3 TB/s)
    const int nseg = by_row ? nr : nc, runlen = (by_row ? nc : nr) * kk;
    const long long seg_src = by_row ? p.s_row : p.s_red;
    const int seg_lds = (by_row ? 32 : nsr) * kk;
    const float* src0 = p.w + (by_row ? (long long)r0s * p.s_row + (long long)c0 * kk : (long long)r0s * kk + (long long)c0 * p.s_red);
    const int total = nseg * runlen;
    for (int base = 0; base < total; base += 256 * 8) {
      float v[8];
      int dst[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = base + j * 256 + (int)threadIdx.x;
        const int seg = idx / runlen, f = idx - seg * runlen;
        dst[j] = idx < total ? seg * seg_lds + f : -1;
        v[j] = idx < total ? src0[seg * seg_src + f] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (dst[j] >= 0) sw[dst[j]] = v[j];
    }
  }
  __syncthreads();
  if (p.rec == 32) {
    // anti-phase image (conv_ap_impl.h): [co tile][16-channel chunk][tap][row 0..63][64 B]; a record's four 16-byte pieces
    // (hi k 0-7, hi k 8-15, lo k 0-7, lo k 8-15) sit in slot q ^ ((row >> 2) & 3): the image IS the kernel's LDS image
    const int nch16 = p.red >> 4;
    for (int it = threadIdx.x; it < p.ntaps * PACK_ROWS * 8; it += 256) {
      const int t = it / (PACK_ROWS * 8), rem = it - t * (PACK_ROWS * 8);
      const int rowi = rem >> 3, vec = rem & 7;
      const int half16 = vec >> 2, q = vec & 3;
      const int c16 = ch * 2 + half16;
      if (c16 >= nch16) continue;
      const int cb = half16 * 16 + (q & 1) * 8;
      const int ts = p.tap_src[t];
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = cb + 2 * j;
        const float v0 = sw[(by_row ? rowi * 32 + cc : cc * nsr + rowi) * kk + ts];
        const float v1 = sw[(by_row ? rowi * 32 + cc + 1 : (cc + 1) * nsr + rowi) * kk + ts];
        uint32_t hi, lo;
        split2(v0, v1, hi, lo);
        o[j] = (q & 2) ? lo : hi;
      }
      const int row = rg * PACK_ROWS + rowi;
      unsigned char* dst = (unsigned char*)p.out + ((((long long)cot * nch16 + c16) * 9 + t) * 64 + row) * 64 + ((q ^ ((row >> 2) & 3)) << 4);
      *(uint4*)dst = make_uint4(o[0], o[1], o[2], o[3]);
    }
    return;
  }
  const bool x3 = p.rec > IG_REC;
  const int nvec = p.rec >> 3;                                // 16-byte vectors per record: 4 hi (+ 4 lo) + 1 pad
  const int per_t = PACK_ROWS * nvec;
  const long long tile0 = (long long)(cot * p.nchunks + ch) * p.ntaps;
  for (int it = threadIdx.x; it < p.ntaps * per_t; it += 256) {
    const int t = it / per_t, rem = it - t * per_t;
    const int rowi = rem / nvec, vec = rem - rowi * nvec;
    const bool pad = vec == nvec - 1, is_lo = x3 && vec >= 4;
    const int cb = (vec & 3) * 8, ri = rowi >> p.pair;
    const int ts = p.tap_src[p.pair ? (rowi & 1) * p.ntaps + t : t];
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v0 = 0.f, v1 = 0.f;
      if (!pad) {
        const int cc = cb + 2 * j;
        v0 = sw[(by_row ? ri * 32 + cc : cc * nsr + ri) * kk + ts];
        v1 = sw[(by_row ? ri * 32 + cc + 1 : (cc + 1) * nsr + ri) * kk + ts];
      }
      uint32_t hi, lo;
      split2(v0, v1, hi, lo);
      o[j] = is_lo ? lo : hi;
    }
    const int row = rg * PACK_ROWS + rowi;
    *(uint4*)(p.out + ((tile0 + t) * p.co_tile + row) * p.rec + vec * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
static inline int pack_table_blocks(const PackParams& p) { return p.n_co_tiles * p.nchunks * (p.co_tile / PACK_ROWS); }

// ==========================================================================================
// host side
// ==========================================================================================
int igemm_dispatch_x3(const IgemmParams& p, const IgemmPlan& pl, int co_blks, int pf, bool pipe, hipStream_t s);
int igemm_dispatch_bf16(const IgemmParams& p, const IgemmPlan& pl, int co_blks, int pf, bool pipe, hipStream_t s);

namespace {

// PCUDA_FAT=1 plans one workgroup per CU with all taps resident (experimental: the weight stash spills)
unsigned long long* g_dbg_clk_host = nullptr;
bool ig_wstay_mode() {   // PCUDA_WSTAY=1: measured slower on the 256x256 32-channel level (0.38 vs 0.33 ms)
  static int m = -1;
  if (m < 0) { const char* e = getenv("PCUDA_WSTAY"); m = (e && atoi(e)) ? 1 : 0; }
  return m != 0;
}
// LDS / workgroup plan: 0 = two (or three) 256-thread workgroups per CU with small weight groups,
// 1 (PCUDA_FAT=1, experimental) = one 256-thread workgroup with two alternating weight buffers,
// 2 (default; PCUDA_W8=0 turns it off) = one 512-thread workgroup with every tap resident where it fits
int ig_plan_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* f = getenv("PCUDA_FAT");
    const char* w = getenv("PCUDA_W8");
    (void)f;   // PCUDA_FAT: the fat variant (mode 1) measured slower everywhere and is no longer built
    mode = (w && !atoi(w)) ? 0 : 2;
  }
  return mode;
}

size_t packed_elems(int rows, int red, int ntaps, int prec) {   // bf16 elements of one packed layout (both planes)
  const int co_tile = 32 * ig_co_blks(rows);
  const int n_co_tiles = cdiv(rows, co_tile), nchunks = cdiv(red, 32);
  return (size_t)n_co_tiles * nchunks * ntaps * co_tile * (ig_rec_bytes(prec == PCUDA_PREC_BF16X3) / 2);
}

// pair: rows = 2 x the source rows, (source row, column class) interleaved; taps.src holds both classes' sources
size_t fill_pack(PackParams& p, const float* w, uint16_t* out, int prec, int rows, int red, long long s_row,
                 long long s_red, const TapSet& taps, bool pair = false) {
  p.w = w; p.out = out;
  p.rows = rows; p.red = red; p.s_row = s_row; p.s_red = s_red;
  p.ntaps = taps.n;
  p.pair = pair ? 1 : 0;
  memcpy(p.tap_src, taps.src, sizeof(p.tap_src));
  p.co_tile = 32 * ig_co_blks(rows);
  p.n_co_tiles = cdiv(rows, p.co_tile);
  p.nchunks = cdiv(red, 32);
  const size_t plane = packed_elems(rows, red, taps.n, prec);
  p.rec = ig_rec_bytes(prec == PCUDA_PREC_BF16X3) / 2;
  return plane;
}

int launch_pack(const float* w, uint16_t* out, int prec, int rows, int red, long long s_row, long long s_red,
                const TapSet& taps, hipStream_t s, bool pair = false) {
  if (taps.n == 0) return PCUDA_OK;
  PackParams p;
  const size_t plane = fill_pack(p, w, out, prec, rows, red, s_row, s_red, taps, pair);
  const int blocks = (int)((plane + 255) / 256 > 4096 ? 4096 : (plane + 255) / 256);
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, s, p);
  PCUDA_CHECK_LAUNCH("pack_kernel");
  return PCUDA_OK;
}

// pick taps-per-group and LDS size; returns <0 when nothing fits.  fat: one workgroup per CU (160 KiB),
// as many taps resident as one 256*WV-vector copy pass holds.
int plan_lds(bool x3, int co_tile, int x_cap, int ntaps, int mode, int* tg_out, size_t* lds_out) {
  const bool fat = mode != 0;
  const size_t rec = ig_rec_bytes(x3), vecs = rec / 16;
  const size_t xb = (size_t)x_cap * rec;
  const size_t wtap = (size_t)co_tile * rec;
  const size_t epi = 4 * (size_t)co_tile * 2 * sizeof(float);
  const size_t tab = 1024;  // per-tap offset table + the epilogue's bias / mean / invstd slices, behind the weight slabs
  if (ntaps < 1) ntaps = 1;
  // one copy pass per group: the kernels' register slots (16-byte vectors per lane) x lanes; a bf16x3 record is 9
  // vectors (both planes), a bf16 record 5
  const int slots = mode == 2 ? (x3 ? 11 : 6) * 512 : (x3 ? (co_tile == 32 ? 8 : 7) : (co_tile == 32 ? 4 : 3)) * 256;
  const int cap = slots / (int)(co_tile * vecs);
  // budgets: 3, 2, 1 workgroups per CU (160 KiB LDS)
  const size_t budgets[3] = {54528, 81920, 163840};
  for (int b = fat ? 2 : 0; b < 3; ++b) {
    if (xb + tab >= budgets[b]) continue;
    int fit = (int)((budgets[b] - xb - tab) / wtap);
    if (fit > cap) fit = cap;
    const int need = b == 0 ? (ntaps < 3 ? ntaps : 3) : 1;
    if (fit < need) continue;
    const int tg = fit > ntaps ? ntaps : fit;
    size_t total = xb + (size_t)tg * wtap + tab;
    if (total < epi) total = epi;
    *tg_out = tg; *lds_out = total;
    return 0;
  }
  return -1;
}


// tile shape / LDS plan of one generic launch: depends only on geometry, taps and precision.
// Minimises the number of MFMA pixel slots (tiles x slots per tile); ties prefer 32-pixel-aligned rows
// (128-B output segments), two pixel blocks per wave, wider tiles.
int plan_igemm_mode(int rows, int lh, int lw, int in_h, int in_w, int in_step, const TapSet& taps, bool x3,
                    int mode_env, IgemmPlan* best, int n = 0) {
  const int co_tile = 32 * ig_co_blks(rows);
  long long best_key = -1;
  for (int npb = 2; npb >= 1; --npb) {
    const int TP = 128 * npb;
    int cand[16];
    const int nc = tile_width_candidates(lw, TP, cand);
    for (int ci = 0; ci < nc; ++ci) {
      IgemmPlan pl;
      pl.npb = npb; pl.tw = cand[ci]; pl.th = TP / pl.tw;
      if (pl.th < 1) continue;
      pl.tiles_x = cdiv(lw, pl.tw);
      pl.tiles_y = cdiv(lh, pl.th);
      pl.ih_t = (pl.th - 1) * in_step + (taps.dy_max - taps.dy_min) + 1;
      pl.iw_t = (pl.tw - 1) * in_step + (taps.dx_max - taps.dx_min) + 1;
      const int full = pl.ih_t * pl.iw_t;
      const int clipped = (pl.ih_t < in_h ? pl.ih_t : in_h) * (pl.iw_t < in_w ? pl.iw_t : in_w) + 1;
      bool ok = false;
      // clamp mode pays ~10 VALU per tap and lane; use it when the halo is mostly padding
      for (int attempt = 0; attempt < 2 && !ok; ++attempt) {
        pl.clamp = attempt == 0 ? ((clipped * 2 <= full) ? 1 : 0) : 1;
        pl.x_cap = pl.clamp ? clipped : full;
        // the eight-wave kernel needs 8 or 16 MFMA tiles per stage and an input tile of <= 2 slots per lane
        int mode = mode_env;
        if (mode == 2) {
          // (only a clamped tile is clipped to the image; an unclamped one stages every halo row, image or padding)
          const int ihc = (pl.clamp && pl.ih_t > in_h) ? in_h : pl.ih_t;
          const bool fits = pl.x_cap <= 1024 && ihc * ((pl.iw_t + 6) / 4) <= 256;
          if (!fits || (npb == 1 && co_tile == 32)) mode = 0;
        }
        pl.fat = mode == 1 ? 1 : 0;
        pl.w8 = mode == 2 ? 1 : 0;
        ok = plan_lds(x3, co_tile, pl.x_cap, taps.n, mode, &pl.tg, &pl.lds) == 0;
        if (pl.clamp) break;
      }
      if (!ok) continue;
      // primary: MFMA pixel slots; then 32-pixel-aligned rows (128-B output segments: measured faster than
      // 16x16 tiles despite their smaller halo); then staged input pixels per output slot (halo overhead,
      // in 1/64ths); then two pixel blocks per wave
      const long long slots = (long long)pl.tiles_x * pl.tiles_y * TP;
      const long long halo = (long long)pl.x_cap * 64 / TP;
      // (then the number of weight groups per stage: every group is a copy and a barrier pair; the 224x224 network's 29x29
      // discriminator maps took a 256-pixel tile whose halo left room for ONE tap of weights per group, 16 groups, over a
      // 128-pixel tile with eight)
      const long long groups = (taps.n + pl.tg - 1) / pl.tg;
      // (eight-wave plans only: fewer (tile, co-tile) items than compute units -- the 16x16 maps at a per-rank batch of 16 --
      // rank behind the 128-pixel tiling that doubles them)
      const long long items = (long long)n * pl.tiles_x * pl.tiles_y * cdiv(rows, co_tile);
      static int nofill = -1;      // PCUDA_NO_UNDERFILL=1: without this term (A/B)
      if (nofill < 0) { const char* e = getenv("PCUDA_NO_UNDERFILL"); nofill = (e && atoi(e)) ? 1 : 0; }
      const long long underfill = (!nofill && n > 0 && items < 256) ? (256 - items + 63) / 64 : 0;      // 0 .. 4, in quarters of the chip
      const long long key = (slots << 28) + (underfill << 25) + ((groups > 3 ? groups : 3) << 22) + ((pl.tw & 31) ? (1ll << 20) : 0) +
                            (halo << 8) + (npb == 1 ? 1 : 0);
      if (best_key < 0 || key < best_key) { best_key = key; *best = pl; }
    }
  }
  return best_key < 0 ? -1 : 0;
}

// Per layer: the eight-wave kernel (mode 2) where the ordinary plan would leave a CU with one 256-thread
// workgroup anyway (its LDS tile > 80 KiB: the stride-2 4x4 layers) or where there are not enough
// (tile, co-tile) items for two workgroups per CU (16x16 maps); measured slower elsewhere (3x3 layers at
// 32x32 and up, the 4-tap dgrad classes), where two independent workgroups per CU overlap better.
int plan_igemm(int rows, int red, int n, int lh, int lw, int in_h, int in_w, int in_step, const TapSet& taps, bool x3,
               IgemmPlan* best, bool allow8 = true) {
  const int mode_env = allow8 ? ig_plan_mode() : 0;
  const bool single_chunk = red <= 32;
  IgemmPlan p0;
  const int rc0 = plan_igemm_mode(rows, lh, lw, in_h, in_w, in_step, taps, x3, mode_env == 2 ? 0 : mode_env, &p0);
  if (mode_env != 2) { *best = p0; return rc0; }
  IgemmPlan p8;
  const int rc8 = plan_igemm_mode(rows, lh, lw, in_h, in_w, in_step, taps, x3, 2, &p8, n);
  if (rc8 < 0 || !p8.w8) { *best = p0; return rc0; }
  if (rc0 < 0) { *best = p8; return rc8; }
  // (counted on the coarser of the two tilings: the four-wave plan may prefer 128-pixel tiles for their weight groups)
  const long long t0 = (long long)p0.tiles_x * p0.tiles_y, t8 = (long long)p8.tiles_x * p8.tiles_y;
  const long long items = (long long)n * (t0 < t8 ? t0 : t8) * cdiv(rows, 32 * ig_co_blks(rows));
  // (one chunk, one co-tile, every tap resident: the eight-wave kernel loads the weights once per workgroup)
  const bool wstay = rows <= 64 && taps.n <= p8.tg && single_chunk;
  const bool use8 = p0.lds > 81920 || items <= 320 || (wstay && ig_wstay_mode());
  *best = use8 ? p8 : p0;
  return 0;
}

// ap_image: the layer's anti-phase image (behind its ordinary packed layout), or NULL
int launch_igemm(IgemmParams& p, int prec, const TapSet& taps, hipStream_t s, const unsigned char* ap_image = nullptr) {
  {
    int rc;
    if (ap_image && prec == PCUDA_PREC_BF16X3 && ap_try_launch(p, taps, ap_image, s, &rc)) return rc;
    if (p.cin == 32 && p.cout == 32 && rs_try_launch(p, prec, taps, s, &rc)) return rc;
  }
  const bool x3 = prec == PCUDA_PREC_BF16X3;
  const int co_blks = ig_co_blks(p.cout);
  const int co_tile = 32 * co_blks;
  IgemmPlan pl;
  if (plan_igemm(p.cout, p.cin, p.n, p.lh, p.lw, p.in_h, p.in_w, p.in_step, taps, x3, &pl, true) < 0)
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "igemm: no tile of this convolution fits LDS (in_step %d, tap span %d)",
               p.in_step, taps.dy_max - taps.dy_min);
  // (a launch the anti-phase kernel was expected to take -- the caller sized its partial sums by THAT kernel's tiles -- but did
  //  not, e.g. misaligned tensors: only if this plan writes the same tiles)
  if (ap_image && p.stats && ap_map_ok(p.n, p.in_h, p.in_w, p.cout) && p.in_step == 1 && !p.fold &&
      (long long)pl.tiles_x * pl.tiles_y * p.n != ap_tiles(p.n, p.in_h, p.in_w))
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv: tensors the anti-phase kernel cannot address on a map whose ordinary plan has other tiles");
  // (the same for the row-streaming kernel of the 32 -> 32 layers: its tiles are its work items)
  if (p.stats && p.cin == 32 && p.cout == 32 && prec == PCUDA_PREC_BF16X3 && taps.n == 9 && p.in_step == 1 && !p.in_shift && !p.fold &&
      !p.pair && taps.dy_max - taps.dy_min == 2 && rs_map_ok(p.n, p.in_h, p.in_w) && p.lh == p.in_h && p.lw == p.in_w &&
      (long long)pl.tiles_x * pl.tiles_y * p.n != rs_tiles(p.n, p.in_h, p.in_w))
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv: tensors the row-streaming kernel cannot address on a map whose ordinary plan has other tiles");
  p.n_co_tiles = cdiv(p.cout, co_tile);
  p.nchunks = cdiv(p.cin, 32);
  p.tw = pl.tw; p.th = pl.th; p.tmagic = 65536 / pl.tw + 1; p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y;
  auto fmagic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d) + 1u; };
  p.m_cot = fmagic(p.n_co_tiles); p.m_tx = fmagic(pl.tiles_x); p.m_ty = fmagic(pl.tiles_y);
  p.ntaps = taps.n;
  memcpy(p.dy, taps.dy, sizeof(p.dy));
  memcpy(p.dx, taps.dx, sizeof(p.dx));
  p.dy_min = taps.dy_min; p.dx_min = taps.dx_min;
  p.ih_t = pl.ih_t; p.iw_t = pl.iw_t; p.clamp = pl.clamp; p.tg = pl.tg;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("PCUDA_DBG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
    static unsigned long long* clkbuf = nullptr;
    if ((dbg & 128) && !clkbuf && hipMalloc((void**)&clkbuf, 64) == hipSuccess) (void)hipMemset(clkbuf, 0, 64);
    p.dbg_clk = clkbuf;
    g_dbg_clk_host = clkbuf;
  }
  const double flops = 2.0 * p.n * (double)p.lh * p.lw * p.cout * (double)p.cin * taps.n;
  char tag[160];
  snprintf(tag, sizeof(tag), "igemm n%d red%d rows%d %dx%d taps%d step%d up%d tw%d npb%d clamp%d tg%d w8%d lds%zu", p.n,
           p.cin, p.cout, p.lh, p.lw, taps.n, p.in_step, p.in_shift, pl.tw, pl.npb, pl.clamp, pl.tg, pl.w8, pl.lds);
  ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
  static int nopipe = -1;
  if (nopipe < 0) { const char* e = getenv("PCUDA_NOPIPE"); nopipe = e ? atoi(e) : 0; }
  const int max_pix = pl.clamp ? pl.x_cap - 1 : pl.ih_t * pl.iw_t;     // largest LDS tile of this launch
  int pf = (max_pix + 255) / 256;
  {   // quad staging: rows of 4k pixels, no upsampling fold, (rows x quads) of a tile in <= 3 x 64 lanes
    static int noxq = -1;
    if (noxq < 0) { const char* e = getenv("PCUDA_NOXQ"); noxq = e ? atoi(e) : 0; }
    const int ih = (pl.clamp && pl.ih_t > p.in_h) ? p.in_h : pl.ih_t;   // rows the kernel stages (unclamped: all of the halo)
    const int pfq = (ih * ((pl.iw_t + 6) / 4) + 63) / 64;
    p.xq = (!noxq && (p.in_w & 3) == 0 && p.in_shift == 0 && pfq <= 3) ? 1 : 0;
    if (p.xq) pf = pfq;
    if (pl.w8) pf = p.xq ? (ih * ((pl.iw_t + 6) / 4) + 127) / 128 : (max_pix + 511) / 512;   // 512 lanes
  }
  // PF = 3 keeps 96 prefetch registers live next to the accumulators: only with one pixel block per wave
  // (one workgroup per CU has 512 registers per lane: PF = 3 next to two pixel blocks fits there)
  // (the unpipelined fallback copies its weight groups in 512-vector passes: any tg of the plan works)
  // (the persistent kernels decode tile indices with multiply-high magic numbers, exact below 2^32 / divisor)
  const long long total_items = (long long)p.n_co_tiles * p.n * pl.tiles_x * pl.tiles_y;
  const int max_div = std::max(p.n_co_tiles, std::max(pl.tiles_x, pl.tiles_y));
  const bool pipe = !nopipe && p.ntaps > 0 && fast_src_ok(&p.x, p.cin) && fast_dst_ok(&p.y, p.cout) &&
                    (pf <= 2 || (pf == 3 && !pl.w8 && (pl.npb == 1 || pl.fat))) && total_items < (1ll << 32) / max_div;
  {   // transposed epilogue: its LDS scratch (4 waves x [32][32*npb] fp32 + the partial-sum slots) aliases the
      // X / W slabs and must end in front of the tap table behind them
    static int note = -1;
    if (note < 0) { const char* e = getenv("PCUDA_NOTE"); note = e ? atoi(e) : 0; }
    // (2x2 fold: y is the half-resolution tensor, stored 8 bytes at a time: even rows, 8-byte aligned planes)
    const bool fold_ok = p.fold && p.ox_mul == 1 && p.ox_off == 0 && !(pl.tw & 3) && !(p.lw & 3) && !(p.out_w & 1) &&
                         !((uintptr_t)p.y.p1 & 7) && !(p.y.sc1 & 1) && !(p.y.sn1 & 1) &&
                         (p.y.c1 >= p.cout || (!((uintptr_t)p.y.p2 & 7) && !(p.y.sc2 & 1) && !(p.y.sn2 & 1)));
    pl.te = (pipe && !pl.w8 && !note &&
             (p.fold ? fold_ok : te_dst_ok(&p.y, p.cout, p.out_w, p.lw, pl.tw, p.ox_mul, p.ox_off))) ? 1 : 0;
    if (pl.te) {
      const size_t rec = ig_rec_bytes(x3), wtap = (size_t)co_tile * rec;
      const size_t need = (size_t)16384 * pl.npb + (size_t)co_tile * 32;
      const size_t have = (size_t)pl.x_cap * rec + (size_t)pl.tg * wtap;
      if (have < need) {
        const int add = (int)((need - have + rec - 1) / rec);
        pl.x_cap += add;
        pl.lds += (size_t)add * rec;
      }
    }
  }
  if (p.fold && !(pipe && pl.te && !pl.w8 && pl.npb == 2 && pl.tw == 32 && pl.th == 8 && !pl.clamp && !p.accumulate && !p.pair))
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_dgrad_fold: this geometry does not run on the 32 x 8-tile transposed-epilogue kernel");
  if (p.mask_a && (pl.te || p.accumulate || p.stats || p.bias || p.fold || p.y.c1 < (p.cout >> p.pair)))
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_dgrad_lrelu: this geometry does not run on a plain-epilogue kernel");
  if (p.red_a && !(pipe && pl.te))
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_dgrad_bnred: this geometry does not run on the transposed-epilogue kernel");
  return x3 ? igemm_dispatch_x3(p, pl, co_blks, pf, pipe, s) : igemm_dispatch_bf16(p, pl, co_blks, pf, pipe, s);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// (the anti-phase image of a layer that has one sits BEHIND its ordinary layout: every launch can still take the ordinary kernels)
extern "C" size_t pcuda_conv2d_packed_fwd_bytes(const pcuda_conv_geom* g, int prec) {
  if (!geom_ok(g)) return 0;
  return packed_elems(g->cout, g->cin, g->k * g->k, prec) * 2 + (ap_layer_ok(g, g->cout, g->cin, prec) ? ap_layer_packed_bytes(g->cout, g->cin) : 0);
}

extern "C" size_t pcuda_conv2d_packed_dgrad_bytes(const pcuda_conv_geom* g, int prec) {
  if (!geom_ok(g)) return 0;
  size_t tot = 0;
  if (dgrad_pair_ok(g)) {
    for (int ry = 0; ry < 2; ++ry) tot += packed_elems(2 * g->cin, g->cout, dgrad_taps(g, ry, 0).n, prec) * 2;
    return tot;
  }
  for (int ry = 0; ry < g->stride; ++ry)
    for (int rx = 0; rx < g->stride; ++rx) {
      TapSet t = dgrad_taps(g, ry, rx);
      tot += packed_elems(g->cin, g->cout, t.n, prec) * 2;
    }
  if (ap_layer_ok(g, g->cin, g->cout, prec)) tot += ap_layer_packed_bytes(g->cin, g->cout);
  return tot;
}

extern "C" int pcuda_conv2d_pack_fwd(const pcuda_conv_geom* g, int prec, const float* w, void* packed,
                                     pcuda_stream_t s) {
  if (!geom_ok(g) || !w || !packed) PCUDA_FAIL(PCUDA_E_BADARG, "pack_fwd: bad geometry or null pointer");
  const int kk = g->k * g->k;
  TapSet t = fwd_taps(g);
  int rc = launch_pack(w, (uint16_t*)packed, prec, g->cout, g->cin, (long long)g->cin * kk, kk, t, (hipStream_t)s);
  if (rc == PCUDA_OK && ap_layer_ok(g, g->cout, g->cin, prec))
    rc = ap_launch_pack(w, (unsigned char*)packed + packed_elems(g->cout, g->cin, kk, prec) * 2, g->cout, g->cin,
                        (long long)g->cin * kk, kk, t, (hipStream_t)s);
  return rc;
}

extern "C" int pcuda_conv2d_pack_dgrad(const pcuda_conv_geom* g, int prec, const float* w, void* packed,
                                       pcuda_stream_t s) {
  if (!geom_ok(g) || !w || !packed) PCUDA_FAIL(PCUDA_E_BADARG, "pack_dgrad: bad geometry or null pointer");
  const int kk = g->k * g->k;
  uint16_t* out = (uint16_t*)packed;
  if (dgrad_pair_ok(g)) {
    for (int ry = 0; ry < 2; ++ry) {
      TapSet t = dgrad_pair_taps(g, ry);
      int rc = launch_pack(w, out, prec, 2 * g->cin, g->cout, kk, (long long)g->cin * kk, t, (hipStream_t)s, true);
      if (rc) return rc;
      out += packed_elems(2 * g->cin, g->cout, t.n, prec);
    }
    return PCUDA_OK;
  }
  for (int ry = 0; ry < g->stride; ++ry)
    for (int rx = 0; rx < g->stride; ++rx) {
      TapSet t = dgrad_taps(g, ry, rx);
      int rc = launch_pack(w, out, prec, g->cin, g->cout, kk, (long long)g->cin * kk, t, (hipStream_t)s);
      if (rc) return rc;
      out += packed_elems(g->cin, g->cout, t.n, prec);
    }
  if (ap_layer_ok(g, g->cin, g->cout, prec))
    return ap_launch_pack(w, (unsigned char*)out, g->cin, g->cout, kk, (long long)g->cin * kk, dgrad_taps(g, 0, 0), (hipStream_t)s);
  return PCUDA_OK;
}

extern "C" int pcuda_conv2d_pack_all(const pcuda_conv_geom* g, int prec, const float* w, void* packed_fwd,
                                     void* packed_dgrad, pcuda_stream_t s) {
  if (!geom_ok(g) || !w || !packed_fwd) PCUDA_FAIL(PCUDA_E_BADARG, "pack_all: bad geometry or null pointer");
  if (g->stride * g->stride + 1 > PACK_MAX_JOBS) {   // strides > 2: one launch per layout
    int rc = pcuda_conv2d_pack_fwd(g, prec, w, packed_fwd, s);
    if (rc == PCUDA_OK && packed_dgrad) rc = pcuda_conv2d_pack_dgrad(g, prec, w, packed_dgrad, s);
    return rc;
  }
  const int kk = g->k * g->k;
  PackJobs jobs;
  jobs.n = 0;
  size_t maxplane = fill_pack(jobs.p[jobs.n++], w, (uint16_t*)packed_fwd, prec, g->cout, g->cin, (long long)g->cin * kk,
                              kk, fwd_taps(g));
  if (packed_dgrad && dgrad_pair_ok(g)) {
    uint16_t* out = (uint16_t*)packed_dgrad;
    for (int ry = 0; ry < 2; ++ry) {
      const size_t plane = fill_pack(jobs.p[jobs.n++], w, out, prec, 2 * g->cin, g->cout, kk, (long long)g->cin * kk,
                                     dgrad_pair_taps(g, ry), true);
      if (plane > maxplane) maxplane = plane;
      out += plane;
    }
  } else if (packed_dgrad) {
    uint16_t* out = (uint16_t*)packed_dgrad;
    for (int ry = 0; ry < g->stride; ++ry)
      for (int rx = 0; rx < g->stride; ++rx) {
        TapSet t = dgrad_taps(g, ry, rx);
        if (t.n == 0) continue;
        const size_t plane = fill_pack(jobs.p[jobs.n++], w, out, prec, g->cin, g->cout, kk, (long long)g->cin * kk, t);
        if (plane > maxplane) maxplane = plane;
        out += plane;
      }
  }
  const int blocks = (int)((maxplane + 255) / 256 > 2048 ? 2048 : (maxplane + 255) / 256);
  hipLaunchKernelGGL(pack_multi_kernel, dim3(blocks, jobs.n), dim3(256), 0, (hipStream_t)s, jobs);
  PCUDA_CHECK_LAUNCH("pack_multi_kernel");
  if (ap_layer_ok(g, g->cout, g->cin, prec)) {
    int rc = ap_launch_pack(w, (unsigned char*)packed_fwd + packed_elems(g->cout, g->cin, kk, prec) * 2, g->cout, g->cin,
                            (long long)g->cin * kk, kk, fwd_taps(g), (hipStream_t)s);
    if (rc) return rc;
  }
  if (packed_dgrad && ap_layer_ok(g, g->cin, g->cout, prec)) {   // (stride 1: one class in front of the image)
    TapSet t = dgrad_taps(g, 0, 0);
    int rc = ap_launch_pack(w, (unsigned char*)packed_dgrad + packed_elems(g->cin, g->cout, t.n, prec) * 2, g->cin, g->cout, kk,
                            (long long)g->cin * kk, t, (hipStream_t)s);
    if (rc) return rc;
  }
  return PCUDA_OK;
}

extern "C" int pcuda_conv2d_fwd_tiles(const pcuda_conv_geom* g, int prec) {
  if (!geom_ok(g)) return 0;
  if (const int d = direct_fwd_tiles(g)) return d;
  if (ap_layer_ok(g, g->cout, g->cin, prec) && ap_map_ok(g->n, g->in_h, g->in_w, g->cout)) return ap_tiles(g->n, g->in_h, g->in_w);
  if (rs_layer_ok(g, g->cout, g->cin, prec) && rs_map_ok(g->n, g->in_h, g->in_w)) return rs_tiles(g->n, g->in_h, g->in_w);
  TapSet t = fwd_taps(g);
  IgemmPlan pl;
  if (plan_igemm(g->cout, g->cin, g->n, g->out_h, g->out_w, g->in_h, g->in_w, g->stride, t, prec == PCUDA_PREC_BF16X3, &pl) < 0)
    return 0;
  return g->n * pl.tiles_x * pl.tiles_y;
}

extern "C" int pcuda_conv2d_forward(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const void* packed_w,
                                    const float* bias, float slope, const pcuda_dst* y, float* bn_partials,
                                    pcuda_stream_t s) {
  if (!geom_ok(g)) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_forward: inconsistent geometry");
  if (!src_ok(x, g->cin) || !dst_ok(y, g->cout) || !packed_w) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_forward: bad tensors");
  if (prec != PCUDA_PREC_BF16X3 && prec != PCUDA_PREC_BF16) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_forward: bad precision");
  {
    int rc;
    if (direct_forward(g, prec, x, packed_w, 0, bias, slope, y,
                       bn_partials, (hipStream_t)s, &rc))
      return rc;
    if (direct_d5_forward(g, prec, x, packed_w, bias, slope, y, bn_partials, (hipStream_t)s, &rc)) return rc;
    if (direct_d1_forward(g, prec, x, packed_w, bias, slope, y, bn_partials, (hipStream_t)s, &rc)) return rc;
  }
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = *x; p.cin = g->cin;
  p.in_h = g->in_h; p.in_w = g->in_w; p.in_shift = g->in_up ? 1 : 0; p.in_row = g->in_w >> p.in_shift;
  p.y = *y; p.cout = g->cout; p.out_w = g->out_w;
  p.lh = g->out_h; p.lw = g->out_w;
  p.oy_mul = p.ox_mul = 1; p.oy_off = p.ox_off = 0;
  p.in_step = g->stride;
  p.wpack = (const uint16_t*)packed_w;
  p.w_lo_off = 0;   // (both planes live in one record)
  p.bias = bias; p.slope = slope; p.accumulate = 0; p.stats = bn_partials;
  p.n = g->n;
  TapSet t = fwd_taps(g);
  const unsigned char* ap_image = ap_layer_ok(g, g->cout, g->cin, prec)
                                      ? (const unsigned char*)packed_w + packed_elems(g->cout, g->cin, g->k * g->k, prec) * 2 : nullptr;
  return launch_igemm(p, prec, t, (hipStream_t)s, ap_image);
}

// mask_a != NULL: the LeakyReLU backward of the layer in front rides in the epilogue (pcuda_conv2d_dgrad_lrelu)
static int dgrad_impl(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                      const pcuda_dst* dx, int accumulate, const float* mask_a, long long mask_sn, float mask_slope,
                      pcuda_stream_t s) {
  if (!geom_ok(g)) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad: inconsistent geometry");
  if (!src_ok(dy, g->cout) || !dst_ok(dx, g->cin) || !packed_w_dgrad) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad: bad tensors");
  if (prec != PCUDA_PREC_BF16X3 && prec != PCUDA_PREC_BF16) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad: bad precision");
  if (!mask_a) {
    int rc;
    if (direct_dgrad(g, prec, dy, packed_w_dgrad, dx, accumulate, (hipStream_t)s, &rc)) return rc;
    if (direct_d1_dgrad(g, prec, dy, packed_w_dgrad, dx, accumulate, packed_elems(g->cin, g->cout, 4, prec), (hipStream_t)s, &rc))
      return rc;
  }
  const uint16_t* wp = (const uint16_t*)packed_w_dgrad;
  const int st = g->stride;
  if (dgrad_pair_ok(g)) {   // (conv_host.h: one launch per row class, rows = (channel, column class))
    for (int ry = 0; ry < 2; ++ry) {
      TapSet t = dgrad_pair_taps(g, ry);
      const int lh = (g->in_h - ry + 1) / 2, lw = (g->in_w + 1) / 2;
      if (lh > 0) {
        IgemmParams p;
        memset(&p, 0, sizeof(p));
        p.x = *dy; p.cin = g->cout;
        p.in_h = g->out_h; p.in_w = g->out_w; p.in_shift = 0; p.in_row = g->out_w;
        p.y = *dx; p.cout = 2 * g->cin; p.out_w = g->in_w;
        p.lh = lh; p.lw = lw; p.pair = 1; p.lw2 = g->in_w / 2;
        p.oy_mul = p.ox_mul = 2; p.oy_off = ry; p.ox_off = 0;
        p.in_step = 1;
        p.wpack = wp; p.w_lo_off = 0;
        p.bias = nullptr; p.slope = 1.f; p.accumulate = accumulate; p.stats = nullptr;
        p.mask_a = mask_a; p.mask_sn = mask_sn; p.mask_slope = mask_slope;
        p.n = g->n;
        int rc = launch_igemm(p, prec, t, (hipStream_t)s);
        if (rc) return rc;
      }
      wp += packed_elems(2 * g->cin, g->cout, t.n, prec);
    }
    return PCUDA_OK;
  }
  for (int ry = 0; ry < st; ++ry)
    for (int rx = 0; rx < st; ++rx) {
      TapSet t = dgrad_taps(g, ry, rx);
      const size_t plane = packed_elems(g->cin, g->cout, t.n, prec);
      const int lh = (g->in_h - ry + st - 1) / st, lw = (g->in_w - rx + st - 1) / st;
      if (lh > 0 && lw > 0) {
        IgemmParams p;
        memset(&p, 0, sizeof(p));
        p.x = *dy; p.cin = g->cout;
        p.in_h = g->out_h; p.in_w = g->out_w; p.in_shift = 0; p.in_row = g->out_w;
        p.y = *dx; p.cout = g->cin; p.out_w = g->in_w;
        p.lh = lh; p.lw = lw;
        p.oy_mul = p.ox_mul = st; p.oy_off = ry; p.ox_off = rx;
        p.in_step = 1;
        p.wpack = wp; p.w_lo_off = 0;
        p.bias = nullptr; p.slope = 1.f; p.accumulate = accumulate; p.stats = nullptr;
        p.mask_a = mask_a; p.mask_sn = mask_sn; p.mask_slope = mask_slope;
        p.n = g->n;
        const unsigned char* ap_image = (!mask_a && ap_layer_ok(g, g->cin, g->cout, prec)) ? (const unsigned char*)(wp + plane) : nullptr;
        int rc = launch_igemm(p, prec, t, (hipStream_t)s, ap_image);
        if (rc) return rc;
      }
      wp += plane;
    }
  return PCUDA_OK;
}

extern "C" int pcuda_conv2d_dgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                                  const pcuda_dst* dx, int accumulate, pcuda_stream_t s) {
  return dgrad_impl(g, prec, dy, packed_w_dgrad, dx, accumulate, nullptr, 0, 1.f, s);
}

// dx = dgrad(dy) * (a > 0 ? 1 : slope): the LeakyReLU backward of the layer in FRONT of this convolution (GAN.py:97-108:
// conv -> LeakyReLU(0.2) -> conv, going back) in the epilogue of the data-gradient kernel -- the gradient with respect to
// the activation is never stored and read back by pcuda_lrelu_bwd.  a: the saved activation [n][cin][in_h][in_w] with the
// plane stride of dx (one destination).  PCUDA_E_UNSUPPORTED where a launch of the layer would not run on a plain-epilogue
// kernel (the caller then runs pcuda_conv2d_dgrad + pcuda_lrelu_bwd; a partly written dx is overwritten).
extern "C" int pcuda_conv2d_dgrad_lrelu(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                                        const pcuda_dst* dx, const float* a, long long a_sn, long long a_sc, float slope,
                                        pcuda_stream_t s) {
  if (!a || !dx) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad_lrelu: bad tensors");
  if (dx->c1 < g->cin || a_sc != dx->sc1 || g->in_up)
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_dgrad_lrelu: one NCHW destination with the activation's plane stride");
  // (a data gradient the anti-phase kernel takes has no masked store: the caller runs that kernel + pcuda_lrelu_bwd)
  if (ap_layer_ok(g, g->cin, g->cout, prec) && ap_map_ok(g->n, g->in_h, g->in_w, g->cin))
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_dgrad_lrelu: this layer's data gradient runs on the anti-phase kernel");
  return dgrad_impl(g, prec, dy, packed_w_dgrad, dx, 0, a, a_sn, slope, s);
}

// timing experiments (PCUDA_DBG bit 128): per-phase cycle sums of igemm_pipe_kernel, read and reset
extern "C" int pcuda_debug_read_clocks(unsigned long long* out8) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (!g_dbg_clk_host) { memcpy(out8, z, sizeof(z)); return 0; }
  if (hipMemcpy(out8, g_dbg_clk_host, sizeof(z), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (hipMemcpy(g_dbg_clk_host, z, sizeof(z), hipMemcpyHostToDevice) != hipSuccess) return -1;
  return 0;
}

// ------------------------------------------------------------------------------------------
// dgrad with the BatchNorm-backward reduce of the layer in FRONT of this convolution fused into its epilogue
// (unet.py:23-30: conv -> LeakyReLU -> BN -> conv: the second convolution's data gradient IS the first BatchNorm's
// incoming gradient, and nothing else adds to it).  Stride-1 layers on the transposed-epilogue kernel only;
// PCUDA_E_UNSUPPORTED otherwise (the caller then runs pcuda_conv2d_dgrad + pcuda_bn_bwd_reduce).
// ------------------------------------------------------------------------------------------
extern "C" int pcuda_conv2d_dgrad_tiles(const pcuda_conv_geom* g, int prec) {
  if (!geom_ok(g) || g->stride != 1) return 0;
  if (const int d = direct_dgrad_tiles(g)) return d;
  if (ap_layer_ok(g, g->cin, g->cout, prec) && ap_map_ok(g->n, g->in_h, g->in_w, g->cin)) return ap_tiles(g->n, g->in_h, g->in_w);
  if (rs_layer_ok(g, g->cin, g->cout, prec) && rs_map_ok(g->n, g->in_h, g->in_w)) return rs_tiles(g->n, g->in_h, g->in_w);
  TapSet t = dgrad_taps(g, 0, 0);
  IgemmPlan pl;
  if (plan_igemm(g->cin, g->cout, g->n, g->in_h, g->in_w, g->out_h, g->out_w, 1, t, prec == PCUDA_PREC_BF16X3, &pl) < 0) return 0;
  return g->n * pl.tiles_x * pl.tiles_y;
}

extern "C" int pcuda_conv2d_dgrad_bnred(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                                        const pcuda_dst* dx, int accumulate, const float* a, long long a_sn, long long a_sc,
                                        const float* mean, const float* invstd, float* red_partials, pcuda_stream_t s) {
  if (!geom_ok(g)) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad_bnred: inconsistent geometry");
  if (!src_ok(dy, g->cout) || !dst_ok(dx, g->cin) || !packed_w_dgrad || !a || !mean || !invstd || !red_partials)
    PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad_bnred: bad tensors");
  if (prec != PCUDA_PREC_BF16X3 && prec != PCUDA_PREC_BF16) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad_bnred: bad precision");
  if (g->stride != 1 || g->in_up || (((uintptr_t)a) & 15) || (a_sn & 3) || (a_sc & 3))
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_dgrad_bnred: stride-1 layers with 16-byte aligned activations only");
  if (direct_dgrad_tiles(g)) {   // (the tile count the caller sized red_partials by is the direct kernel's)
    int rc;
    if (direct_dgrad(g, prec, dy, packed_w_dgrad, dx, accumulate, (hipStream_t)s, &rc, a, a_sn, a_sc, mean, invstd, red_partials))
      return rc;
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_dgrad_bnred: 1x1 layer with tensors the direct kernel does not take");
  }
  TapSet t = dgrad_taps(g, 0, 0);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = *dy; p.cin = g->cout;
  p.in_h = g->out_h; p.in_w = g->out_w; p.in_shift = 0; p.in_row = g->out_w;
  p.y = *dx; p.cout = g->cin; p.out_w = g->in_w;
  p.lh = g->in_h; p.lw = g->in_w;
  p.oy_mul = p.ox_mul = 1; p.oy_off = p.ox_off = 0;
  p.in_step = 1;
  p.wpack = (const uint16_t*)packed_w_dgrad; p.w_lo_off = 0;
  p.bias = nullptr; p.slope = 1.f; p.accumulate = accumulate;
  p.stats = red_partials;
  p.red_a = a; p.red_sn = a_sn; p.red_sc = a_sc; p.red_mean = mean; p.red_invstd = invstd;
  p.n = g->n;
  const unsigned char* ap_image = ap_layer_ok(g, g->cin, g->cout, prec)
                                      ? (const unsigned char*)packed_w_dgrad + packed_elems(g->cin, g->cout, t.n, prec) * 2 : nullptr;
  return launch_igemm(p, prec, t, (hipStream_t)s, ap_image);
}

// ------------------------------------------------------------------------------------------
// dgrad of a layer whose input was read through the nearest-x2 fold (g->in_up), written at the STORED (half) resolution:
// the 2x2 sum that upsample2_bwd_kernel applied to the logical gradient happens in the epilogue.  a != NULL: the
// BatchNorm-backward reduce of the layer that produced the stored tensor rides along (red_partials[tiles][cin][2], tiles =
// pcuda_conv2d_dgrad_tiles).  PCUDA_E_UNSUPPORTED where the plan is not the 32 x 8-tile transposed-epilogue one (the caller
// then runs pcuda_conv2d_dgrad + pcuda_upsample2_bwd).
// ------------------------------------------------------------------------------------------
extern "C" int pcuda_conv2d_dgrad_fold(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                                       const pcuda_dst* dx_half, const float* a, long long a_sn, long long a_sc,
                                       const float* mean, const float* invstd, float* red_partials, pcuda_stream_t s) {
  if (!geom_ok(g)) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad_fold: inconsistent geometry");
  if (!src_ok(dy, g->cout) || !dst_ok(dx_half, g->cin) || !packed_w_dgrad) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad_fold: bad tensors");
  if (prec != PCUDA_PREC_BF16X3 && prec != PCUDA_PREC_BF16) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad_fold: bad precision");
  if (a && (!mean || !invstd || !red_partials)) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad_fold: reduce operands missing");
  if (!g->in_up || g->stride != 1 || (g->in_h & 1) || (g->in_w & 3) || direct_dgrad_tiles(g) ||
      (a && ((((uintptr_t)a) & 7) || (a_sn & 1) || (a_sc & 1))))
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_dgrad_fold: stride-1 MFMA layers behind a nearest-x2 fold, rows of 4k logical pixels");
  TapSet t = dgrad_taps(g, 0, 0);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = *dy; p.cin = g->cout;
  p.in_h = g->out_h; p.in_w = g->out_w; p.in_shift = 0; p.in_row = g->out_w;
  p.y = *dx_half; p.cout = g->cin; p.out_w = g->in_w >> 1;
  p.lh = g->in_h; p.lw = g->in_w;
  p.oy_mul = p.ox_mul = 1; p.oy_off = p.ox_off = 0;
  p.in_step = 1;
  p.wpack = (const uint16_t*)packed_w_dgrad; p.w_lo_off = 0;
  p.bias = nullptr; p.slope = 1.f; p.accumulate = 0;
  p.fold = 1;
  if (a) {
    p.stats = red_partials;
    p.red_a = a; p.red_sn = a_sn; p.red_sc = a_sc; p.red_mean = mean; p.red_invstd = invstd;
  }
  p.n = g->n;
  const unsigned char* ap_image = ap_layer_ok(g, g->cin, g->cout, prec)
                                      ? (const unsigned char*)packed_w_dgrad + packed_elems(g->cin, g->cout, t.n, prec) * 2 : nullptr;
  return launch_igemm(p, prec, t, (hipStream_t)s, ap_image);
}

// ------------------------------------------------------------------------------------------
// batched repack: the caller collects the jobs of all its layers once (host side), keeps them in device memory and
// replays them with one launch after every optimiser step
// ------------------------------------------------------------------------------------------
extern "C" size_t pcuda_conv2d_pack_job_bytes(void) { return sizeof(PackParams); }

extern "C" int pcuda_conv2d_pack_jobs_fill(const pcuda_conv_geom* g, int prec, const float* w, void* packed_fwd,
                                           void* packed_dgrad, void* host_jobs, int max_jobs, int* job_blocks) {
  if (!geom_ok(g) || !w || !packed_fwd || !host_jobs || max_jobs < 1 || !job_blocks)
    PCUDA_FAIL(PCUDA_E_BADARG, "pack_jobs_fill: bad arguments");
  if (prec != PCUDA_PREC_BF16X3 && prec != PCUDA_PREC_BF16) PCUDA_FAIL(PCUDA_E_BADARG, "pack_jobs_fill: bad precision");
  PackParams* jobs = (PackParams*)host_jobs;
  const int kk = g->k * g->k;
  int nj = 0;
  size_t plane = fill_pack(jobs[nj], w, (uint16_t*)packed_fwd, prec, g->cout, g->cin, (long long)g->cin * kk, kk, fwd_taps(g));
  job_blocks[nj] = pack_table_blocks(jobs[nj]); ++nj;
  if (ap_layer_ok(g, g->cout, g->cin, prec)) {   // the anti-phase image behind the forward layout
    if (nj >= max_jobs) PCUDA_FAIL(PCUDA_E_BADARG, "pack_jobs_fill: more layouts than job slots");
    if (!ap_fill_pack(jobs[nj], w, (unsigned char*)packed_fwd + plane * 2, g->cout, g->cin, (long long)g->cin * kk, kk, fwd_taps(g)))
      PCUDA_FAIL(PCUDA_E_BADARG, "pack_jobs_fill: anti-phase image of a layer that is not 3x3");
    job_blocks[nj] = pack_table_blocks(jobs[nj]); ++nj;
  }
  if (packed_dgrad && dgrad_pair_ok(g)) {
    uint16_t* out = (uint16_t*)packed_dgrad;
    for (int ry = 0; ry < 2; ++ry) {
      if (nj >= max_jobs) PCUDA_FAIL(PCUDA_E_BADARG, "pack_jobs_fill: more layouts than job slots");
      plane = fill_pack(jobs[nj], w, out, prec, 2 * g->cin, g->cout, kk, (long long)g->cin * kk, dgrad_pair_taps(g, ry), true);
      job_blocks[nj] = pack_table_blocks(jobs[nj]); ++nj;
      out += plane;
    }
  } else if (packed_dgrad) {
    uint16_t* out = (uint16_t*)packed_dgrad;
    for (int ry = 0; ry < g->stride; ++ry)
      for (int rx = 0; rx < g->stride; ++rx) {
        TapSet t = dgrad_taps(g, ry, rx);
        if (t.n == 0) continue;
        if (nj >= max_jobs) PCUDA_FAIL(PCUDA_E_BADARG, "pack_jobs_fill: more layouts than job slots");
        plane = fill_pack(jobs[nj], w, out, prec, g->cin, g->cout, kk, (long long)g->cin * kk, t);
        job_blocks[nj] = pack_table_blocks(jobs[nj]); ++nj;
        out += plane;
      }
    if (ap_layer_ok(g, g->cin, g->cout, prec)) {   // the anti-phase image behind the (single) data-gradient layout
      if (nj >= max_jobs) PCUDA_FAIL(PCUDA_E_BADARG, "pack_jobs_fill: more layouts than job slots");
      if (!ap_fill_pack(jobs[nj], w, (unsigned char*)out, g->cin, g->cout, kk, (long long)g->cin * kk, dgrad_taps(g, 0, 0)))
        PCUDA_FAIL(PCUDA_E_BADARG, "pack_jobs_fill: anti-phase image of a layer that is not 3x3");
      job_blocks[nj] = pack_table_blocks(jobs[nj]); ++nj;
    }
  }
  return nj;
}

extern "C" int pcuda_conv2d_pack_table(const void* dev_jobs, const int* dev_first_block, int njobs, int total_blocks,
                                       pcuda_stream_t s) {
  if (!dev_jobs || !dev_first_block || njobs < 1 || total_blocks < 1) PCUDA_FAIL(PCUDA_E_BADARG, "pack_table: bad arguments");
  hipLaunchKernelGGL(pack_table_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)s, (const PackParams*)dev_jobs,
                     dev_first_block, njobs);
  PCUDA_CHECK_LAUNCH("pack_table_kernel");
  return PCUDA_OK;
}
