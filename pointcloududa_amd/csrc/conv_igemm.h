// Parameter blocks of the implicit-GEMM convolution kernels (host + device).
#pragma once
#include "common.h"

#define IG_MAX_TAPS 36
#define IG_REC 40          // bf16 per LDS record: 32 channels + 8 pad (80 B: conflict-free b128 reads)
#define IG_REC_BYTES 80
// Forward / dgrad kernels in bf16x3 mode keep BOTH planes of a pixel (or weight row) in ONE 144-byte record:
// hi 64 B | lo 64 B | 16 pad.  36-dword stride: ds_read_b128 over consecutive records is still conflict-free, and the
// haloed input tile plus a tap of weights take 10 % less LDS than two padded 80-byte planes -- which is what lets a
// 64-row tile hold three taps per weight group (three groups per stage instead of five) next to a second workgroup.
// The packed weight image in global memory has the same record (the LDS copy stays a linear 16-byte copy).
template <bool X3>
struct IgRec {
  static constexpr int BYTES = X3 ? 144 : 80;
};
#define IG_LO_OFF 64
static inline int ig_rec_bytes(bool x3) { return x3 ? 144 : 80; }

// One launch of the generic kernel computes, for every logical output pixel (oy,ox) and row r:
//   D[r][oy,ox] = sum_{tap t, channel c} Wp[t][r][c] * X[c][oy*in_step + dy[t]][ox*in_step + dx[t]]
// and stores act(D + bias) at physical (oy*oy_mul + oy_off, ox*ox_mul + ox_off).
// forward:  r = cout, c = cin, in_step = stride, d = k*dil - pad
// dgrad:    r = cin,  c = cout, in_step = 1, one launch per stride-parity class
struct IgemmParams {
  pcuda_src x;
  int cin;                 // reduction channels
  int in_h, in_w;          // logical input plane (bounds for zero padding)
  int in_shift;            // 1: stored plane is (in_h>>1, in_w>>1), read through nearest x2
  int in_row;              // stored row length
  pcuda_dst y;
  int cout;                // rows
  int out_w;               // physical row length of the output planes
  int lh, lw;              // logical output grid
  int oy_mul, oy_off, ox_mul, ox_off;
  int in_step;
  int ntaps, tg;           // taps; taps staged per weight group
  signed char dy[IG_MAX_TAPS], dx[IG_MAX_TAPS];
  int dy_min, dx_min, ih_t, iw_t;   // unclamped LDS tile: origin offset and dims
  const uint16_t* wpack;   // [co_tile][chunk][tap][CO_TILE][IG_REC] bf16 bits; lo plane at +w_lo_off
  long long w_lo_off;
  int nchunks;
  const float* bias;
  float slope;
  int accumulate;
  float* stats;            // [tile][cout][2] partial (sum, sumsq) of the stored values, or NULL
  // BatchNorm-backward reduce fused into a dgrad launch (transposed epilogue only): with red_a set, `stats` receives
  // per tile and row (sum of the stored gradient g, sum of g * (a - mean) * invstd) -- what bn_bwd_reduce computes
  const float* red_a; long long red_sn, red_sc;
  const float* red_mean; const float* red_invstd;
  // LeakyReLU backward of the layer in FRONT fused into a dgrad launch (plain epilogues only): the stored gradient is
  // multiplied by (a > 0 ? 1 : mask_slope), a = the saved activation, laid out like the (single) destination: plane
  // stride y.sc1, image stride mask_sn.  No bias / accumulate / stats with it.
  const float* mask_a; long long mask_sn; float mask_slope;
  int tw, th, tmagic;      // output tile TW x TH (TW*TH <= 128*NPB slots, any TW <= 256); tmagic = 65536/TW + 1
  int tiles_x, tiles_y, n;
  int n_co_tiles;
  // exact division of a tile index by n_co_tiles / tiles_x / tiles_y as one multiply-high (persistent kernels decode a tile
  // per stage: four runtime divisions were ~120 of a stage's ~660 scalar instructions): q = m ? umulhi(v, m) : v with
  // m = floor(2^32 / d) + 1 (0 for d = 1), exact for v < 2^32 / d (the host checks the launch's tile count)
  unsigned m_cot, m_tx, m_ty;
  int clamp;               // 1: LDS tile = tile clipped to the image (+ one zero record)
  int dbg;                 // PCUDA_DBG bits (timing experiments only): 1 no X loads, 2 no MFMA, 4 no epilogue, 8 no W copy
  unsigned long long* dbg_clk;   // PCUDA_DBG bit 128: 8 per-phase cycle sums
  int xq;                  // 1: quad (float4) input staging (in_w % 4 == 0, no upsampling fold)
  int fold;                // 1: the epilogue sums 2x2 blocks of the logical output (the data gradient of a nearest-x2-folded
                           // input, unet.py:111): y is the half-resolution tensor, out_w its row length
  // paired column classes of a stride-2 data gradient (dword-store epilogues): row 2c + rx of the launch is channel c,
  // column parity rx -- one wave stores both halves of every 8 bytes of a destination line back to back.  lw = columns
  // of the even class, lw2 = of the odd one; ox_off = 0, ox_mul = 2.
  int pair, lw2;
};

// wgrad: dW[r][c][tap] = sum_{n,oy,ox} dZ[r][oy,ox] * X[c][oy*stride + dy[t]][ox*stride + dx[t]]
struct WgradParams {
  pcuda_src x;
  int cin;
  int in_h, in_w, in_shift, in_row;
  const float* dz; long long dz_sn, dz_sc;
  int cout, out_h, out_w;
  int stride;
  int ntaps;               // taps handled per block-group (<= 9)
  int ntaps_total, tap_groups;
  signed char dy[IG_MAX_TAPS], dx[IG_MAX_TAPS];
  int ih_t, iw_t;          // LDS tile dims per tap group are computed from the FULL tap span
  int dy_min, dx_min;
  int tw, th, tmagic;      // output tile TW x TH <= 128 slots
  int tiles_x, tiles_y, n;
  int ksplit;              // gridDim.y
  int n_co_tiles, n_chunks;
  float* partial;          // [ksplit][cout][cin][ntaps_total] fp32
  int aligned4;            // dz rows can be read with float4
  int tw16;                // TW % 16 == 0 and TW*TH == 128: a k-step's 16 pixel slots share one tile row
  int xq;                  // 1: quad (float4) staging of the input tile (in_w % 4 == 0, no upsampling fold)
  int dbg;                 // PCUDA_DBG bits (timing experiments only): 16 no X staging, 32 no dZ staging, 64 no MFMA
  int xcd_items;           // > 0: XCD-aware block mapping, work items per XCD (= blocks / 8)
  int xcd_slices;          // > 0: split-K slices per XCD (ksplit / 8), tiles of an XCD's eighth interleaved over them
};

#define WG_ZROW 272   // wgrad: bytes per dZ row in LDS, 128 px bf16 + 16 pad (17*16: conflict-free b128)

struct PackParams {
  const float* w;
  uint16_t* out;
  int rec;                 // bf16 elements per record: 40 (bf16 mode) or 72 (bf16x3: hi | lo | pad)
  int rows, red;           // rows (M) and reduction (K) extents
  long long s_row, s_red;  // element strides in w for row / reduction index (tap stride is 1)
  int ntaps;
  signed char tap_src[IG_MAX_TAPS];
  int pair;                // 1: row r reads source row r >> 1 through tap_src[(r & 1) * ntaps + t] (paired column classes)
  int co_tile;             // 32 or 64
  int nchunks, n_co_tiles;
};

static inline int ig_co_blks(int rows) { return rows > 32 ? 2 : 1; }
// pixel slot -> (row, col) of a TW-wide tile without an integer division (exact for slot < 256, TW <= 256)
#define IG_TY(pl, magic) (((pl) * (magic)) >> 16)
