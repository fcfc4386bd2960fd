// Host side of the anti-phase 3x3 kernel (conv_ap_impl.h): which launches it takes, its packed weight image, its launcher.
//
// It takes the forward and data-gradient launches of the segmenter's aligned 3x3 / stride-1 / pad-1 layers with >= 64 rows
// (unet.py:23,27,116,122 at the 128x128 ... 32x32 levels, the 64-row data gradients of the 256x256 level): the launches
// that igemm_pipe_kernel runs on 64-row tiles.  Everything else (32-row tiles, the nearest-x2 fold on either side,
// dilation, the discriminators' stride-2 layers, maps that are not whole 32 x 8 tiles) stays where it was.
#include <stdlib.h>
#include "conv_ap_impl.h"
#include "conv_host.h"
#include "variants.h"

static int ap_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("PCUDA_AP"); on = (e && !atoi(e)) ? 0 : 1; }
  return on;
}

// A property of the LAYER (kernel size, stride, padding, dilation, channel counts): its packed buffer carries the image
// whatever maps the layer later runs on.  rows / red: output rows and reduction channels of the launch (forward: cout, cin;
// data gradient: cin, cout).
bool ap_layer_ok(const pcuda_conv_geom* g, int rows, int red, int prec) {
  // (in_up layers, unet.py:111: the forward reads the stored half-resolution input through the fold, the data gradient folds
  //  2x2 blocks in its epilogue)
  return ap_enabled() && prec == PCUDA_PREC_BF16X3 && g->k == 3 && g->stride == 1 && g->pad == 1 && g->dil == 1 &&
         rows >= 64 && (rows & 63) == 0 && red >= 16 && (red & 15) == 0 && rows <= AP_MAX_C && red <= AP_MAX_C;
}
size_t ap_layer_packed_bytes(int rows, int red) { return ap_packed_bytes(rows, red); }

// the tap of the anti-phase image at (dy, dx) = (t / 3 - 1, t % 3 - 1) is the launch's tap with that offset
static bool ap_tap_order(const TapSet& taps, signed char* src9) {
  if (taps.n != 9) return false;
  for (int t = 0; t < 9; ++t) {
    int hit = -1;
    for (int i = 0; i < 9; ++i)
      if (taps.dy[i] == t / 3 - 1 && taps.dx[i] == t % 3 - 1) hit = i;
    if (hit < 0) return false;
    src9[t] = taps.src[hit];
  }
  return true;
}

// job record of the batched repack (pack_table_kernel, conv_igemm.hip): rec == 32 marks the anti-phase image
bool ap_fill_pack(PackParams& p, const float* w, unsigned char* out, int rows, int red, long long s_row, long long s_red,
                  const TapSet& taps) {
  memset(&p, 0, sizeof(p));
  signed char src9[9];
  if (!ap_tap_order(taps, src9)) return false;
  p.w = w; p.out = (uint16_t*)out;
  p.rec = 32;
  p.rows = rows; p.red = red; p.s_row = s_row; p.s_red = s_red;
  p.ntaps = 9;
  memcpy(p.tap_src, src9, 9);
  p.pair = 0;
  p.co_tile = 64;
  p.nchunks = (red + 31) / 32;
  p.n_co_tiles = rows / 64;
  return true;
}

// one-off repack of a layer (pcuda_conv2d_pack_fwd / _dgrad / _all): its own small launch behind the other layouts
int ap_launch_pack(const float* w, unsigned char* out, int rows, int red, long long s_row, long long s_red, const TapSet& taps,
                   hipStream_t s) {
  signed char src9[9];
  if (!ap_tap_order(taps, src9)) PCUDA_FAIL(PCUDA_E_BADARG, "anti-phase pack: not a 3x3 tap set");
  bool flip = true, plain = true;
  for (int t = 0; t < 9; ++t) { flip &= src9[t] == 8 - t; plain &= src9[t] == t; }
  if (!flip && !plain) PCUDA_FAIL(PCUDA_E_BADARG, "anti-phase pack: tap order is neither the forward nor the mirrored one");
  ApPackParams pp;
  pp.w = w; pp.s_row = s_row; pp.s_red = s_red; pp.rows = rows; pp.red = red; pp.flip = flip ? 1 : 0; pp.out = out;
  const long long total = (long long)(rows >> 6) * (red >> 4) * 9 * 64 * 4;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(ap_pack_kernel, dim3(blocks), dim3(256), 0, s, pp);
  PCUDA_CHECK_LAUNCH("ap_pack_kernel");
  return PCUDA_OK;
}

// Maps the kernel takes: whole 32 x 8-pixel tiles, an even number of them, inside the cursor's bit fields.  Together with
// ap_layer_ok this decides the TILE COUNT the caller sizes its BatchNorm partial sums by (pcuda_conv2d_fwd_tiles /
// _dgrad_tiles): such a launch either runs here or -- tensors the kernel cannot address -- on the ordinary kernel ONLY IF
// that kernel's plan has the same tiles (launch_igemm checks; otherwise the call fails rather than overrun the partials).
// (+ enough work items -- (tile pair, co tile) -- for the chip: below three quarters of the compute units, the 32x32 maps at a
//  per-rank batch of 16, the ordinary kernels' 128-pixel tiles fill it better than one 2 x 256-pixel item per workgroup on
//  half of it.  PCUDA_AP_MIN_ITEMS overrides the threshold: the kernel-level tests run small maps on the kernel.)
static int ap_min_items() {
  const char* e = getenv("PCUDA_AP_MIN_ITEMS");      // (read per call: tests flip it inside one process)
  if (e) return atoi(e);
  static int cus = 0;
  if (!cus) {
    int d = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&prop, d) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return cus * 3 / 4;
}
bool ap_map_ok(int n, int h, int w, int rows) {
  if (!(ap_enabled() && h >= 8 && (h & 7) == 0 && w >= 32 && (w & 31) == 0 && n < 4096 && (h / 8) < 64 && (w / 32) < 16 &&
        (((long long)n * (h / 8) * (w / 32)) & 1) == 0))
    return false;
  return (long long)n * (h / 8) * (w / 32) / 2 * (rows / 64) >= ap_min_items();
}
int ap_tiles(int n, int h, int w) { return n * (h / 8) * (w / 32); }

// does the kernel take this launch?  (geometry of the maps and the tensors' alignment; the layer property was settled at
// pack time: the caller passes the image)
static bool ap_launch_ok(const IgemmParams& p, const TapSet& taps) {
  signed char src9[9];
  if (!ap_enabled() || !ap_tap_order(taps, src9)) return false;
  if (p.in_step != 1 || p.pair || p.mask_a || p.oy_mul != 1 || p.ox_mul != 1 || p.oy_off || p.ox_off) return false;
  if (p.in_shift && (p.fold || p.in_row != (p.in_w >> 1))) return false;            // forward through the nearest-x2 fold
  if (!p.in_shift && p.in_row != p.in_w) return false;
  if (p.lh != p.in_h || p.lw != p.in_w || p.out_w != (p.fold ? p.in_w >> 1 : p.in_w)) return false;
  if (p.fold && (p.accumulate || p.bias || (p.stats && !p.red_a))) return false;      // data gradient with the 2x2 fold
  if (!ap_geom_ok(p.cout, p.cin, p.in_h, p.in_w) || !ap_map_ok(p.n, p.in_h, p.in_w, p.cout)) return false;
  if (!(p.slope >= 0.f && p.slope <= 1.f)) return false;
  // input: a 16-channel chunk lies in one source, rows of float4, 32-bit element offsets inside an image
  const pcuda_src& x = p.x;
  const int c1 = x.c1 < p.cin ? x.c1 : p.cin;
  if (c1 < p.cin && (c1 & 15)) return false;
  if (((uintptr_t)x.p1 & 15) || (x.sn1 & 3) || (x.sc1 & 3) || x.sc1 >= (1ll << 26)) return false;
  if (c1 < p.cin && (((uintptr_t)x.p2 & 15) || (x.sn2 & 3) || (x.sc2 & 3) || x.sc2 >= (1ll << 26))) return false;
  // output: 16-byte row segments, four consecutive rows in one destination (fold: 8-byte pairs, eight rows)
  const int yc1 = p.y.c1 < p.cout ? p.y.c1 : p.cout;
  if (p.fold) {
    if (((uintptr_t)p.y.p1 & 7) || (p.y.sc1 & 1) || (p.y.sn1 & 1) || (p.out_w & 1)) return false;
    if (yc1 < p.cout && (((uintptr_t)p.y.p2 & 7) || (p.y.sc2 & 1) || (p.y.sn2 & 1) || (yc1 & 7))) return false;
    if (p.red_a && (((uintptr_t)p.red_a & 7) || (p.red_sn & 1) || (p.red_sc & 1))) return false;
  } else {
    if (!te_dst_ok(&p.y, p.cout, p.out_w, p.lw, 32, 1, 0)) return false;
    if (p.red_a && (((uintptr_t)p.red_a & 15) || (p.red_sn & 3) || (p.red_sc & 3))) return false;
  }
  if (yc1 < p.cout && (yc1 & 3)) return false;
  if (p.y.sc1 >= (1ll << 26) || (yc1 < p.cout && p.y.sc2 >= (1ll << 26))) return false;
  if (p.red_a && !p.stats) return false;
  return true;
}

template <int STATS, bool ACC, bool FOLD = false>
static int ap_launch_t(const ApParams& ap, int grid, hipStream_t s) {
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    if (hipFuncSetAttribute((const void*)conv3ap_kernel<STATS, ACC, false, false, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, AP_LDS_BYTES) != hipSuccess)
      PCUDA_FAIL(PCUDA_E_LAUNCH, "conv3ap_kernel: cannot opt in to %d bytes of LDS", AP_LDS_BYTES);
    once.mark(bit);
  }
  hipLaunchKernelGGL((conv3ap_kernel<STATS, ACC, false, false, FOLD>), dim3(grid), dim3(512), AP_LDS_BYTES, s, ap);
  PCUDA_CHECK_LAUNCH("conv3ap_kernel");
  return PCUDA_OK;
}

// returns 1 when the launch was taken (*rc = its status), 0 when it is not this kernel's
int ap_try_launch(const IgemmParams& p, const TapSet& taps, const unsigned char* image, hipStream_t s, int* rc) {
  if (!image || !ap_launch_ok(p, taps)) return 0;
  ApParams ap;
  memset(&ap, 0, sizeof(ap));
  ap.x = p.x; ap.cin = p.cin; ap.H = p.in_h; ap.W = p.in_w;
  ap.y = p.y; ap.cout = p.cout;
  ap.wimg = image;
  ap.bias = p.bias; ap.slope = p.slope; ap.accumulate = p.accumulate;
  ap.stats = p.stats;
  ap.red_a = p.red_a; ap.red_sn = p.red_sn; ap.red_sc = p.red_sc; ap.red_mean = p.red_mean; ap.red_invstd = p.red_invstd;
  ap.tiles_x = p.in_w / 32; ap.tiles_y = p.in_h / 8; ap.n = p.n;
  ap.n_co_tiles = p.cout / 64; ap.nchunks = p.cin / 16;
  ap.total = (p.n * ap.tiles_x * ap.tiles_y / 2) * ap.n_co_tiles;
  ap.up = p.in_shift ? 1 : 0; ap.out_w = p.out_w;
  static int cus = 0;
  if (!cus) {
    int d = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&prop, d) != hipSuccess) { *rc = PCUDA_E_LAUNCH; return 1; }
    cus = prop.multiProcessorCount;
  }
  const int grid = ap.total < cus ? ap.total : cus;
  const double flops = 2.0 * p.n * (double)p.in_h * p.in_w * p.cout * (double)p.cin * 9;
  char tag[160];
  snprintf(tag, sizeof(tag), "conv3ap n%d red%d rows%d %dx%d taps9 up%d fold%d stats%d acc%d items%d", p.n, p.cin, p.cout, p.in_h, p.in_w,
           ap.up, p.fold ? 1 : 0, p.red_a ? 2 : (p.stats ? 1 : 0), p.accumulate ? 1 : 0, ap.total);
  ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
  const int st = p.red_a ? 2 : (p.stats ? 1 : 0);
  if (p.fold) *rc = st == 2 ? ap_launch_t<2, false, true>(ap, grid, s) : ap_launch_t<0, false, true>(ap, grid, s);
  else if (st == 1 && !p.accumulate) *rc = ap_launch_t<1, false>(ap, grid, s);
  else if (st == 1) *rc = ap_launch_t<1, true>(ap, grid, s);
  else if (st == 2 && !p.accumulate) *rc = ap_launch_t<2, false>(ap, grid, s);
  else if (st == 2) *rc = ap_launch_t<2, true>(ap, grid, s);
  else if (!p.accumulate) *rc = ap_launch_t<0, false>(ap, grid, s);
  else *rc = ap_launch_t<0, true>(ap, grid, s);
  note_kernel(p.fold ? "conv3ap+fold" : (st == 2 ? "conv3ap+bnred" : "conv3ap"));
  return 1;
}
