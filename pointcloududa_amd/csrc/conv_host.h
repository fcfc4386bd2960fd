// Host-side geometry helpers shared by conv_igemm.hip and conv_wgrad.hip.
#pragma once
#include <stdlib.h>
#include <string.h>
#include "conv_igemm.h"

static const int LDS_HARD = 160 * 1024;

static inline bool geom_ok(const pcuda_conv_geom* g) {
  if (!g || g->n <= 0 || g->cin <= 0 || g->cout <= 0 || g->k <= 0 || g->stride <= 0 || g->dil <= 0 || g->pad < 0)
    return false;
  if (g->k * g->k > IG_MAX_TAPS) return false;
  const int oh = (g->in_h + 2 * g->pad - g->dil * (g->k - 1) - 1) / g->stride + 1;
  const int ow = (g->in_w + 2 * g->pad - g->dil * (g->k - 1) - 1) / g->stride + 1;
  if (oh != g->out_h || ow != g->out_w || oh <= 0 || ow <= 0) return false;
  if (g->in_up && ((g->in_h & 1) || (g->in_w & 1))) return false;
  return true;
}

static inline bool src_ok(const pcuda_src* s, int c) {
  if (!s || !s->p1 || s->c1 <= 0 || s->c1 > c) return false;
  if (s->c1 < c && !s->p2) return false;
  return true;
}
static inline bool dst_ok(const pcuda_dst* d, int c) {
  if (!d || !d->p1 || d->c1 <= 0 || d->c1 > c) return false;
  if (d->c1 < c && !d->p2) return false;
  return true;
}

struct TapSet {
  int n;
  signed char dy[IG_MAX_TAPS], dx[IG_MAX_TAPS], src[IG_MAX_TAPS];
  int dy_min, dy_max, dx_min, dx_max;
  void finish() {
    dy_min = dx_min = 127; dy_max = dx_max = -127;
    for (int i = 0; i < n; ++i) {
      if (dy[i] < dy_min) dy_min = dy[i];
      if (dy[i] > dy_max) dy_max = dy[i];
      if (dx[i] < dx_min) dx_min = dx[i];
      if (dx[i] > dx_max) dx_max = dx[i];
    }
    if (n == 0) dy_min = dy_max = dx_min = dx_max = 0;
  }
};

static inline TapSet fwd_taps(const pcuda_conv_geom* g) {
  TapSet t; t.n = 0;
  for (int ky = 0; ky < g->k; ++ky)
    for (int kx = 0; kx < g->k; ++kx) {
      t.dy[t.n] = (signed char)(ky * g->dil - g->pad);
      t.dx[t.n] = (signed char)(kx * g->dil - g->pad);
      t.src[t.n] = (signed char)(ky * g->k + kx);
      ++t.n;
    }
  t.finish();
  return t;
}

static inline int posmod(int a, int m) { return ((a % m) + m) % m; }

// taps of the dgrad parity class (ry, rx): dX[s*m + ry] = sum_ky dY[m + (ry + pad - ky*dil)/s] w[ky]
static inline TapSet dgrad_taps(const pcuda_conv_geom* g, int ry, int rx) {
  TapSet t; t.n = 0;
  const int s = g->stride;
  for (int ky = 0; ky < g->k; ++ky) {
    const int vy = ry + g->pad - ky * g->dil;
    if (posmod(vy, s) != 0) continue;
    for (int kx = 0; kx < g->k; ++kx) {
      const int vx = rx + g->pad - kx * g->dil;
      if (posmod(vx, s) != 0) continue;
      t.dy[t.n] = (signed char)(vy / s);
      t.dx[t.n] = (signed char)(vx / s);
      t.src[t.n] = (signed char)(ky * g->k + kx);
      ++t.n;
    }
  }
  t.finish();
  return t;
}

// Stride-2 layers whose two COLUMN parity classes reach the same gradient pixels (k = 4, pad = 2, the discriminators'
// layers, GAN.py:97-105: both classes have the tap offsets {0,1} x {0,1}): the two classes run as ONE launch with rows
// (channel, column parity), so that a wave writes both halves of every 8 bytes of a destination line (a class on its own
// stores every other dword: the lines reach HBM half-written twice, WRITE_SIZE 2x the tensor) and the gradient tile is
// staged twice per layer instead of four times.  A property of (k, stride, pad, dil, cin) only: the packed image of a
// layer and its launches agree.  (cin <= 5 stays on the four-class image: the direct kernels of the first layer read it.)
static inline bool dgrad_pair_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("PCUDA_DGRAD_PAIR"); on = (e && !atoi(e)) ? 0 : 1; }
  return on != 0;
}
static inline bool dgrad_pair_ok(const pcuda_conv_geom* g) {
  if (g->stride != 2 || g->in_up || g->cin <= 5 || !dgrad_pair_enabled()) return false;
  for (int ry = 0; ry < 2; ++ry) {
    const TapSet a = dgrad_taps(g, ry, 0), b = dgrad_taps(g, ry, 1);
    if (a.n < 1 || a.n != b.n || 2 * a.n > IG_MAX_TAPS) return false;
    for (int i = 0; i < a.n; ++i)
      if (a.dy[i] != b.dy[i] || a.dx[i] != b.dx[i]) return false;
  }
  return true;
}
// taps of row class ry for both column classes: offsets once, src[0..n) for rx = 0 and src[n..2n) for rx = 1
static inline TapSet dgrad_pair_taps(const pcuda_conv_geom* g, int ry) {
  TapSet t = dgrad_taps(g, ry, 0);
  const TapSet b = dgrad_taps(g, ry, 1);
  for (int i = 0; i < b.n; ++i) t.src[t.n + i] = b.src[i];
  return t;
}

// candidate output-tile widths: powers of two plus even splits of the row (so a 17- or 33-wide map is
// not padded to 32 / 64)
static inline int tile_width_candidates(int lw, int tile_px, int* out) {
  int n = 0;
  for (int t = 8; t <= 256 && t <= tile_px; t <<= 1) out[n++] = t;
  for (int parts = 1; parts <= 4; ++parts) {
    const int t = (lw + parts - 1) / parts;
    if (t >= 4 && t <= 256 && t <= tile_px) {
      bool dup = false;
      for (int i = 0; i < n; ++i) dup |= out[i] == t;
      if (!dup) out[n++] = t;
    }
  }
  return n;
}


// fast (buffer-load) staging preconditions of the software-pipelined kernels: a 32-channel chunk lies
// in ONE source and every source image spans < 2^30 bytes (32-bit offsets, out-of-bounds marker)
static inline bool fast_src_ok(const pcuda_src* x, int cin) {
  const long long lim = 1ll << 30;
  const int c1 = x->c1 < cin ? x->c1 : cin;
  if (c1 < cin && (c1 & 31)) return false;
  if ((long long)(c1 + 32) * x->sc1 * 4 >= lim) return false;
  if (c1 < cin && (long long)(cin - c1 + 32) * x->sc2 * 4 >= lim) return false;
  return true;
}

// fast epilogue (buffer stores): an 8-row register group never straddles the two destinations and every
// destination image spans < 2^30 bytes
static inline bool fast_dst_ok(const pcuda_dst* y, int cout) {
  const long long lim = 1ll << 30;
  const int c1 = y->c1 < cout ? y->c1 : cout;
  if (c1 < cout && (c1 & 7)) return false;
  if ((long long)(c1 + 64) * y->sc1 * 4 >= lim) return false;
  if (c1 < cout && (long long)(cout - c1 + 64) * y->sc2 * 4 >= lim) return false;
  return true;
}

// tile / LDS plan of one forward or dgrad launch (conv_igemm.hip plans, conv_igemm_impl.h launches)
struct IgemmPlan {
  int npb, tw, th, tiles_x, tiles_y, ih_t, iw_t, clamp, x_cap, tg;
  size_t lds;
  int fat;   // one workgroup per CU, every (or many) taps of weights resident
  int w8;    // eight-wave kernel (one 512-thread workgroup per CU)
  int te;    // transposed epilogue (16-byte row stores through a wave-local LDS transposition)
};

// transposed epilogue preconditions: unit x stride, rows and tiles of 4k pixels, 16-byte aligned planes
static inline bool te_dst_ok(const pcuda_dst* y, int cout, int out_w, int lw, int tw, int ox_mul, int ox_off) {
  if (ox_mul != 1 || ox_off != 0 || (tw & 3) || (out_w & 3) || (lw & 3)) return false;
  if (((uintptr_t)y->p1 & 15) || (y->sc1 & 3) || (y->sn1 & 3)) return false;
  if (y->c1 < cout && (((uintptr_t)y->p2 & 15) || (y->sc2 & 3) || (y->sn2 & 3))) return false;
  return true;
}


// anti-phase 3x3 kernel (conv_ap.hip): layer property, packed image, launcher (returns 1 when it took the launch)
struct IgemmParams;
struct PackParams;
bool ap_layer_ok(const pcuda_conv_geom* g, int rows, int red, int prec);
size_t ap_layer_packed_bytes(int rows, int red);
bool ap_fill_pack(PackParams& p, const float* w, unsigned char* out, int rows, int red, long long s_row, long long s_red,
                  const TapSet& taps);
int ap_launch_pack(const float* w, unsigned char* out, int rows, int red, long long s_row, long long s_red, const TapSet& taps,
                   hipStream_t s);
bool ap_map_ok(int n, int h, int w, int rows);
int ap_tiles(int n, int h, int w);
int ap_try_launch(const IgemmParams& p, const TapSet& taps, const unsigned char* image, hipStream_t s, int* rc);

// row-streaming 3x3 kernel of the 32 -> 32-channel layers (conv_rs.hip): reads the ORDINARY packed layout
bool rs_layer_ok(const pcuda_conv_geom* g, int rows, int red, int prec);
bool rs_map_ok(int n, int h, int w);
int rs_tiles(int n, int h, int w);
int rs_try_launch(const IgemmParams& p, int prec, const TapSet& taps, hipStream_t s, int* rc);

// direct (vector-ALU) kernels of the degenerate layers (conv_direct.hip); each returns 1 when it took the launch
int direct_fwd_tiles(const pcuda_conv_geom* g);
int direct_forward(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const void* packed_w, long long w_lo_off,
                   const float* bias, float slope, const pcuda_dst* y, float* bn_partials, hipStream_t s, int* rc);
int direct_dgrad_tiles(const pcuda_conv_geom* g);
int direct_dgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad, const pcuda_dst* dx,
                 int accumulate, hipStream_t s, int* rc, const float* red_a = nullptr, long long red_sn = 0,
                 long long red_sc = 0, const float* red_mean = nullptr, const float* red_invstd = nullptr,
                 float* red = nullptr);
size_t direct_wgrad_workspace(const pcuda_conv_geom* g);
int direct_wgrad(const pcuda_conv_geom* g, const pcuda_src* x, const float* dz, long long dz_sn, long long dz_sc, float* dw,
                 float* db, int accumulate, void* workspace, hipStream_t s, int* rc);
int direct_d1_dgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad, const pcuda_dst* dx,
                    int accumulate, size_t cls_elems, hipStream_t s, int* rc);
size_t direct_d1_wgrad_workspace(const pcuda_conv_geom* g);
int direct_d1_wgrad(const pcuda_conv_geom* g, const pcuda_src* x, const float* dz, long long dz_sn, long long dz_sc, float* dw,
                    float* db, int accumulate, void* workspace, hipStream_t s, int* rc);
int launch_wgrad_reduce_taps(const float* partial, long long numel, int ksplit, float* dw, int accumulate, int ntaps,
                             const float* db_partial, long long nb, float* db, hipStream_t s);
size_t wgrad3r_workspace(const pcuda_conv_geom* g);
int wgrad3r_try(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy, long long dy_sn, long long dy_sc,
                float* dw, float* db, int accumulate, void* workspace, hipStream_t s, pcuda_reduce_job* defer, int* rc);
size_t wgrad1_workspace(const pcuda_conv_geom* g);
int wgrad1_try(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy, long long dy_sn, long long dy_sc,
               float* dw, float* db, int accumulate, void* workspace, hipStream_t s, pcuda_reduce_job* defer, int* rc);
size_t wgrad3_workspace(const pcuda_conv_geom* g);
int wgrad3_try(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy, long long dy_sn, long long dy_sc,
               float* dw, float* db, int accumulate, void* workspace, hipStream_t s, pcuda_reduce_job* defer, int* rc);
int launch_wgrad_reduce(const float* partial, long long numel, int ksplit, float* dw, int accumulate, const float* db_partial,
                        long long nb, float* db, hipStream_t s);
int direct_d1_forward(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const void* packed_w, const float* bias,
                      float slope, const pcuda_dst* y, float* bn_partials, hipStream_t s, int* rc);
int direct_d5_forward(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const void* packed_w, const float* bias,
                      float slope, const pcuda_dst* y, float* bn_partials, hipStream_t s, int* rc);
