// Weight gradient of the 1x1 / stride-1 layers (the residual 1x1 convolutions of the encoder, unet.py:32 / :41-47 going
// back, and the classifiers, :178): dW[co][ci] = sum over (image, pixel) of dZ[co][px] * X[ci][px] -- a plain NT GEMM over
// pixels whose two operands are both natural rows with the reduction index contiguous.  An MFMA operand fragment (lane =
// row, 8 consecutive reduction elements) is therefore two float4 loads straight from global memory: no pixel records, no
// LDS staging, no transposing reads.  The generic kernel (conv_wgrad_impl.h) staged a 128-pixel record tile per 32-channel
// chunk and ran these layers at 47-49 TFLOP/s and 0.6-2.5 TB/s (profiles/r04_layer_table.txt: four layers, 0.13 ms each);
// they are HBM-bound at a fraction of that time.
//   * a wave owns a [32 CO_B] x [32 CI_B] block of dW (up to 64 x 96: every operand element is loaded once per block) and
//     walks 16-pixel reduction steps; the loads of step i + 1 are in flight during the split and the MFMAs of step i;
//   * the four waves of a workgroup interleave the steps of one contiguous pixel range (a row's 256 consecutive bytes
//     per round) and add their accumulators in a fixed order through LDS: one slab per workgroup;
//   * slabs [slice][co][ci] and the bias partials go through the same reduce as every other weight gradient
//     (launch_wgrad_reduce_taps with one tap): fixed summation order, bit-reproducible;
//   * lazy BatchNorm (affine on load) and the two-source input (zero-copy concatenation) are per-row constants here.
#include "conv_device.h"
#include "conv_host.h"

namespace {

struct W1Params {
  pcuda_src x;
  int cin, cout, n;
  int hw;                 // pixels per plane (a multiple of 16)
  const float* dz; long long dz_sn, dz_sc;
  float* partial;         // [slices][cout][cin]
  float* db_partial;      // [slices][cout] or NULL
  int steps_per_wg;       // 16-pixel steps of one workgroup's range (divides hw / 16)
  int wgs_per_image;
  int n_ci_tiles, tiles;  // (co tile, ci tile) blocks of dW
  int xcd;                // 1: slice-major work items dealt to the XCDs in contiguous eighths
};

template <int NB>
struct W1Raw {
  f32x4 v[NB][2];
};

template <bool X3, int CO_B, int CI_B>
__global__ __launch_bounds__(256, 2) void wgrad1_kernel(const W1Params p) {
  __shared__ float red[CO_B * CI_B * 16 * 64];
  __shared__ float dbs[4][CO_B * 32];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

  int item = blockIdx.x;
  if (p.xcd) item = (blockIdx.x & 7) * ((int)gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = item % p.tiles, slice = item / p.tiles;
  const int cit = tile % p.n_ci_tiles, cot = tile / p.n_ci_tiles;
  const int img = slice / p.wgs_per_image, part = slice - img * p.wgs_per_image;
  const long long px0 = (long long)part * p.steps_per_wg * 16 + 8 * h;      // this lane's first pixel

  // ---- row pointers: dZ rows (A operand), X rows (B operand: source select, lazy-BatchNorm affine per row)
  const float* pa[CO_B];
#pragma unroll
  for (int cb = 0; cb < CO_B; ++cb) {
    const int co = min((cot * CO_B + cb) * 32 + r, p.cout - 1);             // rows past cout: a valid address, never stored
    pa[cb] = p.dz + (long long)img * p.dz_sn + (long long)co * p.dz_sc + px0;
  }
  const float* pb[CI_B];
  float sc[CI_B], sh[CI_B];
#pragma unroll
  for (int cib = 0; cib < CI_B; ++cib) {
    const int ci = min((cit * CI_B + cib) * 32 + r, p.cin - 1);
    const bool first = ci < p.x.c1;
    const float* base = first ? p.x.p1 : p.x.p2;
    const long long sn = first ? p.x.sn1 : p.x.sn2, scs = first ? p.x.sc1 : p.x.sc2;
    const int cl = first ? ci : ci - p.x.c1;
    pb[cib] = base + (long long)img * sn + (long long)cl * scs + px0;
    const float* scp = first ? (const float*)p.x.scale1 : (const float*)p.x.scale2;
    const float* shp = first ? (const float*)p.x.shift1 : (const float*)p.x.shift2;
    sc[cib] = scp ? scp[cl] : 1.f;
    sh[cib] = scp ? shp[cl] : 0.f;
  }
  const bool affine = p.x.scale1 != nullptr || p.x.scale2 != nullptr;      // (uniform)

  f32x16 acc[CO_B][CI_B];
#pragma unroll
  for (int cb = 0; cb < CO_B; ++cb)
#pragma unroll
    for (int cib = 0; cib < CI_B; ++cib)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][cib][i] = 0.f;
  float dbacc[CO_B];
#pragma unroll
  for (int cb = 0; cb < CO_B; ++cb) dbacc[cb] = 0.f;
  const bool do_db = p.db_partial != nullptr && cit == 0;

  // wave w takes steps w, w + 4, ... of the workgroup's range
  const int nsteps = (p.steps_per_wg - w + 3) >> 2;
  W1Raw<CO_B> ra;
  W1Raw<CI_B> rb;
  auto issue = [&](int step) {
    const int off = (w + 4 * step) * 16;
#pragma unroll
    for (int cb = 0; cb < CO_B; ++cb) {
      ra.v[cb][0] = *(const f32x4*)(pa[cb] + off);
      ra.v[cb][1] = *(const f32x4*)(pa[cb] + off + 4);
    }
#pragma unroll
    for (int cib = 0; cib < CI_B; ++cib) {
      rb.v[cib][0] = *(const f32x4*)(pb[cib] + off);
      rb.v[cib][1] = *(const f32x4*)(pb[cib] + off + 4);
    }
  };
  if (nsteps > 0) issue(0);
  for (int st = 0; st < nsteps; ++st) {
    // ---- this step's fragments out of the raw registers (which the next step's loads then overwrite)
    bf16x8 ah[CO_B], al[CO_B], bh[CI_B], bl[CI_B];
#pragma unroll
    for (int cb = 0; cb < CO_B; ++cb) {
      const f32x4 a = ra.v[cb][0], b = ra.v[cb][1];
      if (do_db) dbacc[cb] += ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
      uint4 hi, lo = make_uint4(0, 0, 0, 0);
      if (X3) {
        split2(a[0], a[1], hi.x, lo.x); split2(a[2], a[3], hi.y, lo.y);
        split2(b[0], b[1], hi.z, lo.z); split2(b[2], b[3], hi.w, lo.w);
      } else {
        hi.x = pack_bf16x2(a[0], a[1]); hi.y = pack_bf16x2(a[2], a[3]);
        hi.z = pack_bf16x2(b[0], b[1]); hi.w = pack_bf16x2(b[2], b[3]);
      }
      ah[cb] = __builtin_bit_cast(bf16x8, hi);
      al[cb] = __builtin_bit_cast(bf16x8, lo);
    }
#pragma unroll
    for (int cib = 0; cib < CI_B; ++cib) {
      f32x4 a = rb.v[cib][0], b = rb.v[cib][1];
      if (affine) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = fmaf(a[e], sc[cib], sh[cib]); b[e] = fmaf(b[e], sc[cib], sh[cib]); }
      }
      uint4 hi, lo = make_uint4(0, 0, 0, 0);
      if (X3) {
        split2(a[0], a[1], hi.x, lo.x); split2(a[2], a[3], hi.y, lo.y);
        split2(b[0], b[1], hi.z, lo.z); split2(b[2], b[3], hi.w, lo.w);
      } else {
        hi.x = pack_bf16x2(a[0], a[1]); hi.y = pack_bf16x2(a[2], a[3]);
        hi.z = pack_bf16x2(b[0], b[1]); hi.w = pack_bf16x2(b[2], b[3]);
      }
      bh[cib] = __builtin_bit_cast(bf16x8, hi);
      bl[cib] = __builtin_bit_cast(bf16x8, lo);
    }
    if (st + 1 < nsteps) issue(st + 1);        // in flight during the MFMAs below
#pragma unroll
    for (int cb = 0; cb < CO_B; ++cb)
#pragma unroll
      for (int cib = 0; cib < CI_B; ++cib) {
        if (X3) {
          acc[cb][cib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb], bh[cib], acc[cb][cib], 0, 0, 0);
          acc[cb][cib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bl[cib], acc[cb][cib], 0, 0, 0);
        }
        acc[cb][cib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bh[cib], acc[cb][cib], 0, 0, 0);
      }
  }

  // ---- the four waves' sums in a fixed order: ((w0 + w1) + w2) + w3
  for (int ww = 1; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int cb = 0; cb < CO_B; ++cb)
#pragma unroll
        for (int cib = 0; cib < CI_B; ++cib)
#pragma unroll
          for (int i = 0; i < 16; ++i) red[((cb * CI_B + cib) * 16 + i) * 64 + lane] = acc[cb][cib][i];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int cb = 0; cb < CO_B; ++cb)
#pragma unroll
        for (int cib = 0; cib < CI_B; ++cib)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[cb][cib][i] += red[((cb * CI_B + cib) * 16 + i) * 64 + lane];
    }
    __syncthreads();
  }
  if (do_db) {
#pragma unroll
    for (int cb = 0; cb < CO_B; ++cb) {
      const float s = dbacc[cb] + __shfl_xor(dbacc[cb], 32, 64);          // the row's two pixel octets
      if (h == 0) dbs[w][cb * 32 + r] = s;
    }
  }
  __syncthreads();
  if (w != 0) return;
  // slab [slice][co][ci]: ci on the lanes (128-byte segments), as the generic kernel writes it
#pragma unroll
  for (int cb = 0; cb < CO_B; ++cb)
#pragma unroll
    for (int cib = 0; cib < CI_B; ++cib) {
      const int ci = (cit * CI_B + cib) * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = (cot * CO_B + cb) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (co < p.cout && ci < p.cin) p.partial[((long long)slice * p.cout + co) * p.cin + ci] = acc[cb][cib][i];
      }
    }
  if (do_db && h == 0) {
#pragma unroll
    for (int cb = 0; cb < CO_B; ++cb) {
      const int co = (cot * CO_B + cb) * 32 + r;
      if (co < p.cout)
        p.db_partial[(long long)slice * p.cout + co] =
            ((dbs[0][cb * 32 + r] + dbs[1][cb * 32 + r]) + dbs[2][cb * 32 + r]) + dbs[3][cb * 32 + r];
    }
  }
}

struct W1Plan {
  int co_b, ci_b, n_co_tiles, n_ci_tiles, steps_per_wg, wgs_per_image, slices;
};

bool w1_geom(const pcuda_conv_geom* g) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("PCUDA_NO_WGRAD1"); off = (e && atoi(e)) ? 1 : 0; }
  if (off) return false;
  const long long hw = (long long)g->in_h * g->in_w;
  return g->k == 1 && g->stride == 1 && g->pad == 0 && !g->in_up && (hw & 15) == 0 && hw < (1ll << 28) && g->cin >= 16;
}

W1Plan w1_plan(const pcuda_conv_geom* g) {
  W1Plan w;
  w.co_b = g->cout > 32 ? 2 : 1;
  w.ci_b = g->cin > 64 ? 3 : (g->cin > 32 ? 2 : 1);
  w.n_co_tiles = cdiv(g->cout, 32 * w.co_b);
  w.n_ci_tiles = cdiv(g->cin, 32 * w.ci_b);
  const int tiles = w.n_co_tiles * w.n_ci_tiles;
  const long long spi = (long long)g->in_h * g->in_w / 16, total = spi * g->n;
  // slices: ~768 workgroups in all (measured 768 / 1024 / 2048 / 3072: the single-block layers 0.092 / 0.097 / 0.098 / 0.099 ms,
  // the others flat), the slabs (written once, read once) below an eighth of the operand bytes (8 MB at least), every
  // workgroup at least eight steps
  static int tgt = -1;
  if (tgt < 0) { const char* e = getenv("PCUDA_WG1_BLOCKS"); tgt = e ? atoi(e) : 768; }
  long long want = tgt / tiles;
  const long long welems = (long long)g->cout * g->cin;
  const long long in_bytes = total * 16 * (g->cin + g->cout) * 4;
  long long cap = in_bytes / 8;
  if (cap < (8ll << 20)) cap = 8ll << 20;
  if (want * welems * 4 > cap) want = cap / (welems * 4);
  if (want > total / 8) want = total / 8;
  if (want < 1) want = 1;
  long long d = cdiv(total, want);            // steps per workgroup: the smallest divisor of spi that is >= this
  if (d > spi) d = spi;
  while (spi % d) ++d;
  w.steps_per_wg = (int)d;
  w.wgs_per_image = (int)(spi / d);
  w.slices = w.wgs_per_image * g->n;
  return w;
}

}  // namespace

size_t wgrad1_workspace(const pcuda_conv_geom* g) {
  if (!w1_geom(g)) return 0;
  const W1Plan w = w1_plan(g);
  return ((size_t)w.slices * g->cout * g->cin + (size_t)w.slices * g->cout) * sizeof(float) + 256;
}

// returns 1 when it took the launch (*rc = status)
int wgrad1_try(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy, long long dy_sn, long long dy_sc,
               float* dw, float* db, int accumulate, void* workspace, hipStream_t s, pcuda_reduce_job* defer, int* rc) {
  if (!w1_geom(g)) return 0;
  auto al = [](const void* q, long long sn, long long sc) { return q == nullptr || ((((uintptr_t)q) & 15) == 0 && (sn & 3) == 0 && (sc & 3) == 0); };
  const int c1 = x->c1 < g->cin ? x->c1 : g->cin;
  if (!al(x->p1, x->sn1, x->sc1) || (c1 < g->cin && !al(x->p2, x->sn2, x->sc2)) || !al(dy, dy_sn, dy_sc)) return 0;
  // (a second source needs its own affine pointers resolved per row: both NULL or per-source as given)
  const bool x3 = prec == PCUDA_PREC_BF16X3;
  const W1Plan w = w1_plan(g);
  W1Params p;
  memset(&p, 0, sizeof(p));
  p.x = *x; p.x.c1 = c1;
  p.cin = g->cin; p.cout = g->cout; p.n = g->n; p.hw = g->in_h * g->in_w;
  p.dz = dy; p.dz_sn = dy_sn; p.dz_sc = dy_sc;
  const long long welems = (long long)g->cout * g->cin;
  p.partial = (float*)workspace;
  p.db_partial = db ? (float*)workspace + (size_t)w.slices * welems : nullptr;
  p.steps_per_wg = w.steps_per_wg; p.wgs_per_image = w.wgs_per_image;
  p.n_ci_tiles = w.n_ci_tiles; p.tiles = w.n_co_tiles * w.n_ci_tiles;
  const long long nwg = (long long)p.tiles * w.slices;
  if (nwg >= (1ll << 31)) return 0;
  p.xcd = (nwg >= 16 && (nwg & 7) == 0) ? 1 : 0;
  {
    char tag[160];
    snprintf(tag, sizeof(tag), "wgrad1 n%d cin%d cout%d %dx%d k1 slices%d steps%d tile%dx%d", g->n, g->cin, g->cout, g->in_h,
             g->in_w, w.slices, w.steps_per_wg, 32 * w.co_b, 32 * w.ci_b);
    ProfScope prof(PCUDA_FAM_CONV_WGRAD, 2.0 * g->n * (double)p.hw * g->cout * (double)g->cin, s, tag);
    const dim3 grid((unsigned)nwg);
#define W1_LAUNCH(X3_, CO_, CI_) hipLaunchKernelGGL((wgrad1_kernel<X3_, CO_, CI_>), grid, dim3(256), 0, s, p)
#define W1_CI(X3_, CO_)                                                \
  do {                                                                 \
    if (w.ci_b == 3) W1_LAUNCH(X3_, CO_, 3);                           \
    else if (w.ci_b == 2) W1_LAUNCH(X3_, CO_, 2);                      \
    else W1_LAUNCH(X3_, CO_, 1);                                       \
  } while (0)
    if (x3) { if (w.co_b == 2) W1_CI(true, 2); else W1_CI(true, 1); }
    else { if (w.co_b == 2) W1_CI(false, 2); else W1_CI(false, 1); }
#undef W1_CI
#undef W1_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pcuda_set_error("wgrad1_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; return 1; }
  }
  int nkg = 1;
  while (nkg < 16 && nkg * 2 <= w.slices) nkg <<= 1;
  if (defer) {
    defer->partial = (const float*)workspace; defer->numel = welems; defer->ksplit = w.slices; defer->nkg = nkg;
    defer->dw = dw; defer->accumulate = accumulate; defer->ntaps = 1;
    defer->db_partial = (const float*)p.db_partial; defer->nb = db ? g->cout : 0; defer->db = db;
    *rc = PCUDA_OK;
    return 1;
  }
  *rc = launch_wgrad_reduce_taps((const float*)workspace, welems, w.slices, dw, accumulate, 1, (const float*)p.db_partial,
                                 db ? g->cout : 0, db, s);
  return 1;
}
