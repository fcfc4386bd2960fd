// Weight-gradient kernels of the implicit-GEMM convolution (see conv_igemm.hip for the layout).
#include "conv_igemm.h"
#include "conv_host.h"

// dw[i] (+)= sum_k partial[k][i], deterministic.  A workgroup owns 64 consecutive outputs; its
// blockDim/64 k-groups each sum a strided subset of the slices (8 loads in flight per lane), and the
// k-group partials are combined in a fixed order through LDS.  (One thread per output with a serial
// loop over up to 1024 slices was latency-bound: 0.5 ms for a 9K-weight layer.)
// ntaps > 1: partial is [k][tap][co*cin] and dw is [co*cin][tap] (OIHW): coalesced reads, the
// transpose costs one scattered write per weight.
// The bias gradient rides in the same launch: workgroups past the weight slab's reduce db's [k][cout] slab.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, long long numel,
                                                           int ksplit, float* __restrict__ dw, int accumulate,
                                                           int ntaps, const float* __restrict__ db_partial,
                                                           long long nb, float* __restrict__ db) {
  __shared__ float sh[16][64];
  const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6, nkg = blockDim.x >> 6;
  const long long nblk_w = (numel + 63) / 64;
  long long blk = blockIdx.x;
  if (blk >= nblk_w) {   // uniform per workgroup: switch to the bias problem
    blk -= nblk_w; partial = db_partial; numel = nb; dw = db; ntaps = 1;
  }
  const long long i = blk * 64 + lane;
  float s = 0.f;
  if (i < numel) {
    int k = kg;
    for (; k + 7 * nkg < ksplit; k += 8 * nkg) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(k + u * nkg) * numel + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < ksplit; k += nkg) s += partial[(long long)k * numel + i];
  }
  sh[kg][lane] = s;
  __syncthreads();
  if (kg == 0 && i < numel) {
    float t = 0.f;
    for (int q = 0; q < nkg; ++q) t += sh[q][lane];
    long long o = i;
    if (ntaps > 1) {
      const long long cc = numel / ntaps;          // cout * cin
      const long long tt = i / cc, r = i - tt * cc;
      o = r * ntaps + tt;
    }
    dw[o] = accumulate ? dw[o] + t : t;
  }
}

// the split-K reduce for other translation units (the direct kernels' block partials): partial = [ksplit][numel],
// dw[i] (+)= sum over k in a fixed order; db_partial = [ksplit][nb] or NULL
int launch_wgrad_reduce_taps(const float* partial, long long numel, int ksplit, float* dw, int accumulate, int ntaps,
                             const float* db_partial, long long nb, float* db, hipStream_t s) {
  int nkg = 1;
  while (nkg < 16 && nkg * 2 <= ksplit) nkg <<= 1;
  const unsigned nblk = (unsigned)cdiv(numel, 64) + (db ? (unsigned)cdiv(nb, 64) : 0u);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk), dim3(64 * nkg), 0, s, partial, numel, ksplit, dw, accumulate, ntaps,
                     db_partial, db ? nb : 0, db);
  PCUDA_CHECK_LAUNCH("wgrad_reduce_kernel");
  return PCUDA_OK;
}
int launch_wgrad_reduce(const float* partial, long long numel, int ksplit, float* dw, int accumulate, const float* db_partial,
                        long long nb, float* db, hipStream_t s) {
  return launch_wgrad_reduce_taps(partial, numel, ksplit, dw, accumulate, 1, db_partial, nb, db, s);
}

// ---- the split-K reduces of MANY weight gradients in one launch (a backward pass issues one per layer: 80 launches of
// ~15 us per step, most of it launch and drain).  Jobs travel as kernel arguments; a workgroup finds its job in the
// prefix of 64-output blocks.  Same arithmetic and order per output as wgrad_reduce_kernel with 16 k-groups ... except
// that the number of k-groups is the job's own (nkg), so that the sums are bit-identical to the one-by-one form.
struct ReduceJobs {
  int n;
  int first_block[PCUDA_REDUCE_MAX_JOBS + 1];
  pcuda_reduce_job j[PCUDA_REDUCE_MAX_JOBS];
};

__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(const ReduceJobs jobs) {
  __shared__ float sh[16][64];
  int lo = 0, hi = jobs.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const pcuda_reduce_job& jb = jobs.j[lo];
  const float* partial = jb.partial;
  long long numel = jb.numel;
  float* dw = jb.dw;
  int ntaps = jb.ntaps;
  const int ksplit = jb.ksplit, nkg = jb.nkg;
  const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const long long nblk_w = (numel + 63) / 64;
  long long blk = (int)blockIdx.x - jobs.first_block[lo];
  if (blk >= nblk_w) { blk -= nblk_w; partial = jb.db_partial; numel = jb.nb; dw = jb.db; ntaps = 1; }
  const long long i = blk * 64 + lane;
  float s = 0.f;
  if (i < numel && kg < nkg) {
    int k = kg;
    for (; k + 7 * nkg < ksplit; k += 8 * nkg) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(k + u * nkg) * numel + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < ksplit; k += nkg) s += partial[(long long)k * numel + i];
  }
  sh[kg][lane] = s;
  __syncthreads();
  if (kg == 0 && i < numel) {
    float t = 0.f;
    for (int q = 0; q < nkg; ++q) t += sh[q][lane];
    long long o = i;
    if (ntaps > 1) {
      const long long cc = numel / ntaps;
      const long long tt = i / cc, r = i - tt * cc;
      o = r * ntaps + tt;
    }
    dw[o] = jb.accumulate ? dw[o] + t : t;
  }
}

extern "C" int pcuda_wgrad_reduce_batch(const pcuda_reduce_job* host_jobs, int njobs, pcuda_stream_t s) {
  if (!host_jobs || njobs < 0) PCUDA_FAIL(PCUDA_E_BADARG, "wgrad_reduce_batch: bad arguments");
  int done = 0;
  while (done < njobs) {
    ReduceJobs jobs;
    jobs.n = 0;
    int blocks = 0;
    for (; done < njobs && jobs.n < PCUDA_REDUCE_MAX_JOBS; ++done) {
      const pcuda_reduce_job& jb = host_jobs[done];
      if (jb.ksplit <= 0) continue;      // (the layer's own kernel already wrote dw)
      if (!jb.partial || !jb.dw || jb.numel <= 0 || jb.nkg < 1 || jb.nkg > 16 || (jb.db && !jb.db_partial))
        PCUDA_FAIL(PCUDA_E_BADARG, "wgrad_reduce_batch: bad job %d", done);
      jobs.first_block[jobs.n] = blocks;
      jobs.j[jobs.n++] = jb;
      blocks += (int)cdiv(jb.numel, 64) + (jb.db ? (int)cdiv(jb.nb, 64) : 0);
    }
    if (jobs.n == 0) break;
    jobs.first_block[jobs.n] = blocks;
    ProfScope prof(PCUDA_FAM_POINTWISE, 0.0, (hipStream_t)s, "wgrad reduce batch");
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)s, jobs);
    PCUDA_CHECK_LAUNCH("wgrad_reduce_batch_kernel");
  }
  return PCUDA_OK;
}

namespace {

struct WgradPlan {
  int co_blks, co_tile, n_co_tiles, n_chunks, tap_groups, taps_per_group, tw, th, tiles_x, tiles_y, ksplit;
  int ih_t, iw_t;
};

WgradPlan plan_wgrad(const pcuda_conv_geom* g) {
  WgradPlan w;
  w.co_blks = ig_co_blks(g->cout);
  w.co_tile = 32 * w.co_blks;
  w.n_co_tiles = cdiv(g->cout, w.co_tile);
  w.n_chunks = cdiv(g->cin, 32);
  const int ntaps = g->k * g->k;
  w.tap_groups = ntaps <= 16 ? 1 : cdiv(ntaps, 9);   // 4x4 kernels keep all 16 taps in one block: X and dZ staged once
  w.taps_per_group = cdiv(ntaps, w.tap_groups);
  {   // 128-slot tile with the fewest tiles; ties prefer widths that keep the float4 dZ path (TW % 8 == 0)
    int cand[16];
    const int nc = tile_width_candidates(g->out_w, 128, cand);
    long long best_key = -1;
    w.tw = 32; w.th = 4;
    for (int ci = 0; ci < nc; ++ci) {
      const int tw = cand[ci], th = 128 / tw;
      if (th < 1) continue;
      const int span = (g->k - 1) * g->dil;
      const long long halo = (long long)((th - 1) * g->stride + span + 1) * ((tw - 1) * g->stride + span + 1);
      {   // the staged input tile (clipped to the image where that halves it) + the dZ rows must fit LDS in bf16x3
        const long long ih = (th - 1) * g->stride + span + 1, iw = (tw - 1) * g->stride + span + 1;
        const long long clipped = (ih < g->in_h ? ih : g->in_h) * (iw < g->in_w ? iw : g->in_w) + 1;
        const long long cap = halo < clipped ? halo : clipped;
        if (cap * IG_REC_BYTES * 2 + (long long)w.co_tile * WG_ZROW * 2 > (long long)LDS_HARD) continue;
      }
      // (a staged tile of more than 768 pixels runs unpipelined -- no register prefetch, four waves: the 16-tap layers of the
      // 224x224 network's 57-wide maps took tw = 64 over tw = 57 for its aligned rows and ran at half the rate)
      const long long ih2 = (th - 1) * g->stride + span + 1, iw2 = (tw - 1) * g->stride + span + 1;
      const long long clipped2 = (ih2 < g->in_h ? ih2 : g->in_h) * (iw2 < g->in_w ? iw2 : g->in_w) + 1;
      const long long staged = (clipped2 * 2 <= halo) ? clipped2 : halo;
      const long long key = ((long long)cdiv(g->out_w, tw) * cdiv(g->out_h, th) << 24) + (staged > 768 ? (1ll << 22) : 0) +
                            ((tw & 31) ? (1ll << 20) : 0) + halo;
      if (best_key < 0 || key < best_key) { best_key = key; w.tw = tw; w.th = th; }
    }
  }
  const int TW = w.tw, TH = w.th;
  w.tiles_x = cdiv(g->out_w, TW);
  w.tiles_y = cdiv(g->out_h, TH);
  const int span = (g->k - 1) * g->dil;
  w.ih_t = (TH - 1) * g->stride + span + 1;
  w.iw_t = (TW - 1) * g->stride + span + 1;
  const long long ntiles = (long long)g->n * w.tiles_x * w.tiles_y;
  const int base = w.n_co_tiles * w.n_chunks * w.tap_groups;
  // blocks in flight: 32-row blocks (160 VGPRs, 50 KB of LDS) are resident three per CU, 64-row blocks two per CU --
  // 768 / 1024 blocks are one / two balanced rounds (1024 32-row blocks were one full round plus a third of one)
  static int tgt32 = -1;
  if (tgt32 < 0) { const char* e = getenv("PCUDA_WG_BLOCKS32"); tgt32 = e ? atoi(e) : 768; }
  static int tgt64 = -1;
  if (tgt64 < 0) { const char* e = getenv("PCUDA_WG_BLOCKS64"); tgt64 = e ? atoi(e) : 1024; }
  // plans whose LDS footprint leaves room for ONE workgroup per CU (the 16-tap stride-2 layers of the discriminators on the
  // eight-wave kernel, the dilated bottleneck layers): one round of 256 blocks -- two rounds only doubled their split-K slabs
  // (round 5, scripts/conv_micro.py: d2 / d3 / d4 0.220 / 0.219 / 0.278 -> 0.205 / 0.204 / 0.258 ms, 512->512 at 16x16 dilation 4
  // 0.253 -> 0.241; the two-per-CU plans lose 20-30 % at 256).  Sized on the bf16x3 footprint so that the slab count does
  // not depend on the precision (the workspace query has no precision argument).
  static int tgt1 = -1;
  if (tgt1 < 0) { const char* e = getenv("PCUDA_WG_BLOCKS1"); tgt1 = e ? atoi(e) : 256; }
  bool one_per_cu = false;
  {
    const long long full = (long long)w.ih_t * w.iw_t;
    const long long clipped = (long long)(w.ih_t < g->in_h ? w.ih_t : g->in_h) * (w.iw_t < g->in_w ? w.iw_t : g->in_w) + 1;
    const size_t zb = (size_t)w.co_tile * WG_ZROW * 2;
    bool clamp = clipped * 2 <= full;
    if (!clamp && (size_t)full * IG_REC_BYTES * 2 + zb > (size_t)LDS_HARD) clamp = true;
    one_per_cu = (size_t)(clamp ? clipped : full) * IG_REC_BYTES * 2 + zb > (size_t)LDS_HARD / 2;
  }
  long long ks = (one_per_cu ? tgt1 : (w.co_blks == 1 ? tgt32 : tgt64)) / base;
  if (ks < 1) ks = 1;
  if (ks > ntiles) ks = ntiles;
  // keep the partial slabs (written once, re-read once by the reduce kernel) below ~64 MB (measured:
  // 32 MB costs the wgrad kernels more parallelism than the reduce kernel saves)
  const long long welems = (long long)g->cout * g->cin * ntaps;
  while (ks > 1 && ks * welems * 4 > (64ll << 20)) ks >>= 1;
  if (ks >= 8) ks &= ~7ll;      // multiples of 8: an XCD's slices interleave over its eighth of the tiles (wgrad_kernel)
  w.ksplit = (int)ks;
  return w;
}

}  // namespace

extern "C" size_t pcuda_conv2d_wgrad_workspace_size(const pcuda_conv_geom* g) {
  if (!geom_ok(g)) return 0;
  WgradPlan w = plan_wgrad(g);
  const size_t need = ((size_t)w.ksplit * g->cout * g->cin * g->k * g->k + (size_t)w.ksplit * g->cout) * sizeof(float) + 256;
  size_t dneed = direct_wgrad_workspace(g);
  const size_t d1need = direct_d1_wgrad_workspace(g);
  if (d1need > dneed) dneed = d1need;
  const size_t w3need = wgrad3_workspace(g);
  if (w3need > dneed) dneed = w3need;
  const size_t w1need = wgrad1_workspace(g);
  if (w1need > dneed) dneed = w1need;
  const size_t w3rneed = wgrad3r_workspace(g);
  if (w3rneed > dneed) dneed = w3rneed;
  return need > dneed ? need : dneed;
}

int wgrad_dispatch_x3(const WgradParams& p, int co_blks, bool clamp, int taps_max, int pf, int x_cap, size_t lds,
                      float* dbp, dim3 grid, hipStream_t s);
int wgrad_dispatch_bf16(const WgradParams& p, int co_blks, bool clamp, int taps_max, int pf, int x_cap, size_t lds,
                        float* dbp, dim3 grid, hipStream_t s);

static int wgrad_impl(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy, long long dy_sn, long long dy_sc,
                      float* dw, float* db, int accumulate, void* workspace, size_t workspace_bytes, pcuda_stream_t s_,
                      pcuda_reduce_job* defer);

extern "C" int pcuda_conv2d_wgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy,
                                  long long dy_sn, long long dy_sc, float* dw, float* db, int accumulate,
                                  void* workspace, size_t workspace_bytes, pcuda_stream_t s_) {
  return wgrad_impl(g, prec, x, dy, dy_sn, dy_sc, dw, db, accumulate, workspace, workspace_bytes, s_, nullptr);
}

extern "C" int pcuda_conv2d_wgrad_partial(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy,
                                          long long dy_sn, long long dy_sc, float* dw, float* db, int accumulate,
                                          void* workspace, size_t workspace_bytes, pcuda_reduce_job* job, pcuda_stream_t s_) {
  if (!job) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_wgrad_partial: job is NULL");
  memset(job, 0, sizeof(*job));
  return wgrad_impl(g, prec, x, dy, dy_sn, dy_sc, dw, db, accumulate, workspace, workspace_bytes, s_, job);
}

static int wgrad_impl(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy, long long dy_sn, long long dy_sc,
                      float* dw, float* db, int accumulate, void* workspace, size_t workspace_bytes, pcuda_stream_t s_,
                      pcuda_reduce_job* defer) {
  hipStream_t s = (hipStream_t)s_;
  if (!geom_ok(g)) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_wgrad: inconsistent geometry");
  if (!src_ok(x, g->cin) || !dy || !dw) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_wgrad: bad tensors");
  if (prec != PCUDA_PREC_BF16X3 && prec != PCUDA_PREC_BF16) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_wgrad: bad precision");
  if (!workspace || workspace_bytes < pcuda_conv2d_wgrad_workspace_size(g))
    PCUDA_FAIL(PCUDA_E_WORKSPACE, "conv2d_wgrad: workspace too small (%zu < %zu)", workspace_bytes,
               pcuda_conv2d_wgrad_workspace_size(g));
  {
    int rc;
    if (direct_wgrad(g, x, dy, dy_sn, dy_sc, dw, db, accumulate, workspace, s, &rc)) return rc;
    if (direct_d1_wgrad(g, x, dy, dy_sn, dy_sc, dw, db, accumulate, workspace, s, &rc)) return rc;
    // (experiment, PCUDA_WGRAD3R=1: 3x3 layers with few channels on large maps without LDS staging, conv_wgrad3r.hip)
    if (wgrad3r_try(g, prec, x, dy, dy_sn, dy_sc, dw, db, accumulate, workspace, s, defer, &rc)) return rc;
    // aligned 3x3 / stride-1 layers: the fixed-geometry kernel (conv_wgrad3.hip)
    if (wgrad3_try(g, prec, x, dy, dy_sn, dy_sc, dw, db, accumulate, workspace, s, defer, &rc)) return rc;
    // 1x1 / stride-1 layers on whole 16-pixel steps: NT GEMM over pixels, operands straight from global memory (conv_wgrad1.hip)
    if (wgrad1_try(g, prec, x, dy, dy_sn, dy_sc, dw, db, accumulate, workspace, s, defer, &rc)) return rc;
  }
  const bool x3 = prec == PCUDA_PREC_BF16X3;
  WgradPlan w = plan_wgrad(g);
  TapSet t = fwd_taps(g);
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = *x; p.cin = g->cin;
  p.in_h = g->in_h; p.in_w = g->in_w; p.in_shift = g->in_up ? 1 : 0; p.in_row = g->in_w >> p.in_shift;
  p.dz = dy; p.dz_sn = dy_sn; p.dz_sc = dy_sc;
  p.cout = g->cout; p.out_h = g->out_h; p.out_w = g->out_w;
  p.stride = g->stride;
  p.ntaps = w.taps_per_group; p.ntaps_total = t.n; p.tap_groups = w.tap_groups;
  memcpy(p.dy, t.dy, sizeof(p.dy));
  memcpy(p.dx, t.dx, sizeof(p.dx));
  p.dy_min = t.dy_min; p.dx_min = t.dx_min;
  p.ih_t = w.ih_t; p.iw_t = w.iw_t;
  p.tw = w.tw; p.th = w.th; p.tmagic = 65536 / w.tw + 1; p.tiles_x = w.tiles_x; p.tiles_y = w.tiles_y; p.n = g->n;
  p.ksplit = w.ksplit; p.n_co_tiles = w.n_co_tiles; p.n_chunks = w.n_chunks;
  p.partial = (float*)workspace;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("PCUDA_DBG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
  }
  p.tw16 = ((w.tw & 15) == 0 && w.tw * w.th == 128) ? 1 : 0;
  p.aligned4 = ((g->out_w & 7) == 0 && (w.tw & 7) == 0 && (dy_sn & 3) == 0 && (dy_sc & 3) == 0 && (((uintptr_t)dy) & 15) == 0 &&
                ((long long)g->cout * dy_sc + (long long)g->out_h * g->out_w) * 4 < (1ll << 30)) ? 1 : 0;
  const long long welems = (long long)g->cout * g->cin * t.n;
  float* dbp = db ? (float*)workspace + (size_t)w.ksplit * welems : nullptr;

  const int full = w.ih_t * w.iw_t;
  const int clipped = (w.ih_t < g->in_h ? w.ih_t : g->in_h) * (w.iw_t < g->in_w ? w.iw_t : g->in_w) + 1;
  const size_t mul = x3 ? 2 : 1;
  const size_t zb = (size_t)w.co_tile * WG_ZROW * mul;
  bool clamp = clipped * 2 <= full;
  if (!clamp && (size_t)full * IG_REC_BYTES * mul + zb > (size_t)LDS_HARD) clamp = true;
  const int x_cap = clamp ? clipped : full;
  const size_t lds = (size_t)x_cap * IG_REC_BYTES * mul + zb;
  if (lds > (size_t)LDS_HARD) PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "conv2d_wgrad: tile %dx%d does not fit LDS", w.ih_t, w.iw_t);
  const dim3 grid(w.n_co_tiles * w.n_chunks * w.tap_groups, w.ksplit);
  {
    static int noxcd = -1;
    if (noxcd < 0) { const char* e = getenv("PCUDA_WG_NOXCD"); noxcd = (e && atoi(e)) ? 1 : 0; }
    const long long nblk = (long long)grid.x * grid.y;
    p.xcd_items = (!noxcd && nblk >= 16 && (nblk & 7) == 0) ? (int)(nblk / 8) : 0;
    static int nointer = -1;
    if (nointer < 0) { const char* e = getenv("PCUDA_WG_NOINTER"); nointer = (e && atoi(e)) ? 1 : 0; }
    p.xcd_slices = (p.xcd_items && !nointer && (w.ksplit & 7) == 0) ? w.ksplit / 8 : 0;
  }
  {
    const double flops = 2.0 * g->n * (double)g->out_h * g->out_w * g->cout * (double)g->cin * t.n;
    char tag[160];
    // software-pipelined variants hold the next tile's loads in registers: input tiles <= 256*PF pixels
    static int nopipe = -1;
    if (nopipe < 0) { const char* e = getenv("PCUDA_NOPIPE"); nopipe = (e && atoi(e)) ? 1 : 0; }
    int pf = (nopipe || !fast_src_ok(x, g->cin)) ? 0 : (x_cap <= 256 ? 1 : (x_cap <= 768 ? 3 : 0));
    // 16-tap groups (4x4 kernels): LDS leaves room for one workgroup per CU, so make it eight waves
    static int now8 = -1;
    if (now8 < 0) { const char* e = getenv("PCUDA_WG_NO8"); now8 = (e && atoi(e)) ? 1 : 0; }
    if (!now8 && pf > 0 && w.taps_per_group > 9 && x_cap <= 1024) pf = 100 + (x_cap <= 512 ? 1 : 2);
    {   // quad staging where the (row, quad) items of a tile fit the staging slots of this variant
      static int noxq = -1;
      if (noxq < 0) { const char* e = getenv("PCUDA_NOXQ"); noxq = e ? atoi(e) : 0; }
      const int slots = pf >= 100 ? pf - 100 : pf, per = pf >= 100 ? 128 : 64;
      const int ih = clamp ? (w.ih_t < g->in_h ? w.ih_t : g->in_h) : w.ih_t;
      const int in_w_phys = g->in_up ? g->in_w / 2 : g->in_w;
      p.xq = (!noxq && slots > 0 && !g->in_up && (in_w_phys & 3) == 0 && (ih * ((w.iw_t + 6) / 4) + per - 1) / per <= slots) ? 1 : 0;
    }
    snprintf(tag, sizeof(tag), "wgrad n%d cin%d cout%d %dx%d k%d s%d d%d ksplit%d clamp%d pf%d lds%zu", g->n, g->cin,
             g->cout, g->out_h, g->out_w, g->k, g->stride, g->dil, w.ksplit, clamp ? 1 : 0, pf, lds);
    ProfScope prof(PCUDA_FAM_CONV_WGRAD, flops, s, tag);
    int rc;
    const int taps_max = w.taps_per_group <= 1 ? 1 : w.taps_per_group <= 9 ? 9 : 16;
    rc = x3 ? wgrad_dispatch_x3(p, w.co_blks, clamp, taps_max, pf, x_cap, lds, dbp, grid, s)
            : wgrad_dispatch_bf16(p, w.co_blks, clamp, taps_max, pf, x_cap, lds, dbp, grid, s);
    if (rc) return rc;
  }
  {
    int nkg = 1;
    while (nkg < 16 && nkg * 2 <= w.ksplit) nkg <<= 1;
    if (defer) {      // the caller batches the reduce (pcuda_wgrad_reduce_batch); the workspace must live until then
      defer->partial = (const float*)workspace; defer->numel = welems; defer->ksplit = w.ksplit; defer->nkg = nkg;
      defer->dw = dw; defer->accumulate = accumulate; defer->ntaps = t.n;
      defer->db_partial = (const float*)dbp; defer->nb = db ? g->cout : 0; defer->db = db;
      return PCUDA_OK;
    }
    const unsigned nblk = (unsigned)cdiv(welems, 64) + (db ? (unsigned)cdiv(g->cout, 64) : 0u);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk), dim3(64 * nkg), 0, s, (const float*)workspace, welems, w.ksplit,
                       dw, accumulate, t.n, (const float*)dbp, (long long)(db ? g->cout : 0), db);
    PCUDA_CHECK_LAUNCH("wgrad_reduce_kernel");
  }
  return PCUDA_OK;
}

