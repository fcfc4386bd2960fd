// Anti-phase 3x3 / stride-1 / pad-1 convolution kernel (forward and data gradient of the segmenter's aligned layers,
// unet.py:23,27,116,122) -- round 6.
//
// ONE 512-thread workgroup per CU: waves 0-3 (group A) and waves 4-7 (group B) sit pairwise on the four SIMDs and run in
// ANTI-PHASE.  In every interval between two workgroup barriers one group is in its MFMA segment -- all nine taps of a
// 16-channel chunk, 108 MFMAs per wave, nothing but fragment reads and MFMAs -- while the partner group does the memory
// work of ITS tile: the epilogue of a finished tile, the commit of the next chunk's input (fp32 NCHW -> lazy BatchNorm
// affine -> bf16 hi / lo -> LDS), the next prefetch, half of the next weight chunk by LDS-DMA.  One barrier per interval:
// a stage (one group's chunk) costs two.  The two groups work on ADJACENT 32 x 8-pixel tiles of the same 64-row co-tile,
// so they share the weights: two 36-KiB buffers (chunk s is read while chunk s + 1 lands), filled by global_load_lds
// straight from a packed image that already is the LDS image (no VGPRs, no ds_write).
//
// LDS image (both operands): 64-byte records = 16 channels x (bf16 hi | bf16 lo): pieces 0 / 1 = hi k 0-7 / 8-15, pieces
// 2 / 3 = lo; piece q of record i sits in slot q ^ ((i >> 2) & 3): every ds_read_b128 fragment read over consecutive
// records (a 16-lane group covers 256 B) is conflict-free, and lo = hi ^ 32 in the address.
//   weights : [tap 9][row 64] records per (co-tile, chunk): 36,864 B, x 2 buffers
//   input   : haloed tile 10 x 34 pixels = 340 records per group (21,760 B), x 2 groups
//   scratch : 4 x 8 KiB (the memory group's transposed epilogue), partial sums, bias, the affine table: 157.5 KiB in all.
#pragma once
#include "common.h"

#define AP_HW 34
#define AP_NREC 340
#define AP_WBYTES 36864
#define AP_XBYTES 21760
#define AP_OFF_X (2 * AP_WBYTES)
#define AP_OFF_E (AP_OFF_X + 2 * AP_XBYTES)
#define AP_OFF_RED (AP_OFF_E + 32768)
#define AP_OFF_TAB (AP_OFF_RED + 4096)
#define AP_MAX_C 512
#define AP_OFF_AFF (AP_OFF_TAB + 2 * AP_MAX_C * 4)
#define AP_LDS_BYTES (AP_OFF_AFF + 2 * AP_MAX_C * 4)   // 162,304 of 163,840

struct ApParams {
  pcuda_src x;
  int cin;                 // reduction channels: a multiple of 16; x.c1 a multiple of 16 (a chunk lies in one source)
  int H, W;                // H % 8 == 0, W % 32 == 0
  pcuda_dst y;
  int cout;                // rows: a multiple of 64 (y.c1 a multiple of 4)
  const unsigned char* wimg;   // [n_co_tiles][cin / 16][AP_WBYTES]
  const float* bias;
  float slope;
  int accumulate;
  float* stats;            // [tile][cout][2] or null
  const float* red_a; long long red_sn, red_sc;
  const float* red_mean; const float* red_invstd;
  int tiles_x, tiles_y, n;
  int n_co_tiles, nchunks;
  int total;               // (tile pairs) x (co tiles)
  int up;                  // 1: the stored input is (H / 2) x (W / 2), read through the nearest x2 fold (unet.py:111): the LDS tile holds
                           //    the STORED pixels (6 x 18 records), the fragment addresses map a logical pixel to its record
  int out_w;               // FOLD: row length of the half-resolution output
  int dbg;                 // timing experiments: 1 no MFMA, 2 no epilogue, 4 no commit, 8 no weight DMA
  unsigned long long* dbg_clk;
};

typedef unsigned int ap_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 ap_frag(const unsigned char* p) { return __builtin_bit_cast(bf16x8, *(const uint4*)p); }
// LDS-DMA of 1 KiB per wave instruction (lane l copies 16 bytes from its own address to lds_wave_base + 16 l).  By inline
// assembly: behind the builtin the compiler's waitcnt pass puts s_waitcnt vmcnt(0) in front of the next LDS read of ANY
// address (it cannot tell the epilogue's scratch from the weight buffer) -- the memory segment then sat out the DMA's
// latency before its epilogue instead of beside it.  The readers are ordered by the counted vmcnt wait + barrier at the end
// of the segment.
__device__ __forceinline__ void ap_dma16(const unsigned char* g, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_wave_base), "v"(g) : "memory", "m0");
}
__device__ __forceinline__ void ap_split8(const float* v, ap_u32x4& hi4, ap_u32x4& lo4) {
  uint32_t a, b, c, d, e, f, g, hh;
  split2(v[0], v[1], a, b); split2(v[2], v[3], c, d); split2(v[4], v[5], e, f); split2(v[6], v[7], g, hh);
  hi4 = ap_u32x4{a, c, e, g};
  lo4 = ap_u32x4{b, d, f, hh};
}
// workgroup barrier that waits for this wave's LDS operations only (__syncthreads() also drains vmcnt: the prefetch)
#define AP_BARRIER()                                     \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    asm volatile("" ::: "memory");                       \
  } while (0)

// ds_read_b128 by inline assembly: the compiler's waitcnt pass put s_waitcnt lgkmcnt(0) behind every second tap's reads (the
// reads just issued for the NEXT tap included); the waits of the MFMA segment are placed by hand (AP_WAIT_FRAGS)
#define AP_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// wait until at most N LDS operations are outstanding; the fragment set that is complete then is tied to the wait
#define AP_WAIT_FRAGS(N, fa, fb)                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                          \
               : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]))

// Stage cursor, one scalar register: chunk [0:6) | co tile [6:10) | tile column [10:14) | tile row [14:20) | image [20:32)
#define AP_CUR(chunk, cot, txi, tyi, n) ((unsigned)(chunk) | ((unsigned)(cot) << 6) | ((unsigned)(txi) << 10) | ((unsigned)(tyi) << 14) | ((unsigned)(n) << 20))
#define AP_CUR_CHUNK(c) ((int)((c) & 63u))
#define AP_CUR_COT(c) ((int)(((c) >> 6) & 15u))
#define AP_CUR_X0(c) ((int)(((c) >> 10) & 15u) * 32)
#define AP_CUR_Y0(c) ((int)(((c) >> 14) & 63u) * 8)
#define AP_CUR_N(c) ((int)((c) >> 20))

// STATS: 0 none, 1 BatchNorm partial sums of the stored values, 2 BatchNorm-backward reduce partials (dgrad)
template <int STATS, bool ACC = false, bool DBG = false, bool EXP = false, bool FOLD = false>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(224))) void conv3ap_kernel(const ApParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wv >> 2, w = wv & 3;
  const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS unsigned char*)smem;
  unsigned char* const Xs = smem + AP_OFF_X + g * AP_XBYTES;
  const unsigned xs_lds = lds0 + AP_OFF_X + g * AP_XBYTES;
  float* const tsc = (float*)(smem + AP_OFF_E) + w * 2048;          // [32 rows][64 pixels], the memory group's
  float* const sred = (float*)(smem + AP_OFF_RED) + g * 512;        // [4 waves][64][2]
  float* const stab1 = (float*)(smem + AP_OFF_TAB);                 // per row: bias (forward) or mean (STATS == 2)
  float* const stab2 = stab1 + AP_MAX_C;                            // per row: invstd (STATS == 2)
  float* const ssc = (float*)(smem + AP_OFF_AFF);
  float* const ssh = ssc + AP_MAX_C;
  // the parameters of the prefetch, in registers once (selecting between FIELDS of the by-value argument inside the loop
  // compiled to vector loads from the kernel-argument segment with an s_waitcnt vmcnt(0) behind them)
  const int H = p.H, W = p.W, nchunks = p.nchunks, xc1 = p.x.c1, dbg = (DBG || EXP) ? p.dbg : 0, up = p.up;
  const float* const xp1 = p.x.p1; const float* const xp2 = p.x.p2 ? p.x.p2 : p.x.p1;
  const long long xsn1 = p.x.sn1, xsn2 = p.x.sn2;
  const int xsc1 = (int)p.x.sc1, xsc2 = (int)p.x.sc2;
  const unsigned char* const wimg = p.wimg;

  // XCD-aware persistent schedule (as igemm_pipe_kernel): the 8 XCDs own contiguous eighths of the item list
  const int nx = min(8, (int)gridDim.x);
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int gx = ((int)gridDim.x - xcd + nx - 1) / nx;
  const int lo = (int)((long long)p.total * xcd / nx), hi = (int)((long long)p.total * (xcd + 1) / nx);
  const int L0 = lo + slot;
  const int nitems = L0 < hi ? (hi - L0 + gx - 1) / gx : 0;
  const int S = nitems * nchunks;                                    // stages of each group
  if (S == 0) return;

  // ---- the lazy-BatchNorm affine of every reduction channel (identity where a source has none)
  for (int c = tid; c < p.cin; c += 512) {
    const bool first = c < xc1;
    const float* scp = first ? p.x.scale1 : p.x.scale2;
    const float* shp = first ? p.x.shift1 : p.x.shift2;
    const int cc = first ? c : c - xc1;
    ssc[c] = scp ? scp[cc] : 1.f;
    ssh[c] = scp ? shp[cc] : 0.f;
  }
  // ---- per-row epilogue constants (read by the four waves of a group, which meet at no barrier inside a segment)
  for (int c = tid; c < p.cout; c += 512) {
    stab1[c] = STATS == 2 ? p.red_mean[c] : (p.bias ? p.bias[c] : 0.f);
    stab2[c] = STATS == 2 ? p.red_invstd[c] : 0.f;
  }

  // ---- tile-invariant fragment addresses (LDS byte addresses)
  unsigned xa[2][9];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ly = 2 * w + pb + t / 3, lx = r + (t % 3);                      // haloed logical tile: 10 x 34
      const int pp = up ? ((ly + 1) >> 1) * 18 + ((lx + 1) >> 1) : ly * AP_HW + lx;  // (stored tile: 6 x 18, origin ((y0 - 1) >> 1, (x0 - 1) >> 1))
      xa[pb][t] = xs_lds + pp * 64 + ((h ^ ((pp >> 2) & 3)) << 4);
    }
  const unsigned wa = lds0 + r * 64 + ((h ^ ((r >> 2) & 3)) << 4);

  // ---- tile-invariant staging plan: unit u = (halo row 0..9, channel half, aligned quad -1..8) of the group's 256 lanes
  // (200 units): eight float4 loads (one per channel of the half) of the quad x0 + 4 q .. + 3; consecutive lanes read
  // consecutive quads (160-byte runs).  Quads 0..7 commit four pixels, quad -1 its last (tile column 0), quad 8 its first
  // (tile column 33).
  // (nearest-x2 fold: the stored tile, 6 rows x 6 quads -1..4 x 2 halves = 72 units; quad -1 gives tile column 0, quad 4 column 17)
  const int u = w * 64 + lane;
  const int nq = up ? 6 : 10, nun = up ? 72 : 200, tpitch = up ? 18 : AP_HW, lastcol = up ? 17 : 33;
  const bool uact = u < nun;
  const int uu = min(u, nun - 1);
  const int urow = uu / (2 * nq), uhalf = (uu % (2 * nq)) / nq, uq = uu % nq - 1;
  const unsigned emask = !uact ? 0u : (uq < 0 ? 8u : (uq > nq - 3 ? 1u : 15u));
  unsigned xw[4];                // LDS byte offset (in Xs) of pixel e's hi piece (lo = ^ 32)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int pix = urow * tpitch + min(max(1 + 4 * uq + e, 0), lastcol);
    xw[e] = pix * 64 + ((uhalf ^ ((pix >> 2) & 3)) << 4);
  }
  const int Hs = up ? H >> 1 : H, Ws = up ? W >> 1 : W;   // the stored plane

  // stage cursors (wave-uniform, one register each).  nw: the next stage to prefetch; pw: the stage whose input is in
  // flight; cw: the stage this group computes next (committed in the memory segment in front of it); dw: the tile whose
  // accumulators wait for their epilogue.
  int nL = L0;
  auto decode = [&](int L) -> unsigned {     // chunk 0 of item L (past the end: the last item again)
    const int Lc = min(L, hi - 1);
    const int pp = Lc / p.n_co_tiles;
    const int cot = Lc - pp * p.n_co_tiles;
    const int pt = 2 * pp + g;
    const int tmp = pt / p.tiles_x;
    const int txi = pt - tmp * p.tiles_x;
    const int n = tmp / p.tiles_y;
    const int tyi = tmp - n * p.tiles_y;
    return AP_CUR(0, cot, txi, tyi, n);
  };
  unsigned nw = decode(nL), pw, cw, dw = 0;
  auto advance_n = [&]() {
    if (AP_CUR_CHUNK(nw) + 1 == nchunks) { nL += gx; nw = decode(nL); } else { nw += 1; }
  };

  // The prefetched unit (channel j of the half, 4 consecutive pixels: 8 float4) lands in v[224:255] -- registers the
  // compiler never allocates (amdgpu_num_vgpr(224) on the kernel) --, loaded by inline assembly and read back with v_mov
  // behind a hand-placed counted wait.  The loads are in flight
  // across two barrier intervals; registers the compiler manages cannot hold them: with compiler-visible loads the allocator
  // landed them in a second set of 32 registers and every later reuse of that set waited for the prefetch (and, the DMA
  // being invisible to its counters, for the weight pieces); with "+v"-tied assembly loads it MOVED the still-empty
  // registers between the load and the wait; with AGPRs as the landing zone it split the register file 128 / 128 and spilled.
  bool xin = false;              // the unit in flight lies inside the image
  auto x_issue = [&](unsigned cur) {
    const int c0 = AP_CUR_CHUNK(cur) * 16;
    const int n = AP_CUR_N(cur);
    const bool first = c0 < xc1;
    const float* base = first ? xp1 + (long long)n * xsn1 : xp2 + (long long)n * xsn2;
    const int sc = first ? xsc1 : xsc2;
    const int cl0 = first ? c0 : c0 - xc1;
    const int gy = (up ? (AP_CUR_Y0(cur) >> 1) : AP_CUR_Y0(cur)) - 1 + urow, gxx = (up ? (AP_CUR_X0(cur) >> 1) : AP_CUR_X0(cur)) + 4 * uq;
    xin = ((unsigned)gy < (unsigned)Hs) & ((unsigned)gxx < (unsigned)Ws);
    const int cy = min(max(gy, 0), Hs - 1), cx = min(max(gxx, 0), Ws - 4);
    const unsigned voff = (unsigned)((uhalf * 8) * sc + cy * Ws + cx) * 4u;
    const char* b0 = (const char*)(base + (long long)cl0 * sc) + voff;
    const long long cs4 = (long long)sc * 4;
    asm volatile("global_load_dwordx4 v[224:227], %0, off" ::"v"(b0) : "memory", "v224", "v225", "v226", "v227");
    asm volatile("global_load_dwordx4 v[228:231], %0, off" ::"v"(b0 + cs4) : "memory", "v228", "v229", "v230", "v231");
    asm volatile("global_load_dwordx4 v[232:235], %0, off" ::"v"(b0 + 2 * cs4) : "memory", "v232", "v233", "v234", "v235");
    asm volatile("global_load_dwordx4 v[236:239], %0, off" ::"v"(b0 + 3 * cs4) : "memory", "v236", "v237", "v238", "v239");
    asm volatile("global_load_dwordx4 v[240:243], %0, off" ::"v"(b0 + 4 * cs4) : "memory", "v240", "v241", "v242", "v243");
    asm volatile("global_load_dwordx4 v[244:247], %0, off" ::"v"(b0 + 5 * cs4) : "memory", "v244", "v245", "v246", "v247");
    asm volatile("global_load_dwordx4 v[248:251], %0, off" ::"v"(b0 + 6 * cs4) : "memory", "v248", "v249", "v250", "v251");
    asm volatile("global_load_dwordx4 v[252:255], %0, off" ::"v"(b0 + 7 * cs4) : "memory", "v252", "v253", "v254", "v255");
  };
#define AP_WAIT_X(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  // pixel e of the landed unit: channels 0..7 of the half
#define AP_XREAD1(e, vv, r0, r1, r2, r3, r4, r5, r6, r7)                                                                \
  asm volatile("v_mov_b32 %0, v" #r0 "\n\tv_mov_b32 %1, v" #r1 "\n\tv_mov_b32 %2, v" #r2 "\n\tv_mov_b32 %3, v" #r3        \
               "\n\tv_mov_b32 %4, v" #r4 "\n\tv_mov_b32 %5, v" #r5 "\n\tv_mov_b32 %6, v" #r6 "\n\tv_mov_b32 %7, v" #r7   \
               : "=v"(vv[0]), "=v"(vv[1]), "=v"(vv[2]), "=v"(vv[3]), "=v"(vv[4]), "=v"(vv[5]), "=v"(vv[6]), "=v"(vv[7]))
#define AP_XREAD_0(vv) AP_XREAD1(0, vv, 224, 228, 232, 236, 240, 244, 248, 252)
#define AP_XREAD_1(vv) AP_XREAD1(1, vv, 225, 229, 233, 237, 241, 245, 249, 253)
#define AP_XREAD_2(vv) AP_XREAD1(2, vv, 226, 230, 234, 238, 242, 246, 250, 254)
#define AP_XREAD_3(vv) AP_XREAD1(3, vv, 227, 231, 235, 239, 243, 247, 251, 255)
  auto x_commit = [&](unsigned cur) {   // behind AP_WAIT_X
    const int c0 = AP_CUR_CHUNK(cur) * 16 + uhalf * 8;
    const f32x4 s0 = *(const f32x4*)(ssc + c0), s1 = *(const f32x4*)(ssc + c0 + 4);
    const f32x4 t0 = *(const f32x4*)(ssh + c0), t1 = *(const f32x4*)(ssh + c0 + 4);
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // zero padding is applied AFTER the affine: outside the image both terms are 0
      sc[j] = xin ? s0[j] : 0.f; sc[4 + j] = xin ? s1[j] : 0.f;
      sh[j] = xin ? t0[j] : 0.f; sh[4 + j] = xin ? t1[j] : 0.f;
    }
#define AP_COMMIT_PIXEL(e)                                         \
  {                                                                \
    float xr[8], vv[8];                                            \
    AP_XREAD_##e(xr);                                              \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) vv[j] = fmaf(xr[j], sc[j], sh[j]); \
    ap_u32x4 hi4, lo4;                                             \
    ap_split8(vv, hi4, lo4);                                       \
    if (emask & (1u << e)) {                                       \
      *(ap_u32x4*)(Xs + xw[e]) = hi4;                              \
      *(ap_u32x4*)(Xs + (xw[e] ^ 32)) = lo4;                       \
    }                                                              \
  }
    AP_COMMIT_PIXEL(0) AP_COMMIT_PIXEL(1) AP_COMMIT_PIXEL(2) AP_COMMIT_PIXEL(3)
#undef AP_COMMIT_PIXEL
  };
  f32x16 acc[2][2];
  unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  if (DBG) tlast = __builtin_readcyclecounter();
#define AP_CLK(i)                                                   \
  if (DBG) {                                                        \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    clk[i] += now_ - tlast; tlast = now_;                           \
    __builtin_amdgcn_sched_barrier(0);                              \
  }

  // ---- prologue: weight chunk 0 (36 pieces over the 8 waves), A's first input tile, both groups' first prefetch
  {
    const unsigned char* src = wimg + ((long long)AP_CUR_COT(nw) * nchunks) * AP_WBYTES + lane * 16;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int piece = min(wv + 8 * j, 35);
      ap_dma16(src + piece * 1024, lds0 + piece * 1024);
    }
  }
  cw = pw = nw;                  // stage 0
  x_issue(nw);
  advance_n();
  AP_WAIT_X(0);
  __syncthreads();               // the affine table
  if (g == 0) {                  // A: stage 0 committed, stage 1 in flight; B: stage 0 in flight
    x_commit(cw);
    pw = nw;
    x_issue(nw);
    advance_n();
  }
  AP_BARRIER();
  int sidx = 0;                  // index of the stage this group computes next (= cw)
  bool epi_pending = false, stat_pending = false;

  for (int ph = 0; ph <= 2 * S; ++ph) {
    AP_CLK(0)
    if ((ph & 1) == g) {
      // ======================= MFMA segment: stage sidx = cw =======================
      if (sidx < S && !(dbg & 1)) {
        const unsigned wb = wa + (sidx & 1) * AP_WBYTES, wbl = wb ^ 32;
        bf16x8 fa[2][4], fb[2][4];
        // fa: 0 / 1 = hi rows 0-31 / 32-63, 2 / 3 = lo; fb: 0 / 1 = hi pixel block 0 / 1, 2 / 3 = lo
#define AP_LOAD(bi, t)                                   \
  AP_DSR(fa[bi][0], wb, (t) * 4096);                     \
  AP_DSR(fa[bi][1], wb, (t) * 4096 + 2048);              \
  AP_DSR(fb[bi][0], xa[0][t], 0);                        \
  AP_DSR(fb[bi][1], xa[1][t], 0);                        \
  AP_DSR(fa[bi][2], wbl, (t) * 4096);                    \
  AP_DSR(fa[bi][3], wbl, (t) * 4096 + 2048);             \
  AP_DSR(fb[bi][2], xa[0][t] ^ 32u, 0);                  \
  AP_DSR(fb[bi][3], xa[1][t] ^ 32u, 0);
        // One tap: 12 MFMAs on fragment set bi, and BETWEEN the first eight of them the eight fragment reads of the next tap
        // into set bn, one per MFMA gap.  (Issued as a burst in front of the tap, the reads kept the wave away from the matrix
        // pipe for ~65 cycles per tap: 4.35 k cycles per segment instead of 3.46 k.)  MFMA order: the three products round
        // robin over the four accumulators (lo x hi, hi x lo, hi x hi: the order of igemm_pipe_kernel).
#define AP_M(cb, pb, A, B, C) acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
#define AP_R(stmt) stmt; __builtin_amdgcn_sched_barrier(0);
#define AP_TAP(bi, bn, tn, Z, EXTRA)                                                        \
  AP_M(0, 0, fa[bi][2], fb[bi][0], Z ? zero16 : acc[0][0]) AP_R(AP_DSR(fa[bn][2], wbl, (tn) * 4096))            \
  AP_M(0, 1, fa[bi][2], fb[bi][1], Z ? zero16 : acc[0][1]) AP_R(AP_DSR(fb[bn][0], xa[0][tn], 0))                \
  AP_M(1, 0, fa[bi][3], fb[bi][0], Z ? zero16 : acc[1][0]) AP_R(AP_DSR(fb[bn][1], xa[1][tn], 0))                \
  AP_M(1, 1, fa[bi][3], fb[bi][1], Z ? zero16 : acc[1][1]) AP_R(AP_DSR(fa[bn][3], wbl, (tn) * 4096 + 2048))     \
  AP_M(0, 0, fa[bi][0], fb[bi][2], acc[0][0]) AP_R(AP_DSR(fa[bn][0], wb, (tn) * 4096))                          \
  AP_M(0, 1, fa[bi][0], fb[bi][3], acc[0][1]) AP_R(AP_DSR(fb[bn][2], xa[0][tn] ^ 32u, 0))                       \
  AP_M(1, 0, fa[bi][1], fb[bi][2], acc[1][0]) AP_R(AP_DSR(fb[bn][3], xa[1][tn] ^ 32u, 0))                       \
  AP_M(1, 1, fa[bi][1], fb[bi][3], acc[1][1]) AP_R(AP_DSR(fa[bn][1], wb, (tn) * 4096 + 2048))                   \
  AP_M(0, 0, fa[bi][0], fb[bi][0], acc[0][0]) EXTRA                                                              \
  AP_M(0, 1, fa[bi][0], fb[bi][1], acc[0][1])                                                                    \
  AP_M(1, 0, fa[bi][1], fb[bi][0], acc[1][0])                                                                    \
  AP_M(1, 1, fa[bi][1], fb[bi][1], acc[1][1])                                                                    \
  AP_WAIT_FRAGS(0, fa[bn], fb[bn]);                                                                              \
  __builtin_amdgcn_sched_barrier(0);
#define AP_TAP_LAST(bi, Z)                                                                  \
  AP_M(0, 0, fa[bi][2], fb[bi][0], acc[0][0]) AP_M(0, 1, fa[bi][2], fb[bi][1], acc[0][1])   \
  AP_M(1, 0, fa[bi][3], fb[bi][0], acc[1][0]) AP_M(1, 1, fa[bi][3], fb[bi][1], acc[1][1])   \
  AP_M(0, 0, fa[bi][0], fb[bi][2], acc[0][0]) AP_M(0, 1, fa[bi][0], fb[bi][3], acc[0][1])   \
  AP_M(1, 0, fa[bi][1], fb[bi][2], acc[1][0]) AP_M(1, 1, fa[bi][1], fb[bi][3], acc[1][1])   \
  AP_M(0, 0, fa[bi][0], fb[bi][0], acc[0][0]) AP_M(0, 1, fa[bi][0], fb[bi][1], acc[0][1])   \
  AP_M(1, 0, fa[bi][1], fb[bi][0], acc[1][0]) AP_M(1, 1, fa[bi][1], fb[bi][1], acc[1][1])
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // Group A brings the next weight chunk W(sidx + 1) -- the chunk of the stage whose input it has in flight -- into the
        // buffer both groups left an interval ago, ONE 1-KiB piece per wave and tap, from INSIDE its MFMA stream.  (Round 6
        // measurement: next to a partner that issues MFMAs back to back a wave gets a vector-memory instruction through only
        // every ~390 cycles -- nine DMA pieces cost the memory group 3.3 k cycles per segment, with or without other memory
        // traffic, priority or not; in the MFMA wave's own stream an instruction waits for nobody.)
        const unsigned char* wsrc = wimg + ((long long)AP_CUR_COT(pw) * nchunks + AP_CUR_CHUNK(pw)) * AP_WBYTES + lane * 16 + w * 1024;
        const unsigned wdst = lds0 + ((sidx + 1) & 1) * AP_WBYTES + w * 1024;
        const bool dma_here = g == 0;
#define AP_DMA_TAP(j) if (dma_here) { ap_dma16(wsrc + (j) * 4096, wdst + (j) * 4096); } __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
        AP_LOAD(0, 0)
        AP_WAIT_FRAGS(0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (AP_CUR_CHUNK(cw) == 0) { AP_TAP(0, 1, 1, true, AP_DMA_TAP(0)) } else { AP_TAP(0, 1, 1, false, AP_DMA_TAP(0)) }
        AP_TAP(1, 0, 2, false, AP_DMA_TAP(1))
        // (the scalar bookkeeping of the segment sits between the MFMAs, where it issues for free)
        if (STATS && stat_pending && w == 0) {   // the partial sums of the epilogue that ended at the last barrier
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) { a += sred[(ww * 64 + lane) * 2 + 0]; b += sred[(ww * 64 + lane) * 2 + 1]; }
          const int pt = (AP_CUR_N(dw) * p.tiles_y + (AP_CUR_Y0(dw) >> 3)) * p.tiles_x + (AP_CUR_X0(dw) >> 5);
          float* dst = p.stats + ((long long)pt * p.cout + AP_CUR_COT(dw) * 64 + lane) * 2;
          *(float2*)dst = make_float2(a, b);
        }
        stat_pending = false;
        AP_TAP(0, 1, 3, false, AP_DMA_TAP(2)) AP_TAP(1, 0, 4, false, AP_DMA_TAP(3))
        AP_TAP(0, 1, 5, false, AP_DMA_TAP(4)) AP_TAP(1, 0, 6, false, AP_DMA_TAP(5))
        AP_TAP(0, 1, 7, false, AP_DMA_TAP(6)) AP_TAP(1, 0, 8, false, AP_DMA_TAP(7))
        AP_DMA_TAP(8)
        AP_TAP_LAST(0, false)
      } else if (STATS && stat_pending && w == 0) {   // (A's interval behind its last epilogue)
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { a += sred[(ww * 64 + lane) * 2 + 0]; b += sred[(ww * 64 + lane) * 2 + 1]; }
        const int pt = (AP_CUR_N(dw) * p.tiles_y + (AP_CUR_Y0(dw) >> 3)) * p.tiles_x + (AP_CUR_X0(dw) >> 5);
        float* dst = p.stats + ((long long)pt * p.cout + AP_CUR_COT(dw) * 64 + lane) * 2;
        *(float2*)dst = make_float2(a, b);
        stat_pending = false;
      }
      if (DBG) asm volatile("s_nop 0" ::"v"(acc[0][0][15]), "v"(acc[1][1][15]));
      AP_CLK(1)
      if (sidx < S) {
        epi_pending = AP_CUR_CHUNK(cw) == nchunks - 1;
        dw = epi_pending ? cw : dw;
        cw = pw;                 // the next stage this group computes is the one whose prefetch is in flight
        ++sidx;
      }
      AP_CLK(7)
    } else {
      // ======================= memory segment: everything `cw` needs, the finished tile's epilogue =======================
      // the prefetched unit of `cw` has had a whole MFMA segment to land (and every older store with it)
      AP_WAIT_X(0);
      if (!(dbg & 4) && sidx < S) x_commit(cw);
      AP_CLK(4)
      if (epi_pending && !(dbg & 2)) {
        // Transposed epilogue (as igemm_pipe_kernel's): the wave turns its [32 rows][64 pixels] block through its own
        // scratch so that a lane holds 4 consecutive pixels of one channel: 16-byte stores of whole 128-byte row
        // segments, BatchNorm partial sums from one 16-lane DPP reduction per channel.
        const int co0 = AP_CUR_COT(dw) * 64, dn = AP_CUR_N(dw);
        if constexpr (FOLD) {
          // 2x2 fold (the data gradient of a layer whose input was read through nearest x2, unet.py:111-112: the gradient of the
          // STORED half-resolution tensor is the sum over each 2x2 block of the logical one; as igemm_pipe_kernel's FOLD
          // epilogue).  A wave's two pixel blocks are tile rows 2 w and 2 w + 1: the vertical pair is one add per accumulator
          // register; the row of sums goes through the wave's scratch ([32 channels][32 pixels]) so that a lane holds 4
          // consecutive pixels of one channel, whose two horizontal pairs it stores as 8 bytes.
          const int q = lane & 7, cs = lane >> 3;
          const int ly = AP_CUR_Y0(dw) + 2 * w, lx = AP_CUR_X0(dw) + 4 * q;
          const unsigned pixq = (unsigned)((ly >> 1) * p.out_w + (lx >> 1)) * 4u;
          const int c1 = min(p.y.c1, p.cout);
          const char* const yb1 = (const char*)(p.y.p1 + (long long)dn * p.y.sn1);
          const char* const yb2 = (const char*)(p.y.p2 + (long long)dn * p.y.sn2);
          const unsigned pl1 = (unsigned)p.y.sc1 * 4u, pl2 = (unsigned)p.y.sc2 * 4u;
          const unsigned voff1 = (unsigned)cs * pl1 + pixq, voff2 = (unsigned)cs * pl2 + pixq;
          const char* const ab = STATS == 2 ? (const char*)(p.red_a + (long long)dn * p.red_sn) : nullptr;
          const unsigned aoff = STATS == 2 ? (unsigned)cs * (unsigned)p.red_sc * 4u + pixq : 0u;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) tsc[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[cb][0][i] + acc[cb][1][i];
            __builtin_amdgcn_wave_barrier();
            f32x4 v[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) v[it] = *(const f32x4*)(tsc + (it * 8 + cs) * 32 + 4 * q);
            float s1[4], s2[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int cu = co0 + cb * 32 + it * 8;               // uniform; the 8 channels lie in one destination
              const bool first = cu < c1;
              const char* const sb = first ? yb1 : yb2;
              const unsigned plane = first ? pl1 : pl2;
              const unsigned rel = (unsigned)(first ? cu : cu - c1);
              char* const dptr = (char*)sb + (unsigned long long)rel * plane + (first ? voff1 : voff2);
              const float o0 = v[it][0] + v[it][1], o1 = v[it][2] + v[it][3];
              float a0 = 0.f, a1 = 0.f;
              if (STATS == 2) {
                const float2 av = *(const float2*)(ab + (long long)cu * p.red_sc * 4 + aoff);
                a0 = av.x; a1 = av.y;
              }
              *(float2*)dptr = make_float2(o0, o1);
              if (STATS == 2) {
                const float m = stab1[cu + cs], is = stab2[cu + cs];
                s1[it] = o0 + o1;
                s2[it] = o0 * ((a0 - m) * is) + o1 * ((a1 - m) * is);
              }
            }
            if (STATS == 2) {
#pragma unroll
              for (int it = 0; it < 4; ++it) { s1[it] = row_sum<8>(s1[it]); s2[it] = row_sum<8>(s2[it]); }
              if (q == 0) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
                  *(float2*)(sred + (w * 64 + cb * 32 + it * 8 + cs) * 2) = make_float2(s1[it], s2[it]);
              }
            }
            __builtin_amdgcn_wave_barrier();
          }
          stat_pending = STATS != 0;
        } else {
        const int q = lane & 15, cs = lane >> 4;
        const int ly = AP_CUR_Y0(dw) + 2 * w + (q >> 3), lx = AP_CUR_X0(dw) + 4 * (q & 7);
        const unsigned pixq = (unsigned)(ly * W + lx) * 4u;
        const int c1 = min(p.y.c1, p.cout);
        const char* const yb1 = (const char*)(p.y.p1 + (long long)dn * p.y.sn1);
        const char* const yb2 = (const char*)(p.y.p2 + (long long)dn * p.y.sn2);
        const unsigned pl1 = (unsigned)p.y.sc1 * 4u, pl2 = (unsigned)p.y.sc2 * 4u;
        const unsigned voff1 = (unsigned)cs * pl1 + pixq, voff2 = (unsigned)cs * pl2 + pixq;   // per lane; the plane base is scalar
        const float slope = p.slope;
        const char* const ab = STATS == 2 ? (const char*)(p.red_a + (long long)dn * p.red_sn) : nullptr;
        const unsigned aoff = STATS == 2 ? (unsigned)cs * (unsigned)p.red_sc * 4u + pixq : 0u;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
              tsc[((i & 3) + 8 * (i >> 2) + 4 * h) * 64 + pb * 32 + r] = acc[cb][pb][i];
          __builtin_amdgcn_wave_barrier();
          f32x4 v[8];
          float bia[8];
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            v[it] = *(const f32x4*)(tsc + (it * 4 + cs) * 64 + 4 * q);
            bia[it] = STATS == 2 ? 0.f : stab1[co0 + cb * 32 + it * 4 + cs];
          }
#pragma unroll
          for (int ib = 0; ib < 8; ib += 4) {
            f32x4 av[4], o[4];
            char* dptr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int cu = co0 + cb * 32 + (ib + k) * 4;         // uniform; the 4 channels lie in one destination
              const bool first = cu < c1;                          // (selects on scalars: no branch per store)
              const char* const sb = first ? yb1 : yb2;
              const unsigned plane = first ? pl1 : pl2;
              const unsigned rel = (unsigned)(first ? cu : cu - c1);
              dptr[k] = (char*)sb + (unsigned long long)rel * plane + (first ? voff1 : voff2);
              if (STATS == 2) av[k] = *(const f32x4*)(ab + (long long)cu * p.red_sc * 4 + aoff);
              if (ACC) o[k] = *(const f32x4*)dptr[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float t = v[ib + k][e] + bia[ib + k];
                const float a = fmaxf(t, t * slope);               // LeakyReLU for 0 <= slope <= 1 (the launcher checks): the
                v[ib + k][e] = ACC ? a + o[k][e] : a;              // same value as t > 0 ? t : t * slope, one instruction less
              }
              if (!((DBG || EXP) && (dbg & 16))) *(f32x4*)dptr[k] = v[ib + k];
            }
            if (STATS && !((DBG || EXP) && (dbg & 32))) {
              float s1[4], s2[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const f32x4 t = v[ib + k];
                s1[k] = (t[0] + t[1]) + (t[2] + t[3]);
                if (STATS == 2) {
                  const int cu = co0 + cb * 32 + (ib + k) * 4 + cs;
                  const float m = stab1[cu], is = stab2[cu];
                  s2[k] = (t[0] * ((av[k][0] - m) * is) + t[1] * ((av[k][1] - m) * is)) +
                          (t[2] * ((av[k][2] - m) * is) + t[3] * ((av[k][3] - m) * is));
                } else {
                  s2[k] = (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
                }
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) { s1[k] = row_sum<16>(s1[k]); s2[k] = row_sum<16>(s2[k]); }
              if (q == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const int row = cb * 32 + (ib + k) * 4 + cs;
                  *(float2*)(sred + (w * 64 + row) * 2) = make_float2(s1[k], s2[k]);
                }
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        stat_pending = STATS != 0;
        }
      }
      epi_pending = false;
      AP_CLK(3)
      AP_CLK(2)
      // the prefetch of the stage after `cw` (past the end: a valid tile again, so that the counted wait stays exact)
      if (!(dbg & 4)) {
        pw = nw;
        x_issue(nw);
        advance_n();
      }
      AP_CLK(5)
    }
    AP_BARRIER();
    AP_CLK(6)
  }
  if (STATS && stat_pending && w == 0) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { a += sred[(ww * 64 + lane) * 2 + 0]; b += sred[(ww * 64 + lane) * 2 + 1]; }
    const int pt = (AP_CUR_N(dw) * p.tiles_y + (AP_CUR_Y0(dw) >> 3)) * p.tiles_x + (AP_CUR_X0(dw) >> 5);
    float* dst = p.stats + ((long long)pt * p.cout + AP_CUR_COT(dw) * 64 + lane) * 2;
    *(float2*)dst = make_float2(a, b);
  }
  AP_WAIT_X(0);   // (the unconditional tail prefetch)
  if (DBG && lane == 0 && w == 0 && p.dbg_clk) {
    for (int i = 0; i < 8; ++i) atomicAdd(&p.dbg_clk[g * 8 + i], clk[i]);
  }
}

// ------------------------------------------------------------------------------------------
// packed weight image of the anti-phase kernel: [co tile][16-channel chunk][tap][row 0..63][64 B], the record's pieces
// swizzled as in LDS (the image IS the LDS image: the kernel copies it 1 KiB per wave instruction).  One thread = one
// 16-byte piece.  flip: tap t reads source tap 8 - t (the data gradient's role-swapped, mirrored weights).
// ------------------------------------------------------------------------------------------
struct ApPackParams {
  const float* w;
  long long s_row, s_red;  // element strides of the row (M) and reduction (K) index; the 9 taps are contiguous
  int rows, red, flip;
  unsigned char* out;
};
__global__ void ap_pack_kernel(const ApPackParams p) {
  const int nch = p.red >> 4, ncot = p.rows >> 6;
  const long long total = (long long)ncot * nch * 9 * 64 * 4;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx & 3);
    long long rest = idx >> 2;
    const int row = (int)(rest & 63); rest >>= 6;
    const int t = (int)(rest % 9); rest /= 9;
    const int ch = (int)(rest % nch);
    const int cot = (int)(rest / nch);
    const int ts = p.flip ? 8 - t : t;
    const float* src = p.w + (long long)(cot * 64 + row) * p.s_row + (long long)(ch * 16 + (q & 1) * 8) * p.s_red + ts;
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t hi, lo;
      split2(src[(2 * j) * p.s_red], src[(2 * j + 1) * p.s_red], hi, lo);
      o[j] = (q & 2) ? lo : hi;
    }
    unsigned char* dst = p.out + ((((long long)cot * nch + ch) * 9 + t) * 64 + row) * 64 + ((q ^ ((row >> 2) & 3)) << 4);
    *(uint4*)dst = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
static inline size_t ap_packed_bytes(int rows, int red) { return (size_t)(rows >> 6) * (red >> 4) * AP_WBYTES; }

// geometry the kernel takes (tensor alignment is checked at launch)
static inline bool ap_geom_ok(int rows, int red, int h, int w) {
  return rows >= 64 && (rows & 63) == 0 && red >= 16 && (red & 15) == 0 && rows <= AP_MAX_C && red <= AP_MAX_C && h >= 8 &&
         (h & 7) == 0 && w >= 32 && (w & 31) == 0;
}
