// "Record" activations: 3x3 / stride-1 / pad-1 convolution whose input (and output) tensors live in HBM ALREADY in the
// form the MFMA consumes -- per pixel and 32-channel chunk one 128-byte record, bf16 hi[32] | bf16 lo[32] (hi + lo = the
// fp32 value to 2^-17: the same 4 bytes per element as fp32) -- so that staging is a copy: LDS-DMA
// (global_load_lds_dwordx4: no VGPR destination, no affine, no split, no transposition) instead of ~14 vector
// instructions per staged value, and the epilogue writes records straight from the accumulators (a 32x32 MFMA lane
// already holds 16 output channels of ONE pixel: after one v_permlane32_swap per register pair, two runs of 8 consecutive
// channels = 16-byte pieces of that pixel's record).  Replaces, for the 32- / 64-channel levels of the segmenter
// (unet.py:23-30,116-125 at 256x256 / 128x128), igemm_pipe_kernel's per-ELEMENT work: the layers that were issue-bound at
// 24-27 % MFMA-pipe utilisation (profiles/r03_mfma_counters.csv).
//
// Tensor layout ("R32"):  [N][C / 32][H][W][64 x u16]   (u16[0..31] = bf16 hi of channels 32 cb + 0..31, [32..63] = lo)
// LDS image of a haloed tile: the records row-major, 128 B each, the eight 16-byte pieces of record p XOR-swizzled by
// (p >> 1) & 7 -- piece c sits in slot c ^ ((p >> 1) & 7) -- which makes every ds_read_b128 fragment read (16-lane
// groups over consecutive records) conflict-free; LDS-DMA writes LDS lane-linearly, so the swizzle is applied to the
// SOURCE address of each lane (cdna_hip_programming.md, rule 21).  The packed weights carry the same swizzle in global
// memory and are copied linearly.
#include "common.h"
#include "conv_host.h"

namespace {

constexpr int REC_B = 128;             // bytes per record
constexpr int RC_TH = 8, RC_TW = 32;   // output tile: 4 waves x 2 rows of 32 pixels
constexpr int RC_HH = RC_TH + 2, RC_HW = RC_TW + 2;
constexpr int RC_NREC = RC_HH * RC_HW;                 // 340 halo records
constexpr int RC_XPIECES = (RC_NREC * 8 + 63) / 64;    // 43 one-KiB DMA pieces
constexpr int RC_XBYTES = RC_XPIECES * 1024;           // 44032 (the last piece's tail is scratch)
constexpr int RC_WBYTES = 9 * 32 * REC_B;              // 36864: nine taps x 32 rows
constexpr int RC_NJ = (RC_XPIECES + 3) / 4;            // DMA pieces per wave (11)

struct RConvParams {
  const unsigned char* x;     // R32 input
  long long x_sn;             // bytes per image
  int cb_in;                  // input chunks (cin / 32)
  int H, W;
  const unsigned char* pad;   // [cb_in][128 B]: the record read outside the image (zeros, or -shift/scale of a folded BatchNorm)
  const unsigned char* wpack; // [n_co_tiles][cb_in][9 taps][32 rows][128 B], swizzled
  const float* bias;          // [cout] or null
  float slope;
  unsigned char* y;           // R32 output
  long long y_sn;
  int cout;
  float* stats;               // [tiles][cout][2] partial (sum, sum of squares) of the stored values, or null
  int tiles_x, tiles_y, n, n_co_tiles, total;
  unsigned long long* dbg_clk;   // PCUDA_RC_DBG=1 (scripts/micro/rconv_micro.py): per-phase cycle sums of wave 0 of every workgroup
};

#define RC_CLK(i)                                                        \
  if (DBG) {                                                             \
    __builtin_amdgcn_sched_barrier(0);                                   \
    const unsigned long long now_ = __builtin_readcyclecounter();        \
    clk[i] += now_ - tlast; tlast = now_;                                \
    __builtin_amdgcn_sched_barrier(0);                                   \
  }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that waits for this wave's LDS operations only.  __syncthreads() also drains vmcnt: every barrier
// of a stage would wait for the previous tile's global stores to be acknowledged and for the LDS-DMA prefetch to land
// (cdna_hip_programming.md, "Pipelining across barriers").  LDS-DMA data is ordered for the readers by the issuing
// wave's counted vmcnt wait in front of a barrier (rc_wait_dma below).
#define RC_BARRIER()                                        \
  do {                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
  } while (0)

__device__ __forceinline__ void dma16(const unsigned char* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void split8(const float* v, u32x4& hi4, u32x4& lo4) {
  uint32_t a, b, c, d, e, f, g, hh;
  split2(v[0], v[1], a, b); split2(v[2], v[3], c, d); split2(v[4], v[5], e, f); split2(v[6], v[7], g, hh);
  hi4 = u32x4{a, c, e, g};
  lo4 = u32x4{b, d, f, hh};
}

__device__ __forceinline__ int opaque_zero_v() {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}

__device__ __forceinline__ bf16x8 frag(const unsigned char* p) { return __builtin_bit_cast(bf16x8, *(const uint4*)p); }

template <bool STATS, bool DBG>
__global__ __launch_bounds__(256, 2) void rconv3_kernel(const RConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const Ws = smem;                 // (first: a tap's fragment address is a per-lane register + an immediate)
  unsigned char* const Xs = smem + RC_WBYTES;
  float* const sred = (float*)(Xs + RC_XBYTES);   // [4 waves][32][2]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;

  // XCD-aware persistent schedule (as igemm_pipe_kernel): the 8 XCDs own contiguous eighths of the item list, the
  // workgroups of one XCD interleave over it -- concurrent workgroups of an L2 touch adjacent tiles (shared halo rows)
  const int nx = min(8, (int)gridDim.x);
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int gx = ((int)gridDim.x - xcd + nx - 1) / nx;
  const int lo = (int)((long long)p.total * xcd / nx), hi = (int)((long long)p.total * (xcd + 1) / nx);

  // ---- tile-invariant DMA plan: piece j of this wave fills LDS bytes [(w + 4 j) KiB, + 1 KiB) of the X image
  int xrel[RC_NJ];
#pragma unroll
  for (int j = 0; j < RC_NJ; ++j) {
    const int q = (w + 4 * j) * 64 + lane;
    const int rec = min(q >> 3, RC_NREC - 1), s = q & 7;
    const int c = s ^ ((rec >> 1) & 7);
    const int ry = rec / RC_HW, rx = rec - ry * RC_HW;
    xrel[j] = (ry * W + rx) * REC_B + c * 16;
  }
  // ---- tile-invariant fragment addresses (bytes inside Xs / Ws): ks = 0 hi; ^32 -> ks = 1, ^64 -> lo
  int xa[2][9];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int pp = (2 * w + pb + t / 3) * RC_HW + r + (t % 3);
      xa[pb][t] = pp * REC_B + ((h ^ ((pp >> 1) & 7)) << 4);
    }
  const int wa = r * REC_B + ((h ^ ((r >> 1) & 7)) << 4);   // tap t: + t * 4096 (the key of row t * 32 + r is (r >> 1) & 7)
  const int wah[2] = {wa, wa ^ 32}, wal[2] = {wa ^ 64, wa ^ 96};

  int wres = -1;   // (co-tile, chunk) whose weights are resident in Ws
  f32x16 acc[2];
  unsigned long long clk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();

  struct Item { int cot, pt, n, y0, x0; bool border; };
  auto decode = [&](int L) {
    Item it;
    it.cot = L % p.n_co_tiles; it.pt = L / p.n_co_tiles;
    const int txi = it.pt % p.tiles_x, tmp = it.pt / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    it.n = tmp / p.tiles_y;
    it.y0 = tyi * RC_TH; it.x0 = txi * RC_TW;
    it.border = (it.y0 == 0) | (it.y0 + RC_TH == H) | (it.x0 == 0) | (it.x0 + RC_TW == W);
    return it;
  };
  // stage (item, chunk): the haloed input tile by LDS-DMA, and the weights if another (co-tile, chunk) is resident
  auto issue = [&](const Item& it, int chunk) {
    const unsigned char* xb = p.x + (long long)it.n * p.x_sn +
                              ((long long)chunk * H * W + ((long long)(it.y0 - 1) * W + (it.x0 - 1))) * REC_B;
    if (it.border) {
      const unsigned char* padp = p.pad + chunk * REC_B;
      const int lz = lane + opaque_zero_v();   // (keeps the per-piece record arithmetic below inside the loop: hoisted, it spilled)
#pragma unroll
      for (int j = 0; j < RC_NJ; ++j) {
        if (w + 4 * j < RC_XPIECES) {
          // (border tiles only -- a tenth of a 256x256 map: the lane's record is recomputed rather than kept in registers)
          const int q = (w + 4 * j) * 64 + lz;
          const int rec = min(q >> 3, RC_NREC - 1), c = (q & 7) ^ ((rec >> 1) & 7);
          const int ry = (rec * 1928) >> 16, rx = rec - ry * RC_HW;      // rec / 34, exact for rec < 340
          const bool in = ((unsigned)(it.y0 - 1 + ry) < (unsigned)H) & ((unsigned)(it.x0 - 1 + rx) < (unsigned)W);
          const unsigned char* src = in ? xb + xrel[j] : padp + c * 16;
          dma16(src, Xs + (w + 4 * j) * 1024);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < RC_NJ; ++j)
        if (w + 4 * j < RC_XPIECES) dma16(xb + (unsigned)xrel[j], Xs + (w + 4 * j) * 1024);
    }
    const int wkey = it.cot * p.cb_in + chunk;
    if (wkey != wres) {   // uniform
      const unsigned char* wsrc = p.wpack + (long long)wkey * RC_WBYTES + lane * 16;
#pragma unroll
      for (int j = 0; j < 9; ++j) dma16(wsrc + (w + 4 * j) * 1024, Ws + (w + 4 * j) * 1024);
      wres = wkey;
    }
  };

  int L = lo + slot;
  Item cur = decode(min(L, p.total - 1));
  if (L < hi) issue(cur, 0);
  int cot_b = -1;
  float bia[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bia[i] = 0.f;
  // epilogue staging: wave w turns its 64 output records through its own 8 KiB of the (then idle) X image so that the
  // global stores are whole lines: lane l of store k writes piece (l & 7) of record 8 k + (l >> 3)
  unsigned char* const stg = Xs + w * 8192;
  const int sto0 = (lane >> 3) * REC_B + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);   // store k: + k KiB (record 8 k + (lane >> 3))
  bool first = true;
  for (; L < hi; L += gx) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    if (p.bias && cur.cot != cot_b) {   // uniform; (a persistent workgroup of a one-co-tile layer loads it once)
#pragma unroll
      for (int i = 0; i < 16; ++i) bia[i] = p.bias[cur.cot * 32 + (i & 3) + 8 * (i >> 2) + 4 * h];
      cot_b = cur.cot;
    }
    for (int chunk = 0; chunk < p.cb_in; ++chunk) {
      RC_CLK(1)
      // the stage's DMA pieces have landed: everything but the 8 record stores of the previous tile, which were issued
      // behind them (vmcnt counts loads, stores and LDS-DMA together, in issue order)
      if (chunk == 0 && !first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RC_CLK(2)
      RC_BARRIER();   // the stage's DMA pieces of every wave have landed
      RC_CLK(3)
      // ---- MFMA phase: 9 taps x 2 k-steps x (1 row block x 2 pixel blocks) x 3 products
      // (an opaque zero in the fragment addresses: their three XOR variants per (block, tap) are loop-invariant, were
      // hoisted out of the item loop -- 72 registers -- and spilled)
      const int oz = opaque_zero_v();
      // fragments of step s + 1 (tap, k-step) are requested before the MFMAs of step s: the LDS round trip hides behind
      // six MFMAs (two waves per SIMD do not cover it on their own: 27 % of the stage was MFMA phase at half speed)
      bf16x8 fa[2][2], fb[2][4];   // [buffer][ah, al], [buffer][bh0, bl0, bh1, bl1]
      auto load = [&](int bufi, int st) {
        const int t = st >> 1, ks = st & 1;
        const int x0a = xa[0][t] + oz, x1a = xa[1][t] + oz;
        fa[bufi][0] = frag(Ws + wah[ks] + t * 32 * REC_B);
        fa[bufi][1] = frag(Ws + wal[ks] + t * 32 * REC_B);
        fb[bufi][0] = frag(Xs + (x0a ^ (ks * 32)));
        fb[bufi][1] = frag(Xs + (x0a ^ (ks * 32) ^ 64));
        fb[bufi][2] = frag(Xs + (x1a ^ (ks * 32)));
        fb[bufi][3] = frag(Xs + (x1a ^ (ks * 32) ^ 64));
      };
      load(0, 0);
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        const int cb = st & 1;
        if (st + 1 < 18) load(cb ^ 1, st + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][1], fb[cb][2 * pb], acc[pb], 0, 0, 0);
          acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][0], fb[cb][2 * pb + 1], acc[pb], 0, 0, 0);
          acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][0], fb[cb][2 * pb], acc[pb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (DBG) asm volatile("s_nop 0" ::"v"(acc[0][15]), "v"(acc[1][15]));   // (the MFMAs have completed)
      RC_CLK(4)
      RC_BARRIER();   // every wave is done reading this stage's X / W
      RC_CLK(0)
      if (chunk + 1 < p.cb_in) issue(cur, chunk + 1);
    }
    // ---- epilogue: bias + LeakyReLU, BatchNorm partial sums, records out.  Register i of a lane is output channel
    // (i & 3) + 8 (i >> 2) + 4 h of pixel r of the block: swapping the upper half of register 4 j + e (j even) with the lower
    // half of 4 (j + 1) + e leaves lane (r, h) with channels 8 h .. 8 h + 7 and 16 + 8 h .. 16 + 8 h + 7 of its pixel:
    // four 16-byte pieces of its record, written into the wave's staging block (slot = piece ^ (record & 7):
    // conflict-free both ways) and read back as whole records for 1-KiB global stores.
    const int co0 = cur.cot * 32;
    unsigned char* const yb = p.y + (long long)cur.n * p.y_sn + (long long)cur.cot * H * W * REC_B;
    // values in place: bias + LeakyReLU (0 <= slope <= 1; slope 1: identity)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float t = acc[pb][i] + bia[i];
        acc[pb][i] = fmaxf(t, t * p.slope);
      }
    if (STATS) {
      // Per channel and wave: the sum over the wave's 64 pixels.  Lane (r, h) holds 32 partial values (16 channels x {sum, sum
      // of squares} over its two pixels); they go through the wave's staging block as a [value][lane] table (16-byte chunks
      // XOR-swizzled by the value index: conflict-free both ways) and lane (v, h') adds up value v over the 32 lanes of
      // half h' -- 32 ds_write_b32 + 8 ds_read_b128 + 32 adds instead of 160 dependent DPP adds (21 % of the stage by
      // s_memtime stamps).  Fixed order: deterministic.
#pragma unroll
      for (int v = 0; v < 32; ++v) {
        const float a0 = acc[0][v & 15], a1 = acc[1][v & 15];
        const float val = v < 16 ? a0 + a1 : fmaf(a1, a1, a0 * a0);
        *(float*)(stg + v * 256 + ((((lane >> 2) ^ (v & 15)) << 4) | ((lane & 3) << 2))) = val;
      }
      const int vv = lane & 31;
      float tot = 0.f;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const f32x4 q4 = *(const f32x4*)(stg + vv * 256 + (((h * 8 + jj) ^ (vv & 15)) << 4));
        tot += (q4[0] + q4[1]) + (q4[2] + q4[3]);
      }
      // value vv = (quantity vv >> 4, register i = vv & 15) of the lanes with this h: channel (i & 3) + 8 (i >> 2) + 4 h
      const int i = vv & 15, row = (i & 3) + 8 * (i >> 2) + 4 * h;
      sred[(w * 32 + row) * 2 + (vv >> 4)] = tot;
    }
    RC_CLK(6)
    // records: register i of a lane is output channel (i & 3) + 8 (i >> 2) + 4 h of pixel r of the block; swapping the upper
    // half of register 4 j + e (j even) with the lower half of 4 (j + 1) + e leaves lane (r, h) with channels 8 h .. 8 h + 7
    // and 16 + 8 h .. 16 + 8 h + 7 of its pixel: four 16-byte pieces of its record, written into the wave's staging block
    // (slot = piece ^ (record & 7): conflict-free both ways) and read back as whole records for 1-KiB global stores.
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = acc[pb][i];
      // (inline asm: this hipcc folds the builtin's second result into its first -- the stored pieces repeated channels
      // 8 j .. 8 j + 3, found with scripts/micro/rconv_debug.py; s_nop: VALU write -> permlane-swap read wait states)
      asm volatile("s_nop 1\n\t"
                   "v_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\t"
                   "v_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t"
                   "v_permlane32_swap_b32 %8, %12\n\tv_permlane32_swap_b32 %9, %13\n\t"
                   "v_permlane32_swap_b32 %10, %14\n\tv_permlane32_swap_b32 %11, %15\n\t"
                   "s_nop 1"
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                     "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
      const int rr = pb * 32 + r;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        u32x4 hi4, lo4;
        split8(&v[8 * jj], hi4, lo4);
        *(u32x4*)(stg + rr * REC_B + (((2 * jj + h) ^ (rr & 7)) << 4)) = hi4;
        *(u32x4*)(stg + rr * REC_B + (((4 + 2 * jj + h) ^ (rr & 7)) << 4)) = lo4;
      }
    }
    RC_CLK(7)
    // the wave's two tile rows: 2 x 4 KiB contiguous in global memory
    unsigned char* const row0 = yb + ((long long)(cur.y0 + 2 * w) * W + cur.x0) * REC_B + lane * 16;
    u32x4 o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = *(const u32x4*)(stg + sto0 + k * 1024);
    RC_CLK(8)
    RC_BARRIER();   // staging reads done (the next stage's DMA may overwrite the X image); partial sums visible
    RC_CLK(9)
    if (STATS && tid < 32) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) { a += sred[(ww * 32 + tid) * 2 + 0]; b += sred[(ww * 32 + tid) * 2 + 1]; }
      p.stats[((long long)cur.pt * p.cout + co0 + tid) * 2 + 0] = a;
      p.stats[((long long)cur.pt * p.cout + co0 + tid) * 2 + 1] = b;
    }
    RC_CLK(5)
    // next tile's DMA first, this tile's record stores behind it: the loads are in flight while the stores issue
    if (L + gx < hi) {
      cur = decode(L + gx);
      issue(cur, 0);
    }
    first = false;
    RC_CLK(10)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) *(u32x4*)(row0 + (k >> 2) * (W * REC_B) + (k & 3) * 1024) = o[k];
    __builtin_amdgcn_sched_barrier(0);
    RC_CLK(11)
  }
  if (DBG && tid == 0) {
    for (int i = 0; i < 12; ++i) atomicAdd(&p.dbg_clk[i], clk[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// One-chunk / one-co-tile layers (32 -> 32: the 256x256 level): ONE 512-thread workgroup per CU, two groups of four waves that
// alternate roles at every barrier -- while group A runs the MFMA phase of its tile, group B issues the DMA of ITS next tile
// into its own input image, finishes its previous tile (values, statistics, records out) and waits for that DMA; at the
// barrier they swap.  The 36 KB of weights are resident for the whole launch and shared by both groups, each group owns a
// 43-KB input image and 16 KB of record staging: 155 KB.  Against two independent 256-thread workgroups (rconv3_kernel) the
// matrix pipe always has one group feeding it, and a tile's loads are in flight for the whole of the other group's phase.
template <bool STATS, bool DBG>
__global__ __launch_bounds__(512, 1) void rconv3w_kernel(const RConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wv >> 2, w = wv & 3;                       // group, wave inside the group
  unsigned char* const Ws = smem;
  unsigned char* const Xs = smem + RC_WBYTES + g * RC_XBYTES;
  unsigned char* const stg = smem + RC_WBYTES + 2 * RC_XBYTES + (g * 4 + w) * 4096;      // 32 records per wave
  float* const sred = (float*)(smem + RC_WBYTES + 2 * RC_XBYTES + 8 * 4096) + g * 256;   // [4 waves][32][2] per group
  const int H = p.H, W = p.W;

  const int nx = min(8, (int)gridDim.x);
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int gx = ((int)gridDim.x - xcd + nx - 1) / nx;
  const int lo = (int)((long long)p.total * xcd / nx), hi = (int)((long long)p.total * (xcd + 1) / nx);
  // group g of workgroup `slot` takes items lo + 2 slot + g, + 2 gx, ...: the two groups walk adjacent tiles
  const int Lstep = 2 * gx;
  int L = lo + 2 * slot + g;
  const int n_mine = L < hi ? (hi - L + Lstep - 1) / Lstep : 0;
  const int L0 = lo + 2 * slot;
  const int n_a = L0 < hi ? (hi - L0 + Lstep - 1) / Lstep : 0;      // group A has the most items
  const int nint = 2 * n_a + 1;                                     // barrier intervals, the same for every wave

  int xrel[RC_NJ];
#pragma unroll
  for (int j = 0; j < RC_NJ; ++j) {
    const int q = (w + 4 * j) * 64 + lane;
    const int rec = min(q >> 3, RC_NREC - 1), s = q & 7;
    const int c = s ^ ((rec >> 1) & 7);
    const int ry = rec / RC_HW, rx = rec - ry * RC_HW;
    xrel[j] = (ry * W + rx) * REC_B + c * 16;
  }
  int xa[2][9];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int pp = (2 * w + pb + t / 3) * RC_HW + r + (t % 3);
      xa[pb][t] = pp * REC_B + ((h ^ ((pp >> 1) & 7)) << 4);
    }
  const int wa = r * REC_B + ((h ^ ((r >> 1) & 7)) << 4);
  const int wah[2] = {wa, wa ^ 32}, wal[2] = {wa ^ 64, wa ^ 96};
  const int sto0 = (lane >> 3) * REC_B + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);

  struct Item { int pt, n, y0, x0; bool border; };
  auto decode = [&](int Li) {
    Item it;
    it.pt = Li;
    const int txi = Li % p.tiles_x, tmp = Li / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    it.n = tmp / p.tiles_y;
    it.y0 = tyi * RC_TH; it.x0 = txi * RC_TW;
    it.border = (it.y0 == 0) | (it.y0 + RC_TH == H) | (it.x0 == 0) | (it.x0 + RC_TW == W);
    return it;
  };
  auto issue = [&](const Item& it) {
    const unsigned char* xb = p.x + (long long)it.n * p.x_sn + ((long long)(it.y0 - 1) * W + (it.x0 - 1)) * REC_B;
    if (it.border) {
      const int lz = lane + opaque_zero_v();
#pragma unroll
      for (int j = 0; j < RC_NJ; ++j) {
        if (w + 4 * j < RC_XPIECES) {
          const int q = (w + 4 * j) * 64 + lz;
          const int rec = min(q >> 3, RC_NREC - 1), c = (q & 7) ^ ((rec >> 1) & 7);
          const int ry = (rec * 1928) >> 16, rx = rec - ry * RC_HW;
          const bool in = ((unsigned)(it.y0 - 1 + ry) < (unsigned)H) & ((unsigned)(it.x0 - 1 + rx) < (unsigned)W);
          const unsigned char* src = in ? xb + xrel[j] : p.pad + c * 16;
          dma16(src, Xs + (w + 4 * j) * 1024);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < RC_NJ; ++j)
        if (w + 4 * j < RC_XPIECES) dma16(xb + (unsigned)xrel[j], Xs + (w + 4 * j) * 1024);
    }
  };

  float bia[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bia[i] = p.bias ? p.bias[(i & 3) + 8 * (i >> 2) + 4 * h] : 0.f;
  // ---- prologue: the weights (36 pieces over the 8 waves), each group's first tile
  {
    const unsigned char* wsrc = p.wpack + lane * 16;
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (wv + 8 * j < 36) dma16(wsrc + (wv + 8 * j) * 1024, Ws + (wv + 8 * j) * 1024);
  }
  Item cur = decode(min(L, p.total - 1)), done = cur;
  if (n_mine > 0) issue(cur);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RC_BARRIER();

  f32x16 acc[2];
  unsigned long long clk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
  int k_mine = 0;          // tiles this group has computed
  bool pending = false;    // a computed tile whose epilogue has not run yet
  for (int it = 0; it < nint; ++it) {
    if ((it & 1) == g) {
      // ================= compute interval =================
      RC_CLK(0)
      if (k_mine < n_mine) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
        const int oz = opaque_zero_v();
        bf16x8 fa[2][2], fb[2][4];
        auto load = [&](int bufi, int st) {
          const int t = st >> 1, ks = st & 1;
          const int x0a = xa[0][t] + oz, x1a = xa[1][t] + oz;
          fa[bufi][0] = frag(Ws + wah[ks] + t * 32 * REC_B);
          fa[bufi][1] = frag(Ws + wal[ks] + t * 32 * REC_B);
          fb[bufi][0] = frag(Xs + (x0a ^ (ks * 32)));
          fb[bufi][1] = frag(Xs + (x0a ^ (ks * 32) ^ 64));
          fb[bufi][2] = frag(Xs + (x1a ^ (ks * 32)));
          fb[bufi][3] = frag(Xs + (x1a ^ (ks * 32) ^ 64));
        };
        load(0, 0);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
          const int cb = st & 1;
          if (st + 1 < 18) load(cb ^ 1, st + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][1], fb[cb][2 * pb], acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][0], fb[cb][2 * pb + 1], acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][0], fb[cb][2 * pb], acc[pb], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (DBG) asm volatile("s_nop 0" ::"v"(acc[0][15]), "v"(acc[1][15]));
        RC_CLK(1)
        done = cur;
        ++k_mine;
        pending = true;
      }
    } else if (pending) {
      // ================= memory interval: next tile's DMA first, then this tile's epilogue behind it =================
      RC_CLK(0)
      L += Lstep;
      const bool more = k_mine < n_mine;
      if (more) {
        cur = decode(L);
        issue(cur);
      }
      RC_CLK(2)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float t = acc[pb][i] + bia[i];
          acc[pb][i] = fmaxf(t, t * p.slope);
        }
      if (STATS) {
        // (as rconv3_kernel: a [value][lane] table in the wave's staging block, 16-byte chunks XOR-swizzled by the value
        // index; the block holds 4 KiB = 16 values x 64 lanes, so the sums and the sums of squares go through it in turn)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const float a0 = acc[0][v], a1 = acc[1][v];
            const float val = qq == 0 ? a0 + a1 : fmaf(a1, a1, a0 * a0);
            *(float*)(stg + v * 256 + ((((lane >> 2) ^ v) << 4) | ((lane & 3) << 2))) = val;
          }
          const int vv = lane & 15, part = (lane >> 4) & 1;      // lane (vv, part, h): value vv, 16 of the 32 lanes of half h
          float tot = 0.f;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x4 q4 = *(const f32x4*)(stg + vv * 256 + (((h * 8 + part * 4 + jj) ^ vv) << 4));
            tot += (q4[0] + q4[1]) + (q4[2] + q4[3]);
          }
          tot += __shfl_xor(tot, 16, 64);                        // the two 16-lane parts of the half
          const int row = (vv & 3) + 8 * (vv >> 2) + 4 * h;
          if (part == 0) sred[(w * 32 + row) * 2 + qq] = tot;
        }
      }
      RC_CLK(3)
      unsigned char* const yb = p.y + (long long)done.n * p.y_sn;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = acc[pb][i];
        asm volatile("s_nop 1\n\t"
                     "v_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\t"
                     "v_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t"
                     "v_permlane32_swap_b32 %8, %12\n\tv_permlane32_swap_b32 %9, %13\n\t"
                     "v_permlane32_swap_b32 %10, %14\n\tv_permlane32_swap_b32 %11, %15\n\t"
                     "s_nop 1"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                       "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          u32x4 hi4, lo4;
          split8(&v[8 * jj], hi4, lo4);
          *(u32x4*)(stg + r * REC_B + (((2 * jj + h) ^ (r & 7)) << 4)) = hi4;
          *(u32x4*)(stg + r * REC_B + (((4 + 2 * jj + h) ^ (r & 7)) << 4)) = lo4;
        }
        unsigned char* const row0 = yb + ((long long)(done.y0 + 2 * w + pb) * W + done.x0) * REC_B + lane * 16;
        u32x4 o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = *(const u32x4*)(stg + sto0 + k * 1024);
#pragma unroll
        for (int k = 0; k < 4; ++k) *(u32x4*)(row0 + k * 1024) = o[k];
      }
      pending = false;
      RC_CLK(4)
      // the DMA pieces of the next tile have landed: all but the 8 record stores issued behind them
      if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      RC_CLK(5)
    }
    RC_BARRIER();
    RC_CLK(6)
    if (STATS && (it & 1) != g && w == 0 && lane < 32 && !pending && k_mine > 0 && it >= 1) {
      // (after the barrier that closed this group's memory interval: its four waves' partial sums are complete)
      if (it == 2 * (k_mine - 1) + 1 + g) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { a += sred[(ww * 32 + lane) * 2 + 0]; b += sred[(ww * 32 + lane) * 2 + 1]; }
        p.stats[((long long)done.pt * p.cout + lane) * 2 + 0] = a;
        p.stats[((long long)done.pt * p.cout + lane) * 2 + 1] = b;
      }
    }
  }
  if (DBG && lane == 0 && w == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&p.dbg_clk[i], clk[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// layout conversions at the module boundary (and for tests): NCHW fp32 (+ per-channel affine) <-> R32
// one thread = one 16-byte piece (8 channels of one pixel, hi or lo)
__global__ void rec_from_nchw_kernel(const float* __restrict__ x, long long sn, long long sc, int c, int hw,
                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                     unsigned char* __restrict__ out, long long total) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (n, cb, pixel, g): g = 8-channel group 0..3
  if (gid >= total) return;
  const int g = (int)(gid & 3);
  const long long pix = (gid >> 2) % hw;
  const long long ncb = (gid >> 2) / hw;
  const int cbn = (c + 31) / 32;
  const int cb = (int)(ncb % cbn);
  const long long n = ncb / cbn;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = cb * 32 + g * 8 + j;
    float t = 0.f;
    if (ch < c) {
      t = x[n * sn + (long long)ch * sc + pix];
      if (scale) t = fmaf(t, scale[ch], shift[ch]);
    }
    v[j] = t;
  }
  u32x4 hi4, lo4;
  split8(v, hi4, lo4);
  unsigned char* rec = out + (ncb * hw + pix) * REC_B;
  *(u32x4*)(rec + g * 16) = hi4;
  *(u32x4*)(rec + 64 + g * 16) = lo4;
}

__global__ void rec_to_nchw_kernel(const unsigned char* __restrict__ in, int c, int hw, float* __restrict__ y, long long sn,
                                   long long sc, long long total) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (n, ch, pixel)
  if (gid >= total) return;
  const long long pix = gid % hw;
  const int ch = (int)((gid / hw) % c);
  const long long n = gid / hw / c;
  const int cbn = (c + 31) / 32;
  const unsigned short* rec = (const unsigned short*)(in + ((n * cbn + ch / 32) * hw + pix) * REC_B);
  const float hi = __builtin_bit_cast(float, (unsigned)rec[ch & 31] << 16);
  const float lo = __builtin_bit_cast(float, (unsigned)rec[32 + (ch & 31)] << 16);
  y[n * sn + (long long)ch * sc + pix] = hi + lo;
}

// weights: fp32 OIHW [cout][cin][3][3] (x optional per-input-channel scale: a folded BatchNorm) ->
// [cout / 32][cin / 32][tap][row][128 B] with the LDS swizzle baked in.  One thread = one 16-byte piece.
__global__ void rconv3_pack_kernel(const float* __restrict__ wsrc, int cout, int cin, const float* __restrict__ scale,
                                   unsigned char* __restrict__ out, long long total) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (cot, chunk, tap, row, slot)
  if (gid >= total) return;
  const int s = (int)(gid & 7);
  const int row = (int)((gid >> 3) & 31);
  const int tap = (int)((gid >> 8) % 9);
  const long long cc = (gid >> 8) / 9;
  const int cbn = cin / 32;
  const int chunk = (int)(cc % cbn), cot = (int)(cc / cbn);
  const int q = tap * 32 + row;
  const int c = s ^ ((q >> 1) & 7);          // the piece this slot holds
  const int g = c & 3;
  const bool lo_plane = c >= 4;
  const int co = cot * 32 + row;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = chunk * 32 + g * 8 + j;
    float t = co < cout ? wsrc[((long long)co * cin + ci) * 9 + tap] : 0.f;
    if (scale) t *= scale[ci];
    v[j] = t;
  }
  u32x4 hi4, lo4;
  split8(v, hi4, lo4);
  *(u32x4*)(out + gid * 16) = lo_plane ? lo4 : hi4;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
static unsigned long long* g_rc_dbg = nullptr;
// timing experiments (PCUDA_RC_DBG=1): the six per-phase cycle sums of rconv3_kernel -- top barrier, DMA issue, DMA wait,
// barrier, MFMA phase, epilogue -- over wave 0 of every workgroup since the last call; read + reset
extern "C" int pcuda_rconv3_debug_clocks(unsigned long long* out6) {
  if (!out6) PCUDA_FAIL(PCUDA_E_BADARG, "rconv3_debug_clocks: null");
  memset(out6, 0, 12 * sizeof(unsigned long long));
  if (!g_rc_dbg) return PCUDA_OK;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out6, g_rc_dbg, 96, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemset(g_rc_dbg, 0, 128) != hipSuccess)
    PCUDA_FAIL(PCUDA_E_LAUNCH, "rconv3_debug_clocks: copy failed");
  return PCUDA_OK;
}

extern "C" size_t pcuda_rec_bytes(int n, int c, int h, int w) { return (size_t)n * ((c + 31) / 32) * h * w * REC_B; }

extern "C" int pcuda_rec_from_nchw(const float* x, long long sn, long long sc, int n, int c, int h, int w, const float* scale,
                                   const float* shift, void* out, pcuda_stream_t stream) {
  if (!x || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0 || ((scale == nullptr) != (shift == nullptr)))
    PCUDA_FAIL(PCUDA_E_BADARG, "rec_from_nchw: bad argument");
  const long long total = (long long)n * ((c + 31) / 32) * h * w * 4;
  hipLaunchKernelGGL(rec_from_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, sn, sc,
                     c, h * w, scale, shift, (unsigned char*)out, total);
  PCUDA_CHECK_LAUNCH("rec_from_nchw");
  return PCUDA_OK;
}

extern "C" int pcuda_rec_to_nchw(const void* rec, int n, int c, int h, int w, float* y, long long sn, long long sc,
                                 pcuda_stream_t stream) {
  if (!rec || !y || n <= 0 || c <= 0 || h <= 0 || w <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "rec_to_nchw: bad argument");
  const long long total = (long long)n * c * h * w;
  hipLaunchKernelGGL(rec_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)rec, c, h * w, y, sn, sc, total);
  PCUDA_CHECK_LAUNCH("rec_to_nchw");
  return PCUDA_OK;
}

extern "C" size_t pcuda_rconv3_packed_bytes(int cout, int cin) { return (size_t)((cout + 31) / 32) * (cin / 32) * RC_WBYTES; }

extern "C" int pcuda_rconv3_pack(const float* w, int cout, int cin, const float* in_scale, void* out, pcuda_stream_t stream) {
  if (!w || !out || cout <= 0 || cin <= 0 || (cin & 31)) PCUDA_FAIL(PCUDA_E_BADARG, "rconv3_pack: cin must be a multiple of 32");
  const long long total = (long long)((cout + 31) / 32) * (cin / 32) * 9 * 32 * 8;
  hipLaunchKernelGGL(rconv3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin,
                     in_scale, (unsigned char*)out, total);
  PCUDA_CHECK_LAUNCH("rconv3_pack");
  return PCUDA_OK;
}

extern "C" int pcuda_rconv3_tiles(int n, int h, int w) { return n * (h / RC_TH) * (w / RC_TW); }

extern "C" int pcuda_rconv3_forward(const void* x, int n, int cin, int h, int w, const void* pad_records, const void* wpacked,
                                    const float* bias, float slope, int cout, void* y, float* stats, pcuda_stream_t stream) {
  if (!x || !wpacked || !y || !pad_records || n <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "rconv3_forward: null argument");
  if ((cin & 31) || (cout & 31) || (w % RC_TW) || (h % RC_TH))
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "rconv3_forward: needs cin, cout multiples of 32, rows of 32 k pixels, 8 k rows (got %d -> %d at %dx%d)",
               cin, cout, h, w);
  if (slope < 0.f || slope > 1.f) PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "rconv3_forward: slope outside [0, 1]");
  RConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const unsigned char*)x; p.cb_in = cin / 32; p.H = h; p.W = w;
  p.x_sn = (long long)p.cb_in * h * w * REC_B;
  p.pad = (const unsigned char*)pad_records; p.wpack = (const unsigned char*)wpacked; p.bias = bias; p.slope = slope;
  p.y = (unsigned char*)y; p.cout = cout; p.y_sn = (long long)(cout / 32) * h * w * REC_B;
  p.stats = stats;
  p.tiles_x = w / RC_TW; p.tiles_y = h / RC_TH; p.n = n; p.n_co_tiles = cout / 32;
  p.total = n * p.tiles_x * p.tiles_y * p.n_co_tiles;
  {
    static int dbg = -1;
    static unsigned long long* buf = nullptr;
    if (dbg < 0) { const char* e = getenv("PCUDA_RC_DBG"); dbg = (e && atoi(e)) ? 1 : 0; }
    if (dbg && !buf && hipMalloc(&buf, 16 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemset(buf, 0, 128);
    p.dbg_clk = dbg ? buf : nullptr;
    g_rc_dbg = buf;
  }
  const size_t lds = RC_XBYTES + RC_WBYTES + 4 * 32 * 2 * sizeof(float);
  int ncu = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
  }
  const int grid = p.total < 2 * ncu ? p.total : 2 * ncu;
  char tag[128];
  snprintf(tag, sizeof(tag), "rconv3 n%d cin%d cout%d %dx%d lds%zu", n, cin, cout, h, w, lds);
  ProfScope prof(PCUDA_FAM_CONV_FWD, 2.0 * n * (double)h * w * cout * (double)cin * 9, (hipStream_t)stream, tag);
  auto launch = [&](auto kern) -> int {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD);
    if (e != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "rconv3: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    return PCUDA_OK;
  };
  static int wmode = -1;
  if (wmode < 0) { const char* e = getenv("PCUDA_RC_W"); wmode = (e && !atoi(e)) ? 0 : 1; }
  int rc;
  if (wmode && p.cb_in == 1 && p.n_co_tiles == 1) {
    // one chunk, one co-tile: the two-group kernel, one 512-thread workgroup per CU with the weights resident
    const size_t ldsw = RC_WBYTES + 2 * RC_XBYTES + 8 * 4096 + 2 * 1024;
    const int gridw = p.total < 2 * ncu ? (p.total + 1) / 2 : ncu;
    auto launchw = [&](auto kern) -> int {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD);
      if (e != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "rconv3w: cannot raise dynamic LDS: %s", hipGetErrorString(e));
      hipLaunchKernelGGL(kern, dim3(gridw), dim3(512), ldsw, (hipStream_t)stream, p);
      return PCUDA_OK;
    };
    rc = p.dbg_clk ? (stats ? launchw(rconv3w_kernel<true, true>) : launchw(rconv3w_kernel<false, true>))
                   : (stats ? launchw(rconv3w_kernel<true, false>) : launchw(rconv3w_kernel<false, false>));
  } else {
    rc = p.dbg_clk ? (stats ? launch(rconv3_kernel<true, true>) : launch(rconv3_kernel<false, true>))
                   : (stats ? launch(rconv3_kernel<true, false>) : launch(rconv3_kernel<false, false>));
  }
  if (rc != PCUDA_OK) return rc;
  PCUDA_CHECK_LAUNCH("rconv3_kernel");
  return PCUDA_OK;
}
