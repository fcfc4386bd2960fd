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
#include <stdlib.h>

#include "conv_igemm.h"

// ------------------------------------------------------------------------------------------
// weight repack: fp32 [rows][red][taps] (any strides) -> bf16 [co_tile][chunk][tap][CO_TILE][40]
// ------------------------------------------------------------------------------------------
__global__ void pack_kernel(const PackParams p) {
  const long long total = (long long)p.n_co_tiles * p.nchunks * p.ntaps * p.co_tile * IG_REC;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int col = (int)(idx % IG_REC);
    long long rest = idx / IG_REC;
    int row = (int)(rest % p.co_tile); rest /= p.co_tile;
    int t = (int)(rest % p.ntaps); rest /= p.ntaps;
    int ch = (int)(rest % p.nchunks);
    int cot = (int)(rest / p.nchunks);
    int r = cot * p.co_tile + row, c = ch * 32 + col;
    float v = 0.f;
    if (col < 32 && r < p.rows && c < p.red) v = p.w[r * p.s_row + c * p.s_red + p.tap_src[t]];
    __bf16 hi = (__bf16)v;
    p.out[idx] = __builtin_bit_cast(uint16_t, hi);
    if (p.lo_off) {
      __bf16 lo = (__bf16)(v - (float)hi);
      p.out[p.lo_off + idx] = __builtin_bit_cast(uint16_t, lo);
    }
  }
}

// ------------------------------------------------------------------------------------------
// stage one 32-channel chunk of a haloed input tile into LDS, pixel-major, bf16 hi (+ lo)
// ------------------------------------------------------------------------------------------
template <bool X3>
__device__ __forceinline__ void stage_write(unsigned char* __restrict__ xhi, unsigned char* __restrict__ xlo,
                                            const float (&v)[8], int pix, int g) {
  uint4 hi, lo;
  if (X3) {
    split2(v[0], v[1], hi.x, lo.x); split2(v[2], v[3], hi.y, lo.y);
    split2(v[4], v[5], hi.z, lo.z); split2(v[6], v[7], hi.w, lo.w);
    *(uint4*)(xlo + (size_t)pix * IG_REC_BYTES + g * 16) = lo;
  } else {
    hi.x = pack_bf16x2(v[0], v[1]); hi.y = pack_bf16x2(v[2], v[3]);
    hi.z = pack_bf16x2(v[4], v[5]); hi.w = pack_bf16x2(v[6], v[7]);
  }
  *(uint4*)(xhi + (size_t)pix * IG_REC_BYTES + g * 16) = hi;
}

// PF pixels per thread (npix <= PF*256), everything unrolled.  Instruction-lean by construction:
//  * the channel plane base (p + n*sn + c*sc) is wave-uniform -> SGPR pair; each lane adds ONE 32-bit
//    byte offset computed once per pixel slot -> `global_load_dword v, v_off, s[base]`, no 64-bit VALU;
//  * loads are UNCONDITIONAL on clamped (always valid) addresses and the padding zeros are selected
//    afterwards: a per-lane `if (inb) load` makes hipcc branch around every load and wait for it
//    (cdna_hip_programming.md, "Three .s-level traps" (c));
//  * the lazy-BatchNorm scale/shift are read in uniform control flow (scalar loads), one fma per value.
// Phase 1 issues all 32*PF loads of the chunk, phase 2 applies the affine, splits and writes LDS.
template <bool X3, int PF>
__device__ __forceinline__ void stage_x_chunk_mlp(unsigned char* __restrict__ xhi, unsigned char* __restrict__ xlo,
                                                  const pcuda_src& x, int n, int cin, int chunk, int in_h, int in_w,
                                                  int in_shift, int in_row, int oy0, int ox0, int th, int tw,
                                                  int ngroups, int nwrite, int tid0) {
  const int npix = th * tw;
  const int cbase = chunk * 32;
  const int tid = tid0;   // first pixel slot of this lane (callers may offset it to walk big tiles)
  unsigned voff[PF];
  bool inb[PF];
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int pix = min(tid + s * 256, npix - 1);
    const int iy = pix / tw, ix = pix - iy * tw;
    const int gy = oy0 + iy, gx = ox0 + ix;
    inb[s] = (tid + s * 256 < npix) & ((unsigned)gy < (unsigned)in_h) & ((unsigned)gx < (unsigned)in_w);
    const int cy = min(max(gy, 0), in_h - 1), cx = min(max(gx, 0), in_w - 1);
    voff[s] = (unsigned)((cy >> in_shift) * in_row + (cx >> in_shift)) * 4u;
  }
  const char* b1 = (const char*)(x.p1 + (long long)n * x.sn1);
  const char* b2 = (const char*)(x.p2 + (long long)n * x.sn2);
  float v[PF][4][8];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < ngroups) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = min(cbase + g * 8 + j, cin - 1);   // wave-uniform, clamped
        const char* chan = (c < x.c1) ? b1 + (long long)c * x.sc1 * 4 : b2 + (long long)(c - x.c1) * x.sc2 * 4;
#pragma unroll
        for (int s = 0; s < PF; ++s) v[s][g][j] = *(const float*)(chan + voff[s]);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g >= ngroups && g < nwrite) {   // k-step channels past cin: zeros, not LDS garbage
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (tid + s * 256 < npix) {
          *(uint4*)(xhi + (size_t)(tid + s * 256) * IG_REC_BYTES + g * 16) = make_uint4(0, 0, 0, 0);
          if (X3) *(uint4*)(xlo + (size_t)(tid + s * 256) * IG_REC_BYTES + g * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    if (g < ngroups) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cbase + g * 8 + j;
        const bool cok = c < cin;
        float sc = 1.f, sh = 0.f;
        if (cok) {
          if (c < x.c1) { if (x.scale1) { sc = x.scale1[c]; sh = x.shift1[c]; } }
          else { if (x.scale2) { sc = x.scale2[c - x.c1]; sh = x.shift2[c - x.c1]; } }
        }
#pragma unroll
        for (int s = 0; s < PF; ++s) {
          const float t = fmaf(v[s][g][j], sc, sh);
          v[s][g][j] = (inb[s] & cok) ? t : 0.f;
        }
      }
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (tid + s * 256 < npix) stage_write<X3>(xhi, xlo, v[s][g], tid + s * 256, g);
    }
  }
}

template <bool X3, int MAXPF = 3>
__device__ __forceinline__ void stage_x_chunk(unsigned char* __restrict__ xhi, unsigned char* __restrict__ xlo,
                                              const pcuda_src& x, int n, int cin, int chunk, int in_h, int in_w,
                                              int in_shift, int in_row, int oy0, int ox0, int th, int tw,
                                              int ngroups, int nwrite, int tid) {
  const int npix = th * tw;
  if (MAXPF == 1) {   // register-tight callers (wgrad: 80+ accumulator registers): 32 loads in flight per lane
    for (int pix0 = 0; pix0 < npix; pix0 += 256)
      stage_x_chunk_mlp<X3, 1>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid + pix0);
    return;
  }
  if (npix <= 256) { stage_x_chunk_mlp<X3, 1>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid); return; }
  if (npix <= 512) { stage_x_chunk_mlp<X3, 2>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid); return; }
  if (npix <= 768) { stage_x_chunk_mlp<X3, 3>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid); return; }
  for (int pix0 = 0; pix0 < npix; pix0 += 512)   // big tiles: 512 pixels at a time
    stage_x_chunk_mlp<X3, 2>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid + pix0);
}

// contiguous global -> LDS copy of nvec 16-B vectors, 4 loads per lane in flight
__device__ __forceinline__ void copy_vec16(unsigned char* __restrict__ dst, const uint4* __restrict__ src, int nvec,
                                           int tid) {
  for (int base = 0; base < nvec; base += 1024) {
    uint4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + tid + u * 256;
      r[u] = src[i < nvec ? i : nvec - 1];   // unconditional (clamped) load keeps r[] in registers
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + tid + u * 256;
      if (i < nvec) ((uint4*)dst)[i] = r[u];
    }
  }
}

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) {
  return __builtin_bit_cast(bf16x8, *(const uint4*)p);
}

// ------------------------------------------------------------------------------------------
// forward / dgrad kernel.  256 threads = 4 waves; tile = CO_TILE rows x (128*NPB) logical pixels;
// wave w owns pixels [32*NPB*w, +32*NPB) (NPB 32-pixel MFMA column blocks) for all CO_BLKS row blocks.
// ------------------------------------------------------------------------------------------
template <bool X3, int CO_BLKS, bool CLAMP, int NPB>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p, const int x_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CO_TILE = 32 * CO_BLKS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;

  // XCD-aware block order: blocks that share an XCD (bid % 8) walk adjacent tiles, and the
  // co-tiles of one pixel tile are adjacent, so the haloed input tile is fetched once per L2.
  const unsigned bid = blockIdx.x, nwg = gridDim.x;
  const unsigned xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
  const unsigned L = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
  const int cot = L % p.n_co_tiles;
  const int pt = L / p.n_co_tiles;
  const int txi = pt % p.tiles_x;
  const int tmp = pt / p.tiles_x;
  const int tyi = tmp % p.tiles_y;
  const int n = tmp / p.tiles_y;
  const int TW = p.tw, TH = p.th, TPIX = TW * TH;
  const int y0 = tyi * TH, x0 = txi * TW;

  int oy0 = y0 * p.in_step + p.dy_min, ox0 = x0 * p.in_step + p.dx_min;
  int th = p.ih_t, tw = p.iw_t;
  if (CLAMP) {
    const int y1 = min(oy0 + th, p.in_h), x1 = min(ox0 + tw, p.in_w);
    oy0 = max(oy0, 0); ox0 = max(ox0, 0);
    th = max(y1 - oy0, 0); tw = max(x1 - ox0, 0);
  }
  const int npix = th * tw;

  unsigned char* Xhi = smem;
  unsigned char* Xlo = smem + (size_t)x_cap * IG_REC_BYTES;
  unsigned char* Whi = smem + (size_t)(X3 ? 2 : 1) * x_cap * IG_REC_BYTES;
  unsigned char* Wlo = Whi + (size_t)p.tg * CO_TILE * IG_REC_BYTES;

  // per-lane pixel of each MFMA column block
  int pty[NPB], ptx[NPB], bbase[NPB];
  bool pvalid[NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    const int plr = w * (32 * NPB) + pb * 32 + r;
    pvalid[pb] = plr < TPIX;
    const int pl = min(plr, TPIX - 1);          // idle slots of a non-power-of-two tile read a valid pixel
    pty[pb] = IG_TY(pl, p.tmagic);
    ptx[pb] = pl - pty[pb] * TW;
    bbase[pb] = ((pty[pb] * p.in_step) * tw + ptx[pb] * p.in_step) * IG_REC_BYTES + h * 16;
  }

  f32x16 acc[CO_BLKS][NPB];
#pragma unroll
  for (int cb = 0; cb < CO_BLKS; ++cb)
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][pb][i] = 0.f;

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int cvalid = min(32, p.cin - chunk * 32);
    const int nks = cvalid > 16 ? 2 : 1;
    __syncthreads();   // every wave is done reading the previous chunk's X / W
    if (!(p.dbg & 1))
      stage_x_chunk<X3>(Xhi, Xlo, p.x, n, p.cin, chunk, p.in_h, p.in_w, p.in_shift, p.in_row, oy0, ox0, th, tw,
                        (cvalid + 7) >> 3, nks * 2, tid);
    if (CLAMP && tid < 5) {   // the all-zero record that out-of-image taps read
      *(uint4*)(Xhi + (size_t)npix * IG_REC_BYTES + tid * 16) = make_uint4(0, 0, 0, 0);
      if (X3) *(uint4*)(Xlo + (size_t)npix * IG_REC_BYTES + tid * 16) = make_uint4(0, 0, 0, 0);
    }
    for (int t0 = 0; t0 < p.ntaps; t0 += p.tg) {
      if (t0 > 0) __syncthreads();
      const int tgc = min(p.tg, p.ntaps - t0);
      {
        const long long slab = (long long)CO_TILE * IG_REC;   // bf16 elements per tap
        const uint16_t* src = p.wpack + (((long long)cot * p.nchunks + chunk) * p.ntaps + t0) * slab;
        const int nvec = tgc * CO_TILE * 5;                  // 16-B vectors
        if (!(p.dbg & 8)) {
          copy_vec16(Whi, (const uint4*)src, nvec, tid);
          if (X3) copy_vec16(Wlo, (const uint4*)(src + p.w_lo_off), nvec, tid);
        }
      }
      __syncthreads();
      for (int tl = 0; tl < ((p.dbg & 2) ? 0 : tgc); ++tl) {
        const int t = t0 + tl;
        int baddr[NPB];
        if (CLAMP) {
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) {
            const int gy = (y0 + pty[pb]) * p.in_step + p.dy[t];
            const int gx = (x0 + ptx[pb]) * p.in_step + p.dx[t];
            const bool ok = ((unsigned)gy < (unsigned)p.in_h) & ((unsigned)gx < (unsigned)p.in_w);
            const int idx = ok ? (gy - oy0) * tw + (gx - ox0) : npix;
            baddr[pb] = idx * IG_REC_BYTES + h * 16;
          }
        } else {
          const int toff = ((p.dy[t] - p.dy_min) * tw + (p.dx[t] - p.dx_min)) * IG_REC_BYTES;
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) baddr[pb] = bbase[pb] + toff;
        }
        const int abase = (tl * CO_TILE + r) * IG_REC_BYTES + h * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (ks < nks) {
            bf16x8 ah[CO_BLKS], al[CO_BLKS], bh[NPB], bl[NPB];
#pragma unroll
            for (int cb = 0; cb < CO_BLKS; ++cb) {
              ah[cb] = lds_frag(Whi + abase + cb * 32 * IG_REC_BYTES + ks * 32);
              if (X3) al[cb] = lds_frag(Wlo + abase + cb * 32 * IG_REC_BYTES + ks * 32);
            }
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
              bh[pb] = lds_frag(Xhi + baddr[pb] + ks * 32);
              if (X3) bl[pb] = lds_frag(Xlo + baddr[pb] + ks * 32);
            }
#pragma unroll
            for (int cb = 0; cb < CO_BLKS; ++cb)
#pragma unroll
              for (int pb = 0; pb < NPB; ++pb) {
                if (X3) {
                  acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb], bh[pb], acc[cb][pb], 0, 0, 0);
                  acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bl[pb], acc[cb][pb], 0, 0, 0);
                }
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bh[pb], acc[cb][pb], 0, 0, 0);
              }
          }
        }
      }
    }
  }

  // ---- epilogue: bias + LeakyReLU, NCHW store (register i = one output channel, the 32 lanes of
  // a half-wave = 32 consecutive pixels), optional per-tile BatchNorm partial sums.
  __syncthreads();
  if (p.dbg & 4) {
    if (acc[0][0][0] == 123.456f) p.y.p1[0] = 1.f;   // keep the accumulators live
    return;
  }
  float* sred = (float*)smem;   // [4 waves][CO_TILE][2]
  // per-lane output pixel offsets (elements within a plane), computed once
  int poff[NPB];
  bool pok[NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    const int ly = y0 + pty[pb], lx = x0 + ptx[pb];
    pok[pb] = pvalid[pb] & (ly < p.lh) & (lx < p.lw);
    poff[pb] = (ly * p.oy_mul + p.oy_off) * p.out_w + (lx * p.ox_mul + p.ox_off);
  }
  const int co0 = cot * CO_TILE;
  // fast path: the two half-waves (rows r0 and r0+4) of every register land in the same destination
  // tensor -> the plane base of row r0 is wave-uniform (SGPR) and lanes add a 32-bit offset
  const bool uni = (p.y.c1 >= p.cout) | ((p.y.c1 & 7) == 0);
  float* const yb1 = p.y.p1 + (long long)n * p.y.sn1;
  float* const yb2 = p.y.p2 + (long long)n * p.y.sn2;
#pragma unroll
  for (int cb = 0; cb < CO_BLKS; ++cb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row0 = cb * 32 + (i & 3) + 8 * (i >> 2);   // wave-uniform
      const int row = row0 + 4 * h;
      const int co = co0 + row;
      const bool cok = co < p.cout;
      const float b = (cok && p.bias) ? p.bias[co] : 0.f;
      float* plane;
      if (uni) {
        const int cu = min(co0 + row0, p.cout - 1);
        float* base = (cu < p.y.c1) ? yb1 + (long long)cu * p.y.sc1 : yb2 + (long long)(cu - p.y.c1) * p.y.sc2;
        const long long hs = (cu < p.y.c1) ? p.y.sc1 : p.y.sc2;
        plane = base + (h ? 4 * hs : 0);
      } else {
        const int cc = min(co, p.cout - 1);
        plane = (cc < p.y.c1) ? yb1 + (long long)cc * p.y.sc1 : yb2 + (long long)(cc - p.y.c1) * p.y.sc2;
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        float v = acc[cb][pb][i] + b;
        v = v > 0.f ? v : v * p.slope;
        float* dst = plane + poff[pb];
        if (cok & pok[pb]) {
          if (p.accumulate) v += *dst;
          *dst = v;
          s1 += v;
          s2 += v * v;
        }
      }
      if (p.stats) {
        s1 = half_wave_sum(s1);
        s2 = half_wave_sum(s2);
        if (r == 0) {
          sred[(w * CO_TILE + row) * 2 + 0] = s1;
          sred[(w * CO_TILE + row) * 2 + 1] = s2;
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    if (tid < CO_TILE) {
      const int co = cot * CO_TILE + tid;
      if (co < p.cout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {   // fixed order: deterministic
          s1 += sred[(ww * CO_TILE + tid) * 2 + 0];
          s2 += sred[(ww * CO_TILE + tid) * 2 + 1];
        }
        p.stats[((long long)pt * p.cout + co) * 2 + 0] = s1;
        p.stats[((long long)pt * p.cout + co) * 2 + 1] = s2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Persistent, software-pipelined forward / dgrad kernel (the default).  Same tiling as igemm_kernel,
// but a workgroup walks a strided list of (tile, 32-channel chunk) stages and keeps the NEXT stage's
// input loads in flight (32*PF dwords per lane, in registers) while the current stage copies its
// weights, runs its MFMAs and stores its outputs.  With only 1-2 workgroups per CU (LDS-limited) the
// unpipelined kernel left HBM idle during compute and the matrix cores idle during staging.
// ------------------------------------------------------------------------------------------
template <int PF>
struct XPre {
  float v[PF][4][8];
  bool inb[PF];
};

template <int PF>
__device__ __forceinline__ void xpre_issue(XPre<PF>& pre, const pcuda_src& x, int n, int cin, int chunk, int in_h,
                                           int in_w, int in_shift, int in_row, int oy0, int ox0, int tw, int npix,
                                           int ngroups, int tid) {
  const int cbase = chunk * 32;
  unsigned voff[PF];
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int pix = min(tid + s * 256, npix - 1);
    const int iy = pix / tw, ix = pix - iy * tw;
    const int gy = oy0 + iy, gx = ox0 + ix;
    pre.inb[s] = (tid + s * 256 < npix) & ((unsigned)gy < (unsigned)in_h) & ((unsigned)gx < (unsigned)in_w);
    const int cy = min(max(gy, 0), in_h - 1), cx = min(max(gx, 0), in_w - 1);
    voff[s] = (unsigned)((cy >> in_shift) * in_row + (cx >> in_shift)) * 4u;
  }
  const char* b1 = (const char*)(x.p1 + (long long)n * x.sn1);
  const char* b2 = (const char*)(x.p2 + (long long)n * x.sn2);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < ngroups) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = min(cbase + g * 8 + j, cin - 1);   // wave-uniform, clamped
        const char* chan = (c < x.c1) ? b1 + (long long)c * x.sc1 * 4 : b2 + (long long)(c - x.c1) * x.sc2 * 4;
#pragma unroll
        for (int s = 0; s < PF; ++s) pre.v[s][g][j] = *(const float*)(chan + voff[s]);
      }
    }
  }
}

template <bool X3, int PF>
__device__ __forceinline__ void xpre_commit(XPre<PF>& pre, unsigned char* __restrict__ xhi,
                                            unsigned char* __restrict__ xlo, const pcuda_src& x, int cin, int chunk,
                                            int npix, int ngroups, int nwrite, int tid) {
  const int cbase = chunk * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g >= ngroups && g < nwrite) {   // channels past cin inside a 16-wide k-step: zeros, not LDS garbage
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (tid + s * 256 < npix) {
          *(uint4*)(xhi + (size_t)(tid + s * 256) * IG_REC_BYTES + g * 16) = make_uint4(0, 0, 0, 0);
          if (X3) *(uint4*)(xlo + (size_t)(tid + s * 256) * IG_REC_BYTES + g * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    if (g < ngroups) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cbase + g * 8 + j;
        const bool cok = c < cin;
        float sc = 1.f, sh = 0.f;
        if (cok) {
          if (c < x.c1) { if (x.scale1) { sc = x.scale1[c]; sh = x.shift1[c]; } }
          else { if (x.scale2) { sc = x.scale2[c - x.c1]; sh = x.shift2[c - x.c1]; } }
        }
#pragma unroll
        for (int s = 0; s < PF; ++s) {
          const float t = fmaf(pre.v[s][g][j], sc, sh);
          pre.v[s][g][j] = (pre.inb[s] & cok) ? t : 0.f;
        }
      }
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (tid + s * 256 < npix) stage_write<X3>(xhi, xlo, pre.v[s][g], tid + s * 256, g);
    }
  }
}

struct TileGeom {
  int cot, pt, n, y0, x0, oy0, ox0, th, tw, npix;
};

template <bool CLAMP, int NPB>
__device__ __forceinline__ TileGeom tile_decode(const IgemmParams& p, int L) {
  TileGeom g;
  g.cot = L % p.n_co_tiles;
  g.pt = L / p.n_co_tiles;
  const int txi = g.pt % p.tiles_x;
  const int tmp = g.pt / p.tiles_x;
  const int tyi = tmp % p.tiles_y;
  g.n = tmp / p.tiles_y;
  const int TW = p.tw, TH = p.th;
  g.y0 = tyi * TH; g.x0 = txi * TW;
  g.oy0 = g.y0 * p.in_step + p.dy_min; g.ox0 = g.x0 * p.in_step + p.dx_min;
  g.th = p.ih_t; g.tw = p.iw_t;
  if (CLAMP) {
    const int y1 = min(g.oy0 + g.th, p.in_h), x1 = min(g.ox0 + g.tw, p.in_w);
    g.oy0 = max(g.oy0, 0); g.ox0 = max(g.ox0, 0);
    g.th = max(y1 - g.oy0, 0); g.tw = max(x1 - g.ox0, 0);
  }
  g.npix = g.th * g.tw;
  return g;
}

template <bool X3, int CO_BLKS, bool CLAMP, int NPB, int PF>
__global__ __launch_bounds__(256, 2) void igemm_pipe_kernel(const IgemmParams p, const int x_cap, const int total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CO_TILE = 32 * CO_BLKS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int TW = p.tw, TPIX = p.tw * p.th;

  // XCD-aware persistent schedule: the 8 XCDs own contiguous eighths of the (pixel tile, co tile) list;
  // the workgroups of one XCD (blockIdx % 8) interleave over it, so concurrent workgroups touch
  // adjacent tiles (shared halo rows and weights hit that XCD's L2).
  const int nx = min(8, (int)gridDim.x);                          // XCD groups that actually have workgroups
  const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const int gx = ((int)gridDim.x - xcd + nx - 1) / nx;            // workgroups in this group
  const int lo = (int)((long long)total * xcd / nx), hi = (int)((long long)total * (xcd + 1) / nx);

  unsigned char* Xhi = smem;
  unsigned char* Xlo = smem + (size_t)x_cap * IG_REC_BYTES;
  unsigned char* Whi = smem + (size_t)(X3 ? 2 : 1) * x_cap * IG_REC_BYTES;
  unsigned char* Wlo = Whi + (size_t)p.tg * CO_TILE * IG_REC_BYTES;
  float* sred = (float*)smem;   // [4 waves][CO_TILE][2], reused between a tile's last MFMA and the next commit

  int pty[NPB], ptx[NPB];
  bool pvalid[NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    const int plr = w * (32 * NPB) + pb * 32 + r;
    pvalid[pb] = plr < TPIX;
    const int pl = min(plr, TPIX - 1);
    pty[pb] = IG_TY(pl, p.tmagic);
    ptx[pb] = pl - pty[pb] * TW;
  }

  f32x16 acc[CO_BLKS][NPB];
  XPre<PF> pre;

  int L = lo + slot, chunk = 0;
  bool have = L < hi;
  TileGeom g;
  if (have) {
    g = tile_decode<CLAMP, NPB>(p, L);
    xpre_issue<PF>(pre, p.x, g.n, p.cin, 0, p.in_h, p.in_w, p.in_shift, p.in_row, g.oy0, g.ox0, g.tw, g.npix,
                   (min(32, p.cin) + 7) >> 3, tid);
  }
  while (have) {
    const int cvalid = min(32, p.cin - chunk * 32);
    const int nks = cvalid > 16 ? 2 : 1;
    __syncthreads();   // every wave is done with the previous stage's X / W / reduction scratch
    xpre_commit<X3, PF>(pre, Xhi, Xlo, p.x, p.cin, chunk, g.npix, (cvalid + 7) >> 3, nks * 2, tid);
    if (CLAMP && tid < 5) {
      *(uint4*)(Xhi + (size_t)g.npix * IG_REC_BYTES + tid * 16) = make_uint4(0, 0, 0, 0);
      if (X3) *(uint4*)(Xlo + (size_t)g.npix * IG_REC_BYTES + tid * 16) = make_uint4(0, 0, 0, 0);
    }
    // next stage: its loads stay in flight through everything below
    int nL = L, nchunk = chunk + 1;
    if (nchunk == p.nchunks) { nchunk = 0; nL = L + gx; }
    const bool nhave = nL < hi;
    TileGeom ng = g;
    if (nhave) {
      if (nL != L) ng = tile_decode<CLAMP, NPB>(p, nL);
      xpre_issue<PF>(pre, p.x, ng.n, p.cin, nchunk, p.in_h, p.in_w, p.in_shift, p.in_row, ng.oy0, ng.ox0, ng.tw,
                     ng.npix, (min(32, p.cin - nchunk * 32) + 7) >> 3, tid);
    }

    if (chunk == 0) {
#pragma unroll
      for (int cb = 0; cb < CO_BLKS; ++cb)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[cb][pb][i] = 0.f;
    }
    int bbase[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
      bbase[pb] = ((pty[pb] * p.in_step) * g.tw + ptx[pb] * p.in_step) * IG_REC_BYTES + h * 16;

    for (int t0 = 0; t0 < p.ntaps; t0 += p.tg) {
      if (t0 > 0) __syncthreads();
      const int tgc = min(p.tg, p.ntaps - t0);
      {
        const long long slab = (long long)CO_TILE * IG_REC;
        const uint16_t* src = p.wpack + (((long long)g.cot * p.nchunks + chunk) * p.ntaps + t0) * slab;
        const int nvec = tgc * CO_TILE * 5;
        copy_vec16(Whi, (const uint4*)src, nvec, tid);
        if (X3) copy_vec16(Wlo, (const uint4*)(src + p.w_lo_off), nvec, tid);
      }
      __syncthreads();
      for (int tl = 0; tl < tgc; ++tl) {
        const int t = t0 + tl;
        int baddr[NPB];
        if (CLAMP) {
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) {
            const int gy = (g.y0 + pty[pb]) * p.in_step + p.dy[t];
            const int gxx = (g.x0 + ptx[pb]) * p.in_step + p.dx[t];
            const bool ok = ((unsigned)gy < (unsigned)p.in_h) & ((unsigned)gxx < (unsigned)p.in_w);
            const int idx = ok ? (gy - g.oy0) * g.tw + (gxx - g.ox0) : g.npix;
            baddr[pb] = idx * IG_REC_BYTES + h * 16;
          }
        } else {
          const int toff = ((p.dy[t] - p.dy_min) * g.tw + (p.dx[t] - p.dx_min)) * IG_REC_BYTES;
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) baddr[pb] = bbase[pb] + toff;
        }
        const int abase = (tl * CO_TILE + r) * IG_REC_BYTES + h * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (ks < nks) {
            bf16x8 ah[CO_BLKS], al[CO_BLKS], bh[NPB], bl[NPB];
#pragma unroll
            for (int cb = 0; cb < CO_BLKS; ++cb) {
              ah[cb] = lds_frag(Whi + abase + cb * 32 * IG_REC_BYTES + ks * 32);
              if (X3) al[cb] = lds_frag(Wlo + abase + cb * 32 * IG_REC_BYTES + ks * 32);
            }
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
              bh[pb] = lds_frag(Xhi + baddr[pb] + ks * 32);
              if (X3) bl[pb] = lds_frag(Xlo + baddr[pb] + ks * 32);
            }
#pragma unroll
            for (int cb = 0; cb < CO_BLKS; ++cb)
#pragma unroll
              for (int pb = 0; pb < NPB; ++pb) {
                if (X3) {
                  acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb], bh[pb], acc[cb][pb], 0, 0, 0);
                  acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bl[pb], acc[cb][pb], 0, 0, 0);
                }
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bh[pb], acc[cb][pb], 0, 0, 0);
              }
          }
        }
      }
    }

    if (chunk == p.nchunks - 1) {
      // ---- epilogue of this tile (see igemm_kernel)
      if (p.stats) __syncthreads();   // sred aliases the X tile: wait for every wave's last fragment reads
      int poff[NPB];
      bool pok[NPB];
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        const int ly = g.y0 + pty[pb], lx = g.x0 + ptx[pb];
        pok[pb] = pvalid[pb] & (ly < p.lh) & (lx < p.lw);
        poff[pb] = (ly * p.oy_mul + p.oy_off) * p.out_w + (lx * p.ox_mul + p.ox_off);
      }
      const int co0 = g.cot * CO_TILE;
      const bool uni = (p.y.c1 >= p.cout) | ((p.y.c1 & 7) == 0);
      float* const yb1 = p.y.p1 + (long long)g.n * p.y.sn1;
      float* const yb2 = p.y.p2 + (long long)g.n * p.y.sn2;
#pragma unroll
      for (int cb = 0; cb < CO_BLKS; ++cb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row0 = cb * 32 + (i & 3) + 8 * (i >> 2);
          const int row = row0 + 4 * h;
          const int co = co0 + row;
          const bool cok = co < p.cout;
          const float b = (cok && p.bias) ? p.bias[co] : 0.f;
          float* plane;
          if (uni) {
            const int cu = min(co0 + row0, p.cout - 1);
            float* base = (cu < p.y.c1) ? yb1 + (long long)cu * p.y.sc1 : yb2 + (long long)(cu - p.y.c1) * p.y.sc2;
            const long long hs = (cu < p.y.c1) ? p.y.sc1 : p.y.sc2;
            plane = base + (h ? 4 * hs : 0);
          } else {
            const int cc = min(co, p.cout - 1);
            plane = (cc < p.y.c1) ? yb1 + (long long)cc * p.y.sc1 : yb2 + (long long)(cc - p.y.c1) * p.y.sc2;
          }
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) {
            float v = acc[cb][pb][i] + b;
            v = v > 0.f ? v : v * p.slope;
            float* dst = plane + poff[pb];
            if (cok & pok[pb]) {
              if (p.accumulate) v += *dst;
              *dst = v;
              s1 += v;
              s2 += v * v;
            }
          }
          if (p.stats) {
            s1 = half_wave_sum(s1);
            s2 = half_wave_sum(s2);
            if (r == 0) {
              sred[(w * CO_TILE + row) * 2 + 0] = s1;
              sred[(w * CO_TILE + row) * 2 + 1] = s2;
            }
          }
        }
      }
      if (p.stats) {
        __syncthreads();
        if (tid < CO_TILE) {
          const int co = g.cot * CO_TILE + tid;
          if (co < p.cout) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
              s1 += sred[(ww * CO_TILE + tid) * 2 + 0];
              s2 += sred[(ww * CO_TILE + tid) * 2 + 1];
            }
            p.stats[((long long)g.pt * p.cout + co) * 2 + 0] = s1;
            p.stats[((long long)g.pt * p.cout + co) * 2 + 1] = s2;
          }
        }
      }
    }
    L = nL; chunk = nchunk; g = ng; have = nhave;
  }
}

// ------------------------------------------------------------------------------------------
// wgrad kernel.  Block = (co-tile, 32-channel chunk, tap group) x split-K slice; loops over its
// 128-pixel tiles, keeping dW tiles [32 rows][32 ci] per tap in the accumulators.
// ------------------------------------------------------------------------------------------
#define WG_ZROW 272   // bytes per dZ row in LDS: 128 px bf16 + 16 pad (17*16: conflict-free b128)

__device__ __forceinline__ bf16x8 lds_tr_frag(const unsigned char* p0, const unsigned char* p1) {
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)p0);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)p1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool X3, int CO_BLKS, bool CLAMP, int TAPS_MAX>
__global__ __launch_bounds__(256, (TAPS_MAX <= 9 ? 2 : 1)) void wgrad_kernel(const WgradParams p, const int x_cap, float* db_partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CO_TILE = 32 * CO_BLKS;
  constexpr int NWT = 4 / CO_BLKS;             // waves sharing one row block
  constexpr int MAXT = (TAPS_MAX + NWT - 1) / NWT;   // taps per wave (block handles <= TAPS_MAX taps)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;

  const int bx = blockIdx.x;
  const int tgidx = bx % p.tap_groups;
  const int chunk = (bx / p.tap_groups) % p.n_chunks;
  const int cot = bx / (p.tap_groups * p.n_chunks);
  const int kslice = blockIdx.y;
  const int t_begin = tgidx * p.ntaps;
  const int tcount = min(p.ntaps, p.ntaps_total - t_begin);
  const int cb = w % CO_BLKS, wsub = w / CO_BLKS;
  const int share = (tcount + NWT - 1) / NWT;
  const int my_t0 = t_begin + wsub * share;
  const int my_cnt = max(0, min(share, tcount - wsub * share));

  unsigned char* Xhi = smem;
  unsigned char* Xlo = smem + (size_t)x_cap * IG_REC_BYTES;
  unsigned char* Zhi = smem + (size_t)(X3 ? 2 : 1) * x_cap * IG_REC_BYTES;
  unsigned char* Zlo = Zhi + (size_t)CO_TILE * WG_ZROW;

  const int TW = p.tw, TH = p.th, TPIX = TW * TH;
  const int ntiles = p.n * p.tiles_y * p.tiles_x;
  const int tile_lo = (int)((long long)kslice * ntiles / p.ksplit);
  const int tile_hi = (int)((long long)(kslice + 1) * ntiles / p.ksplit);
  const bool do_db = (db_partial != nullptr) && chunk == 0 && tgidx == 0;

  f32x16 acc[MAXT];
#pragma unroll
  for (int ti = 0; ti < MAXT; ++ti)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ti][i] = 0.f;
  float dbacc[CO_TILE / 16];
#pragma unroll
  for (int j = 0; j < CO_TILE / 16; ++j) dbacc[j] = 0.f;

  // transposed-read lane roles: 16-lane group g -> k half (g>>1), column block (g&1)
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int cvalid = min(32, p.cin - chunk * 32);

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int txi = tile % p.tiles_x;
    const int tmp = tile / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    const int n = tmp / p.tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    int oy0 = y0 * p.stride + p.dy_min, ox0 = x0 * p.stride + p.dx_min;
    int th = p.ih_t, tw = p.iw_t;
    if (CLAMP) {   // LDS tile = halo tile clipped to the image (+ one zero record), as in igemm_kernel
      const int y1 = min(oy0 + th, p.in_h), x1 = min(ox0 + tw, p.in_w);
      oy0 = max(oy0, 0); ox0 = max(ox0, 0);
      th = max(y1 - oy0, 0); tw = max(x1 - ox0, 0);
    }
    const int npix = th * tw;
    __syncthreads();
    if (!(p.dbg & 16))
      stage_x_chunk<X3, 1>(Xhi, Xlo, p.x, n, p.cin, chunk, p.in_h, p.in_w, p.in_shift, p.in_row, oy0, ox0, th, tw,
                           (cvalid + 7) >> 3, 4, tid);
    if (CLAMP && tid < 5) {
      *(uint4*)(Xhi + (size_t)npix * IG_REC_BYTES + tid * 16) = make_uint4(0, 0, 0, 0);
      if (X3) *(uint4*)(Xlo + (size_t)npix * IG_REC_BYTES + tid * 16) = make_uint4(0, 0, 0, 0);
    }
    // dZ tile [CO_TILE][128 px] in natural (pixel-contiguous) order; all loads first, then the writes
    float zv[CO_TILE / 16][8];
    if (p.dbg & 32) {
#pragma unroll
      for (int j = 0; j < CO_TILE / 16; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) zv[j][e] = 1.f;
    } else
#pragma unroll
    for (int j = 0; j < CO_TILE / 16; ++j) {
      const int item = tid + 256 * j;
      const int row = item >> 4, oct = item & 15;
      const int co = cot * CO_TILE + row;
      const int pl = oct * 8;
      const int cco = min(co, p.cout - 1);
      const float* planep = p.dz + (long long)n * p.dz_sn + (long long)cco * p.dz_sc;
      if (p.aligned4) {   // wave-uniform: TW % 8 == 0 and out_w % 8 == 0 -> an octet is inside or outside a row as a whole
        const int ty = IG_TY(pl, p.tmagic), tx = pl - ty * TW;
        const int oy = y0 + ty, ox = x0 + tx;
        const bool ok = (co < p.cout) & (pl < TPIX) & (oy < p.out_h) & (ox + 8 <= p.out_w);
        const float* rowp = planep + (long long)min(oy, p.out_h - 1) * p.out_w + min(ox, p.out_w - 8);
        const float4 a = *(const float4*)rowp, b = *(const float4*)(rowp + 4);
        zv[j][0] = ok ? a.x : 0.f; zv[j][1] = ok ? a.y : 0.f; zv[j][2] = ok ? a.z : 0.f; zv[j][3] = ok ? a.w : 0.f;
        zv[j][4] = ok ? b.x : 0.f; zv[j][5] = ok ? b.y : 0.f; zv[j][6] = ok ? b.z : 0.f; zv[j][7] = ok ? b.w : 0.f;
      } else {            // any tile width: the 8 pixels of an octet may wrap to the next tile row
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ple = min(pl + e, TPIX - 1);
          const int ty = IG_TY(ple, p.tmagic), tx = ple - ty * TW;
          const int oy = y0 + ty, ox = x0 + tx;
          const float t = planep[(long long)min(oy, p.out_h - 1) * p.out_w + min(ox, p.out_w - 1)];
          zv[j][e] = ((co < p.cout) & (pl + e < TPIX) & (oy < p.out_h) & (ox < p.out_w)) ? t : 0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < CO_TILE / 16; ++j) {
      const int item = tid + 256 * j;
      const int row = item >> 4, oct = item & 15;
      const float(&v)[8] = zv[j];
      if (do_db) dbacc[j] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      uint4 hi, lo;
      if (X3) {
        split2(v[0], v[1], hi.x, lo.x); split2(v[2], v[3], hi.y, lo.y);
        split2(v[4], v[5], hi.z, lo.z); split2(v[6], v[7], hi.w, lo.w);
        *(uint4*)(Zlo + row * WG_ZROW + oct * 16) = lo;
      } else {
        hi.x = pack_bf16x2(v[0], v[1]); hi.y = pack_bf16x2(v[2], v[3]);
        hi.z = pack_bf16x2(v[4], v[5]); hi.w = pack_bf16x2(v[6], v[7]);
      }
      *(uint4*)(Zhi + row * WG_ZROW + oct * 16) = hi;
    }
    __syncthreads();

#pragma unroll 1
    for (int ks = 0; ks < ((p.dbg & 64) ? 0 : 8); ++ks) {
      const int aoff = (cb * 32 + r) * WG_ZROW + ks * 32 + h * 16;
      const bf16x8 ah = lds_frag(Zhi + aoff);
      bf16x8 al;
      if (X3) al = lds_frag(Zlo + aoff);
      int rowb[2], rty[2], rtx[2];
      const int colb = ((g & 1) * 16 + 4 * tp) * 2;
#pragma unroll
      for (int sel = 0; sel < 2; ++sel) {
        const int pl = min(ks * 16 + 8 * (g >> 1) + 4 * sel + tq, TPIX - 1);   // idle slots carry dZ = 0
        rty[sel] = IG_TY(pl, p.tmagic); rtx[sel] = pl - rty[sel] * TW;
        rowb[sel] = ((rty[sel] * p.stride) * p.iw_t + rtx[sel] * p.stride) * IG_REC_BYTES + colb;
      }
#pragma unroll
      for (int ti = 0; ti < MAXT; ++ti) {
        if (ti < my_cnt) {   // wave-uniform: EXEC stays all ones for the transposed reads
          const int t = my_t0 + ti;
          int r0, r1;
          if (CLAMP) {
            int ra[2];
#pragma unroll
            for (int sel = 0; sel < 2; ++sel) {
              const int gy = (y0 + rty[sel]) * p.stride + p.dy[t], gx = (x0 + rtx[sel]) * p.stride + p.dx[t];
              const bool ok = ((unsigned)gy < (unsigned)p.in_h) & ((unsigned)gx < (unsigned)p.in_w);
              ra[sel] = (ok ? (gy - oy0) * tw + (gx - ox0) : npix) * IG_REC_BYTES + colb;
            }
            r0 = ra[0]; r1 = ra[1];
          } else {
            const int toff = ((p.dy[t] - p.dy_min) * p.iw_t + (p.dx[t] - p.dx_min)) * IG_REC_BYTES;
            r0 = rowb[0] + toff; r1 = rowb[1] + toff;
          }
          const bf16x8 bh = lds_tr_frag(Xhi + r0, Xhi + r1);
          if (X3) {
            const bf16x8 bl = lds_tr_frag(Xlo + r0, Xlo + r1);
            acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ti], 0, 0, 0);
            acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ti], 0, 0, 0);
          }
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ti], 0, 0, 0);
        }
      }
    }
  }

  // partial slabs: partial[kslice][tap][co][ci] -- ci (the lane index) innermost, so every accumulator
  // register stores two 128-B segments; the OIHW transpose happens once, in the reduce kernel
#pragma unroll
  for (int ti = 0; ti < MAXT; ++ti) {
    if (ti < my_cnt) {
      const int t = my_t0 + ti;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = cot * CO_TILE + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int ci = chunk * 32 + r;
        if (co < p.cout && ci < p.cin)
          p.partial[(((long long)kslice * p.ntaps_total + t) * p.cout + co) * p.cin + ci] = acc[ti][i];
      }
    }
  }
  if (do_db) {
#pragma unroll
    for (int j = 0; j < CO_TILE / 16; ++j) {
      float s = dbacc[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const int row = (tid + 256 * j) >> 4;
      const int co = cot * CO_TILE + row;
      if ((tid & 15) == 0 && co < p.cout) db_partial[(long long)kslice * p.cout + co] = s;
    }
  }
}

// dw[i] (+)= sum_k partial[k][i], deterministic.  A workgroup owns 64 consecutive outputs; its
// blockDim/64 k-groups each sum a strided subset of the slices (8 loads in flight per lane), and the
// k-group partials are combined in a fixed order through LDS.  (One thread per output with a serial
// loop over up to 1024 slices was latency-bound: 0.5 ms for a 9K-weight layer.)
// ntaps > 1: partial is [k][tap][co*cin] and dw is [co*cin][tap] (OIHW): coalesced reads, the
// transpose costs one scattered write per weight.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, long long numel,
                                                           int ksplit, float* __restrict__ dw, int accumulate,
                                                           int ntaps) {
  __shared__ float sh[16][64];
  const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6, nkg = blockDim.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + lane;
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

// ==========================================================================================
// host side
// ==========================================================================================
namespace {

const int LDS_HARD = 160 * 1024;

bool geom_ok(const pcuda_conv_geom* g) {
  if (!g || g->n <= 0 || g->cin <= 0 || g->cout <= 0 || g->k <= 0 || g->stride <= 0 || g->dil <= 0 || g->pad < 0)
    return false;
  if (g->k * g->k > IG_MAX_TAPS) return false;
  const int oh = (g->in_h + 2 * g->pad - g->dil * (g->k - 1) - 1) / g->stride + 1;
  const int ow = (g->in_w + 2 * g->pad - g->dil * (g->k - 1) - 1) / g->stride + 1;
  if (oh != g->out_h || ow != g->out_w || oh <= 0 || ow <= 0) return false;
  if (g->in_up && ((g->in_h & 1) || (g->in_w & 1))) return false;
  return true;
}

bool src_ok(const pcuda_src* s, int c) {
  if (!s || !s->p1 || s->c1 <= 0 || s->c1 > c) return false;
  if (s->c1 < c && !s->p2) return false;
  return true;
}
bool dst_ok(const pcuda_dst* d, int c) {
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

TapSet fwd_taps(const pcuda_conv_geom* g) {
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

inline int posmod(int a, int m) { return ((a % m) + m) % m; }

// taps of the dgrad parity class (ry, rx): dX[s*m + ry] = sum_ky dY[m + (ry + pad - ky*dil)/s] w[ky]
TapSet dgrad_taps(const pcuda_conv_geom* g, int ry, int rx) {
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

size_t packed_elems(int rows, int red, int ntaps) {   // bf16 elements of ONE plane (hi)
  const int co_tile = 32 * ig_co_blks(rows);
  const int n_co_tiles = cdiv(rows, co_tile), nchunks = cdiv(red, 32);
  return (size_t)n_co_tiles * nchunks * ntaps * co_tile * IG_REC;
}

int launch_pack(const float* w, uint16_t* out, int prec, int rows, int red, long long s_row, long long s_red,
                const TapSet& taps, hipStream_t s) {
  if (taps.n == 0) return PCUDA_OK;
  PackParams p;
  p.w = w; p.out = out;
  p.rows = rows; p.red = red; p.s_row = s_row; p.s_red = s_red;
  p.ntaps = taps.n;
  memcpy(p.tap_src, taps.src, sizeof(p.tap_src));
  p.co_tile = 32 * ig_co_blks(rows);
  p.n_co_tiles = cdiv(rows, p.co_tile);
  p.nchunks = cdiv(red, 32);
  const size_t plane = packed_elems(rows, red, taps.n);
  p.lo_off = prec == PCUDA_PREC_BF16X3 ? (long long)plane : 0;
  const int blocks = (int)((plane + 255) / 256 > 4096 ? 4096 : (plane + 255) / 256);
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, s, p);
  PCUDA_CHECK_LAUNCH("pack_kernel");
  return PCUDA_OK;
}

// pick taps-per-group and LDS size; returns <0 when nothing fits
int plan_lds(bool x3, int co_tile, int x_cap, int ntaps, int* tg_out, size_t* lds_out) {
  const size_t mul = x3 ? 2 : 1;
  const size_t xb = (size_t)x_cap * IG_REC_BYTES * mul;
  const size_t wtap = (size_t)co_tile * IG_REC_BYTES * mul;
  const size_t epi = 4 * (size_t)co_tile * 2 * sizeof(float);
  if (ntaps < 1) ntaps = 1;
  // budgets: 3, 2, 1 workgroups per CU (160 KiB LDS)
  const size_t budgets[3] = {54528, 81920, 163840};
  for (int b = 0; b < 3; ++b) {
    if (xb >= budgets[b]) continue;
    const int fit = (int)((budgets[b] - xb) / wtap);
    const int need = b == 0 ? (ntaps < 3 ? ntaps : 3) : 1;
    if (fit < need) continue;
    const int tg = fit > ntaps ? ntaps : fit;
    size_t total = xb + (size_t)tg * wtap;
    if (total < epi) total = epi;
    *tg_out = tg; *lds_out = total;
    return 0;
  }
  return -1;
}

struct IgemmPlan {
  int npb, tw, th, tiles_x, tiles_y, ih_t, iw_t, clamp, x_cap, tg;
  size_t lds;
};

// candidate output-tile widths: powers of two plus even splits of the row (so a 17- or 33-wide map is
// not padded to 32 / 64)
int tile_width_candidates(int lw, int tile_px, int* out) {
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

// tile shape / LDS plan of one generic launch: depends only on geometry, taps and precision.
// Minimises the number of MFMA pixel slots (tiles x slots per tile); ties prefer 32-pixel-aligned rows
// (128-B output segments), two pixel blocks per wave, wider tiles.
int plan_igemm(int rows, int lh, int lw, int in_h, int in_w, int in_step, const TapSet& taps, bool x3,
               IgemmPlan* best) {
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
        ok = plan_lds(x3, co_tile, pl.x_cap, taps.n, &pl.tg, &pl.lds) == 0;
        if (pl.clamp) break;
      }
      if (!ok) continue;
      // primary: MFMA pixel slots; then 32-pixel-aligned rows (128-B output segments: measured faster than
      // 16x16 tiles despite their smaller halo); then staged input pixels per output slot (halo overhead,
      // in 1/64ths); then two pixel blocks per wave
      const long long slots = (long long)pl.tiles_x * pl.tiles_y * TP;
      const long long halo = (long long)pl.x_cap * 64 / TP;
      const long long key = (slots << 24) + ((pl.tw & 31) ? (1ll << 20) : 0) + (halo << 8) + (npb == 1 ? 1 : 0);
      if (best_key < 0 || key < best_key) { best_key = key; *best = pl; }
    }
  }
  return best_key < 0 ? -1 : 0;
}

template <bool X3, int CO_BLKS, bool CLAMP, int NPB>
int launch_igemm_t(const IgemmParams& p, int x_cap, size_t lds, hipStream_t s) {
  auto kern = igemm_kernel<X3, CO_BLKS, CLAMP, NPB>;
  static size_t lds_set = 0;
  if (lds > 32 * 1024 && lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD);
    if (e != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "igemm: cannot raise dynamic LDS to %d: %s", LDS_HARD, hipGetErrorString(e));
    lds_set = LDS_HARD;
  }
  const int grid = p.n_co_tiles * p.n * p.tiles_x * p.tiles_y;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p, x_cap);
  PCUDA_CHECK_LAUNCH("igemm_kernel");
  return PCUDA_OK;
}

template <bool X3, int CO_BLKS>
int launch_igemm_c(const IgemmParams& p, const IgemmPlan& pl, hipStream_t s) {
  if (pl.clamp) return pl.npb == 2 ? launch_igemm_t<X3, CO_BLKS, true, 2>(p, pl.x_cap, pl.lds, s)
                                   : launch_igemm_t<X3, CO_BLKS, true, 1>(p, pl.x_cap, pl.lds, s);
  return pl.npb == 2 ? launch_igemm_t<X3, CO_BLKS, false, 2>(p, pl.x_cap, pl.lds, s)
                     : launch_igemm_t<X3, CO_BLKS, false, 1>(p, pl.x_cap, pl.lds, s);
}

template <bool X3, int CO_BLKS, bool CLAMP, int NPB, int PF>
int launch_pipe_t(const IgemmParams& p, const IgemmPlan& pl, hipStream_t s) {
  auto kern = igemm_pipe_kernel<X3, CO_BLKS, CLAMP, NPB, PF>;
  static size_t lds_set = 0;
  if (pl.lds > 32 * 1024 && pl.lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD);
    if (e != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "igemm_pipe: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    lds_set = LDS_HARD;
  }
  const int total = p.n_co_tiles * p.n * p.tiles_x * p.tiles_y;
  // persistent grid = what is resident at once (registers and LDS both limit it)
  static int occ_cache[4] = {0, 0, 0, 0};   // by LDS class: <=53K, <=80K, <=160K
  const int cls = pl.lds <= 54528 ? 0 : (pl.lds <= 81920 ? 1 : 2);
  if (occ_cache[cls] == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, pl.lds) != hipSuccess || nb < 1) nb = 1;
    const int lds_lim = (int)((size_t)LDS_HARD / pl.lds);
    if (nb > lds_lim) nb = lds_lim;
    if (nb > 4) nb = 4;
    if (nb < 1) nb = 1;
    occ_cache[cls] = nb;
  }
  int grid = occ_cache[cls] * 256;
  if (grid > total) grid = total;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), pl.lds, s, p, pl.x_cap, total);
  PCUDA_CHECK_LAUNCH("igemm_pipe_kernel");
  return PCUDA_OK;
}

template <bool X3, int CO_BLKS, bool CLAMP, int NPB>
int launch_pipe_pf(const IgemmParams& p, const IgemmPlan& pl, int pf, hipStream_t s) {
  if (pf == 1) return launch_pipe_t<X3, CO_BLKS, CLAMP, NPB, 1>(p, pl, s);
  if (pf == 2) return launch_pipe_t<X3, CO_BLKS, CLAMP, NPB, 2>(p, pl, s);
  return launch_pipe_t<X3, CO_BLKS, CLAMP, NPB, 3>(p, pl, s);
}

template <bool X3, int CO_BLKS>
int launch_pipe_c(const IgemmParams& p, const IgemmPlan& pl, int pf, hipStream_t s) {
  if (pl.clamp) return pl.npb == 2 ? launch_pipe_pf<X3, CO_BLKS, true, 2>(p, pl, pf, s)
                                   : launch_pipe_pf<X3, CO_BLKS, true, 1>(p, pl, pf, s);
  return pl.npb == 2 ? launch_pipe_pf<X3, CO_BLKS, false, 2>(p, pl, pf, s)
                     : launch_pipe_pf<X3, CO_BLKS, false, 1>(p, pl, pf, s);
}

int launch_igemm(IgemmParams& p, int prec, const TapSet& taps, hipStream_t s) {
  const bool x3 = prec == PCUDA_PREC_BF16X3;
  const int co_blks = ig_co_blks(p.cout);
  const int co_tile = 32 * co_blks;
  IgemmPlan pl;
  if (plan_igemm(p.cout, p.lh, p.lw, p.in_h, p.in_w, p.in_step, taps, x3, &pl) < 0)
    PCUDA_FAIL(PCUDA_E_UNSUPPORTED, "igemm: no tile of this convolution fits LDS (in_step %d, tap span %d)",
               p.in_step, taps.dy_max - taps.dy_min);
  p.n_co_tiles = cdiv(p.cout, co_tile);
  p.nchunks = cdiv(p.cin, 32);
  p.tw = pl.tw; p.th = pl.th; p.tmagic = 65536 / pl.tw + 1; p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y;
  p.ntaps = taps.n;
  memcpy(p.dy, taps.dy, sizeof(p.dy));
  memcpy(p.dx, taps.dx, sizeof(p.dx));
  p.dy_min = taps.dy_min; p.dx_min = taps.dx_min;
  p.ih_t = pl.ih_t; p.iw_t = pl.iw_t; p.clamp = pl.clamp; p.tg = pl.tg;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("PCUDA_DBG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
  }
  const double flops = 2.0 * p.n * (double)p.lh * p.lw * p.cout * (double)p.cin * taps.n;
  char tag[160];
  snprintf(tag, sizeof(tag), "igemm n%d red%d rows%d %dx%d taps%d step%d up%d tw%d npb%d clamp%d tg%d lds%zu", p.n, p.cin,
           p.cout, p.lh, p.lw, taps.n, p.in_step, p.in_shift, pl.tw, pl.npb, pl.clamp, pl.tg, pl.lds);
  ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
  static int nopipe = -1;
  if (nopipe < 0) { const char* e = getenv("PCUDA_NOPIPE"); nopipe = e ? atoi(e) : 0; }
  const int max_pix = pl.clamp ? pl.x_cap - 1 : pl.ih_t * pl.iw_t;     // largest LDS tile of this launch
  const int pf = (max_pix + 255) / 256;
  // PF = 3 keeps 96 prefetch registers live next to the accumulators: only with one pixel block per wave
  if (!nopipe && p.ntaps > 0 && (pf <= 2 || (pf == 3 && pl.npb == 1))) {
    if (x3) return co_blks == 2 ? launch_pipe_c<true, 2>(p, pl, pf, s) : launch_pipe_c<true, 1>(p, pl, pf, s);
    return co_blks == 2 ? launch_pipe_c<false, 2>(p, pl, pf, s) : launch_pipe_c<false, 1>(p, pl, pf, s);
  }
  if (x3) return co_blks == 2 ? launch_igemm_c<true, 2>(p, pl, s) : launch_igemm_c<true, 1>(p, pl, s);
  return co_blks == 2 ? launch_igemm_c<false, 2>(p, pl, s) : launch_igemm_c<false, 1>(p, pl, s);
}

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
      const long long key = ((long long)cdiv(g->out_w, tw) * cdiv(g->out_h, th) << 24) + ((tw & 31) ? (1ll << 20) : 0) + halo;
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
  long long ks = 1024 / base;
  if (ks < 1) ks = 1;
  if (ks > ntiles) ks = ntiles;
  // keep the partial slabs (written once, re-read once by the reduce kernel) below ~64 MB (measured:
  // 32 MB costs the wgrad kernels more parallelism than the reduce kernel saves)
  const long long welems = (long long)g->cout * g->cin * ntaps;
  while (ks > 1 && ks * welems * 4 > (64ll << 20)) ks >>= 1;
  w.ksplit = (int)ks;
  return w;
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" size_t pcuda_conv2d_packed_fwd_bytes(const pcuda_conv_geom* g, int prec) {
  if (!geom_ok(g)) return 0;
  return packed_elems(g->cout, g->cin, g->k * g->k) * 2 * (prec == PCUDA_PREC_BF16X3 ? 2 : 1);
}

extern "C" size_t pcuda_conv2d_packed_dgrad_bytes(const pcuda_conv_geom* g, int prec) {
  if (!geom_ok(g)) return 0;
  size_t tot = 0;
  for (int ry = 0; ry < g->stride; ++ry)
    for (int rx = 0; rx < g->stride; ++rx) {
      TapSet t = dgrad_taps(g, ry, rx);
      tot += packed_elems(g->cin, g->cout, t.n) * 2 * (prec == PCUDA_PREC_BF16X3 ? 2 : 1);
    }
  return tot;
}

extern "C" int pcuda_conv2d_pack_fwd(const pcuda_conv_geom* g, int prec, const float* w, void* packed,
                                     pcuda_stream_t s) {
  if (!geom_ok(g) || !w || !packed) PCUDA_FAIL(PCUDA_E_BADARG, "pack_fwd: bad geometry or null pointer");
  const int kk = g->k * g->k;
  TapSet t = fwd_taps(g);
  return launch_pack(w, (uint16_t*)packed, prec, g->cout, g->cin, (long long)g->cin * kk, kk, t, (hipStream_t)s);
}

extern "C" int pcuda_conv2d_pack_dgrad(const pcuda_conv_geom* g, int prec, const float* w, void* packed,
                                       pcuda_stream_t s) {
  if (!geom_ok(g) || !w || !packed) PCUDA_FAIL(PCUDA_E_BADARG, "pack_dgrad: bad geometry or null pointer");
  const int kk = g->k * g->k;
  uint16_t* out = (uint16_t*)packed;
  for (int ry = 0; ry < g->stride; ++ry)
    for (int rx = 0; rx < g->stride; ++rx) {
      TapSet t = dgrad_taps(g, ry, rx);
      int rc = launch_pack(w, out, prec, g->cin, g->cout, kk, (long long)g->cin * kk, t, (hipStream_t)s);
      if (rc) return rc;
      out += packed_elems(g->cin, g->cout, t.n) * (prec == PCUDA_PREC_BF16X3 ? 2 : 1);
    }
  return PCUDA_OK;
}

extern "C" int pcuda_conv2d_fwd_tiles(const pcuda_conv_geom* g, int prec) {
  if (!geom_ok(g)) return 0;
  TapSet t = fwd_taps(g);
  IgemmPlan pl;
  if (plan_igemm(g->cout, g->out_h, g->out_w, g->in_h, g->in_w, g->stride, t, prec == PCUDA_PREC_BF16X3, &pl) < 0)
    return 0;
  return g->n * pl.tiles_x * pl.tiles_y;
}

extern "C" int pcuda_conv2d_forward(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const void* packed_w,
                                    const float* bias, float slope, const pcuda_dst* y, float* bn_partials,
                                    pcuda_stream_t s) {
  if (!geom_ok(g)) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_forward: inconsistent geometry");
  if (!src_ok(x, g->cin) || !dst_ok(y, g->cout) || !packed_w) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_forward: bad tensors");
  if (prec != PCUDA_PREC_BF16X3 && prec != PCUDA_PREC_BF16) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_forward: bad precision");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = *x; p.cin = g->cin;
  p.in_h = g->in_h; p.in_w = g->in_w; p.in_shift = g->in_up ? 1 : 0; p.in_row = g->in_w >> p.in_shift;
  p.y = *y; p.cout = g->cout; p.out_w = g->out_w;
  p.lh = g->out_h; p.lw = g->out_w;
  p.oy_mul = p.ox_mul = 1; p.oy_off = p.ox_off = 0;
  p.in_step = g->stride;
  p.wpack = (const uint16_t*)packed_w;
  p.w_lo_off = (long long)packed_elems(g->cout, g->cin, g->k * g->k);
  p.bias = bias; p.slope = slope; p.accumulate = 0; p.stats = bn_partials;
  p.n = g->n;
  TapSet t = fwd_taps(g);
  return launch_igemm(p, prec, t, (hipStream_t)s);
}

extern "C" int pcuda_conv2d_dgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                                  const pcuda_dst* dx, int accumulate, pcuda_stream_t s) {
  if (!geom_ok(g)) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad: inconsistent geometry");
  if (!src_ok(dy, g->cout) || !dst_ok(dx, g->cin) || !packed_w_dgrad) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad: bad tensors");
  if (prec != PCUDA_PREC_BF16X3 && prec != PCUDA_PREC_BF16) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_dgrad: bad precision");
  const uint16_t* wp = (const uint16_t*)packed_w_dgrad;
  const int st = g->stride;
  for (int ry = 0; ry < st; ++ry)
    for (int rx = 0; rx < st; ++rx) {
      TapSet t = dgrad_taps(g, ry, rx);
      const size_t plane = packed_elems(g->cin, g->cout, t.n);
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
        p.wpack = wp; p.w_lo_off = (long long)plane;
        p.bias = nullptr; p.slope = 1.f; p.accumulate = accumulate; p.stats = nullptr;
        p.n = g->n;
        int rc = launch_igemm(p, prec, t, (hipStream_t)s);
        if (rc) return rc;
      }
      wp += plane * (prec == PCUDA_PREC_BF16X3 ? 2 : 1);
    }
  return PCUDA_OK;
}

extern "C" size_t pcuda_conv2d_wgrad_workspace_size(const pcuda_conv_geom* g) {
  if (!geom_ok(g)) return 0;
  WgradPlan w = plan_wgrad(g);
  return ((size_t)w.ksplit * g->cout * g->cin * g->k * g->k + (size_t)w.ksplit * g->cout) * sizeof(float) + 256;
}

template <bool X3, int CO_BLKS, bool CLAMP, int TAPS_MAX>
static int launch_wgrad_t(const WgradParams& p, int x_cap, size_t lds, float* dbp, dim3 grid, hipStream_t s) {
  auto kern = wgrad_kernel<X3, CO_BLKS, CLAMP, TAPS_MAX>;
  static size_t lds_set = 0;
  if (lds > 32 * 1024 && lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HARD);
    if (e != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "wgrad: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    lds_set = LDS_HARD;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p, x_cap, dbp);
  PCUDA_CHECK_LAUNCH("wgrad_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_conv2d_wgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy,
                                  long long dy_sn, long long dy_sc, float* dw, float* db, int accumulate,
                                  void* workspace, size_t workspace_bytes, pcuda_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!geom_ok(g)) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_wgrad: inconsistent geometry");
  if (!src_ok(x, g->cin) || !dy || !dw) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_wgrad: bad tensors");
  if (prec != PCUDA_PREC_BF16X3 && prec != PCUDA_PREC_BF16) PCUDA_FAIL(PCUDA_E_BADARG, "conv2d_wgrad: bad precision");
  if (!workspace || workspace_bytes < pcuda_conv2d_wgrad_workspace_size(g))
    PCUDA_FAIL(PCUDA_E_WORKSPACE, "conv2d_wgrad: workspace too small (%zu < %zu)", workspace_bytes,
               pcuda_conv2d_wgrad_workspace_size(g));
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
  p.aligned4 = ((g->out_w & 7) == 0 && (w.tw & 7) == 0 && (dy_sn & 3) == 0 && (dy_sc & 3) == 0 && (((uintptr_t)dy) & 15) == 0) ? 1 : 0;
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
    const double flops = 2.0 * g->n * (double)g->out_h * g->out_w * g->cout * (double)g->cin * t.n;
    char tag[160];
    snprintf(tag, sizeof(tag), "wgrad n%d cin%d cout%d %dx%d k%d s%d d%d ksplit%d clamp%d lds%zu", g->n, g->cin, g->cout,
             g->out_h, g->out_w, g->k, g->stride, g->dil, w.ksplit, clamp ? 1 : 0, lds);
    ProfScope prof(PCUDA_FAM_CONV_WGRAD, flops, s, tag);
    int rc;
#define WG_GO2(X3_, CB_, CL_) (w.taps_per_group > 9 ? launch_wgrad_t<X3_, CB_, CL_, 16>(p, x_cap, lds, dbp, grid, s) : launch_wgrad_t<X3_, CB_, CL_, 9>(p, x_cap, lds, dbp, grid, s))
#define WG_GO(X3_, CB_) (clamp ? WG_GO2(X3_, CB_, true) : WG_GO2(X3_, CB_, false))
    if (x3) rc = w.co_blks == 2 ? WG_GO(true, 2) : WG_GO(true, 1);
    else rc = w.co_blks == 2 ? WG_GO(false, 2) : WG_GO(false, 1);
#undef WG_GO
#undef WG_GO2
    if (rc) return rc;
  }
  {
    int nkg = 1;
    while (nkg < 16 && nkg * 2 <= w.ksplit) nkg <<= 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(welems, 64)), dim3(64 * nkg), 0, s,
                       (const float*)workspace, welems, w.ksplit, dw, accumulate, t.n);
    PCUDA_CHECK_LAUNCH("wgrad_reduce_kernel");
    if (db) {
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(g->cout, 64)), dim3(64 * nkg), 0, s, (const float*)dbp,
                         (long long)g->cout, w.ksplit, db, accumulate, 1);
      PCUDA_CHECK_LAUNCH("wgrad_reduce_kernel(db)");
    }
  }
  return PCUDA_OK;
}
