// Device-side staging helpers shared by the forward/dgrad and the wgrad kernels.
#pragma once
#include "conv_igemm.h"

// ------------------------------------------------------------------------------------------
// stage one 32-channel chunk of a haloed input tile into LDS, pixel-major, bf16 hi (+ lo)
// ------------------------------------------------------------------------------------------
// REC = bytes per pixel record: 80 (one plane per record: bf16 mode, and both planes of the weight-gradient kernels)
// or 144 (bf16x3 forward / dgrad: hi 64 B | lo 64 B | 16 pad in ONE record, xlo = xhi + 64)
template <bool X3, int REC = IG_REC_BYTES>
__device__ __forceinline__ void stage_write(unsigned char* xhi, unsigned char* xlo,
                                            const float (&v)[8], int pix, int g) {
  uint4 hi, lo;
  if (X3) {
    split2(v[0], v[1], hi.x, lo.x); split2(v[2], v[3], hi.y, lo.y);
    split2(v[4], v[5], hi.z, lo.z); split2(v[6], v[7], hi.w, lo.w);
    *(uint4*)(xlo + (size_t)pix * REC + g * 16) = lo;
  } else {
    hi.x = pack_bf16x2(v[0], v[1]); hi.y = pack_bf16x2(v[2], v[3]);
    hi.z = pack_bf16x2(v[4], v[5]); hi.w = pack_bf16x2(v[6], v[7]);
  }
  *(uint4*)(xhi + (size_t)pix * REC + g * 16) = hi;
}

// wave-uniform channel -> plane pointer / lazy-BatchNorm affine of a two-source input.  Selects on
// scalars (no branches per channel: the branchy form compiled to ~25 scalar instructions and three
// branches per channel).
__device__ __forceinline__ const char* src_chan(const pcuda_src& x, const char* b1, const char* b2, int c) {
  const bool first = c < x.c1;
  const char* cb = first ? b1 : b2;
  return cb + (long long)(first ? c : c - x.c1) * (first ? x.sc1 : x.sc2) * 4;
}
__device__ __forceinline__ void src_affine(const pcuda_src& x, int c, float& sc, float& sh) {
  const bool first = c < x.c1;
  const float* scp = first ? x.scale1 : x.scale2;
  const float* shp = first ? x.shift1 : x.shift2;
  const int cc = first ? c : c - x.c1;
  sc = 1.f; sh = 0.f;
  if (scp) { sc = scp[cc]; sh = shp[cc]; }
}

// PF pixels per thread (npix <= PF*256), everything unrolled.  Instruction-lean by construction:
//  * the channel plane base (p + n*sn + c*sc) is wave-uniform -> SGPR pair; each lane adds ONE 32-bit
//    byte offset computed once per pixel slot -> `global_load_dword v, v_off, s[base]`, no 64-bit VALU;
//  * loads are UNCONDITIONAL on clamped (always valid) addresses and the padding zeros are selected
//    afterwards: a per-lane `if (inb) load` makes hipcc branch around every load and wait for it
//    (cdna_hip_programming.md, "Three .s-level traps" (c));
//  * the lazy-BatchNorm scale/shift are read in uniform control flow (scalar loads), one fma per value.
// Phase 1 issues all 32*PF loads of the chunk, phase 2 applies the affine, splits and writes LDS.
template <bool X3, int PF, int REC = IG_REC_BYTES>
__device__ __forceinline__ void stage_x_chunk_mlp(unsigned char* xhi, unsigned char* xlo,
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
        const char* chan = src_chan(x, b1, b2, c);
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
          *(uint4*)(xhi + (size_t)(tid + s * 256) * REC + g * 16) = make_uint4(0, 0, 0, 0);
          if (X3) *(uint4*)(xlo + (size_t)(tid + s * 256) * REC + g * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    if (g < ngroups) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cbase + g * 8 + j;
        const bool cok = c < cin;
        float sc, sh;
        src_affine(x, min(c, cin - 1), sc, sh);
#pragma unroll
        for (int s = 0; s < PF; ++s) {
          const float t = fmaf(v[s][g][j], sc, sh);
          v[s][g][j] = (inb[s] & cok) ? t : 0.f;
        }
      }
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (tid + s * 256 < npix) stage_write<X3, REC>(xhi, xlo, v[s][g], tid + s * 256, g);
    }
  }
}

template <bool X3, int MAXPF = 3, int REC = IG_REC_BYTES>
__device__ __forceinline__ void stage_x_chunk(unsigned char* xhi, unsigned char* xlo,
                                              const pcuda_src& x, int n, int cin, int chunk, int in_h, int in_w,
                                              int in_shift, int in_row, int oy0, int ox0, int th, int tw,
                                              int ngroups, int nwrite, int tid) {
  const int npix = th * tw;
  if (MAXPF == 1) {   // register-tight callers (wgrad: 80+ accumulator registers): 32 loads in flight per lane
    for (int pix0 = 0; pix0 < npix; pix0 += 256)
      stage_x_chunk_mlp<X3, 1, REC>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid + pix0);
    return;
  }
  if (npix <= 256) { stage_x_chunk_mlp<X3, 1, REC>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid); return; }
  if (npix <= 512) { stage_x_chunk_mlp<X3, 2, REC>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid); return; }
  if (npix <= 768) { stage_x_chunk_mlp<X3, 3, REC>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid); return; }
  for (int pix0 = 0; pix0 < npix; pix0 += 512)   // big tiles: 512 pixels at a time
    stage_x_chunk_mlp<X3, 2, REC>(xhi, xlo, x, n, cin, chunk, in_h, in_w, in_shift, in_row, oy0, ox0, th, tw, ngroups, nwrite, tid + pix0);
}

// ------------------------------------------------------------------------------------------
// Fast staging path (software-pipelined kernels).  Loads go through a buffer resource over the
// chunk's source image: the per-lane pixel offset (32 bit) sits in voffset, the wave-uniform channel
// offset in soffset, so a load costs one scalar add and no vector address arithmetic (the flat-address
// form spent ~10 vector instructions per load on 64-bit pointers).  The hardware range check does the
// zero padding: lanes outside the image (or past the tile) carry voffset = IG_OOB and read 0, channels
// past the source's last plane exceed num_records and read 0.
// Host guarantees (fast_src_ok): a 32-channel chunk never straddles the two sources and every
// source image spans < 2^30 bytes.
// ------------------------------------------------------------------------------------------
#define IG_OOB 0x40000000u

// A zero the optimiser cannot see through.  Added to the scale/shift pointers inside the per-tile commit so
// that their (loop-invariant, wave-uniform) loads are NOT hoisted out of the tile loop: hoisted, the 64 values
// of a chunk overflow the scalar registers and end up spilled to scratch and re-read from there every tile.
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}

template <int PF>
struct XFast {
  float v[PF][32];
  bool inb[PF];
  bool anyout;   // quad path, wave-uniform: some staged quad of this wave lies outside the image (needs the zero select)
};

// issue: npix halo pixels x (8*ngroups) channels of chunk `chunk`, image n; values stay in registers
template <int PF, int NT = 256>
__device__ __forceinline__ void xfast_issue(XFast<PF>& pre, const pcuda_src& x, int n, int cin, int chunk, int in_h,
                                            int in_w, int in_shift, int in_row, int oy0, int ox0, int tw, int npix,
                                            int ngroups, int tid) {
  const int c0 = chunk * 32;
  const bool first = c0 < x.c1;
  const float* base = first ? x.p1 + (long long)n * x.sn1 : x.p2 + (long long)n * x.sn2;
  const int csrc = first ? min(x.c1, cin) : cin - x.c1;   // channels held by this source
  const int cl0 = first ? c0 : c0 - x.c1;                 // chunk's first channel inside the source
  const unsigned plane = (unsigned)(first ? x.sc1 : x.sc2) * 4u;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(csrc * plane), 0x00020000);
  unsigned voff[PF];
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int pix = tid + s * NT;
    const int iy = pix / tw, ix = pix - iy * tw;
    const int gy = oy0 + iy, gx = ox0 + ix;
    pre.inb[s] = (pix < npix) & ((unsigned)gy < (unsigned)in_h) & ((unsigned)gx < (unsigned)in_w);
    voff[s] = pre.inb[s] ? (unsigned)((gy >> in_shift) * in_row + (gx >> in_shift)) * 4u : IG_OOB;
  }
  // always 32 loads per slot, no branches (channels past the source read 0 through the range check): a
  // conditional load would make every later counted wait on OLDER loads collapse to vmcnt(0)
  unsigned soff = (unsigned)cl0 * plane;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
#pragma unroll
    for (int s = 0; s < PF; ++s)
      pre.v[s][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[s], soff, 0));
    soff += plane;
  }
}

// commit: lazy-BatchNorm affine (if the source has one), bf16 hi/lo split, LDS write
template <bool X3, int PF, int NT = 256, int REC = IG_REC_BYTES>
__device__ __forceinline__ void xfast_commit(XFast<PF>& pre, unsigned char* xhi,
                                             unsigned char* xlo, const pcuda_src& x, int cin, int chunk,
                                             int npix, int ngroups, int nwrite, int tid) {
  const int c0 = chunk * 32;
  const bool first = c0 < x.c1;
  const float* scp = first ? x.scale1 : x.scale2;
  const float* shp = first ? x.shift1 : x.shift2;
  const int csrc = first ? min(x.c1, cin) : cin - x.c1;
  const int cl0 = (first ? c0 : c0 - x.c1) + opaque_zero();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g >= ngroups && g < nwrite) {   // channels past cin inside a 16-wide k-step: zeros, not LDS garbage
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (tid + s * NT < npix) {
          *(uint4*)(xhi + (size_t)(tid + s * NT) * REC + g * 16) = make_uint4(0, 0, 0, 0);
          if (X3) *(uint4*)(xlo + (size_t)(tid + s * NT) * REC + g * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    if (g < ngroups) {
      if (scp) {   // uniform
        float sc[8], sh[8];
        if (cl0 + g * 8 + 8 <= csrc) {   // whole group valid: consecutive scalar loads
#pragma unroll
          for (int j = 0; j < 8; ++j) { sc[j] = scp[cl0 + g * 8 + j]; sh[j] = shp[cl0 + g * 8 + j]; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int cc = cl0 + g * 8 + j;
            const bool cok = cc < csrc;
            const float a = scp[min(cc, csrc - 1)], b = shp[min(cc, csrc - 1)];
            sc[j] = cok ? a : 0.f; sh[j] = cok ? b : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int s = 0; s < PF; ++s) {
            const float t = fmaf(pre.v[s][g * 8 + j], sc[j], sh[j]);
            pre.v[s][g * 8 + j] = pre.inb[s] ? t : 0.f;   // zero padding is applied AFTER the affine
          }
      }
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (tid + s * NT < npix) {
          float (&v)[32] = pre.v[s];
          const float vv[8] = {v[g * 8 + 0], v[g * 8 + 1], v[g * 8 + 2], v[g * 8 + 3],
                               v[g * 8 + 4], v[g * 8 + 5], v[g * 8 + 6], v[g * 8 + 7]};
          stage_write<X3, REC>(xhi, xlo, vv, tid + s * NT, g);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Quad staging path (input rows of 4k pixels, no upsampling fold): one float4 load brings 4 consecutive
// pixels of one channel, so a 32-channel chunk of a 340-pixel halo tile is 16 loads per lane instead of
// 64.  (A wave can have at most 64 vector-memory instructions outstanding -- vmcnt is 6 bits -- so the
// dword path stalled in its own issue loop as soon as the weight loads shared the queue.)
// Wave w stages channel group w (8 channels: scale/shift stay scalar); lane + 64*s walks the
// (row, quad) grid of the tile.  Quads are aligned to the IMAGE (x = 4k), hence entirely inside or
// outside it; `lead` = pixels of the first quad left of the tile.
// ------------------------------------------------------------------------------------------
// (register image shared with the dword path: v[s][4*j + e] = pixel e of the quad, channel j of the group)
template <int PF, int NT = 256>
__device__ __forceinline__ void xq_issue(XFast<PF>& pre, const pcuda_src& x, int n, int cin, int chunk, int in_h, int in_w,
                                         int oy0, int ox0, int th, int tw, int tid) {
  // wave -> channel group (wv & 3); with 8 waves the two waves of a group split the (row, quad) items
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), w = wv & 3, lane = (tid & 63) + 64 * (wv >> 2);
  constexpr int IS = NT / 4;   // items per slot and group
  const int c0 = chunk * 32;
  const bool first = c0 < x.c1;
  const float* base = first ? x.p1 + (long long)n * x.sn1 : x.p2 + (long long)n * x.sn2;
  const int csrc = first ? min(x.c1, cin) : cin - x.c1;
  const int cl0 = (first ? c0 : c0 - x.c1) + w * 8;
  const long long sc = first ? x.sc1 : x.sc2;
  const int lead = ox0 & 3, nq = (tw + lead + 3) >> 2, nitems = th * nq;
  const int qmagic = (1 << 16) / nq + 1;   // exact for item < 1024, nq <= 80
  unsigned off[PF];
  bool outside = false;
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int item = lane + IS * s;
    const int iy = (item * qmagic) >> 16, q = item - iy * nq;
    const int gy = oy0 + iy, gx = ox0 - lead + 4 * q;
    pre.inb[s] = (item < nitems) & ((unsigned)gy < (unsigned)in_h) & ((unsigned)gx < (unsigned)in_w);
    off[s] = pre.inb[s] ? (unsigned)(gy * in_w + gx) * 4u : 0u;
    outside |= (item < nitems) & !pre.inb[s];
  }
  pre.anyout = __builtin_amdgcn_ballot_w64(outside) != 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const char* plane = (const char*)(base + (long long)min(cl0 + j, csrc - 1) * sc);   // wave-uniform, clamped
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const float4 t = *(const float4*)(plane + off[s]);
      pre.v[s][4 * j + 0] = t.x; pre.v[s][4 * j + 1] = t.y; pre.v[s][4 * j + 2] = t.z; pre.v[s][4 * j + 3] = t.w;
    }
  }
}

template <bool X3, int PF, int NT = 256, int REC = IG_REC_BYTES>
__device__ __forceinline__ void xq_commit(XFast<PF>& pre, unsigned char* xhi, unsigned char* xlo,
                                          const pcuda_src& x, int cin, int chunk, int ox0, int th, int tw, int nwrite,
                                          int tid) {
  // wave -> channel group (wv & 3); with 8 waves the two waves of a group split the (row, quad) items
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), w = wv & 3, lane = (tid & 63) + 64 * (wv >> 2);
  constexpr int IS = NT / 4;   // items per slot and group
  if (w >= nwrite) return;   // channel groups past the k-steps this chunk runs
  const int c0 = chunk * 32;
  const bool first = c0 < x.c1;
  const float* scp = first ? x.scale1 : x.scale2;
  const float* shp = first ? x.shift1 : x.shift2;
  const int csrc = first ? min(x.c1, cin) : cin - x.c1;
  const int cl0 = (first ? c0 : c0 - x.c1) + w * 8 + opaque_zero();
  const int lead = ox0 & 3, nq = (tw + lead + 3) >> 2, nitems = th * nq;
  const int qmagic = (1 << 16) / nq + 1;
  // Uniform blocks instead of per-element selects: the lazy-BatchNorm affine only where the source has one (the
  // gradient operands of dgrad do not), the zero select only on waves that staged a quad outside the image or a
  // ragged channel group (about a third of the tiles of a 256x256 map).
  if (scp) {
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scp[min(cl0 + j, csrc - 1)]; sh[j] = shp[min(cl0 + j, csrc - 1)]; }
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) pre.v[s][4 * j + e] = fmaf(pre.v[s][4 * j + e], sc[j], sh[j]);
  }
  if (pre.anyout | (cl0 + 8 > csrc)) {   // zero padding is applied AFTER the affine
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = pre.inb[s] & (cl0 + j < csrc);
#pragma unroll
        for (int e = 0; e < 4; ++e) pre.v[s][4 * j + e] = ok ? pre.v[s][4 * j + e] : 0.f;
      }
  }
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int item = lane + IS * s;
    const int iy = (item * qmagic) >> 16, q = item - iy * nq;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ix = 4 * q + e - lead;
      if ((item < nitems) & ((unsigned)ix < (unsigned)tw)) {
        const float vv[8] = {pre.v[s][e], pre.v[s][4 + e], pre.v[s][8 + e], pre.v[s][12 + e],
                             pre.v[s][16 + e], pre.v[s][20 + e], pre.v[s][24 + e], pre.v[s][28 + e]};
        stage_write<X3, REC>(xhi, xlo, vv, iy * tw + ix, w);
      }
    }
  }
}

// contiguous global -> LDS copy of nvec 16-B vectors (packed weights: the global image IS the LDS image).
// Branch-free: lanes past the end re-copy the last vector (same bytes to the same address), so the
// loads of a pass are issued back to back and no wait sits between them.  (With a guarded store the
// compiler sank every load into its store's branch and waited vmcnt(0) after each one.)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// (native vectors, not HIP's uint4 struct: an array of those inside a struct stayed in scratch memory, every
// load followed by a wait and a scratch store)
template <bool X3, int WV>
struct WPass {
  u32x4 hi[WV], lo[WV];
};
template <bool X3, int WV, int NT = 256>
__device__ __forceinline__ void wcopy_issue(WPass<X3, WV>& wp, const uint4* __restrict__ hi,
                                            const uint4* __restrict__ lo, int nvec, int base, int tid) {
#pragma unroll
  for (int u = 0; u < WV; ++u) {
    const int i = min(base + tid + u * NT, nvec - 1);
    wp.hi[u] = ((const u32x4*)hi)[i];
    if (X3) wp.lo[u] = ((const u32x4*)lo)[i];
  }
}
template <bool X3, int WV, int NT = 256>
__device__ __forceinline__ void wcopy_commit(const WPass<X3, WV>& wp, unsigned char* __restrict__ dhi,
                                             unsigned char* __restrict__ dlo, int nvec, int base, int tid) {
#pragma unroll
  for (int u = 0; u < WV; ++u) {
    const int i = min(base + tid + u * NT, nvec - 1);
    ((u32x4*)dhi)[i] = wp.hi[u];
    if (X3) ((u32x4*)dlo)[i] = wp.lo[u];
  }
}
template <bool X3, int WV, int NT = 256>
__device__ __forceinline__ void wcopy(unsigned char* __restrict__ dhi, unsigned char* __restrict__ dlo,
                                      const uint4* __restrict__ hi, const uint4* __restrict__ lo, int nvec, int base0,
                                      int tid) {
  for (int base = base0; base < nvec; base += NT * WV) {
    WPass<X3, WV> wp;
    wcopy_issue<X3, WV, NT>(wp, hi, lo, nvec, base, tid);
    __builtin_amdgcn_sched_barrier(0);   // keep the pass's loads together (the scheduler, short of registers,
    wcopy_commit<X3, WV, NT>(wp, dhi, dlo, nvec, base, tid);   // otherwise emits load / wait / store one vector at a time)
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) {
  return __builtin_bit_cast(bf16x8, *(const uint4*)p);
}

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
        const char* chan = src_chan(x, b1, b2, c);
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
        float sc, sh;
        src_affine(x, min(c, cin - 1), sc, sh);
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

