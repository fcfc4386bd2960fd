// Direct (vector-ALU) kernels for the discriminators' first layer: cin <= 8 class / entropy maps -> 64 channels,
// 4x4, stride 2, pad 2 (GAN.py:97).  On the MFMA kernels 4 or 5 of a 32-deep chunk's channels carry data: its data
// gradient ran at 17 TFLOP/s in four stride-parity launches, its weight gradient at 20 TFLOP/s (0.45 ms).  Here both
// are fp32 FMA loops; dispatched inside pcuda_conv2d_dgrad / pcuda_conv2d_wgrad like the kernels of conv_direct.hip
// (same ABI, same packed weights: w = hi + lo; PCUDA_NODIRECT=1 switches them off).  Fixed reduction order.
#include <stdlib.h>

#include "conv_host.h"
#include "conv_igemm.h"

namespace {

__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

// PCUDA_NODIRECT=1 switches every direct kernel off; PCUDA_DIRECT_MASK selects them one by one (A/B and bisection):
// 1 first-layer forward, 2 first-layer wgrad, 4 classifier forward, 8 classifier dgrad, 16 / 32 discriminator first
// layer dgrad / wgrad, 64 discriminator last layer forward
bool direct_enabled(int bit) {
  static int mask = -1;
  if (mask < 0) {
    const char* e = getenv("PCUDA_NODIRECT");
    const char* m = getenv("PCUDA_DIRECT_MASK");
    mask = (e && atoi(e)) ? 0 : (m ? atoi(m) : 0x7f);
  }
  return (mask & bit) != 0;
}
bool aligned16(const void* p, long long sn, long long sc) { return (((uintptr_t)p) & 15) == 0 && (sn & 3) == 0 && (sc & 3) == 0; }

struct D1Params {
  const float* dz; long long dz_sn, dz_sc;    // [n][cout][oh][ow]
  const float* x; long long x_sn, x_sc;       // wgrad: [n][cin][h][w]
  float* dx; long long dx_sn, dx_sc;          // dgrad: [n][cin][h][w]
  const uint16_t* wpack; int rec;             // dgrad: packed parity-class layouts (below)
  float* partial;                             // wgrad: [blocks][cout * cin * 16]
  int accumulate;
  int n, h, w, oh, ow, cin, cout;
  long long cls_stride;                       // dgrad: bf16 elements between the four parity-class images
  int one;                                    // dgrad: 1 -- the element stride of dz, kept out of the compiler's sight (below)
};

// data gradient.  A 2x2 block of input pixels (2Y + py, 2X + px) sees the same four gradient pixels (Y + a, X + b),
// a, b in {0, 1}, through the taps ky = py + 2(1 - a), kx = px + 2(1 - b).  A thread owns 4 consecutive X (two rows of
// 8 input pixels) for ALL input channels: per output channel 10 gradient values, CIN * 64 FMAs.
// Packed dgrad weights (conv_igemm.hip, pcuda_conv2d_pack_dgrad): one image per parity class (ry, rx), rows = cin (one
// 32-row tile), reduction = cout in 32-deep chunks, taps = the class's (ky, kx) with ky = ry, kx = rx (mod 2), ky
// ascending then kx ascending: element (co, ci, t) at (((co / 32) * 4 + t) * 32 + ci) * rec + co % 32.
template <int CIN>
__global__ __launch_bounds__(256) void d1_dgrad_kernel(const D1Params p) {
  __shared__ float sw[64 * CIN * 16];   // [co][ci][ky][kx]
  const int tid = threadIdx.x;
  for (int i = tid; i < p.cout * CIN * 16; i += 256) {
    const int kx = i & 3, ky = (i >> 2) & 3, ci = (i >> 4) % CIN, co = i / (16 * CIN);
    const int ry = ky & 1, rx = kx & 1, t = (ky >> 1) * 2 + (kx >> 1);
    const long long idx = (long long)(ry * 2 + rx) * p.cls_stride +
                          ((((long long)(co >> 5)) * 4 + t) * 32 + ci) * p.rec + (co & 31);
    const uint16_t* wq = p.wpack + idx;
    float v = bf16_bits_to_float(wq[0]);
    if (p.rec > IG_REC) v += bf16_bits_to_float(wq[32]);
    sw[i] = v;
    PCUDA_KEEP(wq);      // (VMEM address rule, common.h)
  }
  __syncthreads();
  const int n = blockIdx.y;
  const int xg = p.w >> 3;                              // groups of 4 X (8 input pixels) per row
  const int item = blockIdx.x * 256 + tid;
  if (item >= xg * (p.h >> 1)) return;
  const int Y = item / xg, X0 = 4 * (item - Y * xg);
  float acc[CIN][2][8];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[ci][r][e] = 0.f;
  const float* zp = p.dz + (long long)n * p.dz_sn + (long long)Y * p.ow + X0;
  // The five values of a row are read as five DWORD loads (index b * p.one, p.one == 1 at run time: the compiler cannot
  // merge them; rows of dz are only 4-byte aligned) and the row addresses of the loads in flight are kept alive until
  // their data has been used (VMEM address rule, common.h: the first build of this kernel -- one merged dwordx4 + a
  // dword whose destination was the wide load's address register -- returned wrong 16-lane groups in 54 % of its
  // launches on a GPU shared by two processes; scripts/micro/d1_repro.py).
  // (the next channel's ten gradient values are requested before this channel's FMAs: with the loads at the top of each
  // iteration the loop ran at one memory round trip per channel)
  float zn[2][5];
  const float* zq[2];      // rows in flight
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    zq[a] = zp + a * p.ow;
#pragma unroll
    for (int b = 0; b < 5; ++b) zn[a][b] = zq[a][b * p.one];
  }
  for (int co = 0; co < p.cout; ++co) {
    float z[2][5];
    const float* zc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      zc[a] = zq[a];
#pragma unroll
      for (int b = 0; b < 5; ++b) z[a][b] = zn[a][b];
    }
    const int con = min(co + 1, p.cout - 1);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      zq[a] = zp + (long long)con * p.dz_sc + a * p.ow;
#pragma unroll
      for (int b = 0; b < 5; ++b) zn[a][b] = zq[a][b * p.one];
    }
    const float* wc = sw + co * CIN * 16;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const float wv = wc[(ci * 4 + py + 2 * (1 - a)) * 4 + px + 2 * (1 - b)];
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[ci][py][2 * j + px] = fmaf(wv, z[a][j + b], acc[ci][py][2 * j + px]);
            }
    PCUDA_KEEP(zc[0]); PCUDA_KEEP(zc[1]);
  }
  PCUDA_KEEP(zq[0]); PCUDA_KEEP(zq[1]);
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float* d = p.dx + (long long)n * p.dx_sn + (long long)ci * p.dx_sc + (long long)(2 * Y + r) * p.w + 2 * X0;
      f32x4 o0 = {acc[ci][r][0], acc[ci][r][1], acc[ci][r][2], acc[ci][r][3]};
      f32x4 o1 = {acc[ci][r][4], acc[ci][r][5], acc[ci][r][6], acc[ci][r][7]};
      if (p.accumulate) {
        const f32x4 q0 = *(const f32x4*)d, q1 = *(const f32x4*)(d + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o0[e] += q0[e]; o1[e] += q1[e]; }
      }
      *(f32x4*)d = o0;
      *(f32x4*)(d + 4) = o1;
    }
}

// weight gradient.  256 threads own the cout * cin * 16 <= 64 * 4 * 16 sums: thread -> (4 output channels, one
// (ci, ky), the four kx).  A persistent workgroup walks (image, output row) items: the row of dz (all channels) and
// the four input rows it meets are staged in LDS (zero halo), then 4 output pixels per iteration = 64 FMAs against
// seven 16-byte LDS reads.  Block partials + fixed-order reduce.
#define D1_ZP 132    // dz row pitch in LDS (up to 129 + pad to 16 B)
#define D1_XP 276    // x row pitch: 2 zero columns left, up to 256 + 2 right, read 12 wide per 4 output pixels; 276 = 20 (mod 64):
                     // the 16 (channel, ky) rows a wave reads together start on different banks (272 put four on each)
template <int CIN>
__global__ __launch_bounds__(256) void d1_wgrad_kernel(const D1Params p, const int items) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* sz = smem_f;                       // [64][D1_ZP]
  float* sx = smem_f + 64 * D1_ZP;          // [CIN][4 rows][D1_XP]
  const int tid = threadIdx.x;
  const int cg = tid >> 4, jg = tid & 15;   // 4 output channels; (ci, ky)
  const int ci = jg >> 2, ky = jg & 3;
  const bool live = ci < CIN;
  float acc[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[c][k] = 0.f;
  const int ow4 = (p.ow + 3) >> 2;
  // Staging, coalesced and one item ahead: wave w loads rows co = w, w + 4, ... of dz (three 64-lane dword loads per
  // row) and the four input rows of channel w (one 16-byte load per lane and row); the loads of item i + 1 are in
  // flight during item i's FMAs.  (A plain strided copy loop issued one load per iteration and waited for it: the
  // kernel ran at a memory round trip per 256 elements.)
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float rz[16][3];
  f32x4 rx[4];
  const float* xq[4] = {p.x, p.x, p.x, p.x};      // addresses of the input rows in flight
  // VMEM address rule (common.h): these loads stay in flight through a whole item's FMAs.  Every address is a
  // wave-uniform row base (SGPRs) + one of four per-lane offsets that live for the whole kernel (kept after the loop),
  // so no load's address registers can be handed to a destination.
  const unsigned zoff[3] = {(unsigned)min(lane, p.ow - 1), (unsigned)min(lane + 64, p.ow - 1), (unsigned)min(lane + 128, p.ow - 1)};
  const unsigned xoff = (unsigned)min(4 * lane, p.w - 4);
  auto issue = [&](int it) {
    const int n = it / p.oh, oy = it - n * p.oh;
    const float* zb = p.dz + (long long)n * p.dz_sn + (long long)oy * p.ow;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float* zrow = zb + (long long)min(wv + 4 * k, p.cout - 1) * p.dz_sc;      // wave-uniform
#pragma unroll
      for (int u = 0; u < 3; ++u) rz[k][u] = zrow[zoff[u]];
    }
    const float* xb = p.x + (long long)n * p.x_sn + (long long)min(wv, CIN - 1) * p.x_sc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* xrow = xb + (long long)min(max(2 * oy + r - 2, 0), p.h - 1) * p.w;  // wave-uniform
      xq[r] = xrow + xoff;
      rx[r] = *(const f32x4*)xq[r];
    }
  };
  auto commit = [&](int it) {
    const int n = it / p.oh, oy = it - n * p.oh;
    (void)n;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int co = wv + 4 * k;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int c = lane + 64 * u;
        if (c < D1_ZP) sz[co * D1_ZP + c] = (c < p.ow && co < p.cout) ? rz[k][u] : 0.f;
      }
    }
    if (wv < CIN) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = 2 * oy + r - 2;
        const bool ok = (unsigned)yy < (unsigned)p.h && 4 * lane < p.w;
        float* d = sx + (wv * 4 + r) * D1_XP + 2 + 4 * lane;      // (column c of the row holds input column c - 2)
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = ok ? rx[r][e] : 0.f;
        PCUDA_KEEP(xq[r]);
        if (lane < 2) sx[(wv * 4 + r) * D1_XP + lane] = 0.f;                                  // left halo
        if (lane < D1_XP - 2 - 256) sx[(wv * 4 + r) * D1_XP + 2 + 256 + lane] = 0.f;          // right halo / pad
      }
    }
  };
  int it = blockIdx.x;
  if (it < items) issue(it);
  for (; it < items; it += gridDim.x) {
    __syncthreads();   // the previous item's FMAs are done with the tiles
    commit(it);
    __syncthreads();
    const int nit = it + gridDim.x;
    if (nit < items) issue(nit);
    if (live) {
      const float* xr = sx + (ci * 4 + ky) * D1_XP;
      for (int q = 0; q < ow4; ++q) {
        // output pixels ox = 4q .. 4q+3 meet input columns 2ox + kx - 2 (+2 halo) = 8q + 2j + kx, j < 4
        const f32x4 x0 = *(const f32x4*)(xr + 8 * q), x1 = *(const f32x4*)(xr + 8 * q + 4), x2 = *(const f32x4*)(xr + 8 * q + 8);
        const float xv[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 z = *(const f32x4*)(sz + (cg * 4 + c) * D1_ZP + 4 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) acc[c][kx] = fmaf(z[j], xv[2 * j + kx], acc[c][kx]);
        }
      }
    }
  }
  PCUDA_KEEP(zoff[0]); PCUDA_KEEP(zoff[1]); PCUDA_KEEP(zoff[2]); PCUDA_KEEP(xoff);
  if (live) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int co = cg * 4 + c;
        if (co < p.cout) p.partial[(long long)blockIdx.x * (p.cout * CIN * 16) + ((co * CIN + ci) * 4 + ky) * 4 + kx] = acc[c][kx];
      }
  }
}

bool d1_geom(const pcuda_conv_geom* g) {
  return g->k == 4 && g->stride == 2 && g->pad == 2 && g->dil == 1 && !g->in_up && g->cin <= 5 && g->cout <= 64 &&
         (g->cout & 3) == 0 && (g->in_w & 7) == 0 && (g->in_h & 1) == 0 && g->in_w <= 256 && g->out_w <= 129;
}
const int D1_WG_BLOCKS = 768;   // persistent grid of the weight gradient (3 workgroups per CU: ~50 KB of LDS each)

}  // namespace

int direct_d1_dgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad, const pcuda_dst* dx,
                    int accumulate, size_t cls_elems, hipStream_t s, int* rc) {
  *rc = PCUDA_OK;
  if (!direct_enabled(16) || !d1_geom(g) || dy->scale1 || dy->c1 < g->cout || dx->c1 < g->cin) return 0;
  if (!aligned16(dx->p1, dx->sn1, dx->sc1)) return 0;
  D1Params p;
  memset(&p, 0, sizeof(p));
  p.dz = dy->p1; p.dz_sn = dy->sn1; p.dz_sc = dy->sc1;
  p.dx = dx->p1; p.dx_sn = dx->sn1; p.dx_sc = dx->sc1;
  p.wpack = (const uint16_t*)packed_w_dgrad; p.rec = ig_rec_bytes(prec == PCUDA_PREC_BF16X3) / 2;
  p.cls_stride = (long long)cls_elems;
  p.accumulate = accumulate;
  p.one = 1;
  p.n = g->n; p.h = g->in_h; p.w = g->in_w; p.oh = g->out_h; p.ow = g->out_w; p.cin = g->cin; p.cout = g->cout;
  const double flops = 2.0 * g->n * (double)g->out_h * g->out_w * g->cout * (double)g->cin * 16;
  char tag[96];
  snprintf(tag, sizeof(tag), "direct d1 dgrad n%d cin%d cout%d %dx%d", g->n, g->cin, g->cout, g->in_h, g->in_w);
  ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
  const dim3 grid(cdiv((long long)(g->in_w / 8) * (g->in_h / 2), 256), g->n);
  switch (g->cin) {
    case 1: hipLaunchKernelGGL(d1_dgrad_kernel<1>, grid, dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL(d1_dgrad_kernel<2>, grid, dim3(256), 0, s, p); break;
    case 3: hipLaunchKernelGGL(d1_dgrad_kernel<3>, grid, dim3(256), 0, s, p); break;
    case 4: hipLaunchKernelGGL(d1_dgrad_kernel<4>, grid, dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL(d1_dgrad_kernel<5>, grid, dim3(256), 0, s, p); break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { pcuda_set_error("d1_dgrad_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; }
  return 1;
}

size_t direct_d1_wgrad_workspace(const pcuda_conv_geom* g) {
  if (!d1_geom(g) || g->cin > 4) return 0;
  return (size_t)D1_WG_BLOCKS * g->cout * g->cin * 16 * sizeof(float) + 256;
}

int direct_d1_wgrad(const pcuda_conv_geom* g, const pcuda_src* x, const float* dz, long long dz_sn, long long dz_sc, float* dw,
                    float* db, int accumulate, void* workspace, hipStream_t s, int* rc) {
  *rc = PCUDA_OK;
  if (!direct_enabled(32) || !d1_geom(g) || g->cin > 4 || x->scale1 || x->c1 < g->cin || db) return 0;
  if (!aligned16(x->p1, x->sn1, x->sc1)) return 0;
  D1Params p;
  memset(&p, 0, sizeof(p));
  p.dz = dz; p.dz_sn = dz_sn; p.dz_sc = dz_sc;
  p.x = x->p1; p.x_sn = x->sn1; p.x_sc = x->sc1;
  p.partial = (float*)workspace;
  p.n = g->n; p.h = g->in_h; p.w = g->in_w; p.oh = g->out_h; p.ow = g->out_w; p.cin = g->cin; p.cout = g->cout;
  const int items = g->n * g->out_h;
  const int blocks = items < D1_WG_BLOCKS ? items : D1_WG_BLOCKS;
  const size_t lds = (size_t)(64 * D1_ZP + g->cin * 4 * D1_XP) * sizeof(float);
  const double flops = 2.0 * g->n * (double)g->out_h * g->out_w * g->cout * (double)g->cin * 16;
  char tag[96];
  snprintf(tag, sizeof(tag), "direct d1 wgrad n%d cin%d cout%d %dx%d", g->n, g->cin, g->cout, g->in_h, g->in_w);
  ProfScope prof(PCUDA_FAM_CONV_WGRAD, flops, s, tag);
  const int numel = g->cout * g->cin * 16;
#define D1_WG_LAUNCH(C_)                                                                                              \
  {                                                                                                                   \
    /* (the attribute is per DEVICE: set on every launch rather than once per process -- the call is cheap) */          \
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)d1_wgrad_kernel<C_>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
    hipLaunchKernelGGL(d1_wgrad_kernel<C_>, dim3(blocks), dim3(256), lds, s, p, items);                               \
  }
  switch (g->cin) {
    case 1: D1_WG_LAUNCH(1) break;
    case 2: D1_WG_LAUNCH(2) break;
    case 3: D1_WG_LAUNCH(3) break;
    default: D1_WG_LAUNCH(4) break;
  }
#undef D1_WG_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { pcuda_set_error("d1_wgrad_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; return 1; }
  // (the block partials go through the split-K reduce of the MFMA weight-gradient kernels: 16 waves per 64 outputs; a
  // serial loop over the 768 blocks per output ran longer than the FMA kernel itself)
  *rc = launch_wgrad_reduce((const float*)workspace, numel, blocks, dw, accumulate, nullptr, 0, nullptr, s);
  return 1;
}

// ------------------------------------------------------------------------------------------
// The discriminators' LAST layer: cin (256 / 512) -> 1 channel, 4x4, stride 2, pad 2, on a 17x17 (or smaller) map
// (GAN.py:101).  One output row of a 32-row MFMA tile carried data: 0.4-0.7 TFLOP/s, 120 us per launch.  Round 2 ran
// one 1024-thread workgroup per image over 64-channel chunks staged in LDS (131 us: a fraction of the CUs, eight
// barrier-separated round trips); round 3 below.
// Packed forward weights: rows = cout = 1 (row 0 of a 32-row tile), element (ci, t) at ((ci / 32) * 16 + t) * 32 * rec + ci % 32.
// ------------------------------------------------------------------------------------------
namespace {

#define D5_MAXPIX 292     // largest input plane taken (17 x 17 = 289)
struct D5Params {
  const float* x; long long x_sn, x_sc;
  float* y; long long y_sn;
  const uint16_t* wpack; int rec;
  const float* bias;
  float slope;
  int h, w, oh, ow, cin;
};

// Round 3: one workgroup per (image, OUTPUT ROW) instead of one per image.  The one-per-image form put 32-64 workgroups
// on 256 CUs and walked the 512 channels in 8 barrier-separated chunks: 131 us for an 18.9 MB input (3 us of HBM time).
// Here 9 x n workgroups (288-576) each read the 4 input rows their output row sees, straight from global memory (an
// element is used by at most two output pixels of the row; the weights -- 32 KB in fp32 -- are staged in LDS once),
// thread = (output pixel, channel group), fixed-order combine over the channel groups: deterministic.
#define D5R_NT 512
__global__ __launch_bounds__(D5R_NT) void d5_fwd_kernel(const D5Params p) {
  extern __shared__ __attribute__((aligned(16))) float d5_smem[];
  float* swt = d5_smem;                      // [cin][16]
  float* spart = d5_smem + p.cin * 16;       // [ng][ow]
  const int tid = threadIdx.x, n = blockIdx.x, oy = blockIdx.y;
  for (int i = tid; i < p.cin * 16; i += D5R_NT) {
    const int ci = i >> 4, t = i & 15;
    const uint16_t* wq = p.wpack + ((((long long)(ci >> 5)) * 16 + t) * 32 * p.rec + (ci & 31));
    float wv = bf16_bits_to_float(wq[0]);
    if (p.rec > IG_REC) wv += bf16_bits_to_float(wq[32]);
    PCUDA_KEEP(wq);      // (VMEM address rule, common.h)
    swt[i] = wv;
  }
  const int ng = D5R_NT / p.ow;              // channel groups
  const int ox = tid % p.ow, gsel = tid / p.ow;
  const bool live = gsel < ng;
  int off[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int iy = 2 * oy + (t >> 2) - 2, ix = 2 * ox + (t & 3) - 2;
    off[t] = ((unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w) ? iy * p.w + ix : -1;
  }
  __syncthreads();
  float acc = 0.f;
  if (live) {
    const float* xb = p.x + (long long)n * p.x_sn;
    for (int c = gsel; c < p.cin; c += ng) {
      const float* xc = xb + (long long)c * p.x_sc;
      const float* wc = swt + c * 16;
      float xv[16];
      const float* xa[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) { xa[t] = xc + (off[t] >= 0 ? off[t] : 0); xv[t] = *xa[t]; }
#pragma unroll
      for (int t = 0; t < 16; ++t) acc = fmaf(wc[t], off[t] >= 0 ? xv[t] : 0.f, acc);
#pragma unroll
      for (int t = 0; t < 16; ++t) PCUDA_KEEP(xa[t]);      // (VMEM address rule, common.h: every address outlives its load)
    }
    spart[gsel * p.ow + ox] = acc;
  }
  __syncthreads();
  if (tid < p.ow) {
    float s = 0.f;
    for (int k = 0; k < ng; ++k) s += spart[k * p.ow + tid];   // fixed order
    // (VMEM address rule, common.h: SGPR base + a VGPR offset of its own that outlives the load)
    int boff = 0;
    asm volatile("" : "+v"(boff));
    s += p.bias ? p.bias[boff] : 0.f;
    PCUDA_KEEP(boff);
    s = s > 0.f ? s : s * p.slope;
    p.y[(long long)n * p.y_sn + oy * p.ow + tid] = s;
  }
}

bool d5_geom(const pcuda_conv_geom* g) {
  return g->cout == 1 && g->k == 4 && g->stride == 2 && g->pad == 2 && g->dil == 1 && !g->in_up &&
         g->in_h * g->in_w <= D5_MAXPIX && g->out_h * g->out_w <= 128;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Forward of the discriminators' first layer on the matrix cores WITHOUT the unfolded tensor (round 4).  The layer ran as
// unfold_taps_kernel (the 16 taps of the <= 5 input channels written out as 64 channels at 129x129: 136 MB per pass) + a
// 1x1 convolution over that tensor: 66 + 87 us for a layer that reads 33 MB and writes 136 MB.  Here the reduction index is
// (channel, tap): one MFMA k-step per input channel, its 16 values the 4x4 taps.  A lane (output pixel r, k-half h) needs
// the taps ky in {2h, 2h + 1}, kx = 0 .. 3 of its pixel: two runs of four consecutive input floats -- read straight from
// global memory (neighbouring lanes' runs overlap by half: coalesced), split, fed to the MFMA.  The 64 x 16 CIN weights sit
// in registers as fragments for the whole kernel.  No LDS tile, no unfolded tensor; bias + LeakyReLU in the epilogue.
// ------------------------------------------------------------------------------------------
namespace {

struct D1FParams {
  const float* x; long long x_sn, x_sc;
  float* y; long long y_sn, y_sc;
  const uint16_t* wpack; int rec;      // forward layout: [16 taps][64 rows][rec], lo plane at + 32 (bf16x3)
  const float* bias; float slope;
  int n, h, w, oh, ow;
  int tiles_per_image, tiles;
};

template <bool X3, int CIN>
__global__ __launch_bounds__(256) void d1_fwd_kernel(const D1FParams p) {
  __shared__ uint4 wfrag[2][CIN * 2 * 64];      // [plane][(k-step, row block) x lane]: the A fragments, built once per workgroup
  __shared__ float sbias[64];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    uint16_t* const wf = (uint16_t*)&wfrag[0][0];
    for (int e = tid; e < CIN * 2 * 64 * 8; e += 256) {
      const int j = e & 7, ln = (e >> 3) & 63, cb = (e >> 9) & 1, ks = e >> 10;
      const int tap = 8 * (ln >> 5) + j, co = cb * 32 + (ln & 31);
      const uint16_t* wq = p.wpack + ((long long)tap * 64 + co) * p.rec + ks;
      wf[e] = wq[0];
      if (X3) wf[CIN * 2 * 64 * 8 + e] = wq[32];
      PCUDA_KEEP(wq);      // (VMEM address rule, common.h)
    }
  }
  if (tid < 64) {
    const float* bq = p.bias ? p.bias + tid : nullptr;
    sbias[tid] = bq ? *bq : 0.f;
    PCUDA_KEEP(bq);
  }
  __syncthreads();
  bf16x8 ah[CIN][2], al[CIN][2];
#pragma unroll
  for (int ks = 0; ks < CIN; ++ks)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      ah[ks][cb] = __builtin_bit_cast(bf16x8, wfrag[0][(ks * 2 + cb) * 64 + lane]);
      al[ks][cb] = ah[ks][cb];
      if (X3) al[ks][cb] = __builtin_bit_cast(bf16x8, wfrag[1][(ks * 2 + cb) * 64 + lane]);
    }

  const int opix = p.oh * p.ow;
  for (int tile = blockIdx.x * 4 + w; tile < p.tiles; tile += gridDim.x * 4) {
    const int img = tile / p.tiles_per_image, pix = (tile - img * p.tiles_per_image) * 32 + r;
    const bool pv = pix < opix;
    const int pc = min(pix, opix - 1);
    const int oy = pc / p.ow, ox = pc - oy * p.ow;
    // this lane's two input rows (ky = 2 h, 2 h + 1) and its run of four columns (kx = 0 .. 3), two float2 pieces
    const int ix0 = 2 * ox - 2;
    const bool cv0 = (unsigned)ix0 < (unsigned)p.w, cv1 = (unsigned)(ix0 + 2) < (unsigned)p.w;      // (w even: a piece is in or out whole)
    const int cx0 = cv0 ? ix0 : 0, cx1 = cv1 ? ix0 + 2 : 0;
    int rowo[2];
    bool rv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int iy = 2 * oy - 2 + 2 * h + q;
      rv[q] = (unsigned)iy < (unsigned)p.h;
      rowo[q] = (rv[q] ? iy : 0) * p.w;
    }
    const float* xp = p.x + (long long)img * p.x_sn;
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
    float2 v[CIN][2][2];
    const float* pa[CIN][2][2];      // (kept alive behind the loads: VMEM address rule, common.h)
#pragma unroll
    for (int ks = 0; ks < CIN; ++ks)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float* rp = xp + (long long)ks * p.x_sc + rowo[q];
        pa[ks][q][0] = rp + cx0;
        pa[ks][q][1] = rp + cx1;
        v[ks][q][0] = *(const float2*)pa[ks][q][0];
        v[ks][q][1] = *(const float2*)pa[ks][q][1];
      }
#pragma unroll
    for (int ks = 0; ks < CIN; ++ks) {
      float t[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        t[4 * q + 0] = (rv[q] & cv0) ? v[ks][q][0].x : 0.f; t[4 * q + 1] = (rv[q] & cv0) ? v[ks][q][0].y : 0.f;
        t[4 * q + 2] = (rv[q] & cv1) ? v[ks][q][1].x : 0.f; t[4 * q + 3] = (rv[q] & cv1) ? v[ks][q][1].y : 0.f;
      }
      uint4 hi, lo = make_uint4(0, 0, 0, 0);
      if (X3) {
        split2(t[0], t[1], hi.x, lo.x); split2(t[2], t[3], hi.y, lo.y);
        split2(t[4], t[5], hi.z, lo.z); split2(t[6], t[7], hi.w, lo.w);
      } else {
        hi.x = pack_bf16x2(t[0], t[1]); hi.y = pack_bf16x2(t[2], t[3]);
        hi.z = pack_bf16x2(t[4], t[5]); hi.w = pack_bf16x2(t[6], t[7]);
      }
      const bf16x8 bh = __builtin_bit_cast(bf16x8, hi), bl = __builtin_bit_cast(bf16x8, lo);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        if (X3) {
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][cb], bh, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][cb], bl, acc[cb], 0, 0, 0);
        }
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][cb], bh, acc[cb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < CIN; ++ks)
#pragma unroll
      for (int q = 0; q < 2; ++q) { PCUDA_KEEP(pa[ks][q][0]); PCUDA_KEEP(pa[ks][q][1]); }
    // epilogue: register i of row block cb = output channel cb 32 + (i & 3) + 8 (i >> 2) + 4 h, the lane = the pixel:
    // 32 lanes store 128 contiguous bytes of one channel plane
    float* yp = p.y + (long long)img * p.y_sn + pc;
    if (pv) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          float o = acc[cb][i] + sbias[co];
          o = o > 0.f ? o : o * p.slope;
          yp[(long long)co * p.y_sc] = o;
        }
    }
  }
}

}  // namespace

static bool d1_fwd_on(const pcuda_conv_geom* g) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("PCUDA_NO_D1FWD"); off = (e && atoi(e)) ? 1 : 0; }
  return !off && direct_enabled(16) && d1_geom(g) && g->cout == 64;
}

// 1: pcuda_conv2d_forward runs this geometry on d1_fwd_kernel (given plain NCHW fp32 tensors, 8-byte aligned planes): the
// caller then skips the unfold + 1x1 form of the layer (GAN.py's first layer)
extern "C" int pcuda_conv2d_d1_forward_ok(const pcuda_conv_geom* g) { return (g && d1_fwd_on(g)) ? 1 : 0; }

// returns 1 when it took the launch (*rc = status)
int direct_d1_forward(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const void* packed_w, const float* bias,
                      float slope, const pcuda_dst* y, float* bn_partials, hipStream_t s, int* rc) {
  *rc = PCUDA_OK;
  if (!d1_fwd_on(g) || bn_partials || x->scale1 ||
      x->c1 < g->cin || y->c1 < g->cout)
    return 0;
  if ((((uintptr_t)x->p1) & 7) || (x->sn1 & 1) || (x->sc1 & 1)) return 0;      // (float2 pieces at even columns)
  const bool x3 = prec == PCUDA_PREC_BF16X3;
  D1FParams p;
  memset(&p, 0, sizeof(p));
  p.x = x->p1; p.x_sn = x->sn1; p.x_sc = x->sc1;
  p.y = y->p1; p.y_sn = y->sn1; p.y_sc = y->sc1;
  p.wpack = (const uint16_t*)packed_w; p.rec = ig_rec_bytes(x3) / 2;
  p.bias = bias; p.slope = slope;
  p.n = g->n; p.h = g->in_h; p.w = g->in_w; p.oh = g->out_h; p.ow = g->out_w;
  p.tiles_per_image = (g->out_h * g->out_w + 31) / 32;
  p.tiles = p.tiles_per_image * g->n;
  const double flops = 2.0 * g->n * (double)g->out_h * g->out_w * g->cout * (double)g->cin * 16;
  char tag[96];
  snprintf(tag, sizeof(tag), "direct d1 fwd (mfma) n%d cin%d cout%d %dx%d", g->n, g->cin, g->cout, g->in_h, g->in_w);
  ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
  int grid = (p.tiles + 3) / 4;
  if (grid > 2048) grid = 2048;
#define D1F(X3_, C_) hipLaunchKernelGGL((d1_fwd_kernel<X3_, C_>), dim3(grid), dim3(256), 0, s, p)
#define D1F_C(X3_)                                                                                           \
  do {                                                                                                        \
    switch (g->cin) {                                                                                         \
      case 1: D1F(X3_, 1); break; case 2: D1F(X3_, 2); break; case 3: D1F(X3_, 3); break;                    \
      case 4: D1F(X3_, 4); break; default: D1F(X3_, 5); break;                                                \
    }                                                                                                         \
  } while (0)
  if (x3) D1F_C(true); else D1F_C(false);
#undef D1F_C
#undef D1F
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { pcuda_set_error("d1_fwd_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; }
  return 1;
}

int direct_d5_forward(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const void* packed_w, const float* bias,
                      float slope, const pcuda_dst* y, float* bn_partials, hipStream_t s, int* rc) {
  *rc = PCUDA_OK;
  if (!direct_enabled(64) || !d5_geom(g) || bn_partials || x->scale1 || x->c1 < g->cin) return 0;
  D5Params p;
  p.x = x->p1; p.x_sn = x->sn1; p.x_sc = x->sc1;
  p.y = y->p1; p.y_sn = y->sn1;
  p.wpack = (const uint16_t*)packed_w; p.rec = ig_rec_bytes(prec == PCUDA_PREC_BF16X3) / 2;
  p.bias = bias; p.slope = slope;
  p.h = g->in_h; p.w = g->in_w; p.oh = g->out_h; p.ow = g->out_w; p.cin = g->cin;
  const double flops = 2.0 * g->n * (double)g->out_h * g->out_w * (double)g->cin * 16;
  char tag[96];
  snprintf(tag, sizeof(tag), "direct d5 fwd n%d cin%d %dx%d", g->n, g->cin, g->in_h, g->in_w);
  ProfScope prof(PCUDA_FAM_CONV_FWD, flops, s, tag);
  const size_t lds = (size_t)(g->cin * 16 + D5R_NT) * sizeof(float);
  if (lds > 60 * 1024) return 0;      // (cin <= 896; wider layers take the MFMA kernel)
  hipLaunchKernelGGL(d5_fwd_kernel, dim3(g->n, g->out_h), dim3(D5R_NT), lds, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { pcuda_set_error("d5_fwd_kernel: %s", hipGetErrorString(e)); *rc = PCUDA_E_LAUNCH; }
  return 1;
}
