/*
 * libpcuda_hip.so -- C ABI of the MI355X (gfx950) kernels behind the PointCloudUDA
 * adversarial train-step hot path.
 *
 * The reference (sulaimanvesal/PointCloudUDA) has no FFI: its "operators" are the ATen
 * ops issued by the modules under src/networks, src/utils/loss.py and the body of train_epoch
 * (src/train_mscmrseg.py:183-330).  Each entry point below names the reference call
 * site(s) it replaces.  The Python host (pointcloududa_amd/) binds these with ctypes and
 * keeps the reference's nn.Module / function signatures on top.
 *
 * Conventions
 *   - all tensors are device pointers owned by the caller; fp32 unless stated otherwise;
 *     image tensors are NCHW with dense HxW planes and explicit batch/channel strides
 *     (in ELEMENTS), so channel slices of concatenated buffers can be passed without copies;
 *   - every launch goes to the caller's hipStream_t (void* here); nothing allocates, frees
 *     or synchronises; scratch memory is passed in (query the size with *_workspace_size);
 *   - return value: 0 = ok, <0 = error (PCUDA_E_*); never throws;
 *   - precision modes for the MFMA convolutions (operands are split on the fly from fp32):
 *       PCUDA_PREC_BF16X3  a = a_hi + a_lo (two bf16), a*b ~= ah*bh + ah*bl + al*bh, fp32 accumulate
 *                          (~2^-17 relative per product: the parity mode, default)
 *       PCUDA_PREC_BF16    single bf16 MFMA per product, fp32 accumulate (throughput mode)
 */
#ifndef PCUDA_HIP_H
#define PCUDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCUDA_OK 0
#define PCUDA_E_BADARG (-1)
#define PCUDA_E_UNSUPPORTED (-2)
#define PCUDA_E_LAUNCH (-3)
#define PCUDA_E_WORKSPACE (-4)

#define PCUDA_PREC_BF16X3 0
#define PCUDA_PREC_BF16 1

typedef void* pcuda_stream_t; /* hipStream_t */

/* library / device ----------------------------------------------------------------- */
/* ABI version of this header.  Bumped whenever an entry point's signature, a struct layout or a workspace contract
 * changes (5, round 5: pcuda_src / pcuda_dst lost the record fields, the record-convolution entries left the library,
 * pcuda_nn_loss_workspace_floats / pcuda_abi_struct_size / pcuda_last_kernel were added).  A binding must refuse a library
 * whose pcuda_version() differs from the PCUDA_ABI_VERSION it was written against, and compare its own struct sizes
 * with pcuda_abi_struct_size: the library reads these structs from caller memory. */
#define PCUDA_ABI_VERSION 5
int pcuda_version(void);
/* sizeof() of an ABI struct as THIS library was compiled: 0 pcuda_conv_geom, 1 pcuda_src, 2 pcuda_dst, 3 pcuda_pooled,
 * 4 pcuda_reduce_job; 0 for an unknown index */
size_t pcuda_abi_struct_size(int which);
int pcuda_device_count(void);            /* hipGetDeviceCount, 0 when no GPU */
const char* pcuda_last_error(void);      /* text of the last failing call on this thread */
const char* pcuda_build_hash(void);      /* sha256[:16] of the kernel sources the loaded library was compiled from */
long long pcuda_launch_count(int reset); /* kernel launches issued by this library since the last reset (host-side count) */
/* convolution launches that ran on a generic fallback kernel because the geometry's specialised template instantiation is
 * not in this build (csrc/variants.h: the default build holds the variants of the benchmark configurations and tests) */
long long pcuda_fallback_count(void);
/* "<shape tag> | <kernel>" of the most recent convolution launch issued by the calling thread (the tag a profile row
 * carries, and which kernel family the dispatcher picked for it: igemm_pipe / igemm8 / igemm_generic / wgrad[+fallback] /
 * wgrad3 / wgrad3r / wgrad1 / direct ...): lets a kernel-level test assert WHICH kernel produced the result it checks */
const char* pcuda_last_kernel(void);
/* measurement aid (bench.py `clock_ghz_under_load`): 512 workgroups x 4 waves issue `iters` x 4 back-to-back 32x32x16 bf16
 * MFMAs; out2_dev[0] / out2_dev[1] receive the summed shader-clock (s_memtime) and 100-MHz reference-clock (s_memrealtime)
 * ticks of every workgroup's first wave: shader GHz = 0.1 * out2[0] / out2[1].  sink_dev: one float (never written). */
int pcuda_clock_probe(unsigned long long* out2_dev, float* sink_dev, int iters, pcuda_stream_t s);

/* kernel-family timing (HIP events recorded on the launch stream around every launch of
 * a family while enabled; used by bench.py for the live roofline figure) */
#define PCUDA_FAM_CONV_FWD 0   /* implicit-GEMM forward + dgrad launches */
#define PCUDA_FAM_CONV_WGRAD 1
#define PCUDA_FAM_POINTWISE 2
#define PCUDA_FAM_DENSE_F32 3 /* exact-fp32 MFMA k=1 Conv1d of PointNetCls (forward, dgrad, wgrad): fp32 matrix peak, not bf16 */
#define PCUDA_FAM_COUNT 4
int pcuda_prof_enable(int on);
int pcuda_prof_reset(void);
/* synchronises the recorded events; ms = summed kernel time, work = summed algorithmic
 * FLOPs (conv families) or bytes (pointwise), launches = number of launches */
int pcuda_prof_read(int family, double* ms, double* work, long long* launches);
/* writes one CSV row per recorded launch (family, work, ms, shape tag) */
int pcuda_prof_dump(const char* path);
/* timing experiments only (PCUDA_DBG bit 128): the eight per-phase cycle sums the pipelined forward/dgrad
 * kernel accumulates (barrier, X commit, issue, weight copy, barrier, MFMA, epilogue, loop tail); read + reset */
int pcuda_debug_read_clocks(unsigned long long* out8);

/* ------------------------------------------------------------------------------------
 * 2-D convolution: replaces nn.Conv2d forward/backward at
 *   unet.py:23,27,32 (encoder 3x3 / 1x1), :61 (dilated bottleneck), :85 (6x6 head),
 *   :112,116,122 (decoder), :178 (classifier); GAN.py:97-107 (4x4 s2 p2, 3x3 s2 p1).
 * cross-correlation, square kernel, symmetric padding, one dilation, groups = 1.
 * ---------------------------------------------------------------------------------- */
typedef struct pcuda_conv_geom {
  int n;            /* batch */
  int cin, cout;
  int in_h, in_w;   /* LOGICAL input plane (after the optional x2 upsample) */
  int out_h, out_w; /* = floor((in + 2*pad - dil*(k-1) - 1)/stride) + 1 */
  int k, stride, pad, dil;
  int in_up;        /* 1: the stored input is (in_h/2 x in_w/2) and is read through a nearest x2
                       upsample (folds nn.UpsamplingNearest2d, unet.py:111, into the conv) */
} pcuda_conv_geom;

/* source / destination that may be split across two tensors along channels
 * (zero-copy torch.cat(dim=1): unet.py:46,134).  c1 = channels taken from p1; the rest come
 * from p2 (p2 may be NULL when c1 == total).  scale/shift: optional per-channel affine
 * applied on load to in-bounds elements (fused BatchNorm apply), NULL = identity. */
typedef struct pcuda_src {
  const float* p1; long long sn1, sc1; const float* scale1; const float* shift1;
  const float* p2; long long sn2, sc2; const float* scale2; const float* shift2;
  int c1;
} pcuda_src;
typedef struct pcuda_dst {
  float* p1; long long sn1, sc1;
  float* p2; long long sn2, sc2;
  int c1;
} pcuda_dst;

/* packed-weight sizes (bytes) for a given geometry / precision */
size_t pcuda_conv2d_packed_fwd_bytes(const pcuda_conv_geom* g, int prec);
size_t pcuda_conv2d_packed_dgrad_bytes(const pcuda_conv_geom* g, int prec);
/* repack fp32 OIHW weights into the MFMA-fragment layout [co-tile][ci-chunk][tap][row][record]; record =
 * 32 hi | 32 lo | 8 pad bf16 (144 bytes) in PCUDA_PREC_BF16X3, 32 values + 8 pad (80 bytes) in PCUDA_PREC_BF16.
 * Opaque to the caller (sizes from the two queries above); call once per optimiser step */
int pcuda_conv2d_pack_fwd(const pcuda_conv_geom* g, int prec, const float* w, void* packed, pcuda_stream_t s);
int pcuda_conv2d_pack_dgrad(const pcuda_conv_geom* g, int prec, const float* w, void* packed, pcuda_stream_t s);
/* both repacks in ONE launch (forward layout + every dgrad parity class -- for k = 4 / stride 2 / pad 2 one image per ROW
 * parity with the two column classes interleaved row by row; the layout is the library's business: size it with
 * pcuda_conv2d_packed_dgrad_bytes and hand it back to pcuda_conv2d_dgrad); packed_dgrad may be NULL */
int pcuda_conv2d_pack_all(const pcuda_conv_geom* g, int prec, const float* w, void* packed_fwd, void* packed_dgrad,
                          pcuda_stream_t s);
/* Batched repack (one launch for all layers of a network after an optimiser step): fill the job records of a layer
 * (forward layout + every dgrad parity class; packed_dgrad may be NULL) into host memory -- returns the number of jobs
 * written (<= 1 + stride^2) or <0; job_blocks[j] = workgroups job j needs.  Keep the concatenated records in DEVICE
 * memory next to an int table first_block[j] (exclusive prefix sum of job_blocks over all jobs) and replay them with
 * pcuda_conv2d_pack_table (total_blocks = the sum).  The records hold the raw pointers given here. */
size_t pcuda_conv2d_pack_job_bytes(void);
int pcuda_conv2d_pack_jobs_fill(const pcuda_conv_geom* g, int prec, const float* w, void* packed_fwd, void* packed_dgrad,
                                void* host_jobs, int max_jobs, int* job_blocks);
int pcuda_conv2d_pack_table(const void* dev_jobs, const int* dev_first_block, int njobs, int total_blocks, pcuda_stream_t s);

/* 1: pcuda_conv2d_forward takes this geometry (the discriminators' first layer, GAN.py:97: k = 4, stride 2, pad 2, <= 5
 * input channels, 64 outputs) on the direct MFMA kernel that reads the 4x4 taps straight from the image -- the caller then
 * skips the unfold (pcuda_unfold_taps) + 1x1 form of the layer */
int pcuda_conv2d_d1_forward_ok(const pcuda_conv_geom* g);
/* y = lrelu(conv(x) + bias, slope)   (slope = 1 -> no activation; bias may be NULL)
 * bn_partials (optional): per-tile partial sums [ntiles][cout][2] (sum, sum of squares) of the
 * activated output, for the BatchNorm that follows (unet.py:26,30); ntiles from
 * pcuda_conv2d_fwd_tiles().  Deterministic (no atomics). */
int pcuda_conv2d_fwd_tiles(const pcuda_conv_geom* g, int prec);
int pcuda_conv2d_forward(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const void* packed_w,
                         const float* bias, float slope, const pcuda_dst* y, float* bn_partials,
                         pcuda_stream_t s);
/* dx (+)= conv_transpose(dy): the data gradient; for stride 2 this is the transposed
 * convolution the adversarial gradient takes back through the discriminators.
 * dy: [n][cout][out_h][out_w] given as a pcuda_src (c1 = cout); dx may be split like a cat.
 * When g->in_up is set, dx is the gradient of the UPSAMPLED (logical) input. */
int pcuda_conv2d_dgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                       const pcuda_dst* dx, int accumulate, pcuda_stream_t s);
/* The same data gradient with the BatchNorm-backward reduce of the layer in front of this convolution fused into its
 * epilogue (unet.py:23-30: conv -> LeakyReLU -> BN -> conv; the second convolution's dx IS the first BatchNorm's incoming
 * gradient): red_partials[tile][cin][2] = (sum g, sum g * (a - mean) * invstd) over the tile's stored gradient g, the
 * input of pcuda_bn_bwd_finalize with ntiles = pcuda_conv2d_dgrad_tiles().  Stride-1 layers whose rows are multiples of
 * 4 pixels; returns PCUDA_E_UNSUPPORTED otherwise (run pcuda_conv2d_dgrad + pcuda_bn_bwd_reduce instead). */
int pcuda_conv2d_dgrad_tiles(const pcuda_conv_geom* g, int prec);
int pcuda_conv2d_dgrad_bnred(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                             const pcuda_dst* dx, int accumulate, const float* a, long long a_sn, long long a_sc,
                             const float* mean, const float* invstd, float* red_partials, pcuda_stream_t s);
/* dx = pcuda_conv2d_dgrad(dy) * (a > 0 ? 1 : slope): the LeakyReLU backward of the layer in FRONT of this convolution
 * (GAN.py:97-108, conv -> LeakyReLU(0.2) -> conv going back) in the data-gradient kernel's epilogue; the gradient with respect
 * to the activation is never stored.  a: the saved activation [n][cin][in_h][in_w], plane stride a_sc == dx->sc1, one
 * destination.  PCUDA_E_UNSUPPORTED where a launch of the layer takes the transposed (16-byte store) epilogue: the caller runs
 * pcuda_conv2d_dgrad + pcuda_lrelu_bwd instead (a partly written dx is overwritten by them). */
int pcuda_conv2d_dgrad_lrelu(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                             const pcuda_dst* dx, const float* a, long long a_sn, long long a_sc, float slope,
                             pcuda_stream_t s);
/* The data gradient of a layer behind the nearest-x2 fold (g->in_up: the decoder's up-convolutions, unet.py:111-112) written at
 * the STORED, half resolution: dx_half[n][cin][in_h/2][in_w/2] = the 2x2 block sums of the logical gradient (what
 * pcuda_upsample2_bwd computes from pcuda_conv2d_dgrad's output, without that 4x larger tensor going through HBM).  a != NULL:
 * the BatchNorm-backward reduce of the layer that produced the stored tensor rides along as in pcuda_conv2d_dgrad_bnred
 * (red_partials[pcuda_conv2d_dgrad_tiles()][cin][2]).  PCUDA_E_UNSUPPORTED for plans off the 32 x 8-tile transposed epilogue. */
int pcuda_conv2d_dgrad_fold(const pcuda_conv_geom* g, int prec, const pcuda_src* dy, const void* packed_w_dgrad,
                            const pcuda_dst* dx_half, const float* a, long long a_sn, long long a_sc, const float* mean,
                            const float* invstd, float* red_partials, pcuda_stream_t s);
/* dw (+)= sum over batch and space of dy (x) x ; db (+)= sum dy (db may be NULL).
 * workspace: pcuda_conv2d_wgrad_workspace_size() bytes. */
size_t pcuda_conv2d_wgrad_workspace_size(const pcuda_conv_geom* g);
int pcuda_conv2d_wgrad(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy,
                       long long dy_sn, long long dy_sc, float* dw, float* db, int accumulate,
                       void* workspace, size_t workspace_bytes, pcuda_stream_t s);

/* The same weight gradient with its split-K reduce DEFERRED: the partial sums stay in `workspace` (which must live
 * until the reduce has run) and *job describes the reduce; pcuda_wgrad_reduce_batch runs the reduces of many layers in
 * one launch (host array of jobs; same per-output arithmetic and order as pcuda_conv2d_wgrad, bit-identical results).
 * A backward pass issues one weight gradient per layer; no two jobs of one batch may name the same dw.
 * job->ksplit == 0 on return: the layer's kernel wrote dw itself, nothing to reduce. */
#define PCUDA_REDUCE_MAX_JOBS 56
typedef struct {
  const float* partial;      /* [ksplit][numel] */
  long long numel;
  float* dw;
  const float* db_partial;   /* [ksplit][nb] or NULL */
  long long nb;
  float* db;
  int ksplit, nkg, accumulate, ntaps;
} pcuda_reduce_job;
int pcuda_conv2d_wgrad_partial(const pcuda_conv_geom* g, int prec, const pcuda_src* x, const float* dy,
                               long long dy_sn, long long dy_sc, float* dw, float* db, int accumulate,
                               void* workspace, size_t workspace_bytes, pcuda_reduce_job* job, pcuda_stream_t s);
int pcuda_wgrad_reduce_batch(const pcuda_reduce_job* host_jobs, int njobs, pcuda_stream_t s);

/* ------------------------------------------------------------------------------------
 * BatchNorm (training mode) around LeakyReLU: conv -> LeakyReLU -> BatchNorm2d
 * (unet.py:23-30,116-125); also BatchNorm1d of PointNetCls.py (hw = points or 1).
 * ---------------------------------------------------------------------------------- */
/* reduce per-tile partials -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale;
 * running stats updated with momentum (unbiased variance), torch semantics. count = n*h*w */
int pcuda_bn_finalize(const float* partials, int ntiles, int c, long long count, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                      float* mean, float* invstd, float* scale, float* shift, pcuda_stream_t s);
/* partial sums of a tensor a[n][c][hw] (for layers whose producer is not a conv epilogue);
 * returns ntiles written through *ntiles (query with partials == NULL) */
int pcuda_bn_stats(const float* a, long long sn, long long sc, int n, int c, long long hw,
                   float* partials, int* ntiles, pcuda_stream_t s);
/* y = a*scale[c] + shift[c]; optional relu (PointNetCls.py:41-43) */
int pcuda_bn_apply(const float* a, long long a_sn, long long a_sc, const float* scale, const float* shift,
                   int relu, float* y, long long y_sn, long long y_sc, int n, int c, long long hw,
                   pcuda_stream_t s);
/* backward of [z -> a = lrelu(z, slope) -> y = BN(a)] given dy (optionally dy + dy2):
 * pass 1: per-tile partials of (sum dy, sum dy*xhat) -> red[ntiles][c][2]
 * pass 2 (after pcuda_bn_bwd_finalize): dz = lrelu'(a) * scale*(dy - mean_dy - xhat*mean_dyxhat)
 * post_relu: the block is BN -> ReLU instead (PointNetCls): dy is first masked by (y > 0) and
 * `act_slope` applies to nothing (a is then the BN input). */
int pcuda_bn_bwd_reduce(const float* dy, long long dy_sn, long long dy_sc, const float* dy2, long long dy2_sn,
                        long long dy2_sc, const float* a, long long a_sn, long long a_sc, const float* mean,
                        const float* invstd, const float* scale, const float* shift, int post_relu, int n, int c,
                        long long hw, float* red, int* ntiles, pcuda_stream_t s);
/* count < 0: the statistics were FROZEN (eval-mode BatchNorm, nn.BatchNorm2d.eval() in a fine-tuning run: mean / invstd
 * from the running buffers): coef then describes the fixed affine (dz = m * c0 * dy), dgamma / dbeta are unchanged */
int pcuda_bn_bwd_finalize(const float* red, int ntiles, int c, long long count, const float* gamma,
                          const float* invstd, const float* mean, float* dgamma, float* dbeta, int accumulate,
                          float* coef /* [c][3]: dz = m*(c0*dy + c1*a + c2) */, pcuda_stream_t s);
int pcuda_bn_bwd_apply(const float* dy, long long dy_sn, long long dy_sc, const float* dy2, long long dy2_sn,
                       long long dy2_sc, const float* a, long long a_sn, long long a_sc, const float* coef,
                       const float* scale, const float* shift, int post_relu, float act_slope, float* dz,
                       long long dz_sn, long long dz_sc, int n, int c, long long hw, pcuda_stream_t s);
/* dz = dy * (a > 0 ? 1 : slope)   (LeakyReLU backward from the saved OUTPUT, as the in-place
 * nn.LeakyReLU of unet.py:24 / GAN.py:108 does); optional second addend dy2 */
int pcuda_lrelu_bwd(const float* dy, long long dy_sn, long long dy_sc, const float* dy2, long long dy2_sn,
                    long long dy2_sc, const float* a, long long a_sn, long long a_sc, float slope, float* dz,
                    long long dz_sn, long long dz_sc, int n, int c, long long hw, pcuda_stream_t s);
/* per-channel sum over batch and space (bias gradients) db (+)= sum dz */
int pcuda_channel_sum(const float* dz, long long sn, long long sc, int n, int c, long long hw, float* db,
                      int accumulate, float* workspace, size_t workspace_bytes, pcuda_stream_t s);

/* ------------------------------------------------------------------------------------
 * pooling / resampling / adds (unet.py:48 MaxPool2d(2,2); :111 UpsamplingNearest2d backward)
 * ---------------------------------------------------------------------------------- */
/* y = maxpool2x2(x*scale + shift) with argmax code (0..3) saved as uint8 */
int pcuda_maxpool2_fwd(const float* x, long long x_sn, long long x_sc, const float* scale, const float* shift,
                       float* y, long long y_sn, long long y_sc, uint8_t* idx, int n, int c, int h, int w,
                       pcuda_stream_t s);
/* dx (+)= scatter(dy (+ dy2)) through idx; positions not selected get 0 (or keep when accumulate) */
int pcuda_maxpool2_bwd(const float* dy, long long dy_sn, long long dy_sc, const float* dy2, long long dy2_sn,
                       long long dy2_sc, const uint8_t* idx, float* dx, long long dx_sn, long long dx_sc,
                       int accumulate, int n, int c, int h, int w, pcuda_stream_t s);
/* The backward kernels behind a max-pool, with the pool's scatter read IN PLACE (round 4): the gradient of element (y, x) is
 * g[y/2][x/2] (+ g2) where idx[y/2][x/2] == 2 (y & 1) + (x & 1), else 0 -- what pcuda_maxpool2_bwd would have written into a
 * 4x larger tensor for these kernels to read back (unet.py:48 going back into :27-36 / the residual convolution).  `dy` is an
 * optional full-resolution addend (the skip connection's gradient); results equal the two-kernel form bit for bit. */
typedef struct pcuda_pooled {
  const float* g; long long g_sn, g_sc;     /* gradient at the pooled resolution [n][c][h/2][w/2] (strided planes) */
  const float* g2; long long g2_sn, g2_sc;  /* optional second share, summed (NULL: none) */
  const uint8_t* idx;                       /* dense [n][c][h/2][w/2], from pcuda_maxpool2_fwd */
  int h, w;                                 /* the FULL-resolution plane, both even */
} pcuda_pooled;
int pcuda_bn_bwd_reduce_pooled(const pcuda_pooled* pool, const float* dy, long long dy_sn, long long dy_sc, const float* a,
                               long long a_sn, long long a_sc, const float* mean, const float* invstd, int n, int c,
                               float* red, int* ntiles, pcuda_stream_t s);
int pcuda_bn_bwd_apply_pooled(const pcuda_pooled* pool, const float* dy, long long dy_sn, long long dy_sc, const float* a,
                              long long a_sn, long long a_sc, const float* coef, float act_slope, float* dz, long long dz_sn,
                              long long dz_sc, int n, int c, pcuda_stream_t s);
int pcuda_lrelu_bwd_pooled(const pcuda_pooled* pool, const float* dy, long long dy_sn, long long dy_sc, const float* a,
                           long long a_sn, long long a_sc, float slope, float* dz, long long dz_sn, long long dz_sc, int n,
                           int c, pcuda_stream_t s);
/* dx[h][w] (+)= sum of the 2x2 block of dy[2h][2w] (nearest-upsample backward) */
int pcuda_upsample2_bwd(const float* dy, long long dy_sn, long long dy_sc, float* dx, long long dx_sn,
                        long long dx_sc, int accumulate, int n, int c, int h, int w, pcuda_stream_t s);
/* ... with the BatchNorm-backward reduce of the layer that consumes dx fused in (the decoder's conv -> LeakyReLU -> BN
 * blocks, unet.py:116-125, going back): red[ntiles][c][2] = (sum g, sum g * (a - mean) * invstd) per workgroup, the input of
 * pcuda_bn_bwd_finalize; red == NULL only reports ntiles */
int pcuda_upsample2_bwd_bnred(const float* dy, long long dy_sn, long long dy_sc, float* dx, long long dx_sn,
                              long long dx_sc, int accumulate, const float* a, long long a_sn, long long a_sc,
                              const float* mean, const float* invstd, float* red, int* ntiles, int n, int c, int h, int w,
                              pcuda_stream_t s);
/* bilinear resize, align_corners = True (nn.UpsamplingBilinear2d(size=(224, 224)) in OutputDiscriminator, GAN.py:57,78):
 * y dense [n][c][oh][ow]; ATen's arithmetic (src = dst * (in-1)/(out-1)).  bwd: dx = J^T dy, gather form (deterministic) */
int pcuda_bilinear_fwd(const float* x, long long x_sn, long long x_sc, int n, int c, int h, int w, float* y, int oh,
                       int ow, pcuda_stream_t s);
int pcuda_bilinear_bwd(const float* dy, int n, int c, int oh, int ow, float* dx, long long dx_sn, long long dx_sc,
                       int h, int w, pcuda_stream_t s);
/* taps unfolded into channels (u: dense [n][c*k*k][oh][ow]); a k x k layer over few input channels becomes a 1x1
 * layer over c*k*k of them (the discriminators' first layer, GAN.py:96 / :120-124) */
int pcuda_unfold_taps(const float* x, long long x_sn, long long x_sc, int n, int c, int h, int w, int k, int stride,
                      int pad, int dil, float* u, int oh, int ow, pcuda_stream_t s);
/* y = a + b (+ c) (+ d), flat fp32 (bottleneck running sum, unet.py:68-73; gradient joins) */
int pcuda_add4(const float* a, const float* b, const float* c, const float* d, float* y, long long numel,
               pcuda_stream_t s);
/* y = a * b, flat fp32 (applies a precomputed nn.Dropout mask, PointNetCls.py:179,209) */
int pcuda_mul(const float* a, const float* b, float* y, long long numel, pcuda_stream_t s);

/* ------------------------------------------------------------------------------------
 * losses / entropy (train_mscmrseg.py:202-203,222-226; train_mmwhs.py:212-224,242; loss.py)
 * ---------------------------------------------------------------------------------- */
#define PCUDA_ACT_SIGMOID 0
#define PCUDA_ACT_SOFTMAX 1
/* e = -p*log(p + 1e-7) * norm, p = sigmoid|softmax(logits); optionally also writes p */
int pcuda_entropy_fwd(const float* logits, int mode, float norm, float* ent, float* prob, int n, int c,
                      long long hw, pcuda_stream_t s);
/* dlogits (+)= d(ent)/dlogits . dent  (+ dprob through p when dprob != NULL) */
int pcuda_entropy_bwd(const float* logits, int mode, float norm, const float* dent, const float* dprob,
                      float* dlogits, int accumulate, int n, int c, long long hw, pcuda_stream_t s);
/* the same with the gradient of the map's mean on top: dmean (device scalar, may be NULL) is d/d[mean over n and pixels of
 * sum_c ent] -- the entropy terms of train_mmwhs.py:225-230 (-etpls) and :243-247 (-Tetpls); every element receives
 * dmean / (n * hw) in addition to its dent (which may then be NULL) */
int pcuda_entropy_bwd2(const float* logits, int mode, float norm, const float* dent, const float* dprob, const float* dmean,
                       float* dlogits, int accumulate, int n, int c, long long hw, pcuda_stream_t s);
/* out[0] = scale * sum(x[0..numel)), fixed summation order (deterministic): torch.mean(torch.sum(uncertainty_map, dim=1))
 * of train_mmwhs.py:225,243 with scale = 1 / (n * hw) */
size_t pcuda_sum_all_workspace_size(void);
int pcuda_sum_all(const float* x, long long numel, double scale, float* out, void* workspace, size_t workspace_bytes,
                  pcuda_stream_t s);
/* segmentation loss: mode SIGMOID: BCE(sigmoid(o), y) + jaccard(sigmoid(o), y)
 *                    mode SOFTMAX: cross_entropy(softmax(o), argmax y) ("double softmax") + jaccard(softmax(o), y)
 * y: one-hot uint8 [n][c][hw].  out[0] = bce|ce, out[1] = jaccard.  workspace from *_workspace_size.
 * pass 1 (fwd) leaves the per-class sums in the workspace; bwd reuses them. */
size_t pcuda_seg_loss_workspace_size(int n, int c, long long hw);
int pcuda_seg_loss_fwd(const float* logits, const uint8_t* onehot, int mode, int n, int c, long long hw,
                       float* out2, void* workspace, size_t workspace_bytes, pcuda_stream_t s);
/* dlogits = g_main * d(bce|ce)/do + g_jac * d(jaccard)/do */
int pcuda_seg_loss_bwd(const float* logits, const uint8_t* onehot, int mode, int n, int c, long long hw,
                       const float* g_main, const float* g_jac, float* dlogits, const void* workspace,
                       pcuda_stream_t s);
/* mean BCE-with-logits against a constant label; acc (optional) = mean(sigmoid(x) >= .5)
 * (train_mscmrseg.py:224-226,270-273) */
int pcuda_bce_const_fwd(const float* x, long long numel, float label, float* loss, float* acc, pcuda_stream_t s);
int pcuda_bce_const_bwd(const float* x, long long numel, float label, const float* gout, float gscale, float* dx,
                        pcuda_stream_t s);
/* batch_NN_loss (loss.py:40-76): x,y [b][npts][3]; loss scalar; workspaces: idx_ws int32 [2][b][npts],
 * val_ws float [pcuda_nn_loss_workspace_floats(b, npts)] = [2][b][npts] + [2][b][ceil(npts / 64)] (partial sums of the
 * minima per block of 64 points; ABI 5: ask the library instead of sizing it by formula) */
size_t pcuda_nn_loss_workspace_floats(int b, int npts);
int pcuda_nn_loss_fwd(const float* x, const float* y, int b, int npts, float* loss, int* idx_ws, float* val_ws,
                      pcuda_stream_t s);
int pcuda_nn_loss_bwd(const float* x, const float* y, int b, int npts, const int* idx_ws, const float* val_ws,
                      const float* gout, float* dx, pcuda_stream_t s);
/* per-step host metric moved on-device (utils.py:32-40 + metric.py:5-36): hard = (o == max_c o);
 * dice = mean_{c>=1} (2*sum(y*hard)+1)/(sum y + sum hard + 1).  workspace: 3*c doubles. */
int pcuda_dice_metric(const float* logits, const uint8_t* onehot, int n, int c, long long hw, float* dice,
                      void* workspace, size_t workspace_bytes, pcuda_stream_t s);
/* loader-side batch assembly (data_generator_mmwhs.py:265-272, utils.py:7-29, crop_volume :134-137): centre crop
 * (crop = the generator's crop_size, 0 = none), channel-last -> channel-first and integer labels -> one-hot uint8
 * in one pass.  images_hwc [b][h][w][c] fp32, mask_labels [b][h][w] int32 (may be NULL together with onehot),
 * outputs images_chw [b][c][oh][ow], onehot [b][num_classes][oh][ow] */
int pcuda_assemble_batch(const float* images_hwc, const int* mask_labels, int b, int h, int w, int c, int crop,
                         int num_classes, float* images_chw, uint8_t* onehot, pcuda_stream_t s);
/* validation metrics (train_mscmrseg.py:85-92, metric.py:39-82): labels[n][i] = first channel holding the
 * per-pixel maximum of x[n][c][i] (fp32 logits, or a uint8 one-hot mask when x_is_u8); strides in elements */
int pcuda_argmax_labels(const void* x, int x_is_u8, long long sn, long long sc, int n, int c, long long hw,
                        uint8_t* labels, pcuda_stream_t s);
/* dice[k] = 2|pred==k & gt==k| / (|pred==k| + |gt==k|), 0 if both empty (medpy dc), k < c.
 * workspace: 3*c 64-bit counters */
int pcuda_label_dice(const uint8_t* pred, const uint8_t* gt, long long numel, int c, float* dice, void* workspace,
                     size_t workspace_bytes, pcuda_stream_t s);

/* ------------------------------------------------------------------------------------
 * small dense ops of PointNetCls / the point head (PointNetCls.py; unet.py:86,94-95)
 * ---------------------------------------------------------------------------------- */
/* y[m][n] = x[m][k] . w[n][k]^T + b[n]   (nn.Linear; also conv1d k=1 seen as [B*L][C]) */
int pcuda_linear_fwd(const float* x, const float* w, const float* b, float* y, int m, int k, int n, pcuda_stream_t s);
int pcuda_linear_bwd_x(const float* dy, const float* w, float* dx, int m, int k, int n, int accumulate, pcuda_stream_t s);
/* workspace (optional; pcuda_linear_bwd_w_workspace_size bytes) enables a deterministic split-K for
 * long reductions into a small output (the point head's Linear over 300*B rows) */
size_t pcuda_linear_bwd_w_workspace_size(int m, int k, int n);
int pcuda_linear_bwd_w(const float* dy, const float* x, float* dw, float* db, int m, int k, int n, int accumulate,
                       void* workspace, size_t workspace_bytes, pcuda_stream_t s);
/* torch.nn.Conv1d(cin, cout, 1) on dense [b][c][l] point clouds (PointNetCls.py:26-28 STN3d, :76-78 STNkd, :116-131
 * PointNetfeat) in EXACT fp32 on the matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fma chain; no bf16
 * operand split).  w = [cout][cin] (the Conv1d weight with its unit kernel axis dropped), bias may be NULL.
 * fwd: y[b][co][l]; bn_partials (optional) = [pcuda_conv1d_k1_fwd_tiles(b, l)][cout][2] per-tile (sum, sum of squares)
 * of y for the BatchNorm1d that follows (pcuda_bn_finalize reduces them in a fixed order).
 * dgrad: dx[b][ci][l] = sum_co w[co][ci] dy[b][co][l].
 * wgrad: dw[co][ci] (+)= sum_{b,l} dy x, db[co] (+)= sum_{b,l} dy (db may be NULL); split-K slabs in the caller's
 * workspace, summed in a fixed order (deterministic). */
int pcuda_conv1d_k1_fwd_tiles(int b, int l);
int pcuda_conv1d_k1_fwd(const float* x, const float* w, const float* bias, float* y, int b, int cin, int cout, int l,
                        float* bn_partials, pcuda_stream_t s);
int pcuda_conv1d_k1_dgrad(const float* dy, const float* w, float* dx, int b, int cin, int cout, int l, pcuda_stream_t s);
size_t pcuda_conv1d_k1_wgrad_workspace_size(int b, int cin, int cout, int l);
int pcuda_conv1d_k1_wgrad(const float* x, const float* dy, float* dw, float* db, int b, int cin, int cout, int l,
                          int accumulate, void* workspace, size_t workspace_bytes, pcuda_stream_t s);
/* max over the last axis of x[b][c][l] with argmax (PointNetCls.py:44,162) */
int pcuda_max_points_fwd(const float* x, int b, int c, int l, float* y, int* idx, pcuda_stream_t s);
int pcuda_max_points_bwd(const float* dy, const int* idx, int b, int c, int l, float* dx, pcuda_stream_t s);
/* batched small matmul C[b] = op(A[b]) . op(B[b]); A [m][k] (or [k][m] when ta), B [k][n] (or [n][k] when tb) */
int pcuda_bmm(const float* a, const float* bmat, float* c, int batch, int m, int k, int n, int ta, int tb,
              int accumulate, pcuda_stream_t s);

/* jaccard_loss(true, logits=probabilities, eps, activation=False) as a free-standing function (utils/loss.py:5-37 the way
 * train_mscmrseg.py:203 / train_mmwhs.py:218 call it): probs fp32 dense [n][c][hw]; truth one-hot, fp32 or uint8;
 * loss = 1 - mean_c I_c / (S_c - I_c + eps).  bwd: dprobs = gout * d loss / d probs (workspace of the forward call) */
size_t pcuda_jaccard_workspace_size(int c);
int pcuda_jaccard_fwd(const float* probs, const void* truth, int truth_is_u8, int n, int c, long long hw, float eps,
                      float* loss, void* workspace, size_t workspace_bytes, pcuda_stream_t s);
int pcuda_jaccard_bwd(const void* truth, int truth_is_u8, int n, int c, long long hw, float eps, const float* gout,
                      float* dprobs, const void* workspace, pcuda_stream_t s);

/* ------------------------------------------------------------------------------------
 * mask -> surface point cloud sampler (utils/npy2point.py:7-18,101-125)
 * ---------------------------------------------------------------------------------- */
/* canonical surface-vertex list of b binary masks [b][h][w] (uint8, >0 = foreground):
 * verts int32 [b][max_verts][3] rows (z,y,x) in lexicographic order, z in {0,1,2}; counts[b].
 * workspace: b*h*w*4 + 4096 bytes */
int pcuda_surface_vertices(const uint8_t* mask, int b, int h, int w, int* verts, int max_verts, int* counts,
                           void* workspace, size_t workspace_bytes, pcuda_stream_t s);
/* second vertex-list mode: marching-cubes traversal order on the 3-slice stack (cells nested slice / row / column, per
 * cell the edges 6,5,10,0,1,2,3,4,7,8,9,11, one vertex per crossing edge at its background end, duplicates included);
 * replaces mcubes.marching_cubes(vol, 0) at utils/npy2point.py:112,121 as far as the un-vendored library's published
 * algorithm pins it (parity unpinned, oracle/sampler.py).  verts int32 [b][max_verts][3] rows (slice,row,col); counts[b]
 * = vertices found (may exceed max_verts: only the first max_verts are written) */
int pcuda_surface_vertices_mc(const uint8_t* mask, int b, int h, int w, int* verts, int max_verts, int* counts,
                              pcuda_stream_t s);
/* farthest point sampling (graipher): pts float64 [b][npts_max][3] with counts[b] valid rows;
 * first[b] = start index; out idx int32 [b][k].  Bit-exact with numpy float64 (no FMA contraction,
 * first-occurrence argmax).  counts[b] == 0 -> idx all -1 */
int pcuda_fps(const double* pts, const int* counts, const int* first, int b, int npts_max, int k, int* idx,
              pcuda_stream_t s);

/* ------------------------------------------------------------------------------------
 * optimisers on flat fp32 buffers (train_mscmrseg.py:427-455: Adam(b=(.9,.99)); SGD(m, wd))
 * ---------------------------------------------------------------------------------- */
int pcuda_adam_step(float* p, const float* g, float* m, float* v, long long numel, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, pcuda_stream_t s);
/* same update with the step count held in device memory: *step_dev is incremented on the stream first and then
 * used for the bias corrections, so a captured graph of the train step replays correctly */
int pcuda_adam_step_dev(float* p, const float* g, float* m, float* v, long long numel, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int* step_dev, float grad_scale,
                        pcuda_stream_t s);
int pcuda_sgd_step(float* p, const float* g, float* mom, long long numel, float lr, float momentum,
                   float weight_decay, int first_step, float grad_scale, pcuda_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* PCUDA_HIP_H */
