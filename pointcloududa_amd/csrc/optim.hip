// Fused optimiser steps on flat fp32 parameter / gradient buffers.
// Adam(betas=(0.9,0.99)) for the segmenter, SGD(momentum, weight_decay=5e-4) for the
// discriminators (train_mscmrseg.py:427-455).  Arithmetic follows torch.optim's single-tensor
// path so one step from identical state matches to rounding.
#include "common.h"

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long numel,
                                                   float lr_over_bc1, float beta1, float beta2, float eps,
                                                   float weight_decay, float inv_sqrt_bc2, float grad_scale) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < numel; i += 256ll * gridDim.x) {
    float gi = g[i] * grad_scale;
    const float pi = p[i];
    if (weight_decay != 0.f) gi += weight_decay * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;        // exp_avg.lerp_(grad, 1-beta1)
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;   // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = pi - lr_over_bc1 * (mi / denom);
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ mom, long long numel, float lr, float momentum,
                                                  float weight_decay, int first_step, float grad_scale) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < numel; i += 256ll * gridDim.x) {
    const float pi = p[i];
    float gi = g[i] * grad_scale + weight_decay * pi;
    if (momentum != 0.f) {
      const float b = first_step ? gi : momentum * mom[i] + gi;
      mom[i] = b;
      gi = b;
    }
    p[i] = pi - lr * gi;
  }
}

extern "C" int pcuda_adam_step(float* p, const float* g, float* m, float* v, long long numel, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, float grad_scale,
                               pcuda_stream_t s) {
  if (!p || !g || !m || !v || numel <= 0 || step < 1) PCUDA_FAIL(PCUDA_E_BADARG, "adam_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float lr_over_bc1 = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const int blocks = (int)(cdiv(numel, 256) > 8192 ? 8192 : cdiv(numel, 256));
  ProfScope prof(PCUDA_FAM_POINTWISE, 28.0 * (double)numel, (hipStream_t)s);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, numel, lr_over_bc1, beta1,
                     beta2, eps, weight_decay, inv_sqrt_bc2, grad_scale);
  PCUDA_CHECK_LAUNCH("adam_kernel");
  return PCUDA_OK;
}

// Adam with the step count in device memory (incremented here, ahead of the update): a captured hipGraph of the
// train step replays correctly, which a host-side count baked into the kernel arguments would not.
__global__ void inc_i32_kernel(int* p) { *p += 1; }
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long long numel,
                                                       float lr, float beta1, float beta2, float eps, float weight_decay,
                                                       const int* __restrict__ step_dev, float grad_scale) {
  const int step = *step_dev;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float lr_over_bc1 = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < numel; i += 256ll * gridDim.x) {
    float gi = g[i] * grad_scale;
    const float pi = p[i];
    if (weight_decay != 0.f) gi += weight_decay * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = pi - lr_over_bc1 * (mi / denom);
  }
}

extern "C" int pcuda_adam_step_dev(float* p, const float* g, float* m, float* v, long long numel, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int* step_dev, float grad_scale,
                                   pcuda_stream_t s) {
  if (!p || !g || !m || !v || !step_dev || numel <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "adam_step_dev: bad arguments");
  const int blocks = (int)(cdiv(numel, 256) > 8192 ? 8192 : cdiv(numel, 256));
  ProfScope prof(PCUDA_FAM_POINTWISE, 28.0 * (double)numel, (hipStream_t)s);
  hipLaunchKernelGGL(inc_i32_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, step_dev);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, numel, lr, beta1, beta2,
                     eps, weight_decay, (const int*)step_dev, grad_scale);
  PCUDA_CHECK_LAUNCH("adam_dev_kernel");
  return PCUDA_OK;
}

extern "C" int pcuda_sgd_step(float* p, const float* g, float* mom, long long numel, float lr, float momentum,
                              float weight_decay, int first_step, float grad_scale, pcuda_stream_t s) {
  if (!p || !g || (momentum != 0.f && !mom) || numel <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "sgd_step: bad arguments");
  const int blocks = (int)(cdiv(numel, 256) > 8192 ? 8192 : cdiv(numel, 256));
  ProfScope prof(PCUDA_FAM_POINTWISE, 20.0 * (double)numel, (hipStream_t)s);
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, p, g, mom, numel, lr, momentum,
                     weight_decay, first_step, grad_scale);
  PCUDA_CHECK_LAUNCH("sgd_kernel");
  return PCUDA_OK;
}
