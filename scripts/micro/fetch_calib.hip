// FETCH_SIZE / WRITE_SIZE calibration for the load forms the convolution kernels use (MI355X guide, HBM section: on
// gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read; other widths are uncalibrated).  Every kernel reads
// (or writes) a KNOWN byte count from a 1 GiB buffer -- far past the 256 MiB infinity cache -- exactly once:
//   k_x4      global_load_dwordx4, 16 B per lane, consecutive lanes consecutive (pointwise kernels, dZ rows of wgrad)
//   k_x1      global_load_dword, 4 B per lane, consecutive (the dword staging path)
//   k_buf1    raw_buffer_load_b32 with the lane offset in voffset (the buffer staging path of igemm / wgrad)
//   k_rows    dwordx4 loads of 160-byte row segments at a 1 KiB pitch starting 16 B before a 128-B line (the quad staging
//             of a 32-pixel-wide tile with its halo: 3 lines touched per 160 useful bytes)
//   k_w4      global_store_dwordx4
// run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib   (then WRITE_SIZE in a second pass)
// The program prints the bytes each kernel requested; scripts/micro/fetch_calib.sh divides the counters by them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_x4(const float4* __restrict__ p, long long n4, float* out) {
  float s = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.678f) out[0] = s;
}
__global__ void k_x1(const float* __restrict__ p, long long n, float* out) {
  float s = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ void k_buf1(const float* __restrict__ p, long long n, float* out) {
  // 1 GiB in 16 windows of 64 MiB: a buffer resource spans < 2^30 bytes here, as in the library
  float s = 0.f;
  const long long win = 1ll << 24;   // elements per window
  for (long long w0 = 0; w0 < n; w0 += win) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p + w0), 0, (int)(win * 4), 0x00020000);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < win; i += (long long)gridDim.x * blockDim.x)
      s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(i * 4), 0, 0));
  }
  if (s == 12345.678f) out[0] = s;
}
// rows of 1 KiB; each row: 10 lanes x 16 B starting at byte 112 (16 B before the line at 128): 160 B across 3 lines
__global__ void k_rows(const char* __restrict__ p, long long nrows, float* out) {
  float s = 0.f;
  const int q = threadIdx.x % 10, rl = threadIdx.x / 10;      // 250 of 256 threads: 25 rows per block pass
  if (rl < 25)
    for (long long r = blockIdx.x * 25ll + rl; r < nrows; r += gridDim.x * 25ll) {
      const float4 v = *(const float4*)(p + r * 1024 + 112 + q * 16);
      s += v.x + v.y + v.z + v.w;
    }
  if (s == 12345.678f) out[0] = s;
}
__global__ void k_w4(float4* __restrict__ p, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
  const long long bytes = 1ll << 30;
  char* buf; float* out;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(buf, 0, bytes));
  char* spoil; CHECK(hipMalloc(&spoil, bytes));      // a second buffer written between the kernels: evicts the cache
  const int grid = 256 * 16;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(spoil, rep, bytes));
    k_x4<<<grid, 256>>>((const float4*)buf, bytes / 16, out);
    CHECK(hipMemset(spoil, rep + 2, bytes));
    k_x1<<<grid, 256>>>((const float*)buf, bytes / 4, out);
    CHECK(hipMemset(spoil, rep + 4, bytes));
    k_buf1<<<grid, 256>>>((const float*)buf, bytes / 4, out);
    CHECK(hipMemset(spoil, rep + 6, bytes));
    k_rows<<<grid, 256>>>(buf, bytes / 1024, out);
    CHECK(hipMemset(spoil, rep + 8, bytes));
    k_w4<<<grid, 256>>>((float4*)buf, bytes / 16);
    CHECK(hipDeviceSynchronize());
  }
  printf("requested_bytes k_x4 %lld k_x1 %lld k_buf1 %lld k_rows_useful %lld k_rows_lines %lld k_w4 %lld\n", bytes, bytes, bytes,
         (bytes / 1024) * 160, (bytes / 1024) * 384, bytes);
  return 0;
}
