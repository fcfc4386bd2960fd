// Stage-complete micro-benchmark of the anti-phase 3x3 kernel (csrc/conv_ap_impl.h): real global -> LDS staging from fp32
// NCHW with the lazy affine, real weight DMA, real transposed epilogue with bias + LeakyReLU + BatchNorm partial sums.
// Checks a sample of outputs and the partial sums against a float64 reference on the host, then times the launch.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-memory-clause \
//          -Ipointcloududa_amd/csrc scripts/micro/conv3ap_micro.hip -o scripts/micro/bin/conv3ap_micro
//   run:   conv3ap_micro n cin cout H W [iters] [dbg bits] [clk]
#include "conv_ap_impl.h"
#include <vector>
#include <random>
#include <cmath>
#include <cstdlib>

long long g_pcuda_launches = 0;
void pcuda_set_error(const char*, ...) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2, cin = argc > 2 ? atoi(argv[2]) : 64, cout = argc > 3 ? atoi(argv[3]) : 64;
  const int H = argc > 4 ? atoi(argv[4]) : 32, W = argc > 5 ? atoi(argv[5]) : 32;
  const int iters = argc > 6 ? atoi(argv[6]) : 20, dbg = argc > 7 ? atoi(argv[7]) : 0, clkmode = argc > 8 ? atoi(argv[8]) : 0;
  if (!ap_geom_ok(cout, cin, H, W)) { printf("geometry not supported\n"); return 1; }
  const size_t nx = (size_t)n * cin * H * W, ny = (size_t)n * cout * H * W, nw = (size_t)cout * cin * 9;
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> hx(nx), hw(nw), hb(cout), hsc(cin), hsh(cin);
  for (auto& v : hx) v = U(rng);
  const float ws = sqrtf(2.f / (9.f * cin));
  for (auto& v : hw) v = U(rng) * ws * 1.7f;
  for (auto& v : hb) v = U(rng) * 0.1f;
  for (auto& v : hsc) v = 1.f + 0.3f * U(rng);
  for (auto& v : hsh) v = 0.2f * U(rng);
  const int datamode = getenv("AP_DATA") ? atoi(getenv("AP_DATA")) : 0;   // 1: zero input and weights (power / clock experiment), 2: zero weights only
  if (datamode == 1) { for (auto& v : hx) v = 0.f; for (auto& v : hsh) v = 0.f; }
  if (datamode >= 1) for (auto& v : hw) v = 0.f;
  float *dx, *dw, *db, *dsc, *dsh, *dy, *dst;
  unsigned char* dimg;
  const int tiles_x = W / 32, tiles_y = H / 8, ntiles = n * tiles_x * tiles_y;
  CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, cout * 4)); CK(hipMalloc(&dsc, cin * 4));
  CK(hipMalloc(&dsh, cin * 4)); CK(hipMalloc(&dy, ny * 4)); CK(hipMalloc(&dst, (size_t)ntiles * cout * 2 * 4));
  CK(hipMalloc(&dimg, ap_packed_bytes(cout, cin)));
  CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, hsc.data(), cin * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsh, hsh.data(), cin * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0xff, ny * 4)); CK(hipMemset(dst, 0xff, (size_t)ntiles * cout * 8));
  {
    ApPackParams pp; pp.w = dw; pp.s_row = (long long)cin * 9; pp.s_red = 9; pp.rows = cout; pp.red = cin; pp.flip = 0; pp.out = dimg;
    hipLaunchKernelGGL(ap_pack_kernel, dim3(1024), dim3(256), 0, 0, pp);
    CK(hipDeviceSynchronize());
  }
  ApParams p;
  memset(&p, 0, sizeof(p));
  p.x.p1 = dx; p.x.sn1 = (long long)cin * H * W; p.x.sc1 = (long long)H * W; p.x.scale1 = dsc; p.x.shift1 = dsh; p.x.c1 = cin;
  p.cin = cin; p.H = H; p.W = W;
  p.y.p1 = dy; p.y.sn1 = (long long)cout * H * W; p.y.sc1 = (long long)H * W; p.y.c1 = cout;
  p.cout = cout; p.wimg = dimg; p.bias = db; p.slope = 0.01f; p.stats = dst;
  p.tiles_x = tiles_x; p.tiles_y = tiles_y; p.n = n; p.n_co_tiles = cout / 64; p.nchunks = cin / 16;
  p.total = (ntiles / 2) * p.n_co_tiles;
  p.dbg = dbg;
  unsigned long long* dclk;
  CK(hipMalloc(&dclk, 16 * 8)); CK(hipMemset(dclk, 0, 128));
  p.dbg_clk = dclk;
  if (ntiles & 1) { printf("odd tile count\n"); return 1; }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int grid = std::min(prop.multiProcessorCount, p.total);
  auto kern = clkmode == 1 ? conv3ap_kernel<1, false, true, false> : (clkmode == 2 ? conv3ap_kernel<1, false, false, true> : conv3ap_kernel<1, false, false, false>);
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, AP_LDS_BYTES));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), AP_LDS_BYTES, 0, p);
  CK(hipDeviceSynchronize());
  printf("n %d cin %d cout %d %dx%d: grid %d, items %d, LDS %d\n", n, cin, cout, H, W, grid, p.total, AP_LDS_BYTES);
  if (!dbg) {
    std::vector<float> hy(ny), hst((size_t)ntiles * cout * 2);
    CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hst.data(), dst, hst.size() * 4, hipMemcpyDeviceToHost));
    double ymax = 0; for (auto v : hy) ymax = std::max(ymax, (double)fabs(v));
    double worst = 0; int nan = 0;
    std::uniform_int_distribution<int> Rn(0, n - 1), Rc(0, cout - 1), Ry(0, H - 1), Rx(0, W - 1);
    const int nsamp = 6000;
    for (int sidx = 0; sidx < nsamp; ++sidx) {
      int in_ = Rn(rng), co = Rc(rng), y = Ry(rng), x = Rx(rng);
      if (sidx < 64) { y = (sidx & 1) ? H - 1 : 0; x = (sidx & 2) ? W - 1 : 0; }   // corners
      double a = hb[co];
      for (int ci = 0; ci < cin; ++ci)
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            const int iy = y + ky - 1, ix = x + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const float xv = fmaf(hx[((size_t)(in_ * cin + ci) * H + iy) * W + ix], hsc[ci], hsh[ci]);
            a += (double)xv * hw[((size_t)co * cin + ci) * 9 + ky * 3 + kx];
          }
      const double ref = a > 0 ? a : a * 0.01;
      const float got = hy[((size_t)(in_ * cout + co) * H + y) * W + x];
      if (!(got == got)) ++nan;
      worst = std::max(worst, fabs(got - ref));
    }
    // partial sums against the stored values
    double sworst = 0, smax = 0;
    for (int pt = 0; pt < ntiles; pt += std::max(1, ntiles / 64)) {
      const int txi = pt % tiles_x, tmp = pt / tiles_x, tyi = tmp % tiles_y, in_ = tmp / tiles_y;
      for (int co = 0; co < cout; co += 7) {
        double s1 = 0, s2 = 0;
        for (int yy = 0; yy < 8; ++yy)
          for (int xx = 0; xx < 32; ++xx) {
            const double v = hy[((size_t)(in_ * cout + co) * H + tyi * 8 + yy) * W + txi * 32 + xx];
            s1 += v; s2 += v * v;
          }
        sworst = std::max(sworst, std::max(fabs(hst[((size_t)pt * cout + co) * 2] - s1), fabs(hst[((size_t)pt * cout + co) * 2 + 1] - s2)));
        smax = std::max(smax, std::max(fabs(s1), fabs(s2)));
      }
    }
    printf("check: max |y| %.3f, worst abs err %.3e (rel %.2e), NaN %d; stats worst %.3e of %.2f\n", ymax, worst, worst / ymax, nan, sworst, smax);
    if (worst / ymax > 1e-4 || nan || sworst / smax > 1e-4) printf("CHECK FAILED\n"); else printf("CHECK OK\n");
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), AP_LDS_BYTES, 0, p);
  CK(hipMemset(dclk, 0, 128));
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), AP_LDS_BYTES, 0, p);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, fl = 2.0 * n * H * W * (double)cout * cin * 9;
  printf("time %.1f us  %.1f TFLOP/s algorithmic  (%.2f TB/s of in + out bytes)  dbg %d\n", us, fl / us * 1e-6, (nx + ny) * 4.0 / us * 1e-6, dbg);
  if (clkmode == 1) {
    unsigned long long hc[16]; CK(hipMemcpy(hc, dclk, 128, hipMemcpyDeviceToHost));
    const char* names[8] = {"loop top", "mfma seg", "w dma issue", "epilogue", "x wait + commit", "x issue", "barrier", "mfma tail (B: dma wait)"};
    for (int g = 0; g < 2; ++g) {
      double tot = 0; for (int i = 0; i < 8; ++i) tot += hc[g * 8 + i];
      printf("group %c:", 'A' + g);
      for (int i = 0; i < 8; ++i) printf("  %s %.1f%%", names[i], 100.0 * hc[g * 8 + i] / tot);
      printf("  (%.0f cycles per launch and workgroup)\n", tot / iters / grid);
    }
  }
  return 0;
}
