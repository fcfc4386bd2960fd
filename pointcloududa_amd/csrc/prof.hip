// Error text + per-family kernel timing with HIP events on the launch stream.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void pcuda_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pcuda_last_error(void) { return g_err; }

#include <set>
#include "variants.h"
void variant_log(const char* family, unsigned key) {
  static const char* path = getenv("PCUDA_VARIANT_LOG");
  if (!path || !*path) return;
  static std::mutex mu;
  static std::set<std::pair<std::string, unsigned>> seen;
  std::lock_guard<std::mutex> lk(mu);
  if (!seen.insert({family, key}).second) return;
  if (FILE* f = fopen(path, "a")) {
    fprintf(f, "%s %u\n", family, key);
    fclose(f);
  }
}
static long long g_fallbacks = 0;
extern "C" long long pcuda_fallback_count(void) { return __atomic_load_n(&g_fallbacks, __ATOMIC_RELAXED); }
void variant_fallback_note(const char* family, unsigned key) {
  __atomic_fetch_add(&g_fallbacks, 1ll, __ATOMIC_RELAXED);
  static std::mutex mu;
  static std::set<std::pair<std::string, unsigned>> seen;
  std::lock_guard<std::mutex> lk(mu);
  if (!seen.insert({family, key}).second) return;
  fprintf(stderr, "libpcuda_hip: %s variant %u is not in this build (pruned to the shapes of the benchmark configurations and "
                  "tests): running on the generic kernel; `make FULL=1` builds every variant\n", family, key);
}

long long g_pcuda_launches = 0;
extern "C" long long pcuda_launch_count(int reset) {
  return reset ? __atomic_exchange_n(&g_pcuda_launches, 0ll, __ATOMIC_RELAXED) : __atomic_load_n(&g_pcuda_launches, __ATOMIC_RELAXED);
}
// sha256[:16] over the kernel sources this library was compiled from (the Makefile writes build/srchash.h with the
// same recipe as pointcloududa_amd._lib.csrc_hash): a profile is stamped with THIS, not with the tree's hash
#include "srchash.h"
extern "C" const char* pcuda_build_hash(void) { return PCUDA_SRC_HASH; }
extern "C" int pcuda_version(void) { return PCUDA_ABI_VERSION; }
extern "C" size_t pcuda_abi_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(pcuda_conv_geom);
    case 1: return sizeof(pcuda_src);
    case 2: return sizeof(pcuda_dst);
    case 3: return sizeof(pcuda_pooled);
    case 4: return sizeof(pcuda_reduce_job);
    default: return 0;
  }
}
// which kernel the dispatcher picked for the calling thread's most recent convolution launch (tests assert it)
static thread_local char g_last_tag[200] = "";
static thread_local char g_last_kern[64] = "";
static thread_local char g_last_both[272] = "";
void note_kernel(const char* name) {
  snprintf(g_last_kern, sizeof(g_last_kern), "%s", name ? name : "");
}
extern "C" const char* pcuda_last_kernel(void) {
  snprintf(g_last_both, sizeof(g_last_both), "%s | %s", g_last_tag, g_last_kern);
  return g_last_both;
}
extern "C" int pcuda_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---- shader clock under a matrix-core load (bench.py: `clock_ghz_under_load`).  Boxes of one pool differ by up to 10 % on one
// binary; this probe says how fast THIS box clocks its CUs while every SIMD issues MFMAs back to back, so that two rounds'
// lines can be normalised.  s_memtime counts shader-engine clocks, s_memrealtime the constant 100-MHz reference.
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* out2, float* sink, int iters) {
  probe_f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  probe_bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)(float)(i + 1); }
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
  asm volatile("s_nop 0" ::"v"(s));
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) {
    atomicAdd(&out2[0], c1 - c0);
    atomicAdd(&out2[1], r1 - r0);
  }
}

extern "C" int pcuda_clock_probe(unsigned long long* out2_dev, float* sink_dev, int iters, pcuda_stream_t s) {
  if (!out2_dev || !sink_dev || iters <= 0) PCUDA_FAIL(PCUDA_E_BADARG, "clock_probe: bad arguments");
  if (hipMemsetAsync(out2_dev, 0, 16, (hipStream_t)s) != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "clock_probe: memset failed");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(512), dim3(256), 0, (hipStream_t)s, out2_dev, sink_dev, iters);
  PCUDA_CHECK_LAUNCH("clock_probe_kernel");
  return PCUDA_OK;
}

namespace {
struct Rec {
  hipEvent_t e0, e1;
  int fam;
  double work;
  std::string tag;
};
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_free;

hipEvent_t get_event() {
  if (!g_free.empty()) {
    hipEvent_t e = g_free.back();
    g_free.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

ProfScope::ProfScope(int family, double work, hipStream_t s, const char* tag) : fam(family), ev0(nullptr), stream(s) {
  if (tag) {   // (tagged launches are the convolution families: tens of nanoseconds next to a launch)
    snprintf(g_last_tag, sizeof(g_last_tag), "%s", tag);
    const char* sp = strchr(tag, ' ');   // default kernel name = the tag's first word(s): "wgrad3r", "direct", ...
    snprintf(g_last_kern, sizeof(g_last_kern), "%.*s", sp ? (int)(sp - tag) : (int)strlen(tag), tag);
  }
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  r.e0 = get_event();
  r.e1 = get_event();
  r.fam = family;
  r.work = work;
  if (tag) r.tag = tag;
  if (!r.e0 || !r.e1) return;
  (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
  ev0 = (void*)r.e1;
}

ProfScope::~ProfScope() {
  if (ev0) (void)hipEventRecord((hipEvent_t)ev0, stream);
}

extern "C" int pcuda_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  return PCUDA_OK;
}

extern "C" int pcuda_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.e1);
    g_free.push_back(r.e0);
    g_free.push_back(r.e1);
  }
  g_recs.clear();
  return PCUDA_OK;
}

extern "C" int pcuda_prof_read(int family, double* ms, double* work, long long* launches) {
  if (family < 0 || family >= PCUDA_FAM_COUNT) PCUDA_FAIL(PCUDA_E_BADARG, "prof_read: bad family %d", family);
  std::lock_guard<std::mutex> lk(g_mu);
  double t = 0, w = 0;
  long long n = 0;
  for (auto& r : g_recs) {
    if (r.fam != family) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "prof_read: event sync failed");
    float dt = 0.f;
    if (hipEventElapsedTime(&dt, r.e0, r.e1) != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "prof_read: elapsed failed");
    t += dt;
    w += r.work;
    ++n;
  }
  if (ms) *ms = t;
  if (work) *work = w;
  if (launches) *launches = n;
  return PCUDA_OK;
}

// debugging aid: one CSV line per recorded launch (family,work,ms,tag)
extern "C" int pcuda_prof_dump(const char* path) {
  std::lock_guard<std::mutex> lk(g_mu);
  FILE* f = fopen(path, "w");
  if (!f) PCUDA_FAIL(PCUDA_E_BADARG, "prof_dump: cannot open %s", path);
  fprintf(f, "family,work,ms,tag\n");
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.e1);
    float dt = 0.f;
    (void)hipEventElapsedTime(&dt, r.e0, r.e1);
    fprintf(f, "%d,%.6g,%.6f,%s\n", r.fam, r.work, dt, r.tag.c_str());
  }
  fclose(f);
  return PCUDA_OK;
}
