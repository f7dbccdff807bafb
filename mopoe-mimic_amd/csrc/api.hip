// C-ABI plumbing: version, error text, launch checks, and the optional event-based kernel profiling that
// bench.py uses to measure the implicit-GEMM kernels live.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.hpp"

namespace mopoe {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MOPOE_ERR_LAUNCH;
  }
  return MOPOE_OK;
}

// ---- profiling ------------------------------------------------------------------------------------------
struct ProfRec {
  hipEvent_t start, stop;
  double flops, bytes;
  int kind;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_pool;   // created lazily, reused
static size_t g_prof_used = 0;

ProfScope::ProfScope(hipStream_t s, double flops, int kind, double bytes) : stream(s), slot(-1) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_used == g_prof_pool.size()) {
    ProfRec r;
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
    r.flops = 0;
    r.bytes = 0;
    r.kind = 0;
    g_prof_pool.push_back(r);
  }
  slot = (int)g_prof_used++;
  g_prof_pool[slot].flops = flops;
  g_prof_pool[slot].bytes = bytes;
  g_prof_pool[slot].kind = kind;
  (void)hipEventRecord(g_prof_pool[slot].start, stream);
}

ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof_pool[slot].stop, stream);
}

// one device timestamp (the 100 MHz constant clock) into *slot when the stream reaches this point
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

}  // namespace mopoe

using namespace mopoe;

extern "C" int mopoe_prof_stamp(uint64_t* slot, void* stream) {
  if (!slot) { set_error("prof_stamp: null slot"); return MOPOE_ERR_ARG; }
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
  return check_launch("prof_stamp");
}

extern "C" int mopoe_abi_version(void) { return MOPOE_ABI_VERSION; }

extern "C" const char* mopoe_last_error(void) { return g_err; }

extern "C" int mopoe_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  if (on) g_prof_used = 0;
  return MOPOE_OK;
}

extern "C" int mopoe_prof_collect(int64_t* launches, double* total_ms, double* total_flops, double* total_bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int k = 0; k < PROF_NKINDS; ++k) { launches[k] = 0; total_ms[k] = 0; total_flops[k] = 0; if (total_bytes) total_bytes[k] = 0; }
  for (size_t i = 0; i < g_prof_used; ++i) {
    if (hipEventSynchronize(g_prof_pool[i].stop) != hipSuccess) { set_error("prof_collect: event sync failed"); return MOPOE_ERR_LAUNCH; }
    float t = 0;
    if (hipEventElapsedTime(&t, g_prof_pool[i].start, g_prof_pool[i].stop) != hipSuccess) { set_error("prof_collect: elapsed failed"); return MOPOE_ERR_LAUNCH; }
    const int k = g_prof_pool[i].kind;
    launches[k] += 1;
    total_ms[k] += t;
    total_flops[k] += g_prof_pool[i].flops;
    if (total_bytes) total_bytes[k] += g_prof_pool[i].bytes;
  }
  g_prof_used = 0;
  return MOPOE_OK;
}
