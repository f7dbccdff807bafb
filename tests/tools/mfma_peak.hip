// fp32 MFMA issue-rate probe (tuning aid, not a test).  Every wave runs ITER x 8 x 2 independent
// v_mfma_f32_32x32x2_f32 with NV plain VALU ops (v_fma_f32 on private registers) placed after each MFMA; no memory.
// Answers two questions the GEMM kernels are held against:
//   * the practical fp32 MFMA ceiling with constant vs pseudo-random operands (clock management), and
//   * whether VALU work of the SAME or of ANOTHER wave hides beside fp32 MFMAs or adds to them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0, int rnd) {
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0, b = b0;
  if (rnd) {  // per-lane pseudo-random mantissas
    unsigned h = (threadIdx.x + blockIdx.x * 256u) * 2654435761u;
    a = __uint_as_float(0x3f800000u | (h >> 9)) - 1.5f;
    b = __uint_as_float(0x3f800000u | ((h * 40503u) >> 9)) - 1.5f;
  }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, b);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) out[0] = s;
}
template <int NV>
int run(float* out, int wps, int rnd, const char* tag) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 3000, blocks = 256 * wps;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_loop<NV>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f, rnd);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  double flops = (double)blocks * 4 * iters * 8 * 2 * (32.0 * 32 * 2 * 2);
  printf("%-8s waves/SIMD %d  VALU per MFMA %d: %7.2f ms  %6.1f TFLOP/s\n", tag, wps, NV, ms, flops / ms / 1e9);
  return 0;
}
int main() {
  float* out; CK(hipMalloc(&out, 4));
  for (int rnd = 0; rnd < 2; ++rnd) {
    const char* tag = rnd ? "random" : "constant";
    for (int wps : {1, 2, 4}) {
      if (run<0>(out, wps, rnd, tag)) return 1;
      if (run<2>(out, wps, rnd, tag)) return 1;
      if (run<4>(out, wps, rnd, tag)) return 1;
      if (run<8>(out, wps, rnd, tag)) return 1;
      if (run<16>(out, wps, rnd, tag)) return 1;
    }
  }
  return 0;
}
