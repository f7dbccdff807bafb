// fp32 MFMA + LDS traffic probe (tuning aid, not a test): the GEMM main loop's MFMA block without global memory.
// Per chunk: 8 k-steps x [NR ds_read2_b32 -> s_waitcnt -> 2 MFMA], then NW ds_write_b32 and (BAR) one s_barrier.
// Reports TFLOP/s so that the cost of each ingredient beside the 64-cycle fp32 MFMA can be read off.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NR, int NW, int BAR, int THREADS>
__global__ __launch_bounds__(THREADS) void loop(float* out, int iters) {
  __shared__ float lds[2][16][260];
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 16 * 260; i += THREADS) (&lds[0][0][0])[i] = 1.0f + (i & 7);
  __syncthreads();
  float a0 = 1.f + lane, a1 = 2.f, b0 = 0.5f;
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = kk * 2 + lhi;
      if (NR >= 1) { a0 = lds[cur][k][(wave & 3) * 64 + l31]; a1 = lds[cur][k][(wave & 3) * 64 + 32 + l31]; }
      if (NR >= 2) { b0 = lds[cur][k][128 + (wave >> 2) * 32 + l31]; }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) lds[cur ^ 1][w][threadIdx.x & 255] = acc[0][w & 15];
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}
template <int NR, int NW, int BAR, int THREADS>
int run(float* out, int blocks_per_cu) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((loop<NR, NW, BAR, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  double flops = (double)blocks * (THREADS / 64) * iters * 16 * (32.0 * 32 * 2 * 2);
  printf("threads %d x %d blocks/CU  ds_read2/k-step %d  ds_write/chunk %d  barrier %d: %7.2f ms  %6.1f TFLOP/s\n", THREADS, blocks_per_cu,
         NR, NW, BAR, ms, flops / ms / 1e9);
  return 0;
}
int main() {
  float* out; CK(hipMalloc(&out, 4));
  if (run<0, 0, 0, 512>(out, 2)) return 1;
  if (run<1, 0, 0, 512>(out, 2)) return 1;
  if (run<2, 0, 0, 512>(out, 2)) return 1;
  if (run<2, 5, 0, 512>(out, 2)) return 1;
  if (run<2, 5, 1, 512>(out, 2)) return 1;
  if (run<2, 5, 1, 512>(out, 1)) return 1;
  if (run<2, 5, 1, 256>(out, 3)) return 1;
  if (run<2, 5, 1, 256>(out, 4)) return 1;
  if (run<2, 10, 1, 512>(out, 2)) return 1;
  return 0;
}
