// How long do dependent VALU instructions of the split3 sequence take on one wave per SIMD?  (not a test)
//   hipcc -O3 --offload-arch=gfx950 -o tests/tools/valu_dep_probe tests/tools/valu_dep_probe.hip && tests/tools/valu_dep_probe
// mode 0: the 11-instruction split of one pair, dependent chain as the compiler orders it
// mode 1: two pairs interleaved by hand (22 instructions, dependent instructions >= 2 apart)
// mode 2: 11 independent v_add_f32 (issue floor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = seed * 1.5f + threadIdx.x, b0 = seed * 2.5f + threadIdx.x, b1 = seed * 3.5f + threadIdx.x;
  unsigned acc = 0;
  float f[11];
  for (int k = 0; k < 11; ++k) f[k] = seed + k;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        unsigned h, m, l; float r0, r1, q0, q1;
        asm volatile(
            "v_cvt_pk_bf16_f32 %0, %7, %8\n\t"
            "v_lshlrev_b32 %3, 16, %0\n\t"
            "v_and_b32 %4, 0xffff0000, %0\n\t"
            "v_sub_f32 %3, %7, %3\n\t"
            "v_sub_f32 %4, %8, %4\n\t"
            "v_cvt_pk_bf16_f32 %1, %3, %4\n\t"
            "v_lshlrev_b32 %5, 16, %1\n\t"
            "v_and_b32 %6, 0xffff0000, %1\n\t"
            "v_sub_f32 %5, %3, %5\n\t"
            "v_sub_f32 %6, %4, %6\n\t"
            "v_cvt_pk_bf16_f32 %2, %5, %6"
            : "=&v"(h), "=&v"(m), "=&v"(l), "=&v"(r0), "=&v"(r1), "=&v"(q0), "=&v"(q1) : "v"(r ? b0 : a0), "v"(r ? b1 : a1));
        acc ^= h ^ m ^ l;
      }
    } else if (MODE == 1) {
      unsigned h, m, l, h2, m2, l2; float r0, r1, q0, q1, s0, s1, t0, t1;
      asm volatile(
          "v_cvt_pk_bf16_f32 %0, %14, %15\n\t"
          "v_cvt_pk_bf16_f32 %7, %16, %17\n\t"
          "v_lshlrev_b32 %3, 16, %0\n\t"
          "v_and_b32 %4, 0xffff0000, %0\n\t"
          "v_lshlrev_b32 %10, 16, %7\n\t"
          "v_and_b32 %11, 0xffff0000, %7\n\t"
          "v_sub_f32 %3, %14, %3\n\t"
          "v_sub_f32 %4, %15, %4\n\t"
          "v_sub_f32 %10, %16, %10\n\t"
          "v_sub_f32 %11, %17, %11\n\t"
          "v_cvt_pk_bf16_f32 %1, %3, %4\n\t"
          "v_cvt_pk_bf16_f32 %8, %10, %11\n\t"
          "v_lshlrev_b32 %5, 16, %1\n\t"
          "v_and_b32 %6, 0xffff0000, %1\n\t"
          "v_lshlrev_b32 %12, 16, %8\n\t"
          "v_and_b32 %13, 0xffff0000, %8\n\t"
          "v_sub_f32 %5, %3, %5\n\t"
          "v_sub_f32 %6, %4, %6\n\t"
          "v_sub_f32 %12, %10, %12\n\t"
          "v_sub_f32 %13, %11, %13\n\t"
          "v_cvt_pk_bf16_f32 %2, %5, %6\n\t"
          "v_cvt_pk_bf16_f32 %9, %12, %13"
          : "=&v"(h), "=&v"(m), "=&v"(l), "=&v"(r0), "=&v"(r1), "=&v"(q0), "=&v"(q1),
            "=&v"(h2), "=&v"(m2), "=&v"(l2), "=&v"(s0), "=&v"(s1), "=&v"(t0), "=&v"(t1)
          : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
      acc ^= h ^ m ^ l ^ h2 ^ m2 ^ l2;
    } else if (MODE == 3) {
      unsigned h, m, l, h2, m2, l2; float r0, r1, q0, q1, s0, s1, t0, t1;
      asm volatile(
          "v_cvt_pk_bf16_f32 %0, %14, %15\n\t"
          "v_cvt_pk_bf16_f32 %7, %16, %17\n\t"
          "v_lshlrev_b32 %3, 16, %0\n\t"
          "v_and_b32 %4, %18, %0\n\t"
          "v_lshlrev_b32 %10, 16, %7\n\t"
          "v_and_b32 %11, %18, %7\n\t"
          "v_sub_f32 %3, %14, %3\n\t"
          "v_sub_f32 %4, %15, %4\n\t"
          "v_sub_f32 %10, %16, %10\n\t"
          "v_sub_f32 %11, %17, %11\n\t"
          "v_cvt_pk_bf16_f32 %1, %3, %4\n\t"
          "v_cvt_pk_bf16_f32 %8, %10, %11\n\t"
          "v_lshlrev_b32 %5, 16, %1\n\t"
          "v_and_b32 %6, %18, %1\n\t"
          "v_lshlrev_b32 %12, 16, %8\n\t"
          "v_and_b32 %13, %18, %8\n\t"
          "v_sub_f32 %5, %3, %5\n\t"
          "v_sub_f32 %6, %4, %6\n\t"
          "v_sub_f32 %12, %10, %12\n\t"
          "v_sub_f32 %13, %11, %13\n\t"
          "v_cvt_pk_bf16_f32 %2, %5, %6\n\t"
          "v_cvt_pk_bf16_f32 %9, %12, %13"
          : "=&v"(h), "=&v"(m), "=&v"(l), "=&v"(r0), "=&v"(r1), "=&v"(q0), "=&v"(q1),
            "=&v"(h2), "=&v"(m2), "=&v"(l2), "=&v"(s0), "=&v"(s1), "=&v"(t0), "=&v"(t1)
          : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "s"(0xffff0000u));
      acc ^= h ^ m ^ l ^ h2 ^ m2 ^ l2;
    } else if (MODE >= 4 && MODE <= 6) {
      unsigned u[11];
#pragma unroll
      for (int k = 0; k < 11; ++k) u[k] = __float_as_uint(f[k]);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (MODE == 4)
          asm volatile("v_and_b32 %0, 0xffff0000, %0\n\tv_and_b32 %1, 0xffff0000, %1\n\tv_and_b32 %2, 0xffff0000, %2\n\tv_and_b32 %3, 0xffff0000, %3\n\t"
                       "v_and_b32 %4, 0xffff0000, %4\n\tv_and_b32 %5, 0xffff0000, %5\n\tv_and_b32 %6, 0xffff0000, %6\n\tv_and_b32 %7, 0xffff0000, %7\n\t"
                       "v_and_b32 %8, 0xffff0000, %8\n\tv_and_b32 %9, 0xffff0000, %9\n\tv_and_b32 %10, 0xffff0000, %10"
                       : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]), "+v"(u[8]), "+v"(u[9]), "+v"(u[10]));
        else if (MODE == 5)
          asm volatile("v_cvt_pk_bf16_f32 %0, %0, %11\n\tv_cvt_pk_bf16_f32 %1, %1, %11\n\tv_cvt_pk_bf16_f32 %2, %2, %11\n\tv_cvt_pk_bf16_f32 %3, %3, %11\n\t"
                       "v_cvt_pk_bf16_f32 %4, %4, %11\n\tv_cvt_pk_bf16_f32 %5, %5, %11\n\tv_cvt_pk_bf16_f32 %6, %6, %11\n\tv_cvt_pk_bf16_f32 %7, %7, %11\n\t"
                       "v_cvt_pk_bf16_f32 %8, %8, %11\n\tv_cvt_pk_bf16_f32 %9, %9, %11\n\tv_cvt_pk_bf16_f32 %10, %10, %11"
                       : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]), "+v"(u[8]), "+v"(u[9]), "+v"(u[10]) : "v"(seed));
        else
          asm volatile("v_lshlrev_b32 %0, 16, %0\n\tv_lshlrev_b32 %1, 16, %1\n\tv_lshlrev_b32 %2, 16, %2\n\tv_lshlrev_b32 %3, 16, %3\n\t"
                       "v_lshlrev_b32 %4, 16, %4\n\tv_lshlrev_b32 %5, 16, %5\n\tv_lshlrev_b32 %6, 16, %6\n\tv_lshlrev_b32 %7, 16, %7\n\t"
                       "v_lshlrev_b32 %8, 16, %8\n\tv_lshlrev_b32 %9, 16, %9\n\tv_lshlrev_b32 %10, 16, %10"
                       : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]), "+v"(u[8]), "+v"(u[9]), "+v"(u[10]));
      }
#pragma unroll
      for (int k = 0; k < 11; ++k) f[k] = __uint_as_float(u[k] | 0x3f800000u);
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        asm volatile(
            "v_add_f32 %0, %0, %11\n\tv_add_f32 %1, %1, %11\n\tv_add_f32 %2, %2, %11\n\tv_add_f32 %3, %3, %11\n\t"
            "v_add_f32 %4, %4, %11\n\tv_add_f32 %5, %5, %11\n\tv_add_f32 %6, %6, %11\n\tv_add_f32 %7, %7, %11\n\t"
            "v_add_f32 %8, %8, %11\n\tv_add_f32 %9, %9, %11\n\tv_add_f32 %10, %10, %11"
            : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10])
            : "v"(seed));
    }
    a0 += 1.f; b0 += 1.f;
  }
  float s = 0.f;
  for (int k = 0; k < 11; ++k) s += f[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(acc & 0x3fffffffu) + a0 + b0;
}

template <int MODE>
double run(float* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 256>>>(d, 1000, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<256, 256>>>(d, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 256 * 256 * 4);
  const int iters = 200000;
  const double m0 = run<0>(d, iters), m1 = run<1>(d, iters), m2 = run<2>(d, iters), m3 = run<3>(d, iters), m4 = run<4>(d, iters),
               m5 = run<5>(d, iters), m6 = run<6>(d, iters);
  // 22 split instructions (+ 2 adds, loop overhead) per iteration, one wave per SIMD
  printf("per iteration of 22 VALU instructions, one wave per SIMD (ns; x clock = cycles):\n");
  printf("  mode 0 (two dependent chains of 11, one after the other): %.1f ns  = %.1f cycles/instr at 2.4 GHz\n", m0 * 1e6 / iters, m0 * 1e6 / iters * 2.4 / 22);
  printf("  mode 1 (the two chains interleaved):                      %.1f ns  = %.1f cycles/instr\n", m1 * 1e6 / iters, m1 * 1e6 / iters * 2.4 / 22);
  printf("  mode 2 (22 independent v_add_f32):                        %.1f ns  = %.1f cycles/instr\n", m2 * 1e6 / iters, m2 * 1e6 / iters * 2.4 / 22);
  printf("  mode 3 (mode 1 with the mask in an SGPR):                 %.1f ns  = %.1f cycles/instr\n", m3 * 1e6 / iters, m3 * 1e6 / iters * 2.4 / 22);
  printf("  mode 4 (22 x v_and_b32 with a 32-bit literal, + 22 v_or):  %.1f ns\n", m4 * 1e6 / iters);
  printf("  mode 5 (22 x v_cvt_pk_bf16_f32, + 22 v_or):                %.1f ns\n", m5 * 1e6 / iters);
  printf("  mode 6 (22 x v_lshlrev_b32, + 22 v_or):                    %.1f ns\n", m6 * 1e6 / iters);
  return 0;
}
