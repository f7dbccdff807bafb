// Shared device/host helpers for libmopoe_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mopoe_hip.h"

#define MOPOE_WAVE 64

namespace mopoe {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---- BatchNorm coefficients from a mopoe_bn_ref --------------------------------------------------
struct BnC {
  float mean, rstd, scale, shift;
};

__device__ __forceinline__ BnC bn_coef(const mopoe_bn_ref& b, int c) {
  BnC r;
#ifdef MOPOE_DBG_CONST_COEF   // timing experiment only (A/B library, wrong results): what the coefficient prologues cost
  r.mean = 0.f; r.rstd = 1.f; r.scale = 1.f; r.shift = 0.f;
  return r;
#endif
  if (b.mode == 1 || b.mode == 3) {
    const double m = b.sums[c] * b.inv_count;
    double v = b.sums[b.C + c] * b.inv_count - m * m;
    v = v < 0.0 ? 0.0 : v;
    r.mean = (float)m;
    r.rstd = (float)(1.0 / sqrt(v + (double)b.eps));
  } else {
    r.mean = b.rmean[c];
    r.rstd = 1.0f / sqrtf(b.rvar[c] + b.eps);
  }
  const float g = b.gamma[c];
  r.scale = g * r.rstd;
  r.shift = b.beta[c] - r.mean * r.scale;
  if (b.mode == 3) {
    // the tensor that will be read is the ACTIVATION y = relu(scale * x + shift) of this (batch-statistics) BatchNorm, not x
    // (conv2's input gradient when the block front is recomputed, pointwise.hip): [y > 0] is the ReLU mask and
    // xhat = (y - beta) / gamma wherever the mask is set -- the only places the masked gradient looks at xhat
    BnC y;
    y.mean = b.beta[c];
    y.rstd = g != 0.f ? 1.0f / g : 0.f;
    y.scale = 1.f;
    y.shift = 0.f;
    return y;
  }
  return r;
}

__device__ __forceinline__ float mask_at(const mopoe_mask_ref& m, long row, int c, int C) {
  if (m.kind == 0) return 1.0f;
  if (m.kind == 1) return m.mask[(row / m.rows_per_sample) * (long)C + c];
  return m.mask[row * (long)C + c];
}

#ifdef MOPOE_DBG_NO_STAT_ATOMICS   // timing experiment only (A/B library; statistics are then wrong): DESIGN section 4, glue kernels
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { if (v == 1.2345e300) *p = v; }
#else
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
#endif

// ---- bf16 storage (BASELINE configs #3, #5): activations / activation gradients / MFMA operands are bfloat16 in HBM,
// every sum stays fp32 (statistics fp64).  A bf16 value is the upper half of the fp32 with the same value; rounding to
// bf16 is round-to-nearest-even (the compiler's __bf16 conversion: v_cvt_pk_bf16_f32, NaN stays NaN).
typedef unsigned short bf16_t;     // storage type at the C ABI (a plain 16-bit pattern)
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 f = {lo, hi};      // converted as a vector: ONE v_cvt_pk_bf16_f32 (two scalar conversions cost two + a v_perm)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16(f, 0.f) & 0xffffu); }
// the value a float has after a round trip through bf16 storage
__device__ __forceinline__ float round_bf16(float f) { return bf16_lo(pack_bf16(f, 0.f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- profiling (api.hip) ----------------------------------------------------------------------------
// one kind per kernel instantiation, so that the names bench.py reports are the names rocprofv3 prints:
//   gather, vector path:  tile * 4 + spec            (tile 0..7, spec 0..3)        kinds  0..31
//   gather, scalar path:  32 + {0: 128x128, 1: 256x64, 2: 64x64}                   kinds 32..34
//   wgrad,  vector path:  36 + (128x128 ? 0 : 3) + spec   (spec 0..2)              kinds 36..41
//   wgrad,  scalar path:  42 + (128x128 ? 0 : 1)                                   kinds 42..43
//   direct (LDS-free) gather: 44 + (tile - 8) * 4 + spec                           kinds 44..59
//   bf16 gather:          60 + tile * 3 + (spec - 1)     (tile 0..4, spec 1..3)    kinds 60..74
//   bf16 wgrad:           75 + (128x128 ? 0 : 2) + (BN+ReLU on x ? 1 : 0)          kinds 75..78
//   bf16 gather, LDS-DMA: 80 + (tile - 5) * 2 + (input gradient ? 1 : 0)           kinds 80..93
//   the same with BN + ReLU on the operand (tiles 5, 7, 9): 94 + (tile - 5) / 2     kinds 94..96
//   bf16 wgrad, LDS-DMA:  100 + (128x128 ? 0 : 2) + (BN+ReLU on x ? 1 : 0)         kinds 100..103
//   fp32 gather, LDS-DMA: 104 + (tile - 12) * 3 + (spec - 1)                       kinds 104..115
//   fp32 wgrad, LDS-DMA:  116 + (128x128 ? 0 : 2) + (BN+ReLU on x ? 1 : 0)         kinds 116..119
//   fp32 gather, LDS-DMA, products on the bf16 pipe: 130 + (tile - 16) * 2 + (input gradient ? 1 : 0)   kinds 130..137
//   fp32 wgrad,  LDS-DMA, products on the bf16 pipe: 138 + (128x128 ? 0 : 1)       kinds 138..139
//   fp32 wgrad, four taps per block, products on the bf16 pipe (tile 9)            kinds 140..143: (S tile 128 ? 2 : 0) + (transposed conv ? 1 : 0)
enum { PROF_GATHER_VEC = 0, PROF_GATHER_SCALAR = 32, PROF_WGRAD_VEC = 36, PROF_WGRAD_SCALAR = 42, PROF_DIRECT = 44,
       PROF_BF16_GATHER = 60, PROF_BF16_WGRAD = 75, PROF_BF16_GLDS = 80, PROF_BF16_GLDS_X = 94, PROF_BF16_WGRAD_GLDS = 100, PROF_F32_GLDS = 104, PROF_F32_WGRAD_GLDS = 116, PROF_BF16_WGRAD_GLDS_MERGE = 120, PROF_BF16_WGRAD_PARITY = 122, PROF_PW_FRONT = 124, PROF_F32_GLDS_EMU = 130, PROF_F32_WGRAD_GLDS_EMU = 138, PROF_F32_WGRAD_PARITY = 140, PROF_NKINDS = MOPOE_PROF_KINDS };
struct ProfScope {
  hipStream_t stream;
  int slot;
  // flops / bytes: the launch's ALGORITHMIC work (one read of the input activation + one write of the
  // output per pass; bytes = 0 where not stated)
  ProfScope(hipStream_t s, double flops, int kind, double bytes = 0.0);
  ~ProfScope();
};

}  // namespace mopoe
