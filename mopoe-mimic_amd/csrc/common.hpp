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
  if (b.mode == 1) {
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
  return r;
}

__device__ __forceinline__ float mask_at(const mopoe_mask_ref& m, long row, int c, int C) {
  if (m.kind == 0) return 1.0f;
  if (m.kind == 1) return m.mask[(row / m.rows_per_sample) * (long)C + c];
  return m.mask[row * (long)C + c];
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

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
enum { PROF_GATHER_VEC = 0, PROF_GATHER_SCALAR = 32, PROF_WGRAD_VEC = 36, PROF_WGRAD_SCALAR = 42, PROF_DIRECT = 44, PROF_NKINDS = MOPOE_PROF_KINDS };
struct ProfScope {
  hipStream_t stream;
  int slot;
  ProfScope(hipStream_t s, double flops, int kind);
  ~ProfScope();
};

}  // namespace mopoe
