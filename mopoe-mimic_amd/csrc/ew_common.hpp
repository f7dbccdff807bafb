// Shared pieces of the row-streaming kernels: vector load/store, column layout, per-block column reduction.
#pragma once
#include <stdlib.h>
#include <initializer_list>
#include "common.hpp"

namespace mopoe {

constexpr int EW_THREADS = 256;
constexpr int EW_MAX_BLOCKS = 512;

// VEC consecutive channels of a row as floats, whatever the storage type T (float or bf16_t) is
template <typename T, int VEC>
struct VecT;
template <>
struct VecT<float, 4> {
  float v[4];
  typedef float4 Raw;     // what a load leaves in registers until the row is computed (ldr / un)
  __device__ static Raw ldr(const float* p) { return *reinterpret_cast<const float4*>(p); }
  __device__ static VecT un(const Raw& t) { VecT r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r; }
  __device__ static VecT ld(const float* p) { return un(ldr(p)); }
  __device__ void st(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <>
struct VecT<float, 1> {
  float v[1];
  typedef float Raw;
  __device__ static Raw ldr(const float* p) { return *p; }
  __device__ static VecT un(const Raw& t) { VecT r; r.v[0] = t; return r; }
  __device__ static VecT ld(const float* p) { VecT r; r.v[0] = *p; return r; }
  __device__ void st(float* p) const { *p = v[0]; }
};
template <>
struct VecT<bf16_t, 8> {   // 16 bytes per lane
  float v[8];
  typedef uint4 Raw;
  __device__ static Raw ldr(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static VecT ld(const bf16_t* p) { return un(ldr(p)); }
  __device__ static VecT un(const Raw& t) {
    VecT r;
    r.v[0] = bf16_lo(t.x); r.v[1] = bf16_hi(t.x); r.v[2] = bf16_lo(t.y); r.v[3] = bf16_hi(t.y);
    r.v[4] = bf16_lo(t.z); r.v[5] = bf16_hi(t.z); r.v[6] = bf16_lo(t.w); r.v[7] = bf16_hi(t.w);
    return r;
  }
  __device__ void st(bf16_t* p) const {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
  }
};
template <>
struct VecT<bf16_t, 4> {   // 8 bytes per lane (edge kernels: 16 lanes x 4 channels per pixel)
  float v[4];
  typedef uint2 Raw;
  __device__ static Raw ldr(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  __device__ static VecT un(const Raw& t) {
    VecT r; r.v[0] = bf16_lo(t.x); r.v[1] = bf16_hi(t.x); r.v[2] = bf16_lo(t.y); r.v[3] = bf16_hi(t.y); return r;
  }
  __device__ static VecT ld(const bf16_t* p) { return un(ldr(p)); }
  __device__ void st(bf16_t* p) const { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])); }
};
template <int VEC>
using Vec = VecT<float, VEC>;
// what a value reads back as after it has been stored (statistics are taken over the STORED tensor)
template <typename T> __device__ __forceinline__ float stored(float f);
template <> __device__ __forceinline__ float stored<float>(float f) { return f; }
template <> __device__ __forceinline__ float stored<bf16_t>(float f) { return round_bf16(f); }

// column layout of a block: `cols` vector-columns per pass, `rpp` rows per pass
struct ColLayout {
  int Cv, cols, rpp, tc, tr;
  __device__ ColLayout(int C, int VEC) {
    Cv = (C + VEC - 1) / VEC;
    cols = Cv < EW_THREADS ? Cv : EW_THREADS;
    rpp = EW_THREADS / cols;
    tc = threadIdx.x % cols;
    tr = threadIdx.x / cols;
  }
};

// reduce NACC per-thread partials (per channel of the thread's vector) over the block's row dimension
// and add them to outd[k][channel] (double atomics) / outf[k][channel] (float atomics).  The final adds are
// issued by consecutive threads on consecutive addresses (one wave instruction covers 64 channels).
template <int VEC, int NACC>
__device__ void block_col_reduce(const ColLayout& L, bool active, int cbase, int C, float (&part)[NACC][VEC],
                                 double* const (&outd)[NACC], float* const (&outf)[NACC]) {
  __shared__ float red[EW_THREADS * NACC * VEC];
  __shared__ float tot[NACC][EW_THREADS * VEC];
  __syncthreads();
  for (int k = 0; k < NACC; ++k)
    for (int e = 0; e < VEC; ++e) red[(threadIdx.x * NACC + k) * VEC + e] = active ? part[k][e] : 0.f;
  __syncthreads();
  // thread t sums column (t % ncol) of accumulator (t / ncol) over the rpp row-lanes
  const int ncol = L.cols * VEC;  // channels covered by this pass
  for (int idx = threadIdx.x; idx < ncol * NACC; idx += EW_THREADS) {
    const int k = idx / ncol, cc = idx - k * ncol;
    const int tc = cc / VEC, e = cc - tc * VEC;
    float s = 0.f;
    for (int r = 0; r < L.rpp; ++r) s += red[((r * L.cols + tc) * NACC + k) * VEC + e];
    tot[k][cc] = s;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < ncol * NACC; idx += EW_THREADS) {
    const int k = idx / ncol, cc = idx - k * ncol;
    const int c = cbase * VEC + cc;
    if (c < C) {
      if (outd[k]) atomic_add_f64(outd[k] + c, (double)tot[k][cc]);
      if (outf[k]) unsafeAtomicAdd(outf[k] + c, tot[k][cc]);
    }
  }
}


// grid of a row-streaming kernel: at least EW_ROWS_PER_THREAD rows per thread (the per-block column reduction and its
// atomics amortise over them), at most EW_MAX_BLOCKS blocks.  Both are tuning knobs (env MOPOE_EW_MAX_BLOCKS /
// MOPOE_EW_ROWS_PER_THREAD, read once): tests/tools/glue_time.py sweeps them.
static inline int ew_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && atoi(v) > 0 ? atoi(v) : dflt;
}
static inline int ew_grid(long rows, int C, int VEC) {
  // Block count of a row-streaming kernel.  Every block ends in one atomic per column and accumulator on the SAME addresses,
  // and same-address atomics serialise at the memory side at ~18 ns each (measured: +9.2 us per 512 additional blocks,
  // profiles/r03_glue_sweep.txt): a tail of blocks x 18 ns that only the largest tensors amortise.  One block per CU (256,
  // four rows in flight per thread) streams tensors of up to ~100 MB per stream at the same rate as two per CU; beyond that
  // the second block per CU is worth more than its share of the tail.
  static const int max_blocks_env = ew_env("MOPOE_EW_MAX_BLOCKS", 0);
  static const int rows_per_thread = ew_env("MOPOE_EW_ROWS_PER_THREAD", 8);
  const int max_blocks = max_blocks_env > 0 ? max_blocks_env : ((double)rows * C * (VEC == 8 ? 2.0 : 4.0) > 96e6 ? EW_MAX_BLOCKS : 256);
  const int Cv = (C + VEC - 1) / VEC;
  const int cols = Cv < EW_THREADS ? Cv : EW_THREADS;
  const int rpp = EW_THREADS / cols;
  long blocks = (rows + rpp - 1) / rpp;
  blocks = (blocks + rows_per_thread - 1) / rows_per_thread;
  if (blocks < 1) blocks = 1;
  if (blocks > max_blocks) blocks = max_blocks;
  return (int)blocks;
}

static inline bool vec_ok(int C, std::initializer_list<const void*> ptrs, int vec = 4) {
  if (C % vec != 0) return false;
  for (const void* p : ptrs)
    if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return false;
  return true;
}

}  // namespace mopoe
