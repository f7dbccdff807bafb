// Shared pieces of the row-streaming kernels: vector load/store, column layout, per-block column reduction.
#pragma once
#include <initializer_list>
#include "common.hpp"

namespace mopoe {

constexpr int EW_THREADS = 256;
constexpr int EW_MAX_BLOCKS = 512;

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  float v[4];
  __device__ static Vec ld(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    Vec r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
  }
  __device__ void st(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <>
struct Vec<1> {
  float v[1];
  __device__ static Vec ld(const float* p) { Vec r; r.v[0] = *p; return r; }
  __device__ void st(float* p) const { *p = v[0]; }
};

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


static inline int ew_grid(long rows, int C, int VEC) {
  const int Cv = (C + VEC - 1) / VEC;
  const int cols = Cv < EW_THREADS ? Cv : EW_THREADS;
  const int rpp = EW_THREADS / cols;
  long blocks = (rows + rpp - 1) / rpp;
  // keep several rows per thread so the per-block reduction / atomics amortise
  blocks = (blocks + 7) / 8;
  if (blocks < 1) blocks = 1;
  if (blocks > EW_MAX_BLOCKS) blocks = EW_MAX_BLOCKS;
  return (int)blocks;
}

static inline bool vec_ok(int C, std::initializer_list<const void*> ptrs) {
  if (C % 4 != 0) return false;
  for (const void* p : ptrs)
    if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return false;
  return true;
}

}  // namespace mopoe
