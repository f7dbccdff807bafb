// Pieces shared by the implicit-GEMM kernel families (conv_gemm.hip: fp32 MFMA; conv_gemm_bf16.hip: bf16 MFMA).
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <initializer_list>
#include <type_traits>

#include "common.hpp"

namespace mopoe {

// edge.hip: streaming kernels for the single-channel image-side layers (T = storage type of the wide tensor)
bool edge_supported(const mopoe_conv_geom* g, int C, std::initializer_list<const void*> ptrs);
template <typename T> int edge_expand(const float* scal, const float* W, T* out, const mopoe_conv_geom* g, int C, double* stats, hipStream_t st);
template <typename T> int edge_wgrad(const T* vec, const float* scal, float* dW, const mopoe_conv_geom* g, int C, hipStream_t st, bool dw_is_zero = false);
template <typename T> int edge_reduce(const T* x, const float* W, const float* bias, float* out, const mopoe_conv_geom* g, int C, hipStream_t st);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_BN_C = 1024;
// voffset past every buffer (operands are < 2 GiB, checked on the host): the hardware range check returns zeros.
// 2^31 rather than ~0 so that voffset + soffset cannot wrap whichever of the two the range check includes.
constexpr unsigned OOB = 0x80000000u;
constexpr size_t WS_RECOMMENDED = 64u << 20;
// head of the conv workspace: one int arrival counter per output tile of a split reduction (zero between launches)
constexpr size_t WS_COUNTER_BYTES = 64u << 10;

// ---- which taps a block multiplies, and from where --------------------------------------------------------------
// form 0: every tap, source pixel = output pixel * stride - pad + tap.  form 1 (sub-pixel phase `phase` of the big
// grid): only the taps congruent to the phase, source pixel = (output pixel + pad - tap) / stride, walked backwards.
struct TapWalk {
  int nty, ntx;          // taps of this phase along y / x
  int ky0, kx0;          // first tap
  int kstep_y, kstep_x;  // tap stride
  int dsgn;              // +1: source moves forward with the tap, -1: backward
  int cy, cx;            // source offset of the first tap (form 1)
  int phy, phx;          // the phase (form 1)
};
template <typename Args>
__device__ __forceinline__ TapWalk tap_walk(const Args& a, int phase) {
  TapWalk w;
  w.cy = w.cx = w.phy = w.phx = 0;
  if (a.form == 0) {
    w.nty = a.kh; w.ntx = a.kw; w.ky0 = 0; w.kx0 = 0; w.kstep_y = 1; w.kstep_x = 1; w.dsgn = 1;
  } else {
    w.phy = phase / a.sw; w.phx = phase % a.sw;
    const int ry = (w.phy + a.ph) % a.sh, rx = (w.phx + a.pw) % a.sw;
    w.nty = ry < a.kh ? (a.kh - ry + a.sh - 1) / a.sh : 0;
    w.ntx = rx < a.kw ? (a.kw - rx + a.sw - 1) / a.sw : 0;
    w.ky0 = ry; w.kx0 = rx; w.kstep_y = a.sh; w.kstep_x = a.sw; w.dsgn = -1;
    w.cy = (w.phy + a.ph - ry) / a.sh; w.cx = (w.phx + a.pw - rx) / a.sw;
  }
  return w;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline int validate_geom(const mopoe_conv_geom* g) {
  if (!g || g->N <= 0 || g->Hs <= 0 || g->Ws <= 0 || g->Hb <= 0 || g->Wb <= 0 || g->Cin <= 0 || g->Cout <= 0 ||
      g->kh <= 0 || g->kw <= 0 || g->sh <= 0 || g->sw <= 0 || g->ph < 0 || g->pw < 0) {
    set_error("conv geometry: non-positive field");
    return MOPOE_ERR_ARG;
  }
  if (g->Hb % g->sh != 0 || g->Wb % g->sw != 0) {
    set_error("conv geometry: big grid (%d,%d) must be a multiple of the stride (%d,%d)", g->Hb, g->Wb, g->sh, g->sw);
    return MOPOE_ERR_ARG;
  }
  // every small-grid pixel must map inside the padded big grid
  if ((g->Hs - 1) * g->sh - g->ph + g->kh - 1 >= g->Hb + g->ph + g->sh || (g->Ws - 1) * g->sw - g->pw + g->kw - 1 >= g->Wb + g->pw + g->sw) {
    set_error("conv geometry: small grid does not fit the big grid");
    return MOPOE_ERR_ARG;
  }
  return 0;
}


// XCD-aware block numbering.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one L2:
// MI355X_MICROARCH.md, Workgroup dispatch): the logical id returned here gives each XCD a CONTIGUOUS range of the grid,
// so blocks that read the same operand panels (all taps / channel tiles of one pixel chunk of a weight gradient) fill
// one L2 instead of eight.  Bijective for any grid size (cdna_hip_programming.md T1); a pure speed choice.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned orig, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u;
  return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (orig >> 3);
}

}  // namespace mopoe
