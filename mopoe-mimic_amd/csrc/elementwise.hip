// HBM-bound glue kernels of the residual blocks: every tensor is a row-major [rows][C] matrix; a thread
// owns a fixed group of VEC consecutive channels (so per-channel BatchNorm coefficients live in
// registers) and walks rows with a grid stride; loads/stores are 16 bytes per lane when C % 4 == 0.
// Column reductions (BN statistics, BN-backward sums, bias gradients) are reduced per block through
// LDS and leave the block as one atomic per column.
#include <type_traits>

#include "ew_common.hpp"

namespace mopoe {

// ---- out = a*bn(s) + b*m  (+ stats of out) -----------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(EW_THREADS) void block_out_fwd_kernel(const T* s, const T* m, T* out, long rows,
                                                                 int C, mopoe_bn_ref bn, float a, float b, double* stats) {
  const ColLayout L(C, VEC);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float sc[VEC], sh[VEC], part[2][VEC];
    for (int e = 0; e < VEC; ++e) {
      sc[e] = sh[e] = 0.f; part[0][e] = part[1][e] = 0.f;
      const int c = cv * VEC + e;
      if (active && c < C) { const BnC k = bn_coef(bn, c); sc[e] = a * k.scale; sh[e] = a * k.shift; }
    }
    if (active) {
      for (long r = (long)blockIdx.x * L.rpp + L.tr; r < rows; r += (long)gridDim.x * L.rpp) {
        const long off = r * C + (long)cv * VEC;
        const VecT<T, VEC> vs = VecT<T, VEC>::ld(s + off), vm = VecT<T, VEC>::ld(m + off);
        VecT<T, VEC> o;
        for (int e = 0; e < VEC; ++e) {
          o.v[e] = stored<T>(fmaf(vs.v[e], sc[e], sh[e]) + b * vm.v[e]);
          part[0][e] += o.v[e];
          part[1][e] += o.v[e] * o.v[e];
        }
        o.st(out + off);
      }
    }
    if (stats) {
      double* const od[2] = {stats, stats + C};
      float* const of[2] = {nullptr, nullptr};
      block_col_reduce<VEC, 2>(L, active, cbase, C, part, od, of);
    }
  }
}

// ---- sums += {sum g, sum g*shat} -----------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_reduce_kernel(const T* g, const T* s, long rows, int C,
                                                                 mopoe_bn_ref bn, double* sums) {
  const ColLayout L(C, VEC);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float mean[VEC], rstd[VEC], part[2][VEC];
    for (int e = 0; e < VEC; ++e) {
      mean[e] = rstd[e] = 0.f; part[0][e] = part[1][e] = 0.f;
      const int c = cv * VEC + e;
      if (active && c < C) { const BnC k = bn_coef(bn, c); mean[e] = k.mean; rstd[e] = k.rstd; }
    }
    if (active) {
      for (long r = (long)blockIdx.x * L.rpp + L.tr; r < rows; r += (long)gridDim.x * L.rpp) {
        const long off = r * C + (long)cv * VEC;
        const VecT<T, VEC> vg = VecT<T, VEC>::ld(g + off), vs = VecT<T, VEC>::ld(s + off);
        for (int e = 0; e < VEC; ++e) {
          part[0][e] += vg.v[e];
          part[1][e] += vg.v[e] * ((vs.v[e] - mean[e]) * rstd[e]);
        }
      }
    }
    double* const od[2] = {sums, sums + C};
    float* const of[2] = {nullptr, nullptr};
    block_col_reduce<VEC, 2>(L, active, cbase, C, part, od, of);
  }
}

// ---- dm = b*g*mask ; ds = a*BNbwd(g; s) ---------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(EW_THREADS) void block_out_bwd_kernel(const T* g, const T* s, T* dm, T* ds,
                                                                 long rows, int C, mopoe_bn_ref bn, const double* sums,
                                                                 mopoe_mask_ref mask, float a, float b, float* dgamma,
                                                                 float* dbeta, float* colsum_dm, float* colsum_ds) {
  const ColLayout L(C, VEC);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float mean[VEC], rstd[VEC], gr[VEC], k1[VEC], k2[VEC], part[2][VEC];
    for (int e = 0; e < VEC; ++e) {
      mean[e] = rstd[e] = gr[e] = k1[e] = k2[e] = 0.f; part[0][e] = part[1][e] = 0.f;
      const int c = cv * VEC + e;
      if (active && c < C) {
        const BnC k = bn_coef(bn, c);
        mean[e] = k.mean; rstd[e] = k.rstd; gr[e] = a * k.scale;  // a * gamma * rstd
        if (bn.mode == 1) { k1[e] = (float)(sums[c] * bn.inv_count); k2[e] = (float)(sums[C + c] * bn.inv_count); }
        if (blockIdx.x == 0 && L.tr == 0) {
          dgamma[c] = a * (float)sums[C + c];
          dbeta[c] = a * (float)sums[c];
        }
      }
    }
    if (active) {
      for (long r = (long)blockIdx.x * L.rpp + L.tr; r < rows; r += (long)gridDim.x * L.rpp) {
        const long off = r * C + (long)cv * VEC;
        const VecT<T, VEC> vg = VecT<T, VEC>::ld(g + off), vs = VecT<T, VEC>::ld(s + off);
        VecT<T, VEC> om, os;
        const float* mrow = nullptr;
        if (mask.kind == 1) mrow = mask.mask + (long)((unsigned)r / (unsigned)mask.rows_per_sample) * C;
        else if (mask.kind == 2) mrow = mask.mask + r * C;
        for (int e = 0; e < VEC; ++e) {
          const float mk = mrow ? mrow[cv * VEC + e] : 1.f;
          om.v[e] = stored<T>(b * vg.v[e] * mk);
          const float shat = (vs.v[e] - mean[e]) * rstd[e];
          os.v[e] = stored<T>(gr[e] * (vg.v[e] - k1[e] - shat * k2[e]));
          part[0][e] += om.v[e];
          part[1][e] += os.v[e];
        }
        om.st(dm + off);
        os.st(ds + off);
      }
    }
    if (colsum_dm || colsum_ds) {
      double* const od[2] = {nullptr, nullptr};
      float* const of[2] = {colsum_dm, colsum_ds};
      block_col_reduce<VEC, 2>(L, active, cbase, C, part, od, of);
    }
  }
}

// ---- dx = mask * BNbwd(dy; x) + add --------------------------------------------------------------------
// NEXT: dx is the gradient entering the previous residual block, whose first backward step is the pair of column
// reductions {sum dx, sum dx * shat} over its shortcut output s (bn_bwd_reduce).  Producing them here saves that
// kernel and its re-read of dx.
template <typename T, int VEC, bool NEXT>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_apply_kernel(const T* dy, const T* x, const T* add,
                                                                T* dx, long rows, int C, mopoe_bn_ref bn,
                                                                const double* sums, mopoe_mask_ref mask, float* dgamma,
                                                                float* dbeta, float* colsum_dx, const T* next_s,
                                                                mopoe_bn_ref next_bn, double* next_sums) {
  constexpr int NACC = NEXT ? 3 : 1;
  const ColLayout L(C, VEC);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float mean[VEC], rstd[VEC], gr[VEC], k1[VEC], k2[VEC], nmean[VEC], nrstd[VEC], part[NACC][VEC];
    for (int e = 0; e < VEC; ++e) {
      mean[e] = rstd[e] = gr[e] = k1[e] = k2[e] = nmean[e] = nrstd[e] = 0.f;
      for (int k = 0; k < NACC; ++k) part[k][e] = 0.f;
      const int c = cv * VEC + e;
      if (active && c < C) {
        const BnC k = bn_coef(bn, c);
        mean[e] = k.mean; rstd[e] = k.rstd; gr[e] = k.scale;
        if (bn.mode == 1) { k1[e] = (float)(sums[c] * bn.inv_count); k2[e] = (float)(sums[C + c] * bn.inv_count); }
        if (NEXT) { const BnC kn = bn_coef(next_bn, c); nmean[e] = kn.mean; nrstd[e] = kn.rstd; }
        if (blockIdx.x == 0 && L.tr == 0) {
          dgamma[c] = (float)sums[C + c];
          dbeta[c] = (float)sums[c];
        }
      }
    }
    if (active) {
      for (long r = (long)blockIdx.x * L.rpp + L.tr; r < rows; r += (long)gridDim.x * L.rpp) {
        const long off = r * C + (long)cv * VEC;
        const VecT<T, VEC> vd = VecT<T, VEC>::ld(dy + off), vx = VecT<T, VEC>::ld(x + off);
        VecT<T, VEC> va, vs, o;
        if (add) va = VecT<T, VEC>::ld(add + off);
        if (NEXT) vs = VecT<T, VEC>::ld(next_s + off);
        const float* mrow = nullptr;
        if (mask.kind == 1) mrow = mask.mask + (long)((unsigned)r / (unsigned)mask.rows_per_sample) * C;
        else if (mask.kind == 2) mrow = mask.mask + r * C;
        for (int e = 0; e < VEC; ++e) {
          const float xhat = (vx.v[e] - mean[e]) * rstd[e];
          float v = gr[e] * (vd.v[e] - k1[e] - xhat * k2[e]);
          if (mrow) v *= mrow[cv * VEC + e];
          if (add) v += va.v[e];
          v = stored<T>(v);
          o.v[e] = v;
          part[0][e] += v;
          if constexpr (NEXT) { part[1][e] += v; part[2][e] += v * ((vs.v[e] - nmean[e]) * nrstd[e]); }
        }
        o.st(dx + off);
      }
    }
    if constexpr (NEXT) {
      double* const od[3] = {nullptr, next_sums, next_sums + C};
      float* const of[3] = {colsum_dx, nullptr, nullptr};
      block_col_reduce<VEC, 3>(L, active, cbase, C, part, od, of);
    } else {
      if (colsum_dx) {
        double* const od[1] = {nullptr};
        float* const of[1] = {colsum_dx};
        block_col_reduce<VEC, 1>(L, active, cbase, C, part, od, of);
      }
    }
  }
}

// ---- column sums -----------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(EW_THREADS) void colsum_kernel(const T* x, float* out, long rows, int C) {
  const ColLayout L(C, VEC);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float part[1][VEC];
    for (int e = 0; e < VEC; ++e) part[0][e] = 0.f;
    if (active) {
      for (long r = (long)blockIdx.x * L.rpp + L.tr; r < rows; r += (long)gridDim.x * L.rpp) {
        const VecT<T, VEC> v = VecT<T, VEC>::ld(x + r * C + (long)cv * VEC);
        for (int e = 0; e < VEC; ++e) part[0][e] += v.v[e];
      }
    }
    double* const od[1] = {nullptr};
    float* const of[1] = {out};
    block_col_reduce<VEC, 1>(L, active, cbase, C, part, od, of);
  }
}

// ---- running statistics ------------------------------------------------------------------------------------
// the records travel in the kernel arguments (no device-side table: nothing to copy, nothing to keep alive, and the
// launch can sit inside a hipGraph capture)
constexpr int RUN_DESC_PER_LAUNCH = 32;
struct RunDescPack { mopoe_bn_running_desc d[RUN_DESC_PER_LAUNCH]; };

__global__ void bn_running_kernel(const RunDescPack pack, float momentum) {
  const mopoe_bn_running_desc d = pack.d[blockIdx.x];
  const double inv = 1.0 / (double)d.count;
  const double unb = d.count > 1 ? (double)d.count / (double)(d.count - 1) : 1.0;
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const double m = d.sums[c] * inv;
    double v = d.sums[d.C + c] * inv - m * m;
    v = v < 0.0 ? 0.0 : v;
    d.rmean[c] = (1.f - momentum) * d.rmean[c] + momentum * (float)m;
    d.rvar[c] = (1.f - momentum) * d.rvar[c] + momentum * (float)(v * unb);
  }
}

}  // namespace mopoe

using namespace mopoe;

#define EW_ARGCHECK(cond, msg) \
  if (!(cond)) { set_error(msg); return MOPOE_ERR_ARG; }

// ---- typed launchers: T = float (VEC 4, or 1 for ragged channel counts) or bf16_t (VEC 8; C % 8 == 0 required) -------
template <typename T> struct EwVec { static constexpr int wide = 4; };
template <> struct EwVec<bf16_t> { static constexpr int wide = 8; };

template <typename T>
static bool ew_wide_ok(int C, std::initializer_list<const void*> ptrs, const char* what, int* rc) {
  const bool ok = vec_ok(C, ptrs, EwVec<T>::wide);
  *rc = MOPOE_OK;
  if (!ok && std::is_same<T, bf16_t>::value) {
    set_error("%s (bf16): needs C %% 8 == 0 (C = %d) and 16-byte aligned tensors", what, C);
    *rc = MOPOE_ERR_ARG;
  }
  return ok;
}

template <typename T>
static int block_out_fwd_t(const T* s, const T* m, T* out, int64_t rows, int32_t C, const mopoe_bn_ref* bn_s, float a,
                           float b, double* out_stats, void* stream) {
  EW_ARGCHECK(s && m && out && bn_s && bn_s->mode != 0 && bn_s->C == C && rows > 0, "block_out_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  constexpr int W = EwVec<T>::wide;
  if (ew_wide_ok<T>(C, {s, m, out}, "block_out_fwd", &rc))
    hipLaunchKernelGGL((block_out_fwd_kernel<T, W>), dim3(ew_grid(rows, C, W)), dim3(EW_THREADS), 0, st, s, m, out, (long)rows, C, *bn_s, a, b, out_stats);
  else if constexpr (std::is_same<T, float>::value)
    hipLaunchKernelGGL((block_out_fwd_kernel<float, 1>), dim3(ew_grid(rows, C, 1)), dim3(EW_THREADS), 0, st, s, m, out, (long)rows, C, *bn_s, a, b, out_stats);
  else return rc;
  return check_launch("block_out_fwd");
}

template <typename T>
static int bn_bwd_reduce_t(const T* g, const T* s, int64_t rows, int32_t C, const mopoe_bn_ref* bn_s, double* sums,
                           void* stream) {
  EW_ARGCHECK(g && s && sums && bn_s && bn_s->mode != 0 && bn_s->C == C && rows > 0, "bn_bwd_reduce: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  constexpr int W = EwVec<T>::wide;
  if (ew_wide_ok<T>(C, {g, s}, "bn_bwd_reduce", &rc))
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, W>), dim3(ew_grid(rows, C, W)), dim3(EW_THREADS), 0, st, g, s, (long)rows, C, *bn_s, sums);
  else if constexpr (std::is_same<T, float>::value)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 1>), dim3(ew_grid(rows, C, 1)), dim3(EW_THREADS), 0, st, g, s, (long)rows, C, *bn_s, sums);
  else return rc;
  return check_launch("bn_bwd_reduce");
}

template <typename T>
static int block_out_bwd_t(const T* g, const T* s, T* dm, T* ds, int64_t rows, int32_t C, const mopoe_bn_ref* bn_s,
                           const double* sums, const mopoe_mask_ref* mask, float a, float b, float* dgamma, float* dbeta,
                           float* colsum_dm, float* colsum_ds, void* stream) {
  EW_ARGCHECK(g && s && dm && ds && sums && dgamma && dbeta && bn_s && bn_s->mode != 0 && bn_s->C == C && rows > 0,
              "block_out_bwd: bad arguments");
  mopoe_mask_ref mk = mask ? *mask : mopoe_mask_ref{nullptr, 0, 1};
  hipStream_t st = (hipStream_t)stream;
  int rc;
  constexpr int W = EwVec<T>::wide;
  if (ew_wide_ok<T>(C, {g, s, dm, ds}, "block_out_bwd", &rc))
    hipLaunchKernelGGL((block_out_bwd_kernel<T, W>), dim3(ew_grid(rows, C, W)), dim3(EW_THREADS), 0, st, g, s, dm, ds, (long)rows, C, *bn_s, sums, mk, a, b, dgamma, dbeta, colsum_dm, colsum_ds);
  else if constexpr (std::is_same<T, float>::value)
    hipLaunchKernelGGL((block_out_bwd_kernel<float, 1>), dim3(ew_grid(rows, C, 1)), dim3(EW_THREADS), 0, st, g, s, dm, ds, (long)rows, C, *bn_s, sums, mk, a, b, dgamma, dbeta, colsum_dm, colsum_ds);
  else return rc;
  return check_launch("block_out_bwd");
}

template <typename T>
static int bn_bwd_apply_t(const T* dy, const T* x, const T* add, T* dx, int64_t rows, int32_t C, const mopoe_bn_ref* bn,
                          const double* sums, const mopoe_mask_ref* mask, float* dgamma, float* dbeta, float* colsum_dx,
                          const T* next_s, const mopoe_bn_ref* next_bn, double* next_sums, void* stream) {
  EW_ARGCHECK(dy && x && dx && sums && dgamma && dbeta && bn && bn->mode != 0 && bn->C == C && rows > 0,
              "bn_bwd_apply: bad arguments");
  const bool next = next_s != nullptr;
  EW_ARGCHECK(!next || (next_bn && next_sums && next_bn->mode != 0 && next_bn->C == C),
              "bn_bwd_apply: next_s needs next_bn (C channels) and next_sums");
  mopoe_mask_ref mk = mask ? *mask : mopoe_mask_ref{nullptr, 0, 1};
  const mopoe_bn_ref nb = next ? *next_bn : mopoe_bn_ref{};
  hipStream_t st = (hipStream_t)stream;
  int rc;
  constexpr int W = EwVec<T>::wide;
  const bool vec = ew_wide_ok<T>(C, {dy, x, add, dx, next_s}, "bn_bwd_apply", &rc);
  if (!vec && !std::is_same<T, float>::value) return rc;
  const dim3 grid(ew_grid(rows, C, vec ? W : 1)), blk(EW_THREADS);
#define MOPOE_APPLY(V_, N_) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, V_, N_>), grid, blk, 0, st, dy, x, add, dx, (long)rows, C, *bn, sums, mk, dgamma, dbeta, colsum_dx, next_s, nb, next_sums)
  if (vec) { if (next) MOPOE_APPLY(W, true); else MOPOE_APPLY(W, false); }
  else if constexpr (std::is_same<T, float>::value) { if (next) MOPOE_APPLY(1, true); else MOPOE_APPLY(1, false); }
#undef MOPOE_APPLY
  return check_launch("bn_bwd_apply");
}

template <typename T>
static int colsum_t(const T* x, float* out, int64_t rows, int32_t C, int32_t out_is_zero, void* stream) {
  EW_ARGCHECK(x && out && rows > 0 && C > 0, "colsum: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (!out_is_zero && hipMemsetAsync(out, 0, sizeof(float) * C, st) != hipSuccess) { set_error("colsum memset failed"); return MOPOE_ERR_LAUNCH; }
  int rc;
  constexpr int W = EwVec<T>::wide;
  if (ew_wide_ok<T>(C, {x}, "colsum", &rc))
    hipLaunchKernelGGL((colsum_kernel<T, W>), dim3(ew_grid(rows, C, W)), dim3(EW_THREADS), 0, st, x, out, (long)rows, C);
  else if constexpr (std::is_same<T, float>::value)
    hipLaunchKernelGGL((colsum_kernel<float, 1>), dim3(ew_grid(rows, C, 1)), dim3(EW_THREADS), 0, st, x, out, (long)rows, C);
  else return rc;
  return check_launch("colsum");
}

extern "C" int mopoe_block_out_fwd(const float* s, const float* m, float* out, int64_t rows, int32_t C,
                                   const mopoe_bn_ref* bn_s, float a, float b, double* out_stats, void* stream) {
  return block_out_fwd_t<float>(s, m, out, rows, C, bn_s, a, b, out_stats, stream);
}
extern "C" int mopoe_block_out_fwd_bf16(const uint16_t* s, const uint16_t* m, uint16_t* out, int64_t rows, int32_t C,
                                        const mopoe_bn_ref* bn_s, float a, float b, double* out_stats, void* stream) {
  return block_out_fwd_t<bf16_t>(s, m, out, rows, C, bn_s, a, b, out_stats, stream);
}

extern "C" int mopoe_bn_bwd_reduce(const float* g, const float* s, int64_t rows, int32_t C, const mopoe_bn_ref* bn_s,
                                   double* sums, void* stream) {
  return bn_bwd_reduce_t<float>(g, s, rows, C, bn_s, sums, stream);
}
extern "C" int mopoe_bn_bwd_reduce_bf16(const uint16_t* g, const uint16_t* s, int64_t rows, int32_t C,
                                        const mopoe_bn_ref* bn_s, double* sums, void* stream) {
  return bn_bwd_reduce_t<bf16_t>(g, s, rows, C, bn_s, sums, stream);
}

extern "C" int mopoe_block_out_bwd(const float* g, const float* s, float* dm, float* ds, int64_t rows, int32_t C,
                                   const mopoe_bn_ref* bn_s, const double* sums, const mopoe_mask_ref* mask, float a,
                                   float b, float* dgamma, float* dbeta, float* colsum_dm, float* colsum_ds,
                                   void* stream) {
  return block_out_bwd_t<float>(g, s, dm, ds, rows, C, bn_s, sums, mask, a, b, dgamma, dbeta, colsum_dm, colsum_ds, stream);
}
extern "C" int mopoe_block_out_bwd_bf16(const uint16_t* g, const uint16_t* s, uint16_t* dm, uint16_t* ds, int64_t rows,
                                        int32_t C, const mopoe_bn_ref* bn_s, const double* sums,
                                        const mopoe_mask_ref* mask, float a, float b, float* dgamma, float* dbeta,
                                        float* colsum_dm, float* colsum_ds, void* stream) {
  return block_out_bwd_t<bf16_t>(g, s, dm, ds, rows, C, bn_s, sums, mask, a, b, dgamma, dbeta, colsum_dm, colsum_ds, stream);
}

extern "C" int mopoe_bn_bwd_apply(const float* dy, const float* x, const float* add, float* dx, int64_t rows, int32_t C,
                                  const mopoe_bn_ref* bn, const double* sums, const mopoe_mask_ref* mask, float* dgamma,
                                  float* dbeta, float* colsum_dx, const float* next_s, const mopoe_bn_ref* next_bn,
                                  double* next_sums, void* stream) {
  return bn_bwd_apply_t<float>(dy, x, add, dx, rows, C, bn, sums, mask, dgamma, dbeta, colsum_dx, next_s, next_bn, next_sums, stream);
}
extern "C" int mopoe_bn_bwd_apply_bf16(const uint16_t* dy, const uint16_t* x, const uint16_t* add, uint16_t* dx,
                                       int64_t rows, int32_t C, const mopoe_bn_ref* bn, const double* sums,
                                       const mopoe_mask_ref* mask, float* dgamma, float* dbeta, float* colsum_dx,
                                       const uint16_t* next_s, const mopoe_bn_ref* next_bn, double* next_sums,
                                       void* stream) {
  return bn_bwd_apply_t<bf16_t>(dy, x, add, dx, rows, C, bn, sums, mask, dgamma, dbeta, colsum_dx, next_s, next_bn, next_sums, stream);
}

extern "C" int mopoe_bn_running_update(const mopoe_bn_running_desc* desc, int32_t n, float momentum, void* stream) {
  EW_ARGCHECK(desc && n > 0, "bn_running_update: bad arguments");
  for (int32_t base = 0; base < n; base += RUN_DESC_PER_LAUNCH) {
    const int32_t m = n - base < RUN_DESC_PER_LAUNCH ? n - base : RUN_DESC_PER_LAUNCH;
    RunDescPack pack = {};
    for (int32_t i = 0; i < m; ++i) {
      pack.d[i] = desc[base + i];
      EW_ARGCHECK(pack.d[i].sums && pack.d[i].rmean && pack.d[i].rvar && pack.d[i].C > 0 && pack.d[i].count > 0,
                  "bn_running_update: bad record");
    }
    hipLaunchKernelGGL(bn_running_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, pack, momentum);
    if (int rc = check_launch("bn_running_update")) return rc;
  }
  return MOPOE_OK;
}

extern "C" int mopoe_colsum(const float* x, float* out, int64_t rows, int32_t C, int32_t out_is_zero, void* stream) {
  return colsum_t<float>(x, out, rows, C, out_is_zero, stream);
}
extern "C" int mopoe_colsum_bf16(const uint16_t* x, float* out, int64_t rows, int32_t C, int32_t out_is_zero, void* stream) {
  return colsum_t<bf16_t>(x, out, rows, C, out_is_zero, stream);
}
