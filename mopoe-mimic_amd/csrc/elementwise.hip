// HBM-bound glue kernels of the residual blocks: every tensor is a row-major [rows][C] matrix; a thread
// owns a fixed group of VEC consecutive channels (so per-channel BatchNorm coefficients live in
// registers) and walks rows with a grid stride; loads/stores are 16 bytes per lane when C % 4 == 0.
// Column reductions (BN statistics, BN-backward sums, bias gradients) are reduced per block through
// LDS and leave the block as one atomic per column.
#include <type_traits>

#include "ew_common.hpp"

namespace mopoe {

// The row loop of every kernel below: a block owns chunks of U * rpp consecutive rows; a thread first issues the loads of
// all its U rows of a chunk (U x streams 16-byte loads in flight per lane: HBM latency is hidden by bytes in flight, and at
// 8-16 waves per CU one row per thread leaves the chip at a third of its bandwidth), then computes and stores them.  Rows
// past the end are loaded from the last row (a valid address) and dropped.
template <int VEC>
__device__ __forceinline__ void ld_maskv(const float* p, float (&m)[VEC]) {
  if constexpr (VEC % 4 == 0) {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
      m[4 * q] = t.x; m[4 * q + 1] = t.y; m[4 * q + 2] = t.z; m[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = p[e];
  }
}
__device__ __forceinline__ const float* mask_row(const mopoe_mask_ref& mask, long r, int C) {
  if (mask.kind == 1) return mask.mask + (long)((unsigned)r / (unsigned)mask.rows_per_sample) * C;
  if (mask.kind == 2) return mask.mask + r * C;
  return nullptr;
}

// Per-channel BatchNorm coefficients, computed ONCE per block into LDS (each needs an fp64 division and square root, ~100
// instructions; per thread and channel that was up to 16 of them -- 3 us of a 10-us launch in the bf16 family).
// cf[0..5][c] = mean, rstd, scale (gamma * rstd), shift, k1 = sum0 / count, k2 = sum1 / count  (k1, k2 only with `sums`).
constexpr int EW_COEF_C = 640;   // channel counts beyond this take the per-thread path (no BASELINE shape does)
__device__ __forceinline__ void coef_one(const mopoe_bn_ref& bn, const double* sums, int c, int C, float (&o)[6]) {
  const BnC k = bn_coef(bn, c);
  o[0] = k.mean; o[1] = k.rstd; o[2] = k.scale; o[3] = k.shift; o[4] = o[5] = 0.f;
  if (sums && bn.mode == 1) { o[4] = (float)(sums[c] * bn.inv_count); o[5] = (float)(sums[C + c] * bn.inv_count); }
}
template <int NF>
__device__ __forceinline__ void coef_table(const mopoe_bn_ref& bn, const double* sums, int C, float (*cf)[EW_COEF_C]) {
  if (C > EW_COEF_C) return;
  for (int c = threadIdx.x; c < C; c += EW_THREADS) {
    float o[6];
    coef_one(bn, sums, c, C, o);
#pragma unroll
    for (int f = 0; f < NF; ++f) cf[f][c] = o[f];
  }
  __syncthreads();
}
// this thread's coefficients of channel c: from the table, or computed here when the table was not built
template <int NF>
__device__ __forceinline__ void coef_get(const mopoe_bn_ref& bn, const double* sums, int c, int C,
                                         const float (*cf)[EW_COEF_C], float (&o)[6]) {
  if (C <= EW_COEF_C) {
#pragma unroll
    for (int f = 0; f < 6; ++f) o[f] = f < NF ? cf[f][c] : 0.f;
  } else {
    coef_one(bn, sums, c, C, o);
  }
}

// ---- out = a*bn(s) + b*m  (+ stats of out) -----------------------------------------------------------
template <typename T, int VEC, int U>
__global__ __launch_bounds__(EW_THREADS) void block_out_fwd_kernel(const T* __restrict__ s, const T* __restrict__ m,
                                                                 T* __restrict__ out, long rows, int C, mopoe_bn_ref bn,
                                                                 float a, float b, double* stats) {
  const ColLayout L(C, VEC);
  __shared__ float cf[4][EW_COEF_C];
  coef_table<4>(bn, nullptr, C, cf);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float sc[VEC], sh[VEC], part[2][VEC];
    for (int e = 0; e < VEC; ++e) {
      sc[e] = sh[e] = 0.f; part[0][e] = part[1][e] = 0.f;
      const int c = cv * VEC + e;
      if (active && c < C) { float k[6]; coef_get<4>(bn, nullptr, c, C, cf, k); sc[e] = a * k[2]; sh[e] = a * k[3]; }
    }
    if (active) {
      const long step = (long)gridDim.x * L.rpp * U;
      for (long rb = (long)blockIdx.x * L.rpp * U + L.tr; rb < rows; rb += step) {
        typename VecT<T, VEC>::Raw vs[U], vm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          const long off = (r < rows ? r : rows - 1) * C + (long)cv * VEC;
          vs[u] = VecT<T, VEC>::ldr(s + off); vm[u] = VecT<T, VEC>::ldr(m + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          if (r >= rows) break;
          const VecT<T, VEC> ws = VecT<T, VEC>::un(vs[u]), wm = VecT<T, VEC>::un(vm[u]);
          VecT<T, VEC> o;
          for (int e = 0; e < VEC; ++e) {
            o.v[e] = stored<T>(fmaf(ws.v[e], sc[e], sh[e]) + b * wm.v[e]);
            part[0][e] += o.v[e];
            part[1][e] += o.v[e] * o.v[e];
          }
          o.st(out + r * C + (long)cv * VEC);
        }
      }
    }
    if (stats) {
      double* const od[2] = {stats, stats + C};
      float* const of[2] = {nullptr, nullptr};
      block_col_reduce<VEC, 2>(L, active, cbase, C, part, od, of);
    }
  }
}

// ---- out = relu(bn(x)) ------------------------------------------------------------------------------------
// The operand of a block's second conv, written out once where reading it through the conv's BN -> ReLU-on-load form costs
// more than one more pass over it (the up-sampling convs: every input element is gathered by 16 (tap, phase) pairs and K
// per tile is too short to hide the register-staged loads -- DESIGN section 4).  Same arithmetic and the same rounding point
// as the on-load transform: fmaf, max, one rounding when stored.
template <typename T, int VEC, int U>
__global__ __launch_bounds__(EW_THREADS) void bn_relu_apply_kernel(const T* __restrict__ x, T* __restrict__ out, long rows, int C,
                                                                 mopoe_bn_ref bn) {
  const ColLayout L(C, VEC);
  __shared__ float cf[4][EW_COEF_C];
  coef_table<4>(bn, nullptr, C, cf);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float sc[VEC], sh[VEC];
    for (int e = 0; e < VEC; ++e) {
      sc[e] = sh[e] = 0.f;
      const int c = cv * VEC + e;
      if (active && c < C) { float k[6]; coef_get<4>(bn, nullptr, c, C, cf, k); sc[e] = k[2]; sh[e] = k[3]; }
    }
    if (active) {
      const long step = (long)gridDim.x * L.rpp * U;
      for (long rb = (long)blockIdx.x * L.rpp * U + L.tr; rb < rows; rb += step) {
        typename VecT<T, VEC>::Raw vx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          vx[u] = VecT<T, VEC>::ldr(x + (r < rows ? r : rows - 1) * C + (long)cv * VEC);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          if (r >= rows) break;
          const VecT<T, VEC> w = VecT<T, VEC>::un(vx[u]);
          VecT<T, VEC> o;
          for (int e = 0; e < VEC; ++e) o.v[e] = stored<T>(fmaxf(fmaf(w.v[e], sc[e], sh[e]), 0.f));
          o.st(out + r * C + (long)cv * VEC);
        }
      }
    }
  }
}

// ---- sums += {sum g, sum g*shat} -----------------------------------------------------------------------
template <typename T, int VEC, int U>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ s, long rows,
                                                                 int C, mopoe_bn_ref bn, double* sums) {
  const ColLayout L(C, VEC);
  __shared__ float cf[2][EW_COEF_C];
  coef_table<2>(bn, nullptr, C, cf);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float mean[VEC], rstd[VEC], part[2][VEC];
    for (int e = 0; e < VEC; ++e) {
      mean[e] = rstd[e] = 0.f; part[0][e] = part[1][e] = 0.f;
      const int c = cv * VEC + e;
      if (active && c < C) { float k[6]; coef_get<2>(bn, nullptr, c, C, cf, k); mean[e] = k[0]; rstd[e] = k[1]; }
    }
    if (active) {
      const long step = (long)gridDim.x * L.rpp * U;
      for (long rb = (long)blockIdx.x * L.rpp * U + L.tr; rb < rows; rb += step) {
        typename VecT<T, VEC>::Raw vg[U], vs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          const long off = (r < rows ? r : rows - 1) * C + (long)cv * VEC;
          vg[u] = VecT<T, VEC>::ldr(g + off); vs[u] = VecT<T, VEC>::ldr(s + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (rb + (long)u * L.rpp >= rows) break;
          const VecT<T, VEC> wg = VecT<T, VEC>::un(vg[u]), ws = VecT<T, VEC>::un(vs[u]);
          for (int e = 0; e < VEC; ++e) {
            part[0][e] += wg.v[e];
            part[1][e] += wg.v[e] * ((ws.v[e] - mean[e]) * rstd[e]);
          }
        }
      }
    }
    double* const od[2] = {sums, sums + C};
    float* const of[2] = {nullptr, nullptr};
    block_col_reduce<VEC, 2>(L, active, cbase, C, part, od, of);
  }
}

// ---- dm = b*g*mask ; ds = a*BNbwd(g; s) ---------------------------------------------------------------
template <typename T, int VEC, int U>
__global__ __launch_bounds__(EW_THREADS) void block_out_bwd_kernel(const T* __restrict__ g, const T* __restrict__ s,
                                                                 T* __restrict__ dm, T* __restrict__ ds, long rows, int C,
                                                                 mopoe_bn_ref bn, const double* sums, mopoe_mask_ref mask,
                                                                 float a, float b, float* dgamma, float* dbeta,
                                                                 float* colsum_dm, float* colsum_ds) {
  const ColLayout L(C, VEC);
  __shared__ float cf[6][EW_COEF_C];
  coef_table<6>(bn, sums, C, cf);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float mean[VEC], rstd[VEC], gr[VEC], k1[VEC], k2[VEC], part[2][VEC];
    for (int e = 0; e < VEC; ++e) {
      mean[e] = rstd[e] = gr[e] = k1[e] = k2[e] = 0.f; part[0][e] = part[1][e] = 0.f;
      const int c = cv * VEC + e;
      if (active && c < C) {
        float k[6];
        coef_get<6>(bn, sums, c, C, cf, k);
        mean[e] = k[0]; rstd[e] = k[1]; gr[e] = a * k[2];  // a * gamma * rstd
        k1[e] = k[4]; k2[e] = k[5];
        if (blockIdx.x == 0 && L.tr == 0) {
          dgamma[c] = a * (float)sums[C + c];
          dbeta[c] = a * (float)sums[c];
        }
      }
    }
    if (active) {
      const long step = (long)gridDim.x * L.rpp * U;
      for (long rb = (long)blockIdx.x * L.rpp * U + L.tr; rb < rows; rb += step) {
        typename VecT<T, VEC>::Raw vg[U], vs[U];
        float mk[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          const long rc = r < rows ? r : rows - 1;
          const long off = rc * C + (long)cv * VEC;
          vg[u] = VecT<T, VEC>::ldr(g + off); vs[u] = VecT<T, VEC>::ldr(s + off);
          const float* mrow = mask_row(mask, rc, C);
          if (mrow) ld_maskv<VEC>(mrow + cv * VEC, mk[u]);
          else for (int e = 0; e < VEC; ++e) mk[u][e] = 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          if (r >= rows) break;
          const VecT<T, VEC> wg = VecT<T, VEC>::un(vg[u]), ws = VecT<T, VEC>::un(vs[u]);
          VecT<T, VEC> om, os;
          for (int e = 0; e < VEC; ++e) {
            om.v[e] = stored<T>(b * wg.v[e] * mk[u][e]);
            const float shat = (ws.v[e] - mean[e]) * rstd[e];
            os.v[e] = stored<T>(gr[e] * (wg.v[e] - k1[e] - shat * k2[e]));
            part[0][e] += om.v[e];
            part[1][e] += os.v[e];
          }
          const long off = r * C + (long)cv * VEC;
          om.st(dm + off);
          os.st(ds + off);
        }
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
template <typename T, int VEC, bool NEXT, int U>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                const T* __restrict__ add, T* __restrict__ dx, long rows,
                                                                int C, mopoe_bn_ref bn, const double* sums,
                                                                mopoe_mask_ref mask, float* dgamma, float* dbeta,
                                                                float* colsum_dx, const T* __restrict__ next_s,
                                                                mopoe_bn_ref next_bn, double* next_sums) {
  constexpr int NACC = NEXT ? 3 : 1;
  const ColLayout L(C, VEC);
  __shared__ float cf[6][EW_COEF_C];
  __shared__ float cfn[NEXT ? 2 : 1][NEXT ? EW_COEF_C : 1];
  coef_table<6>(bn, sums, C, cf);
  if constexpr (NEXT) coef_table<2>(next_bn, nullptr, C, cfn);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float mean[VEC], rstd[VEC], gr[VEC], k1[VEC], k2[VEC], nmean[VEC], nrstd[VEC], part[NACC][VEC];
    for (int e = 0; e < VEC; ++e) {
      mean[e] = rstd[e] = gr[e] = k1[e] = k2[e] = nmean[e] = nrstd[e] = 0.f;
      for (int k = 0; k < NACC; ++k) part[k][e] = 0.f;
      const int c = cv * VEC + e;
      if (active && c < C) {
        float k[6];
        coef_get<6>(bn, sums, c, C, cf, k);
        mean[e] = k[0]; rstd[e] = k[1]; gr[e] = k[2]; k1[e] = k[4]; k2[e] = k[5];
        if constexpr (NEXT) { float kn[6]; coef_get<2>(next_bn, nullptr, c, C, cfn, kn); nmean[e] = kn[0]; nrstd[e] = kn[1]; }
        if (blockIdx.x == 0 && L.tr == 0) {
          dgamma[c] = (float)sums[C + c];
          dbeta[c] = (float)sums[c];
        }
      }
    }
    if (active) {
      const long step = (long)gridDim.x * L.rpp * U;
      const bool has_mask = mask.kind != 0;
      for (long rb = (long)blockIdx.x * L.rpp * U + L.tr; rb < rows; rb += step) {
        typename VecT<T, VEC>::Raw vd[U], vx[U], va[U], vs[U];
        float mk[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          const long rc = r < rows ? r : rows - 1;
          const long off = rc * C + (long)cv * VEC;
          vd[u] = VecT<T, VEC>::ldr(dy + off); vx[u] = VecT<T, VEC>::ldr(x + off);
          if (add) va[u] = VecT<T, VEC>::ldr(add + off);
          if (NEXT) vs[u] = VecT<T, VEC>::ldr(next_s + off);
          if (has_mask) ld_maskv<VEC>(mask_row(mask, rc, C) + cv * VEC, mk[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          if (r >= rows) break;
          const VecT<T, VEC> wd = VecT<T, VEC>::un(vd[u]), wx = VecT<T, VEC>::un(vx[u]);
          VecT<T, VEC> wa, ws, o;
          if (add) wa = VecT<T, VEC>::un(va[u]);
          if (NEXT) ws = VecT<T, VEC>::un(vs[u]);
          for (int e = 0; e < VEC; ++e) {
            const float xhat = (wx.v[e] - mean[e]) * rstd[e];
            float v = gr[e] * (wd.v[e] - k1[e] - xhat * k2[e]);
            if (has_mask) v *= mk[u][e];
            if (add) v += wa.v[e];
            v = stored<T>(v);
            o.v[e] = v;
            part[0][e] += v;
            if constexpr (NEXT) { part[1][e] += v; part[2][e] += v * ((ws.v[e] - nmean[e]) * nrstd[e]); }
          }
          o.st(dx + r * C + (long)cv * VEC);
        }
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
template <typename T, int VEC, int U>
__global__ __launch_bounds__(EW_THREADS) void colsum_kernel(const T* __restrict__ x, float* out, long rows, int C) {
  const ColLayout L(C, VEC);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float part[1][VEC];
    for (int e = 0; e < VEC; ++e) part[0][e] = 0.f;
    if (active) {
      const long step = (long)gridDim.x * L.rpp * U;
      for (long rb = (long)blockIdx.x * L.rpp * U + L.tr; rb < rows; rb += step) {
        typename VecT<T, VEC>::Raw v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long r = rb + (long)u * L.rpp;
          v[u] = VecT<T, VEC>::ldr(x + (r < rows ? r : rows - 1) * C + (long)cv * VEC);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (rb + (long)u * L.rpp >= rows) break;
          const VecT<T, VEC> w = VecT<T, VEC>::un(v[u]);
          for (int e = 0; e < VEC; ++e) part[0][e] += w.v[e];
        }
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

// rows a thread has in flight per loop iteration (MOPOE_EW_UNROLL = 1 | 2 | 4, read once; tests/tools/glue_time.py sweeps it)
static inline int ew_unroll() {
  static const int u = ew_env("MOPOE_EW_UNROLL", 4);
  return u >= 8 ? 8 : (u >= 4 ? 4 : (u >= 2 ? 2 : 1));
}
#define EW_DISPATCH_U(LAUNCH_)       \
  switch (ew_unroll()) {             \
    case 8: LAUNCH_(8); break;       \
    case 4: LAUNCH_(4); break;       \
    case 2: LAUNCH_(2); break;       \
    default: LAUNCH_(1); break;      \
  }

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
static int bn_relu_apply_t(const T* x, T* out, int64_t rows, int32_t C, const mopoe_bn_ref* bn, void* stream) {
  EW_ARGCHECK(x && out && bn && bn->mode != 0 && bn->C == C && rows > 0, "bn_relu_apply: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  constexpr int W = EwVec<T>::wide;
#define MOPOE_L(U_) hipLaunchKernelGGL((bn_relu_apply_kernel<T, W, U_>), dim3(ew_grid(rows, C, W)), dim3(EW_THREADS), 0, st, x, out, (long)rows, C, *bn)
  if (ew_wide_ok<T>(C, {x, out}, "bn_relu_apply", &rc)) { EW_DISPATCH_U(MOPOE_L) }
#undef MOPOE_L
  else if constexpr (std::is_same<T, float>::value)
    hipLaunchKernelGGL((bn_relu_apply_kernel<float, 1, 1>), dim3(ew_grid(rows, C, 1)), dim3(EW_THREADS), 0, st, x, out, (long)rows, C, *bn);
  else return rc;
  return check_launch("bn_relu_apply");
}

template <typename T>
static int block_out_fwd_t(const T* s, const T* m, T* out, int64_t rows, int32_t C, const mopoe_bn_ref* bn_s, float a,
                           float b, double* out_stats, void* stream) {
  EW_ARGCHECK(s && m && out && bn_s && bn_s->mode != 0 && bn_s->C == C && rows > 0, "block_out_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  constexpr int W = EwVec<T>::wide;
#define MOPOE_L(U_) hipLaunchKernelGGL((block_out_fwd_kernel<T, W, U_>), dim3(ew_grid(rows, C, W)), dim3(EW_THREADS), 0, st, s, m, out, (long)rows, C, *bn_s, a, b, out_stats)
  if (ew_wide_ok<T>(C, {s, m, out}, "block_out_fwd", &rc)) { EW_DISPATCH_U(MOPOE_L) }
#undef MOPOE_L
  else if constexpr (std::is_same<T, float>::value)
    hipLaunchKernelGGL((block_out_fwd_kernel<float, 1, 1>), dim3(ew_grid(rows, C, 1)), dim3(EW_THREADS), 0, st, s, m, out, (long)rows, C, *bn_s, a, b, out_stats);
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
#define MOPOE_L(U_) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, W, U_>), dim3(ew_grid(rows, C, W)), dim3(EW_THREADS), 0, st, g, s, (long)rows, C, *bn_s, sums)
  if (ew_wide_ok<T>(C, {g, s}, "bn_bwd_reduce", &rc)) { EW_DISPATCH_U(MOPOE_L) }
#undef MOPOE_L
  else if constexpr (std::is_same<T, float>::value)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 1, 1>), dim3(ew_grid(rows, C, 1)), dim3(EW_THREADS), 0, st, g, s, (long)rows, C, *bn_s, sums);
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
#define MOPOE_L(U_) hipLaunchKernelGGL((block_out_bwd_kernel<T, W, U_>), dim3(ew_grid(rows, C, W)), dim3(EW_THREADS), 0, st, g, s, dm, ds, (long)rows, C, *bn_s, sums, mk, a, b, dgamma, dbeta, colsum_dm, colsum_ds)
  if (ew_wide_ok<T>(C, {g, s, dm, ds, mk.mask}, "block_out_bwd", &rc)) { EW_DISPATCH_U(MOPOE_L) }
#undef MOPOE_L
  else if constexpr (std::is_same<T, float>::value)
    hipLaunchKernelGGL((block_out_bwd_kernel<float, 1, 1>), dim3(ew_grid(rows, C, 1)), dim3(EW_THREADS), 0, st, g, s, dm, ds, (long)rows, C, *bn_s, sums, mk, a, b, dgamma, dbeta, colsum_dm, colsum_ds);
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
  const bool vec = ew_wide_ok<T>(C, {dy, x, add, dx, next_s, mk.mask}, "bn_bwd_apply", &rc);
  if (!vec && !std::is_same<T, float>::value) return rc;
  const dim3 grid(ew_grid(rows, C, vec ? W : 1)), blk(EW_THREADS);
#define MOPOE_APPLY(V_, N_, U_) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, V_, N_, U_>), grid, blk, 0, st, dy, x, add, dx, (long)rows, C, *bn, sums, mk, dgamma, dbeta, colsum_dx, next_s, nb, next_sums)
#define MOPOE_LN(U_) MOPOE_APPLY(W, true, U_)
#define MOPOE_LP(U_) MOPOE_APPLY(W, false, U_)
  if (vec) { if (next) { EW_DISPATCH_U(MOPOE_LN) } else { EW_DISPATCH_U(MOPOE_LP) } }
  else if constexpr (std::is_same<T, float>::value) { if (next) MOPOE_APPLY(1, true, 1); else MOPOE_APPLY(1, false, 1); }
#undef MOPOE_LN
#undef MOPOE_LP
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
#define MOPOE_L(U_) hipLaunchKernelGGL((colsum_kernel<T, W, U_>), dim3(ew_grid(rows, C, W)), dim3(EW_THREADS), 0, st, x, out, (long)rows, C)
  if (ew_wide_ok<T>(C, {x}, "colsum", &rc)) { EW_DISPATCH_U(MOPOE_L) }
#undef MOPOE_L
  else if constexpr (std::is_same<T, float>::value)
    hipLaunchKernelGGL((colsum_kernel<float, 1, 1>), dim3(ew_grid(rows, C, 1)), dim3(EW_THREADS), 0, st, x, out, (long)rows, C);
  else return rc;
  return check_launch("colsum");
}

extern "C" int mopoe_bn_relu_apply(const float* x, float* out, int64_t rows, int32_t C, const mopoe_bn_ref* bn, void* stream) {
  return bn_relu_apply_t<float>(x, out, rows, C, bn, stream);
}
extern "C" int mopoe_bn_relu_apply_bf16(const uint16_t* x, uint16_t* out, int64_t rows, int32_t C, const mopoe_bn_ref* bn,
                                        void* stream) {
  return bn_relu_apply_t<bf16_t>(x, out, rows, C, bn, stream);
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
