// Bandwidth-bound edge convolutions: the image-side layers with ONE channel on one side
//   stem  Conv2d(1 -> C, k3 s2 p1)                 (reference FeatureExtractorImg.py:29-34)
//   head  ConvTranspose2d(C -> 1, k3 s2 p1 op1)    (reference DataGeneratorImg.py:84-90)
// A GEMM tile would be >90 % padding here, so these are streaming kernels over the [pixels][C] tensor:
//   edge_expand : out[p][c]  = sum_tap scal[gather(p,tap)] * W[tap][c]      (stem forward, head dgrad)
//   edge_wgrad  : dW[tap][c] = sum_p   vec[p][c] * scal[gather(p,tap)]      (stem wgrad, head wgrad)
//   edge_reduce : out[q]     = b + sum_{tap hits p} sum_c x[p][c] * W[tap][c]  (head forward)
// p runs over the small grid, gather(p, tap) = p*stride - pad + tap on the big (single-channel) grid.
#include "ew_common.hpp"

namespace mopoe {

struct EdgeGeom {
  int N, Hs, Ws, Hb, Wb, C, kh, kw, sh, sw, ph, pw;
};

constexpr int MAXT = 9;

// thread owns VEC(4) fixed channels and walks small-grid pixels
template <typename T, int NT, int KW>
__global__ __launch_bounds__(EW_THREADS) void edge_expand_kernel(const float* __restrict__ scal, const float* __restrict__ W,
                                                                T* __restrict__ out, const EdgeGeom g, double* stats) {
  const long rows = (long)g.N * g.Hs * g.Ws;
  const ColLayout L(g.C, 4);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float w[NT][4], part[2][4];
    for (int e = 0; e < 4; ++e) part[0][e] = part[1][e] = 0.f;
    if (active)
      for (int t = 0; t < NT; ++t) {
        const Vec<4> v = Vec<4>::ld(W + (long)t * g.C + cv * 4);
        for (int e = 0; e < 4; ++e) w[t][e] = v.v[e];
      }
    if (active) {
      const unsigned hw = g.Hs * g.Ws;
      for (long r = (long)blockIdx.x * L.rpp + L.tr; r < rows; r += (long)gridDim.x * L.rpp) {
        const unsigned n = (unsigned)r / hw, rem = (unsigned)r - n * hw;
        const int qy = rem / g.Ws, qx = rem - qy * g.Ws;
        const float* src = scal + (long)n * g.Hb * g.Wb;
        VecT<T, 4> o;
        for (int e = 0; e < 4; ++e) o.v[e] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int by = qy * g.sh - g.ph + t / KW, bx = qx * g.sw - g.pw + t % KW;
          float sv = 0.f;
          if (by >= 0 && by < g.Hb && bx >= 0 && bx < g.Wb) sv = src[by * g.Wb + bx];
          for (int e = 0; e < 4; ++e) o.v[e] = fmaf(sv, w[t][e], o.v[e]);
        }
        for (int e = 0; e < 4; ++e) { o.v[e] = stored<T>(o.v[e]); part[0][e] += o.v[e]; part[1][e] += o.v[e] * o.v[e]; }
        o.st(out + r * g.C + (long)cv * 4);
      }
    }
    if (stats) {
      double* const od[2] = {stats, stats + g.C};
      float* const of[2] = {nullptr, nullptr};
      block_col_reduce<4, 2>(L, active, cbase, g.C, part, od, of);
    }
  }
}

template <typename T, int NT, int KW>
__global__ __launch_bounds__(EW_THREADS) void edge_wgrad_kernel(const T* __restrict__ vec, const float* __restrict__ scal,
                                                               float* __restrict__ dW, const EdgeGeom g) {
  const long rows = (long)g.N * g.Hs * g.Ws;
  const ColLayout L(g.C, 4);
  for (int cbase = 0; cbase < L.Cv; cbase += L.cols) {
    const int cv = cbase + L.tc;
    const bool active = cv < L.Cv && L.tr < L.rpp;
    float part[NT][4];
    for (int t = 0; t < NT; ++t)
      for (int e = 0; e < 4; ++e) part[t][e] = 0.f;
    if (active) {
      const unsigned hw = g.Hs * g.Ws;
      for (long r = (long)blockIdx.x * L.rpp + L.tr; r < rows; r += (long)gridDim.x * L.rpp) {
        const unsigned n = (unsigned)r / hw, rem = (unsigned)r - n * hw;
        const int qy = rem / g.Ws, qx = rem - qy * g.Ws;
        const float* src = scal + (long)n * g.Hb * g.Wb;
        const VecT<T, 4> v = VecT<T, 4>::ld(vec + r * g.C + (long)cv * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int by = qy * g.sh - g.ph + t / KW, bx = qx * g.sw - g.pw + t % KW;
          float sv = 0.f;
          if (by >= 0 && by < g.Hb && bx >= 0 && bx < g.Wb) sv = src[by * g.Wb + bx];
          for (int e = 0; e < 4; ++e) part[t][e] = fmaf(sv, v.v[e], part[t][e]);
        }
      }
    }
    double* od[NT];
    float* of[NT];
    for (int t = 0; t < NT; ++t) { od[t] = nullptr; of[t] = dW + (long)t * g.C; }
    double* const (&odr)[NT] = od;
    float* const (&ofr)[NT] = of;
    block_col_reduce<4, NT>(L, active, cbase, g.C, part, odr, ofr);
  }
}

// head forward: 16 lanes per output pixel (4 channels each, C <= 64*... loops over channel groups), shuffle reduce
template <typename T, int NT, int KW, int S>
__global__ __launch_bounds__(256) void edge_reduce_kernel(const T* x, const float* W, const float* bias, float* out,
                                                        const EdgeGeom g) {
  const long total = (long)g.N * g.Hb * g.Wb;
  const int sub = threadIdx.x & 15;                         // lane within the 16-lane group
  const long grp0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const long ngrp = ((long)gridDim.x * blockDim.x) >> 4;
  const float b0 = bias ? bias[0] : 0.f;
  const unsigned hwb = g.Hb * g.Wb;
  for (long q = grp0; q < total; q += ngrp) {
    const unsigned n = (unsigned)q / hwb, rem = (unsigned)q - n * hwb;
    const int oy = rem / g.Wb, ox = rem - oy * g.Wb;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ky = t / KW, kx = t % KW;
      const int ny = oy + g.ph - ky, nx = ox + g.pw - kx;   // = small * stride
      if (ny < 0 || nx < 0 || ny % S != 0 || nx % S != 0) continue;
      const int qy = ny / S, qx = nx / S;
      if (qy >= g.Hs || qx >= g.Ws) continue;
      const T* xr = x + (((long)n * g.Hs + qy) * g.Ws + qx) * g.C;
      const float* wr = W + (long)t * g.C;
      for (int c = sub * 4; c < g.C; c += 64) {
        const VecT<T, 4> xv = VecT<T, 4>::ld(xr + c);
        const float4 wv = *reinterpret_cast<const float4*>(wr + c);
        acc = fmaf(xv.v[0], wv.x, acc); acc = fmaf(xv.v[1], wv.y, acc);
        acc = fmaf(xv.v[2], wv.z, acc); acc = fmaf(xv.v[3], wv.w, acc);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0) out[q] = acc + b0;
  }
}

// head forward for the reference's geometry (k3 s2 p1, output_padding 1: big grid = 2 x small grid): 16 lanes per 2x2
// OUTPUT QUAD.  With oy = 2 qy - 1 + ky the quad (2a..2a+1, 2b..2b+1) reads exactly the four input pixels
// (a..a+1, b..b+1), once each, instead of every output pixel re-reading up to four inputs (2.25x less L2 traffic):
//   out[2a  ][2b  ] = x[a][b].W11
//   out[2a  ][2b+1] = x[a][b+1].W10 + x[a][b].W12
//   out[2a+1][2b  ] = x[a+1][b].W01 + x[a][b].W21
//   out[2a+1][2b+1] = x[a+1][b+1].W00 + x[a+1][b].W02 + x[a][b+1].W20 + x[a][b].W22
template <typename T, bool C64>   // C64: C <= 64, one channel group per lane: the nine tap vectors stay in registers
__global__ __launch_bounds__(256) void edge_reduce_quad_kernel(const T* x, const float* W, const float* bias, float* out,
                                                             const EdgeGeom g) {
  const long total = (long)g.N * g.Hs * g.Ws;               // quads
  const int sub = threadIdx.x & 15;
  const long grp0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const long ngrp = ((long)gridDim.x * blockDim.x) >> 4;
  const float b0 = bias ? bias[0] : 0.f;
  const unsigned hws = g.Hs * g.Ws;
  float4 wreg[9];
  if (C64) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
      wreg[tp] = sub * 4 < g.C ? *reinterpret_cast<const float4*>(W + (long)tp * g.C + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (long q = grp0; q < total; q += ngrp) {
    const unsigned n = (unsigned)q / hws, rem = (unsigned)q - n * hws;
    const int a = rem / g.Ws, b = rem - a * g.Ws;
    const bool hy = a + 1 < g.Hs, hx = b + 1 < g.Ws;
    const T* x00 = x + (((long)n * g.Hs + a) * g.Ws + b) * g.C;
    float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
    for (int c = sub * 4; c < g.C; c += 64) {
      VecT<T, 4> z;
      z.v[0] = z.v[1] = z.v[2] = z.v[3] = 0.f;
      const VecT<T, 4> v00 = VecT<T, 4>::ld(x00 + c);
      const VecT<T, 4> v01 = hx ? VecT<T, 4>::ld(x00 + g.C + c) : z;
      const VecT<T, 4> v10 = hy ? VecT<T, 4>::ld(x00 + (long)g.Ws * g.C + c) : z;
      const VecT<T, 4> v11 = (hx && hy) ? VecT<T, 4>::ld(x00 + (long)(g.Ws + 1) * g.C + c) : z;
      auto dot = [&](const VecT<T, 4>& u, int tap) {
        const float4 w = C64 ? wreg[tap] : *reinterpret_cast<const float4*>(W + (long)tap * g.C + c);
        return fmaf(u.v[0], w.x, fmaf(u.v[1], w.y, fmaf(u.v[2], w.z, u.v[3] * w.w)));
      };
      o00 += dot(v00, 4);
      o01 += dot(v01, 3) + dot(v00, 5);
      o10 += dot(v10, 1) + dot(v00, 7);
      o11 += dot(v11, 0) + dot(v10, 2) + dot(v01, 6) + dot(v00, 8);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      o00 += __shfl_xor(o00, o, 64); o01 += __shfl_xor(o01, o, 64);
      o10 += __shfl_xor(o10, o, 64); o11 += __shfl_xor(o11, o, 64);
    }
    if (sub == 0) {
      float* o = out + ((long)n * g.Hb + 2 * a) * g.Wb + 2 * b;
      *reinterpret_cast<float2*>(o) = make_float2(o00 + b0, o01 + b0);
      *reinterpret_cast<float2*>(o + g.Wb) = make_float2(o10 + b0, o11 + b0);
    }
  }
}

static EdgeGeom edge_geom(const mopoe_conv_geom* g, int C) {
  return EdgeGeom{g->N, g->Hs, g->Ws, g->Hb, g->Wb, C, g->kh, g->kw, g->sh, g->sw, g->ph, g->pw};
}

// ---- entry points used by the conv dispatchers (conv_gemm.hip: T = float, conv_gemm_bf16.hip: T = bf16_t) ---------
// The wide ([pixels][C]) tensor has storage type T; the single-channel image side, the taps W[9][C] and the tap
// gradients stay fp32 in both cases.
bool edge_supported(const mopoe_conv_geom* g, int C, std::initializer_list<const void*> ptrs) {
  return g->kh == 3 && g->kw == 3 && g->sh == 2 && g->sw == 2 && C % 4 == 0 && C >= 4 && vec_ok(C, ptrs);
}

template <typename T>
int edge_expand(const float* scal, const float* W, T* out, const mopoe_conv_geom* g, int C, double* stats,
                hipStream_t st) {
  const EdgeGeom eg = edge_geom(g, C);
  const long rows = (long)g->N * g->Hs * g->Ws;
  hipLaunchKernelGGL((edge_expand_kernel<T, 9, 3>), dim3(ew_grid(rows, C, 4)), dim3(EW_THREADS), 0, st, scal, W, out, eg, stats);
  return check_launch("edge_expand");
}

template <typename T>
int edge_wgrad(const T* vec, const float* scal, float* dW, const mopoe_conv_geom* g, int C, hipStream_t st) {
  const EdgeGeom eg = edge_geom(g, C);
  const long rows = (long)g->N * g->Hs * g->Ws;
  if (hipMemsetAsync(dW, 0, sizeof(float) * 9 * C, st) != hipSuccess) { set_error("edge_wgrad memset failed"); return MOPOE_ERR_LAUNCH; }
  hipLaunchKernelGGL((edge_wgrad_kernel<T, 9, 3>), dim3(ew_grid(rows, C, 4)), dim3(EW_THREADS), 0, st, vec, scal, dW, eg);
  return check_launch("edge_wgrad");
}

template <typename T>
int edge_reduce(const T* x, const float* W, const float* bias, float* out, const mopoe_conv_geom* g, int C,
                hipStream_t st) {
  const EdgeGeom eg = edge_geom(g, C);
  if (g->ph == 1 && g->pw == 1 && g->Hb == 2 * g->Hs && g->Wb == 2 * g->Ws && (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
    const long quads = (long)g->N * g->Hs * g->Ws;
    long qb = (quads * 16 + 255) / 256;
    if (qb > 4096) qb = 4096;
    if (C <= 64) hipLaunchKernelGGL((edge_reduce_quad_kernel<T, true>), dim3((unsigned)qb), dim3(256), 0, st, x, W, bias, out, eg);
    else hipLaunchKernelGGL((edge_reduce_quad_kernel<T, false>), dim3((unsigned)qb), dim3(256), 0, st, x, W, bias, out, eg);
    return check_launch("edge_reduce_quad");
  }
  const long total = (long)g->N * g->Hb * g->Wb;
  long blocks = (total * 16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((edge_reduce_kernel<T, 9, 3, 2>), dim3((unsigned)blocks), dim3(256), 0, st, x, W, bias, out, eg);
  return check_launch("edge_reduce");
}

template int edge_expand<float>(const float*, const float*, float*, const mopoe_conv_geom*, int, double*, hipStream_t);
template int edge_expand<bf16_t>(const float*, const float*, bf16_t*, const mopoe_conv_geom*, int, double*, hipStream_t);
template int edge_wgrad<float>(const float*, const float*, float*, const mopoe_conv_geom*, int, hipStream_t);
template int edge_wgrad<bf16_t>(const bf16_t*, const float*, float*, const mopoe_conv_geom*, int, hipStream_t);
template int edge_reduce<float>(const float*, const float*, const float*, float*, const mopoe_conv_geom*, int, hipStream_t);
template int edge_reduce<bf16_t>(const bf16_t*, const float*, const float*, float*, const mopoe_conv_geom*, int, hipStream_t);

}  // namespace mopoe
