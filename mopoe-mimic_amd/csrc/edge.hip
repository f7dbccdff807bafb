// Bandwidth-bound edge convolutions: the image-side layers with ONE channel on one side
//   stem  Conv2d(1 -> C, k3 s2 p1)                 (reference FeatureExtractorImg.py:29-34)
//   head  ConvTranspose2d(C -> 1, k3 s2 p1 op1)    (reference DataGeneratorImg.py:84-90)
// A GEMM tile would be >90 % padding here, so these are streaming kernels over the [pixels][C] tensor:
//   edge_expand : out[p][c]  = sum_tap scal[gather(p,tap)] * W[tap][c]      (stem forward, head dgrad)
//   edge_wgrad  : dW[tap][c] = sum_p   vec[p][c] * scal[gather(p,tap)]      (stem wgrad, head wgrad)
//   edge_reduce : out[q]     = b + sum_{tap hits p} sum_c x[p][c] * W[tap][c]  (head forward)
// p runs over the small grid, gather(p, tap) = p*stride - pad + tap on the big (single-channel) grid.
#include <algorithm>
#include "ew_common.hpp"

namespace mopoe {

struct EdgeGeom {
  int N, Hs, Ws, Hb, Wb, C, kh, kw, sh, sw, ph, pw;
};

constexpr int EDGE_MFMA_BLOCKS = 512;   // 2 blocks of 4 waves per CU: per-block column atomics grow with the grid

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


// =====================================================================================================================
// MFMA forms for C = 64 (DIM_img of every BASELINE config).  The streaming kernels above spend ~29 vector instructions
// per output element on the 9-tap gather and run at ~1 TB/s; here the taps are the K dimension of v_mfma_f32_32x32x2_f32
// (exact fp32 products and sums, as before), the gather is ONE scalar load per lane per MFMA, and the wide tensor moves
// in full rows.  rocprofv3 (C3, B = 256): expand 124 -> see profiles/, wgrad 124 -> see profiles/.
// =====================================================================================================================
typedef float f32x16e __attribute__((ext_vector_type(16)));

// out[p][c] = sum_t scal[gather(p, t)] * W[t][c]:  M = 32 pixels per wave tile, N = 2 x 32 channels, K = 9 taps (5 x K2)
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void edge_expand_mfma_kernel(const float* __restrict__ scal, const float* __restrict__ W,
                                                                 T* __restrict__ out, const EdgeGeom g, double* stats) {
  constexpr int C = 64, STG_LD = C + 4;
  __shared__ __attribute__((aligned(16))) float stg_all[NW * 32 * STG_LD];
  __shared__ float red[2][NW][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  float* const stg = stg_all + wave * 32 * STG_LD;
  // B operand: B[k = lhi][n = l31] of step i = W[tap 2 i + lhi][32 j + l31]
  float b[5][2];
  int ky[5], kx[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int t = 2 * i + lhi;
    ky[i] = t / 3; kx[i] = t - 3 * ky[i];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[i][j] = t < 9 ? W[t * C + 32 * j + l31] : 0.f;
  }
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  const long rows = (long)g.N * g.Hs * g.Ws;
  const long ntiles = (rows + 31) / 32;
  const unsigned hw = (unsigned)(g.Hs * g.Ws);
  const int rsub = lane >> 3, c8 = lane & 7;      // store phase: 8 lanes per row, 8 rows per pass
  for (long tile = (long)blockIdx.x * NW + wave; tile < ntiles; tile += (long)gridDim.x * NW) {
    const long p = tile * 32 + l31;
    const bool valid = p < rows;
    const unsigned pu = valid ? (unsigned)p : 0u;
    const unsigned n = pu / hw, rem = pu - n * hw;
    const int qy = (int)(rem / (unsigned)g.Ws), qx = (int)(rem - (unsigned)qy * (unsigned)g.Ws);
    const float* src = scal + (long)n * g.Hb * g.Wb;
    float a[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int by = qy * g.sh - g.ph + ky[i], bx = qx * g.sw - g.pw + kx[i];
      const bool ok = valid && (2 * i + lhi < 9) && (unsigned)by < (unsigned)g.Hb && (unsigned)bx < (unsigned)g.Wb;
      a[i] = ok ? src[by * g.Wb + bx] : 0.f;
    }
    f32x16e acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i][j], acc[j], 0, 0, 0);
    // accumulator element r of lane (l31, lhi): pixel row (r & 3) + 8 (r >> 2) + 4 lhi of the tile, channel 32 j + l31.
    // Rows past the end hold exact zeros (their taps were zero): they add nothing to the statistics.
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = stored<T>(acc[j][r]);
        s1[j] += v;
        s2[j] = fmaf(v, v, s2[j]);
        stg[((r & 3) + 8 * (r >> 2) + 4 * lhi) * STG_LD + 32 * j + l31] = v;
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int rr = pass * 8 + rsub;
      const long prow = tile * 32 + rr;
      const float4 q0 = *reinterpret_cast<const float4*>(&stg[rr * STG_LD + c8 * 8]);
      const float4 q1 = *reinterpret_cast<const float4*>(&stg[rr * STG_LD + c8 * 8 + 4]);
      if (prow < rows) {
        T* dst = out + prow * C + c8 * 8;
        if constexpr (sizeof(T) == 2) {
          *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16(q0.x, q0.y), pack_bf16(q0.z, q0.w), pack_bf16(q1.x, q1.y), pack_bf16(q1.z, q1.w));
        } else {
          *reinterpret_cast<float4*>(dst) = q0;
          *reinterpret_cast<float4*>(dst + 4) = q1;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (stats) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s1[j] += __shfl_xor(s1[j], 32, 64);
      s2[j] += __shfl_xor(s2[j], 32, 64);
      if (lhi == 0) { red[0][wave][32 * j + l31] = s1[j]; red[1][wave][32 * j + l31] = s2[j]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
      const int k = threadIdx.x / C, c = threadIdx.x - k * C;
      float tsum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tsum += red[k][w][c];
      atomic_add_f64(stats + k * C + c, (double)tsum);
    }
  }
}

// dW[t][c] = sum_p vec[p][c] * scal[gather(p, t)] on v_mfma_f32_16x16x4_f32:  M = 9 taps (of 16), N = 4 x 16 channels,
// K = pixels, 4 per MFMA.  Lane (n = lane % 16, kq = lane / 16) reads the FOUR neighbouring channels 4 n .. 4 n + 3 of
// pixel p0 + kq with one 8 / 16-byte load and feeds them to four MFMAs (column n of MFMA j is channel 4 n + j); lanes
// n < 9 also fetch the tap value scal[gather(p0 + kq, n)].  A wave walks runs of 16 pixels of one image row (4 K steps),
// the loads of the next run are issued before the MFMAs of the current one.
typedef float f32x4e __attribute__((ext_vector_type(4)));
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void edge_wgrad_mfma_kernel(const T* __restrict__ vec, const float* __restrict__ scal,
                                                                float* __restrict__ dW, const EdgeGeom g) {
  constexpr int C = 64;
  __shared__ float red[NW][9][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, kq = lane >> 4;
  const bool tap_lane = n16 < 9;
  const int ky = tap_lane ? n16 / 3 : 0, kx = tap_lane ? n16 - 3 * (n16 / 3) : 0;
  const long rows = (long)g.N * g.Hs * g.Ws;
  const long nruns = rows / 16;                       // (Ws % 16 == 0: a run never leaves its image row)
  f32x4e acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
  float av[4], bv[4][4];
  auto load_run = [&](long run) {
    const long p0 = run * 16;
    const unsigned row = (unsigned)(p0 / g.Ws);        // image row index n * Hs + qy
    const int qx0 = (int)(p0 - (long)row * g.Ws);
    const unsigned n = row / (unsigned)g.Hs;
    const int qy = (int)(row - n * (unsigned)g.Hs);
    const int by = qy * g.sh - g.ph + ky;
    const bool rowok = tap_lane && (unsigned)by < (unsigned)g.Hb;
    const float* srow = scal + ((long)n * g.Hb + (rowok ? by : 0)) * g.Wb;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int bx = (qx0 + 4 * u + kq) * g.sw - g.pw + kx;
      av[u] = (rowok && (unsigned)bx < (unsigned)g.Wb) ? srow[bx] : 0.f;
      const T* vp = vec + (p0 + 4 * u + kq) * C + 4 * n16;
      if constexpr (sizeof(T) == 2) {
        const uint2 w2 = *reinterpret_cast<const uint2*>(vp);
        bv[u][0] = bf16_lo(w2.x); bv[u][1] = bf16_hi(w2.x); bv[u][2] = bf16_lo(w2.y); bv[u][3] = bf16_hi(w2.y);
      } else {
        const float4 w4 = *reinterpret_cast<const float4*>(vp);
        bv[u][0] = w4.x; bv[u][1] = w4.y; bv[u][2] = w4.z; bv[u][3] = w4.w;
      }
    }
  };
  const long stride = (long)gridDim.x * NW;
  long run = (long)blockIdx.x * NW + wave;
  if (run < nruns) load_run(run);
  while (run < nruns) {
    float ca[4], cb[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ca[u] = av[u];
#pragma unroll
      for (int j = 0; j < 4; ++j) cb[u][j] = bv[u][j];
    }
    run += stride;
    if (run < nruns) load_run(run);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[u], cb[u][j], acc[j], 0, 0, 0);
  }
  // accumulator element r of lane (n16, kq): tap 4 kq + r, channel 4 n16 + j
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * kq + r < 9) red[wave][4 * kq + r][4 * n16 + j] = acc[j][r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 9 * C; idx += 64 * NW) {
    const int t = idx / C, c = idx - t * C;
    float tsum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tsum += red[w][t][c];
    unsafeAtomicAdd(dW + t * C + c, tsum);
  }
}

// head forward out[2a + dy][2b + dx] = b + (the quad formula above) on v_mfma_f32_16x16x4_f32: a block owns RT = 7 rows
// of one image plus the halo row below; P[pixel][tap] = sum_c x[pixel][c] W[tap][c] (M = 16 pixels, N = 9 taps of 16,
// K = 64 channels: lane (pixel i = lane % 16, kq = lane / 16) reads the 16 consecutive channels 16 kq .. 16 kq + 15 of its
// pixel with 16-byte loads, K step s multiplies channel 16 kq + s) goes to LDS, then every thread assembles output quads
// from the four pixels (a..a+1, b..b+1) it touches and writes them as float2 rows.
template <typename T>
__global__ __launch_bounds__(256) void edge_reduce_mfma_kernel(const T* __restrict__ x, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             const EdgeGeom g, int row_blocks) {
  constexpr int C = 64, RT = 7, PL = 12;
  extern __shared__ __attribute__((aligned(16))) float P[];   // [(RT + 1) * Ws][PL]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int n = blockIdx.x / row_blocks, a0 = (blockIdx.x - n * row_blocks) * RT;
  // B operand of K step s: B[k = kq][j = i16] = W[tap i16][channel 16 kq + s]
  float bw[16];
#pragma unroll
  for (int sidx = 0; sidx < 16; ++sidx) bw[sidx] = i16 < 9 ? W[i16 * C + 16 * kq + sidx] : 0.f;
  const int ntile = (RT + 1) * g.Ws / 16;
  for (int tile = wave; tile < ntile; tile += 4) {
    const int q0 = tile * 16;
    const int ar = q0 / g.Ws, b = q0 - ar * g.Ws + i16;
    const int a = a0 + ar;
    f32x4e acc;
    acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
    if (a < g.Hs) {     // wave-uniform: a tile never leaves its image row (Ws % 16 == 0)
      const T* xp = x + (((long)n * g.Hs + a) * g.Ws + b) * C + 16 * kq;
      float av[16];
      if constexpr (sizeof(T) == 2) {
        const uint4 u0 = *reinterpret_cast<const uint4*>(xp), u1 = *reinterpret_cast<const uint4*>(xp + 8);
        av[0] = bf16_lo(u0.x); av[1] = bf16_hi(u0.x); av[2] = bf16_lo(u0.y); av[3] = bf16_hi(u0.y);
        av[4] = bf16_lo(u0.z); av[5] = bf16_hi(u0.z); av[6] = bf16_lo(u0.w); av[7] = bf16_hi(u0.w);
        av[8] = bf16_lo(u1.x); av[9] = bf16_hi(u1.x); av[10] = bf16_lo(u1.y); av[11] = bf16_hi(u1.y);
        av[12] = bf16_lo(u1.z); av[13] = bf16_hi(u1.z); av[14] = bf16_lo(u1.w); av[15] = bf16_hi(u1.w);
      } else {
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
          const float4 f = *reinterpret_cast<const float4*>(xp + 4 * v4);
          av[4 * v4] = f.x; av[4 * v4 + 1] = f.y; av[4 * v4 + 2] = f.z; av[4 * v4 + 3] = f.w;
        }
      }
#pragma unroll
      for (int sidx = 0; sidx < 16; ++sidx) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sidx], bw[sidx], acc, 0, 0, 0);
    }
    // accumulator element r of lane (i16, kq): pixel q0 + 4 kq + r, tap i16
    if (i16 < 9) {
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(q0 + 4 * kq + r) * PL + i16] = acc[r];
    }
  }
  __syncthreads();
  const float b0 = bias ? bias[0] : 0.f;
  const int nquad = RT * g.Ws;
  for (int q = threadIdx.x; q < nquad; q += 256) {
    const int ar = q / g.Ws, b = q - ar * g.Ws;
    const int a = a0 + ar;
    if (a >= g.Hs) break;
    const float* p00 = P + q * PL;
    const float* p10 = p00 + g.Ws * PL;               // halo row: zeros when a + 1 == Hs
    const bool hx = b + 1 < g.Ws;
    auto at = [&](const float* pp, int t, bool ok) { return ok ? pp[t] : 0.f; };
    const float o00 = p00[4];
    const float o01 = at(p00 + PL, 3, hx) + p00[5];
    const float o10 = p10[1] + p00[7];
    const float o11 = at(p10 + PL, 0, hx) + p10[2] + at(p00 + PL, 6, hx) + p00[8];
    float* o = out + ((long)n * g.Hb + 2 * a) * g.Wb + 2 * b;
    *reinterpret_cast<float2*>(o) = make_float2(o00 + b0, o01 + b0);
    *reinterpret_cast<float2*>(o + g.Wb) = make_float2(o10 + b0, o11 + b0);
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
  static const bool use_mfma = !getenv("MOPOE_EDGE_VALU");
  if (use_mfma && C == 64 && vec_ok(C, {out}, 8)) {
    // with statistics: 8-wave blocks, one per CU -- the per-block column atomics (128 doubles) serialise per address
    static const int stat_blocks = ew_env("MOPOE_EDGE_STAT_BLOCKS", 0);
    if (stats) {
      const long blocks = std::min<long>((rows + 255) / 256, stat_blocks > 0 ? stat_blocks : (rows >= (1L << 19) ? 512 : 256));
      hipLaunchKernelGGL((edge_expand_mfma_kernel<T, 8>), dim3((unsigned)blocks), dim3(512), 0, st, scal, W, out, eg, stats);
    } else {
      const long blocks = std::min<long>((rows + 127) / 128, EDGE_MFMA_BLOCKS);
      hipLaunchKernelGGL((edge_expand_mfma_kernel<T, 4>), dim3((unsigned)blocks), dim3(256), 0, st, scal, W, out, eg, stats);
    }
    return check_launch("edge_expand_mfma");
  }
  hipLaunchKernelGGL((edge_expand_kernel<T, 9, 3>), dim3(ew_grid(rows, C, 4)), dim3(EW_THREADS), 0, st, scal, W, out, eg, stats);
  return check_launch("edge_expand");
}

template <typename T>
int edge_wgrad(const T* vec, const float* scal, float* dW, const mopoe_conv_geom* g, int C, hipStream_t st, bool dw_is_zero) {
  const EdgeGeom eg = edge_geom(g, C);
  const long rows = (long)g->N * g->Hs * g->Ws;
  if (!dw_is_zero && hipMemsetAsync(dW, 0, sizeof(float) * 9 * C, st) != hipSuccess) { set_error("edge_wgrad memset failed"); return MOPOE_ERR_LAUNCH; }
  static const bool use_mfma = !getenv("MOPOE_EDGE_VALU");
  if (use_mfma && C == 64 && g->Ws % 16 == 0 && vec_ok(C, {vec}, 8)) {
    static const int wg_blocks = ew_env("MOPOE_EDGE_WGRAD_BLOCKS", EDGE_MFMA_BLOCKS);
    const long blocks = std::min<long>((rows / 16 + 7) / 8, wg_blocks);
    hipLaunchKernelGGL((edge_wgrad_mfma_kernel<T, 8>), dim3((unsigned)blocks), dim3(512), 0, st, vec, scal, dW, eg);
    return check_launch("edge_wgrad_mfma");
  }
  hipLaunchKernelGGL((edge_wgrad_kernel<T, 9, 3>), dim3(ew_grid(rows, C, 4)), dim3(EW_THREADS), 0, st, vec, scal, dW, eg);
  return check_launch("edge_wgrad");
}

template <typename T>
int edge_reduce(const T* x, const float* W, const float* bias, float* out, const mopoe_conv_geom* g, int C,
                hipStream_t st) {
  const EdgeGeom eg = edge_geom(g, C);
  static const bool use_mfma = !getenv("MOPOE_EDGE_VALU");
  if (use_mfma && C == 64 && g->ph == 1 && g->pw == 1 && g->Hb == 2 * g->Hs && g->Wb == 2 * g->Ws && g->Ws % 16 == 0 &&
      g->Ws <= 128 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && vec_ok(C, {x}, 8)) {   // (<= 48 KiB of dynamic LDS)
    const int row_blocks = (g->Hs + 6) / 7;
    const size_t lds = (size_t)8 * g->Ws * 12 * sizeof(float);
    hipLaunchKernelGGL((edge_reduce_mfma_kernel<T>), dim3((unsigned)(g->N * row_blocks)), dim3(256), lds, st, x, W, bias, out, eg, row_blocks);
    return check_launch("edge_reduce_mfma");
  }
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
template int edge_wgrad<float>(const float*, const float*, float*, const mopoe_conv_geom*, int, hipStream_t, bool);
template int edge_wgrad<bf16_t>(const bf16_t*, const float*, float*, const mopoe_conv_geom*, int, hipStream_t, bool);
template int edge_reduce<float>(const float*, const float*, const float*, float*, const mopoe_conv_geom*, int, hipStream_t);
template int edge_reduce<bf16_t>(const bf16_t*, const float*, const float*, float*, const mopoe_conv_geom*, int, hipStream_t);

}  // namespace mopoe
