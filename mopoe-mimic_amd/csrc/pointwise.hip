// The FRONT of a residual block -- bn1 -> relu -> conv1 (1x1) -> dropout -> bn2 -> relu -- as streaming kernels (round 4).
//
// Reference: mimic/networks/ResidualBlocks.py:84-97,118-131 (2-D blocks; the 1-D ones :20-33,51-65 have the same front):
//     h1 = relu(bn1(x));  d1 = drop1(conv1(h1));  a2 = relu(bn2(d1))  ->  conv2(a2) ...
// The 1x1 conv on the block's big input grid is HBM-bound (64 channels: 0.5 FLOP per byte and pass in bf16), and until round 3
// its neighbourhood cost twelve passes over [rows, C] tensors per block and step:
//     forward   conv1 (read x, write d1) + bn_relu_apply (read d1, write a2)                                     4
//     backward  bn_bwd_apply (read dh2, d1; write dc1) + conv1 dgrad (read dc1, x; write dh1) + conv1 wgrad (read x, dc1)   8
// d1 exists only to be normalised: its batch statistics need a pass of their own, everything else can be RECOMPUTED from x
// (64 MFMAs per 32 pixels, a few per cent of the pass's HBM time).  Here:
//     forward   pw_front_fwd<WRITE = false>: read x -> statistics of d1 (nothing written)                       1
//               pw_front_fwd<WRITE = true>:  read x -> d1 again (bit-identical: same instructions) -> a2        2
//     backward  pw_front_bwd: read x, dh2 -> d1 again -> dc1 -> dh1 (written), dW1 += h1^T dc1, bias gradient,
//               the two BatchNorm-backward sums of bn1                                                          3
// six passes instead of twelve, and d1 / dc1 never exist in HBM.  (conv2's input gradient takes its ReLU mask and x-hat from a2
// instead of d1: mopoe_bn_ref mode 3.)
//
// Layout.  One wave owns 32 consecutive pixels (rows).  Lane l = (p = l % 32, h = l / 32) holds, for every 16-channel group
// s, the 8 consecutive channels 16 s + 8 h + j of pixel p -- a 16-byte piece of the row, which is at once
//   * what a global_load_dwordx4 / global_store_dwordx4 moves,
//   * the B operand of v_mfma_f32_32x32x16_bf16 for k-step s (lane = column p, k = 8 h + j), and
//   * with the rows of the A operand (the weights) listed in the order sigma below, what the accumulator gives back:
//     register r of lane (p, h) of output tile t is row m = 8 (r / 4) + 4 h + r % 4, which sigma maps to channel
//     32 t + 16 (r / 8) + 8 h + r % 8: registers 0-7 / 8-15 of tile t ARE groups 2 t / 2 t + 1 of the same piece layout.
// So x -> h1 -> (MFMA) d1 -> dc1 -> (MFMA) dh1 runs register to register, every global access is a 16-byte piece of a row,
// and no LDS transposition sits between the two GEMMs (the trick of fused attention kernels: P feeds PV as it lies).
// Reductions over pixels (statistics, BatchNorm-backward sums, the bias gradient) and the weight gradient (K = pixels) need
// the transposed view: each wave writes the tiles concerned into a private, swizzled [32 pixels][C] LDS patch and reads
// columns (one channel per lane) or hardware-transposed fragments (ds_read_b64_tr_b16) from it.
#include "gemm_common.hpp"

namespace mopoe {

typedef __bf16 pw_bf16x8 __attribute__((ext_vector_type(8)));
typedef short pw_s16x4 __attribute__((ext_vector_type(4)));
typedef short pw_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) pw_s16x4 pw_lds_s16x4;

struct PwArgs {
  const bf16_t* x;        // [R, C] the block's input
  const bf16_t* dh2;      // (bwd) [R, C] gradient of bn2's output with its ReLU mask applied (conv2's input gradient)
  bf16_t* out;            // fwd WRITE: a2 [R, C]; bwd: dh1 [R, C]
  const bf16_t* W;        // [C][C] conv1's packed weight (ci rows, co columns), bf16 copy
  const float* bias;      // [C] or null
  mopoe_bn_ref bn1, bn2;
  mopoe_mask_ref mask1;   // kind 0 | 1 (per sample and channel); rows_per_sample a multiple of 32
  double* stats_d1;       // fwd !WRITE: [2][C] += {sum d1, sum d1^2}
  const double* sums2;    // bwd: [2][C] {sum dh2, sum dh2 * xhat2} (complete)
  double* sums1;          // bwd: [2][C] += {sum dh1, sum dh1 * xhat1}
  float* dW;              // bwd: [C][C] += h1^T dc1
  float* dbias;           // bwd: [C] += column sums of dc1, or null
  float* dgamma2;         // bwd: [C] = sum dh2 xhat2, [C] = sum dh2 (the gradients of bn2's affine parameters), or null
  float* dbeta2;
  long R;
  int ntiles;
  unsigned x_bytes;
};

__device__ __forceinline__ unsigned pw_bn_relu2(unsigned u, float s0, float s1, float t0, float t1) {
  const float a = fmaxf(fmaf(bf16_lo(u), s0, t0), 0.f), b = fmaxf(fmaf(bf16_hi(u), s1, t1), 0.f);
  return pack_bf16(a, b);
}
// sigma: row m of the A operand of output tile t <-> channel
__device__ __forceinline__ int pw_sigma(int t, int m) {
  const int q = m >> 3, h = (m >> 2) & 1, e = m & 3;
  return 32 * t + 16 * (q >> 1) + 8 * h + 4 * (q & 1) + e;
}
// swizzle of the [32][C] LDS patches (C = 64: 128-byte rows, 8 sixteen-byte slots): slot' = slot ^ sw(p).  Bit 2 moves by bit 1
// of the row (the four rows of a transposed read land in four quarters of the 256-byte bank row), the low two bits by
// (p & 1) + 2 ((p >> 2) & 1) (eight consecutive rows of a ds_write_b128 lane group land in eight different slots).
__device__ __forceinline__ int pw_sw64(int p) { return (((p >> 1) & 1) << 2) ^ ((p & 1) | (((p >> 2) & 1) << 1)); }

// ---------------------------------------------------------------------------------------------------------------------
// forward: statistics pass (WRITE = false) / a2 pass (WRITE = true)
// ---------------------------------------------------------------------------------------------------------------------
template <int C, bool WRITE>
__global__ __launch_bounds__(512, 2) void pw_front_fwd_bf16_kernel(const PwArgs a) {
  static_assert(C == 64, "patch swizzle and lane <-> channel maps are written for 64 channels");
  constexpr int NG = C / 16, NT = C / 32, NW = 8;
  constexpr int IMG_BYTES = NT * NG * 1024;            // A-operand fragments of the weight, one 1-KB piece each
  constexpr int OFF_TAB = IMG_BYTES;                   // scale1, shift1, bias, scale2, shift2: 5 x C floats
  constexpr int OFF_PATCH = OFF_TAB + 5 * C * 4;
  constexpr int PATCH_BYTES = 32 * C * 2;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[OFF_PATCH + (WRITE ? 0 : NW * PATCH_BYTES)];
  float* const tab = reinterpret_cast<float*>(smem + OFF_TAB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;

  for (int c = tid; c < C; c += 512) {
    const BnC k1 = bn_coef(a.bn1, c);
    tab[c] = k1.scale; tab[C + c] = k1.shift;
    tab[2 * C + c] = a.bias ? a.bias[c] : 0.f;
    if (WRITE) { const BnC k2 = bn_coef(a.bn2, c); tab[3 * C + c] = k2.scale; tab[4 * C + c] = k2.shift; }
  }
  // fragment (t, s), lane (m, hh): W[ci = 16 s + 8 hh + j][co = sigma_t(m)], j = 0..7
  for (int idx = tid; idx < NT * NG * 64; idx += 512) {
    const int f = idx >> 6, ln = idx & 63, t = f / NG, s = f % NG, m = ln & 31, hh = ln >> 5;
    const int co = pw_sigma(t, m);
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = 16 * s + 8 * hh + 2 * j;
      w[j] = (unsigned)a.W[ci * C + co] | ((unsigned)a.W[(ci + 1) * C + co] << 16);
    }
    *reinterpret_cast<uint4*>(smem + idx * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
  unsigned char* const patch = smem + OFF_PATCH + (WRITE ? 0 : wave * PATCH_BYTES);
  double S1 = 0.0, S2 = 0.0;      // (!WRITE) running sums of channel `lane`

  auto load_x = [&](int tile, u32x4 (&xr)[NG]) {
    const long row = (long)tile * 32 + p;
    const unsigned base = row < a.R ? (unsigned)(row * C + 8 * h) * 2u : OOB;
#pragma unroll
    for (int s = 0; s < NG; ++s) xr[s] = __builtin_amdgcn_raw_buffer_load_b128(srdX, base, (unsigned)(32 * s), 0);
  };

  const int stride = gridDim.x * NW;
  int tile = blockIdx.x * NW + wave;
  u32x4 xc[NG], xn[NG];
  if (tile < a.ntiles) load_x(tile, xc);
  for (; tile < a.ntiles; tile += stride) {
    const bool more = tile + stride < a.ntiles;
    if (more) load_x(tile + stride, xn);
    const long row = (long)tile * 32 + p;
    const bool ok = row < a.R;
    // Dropout2d multipliers of this tile's sample (32 | rows_per_sample: one sample per tile)
    const float* mrow = a.mask1.kind == 1 ? a.mask1.mask + ((long)tile * 32 / a.mask1.rows_per_sample) * C : nullptr;
    // h1 = relu(bn1(x)) as the B operand of k-step s
    pw_bf16x8 hb[NG];
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const int c0 = 16 * s + 8 * h;
      const float4 sc0 = *reinterpret_cast<const float4*>(&tab[c0]), sc1 = *reinterpret_cast<const float4*>(&tab[c0 + 4]);
      const float4 sh0 = *reinterpret_cast<const float4*>(&tab[C + c0]), sh1 = *reinterpret_cast<const float4*>(&tab[C + c0 + 4]);
      u32x4 r;
      r.x = pw_bn_relu2(xc[s].x, sc0.x, sc0.y, sh0.x, sh0.y);
      r.y = pw_bn_relu2(xc[s].y, sc0.z, sc0.w, sh0.z, sh0.w);
      r.z = pw_bn_relu2(xc[s].z, sc1.x, sc1.y, sh1.x, sh1.y);
      r.w = pw_bn_relu2(xc[s].w, sc1.z, sc1.w, sh1.z, sh1.w);
      if (!ok) r = u32x4{0u, 0u, 0u, 0u};
      hb[s] = __builtin_bit_cast(pw_bf16x8, r);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        const pw_bf16x8 wa = *reinterpret_cast<const pw_bf16x8*>(smem + ((t * NG + s) * 64 + lane) * 16);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hb[s], acc[t], 0, 0, 0);
      }
    }
    // groups: d1 = round(mask * (acc + bias)); registers 8 (s & 1) .. + 7 of tile s / 2 are channels 16 s + 8 h + 0..7
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const int c0 = 16 * s + 8 * h;
      float v[8];
      const float4 b0 = *reinterpret_cast<const float4*>(&tab[2 * C + c0]), b1 = *reinterpret_cast<const float4*>(&tab[2 * C + c0 + 4]);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = acc[s >> 1][8 * (s & 1) + j] + bb[j];
      if (mrow) {
        const float4 m0 = *reinterpret_cast<const float4*>(mrow + c0), m1 = *reinterpret_cast<const float4*>(mrow + c0 + 4);
        v[0] *= m0.x; v[1] *= m0.y; v[2] *= m0.z; v[3] *= m0.w; v[4] *= m1.x; v[5] *= m1.y; v[6] *= m1.z; v[7] *= m1.w;
      }
      u32x4 d;      // d1, rounded to its storage type (the tensor round 3 kept in HBM)
      d.x = pack_bf16(v[0], v[1]); d.y = pack_bf16(v[2], v[3]); d.z = pack_bf16(v[4], v[5]); d.w = pack_bf16(v[6], v[7]);
      if (!WRITE) {
        if (!ok) d = u32x4{0u, 0u, 0u, 0u};
        const int slot = (2 * s + h) ^ pw_sw64(p);
        *reinterpret_cast<u32x4*>(patch + p * 128 + slot * 16) = d;
      } else {
        const float4 s0 = *reinterpret_cast<const float4*>(&tab[3 * C + c0]), s1 = *reinterpret_cast<const float4*>(&tab[3 * C + c0 + 4]);
        const float4 t0 = *reinterpret_cast<const float4*>(&tab[4 * C + c0]), t1 = *reinterpret_cast<const float4*>(&tab[4 * C + c0 + 4]);
        u32x4 o;
        o.x = pw_bn_relu2(d.x, s0.x, s0.y, t0.x, t0.y);
        o.y = pw_bn_relu2(d.y, s0.z, s0.w, t0.z, t0.w);
        o.z = pw_bn_relu2(d.z, s1.x, s1.y, t1.x, t1.y);
        o.w = pw_bn_relu2(d.w, s1.z, s1.w, t1.z, t1.w);
        if (ok) *reinterpret_cast<u32x4*>(a.out + row * C + c0) = o;
      }
    }
    if (!WRITE) {
      // column sums of this tile's d1: lane = channel
      const int c = lane;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
      for (int pp = 0; pp < 32; ++pp) {
        const int slot = (c >> 3) ^ pw_sw64(pp);
        const float v = bf16_to_f32(*reinterpret_cast<const bf16_t*>(patch + pp * 128 + slot * 16 + (c & 7) * 2));
        s1 += v;
        s2 = fmaf(v, v, s2);
      }
      S1 += (double)s1;
      S2 += (double)s2;
    }
    if (more) {
#pragma unroll
      for (int s = 0; s < NG; ++s) xc[s] = xn[s];
    }
  }
  if (!WRITE) {
    // block reduction (8 waves -> one atomic per channel and block: same-address atomics serialise at ~18 ns each)
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem);     // (the weight image is dead)
    red[(wave * 2 + 0) * C + lane] = S1;
    red[(wave * 2 + 1) * C + lane] = S2;
    __syncthreads();
    if (tid < 2 * C) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[(w * 2 + tid / C) * C + tid % C];
      atomic_add_f64(a.stats_d1 + tid, s);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward: dh2, x -> dh1, sums1, dW1, dbias
// ---------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512, 2) void pw_front_bwd_bf16_kernel(const PwArgs a) {
  static_assert(C == 64, "patch swizzle and lane <-> channel maps are written for 64 channels");
  constexpr int NG = C / 16, NT = C / 32, NW = 8;
  constexpr int IMG_BYTES = NT * NG * 1024;
  constexpr int OFF_IMG2 = IMG_BYTES;                  // A fragments of the second GEMM (rows = ci, k = co)
  constexpr int OFF_TAB = 2 * IMG_BYTES;               // scale1, shift1, bias, ga, gb, gc: 6 x C floats
  constexpr int OFF_PATCH = OFF_TAB + 6 * C * 4;
  constexpr int PATCH_BYTES = 32 * C * 2;              // per wave: three patches (x, h1, dc1 / dh1)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[OFF_PATCH + NW * 3 * PATCH_BYTES];
  float* const tab = reinterpret_cast<float*>(smem + OFF_TAB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;

  for (int c = tid; c < C; c += 512) {
    const BnC k1 = bn_coef(a.bn1, c);
    const BnC k2 = bn_coef(a.bn2, c);
    tab[c] = k1.scale; tab[C + c] = k1.shift;
    tab[2 * C + c] = a.bias ? a.bias[c] : 0.f;
    // dc1 = mask * ga * (dh2 - c1 - xhat2 * c2), xhat2 = (d1 - mean2) * rstd2, c1 = sum dh2 / n, c2 = sum dh2 xhat2 / n
    //     = mask * (ga * dh2 + gb + gc * d1)
    // (running statistics -- a backward in eval mode -- : bn2 is a fixed affine map, no batch terms)
    const double c1 = a.bn2.mode == 1 ? a.sums2[c] * a.bn2.inv_count : 0.0, c2 = a.bn2.mode == 1 ? a.sums2[C + c] * a.bn2.inv_count : 0.0;
    const float ga = k2.scale;                                   // gamma2 * rstd2
    tab[3 * C + c] = ga;
    tab[4 * C + c] = (float)(-(double)ga * c1 + (double)ga * c2 * (double)k2.rstd * (double)k2.mean);
    tab[5 * C + c] = (float)(-(double)ga * c2 * (double)k2.rstd);
  }
  for (int idx = tid; idx < NT * NG * 64; idx += 512) {
    const int f = idx >> 6, ln = idx & 63, t = f / NG, s = f % NG, m = ln & 31, hh = ln >> 5;
    const int cs = pw_sigma(t, m);
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {      // first GEMM: W[ci = 16 s + 8 hh + 2 j (+ 1)][co = sigma]
      const int ci = 16 * s + 8 * hh + 2 * j;
      w[j] = (unsigned)a.W[ci * C + cs] | ((unsigned)a.W[(ci + 1) * C + cs] << 16);
    }
    *reinterpret_cast<uint4*>(smem + idx * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    // second GEMM: W[ci = sigma][co = 16 s + 8 hh + j]: eight consecutive columns of one row
    *reinterpret_cast<uint4*>(smem + OFF_IMG2 + idx * 16) = *reinterpret_cast<const uint4*>(a.W + cs * C + 16 * s + 8 * hh);
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdD = __builtin_amdgcn_make_buffer_rsrc((void*)a.dh2, 0, (int)a.x_bytes, 0x00020000);
  unsigned char* const pX = smem + OFF_PATCH + wave * 3 * PATCH_BYTES;   // raw x           (column sums: sum dh1 * x)
  unsigned char* const pH = pX + PATCH_BYTES;                            // h1              (weight gradient, rows of the A operand)
  unsigned char* const pD = pH + PATCH_BYTES;                            // dc1, then dh1   (weight gradient B operand; column sums)

  // transposed-fragment addresses: element j of lane l = patch[k0 + 8 (l / 32) + j][c0 + l % 32]; this lane reads 8 bytes of row
  // 16 ks + 8 h + q4 (+ 4) at column c0 + 16 g16 + 4 (i16 & 3).  The swizzle depends on the row: one address per (k-step, half)
  const int i16 = lane & 15, g16 = (lane >> 4) & 1, q4 = i16 >> 2;
  unsigned troff[NT][2][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int row = 16 * ks + 8 * h + q4 + 4 * hf;
        const int colb = (32 * t + 16 * g16 + 4 * (i16 & 3)) * 2;
        troff[t][ks][hf] = (unsigned)(row * 128 + ((((colb >> 4) ^ pw_sw64(row)) << 4) | (colb & 15)));
      }

  f32x16 wacc[NT][NT];     // dW tile [ci tile][co tile]
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) wacc[i][j][r] = 0.f;
  double Sg = 0.0, Sgx = 0.0, Sb = 0.0;     // channel `lane`: sum dh1, sum dh1 * x, sum dc1

  auto load_rows = [&](const __amdgpu_buffer_rsrc_t& srd, int tile, u32x4 (&r)[NG]) {
    const long row = (long)tile * 32 + p;
    const unsigned base = row < a.R ? (unsigned)(row * C + 8 * h) * 2u : OOB;
#pragma unroll
    for (int s = 0; s < NG; ++s) r[s] = __builtin_amdgcn_raw_buffer_load_b128(srd, base, (unsigned)(32 * s), 0);
  };

  const int stride = gridDim.x * NW;
  int tile = blockIdx.x * NW + wave;
  u32x4 xr[NG], gr[NG], xn[NG], gn[NG];
  if (tile < a.ntiles) { load_rows(srdX, tile, xr); load_rows(srdD, tile, gr); }
  for (; tile < a.ntiles; tile += stride) {
    // the next tile's rows are requested before this tile's arithmetic: the wave's share of the HBM stream stays in flight
    const bool more = tile + stride < a.ntiles;
    if (more) { load_rows(srdX, tile + stride, xn); load_rows(srdD, tile + stride, gn); }
    const long row = (long)tile * 32 + p;
    const bool ok = row < a.R;
    const float* mrow = a.mask1.kind == 1 ? a.mask1.mask + ((long)tile * 32 / a.mask1.rows_per_sample) * C : nullptr;
    pw_bf16x8 hb[NG];
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const int c0 = 16 * s + 8 * h;
      const float4 sc0 = *reinterpret_cast<const float4*>(&tab[c0]), sc1 = *reinterpret_cast<const float4*>(&tab[c0 + 4]);
      const float4 sh0 = *reinterpret_cast<const float4*>(&tab[C + c0]), sh1 = *reinterpret_cast<const float4*>(&tab[C + c0 + 4]);
      u32x4 r;
      r.x = pw_bn_relu2(xr[s].x, sc0.x, sc0.y, sh0.x, sh0.y);
      r.y = pw_bn_relu2(xr[s].y, sc0.z, sc0.w, sh0.z, sh0.w);
      r.z = pw_bn_relu2(xr[s].z, sc1.x, sc1.y, sh1.x, sh1.y);
      r.w = pw_bn_relu2(xr[s].w, sc1.z, sc1.w, sh1.z, sh1.w);
      if (!ok) r = u32x4{0u, 0u, 0u, 0u};
      hb[s] = __builtin_bit_cast(pw_bf16x8, r);
      const int slot = (2 * s + h) ^ pw_sw64(p);
      *reinterpret_cast<u32x4*>(pX + p * 128 + slot * 16) = ok ? xr[s] : u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(pH + p * 128 + slot * 16) = r;
    }
    // d1 again (the same instructions as the forward's two passes)
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        const pw_bf16x8 wa = *reinterpret_cast<const pw_bf16x8*>(smem + ((t * NG + s) * 64 + lane) * 16);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hb[s], acc[t], 0, 0, 0);
      }
    }
    pw_bf16x8 db[NG];      // dc1 as the B operand of the second GEMM's k-step s (k = co)
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const int c0 = 16 * s + 8 * h;
      float v[8], m8[8];
      const float4 b0 = *reinterpret_cast<const float4*>(&tab[2 * C + c0]), b1 = *reinterpret_cast<const float4*>(&tab[2 * C + c0 + 4]);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = acc[s >> 1][8 * (s & 1) + j] + bb[j]; m8[j] = 1.f; }
      if (mrow) {
        const float4 m0 = *reinterpret_cast<const float4*>(mrow + c0), m1 = *reinterpret_cast<const float4*>(mrow + c0 + 4);
        m8[0] = m0.x; m8[1] = m0.y; m8[2] = m0.z; m8[3] = m0.w; m8[4] = m1.x; m8[5] = m1.y; m8[6] = m1.z; m8[7] = m1.w;
      }
      const float4 ga0 = *reinterpret_cast<const float4*>(&tab[3 * C + c0]), ga1 = *reinterpret_cast<const float4*>(&tab[3 * C + c0 + 4]);
      const float4 gb0 = *reinterpret_cast<const float4*>(&tab[4 * C + c0]), gb1 = *reinterpret_cast<const float4*>(&tab[4 * C + c0 + 4]);
      const float4 gc0 = *reinterpret_cast<const float4*>(&tab[5 * C + c0]), gc1 = *reinterpret_cast<const float4*>(&tab[5 * C + c0 + 4]);
      const float ga[8] = {ga0.x, ga0.y, ga0.z, ga0.w, ga1.x, ga1.y, ga1.z, ga1.w};
      const float gb[8] = {gb0.x, gb0.y, gb0.z, gb0.w, gb1.x, gb1.y, gb1.z, gb1.w};
      const float gc[8] = {gc0.x, gc0.y, gc0.z, gc0.w, gc1.x, gc1.y, gc1.z, gc1.w};
      const unsigned gw[4] = {gr[s].x, gr[s].y, gr[s].z, gr[s].w};
      float dc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d1 = round_bf16(v[j] * m8[j]);                       // the stored d1 of round 3, bit for bit
        const float g = (j & 1) ? bf16_hi(gw[j >> 1]) : bf16_lo(gw[j >> 1]);
        dc[j] = ok ? m8[j] * fmaf(gc[j], d1, fmaf(ga[j], g, gb[j])) : 0.f;
      }
      u32x4 d;
      d.x = pack_bf16(dc[0], dc[1]); d.y = pack_bf16(dc[2], dc[3]); d.z = pack_bf16(dc[4], dc[5]); d.w = pack_bf16(dc[6], dc[7]);
      db[s] = __builtin_bit_cast(pw_bf16x8, d);
      const int slot = (2 * s + h) ^ pw_sw64(p);
      *reinterpret_cast<u32x4*>(pD + p * 128 + slot * 16) = d;
    }
    // dh1 = [h1 > 0] * (dc1 W^T): rows of the A operand = ci in sigma order, k = co
    f32x16 acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        const pw_bf16x8 wa = *reinterpret_cast<const pw_bf16x8*>(smem + OFF_IMG2 + ((t * NG + s) * 64 + lane) * 16);
        acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, db[s], acc2[t], 0, 0, 0);
      }
    }
    // weight gradient: dW[ci][co] += sum over the tile's 32 pixels of h1[p][ci] dc1[p][co]  (K = pixels: transposed fragments)
    // and the bias gradient (column sums of dc1) -- both before dh1 overwrites the dc1 patch
    {
      asm volatile("" ::: "memory");       // (the patch stores above are ordinary C++ stores: keep the transposed reads behind them)
      pw_bf16x8 fa[NT][2], fb[NT][2];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const pw_s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s16x4*)(pH + troff[t][ks][0]));
          const pw_s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s16x4*)(pH + troff[t][ks][1]));
          const pw_s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s16x4*)(pD + troff[t][ks][0]));
          const pw_s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s16x4*)(pD + troff[t][ks][1]));
          fa[t][ks] = __builtin_bit_cast(pw_bf16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
          fb[t][ks] = __builtin_bit_cast(pw_bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            wacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb[j][ks], wacc[i][j], 0, 0, 0);
      if (a.dbias) {
        const int c = lane;
        float sb = 0.f;
#pragma unroll 8
        for (int pp = 0; pp < 32; ++pp) {
          const int slot = (c >> 3) ^ pw_sw64(pp);
          sb += bf16_to_f32(*reinterpret_cast<const bf16_t*>(pD + pp * 128 + slot * 16 + (c & 7) * 2));
        }
        Sb += (double)sb;
      }
    }
    asm volatile("" ::: "memory");
    // dh1: mask by h1 > 0, round, store, and into the patch for the two column sums
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const int c0 = 16 * s + 8 * h;
      const u32x4 hr = __builtin_bit_cast(u32x4, hb[s]);
      const unsigned hw[4] = {hr.x, hr.y, hr.z, hr.w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned hbits = (j & 1) ? (hw[j >> 1] >> 16) : (hw[j >> 1] & 0xffffu);
        v[j] = (hbits != 0u && hbits != 0x8000u) ? acc2[s >> 1][8 * (s & 1) + j] : 0.f;     // relu(.) > 0
      }
      u32x4 o;
      o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
      if (!ok) o = u32x4{0u, 0u, 0u, 0u};
      if (ok) *reinterpret_cast<u32x4*>(a.out + row * C + c0) = o;
      const int slot = (2 * s + h) ^ pw_sw64(p);
      *reinterpret_cast<u32x4*>(pD + p * 128 + slot * 16) = o;
    }
    {
      const int c = lane;
      float sg = 0.f, sgx = 0.f;
#pragma unroll 8
      for (int pp = 0; pp < 32; ++pp) {
        const int slot = (c >> 3) ^ pw_sw64(pp);
        const int off = pp * 128 + slot * 16 + (c & 7) * 2;
        const float g = bf16_to_f32(*reinterpret_cast<const bf16_t*>(pD + off));
        const float xv = bf16_to_f32(*reinterpret_cast<const bf16_t*>(pX + off));
        sg += g;
        sgx = fmaf(g, xv, sgx);
      }
      Sg += (double)sg;
      Sgx += (double)sgx;
    }
    if (more) {
#pragma unroll
      for (int s = 0; s < NG; ++s) { xr[s] = xn[s]; gr[s] = gn[s]; }
    }
  }

  // ---- block reductions, then one set of atomics per block
  __syncthreads();
  {
    double* red = reinterpret_cast<double*>(smem + OFF_PATCH);        // (patches are dead) [NW][3][C]
    red[(wave * 3 + 0) * C + lane] = Sg;
    red[(wave * 3 + 1) * C + lane] = Sgx;
    red[(wave * 3 + 2) * C + lane] = Sb;
    __syncthreads();
    if (tid < C) {
      double g = 0.0, gx = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { g += red[(w * 3 + 0) * C + tid]; gx += red[(w * 3 + 1) * C + tid]; b += red[(w * 3 + 2) * C + tid]; }
      // sum dh1 * xhat1 = rstd1 * (sum dh1 x - mean1 * sum dh1)
      const BnC k1 = bn_coef(a.bn1, tid);
      atomic_add_f64(a.sums1 + tid, g);
      atomic_add_f64(a.sums1 + C + tid, (double)k1.rstd * (gx - (double)k1.mean * g));
      if (a.dbias) unsafeAtomicAdd(a.dbias + tid, (float)b);
      if (blockIdx.x == 0 && a.dgamma2) { a.dgamma2[tid] = (float)a.sums2[C + tid]; a.dbeta2[tid] = (float)a.sums2[tid]; }
    }
    __syncthreads();
  }
  // weight gradient: the eight waves' tiles summed through LDS (two rounds of four waves), then atomics
  {
    float* wred = reinterpret_cast<float*>(smem + OFF_PATCH);          // [4][C][C] floats = 64 KB
    const int l31 = lane & 31;
    for (int round = 0; round < 2; ++round) {
      if ((wave >> 2) == round) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ci = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h, co = 32 * j + l31;
              float* dst = wred + ((wave & 3) * C + ci) * C + co;
              if (round == 0) *dst = wacc[i][j][r]; else *dst += wacc[i][j][r];
            }
      }
      __syncthreads();
    }
    for (int idx = tid; idx < C * C; idx += 512) {
      const float s = wred[idx] + wred[C * C + idx] + wred[2 * C * C + idx] + wred[3 * C * C + idx];
      unsafeAtomicAdd(a.dW + idx, s);
    }
  }
}

// =====================================================================================================================
// fp32 family (BASELINE configs #2, #4): the same three kernels on v_mfma_f32_32x32x2_f32 (exact fp32 products and sums).
//
// Lane (p, h) owns, for g = 0..7, the 16-byte piece of pixel p's row that holds channels 8 g + 4 h + e (e = 0..3): 32 values,
// value u = 4 g + e.  That IS the B operand of MFMA number u (k = h), and the accumulator of output tile t returns register
// r = 4 q + e as channel 32 t + 8 q + 4 h + e, i.e. value u = 4 (4 t + q) + e of the same layout -- no row permutation needed.
// A MFMA does 64 cycles of work for 2 k: a 64 x 64 product per 32 pixels is 64 MFMAs = 4096 cycles per wave, so with the
// three products of the backward pass the matrix pipe and the HBM stream take about the same time (7.4 against 8.1 bytes per
// clock and CU): these kernels run one wave per SIMD with the next tile's rows in flight, and 350 registers.
// LDS patches are fp32 [32][64] (256-byte rows, 16-byte slot ^ (p & 7)); fragments of the weight gradient are plain
// ds_read_b32 (lane = channel: no transposition instruction is needed in fp32).
// =====================================================================================================================
struct PwArgsF {
  const float* x; const float* dh2; float* out; const float* W; const float* bias;
  mopoe_bn_ref bn1, bn2;
  mopoe_mask_ref mask1;
  double* stats_d1; const double* sums2; double* sums1; float* dW; float* dbias; float* dgamma2; float* dbeta2;
  long R; int ntiles; unsigned x_bytes;
};

template <bool WRITE>
__global__ __launch_bounds__(256, 1) void pw_front_fwd_f32_kernel(const PwArgsF a) {
  constexpr int C = 64, NP = 8, NT = 2, NW = 4;       // pieces per lane, output tiles, waves
  constexpr int IMG_BYTES = NT * 8 * 1024;            // A operands: per tile 32 MFMAs = 8 groups of 4, 1 KB per group
  constexpr int OFF_TAB = IMG_BYTES;                  // scale1, shift1, bias, scale2, shift2
  constexpr int OFF_PATCH = OFF_TAB + 5 * C * 4;
  constexpr int PATCH_BYTES = 32 * C * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[OFF_PATCH + (WRITE ? 0 : NW * PATCH_BYTES)];
  float* const tab = reinterpret_cast<float*>(smem + OFF_TAB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  for (int c = tid; c < C; c += 256) {
    const BnC k1 = bn_coef(a.bn1, c);
    tab[c] = k1.scale; tab[C + c] = k1.shift;
    tab[2 * C + c] = a.bias ? a.bias[c] : 0.f;
    if (WRITE) { const BnC k2 = bn_coef(a.bn2, c); tab[3 * C + c] = k2.scale; tab[4 * C + c] = k2.shift; }
  }
  // group (t, gq) = MFMAs u = 4 gq .. 4 gq + 3 of tile t; lane (m, kh) holds W[ci = 8 gq + 4 kh + e][co = 32 t + m], e = 0..3
  for (int idx = tid; idx < NT * 8 * 64; idx += 256) {
    const int grp = idx >> 6, ln = idx & 63, t = grp >> 3, gq = grp & 7, m = ln & 31, kh = ln >> 5;
    float4 w;
    const int ci = 8 * gq + 4 * kh, co = 32 * t + m;
    w.x = a.W[(ci + 0) * C + co]; w.y = a.W[(ci + 1) * C + co]; w.z = a.W[(ci + 2) * C + co]; w.w = a.W[(ci + 3) * C + co];
    *reinterpret_cast<float4*>(smem + idx * 16) = w;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
  unsigned char* const patch = smem + OFF_PATCH + (WRITE ? 0 : wave * PATCH_BYTES);
  double S1 = 0.0, S2 = 0.0;
  auto load_x = [&](int tile, float4 (&xr)[NP]) {
    const long row = (long)tile * 32 + p;
    const unsigned base = row < a.R ? (unsigned)(row * C + 4 * h) * 4u : OOB;
#pragma unroll
    for (int g = 0; g < NP; ++g) xr[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srdX, base, (unsigned)(32 * g), 0));
  };
  const int stride = gridDim.x * NW;
  int tile = blockIdx.x * NW + wave;
  float4 xc[NP], xn[NP];
  if (tile < a.ntiles) load_x(tile, xc);
  for (; tile < a.ntiles; tile += stride) {
    const bool more = tile + stride < a.ntiles;
    if (more) load_x(tile + stride, xn);
    const long row = (long)tile * 32 + p;
    const bool ok = row < a.R;
    const float* mrow = a.mask1.kind == 1 ? a.mask1.mask + ((long)tile * 32 / a.mask1.rows_per_sample) * C : nullptr;
    float hv[32];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      const int c0 = 8 * g + 4 * h;
      const float4 sc = *reinterpret_cast<const float4*>(&tab[c0]), sh = *reinterpret_cast<const float4*>(&tab[C + c0]);
      hv[4 * g + 0] = ok ? fmaxf(fmaf(xc[g].x, sc.x, sh.x), 0.f) : 0.f;
      hv[4 * g + 1] = ok ? fmaxf(fmaf(xc[g].y, sc.y, sh.y), 0.f) : 0.f;
      hv[4 * g + 2] = ok ? fmaxf(fmaf(xc[g].z, sc.z, sh.z), 0.f) : 0.f;
      hv[4 * g + 3] = ok ? fmaxf(fmaf(xc[g].w, sc.w, sh.w), 0.f) : 0.f;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 w = *reinterpret_cast<const float4*>(smem + ((t * 8 + gq) * 64 + lane) * 16);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, hv[4 * gq + 0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, hv[4 * gq + 1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, hv[4 * gq + 2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, hv[4 * gq + 3], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < NP; ++g) {       // piece g = 4 t + q: registers 4 q .. 4 q + 3 of tile t
      const int c0 = 8 * g + 4 * h, t = g >> 2, q = g & 3;
      const float4 bb = *reinterpret_cast<const float4*>(&tab[2 * C + c0]);
      float4 d;
      d.x = acc[t][4 * q + 0] + bb.x; d.y = acc[t][4 * q + 1] + bb.y; d.z = acc[t][4 * q + 2] + bb.z; d.w = acc[t][4 * q + 3] + bb.w;
      if (mrow) { const float4 m = *reinterpret_cast<const float4*>(mrow + c0); d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w; }
      if (!WRITE) {
        if (!ok) d = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(patch + p * 256 + (((2 * g + h) ^ (p & 7)) << 4)) = d;
      } else {
        const float4 s2 = *reinterpret_cast<const float4*>(&tab[3 * C + c0]), t2 = *reinterpret_cast<const float4*>(&tab[4 * C + c0]);
        float4 o;
        o.x = fmaxf(fmaf(d.x, s2.x, t2.x), 0.f); o.y = fmaxf(fmaf(d.y, s2.y, t2.y), 0.f);
        o.z = fmaxf(fmaf(d.z, s2.z, t2.z), 0.f); o.w = fmaxf(fmaf(d.w, s2.w, t2.w), 0.f);
        if (ok) *reinterpret_cast<float4*>(a.out + row * C + c0) = o;
      }
    }
    if (!WRITE) {
      const int c = lane;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
      for (int pp = 0; pp < 32; ++pp) {
        const float v = *reinterpret_cast<const float*>(patch + pp * 256 + ((((c >> 2) ^ (pp & 7)) << 4) | ((c & 3) << 2)));
        s1 += v;
        s2 = fmaf(v, v, s2);
      }
      S1 += (double)s1;
      S2 += (double)s2;
    }
    if (more) {
#pragma unroll
      for (int g = 0; g < NP; ++g) xc[g] = xn[g];
    }
  }
  if (!WRITE) {
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem);
    red[(wave * 2 + 0) * C + lane] = S1;
    red[(wave * 2 + 1) * C + lane] = S2;
    __syncthreads();
    if (tid < 2 * C) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[(w * 2 + tid / C) * C + tid % C];
      atomic_add_f64(a.stats_d1 + tid, s);
    }
  }
}

__global__ __launch_bounds__(256, 1) void pw_front_bwd_f32_kernel(const PwArgsF a) {
  constexpr int C = 64, NP = 8, NT = 2, NW = 4;
  constexpr int IMG_BYTES = NT * 8 * 1024;
  constexpr int OFF_IMG2 = IMG_BYTES;
  constexpr int OFF_TAB = 2 * IMG_BYTES;              // scale1, shift1, bias, ga, gb, gc, xh_a, xh_b: 8 x C floats
  constexpr int OFF_PATCH = OFF_TAB + 8 * C * 4;
  constexpr int PATCH_BYTES = 32 * C * 4;             // per wave: h1, dc1 / dh1
  __shared__ __attribute__((aligned(1024))) unsigned char smem[OFF_PATCH + NW * 2 * PATCH_BYTES];
  float* const tab = reinterpret_cast<float*>(smem + OFF_TAB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  for (int c = tid; c < C; c += 256) {
    const BnC k1 = bn_coef(a.bn1, c);
    const BnC k2 = bn_coef(a.bn2, c);
    tab[c] = k1.scale; tab[C + c] = k1.shift;
    tab[2 * C + c] = a.bias ? a.bias[c] : 0.f;
    // (running statistics -- a backward in eval mode -- : bn2 is a fixed affine map, no batch terms)
    const double c1 = a.bn2.mode == 1 ? a.sums2[c] * a.bn2.inv_count : 0.0, c2 = a.bn2.mode == 1 ? a.sums2[C + c] * a.bn2.inv_count : 0.0;
    const float ga = k2.scale;
    tab[3 * C + c] = ga;
    tab[4 * C + c] = (float)(-(double)ga * c1 + (double)ga * c2 * (double)k2.rstd * (double)k2.mean);
    tab[5 * C + c] = (float)(-(double)ga * c2 * (double)k2.rstd);
    // xhat1 = (x - mean1) rstd1 = (h1 - beta1) / gamma1 wherever h1 > 0 (the only places dh1 is not zero): xh_a h1 + xh_b
    const float g1 = a.bn1.gamma[c], b1 = a.bn1.beta[c];
    tab[6 * C + c] = g1 != 0.f ? 1.0f / g1 : 0.f;
    tab[7 * C + c] = g1 != 0.f ? -b1 / g1 : 0.f;
  }
  for (int idx = tid; idx < NT * 8 * 64; idx += 256) {
    const int grp = idx >> 6, ln = idx & 63, t = grp >> 3, gq = grp & 7, m = ln & 31, kh = ln >> 5;
    const int k0 = 8 * gq + 4 * kh, mm = 32 * t + m;
    float4 w1, w2;      // first product: W[ci = k0 + e][co = mm]; second: W[ci = mm][co = k0 + e]
    w1.x = a.W[(k0 + 0) * C + mm]; w1.y = a.W[(k0 + 1) * C + mm]; w1.z = a.W[(k0 + 2) * C + mm]; w1.w = a.W[(k0 + 3) * C + mm];
    w2 = *reinterpret_cast<const float4*>(a.W + mm * C + k0);
    *reinterpret_cast<float4*>(smem + idx * 16) = w1;
    *reinterpret_cast<float4*>(smem + OFF_IMG2 + idx * 16) = w2;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdD = __builtin_amdgcn_make_buffer_rsrc((void*)a.dh2, 0, (int)a.x_bytes, 0x00020000);
  unsigned char* const pH = smem + OFF_PATCH + wave * 2 * PATCH_BYTES;
  unsigned char* const pD = pH + PATCH_BYTES;
  f32x16 wacc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) wacc[i][j][r] = 0.f;
  double Sg = 0.0, Sgx = 0.0, Sb = 0.0;
  auto load_rows = [&](const __amdgpu_buffer_rsrc_t& srd, int tile, float4 (&r)[NP]) {
    const long row = (long)tile * 32 + p;
    const unsigned base = row < a.R ? (unsigned)(row * C + 4 * h) * 4u : OOB;
#pragma unroll
    for (int g = 0; g < NP; ++g) r[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srd, base, (unsigned)(32 * g), 0));
  };
  const int stride = gridDim.x * NW;
  int tile = blockIdx.x * NW + wave;
  float4 xr[NP], gr[NP], xn[NP], gn[NP];
  if (tile < a.ntiles) { load_rows(srdX, tile, xr); load_rows(srdD, tile, gr); }
  for (; tile < a.ntiles; tile += stride) {
    const bool more = tile + stride < a.ntiles;
    if (more) { load_rows(srdX, tile + stride, xn); load_rows(srdD, tile + stride, gn); }
    const long row = (long)tile * 32 + p;
    const bool ok = row < a.R;
    const float* mrow = a.mask1.kind == 1 ? a.mask1.mask + ((long)tile * 32 / a.mask1.rows_per_sample) * C : nullptr;
    float hv[32];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      const int c0 = 8 * g + 4 * h;
      const float4 sc = *reinterpret_cast<const float4*>(&tab[c0]), sh = *reinterpret_cast<const float4*>(&tab[C + c0]);
      hv[4 * g + 0] = ok ? fmaxf(fmaf(xr[g].x, sc.x, sh.x), 0.f) : 0.f;
      hv[4 * g + 1] = ok ? fmaxf(fmaf(xr[g].y, sc.y, sh.y), 0.f) : 0.f;
      hv[4 * g + 2] = ok ? fmaxf(fmaf(xr[g].z, sc.z, sh.z), 0.f) : 0.f;
      hv[4 * g + 3] = ok ? fmaxf(fmaf(xr[g].w, sc.w, sh.w), 0.f) : 0.f;
      *reinterpret_cast<float4*>(pH + p * 256 + (((2 * g + h) ^ (p & 7)) << 4)) = make_float4(hv[4 * g], hv[4 * g + 1], hv[4 * g + 2], hv[4 * g + 3]);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 w = *reinterpret_cast<const float4*>(smem + ((t * 8 + gq) * 64 + lane) * 16);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, hv[4 * gq + 0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, hv[4 * gq + 1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, hv[4 * gq + 2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, hv[4 * gq + 3], acc[t], 0, 0, 0);
      }
    }
    float dcv[32];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      const int c0 = 8 * g + 4 * h, t = g >> 2, q = g & 3;
      const float4 bb = *reinterpret_cast<const float4*>(&tab[2 * C + c0]);
      float4 m4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (mrow) m4 = *reinterpret_cast<const float4*>(mrow + c0);
      const float4 ga = *reinterpret_cast<const float4*>(&tab[3 * C + c0]), gb = *reinterpret_cast<const float4*>(&tab[4 * C + c0]);
      const float4 gc = *reinterpret_cast<const float4*>(&tab[5 * C + c0]);
      const float d0 = (acc[t][4 * q + 0] + bb.x) * m4.x, d1 = (acc[t][4 * q + 1] + bb.y) * m4.y;
      const float d2 = (acc[t][4 * q + 2] + bb.z) * m4.z, d3 = (acc[t][4 * q + 3] + bb.w) * m4.w;
      dcv[4 * g + 0] = ok ? m4.x * fmaf(gc.x, d0, fmaf(ga.x, gr[g].x, gb.x)) : 0.f;
      dcv[4 * g + 1] = ok ? m4.y * fmaf(gc.y, d1, fmaf(ga.y, gr[g].y, gb.y)) : 0.f;
      dcv[4 * g + 2] = ok ? m4.z * fmaf(gc.z, d2, fmaf(ga.z, gr[g].z, gb.z)) : 0.f;
      dcv[4 * g + 3] = ok ? m4.w * fmaf(gc.w, d3, fmaf(ga.w, gr[g].w, gb.w)) : 0.f;
      *reinterpret_cast<float4*>(pD + p * 256 + (((2 * g + h) ^ (p & 7)) << 4)) = make_float4(dcv[4 * g], dcv[4 * g + 1], dcv[4 * g + 2], dcv[4 * g + 3]);
    }
    f32x16 acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 w = *reinterpret_cast<const float4*>(smem + OFF_IMG2 + ((t * 8 + gq) * 64 + lane) * 16);
        acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, dcv[4 * gq + 0], acc2[t], 0, 0, 0);
        acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, dcv[4 * gq + 1], acc2[t], 0, 0, 0);
        acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, dcv[4 * gq + 2], acc2[t], 0, 0, 0);
        acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, dcv[4 * gq + 3], acc2[t], 0, 0, 0);
      }
    }
    // weight gradient (K = the tile's 32 pixels, 2 per MFMA): lane (channel, kh) reads pixel 2 v + kh of the two patches
    {
      asm volatile("" ::: "memory");
      const int m = lane & 31, kh = lane >> 5;
#pragma unroll 4
      for (int v = 0; v < 16; ++v) {
        const int pr = 2 * v + kh;
        float av[NT], bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int c = 32 * t + m;
          const int off = pr * 256 + ((((c >> 2) ^ (pr & 7)) << 4) | ((c & 3) << 2));
          av[t] = *reinterpret_cast<const float*>(pH + off);
          bv[t] = *reinterpret_cast<const float*>(pD + off);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) wacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], wacc[i][j], 0, 0, 0);
      }
      if (a.dbias) {
        const int c = lane;
        float sb = 0.f;
#pragma unroll 8
        for (int pp = 0; pp < 32; ++pp) sb += *reinterpret_cast<const float*>(pD + pp * 256 + ((((c >> 2) ^ (pp & 7)) << 4) | ((c & 3) << 2)));
        Sb += (double)sb;
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      const int c0 = 8 * g + 4 * h, t = g >> 2, q = g & 3;
      float4 o;
      o.x = hv[4 * g + 0] > 0.f ? acc2[t][4 * q + 0] : 0.f; o.y = hv[4 * g + 1] > 0.f ? acc2[t][4 * q + 1] : 0.f;
      o.z = hv[4 * g + 2] > 0.f ? acc2[t][4 * q + 2] : 0.f; o.w = hv[4 * g + 3] > 0.f ? acc2[t][4 * q + 3] : 0.f;
      if (ok) *reinterpret_cast<float4*>(a.out + row * C + c0) = o;
      *reinterpret_cast<float4*>(pD + p * 256 + (((2 * g + h) ^ (p & 7)) << 4)) = o;
    }
    {
      const int c = lane;
      float sg = 0.f, sgh = 0.f;
#pragma unroll 8
      for (int pp = 0; pp < 32; ++pp) {
        const int off = pp * 256 + ((((c >> 2) ^ (pp & 7)) << 4) | ((c & 3) << 2));
        const float g = *reinterpret_cast<const float*>(pD + off);
        sg += g;
        sgh = fmaf(g, *reinterpret_cast<const float*>(pH + off), sgh);
      }
      Sg += (double)sg;
      Sgx += (double)sgh;      // sum dh1 * h1
    }
    if (more) {
#pragma unroll
      for (int g = 0; g < NP; ++g) { xr[g] = xn[g]; gr[g] = gn[g]; }
    }
  }
  __syncthreads();
  {
    double* red = reinterpret_cast<double*>(smem + OFF_PATCH);
    red[(wave * 3 + 0) * C + lane] = Sg;
    red[(wave * 3 + 1) * C + lane] = Sgx;
    red[(wave * 3 + 2) * C + lane] = Sb;
    __syncthreads();
    if (tid < C) {
      double g = 0.0, gh = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { g += red[(w * 3 + 0) * C + tid]; gh += red[(w * 3 + 1) * C + tid]; b += red[(w * 3 + 2) * C + tid]; }
      // sum dh1 * xhat1 with xhat1 = xh_a * h1 + xh_b on the support of dh1
      atomic_add_f64(a.sums1 + tid, g);
      atomic_add_f64(a.sums1 + C + tid, (double)tab[6 * C + tid] * gh + (double)tab[7 * C + tid] * g);
      if (a.dbias) unsafeAtomicAdd(a.dbias + tid, (float)b);
      if (blockIdx.x == 0 && a.dgamma2) { a.dgamma2[tid] = (float)a.sums2[C + tid]; a.dbeta2[tid] = (float)a.sums2[tid]; }
    }
    __syncthreads();
  }
  {
    float* wred = reinterpret_cast<float*>(smem + OFF_PATCH);          // [4][C][C] floats = 64 KB (the patches: 64 KB)
    const int l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h, co = 32 * j + l31;
          wred[(wave * C + ci) * C + co] = wacc[i][j][r];
        }
    __syncthreads();
    for (int idx = tid; idx < C * C; idx += 256) {
      const float s = wred[idx] + wred[C * C + idx] + wred[2 * C * C + idx] + wred[3 * C * C + idx];
      unsafeAtomicAdd(a.dW + idx, s);
    }
  }
}

static int pw_check(const char* what, const void* x, const void* W, int64_t rows, int32_t C, const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2,
                    const mopoe_mask_ref* mask) {
  if (!x || !W || rows <= 0 || !bn1 || bn1->mode == 0 || bn1->C != C) { set_error("%s: bad arguments", what); return MOPOE_ERR_ARG; }
  if (C != 64) { set_error("%s: built for 64 channels (C = %d)", what, C); return MOPOE_ERR_ARG; }
  if (bn2 && (bn2->mode == 0 || bn2->C != C)) { set_error("%s: bn2 channel mismatch", what); return MOPOE_ERR_ARG; }
  if (mask && mask->kind != 0 && (mask->kind != 1 || !mask->mask || mask->rows_per_sample % 32 != 0)) {
    set_error("%s: dropout mask must be absent or per (sample, channel) with rows_per_sample a multiple of 32", what); return MOPOE_ERR_ARG;
  }
  if ((size_t)rows * C * 2 >= (1ull << 31) || ((uintptr_t)x & 15) || ((uintptr_t)W & 15)) { set_error("%s: tensors must be 16-byte aligned and smaller than 2 GiB", what); return MOPOE_ERR_ARG; }
  return 0;
}

static int pw_grid(int ntiles, int per_cu) {
  const int blocks = (ntiles + 7) / 8;
  return blocks < 256 * per_cu ? blocks : 256 * per_cu;       // 8-wave blocks, each wave walking its tiles
}

}  // namespace mopoe

using namespace mopoe;

extern "C" int mopoe_block_front_stats_bf16(const uint16_t* x, const uint16_t* w1, const float* bias, int64_t rows, int32_t C,
                                            const mopoe_bn_ref* bn1, const mopoe_mask_ref* mask1, double* stats_d1, void* stream) {
  if (int rc = pw_check("block_front_stats_bf16", x, w1, rows, C, bn1, nullptr, mask1)) return rc;
  if (!stats_d1) { set_error("block_front_stats_bf16: null statistics"); return MOPOE_ERR_ARG; }
  PwArgs a = {};
  a.x = x; a.W = w1; a.bias = bias; a.bn1 = *bn1; a.stats_d1 = stats_d1; a.R = rows; a.ntiles = (int)((rows + 31) / 32);
  a.x_bytes = (unsigned)((size_t)rows * C * 2);
  if (mask1) a.mask1 = *mask1;
  ProfScope prof((hipStream_t)stream, 2.0 * (double)rows * C * C, PROF_PW_FRONT, (double)rows * C * 2.0);
  hipLaunchKernelGGL((pw_front_fwd_bf16_kernel<64, false>), dim3(pw_grid(a.ntiles, 2)), dim3(512), 0, (hipStream_t)stream, a);
  return check_launch("block_front_stats_bf16");
}

extern "C" int mopoe_block_front_apply_bf16(const uint16_t* x, const uint16_t* w1, const float* bias, uint16_t* a2, int64_t rows, int32_t C,
                                            const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2, const mopoe_mask_ref* mask1, void* stream) {
  if (int rc = pw_check("block_front_apply_bf16", x, w1, rows, C, bn1, bn2, mask1)) return rc;
  if (!a2 || !bn2 || ((uintptr_t)a2 & 15)) { set_error("block_front_apply_bf16: bad output / bn2"); return MOPOE_ERR_ARG; }
  PwArgs a = {};
  a.x = x; a.W = w1; a.bias = bias; a.out = a2; a.bn1 = *bn1; a.bn2 = *bn2; a.R = rows; a.ntiles = (int)((rows + 31) / 32);
  a.x_bytes = (unsigned)((size_t)rows * C * 2);
  if (mask1) a.mask1 = *mask1;
  ProfScope prof((hipStream_t)stream, 2.0 * (double)rows * C * C, PROF_PW_FRONT + 1, (double)rows * C * 4.0);
  hipLaunchKernelGGL((pw_front_fwd_bf16_kernel<64, true>), dim3(pw_grid(a.ntiles, 2)), dim3(512), 0, (hipStream_t)stream, a);
  return check_launch("block_front_apply_bf16");
}

extern "C" int mopoe_block_front_bwd_bf16(const uint16_t* x, const uint16_t* dh2, const uint16_t* w1, const float* bias, uint16_t* dh1,
                                          int64_t rows, int32_t C, const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2,
                                          const mopoe_mask_ref* mask1, const double* sums2, double* sums1, float* dw1, float* dbias,
                                          float* dgamma2, float* dbeta2, void* stream) {
  if (int rc = pw_check("block_front_bwd_bf16", x, w1, rows, C, bn1, bn2, mask1)) return rc;
  if (!dh2 || !dh1 || !bn2 || !sums2 || !sums1 || !dw1 || ((uintptr_t)dh2 & 15) || ((uintptr_t)dh1 & 15)) {
    set_error("block_front_bwd_bf16: bad arguments"); return MOPOE_ERR_ARG;
  }
  PwArgs a = {};
  a.x = x; a.dh2 = dh2; a.W = w1; a.bias = bias; a.out = dh1; a.bn1 = *bn1; a.bn2 = *bn2; a.sums2 = sums2; a.sums1 = sums1;
  a.dW = dw1; a.dbias = dbias; a.dgamma2 = dgamma2 && dbeta2 ? dgamma2 : nullptr; a.dbeta2 = dbeta2; a.R = rows; a.ntiles = (int)((rows + 31) / 32);
  a.x_bytes = (unsigned)((size_t)rows * C * 2);
  if (mask1) a.mask1 = *mask1;
  ProfScope prof((hipStream_t)stream, 6.0 * (double)rows * C * C, PROF_PW_FRONT + 2, (double)rows * C * 6.0);
  hipLaunchKernelGGL((pw_front_bwd_bf16_kernel<64>), dim3(pw_grid(a.ntiles, 1)), dim3(512), 0, (hipStream_t)stream, a);
  return check_launch("block_front_bwd_bf16");
}

static PwArgsF pw_args_f32(const float* x, const float* w1, const float* bias, int64_t rows, const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2,
                           const mopoe_mask_ref* mask1) {
  PwArgsF a = {};
  a.x = x; a.W = w1; a.bias = bias; a.bn1 = *bn1; a.R = rows; a.ntiles = (int)((rows + 31) / 32);
  a.x_bytes = (unsigned)((size_t)rows * 64 * 4);
  if (bn2) a.bn2 = *bn2;
  if (mask1) a.mask1 = *mask1;
  return a;
}
static int pw_check_f32(const char* what, const void* x, const void* W, int64_t rows, int32_t C, const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2,
                        const mopoe_mask_ref* mask) {
  if (int rc = pw_check(what, x, W, rows, C, bn1, bn2, mask)) return rc;
  if ((size_t)rows * C * 4 >= (1ull << 31)) { set_error("%s: tensors must be smaller than 2 GiB", what); return MOPOE_ERR_ARG; }
  return 0;
}
static int pw_grid4(int ntiles, int per_cu) {
  const int blocks = (ntiles + 3) / 4;
  return blocks < 256 * per_cu ? blocks : 256 * per_cu;       // 4-wave blocks (the backward: one per CU, one wave per SIMD)
}

extern "C" int mopoe_block_front_stats(const float* x, const float* w1, const float* bias, int64_t rows, int32_t C,
                                       const mopoe_bn_ref* bn1, const mopoe_mask_ref* mask1, double* stats_d1, void* stream) {
  if (int rc = pw_check_f32("block_front_stats", x, w1, rows, C, bn1, nullptr, mask1)) return rc;
  if (!stats_d1) { set_error("block_front_stats: null statistics"); return MOPOE_ERR_ARG; }
  PwArgsF a = pw_args_f32(x, w1, bias, rows, bn1, nullptr, mask1);
  a.stats_d1 = stats_d1;
  ProfScope prof((hipStream_t)stream, 2.0 * (double)rows * C * C, PROF_PW_FRONT + 3, (double)rows * C * 4.0);
  hipLaunchKernelGGL((pw_front_fwd_f32_kernel<false>), dim3(pw_grid4(a.ntiles, 2)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("block_front_stats");
}

extern "C" int mopoe_block_front_apply(const float* x, const float* w1, const float* bias, float* a2, int64_t rows, int32_t C,
                                       const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2, const mopoe_mask_ref* mask1, void* stream) {
  if (int rc = pw_check_f32("block_front_apply", x, w1, rows, C, bn1, bn2, mask1)) return rc;
  if (!a2 || !bn2 || ((uintptr_t)a2 & 15)) { set_error("block_front_apply: bad output / bn2"); return MOPOE_ERR_ARG; }
  PwArgsF a = pw_args_f32(x, w1, bias, rows, bn1, bn2, mask1);
  a.out = a2;
  ProfScope prof((hipStream_t)stream, 2.0 * (double)rows * C * C, PROF_PW_FRONT + 4, (double)rows * C * 8.0);
  hipLaunchKernelGGL((pw_front_fwd_f32_kernel<true>), dim3(pw_grid4(a.ntiles, 2)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("block_front_apply");
}

extern "C" int mopoe_block_front_bwd(const float* x, const float* dh2, const float* w1, const float* bias, float* dh1, int64_t rows, int32_t C,
                                     const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2, const mopoe_mask_ref* mask1, const double* sums2,
                                     double* sums1, float* dw1, float* dbias, float* dgamma2, float* dbeta2, void* stream) {
  if (int rc = pw_check_f32("block_front_bwd", x, w1, rows, C, bn1, bn2, mask1)) return rc;
  if (!dh2 || !dh1 || !bn2 || !sums2 || !sums1 || !dw1 || ((uintptr_t)dh2 & 15) || ((uintptr_t)dh1 & 15)) {
    set_error("block_front_bwd: bad arguments"); return MOPOE_ERR_ARG;
  }
  PwArgsF a = pw_args_f32(x, w1, bias, rows, bn1, bn2, mask1);
  a.dh2 = dh2; a.out = dh1; a.sums2 = sums2; a.sums1 = sums1; a.dW = dw1; a.dbias = dbias;
  a.dgamma2 = dgamma2 && dbeta2 ? dgamma2 : nullptr; a.dbeta2 = dbeta2;
  ProfScope prof((hipStream_t)stream, 6.0 * (double)rows * C * C, PROF_PW_FRONT + 5, (double)rows * C * 12.0);
  hipLaunchKernelGGL(pw_front_bwd_f32_kernel, dim3(pw_grid4(a.ntiles, 1)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("block_front_bwd");
}
