// Implicit-GEMM convolution family on bf16 MFMA (v_mfma_f32_32x32x16_bf16) for gfx950: bf16 operands from HBM, fp32
// accumulation, bf16 (or fp32) results -- BASELINE configs #3 and #5.
//
// Same problem statement as conv_gemm.hip (activations = channels-last row matrices [pixels][C], weights packed
// Wp[tap][Cin][Cout], gather forms 0 / 1, weight gradient), different machine mapping: one bf16 MFMA consumes, per
// lane, EIGHT consecutive K elements of its row / column (A[row = lane % 32][k = 8 (lane / 32) + j], B[k][col]),
// i.e. 16 bytes, so
//   * an operand whose K runs along memory (activations: K = channels; weights of the input gradient: K = Cout)
//     is staged as [row][k] LDS rows of 80 bytes (64 + 16 pad: the 16 rows a ds_read_b128 lane group touches then
//     sit in 16 different 16-byte bank slots) and read with one ds_read_b128 per fragment;
//   * an operand whose K is the STRIDED index (forward weights Wp[k = ci][n = co]; both operands of the weight
//     gradient, where K = pixels) is staged exactly as it lies in memory, [k][n] rows, and read with the hardware
//     transpose ds_read_b64_tr_b16 (two per fragment); the row stride is 64 bytes more than a multiple of 256 so the four
//     k-rows of one transposed read fall into four different quarters of the bank row.
// Nothing is re-laid out in HBM: the bf16 copy of a weight tensor serves forward, input gradient and (as the
// destination layout) the weight gradient.
// Rounding points (the test oracle has a bf16 mode that rounds at the same places): operands after
// BN+ReLU, and every stored result; bias, dropout mask, statistics and the BN-backward sums are fp32 / fp64 as in the
// fp32 family, and statistics are taken over the STORED (rounded) values.
// Requirements (all BASELINE configurations meet them; anything else is rejected with MOPOE_ERR_ARG, there is no
// scalar fallback): K channels a multiple of 32, N channels a multiple of 8, 16-byte aligned tensors < 2 GiB.
#include "gemm_common.hpp"

#ifndef PERSIST_BLOCKS_BF16
#define PERSIST_BLOCKS_BF16 512   // upper bound on blocks of one launch (see conv_gemm.hip: PERSIST_BLOCKS)
#endif

namespace mopoe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BKH = 32;   // K chunk in bf16 elements (two 16-deep MFMA steps); Ck % BKH == 0 is required

struct GemmArgsH {
  const bf16_t* X;       // gathered operand [rows_x][ldx]
  const bf16_t* W;       // Wp[tap][Cin_w][Cout_w]
  void* Y;               // bf16_t or float [rows_total][ldy]
  const float* bias;
  int N, Hx, Wx, Hy, Wy, Ck, Cn, Cin_w, Cout_w;
  int ldx, ldy;
  int kh, kw, sh, sw, ph, pw;
  int form;
  int Hq, Wq;
  long rows_per_phase;
  mopoe_bn_ref bn_in;
  mopoe_mask_ref mask;
  double* out_stats;
  mopoe_bn_ref relu_bn;
  const bf16_t* xin;     // [rows_total][ldy]
  double* bwd_sums;
  unsigned x_bytes, w_bytes;
  int nsplit;
  float* partial;        // split reduction: [nsplit][rows_total][Cn] fp32 slabs (else nullptr)
  int* counters;         // split reduction: one arrival counter per output tile, zero on entry, left zero
  long rows_total;
  int out_f32;
  int mix;               // forward only: y = mix_a * bn(xin) + mix_b * mask * (conv + bias); relu_bn carries that bn
  float mix_a, mix_b;
  int xcd_remap;         // 1: XCD-aware block numbering
};

// storage helpers of gemm_epilogue_rows.inc for this family: bf16 (or, on request, fp32) results, bf16 xin
__device__ __forceinline__ float epi_round(const GemmArgsH& a, float x) { return a.out_f32 ? x : round_bf16(x); }
typedef uint4 EpiXinRaw;
constexpr bool EPI_PREFETCH_XIN = false;   // measured on C3 / C5: requesting xin ahead of the transposition costs registers, -4 % (rejected)
__device__ __forceinline__ EpiXinRaw epi_xin_ld(const GemmArgsH& a, long yrow, int ncol, bool /*ok_hi*/) {
  return *reinterpret_cast<const uint4*>(a.xin + yrow * a.ldy + ncol);
}
__device__ __forceinline__ void epi_xin_unpack(const EpiXinRaw& xu, float (&xi)[8]) {
  xi[0] = bf16_lo(xu.x); xi[1] = bf16_hi(xu.x); xi[2] = bf16_lo(xu.y); xi[3] = bf16_hi(xu.y);
  xi[4] = bf16_lo(xu.z); xi[5] = bf16_hi(xu.z); xi[6] = bf16_lo(xu.w); xi[7] = bf16_hi(xu.w);
}
__device__ __forceinline__ void epi_store8(const GemmArgsH& a, long yrow, int ncol, bool /*ok_hi*/, const float (&v)[8]) {
  if (a.out_f32) {
    float* dst = reinterpret_cast<float*>(a.Y) + yrow * a.ldy + ncol;
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.Y) + yrow * a.ldy + ncol) =
        make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
  }
}

__device__ __forceinline__ uint4 bldq(__amdgpu_buffer_rsrc_t srd, unsigned byte_off, unsigned s_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, byte_off, s_off, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// 8 bf16 (one uint4) -> relu(x * scale + shift) -> 8 bf16; zero when !ok (spatial padding stays zero AFTER the transform).
// Six VALU instructions per pair instead of eight: one packed fp32 fma, one packed conversion (round to nearest even), and the
// ReLU on the packed result as a signed 16-bit max (rounding keeps the sign, so relu(round(y)) == round(relu(y)); -0 becomes +0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bn_relu2(unsigned u, float s0, float s1, float t0, float t1) {
  f32x2 x, s, t;
  x.x = bf16_lo(u); x.y = bf16_hi(u);
  s.x = s0; s.y = s1; t.x = t0; t.y = t1;
  const f32x2 y = __builtin_elementwise_fma(x, s, t);
  const s16x2 p = __builtin_bit_cast(s16x2, pack_bf16(y.x, y.y));
  const s16x2 zero = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(p, zero));
}
__device__ __forceinline__ uint4 bn_relu8(uint4 u, const float4& sc0, const float4& sc1, const float4& sh0, const float4& sh1, bool ok) {
  uint4 r;
  r.x = bn_relu2(u.x, sc0.x, sc0.y, sh0.x, sh0.y);
  r.y = bn_relu2(u.y, sc0.z, sc0.w, sh0.z, sh0.w);
  r.z = bn_relu2(u.z, sc1.x, sc1.y, sh1.x, sh1.y);
  r.w = bn_relu2(u.w, sc1.z, sc1.w, sh1.z, sh1.w);
  return ok ? r : make_uint4(0u, 0u, 0u, 0u);
}

// fragment of the K-strided operand: element j of lane l = tile[k0 + 8 (l / 32) + j][c0 + l % 32] from an LDS image
// stored [k][column] with row stride LD elements; two hardware-transposed 4 x 16 block reads
template <int LD>
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int k0, int c0, int lane) {
  const int i16 = lane & 15, g16 = (lane >> 4) & 1, lhi = lane >> 5;
  const bf16_t* p = tile + (k0 + lhi * 8 + (i16 >> 2)) * LD + c0 + g16 * 16 + (i16 & 3) * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * LD));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// SPEC: 1 forward, plain operand   2 forward, BN+ReLU on the operand   3 input gradient (weights K-contiguous)
template <int BM, int BN, int WGM, int WGN, int SPEC>
__global__ __launch_bounds__(64 * WGM * WGN) void gather_gemm_bf16_kernel(const GemmArgsH a) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TI = WM / 32, TJ = WN / 32;
  static_assert(TI >= 1 && TJ >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile is made of 32x32 MFMA tiles");
  constexpr bool xform = SPEC == 2;
  constexpr bool w_nk = SPEC == 3;
  constexpr int A_LD = BKH + 8;                  // 80-byte rows
  constexpr int BNK_LD = BKH + 8;
  constexpr int BKN_LD = BN + 32;                // row stride = 64 (mod 256) bytes
  constexpr int A_PER_THR = BM * 4 / NT;         // 16-byte pieces per thread (4 per tile row)
  static_assert((BM * 4) % NT == 0 && A_PER_THR >= 1, "A tile pieces divide over the block");
  constexpr int N8 = BN / 8;                     // pieces per k-row of the [k][n] weight tile
  constexpr int B_PER_THR = w_nk ? BN * 4 / NT : (BKH * N8) / NT;
  static_assert(w_nk ? (BN * 4) % NT == 0 : (BKH * N8) % NT == 0, "B tile pieces divide over the block");
  static_assert(B_PER_THR >= 1, "B tile");
  constexpr int B_ELEMS = w_nk ? BN * BNK_LD : BKH * BKN_LD;

  // epilogue staging: each wave transposes its accumulators through a private [32 rows][WN + 4] fp32 patch, so that the
  // result leaves the chip row-major in 16-byte pieces (and the BN/ReLU operand, the dropout mask and the bias arrive the
  // same way); the patches overlay the operand tiles, which are dead by then
  constexpr int STG_LD = WN + 4;
  constexpr int A_BYTES = 2 * BM * A_LD * 2, B_BYTES = 2 * B_ELEMS * 2;
  constexpr int STG_BYTES = (NT / 64) * 32 * STG_LD * 4;
  constexpr int SMEM_BYTES = (A_BYTES + B_BYTES) > STG_BYTES ? (A_BYTES + B_BYTES) : STG_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  __shared__ __attribute__((aligned(16))) float bnS[xform ? MAX_BN_C : 4];
  __shared__ __attribute__((aligned(16))) float bnT[xform ? MAX_BN_C : 4];
  __shared__ __attribute__((aligned(16))) float epi[5][BN];   // per output column: mean, rstd, scale, shift (relu_bn), bias
  __shared__ int s_ticket;
  bool was_last = false;   // split reduction: this block finished at least one tile (it then owns column statistics)
  bf16_t* const As0 = reinterpret_cast<bf16_t*>(smem);
  bf16_t* const Bs0 = reinterpret_cast<bf16_t*>(smem + A_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  // XCD-aware logical block (gemm_common.hpp: xcd_swizzle): each XCD owns a contiguous range of M-tile groups together with
  // ALL their column tiles, phases and splits -- the blocks that gather the same activation rows fill one L2, not eight
  unsigned lbx = blockIdx.x, lby = blockIdx.y, lbz = blockIdx.z;
  if (a.xcd_remap) {
    const unsigned inner = gridDim.y * gridDim.z;
    const unsigned sw = xcd_swizzle(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * inner);
    lbx = sw / inner;
    const unsigned rem = sw - lbx * inner;
    lby = rem % gridDim.y;
    lbz = rem / gridDim.y;
  }
  const int n0 = lby * BN;
  const int phase = lbz / a.nsplit;
  const int split = lbz - phase * a.nsplit;

  const TapWalk tw = tap_walk(a, phase);
  const int nty = tw.nty, ntx = tw.ntx, ky0 = tw.ky0, kx0 = tw.kx0, kstep_y = tw.kstep_y, kstep_x = tw.kstep_x;
  const int dsgn = tw.dsgn, cy = tw.cy, cx = tw.cx, phy = tw.phy, phx = tw.phx;
  const int nkc = a.Ck / BKH;
  const int total_all = nty * ntx * nkc;
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (int)a.w_bytes, 0x00020000);
  const int per_split = (total_all + a.nsplit - 1) / a.nsplit;
  const int it_beg = split * per_split;
  const int it_end = it_beg + per_split < total_all ? it_beg + per_split : total_all;
  const int total = it_end > it_beg ? it_end - it_beg : 0;

  if (xform) {
    for (int c = tid; c < a.Ck; c += NT) {
      const BnC k = bn_coef(a.bn_in, c);
      bnS[c] = k.scale;
      bnT[c] = k.shift;
    }
    __syncthreads();
  }

  const int kq = tid & 3;            // 16-byte piece within a 64-byte tile row
  const int trow = tid >> 2;
  const int l31 = lane & 31, lhi = lane >> 5;
  const bool do_relu_bn = a.relu_bn.mode != 0 && !a.mix;
  const int hw = a.Hq * a.Wq;

  // epilogue constants per output column of this block (zero for columns past Cn)
  for (int c = tid; c < BN; c += NT) {
    const int n = n0 + c;
    BnC k = BnC{0.f, 0.f, 0.f, 0.f};
    if (n < a.Cn && (do_relu_bn || a.mix)) k = bn_coef(a.relu_bn, n);
    if (a.mix) { k.scale *= a.mix_a; k.shift *= a.mix_a; }
    epi[0][c] = k.mean; epi[1][c] = k.rstd; epi[2][c] = k.scale; epi[3][c] = k.shift;
    epi[4][c] = (n < a.Cn && a.bias) ? a.bias[n] : 0.f;
  }
  __syncthreads();
  // row-major epilogue: E_LPR lanes share one output row, each owns 8 consecutive columns (one 16-byte bf16 piece)
  constexpr int E_LPR = WN / 8, E_RPP = 64 / E_LPR, E_NPASS = 32 / E_RPP;
  const int c8 = lane % E_LPR, rsub = lane / E_LPR;
  const int ecol = wn * WN + c8 * 8;             // first of this lane's columns within the block tile
  const int ncol = n0 + ecol;
  const bool ok_lo = ncol < a.Cn, ok_hi = ok_lo;   // (Cn % 8 == 0: the 8 columns are valid together)
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  float* const stg = reinterpret_cast<float*>(smem) + wave * 32 * STG_LD;

  // weight operand: per-thread byte offsets are fixed for the whole kernel, the K / tap advance is the scalar offset
  unsigned voffB[B_PER_THR];
#pragma unroll
  for (int i = 0; i < B_PER_THR; ++i) {
    if (w_nk) {
      const int n = n0 + trow + i * (NT / 4);
      voffB[i] = n < a.Cn ? ((unsigned)n * (unsigned)a.Cout_w + (unsigned)kq * 8u) * 2u : OOB;
    } else {
      const int p = tid + i * NT;
      const int k = p / N8, n = n0 + (p % N8) * 8;
      voffB[i] = n < a.Cn ? ((unsigned)k * (unsigned)a.Cout_w + (unsigned)n) * 2u : OOB;
    }
  }

  const long nMt = (a.rows_per_phase + BM - 1) / BM;
  for (long mt = lbx; mt < nMt; mt += gridDim.x) {
    const long m0 = mt * BM;

    int ry0[A_PER_THR], rx0[A_PER_THR], rbase[A_PER_THR];
    bool rvalid[A_PER_THR];
#pragma unroll
    for (int i = 0; i < A_PER_THR; ++i) {
      const long m = m0 + trow + i * (NT / 4);
      rvalid[i] = m < a.rows_per_phase;
      const unsigned mm = rvalid[i] ? (unsigned)m : 0u;
      const int n = (int)(mm / (unsigned)hw);
      const int rem = (int)(mm - (unsigned)n * (unsigned)hw);
      const int qy = rem / a.Wq, qx = rem - qy * a.Wq;
      if (a.form == 0) { ry0[i] = qy * a.sh - a.ph; rx0[i] = qx * a.sw - a.pw; }
      else             { ry0[i] = qy + cy;          rx0[i] = qx + cx; }
      rbase[i] = (n * a.Hx + ry0[i]) * a.Wx + rx0[i];
    }

    uint4 ra[A_PER_THR], rb[B_PER_THR];
    float4 psc0 = make_float4(0.f, 0.f, 0.f, 0.f), psc1 = psc0, psh0 = psc0, psh1 = psc0;
    bool pend_ok[A_PER_THR];
    int ld_tap = it_beg / nkc;
    int ld_kc = (it_beg - ld_tap * nkc) * BKH;
    bool tap_dirty = true;
    unsigned offA[A_PER_THR], offW = 0;

    auto load_tiles = [&]() {
      if (tap_dirty) {
        const int jy = ld_tap / ntx, jx = ld_tap - jy * ntx;
        const int wtap = (ky0 + kstep_y * jy) * a.kw + (kx0 + kstep_x * jx);
        const int tapoff = dsgn * (jy * a.Wx + jx);
#pragma unroll
        for (int i = 0; i < A_PER_THR; ++i) {
          const int iy = ry0[i] + dsgn * jy, ix = rx0[i] + dsgn * jx;
          const bool ok = rvalid[i] & ((unsigned)iy < (unsigned)a.Hx) & ((unsigned)ix < (unsigned)a.Wx);
          offA[i] = ok ? ((unsigned)(rbase[i] + tapoff) * (unsigned)a.ldx + (unsigned)kq * 8u) * 2u : OOB;
          pend_ok[i] = ok;
        }
        offW = (unsigned)wtap * (unsigned)a.Cin_w * (unsigned)a.Cout_w * 2u;
        tap_dirty = false;
      }
      const unsigned sA = (unsigned)ld_kc * 2u;
      const unsigned sB = offW + (w_nk ? (unsigned)ld_kc * 2u : (unsigned)ld_kc * (unsigned)a.Cout_w * 2u);
      if (xform) {
        psc0 = *reinterpret_cast<const float4*>(&bnS[ld_kc + kq * 8]);
        psc1 = *reinterpret_cast<const float4*>(&bnS[ld_kc + kq * 8 + 4]);
        psh0 = *reinterpret_cast<const float4*>(&bnT[ld_kc + kq * 8]);
        psh1 = *reinterpret_cast<const float4*>(&bnT[ld_kc + kq * 8 + 4]);
      }
#pragma unroll
      for (int i = 0; i < A_PER_THR; ++i) ra[i] = bldq(srdX, offA[i], sA);
#pragma unroll
      for (int i = 0; i < B_PER_THR; ++i) rb[i] = bldq(srdW, voffB[i], sB);
      ld_kc += BKH;
      if (ld_kc >= a.Ck) { ld_kc = 0; ++ld_tap; tap_dirty = true; }
    };

    auto store_tiles = [&](auto bufc) {
      constexpr int buf = decltype(bufc)::value;
#pragma unroll
      for (int i = 0; i < A_PER_THR; ++i) {
        const int r = trow + i * (NT / 4);
        uint4 v = ra[i];
        if (xform) v = bn_relu8(v, psc0, psc1, psh0, psh1, pend_ok[i]);
        *reinterpret_cast<uint4*>(&As0[buf * BM * A_LD + r * A_LD + kq * 8]) = v;
      }
#pragma unroll
      for (int i = 0; i < B_PER_THR; ++i) {
        if (w_nk) {
          const int r = trow + i * (NT / 4);
          *reinterpret_cast<uint4*>(&Bs0[buf * B_ELEMS + r * BNK_LD + kq * 8]) = rb[i];
        } else {
          const int p = tid + i * NT;
          *reinterpret_cast<uint4*>(&Bs0[buf * B_ELEMS + (p / N8) * BKN_LD + (p % N8) * 8]) = rb[i];
        }
      }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (total > 0) {
      load_tiles();
      store_tiles(std::integral_constant<int, 0>{});
    }
    __syncthreads();

    auto chunk = [&](int it, auto curc) {
      constexpr int cur = decltype(curc)::value;
      // the scale/shift registers (psc/psh) and the validity flags (pend_ok, recomputed only when the tap changes)
      // belong to the load issued HERE and are consumed by the store at the end of this chunk
      if (it + 1 < total) load_tiles();
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the MFMA block
#pragma unroll
      for (int s = 0; s < BKH / 16; ++s) {
        bf16x8 av[TI], bv[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
          av[i] = *reinterpret_cast<const bf16x8*>(&As0[cur * BM * A_LD + (wm * WM + i * 32 + l31) * A_LD + s * 16 + lhi * 8]);
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          if (w_nk) bv[j] = *reinterpret_cast<const bf16x8*>(&Bs0[cur * B_ELEMS + (wn * WN + j * 32 + l31) * BNK_LD + s * 16 + lhi * 8]);
          else bv[j] = tr_frag<BKN_LD>(&Bs0[cur * B_ELEMS], s * 16, wn * WN + j * 32, lane);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
      if (it + 1 < total) store_tiles(std::integral_constant<int, cur ^ 1>{});
      __syncthreads();
    };
    int it = 0;
    for (; it + 1 < total; it += 2) {
      chunk(it, std::integral_constant<int, 0>{});
      chunk(it + 1, std::integral_constant<int, 1>{});
    }
    if (it < total) chunk(it, std::integral_constant<int, 0>{});

#include "gemm_epilogue_rows.inc"
  }

#include "gemm_colstats_rows.inc"
}

// =====================================================================================================
// weight gradient: dWp[tap][ci][co] (fp32) = sum over pixels of T(x)[pixel][ci] * dy[pixel][co]
// K = pixels is the strided index of BOTH operands: both tiles are staged [pixel][channel] as they lie in memory and
// both fragments come from transposed reads.  32 pixels per chunk; the pixel -> (image, y, x) split is shifts and
// masks when the small grid is a power of two (every BASELINE shape), integer divisions otherwise.
// =====================================================================================================
struct WgradArgsH {
  const bf16_t* Xs;
  const bf16_t* Dy;
  float* dW;
  int N, Hs, Ws, Hb, Wb, Cin, Cout, kh, kw, sh, sw, ph, pw;
  int x_is_big;
  long Ms;
  long chunk;
  int nJ;
  int atomic;
  unsigned x_bytes, dy_bytes;
  int lg_ws, lg_hw;      // log2 of Ws and Hs*Ws (POW2 kernels)
  mopoe_bn_ref bn_in;  int xcd_remap;         // 1: XCD-aware block numbering (gemm_common.hpp: xcd_swizzle)
};

template <int BI, int BJ, bool XFORM, bool POW2>
__global__ __launch_bounds__(256) void wgrad_gemm_bf16_kernel(const WgradArgsH a) {
  constexpr int WI = BI / 2, WJ = BJ / 2, TI = WI / 32, TJ = WJ / 32;
  constexpr int I_LD = BI + 32, J_LD = BJ + 32;   // row stride = 64 (mod 256) bytes
  constexpr int I8 = BI / 8, J8 = BJ / 8;
  constexpr int I_PER_THR = (BKH * I8) / 256, J_PER_THR = (BKH * J8) / 256;
  static_assert(BI == BJ && I_PER_THR >= 1, "square tiles: both operands cover the same pixels per thread");

  __shared__ __attribute__((aligned(16))) bf16_t Is[2][BKH * I_LD];
  __shared__ __attribute__((aligned(16))) bf16_t Js[2][BKH * J_LD];
  __shared__ __attribute__((aligned(16))) float bnS[XFORM ? MAX_BN_C : 4];
  __shared__ __attribute__((aligned(16))) float bnT[XFORM ? MAX_BN_C : 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  // logical block id: x = channel tile, y = tap fastest, z = pixel chunk slowest, each XCD owning a contiguous range of it:
  // every block of one pixel chunk (all taps, all channel tiles) then reads its activation / gradient rows through ONE L2
  const unsigned lin = a.xcd_remap ? xcd_swizzle(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                                 gridDim.x * gridDim.y * gridDim.z)
                                   : blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned bx = lin % gridDim.x, byz = lin / gridDim.x;
  const unsigned by = byz % gridDim.y, bz = byz / gridDim.y;
  const int it_i = bx / a.nJ, it_j = bx % a.nJ;
  const int i0 = it_i * BI, j0 = it_j * BJ;
  const int tap = by;
  const int ky = tap / a.kw, kx = tap % a.kw;
  const long mbeg = (long)bz * a.chunk;
  const long mend = mbeg + a.chunk < a.Ms ? mbeg + a.chunk : a.Ms;
  const int total = (int)((mend - mbeg + BKH - 1) / BKH);

  if (XFORM) {
    for (int c = tid; c < a.Cin; c += 256) {
      const BnC k = bn_coef(a.bn_in, c);
      bnS[c] = k.scale;
      bnT[c] = k.shift;
    }
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)a.Xs, 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdD = __builtin_amdgcn_make_buffer_rsrc((void*)a.Dy, 0, (int)a.dy_bytes, 0x00020000);

  const int c8 = tid % I8;                        // this thread's 8-channel group (same for both operands: BI == BJ)
  const int slot0 = tid / I8;                     // its pixel slot within a pass
  constexpr int PPP = 256 / I8;                   // pixels per pass
  const int ci = i0 + c8 * 8, cj = j0 + c8 * 8;
  const bool ci_ok = ci < a.Cin, cj_ok = cj < a.Cout;
  float4 sc0 = make_float4(0.f, 0.f, 0.f, 0.f), sc1 = sc0, sh0 = sc0, sh1 = sc0;
  if (XFORM && ci_ok) {
    sc0 = *reinterpret_cast<const float4*>(&bnS[ci]); sc1 = *reinterpret_cast<const float4*>(&bnS[ci + 4]);
    sh0 = *reinterpret_cast<const float4*>(&bnT[ci]); sh1 = *reinterpret_cast<const float4*>(&bnT[ci + 4]);
  }
  const unsigned hw = (unsigned)(a.Hs * a.Ws);

  uint4 ri[I_PER_THR], rj[J_PER_THR];
  bool okx_reg[I_PER_THR];
  unsigned m_next = (unsigned)mbeg;               // first pixel of the next chunk to load

  auto load_tiles = [&]() {
#pragma unroll
    for (int t = 0; t < I_PER_THR; ++t) {
      const unsigned m = m_next + (unsigned)(slot0 + t * PPP);
      const bool inm = (long)m < mend;
      unsigned n, qy, qx;
      if (POW2) {
        n = m >> a.lg_hw;
        const unsigned rem = m & (hw - 1u);
        qy = rem >> a.lg_ws;
        qx = rem & ((unsigned)a.Ws - 1u);
      } else {
        n = m / hw;
        const unsigned rem = m - n * hw;
        qy = rem / (unsigned)a.Ws;
        qx = rem - qy * (unsigned)a.Ws;
      }
      const int by = (int)qy * a.sh - a.ph + ky, bx = (int)qx * a.sw - a.pw + kx;
      const bool inb = ((unsigned)by < (unsigned)a.Hb) & ((unsigned)bx < (unsigned)a.Wb);
      const unsigned brow = (n * (unsigned)a.Hb + (unsigned)by) * (unsigned)a.Wb + (unsigned)bx;
      const unsigned rowx = a.x_is_big ? brow : m, rowd = a.x_is_big ? m : brow;
      const bool okx = inm & ci_ok & (a.x_is_big ? inb : true);
      const bool okd = inm & cj_ok & (a.x_is_big ? true : inb);
      ri[t] = bldq(srdX, okx ? (rowx * (unsigned)a.Cin + (unsigned)ci) * 2u : OOB, 0u);
      rj[t] = bldq(srdD, okd ? (rowd * (unsigned)a.Cout + (unsigned)cj) * 2u : OOB, 0u);
      okx_reg[t] = okx;
    }
    m_next += BKH;
  };

  auto store_tiles = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int t = 0; t < I_PER_THR; ++t) {
      const int p = slot0 + t * PPP;
      uint4 v = ri[t];
      if (XFORM) v = bn_relu8(v, sc0, sc1, sh0, sh1, okx_reg[t]);
      *reinterpret_cast<uint4*>(&Is[buf][p * I_LD + c8 * 8]) = v;
      *reinterpret_cast<uint4*>(&Js[buf][p * J_LD + c8 * 8]) = rj[t];
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (total > 0) {
    load_tiles();
    store_tiles(std::integral_constant<int, 0>{});
  }
  __syncthreads();

  auto chunk = [&](int it, auto curc) {
    constexpr int cur = decltype(curc)::value;
    if (it + 1 < total) load_tiles();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < BKH / 16; ++s) {
      bf16x8 av[TI], bv[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) av[i] = tr_frag<I_LD>(&Is[cur][0], s * 16, wi * WI + i * 32, lane);
#pragma unroll
      for (int j = 0; j < TJ; ++j) bv[j] = tr_frag<J_LD>(&Js[cur][0], s * 16, wj * WJ + j * 32, lane);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (it + 1 < total) store_tiles(std::integral_constant<int, cur ^ 1>{});
    __syncthreads();
  };
  int it = 0;
  for (; it + 1 < total; it += 2) {
    chunk(it, std::integral_constant<int, 0>{});
    chunk(it + 1, std::integral_constant<int, 1>{});
  }
  if (it < total) chunk(it, std::integral_constant<int, 0>{});

  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int co = j0 + wj * WJ + j * 32 + l31;
    if (co >= a.Cout) continue;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cii = i0 + wi * WI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (cii >= a.Cin) continue;
        float* dst = a.dW + ((long)tap * a.Cin + cii) * a.Cout + co;
        if (a.atomic) unsafeAtomicAdd(dst, acc[i][j][r]);
        else *dst = acc[i][j][r];
      }
    }
  }
}

#include "conv_gemm_bf16_glds.inc"

// =====================================================================================================
// host-side launchers
// =====================================================================================================
static int ilog2_exact(long v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1L << l) < v) ++l;
  return l;
}

// tiles: 0 = 128x128 (4 waves)  1 = 256x64 (4 waves)  2 = 64x64 (4 waves)  3 = 256x128 (8 waves)  4 = 128x64 (4 waves)
// LDS-DMA family (conv_gemm_bf16_glds.inc; operands without a transform on load, K channels a multiple of 64):
//   5 = 128x128, 2 buffers   6 = 128x128, 3 buffers   7 = 256x128 (8 waves), 2 buffers   8 = 256x128, 3 buffers
//   9 = 128x64, 2 buffers   10 = 128x64, 3 buffers   11 = 64x64, 4 buffers (the deep layers: few rows, long reductions)
constexpr int BF16_NTILES = 12;
static int launch_gather_bf16(const bf16_t* X, const bf16_t* W, const float* bias, void* Y, int out_f32,
                              const mopoe_conv_geom* g, int dest_on_small, int Ck, int Cn, int w_nk,
                              const mopoe_bn_ref* bn_in, const mopoe_mask_ref* mask, double* out_stats,
                              const mopoe_bn_ref* relu_bn, const bf16_t* xin, double* bwd_sums,
                              const mopoe_conv_plan* plan, void* ws, size_t ws_bytes, hipStream_t stream,
                              const mopoe_mix_ref* mix = nullptr) {
  GemmArgsH a;
  a.X = X; a.W = W; a.Y = Y; a.bias = bias; a.out_f32 = out_f32;
  a.N = g->N; a.Ck = Ck; a.Cn = Cn; a.Cin_w = g->Cin; a.Cout_w = g->Cout;
  a.ldx = Ck; a.ldy = Cn;
  a.kh = g->kh; a.kw = g->kw; a.sh = g->sh; a.sw = g->sw; a.ph = g->ph; a.pw = g->pw;
  int nphase;
  if (dest_on_small) {
    a.form = 0; a.Hx = g->Hb; a.Wx = g->Wb; a.Hy = g->Hs; a.Wy = g->Ws; a.Hq = g->Hs; a.Wq = g->Ws; nphase = 1;
  } else {
    a.form = 1; a.Hx = g->Hs; a.Wx = g->Ws; a.Hy = g->Hb; a.Wy = g->Wb; a.Hq = g->Hb / g->sh; a.Wq = g->Wb / g->sw;
    nphase = g->sh * g->sw;
  }
  a.rows_per_phase = (long)g->N * a.Hq * a.Wq;
  a.rows_total = (long)g->N * a.Hy * a.Wy;
  const size_t xb = (size_t)g->N * a.Hx * a.Wx * Ck * sizeof(bf16_t);
  const size_t wb = (size_t)g->kh * g->kw * g->Cin * g->Cout * sizeof(bf16_t);
  if (Ck % BKH != 0 || Cn % 8 != 0 || g->Cout % 8 != 0) {
    set_error("bf16 conv: K channels must be a multiple of 32 and N channels of 8 (Ck = %d, Cn = %d)", Ck, Cn);
    return MOPOE_ERR_ARG;
  }
  if (!aligned16(X) || !aligned16(W) || !aligned16(Y) || (xin && !aligned16(xin)) || xb >= (1ull << 31) || wb >= (1ull << 31)) {
    set_error("bf16 conv: tensors must be 16-byte aligned and smaller than 2 GiB");
    return MOPOE_ERR_ARG;
  }
  a.x_bytes = (unsigned)xb;
  a.w_bytes = (unsigned)wb;
  mopoe_bn_ref none = {};
  mopoe_mask_ref nomask = {nullptr, 0, 1};
  a.bn_in = bn_in ? *bn_in : none;
  a.mask = mask ? *mask : nomask;
  a.out_stats = out_stats;
  a.relu_bn = relu_bn ? *relu_bn : none;
  a.xin = xin; a.bwd_sums = bwd_sums;
  a.mix = 0; a.mix_a = a.mix_b = 0.f;
  static const bool xcd_remap_g = !getenv("MOPOE_NO_XCD_REMAP");   // (A/B switch)
  a.xcd_remap = xcd_remap_g ? 1 : 0;
  if (mix) {   // residual mix in the epilogue: the shortcut's BN rides in relu_bn, its (bf16) tensor in xin
    if (!mix->s || mix->bn.mode == 0 || relu_bn || xin || out_f32 || !aligned16(mix->s)) { set_error("conv_fwd_mix_bf16: needs an aligned bf16 s, its BatchNorm and a bf16 result"); return MOPOE_ERR_ARG; }
    a.mix = 1; a.mix_a = mix->a; a.mix_b = mix->b;
    a.relu_bn = mix->bn; a.xin = (const bf16_t*)mix->s;
  }
  a.nsplit = 1; a.partial = nullptr; a.counters = nullptr;
  if (a.bn_in.mode != 0 && (a.bn_in.C != Ck || Ck > MAX_BN_C)) { set_error("bn_in channel mismatch (%d vs %d)", a.bn_in.C, Ck); return MOPOE_ERR_ARG; }
  if (a.relu_bn.mode != 0 && (a.relu_bn.C != Cn || !a.xin)) { set_error("relu_bn needs xin and C == %d", Cn); return MOPOE_ERR_ARG; }
  if (a.mask.kind != 0 && !a.mask.mask) { set_error("mask pointer missing"); return MOPOE_ERR_ARG; }
  if (a.mask.kind == 1 && a.mask.rows_per_sample != a.Hy * a.Wy) { set_error("channel mask: rows_per_sample must be Hout*Wout"); return MOPOE_ERR_ARG; }
  if (a.rows_total >= (1L << 31) || (long)g->N * a.Hx * a.Wx >= (1L << 31)) { set_error("conv: more than 2^31 rows"); return MOPOE_ERR_ARG; }

  int cfg;
  if (Cn > 64) cfg = a.rows_per_phase >= 256L * 128 ? 3 : (a.rows_per_phase > 64 ? 0 : 2);
  else cfg = a.rows_per_phase >= 256L * 64 ? 1 : 2;
  const bool glds_ok = Ck % 64 == 0;     // (with BN on load: tiles 5, 7, 9 (two buffers) and 11 (four; the three-buffer forms exceed 256 registers))
  static const bool glds_default = !getenv("MOPOE_BF16_NO_GLDS");   // (A/B switch for the static heuristic)
  if (glds_ok && glds_default) cfg = cfg == 0 ? 5 : (cfg == 3 ? 7 : (cfg == 4 ? 9 : cfg));
  if (plan && plan->tile >= 0) {
    if (plan->tile >= BF16_NTILES) { set_error("bf16 conv plan: tile %d (0..%d)", plan->tile, BF16_NTILES - 1); return MOPOE_ERR_ARG; }
    // (tile 8 = 256x128 with three buffers spills registers: never offered by the tuner, no longer reachable through a plan)
    if (plan->tile == 8) { set_error("bf16 conv plan: tile 8 (256x128, three LDS buffers) is not built: it spills registers; use 7"); return MOPOE_ERR_ARG; }
    if (plan->tile >= 5 && (!glds_ok || (a.bn_in.mode != 0 && (plan->tile == 6 || plan->tile == 8 || plan->tile == 10)))) {
      set_error("bf16 conv plan: tile %d (LDS-DMA family) needs K channels %% 64 == 0 (Ck = %d) and, with BN on load, one of the tiles 5, 7, 9, 11", plan->tile, Ck);
      return MOPOE_ERR_ARG;
    }
    cfg = plan->tile;
  }
  static const int TILE_BM[BF16_NTILES] = {128, 256, 64, 256, 128, 128, 128, 256, 256, 128, 128, 64};
  static const int TILE_BN[BF16_NTILES] = {128, 64, 64, 128, 64, 128, 128, 128, 128, 64, 64, 64};
  const int bm = TILE_BM[cfg], bn = TILE_BN[cfg];
  const long nMt = ceil_div(a.rows_per_phase, bm);
  const int nNt = ceil_div(Cn, bn);
  const int nkc = Ck / (cfg >= 5 ? 64 : BKH);    // K chunks per tap (the LDS-DMA tiles walk K 64 deep)
  const int iters = (dest_on_small ? g->kh * g->kw : std::max(1, (g->kh / g->sh) * (g->kw / g->sw))) * nkc;
  const long blocks = nMt * nNt * nphase;
  const size_t per = (size_t)a.rows_total * Cn * sizeof(float);
  // workspace = [arrival counters: WS_COUNTER_BYTES, zero between launches][fp32 slabs]
  const bool can_split = ws && ws_bytes > WS_COUNTER_BYTES && blocks <= (long)(WS_COUNTER_BYTES / sizeof(int));
  const size_t slab_bytes = can_split ? ws_bytes - WS_COUNTER_BYTES : 0;
  long ns = 1;
  if (plan && plan->split > 0) {
    ns = std::min<long>(plan->split, iters);
    if (ns >= 2 && (!can_split || (size_t)ns * per > slab_bytes)) {
      set_error("bf16 conv plan: split %ld needs %zu workspace bytes (have %zu) and at most %zu output tiles (have %ld)", ns,
                (size_t)ns * per + WS_COUNTER_BYTES, ws ? ws_bytes : (size_t)0, WS_COUNTER_BYTES / sizeof(int), blocks);
      return MOPOE_ERR_ARG;
    }
  } else if (can_split && blocks < 256 && iters >= 8) {
    ns = std::min<long>((512 + blocks - 1) / blocks, (long)iters / 4);
    if ((size_t)ns * per > slab_bytes) ns = (long)(slab_bytes / per);
  }
  if (ns >= 2) {
    a.nsplit = (int)ns;
    a.counters = (int*)ws;
    a.partial = (float*)((char*)ws + WS_COUNTER_BYTES);
  }
  // resident blocks: 8-wave tiles 2 per CU, 4-wave register-staged tiles 3 per CU; the LDS-DMA tiles by their LDS footprint
  // (64 KB -> 2 per CU, 48 KB -> 3 per CU, 96 KB and more -> 1 per CU)
  static const int GLDS_PER_CU[BF16_NTILES] = {0, 0, 0, 0, 0, 2, 1, 1, 1, 3, 2, 2};
  const long persist = cfg >= 5 ? 256L * GLDS_PER_CU[cfg] : (cfg == 3 ? PERSIST_BLOCKS_BF16 : PERSIST_BLOCKS_BF16 * 3 / 2);
  long gx = std::min<long>(nMt, std::max<long>(1, persist / ((long)nNt * nphase * a.nsplit)));
  const double taps_eff = dest_on_small ? (double)g->kh * g->kw : (double)g->kh * g->kw / ((double)g->sh * g->sw);
  const double flops = 2.0 * (double)g->N * a.Hy * a.Wy * (double)Cn * (double)Ck * taps_eff;
  {
    const int spec = w_nk ? 3 : (a.bn_in.mode != 0 ? 2 : 1);
    const double abytes = (double)xb + (double)a.rows_total * Cn * (out_f32 ? 4.0 : 2.0);
    static const int GLDS_X_SLOT[BF16_NTILES] = {0, 0, 0, 0, 0, 0, 0, 1, 0, 2, 3, 4};
    ProfScope prof(stream, flops, cfg >= 5 ? (spec == 2 ? PROF_BF16_GLDS_X + GLDS_X_SLOT[cfg] : PROF_BF16_GLDS + (cfg - 5) * 2 + (spec == 3 ? 1 : 0))
                                           : PROF_BF16_GATHER + cfg * 3 + (spec - 1), abytes);
    dim3 grid((unsigned)gx, nNt, nphase * a.nsplit);
#define MOPOE_LAUNCH_G(BM_, BN_, WM_, WN_, ST_)                                                                                           \
  do {                                                                                                                                  \
    if (spec == 1) hipLaunchKernelGGL((gather_gemm_bf16_glds_kernel<BM_, BN_, WM_, WN_, 1, ST_>), grid, dim3(64 * WM_ * WN_), 0, stream, a); \
    else hipLaunchKernelGGL((gather_gemm_bf16_glds_kernel<BM_, BN_, WM_, WN_, 3, ST_>), grid, dim3(64 * WM_ * WN_), 0, stream, a);          \
  } while (0)
#define MOPOE_LAUNCH_GX(BM_, BN_, WM_, WN_, ST_) hipLaunchKernelGGL((gather_gemm_bf16_glds_kernel<BM_, BN_, WM_, WN_, 2, ST_>), grid, dim3(64 * WM_ * WN_), 0, stream, a)
    if (cfg >= 5) {
      if (spec == 2) {
        if (cfg == 5) MOPOE_LAUNCH_GX(128, 128, 2, 2, 2);
        else if (cfg == 7) MOPOE_LAUNCH_GX(256, 128, 4, 2, 2);
        else if (cfg == 9) MOPOE_LAUNCH_GX(128, 64, 4, 1, 2);
        else MOPOE_LAUNCH_GX(64, 64, 2, 2, 4);
      }
      else if (cfg == 5) MOPOE_LAUNCH_G(128, 128, 2, 2, 2);
      else if (cfg == 6) MOPOE_LAUNCH_G(128, 128, 2, 2, 3);
      else if (cfg == 7) MOPOE_LAUNCH_G(256, 128, 4, 2, 2);
      else if (cfg == 9) MOPOE_LAUNCH_G(128, 64, 4, 1, 2);
      else if (cfg == 10) MOPOE_LAUNCH_G(128, 64, 4, 1, 3);
      else MOPOE_LAUNCH_G(64, 64, 2, 2, 4);
      if (int rc = check_launch("gather_gemm_bf16_glds")) return rc;
      return MOPOE_OK;
    }
#undef MOPOE_LAUNCH_G
#define MOPOE_LAUNCH_H(BM_, BN_, WM_, WN_)                                                                                          \
  do {                                                                                                                            \
    if (spec == 1) hipLaunchKernelGGL((gather_gemm_bf16_kernel<BM_, BN_, WM_, WN_, 1>), grid, dim3(64 * WM_ * WN_), 0, stream, a);      \
    else if (spec == 2) hipLaunchKernelGGL((gather_gemm_bf16_kernel<BM_, BN_, WM_, WN_, 2>), grid, dim3(64 * WM_ * WN_), 0, stream, a); \
    else hipLaunchKernelGGL((gather_gemm_bf16_kernel<BM_, BN_, WM_, WN_, 3>), grid, dim3(64 * WM_ * WN_), 0, stream, a);                \
  } while (0)
    if (cfg == 0) MOPOE_LAUNCH_H(128, 128, 2, 2);
    else if (cfg == 1) MOPOE_LAUNCH_H(256, 64, 4, 1);
    else if (cfg == 3) MOPOE_LAUNCH_H(256, 128, 4, 2);
    else if (cfg == 4) MOPOE_LAUNCH_H(128, 64, 4, 1);
    else MOPOE_LAUNCH_H(64, 64, 2, 2);
#undef MOPOE_LAUNCH_H
    if (int rc = check_launch("gather_gemm_bf16")) return rc;
  }
  return MOPOE_OK;
}

}  // namespace mopoe

using namespace mopoe;

extern "C" int mopoe_conv_fwd_bf16(const uint16_t* x, const uint16_t* wp, const float* bias, void* y, int32_t y_is_f32,
                                   const mopoe_conv_geom* g, const mopoe_bn_ref* bn_in, const mopoe_mask_ref* mask,
                                   double* out_stats, const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  if (int rc = validate_geom(g)) return rc;
  if (!x || !wp || !y) { set_error("conv_fwd_bf16: null pointer"); return MOPOE_ERR_ARG; }
  return launch_gather_bf16(x, wp, bias, y, y_is_f32, g, g->transposed ? 0 : 1, g->Cin, g->Cout, /*w_nk=*/0, bn_in, mask,
                            out_stats, nullptr, nullptr, nullptr, plan, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mopoe_conv_fwd_mix_bf16(const uint16_t* x, const uint16_t* wp, const float* bias, uint16_t* y,
                                       const mopoe_conv_geom* g, const mopoe_bn_ref* bn_in, const mopoe_mask_ref* mask,
                                       const mopoe_mix_ref* mix, double* out_stats, const mopoe_conv_plan* plan,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = validate_geom(g)) return rc;
  if (!x || !wp || !y || !mix) { set_error("conv_fwd_mix_bf16: null pointer"); return MOPOE_ERR_ARG; }
  return launch_gather_bf16(x, wp, bias, y, 0, g, g->transposed ? 0 : 1, g->Cin, g->Cout, /*w_nk=*/0, bn_in, mask,
                            out_stats, nullptr, nullptr, nullptr, plan, workspace, workspace_bytes, (hipStream_t)stream, mix);
}

extern "C" int mopoe_conv_dgrad_bf16(const uint16_t* dy, const uint16_t* wp, void* dx, int32_t dx_is_f32,
                                     const mopoe_conv_geom* g, const mopoe_bn_ref* relu_bn, const uint16_t* xin,
                                     double* bwd_sums, const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  if (int rc = validate_geom(g)) return rc;
  if (!dy || !wp || !dx) { set_error("conv_dgrad_bf16: null pointer"); return MOPOE_ERR_ARG; }
  return launch_gather_bf16(dy, wp, nullptr, dx, dx_is_f32, g, g->transposed ? 1 : 0, g->Cout, g->Cin, /*w_nk=*/1, nullptr,
                            nullptr, nullptr, relu_bn, xin, bwd_sums, plan, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mopoe_conv_wgrad_bf16(const uint16_t* x, const uint16_t* dy, float* dwp, const mopoe_conv_geom* g,
                                     const mopoe_bn_ref* bn_in, int32_t dwp_is_zero, const mopoe_conv_plan* plan,
                                     void* stream_) {
  if (int rc = validate_geom(g)) return rc;
  if (!x || !dy || !dwp) { set_error("conv_wgrad_bf16: null pointer"); return MOPOE_ERR_ARG; }
  hipStream_t stream = (hipStream_t)stream_;
  WgradArgsH a;
  a.Xs = x; a.Dy = dy; a.dW = dwp;
  a.N = g->N; a.Hs = g->Hs; a.Ws = g->Ws; a.Hb = g->Hb; a.Wb = g->Wb; a.Cin = g->Cin; a.Cout = g->Cout;
  a.kh = g->kh; a.kw = g->kw; a.sh = g->sh; a.sw = g->sw; a.ph = g->ph; a.pw = g->pw;
  a.x_is_big = g->transposed ? 0 : 1;
  a.Ms = (long)g->N * g->Hs * g->Ws;
  mopoe_bn_ref none = {};
  a.bn_in = bn_in ? *bn_in : none;
  if (a.bn_in.mode != 0 && (a.bn_in.C != g->Cin || g->Cin > MAX_BN_C)) { set_error("wgrad bn_in channel mismatch"); return MOPOE_ERR_ARG; }
  if (g->Cin % 8 != 0 || g->Cout % 8 != 0) { set_error("bf16 wgrad: channel counts must be multiples of 8 (%d, %d)", g->Cin, g->Cout); return MOPOE_ERR_ARG; }
  const size_t rows_x = (size_t)g->N * (g->transposed ? g->Hs * g->Ws : g->Hb * g->Wb);
  const size_t rows_dy = (size_t)g->N * (g->transposed ? g->Hb * g->Wb : g->Hs * g->Ws);
  const size_t xb = rows_x * g->Cin * sizeof(bf16_t), db = rows_dy * g->Cout * sizeof(bf16_t);
  if (!aligned16(x) || !aligned16(dy) || xb >= (1ull << 31) || db >= (1ull << 31)) {
    set_error("bf16 wgrad: tensors must be 16-byte aligned and smaller than 2 GiB");
    return MOPOE_ERR_ARG;
  }
  a.x_bytes = (unsigned)xb;
  a.dy_bytes = (unsigned)db;
  a.lg_ws = ilog2_exact(g->Ws);
  a.lg_hw = ilog2_exact((long)g->Hs * g->Ws);
  const bool pow2 = a.lg_ws >= 0 && a.lg_hw >= 0;
  const int taps = g->kh * g->kw;
  bool big = g->Cin > 64 && g->Cout > 64;
  if (plan && (plan->tile == 2 || plan->tile == 6)) big = false;
  if (plan && plan->tile == 5) {
    // the 128 x 128 LDS-DMA tile on a layer with <= 64 channels on one side would run half-empty MFMA tiles: refused, as the
    // tuner never offers it (mimic_amd/ops.py: _wgrad_candidates)
    if (!big) { set_error("bf16 wgrad plan: tile 5 (128x128 on LDS-DMA) needs more than 64 channels on both sides (%d, %d)", g->Cin, g->Cout); return MOPOE_ERR_ARG; }
  }
  // plan tiles: 0 = 128x128, 2 = 64x64 (register-staged, 32 pixels per chunk); 5 = 128x128, 6 = 64x64 on LDS-DMA (64 pixels
  // per stage).  Default: the LDS-DMA form (MOPOE_BF16_NO_GLDS = the register-staged one).
  static const bool glds_default = !getenv("MOPOE_BF16_NO_GLDS");
  const bool glds = plan && plan->tile >= 0 ? plan->tile >= 5 : glds_default;
  // tile 7: the 128 tile on LDS-DMA with TWO taps per block on the side of the gathered operand when that side has 64 channels
  // (conv_gemm_bf16_glds.inc, MERGE): convs with Cin = 64 (1), transposed convs with Cout = 64 (2); plain operands, even tap count
  const bool xf_ = a.bn_in.mode != 0;
  const int merge_ok = (!xf_ && taps % 2 == 0) ? (a.x_is_big ? (g->Cin == 64 ? 1 : 0) : (g->Cout == 64 ? 2 : 0)) : 0;
  if (plan && plan->tile > 9) { set_error("bf16 wgrad plan: tile %d (0, 2, 5, 6, 7, 8, 9)", plan->tile); return MOPOE_ERR_ARG; }
  // tiles 8 / 9: four taps (one parity class of a k4 s2 p1 kernel) per block, 64 gathered channels x 64 / 128 channels of the
  // small-grid operand (conv_gemm_bf16_glds.inc: wgrad_parity_bf16_kernel)
  if (plan && plan->tile >= 8) {
    const bool ok = !xf_ && g->kh == 4 && g->kw == 4 && g->sh == 2 && g->sw == 2 && g->ph == 1 && g->pw == 1 &&
                    g->Hs % 8 == 0 && g->Ws % 8 == 0 && g->Hb == 2 * g->Hs && g->Wb == 2 * g->Ws;
    if (!ok) {
      set_error("bf16 wgrad plan: tiles 8 / 9 (four taps per block) need a plain operand, k4 s2 p1 and a small grid of whole 8 x 8 tiles");
      return MOPOE_ERR_ARG;
    }
    const int Cg = a.x_is_big ? g->Cin : g->Cout, Csm = a.x_is_big ? g->Cout : g->Cin;
    const int cs = plan->tile == 9 ? 128 : 64;
    if (cs == 128 && Csm % 128 != 0) { set_error("bf16 wgrad plan: tile 9 needs a multiple of 128 channels on the small-grid operand"); return MOPOE_ERR_ARG; }
    const long ntiles = (long)g->N * (g->Hs / 8) * (g->Ws / 8);
    const long cblocks = (long)ceil_div(Cg, 64) * ceil_div(Csm, cs) * 4;
    long split = plan->split > 0 ? plan->split : (512 + cblocks - 1) / cblocks;
    if (split > ntiles) split = ntiles;
    if (split < 1) split = 1;
    a.chunk = (ntiles + split - 1) / split;             // (in 8 x 8 tiles)
    split = (ntiles + a.chunk - 1) / a.chunk;
    a.atomic = split > 1;
    static const bool xcd_remap8 = !getenv("MOPOE_NO_XCD_REMAP");
    a.xcd_remap = xcd_remap8 ? 1 : 0;
    if (a.atomic && !dwp_is_zero) {
      if (hipMemsetAsync(dwp, 0, (size_t)taps * g->Cin * g->Cout * sizeof(float), stream) != hipSuccess) { set_error("wgrad memset failed"); return MOPOE_ERR_LAUNCH; }
    }
    ProfScope prof(stream, 2.0 * (double)a.Ms * g->Cin * (double)g->Cout * taps, PROF_BF16_WGRAD_PARITY + (cs == 128 ? 1 : 0), (double)xb + (double)db);
    const dim3 grid((unsigned)(ceil_div(Cg, 64) * ceil_div(Csm, cs)), 4, (unsigned)split);
    if (cs == 128 && a.x_is_big) hipLaunchKernelGGL((wgrad_parity_bf16_kernel<128, true>), grid, dim3(256), 0, stream, a);
    else if (cs == 128) hipLaunchKernelGGL((wgrad_parity_bf16_kernel<128, false>), grid, dim3(256), 0, stream, a);
    else if (a.x_is_big) hipLaunchKernelGGL((wgrad_parity_bf16_kernel<64, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((wgrad_parity_bf16_kernel<64, false>), grid, dim3(256), 0, stream, a);
    return check_launch("wgrad_parity_bf16 (four taps per block)");
  }
  if (plan && plan->tile == 7 && !merge_ok) {
    set_error("bf16 wgrad plan: tile 7 (two taps per block) needs a plain operand, an even tap count and 64 channels on the gathered side");
    return MOPOE_ERR_ARG;
  }
  const int merge = plan && plan->tile == 7 ? merge_ok : 0;
  if (merge) big = true;
  const int T = big ? 128 : 64;
  const int nI = merge == 1 ? 1 : ceil_div(g->Cin, T), nJ = merge == 2 ? 1 : ceil_div(g->Cout, T);
  a.nJ = nJ;
  const long tiles = (long)nI * nJ * (merge ? taps / 2 : taps);
  long split = (1024 + tiles - 1) / tiles;
  if (plan && plan->split > 0) split = plan->split;
  const int kp = glds ? 64 : BKH;                 // pixels per chunk
  const long max_split = (a.Ms + 4 * kp - 1) / (4 * kp);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  long chunk = (a.Ms + split - 1) / split;
  chunk = (chunk + kp - 1) / kp * kp;
  split = (a.Ms + chunk - 1) / chunk;
  a.chunk = chunk;
  a.atomic = split > 1;
  static const bool xcd_remap = !getenv("MOPOE_NO_XCD_REMAP");   // (A/B switch)
  a.xcd_remap = xcd_remap ? 1 : 0;
  const size_t bytes = (size_t)taps * g->Cin * g->Cout * sizeof(float);
  if (a.atomic && !dwp_is_zero) {
    if (hipMemsetAsync(dwp, 0, bytes, stream) != hipSuccess) { set_error("wgrad memset failed"); return MOPOE_ERR_LAUNCH; }
  }
  const double flops = 2.0 * (double)a.Ms * g->Cin * (double)g->Cout * taps;
  const bool xf = a.bn_in.mode != 0;
  ProfScope prof(stream, flops, merge ? PROF_BF16_WGRAD_GLDS_MERGE + merge - 1
                                      : (glds ? PROF_BF16_WGRAD_GLDS : PROF_BF16_WGRAD) + (big ? 0 : 2) + (xf ? 1 : 0), (double)xb + (double)db);
  dim3 grid(nI * nJ, merge ? taps / 2 : taps, (unsigned)split);
  if (merge) {
    if (merge == 1 && pow2) hipLaunchKernelGGL((wgrad_gemm_bf16_glds_kernel<128, false, true, 1>), grid, dim3(256), 0, stream, a);
    else if (merge == 1) hipLaunchKernelGGL((wgrad_gemm_bf16_glds_kernel<128, false, false, 1>), grid, dim3(256), 0, stream, a);
    else if (pow2) hipLaunchKernelGGL((wgrad_gemm_bf16_glds_kernel<128, false, true, 2>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((wgrad_gemm_bf16_glds_kernel<128, false, false, 2>), grid, dim3(256), 0, stream, a);
    return check_launch("wgrad_gemm_bf16_glds (two taps per block)");
  }
  if (glds) {
#define MOPOE_LAUNCH_WG(T_)                                                                                             \
  do {                                                                                                                 \
    if (xf && pow2) hipLaunchKernelGGL((wgrad_gemm_bf16_glds_kernel<T_, true, true>), grid, dim3(256), 0, stream, a);      \
    else if (xf) hipLaunchKernelGGL((wgrad_gemm_bf16_glds_kernel<T_, true, false>), grid, dim3(256), 0, stream, a);        \
    else if (pow2) hipLaunchKernelGGL((wgrad_gemm_bf16_glds_kernel<T_, false, true>), grid, dim3(256), 0, stream, a);      \
    else hipLaunchKernelGGL((wgrad_gemm_bf16_glds_kernel<T_, false, false>), grid, dim3(256), 0, stream, a);               \
  } while (0)
    if (big) MOPOE_LAUNCH_WG(128);
    else MOPOE_LAUNCH_WG(64);
#undef MOPOE_LAUNCH_WG
    return check_launch("wgrad_gemm_bf16_glds");
  }
#define MOPOE_LAUNCH_WH(T_)                                                                                              \
  do {                                                                                                                 \
    if (xf && pow2) hipLaunchKernelGGL((wgrad_gemm_bf16_kernel<T_, T_, true, true>), grid, dim3(256), 0, stream, a);        \
    else if (xf) hipLaunchKernelGGL((wgrad_gemm_bf16_kernel<T_, T_, true, false>), grid, dim3(256), 0, stream, a);          \
    else if (pow2) hipLaunchKernelGGL((wgrad_gemm_bf16_kernel<T_, T_, false, true>), grid, dim3(256), 0, stream, a);        \
    else hipLaunchKernelGGL((wgrad_gemm_bf16_kernel<T_, T_, false, false>), grid, dim3(256), 0, stream, a);                 \
  } while (0)
  if (big) MOPOE_LAUNCH_WH(128);
  else MOPOE_LAUNCH_WH(64);
#undef MOPOE_LAUNCH_WH
  return check_launch("wgrad_gemm_bf16");
}

// ---- image-side edge layers with the wide tensor in bf16 (the single-channel image, the taps and the tap gradients
// stay fp32): stem forward / head input gradient, stem / head weight gradient, head forward ------------------------------
extern "C" int mopoe_edge_expand_bf16(const float* scal, const float* w, uint16_t* out, const mopoe_conv_geom* g, int32_t C,
                                      double* stats, void* stream) {
  if (int rc = validate_geom(g)) return rc;
  if (!scal || !w || !out || !edge_supported(g, C, {w, out})) { set_error("edge_expand_bf16: needs k3 s2, C %% 4 == 0, aligned tensors"); return MOPOE_ERR_ARG; }
  return edge_expand<bf16_t>(scal, w, out, g, C, stats, (hipStream_t)stream);
}
extern "C" int mopoe_edge_wgrad_bf16(const uint16_t* vec, const float* scal, float* dw, const mopoe_conv_geom* g, int32_t C, int32_t dw_is_zero,
                                     void* stream) {
  if (int rc = validate_geom(g)) return rc;
  if (!vec || !scal || !dw || !edge_supported(g, C, {vec, dw})) { set_error("edge_wgrad_bf16: needs k3 s2, C %% 4 == 0, aligned tensors"); return MOPOE_ERR_ARG; }
  return edge_wgrad<bf16_t>(vec, scal, dw, g, C, (hipStream_t)stream, dw_is_zero != 0);
}
extern "C" int mopoe_edge_reduce_bf16(const uint16_t* x, const float* w, const float* bias, float* out,
                                      const mopoe_conv_geom* g, int32_t C, void* stream) {
  if (int rc = validate_geom(g)) return rc;
  if (!x || !w || !out || !edge_supported(g, C, {x, w})) { set_error("edge_reduce_bf16: needs k3 s2, C %% 4 == 0, aligned tensors"); return MOPOE_ERR_ARG; }
  return edge_reduce<bf16_t>(x, w, bias, out, g, C, (hipStream_t)stream);
}
