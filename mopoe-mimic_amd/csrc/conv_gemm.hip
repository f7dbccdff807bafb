// Implicit-GEMM convolution family on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// All activations are channels-last row matrices [pixels][C]; weights are packed Wp[tap][Cin][Cout].
//
//   gather_gemm  : Y[M][Cn] = sum_taps  T(X)[gather(m, tap)][Ck] * W_tap[Ck][Cn]
//       form 0 ("gather from the big grid"): Y lives on the conv's small grid, X on the big grid.
//               conv forward, convT input-gradient.
//       form 1 ("sub-pixel phases"):        Y lives on the big grid, X on the small grid; the output
//               pixels are processed phase by phase (oy % sh, ox % sw) so that only taps that hit a
//               real input pixel are multiplied.  convT forward, conv input-gradient.
//   wgrad_gemm   : dWp[tap][Cin][Cout] = sum_pixels T(x)[..][Cin]^T * dy[..][Cout]
//
// Blocks are 4 or 8 waves; a wave owns a (BM/WGM)x(BN/WGN) sub-tile made of 32x32 MFMA tiles; K is consumed in
// chunks of 16 (or 32) through double-buffered LDS tiles stored K-major ([k][m], [k][n]) so that every MFMA operand
// read is a conflict-free ds_read of consecutive lanes.  BatchNorm+ReLU of the operand, bias, dropout mask, BN
// statistics and the ReLU/BN backward reductions are fused into the operand load / epilogue (include/mopoe_hip.h).
//
// What shapes the main loops (measured): VALU work does not hide beside fp32 MFMAs on this
// chip, so the loops carry as few vector instructions per MFMA as possible -- the K advance of every operand is a
// scalar add in the buffer load's soffset, per-thread byte offsets change only with the tap (gather) or never
// (weights, wgrad's pixel slots), out-of-range rows read zeros through the hardware range check (voffset 2^31),
// LDS addresses are base + immediate (loop unrolled by two), and the mode flags are template specialisations.
// The tile and the split of the reduction are launch-plan arguments (mopoe_conv_plan): the host mirror measures
// the candidates per layer; the heuristics in launch_gather / mopoe_conv_wgrad are only the fallback.
#include "gemm_common.hpp"

namespace mopoe {

constexpr int BK = 16;            // K-chunk of the wgrad kernel and of the 64x64 gather tile
#ifndef TILE_N64_REMAINDER
#define TILE_N64_REMAINDER 64
#endif
#ifndef SPLIT_BLOCKS
#define SPLIT_BLOCKS 256   // grids below this many blocks split the tap x channel reduction
#endif
#ifndef SPLIT_TARGET
#define SPLIT_TARGET 512   // ... until about this many blocks are in flight
#endif
#ifndef GEMM_8WAVES
#define GEMM_8WAVES 1          // 512-thread blocks (8 waves, 64x32 per wave): 4 waves per SIMD hide the load/store phases
#endif
#ifndef PERSIST_BLOCKS
// upper bound on blocks of one launch (each block walks the M tiles it owns).  512 = exactly the 2 x 8-wave blocks a CU
// holds: fastest for a kernel that is alone on the chip, but inside the replayed step such a kernel keeps every CU for its
// whole duration and the other branches' small kernels wait.  With 2048 blocks retire 4x as often: C2 step +1...2 %
// (5250-5260 -> 5305-5370 samples/s; 4096: 5339, 16384: 5324); the bf16 family measured the other way and stays at 512.
#define PERSIST_BLOCKS 2048
#endif
#ifndef GEMM_MIN_WAVES
#define GEMM_MIN_WAVES 3
#endif
#ifndef GEMM_BK_BIG
#define GEMM_BK_BIG 16           // K-chunk of the 128x128 / 256x64 gather tiles
#endif
#ifndef GEMM_LDS_PAD
#define GEMM_LDS_PAD 4
#endif
constexpr int LDS_PAD = GEMM_LDS_PAD;

struct GemmArgs {
  const float* X;
  const float* W;
  float* Y;
  const float* bias;
  int N, Hx, Wx, Hy, Wy, Ck, Cn, Cin_w, Cout_w;
  int kh, kw, sh, sw, ph, pw;
  int form, w_nk;
  int Hq, Wq;            // rows of one phase: N*Hq*Wq (form 0: Hq=Hy, Wq=Wy)
  long rows_per_phase;
  int vecA, vecB;        // 16-byte loads legal for the activation / weight operand
  mopoe_bn_ref bn_in;
  mopoe_mask_ref mask;
  double* out_stats;
  mopoe_bn_ref relu_bn;
  const float* xin;
  double* bwd_sums;
  unsigned x_bytes, w_bytes;   // sizes of X and W for the buffer descriptors (vector path)
  int nsplit;            // split-K factor (1 = none)
  float* partial;        // split-K: [nsplit][rows_total][Cn] raw partial sums (else nullptr)
  int* counters;         // vector path: one arrival counter per output tile of a split reduction (zero on entry, left zero)
  long rows_total;       // N*Hy*Wy
  int mix;               // forward only (vector path): y = mix_a * bn(xin) + mix_b * mask * (conv + bias); relu_bn carries bn
  float mix_a, mix_b;
  int xcd_remap;         // 1: XCD-aware block numbering
};

// storage helpers of gemm_epilogue_rows.inc for this family: fp32 results and fp32 xin, rows of Cn floats; a lane's 8
// columns are two float4 halves, the upper one may lie past Cn (Cn % 4 == 0 on the vector path)
__device__ __forceinline__ float epi_round(const GemmArgs&, float x) { return x; }
struct EpiXinRaw { float4 lo, hi; };
constexpr bool EPI_PREFETCH_XIN = false;
__device__ __forceinline__ EpiXinRaw epi_xin_ld(const GemmArgs& a, long yrow, int ncol, bool ok_hi) {
  const float* p = a.xin + yrow * a.Cn + ncol;
  EpiXinRaw r;
  r.lo = *reinterpret_cast<const float4*>(p);
  r.hi = ok_hi ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  return r;
}
__device__ __forceinline__ void epi_xin_unpack(const EpiXinRaw& r, float (&xi)[8]) {
  xi[0] = r.lo.x; xi[1] = r.lo.y; xi[2] = r.lo.z; xi[3] = r.lo.w; xi[4] = r.hi.x; xi[5] = r.hi.y; xi[6] = r.hi.z; xi[7] = r.hi.w;
}
__device__ __forceinline__ void epi_store8(const GemmArgs& a, long yrow, int ncol, bool ok_hi, const float (&v)[8]) {
  float* dst = a.Y + yrow * a.Cn + ncol;
  *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  if (ok_hi) *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
}


// 16-byte buffer load with hardware bounds check: no branch, no exec masking, zero for off >= bytes
__device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t srd, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// same with a wave-uniform byte offset in the instruction's scalar-offset slot: the per-iteration advance along K
// costs one SALU add instead of one VALU add per load (VALU work does not hide beside fp32 MFMAs on this chip:
// tests/tools/mfma_peak.hip measures +3.4 cycles per VALU op on a 64-cycle MFMA, from any wave of the SIMD)
__device__ __forceinline__ float4 bld4s(__amdgpu_buffer_rsrc_t srd, unsigned byte_off, unsigned s_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, byte_off, s_off, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float4 ld4(const float* p, int nvalid, bool vec) {
  // loads up to 4 consecutive floats starting at p; elements >= nvalid are zero
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid >= 4 && vec) {
    v = *reinterpret_cast<const float4*>(p);
  } else {
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
  }
  return v;
}

// Tile configuration: BM x BN block tile, WGM x WGN waves (WGM*WGN == 4), each wave owns a
// (BM/WGM) x (BN/WGN) sub-tile of 32x32 MFMA tiles.
// SPEC fixes the three mode flags of the main loop at compile time (0 = all of them at run time):
//   1 forward, plain operand   2 forward, BN+ReLU on the operand   3 input gradient (weights K-contiguous)
// each with the fast K addressing (Ck a multiple of the chunk).  Without it the loop carries the untaken
// variants' register copies and zero-fills as VALU work, which on this chip does not hide beside fp32 MFMAs.
template <int BM, int BN, int WGM, int WGN, int GBK, bool VEC, int SPEC = 0>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN == 4 ? GEMM_MIN_WAVES : ((BM / WGM) * (BN / WGN) > 64 * 32 ? 2 : 4)))
void gather_gemm_kernel(const GemmArgs a) {
  static_assert(SPEC == 0 || VEC, "specialised loops exist for the vector path only");
  constexpr int NT = 64 * WGM * WGN;            // threads per block (4 or 8 waves)
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves per block");
  constexpr int WM = BM / WGM, WN = BN / WGN;     // wave tile
  constexpr int TI = WM / 32, TJ = WN / 32;       // MFMA tiles per wave
  constexpr int A_LD = BM + LDS_PAD, B_LD = BN + LDS_PAD;
  constexpr int KQ = GBK / 4;                     // float4 per tile row along K
  constexpr int RPP = NT / KQ;                    // tile rows covered by one pass of the block
  constexpr int A_PER_THR = BM / RPP;             // float4 loads per thread for the A tile
  static_assert(BM % RPP == 0, "A tile rows must be a multiple of the rows covered per pass");
  constexpr int B_PER_THR_NK = (BN + RPP - 1) / RPP;   // weights with K contiguous (dgrad)
  constexpr int N4 = BN / 4;                      // weights with N contiguous (forward): float4 per k-row
  constexpr int B_PER_THR_KN = (GBK * N4 + NT - 1) / NT;
  constexpr int B_PER_THR = B_PER_THR_NK > B_PER_THR_KN ? B_PER_THR_NK : B_PER_THR_KN;

  // operand tiles As[2][GBK][A_LD], Bs[2][GBK][B_LD]; the vector path's epilogue overlays them with one
  // [32][WN + 4] fp32 staging patch per wave (gemm_epilogue_rows.inc)
  constexpr int STG_LD = WN + 4;
  constexpr int A_FLOATS = 2 * GBK * A_LD, B_FLOATS = 2 * GBK * B_LD;
  constexpr int STG_FLOATS = VEC ? (NT / 64) * 32 * STG_LD : 0;
  constexpr int SMEM_FLOATS = (A_FLOATS + B_FLOATS) > STG_FLOATS ? (A_FLOATS + B_FLOATS) : STG_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  __shared__ __attribute__((aligned(16))) float bnS[MAX_BN_C];
  __shared__ __attribute__((aligned(16))) float bnT[MAX_BN_C];
  __shared__ __attribute__((aligned(16))) float epi[5][VEC ? BN : 4];   // per block column: mean, rstd, scale, shift (relu_bn), bias
  __shared__ int s_ticket;
  float* const As0 = smem;
  float* const Bs0 = smem + A_FLOATS;
#define AS_(buf, k, r) As0[((buf) * GBK + (k)) * A_LD + (r)]
#define BS_(buf, k, r) Bs0[((buf) * GBK + (k)) * B_LD + (r)]

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
  const int nkc = (a.Ck + GBK - 1) / GBK;
  const int total_all = nty * ntx * nkc;
  // buffer descriptors (wave-uniform: kernel arguments only) for the branch-free vector path
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (int)a.w_bytes, 0x00020000);
  // split-K: this block reduces iterations [it_beg, it_end) of the flattened (tap, k-chunk) space
  const int per_split = (total_all + a.nsplit - 1) / a.nsplit;
  const int it_beg = split * per_split;
  const int it_end = it_beg + per_split < total_all ? it_beg + per_split : total_all;
  const int total = it_end > it_beg ? it_end - it_beg : 0;

  // ---- BN(+ReLU) table for the operand transform -----------------------------------------------
  const bool xform = SPEC ? (SPEC == 2) : (a.bn_in.mode != 0);
  const int w_nk = SPEC ? (SPEC == 3 ? 1 : 0) : a.w_nk;
  if (xform) {
    for (int c = tid; c < ((a.Ck + 3) & ~3); c += NT) {
      BnC k = BnC{0.f, 0.f, 0.f, 0.f};
      if (c < a.Ck) k = bn_coef(a.bn_in, c);
      bnS[c] = k.scale;
      bnT[c] = k.shift;
    }
    __syncthreads();
  }

  const int kq = tid % KQ;
  const int trow = tid / KQ;
  const int l31 = lane & 31, lhi = lane >> 5;
  const bool do_relu_bn = a.relu_bn.mode != 0 && !a.mix;
  const int hw = a.Hq * a.Wq;

  // per-column epilogue constants and running column sums (persist across this block's M tiles)
  static_assert(TJ <= 8, "column sums live in s1[8] / s2[8] on both epilogue paths");
  float cbias[TJ], s1[8], s2[8];
  BnC rbc[TJ];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {   // scalar path: per-column constants of the MFMA-layout epilogue (gemm_epilogue.inc)
    const int n = n0 + wn * WN + j * 32 + l31;
    cbias[j] = (!VEC && n < a.Cn && a.bias && !a.partial) ? a.bias[n] : 0.f;
    rbc[j] = BnC{0.f, 0.f, 0.f, 0.f};
    if (!VEC && n < a.Cn && do_relu_bn && !a.partial) rbc[j] = bn_coef(a.relu_bn, n);
  }
  // vector path: row-major epilogue (gemm_epilogue_rows.inc): LPR lanes share one output row, each owns 8 columns
  constexpr int E_LPR = WN / 8, E_RPP = 64 / E_LPR, E_NPASS = 32 / E_RPP;
  const int c8 = lane % E_LPR, rsub = lane / E_LPR;
  const int ecol = wn * WN + c8 * 8;
  const int ncol = n0 + ecol;
  const bool ok_lo = ncol < a.Cn, ok_hi = ncol + 4 < a.Cn;
  float* const stg = smem + wave * 32 * STG_LD;
  bool was_last = false;
  if constexpr (VEC) {
    for (int c = tid; c < BN; c += NT) {
      const int n = n0 + c;
      BnC k = BnC{0.f, 0.f, 0.f, 0.f};
      if (n < a.Cn && (do_relu_bn || a.mix)) k = bn_coef(a.relu_bn, n);
      if (a.mix) { k.scale *= a.mix_a; k.shift *= a.mix_a; }
      epi[0][c] = k.mean; epi[1][c] = k.rstd; epi[2][c] = k.scale; epi[3][c] = k.shift;
      epi[4][c] = (n < a.Cn && a.bias) ? a.bias[n] : 0.f;
    }
    __syncthreads();
  }

  // fast operand addressing (vector path, Ck a multiple of the K chunk): per-thread byte offsets are fixed per tap
  // (A) or for the whole kernel (B); the K advance lives in the scalar offset of the buffer load
  const bool fastk = SPEC ? true : (VEC && (a.Ck % GBK == 0));
  unsigned voffB[B_PER_THR];
#pragma unroll
  for (int i = 0; i < B_PER_THR; ++i) voffB[i] = OOB;
  if (w_nk == 0) {
#pragma unroll
    for (int i = 0; i < B_PER_THR_KN; ++i) {
      const int k = tid / N4 + i * (NT / N4);
      const int n = n0 + (tid % N4) * 4;
      if ((k < GBK) & (n < a.Cn)) voffB[i] = ((unsigned)k * (unsigned)a.Cout_w + (unsigned)n) * 4u;
    }
  } else {
#pragma unroll
    for (int i = 0; i < B_PER_THR_NK; ++i) {
      const int n = n0 + trow + i * RPP;
      if ((trow + i * RPP < BN) & (n < a.Cn)) voffB[i] = ((unsigned)n * (unsigned)a.Cout_w + (unsigned)kq * 4u) * 4u;
    }
  }

  const long nMt = (a.rows_per_phase + BM - 1) / BM;
  for (long mt = lbx; mt < nMt; mt += gridDim.x) {
    const long m0 = mt * BM;

    // ---- per-thread A rows -------------------------------------------------------------------------
    int rn[A_PER_THR], ry0[A_PER_THR], rx0[A_PER_THR], rbase[A_PER_THR];
    bool rvalid[A_PER_THR];
#pragma unroll
    for (int i = 0; i < A_PER_THR; ++i) {
      const long m = m0 + trow + i * RPP;
      rvalid[i] = m < a.rows_per_phase;
      const unsigned mm = rvalid[i] ? (unsigned)m : 0u;   // rows_per_phase < 2^31 (checked on the host)
      const int n = (int)(mm / (unsigned)hw);
      const int rem = (int)(mm - (unsigned)n * (unsigned)hw);
      const int qy = rem / a.Wq, qx = rem - qy * a.Wq;
      rn[i] = n;
      if (a.form == 0) { ry0[i] = qy * a.sh - a.ph; rx0[i] = qx * a.sw - a.pw; }
      else             { ry0[i] = qy + cy;          rx0[i] = qx + cx; }
      rbase[i] = (n * a.Hx + ry0[i]) * a.Wx + rx0[i];   // source row of tap (0,0); < 2^31 (checked on the host)
    }

    float4 ra[A_PER_THR], rb[B_PER_THR];
    // BN+ReLU of the operand is applied when the registers are written to LDS (store_tiles), i.e. AFTER the
    // MFMA block that the global loads overlap with -- applying it at load time would put a vmcnt(0) wait
    // right behind every load
    float4 pend_sc = make_float4(0.f, 0.f, 0.f, 0.f), pend_sh = pend_sc;
    bool pend_ok[A_PER_THR];

    // (tap, k-chunk) position of the NEXT load: advanced incrementally; the per-tap part (source pixel of every
    // row, bounds, weight slice) is recomputed only when the tap changes, the per-chunk part is one add
    int ld_tap = it_beg / nkc;
    int ld_kc = (it_beg - ld_tap * nkc) * GBK;
    bool tap_dirty = true;
    unsigned offA[A_PER_THR], offW = 0;
    bool okA[A_PER_THR];

    auto load_tiles = [&](int /*it*/) {
      if (tap_dirty) {
        const int jy = ld_tap / ntx, jx = ld_tap - jy * ntx;
        const int wtap = (ky0 + kstep_y * jy) * a.kw + (kx0 + kstep_x * jx);
        const int tapoff = dsgn * (jy * a.Wx + jx);
#pragma unroll
        for (int i = 0; i < A_PER_THR; ++i) {
          const int iy = ry0[i] + dsgn * jy, ix = rx0[i] + dsgn * jx;
          okA[i] = rvalid[i] & ((unsigned)iy < (unsigned)a.Hx) & ((unsigned)ix < (unsigned)a.Wx);
          offA[i] = (unsigned)(rbase[i] + tapoff) * (unsigned)a.Ck * 4u;
        }
        offW = (unsigned)wtap * (unsigned)a.Cin_w * (unsigned)a.Cout_w * 4u;
        if (fastk) {
#pragma unroll
          for (int i = 0; i < A_PER_THR; ++i) {
            offA[i] = okA[i] ? offA[i] + (unsigned)kq * 16u : OOB;
            pend_ok[i] = okA[i];
          }
        }
        tap_dirty = false;
      }
      if (fastk) {
        const unsigned sA = (unsigned)ld_kc * 4u;
        const unsigned sB = offW + (w_nk == 0 ? (unsigned)ld_kc * (unsigned)a.Cout_w * 4u : (unsigned)ld_kc * 4u);
        if (xform) {
          pend_sc = *reinterpret_cast<const float4*>(&bnS[ld_kc + kq * 4]);
          pend_sh = *reinterpret_cast<const float4*>(&bnT[ld_kc + kq * 4]);
        }
        if constexpr (VEC) {
#pragma unroll
          for (int i = 0; i < A_PER_THR; ++i) ra[i] = bld4s(srdX, offA[i], sA);
          if (w_nk == 0) {
#pragma unroll
            for (int i = 0; i < B_PER_THR_KN; ++i) rb[i] = bld4s(srdW, voffB[i], sB);
          } else {
#pragma unroll
            for (int i = 0; i < B_PER_THR_NK; ++i) rb[i] = bld4s(srdW, voffB[i], sB);
          }
        }
        ld_kc += GBK;
        if (ld_kc >= nkc * GBK) { ld_kc = 0; ++ld_tap; tap_dirty = true; }
        return;
      }
      const int kc = ld_kc;
      const int ck = kc + kq * 4;
      const int nvk = a.Ck - ck;
      if (xform) {
        const int cks = nvk > 0 ? ck : 0;
        pend_sc = *reinterpret_cast<const float4*>(&bnS[cks]);
        pend_sh = *reinterpret_cast<const float4*>(&bnT[cks]);
      }
#pragma unroll
      for (int i = 0; i < A_PER_THR; ++i) {
        const bool ok = okA[i] & (nvk > 0);
        float4 v;
        if constexpr (VEC) {
          // Ck % 4 == 0 here, so a 4-channel group is either fully inside or fully outside
          v = bld4(srdX, ok ? offA[i] + (unsigned)ck * 4u : OOB);
        } else {
          v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) v = ld4(a.X + (long)(offA[i] >> 2) + ck, nvk, false);
        }
        pend_ok[i] = ok;
        ra[i] = v;
      }
      if (w_nk == 0) {
#pragma unroll
        for (int i = 0; i < B_PER_THR_KN; ++i) {
          const int k = tid / N4 + i * (NT / N4);
          const int n = n0 + (tid % N4) * 4;
          const int kk = kc + k;
          const bool ok = (k < GBK) & (kk < a.Ck) & (n < a.Cn);
          if constexpr (VEC) {
            rb[i] = bld4(srdW, ok ? offW + ((unsigned)kk * (unsigned)a.Cout_w + (unsigned)n) * 4u : OOB);
          } else {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = ld4(a.W + (long)(offW >> 2) + (long)kk * a.Cout_w + n, a.Cn - n, false);
            rb[i] = v;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < B_PER_THR_NK; ++i) {
          const int n = n0 + trow + i * RPP;
          const bool ok = (trow + i * RPP < BN) & (n < a.Cn) & (nvk > 0);
          if constexpr (VEC) {
            rb[i] = bld4(srdW, ok ? offW + ((unsigned)n * (unsigned)a.Cout_w + (unsigned)ck) * 4u : OOB);
          } else {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = ld4(a.W + (long)(offW >> 2) + (long)n * a.Cout_w + ck, nvk, false);
            rb[i] = v;
          }
        }
      }
      // advance to the next chunk
      ld_kc += GBK;
      if (ld_kc >= nkc * GBK) { ld_kc = 0; ++ld_tap; tap_dirty = true; }
    };

    auto store_tiles = [&](auto bufc) {
      constexpr int buf = decltype(bufc)::value;
#pragma unroll
      for (int i = 0; i < A_PER_THR; ++i) {
        const int r = trow + i * RPP;
        if (xform) {
          // table entries past Ck are zero, and so are the operand lanes past Ck: relu(0*0+0) = 0;
          // spatial padding must stay zero AFTER the transform
          const bool ok = pend_ok[i];
          ra[i].x = ok ? fmaxf(fmaf(ra[i].x, pend_sc.x, pend_sh.x), 0.f) : 0.f;
          ra[i].y = ok ? fmaxf(fmaf(ra[i].y, pend_sc.y, pend_sh.y), 0.f) : 0.f;
          ra[i].z = ok ? fmaxf(fmaf(ra[i].z, pend_sc.z, pend_sh.z), 0.f) : 0.f;
          ra[i].w = ok ? fmaxf(fmaf(ra[i].w, pend_sc.w, pend_sh.w), 0.f) : 0.f;
        }
        AS_(buf, kq * 4 + 0, r) = ra[i].x;
        AS_(buf, kq * 4 + 1, r) = ra[i].y;
        AS_(buf, kq * 4 + 2, r) = ra[i].z;
        AS_(buf, kq * 4 + 3, r) = ra[i].w;
      }
      if (w_nk == 0) {
#pragma unroll
        for (int i = 0; i < B_PER_THR_KN; ++i) {
          const int k = tid / N4 + i * (NT / N4);
          if (k < GBK) *reinterpret_cast<float4*>(&BS_(buf, k, (tid % N4) * 4)) = rb[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < B_PER_THR_NK; ++i) {
          const int r = trow + i * RPP;
          if (r >= BN) continue;
          BS_(buf, kq * 4 + 0, r) = rb[i].x;
          BS_(buf, kq * 4 + 1, r) = rb[i].y;
          BS_(buf, kq * 4 + 2, r) = rb[i].z;
          BS_(buf, kq * 4 + 3, r) = rb[i].w;
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
      load_tiles(it_beg);
      store_tiles(std::integral_constant<int, 0>{});
    }
    __syncthreads();

    // one K chunk: prefetch the next chunk into registers, MFMA over the current LDS buffer, stage the prefetched
    // chunk into the other buffer.  The buffer index is a compile-time constant (the loop below is unrolled by
    // two) so that every LDS address is one per-thread base register plus an immediate.
    auto chunk = [&](int it, auto curc) {
      constexpr int cur = decltype(curc)::value;
      if (it + 1 < total) load_tiles(it_beg + it + 1);
      // the prefetch must be in flight during the whole MFMA block: without this fence the scheduler sinks the
      // buffer loads down to their first use (the LDS stores below) and the wave sits out their full latency
      __builtin_amdgcn_sched_barrier(0);
#ifdef GEMM_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int kk = 0; kk < GBK / 2; ++kk) {
        const int k = kk * 2 + lhi;
        float av[TI], bv[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) av[i] = AS_(cur, k, wm * WM + i * 32 + l31);
#pragma unroll
        for (int j = 0; j < TJ; ++j) bv[j] = BS_(cur, k, wn * WN + j * 32 + l31);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
#ifdef GEMM_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      if (it + 1 < total) store_tiles(std::integral_constant<int, cur ^ 1>{});
      __syncthreads();
    };
    int it = 0;
    for (; it + 1 < total; it += 2) {
      chunk(it, std::integral_constant<int, 0>{});
      chunk(it + 1, std::integral_constant<int, 1>{});
    }
    if (it < total) chunk(it, std::integral_constant<int, 0>{});

    if constexpr (VEC) {
#include "gemm_epilogue_rows.inc"
    } else {
#include "gemm_epilogue.inc"
    }
  }

  if constexpr (VEC) {
#include "gemm_colstats_rows.inc"
  } else {
#include "gemm_colstats.inc"
  }
#undef AS_
#undef BS_
}

// =====================================================================================================
// direct_gemm: the same implicit GEMM with both operands streamed from global memory straight into the MFMA
// registers -- no LDS tiles, no barriers in the K loop, (almost) no vector instructions beside the MFMAs.
//
// v_mfma_f32_32x32x2_f32 takes, per lane, A[row = lane % 32][k = lane / 32] and B[k = lane / 32][col = lane % 32]:
// which two K indices one instruction multiplies is free as long as A and B agree.  A lane of the upper half
// (lhi = 1) therefore owns K offsets +4..+7 of an 8-deep step and the lower half +0..+3: every lane loads ONE
// 16-byte run of its own row (rows are K-contiguous in the channels-last layout) and the four components feed four
// MFMAs.  Weights: K-contiguous for the input gradient (same 16-byte trick), N-contiguous for the forward (one
// 4-byte load per MFMA, 32 consecutive columns per half-wave).  The K advance is the loads' scalar offset; the
// per-lane byte offsets change only with the tap.  Reuse between the waves of a block (same rows / same columns)
// is served by the CU's vector L1.
// SPEC: 1 forward, plain operand   2 forward, BN+ReLU on the operand   3 input gradient.  Needs Ck % 8 == 0.
// =====================================================================================================
__device__ __forceinline__ float bld1s(__amdgpu_buffer_rsrc_t srd, unsigned byte_off, unsigned s_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srd, byte_off, s_off, 0));
}

template <int WGM, int WGN, int TI, int TJ, int SPEC>
__global__ __launch_bounds__(64 * WGM * WGN, (TI * TJ >= 4 ? 3 : 4)) void direct_gemm_kernel(const GemmArgs a) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = TI * 32, WN = TJ * 32, BM = WGM * WM, BN = WGN * WN;
  constexpr bool xform = SPEC == 2;
  constexpr int w_nk = SPEC == 3 ? 1 : 0;
  constexpr int KS = 8;                          // K floats per step (4 per half-wave)

  __shared__ __attribute__((aligned(16))) float bnS[xform ? MAX_BN_C : 4];
  __shared__ __attribute__((aligned(16))) float bnT[xform ? MAX_BN_C : 4];

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
  const int nkc = a.Ck / KS;                     // Ck % 8 == 0 (host)
  const int total_all = nty * ntx * nkc;
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (int)a.w_bytes, 0x00020000);
  const int per_split = (total_all + a.nsplit - 1) / a.nsplit;
  const int it_beg = split * per_split;
  const int it_end = it_beg + per_split < total_all ? it_beg + per_split : total_all;
  const int total = it_end > it_beg ? it_end - it_beg : 0;

  if (xform) {
    for (int c = tid; c < ((a.Ck + 3) & ~3); c += NT) {
      BnC k = BnC{0.f, 0.f, 0.f, 0.f};
      if (c < a.Ck) k = bn_coef(a.bn_in, c);
      bnS[c] = k.scale;
      bnT[c] = k.shift;
    }
    __syncthreads();
  }

  const int l31 = lane & 31, lhi = lane >> 5;
  const bool do_relu_bn = a.relu_bn.mode != 0 && !a.mix;
  const int hw = a.Hq * a.Wq;

  float cbias[TJ], s1[TJ], s2[TJ];
  BnC rbc[TJ];
  unsigned voffB[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int n = n0 + wn * WN + j * 32 + l31;
    s1[j] = s2[j] = 0.f;
    cbias[j] = (n < a.Cn && a.bias && !a.partial) ? a.bias[n] : 0.f;
    rbc[j] = BnC{0.f, 0.f, 0.f, 0.f};
    if (n < a.Cn && do_relu_bn && !a.partial) rbc[j] = bn_coef(a.relu_bn, n);
    if (n < a.Cn) voffB[j] = w_nk ? ((unsigned)n * (unsigned)a.Cout_w + (unsigned)lhi * 4u) * 4u
                                  : ((unsigned)lhi * 4u * (unsigned)a.Cout_w + (unsigned)n) * 4u;
    else voffB[j] = OOB;
  }

  const long nMt = (a.rows_per_phase + BM - 1) / BM;
  for (long mt = lbx; mt < nMt; mt += gridDim.x) {
    const long m0 = mt * BM;
    int ry0[TI], rx0[TI], rbase[TI];
    bool rvalid[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const long m = m0 + wm * WM + i * 32 + l31;
      rvalid[i] = m < a.rows_per_phase;
      const unsigned mm = rvalid[i] ? (unsigned)m : 0u;
      const int n = (int)(mm / (unsigned)hw);
      const int rem = (int)(mm - (unsigned)n * (unsigned)hw);
      const int qy = rem / a.Wq, qx = rem - qy * a.Wq;
      if (a.form == 0) { ry0[i] = qy * a.sh - a.ph; rx0[i] = qx * a.sw - a.pw; }
      else             { ry0[i] = qy + cy;          rx0[i] = qx + cx; }
      rbase[i] = (n * a.Hx + ry0[i]) * a.Wx + rx0[i];
    }

    int ld_tap = it_beg / nkc;
    int ld_kc = (it_beg - ld_tap * nkc) * KS;
    bool tap_dirty = true;
    unsigned offA[TI], offW = 0;
    bool okA[TI];
    // two register sets: step s+1 is in flight while step s is multiplied
    float4 ra[2][TI], rsc[2], rsh[2];
    float rbd[2][TJ][4];
    float4 rb4[2][TJ];
    bool rok[2][TI];

    auto load_step = [&](auto bufc) {
      constexpr int buf = decltype(bufc)::value;
      if (tap_dirty) {
        const int jy = ld_tap / ntx, jx = ld_tap - jy * ntx;
        const int wtap = (ky0 + kstep_y * jy) * a.kw + (kx0 + kstep_x * jx);
        const int tapoff = dsgn * (jy * a.Wx + jx);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int iy = ry0[i] + dsgn * jy, ix = rx0[i] + dsgn * jx;
          okA[i] = rvalid[i] & ((unsigned)iy < (unsigned)a.Hx) & ((unsigned)ix < (unsigned)a.Wx);
          offA[i] = okA[i] ? (unsigned)(rbase[i] + tapoff) * (unsigned)a.Ck * 4u + (unsigned)lhi * 16u : OOB;
        }
        offW = (unsigned)wtap * (unsigned)a.Cin_w * (unsigned)a.Cout_w * 4u;
        tap_dirty = false;
      }
      const unsigned sA = (unsigned)ld_kc * 4u;
#pragma unroll
      for (int i = 0; i < TI; ++i) { ra[buf][i] = bld4s(srdX, offA[i], sA); rok[buf][i] = okA[i]; }
      if (w_nk) {
        const unsigned sB = offW + (unsigned)ld_kc * 4u;
#pragma unroll
        for (int j = 0; j < TJ; ++j) rb4[buf][j] = bld4s(srdW, voffB[j], sB);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned sB = offW + (unsigned)(ld_kc + e) * (unsigned)a.Cout_w * 4u;
#pragma unroll
          for (int j = 0; j < TJ; ++j) rbd[buf][j][e] = bld1s(srdW, voffB[j], sB);
        }
      }
      if (xform) {
        rsc[buf] = *reinterpret_cast<const float4*>(&bnS[ld_kc + lhi * 4]);
        rsh[buf] = *reinterpret_cast<const float4*>(&bnT[ld_kc + lhi * 4]);
      }
      ld_kc += KS;
      if (ld_kc >= a.Ck) { ld_kc = 0; ++ld_tap; tap_dirty = true; }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mma_step = [&](auto bufc) {
      constexpr int buf = decltype(bufc)::value;
      float av[TI][4];
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        float4 v = ra[buf][i];
        if (xform) {   // spatial padding must stay zero AFTER the transform
          const bool ok = rok[buf][i];
          v.x = ok ? fmaxf(fmaf(v.x, rsc[buf].x, rsh[buf].x), 0.f) : 0.f;
          v.y = ok ? fmaxf(fmaf(v.y, rsc[buf].y, rsh[buf].y), 0.f) : 0.f;
          v.z = ok ? fmaxf(fmaf(v.z, rsc[buf].z, rsh[buf].z), 0.f) : 0.f;
          v.w = ok ? fmaxf(fmaf(v.w, rsc[buf].w, rsh[buf].w), 0.f) : 0.f;
        }
        av[i][0] = v.x; av[i][1] = v.y; av[i][2] = v.z; av[i][3] = v.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          float bv;
          if (w_nk) bv = e == 0 ? rb4[buf][j].x : (e == 1 ? rb4[buf][j].y : (e == 2 ? rb4[buf][j].z : rb4[buf][j].w));
          else bv = rbd[buf][j][e];
#pragma unroll
          for (int i = 0; i < TI; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv, acc[i][j], 0, 0, 0);
        }
      }
    };

    if (total > 0) load_step(std::integral_constant<int, 0>{});
    for (int it = 0; it < total; it += 2) {
      if (it + 1 < total) load_step(std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      mma_step(std::integral_constant<int, 0>{});
      if (it + 1 < total) {
        if (it + 2 < total) load_step(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        mma_step(std::integral_constant<int, 1>{});
      }
    }

#include "gemm_epilogue.inc"
  }

#include "gemm_colstats.inc"
}

// =====================================================================================================
// weight gradient
// =====================================================================================================
struct WgradArgs {
  const float* Xs;   // layer input  x  [rows][Cin]
  const float* Dy;   // output grad  dy [rows][Cout]
  float* dW;         // [taps][Cin][Cout]
  int N, Hs, Ws, Hb, Wb, Cin, Cout, kh, kw, sh, sw, ph, pw;
  int x_is_big;      // 1: conv (x on the big grid, gathered); 0: convT (dy on the big grid, gathered)
  long Ms;           // N*Hs*Ws pixels of the small grid (the reduction length)
  long chunk;        // pixels per split (multiple of BK)
  int nJ;            // number of J (Cout) tiles
  int atomic;        // 1: accumulate with atomics (split reduction)
  int vecI, vecJ;
  unsigned x_bytes, dy_bytes;   // operand sizes for the buffer descriptors (vector path)
  int fast;          // 1: every 16-pixel K chunk is a run inside one image row, or whole rows of one image
  mopoe_bn_ref bn_in;  int xcd_remap;         // 1: XCD-aware block numbering (gemm_common.hpp: xcd_swizzle)
};

// SPEC: 0 = mode flags at run time; 1 / 2 = fast pixel addressing without / with BN+ReLU on x (see gather_gemm_kernel)
template <int BI, int BJ, bool VEC, int SPEC = 0>
__global__ __launch_bounds__(256) void wgrad_gemm_kernel(const WgradArgs a) {
  static_assert(SPEC == 0 || VEC, "specialised loops exist for the vector path only");
  constexpr int WI = BI / 2, WJ = BJ / 2;
  constexpr int TI = WI / 32, TJ = WJ / 32;
  constexpr int I_LD = BI + LDS_PAD, J_LD = BJ + LDS_PAD;
  constexpr int I4 = BI / 4, J4 = BJ / 4;          // float4 per pixel row
  constexpr int I_PER_THR = (BK * I4) / 256, J_PER_THR = (BK * J4) / 256;

  __shared__ __attribute__((aligned(16))) float Is[2][BK][I_LD];
  __shared__ __attribute__((aligned(16))) float Js[2][BK][J_LD];
  __shared__ __attribute__((aligned(16))) float bnS[MAX_BN_C];
  __shared__ __attribute__((aligned(16))) float bnT[MAX_BN_C];

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
  const int total = (int)((mend - mbeg + BK - 1) / BK);

  const bool xform = SPEC ? (SPEC == 2) : (a.bn_in.mode != 0);
  const bool fast = SPEC ? true : (VEC && a.fast != 0);
  if (xform) {
    for (int c = tid; c < ((a.Cin + 3) & ~3); c += 256) {
      BnC k = BnC{0.f, 0.f, 0.f, 0.f};
      if (c < a.Cin) k = bn_coef(a.bn_in, c);
      bnS[c] = k.scale;
      bnT[c] = k.shift;
    }
    __syncthreads();
  }

  float4 ri[I_PER_THR], rj[J_PER_THR];
  bool pend_ok[I_PER_THR];
  const int hw = a.Hs * a.Ws;
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void*)a.Xs, 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdD = __builtin_amdgcn_make_buffer_rsrc((void*)a.Dy, 0, (int)a.dy_bytes, 0x00020000);

  static_assert(BI == BJ, "both operand tiles cover the same pixels per thread");
  // Pixel coordinates of this thread's rows, advanced incrementally (16 pixels per K-step) instead of being
  // re-derived with integer divisions every iteration.
  constexpr int PPP = 256 / I4;                 // pixels covered per pass
  int pn[I_PER_THR], pqy[I_PER_THR], pqx[I_PER_THR];
  long pm[I_PER_THR];
#pragma unroll
  for (int t = 0; t < I_PER_THR; ++t) {
    pm[t] = mbeg + tid / I4 + t * PPP;
    const unsigned m = (unsigned)(pm[t] < a.Ms ? pm[t] : 0);
    const unsigned n = m / (unsigned)hw;
    const unsigned rem = m - n * (unsigned)hw;
    pn[t] = (int)n; pqy[t] = rem / a.Ws; pqx[t] = rem - pqy[t] * a.Ws;
  }

  // BN+ReLU of x is applied at LDS-store time (see gather_gemm_kernel): loads stay back-to-back
  float4 sc4 = make_float4(0.f, 0.f, 0.f, 0.f), sh4 = sc4;
  const int ci = i0 + (tid % I4) * 4, cj = j0 + (tid % J4) * 4;
  if (xform) {
    const int cs = ci < a.Cin ? ci : 0;
    sc4 = *reinterpret_cast<const float4*>(&bnS[cs]);
    sh4 = *reinterpret_cast<const float4*>(&bnT[cs]);
  }

  // ---- fast pixel addressing ---------------------------------------------------------------------------------
  // The order in which pixels are reduced is free, and when Ws is a multiple of the 16-pixel chunk (or divides it,
  // with Hs*Ws a multiple of 16) every chunk is a run inside one image row (or whole rows of one image).  The
  // chunk's origin (image, row, column) is then wave-uniform and advances with scalar adds; it enters the buffer
  // loads through their scalar offset.  Each thread keeps constant byte offsets for its pixel slots; per chunk it
  // only re-evaluates the bounds of the gathered operand.  The gathered operand's descriptor starts (ph*Wb + pw)
  // rows BEFORE the tensor so that both offset parts stay non-negative; those rows are never touched (the bounds
  // test sends them to the OOB offset).
  const int rpc = a.Ws >= BK ? 1 : BK / a.Ws;              // image rows per chunk (fast path)
  const int big_c = a.x_is_big ? a.Cin : a.Cout;           // channels of the gathered (big-grid) operand
  const unsigned shift_rows = (unsigned)(a.ph * a.Wb + a.pw);
  const unsigned shift_bytes = shift_rows * (unsigned)big_c * 4u;
  const __amdgpu_buffer_rsrc_t srdBig = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)(a.x_is_big ? a.Xs : a.Dy) - shift_bytes), 0,
      (int)((a.x_is_big ? a.x_bytes : a.dy_bytes) + shift_bytes), 0x00020000);
  int f_ly[I_PER_THR], f_lx[I_PER_THR];          // lane part of the big-grid coordinates (may be negative)
  unsigned f_vbig[I_PER_THR], f_vsmall_i[I_PER_THR], f_vsmall_j[I_PER_THR];
  int f_slot[I_PER_THR];
  int f_n = 0, f_qy = 0, f_qx = 0;               // chunk origin (wave-uniform)
  unsigned f_m0 = 0;                             // first pixel of the chunk
  if (fast) {
#pragma unroll
    for (int t = 0; t < I_PER_THR; ++t) {
      const int p = tid / I4 + t * PPP;
      const int pdy = a.Ws >= BK ? 0 : p / a.Ws, pqx = a.Ws >= BK ? p : p - (p / a.Ws) * a.Ws;
      f_slot[t] = p;
      f_ly[t] = pdy * a.sh + ky - a.ph;
      f_lx[t] = pqx * a.sw + kx - a.pw;
      const unsigned lane_rows = (unsigned)((f_ly[t] + a.ph) * a.Wb + (f_lx[t] + a.pw));
      const int cbig = a.x_is_big ? ci : cj;
      f_vbig[t] = (cbig < big_c) ? (lane_rows * (unsigned)big_c + (unsigned)cbig) * 4u : OOB;
      f_vsmall_i[t] = (ci < a.Cin) ? ((unsigned)p * (unsigned)a.Cin + (unsigned)ci) * 4u : OOB;
      f_vsmall_j[t] = (cj < a.Cout) ? ((unsigned)p * (unsigned)a.Cout + (unsigned)cj) * 4u : OOB;
    }
    const unsigned m0 = (unsigned)mbeg;
    f_m0 = m0;
    f_n = (int)(m0 / (unsigned)hw);
    const unsigned rem = m0 - (unsigned)f_n * (unsigned)hw;
    f_qy = (int)(rem / (unsigned)a.Ws);
    f_qx = (int)(rem - (unsigned)f_qy * (unsigned)a.Ws);
  }

  auto load_tiles = [&](int /*it*/) {
    if (fast) {
      if constexpr (VEC) {
        const int nleft = (int)((unsigned)mend - f_m0);     // pixels of this split from the chunk origin on
        const int s_by = f_qy * a.sh, s_bx = f_qx * a.sw;
        const unsigned s_big = (unsigned)((f_n * a.Hb + s_by) * a.Wb + s_bx) * (unsigned)big_c * 4u;
        const unsigned s_small_i = f_m0 * (unsigned)a.Cin * 4u, s_small_j = f_m0 * (unsigned)a.Cout * 4u;
#pragma unroll
        for (int t = 0; t < I_PER_THR; ++t) {
          const bool inm = f_slot[t] < nleft;
          const bool inb = ((unsigned)(s_by + f_ly[t]) < (unsigned)a.Hb) & ((unsigned)(s_bx + f_lx[t]) < (unsigned)a.Wb);
          const unsigned vb = (inm & inb) ? f_vbig[t] : OOB;
          if (a.x_is_big) {
            ri[t] = bld4s(srdBig, vb, s_big);
            rj[t] = bld4s(srdD, inm ? f_vsmall_j[t] : OOB, s_small_j);
            pend_ok[t] = (inm & inb) & (ci < a.Cin);
          } else {
            ri[t] = bld4s(srdX, inm ? f_vsmall_i[t] : OOB, s_small_i);
            rj[t] = bld4s(srdBig, vb, s_big);
            pend_ok[t] = inm & (ci < a.Cin);
          }
        }
        // next chunk (wave-uniform)
        f_m0 += BK;
        if (a.Ws >= BK) { f_qx += BK; if (f_qx >= a.Ws) { f_qx = 0; ++f_qy; } }
        else f_qy += rpc;
        if (f_qy >= a.Hs) { f_qy = 0; ++f_n; }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < I_PER_THR; ++t) {
      const bool inm = pm[t] < mend;
      const int by = pqy[t] * a.sh - a.ph + ky, bx = pqx[t] * a.sw - a.pw + kx;
      const bool inb = ((unsigned)by < (unsigned)a.Hb) & ((unsigned)bx < (unsigned)a.Wb);
      const int srow = (int)pm[t];
      const int brow = (pn[t] * a.Hb + by) * a.Wb + bx;
      const int rowx = a.x_is_big ? brow : srow, rowd = a.x_is_big ? srow : brow;
      const bool okx = inm & (ci < a.Cin) & (a.x_is_big ? inb : true);
      const bool okd = inm & (cj < a.Cout) & (a.x_is_big ? true : inb);
      pend_ok[t] = okx;
      if constexpr (VEC) {
        ri[t] = bld4(srdX, okx ? ((unsigned)rowx * (unsigned)a.Cin + (unsigned)ci) * 4u : OOB);
        rj[t] = bld4(srdD, okd ? ((unsigned)rowd * (unsigned)a.Cout + (unsigned)cj) * 4u : OOB);
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), w = v;
        if (okx) v = ld4(a.Xs + (long)rowx * a.Cin + ci, a.Cin - ci, false);
        if (okd) w = ld4(a.Dy + (long)rowd * a.Cout + cj, a.Cout - cj, false);
        ri[t] = v; rj[t] = w;
      }
      // advance this row by BK pixels
      pm[t] += BK;
      pqx[t] += BK;
      if (pqx[t] >= a.Ws) {
        if (a.Ws >= BK) {
          pqx[t] -= a.Ws; ++pqy[t];
        } else {
          const int carry = pqx[t] / a.Ws;
          pqx[t] -= carry * a.Ws; pqy[t] += carry;
        }
        if (pqy[t] >= a.Hs) {
          const int carry = pqy[t] / a.Hs;
          pqy[t] -= carry * a.Hs; pn[t] += carry;
        }
      }
    }
  };

  auto store_tiles = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int t = 0; t < I_PER_THR; ++t) {
      const int p = tid / I4 + t * (256 / I4);
      if (xform) {
        const bool ok = pend_ok[t];
        ri[t].x = ok ? fmaxf(fmaf(ri[t].x, sc4.x, sh4.x), 0.f) : 0.f;
        ri[t].y = ok ? fmaxf(fmaf(ri[t].y, sc4.y, sh4.y), 0.f) : 0.f;
        ri[t].z = ok ? fmaxf(fmaf(ri[t].z, sc4.z, sh4.z), 0.f) : 0.f;
        ri[t].w = ok ? fmaxf(fmaf(ri[t].w, sc4.w, sh4.w), 0.f) : 0.f;
      }
      *reinterpret_cast<float4*>(&Is[buf][p][(tid % I4) * 4]) = ri[t];
    }
#pragma unroll
    for (int t = 0; t < J_PER_THR; ++t) {
      const int p = tid / J4 + t * (256 / J4);
      *reinterpret_cast<float4*>(&Js[buf][p][(tid % J4) * 4]) = rj[t];
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
    load_tiles(0);
    store_tiles(std::integral_constant<int, 0>{});
  }
  __syncthreads();

  const int l31 = lane & 31, lhi = lane >> 5;
  auto chunk = [&](int it, auto curc) {
    constexpr int cur = decltype(curc)::value;
    if (it + 1 < total) load_tiles(it + 1);
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the MFMA block (see gather_gemm_kernel)
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = kk * 2 + lhi;
      float av[TI], bv[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) av[i] = Is[cur][k][wi * WI + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < TJ; ++j) bv[j] = Js[cur][k][wj * WJ + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
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

#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int co = j0 + wj * WJ + j * 32 + l31;
    if (co >= a.Cout) continue;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = i0 + wi * WI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (ci >= a.Cin) continue;
        float* dst = a.dW + ((long)tap * a.Cin + ci) * a.Cout + co;
        if (a.atomic) unsafeAtomicAdd(dst, acc[i][j][r]);
        else *dst = acc[i][j][r];
      }
    }
  }
}

// =====================================================================================================
// host-side launchers
// =====================================================================================================
// ---- split-K epilogue: Y = mask * (sum_s partial[s] + bias), optional ReLU/BN-backward masking + sums ------
// block = 64 columns x 4 row-lanes, grid-stride over rows (at most EPI_MAX_BLOCKS_Y blocks per column group so
// that the column statistics leave as few same-address atomics as possible); the nsplit partials of one
// element are independent loads.
#ifndef EPI_MAX_BLOCKS_Y
#define EPI_MAX_BLOCKS_Y 256
#endif
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmArgs a) {
  __shared__ float cs[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const bool nok = n < a.Cn;
  const bool do_relu_bn = a.relu_bn.mode != 0;
  const float bias = (nok && a.bias) ? a.bias[n] : 0.f;
  BnC rb = {0.f, 0.f, 0.f, 0.f};
  if (nok && do_relu_bn) rb = bn_coef(a.relu_bn, n);
  const long stride = a.rows_total * (long)a.Cn;
  float s1 = 0.f, s2 = 0.f;
  if (nok) {
    for (long row = (long)blockIdx.y * 4 + ty; row < a.rows_total; row += (long)gridDim.y * 4) {
      const float* p = a.partial + row * a.Cn + n;
      float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int s = 0;
      for (; s + 8 <= a.nsplit; s += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc8[u] += p[(long)(s + u) * stride];
      }
      for (; s < a.nsplit; ++s) acc8[s & 7] += p[(long)s * stride];
      float x = bias + (((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7])));
      if (a.mask.kind != 0) x *= mask_at(a.mask, row, n, a.Cn);
      if (do_relu_bn) {
        const float xi = a.xin[row * a.Cn + n];
        x = (fmaf(xi, rb.scale, rb.shift) > 0.f) ? x : 0.f;
        s1 += x;
        s2 += x * ((xi - rb.mean) * rb.rstd);
      } else {
        s1 += x;
        s2 += x * x;
      }
      a.Y[row * a.Cn + n] = x;
    }
  }
  double* sums = do_relu_bn ? a.bwd_sums : a.out_stats;
  if (sums) {
    cs[0][ty][tx] = s1;
    cs[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && nok) {
      atomic_add_f64(sums + n, (double)((cs[0][0][tx] + cs[0][1][tx]) + (cs[0][2][tx] + cs[0][3][tx])));
      atomic_add_f64(sums + a.Cn + n, (double)((cs[1][0][tx] + cs[1][1][tx]) + (cs[1][2][tx] + cs[1][3][tx])));
    }
  }
}


// ---- LDS-DMA family (tiles 12..15) --------------------------------------------------------------------------------------
#include "gemm_glds_common.inc"
#include "conv_gemm_glds.inc"
#include "conv_gemm_glds_parity.inc"


// dest_on_small: 1 -> form 0 (Y on the small grid), 0 -> form 1 (Y on the big grid)
static int launch_gather(const float* X, const float* W, const float* bias, float* Y, const mopoe_conv_geom* g,
                         int dest_on_small, int Ck, int Cn, int w_nk, const mopoe_bn_ref* bn_in,
                         const mopoe_mask_ref* mask, double* out_stats, const mopoe_bn_ref* relu_bn,
                         const float* xin, double* bwd_sums, const mopoe_conv_plan* plan, void* ws, size_t ws_bytes,
                         hipStream_t stream, const mopoe_mix_ref* mix = nullptr) {
  // the head of the workspace holds the arrival counters of in-kernel split reductions (kept zero by the kernels)
  int* counters = nullptr;
  if (ws && ws_bytes > WS_COUNTER_BYTES) { counters = (int*)ws; ws = (char*)ws + WS_COUNTER_BYTES; ws_bytes -= WS_COUNTER_BYTES; }
  else { ws = nullptr; ws_bytes = 0; }
  GemmArgs a;
  a.X = X; a.W = W; a.Y = Y; a.bias = bias;
  a.N = g->N; a.Ck = Ck; a.Cn = Cn; a.Cin_w = g->Cin; a.Cout_w = g->Cout;
  a.kh = g->kh; a.kw = g->kw; a.sh = g->sh; a.sw = g->sw; a.ph = g->ph; a.pw = g->pw;
  a.w_nk = w_nk;
  int nphase;
  if (dest_on_small) {
    a.form = 0; a.Hx = g->Hb; a.Wx = g->Wb; a.Hy = g->Hs; a.Wy = g->Ws; a.Hq = g->Hs; a.Wq = g->Ws; nphase = 1;
  } else {
    a.form = 1; a.Hx = g->Hs; a.Wx = g->Ws; a.Hy = g->Hb; a.Wy = g->Wb; a.Hq = g->Hb / g->sh; a.Wq = g->Wb / g->sw;
    nphase = g->sh * g->sw;
  }
  a.rows_per_phase = (long)g->N * a.Hq * a.Wq;
  a.rows_total = (long)g->N * a.Hy * a.Wy;
  a.vecA = (Ck % 4 == 0) && aligned16(X);
  a.vecB = (g->Cout % 4 == 0) && aligned16(W);
  const size_t xb = (size_t)g->N * a.Hx * a.Wx * Ck * sizeof(float);
  const size_t wb = (size_t)g->kh * g->kw * g->Cin * g->Cout * sizeof(float);
  // branch-free buffer loads need 16-byte alignment, channel counts that are multiples of 4 and < 2 GiB operands
  // (the vector path's row-major epilogue moves the result, xin and the mask in float4 pieces: Cn % 4 == 0, aligned rows)
  const bool vec = a.vecA && a.vecB && xb < (1ull << 31) && wb < (1ull << 31) && Cn % 4 == 0 && aligned16(Y) &&
                   (!xin || aligned16(xin)) && (!mask || mask->kind == 0 || aligned16(mask->mask));
  a.x_bytes = (unsigned)std::min<size_t>(xb, 0x7fffffffu);
  a.w_bytes = (unsigned)std::min<size_t>(wb, 0x7fffffffu);
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
  if (mix) {   // residual mix in the epilogue: the shortcut's BN rides in relu_bn, its tensor in xin (vector path only)
    if (!mix->s || mix->bn.mode == 0 || relu_bn || xin) { set_error("conv_fwd_mix: needs s and its BatchNorm"); return MOPOE_ERR_ARG; }
    if (!vec || !aligned16(mix->s)) { set_error("conv_fwd_mix: channel counts must be multiples of 4 and tensors 16-byte aligned"); return MOPOE_ERR_ARG; }
    a.mix = 1; a.mix_a = mix->a; a.mix_b = mix->b;
    a.relu_bn = mix->bn; a.xin = (const float*)mix->s;
  }
  a.nsplit = 1; a.partial = nullptr; a.counters = nullptr;
  static const bool dbg_nostats = getenv("MOPOE_DEBUG_NOSTATS") != nullptr;  // timing experiments only
  if (dbg_nostats) { a.out_stats = nullptr; a.bwd_sums = nullptr; }
  if (a.bn_in.mode != 0 && (a.bn_in.C != Ck || Ck > MAX_BN_C)) { set_error("bn_in channel mismatch (%d vs %d)", a.bn_in.C, Ck); return MOPOE_ERR_ARG; }
  if (a.relu_bn.mode != 0 && (a.relu_bn.C != Cn || !a.xin)) { set_error("relu_bn needs xin and C == %d", Cn); return MOPOE_ERR_ARG; }
  if (a.mask.kind != 0 && !a.mask.mask) { set_error("mask pointer missing"); return MOPOE_ERR_ARG; }
  if (a.mask.kind == 1 && a.mask.rows_per_sample != a.Hy * a.Wy) { set_error("channel mask: rows_per_sample must be Hout*Wout"); return MOPOE_ERR_ARG; }
  if (a.rows_total >= (1L << 31) || (long)g->N * a.Hx * a.Wx >= (1L << 31)) { set_error("conv: more than 2^31 rows"); return MOPOE_ERR_ARG; }

  // ---- tile choice -------------------------------------------------------------------------------------------
  // 0: 128x128 (2x2 waves)   1: 256x64 (4x1 waves; narrow outputs, many rows)   2: 64x64 (2x2 waves)
  // Wide outputs always take the 128x128 tile (best operand reuse); when that leaves the chip under-filled the
  // tap x channel reduction is split across blocks instead of shrinking the tile.
  int cfg;
  bool emu = false;
  if (Cn > 64) cfg = (a.rows_per_phase > 64 || Cn >= 256) ? 0 : 2;
  else cfg = a.rows_per_phase >= 256L * 64 ? 1 : 2;
#ifdef TILE_N64_REMAINDER
  // a 128-wide tile would be half empty for the last 64 columns (Cout = 192, 320): use 64-wide tiles instead
  if (cfg == 0 && (Cn % 128) == 64 && a.rows_per_phase >= 256L * TILE_N64_REMAINDER) cfg = 1;
#endif
  if (plan && plan->tile >= 0) {
    if (plan->tile > 19) { set_error("conv plan: tile %d (see mopoe_conv_plan in mopoe_hip.h: 0..19)", plan->tile); return MOPOE_ERR_ARG; }
    cfg = plan->tile;
    // 16..19 = tiles 12..15 with the fp32 products on the bf16 matrix pipe (conv_gemm_glds.inc, EMU); not with BN on load
    if (cfg >= 16) {
      if (a.bn_in.mode != 0) { set_error("conv plan: tile %d (split-bf16 products) has no BN-on-load form", cfg); return MOPOE_ERR_ARG; }
      emu = true;
      cfg -= 4;
    }
    // 12..15 = LDS-DMA family (conv_gemm_glds.inc): vector path, K channels a multiple of 32; with BN on load the tiles with
    // two or four buffers (12, 14, 15).  A plan that asks for one where it does not apply is refused (the tuner never offers it).
    if (cfg >= 12 && (!vec || Ck % 32 != 0 || (a.bn_in.mode != 0 && cfg == 13))) {
      set_error("conv plan: tile %d (LDS-DMA family) needs the vector path, K channels %% 32 == 0 (Ck = %d) and, with BN on load, tile 12, 14 or 15", cfg, Ck);
      return MOPOE_ERR_ARG;
    }
    if (cfg >= 3 && !vec) cfg = cfg == 3 ? 0 : 2;   // the extra tiles exist for the vector path only
    if ((cfg == 5 || cfg == 6) && Ck % 32 != 0) cfg -= 3;   // 5, 6 = tiles 2, 4 with a 32-deep K chunk
    if (cfg >= 8 && cfg < 12 && (!vec || Ck % 8 != 0 || a.mix)) cfg = (cfg == 8) ? 0 : (cfg == 9 ? 1 : (cfg == 10 ? 2 : 4));   // 8..11 = LDS-free kernels (no residual mix)
  }
  static const int TILE_BM[16] = {128, 256, 64, 256, 128, 64, 128, 128, 128, 256, 64, 128, 128, 128, 64, 256};
  static const int TILE_BN[16] = {128, 64, 64, 128, 64, 64, 64, 128, 128, 64, 64, 64, 128, 64, 64, 128};
  const int bm = TILE_BM[cfg], bn = TILE_BN[cfg];
  const long nMt = ceil_div(a.rows_per_phase, bm);
  const int nNt = ceil_div(Cn, bn);
  // ---- split-K for grids that cannot fill the chip --------------------------------------------------------------
  const int gbk = (cfg == 5 || cfg == 6 || cfg >= 12) ? 32 : (cfg >= 8 ? 8 : 16);
  const int nkc = ceil_div(Ck, gbk);
  const int iters = (dest_on_small ? g->kh * g->kw : std::max(1, (g->kh / g->sh) * (g->kw / g->sw))) * nkc;
  const long blocks = nMt * nNt * nphase;
  const size_t per = (size_t)a.rows_total * Cn * sizeof(float);
  if (plan && plan->split > 0) {
    long ns = std::min<long>(plan->split, iters);
    if (ns >= 2 && (!ws || (size_t)ns * per > ws_bytes)) {
      set_error("conv plan: split %ld needs %zu workspace bytes (have %zu)", ns, (size_t)ns * per, ws ? ws_bytes : (size_t)0);
      return MOPOE_ERR_ARG;
    }
    if (ns >= 2) { a.nsplit = (int)ns; a.partial = (float*)ws; }
  } else if (ws && blocks < SPLIT_BLOCKS && iters * gbk >= 256) {
    long ns = std::min<long>((SPLIT_TARGET + blocks - 1) / blocks, (long)iters * gbk / 128);
    if ((size_t)ns * per > ws_bytes) ns = (long)(ws_bytes / per);
    if (ns >= 2) { a.nsplit = (int)ns; a.partial = (float*)ws; }
  }
  // vector path, LDS kernels: the reduction finishes in the tile's last-arriving block (gemm_epilogue_rows.inc); the
  // scalar path and the LDS-free kernels park raw slabs for splitk_epilogue_kernel
  const bool in_kernel_reduce = vec && (cfg < 8 || cfg >= 12);
  if (a.partial && in_kernel_reduce) {
    if (blocks > (long)(WS_COUNTER_BYTES / sizeof(int))) { set_error("conv: split reduction over %ld output tiles (at most %zu)", blocks, WS_COUNTER_BYTES / sizeof(int)); return MOPOE_ERR_ARG; }
    a.counters = counters;
  }
  // ---- persistent M loop: at most ~1024 blocks in flight, column statistics leave a block once -------------------
  const long persist = cfg >= 12 ? (cfg == 15 ? 256 : 512)       // LDS-DMA tiles: resident blocks by LDS footprint (1 or 2 per CU)
                                 : (cfg >= 8 ? 4096 : ((cfg == 2 || cfg >= 4) ? PERSIST_BLOCKS * 3 / 2 : PERSIST_BLOCKS));   // 4-wave blocks: 3 per CU
  long gx = std::min<long>(nMt, std::max<long>(1, persist / ((long)nNt * nphase * a.nsplit)));
  // algorithmic flops: every (output pixel, tap that exists) pair
  double taps_eff = dest_on_small ? (double)g->kh * g->kw : (double)g->kh * g->kw / ((double)g->sh * g->sw);
  const double flops = 2.0 * (double)g->N * a.Hy * a.Wy * (double)Cn * (double)Ck * taps_eff;
  {
    const int spec = (vec && Ck % gbk == 0) ? (w_nk ? 3 : (a.bn_in.mode != 0 ? 2 : 1)) : 0;
    // algorithmic bytes: one read of the gathered activation + one write of the result
    const double abytes = ((double)g->N * a.Hx * a.Wx * Ck + (double)a.rows_total * Cn) * sizeof(float);
    ProfScope prof(stream, flops, emu ? PROF_F32_GLDS_EMU + (cfg - 12) * 2 + (spec == 3 ? 1 : 0)
                                      : cfg >= 12 ? PROF_F32_GLDS + (cfg - 12) * 3 + (spec - 1)
                                            : (cfg >= 8 ? PROF_DIRECT + (cfg - 8) * 4 + spec
                                                        : (vec ? PROF_GATHER_VEC + cfg * 4 + spec : PROF_GATHER_SCALAR + cfg)), abytes);
    dim3 grid((unsigned)gx, nNt, nphase * a.nsplit);
    // specialised main loops (spec != 0): vector path with Ck a multiple of the K chunk (every layer of the four
    // networks except the image-side edge layers, which do not come here, and the vocabulary projection's input gradient)
#define MOPOE_LAUNCH_TILE(BM_, BN_, WM_, WN_, BK_, THREADS_)                                                                    \
  do {                                                                                                                          \
    if (spec == 1) hipLaunchKernelGGL((gather_gemm_kernel<BM_, BN_, WM_, WN_, BK_, true, 1>), grid, dim3(THREADS_), 0, stream, a);      \
    else if (spec == 2) hipLaunchKernelGGL((gather_gemm_kernel<BM_, BN_, WM_, WN_, BK_, true, 2>), grid, dim3(THREADS_), 0, stream, a); \
    else if (spec == 3) hipLaunchKernelGGL((gather_gemm_kernel<BM_, BN_, WM_, WN_, BK_, true, 3>), grid, dim3(THREADS_), 0, stream, a); \
    else hipLaunchKernelGGL((gather_gemm_kernel<BM_, BN_, WM_, WN_, 16, true, 0>), grid, dim3(THREADS_), 0, stream, a);                \
  } while (0)
#define MOPOE_LAUNCH_DIRECT(WM_, WN_, TI_, TJ_)                                                                                   \
  do {                                                                                                                          \
    if (spec == 1) hipLaunchKernelGGL((direct_gemm_kernel<WM_, WN_, TI_, TJ_, 1>), grid, dim3(64 * WM_ * WN_), 0, stream, a);      \
    else if (spec == 2) hipLaunchKernelGGL((direct_gemm_kernel<WM_, WN_, TI_, TJ_, 2>), grid, dim3(64 * WM_ * WN_), 0, stream, a); \
    else hipLaunchKernelGGL((direct_gemm_kernel<WM_, WN_, TI_, TJ_, 3>), grid, dim3(64 * WM_ * WN_), 0, stream, a);                \
  } while (0)
#define MOPOE_LAUNCH_GLDS(BM_, BN_, WM_, WN_, ST_)                                                                                       \
  do {                                                                                                                                \
    if (spec == 1) hipLaunchKernelGGL((gather_gemm_f32_glds_kernel<BM_, BN_, WM_, WN_, 1, ST_>), grid, dim3(64 * WM_ * WN_), 0, stream, a); \
    else hipLaunchKernelGGL((gather_gemm_f32_glds_kernel<BM_, BN_, WM_, WN_, 3, ST_>), grid, dim3(64 * WM_ * WN_), 0, stream, a);          \
  } while (0)
#define MOPOE_LAUNCH_GLDS_EMU(BM_, BN_, WM_, WN_, ST_)                                                                                   \
  do {                                                                                                                                \
    if (spec == 1) hipLaunchKernelGGL((gather_gemm_f32_glds_kernel<BM_, BN_, WM_, WN_, 1, ST_, 1>), grid, dim3(64 * WM_ * WN_), 0, stream, a); \
    else hipLaunchKernelGGL((gather_gemm_f32_glds_kernel<BM_, BN_, WM_, WN_, 3, ST_, 1>), grid, dim3(64 * WM_ * WN_), 0, stream, a);          \
  } while (0)
    if (emu) {
      // (the two wide tiles: one 4-wave block per CU -- 512 registers per wave for the software pipeline -- and the LDS that frees
      // spent on stages: 4 x 32 KB, 3 x 48 KB)
      if (cfg == 12) MOPOE_LAUNCH_GLDS_EMU(128, 128, 2, 2, 2);
      else if (cfg == 13) MOPOE_LAUNCH_GLDS_EMU(128, 64, 2, 2, 3);
      else if (cfg == 14) MOPOE_LAUNCH_GLDS_EMU(64, 64, 2, 2, 4);
      else MOPOE_LAUNCH_GLDS_EMU(256, 128, 4, 2, 3);
    }
#undef MOPOE_LAUNCH_GLDS_EMU
    else if (cfg >= 12) {
      if (spec == 2) {
        if (cfg == 12) hipLaunchKernelGGL((gather_gemm_f32_glds_kernel<128, 128, 2, 2, 2, 2>), grid, dim3(256), 0, stream, a);
        else if (cfg == 14) hipLaunchKernelGGL((gather_gemm_f32_glds_kernel<64, 64, 2, 2, 2, 4>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((gather_gemm_f32_glds_kernel<256, 128, 4, 2, 2, 2>), grid, dim3(512), 0, stream, a);
      }
      else if (cfg == 12) MOPOE_LAUNCH_GLDS(128, 128, 2, 2, 2);
      else if (cfg == 13) MOPOE_LAUNCH_GLDS(128, 64, 2, 2, 3);
      else if (cfg == 14) MOPOE_LAUNCH_GLDS(64, 64, 2, 2, 4);
      else MOPOE_LAUNCH_GLDS(256, 128, 4, 2, 2);
    }
#undef MOPOE_LAUNCH_GLDS
    else if (cfg == 8) MOPOE_LAUNCH_DIRECT(2, 2, 2, 2);
    else if (cfg == 9) MOPOE_LAUNCH_DIRECT(4, 1, 2, 2);
    else if (cfg == 10) MOPOE_LAUNCH_DIRECT(2, 2, 1, 1);
    else if (cfg == 11) MOPOE_LAUNCH_DIRECT(2, 1, 2, 2);
    else if (vec) {
      if (cfg == 0) MOPOE_LAUNCH_TILE(128, 128, 2, 4, 16, 512);
      else if (cfg == 1) MOPOE_LAUNCH_TILE(256, 64, 4, 2, 16, 512);
      else if (cfg == 3) MOPOE_LAUNCH_TILE(256, 128, 4, 2, 16, 512);
      else if (cfg == 4) MOPOE_LAUNCH_TILE(128, 64, 2, 2, 16, 256);
      else if (cfg == 5) MOPOE_LAUNCH_TILE(64, 64, 2, 2, 32, 256);
      else if (cfg == 6) MOPOE_LAUNCH_TILE(128, 64, 2, 2, 32, 256);
      else if (cfg == 7) MOPOE_LAUNCH_TILE(128, 128, 2, 2, 16, 256);
      else MOPOE_LAUNCH_TILE(64, 64, 2, 2, 16, 256);
    } else {
      if (cfg == 0) hipLaunchKernelGGL((gather_gemm_kernel<128, 128, 2, 2, GEMM_BK_BIG, false>), grid, dim3(256), 0, stream, a);
      else if (cfg == 1) hipLaunchKernelGGL((gather_gemm_kernel<256, 64, 4, 1, 16, false>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((gather_gemm_kernel<64, 64, 2, 2, 16, false>), grid, dim3(256), 0, stream, a);
    }
#undef MOPOE_LAUNCH_TILE
#undef MOPOE_LAUNCH_DIRECT
    if (int rc = check_launch("gather_gemm")) return rc;
    if (a.partial && !in_kernel_reduce) {
      dim3 eg(ceil_div(Cn, 64), std::min<long>(ceil_div(a.rows_total, 4), EPI_MAX_BLOCKS_Y));
      hipLaunchKernelGGL(splitk_epilogue_kernel, eg, dim3(256), 0, stream, a);
      if (int rc = check_launch("splitk_epilogue")) return rc;
    }
  }
  return MOPOE_OK;
}

}  // namespace mopoe

using namespace mopoe;

extern "C" size_t mopoe_conv_workspace_bytes(void) { return WS_RECOMMENDED; }

extern "C" int mopoe_conv_fwd(const float* x, const float* wp, const float* bias, float* y, const mopoe_conv_geom* g,
                              const mopoe_bn_ref* bn_in, const mopoe_mask_ref* mask, double* out_stats,
                              const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = validate_geom(g)) return rc;
  if (!x || !wp || !y) { set_error("conv_fwd: null pointer"); return MOPOE_ERR_ARG; }
  const bool plain = (!bn_in || bn_in->mode == 0) && (!mask || mask->kind == 0);
  // image-side edge layers: one channel on one side -> streaming kernels instead of a 98 %-padding GEMM tile
  if (plain && !g->transposed && g->Cin == 1 && !bias && edge_supported(g, g->Cout, {wp, y}))
    return edge_expand(x, wp, y, g, g->Cout, out_stats, (hipStream_t)stream);
  if (plain && g->transposed && g->Cout == 1 && !out_stats && edge_supported(g, g->Cin, {x, wp}))
    return edge_reduce(x, wp, bias, y, g, g->Cin, (hipStream_t)stream);
  // Conv: output on the small grid.  ConvTranspose: output on the big grid (phases).
  return launch_gather(x, wp, bias, y, g, g->transposed ? 0 : 1, g->Cin, g->Cout, /*w_nk=*/0, bn_in, mask, out_stats,
                       nullptr, nullptr, nullptr, plan, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mopoe_conv_fwd_mix(const float* x, const float* wp, const float* bias, float* y, const mopoe_conv_geom* g,
                                  const mopoe_bn_ref* bn_in, const mopoe_mask_ref* mask, const mopoe_mix_ref* mix,
                                  double* out_stats, const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  if (int rc = validate_geom(g)) return rc;
  if (!x || !wp || !y || !mix) { set_error("conv_fwd_mix: null pointer"); return MOPOE_ERR_ARG; }
  return launch_gather(x, wp, bias, y, g, g->transposed ? 0 : 1, g->Cin, g->Cout, /*w_nk=*/0, bn_in, mask, out_stats,
                       nullptr, nullptr, nullptr, plan, workspace, workspace_bytes, (hipStream_t)stream, mix);
}

extern "C" int mopoe_conv_dgrad(const float* dy, const float* wp, float* dx, const mopoe_conv_geom* g,
                                const mopoe_bn_ref* relu_bn, const float* xin, double* bwd_sums,
                                const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = validate_geom(g)) return rc;
  if (!dy || !wp || !dx) { set_error("conv_dgrad: null pointer"); return MOPOE_ERR_ARG; }
  if (g->transposed && g->Cout == 1 && (!relu_bn || relu_bn->mode == 0) && edge_supported(g, g->Cin, {wp, dx}))
    return edge_expand(dy, wp, dx, g, g->Cin, nullptr, (hipStream_t)stream);
  // input gradient of a Conv lives on the big grid (phases); of a ConvTranspose on the small grid.
  return launch_gather(dy, wp, nullptr, dx, g, g->transposed ? 1 : 0, g->Cout, g->Cin, /*w_nk=*/1, nullptr, nullptr,
                       nullptr, relu_bn, xin, bwd_sums, plan, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mopoe_conv_wgrad(const float* x, const float* dy, float* dwp, const mopoe_conv_geom* g,
                                const mopoe_bn_ref* bn_in, int32_t dwp_is_zero, const mopoe_conv_plan* plan,
                                void* stream_) {
  if (int rc = validate_geom(g)) return rc;
  if (!x || !dy || !dwp) { set_error("conv_wgrad: null pointer"); return MOPOE_ERR_ARG; }
  hipStream_t stream = (hipStream_t)stream_;
  if ((!bn_in || bn_in->mode == 0) && !g->transposed && g->Cin == 1 && edge_supported(g, g->Cout, {dy, dwp}))
    return edge_wgrad(dy, x, dwp, g, g->Cout, stream, dwp_is_zero != 0);
  if ((!bn_in || bn_in->mode == 0) && g->transposed && g->Cout == 1 && edge_supported(g, g->Cin, {x, dwp}))
    return edge_wgrad(x, dy, dwp, g, g->Cin, stream, dwp_is_zero != 0);
  WgradArgs a;
  a.Xs = x; a.Dy = dy; a.dW = dwp;
  a.N = g->N; a.Hs = g->Hs; a.Ws = g->Ws; a.Hb = g->Hb; a.Wb = g->Wb; a.Cin = g->Cin; a.Cout = g->Cout;
  a.kh = g->kh; a.kw = g->kw; a.sh = g->sh; a.sw = g->sw; a.ph = g->ph; a.pw = g->pw;
  a.x_is_big = g->transposed ? 0 : 1;
  a.Ms = (long)g->N * g->Hs * g->Ws;
  mopoe_bn_ref none = {};
  a.bn_in = bn_in ? *bn_in : none;
  if (a.bn_in.mode != 0 && (a.bn_in.C != g->Cin || g->Cin > MAX_BN_C)) { set_error("wgrad bn_in channel mismatch"); return MOPOE_ERR_ARG; }
  a.vecI = (g->Cin % 4 == 0) && aligned16(x);
  a.vecJ = (g->Cout % 4 == 0) && aligned16(dy);
  const size_t rows_x = (size_t)g->N * (g->transposed ? g->Hs * g->Ws : g->Hb * g->Wb);
  const size_t rows_dy = (size_t)g->N * (g->transposed ? g->Hb * g->Wb : g->Hs * g->Ws);
  const size_t xb = rows_x * g->Cin * sizeof(float), db = rows_dy * g->Cout * sizeof(float);
  const bool vec = a.vecI && a.vecJ && xb < (1ull << 31) && db < (1ull << 31);
  a.x_bytes = (unsigned)std::min<size_t>(xb, 0x7fffffffu);
  a.dy_bytes = (unsigned)std::min<size_t>(db, 0x7fffffffu);
  if (rows_x >= (1ull << 31) || rows_dy >= (1ull << 31)) { set_error("wgrad: more than 2^31 rows"); return MOPOE_ERR_ARG; }
  // fast pixel addressing: chunks of BK = 16 pixels never straddle an image row (or cover whole rows of one image)
  const int hw_s = g->Hs * g->Ws;
  a.fast = (vec && ((g->Ws % BK == 0) || (BK % g->Ws == 0 && hw_s % BK == 0))) ? 1 : 0;
  const int taps = g->kh * g->kw;
  bool big = g->Cin > 64 && g->Cout > 64;
  // plan tiles: 0 = 128x128, 2 = 64x64 (register-staged, 16 pixels per chunk); 5 = 128x128, 6 = 64x64 on LDS-DMA (32 pixels per
  // stage: conv_gemm_glds.inc; vector path only)
  // 9 = four taps (one parity class of a k4 s2 p1 kernel) per block, 64 x 64 channels, products on the bf16 matrix pipe
  // (conv_gemm_glds_parity.inc: wgrad_parity_f32_kernel)
  if (plan && (plan->tile == 9 || plan->tile == 10)) {
    const bool ok = a.bn_in.mode == 0 && vec && g->kh == 4 && g->kw == 4 && g->sh == 2 && g->sw == 2 && g->ph == 1 && g->pw == 1 &&
                    g->Hs % 8 == 0 && g->Ws % 8 == 0 && g->Hb == 2 * g->Hs && g->Wb == 2 * g->Ws;
    if (!ok) {
      set_error("wgrad plan: tiles 9 / 10 (four taps per block) need a plain operand, the vector path, k4 s2 p1 and a small grid of whole 8 x 8 tiles");
      return MOPOE_ERR_ARG;
    }
    const int Cg = a.x_is_big ? g->Cin : g->Cout, Csm = a.x_is_big ? g->Cout : g->Cin;
    const int cs = plan->tile == 10 ? 128 : 64;
    if (cs == 128 && Csm % 128 != 0) { set_error("wgrad plan: tile 10 needs a multiple of 128 channels on the small-grid operand (%d)", Csm); return MOPOE_ERR_ARG; }
    const long ntiles = (long)g->N * (g->Hs / 8) * (g->Ws / 8);
    const long cblocks = (long)ceil_div(Cg, 64) * ceil_div(Csm, cs) * 4;
    long split = plan->split > 0 ? plan->split : (256 + cblocks - 1) / cblocks;
    if (split > ntiles) split = ntiles;
    if (split < 1) split = 1;
    a.chunk = (ntiles + split - 1) / split;             // (in 8 x 8 tiles)
    split = (ntiles + a.chunk - 1) / a.chunk;
    a.atomic = split > 1;
    static const bool xcd_remap9 = !getenv("MOPOE_NO_XCD_REMAP");
    a.xcd_remap = xcd_remap9 ? 1 : 0;
    if (a.atomic && !dwp_is_zero) {
      if (hipMemsetAsync(dwp, 0, (size_t)taps * g->Cin * g->Cout * sizeof(float), stream) != hipSuccess) { set_error("wgrad memset failed"); return MOPOE_ERR_LAUNCH; }
    }
    ProfScope prof(stream, 2.0 * (double)a.Ms * g->Cin * (double)g->Cout * taps, PROF_F32_WGRAD_PARITY + (cs == 128 ? 2 : 0) + (a.x_is_big ? 0 : 1), (double)xb + (double)db);
    const dim3 grid((unsigned)(ceil_div(Cg, 64) * ceil_div(Csm, cs)), 4, (unsigned)split);
    if (cs == 128 && a.x_is_big) hipLaunchKernelGGL((wgrad_parity_f32_kernel<128, true>), grid, dim3(256), 0, stream, a);
    else if (cs == 128) hipLaunchKernelGGL((wgrad_parity_f32_kernel<128, false>), grid, dim3(256), 0, stream, a);
    else if (a.x_is_big) hipLaunchKernelGGL((wgrad_parity_f32_kernel<64, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((wgrad_parity_f32_kernel<64, false>), grid, dim3(256), 0, stream, a);
    return check_launch("wgrad_parity_f32 (four taps per block)");
  }
  // 7, 8 = tiles 5, 6 with the fp32 products on the bf16 matrix pipe (EMU; plain operand only)
  const bool wemu = plan && (plan->tile == 7 || plan->tile == 8);
  if (wemu && a.bn_in.mode != 0) { set_error("wgrad plan: tile %d (split-bf16 products) has no BN-on-load form", plan->tile); return MOPOE_ERR_ARG; }
  if (plan && (plan->tile == 2 || plan->tile == 6 || plan->tile == 8)) big = false;   // the plan may ask for 64x64 tiles on wide layers too
  else if (plan && (plan->tile == 5 || plan->tile == 7)) {
    // (mirrors the tuner, mimic_amd/ops.py: _wgrad_candidates -- the 128 tile is never offered with <= 64 channels on a side)
    if (!big) { set_error("wgrad plan: tile %d (128x128 on LDS-DMA) needs more than 64 channels on both sides (%d, %d)", plan->tile, g->Cin, g->Cout); return MOPOE_ERR_ARG; }
  }
  else if (plan && plan->tile > 2) { set_error("wgrad plan: tile %d (see mopoe_conv_plan in mopoe_hip.h: -1, 0, 2, 5, 6, 7, 8, 9, 10)", plan->tile); return MOPOE_ERR_ARG; }
  const bool glds = plan && plan->tile >= 5;
  if (glds && !vec) { set_error("wgrad plan: the LDS-DMA tiles need the vector path (channel counts %% 4 == 0, aligned tensors)"); return MOPOE_ERR_ARG; }
  const int T = big ? 128 : 64;
  const int nI = ceil_div(g->Cin, T), nJ = ceil_div(g->Cout, T);
  a.nJ = nJ;
  const long tiles = (long)nI * nJ * taps;
  // split the pixel reduction until ~1024 blocks are in flight, keeping >= 8 K-chunks per block
  long split = (1024 + tiles - 1) / tiles;
  if (plan && plan->split > 0) split = plan->split;
  const int kp = glds ? 32 : BK;                  // pixels per chunk
  const long max_split = glds ? (a.Ms + 4 * kp - 1) / (4 * kp) : (a.Ms + 8 * BK - 1) / (8 * BK);
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
  const int spec = a.fast ? (a.bn_in.mode != 0 ? 2 : 1) : 0;
  const double abytes = ((double)g->N * g->Hs * g->Ws * (g->transposed ? g->Cin : g->Cout)
                         + (double)g->N * g->Hb * g->Wb * (g->transposed ? g->Cout : g->Cin)) * sizeof(float);   // both operands, once
  ProfScope prof(stream, flops, wemu ? PROF_F32_WGRAD_GLDS_EMU + (big ? 0 : 1)
                                : glds ? PROF_F32_WGRAD_GLDS + (big ? 0 : 2) + (a.bn_in.mode != 0 ? 1 : 0)
                                     : (vec ? PROF_WGRAD_VEC + (big ? 0 : 3) + spec : PROF_WGRAD_SCALAR + (big ? 0 : 1)), abytes);
  dim3 grid(nI * nJ, taps, (unsigned)split);
  if (wemu) {
    if (big) hipLaunchKernelGGL((wgrad_gemm_f32_glds_kernel<128, false, 2, 1>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((wgrad_gemm_f32_glds_kernel<64, false, 4, 1>), grid, dim3(256), 0, stream, a);
  } else if (glds) {
    const bool xf = a.bn_in.mode != 0;
    if (big) {
      if (xf) hipLaunchKernelGGL((wgrad_gemm_f32_glds_kernel<128, true, 2>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((wgrad_gemm_f32_glds_kernel<128, false, 2>), grid, dim3(256), 0, stream, a);
    } else {
      if (xf) hipLaunchKernelGGL((wgrad_gemm_f32_glds_kernel<64, true, 2>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((wgrad_gemm_f32_glds_kernel<64, false, 4>), grid, dim3(256), 0, stream, a);
    }
  } else if (vec) {
    if (big) {
      if (spec == 1) hipLaunchKernelGGL((wgrad_gemm_kernel<128, 128, true, 1>), grid, dim3(256), 0, stream, a);
      else if (spec == 2) hipLaunchKernelGGL((wgrad_gemm_kernel<128, 128, true, 2>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((wgrad_gemm_kernel<128, 128, true, 0>), grid, dim3(256), 0, stream, a);
    } else {
      if (spec == 1) hipLaunchKernelGGL((wgrad_gemm_kernel<64, 64, true, 1>), grid, dim3(256), 0, stream, a);
      else if (spec == 2) hipLaunchKernelGGL((wgrad_gemm_kernel<64, 64, true, 2>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((wgrad_gemm_kernel<64, 64, true, 0>), grid, dim3(256), 0, stream, a);
    }
  } else {
    if (big) hipLaunchKernelGGL((wgrad_gemm_kernel<128, 128, false>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((wgrad_gemm_kernel<64, 64, false>), grid, dim3(256), 0, stream, a);
  }
  return check_launch("wgrad_gemm");
}
