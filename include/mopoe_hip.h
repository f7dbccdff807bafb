/* libmopoe_hip.so -- C ABI of the MI355X (gfx950) MoPoE joint-ELBO hot path.
 *
 * The reference (Jimmy2027/MoPoE-MIMIC) is pure Python on PyTorch: it has no FFI.  Each entry point
 * below names the reference call site(s) whose arithmetic it replaces (paths relative to the
 * reference repository root); INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer into caller-owned memory
 *     (PyTorch's caching allocator in the shipped host code); the library never allocates, frees or
 *     retains a pointer past return; outputs and workspaces are caller-allocated.
 *   - work is enqueued on the caller's `stream` (a hipStream_t passed as void*); no host sync inside.
 *   - return 0 on success, a negative code on error (mopoe_last_error() gives the text); nothing
 *     throws across the ABI.
 *   - activations are fp32, channels-last: a [N,H,W,C] tensor is a row-major [N*H*W, C] matrix
 *     ("rows" = pixels, "cols" = channels); 1-D sequences use H = 1.  Image inputs/outputs with C = 1
 *     are therefore bit-identical to the reference's NCHW tensors.
 *   - conv weights are "packed": Wp[kh*kw][Cin][Cout] (tap-major, Cout contiguous), for Conv and
 *     ConvTranspose alike (Cin/Cout are those of the FORWARD op).  mimic_amd converts to and from
 *     the reference's [Cout,Cin,kh,kw] / [Cin,Cout,kh,kw] layouts in state_dict hooks.
 *   - "stats"/"sums" buffers are double[2*C]: {sum_c, sumsq_c} (or the two BN-backward sums); the
 *     kernels ACCUMULATE into them with atomics, the caller zeroes them.
 */
#ifndef MOPOE_HIP_H
#define MOPOE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOPOE_ABI_VERSION 19

/* error codes */
#define MOPOE_OK 0
#define MOPOE_ERR_ARG (-1)     /* inconsistent shapes / unsupported geometry */
#define MOPOE_ERR_LAUNCH (-2)  /* hipLaunch / hipMemset failed */
#define MOPOE_ERR_DEVICE (-3)  /* no gfx950 device / wrong arch */

int mopoe_abi_version(void);
const char* mopoe_last_error(void);

/* Geometry of one Conv{1,2}d / ConvTranspose{1,2}d.  "small" grid = conv output / convT input,
 * "big" grid = conv input / convT output; big = small*stride - pad + tap. */
typedef struct {
  int32_t N;             /* batch */
  int32_t Hs, Ws;        /* small grid */
  int32_t Hb, Wb;        /* big grid  */
  int32_t Cin, Cout;     /* channels of the forward op */
  int32_t kh, kw;
  int32_t sh, sw;        /* stride */
  int32_t ph, pw;        /* padding */
  int32_t transposed;    /* 0: Conv (big -> small), 1: ConvTranspose (small -> big) */
} mopoe_conv_geom;

/* A BatchNorm whose normalisation is applied/inverted inside another kernel.
 * mode 0: absent.  mode 1: batch statistics from `sums` (train).  mode 2: running stats (eval).
 * mode 3 (relu_bn of mopoe_conv_dgrad* only): as mode 1, but the tensor passed as `xin` is the ACTIVATION
 *         y = relu(bn(x)) instead of x: the ReLU mask is [y > 0] and xhat = (y - beta) / gamma where it is set. */
typedef struct {
  const double* sums;    /* [2*C] sum, sumsq over `count` rows (mode 1) */
  const float* gamma;    /* [C] */
  const float* beta;     /* [C] */
  const float* rmean;    /* [C] (mode 2) */
  const float* rvar;     /* [C] (mode 2) */
  double inv_count;      /* 1 / rows (mode 1) */
  float eps;
  int32_t C;
  int32_t mode;
} mopoe_bn_ref;

/* dropout multiplier applied in an epilogue: kind 0 none, 1 per-(sample,channel) [N,C]
 * (nn.Dropout2d), 2 per element [rows,C] (nn.Dropout).  Values are the multiplier itself (0 or 2). */
typedef struct {
  const float* mask;
  int32_t kind;
  int32_t rows_per_sample;
} mopoe_mask_ref;

/* ---- convolution family (implicit-GEMM on fp32 MFMA) --------------------------------------------
 * Replaces torch.nn.Conv1d/Conv2d/ConvTranspose1d/ConvTranspose2d/Linear forward and their autograd
 * backward as called from mimic/networks/ResidualBlocks.py:20-33,51-65,84-97,118-131,
 * FeatureExtractorImg.py:61-81, DataGeneratorImg.py:93-98, word_encoding/mmvae_text_enc.py:58-85,
 * word_encoding/DataGeneratorText.py:83-98, FeatureCompressor.py:21-28,
 * ConvNetworksImgMimic.py:51-52, ConvNetworksTextMimic.py:57-58.
 *
 * y = mask * (conv(act(x)) + bias), act(x) = relu(bn(x)) when bn_in.mode != 0 (the BN -> ReLU that
 * precedes every conv inside a residual block is fused into the operand load).
 * out_stats (optional) += {sum, sumsq} of the stored y per output channel.
 *
 * workspace: caller-owned scratch (mopoe_conv_workspace_bytes() is the recommended size; may be NULL/0).  Its first
 * 64 KiB hold the arrival counters of in-kernel split reductions: they must be ZERO when a workspace is first handed
 * over, every launch leaves them zero, and a workspace must not be shared by launches that may run concurrently
 * (one workspace per stream).
 * Layers whose output grid cannot fill the chip split the tap x channel reduction across blocks, park
 * the partial sums there and finish (bias, mask, statistics) in a second small kernel. */
size_t mopoe_conv_workspace_bytes(void);

/* Launch plan of one conv op (NULL = the library's static heuristic).  Every plan computes the same sums in
 * a different association order; the host mirror times the candidates the first time it meets a
 * (op, geometry, fusion) triple during warm-up and keeps the fastest (mimic_amd/ops.py, MOPOE_AUTOTUNE).
 * fp32 family (mopoe_conv_fwd / _fwd_mix / _dgrad):
 *   tile  -1 auto | 0 = 128x128 | 1 = 256x64 | 2 = 64x64 | 3 = 256x128 | 4 = 128x64 output tile (0, 1, 3: 8 waves per block;
 *          2, 4: 4 waves) | 5, 6 = tiles 2, 4 with a 32-deep K chunk (K channels % 32 == 0) | 7 = 128x128 with 4 waves
 *          | 8..11 = 128x128, 256x64, 64x64, 128x64 tiles of the LDS-free kernel (operands streamed from global memory into the
 *            MFMA registers; channel counts multiples of 8)
 *          | 12..15 = 128x128, 128x64, 64x64, 256x128 on LDS-DMA (csrc/conv_gemm_glds.inc: buffer_load ... lds, 32-deep stages,
 *            2-4 LDS buffers; K channels % 32 == 0, channel counts % 4 == 0; with BN -> ReLU on load: 12, 14, 15 only)
 *          | 16..19 = tiles 12..15 with the fp32 products computed on the bf16 matrix pipe: each fp32 operand value is split
 *            EXACTLY into three bf16 parts in registers (a = h + m + l) and six of the nine partial products (all but m l, l m, l l,
 *            each below 2^-24 |a b|) are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- results within fp32 rounding of the
 *            fp32-MFMA tiles', measured closer to fp64 than theirs; plain operand forms only (no BN -> ReLU on load); 19 has 3 buffers
 *   split  0 auto | n >= 1 blocks sharing one tile's tap x channel reduction, finished by the last-arriving block (needs workspace)
 * fp32 weight gradient (mopoe_conv_wgrad):
 *   tile  -1 auto | 0 = 128x128 | 2 = 64x64 (Cin x Cout tile of one tap, register-staged, 16 pixels per chunk)
 *          | 5 = 128x128, 6 = 64x64 on LDS-DMA (32 pixels per stage; channel counts % 4 == 0; 5 needs > 64 channels on both sides)
 *          | 7, 8 = tiles 5, 6 with the fp32 products on the bf16 matrix pipe (as tiles 16..19 above; plain operand only)
 *          | 9, 10 = FOUR taps (one parity class of a k4 s2 p1 kernel) per block, 64 gathered channels x 64 / 128 channels of the
 *            small-grid operand, products on the bf16 matrix pipe (csrc/conv_gemm_glds_parity.inc; plain operand; channel counts
 *            % 4 == 0; small grid of whole 8 x 8 tiles; 10 needs a multiple of 128 channels there); split = blocks sharing the
 *            pixel TILES
 *   split  0 auto | n >= 1 blocks sharing one tile's pixel reduction (atomics into dwp)
 * bf16 family (mopoe_conv_fwd_bf16 / _fwd_mix_bf16 / _dgrad_bf16):
 *   tile  -1 auto | 0 = 128x128 | 1 = 256x64 | 2 = 64x64 | 3 = 256x128 | 4 = 128x64 (register-staged, 32-deep K chunk)
 *          | 5 = 128x128 (2 buffers) | 6 = 128x128 (3) | 7 = 256x128 (2) | 9 = 128x64 (2) | 10 = 128x64 (3) | 11 = 64x64 (4) on LDS-DMA
 *            (csrc/conv_gemm_bf16_glds.inc: 64-deep stages; K channels % 64 == 0; with BN -> ReLU on load: 5, 7, 9, 11 only);
 *            8 (256x128, 3 buffers) is refused: it spills registers
 * bf16 weight gradient (mopoe_conv_wgrad_bf16):
 *   tile  -1 auto | 0 = 128x128 | 2 = 64x64 (register-staged) | 5 = 128x128 (needs > 64 channels on both sides), 6 = 64x64 on LDS-DMA
 *          | 7 = 128x128 with TWO taps per block on the gathered operand's side (that side has 64 channels; plain operand; even tap count)
 *          | 8, 9 = FOUR taps (one parity class of a k4 s2 p1 kernel) per block, 64 gathered channels x 64 / 128 channels of the
 *            small-grid operand (plain operand; small grid of whole 8 x 8 tiles; 9 needs a multiple of 128 channels there);
 *            split = blocks sharing the pixel TILES
 */
typedef struct {
  int32_t tile;
  int32_t split;
} mopoe_conv_plan;

int mopoe_conv_fwd(const float* x, const float* wp, const float* bias, float* y,
                   const mopoe_conv_geom* g, const mopoe_bn_ref* bn_in, const mopoe_mask_ref* mask,
                   double* out_stats, const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes,
                   void* stream);

/* conv2 of a residual block with the residual mix in its epilogue (ResidualBlocks.py:31-32,63-64,95-96,129-130:
 * `out = self.a * residual + self.b * out`, residual = BN(shortcut conv), a = 2.0, b = 0.3 at every call site):
 *   y = a * bn(s) + b * mask * (conv(act(x)) + bias),   out_stats += {sum, sumsq} of the stored y
 * s: the shortcut conv's output, y's shape and storage type; bn: its BatchNorm (batch statistics complete, i.e. the
 * shortcut conv was enqueued earlier on the same stream).  drop2(conv2(.)) is never written to memory.
 * Needs the vector path (channel counts multiples of 4, 16-byte aligned tensors), else MOPOE_ERR_ARG. */
typedef struct {
  const void* s;
  mopoe_bn_ref bn;
  float a, b;
} mopoe_mix_ref;
int mopoe_conv_fwd_mix(const float* x, const float* wp, const float* bias, float* y, const mopoe_conv_geom* g,
                       const mopoe_bn_ref* bn_in, const mopoe_mask_ref* mask, const mopoe_mix_ref* mix,
                       double* out_stats, const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes,
                       void* stream);

/* dx = d(conv)/d(input) applied to dy.  If relu_bn.mode != 0 the ReLU that fed the conv is inverted in
 * the epilogue, dx *= [bn(xin) > 0], and bwd_sums (optional) += {sum dx, sum dx*xhat} per input channel
 * (the two reductions BatchNorm's backward needs), xhat = (xin - mean) * rstd. */
int mopoe_conv_dgrad(const float* dy, const float* wp, float* dx, const mopoe_conv_geom* g,
                     const mopoe_bn_ref* relu_bn, const float* xin, double* bwd_sums,
                     const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes, void* stream);

/* dwp[kh*kw][Cin][Cout] = d(conv)/d(weight); x is transformed by relu(bn(x)) when bn_in.mode != 0.
 * dwp is overwritten; when the pixel reduction is split across blocks the partial products are accumulated
 * with atomics into a zero-filled dwp: the library zero-fills it unless the caller states dwp_is_zero != 0
 * (callers that carve all weight gradients of a network out of one zeroed arena save ~100 memsets a step). */
int mopoe_conv_wgrad(const float* x, const float* dy, float* dwp, const mopoe_conv_geom* g,
                     const mopoe_bn_ref* bn_in, int32_t dwp_is_zero, const mopoe_conv_plan* plan, void* stream);

/* ---- residual-block glue (HBM-bound elementwise + column reductions) -----------------------------
 * out = a * bn_s(s) + b * m           (ResidualBlocks.py:31-32,63-64,95-96,129-130 with the
 *                                       shortcut BatchNorm of make_res_block_* fused)
 * out_stats (optional) += {sum, sumsq} of out (feeds the next block's bn1). */
int mopoe_block_out_fwd(const float* s, const float* m, float* out, int64_t rows, int32_t C,
                        const mopoe_bn_ref* bn_s, float a, float b, double* out_stats, void* stream);

/* out = relu(bn(x)): the operand of a block's second conv written out once (the BatchNorm + ReLU in front of the k4
 * conv, ResidualBlocks.py:26-29,58-61,90-93,124-127), for the layers where the conv's BN -> ReLU-on-load form costs more
 * than this pass (the up-sampling convs); the conv and its weight gradient then take `out` as a plain operand. */
int mopoe_bn_relu_apply(const float* x, float* out, int64_t rows, int32_t C, const mopoe_bn_ref* bn, void* stream);

/* sums += {sum g, sum g*shat} over rows: the reductions for the shortcut BatchNorm's backward. */
int mopoe_bn_bwd_reduce(const float* g, const float* s, int64_t rows, int32_t C,
                        const mopoe_bn_ref* bn_s, double* sums, void* stream);

/* backward of mopoe_block_out_fwd:
 *   dm = b * g * mask                     (gradient w.r.t. the main conv2 output, before dropout2)
 *   ds = a * BatchNormBackward(g; s)      (gradient w.r.t. the shortcut conv output)
 *   dgamma_s = a * sums[1], dbeta_s = a * sums[0]
 * colsum_dm / colsum_ds (optional, float[C], caller-zeroed) += column sums (= conv bias gradients). */
int mopoe_block_out_bwd(const float* g, const float* s, float* dm, float* ds, int64_t rows, int32_t C,
                        const mopoe_bn_ref* bn_s, const double* sums, const mopoe_mask_ref* mask,
                        float a, float b, float* dgamma, float* dbeta, float* colsum_dm,
                        float* colsum_ds, void* stream);

/* dx = mask * BatchNormBackward(dy; x) + add, with dy already ReLU-masked and sums = {sum dy,
 * sum dy*xhat} (both from mopoe_conv_dgrad's epilogue).  dgamma = sums[1], dbeta = sums[0].
 * colsum_dx (optional) += column sums of dx.
 * next_s / next_bn / next_sums (optional, all or none): dx is the gradient entering the PREVIOUS residual block, whose
 * backward starts with mopoe_bn_bwd_reduce(dx, s_prev, bn_s_prev); passing that block's shortcut output and BatchNorm
 * here accumulates next_sums (caller-zeroed double[2][C]) += {sum dx, sum dx*shat_prev} in the same pass. */
int mopoe_bn_bwd_apply(const float* dy, const float* x, const float* add, float* dx, int64_t rows,
                       int32_t C, const mopoe_bn_ref* bn, const double* sums,
                       const mopoe_mask_ref* mask, float* dgamma, float* dbeta, float* colsum_dx,
                       const float* next_s, const mopoe_bn_ref* next_bn, double* next_sums, void* stream);

/* running_mean/var momentum update for `n` BatchNorm layers in one launch (torch.nn.BatchNorm
 * train-mode side effect).  desc is a HOST array of n records {sums*, rmean*, rvar*, C, count} (device pointers
 * inside); it is copied into the kernel arguments before the call returns. */
typedef struct {
  const double* sums;
  float* rmean;
  float* rvar;
  int32_t C;
  int32_t count;
} mopoe_bn_running_desc;
int mopoe_bn_running_update(const mopoe_bn_running_desc* desc, int32_t n, float momentum, void* stream);

/* column sums of a [rows, C] matrix into float out[C] (overwritten). */
int mopoe_colsum(const float* x, float* out, int64_t rows, int32_t C, int32_t out_is_zero, void* stream);

/* ---- latent space ----------------------------------------------------------------------------------
 * One kernel for BaseMMVae.inference's subset loop (mimic/utils/BaseMMVae.py:148-177), mm_div.poe
 * (evaluation/divergence_measures/mm_div.py:10-17), utils.mixture_component_selection
 * (utils/utils.py:55-77), calc_group_divergence_moe / calc_kl_divergence (mm_div.py:90-110,
 * kl_div.py:8-16), losses.calc_klds (evaluation/losses.py:24-31) and utils.reparameterize
 * (utils/utils.py:45-48).
 *   mu_in/lv_in[3]: per-modality (mu, logvar) [B,D] in the order PA, Lateral, text; NULL if absent.
 *   K = number of subsets whose members are all present (order: PA, Lateral, text, Lateral_PA,
 *       PA_text, Lateral_text, Lateral_PA_text).
 *   row_start[K+1]: HOST array (copied into the launch), rows [row_start[k], row_start[k+1]) of the
 *       joint posterior come from subset k (the floor(B*w_k) partition).  w[K]: HOST array of the
 *       re-normalised mixture weights.
 *   outputs: mus/lvs [K,B,D]; joint_mu/joint_lv/z [B,D]; klds [K] = KL_k / norm;
 *       joint_div [1] = sum_k w[k]*klds[k].  eps [B,D] is the N(0,1) noise.  kl_ws: double[K+1]
 *       workspace that must be zero on entry and is left zero. */
int mopoe_latent_fwd(const float* const mu_in[3], const float* const lv_in[3], const float* eps,
                     int32_t B, int32_t D, const int32_t* row_start, const float* w, float norm,
                     float* mus, float* lvs, float* joint_mu, float* joint_lv, float* z, float* klds,
                     float* joint_div, double* kl_ws, void* stream);

/* backward: any of g_mus, g_lvs, g_joint_mu, g_joint_lv, g_z, g_klds, g_joint_div may be NULL.
 * d_mu_in/d_lv_in[3] are overwritten for present modalities. */
int mopoe_latent_bwd(const float* const mu_in[3], const float* const lv_in[3], const float* eps,
                     int32_t B, int32_t D, const int32_t* row_start, const float* w, float norm,
                     const float* g_mus, const float* g_lvs, const float* g_joint_mu,
                     const float* g_joint_lv, const float* g_z, const float* g_klds,
                     const float* g_joint_div, float* const d_mu_in[3], float* const d_lv_in[3],
                     void* stream);

/* ---- likelihoods -----------------------------------------------------------------------------------
 * Laplace(loc = x_hat, scale): out[0] = -sum log p(x | x_hat) / norm
 * (modalities/Modality.py:25-30 with dist.Laplace, networks/VAEtrimodalMimic.py:55-57,
 *  evaluation/losses.py:17).  ws: double[2], zero on entry, left zero. */
int mopoe_laplace_nll_fwd(const float* x_hat, const float* x, int64_t n, float scale, float norm,
                          float* out, double* ws, void* stream);
/* d x_hat = g[0] * sign(x_hat - x) / (scale * norm) */
int mopoe_laplace_nll_bwd(const float* x_hat, const float* x, const float* g, int64_t n, float scale,
                          float norm, float* dx_hat, void* stream);

/* row-wise log-softmax over [rows, V] (nn.LogSoftmax(dim=1) of word_encoding/DataGeneratorText.py:77
 * in channels-last form); y may alias x. */
int mopoe_logsoftmax_fwd(const float* x, float* y, int64_t rows, int32_t V, void* stream);
/* dx = dy - exp(y) * rowsum(dy); dx may alias dy. */
int mopoe_logsoftmax_bwd(const float* dy, const float* y, float* dx, int64_t rows, int32_t V,
                         void* stream);
/* OneHotCategorical NLL of float-encoded token ids (modalities/MimicText.py:37-40):
 * out[0] = -sum_r logp[r, ids[r]] / norm.  ws: double[2], zero on entry, left zero. */
int mopoe_token_nll_fwd(const float* logp, const float* ids, int64_t rows, int32_t V, float norm,
                        float* out, double* ws, void* stream);
/* dlogp = -g[0]/norm at [r, ids[r]], zero elsewhere (dlogp is overwritten). */
int mopoe_token_nll_bwd(const float* ids, const float* g, int64_t rows, int32_t V, float norm,
                        float* dlogp, void* stream);

/* OneHotCategorical NLL of a DENSE target (text_encoding='char': modalities/MimicText.py:37-40 passes the
 * [B, L, num_features] one-hot tensor itself, modalities/Modality.py:25-30 sums target * log p):
 * out[0] = -sum(target * logp) / norm over n elements.  ws: double[2], zero on entry, left zero.
 * backward: dlogp = -g[0] / norm * target (overwritten). */
int mopoe_dense_nll_fwd(const float* logp, const float* target, int64_t n, float norm, float* out, double* ws,
                        void* stream);
int mopoe_dense_nll_bwd(const float* target, const float* g, int64_t n, float norm, float* dlogp, void* stream);

/* Per-row log-probabilities for the importance-sampled likelihood estimator (reference:
 * mimic/utils/likelihood.py:119-120,185-186 `likelihood.log_prob(x_rep).view(B*K, -1).sum(dim=1)` on a target repeated
 * K times).  Row r of the decoder output is scored against target row r % target_rows; nothing is repeated in memory.
 *   laplace: out[r] = sum_j ( -log(2 scale) - |x[r % B][j] - x_hat[r][j]| / scale )      x_hat [rows, per_row], x [B, per_row]
 *   token:   out[r] = sum_l logp[r][l][ids[r % B][l]]                                    logp [rows, L, V], ids [B, L] (float) */
int mopoe_laplace_logprob_rows(const float* x_hat, const float* x, int64_t rows, int64_t per_row,
                               int64_t target_rows, float scale, float* out, void* stream);
int mopoe_token_logprob_rows(const float* logp, const float* ids, int64_t rows, int32_t L, int32_t V,
                             int64_t target_rows, float* out, void* stream);
/* text_encoding='char' (MimicText.py:37-40 skips the one-hot step; utils/likelihood.py:103-104): the target IS a dense
 * [B, L, num_features] tensor: out[r] = sum_i target[(r % target_rows) per_row + i] * logp[r per_row + i] */
/* Gradient of the LOGITS for the token NLL in one pass (DataGeneratorText.py:64-77 LogSoftmax + MimicText.py:37-40 +
 * Modality.py:25-30 backward): dx[r, v] = g[0] / norm * (exp(logp[r, v]) - [v == ids[r]]); dx fp32 or bf16 (dx_is_bf16).
 * Replaces mopoe_token_nll_bwd (a memset + scatter of a [rows, V] tensor) followed by mopoe_logsoftmax_bwd. */
int mopoe_token_softmax_grad(const float* logp, const float* ids, const float* g, int64_t rows, int32_t V, float norm,
                             void* dx, int32_t dx_is_bf16, void* stream);
/* The FRONT of a residual block, bn1 -> relu -> conv1 (1x1) -> dropout -> bn2 -> relu (reference ResidualBlocks.py:84-97,118-131;
 * 1-D :20-33,51-65), as streaming kernels that never write d1 = drop1(conv1(.)) (csrc/pointwise.hip; bf16 family, C = 64,
 * dropout mask absent or per (sample, channel) with rows_per_sample % 32 == 0; x, a2, dh2, dh1: [rows, C] bf16; w1: conv1's
 * packed weight [C][C] (bf16 copy); bias [C] or NULL):
 *   mopoe_block_front_stats_bf16   stats_d1[2][C] += {sum, sumsq} of d1 (rounded to bf16 as if stored)       reads x
 *   mopoe_block_front_apply_bf16   a2 = relu(bn2(d1)), d1 recomputed bit for bit                             reads x, writes a2
 *   mopoe_block_front_bwd_bf16     dh2 = gradient of bn2's output with its ReLU mask applied and sums2 = {sum dh2, sum dh2 xhat2}
 *                                  (what mopoe_conv_dgrad_bf16 with relu_bn mode 3 / xin = a2 produces) ->
 *                                  dh1 = [relu(bn1(x)) > 0] * (dc1 W1^T) with dc1 = mask * bn2-backward(dh2), rounded to bf16;
 *                                  sums1[2][C] += {sum dh1, sum dh1 xhat1}; dw1[C][C] += relu(bn1(x))^T dc1; dbias[C] += colsum(dc1);
 *                                  dgamma2[C] = sums2[1], dbeta2[C] = sums2[0] (bn2's affine gradients; both optional)
 * replace mopoe_conv_fwd_bf16 (1x1) + mopoe_bn_relu_apply_bf16 and mopoe_bn_bwd_apply_bf16 + mopoe_conv_dgrad_bf16 (1x1) +
 * mopoe_conv_wgrad_bf16 (1x1): 6 passes over [rows, C] instead of 12.  The gradients of bn2's affine parameters are sums2 itself. */
int mopoe_block_front_stats_bf16(const uint16_t* x, const uint16_t* w1, const float* bias, int64_t rows, int32_t C,
                                 const mopoe_bn_ref* bn1, const mopoe_mask_ref* mask1, double* stats_d1, void* stream);
int mopoe_block_front_apply_bf16(const uint16_t* x, const uint16_t* w1, const float* bias, uint16_t* a2, int64_t rows, int32_t C,
                                 const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2, const mopoe_mask_ref* mask1, void* stream);
int mopoe_block_front_bwd_bf16(const uint16_t* x, const uint16_t* dh2, const uint16_t* w1, const float* bias, uint16_t* dh1,
                               int64_t rows, int32_t C, const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2,
                               const mopoe_mask_ref* mask1, const double* sums2, double* sums1, float* dw1, float* dbias,
                               float* dgamma2, float* dbeta2, void* stream);
/* the same three kernels of the fp32 family (v_mfma_f32_32x32x2_f32; x, a2, dh2, dh1 and w1 fp32) */
int mopoe_block_front_stats(const float* x, const float* w1, const float* bias, int64_t rows, int32_t C,
                            const mopoe_bn_ref* bn1, const mopoe_mask_ref* mask1, double* stats_d1, void* stream);
int mopoe_block_front_apply(const float* x, const float* w1, const float* bias, float* a2, int64_t rows, int32_t C,
                            const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2, const mopoe_mask_ref* mask1, void* stream);
int mopoe_block_front_bwd(const float* x, const float* dh2, const float* w1, const float* bias, float* dh1, int64_t rows, int32_t C,
                          const mopoe_bn_ref* bn1, const mopoe_bn_ref* bn2, const mopoe_mask_ref* mask1, const double* sums2,
                          double* sums1, float* dw1, float* dbias, float* dgamma2, float* dbeta2, void* stream);
/* The vocabulary head WITHOUT a materialised log-softmax (reference word_encoding/DataGeneratorText.py:64-67,76-77: Conv1d k1
 * -> LogSoftmax; mimic/modalities/MimicText.py:37-40: one_hot x log-probabilities; Modality.py:25-30).  The head GEMM writes
 * the LOGITS [rows, V] once in the family's storage type (is_bf16: uint16 bf16 patterns, else float); V a multiple of 8
 * (bf16) / 4 (fp32), rows 16-byte aligned (the padded head: pad columns carry a bias of -1e30).
 *   mopoe_lse_rows:                   lse[r] = log sum_v exp(logits[r, v])                          (one pass over the logits)
 *   mopoe_token_nll_logits_fwd:       out[0] = sum_r (lse[r] - logits[r, ids[r]]) / norm            (a gather)
 *   mopoe_token_softmax_grad_logits:  dx[r, v] = g[0] / norm * (exp(logits[r, v] - lse[r]) - [v == ids[r]]), dx in the storage
 *                                     type, may alias logits
 * replace mopoe_logsoftmax_fwd + mopoe_token_nll_fwd + mopoe_token_softmax_grad on the training path. */
int mopoe_lse_rows(const void* logits, int32_t is_bf16, int64_t rows, int32_t V, float* lse, void* stream);
int mopoe_token_nll_logits_fwd(const void* logits, int32_t is_bf16, const float* lse, const float* ids, int64_t rows, int32_t V,
                               float norm, float* out, double* ws, void* stream);
int mopoe_token_softmax_grad_logits(const void* logits, int32_t is_bf16, const float* lse, const float* ids, const float* g,
                                    int64_t rows, int32_t V, float norm, void* dx, void* stream);
int mopoe_dense_logprob_rows(const float* logp, const float* target, int64_t rows, int64_t per_row,
                             int64_t target_rows, float* out, void* stream);

/* ---- embedding (word_encoding/mmvae_text_enc.py:27-28,73) ------------------------------------------
 * out[r, :] = table[(int)ids[r], :]; backward scatter-adds into dtable (overwritten), skipping
 * padding_idx. */
int mopoe_embedding_fwd(const float* ids, const float* table, float* out, int64_t rows, int32_t V,
                        int32_t D, void* stream);
int mopoe_embedding_bwd(const float* ids, const float* gout, float* dtable, int64_t rows, int32_t V,
                        int32_t D, int32_t padding_idx, void* stream);


/* ==== bf16 storage family (BASELINE configs #3, #5) ===================================================================
 * Same operations with activations, activation gradients and the MFMA operands stored as bfloat16 (uint16_t bit
 * patterns at this ABI), fp32 accumulation on v_mfma_f32_32x32x16_bf16.  Still fp32 / fp64: bias, dropout masks, BN
 * affine parameters and statistics (taken over the STORED, i.e. rounded, values), BN-backward sums, weight gradients
 * (mopoe_conv_wgrad_bf16 writes fp32: they feed Adam on the fp32 master weights), the latent kernel and the
 * likelihoods.  Weights: a bf16 copy of the packed fp32 master tensor Wp[kh*kw][Cin][Cout] -- one layout serves
 * forward (transposed LDS reads), input gradient and the weight gradient's destination.
 * Requirements: K channels % 32 == 0, N channels % 8 == 0 (glue kernels: C % 8 == 0), 16-byte aligned tensors < 2 GiB;
 * anything else returns MOPOE_ERR_ARG (there is no scalar fallback in this family).
 * y_is_f32 / dx_is_f32 != 0: the result is written as fp32 (latent projections, gradients entering the latent kernel).
 * Launch plans: tile -1 auto | 0 = 128x128 | 1 = 256x64 | 2 = 64x64 | 3 = 256x128 (8 waves) | 4 = 128x64; split as above.
 * Replaces the same reference call sites as the fp32 family (the reference itself has no bf16 mode: BASELINE.json
 * configs #3/#5 define it; SURVEY section 8c sets the tolerance, rtol 2e-2 against the fp32 reference arithmetic). */
int mopoe_conv_fwd_bf16(const uint16_t* x, const uint16_t* wp, const float* bias, void* y, int32_t y_is_f32,
                        const mopoe_conv_geom* g, const mopoe_bn_ref* bn_in, const mopoe_mask_ref* mask,
                        double* out_stats, const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes,
                        void* stream);
int mopoe_conv_fwd_mix_bf16(const uint16_t* x, const uint16_t* wp, const float* bias, uint16_t* y,
                            const mopoe_conv_geom* g, const mopoe_bn_ref* bn_in, const mopoe_mask_ref* mask,
                            const mopoe_mix_ref* mix, double* out_stats, const mopoe_conv_plan* plan, void* workspace,
                            size_t workspace_bytes, void* stream);
int mopoe_conv_dgrad_bf16(const uint16_t* dy, const uint16_t* wp, void* dx, int32_t dx_is_f32, const mopoe_conv_geom* g,
                          const mopoe_bn_ref* relu_bn, const uint16_t* xin, double* bwd_sums,
                          const mopoe_conv_plan* plan, void* workspace, size_t workspace_bytes, void* stream);
int mopoe_conv_wgrad_bf16(const uint16_t* x, const uint16_t* dy, float* dwp, const mopoe_conv_geom* g,
                          const mopoe_bn_ref* bn_in, int32_t dwp_is_zero, const mopoe_conv_plan* plan, void* stream);
/* image-side edge layers (FeatureExtractorImg.py:29-34 stem, DataGeneratorImg.py:84-90 head): the single-channel image,
 * the 3x3 taps w[9][C] and their gradients stay fp32, the wide [pixels][C] tensor is bf16.
 *   expand: out[p][c] = sum_tap scal[gather(p, tap)] * w[tap][c]     stem forward; head input gradient
 *   wgrad:  dw[tap][c] = sum_p vec[p][c] * scal[gather(p, tap)]      stem / head weight gradient
 *   reduce: out[q] = bias + sum_tap sum_c x[p(q, tap)][c] * w[tap][c]  head forward */
int mopoe_edge_expand_bf16(const float* scal, const float* w, uint16_t* out, const mopoe_conv_geom* g, int32_t C,
                           double* stats, void* stream);
int mopoe_edge_wgrad_bf16(const uint16_t* vec, const float* scal, float* dw, const mopoe_conv_geom* g, int32_t C,
                          int32_t dw_is_zero, void* stream);
int mopoe_edge_reduce_bf16(const uint16_t* x, const float* w, const float* bias, float* out, const mopoe_conv_geom* g,
                           int32_t C, void* stream);
/* residual-block glue on bf16 tensors (same arithmetic in fp32 registers; results rounded once when stored) */
int mopoe_bn_relu_apply_bf16(const uint16_t* x, uint16_t* out, int64_t rows, int32_t C, const mopoe_bn_ref* bn, void* stream);
int mopoe_block_out_fwd_bf16(const uint16_t* s, const uint16_t* m, uint16_t* out, int64_t rows, int32_t C,
                             const mopoe_bn_ref* bn_s, float a, float b, double* out_stats, void* stream);
int mopoe_bn_bwd_reduce_bf16(const uint16_t* g, const uint16_t* s, int64_t rows, int32_t C, const mopoe_bn_ref* bn_s,
                             double* sums, void* stream);
int mopoe_block_out_bwd_bf16(const uint16_t* g, const uint16_t* s, uint16_t* dm, uint16_t* ds, int64_t rows, int32_t C,
                             const mopoe_bn_ref* bn_s, const double* sums, const mopoe_mask_ref* mask, float a, float b,
                             float* dgamma, float* dbeta, float* colsum_dm, float* colsum_ds, void* stream);
int mopoe_bn_bwd_apply_bf16(const uint16_t* dy, const uint16_t* x, const uint16_t* add, uint16_t* dx, int64_t rows,
                            int32_t C, const mopoe_bn_ref* bn, const double* sums, const mopoe_mask_ref* mask,
                            float* dgamma, float* dbeta, float* colsum_dx, const uint16_t* next_s,
                            const mopoe_bn_ref* next_bn, double* next_sums, void* stream);
int mopoe_colsum_bf16(const uint16_t* x, float* out, int64_t rows, int32_t C, int32_t out_is_zero, void* stream);
/* embedding with a bf16 activation: out[r, :] = bf16(table[ids[r], :]) (fp32 table); backward scatter-adds the bf16
 * gradient rows into the fp32 dtable (overwritten), skipping padding_idx */
/* log-softmax backward with the gradient written as bf16 (it enters the vocabulary head's GEMMs as a bf16 operand);
 * dy and y stay fp32 */
int mopoe_logsoftmax_bwd_bf16out(const float* dy, const float* y, uint16_t* dx, int64_t rows, int32_t V, void* stream);
int mopoe_embedding_fwd_bf16(const float* ids, const float* table, uint16_t* out, int64_t rows, int32_t V, int32_t D,
                             void* stream);
int mopoe_embedding_bwd_bf16(const float* ids, const uint16_t* gout, float* dtable, int64_t rows, int32_t V, int32_t D,
                             int32_t padding_idx, void* stream);

/* ---- optimiser step ------------------------------------------------------------------------------------
 * Replaces exp.optimizer.step() of the reference's train loop (mimic/run_epochs.py:131) for the optimiser
 * mimic/utils/experiment.py:171-178 builds: optim.Adam(params, lr, betas), no weight decay, no amsgrad.  Arithmetic of
 * PyTorch's fused capturable Adam: step += 1; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
 * p -= (lr / (1 - b1^step)) * m / (sqrt(v) / sqrt(1 - b2^step) + eps).
 * One record per parameter tensor (HOST array; the records travel in the kernel arguments, 64 per launch, so no device
 * table has to be built or kept alive and a hipGraph capture bakes the pointers into its nodes).  g == NULL: the tensor
 * and its state are left untouched (optim.Adam skips parameters whose .grad is None).  p16 (optional): the bf16 copy of
 * the updated parameter is written in the same pass (the bf16 family's MFMA operands).
 * step: device scalar (float), incremented first (NULL: a further part of a step whose counter and coefficients an
 * earlier call with nseg >= 0 has already set in `coef`: the tensors of one step may be updated by several calls).  lr_dev: device scalar or NULL (then `lr`).  coef: float[2] device scratch.
 * The hyper-parameters are doubles, as in optim.Adam: 1 - beta2 formed from the float 0.999 is off by 1.3e-5 relative. */
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  uint16_t* p16;
  int64_t n;
} mopoe_adam_seg;
int mopoe_adam_step(const mopoe_adam_seg* segs, int32_t nseg, float* step, const float* lr_dev, double lr, double beta1,
                    double beta2, double eps, float* coef, void* stream);

/* ---- profiling support for bench.py ------------------------------------------------------------------
 * When enabled, every launch of the implicit-GEMM kernels is bracketed by HIP events on the launch
 * stream.  mopoe_prof_collect synchronises those events and fills, per kernel instantiation (arrays of
 * MOPOE_PROF_KINDS entries), the number of launches, their summed duration (ms), their summed algorithmic FLOPs and
 * their summed algorithmic bytes (one read of the input activation + one write of the result; stated by the
 * bf16 family, 0 elsewhere) since the last collect.  Kinds (the host mirror turns them into the template names rocprofv3 prints):
 *   0..31  gather_gemm_kernel, vector path: tile * 4 + spec (tile as in mopoe_conv_plan; spec 0 = run-time modes,
 *          1 forward, 2 forward with BN+ReLU on the operand, 3 input gradient)
 *   32..34 gather_gemm_kernel, scalar path: 128x128, 256x64, 64x64
 *   36..41 wgrad_gemm_kernel, vector path: (128x128 ? 0 : 3) + spec (0 run-time modes, 1 plain, 2 BN+ReLU on x)
 *   42..43 wgrad_gemm_kernel, scalar path: 128x128, 64x64
 *   44..59 direct_gemm_kernel: (tile - 8) * 4 + spec
 *   60..74 gather_gemm_bf16_kernel: tile * 3 + (spec - 1)   (tiles 0..4 of the bf16 family)
 *   75..78 wgrad_gemm_bf16_kernel: (128x128 ? 0 : 2) + (BN+ReLU on x ? 1 : 0)
 *   80..93 gather_gemm_bf16_glds_kernel, plain operand: (tile - 5) * 2 + (input gradient ? 1 : 0)   (bf16 tiles 5..11)
 *   94..98 gather_gemm_bf16_glds_kernel with BN+ReLU on load: tiles 5, 7, 9, 10, 11 -> 94..98
 *   100..103 wgrad_gemm_bf16_glds_kernel: (128x128 ? 0 : 2) + (BN+ReLU on x ? 1 : 0)
 *   104..115 gather_gemm_f32_glds_kernel: (tile - 12) * 3 + (spec - 1)   (fp32 tiles 12..15)
 *   116..119 wgrad_gemm_f32_glds_kernel: (128x128 ? 0 : 2) + (BN+ReLU on x ? 1 : 0)
 *   120..121 wgrad_gemm_bf16_glds_kernel, two taps per block (tile 7): gathered side = activations / gradient rows
 *   122..123 wgrad_parity_bf16_kernel (tiles 8 / 9: four taps per block), S tile 64 / 128
 *   124..126 pw_front_fwd_bf16_kernel<64, false> (statistics pass), <64, true> (a2 pass), pw_front_bwd_bf16_kernel<64>
 *   127..129 pw_front_fwd_f32_kernel<false>, <true>, pw_front_bwd_f32_kernel
 *   130..137 gather_gemm_f32_glds_kernel<..., EMU = 1> (fp32 tiles 16..19): (tile - 16) * 2 + (input gradient ? 1 : 0)
 *   138..139 wgrad_gemm_f32_glds_kernel<..., EMU = 1> (fp32 wgrad tiles 7, 8)
 *   140..143 wgrad_parity_f32_kernel<CS, conv ? true : false> (fp32 wgrad tiles 9 / 10): (CS == 128 ? 2 : 0) + (transposed ? 1 : 0) */
#define MOPOE_PROF_KINDS 144
int mopoe_prof_enable(int32_t on);
/* Device timestamp (ticks of the 100 MHz constant clock) written to *slot when `stream` reaches this point: a one-thread
 * kernel, so it can be captured into a hipGraph -- the only way to see WHEN the branches of a replayed graph run without
 * a profiler in the process (tests/tools/net_timeline.py). */
int mopoe_prof_stamp(uint64_t* slot, void* stream);
int mopoe_prof_collect(int64_t* launches, double* total_ms, double* total_flops, double* total_bytes /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* MOPOE_HIP_H */
