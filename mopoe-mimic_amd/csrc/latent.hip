// Fused latent-space kernel: per-subset product of experts, KL to N(0,I), mixture selection by row
// range and reparameterisation, forward and backward.  One thread per (row b, latent dim d); the seven
// KL sums are reduced with wavefront shuffles -> LDS -> one double atomic per subset per block; the last
// block to finish turns them into klds[] / joint_divergence (so the forward is a single launch).
//
// Reference arithmetic: mimic/evaluation/divergence_measures/mm_div.py:10-17 (poe, eps = 1e-8, applied to
// single-expert subsets too), kl_div.py:8-16, mimic/utils/utils.py:45-48,55-77, BaseMMVae.py:139-196.
#include "common.hpp"

namespace mopoe {

constexpr int MAXK = 7;
constexpr float POE_EPS = 1e-8f;
// subset bitmasks in the reference's subset order; bit0 PA, bit1 Lateral, bit2 text
__constant__ int kSubsetMask[MAXK] = {1, 2, 4, 3, 5, 6, 7};
// members are accumulated in sorted-name order: Lateral (1), PA (0), text (2)
__constant__ int kMemberOrder[3] = {1, 0, 2};

struct LatentArgs {
  const float* mu[3];
  const float* lv[3];
  const float* eps;
  int B, D, K;
  int row_start[MAXK + 1];
  float w[MAXK];
  int subset[MAXK];   // bitmask of active subset k
  float norm;
};

struct LatentFwdOut {
  float *mus, *lvs, *jm, *jl, *z, *klds, *jd;
  double* ws;  // [K] sums + [1] arrival counter (as double)
};

__global__ __launch_bounds__(256) void latent_fwd_kernel(const LatentArgs a, const LatentFwdOut o, int nblocks) {
  __shared__ float red[4][MAXK];
  __shared__ int is_last;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.B * a.D;
  const bool ok = idx < total;
  const long i = ok ? idx : 0;
  const int b = (int)(i / a.D);
  float m3[3], T3[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    m3[s] = 0.f; T3[s] = 0.f;
    if (a.mu[s]) {
      m3[s] = a.mu[s][i];
      T3[s] = 1.0f / (expf(a.lv[s][i]) + POE_EPS);
    }
  }
  float klp[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    klp[k] = 0.f;
    if (k < a.K) {
      const int sm = a.subset[k];
      float tsum = 0.f, msum = 0.f;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int s = kMemberOrder[q];
        if (sm & (1 << s)) { tsum += T3[s]; msum += m3[s] * T3[s]; }
      }
      const float mu = msum / tsum;
      const float lv = logf(1.0f / tsum);
      if (ok) {
        o.mus[(long)k * total + i] = mu;
        o.lvs[(long)k * total + i] = lv;
        klp[k] = 1.0f - expf(lv) - mu * mu + lv;
        if (b >= a.row_start[k] && b < a.row_start[k + 1]) {
          o.jm[i] = mu;
          o.jl[i] = lv;
          o.z[i] = a.eps[i] * expf(0.5f * lv) + mu;
        }
      }
    }
  }
  // block reduction of the K partial KL sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const float s = wave_sum(klp[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < a.K) {
    const int k = threadIdx.x;
    atomic_add_f64(o.ws + k, (double)(red[0][k] + red[1][k] + red[2][k] + red[3][k]));
  }
  // last-block finalisation (agent-scope release/acquire around the arrival counter)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned prev = atomicAdd(reinterpret_cast<unsigned*>(o.ws + MAXK), 1u);
    is_last = (prev == (unsigned)(nblocks - 1));
  }
  __syncthreads();
  if (is_last && threadIdx.x == 0) {
    __threadfence();
    float jd = 0.f;
    for (int k = 0; k < a.K; ++k) {
      const double s = __hip_atomic_load(o.ws + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float kl = (float)(-0.5 * s / (double)a.norm);
      o.klds[k] = kl;
      jd += a.w[k] * kl;
      __hip_atomic_store(o.ws + k, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    o.jd[0] = jd;
    __hip_atomic_store(reinterpret_cast<unsigned*>(o.ws + MAXK), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct LatentBwdIn {
  const float *g_mus, *g_lvs, *g_jm, *g_jl, *g_z, *g_klds, *g_jd;
  float* dmu[3];
  float* dlv[3];
};

__global__ __launch_bounds__(256) void latent_bwd_kernel(const LatentArgs a, const LatentBwdIn g) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.B * a.D;
  if (idx >= total) return;
  const long i = idx;
  const int b = (int)(i / a.D);
  float m3[3], T3[3], e3[3], dm3[3], dT3[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    m3[s] = T3[s] = e3[s] = dm3[s] = dT3[s] = 0.f;
    if (a.mu[s]) {
      m3[s] = a.mu[s][i];
      e3[s] = expf(a.lv[s][i]);
      T3[s] = 1.0f / (e3[s] + POE_EPS);
    }
  }
  const float gjd = g.g_jd ? g.g_jd[0] : 0.f;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    if (k >= a.K) continue;
    const int sm = a.subset[k];
    float tsum = 0.f, msum = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int s = kMemberOrder[q];
      if (sm & (1 << s)) { tsum += T3[s]; msum += m3[s] * T3[s]; }
    }
    const float mu = msum / tsum;
    const float lv = logf(1.0f / tsum);
    // upstream gradient on (mu_k, lv_k)
    float gmu = g.g_mus ? g.g_mus[(long)k * total + i] : 0.f;
    float glv = g.g_lvs ? g.g_lvs[(long)k * total + i] : 0.f;
    const float gk = (g.g_klds ? g.g_klds[k] : 0.f) + gjd * a.w[k];
    gmu += gk * mu / a.norm;
    glv += gk * 0.5f * (expf(lv) - 1.0f) / a.norm;
    if (b >= a.row_start[k] && b < a.row_start[k + 1]) {
      if (g.g_jm) gmu += g.g_jm[i];
      if (g.g_jl) glv += g.g_jl[i];
      if (g.g_z) {
        const float gz = g.g_z[i];
        gmu += gz;
        glv += gz * 0.5f * a.eps[i] * expf(0.5f * lv);
      }
    }
    // back through the product of experts: mu = (sum m_s T_s)/tsum, lv = -log(tsum)
    const float inv_t = 1.0f / tsum;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (sm & (1 << s)) {
        dm3[s] += gmu * T3[s] * inv_t;
        dT3[s] += gmu * (m3[s] - mu) * inv_t - glv * inv_t;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (a.mu[s]) {
      g.dmu[s][i] = dm3[s];
      g.dlv[s][i] = dT3[s] * (-T3[s] * T3[s] * e3[s]);  // dT/dlv = -exp(lv) / (exp(lv)+eps)^2
    }
  }
}

static int fill_args(LatentArgs& a, const float* const mu_in[3], const float* const lv_in[3], const float* eps, int B,
                     int D, const int32_t* row_start, const float* w, float norm) {
  int avail = 0;
  for (int s = 0; s < 3; ++s) {
    a.mu[s] = mu_in[s];
    a.lv[s] = lv_in[s];
    if ((mu_in[s] == nullptr) != (lv_in[s] == nullptr)) { set_error("latent: mu/logvar presence mismatch"); return MOPOE_ERR_ARG; }
    if (mu_in[s]) avail |= 1 << s;
  }
  if (!avail || !eps || B <= 0 || D <= 0 || !row_start || !w || norm <= 0.f) { set_error("latent: bad arguments"); return MOPOE_ERR_ARG; }
  static const int masks[MAXK] = {1, 2, 4, 3, 5, 6, 7};
  int K = 0;
  for (int k = 0; k < MAXK; ++k)
    if ((masks[k] & ~avail) == 0) a.subset[K++] = masks[k];
  for (int k = K; k < MAXK; ++k) a.subset[k] = 0;
  a.K = K;
  for (int k = 0; k <= MAXK; ++k) a.row_start[k] = k <= K ? row_start[k] : B;
  if (a.row_start[0] != 0 || a.row_start[K] != B) { set_error("latent: row_start must span [0, B]"); return MOPOE_ERR_ARG; }
  for (int k = 0; k < MAXK; ++k) a.w[k] = k < K ? w[k] : 0.f;
  a.eps = eps; a.B = B; a.D = D; a.norm = norm;
  return 0;
}

}  // namespace mopoe

using namespace mopoe;

extern "C" int mopoe_latent_fwd(const float* const mu_in[3], const float* const lv_in[3], const float* eps, int32_t B,
                                int32_t D, const int32_t* row_start, const float* w, float norm, float* mus, float* lvs,
                                float* joint_mu, float* joint_lv, float* z, float* klds, float* joint_div, double* kl_ws,
                                void* stream) {
  LatentArgs a;
  if (int rc = fill_args(a, mu_in, lv_in, eps, B, D, row_start, w, norm)) return rc;
  if (!mus || !lvs || !joint_mu || !joint_lv || !z || !klds || !joint_div || !kl_ws) { set_error("latent_fwd: null output"); return MOPOE_ERR_ARG; }
  LatentFwdOut o = {mus, lvs, joint_mu, joint_lv, z, klds, joint_div, kl_ws};
  const int nblocks = ceil_div((long)B * D, 256);
  hipLaunchKernelGGL(latent_fwd_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, a, o, nblocks);
  return check_launch("latent_fwd");
}

extern "C" int mopoe_latent_bwd(const float* const mu_in[3], const float* const lv_in[3], const float* eps, int32_t B,
                                int32_t D, const int32_t* row_start, const float* w, float norm, const float* g_mus,
                                const float* g_lvs, const float* g_joint_mu, const float* g_joint_lv, const float* g_z,
                                const float* g_klds, const float* g_joint_div, float* const d_mu_in[3],
                                float* const d_lv_in[3], void* stream) {
  LatentArgs a;
  if (int rc = fill_args(a, mu_in, lv_in, eps, B, D, row_start, w, norm)) return rc;
  LatentBwdIn g = {g_mus, g_lvs, g_joint_mu, g_joint_lv, g_z, g_klds, g_joint_div, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  for (int s = 0; s < 3; ++s) {
    g.dmu[s] = d_mu_in[s];
    g.dlv[s] = d_lv_in[s];
    if (mu_in[s] && (!d_mu_in[s] || !d_lv_in[s])) { set_error("latent_bwd: missing gradient buffer"); return MOPOE_ERR_ARG; }
  }
  const int nblocks = ceil_div((long)B * D, 256);
  hipLaunchKernelGGL(latent_bwd_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, a, g);
  return check_launch("latent_bwd");
}
