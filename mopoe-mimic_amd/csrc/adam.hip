// Adam over every parameter tensor of the model in a handful of launches (gfx950).
//
// Replaces optim.Adam.step of the reference (mimic/utils/experiment.py:171-178, driven by mimic/run_epochs.py:131) with
// the arithmetic of PyTorch's fused, capturable Adam (float state, the two moment updates and the final `+ eps` carried
// in double like that kernel's double-typed hyper-parameters make them): 7 streams of 4 bytes per parameter, HBM-bound.
// The multi-tensor kernel it replaces needs ~25 launches of <= 320 blocks for the 397 tensors / 65 M parameters of
// BASELINE config #2 (each launch ends in a tail that leaves most CUs idle: 2.8 TB/s); here the tensor records travel
// in the kernel arguments, ADAM_SEGS_PER_LAUNCH per launch, a block owns one 4096-element chunk of one tensor and finds
// it with two ballots over the chunk prefix table.
#include "common.hpp"

namespace mopoe {

constexpr int ADAM_SEGS_PER_LAUNCH = 64;     // 64 * (48 + 4) B + scalars < 4 KiB of kernel arguments
constexpr int ADAM_CHUNK = 4096;             // elements per block: 256 threads x 4 float4
constexpr int ADAM_THREADS = 256;

struct AdamPack {
  mopoe_adam_seg seg[ADAM_SEGS_PER_LAUNCH];
  int chunk_start[ADAM_SEGS_PER_LAUNCH];     // first block of each tensor; INT_MAX past the last one
};

// step += 1; coef = {lr / bias_correction1, sqrt(bias_correction2)} for that step
__global__ void adam_prep_kernel(float* step, const float* lr_dev, double lr_host, double beta1, double beta2, float* coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float s = step[0] + 1.f;
  step[0] = s;
  const double lr = lr_dev ? (double)lr_dev[0] : lr_host;
  const float bc1 = (float)(1.0 - pow(beta1, (double)s));
  const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)s));
  coef[0] = (float)(lr / (double)bc1);
  coef[1] = bc2s;
}

struct AdamHyper {
  double beta1, beta2, eps;
  float step_size, bc2_sqrt;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamHyper& h) {
  m = (float)(h.beta1 * (double)m + (1.0 - h.beta1) * (double)g);
  v = (float)(h.beta2 * (double)v + (1.0 - h.beta2) * (double)g * (double)g);
  const float denom = (float)((double)(sqrtf(v) / h.bc2_sqrt) + h.eps);
  p -= h.step_size * m / denom;
}

__global__ __launch_bounds__(ADAM_THREADS) void adam_kernel(const AdamPack pack, int nseg, const float* __restrict__ coef,
                                                            double beta1, double beta2, double eps) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = (int)blockIdx.x;
  // which tensor: the number of prefix entries <= b (ADAM_SEGS_PER_LAUNCH == 64: one entry per lane)
  static_assert(ADAM_SEGS_PER_LAUNCH == 64, "one prefix entry per lane");
  const unsigned long long le = __ballot(pack.chunk_start[lane] <= b);
  const int si = __builtin_amdgcn_readfirstlane(__popcll(le) - 1);
  if (si < 0 || si >= nseg) return;
  const mopoe_adam_seg s = pack.seg[si];
  const long base = (long)(b - pack.chunk_start[si]) * ADAM_CHUNK;
  AdamHyper h;
  h.beta1 = beta1; h.beta2 = beta2; h.eps = eps;
  h.step_size = coef[0]; h.bc2_sqrt = coef[1];

  const bool vec = ((((uintptr_t)s.p | (uintptr_t)s.g | (uintptr_t)s.m | (uintptr_t)s.v) & 15) == 0) &&
                   (s.p16 == nullptr || (((uintptr_t)s.p16) & 7) == 0);
  if (vec) {
    float4 p[4], g[4], m[4], v[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long e = base + (long)(i * ADAM_THREADS + tid) * 4;
      ok[i] = e + 3 < s.n;
      if (ok[i]) {
        g[i] = *reinterpret_cast<const float4*>(s.g + e);
        p[i] = *reinterpret_cast<const float4*>(s.p + e);
        m[i] = *reinterpret_cast<const float4*>(s.m + e);
        v[i] = *reinterpret_cast<const float4*>(s.v + e);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!ok[i]) continue;
      const long e = base + (long)(i * ADAM_THREADS + tid) * 4;
      adam_one(p[i].x, g[i].x, m[i].x, v[i].x, h);
      adam_one(p[i].y, g[i].y, m[i].y, v[i].y, h);
      adam_one(p[i].z, g[i].z, m[i].z, v[i].z, h);
      adam_one(p[i].w, g[i].w, m[i].w, v[i].w, h);
      *reinterpret_cast<float4*>(s.p + e) = p[i];
      *reinterpret_cast<float4*>(s.m + e) = m[i];
      *reinterpret_cast<float4*>(s.v + e) = v[i];
      if (s.p16) {
        uint2 o;
        o.x = pack_bf16(p[i].x, p[i].y);
        o.y = pack_bf16(p[i].z, p[i].w);
        *reinterpret_cast<uint2*>(s.p16 + e) = o;
      }
    }
    // the last n % 4 elements of the tensor: one thread of its last block
    const long tail = s.n & ~3L;
    if (tid == 0 && tail >= base && tail < base + ADAM_CHUNK) {
      for (long e = tail; e < s.n; ++e) {
        float pp = s.p[e], mm = s.m[e], vv = s.v[e];
        adam_one(pp, s.g[e], mm, vv, h);
        s.p[e] = pp; s.m[e] = mm; s.v[e] = vv;
        if (s.p16) s.p16[e] = f32_to_bf16(pp);
      }
    }
  } else {
    for (int i = tid; i < ADAM_CHUNK; i += ADAM_THREADS) {
      const long e = base + i;
      if (e >= s.n) break;
      float pp = s.p[e], mm = s.m[e], vv = s.v[e];
      adam_one(pp, s.g[e], mm, vv, h);
      s.p[e] = pp; s.m[e] = mm; s.v[e] = vv;
      if (s.p16) s.p16[e] = f32_to_bf16(pp);
    }
  }
}

}  // namespace mopoe

using namespace mopoe;

extern "C" int mopoe_adam_step(const mopoe_adam_seg* segs, int32_t nseg, float* step, const float* lr_dev, double lr,
                               double beta1, double beta2, double eps, float* coef, void* stream) {
  if ((!segs && nseg != 0) || nseg < 0 || !coef) { set_error("adam_step: bad arguments"); return MOPOE_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (step) {   // (step == NULL: a further part of a step whose counter and coefficients an earlier call has set)
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, st, step, lr_dev, lr, beta1, beta2, coef);
    if (int rc = check_launch("adam_prep")) return rc;
  }
  int32_t i = 0;
  while (i < nseg) {
    AdamPack pack;
    int n = 0;
    long blocks = 0;
    for (; i < nseg && n < ADAM_SEGS_PER_LAUNCH; ++i) {
      const mopoe_adam_seg& s = segs[i];
      if (!s.g || s.n == 0) continue;                 // no gradient: the tensor and its state stay as they are
      if (!s.p || !s.m || !s.v || s.n < 0) { set_error("adam_step: record %d is incomplete", (int)i); return MOPOE_ERR_ARG; }
      const long nb = (s.n + ADAM_CHUNK - 1) / ADAM_CHUNK;
      if (nb > 0x7fffffffL) { set_error("adam_step: record %d has more than 2^43 elements", (int)i); return MOPOE_ERR_ARG; }
      if (blocks + nb > 0x7fffffffL) break;           // (a launch's grid: next launch takes the rest)
      pack.seg[n] = s;
      pack.chunk_start[n] = (int)blocks;
      blocks += nb;
      ++n;
    }
    for (int k = n; k < ADAM_SEGS_PER_LAUNCH; ++k) { pack.seg[k] = mopoe_adam_seg{}; pack.chunk_start[k] = 0x7fffffff; }
    if (n == 0) continue;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(ADAM_THREADS), 0, st, pack, n, coef, beta1, beta2, eps);
    if (int rc = check_launch("adam_step")) return rc;
  }
  return MOPOE_OK;
}
