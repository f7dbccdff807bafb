// Likelihood reductions and the text-side row kernels.
//   Laplace NLL (image modalities), row-wise log-softmax over the vocabulary, token NLL gather
//   (OneHotCategorical without the one-hot), embedding gather / scatter-add.
// All are HBM-bound streaming kernels: 16-byte loads where alignment allows, wavefront-shuffle
// reductions, one double atomic per block; scalar results are finalised by the last block.
#include "common.hpp"

namespace mopoe {

// finalise a scalar reduction: ws[0] = running double sum, ws[1] = arrival counter (both left zero)
__device__ __forceinline__ void finish_scalar(double block_sum, double* ws, int nblocks, float scale, float* out) {
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    atomic_add_f64(ws, block_sum);
    __threadfence();
    const unsigned prev = atomicAdd(reinterpret_cast<unsigned*>(ws + 1), 1u);
    is_last = (prev == (unsigned)(nblocks - 1));
  }
  __syncthreads();
  if (is_last && threadIdx.x == 0) {
    __threadfence();
    const double s = __hip_atomic_load(ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[0] = (float)(s * (double)scale);
    __hip_atomic_store(ws, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned*>(ws + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ double block_sum_256(float v) {
  __shared__ double part[4];
  const double s = wave_sum_d((double)v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  return part[0] + part[1] + part[2] + part[3];
}

// ---- Laplace ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void laplace_nll_fwd_kernel(const float* xh, const float* x, long n, float inv_scale,
                                                            float log2b, float inv_norm, float* out, double* ws,
                                                            int nblocks, int vec) {
  float acc = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const long n4 = n >> 2;
    for (; i < n4; i += stride) {
      const float4 a = reinterpret_cast<const float4*>(xh)[i], b = reinterpret_cast<const float4*>(x)[i];
      acc += fabsf(b.x - a.x) + fabsf(b.y - a.y) + fabsf(b.z - a.z) + fabsf(b.w - a.w);
    }
    // tail (n % 4 elements)
    const long t = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) acc += fabsf(x[t] - xh[t]);
  } else {
    for (; i < n; i += stride) acc += fabsf(x[i] - xh[i]);
  }
  const double bs = block_sum_256(acc);
  // sum over elements of (log(2b) + |x - xh|/b) / norm
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    // the constant term is added once, exactly
    atomic_add_f64(ws, (double)n * (double)log2b / (double)inv_scale);
  }
  finish_scalar(bs, ws, nblocks, inv_scale * inv_norm, out);
}

__global__ __launch_bounds__(256) void laplace_nll_bwd_kernel(const float* xh, const float* x, const float* g, long n,
                                                            float coef, float* dxh) {
  const float c = g[0] * coef;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = xh[i] - x[i];
    dxh[i] = d > 0.f ? c : (d < 0.f ? -c : 0.f);
  }
}

template <typename T> __device__ __forceinline__ T to_store(float f);
template <> __device__ __forceinline__ float to_store<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t to_store<bf16_t>(float f) { return f32_to_bf16(f); }
__device__ __forceinline__ float from_store(float f) { return f; }
__device__ __forceinline__ float from_store(bf16_t h) { return bf16_to_f32(h); }

// ---- dense categorical NLL (char text encoding: the target is a [B, L, F] one-hot / dense tensor) ----------------------
// out = -sum(target * logp) / norm;  dlogp = -g / norm * target
__global__ __launch_bounds__(256) void dense_nll_fwd_kernel(const float* logp, const float* tgt, long n, float inv_norm,
                                                          float* out, double* ws, int nblocks) {
  float acc = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float t = tgt[i];
    if (t != 0.f) acc = fmaf(t, logp[i], acc);     // (a one-hot zero must not meet a -inf log-probability)
  }
  finish_scalar(block_sum_256(acc), ws, nblocks, -inv_norm, out);
}

__global__ __launch_bounds__(256) void dense_nll_bwd_kernel(const float* tgt, const float* g, long n, float inv_norm, float* dlogp) {
  const float c = -g[0] * inv_norm;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dlogp[i] = c * tgt[i];
}

// ---- log-softmax over rows of [rows][V]: one 256-thread block per row, row kept in registers --------
// NPT = elements per thread (compile-time so the row really stays in VGPRs); V <= 256 * NPT.
template <int NPT>
__global__ __launch_bounds__(256) void logsoftmax_fwd_kernel(const float* x, float* y, int V) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const float* xr = x + row * V;
  float* yr = y + row * V;
  float v[NPT];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int c = threadIdx.x + k * 256;
    v[k] = c < V ? xr[c] : -INFINITY;
    mx = fmaxf(mx, v[k]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < NPT; ++k) se += expf(v[k] - mx);  // exp(-inf) = 0 for the padding lanes
  se = wave_sum(se);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = se;
  __syncthreads();
  const float lse = mx + logf(red[0] + red[1] + red[2] + red[3]);
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int c = threadIdx.x + k * 256;
    if (c < V) yr[c] = v[k] - lse;
  }
}

template <int NPT, typename TO>
__global__ __launch_bounds__(256) void logsoftmax_bwd_kernel(const float* dy, const float* y, TO* dx, int V) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const float* dr = dy + row * V;
  const float* yr = y + row * V;
  TO* xr = dx + row * V;
  float v[NPT];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int c = threadIdx.x + k * 256;
    v[k] = c < V ? dr[c] : 0.f;
    s += v[k];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int c = threadIdx.x + k * 256;
    if (c < V) xr[c] = to_store<TO>(v[k] - expf(yr[c]) * s);
  }
}

// ---- gradient of the logits for the token NLL, in one pass ---------------------------------------------------------------
// loss = -sum_r logp[r, id_r] / norm with upstream gradient g:  dlogits[r, v] = (g / norm) * (exp(logp[r, v]) - [v == id_r])
// (= log-softmax backward of the one-hot gradient -g / norm, without ever writing that [rows, V] tensor or its memset)
template <int NPT, typename TO>
__global__ __launch_bounds__(256) void token_softmax_grad_kernel(const float* logp, const float* ids, const float* g, int V,
                                                               float inv_norm, TO* dx) {
  const long row = blockIdx.x;
  const float c = g[0] * inv_norm;
  int t = (int)ids[row];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  const float* yr = logp + row * V;
  TO* xr = dx + row * V;
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int col = threadIdx.x + k * 256;
    if (col < V) xr[col] = to_store<TO>(c * (expf(yr[col]) - (col == t ? 1.f : 0.f)));
  }
}

// ---- vocabulary head without a materialised log-softmax (round 4) ---------------------------------------------------------
// The head GEMM writes the LOGITS once, in the family's storage type (bf16 / fp32).  Training needs two numbers per row of
// them -- logsumexp and the target's logit -- and, in the backward, softmax - onehot: the [rows, V] fp32 log-probability
// tensor (461 MB at config #3, written and re-read three times) is never made.  Rows are walked in 16-byte vectors
// (V % EV == 0, EV = 16 / sizeof(T)); one 256-thread block per row, any V: each thread keeps an online (max, sum).
template <typename T> struct RowVec;
template <> struct RowVec<float> {
  static constexpr int EV = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct RowVec<bf16_t> {
  static constexpr int EV = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f32_to_bf16(v[2 * i]) | ((unsigned)f32_to_bf16(v[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// lse[row] = log sum_v exp(x[row, v])   (columns padded with a very negative bias contribute exp(..) = 0)
template <typename T>
__global__ __launch_bounds__(256) void lse_rows_kernel(const T* x, float* lse, int V) {
  constexpr int EV = RowVec<T>::EV;
  __shared__ float red_m[4], red_s[4];
  const long row = blockIdx.x;
  const T* xr = x + row * V;
  float m = -INFINITY, s = 0.f;
  for (int c = threadIdx.x * EV; c < V; c += 256 * EV) {
    float v[EV];
    RowVec<T>::load(xr + c, v);
    float vm = v[0];
#pragma unroll
    for (int e = 1; e < EV; ++e) vm = fmaxf(vm, v[e]);
    const float mn = fmaxf(m, vm);
    float add = 0.f;
#pragma unroll
    for (int e = 0; e < EV; ++e) add += __expf(v[e] - mn);
    s = s * __expf(m - mn) + add;      // (m = -inf on the first vector: exp(-inf) = 0, s = 0)
    m = mn;
  }
  // combine the threads' (m, s): wave, then block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float mo = __shfl_xor(m, o, 64), so = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, mo);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (mo == -INFINITY ? 0.f : so * __expf(mo - mn));
    m = mn;
  }
  if ((threadIdx.x & 63) == 0) { red_m[threadIdx.x >> 6] = m; red_s[threadIdx.x >> 6] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float ss = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) ss += red_m[w] == -INFINITY ? 0.f : red_s[w] * expf(red_m[w] - mm);
    lse[row] = mm + logf(ss);
  }
}

// out = sum_r (lse[r] - x[r, id_r]) / norm   (= -sum_r log p(id_r) / norm)
template <typename T>
__global__ __launch_bounds__(256) void token_nll_logits_fwd_kernel(const T* x, const float* lse, const float* ids, long rows, int V,
                                                                 float inv_norm, float* out, double* ws, int nblocks) {
  float acc = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
    int t = (int)ids[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    acc += lse[r] - from_store(x[r * V + t]);
  }
  finish_scalar(block_sum_256(acc), ws, nblocks, inv_norm, out);
}

// dx[r, v] = g / norm * (exp(x[r, v] - lse[r]) - [v == id_r]);  dx may alias x (each vector is read, then written, by one thread)
template <typename T>
__global__ __launch_bounds__(256) void token_softmax_grad_logits_kernel(const T* x, const float* lse, const float* ids, const float* g,
                                                                      int V, float inv_norm, T* dx) {
  constexpr int EV = RowVec<T>::EV;
  const long row = blockIdx.x;
  const float c = g[0] * inv_norm, l = lse[row];
  int t = (int)ids[row];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  const T* xr = x + row * V;
  T* dr = dx + row * V;
  for (int col = threadIdx.x * EV; col < V; col += 256 * EV) {
    float v[EV];
    RowVec<T>::load(xr + col, v);
#pragma unroll
    for (int e = 0; e < EV; ++e) v[e] = c * (__expf(v[e] - l) - (col + e == t ? 1.f : 0.f));
    RowVec<T>::store(dr + col, v);
  }
}

// ---- token NLL -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void token_nll_fwd_kernel(const float* logp, const float* ids, long rows, int V,
                                                          float inv_norm, float* out, double* ws, int nblocks) {
  float acc = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
    int t = (int)ids[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    acc += logp[r * V + t];
  }
  const double bs = block_sum_256(acc);
  finish_scalar(bs, ws, nblocks, -inv_norm, out);
}

__global__ __launch_bounds__(256) void token_nll_bwd_kernel(const float* ids, const float* g, long rows, int V,
                                                          float inv_norm, float* dlogp) {
  const float c = -g[0] * inv_norm;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
    int t = (int)ids[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    dlogp[r * V + t] = c;
  }
}

// ---- embedding -------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const float* ids, const float* table, T* out, long rows,
                                                          int V, int D) {
  // one wave per row chunk: lanes walk the D channels
  const long total = rows * D;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long r = i / D;
    const int d = (int)(i - r * D);
    int t = (int)ids[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    out[i] = to_store<T>(table[(long)t * D + d]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* ids, const T* gout, float* dtable, long rows,
                                                          int V, int D, int padding_idx) {
  const long total = rows * D;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long r = i / D;
    const int d = (int)(i - r * D);
    int t = (int)ids[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    if (t != padding_idx) unsafeAtomicAdd(dtable + (long)t * D + d, from_store(gout[i]));
  }
}

// ---- per-row log-probabilities (importance-sampled likelihood estimator) -----------------------------------------
// out[r] = sum_j ( -log(2b) - |x[(r % B) P + j] - xhat[r P + j]| / b ): one block per row; the K repeats of the
// target are an index (r % B), not a materialised [K,B,...] copy
__global__ __launch_bounds__(256) void laplace_logprob_rows_kernel(const float* xh, const float* x, long P, long B, float inv_scale,
                                                                 float log2b, float* out, int vec) {
  const long r = blockIdx.x;
  const float* a = xh + r * P;
  const float* b = x + (r % B) * P;
  float acc = 0.f;
  if (vec) {
    const long n4 = P >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
      acc += fabsf(v.x - u.x) + fabsf(v.y - u.y) + fabsf(v.z - u.z) + fabsf(v.w - u.w);
    }
  } else {
    for (long i = threadIdx.x; i < P; i += 256) acc += fabsf(b[i] - a[i]);
  }
  const double s = block_sum_256(acc);
  if (threadIdx.x == 0) out[r] = (float)(-(double)P * (double)log2b - s * (double)inv_scale);
}

// out[r] = sum_l logp[(r L + l) V + ids[(r % B) L + l]]
__global__ __launch_bounds__(256) void token_logprob_rows_kernel(const float* logp, const float* ids, int L, int V, long B, float* out) {
  const long r = blockIdx.x;
  float acc = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) {
    int id = (int)ids[(r % B) * L + l];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    acc += logp[(r * L + l) * (long)V + id];
  }
  const double s = block_sum_256(acc);
  if (threadIdx.x == 0) out[r] = (float)s;
}

// out[r] = sum_i target[(r % B) P + i] * logp[r P + i]   (dense / one-hot char targets, P = L * num_features)
__global__ __launch_bounds__(256) void dense_logprob_rows_kernel(const float* logp, const float* tgt, long P, long B, float* out) {
  const long r = blockIdx.x;
  const float* a = logp + r * P;
  const float* b = tgt + (r % B) * P;
  float acc = 0.f;
  for (long i = threadIdx.x; i < P; i += 256) acc += a[i] * b[i];
  const double s = block_sum_256(acc);
  if (threadIdx.x == 0) out[r] = (float)s;
}

// grid of a kernel that ends in finish_scalar(): every block adds to ONE double and bumps ONE counter -- same-address atomics
// serialise at the memory side at ~18 ns each, so the grid is capped at one block per CU (round 4: laplace_nll_fwd ran 1024
// blocks = ~35 us of atomic tail behind 8 us of streaming, on the critical path between the step's forward and backward)
static int reduce_grid(long n, int per_thread) {
  long blocks = (n + 256L * per_thread - 1) / (256L * per_thread);
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  return (int)blocks;
}

static int stream_grid(long n, int per_thread) {
  long blocks = (n + 256L * per_thread - 1) / (256L * per_thread);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  return (int)blocks;
}

}  // namespace mopoe

using namespace mopoe;

extern "C" int mopoe_laplace_nll_fwd(const float* x_hat, const float* x, int64_t n, float scale, float norm, float* out,
                                     double* ws, void* stream) {
  if (!x_hat || !x || !out || !ws || n <= 0 || scale <= 0.f || norm <= 0.f) { set_error("laplace_nll_fwd: bad arguments"); return MOPOE_ERR_ARG; }
  const int vec = ((reinterpret_cast<uintptr_t>(x_hat) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
  const int nb = reduce_grid(n, 16);
  hipLaunchKernelGGL(laplace_nll_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x_hat, x, (long)n, 1.0f / scale,
                     logf(2.0f * scale), 1.0f / norm, out, ws, nb, vec);
  return check_launch("laplace_nll_fwd");
}

extern "C" int mopoe_laplace_nll_bwd(const float* x_hat, const float* x, const float* g, int64_t n, float scale,
                                     float norm, float* dx_hat, void* stream) {
  if (!x_hat || !x || !g || !dx_hat || n <= 0) { set_error("laplace_nll_bwd: bad arguments"); return MOPOE_ERR_ARG; }
  hipLaunchKernelGGL(laplace_nll_bwd_kernel, dim3(stream_grid(n, 8)), dim3(256), 0, (hipStream_t)stream, x_hat, x, g,
                     (long)n, 1.0f / (scale * norm), dx_hat);
  return check_launch("laplace_nll_bwd");
}

extern "C" int mopoe_dense_nll_fwd(const float* logp, const float* target, int64_t n, float norm, float* out, double* ws,
                                   void* stream) {
  if (!logp || !target || !out || !ws || n <= 0 || norm <= 0.f) { set_error("dense_nll_fwd: bad arguments"); return MOPOE_ERR_ARG; }
  const int nb = reduce_grid(n, 16);
  hipLaunchKernelGGL(dense_nll_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, logp, target, (long)n, 1.0f / norm, out, ws, nb);
  return check_launch("dense_nll_fwd");
}

extern "C" int mopoe_dense_nll_bwd(const float* target, const float* g, int64_t n, float norm, float* dlogp, void* stream) {
  if (!target || !g || !dlogp || n <= 0 || norm <= 0.f) { set_error("dense_nll_bwd: bad arguments"); return MOPOE_ERR_ARG; }
  hipLaunchKernelGGL(dense_nll_bwd_kernel, dim3(stream_grid(n, 8)), dim3(256), 0, (hipStream_t)stream, target, g, (long)n,
                     1.0f / norm, dlogp);
  return check_launch("dense_nll_bwd");
}

extern "C" int mopoe_logsoftmax_fwd(const float* x, float* y, int64_t rows, int32_t V, void* stream) {
  if (!x || !y || rows <= 0 || V <= 0 || V > 256 * 32) { set_error("logsoftmax_fwd: bad arguments (V <= 8192)"); return MOPOE_ERR_ARG; }
  const dim3 grid((unsigned)rows), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (V <= 256 * 4) hipLaunchKernelGGL(logsoftmax_fwd_kernel<4>, grid, blk, 0, st, x, y, V);
  else if (V <= 256 * 16) hipLaunchKernelGGL(logsoftmax_fwd_kernel<16>, grid, blk, 0, st, x, y, V);
  else hipLaunchKernelGGL(logsoftmax_fwd_kernel<32>, grid, blk, 0, st, x, y, V);
  return check_launch("logsoftmax_fwd");
}

extern "C" int mopoe_logsoftmax_bwd(const float* dy, const float* y, float* dx, int64_t rows, int32_t V, void* stream) {
  if (!dy || !y || !dx || rows <= 0 || V <= 0 || V > 256 * 32) { set_error("logsoftmax_bwd: bad arguments (V <= 8192)"); return MOPOE_ERR_ARG; }
  const dim3 grid((unsigned)rows), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (V <= 256 * 4) hipLaunchKernelGGL((logsoftmax_bwd_kernel<4, float>), grid, blk, 0, st, dy, y, dx, V);
  else if (V <= 256 * 16) hipLaunchKernelGGL((logsoftmax_bwd_kernel<16, float>), grid, blk, 0, st, dy, y, dx, V);
  else hipLaunchKernelGGL((logsoftmax_bwd_kernel<32, float>), grid, blk, 0, st, dy, y, dx, V);
  return check_launch("logsoftmax_bwd");
}

extern "C" int mopoe_logsoftmax_bwd_bf16out(const float* dy, const float* y, uint16_t* dx, int64_t rows, int32_t V, void* stream) {
  if (!dy || !y || !dx || rows <= 0 || V <= 0 || V > 256 * 32) { set_error("logsoftmax_bwd_bf16out: bad arguments (V <= 8192)"); return MOPOE_ERR_ARG; }
  const dim3 grid((unsigned)rows), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (V <= 256 * 4) hipLaunchKernelGGL((logsoftmax_bwd_kernel<4, bf16_t>), grid, blk, 0, st, dy, y, dx, V);
  else if (V <= 256 * 16) hipLaunchKernelGGL((logsoftmax_bwd_kernel<16, bf16_t>), grid, blk, 0, st, dy, y, dx, V);
  else hipLaunchKernelGGL((logsoftmax_bwd_kernel<32, bf16_t>), grid, blk, 0, st, dy, y, dx, V);
  return check_launch("logsoftmax_bwd_bf16out");
}

extern "C" int mopoe_token_softmax_grad(const float* logp, const float* ids, const float* g, int64_t rows, int32_t V, float norm,
                                        void* dx, int32_t dx_is_bf16, void* stream) {
  if (!logp || !ids || !g || !dx || rows <= 0 || V <= 0 || V > 256 * 32 || norm <= 0.f || rows > 0x7fffffffL) {
    set_error("token_softmax_grad: bad arguments (V <= 8192)"); return MOPOE_ERR_ARG;
  }
  const dim3 grid((unsigned)rows), blk(256);
  hipStream_t st = (hipStream_t)stream;
  const float inv = 1.0f / norm;
#define MOPOE_TSG(NPT_)                                                                                                          \
  do {                                                                                                                           \
    if (dx_is_bf16) hipLaunchKernelGGL((token_softmax_grad_kernel<NPT_, bf16_t>), grid, blk, 0, st, logp, ids, g, V, inv, (bf16_t*)dx); \
    else hipLaunchKernelGGL((token_softmax_grad_kernel<NPT_, float>), grid, blk, 0, st, logp, ids, g, V, inv, (float*)dx);          \
  } while (0)
  if (V <= 256 * 4) MOPOE_TSG(4);
  else if (V <= 256 * 16) MOPOE_TSG(16);
  else MOPOE_TSG(32);
#undef MOPOE_TSG
  return check_launch("token_softmax_grad");
}

extern "C" int mopoe_lse_rows(const void* logits, int32_t is_bf16, int64_t rows, int32_t V, float* lse, void* stream) {
  const int ev = is_bf16 ? 8 : 4;
  if (!logits || !lse || rows <= 0 || rows > 0x7fffffffL || V <= 0 || V % ev != 0 || ((uintptr_t)logits & 15)) {
    set_error("lse_rows: bad arguments (V must be a multiple of %d, rows 16-byte aligned)", ev); return MOPOE_ERR_ARG;
  }
  const dim3 grid((unsigned)rows), blk(256);
  if (is_bf16) hipLaunchKernelGGL(lse_rows_kernel<bf16_t>, grid, blk, 0, (hipStream_t)stream, (const bf16_t*)logits, lse, V);
  else hipLaunchKernelGGL(lse_rows_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const float*)logits, lse, V);
  return check_launch("lse_rows");
}

extern "C" int mopoe_token_nll_logits_fwd(const void* logits, int32_t is_bf16, const float* lse, const float* ids, int64_t rows,
                                          int32_t V, float norm, float* out, double* ws, void* stream) {
  if (!logits || !lse || !ids || !out || !ws || rows <= 0 || V <= 0 || norm <= 0.f) { set_error("token_nll_logits_fwd: bad arguments"); return MOPOE_ERR_ARG; }
  const int nb = reduce_grid(rows, 1);
  if (is_bf16) hipLaunchKernelGGL(token_nll_logits_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits,
                                  lse, ids, (long)rows, V, 1.0f / norm, out, ws, nb);
  else hipLaunchKernelGGL(token_nll_logits_fwd_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)logits, lse, ids,
                          (long)rows, V, 1.0f / norm, out, ws, nb);
  return check_launch("token_nll_logits_fwd");
}

extern "C" int mopoe_token_softmax_grad_logits(const void* logits, int32_t is_bf16, const float* lse, const float* ids, const float* g,
                                               int64_t rows, int32_t V, float norm, void* dx, void* stream) {
  const int ev = is_bf16 ? 8 : 4;
  if (!logits || !lse || !ids || !g || !dx || rows <= 0 || rows > 0x7fffffffL || V <= 0 || V % ev != 0 || norm <= 0.f ||
      ((uintptr_t)logits & 15) || ((uintptr_t)dx & 15)) {
    set_error("token_softmax_grad_logits: bad arguments (V must be a multiple of %d, rows 16-byte aligned)", ev); return MOPOE_ERR_ARG;
  }
  const dim3 grid((unsigned)rows), blk(256);
  if (is_bf16) hipLaunchKernelGGL(token_softmax_grad_logits_kernel<bf16_t>, grid, blk, 0, (hipStream_t)stream, (const bf16_t*)logits, lse,
                                  ids, g, V, 1.0f / norm, (bf16_t*)dx);
  else hipLaunchKernelGGL(token_softmax_grad_logits_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const float*)logits, lse, ids, g, V,
                          1.0f / norm, (float*)dx);
  return check_launch("token_softmax_grad_logits");
}

extern "C" int mopoe_token_nll_fwd(const float* logp, const float* ids, int64_t rows, int32_t V, float norm, float* out,
                                   double* ws, void* stream) {
  if (!logp || !ids || !out || !ws || rows <= 0 || V <= 0 || norm <= 0.f) { set_error("token_nll_fwd: bad arguments"); return MOPOE_ERR_ARG; }
  const int nb = reduce_grid(rows, 1);
  hipLaunchKernelGGL(token_nll_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, logp, ids, (long)rows, V,
                     1.0f / norm, out, ws, nb);
  return check_launch("token_nll_fwd");
}

extern "C" int mopoe_token_nll_bwd(const float* ids, const float* g, int64_t rows, int32_t V, float norm, float* dlogp,
                                   void* stream) {
  if (!ids || !g || !dlogp || rows <= 0 || V <= 0) { set_error("token_nll_bwd: bad arguments"); return MOPOE_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dlogp, 0, sizeof(float) * (size_t)rows * V, st) != hipSuccess) { set_error("token_nll_bwd memset failed"); return MOPOE_ERR_LAUNCH; }
  hipLaunchKernelGGL(token_nll_bwd_kernel, dim3(stream_grid(rows, 1)), dim3(256), 0, st, ids, g, (long)rows, V, 1.0f / norm, dlogp);
  return check_launch("token_nll_bwd");
}

extern "C" int mopoe_laplace_logprob_rows(const float* x_hat, const float* x, int64_t rows, int64_t per_row, int64_t target_rows,
                                          float scale, float* out, void* stream) {
  if (!x_hat || !x || !out || rows <= 0 || per_row <= 0 || target_rows <= 0 || scale <= 0.f || rows > 0x7fffffffL) {
    set_error("laplace_logprob_rows: bad arguments"); return MOPOE_ERR_ARG;
  }
  const int vec = (per_row % 4 == 0) && ((reinterpret_cast<uintptr_t>(x_hat) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
  hipLaunchKernelGGL(laplace_logprob_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x_hat, x, (long)per_row,
                     (long)target_rows, 1.0f / scale, logf(2.0f * scale), out, vec);
  return check_launch("laplace_logprob_rows");
}

extern "C" int mopoe_dense_logprob_rows(const float* logp, const float* target, int64_t rows, int64_t per_row, int64_t target_rows,
                                        float* out, void* stream) {
  if (!logp || !target || !out || rows <= 0 || per_row <= 0 || target_rows <= 0 || rows > 0x7fffffffL) {
    set_error("dense_logprob_rows: bad arguments"); return MOPOE_ERR_ARG;
  }
  hipLaunchKernelGGL(dense_logprob_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logp, target, (long)per_row,
                     (long)target_rows, out);
  return check_launch("dense_logprob_rows");
}

extern "C" int mopoe_token_logprob_rows(const float* logp, const float* ids, int64_t rows, int32_t L, int32_t V, int64_t target_rows,
                                        float* out, void* stream) {
  if (!logp || !ids || !out || rows <= 0 || L <= 0 || V <= 0 || target_rows <= 0 || rows > 0x7fffffffL) {
    set_error("token_logprob_rows: bad arguments"); return MOPOE_ERR_ARG;
  }
  hipLaunchKernelGGL(token_logprob_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logp, ids, L, V,
                     (long)target_rows, out);
  return check_launch("token_logprob_rows");
}

extern "C" int mopoe_embedding_fwd(const float* ids, const float* table, float* out, int64_t rows, int32_t V, int32_t D,
                                   void* stream) {
  if (!ids || !table || !out || rows <= 0 || V <= 0 || D <= 0) { set_error("embedding_fwd: bad arguments"); return MOPOE_ERR_ARG; }
  hipLaunchKernelGGL(embedding_fwd_kernel<float>, dim3(stream_grid(rows * D, 4)), dim3(256), 0, (hipStream_t)stream, ids, table,
                     out, (long)rows, V, D);
  return check_launch("embedding_fwd");
}

extern "C" int mopoe_embedding_fwd_bf16(const float* ids, const float* table, uint16_t* out, int64_t rows, int32_t V,
                                        int32_t D, void* stream) {
  if (!ids || !table || !out || rows <= 0 || V <= 0 || D <= 0) { set_error("embedding_fwd_bf16: bad arguments"); return MOPOE_ERR_ARG; }
  hipLaunchKernelGGL(embedding_fwd_kernel<bf16_t>, dim3(stream_grid(rows * D, 4)), dim3(256), 0, (hipStream_t)stream, ids,
                     table, out, (long)rows, V, D);
  return check_launch("embedding_fwd_bf16");
}

extern "C" int mopoe_embedding_bwd(const float* ids, const float* gout, float* dtable, int64_t rows, int32_t V, int32_t D,
                                   int32_t padding_idx, void* stream) {
  if (!ids || !gout || !dtable || rows <= 0 || V <= 0 || D <= 0) { set_error("embedding_bwd: bad arguments"); return MOPOE_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dtable, 0, sizeof(float) * (size_t)V * D, st) != hipSuccess) { set_error("embedding_bwd memset failed"); return MOPOE_ERR_LAUNCH; }
  hipLaunchKernelGGL(embedding_bwd_kernel<float>, dim3(stream_grid(rows * D, 4)), dim3(256), 0, st, ids, gout, dtable, (long)rows,
                     V, D, padding_idx);
  return check_launch("embedding_bwd");
}

extern "C" int mopoe_embedding_bwd_bf16(const float* ids, const uint16_t* gout, float* dtable, int64_t rows, int32_t V,
                                        int32_t D, int32_t padding_idx, void* stream) {
  if (!ids || !gout || !dtable || rows <= 0 || V <= 0 || D <= 0) { set_error("embedding_bwd_bf16: bad arguments"); return MOPOE_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dtable, 0, sizeof(float) * (size_t)V * D, st) != hipSuccess) { set_error("embedding_bwd memset failed"); return MOPOE_ERR_LAUNCH; }
  hipLaunchKernelGGL(embedding_bwd_kernel<bf16_t>, dim3(stream_grid(rows * D, 4)), dim3(256), 0, st, ids, gout, dtable,
                     (long)rows, V, D, padding_idx);
  return check_launch("embedding_bwd_bf16");
}
