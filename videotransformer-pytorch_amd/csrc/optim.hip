// optim.hip -- multi-tensor optimizer step and per-parameter gradient norms / clipping.
//
// The step right after backward.  Reference:
//   model_trainer.py:155-170  clip_gradients: per PARAMETER  n_i = ||g_i||_2; if clip / (n_i + 1e-6) < 1 the
//                             gradient is scaled by it; returns ||(n_0, n_1, ...)||_2  (~250 torch.norm launches)
//   optimizer.py:31-38        torch.optim.SGD(momentum=0.9, nesterov=True, weight_decay) or AdamW(betas=(0.9,0.999))
//                             over two parameter groups (no-decay / decay), optimizer.py:42-62
//   model_trainer.py:147-151  the decay group's weight_decay follows a cosine schedule (read per step)
// Here: every parameter of the model is one entry of a device table {param, grad, state pointers, numel, lr,
// weight decay}; three launches per step whatever the number of parameters:
//   mt_sqnorm   : sum of squares of every 4096-element chunk of every gradient     (reads G)
//   mt_norms    : per-parameter norms (fixed summation order) + the norm of norms  (tiny)
//   mt_sgd/adamw: clip coefficient from the norm, update, state update             (reads G,P,S; writes P,S)
// HBM-bound: SGD moves 5 x 4 bytes per parameter element (+4 for the norm pass), AdamW 7 x 4.
#include <math.h>
#include "common.h"

namespace vtx {

constexpr int MT_CHUNK = 4096;     // elements per workgroup
constexpr int MT_THREADS = 256;

// block -> (tensor, chunk within the tensor): chunk_start is the exclusive prefix sum of chunks per tensor
__device__ inline int mt_find(const int* __restrict__ chunk_start, int n_tensors, int b) {
  int lo = 0, hi = n_tensors;                  // largest t with chunk_start[t] <= b
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_start[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ inline float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float a = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < MT_THREADS / 64; ++w) a += red[w];
  }
  return a;                                    // valid in thread 0
}

__global__ __launch_bounds__(MT_THREADS) void mt_sqnorm_kernel(const vtx_mt_tensor* __restrict__ tab, const int* __restrict__ chunk_start,
                                                               int n_tensors, float* __restrict__ partial) {
  __shared__ float red[MT_THREADS / 64];
  const int b = blockIdx.x;
  const int t = mt_find(chunk_start, n_tensors, b);
  const long n = tab[t].n;
  const long base = (long)(b - chunk_start[t]) * MT_CHUNK;
  const float* g = reinterpret_cast<const float*>(tab[t].g) + base;
  const long left = n - base;
  float a = 0.f;
  if (left >= MT_CHUNK && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
#pragma unroll
    for (int i = 0; i < MT_CHUNK / (MT_THREADS * 4); ++i) {
      const float4 v = *reinterpret_cast<const float4*>(g + (i * MT_THREADS + threadIdx.x) * 4);
      a += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    for (long i = threadIdx.x; i < left && i < MT_CHUNK; i += MT_THREADS) a += g[i] * g[i];
  }
  const float s = block_sum(a, red);
  if (threadIdx.x == 0) partial[b] = s;
}

// norms[t] = sqrt(sum of the tensor's chunk partials, in chunk order); norms[n_tensors] = ||norms[0..n)||_2
__global__ __launch_bounds__(MT_THREADS) void mt_norms_kernel(const int* __restrict__ chunk_start, int n_tensors,
                                                              const float* __restrict__ partial, float* __restrict__ norms) {
  __shared__ float red[MT_THREADS / 64];
  float sq = 0.f;
  for (int t = threadIdx.x; t < n_tensors; t += MT_THREADS) {
    float a = 0.f;
    for (int c = chunk_start[t]; c < chunk_start[t + 1]; ++c) a += partial[c];
    norms[t] = sqrtf(a);
    sq += a;
  }
  const float s = block_sum(sq, red);
  if (threadIdx.x == 0) norms[n_tensors] = sqrtf(s);
}

struct MtHyper {
  float clip;          // per-parameter clip threshold, <= 0: no clipping
  float momentum;      // SGD
  int nesterov;        // SGD
  int first_step;      // SGD: momentum buffer := gradient
  float beta1, beta2, eps, bc1, bc2;   // AdamW: bias corrections 1 - beta^step
};

template <bool ADAMW>
__global__ __launch_bounds__(MT_THREADS) void mt_step_kernel(const vtx_mt_tensor* __restrict__ tab, const int* __restrict__ chunk_start,
                                                             int n_tensors, const float* __restrict__ norms, MtHyper h) {
  const int b = blockIdx.x;
  const int t = mt_find(chunk_start, n_tensors, b);
  const vtx_mt_tensor e = tab[t];
  const long base = (long)(b - chunk_start[t]) * MT_CHUNK;
  float coef = 1.f;
  if (h.clip > 0.f) {                           // model_trainer.py:165-168
    const float c = h.clip / (norms[t] + 1e-6f);
    if (c < 1.f) coef = c;
  }
  float* p = reinterpret_cast<float*>(e.p) + base;
  const float* g = reinterpret_cast<const float*>(e.g) + base;
  float* s1 = reinterpret_cast<float*>(e.s1) + base;
  float* s2 = ADAMW ? reinterpret_cast<float*>(e.s2) + base : nullptr;
  const long left = e.n - base;
  const float lr = e.lr, wd = e.wd;
  auto upd = [&](float& pv, float gv, float& m, float& v) {
    gv *= coef;
    if (ADAMW) {                                // torch.optim.AdamW: decoupled decay, then the Adam update
      pv *= 1.f - lr * wd;
      m = h.beta1 * m + (1.f - h.beta1) * gv;
      v = h.beta2 * v + (1.f - h.beta2) * gv * gv;
      const float denom = sqrtf(v) / sqrtf(h.bc2) + h.eps;
      pv -= (lr / h.bc1) * (m / denom);
    } else {                                    // torch.optim.SGD (dampening 0)
      gv += wd * pv;
      m = h.first_step ? gv : h.momentum * m + gv;
      const float d = h.nesterov ? gv + h.momentum * m : m;
      pv -= lr * d;
    }
  };
  const bool vec = left >= MT_CHUNK && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(s1) |
                                         reinterpret_cast<uintptr_t>(s2)) & 15u) == 0;
  if (vec) {
#pragma unroll
    for (int i = 0; i < MT_CHUNK / (MT_THREADS * 4); ++i) {
      const int o = (i * MT_THREADS + threadIdx.x) * 4;
      float4 pv = *reinterpret_cast<float4*>(p + o);
      const float4 gv = *reinterpret_cast<const float4*>(g + o);
      float4 m = h.first_step && !ADAMW ? make_float4(0, 0, 0, 0) : *reinterpret_cast<float4*>(s1 + o);
      float4 v = make_float4(0, 0, 0, 0);
      if (ADAMW) v = *reinterpret_cast<float4*>(s2 + o);
      upd(pv.x, gv.x, m.x, v.x); upd(pv.y, gv.y, m.y, v.y); upd(pv.z, gv.z, m.z, v.z); upd(pv.w, gv.w, m.w, v.w);
      *reinterpret_cast<float4*>(p + o) = pv;
      *reinterpret_cast<float4*>(s1 + o) = m;
      if (ADAMW) *reinterpret_cast<float4*>(s2 + o) = v;
    }
  } else {
    for (long i = threadIdx.x; i < left && i < MT_CHUNK; i += MT_THREADS) {
      float pv = p[i], m = (h.first_step && !ADAMW) ? 0.f : s1[i], v = ADAMW ? s2[i] : 0.f;
      upd(pv, g[i], m, v);
      p[i] = pv;
      s1[i] = m;
      if (ADAMW) s2[i] = v;
    }
  }
}

}  // namespace vtx

using namespace vtx;

extern "C" int vtx_mt_chunks(long numel) { return numel <= 0 ? 0 : (int)((numel + MT_CHUNK - 1) / MT_CHUNK); }

static int mt_check(const vtx_mt_tensor* tab, const int* chunk_start, int n_tensors, int n_chunks, const char* who) {
  VTX_REQUIRE(tab && chunk_start && n_tensors > 0 && n_chunks > 0, VTX_EINVAL, "%s: empty table", who);
  return VTX_OK;
}

extern "C" int vtx_mt_grad_norms(const vtx_mt_tensor* tab, const int* chunk_start, int n_tensors, int n_chunks,
                                 float* partial, float* norms, void* stream) {
  int rc = mt_check(tab, chunk_start, n_tensors, n_chunks, "mt_grad_norms");
  if (rc) return rc;
  VTX_REQUIRE(partial && norms, VTX_EINVAL, "mt_grad_norms: null output");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(mt_sqnorm_kernel, dim3(n_chunks), dim3(MT_THREADS), 0, st, tab, chunk_start, n_tensors, partial);
  rc = check_launch("mt_sqnorm");
  if (rc) return rc;
  hipLaunchKernelGGL(mt_norms_kernel, dim3(1), dim3(MT_THREADS), 0, st, chunk_start, n_tensors, partial, norms);
  return check_launch("mt_norms");
}

extern "C" int vtx_mt_sgd_step(const vtx_mt_tensor* tab, const int* chunk_start, int n_tensors, int n_chunks, const float* norms,
                               float clip, float momentum, int nesterov, int first_step, void* stream) {
  int rc = mt_check(tab, chunk_start, n_tensors, n_chunks, "mt_sgd_step");
  if (rc) return rc;
  VTX_REQUIRE(clip <= 0.f || norms, VTX_EINVAL, "mt_sgd_step: clipping needs the norms of vtx_mt_grad_norms");
  MtHyper h = {};
  h.clip = clip; h.momentum = momentum; h.nesterov = nesterov; h.first_step = first_step;
  hipLaunchKernelGGL(mt_step_kernel<false>, dim3(n_chunks), dim3(MT_THREADS), 0, as_stream(stream), tab, chunk_start, n_tensors, norms, h);
  return check_launch("mt_sgd_step");
}

extern "C" int vtx_mt_adamw_step(const vtx_mt_tensor* tab, const int* chunk_start, int n_tensors, int n_chunks, const float* norms,
                                 float clip, float beta1, float beta2, float eps, int step, void* stream) {
  int rc = mt_check(tab, chunk_start, n_tensors, n_chunks, "mt_adamw_step");
  if (rc) return rc;
  VTX_REQUIRE(step >= 1, VTX_EINVAL, "mt_adamw_step: step counts from 1");
  VTX_REQUIRE(clip <= 0.f || norms, VTX_EINVAL, "mt_adamw_step: clipping needs the norms of vtx_mt_grad_norms");
  MtHyper h = {};
  h.clip = clip; h.beta1 = beta1; h.beta2 = beta2; h.eps = eps;
  h.bc1 = (float)(1.0 - pow((double)beta1, (double)step));       // as torch.optim.AdamW: python doubles
  h.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(mt_step_kernel<true>, dim3(n_chunks), dim3(MT_THREADS), 0, as_stream(stream), tab, chunk_start, n_tensors, norms, h);
  return check_launch("mt_adamw_step");
}
