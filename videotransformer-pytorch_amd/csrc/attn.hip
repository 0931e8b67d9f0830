// attn.hip -- softmax(q k^T * scale) v per (sequence, head): forward, materialised
// probabilities, and backward.  fp32-exact VALU formulation (one thread per
// query row, keys/values staged through LDS in tiles, chunked online softmax).
//
// Reference: Attention.forward transformer.py:165-177 and the regrouping around
// it: '(b p) t d' :250 (temporal, contiguous T-token sequences), '(b t) p d' +
// per-frame cls replication :352-356 and the scatter back :375 (spatial).  The
// regrouping is done here by row arithmetic (in_row/out_row); no tensor is
// rearranged in memory.
//
// Small sequences (temporal attention, L = T = 8) pack G = 256/L sequences per
// workgroup; long ones (L = 197 spatial, 1569 joint) use one sequence per
// workgroup and 64-key tiles.  Algorithmic bytes per (sequence, head):
// read 3*L*hd, write L*hd elements (+ L fp32 lse).
#include <stdlib.h>
#include "attn_common.h"

namespace vtx {

constexpr int AT_THREADS = 256;    // long sequences: one sequence per workgroup
constexpr int AT_PACK_THREADS = 128;  // short sequences: G = 128/L sequences per workgroup (64 KB of LDS tiles)
constexpr int AT_KT = 64;      // keys (or queries, in the dk/dv pass) per LDS tile
constexpr int AT_CH = 8;       // online-softmax chunk

template <int HD> struct Tile {           // one [rows][HD] fp32 tile per packed sequence, padded
  static constexpr int SEQ_PAD = 4;       // floats between packed sequences (bank spread for broadcast reads)
};

// Cooperative load of `n` rows x HD elements (type T, global) into fp32 LDS rows.
// 8 (bf16) or 16 (fp32) threads per row; rows addressed through rowfn(r).
template <typename T, int HD, typename RowFn>
__device__ inline void load_rows(float* lds, int lds_ld, int n, const T* base, long ld, int col0, RowFn rowfn) {
  constexpr int CH = HD / 8;              // 8-element chunks per row
  for (int id = threadIdx.x; id < n * CH; id += blockDim.x) {
    const int r = id / CH, c = id - r * CH;
    float v[8];
    load8(base + rowfn(r) * ld + col0 + c * 8, v);
    float* d = lds + r * lds_ld + c * 8;
    *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

template <int HD> __device__ inline float dot_lds(const float (&q)[HD], const float* k) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int e = 0; e < HD; e += 4) {
    const float4 kv = *reinterpret_cast<const float4*>(k + e);
    s0 += q[e] * kv.x; s1 += q[e + 1] * kv.y; s2 += q[e + 2] * kv.z; s3 += q[e + 3] * kv.w;
  }
  return (s0 + s1) + (s2 + s3);
}
template <int HD> __device__ inline void axpy_lds(float (&acc)[HD], float a, const float* v) {
#pragma unroll
  for (int e = 0; e < HD; e += 4) {
    const float4 x = *reinterpret_cast<const float4*>(v + e);
    acc[e] += a * x.x; acc[e + 1] += a * x.y; acc[e + 2] += a * x.z; acc[e + 3] += a * x.w;
  }
}
template <typename T, int HD> __device__ inline void load_vec(const T* p, float (&v)[HD]) {
#pragma unroll
  for (int c = 0; c < HD / 8; ++c) {
    float t8[8];
    load8(p + c * 8, t8);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[c * 8 + j] = t8[j];
  }
}
template <typename T, int HD> __device__ inline void store_vec(T* p, const float (&v)[HD]) {
#pragma unroll
  for (int c = 0; c < HD / 8; ++c) {
    float t8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t8[j] = v[c * 8 + j];
    store8(p + c * 8, t8);
  }
}

// Work decomposition shared by all passes: blockIdx.x -> (sequence group, row block), blockIdx.y -> head.
// Thread -> (g, i): packed sequence g of the group and row i (query or key) of that sequence.
struct Who { int s0, g, i, nseq, rows_blk; bool active; };
__device__ inline Who who_am_i(const AttnP& p) {
  Who w;
  if (p.G > 1) {
    w.s0 = blockIdx.x * p.G;
    w.nseq = min(p.G, p.S - w.s0);
    w.g = threadIdx.x / p.L; w.i = threadIdx.x - w.g * p.L;
    w.active = w.g < w.nseq;
    w.rows_blk = 0;
  } else {
    const int nblk = (p.L + blockDim.x - 1) / blockDim.x;
    w.s0 = blockIdx.x / nblk; w.nseq = 1; w.g = 0;
    w.rows_blk = (blockIdx.x - w.s0 * nblk) * blockDim.x;
    w.i = w.rows_blk + threadIdx.x;
    w.active = w.i < p.L;
  }
  return w;
}

// PASS 0: forward (out, lse).  PASS 1: probabilities from a saved lse.
template <typename T, int HD, int PASS>
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(AttnP p, const T* __restrict__ qkv, T* __restrict__ out,
                                                              float* __restrict__ lse, float* __restrict__ probs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int h = blockIdx.y, D = p.H * HD;
  const Who w = who_am_i(p);
  const int KT = p.G > 1 ? p.L : AT_KT;
  const int seq_ld = KT * HD + Tile<HD>::SEQ_PAD;
  float* Ks = sm;
  float* Vs = sm + p.G * seq_ld;
  const int s = w.s0 + w.g;

  float q[HD];
  if (w.active) {
    load_vec<T, HD>(qkv + in_row(p, s, w.i) * p.ld_qkv + h * HD, q);
#pragma unroll
    for (int e = 0; e < HD; ++e) q[e] *= p.scale;
  }
  float acc[HD];
#pragma unroll
  for (int e = 0; e < HD; ++e) acc[e] = 0.f;
  float m = -INFINITY, l = 0.f;
  float my_lse = 0.f;
  if (PASS == 1 && w.active) my_lse = lse[((long)s * p.H + h) * p.L + w.i];

  for (int k0 = 0; k0 < p.L; k0 += KT) {
    const int nk = min(KT, p.L - k0);
    __syncthreads();
    for (int g = 0; g < w.nseq; ++g) {
      const int sg = w.s0 + g;
      load_rows<T, HD>(Ks + g * seq_ld, HD, nk, qkv, p.ld_qkv, D + h * HD, [&](int r) { return in_row(p, sg, k0 + r); });
      if (PASS == 0)
        load_rows<T, HD>(Vs + g * seq_ld, HD, nk, qkv, p.ld_qkv, 2 * D + h * HD, [&](int r) { return in_row(p, sg, k0 + r); });
    }
    __syncthreads();
    if (!w.active) continue;
    const float* Kg = Ks + w.g * seq_ld;
    const float* Vg = Vs + w.g * seq_ld;
    if (PASS == 1) {
      float* pr = probs + (((long)s * p.H + h) * p.L + w.i) * p.L + k0;
      for (int j = 0; j < nk; ++j) pr[j] = __expf(dot_lds<HD>(q, Kg + j * HD) - my_lse);
      continue;
    }
    for (int j0 = 0; j0 < nk; j0 += AT_CH) {
      float sc[AT_CH];
      float cmax = -INFINITY;
#pragma unroll
      for (int j = 0; j < AT_CH; ++j) {
        if (j0 + j < nk) { sc[j] = dot_lds<HD>(q, Kg + (j0 + j) * HD); cmax = fmaxf(cmax, sc[j]); }
        else sc[j] = -INFINITY;
      }
      const float mn = fmaxf(m, cmax);
      const float alpha = __expf(m - mn);      // m = -inf on the first chunk -> 0
      l *= alpha;
#pragma unroll
      for (int e = 0; e < HD; ++e) acc[e] *= alpha;
      m = mn;
#pragma unroll
      for (int j = 0; j < AT_CH; ++j) {
        if (j0 + j < nk) {
          const float pj = __expf(sc[j] - m);
          l += pj;
          axpy_lds<HD>(acc, pj, Vg + (j0 + j) * HD);
        }
      }
    }
  }
  if (PASS == 0 && w.active) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int e = 0; e < HD; ++e) acc[e] *= inv;
    store_vec<T, HD>(out + out_row(p, s, w.i) * p.ld_out + h * HD, acc);
    lse[((long)s * p.H + h) * p.L + w.i] = m + __logf(l);
  }
}

// Backward pass A: thread per query.  dq_i = scale * sum_j p_ij (dO_i.v_j - delta_i) k_j; also writes delta.
template <typename T, int HD>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_kernel(AttnP p, const T* __restrict__ qkv, const T* __restrict__ o,
                                                                 const T* __restrict__ dout, const float* __restrict__ lse,
                                                                 float* __restrict__ delta, T* __restrict__ dqkv,
                                                                 T* __restrict__ dqkv_cls) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int h = blockIdx.y, D = p.H * HD;
  const Who w = who_am_i(p);
  const int KT = p.G > 1 ? p.L : AT_KT;
  const int seq_ld = KT * HD + Tile<HD>::SEQ_PAD;
  float* Ks = sm;
  float* Vs = sm + p.G * seq_ld;
  const int s = w.s0 + w.g;

  float q[HD], dO[HD], dq[HD];
  float my_lse = 0.f, my_delta = 0.f;
#pragma unroll
  for (int e = 0; e < HD; ++e) dq[e] = 0.f;
  if (w.active) {
    load_vec<T, HD>(qkv + in_row(p, s, w.i) * p.ld_qkv + h * HD, q);
    const long orow = out_row(p, s, w.i);
    load_vec<T, HD>(dout + orow * p.ld_dout + h * HD, dO);
    {
      float ov[HD];
      load_vec<T, HD>(o + orow * p.ld_out + h * HD, ov);
#pragma unroll
      for (int e = 0; e < HD; ++e) my_delta += ov[e] * dO[e];
    }
#pragma unroll
    for (int e = 0; e < HD; ++e) q[e] *= p.scale;
    const long li = ((long)s * p.H + h) * p.L + w.i;
    my_lse = lse[li];
    delta[li] = my_delta;
  }
  for (int k0 = 0; k0 < p.L; k0 += KT) {
    const int nk = min(KT, p.L - k0);
    __syncthreads();
    for (int g = 0; g < w.nseq; ++g) {
      const int sg = w.s0 + g;
      load_rows<T, HD>(Ks + g * seq_ld, HD, nk, qkv, p.ld_qkv, D + h * HD, [&](int r) { return in_row(p, sg, k0 + r); });
      load_rows<T, HD>(Vs + g * seq_ld, HD, nk, qkv, p.ld_qkv, 2 * D + h * HD, [&](int r) { return in_row(p, sg, k0 + r); });
    }
    __syncthreads();
    if (!w.active) continue;
    const float* Kg = Ks + w.g * seq_ld;
    const float* Vg = Vs + w.g * seq_ld;
    for (int j = 0; j < nk; ++j) {
      const float pij = __expf(dot_lds<HD>(q, Kg + j * HD) - my_lse);
      const float dp = dot_lds<HD>(dO, Vg + j * HD);
      axpy_lds<HD>(dq, pij * (dp - my_delta), Kg + j * HD);
    }
  }
  if (w.active) {
#pragma unroll
    for (int e = 0; e < HD; ++e) dq[e] *= p.scale;
    T* dst = (p.mode == VTX_ATTN_SPACE && w.i == 0) ? dqkv_cls + (long)s * p.ld_dqkv + h * HD
                                                    : dqkv + in_row(p, s, w.i) * p.ld_dqkv + h * HD;
    store_vec<T, HD>(dst, dq);
  }
}

// Backward pass B: thread per key.  dv_j = sum_i p_ij dO_i;  dk_j = scale * sum_i p_ij (dO_i.v_j - delta_i) q_i.
template <typename T, int HD>
__global__ __launch_bounds__(AT_THREADS, 1) void attn_bwd_dkv_kernel(AttnP p, const T* __restrict__ qkv,
                                                                     const T* __restrict__ dout, const float* __restrict__ lse,
                                                                     const float* __restrict__ delta, T* __restrict__ dqkv,
                                                                     T* __restrict__ dqkv_cls) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int h = blockIdx.y, D = p.H * HD;
  const Who w = who_am_i(p);
  const int QT = p.G > 1 ? p.L : AT_KT;
  const int seq_ld = QT * HD + Tile<HD>::SEQ_PAD;
  float* Qs = sm;
  float* Os = sm + p.G * seq_ld;
  float* Ls = Os + p.G * seq_ld;            // [G][QT] lse
  float* Ds = Ls + p.G * QT;                // [G][QT] delta
  const int s = w.s0 + w.g;

  float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
  for (int e = 0; e < HD; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
  if (w.active) {
    const long r = in_row(p, s, w.i);
    load_vec<T, HD>(qkv + r * p.ld_qkv + D + h * HD, k);
    load_vec<T, HD>(qkv + r * p.ld_qkv + 2 * D + h * HD, v);
#pragma unroll
    for (int e = 0; e < HD; ++e) k[e] *= p.scale;      // s_ij = q_i . (scale k_j)
  }
  for (int q0 = 0; q0 < p.L; q0 += QT) {
    const int nq = min(QT, p.L - q0);
    __syncthreads();
    for (int g = 0; g < w.nseq; ++g) {
      const int sg = w.s0 + g;
      load_rows<T, HD>(Qs + g * seq_ld, HD, nq, qkv, p.ld_qkv, h * HD, [&](int r) { return in_row(p, sg, q0 + r); });
      load_rows<T, HD>(Os + g * seq_ld, HD, nq, dout, p.ld_dout, h * HD, [&](int r) { return out_row(p, sg, q0 + r); });
      for (int r = threadIdx.x; r < nq; r += blockDim.x) {
        const long li = ((long)sg * p.H + h) * p.L + q0 + r;
        Ls[g * QT + r] = lse[li];
        Ds[g * QT + r] = delta[li];
      }
    }
    __syncthreads();
    if (!w.active) continue;
    const float* Qg = Qs + w.g * seq_ld;
    const float* Og = Os + w.g * seq_ld;
    for (int i = 0; i < nq; ++i) {
      const float pij = __expf(dot_lds<HD>(k, Qg + i * HD) - Ls[w.g * QT + i]);
      const float dp = dot_lds<HD>(v, Og + i * HD);
      axpy_lds<HD>(dv, pij, Og + i * HD);
      axpy_lds<HD>(dk, pij * (dp - Ds[w.g * QT + i]), Qg + i * HD);
    }
  }
  if (w.active) {
#pragma unroll
    for (int e = 0; e < HD; ++e) dk[e] *= p.scale;
    T* base = (p.mode == VTX_ATTN_SPACE && w.i == 0) ? dqkv_cls + (long)s * p.ld_dqkv
                                                     : dqkv + in_row(p, s, w.i) * p.ld_dqkv;
    store_vec<T, HD>(base + D + h * HD, dk);
    store_vec<T, HD>(base + 2 * D + h * HD, dv);
  }
}

// bf16, head_dim 64, 33..256 tokens -> MFMA kernels (attn_mfma.hip); VTX_ATTN_VALU=1 forces the VALU path.
static bool use_mfma(int dtype, int L, int hd) {
  if (options().attn_valu) return false;
  return attn_mfma_eligible(dtype, L, hd);
}

static bool use_small(int dtype, int mode, int L, int hd) {
  if (options().attn_valu) return false;
  return attn_small_eligible(dtype, mode, L, hd);
}

static int make_params(const vtx_attn_desc* d, AttnP& p, const char* who) {
  VTX_REQUIRE(d->S > 0 && d->L > 0 && d->H > 0, VTX_EINVAL, "%s: bad shape S=%d L=%d H=%d", who, d->S, d->L, d->H);
  VTX_REQUIRE(d->hd == 64, VTX_EINVAL, "%s: head_dim %d unsupported (64 only)", who, d->hd);
  VTX_REQUIRE(d->mode == VTX_ATTN_CONTIG || d->mode == VTX_ATTN_SPACE, VTX_EINVAL, "%s: bad mode", who);
  VTX_REQUIRE(d->dtype == VTX_F32 || d->dtype == VTX_BF16, VTX_EINVAL, "%s: bad dtype", who);
  if (d->mode == VTX_ATTN_SPACE)
    VTX_REQUIRE(d->B > 0 && d->T > 0 && d->P > 0 && d->S == d->B * d->T && d->L == d->P + 1, VTX_EINVAL,
                "%s: SPACE mode needs S == B*T and L == P+1", who);
  const long vec = d->dtype == VTX_BF16 ? 8 : 4;
  VTX_REQUIRE(d->qkv && aligned16(d->qkv) && d->ld_qkv % vec == 0, VTX_EALIGN, "%s: qkv alignment", who);
  p.mode = d->mode; p.S = d->S; p.L = d->L; p.H = d->H; p.B = d->B; p.T = d->T; p.P = d->P;
  p.ld_qkv = d->ld_qkv; p.ld_out = d->ld_out; p.ld_dout = 0; p.ld_dqkv = 0; p.scale = d->scale;
  p.G = d->L <= AT_KT ? AT_PACK_THREADS / d->L : 1;
  return VTX_OK;
}
static dim3 attn_grid(const AttnP& p) {
  if (p.G > 1) return dim3(cdiv(p.S, p.G), p.H);
  return dim3(p.S * cdiv(p.L, AT_THREADS), p.H);
}
static dim3 attn_block(const AttnP& p) { return dim3(p.G > 1 ? AT_PACK_THREADS : AT_THREADS); }
static size_t attn_lds(const AttnP& p, int hd, bool with_stats) {
  const int KT = p.G > 1 ? p.L : AT_KT;
  size_t b = (size_t)2 * p.G * (KT * hd + 4) * sizeof(float);
  if (with_stats) b += (size_t)2 * p.G * KT * sizeof(float);
  return b;
}

}  // namespace vtx

using namespace vtx;

extern "C" int vtx_attn_fwd(const vtx_attn_desc* d, void* stream) {
  VTX_REQUIRE(d != nullptr, VTX_EINVAL, "attn_fwd: null descriptor");
  AttnP p;
  int rc = make_params(d, p, "attn_fwd");
  if (rc) return rc;
  VTX_REQUIRE(d->out && d->lse && aligned16(d->out), VTX_EINVAL, "attn_fwd: out/lse required");
  hipStream_t st = as_stream(stream);
  const dim3 grid = attn_grid(p), block = attn_block(p);
  const size_t lds = attn_lds(p, 64, false);
  if (use_small(d->dtype, d->mode, d->L, d->hd)) {
    rc = attn_fwd_small_launch(p, d->qkv, d->out, d->lse, st);
  } else if (use_mfma(d->dtype, d->L, d->hd)) {
    rc = attn_fwd_mfma_launch(p, d->qkv, d->out, d->lse, st);
  } else {
    if (d->dtype == VTX_F32)
      hipLaunchKernelGGL((attn_fwd_kernel<float, 64, 0>), grid, block, lds, st, p, (const float*)d->qkv, (float*)d->out, d->lse, nullptr);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<bf16raw, 64, 0>), grid, block, lds, st, p, (const bf16raw*)d->qkv, (bf16raw*)d->out, d->lse, nullptr);
    rc = check_launch("attn_fwd");
  }
  if (rc || !d->probs) return rc;
  if (d->dtype == VTX_F32)
    hipLaunchKernelGGL((attn_fwd_kernel<float, 64, 1>), grid, block, lds, st, p, (const float*)d->qkv, (float*)d->out, d->lse, d->probs);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<bf16raw, 64, 1>), grid, block, lds, st, p, (const bf16raw*)d->qkv, (bf16raw*)d->out, d->lse, d->probs);
  return check_launch("attn_probs");
}

extern "C" int vtx_attn_bwd(const vtx_attn_bwd_desc* d, void* stream) {
  VTX_REQUIRE(d != nullptr, VTX_EINVAL, "attn_bwd: null descriptor");
  AttnP p;
  int rc = make_params(&d->f, p, "attn_bwd");
  if (rc) return rc;
  VTX_REQUIRE(d->f.out && d->f.lse && d->dout && d->dqkv && d->delta, VTX_EINVAL, "attn_bwd: null pointer");
  VTX_REQUIRE(d->f.mode != VTX_ATTN_SPACE || d->dqkv_cls, VTX_EINVAL, "attn_bwd: SPACE mode needs dqkv_cls");
  VTX_REQUIRE(aligned16(d->dout) && aligned16(d->dqkv), VTX_EALIGN, "attn_bwd: alignment");
  p.ld_dout = d->ld_dout; p.ld_dqkv = d->ld_dqkv;
  hipStream_t st = as_stream(stream);
  const dim3 grid = attn_grid(p), block = attn_block(p);
  const size_t lds_a = attn_lds(p, 64, false), lds_b = attn_lds(p, 64, true);
  if (use_small(d->f.dtype, d->f.mode, d->f.L, d->f.hd))
    return attn_bwd_small_launch(p, d->f.qkv, d->f.out, d->dout, d->f.lse, d->dqkv, st);
  if (use_mfma(d->f.dtype, d->f.L, d->f.hd))
    return attn_bwd_mfma_launch(p, d->f.qkv, d->f.out, d->dout, d->f.lse, d->delta, d->dqkv, d->dqkv_cls, st);
  if (d->f.dtype == VTX_F32) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<float, 64>), grid, block, lds_a, st, p, (const float*)d->f.qkv, (const float*)d->f.out,
                       (const float*)d->dout, d->f.lse, d->delta, (float*)d->dqkv, (float*)d->dqkv_cls);
    rc = check_launch("attn_bwd_dq");
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<float, 64>), grid, block, lds_b, st, p, (const float*)d->f.qkv, (const float*)d->dout,
                       d->f.lse, d->delta, (float*)d->dqkv, (float*)d->dqkv_cls);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16raw, 64>), grid, block, lds_a, st, p, (const bf16raw*)d->f.qkv, (const bf16raw*)d->f.out,
                       (const bf16raw*)d->dout, d->f.lse, d->delta, (bf16raw*)d->dqkv, (bf16raw*)d->dqkv_cls);
    rc = check_launch("attn_bwd_dq");
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<bf16raw, 64>), grid, block, lds_b, st, p, (const bf16raw*)d->f.qkv, (const bf16raw*)d->dout,
                       d->f.lse, d->delta, (bf16raw*)d->dqkv, (bf16raw*)d->dqkv_cls);
  }
  return check_launch("attn_bwd_dkv");
}
