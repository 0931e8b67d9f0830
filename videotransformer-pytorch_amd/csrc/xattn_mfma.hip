// xattn_mfma.hip -- bf16 MFMA cross attention, softmax(q k^T hd^-0.5) v with Lq != Lk, head_dim 96 or 64.
//
// The pooled attention of the MViT-B backbone of MaskFeat (reference video_transformer.py:763-786 builds the blocks from
// pytorchvideo's MultiScaleAttention; semantics restated in oracle/mvit_oracle.py -- parity unpinned by the reference):
// queries and keys are pooled with different strides, so a (clip, head) has Lq in {25 089, 6 273, 1 569} queries against
// Lk = 393 keys, head_dim 96.  Tokens are [B, L, heads * hd] with the heads interleaved in the feature axis.
//
// The structure is the one of attn_mfma.hip (see there for the layout argument): the score tile is computed TRANSPOSED
// (S^T = K Q^T, v_mfma_f32_32x32x16_bf16) so that a lane owns one query column -- row statistics stay in the lane, and the
// probabilities are already the B operand of O^T = V^T P^T; transposed operands come out of LDS through
// ds_read_b64_tr_b16 with the same key permutation on both sides.  What differs:
//   * queries are many and keys few: a workgroup (4 waves x 32 queries) streams the keys through LDS in chunks of 128
//     (forward, dq); the dk / dv kernel gives a wave one 32-key tile and streams ITS share of the queries in chunks of
//     128, several workgroups splitting the query range and leaving fp32 partials that a fixed-order reduction sums
//     (deterministic, no atomics);
//   * rows are hd elements wide (192 B for hd 96): 16-byte chunks are XOR-swizzled inside aligned groups of four
//     chunks with (row >> 2) & 3, which keeps the row-wise ds_read_b128 fragment reads conflict-free for any hd that is
//     a multiple of 32.
// FLOPs per (clip, head): forward 4 Lq Lk hd, backward 10 Lq Lk hd (+ 4 for the recomputed scores).
// The fp32 VALU kernels of mvit.hip keep hd floats of three operands per thread in registers (spilled at hd 96):
// 26 ms per backward launch at 32 clips, 77 % of the MaskFeat step; they stay as the exact-fp32 path.
#include "common.h"

namespace vtx {

namespace xa {

constexpr int THREADS = 256;
constexpr int CHUNK = 128;                 // rows of the streamed operand per LDS chunk (4 tiles of 32)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int HD> __device__ inline int sw_chunk(int row, int chunk) { return (chunk & ~3) | ((chunk & 3) ^ ((row >> 2) & 3)); }
template <int HD> __device__ inline int off(int row, int col) { return row * HD + (sw_chunk<HD>(row, col >> 3) << 3) + (col & 7); }

__device__ inline int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ inline void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
__device__ inline bf16x8 pack8(const float* f) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)f[j];
  return v;
}
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

// Fill TWO swizzled [CHUNK][HD] tiles with rows row0 .. row0 + nrows - 1 of two [L, ld] matrices (column origin folded
// into the base pointers); rows beyond nrows are zero.  All global loads are issued before the first LDS store.
template <int HD>
__device__ inline void fill2(bf16raw* t0, bf16raw* t1, const bf16raw* b0, const bf16raw* b1, long ld, long row0, int nrows) {
  constexpr int CPR = HD / 8, PER = CHUNK * CPR / THREADS;
  uint4 v0[PER], v1[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int id = threadIdx.x + i * THREADS;
    const int r = id / CPR, c = id - r * CPR;
    v0[i] = make_uint4(0, 0, 0, 0);
    v1[i] = make_uint4(0, 0, 0, 0);
    if (r < nrows) {
      v0[i] = *reinterpret_cast<const uint4*>(b0 + (row0 + r) * ld + c * 8);
      v1[i] = *reinterpret_cast<const uint4*>(b1 + (row0 + r) * ld + c * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int id = threadIdx.x + i * THREADS;
    const int r = id / CPR, c = id - r * CPR;
    const int o = r * HD + (sw_chunk<HD>(r, c) << 3);
    *reinterpret_cast<uint4*>(t0 + o) = v0[i];
    *reinterpret_cast<uint4*>(t1 + o) = v1[i];
  }
}

// Row-wise operand fragment: lane (l & 31) -> row row0 + (l & 31), 8 consecutive columns ks * 16 + 8 * (l >> 5).
template <int HD> __device__ inline bf16x8 frag_rows(const bf16raw* lds, int row0, int ks, int lane) {
  const int row = row0 + (lane & 31);
  return *reinterpret_cast<const bf16x8*>(lds + row * HD + (sw_chunk<HD>(row, ks * 2 + (lane >> 5)) << 3));
}
// Transposed operand fragment: A[i = column col0 + (l & 31)][k], k over the 16 rows row0 .. row0 + 15 in the permuted
// order {0-3, 8-11 | 4-7, 12-15} (lower | upper half-wave) -- the order the accumulator layout gives the other operand.
template <int HD> __device__ inline bf16x8 frag_cols(const bf16raw* lds, int row0, int col0, int lane) {
  const int r = row0 + 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int c = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off<HD>(r, c)));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off<HD>(r + 8, c)));
  return u.v;
}
// The KS row-wise fragments of one row of a [L, ld] matrix straight from global memory (this lane's row, clamped by the caller).
template <int HD> __device__ inline void load_row(bf16x8 (&f)[HD / 16], const bf16raw* row_ptr, int lane) {
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) {
    union { bf16x8 v; uint4 u; } x;
    x.u = *reinterpret_cast<const uint4*>(row_ptr + ks * 16 + 8 * (lane >> 5));
    f[ks] = x.v;
  }
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ inline float dot8(const bf16x8& a, const bf16x8& b, float acc) {
  union { bf16x8 v; bf16x2 h[4]; } ua, ub;
  ua.v = a; ub.v = b;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_fdot2_f32_bf16(ua.h[j], ub.h[j], acc, false);
  return acc;
}

// Store a [32 x HD] result held transposed (lane & 31 = row, registers = HD columns in HD / 32 accumulator tiles) as whole
// rows through a wave-private swizzled staging tile.  ptr_of_row(r) -> destination of tile row r, or nullptr.
template <int HD, typename PtrFn>
__device__ inline void store_rows_T(bf16raw* stg, const f32x16 (&acc)[HD / 32], float mul, int lane, PtrFn ptr_of_row) {
  const int row = lane & 31;
#pragma unroll
  for (int nt = 0; nt < HD / 32; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = nt * 32 + 8 * g + 4 * (lane >> 5);
      union { bf16x4 v; uint2 u; } w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w.v[j] = (__bf16)(acc[nt][4 * g + j] * mul);
      *reinterpret_cast<uint2*>(stg + off<HD>(row, col)) = w.u;
    }
  wave_lds_sync();
  constexpr int CPR = HD / 8;
#pragma unroll
  for (int i = 0; i < 32 * CPR / 64; ++i) {
    const int id = i * 64 + lane;
    const int r = id / CPR, c = id - r * CPR;
    const uint4 v = *reinterpret_cast<const uint4*>(stg + r * HD + (sw_chunk<HD>(r, c) << 3));
    bf16raw* dst = ptr_of_row(r);
    if (dst) *reinterpret_cast<uint4*>(dst + c * 8) = v;
  }
  wave_lds_sync();
}

// ------------------------------------------------------------------------------------------------ forward
template <int HD>
__global__ __launch_bounds__(THREADS, 2) void fwd_kernel(int Lq, int Lk, int heads, float scale, const bf16raw* __restrict__ q,
                                                         const bf16raw* __restrict__ k, const bf16raw* __restrict__ v,
                                                         bf16raw* __restrict__ out, float* __restrict__ lse) {
  constexpr int KS = HD / 16, NT = HD / 32, KC = CHUNK / 32;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  bf16raw* Ks = reinterpret_cast<bf16raw*>(sm);
  bf16raw* Vs = Ks + CHUNK * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16raw* stg = Vs + CHUNK * HD + wave * 32 * HD;
  const int b = blockIdx.z, h = blockIdx.y;
  const long C = (long)heads * HD;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qi = q0 + (lane & 31);
  const int qc = qi < Lq ? qi : Lq - 1;              // a padded query keeps to its lane and is never stored
  bf16x8 qf[KS];
  load_row<HD>(qf, q + ((long)b * Lq + qc) * C + h * HD, lane);
  const bf16raw* kb = k + (long)b * Lk * C + h * HD;
  const bf16raw* vb = v + (long)b * Lk * C + h * HD;
  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) zero16(acc[n]);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float c2 = scale * LOG2E;
  float m = -1e30f, l = 0.f;                         // running max of the RAW scores (scale > 0)
  for (int k0 = 0; k0 < Lk; k0 += CHUNK) {
    const int nrows = min(CHUNK, Lk - k0);
    __syncthreads();                                 // the previous chunk is consumed
    fill2<HD>(Ks, Vs, kb, vb, C, k0, nrows);
    __syncthreads();
    const int nt = (nrows + 31) >> 5;
    f32x16 st[KC];
#pragma unroll
    for (int t = 0; t < KC; ++t)
      if (t < nt) {
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Ks, t * 32, 0, lane), qf[0], zero, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks)
          st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Ks, t * 32, ks, lane), qf[ks], st[t], 0, 0, 0);
      }
    float bm = -1e30f;
#pragma unroll
    for (int t = 0; t < KC; ++t)
      if (t < nt) {
        if (t * 32 + 32 > nrows) {                   // the tile that holds padded keys
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t * 32 + crow(r, lane) >= nrows) st[t][r] = -1e30f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) bm = fmaxf(bm, st[t][r]);
      }
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float mn = fmaxf(m, bm);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c2);
    m = mn;
    const float mc = mn * c2;
    float bl = 0.f;
#pragma unroll
    for (int t = 0; t < KC; ++t)
      if (t < nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(st[t][r], c2, -mc)); st[t][r] = e; bl += e; }
      }
    bl += __shfl_xor(bl, 32, 64);
    l = l * alpha + bl;
    if (k0 > 0) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] *= alpha;
    }
#pragma unroll
    for (int t = 0; t < KC; ++t)
      if (t < nt) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float pf[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = st[t][8 * s2 + j];
          const bf16x8 pb = pack8(pf);
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<HD>(Vs, t * 32 + 16 * s2, n * 32, lane), pb, acc[n], 0, 0, 0);
        }
      }
  }
  store_rows_T<HD>(stg, acc, 1.0f / l, lane, [&](int r) -> bf16raw* {
    const int qq = q0 + r;
    return qq < Lq ? out + ((long)b * Lq + qq) * C + h * HD : nullptr;
  });
  if (qi < Lq && lane < 32) lse[((long)b * heads + h) * Lq + qi] = (m * c2) * LN2 + __logf(l);
}

// ------------------------------------------------------------------------------------------------ backward: dq (+ delta)
template <int HD>
__global__ __launch_bounds__(THREADS, 2) void bwd_dq_kernel(int Lq, int Lk, int heads, float scale, const bf16raw* __restrict__ q,
                                                            const bf16raw* __restrict__ k, const bf16raw* __restrict__ v,
                                                            const bf16raw* __restrict__ o, const bf16raw* __restrict__ dout,
                                                            const float* __restrict__ lse, float* __restrict__ delta,
                                                            bf16raw* __restrict__ dq) {
  constexpr int KS = HD / 16, NT = HD / 32, KC = CHUNK / 32;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  bf16raw* Ks = reinterpret_cast<bf16raw*>(sm);
  bf16raw* Vs = Ks + CHUNK * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16raw* stg = Vs + CHUNK * HD + wave * 32 * HD;
  const int b = blockIdx.z, h = blockIdx.y;
  const long C = (long)heads * HD;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qi = q0 + (lane & 31);
  const int qc = qi < Lq ? qi : Lq - 1;
  const long roff = ((long)b * Lq + qc) * C + h * HD;
  bf16x8 qf[KS], df[KS];
  float dl = 0.f;                                    // delta = rowsum(dO * O): this lane's half of the row's columns
  {
    bf16x8 of[KS];
    load_row<HD>(qf, q + roff, lane);
    load_row<HD>(df, dout + roff, lane);
    load_row<HD>(of, o + roff, lane);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dl = dot8(df[ks], of[ks], dl);
  }
  dl += __shfl_xor(dl, 32, 64);
  const long li = ((long)b * heads + h) * Lq + qc;
  const float l2 = lse[li] * LOG2E;
  if (qi < Lq && lane < 32) delta[li] = dl;
  const bf16raw* kb = k + (long)b * Lk * C + h * HD;
  const bf16raw* vb = v + (long)b * Lk * C + h * HD;
  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) zero16(acc[n]);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float c2 = scale * LOG2E;
  for (int k0 = 0; k0 < Lk; k0 += CHUNK) {
    const int nrows = min(CHUNK, Lk - k0);
    __syncthreads();
    fill2<HD>(Ks, Vs, kb, vb, C, k0, nrows);
    __syncthreads();
    const int nt = (nrows + 31) >> 5;
#pragma unroll
    for (int t = 0; t < KC; ++t)
      if (t < nt) {
        f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Ks, t * 32, 0, lane), qf[0], zero, 0, 0, 0);
        f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Vs, t * 32, 0, lane), df[0], zero, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Ks, t * 32, ks, lane), qf[ks], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Vs, t * 32, ks, lane), df[ks], dp, 0, 0, 0);
        }
        // dS = P (dP - delta); the softmax scale is applied once to dq at the store.  Padded keys have zero K rows
        // (they add nothing to dq) but their probability is masked so that the bf16 pack stays clean.
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(fmaf(st[r], c2, -l2)) * (dp[r] - dl);
        if (t * 32 + 32 > nrows) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t * 32 + crow(r, lane) >= nrows) ds[r] = 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 db = pack8(ds + 8 * s2);
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<HD>(Ks, t * 32 + 16 * s2, n * 32, lane), db, acc[n], 0, 0, 0);
        }
      }
  }
  store_rows_T<HD>(stg, acc, scale, lane, [&](int r) -> bf16raw* {
    const int qq = q0 + r;
    return qq < Lq ? dq + ((long)b * Lq + qq) * C + h * HD : nullptr;
  });
}

// ------------------------------------------------------------------------------------------------ backward: dk, dv
// Lanes = keys: wave w of workgroup (key group kg, query split sp) owns key tile kg * 4 + w and walks the queries
// [sp * q_per, (sp + 1) * q_per) in chunks of 128 through LDS.  fp32 partials part_k / part_v [split][B, Lk, heads * hd].
template <int HD>
__global__ __launch_bounds__(THREADS, 2) void bwd_dkv_kernel(int Lq, int Lk, int heads, float scale, const bf16raw* __restrict__ q,
                                                             const bf16raw* __restrict__ k, const bf16raw* __restrict__ v,
                                                             const bf16raw* __restrict__ dout, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, int kgroups, int q_per,
                                                             float* __restrict__ part_k, float* __restrict__ part_v, long part_stride) {
  constexpr int KS = HD / 16, NT = HD / 32, QC = CHUNK / 32;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  bf16raw* Qs = reinterpret_cast<bf16raw*>(sm);
  bf16raw* Os = Qs + CHUNK * HD;
  float* Ls = reinterpret_cast<float*>(Os + CHUNK * HD);     // lse * log2(e), +huge on padded rows
  float* Ds = Ls + CHUNK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const long C = (long)heads * HD;
  const int sp = blockIdx.x / kgroups, kg = blockIdx.x - sp * kgroups;
  const int key0 = (kg * 4 + wave) * 32;
  const bool wave_active = key0 < Lk;                // wave-uniform; idle waves still help to fill and meet the barriers
  const int key = key0 + (lane & 31);
  const int kc = key < Lk ? key : Lk - 1;            // a padded key keeps to its lane and is never stored
  bf16x8 kf[KS], vf[KS];
  load_row<HD>(kf, k + ((long)b * Lk + kc) * C + h * HD, lane);
  load_row<HD>(vf, v + ((long)b * Lk + kc) * C + h * HD, lane);
  const int q_begin = sp * q_per, q_end = min(Lq, q_begin + q_per);
  const bf16raw* qb = q + (long)b * Lq * C + h * HD;
  const bf16raw* ob = dout + (long)b * Lq * C + h * HD;
  const float* lb = lse + ((long)b * heads + h) * Lq;
  const float* db_ = delta + ((long)b * heads + h) * Lq;
  f32x16 dk[NT], dv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { zero16(dk[n]); zero16(dv[n]); }
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float c2 = scale * LOG2E;
  for (int qs = q_begin; qs < q_end; qs += CHUNK) {
    const int nrows = min(CHUNK, q_end - qs);
    __syncthreads();
    fill2<HD>(Qs, Os, qb, ob, C, qs, nrows);
    if (threadIdx.x < CHUNK) {
      const int i = threadIdx.x;
      Ls[i] = i < nrows ? lb[qs + i] * LOG2E : 1e30f;          // padded query rows: P = exp2(. - huge) = 0
      Ds[i] = i < nrows ? db_[qs + i] : 0.f;
    }
    __syncthreads();
    if (!wave_active) continue;
    const int nt = (nrows + 31) >> 5;
#pragma unroll
    for (int t = 0; t < QC; ++t)
      if (t < nt) {
        f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Qs, t * 32, 0, lane), kf[0], zero, 0, 0, 0);
        f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Os, t * 32, 0, lane), vf[0], zero, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Qs, t * 32, ks, lane), kf[ks], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Os, t * 32, ks, lane), vf[ks], dp, 0, 0, 0);
        }
        float pr[16], ds[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int qrow = t * 32 + 8 * g + 4 * (lane >> 5);
          const float4 l4 = *reinterpret_cast<const float4*>(Ls + qrow);
          const float4 d4 = *reinterpret_cast<const float4*>(Ds + qrow);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            const float e = __builtin_amdgcn_exp2f(fmaf(st[r], c2, -lv[j]));
            pr[r] = e;
            ds[r] = e * (dp[r] - dvv[j]);            // the softmax scale is applied once to dk at the store
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pb = pack8(pr + 8 * s2);
          const bf16x8 dsb = pack8(ds + 8 * s2);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            dv[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<HD>(Os, t * 32 + 16 * s2, n * 32, lane), pb, dv[n], 0, 0, 0);
            dk[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<HD>(Qs, t * 32 + 16 * s2, n * 32, lane), dsb, dk[n], 0, 0, 0);
          }
        }
      }
  }
  if (wave_active && key < Lk) {                     // fp32 partial rows: lane = key row, 4 consecutive columns per store
    float* pk = part_k + (long)sp * part_stride + ((long)b * Lk + key) * C + h * HD;
    float* pv = part_v + (long)sp * part_stride + ((long)b * Lk + key) * C + h * HD;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = n * 32 + 8 * g + 4 * (lane >> 5);
        *reinterpret_cast<float4*>(pk + col) = make_float4(dk[n][4 * g] * scale, dk[n][4 * g + 1] * scale, dk[n][4 * g + 2] * scale,
                                                           dk[n][4 * g + 3] * scale);
        *reinterpret_cast<float4*>(pv + col) = make_float4(dv[n][4 * g], dv[n][4 * g + 1], dv[n][4 * g + 2], dv[n][4 * g + 3]);
      }
  }
}

template <int HD> constexpr size_t lds_fwd() { return (size_t)(2 * CHUNK * HD + 4 * 32 * HD) * 2; }
template <int HD> constexpr size_t lds_dkv() { return (size_t)(2 * CHUNK * HD) * 2 + 2 * CHUNK * 4; }

template <auto Kernel> static void allow(size_t lds) {
  static std::atomic<unsigned long long> seen{0};
  if (lds > 65536 && first_launch_on_device(seen))
    hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace xa

bool xattn_mfma_eligible(int dtype, int hd) { return dtype == VTX_BF16 && (hd == 96 || hd == 64); }

// query splits of the dk / dv kernel: ~4 workgroups per CU, at least four 128-query chunks per split
int xattn_mfma_splits(int B, int Lq, int Lk, int heads) {
  const int kgroups = cdiv(cdiv(Lk, 32), 4);
  const long blocks = (long)kgroups * heads * B;
  int s = (int)(1024 / (blocks > 0 ? blocks : 1));
  const int max_s = cdiv(Lq, 4 * xa::CHUNK);
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}

template <int HD>
static int xattn_fwd_t(const vtx_xattn_desc* d, hipStream_t st) {
  const dim3 g(cdiv(d->Lq, 128), d->heads, d->B), blk(xa::THREADS);
  xa::allow<xa::fwd_kernel<HD>>(xa::lds_fwd<HD>());
  hipLaunchKernelGGL((xa::fwd_kernel<HD>), g, blk, xa::lds_fwd<HD>(), st, d->Lq, d->Lk, d->heads, d->scale, (const bf16raw*)d->q,
                     (const bf16raw*)d->k, (const bf16raw*)d->v, (bf16raw*)d->out, d->lse);
  return check_launch("xattn_fwd_mfma");
}
int xattn_fwd_mfma_launch(const vtx_xattn_desc* d, hipStream_t st) {
  return d->hd == 96 ? xattn_fwd_t<96>(d, st) : xattn_fwd_t<64>(d, st);
}

// dq + delta, then the dk / dv partials (part_k, part_v: [nsplit][B, Lk, heads * hd] fp32); the caller reduces them.
template <int HD>
static int xattn_bwd_t(const vtx_xattn_desc* d, const void* dout, float* delta, void* dq, int nsplit, float* part_k, float* part_v,
                       long part_stride, hipStream_t st) {
  const dim3 gq(cdiv(d->Lq, 128), d->heads, d->B), blk(xa::THREADS);
  xa::allow<xa::bwd_dq_kernel<HD>>(xa::lds_fwd<HD>());
  hipLaunchKernelGGL((xa::bwd_dq_kernel<HD>), gq, blk, xa::lds_fwd<HD>(), st, d->Lq, d->Lk, d->heads, d->scale, (const bf16raw*)d->q,
                     (const bf16raw*)d->k, (const bf16raw*)d->v, (const bf16raw*)d->out, (const bf16raw*)dout, d->lse, delta,
                     (bf16raw*)dq);
  int rc = check_launch("xattn_bwd_dq_mfma");
  if (rc) return rc;
  const int kgroups = cdiv(cdiv(d->Lk, 32), 4);
  const int q_per = cdiv(cdiv(d->Lq, nsplit), xa::CHUNK) * xa::CHUNK;
  const dim3 gk(kgroups * nsplit, d->heads, d->B);
  xa::allow<xa::bwd_dkv_kernel<HD>>(xa::lds_dkv<HD>());
  hipLaunchKernelGGL((xa::bwd_dkv_kernel<HD>), gk, blk, xa::lds_dkv<HD>(), st, d->Lq, d->Lk, d->heads, d->scale, (const bf16raw*)d->q,
                     (const bf16raw*)d->k, (const bf16raw*)d->v, (const bf16raw*)dout, d->lse, delta, kgroups, q_per, part_k, part_v,
                     part_stride);
  return check_launch("xattn_bwd_dkv_mfma");
}
int xattn_bwd_mfma_launch(const vtx_xattn_desc* d, const void* dout, float* delta, void* dq, int nsplit, float* part_k, float* part_v,
                          long part_stride, hipStream_t st) {
  return d->hd == 96 ? xattn_bwd_t<96>(d, dout, delta, dq, nsplit, part_k, part_v, part_stride, st)
                     : xattn_bwd_t<64>(d, dout, delta, dq, nsplit, part_k, part_v, part_stride, st);
}

}  // namespace vtx
