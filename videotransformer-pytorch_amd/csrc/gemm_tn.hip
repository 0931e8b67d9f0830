// gemm_tn.hip -- weight gradients: C[N1,N2] (fp32) (+)= sum_m A[m][n1] * B[m][n2].
//
// Replaces autograd's mm(grad_out^T, input) for every nn.Linear on the path
// (reference transformer.py:160,162,225,501,505; patch projection :116-126;
// video_transformer.py:855).  The reduction runs over the token rows M, which
// are NOT contiguous for either operand, so the MFMA operand fragments are
// transposed reads of row-major LDS tiles:
//   bf16: tiles [64 m][128 n] bf16 (row stride 160 elements = 320 B so that the
//         4 rows x 64 B touched by each half-wave of ds_read_b64_tr_b16 fall in
//         four disjoint bank quarters); the hardware transpose read
//         ds_read_b64_tr_b16 delivers 4 consecutive-m values of one column per
//         lane, two reads build the 8-deep K fragment of v_mfma_f32_32x32x16_bf16.
//         VTX_TN_SAFE=1 selects plain 2-byte LDS gathers instead (slow, no
//         dependence on the tr instruction) -- kept as a diagnostic path.
//   fp32: tiles [16 m][128 n]; v_mfma_f32_32x32x2_f32 takes one element per lane,
//         read straight along the row (conflict free, no transpose needed).
// M is split over `splits` workgroup slices (fp32 slabs in the workspace) that a
// second kernel sums in a fixed order: deterministic, no atomics.
// Algorithmic FLOPs per launch: 2*M*N1*N2.
#include <stdlib.h>
#include <string>
#include "gemm_common.h"

namespace vtx {

constexpr int TN_BKM = 64;        // m rows per tile (bf16)
constexpr int TN_LD = 160;        // padded row length (bf16 elements)
constexpr int TN_BKM32 = 16;      // m rows per tile (fp32)
constexpr int TN_LD32 = 128;

struct TnOut {
  float* slab; long slab_stride;  // [splits][N1*N2]
  int N1, N2;
  float* cslab;                   // [splits][N1] column sums of A (bias gradient), or nullptr
  int cs_fold;                    // ring / ping-pong kernels: [splits][cs_fold][N1], copy t2 = the K tiles kt % tiles2 == t2
};

__device__ inline void tn_store(const TnOut& o, const float* stage, int split, int r_base, int c_base, int lane) {
  float* dst = o.slab + (long)split * o.slab_stride;
#pragma unroll 1
  for (int e = 0; e < 8; ++e) {
    const int rw = e * 8 + (lane >> 3);
    const int r = r_base + rw, c = c_base + (lane & 7) * 8;
    if (r >= o.N1 || c >= o.N2) continue;
    const float* s = stage + rw * STAGE_LD + (lane & 7) * 8;
    float* d = dst + (long)r * o.N2 + c;
    *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(s);
    *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(s + 4);
  }
}

template <bool SAFE>
__global__ __launch_bounds__(NT_THREADS) void gemm_tn_bf16_kernel(
    int M, int m_per_split, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap,
    const bf16raw* __restrict__ B, long ldb, vtx_rowmap bmap, int tiles2, int tiles12, TnOut out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* As = reinterpret_cast<bf16raw*>(smem);          // [64][160]
  bf16raw* Bs = As + TN_BKM * TN_LD;                       // [64][160]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int split = blockIdx.x / tiles12;
  const int tile = blockIdx.x - split * tiles12;
  const int t1 = tile / tiles2, t2 = tile - t1 * tiles2;
  const int r0 = t1 * 128, c0 = t2 * 128;               // output tile origin (n1, n2)
  const int m_begin = split * m_per_split;
  const int m_end = min(M, m_begin + m_per_split);

  // loader: 64 rows x 16 chunks(16 B) = 1024 chunks -> 4 per thread per operand
  const int lc = tid & 15, lr = tid >> 4;
  const bool a_col_ok = (r0 + lc * 8) < out.N1;
  const bool b_col_ok = (c0 + lc * 8) < out.N2;
  const int a_col = a_col_ok ? r0 + lc * 8 : 0;
  const int b_col = b_col_ok ? c0 + lc * 8 : 0;
  uint4 ra[4], rb[4];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
#define TN_GLOAD16(mt_)                                                                                          \
  {                                                                                                              \
    const int mt__ = (mt_);                                                                                      \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                           \
      const int m = mt__ + lr + 16 * it;                                                                         \
      const bool ok = m < m_end;                                                                                 \
      const int mc = m < M ? m : M - 1; /* always-valid address; zeroed below */                                 \
      ra[it] = *reinterpret_cast<const uint4*>(A + map_row(amap, mc) * lda + a_col);                             \
      rb[it] = *reinterpret_cast<const uint4*>(B + map_row(bmap, mc) * ldb + b_col);                             \
      if (!(ok && a_col_ok)) ra[it] = zero4;                                                                     \
      if (!(ok && b_col_ok)) rb[it] = zero4;                                                                     \
    }                                                                                                            \
  }
#define TN_LSTORE16()                                                  \
  {                                                                    \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                 \
      const int off = (lr + 16 * it) * TN_LD + lc * 8;                 \
      *reinterpret_cast<uint4*>(As + off) = ra[it];                    \
      *reinterpret_cast<uint4*>(Bs + off) = rb[it];                    \
    }                                                                  \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Transposed fragment addressing.  MFMA operand: lane l holds column (l&31) of
  // the 32-wide tile for k = 8*(l>>5) + j, j = 0..7.  ds_read_b64_tr_b16: in
  // each 16-lane group, lane r supplies the address of 4 contiguous elements of
  // row (r>>2), columns 4*(r&3)..+3 of a [4][16] block; lane i receives column i.
  const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int sf_row = 8 * (lane >> 5);           // SAFE path: own column, 8 rows
  const int sf_col = lane & 31;

  const bool do_cs = out.cslab != nullptr && t2 == 0;     // block-uniform
  float csum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  TN_GLOAD16(m_begin);
  for (int mt = m_begin; mt < m_end; mt += TN_BKM) {
    TN_LSTORE16();
    if (do_cs) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t w[4] = {ra[it].x, ra[it].y, ra[it].z, ra[it].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          csum[2 * j] += __uint_as_float(w[j] << 16);
          csum[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
        }
      }
    }
    __syncthreads();
    if (mt + TN_BKM < m_end) TN_GLOAD16(mt + TN_BKM);
#pragma unroll
    for (int ks = 0; ks < TN_BKM / 16; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (SAFE) {
          const bf16raw* pa = As + (ks * 16 + sf_row) * TN_LD + wm * 64 + i * 32 + sf_col;
          const bf16raw* pb = Bs + (ks * 16 + sf_row) * TN_LD + wn * 64 + i * 32 + sf_col;
          union { bf16x8 v; bf16raw s[8]; } ua, ub;
#pragma unroll
          for (int j = 0; j < 8; ++j) { ua.s[j] = pa[j * TN_LD]; ub.s[j] = pb[j * TN_LD]; }
          af[i] = ua.v; bfr[i] = ub.v;
        } else {
          const bf16raw* pa = As + (ks * 16 + tr_row) * TN_LD + wm * 64 + i * 32 + tr_col;
          const bf16raw* pb = Bs + (ks * 16 + tr_row) * TN_LD + wn * 64 + i * 32 + tr_col;
          union { bf16x8 v; s16x4 h[2]; } ua, ub;
          ua.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
          ua.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa + 4 * TN_LD));
          ub.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
          ub.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb + 4 * TN_LD));
          af[i] = ua.v; bfr[i] = ub.v;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  if (do_cs) {           // reduce the 16 row-lanes of every column chunk through LDS
    float* red = reinterpret_cast<float*>(smem);            // [16][128]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[lr * 128 + lc * 8 + j] = csum[j];
    __syncthreads();
    if (tid < 128 && r0 + tid < out.N1) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += red[r * 128 + tid];
      out.cslab[(long)split * out.slab_stride + r0 + tid] = a;
    }
    __syncthreads();
  }
  float* stage = reinterpret_cast<float*>(smem) + wave * 64 * STAGE_LD;
  stage_acc(stage, acc, lane);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  tn_store(out, stage, split, r0 + wm * 64, c0 + wn * 64, lane);
}

// ---- bf16 with LDS-DMA staging -----------------------------------------------------------
// Tiles [64 m][128 n] bf16, 256-B rows, no padding (a DMA piece = 1 KiB = 4 whole rows).  The
// 16-B chunk index is XOR-ed with (row&3)<<2 on the SOURCE side so the 4 rows x 64 B a half-wave
// touches in one ds_read_b64_tr_b16 fall into the four different 64-B bank quarters.  Full
// 64-row tiles are staged by DMA into a double buffer; the (at most one) ragged last tile of a
// split goes through registers with zero fill.
__device__ inline void tn_dma16(const bf16raw* src, bf16raw* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
constexpr int TN_DLD = 128;   // elements per LDS row in the DMA layout

__global__ __launch_bounds__(NT_THREADS) void gemm_tn_bf16_dma_kernel(
    int M, int m_per_split, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap,
    const bf16raw* __restrict__ B, long ldb, vtx_rowmap bmap, int tiles2, int tiles12, TnOut out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* As = reinterpret_cast<bf16raw*>(smem);          // [2][64][128]
  bf16raw* Bs = As + 2 * TN_BKM * TN_DLD;                  // [2][64][128]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int split = blockIdx.x / tiles12;
  const int tile = blockIdx.x - split * tiles12;
  const int t1 = tile / tiles2, t2 = tile - t1 * tiles2;
  const int r0 = t1 * 128, c0 = t2 * 128;
  const int m_begin = split * m_per_split;
  const int m_end = min(M, m_begin + m_per_split);

  // DMA assignment: piece p = wave*4 + j covers tile rows 4p..4p+3; lane -> row 4p + (lane>>4), physical chunk lane&15
  const int prow = lane >> 4;                               // == row & 3
  const int lchunk = (lane & 15) ^ (prow << 2);             // logical chunk this lane fetches
  const int a_col = (r0 + lchunk * 8) < out.N1 ? r0 + lchunk * 8 : 0;   // out-of-range columns: any valid address
  const int b_col = (c0 + lchunk * 8) < out.N2 ? c0 + lchunk * 8 : 0;
#define TN_STAGE_DMA(buf_, mt_)                                                                       \
  {                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
      const int m = (mt_) + (wave * 4 + j) * 4 + prow;                                                \
      tn_dma16(A + map_row(amap, m) * lda + a_col, As + (buf_) * TN_BKM * TN_DLD + (wave * 4 + j) * 4 * TN_DLD); \
      tn_dma16(B + map_row(bmap, m) * ldb + b_col, Bs + (buf_) * TN_BKM * TN_DLD + (wave * 4 + j) * 4 * TN_DLD); \
    }                                                                                                 \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed fragment addressing (see gemm_tn_bf16_kernel); row&3 == (lane&15)>>2 for both reads
  const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int tr_sw = ((lane & 15) >> 2) << 2;
  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = wm * 64 + i * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int cb = wn * 64 + i * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    a_off[i] = tr_row * TN_DLD + (((ca >> 3) ^ tr_sw) << 3) + (ca & 7);
    b_off[i] = tr_row * TN_DLD + (((cb >> 3) ^ tr_sw) << 3) + (cb & 7);
  }
  // column sums of A (bias gradient): thread owns logical chunk cs_chunk of rows (tid>>4) + 16*it
  const bool do_cs = out.cslab != nullptr && t2 == 0;
  const int cs_row = tid >> 4;
  const int cs_pc = tid & 15;
  const int cs_chunk = cs_pc ^ ((cs_row & 3) << 2);
  float csum[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  const int n_full = (m_end - m_begin) > 0 ? (m_end - m_begin) / TN_BKM : 0;
  const int rem = (m_end - m_begin) > 0 ? (m_end - m_begin) - n_full * TN_BKM : 0;
  const int n_tiles = n_full + (rem ? 1 : 0);
  // ragged tile loader (registers, zero fill): thread -> row tid>>4 + 16*it, physical chunk tid&15
  auto stage_ragged = [&](int buf, int mt) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = (tid >> 4) + 16 * it;
      const int m = mt + row;
      const int lc = (tid & 15) ^ ((row & 3) << 2);
      uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
      if (m < m_end) {
        if (r0 + lc * 8 < out.N1) va = *reinterpret_cast<const uint4*>(A + map_row(amap, m) * lda + r0 + lc * 8);
        if (c0 + lc * 8 < out.N2) vb = *reinterpret_cast<const uint4*>(B + map_row(bmap, m) * ldb + c0 + lc * 8);
      }
      *reinterpret_cast<uint4*>(As + buf * TN_BKM * TN_DLD + row * TN_DLD + (tid & 15) * 8) = va;
      *reinterpret_cast<uint4*>(Bs + buf * TN_BKM * TN_DLD + row * TN_DLD + (tid & 15) * 8) = vb;
    }
  };
  if (n_tiles > 0) {
    if (n_full > 0) { TN_STAGE_DMA(0, m_begin); } else { stage_ragged(0, m_begin); }
  }
  __syncthreads();
  for (int ti = 0; ti < n_tiles; ++ti) {
    const int buf = ti & 1;
    if (ti + 1 < n_tiles) {
      const int mt = m_begin + (ti + 1) * TN_BKM;
      if (ti + 1 < n_full) { TN_STAGE_DMA(buf ^ 1, mt); } else { stage_ragged(buf ^ 1, mt); }
    }
    const bf16raw* Ab = As + buf * TN_BKM * TN_DLD;
    const bf16raw* Bb = Bs + buf * TN_BKM * TN_DLD;
    if (do_cs) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint4 v = *reinterpret_cast<const uint4*>(Ab + (cs_row + 16 * it) * TN_DLD + cs_pc * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          csum[2 * j] += __uint_as_float(w[j] << 16);
          csum[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < TN_BKM / 16; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16raw* pa = Ab + ks * 16 * TN_DLD + a_off[i];
        const bf16raw* pb = Bb + ks * 16 * TN_DLD + b_off[i];
        union { bf16x8 v; s16x4 h[2]; } ua, ub;
        ua.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
        ua.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa + 4 * TN_DLD));
        ub.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
        ub.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb + 4 * TN_DLD));
        af[i] = ua.v; bfr[i] = ub.v;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#undef TN_STAGE_DMA
  if (do_cs) {
    float* red = reinterpret_cast<float*>(smem);            // [16][128]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[cs_row * 128 + cs_chunk * 8 + j] = csum[j];
    __syncthreads();
    if (tid < 128 && r0 + tid < out.N1) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += red[r * 128 + tid];
      out.cslab[(long)split * out.slab_stride + r0 + tid] = a;
    }
    __syncthreads();
  }
  float* stage = reinterpret_cast<float*>(smem) + wave * 64 * STAGE_LD;
  stage_acc(stage, acc, lane);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  tn_store(out, stage, split, r0 + wm * 64, c0 + wn * 64, lane);
}

// ---- bf16, 256x128 output tile, LDS-DMA ring with counted vmcnt ----------------------------
// Mirror of gemm_nt's <4,3,32> ring: 8 waves (4x2, each 64x64), 32 token rows per stage
// (A [32][256] + B [32][128] bf16 = 24 KB), 3 stages = 72 KB -> two workgroups per CU.  Chunk
// swizzle ^((row&3)<<2) on the DMA source side keeps the transpose reads conflict free.
// Transpose reads are issued as inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS read it emits itself
// while LDS-DMA requests are outstanding (it cannot prove the read does not alias the DMA destination), which drains
// the whole lookahead at the top of every load section -- measured: DMA-only 483 us + MFMA-only 486 us = 1047 us for
// 150528x768x3072 instead of overlapping.  The price: the compiler does not count these reads either, so every load
// section ends in an explicit lgkmcnt(0) and the fragments are pinned behind it (TP_PIN) before the MFMAs use them.
typedef __attribute__((address_space(3))) char tn_lds_char;
template <int OFF> __device__ inline s16x4 tn_tr_read(unsigned addr) {
  s16x4 d;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
  return d;
}
template <int N> __device__ inline void tn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
constexpr int TR_BKM = 32, TR_NBUF = 3, TR_A_LD = 256, TR_B_LD = 128;
constexpr int TR_STAGE = TR_BKM * (TR_A_LD + TR_B_LD);      // elements per stage

__global__ __launch_bounds__(512, 4) void gemm_tn_bf16_ring_kernel(
    int M, int m_per_split, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap,
    const bf16raw* __restrict__ B, long ldb, vtx_rowmap bmap, int tiles2, int tiles12, TnOut out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* ring = reinterpret_cast<bf16raw*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);          // split-major runs per XCD (see the pp kernel)
  const int split = bid / tiles12;
  const int tile = bid - split * tiles12;
  const int t1 = tile / tiles2, t2 = tile - t1 * tiles2;
  const int r0 = t1 * 256, c0 = t2 * 128;
  const int m_begin = split * m_per_split;
  const int m_end = min(M, m_begin + m_per_split);

  // DMA: A piece = 2 rows x 512 B (lane -> row lane>>5, physical chunk lane&31), 2 pieces per wave;
  //      B piece = 4 rows x 256 B (lane -> row lane>>4, physical chunk lane&15), 1 piece per wave.
  const int a_prow = lane >> 5;                       // row within the A piece; piece p covers rows 2p, 2p+1
  const int b_prow = lane >> 4;
  int a_col[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 2 + a_prow;
    const int lc = (lane & 31) ^ ((row & 3) << 2);
    a_col[j] = (r0 + lc * 8) < out.N1 ? r0 + lc * 8 : 0;
  }
  const int b_row = wave * 4 + b_prow;
  const int b_lc = (lane & 15) ^ ((b_row & 3) << 2);
  const int b_col = (c0 + b_lc * 8) < out.N2 ? c0 + b_lc * 8 : 0;
  auto stage = [&](int buf, int mt) {
    bf16raw* Ab = ring + buf * TR_STAGE;
    bf16raw* Bb = Ab + TR_BKM * TR_A_LD;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave * 2 + j) * 2 + a_prow;
      tn_dma16(A + map_row(amap, mt + row) * lda + a_col[j], Ab + (wave * 2 + j) * 512);
    }
    tn_dma16(B + map_row(bmap, mt + b_row) * ldb + b_col, Bb + wave * 512);
  };
  // ragged tile (registers, zero fill)
  auto stage_ragged = [&](int buf, int mt) {
    bf16raw* Ab = ring + buf * TR_STAGE;
    bf16raw* Bb = Ab + TR_BKM * TR_A_LD;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = (tid >> 5) + 16 * it;
      const int lc = (tid & 31) ^ ((row & 3) << 2);
      uint4 va = make_uint4(0, 0, 0, 0);
      if (mt + row < m_end && r0 + lc * 8 < out.N1)
        va = *reinterpret_cast<const uint4*>(A + map_row(amap, mt + row) * lda + r0 + lc * 8);
      *reinterpret_cast<uint4*>(Ab + row * TR_A_LD + (tid & 31) * 8) = va;
    }
    {
      const int row = tid >> 4;
      const int lc = (tid & 15) ^ ((row & 3) << 2);
      uint4 vb = make_uint4(0, 0, 0, 0);
      if (mt + row < m_end && c0 + lc * 8 < out.N2)
        vb = *reinterpret_cast<const uint4*>(B + map_row(bmap, mt + row) * ldb + c0 + lc * 8);
      *reinterpret_cast<uint4*>(Bb + row * TR_B_LD + (tid & 15) * 8) = vb;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int tr_sw = ((lane & 15) >> 2) << 2;
  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = wm * 64 + i * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int cb = wn * 64 + i * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    a_off[i] = tr_row * TR_A_LD + (((ca >> 3) ^ tr_sw) << 3) + (ca & 7);
    b_off[i] = tr_row * TR_B_LD + (((cb >> 3) ^ tr_sw) << 3) + (cb & 7);
  }
  const unsigned lds_b = (unsigned)(unsigned long)(tn_lds_char*)smem;
  // bias gradient: the tiles2 workgroups of a row tile read the same A rows; each sums the stages ti % tiles2 == t2
  // into its own partial copy (a single column tile doing all of it finishes last and holds the launch: 12 % on 3072x768)
  const bool want_cs = out.cslab != nullptr;
  int cs_next = t2;
  const int cs_row = tid >> 5, cs_pc = tid & 31;                  // rows cs_row, cs_row+16 (same row&3)
  const int cs_chunk = cs_pc ^ ((cs_row & 3) << 2);
  float csum[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  const int span = m_end - m_begin;
  const int n_full = span > 0 ? span / TR_BKM : 0;
  const int n_tiles = n_full + ((span > 0 && span % TR_BKM) ? 1 : 0);
  // prologue: tiles 0 .. NBUF-2.  A ragged tile (always the last one) goes through registers.
#pragma unroll
  for (int s = 0; s < TR_NBUF - 1; ++s)
    if (s < n_full) stage(s, m_begin + s * TR_BKM);
  int buf = 0;
  for (int ti = 0; ti < n_tiles; ++ti) {
    if (ti >= n_full) {                                                  // only ti == n_tiles-1; its slot was drained
      stage_ragged(buf, m_begin + ti * TR_BKM);
      __builtin_amdgcn_s_waitcnt(0xc07f);                                // ds_writes visible before the raw barrier
    }
    if (ti + TR_NBUF - 2 < n_full) tn_wait_vmcnt<3 * (TR_NBUF - 2)>(); else tn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    {
      const int nxt = ti + TR_NBUF - 1;
      int nbuf = buf + TR_NBUF - 1; if (nbuf >= TR_NBUF) nbuf -= TR_NBUF;
      if (nxt < n_full) stage(nbuf, m_begin + nxt * TR_BKM);
    }
    const unsigned stage_b = lds_b + 2u * (unsigned)(buf * TR_STAGE);
    const bool do_cs = want_cs && ti == cs_next;
    if (do_cs) {
      cs_next += tiles2;
      u32x4 v_[2];
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v_[0]), "=&v"(v_[1]) : "v"(stage_b + 2u * (cs_row * TR_A_LD + cs_pc * 8)) : "memory");
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          csum[2 * j] += __uint_as_float(v_[it][j] << 16);
          csum[2 * j + 1] += __uint_as_float(v_[it][j] & 0xffff0000u);
        }
    }
    // per 16-deep k step: 8 transpose reads (inline asm: see tn_tr_read), an explicit wait that pins the fragments, 4 MFMAs
#define TR_KSTEP(KA_, KB_)                                                                               \
    {                                                                                                    \
      bf16x8 af[2], bfr[2];                                                                              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                    \
        const unsigned pa = stage_b + 2u * a_off[i], pb = stage_b + 2u * (TR_BKM * TR_A_LD + b_off[i]);  \
        union { bf16x8 v; s16x4 h[2]; } u_;                                                              \
        u_.h[0] = tn_tr_read<KA_>(pa); u_.h[1] = tn_tr_read<KA_ + 2048>(pa); af[i] = u_.v;               \
        u_.h[0] = tn_tr_read<KB_>(pb); u_.h[1] = tn_tr_read<KB_ + 1024>(pb); bfr[i] = u_.v;              \
      }                                                                                                  \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bfr[0]), "+v"(bfr[1]));       \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)        \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);        \
    }
    TR_KSTEP(0, 0)
    TR_KSTEP(8192, 4096)
#undef TR_KSTEP
    if (++buf == TR_NBUF) buf = 0;
  }
  __syncthreads();
  if (want_cs) {
    float* red = reinterpret_cast<float*>(smem);            // [16][256]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[cs_row * 256 + cs_chunk * 8 + j] = csum[j];
    __syncthreads();
    if (tid < 256 && r0 + tid < out.N1) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += red[r * 256 + tid];
      out.cslab[(long)split * out.slab_stride + (long)t2 * out.N1 + r0 + tid] = a;
    }
    __syncthreads();
  }
  float* stg = reinterpret_cast<float*>(smem) + wave * 32 * STAGE_LD;
  float* dst = out.slab + (long)split * out.slab_stride;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    stage_acc_half(stg, acc[mi], lane);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int e = 0; e < 4; ++e) {
      const int rw = e * 8 + (lane >> 3);
      const int r = r0 + wm * 64 + mi * 32 + rw, c = c0 + wn * 64 + (lane & 7) * 8;
      if (r >= out.N1 || c >= out.N2) continue;
      const float* sp = stg + rw * STAGE_LD + (lane & 7) * 8;
      float* d = dst + (long)r * out.N2 + c;
      *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(sp);
      *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(sp + 4);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- bf16, 256x256 output tile, two staggered wave groups (mirror of gemm_nt's pp kernel) -------
// out[256 n1][256 n2] partial over a span of token rows.  8 waves as 2(n1) x 4(n2), 128x64 per wave,
// groups wr = wave>>2 one barrier apart; a K tile = 64 token rows in four phases
// (A0 x B0 | A0 x B1 | A1 x B1 | A1 x B0, 8 MFMAs each).  LDS regions are 2 x [64 token rows][64 columns]
// (A0 = columns wr*128 + [0,64) of both wave rows, A1 the other halves; B0 = columns [0,128), B1 = columns
// [128,256) of the tile, wave column wc owning wc*32 + [0,32) of each: every DMA row segment is whole 128-B lines), 16-B chunks XOR-swizzled with ((row>>1)&1)<<2 on the DMA
// source side; fragments come out through `ds_read_b64_tr_b16` (the reduction index is the LDS row).
// Same region-wise lookahead / vmcnt / barrier protocol as gemm_nt_bf16_pp_kernel (see there).
// A ragged last tile is fetched with clamped row indices and the rows beyond the span are zeroed in
// LDS by the wave that fetched them (right after its vmcnt wait, before the publishing barrier).
// Bias gradient: every workgroup sums the A regions over the token rows (two ds_read_b128 + 16 adds per thread and
// region, in the load sections) of the K tiles kt % tiles2 == t2 -- the tiles2 workgroups of one row tile read the same
// A rows, so each takes a 1/tiles2 share into its own partial copy instead of column tile 0 doing it all and finishing
// last (one resident round: the launch ends with its slowest workgroup; measured 8 % on 768x3072).
constexpr int TP_BK = 64, TP_THREADS = 512;
constexpr int TP_REGION = TP_BK * 128;           // elements
constexpr int TP_BUF = 4 * TP_REGION;            // [A0 | B0 | B1 | A1]
constexpr int TP_RING_BYTES = 2 * TP_BUF * 2;    // 131072
constexpr int TP_STG_LD = 64;
constexpr int TP_LDS_BYTES = TP_RING_BYTES + 8 * 16 * TP_STG_LD * 4;

__device__ inline void tp_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <bool TRACE>
__global__ __launch_bounds__(TP_THREADS, 2) void gemm_tn_bf16_pp_kernel(
    int M, int m_per_split, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap,
    const bf16raw* __restrict__ B, long ldb, vtx_rowmap bmap, int tiles2, int tiles12, TnOut out, long long* __restrict__ trace) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* lds = reinterpret_cast<bf16raw*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // phase timeline of K tiles 8..15 (tools/tn_timeline.py): shader-clock stamps of wave 0 at 13 points per K tile
  int trace_kt = -1;
  auto stamp = [&](int e) {
    if constexpr (TRACE) {
      if (tid == 0 && trace_kt >= 0)
        trace[((long)blockIdx.x * 8 + trace_kt) * 24 + e] = (long long)__builtin_amdgcn_s_memtime();
    }
  };
  // XCD-aware order: the tiles of one split read the same token rows, so give every XCD a contiguous
  // run of the split-major work list (its L2 then serves each A / B row to all the tiles that need it;
  // round-robin placement re-fetches every panel once per XCD: 1.85 GB instead of 0.39 GB at 768x3072)
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid / tiles12;
  const int tile = bid - split * tiles12;
  const int t1 = tile / tiles2, t2 = tile - t1 * tiles2;
  const int r0 = t1 * 256, c0 = t2 * 256;
  const int m_begin = split * m_per_split;
  const int m_end = (split == (int)(gridDim.x / tiles12) - 1) ? M : m_begin + m_per_split;   // last split: + remainder
  const int span = m_end - m_begin;
  const int nk = (span + TP_BK - 1) / TP_BK;      // >= 2 (host guarantees m_per_split >= 128)
  const int last_valid = span - (nk - 1) * TP_BK; // rows of the last tile that are inside the span

  // A region is two sub-blocks of [64 token rows][64 columns] (128-B rows), one per column half.
  // DMA: piece p = 8 token rows x 128 B of sub-block p/8 (the same request shape as gemm_nt's: one full
  // line per row); this wave owns pieces 2*wave, 2*wave+1.  lane -> row 8*(p%8) + lane/8, physical
  // chunk lane%8, logical chunk = physical ^ (((row>>1)&1)<<2): rows r, r+2 swap 64-B halves, which puts
  // the four rows a transpose read touches in one LDS cycle on four disjoint bank quarters.
  // The wave's two pieces are rows prow0 and prow0 + 8 of the same sub-block (2*wave is even, so 2*wave + 1 does not
  // leave the sub-block) and (prow >> 1) & 1 is the same for both: ONE column offset per operand serves both pieces
  // and both region halves (+64 for A1, +128 for B1).
  const int pc0 = wave * 2, sub = pc0 >> 3;
  const int prow0 = (pc0 & 7) * 8 + (lane >> 3);
  const int rc = sub * 64 + ((lane & 7) ^ (((prow0 >> 1) & 1) << 2)) * 8;   // region column 0..127
  const int acol_l = (rc >> 6) * 128 + (rc & 63);        // lane part of the A column (tile origin r0 is added as a scalar); A1: + 64
  // B column = c0 + rc (B1: + 128): wave wc owns columns wc*32+[0,32) of each half
  // per-lane byte offsets of piece 0 from the (scalar) address of the K tile's first row; piece 1 = 8 rows further
  const unsigned voff_a = 2u * (unsigned)(prow0 * (int)lda + acol_l), voff_b = 2u * (unsigned)(prow0 * (int)ldb + rc);
  // Row maps without a division per DMA: phys(m) = base + m + (m / grp) * skip.  Every region kind is requested for K
  // tiles 0, 1, 2, ... in order, so the state of its NEXT request lives in SGPRs and is stepped by one tile after each
  // request (`advance`, pure SALU, called from the following load section): the scalar byte address of the tile's first
  // row (column origin included), how many of its 64 rows come before the next row-map group (`crs`, >= 64: all) and
  // how many are inside M (`vld`).  The request itself (`issue`, which sits between the MFMAs of an MMA section) is
  // branch-free: per piece a lane picks one of two scalar row addresses (this group / the next one) and one of two
  // constant offsets (its own row / row 0 of the tile for rows beyond M, which are zeroed in LDS afterwards) -- about a
  // dozen VALU instructions per region.  It used to be ~100 instructions of per-lane 64-bit address arithmetic and
  // branches, which made every MMA section 2.3x as long as its 8 MFMAs (tools/tn_timeline.py: 600 cycles).
  int trm[4], crs[4], vld[4];
  const char* sp[4];
  const long step64_a = 2L * TP_BK * lda, step64_b = 2L * TP_BK * ldb;          // bytes per K tile
  const long skipb_a = 2L * amap.skip * lda, skipb_b = 2L * bmap.skip * ldb;    // bytes per row-map group crossing
  constexpr int TP_FAR = 1 << 28;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool isA = k == 0 || k == 3;
    const vtx_rowmap& mp = isA ? amap : bmap;
    const int g = mp.grp;
    const int q = g > 0 ? m_begin / g : 0;
    trm[k] = g > 0 ? m_begin - q * g : 0;
    crs[k] = g > 0 ? g - trm[k] : TP_FAR;
    vld[k] = M - m_begin;
    const long phys0 = (long)mp.base + m_begin + (long)q * mp.skip;
    sp[k] = reinterpret_cast<const char*>(isA ? A + phys0 * lda + r0 + (k == 3 ? 64 : 0) : B + phys0 * ldb + c0 + (k == 2 ? 128 : 0));
  }
  auto issue_piece = [&](int kind, int kt, int j) {   // kind: 0 = A0, 1 = B0, 2 = B1, 3 = A1; piece j of this wave
    bf16raw* dst = lds + (kt & 1) * TP_BUF + kind * TP_REGION + wave * 1024;
    const bool isA = kind == 0 || kind == 3;
    const char* lo = sp[kind];
    const char* hi = lo + (isA ? skipb_a : skipb_b);
    const unsigned vo = isA ? voff_a : voff_b, vc = 2u * (unsigned)(isA ? acol_l : rc);
    const unsigned step8 = 16u * (unsigned)(isA ? lda : ldb);
    const int row = prow0 + 8 * j;
    // Regular K tile (all 64 rows before the next row-map group and inside M -- a scalar test; all but one tile in
    // ~24 of the spatial / temporal row maps and every tile of the identity map): scalar base + the lane's constant
    // offset, i.e. the request is `s_mov m0` + one saddr-form DMA instruction and fits an MFMA gap.  The general
    // form below is a dozen dependent vector instructions per piece: two of them per MMA section stretched it by a
    // third (tools/tn_timeline.py: 500 cycles for 8 MFMAs).
    if (crs[kind] >= TP_BK && vld[kind] >= TP_BK) {
      // (inline asm: written with the builtin, hipcc merges this branch with the general one below and the request is
      // again a per-lane 64-bit address built from a dozen instructions)
      const unsigned m0v = (unsigned)(unsigned long)(tn_lds_char*)(dst + j * 512);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :: "v"(vo + j * step8), "s"(lo), "s"(m0v) : "memory", "m0");
      return;
    }
    const char* base = row >= crs[kind] ? hi : lo;
    const unsigned o = row < vld[kind] ? vo + j * step8 : vc;
    tn_dma16(reinterpret_cast<const bf16raw*>(base + o), dst + j * 512);
  };
  auto issue = [&](int kind, int kt) { issue_piece(kind, kt, 0); issue_piece(kind, kt, 1); };
  auto advance = [&](int kind) {                  // step the request state of `kind` to its next K tile
    const bool isA = kind == 0 || kind == 3;
    const int g = isA ? amap.grp : bmap.grp;
    sp[kind] += isA ? step64_a : step64_b;
    vld[kind] -= TP_BK;
    if (g > 0) {
      trm[kind] += TP_BK;
      if (trm[kind] >= g) { trm[kind] -= g; sp[kind] += isA ? skipb_a : skipb_b; }
      crs[kind] = g - trm[kind];
    }
  };
  auto zero_tail = [&](int kt) {                  // rows >= last_valid of tile kt (this wave's own pieces)
#pragma unroll
    for (int kind = 0; kind < 4; ++kind)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (prow0 + 8 * j >= last_valid)
          *reinterpret_cast<uint4*>(lds + (kt & 1) * TP_BUF + kind * TP_REGION + wave * 1024 + j * 512 + lane * 8) =
              make_uint4(0, 0, 0, 0);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed fragment addressing inside a region: token row ks*16 + tr_row (+4), column c
  const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int tr_sw = ((lane >> 3) & 1) << 2;                  // ((row>>1)&1)<<2 for rows tr_row, tr_row+4 (+16 ks)
  const int cl = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);   // column of this lane inside a 32-column group
  int fa_off[2], fb_off;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int cc = i * 32 + cl;                              // column inside sub-block wr
    fa_off[i] = wr * 4096 + tr_row * 64 + (((cc >> 3) ^ tr_sw) << 3) + (cc & 7);
  }
  {
    const int cc = (wc & 1) * 32 + cl;                       // column inside sub-block wc>>1
    fb_off = (wc >> 1) * 4096 + tr_row * 64 + (((cc >> 3) ^ tr_sw) << 3) + (cc & 7);
  }
  const unsigned lds_b = (unsigned)(unsigned long)(tn_lds_char*)smem;
  const unsigned fa_addr[2] = {lds_b + 2u * fa_off[0], lds_b + 2u * fa_off[1]}, fb_addr = lds_b + 2u * fb_off;
  // column sums: thread -> sub-block tid>>8, physical chunk tid&7 of rows (tid>>3)&31 and +32
  const bool want_cs = out.cslab != nullptr;
  bool do_cs = false;                              // this K tile is one of ours
  int cs_next = t2;                                // next K tile whose A regions this workgroup sums
  const int cs_sub = tid >> 8, cs_row = (tid >> 3) & 31, cs_pc = tid & 7;
  float cs0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cs1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TP_COLSUM(buf_, kind_, cs_)                                                                      \
  if (do_cs) {                                                                                           \
    u32x4 v_[2];                                                                                         \
    /* the address is rebuilt from the thread index here: kept live across the loop it is spilled, and a scratch reload \
       waits for vmcnt(0), i.e. for the whole DMA look-ahead */                                          \
    unsigned t_ = threadIdx.x;                                                                           \
    asm volatile("" : "+v"(t_));                                                                         \
    const unsigned ca_ = lds_b + 2u * ((t_ >> 8) * 4096 + ((t_ >> 3) & 31) * 64 + (t_ & 7) * 8);        \
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096\n\ts_waitcnt lgkmcnt(0)"        \
                 : "=&v"(v_[0]), "=&v"(v_[1]) : "v"(ca_ + 2u * ((buf_) * TP_BUF + (kind_) * TP_REGION)) : "memory"); \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) _Pragma("unroll") for (int j = 0; j < 4; ++j) {     \
      cs_[2 * j] += __uint_as_float(v_[it][j] << 16);                                                    \
      cs_[2 * j + 1] += __uint_as_float(v_[it][j] & 0xffff0000u);                                        \
    }                                                                                                    \
  }
#define TP_READ_A(buf_, kind_)                                                                           \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
    const unsigned a_ = fa_addr[i] + 2u * ((buf_) * TP_BUF + (kind_) * TP_REGION);                       \
    union { bf16x8 v; s16x4 h[2]; } u_;                                                                  \
    u_.h[0] = tn_tr_read<0>(a_);    u_.h[1] = tn_tr_read<512>(a_);         fa[i][0] = u_.v;              \
    u_.h[0] = tn_tr_read<2048>(a_); u_.h[1] = tn_tr_read<2048 + 512>(a_);  fa[i][1] = u_.v;              \
    u_.h[0] = tn_tr_read<4096>(a_); u_.h[1] = tn_tr_read<4096 + 512>(a_);  fa[i][2] = u_.v;              \
    u_.h[0] = tn_tr_read<6144>(a_); u_.h[1] = tn_tr_read<6144 + 512>(a_);  fa[i][3] = u_.v;              \
  }
#define TP_READ_B(buf_, kind_, fb_)                                                                      \
  {                                                                                                      \
    const unsigned a_ = fb_addr + 2u * ((buf_) * TP_BUF + (kind_) * TP_REGION);                          \
    union { bf16x8 v; s16x4 h[2]; } u_;                                                                  \
    u_.h[0] = tn_tr_read<0>(a_);    u_.h[1] = tn_tr_read<512>(a_);         fb_[0] = u_.v;                \
    u_.h[0] = tn_tr_read<2048>(a_); u_.h[1] = tn_tr_read<2048 + 512>(a_);  fb_[1] = u_.v;                \
    u_.h[0] = tn_tr_read<4096>(a_); u_.h[1] = tn_tr_read<4096 + 512>(a_);  fb_[2] = u_.v;                \
    u_.h[0] = tn_tr_read<6144>(a_); u_.h[1] = tn_tr_read<6144 + 512>(a_);  fb_[3] = u_.v;                \
  }
#define TP_READ_B_Q(buf_, kind_, fb_, Q_)                                                                \
  {                                                                                                      \
    const unsigned a_ = fb_addr + 2u * ((buf_) * TP_BUF + (kind_) * TP_REGION);                          \
    union { bf16x8 v; s16x4 h[2]; } u_;                                                                  \
    u_.h[0] = tn_tr_read<(Q_) * 2048>(a_); u_.h[1] = tn_tr_read<(Q_) * 2048 + 512>(a_); fb_[Q_] = u_.v;  \
  }
#define TP_PIN_A() asm volatile("" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]),      \
                                     "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]))
#define TP_PIN_B(fb_) asm volatile("" : "+v"(fb_[0]), "+v"(fb_[1]), "+v"(fb_[2]), "+v"(fb_[3]))
// 8 MFMAs with a hook after each of the first six: the two DMA pieces of the section's request and the four fragments
// of a B prefetch are spread one per gap, so that no gap holds more than the ~8 issue slots one MFMA (32 cycles)
// covers -- everything in one lump after the second MFMA left the matrix pipe idle for the length of the lump.
#define TP_MFMA1(i_, ks_, i0_, j_, fb_)                                                                  \
  acc[(i0_) + (i_)][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i_][ks_], fb_[ks_], acc[(i0_) + (i_)][j_], 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
#ifndef VTX_TP_PRIO
// Wave priority around the MMA sections.  Round 5, process-level A/B on the step (tools/micro/lib_ab.sh, two rounds, same box): with
// `s_setprio 1` around every MMA section (rounds 2 - 4) the weight-gradient launches sum to 32.20 / 32.30 ms per step, without any
// priority 32.01 / 32.01, static priority for the younger group 32.42 / 32.35, the LOAD sections at 1 32.23 / 32.18 -- here the load
// section (16 transpose reads) is the longer side of the ping-pong and prioritising the partner's MFMAs only delays it.  (The NT kernel
// is the other way round: no priority costs it 5 %, 478 vs 454 us per launch.)
#define VTX_TP_PRIO 0   // 0 = none (default); 1 = s_setprio 1 around every MMA section; 2 = static (the younger group at 1); 3 = the load sections at 1
#endif
#define TP_PRIO_ON() do { if (VTX_TP_PRIO == 1) __builtin_amdgcn_s_setprio(1); else if (VTX_TP_PRIO == 3) __builtin_amdgcn_s_setprio(0); } while (0)
#define TP_PRIO_OFF() do { if (VTX_TP_PRIO == 1) __builtin_amdgcn_s_setprio(0); else if (VTX_TP_PRIO == 3) __builtin_amdgcn_s_setprio(1); } while (0)
#define TP_MMA(i0_, j_, fb_, H0_, H1_, H2_, H3_, H4_, H5_)                                               \
  TP_PRIO_ON();                                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                     \
  TP_MFMA1(0, 0, i0_, j_, fb_) H0_; __builtin_amdgcn_sched_barrier(0);                                   \
  TP_MFMA1(1, 0, i0_, j_, fb_) H1_; __builtin_amdgcn_sched_barrier(0);                                   \
  TP_MFMA1(0, 1, i0_, j_, fb_) H2_; __builtin_amdgcn_sched_barrier(0);                                   \
  TP_MFMA1(1, 1, i0_, j_, fb_) H3_; __builtin_amdgcn_sched_barrier(0);                                   \
  TP_MFMA1(0, 2, i0_, j_, fb_) H4_; __builtin_amdgcn_sched_barrier(0);                                   \
  TP_MFMA1(1, 2, i0_, j_, fb_) H5_; __builtin_amdgcn_sched_barrier(0);                                   \
  TP_MFMA1(0, 3, i0_, j_, fb_)                                                                           \
  TP_MFMA1(1, 3, i0_, j_, fb_)                                                                           \
  TP_PRIO_OFF();
#define TP_BAR() __builtin_amdgcn_s_barrier()

  bf16x8 fa[2][4], fb0[4], fb1[4];
  issue(0, 0); advance(0); issue(1, 0); advance(1); issue(2, 0); advance(2); issue(3, 0); advance(3);
  issue(0, 1); advance(0); issue(1, 1); advance(1); issue(2, 1); advance(2);
  // Waits are per region and counted (2 DMA instructions per region; issue order per tile A0 B0 B1 A1, then in the loop
  // A1(kt+1) in P1, A0(kt+2) in P2, B0(kt+2) in P3, B1(kt+2) in P4).  The other group runs ONE BARRIER behind, so a
  // region is complete only two barriers after this wave's own wait: every wait sits a full phase ahead of the first
  // read -- and the B fragments prefetched inside an MMA section are first read one barrier earlier than a load section
  // would read them, so B1(kt+1) is waited for in P4(kt), A0 / B0(kt+1) in P3(kt), A1(kt) in P2(kt).
  tn_wait_vmcnt<8>();                             // A0(0), B0(0), B1(0) landed; A1(0) A0(1) B0(1) B1(1) in flight
  TP_BAR();
  if (wr == 1) TP_BAR();
  if (VTX_TP_PRIO == 2 && wr == 1) __builtin_amdgcn_s_setprio(1);
  if (VTX_TP_PRIO == 3) __builtin_amdgcn_s_setprio(1);
  // The load sections must not outlast the partner group's 8-MFMA section (256 cycles), and a transpose read costs
  // ~14 cycles with four waves reading: 24 reads (A0 + B0) in P1 made every phase load-bound (tools/tn_timeline.py:
  // 4900 cycles per K tile).  The A fragments have to be read where they are (fa is busy in every MMA section), but a
  // B fragment set is free one phase before it is needed: B1 is read during P1's MFMAs (into Y), the NEXT tile's B0
  // during P4's (into the registers B1 just left), so the load sections are 16 / 0 / 16 / 0 reads.  X / Y swap roles
  // every K tile (the loop is unrolled by two).
  // Requests: an LDS-DMA piece between two MFMAs stretches the MMA section by ~60 cycles (8 bare MFMAs + partner: 320,
  // with two pieces 450), and the MMA sections are the critical path (load sections end in a barrier wait); the two
  // read-free load sections P2 / P4 take one region's request each (A1(kt+1), B0(kt+2): same issue order, in front of
  // the section's counted wait), the other two stay in the MFMA gaps of P2 / P4 -- four pieces in one load section
  // made it the longer side (750 cycles), and fragment reads moved INTO the gaps cost ~10 cycles each there (measured:
  // all 48 in gaps 3261 us per layer against 2989).
#define TP_KTILE(kt_, X_, Y_)                                                                            \
  {                                                                                                      \
    const int kt = (kt_);                                                                                \
    const int buf = kt & 1;                                                                              \
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;                                                       \
    do_cs = want_cs && kt == cs_next;                                                                    \
    if (do_cs) cs_next += tiles2;                                                                        \
    if constexpr (TRACE) trace_kt = (kt >= 8 && kt < 16) ? kt - 8 : -1;                                  \
    stamp(0);                                                                                            \
    /* P1: reads A0 (B0 is in X_ already); B1(kt) -- waited for in P4 of the previous tile -- is read during the MFMAs */ \
    if (kt >= 1 && n1) advance(2);                   /* B1(kt+1) was requested in P4 of the previous tile */ \
    TP_READ_A(buf, 0);                                                                                   \
    TP_COLSUM(buf, 0, cs0);                                                                              \
    tp_lgkm0();                                                                                          \
    stamp(1);                                                                                            \
    stamp(2);                                                                                            \
    TP_BAR();                                                                                            \
    stamp(3);                                                                                            \
    TP_PIN_A(); TP_PIN_B(X_);                                                                            \
    TP_MMA(0, 0, X_, , ,                                                                                 \
           TP_READ_B_Q(buf, 2, Y_, 0), TP_READ_B_Q(buf, 2, Y_, 1), TP_READ_B_Q(buf, 2, Y_, 2), TP_READ_B_Q(buf, 2, Y_, 3)); \
    stamp(15);                                                                                           \
    TP_BAR();                                                                                            \
    stamp(4);                                                                                            \
    /* P2: no reads (B1 arrived in Y_ during P1); P3 will read A1(kt) */                                 \
    if (n1) { issue(3, kt + 1); advance(3); }        /* A1(kt+1): requested here, not between the MFMAs (see above) */ \
    tp_lgkm0();                                                                                          \
    stamp(5);                                                                                            \
    if (n1) tn_wait_vmcnt<8>(); else tn_wait_vmcnt<0>();                                                 \
    stamp(6);                                                                                            \
    TP_BAR();                                                                                            \
    stamp(7);                                                                                            \
    TP_PIN_B(Y_);                                                                                        \
    TP_MMA(0, 1, Y_, if (n2) issue_piece(0, kt + 2, 0), if (n2) issue_piece(0, kt + 2, 1), , , , );      \
    stamp(16);                                                                                           \
    TP_BAR();                                                                                            \
    stamp(8);                                                                                            \
    /* P3: reads A1 */                                                                                   \
    if (n2) advance(0);                                                                                  \
    TP_READ_A(buf, 3);                                                                                   \
    TP_COLSUM(buf, 3, cs1);                                                                              \
    if (n2) {                                                                                            \
      tn_wait_vmcnt<6>();                            /* A0(kt+1), B0(kt+1) landed */                     \
    } else if (n1) {                                                                                     \
      tn_wait_vmcnt<0>();                            /* all of the last tile */                          \
      if (last_valid < TP_BK) zero_tail(kt + 1);                                                         \
    }                                                                                                    \
    tp_lgkm0();                                                                                          \
    stamp(9);                                                                                            \
    TP_BAR();                                                                                            \
    stamp(10);                                                                                           \
    TP_PIN_A();                                                                                          \
    TP_MMA(2, 1, Y_, , , , , , );                                                                        \
    stamp(17);                                                                                           \
    TP_BAR();                                                                                            \
    stamp(11);                                                                                           \
    /* P4: no reads of its own; B0(kt+1) (published by P3's barrier) is read into Y_ during the MFMAs */ \
    if (n2) {                                                                                            \
      issue(1, kt + 2); advance(1);                  /* B0(kt+2): this section has no reads, see above */ \
      tn_wait_vmcnt<6>();                            /* B1(kt+1) landed */                               \
    }                                                                                                    \
    stamp(12);                                                                                           \
    TP_BAR();                                                                                            \
    stamp(13);                                                                                           \
    TP_MMA(2, 0, X_, if (n2) issue_piece(2, kt + 2, 0), if (n2) issue_piece(2, kt + 2, 1),              \
           if (n1) TP_READ_B_Q(buf ^ 1, 1, Y_, 0), if (n1) TP_READ_B_Q(buf ^ 1, 1, Y_, 1),               \
           if (n1) TP_READ_B_Q(buf ^ 1, 1, Y_, 2), if (n1) TP_READ_B_Q(buf ^ 1, 1, Y_, 3));              \
    stamp(18);                                                                                           \
    TP_BAR();                                                                                            \
    stamp(14);                                                                                           \
  }
  TP_READ_B(0, 1, fb0);                            // B0 of K tile 0 (published by the barrier above); covered by P1's lgkmcnt(0)
  for (int kt2 = 0; kt2 < nk; kt2 += 2) {
    TP_KTILE(kt2, fb0, fb1)
    if (kt2 + 1 < nk) TP_KTILE(kt2 + 1, fb1, fb0)
  }
#undef TP_KTILE
  if (wr == 0) TP_BAR();
  if (VTX_TP_PRIO >= 2) __builtin_amdgcn_s_setprio(0);
#undef TP_COLSUM
#undef TP_READ_B_Q
#undef TP_MFMA1
#undef TP_PIN_A
#undef TP_PIN_B
#undef TP_READ_A
#undef TP_READ_B
#undef TP_MMA
#undef TP_BAR
  if (want_cs) {
    // logical chunk of this thread = cs_pc ^ (((cs_row>>1)&1)<<2) (rows cs_row, cs_row+32 share bit 1);
    // sub-block s of A0 = tile columns s*128 + [0,64), of A1 = s*128 + 64 + [0,64)
    float* red = reinterpret_cast<float*>(smem);            // [32][256] floats = 32 KB (ring is free)
    const int tc = cs_sub * 128 + (cs_pc ^ (((cs_row >> 1) & 1) << 2)) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[cs_row * 256 + tc + j] = cs0[j];
      red[cs_row * 256 + tc + 64 + j] = cs1[j];
    }
    __syncthreads();
    if (tid < 256 && r0 + tid < out.N1) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) a += red[r * 256 + tid];
      out.cslab[(long)split * out.slab_stride + (long)t2 * out.N1 + r0 + tid] = a;
    }
    __syncthreads();
  }
  float* stg = reinterpret_cast<float*>(smem + TP_RING_BYTES) + wave * 16 * TP_STG_LD;
  float* dst = out.slab + (long)split * out.slab_stride;
#define TP_EPI(mi_, half_)                                                                               \
  {                                                                                                      \
    const int col = lane & 31, rhalf = (lane >> 5) * 4;                                                  \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int r = 0; r < 8; ++r)       \
        stg[((r & 3) + 8 * (r >> 2) + rhalf) * TP_STG_LD + ni * 32 + col] = acc[mi_][ni][8 * (half_) + r]; \
    tp_lgkm0();                                                                                          \
    __builtin_amdgcn_wave_barrier();                                                                     \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                      \
      const int rw = e * 8 + (lane >> 3);                                                                \
      const int r = r0 + wr * 128 + (mi_) * 32 + (half_) * 16 + rw;                                      \
      const int c = c0 + ((lane & 7) >> 2) * 128 + wc * 32 + (lane & 3) * 8;                             \
      if (r < out.N1 && c < out.N2) {                                                                    \
        const float* sp = stg + rw * TP_STG_LD + (lane & 7) * 8;                                         \
        float* d = dst + (long)r * out.N2 + c;                                                           \
        *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(sp);                            \
        *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(sp + 4);                    \
      }                                                                                                  \
    }                                                                                                    \
    tp_lgkm0();                                                                                          \
    __builtin_amdgcn_wave_barrier();                                                                     \
  }
  TP_EPI(0, 0) TP_EPI(0, 1) TP_EPI(1, 0) TP_EPI(1, 1) TP_EPI(2, 0) TP_EPI(2, 1) TP_EPI(3, 0) TP_EPI(3, 1)
#undef TP_EPI
}

// ---- bf16, 256x256 output tile, ONE wave per SIMD with 128x128 wave tiles (round 5) --------------
// The ping-pong kernel above needs 1.5 transpose reads per MFMA (a 128x64 wave tile takes 4 A + 2 B fragments per 8
// MFMAs, two `ds_read_b64_tr_b16` each) and its two load sections of 16 reads (~380 cycles) outlast the partner group's
// 8-MFMA section (256): ~3500 cycles per 64-row K tile against the 2048 its 64 MFMAs per SIMD take.  Here
// a workgroup is 4 waves, one per SIMD, each owning a 128x128 block of the same 256x256 tile: 4 A + 4 B fragments per
// 16 MFMAs = 1.0 read per MFMA, all 256 accumulator registers of the wave in the AGPR half of the 512-register file.
// There is no partner wave to hide behind, so the wave pipelines against itself: a K tile is four phases of 16 MFMAs
//     P1  A0 x B0     P2  A0 x B1     P3  A1 x B1     P4  A1 x B0          (regions as in the ping-pong kernel)
// and the 16 transpose reads of the fragment set the NEXT phase needs (B1(kt) | A1(kt) | A0(kt+1) | B0(kt+1)) sit one
// fragment (two reads) per gap behind the first eight MFMAs of the current phase, the four LDS-DMA pieces that refill
// the region the PREVIOUS phase has just released (B0(kt+2) | B1(kt+2) | A1(kt+2) | A0(kt+3): per tile the same A0 B0 B1
// A1 order as above) behind the next four.  Every phase ends with lgkmcnt(0) (the set is complete, the region's last
// reader is done), a counted vmcnt (this wave's pieces of the region the next phase reads have landed: six regions = 24
// requests were issued after it), and ONE barrier that publishes both facts.  Same LDS image, piece shapes, swizzle,
// row-state machine and slab / column-sum outputs as the ping-pong kernel (a wave owns four pieces per region instead
// of two); same products summed in the same order per accumulator: bit-identical weight slabs (the column sums take four
// rows per thread and K tile instead of two: equal to fp32 rounding).
// MEASURED (round 5, interleaved A/B at 96 clips, tools/tn_compare.py, profiles/round5_tn_w4_ab.txt): 6 - 10 % SLOWER than the
// ping-pong kernel on every training shape (768x3072 678 vs 634 us, 2304x768 519 vs 489, 768x768 197 vs 182; in the step 132.0
// vs 129.7 ms).  3600 - 3700 cycles per K tile against the 2048 of its 64 MFMAs: what a lone wave cannot hide is the ISSUE cost
// of the LDS-DMA requests (~60 cycles per 1-KiB piece among MFMAs, MI355X_MICROARCH.md; a K tile is 64 pieces whoever issues
// them: 16 per wave here = ~960 cycles in which this SIMD's matrix pipe has no other client, 8 per wave in the ping-pong kernel
// where the partner wave's MFMAs run meanwhile) plus ~10 cycles per transpose read.  The reads-per-MFMA argument was right and
// beside the point: with LDS-DMA operands the two-waves-per-SIMD structure is what hides the request issue.  Kept as
// gemm_tn=w4 (tested, never the default).
constexpr int TW_THREADS = 256;
constexpr int TW_LDS_BYTES = TP_RING_BYTES + 4 * 16 * TP_STG_LD * 4;

template <bool WANT_CS>
__global__ __launch_bounds__(TW_THREADS, 1) void gemm_tn_bf16_w4_kernel(
    int M, int m_per_split, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap,
    const bf16raw* __restrict__ B, long ldb, vtx_rowmap bmap, int tiles2, int tiles12, TnOut out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* lds = reinterpret_cast<bf16raw*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid / tiles12;
  const int tile = bid - split * tiles12;
  const int t1 = tile / tiles2, t2 = tile - t1 * tiles2;
  const int r0 = t1 * 256, c0 = t2 * 256;
  const int m_begin = split * m_per_split;
  const int m_end = (split == (int)(gridDim.x / tiles12) - 1) ? M : m_begin + m_per_split;   // last split: + remainder
  const int span = m_end - m_begin;
  const int nk = (span + TP_BK - 1) / TP_BK;      // >= 2 (host guarantees m_per_split >= 128)
  const int last_valid = span - (nk - 1) * TP_BK; // rows of the last tile that are inside the span

  // DMA pieces: a region is 16 pieces of 8 token rows x 128 B (piece p: sub-block p / 8, rows 8 * (p % 8) + lane / 8);
  // this wave owns pieces 4 * wave .. 4 * wave + 3 -- one sub-block, rows prow0 + 8 j, the same swizzle bit for all four.
  const int pc0 = wave * 4, sub = pc0 >> 3;
  const int prow0 = (pc0 & 7) * 8 + (lane >> 3);
  const int rc = sub * 64 + ((lane & 7) ^ (((prow0 >> 1) & 1) << 2)) * 8;   // region column 0..127
  const int acol_l = (rc >> 6) * 128 + (rc & 63);
  const unsigned voff_a = 2u * (unsigned)(prow0 * (int)lda + acol_l), voff_b = 2u * (unsigned)(prow0 * (int)ldb + rc);
  int trm[4], crs[4], vld[4];
  const char* sp[4];
  const long step64_a = 2L * TP_BK * lda, step64_b = 2L * TP_BK * ldb;
  const long skipb_a = 2L * amap.skip * lda, skipb_b = 2L * bmap.skip * ldb;
  constexpr int TP_FAR = 1 << 28;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool isA = k == 0 || k == 3;
    const vtx_rowmap& mp = isA ? amap : bmap;
    const int g = mp.grp;
    const int q = g > 0 ? m_begin / g : 0;
    trm[k] = g > 0 ? m_begin - q * g : 0;
    crs[k] = g > 0 ? g - trm[k] : TP_FAR;
    vld[k] = M - m_begin;
    const long phys0 = (long)mp.base + m_begin + (long)q * mp.skip;
    sp[k] = reinterpret_cast<const char*>(isA ? A + phys0 * lda + r0 + (k == 3 ? 64 : 0) : B + phys0 * ldb + c0 + (k == 2 ? 128 : 0));
  }
  auto issue_piece = [&](int kind, int kt, int j) {   // kind: 0 = A0, 1 = B0, 2 = B1, 3 = A1; piece j (0..3) of this wave
    bf16raw* dst = lds + (kt & 1) * TP_BUF + kind * TP_REGION + wave * 2048;
    const bool isA = kind == 0 || kind == 3;
    const char* lo = sp[kind];
    const char* hi = lo + (isA ? skipb_a : skipb_b);
    const unsigned vo = isA ? voff_a : voff_b, vc = 2u * (unsigned)(isA ? acol_l : rc);
    const unsigned step8 = 16u * (unsigned)(isA ? lda : ldb);
    const int row = prow0 + 8 * j;
    if (crs[kind] >= TP_BK && vld[kind] >= TP_BK) {   // regular K tile: scalar base + the lane's constant offset (see the ping-pong kernel)
      const unsigned m0v = (unsigned)(unsigned long)(tn_lds_char*)(dst + j * 512);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :: "v"(vo + j * step8), "s"(lo), "s"(m0v) : "memory", "m0");
      return;
    }
    const char* base = row >= crs[kind] ? hi : lo;
    const unsigned o = row < vld[kind] ? vo + j * step8 : vc;
    tn_dma16(reinterpret_cast<const bf16raw*>(base + o), dst + j * 512);
  };
  auto issue = [&](int kind, int kt) { issue_piece(kind, kt, 0); issue_piece(kind, kt, 1); issue_piece(kind, kt, 2); issue_piece(kind, kt, 3); };
  auto advance = [&](int kind) {
    const bool isA = kind == 0 || kind == 3;
    const int g = isA ? amap.grp : bmap.grp;
    sp[kind] += isA ? step64_a : step64_b;
    vld[kind] -= TP_BK;
    if (g > 0) {
      trm[kind] += TP_BK;
      if (trm[kind] >= g) { trm[kind] -= g; sp[kind] += isA ? skipb_a : skipb_b; }
      crs[kind] = g - trm[kind];
    }
  };
  auto zero_tail = [&](int kt) {                  // rows >= last_valid of tile kt (this wave's own pieces)
#pragma unroll
    for (int kind = 0; kind < 4; ++kind)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (prow0 + 8 * j >= last_valid)
          *reinterpret_cast<uint4*>(lds + (kt & 1) * TP_BUF + kind * TP_REGION + wave * 2048 + j * 512 + lane * 8) =
              make_uint4(0, 0, 0, 0);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed fragment addressing inside a region (as in the ping-pong kernel): token row ks*16 + tr_row (+4), column c
  const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int tr_sw = ((lane >> 3) & 1) << 2;
  const int cl = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const unsigned lds_b = (unsigned)(unsigned long)(tn_lds_char*)smem;
  unsigned fa_addr[2], fb_addr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int cc = i * 32 + cl;                              // column inside the 64-column sub-block
    const int off = tr_row * 64 + (((cc >> 3) ^ tr_sw) << 3) + (cc & 7);
    fa_addr[i] = lds_b + 2u * (wr * 4096 + off);             // A regions: sub-block wr
    fb_addr[i] = lds_b + 2u * (wc * 4096 + off);             // B regions: sub-block wc
  }
  // column sums (bias gradient) of the A regions of the K tiles kt % tiles2 == t2: thread -> sub-block tid >> 7,
  // physical chunk tid & 7 of rows ((tid >> 3) & 15) + {0, 16, 32, 48}
  constexpr bool want_cs = WANT_CS;
  int cs_next = t2;
  const int cs_sub = tid >> 7, cs_row = (tid >> 3) & 15, cs_pc = tid & 7;
  const unsigned cs_addr = lds_b + 2u * (cs_sub * 4096 + cs_row * 64 + cs_pc * 8);
  float cs0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cs1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u32x4 csv[4];
#define TW_CS_READ(buf_, kind_)                                                                          \
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\t" \
               "ds_read_b128 %3, %4 offset:6144"                                                         \
               : "=&v"(csv[0]), "=&v"(csv[1]), "=&v"(csv[2]), "=&v"(csv[3])                             \
               : "v"(cs_addr + 2u * ((buf_) * TP_BUF + (kind_) * TP_REGION)) : "memory")
#define TW_CS_WAIT() { tp_lgkm0(); asm volatile("" : "+v"(csv[0]), "+v"(csv[1]), "+v"(csv[2]), "+v"(csv[3])); }
#define TW_CS_ADD(cs_, q_)                                                                               \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                        \
    cs_[2 * j] += __uint_as_float(csv[q_][j] << 16);                                                     \
    cs_[2 * j + 1] += __uint_as_float(csv[q_][j] & 0xffff0000u);                                         \
  }
// (the region kind and the 16-row step travel in the instruction's 16-bit offset field: one address register per operand
// half and ring buffer instead of one per region, which hipcc hoists out of the loop and spills)
#define TW_FRAG(dst_, addr_, kind_, ks_)                                                                 \
  {                                                                                                      \
    union { bf16x8 v; s16x4 h[2]; } u_;                                                                  \
    u_.h[0] = tn_tr_read<(kind_) * 16384 + (ks_) * 2048>(addr_);                                         \
    u_.h[1] = tn_tr_read<(kind_) * 16384 + (ks_) * 2048 + 512>(addr_); dst_ = u_.v;                      \
  }
#define TW_PIN(S_) asm volatile("" : "+v"(S_[0][0]), "+v"(S_[0][1]), "+v"(S_[0][2]), "+v"(S_[0][3]),     \
                                     "+v"(S_[1][0]), "+v"(S_[1][1]), "+v"(S_[1][2]), "+v"(S_[1][3]))
// the 256 accumulator registers live in the AGPR half of the file: without this the allocator splits their live ranges
// between the two halves and moves whole tuples at the loop header
#define TW_PIN_ACC()                                                                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                       \
    asm volatile("" : "+a"(acc[i_][0]), "+a"(acc[i_][1]), "+a"(acc[i_][2]), "+a"(acc[i_][3]));
#define TW_MF(i2_, j2_, ks_, AS_, i0_, BS_, j0_)                                                         \
  acc[(i0_) + (i2_)][(j0_) + (j2_)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AS_[i2_][ks_], BS_[j2_][ks_], \
                                                                            acc[(i0_) + (i2_)][(j0_) + (j2_)], 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
// 16 MFMAs, one hook behind each: consecutive MFMAs go to different accumulators (an accumulator comes round every
// fourth instruction = 128 cycles, twice its 64-cycle latency)
#define TW_MMA(AS_, i0_, BS_, j0_, H0_, H1_, H2_, H3_, H4_, H5_, H6_, H7_, H8_, H9_, H10_, H11_, H12_, H13_, H14_, H15_) \
  __builtin_amdgcn_sched_barrier(0);                                                                     \
  TW_MF(0, 0, 0, AS_, i0_, BS_, j0_) H0_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(1, 0, 0, AS_, i0_, BS_, j0_) H1_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(0, 1, 0, AS_, i0_, BS_, j0_) H2_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(1, 1, 0, AS_, i0_, BS_, j0_) H3_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(0, 0, 1, AS_, i0_, BS_, j0_) H4_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(1, 0, 1, AS_, i0_, BS_, j0_) H5_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(0, 1, 1, AS_, i0_, BS_, j0_) H6_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(1, 1, 1, AS_, i0_, BS_, j0_) H7_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(0, 0, 2, AS_, i0_, BS_, j0_) H8_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(1, 0, 2, AS_, i0_, BS_, j0_) H9_;  __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(0, 1, 2, AS_, i0_, BS_, j0_) H10_; __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(1, 1, 2, AS_, i0_, BS_, j0_) H11_; __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(0, 0, 3, AS_, i0_, BS_, j0_) H12_; __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(1, 0, 3, AS_, i0_, BS_, j0_) H13_; __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(0, 1, 3, AS_, i0_, BS_, j0_) H14_; __builtin_amdgcn_sched_barrier(0);                            \
  TW_MF(1, 1, 3, AS_, i0_, BS_, j0_) H15_; __builtin_amdgcn_sched_barrier(0);
// end of a phase: the set just read is complete and its region released (lgkmcnt), this wave's pieces of the region the
// next phase reads have landed (six regions were requested after it; in the last two tiles nothing more is requested:
// everything), one barrier publishes both
#define TW_END(n2_)                                                                                      \
  tp_lgkm0();                                                                                            \
  if (n2_) tn_wait_vmcnt<24>(); else tn_wait_vmcnt<0>();                                                 \
  __builtin_amdgcn_s_barrier();

  bf16x8 fa0[2][4], fa1[2][4], fbx[2][4], fby[2][4];
  // prologue: all of K tiles 0 and 1 (the ring's eight slots), then the sets P1(0) multiplies, then A0(2) into the slot
  // A0(0) has just left (the loop continues the request order with B0(2) in P1(0))
  issue(0, 0); advance(0); issue(1, 0); advance(1); issue(2, 0); advance(2); issue(3, 0); advance(3);
  issue(0, 1); advance(0); issue(1, 1); advance(1); issue(2, 1); advance(2); issue(3, 1); advance(3);
  tn_wait_vmcnt<24>();                            // A0(0), B0(0) landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  {
    const unsigned ra0 = fa_addr[0], ra1 = fa_addr[1], rb0 = fb_addr[0], rb1 = fb_addr[1];
    TW_FRAG(fa0[0][0], ra0, 0, 0) TW_FRAG(fa0[1][0], ra1, 0, 0) TW_FRAG(fa0[0][1], ra0, 0, 1) TW_FRAG(fa0[1][1], ra1, 0, 1)
    TW_FRAG(fa0[0][2], ra0, 0, 2) TW_FRAG(fa0[1][2], ra1, 0, 2) TW_FRAG(fa0[0][3], ra0, 0, 3) TW_FRAG(fa0[1][3], ra1, 0, 3)
    TW_FRAG(fbx[0][0], rb0, 1, 0) TW_FRAG(fbx[1][0], rb1, 1, 0) TW_FRAG(fbx[0][1], rb0, 1, 1) TW_FRAG(fbx[1][1], rb1, 1, 1)
    TW_FRAG(fbx[0][2], rb0, 1, 2) TW_FRAG(fbx[1][2], rb1, 1, 2) TW_FRAG(fbx[0][3], rb0, 1, 3) TW_FRAG(fbx[1][3], rb1, 1, 3)
    tp_lgkm0();
    if (want_cs && cs_next == 0) {
      TW_CS_READ(0, 0);
      TW_CS_WAIT();
      TW_CS_ADD(cs0, 0) TW_CS_ADD(cs0, 1) TW_CS_ADD(cs0, 2) TW_CS_ADD(cs0, 3)
    }
    tn_wait_vmcnt<20>();                          // B1(0), which P1(0) reads, has landed too (five regions were requested after it)
    __builtin_amdgcn_s_barrier();                 // ... and every wave has read A0(0): its slot takes A0(2)
    TW_PIN(fa0); TW_PIN(fbx);
    if (nk > 2) { issue(0, 2); advance(0); }
  }
  // X_ holds B0(kt), Y_ receives B1(kt) in P1 and B0(kt + 1) in P4: the two sets swap names every K tile
#define TW_KTILE(kt_, X_, Y_)                                                                            \
  {                                                                                                      \
    const int kt = (kt_);                                                                                \
    const int buf = kt & 1;                                                                              \
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk, n3 = kt + 3 < nk;                                     \
    TW_PIN_ACC();                                                                                        \
    /* P1: A0 x B0; reads B1(kt) -> Y_; requests B0(kt + 2) */                                           \
    {                                                                                                    \
      const unsigned r0_ = fb_addr[0] + 2u * (buf * TP_BUF), r1_ = fb_addr[1] + 2u * (buf * TP_BUF); \
      TW_MMA(fa0, 0, X_, 0,                                                                              \
             TW_FRAG(Y_[0][0], r0_, 2, 0), TW_FRAG(Y_[1][0], r1_, 2, 0), TW_FRAG(Y_[0][1], r0_, 2, 1), TW_FRAG(Y_[1][1], r1_, 2, 1), \
             TW_FRAG(Y_[0][2], r0_, 2, 2), TW_FRAG(Y_[1][2], r1_, 2, 2), TW_FRAG(Y_[0][3], r0_, 2, 3), TW_FRAG(Y_[1][3], r1_, 2, 3), \
             if (n2) issue_piece(1, kt + 2, 0), if (n2) issue_piece(1, kt + 2, 1), if (n2) issue_piece(1, kt + 2, 2), \
             if (n2) issue_piece(1, kt + 2, 3), if (n2) advance(1), , , )                                \
      TW_END(n2)                                                                                         \
      TW_PIN(Y_);                                                                                        \
    }                                                                                                    \
    /* P2: A0 x B1; reads A1(kt) -> fa1 (+ its column sums); requests B1(kt + 2) */                      \
    {                                                                                                    \
      const unsigned r0_ = fa_addr[0] + 2u * (buf * TP_BUF), r1_ = fa_addr[1] + 2u * (buf * TP_BUF); \
      const bool cs_ = want_cs && kt == cs_next;                                                         \
      TW_MMA(fa0, 0, Y_, 2,                                                                              \
             TW_FRAG(fa1[0][0], r0_, 3, 0), TW_FRAG(fa1[1][0], r1_, 3, 0), TW_FRAG(fa1[0][1], r0_, 3, 1), TW_FRAG(fa1[1][1], r1_, 3, 1), \
             TW_FRAG(fa1[0][2], r0_, 3, 2), TW_FRAG(fa1[1][2], r1_, 3, 2), TW_FRAG(fa1[0][3], r0_, 3, 3), TW_FRAG(fa1[1][3], r1_, 3, 3), \
             if (cs_) TW_CS_READ(buf, 3); if (n2) issue_piece(2, kt + 2, 0), if (n2) issue_piece(2, kt + 2, 1),  \
             if (n2) issue_piece(2, kt + 2, 2), if (n2) issue_piece(2, kt + 2, 3),                       \
             if (n2) advance(2); if (cs_) { TW_CS_WAIT() TW_CS_ADD(cs1, 0) }, if (cs_) { TW_CS_ADD(cs1, 1) }, \
             if (cs_) { TW_CS_ADD(cs1, 2) }, if (cs_) { TW_CS_ADD(cs1, 3) })                             \
      if (cs_) cs_next += tiles2;                                                                        \
      if (!n2) {                                   /* the tile after this one is the last: wait for all of it, zero its tail */ \
        tn_wait_vmcnt<0>();                                                                              \
        if (n1 && last_valid < TP_BK) zero_tail(kt + 1);                                                 \
      }                                                                                                  \
      TW_END(n2)                                                                                         \
      TW_PIN(fa1);                                                                                       \
    }                                                                                                    \
    /* P3: A1 x B1; reads A0(kt + 1) -> fa0 (+ its column sums); requests A1(kt + 2) */                  \
    {                                                                                                    \
      const unsigned r0_ = fa_addr[0] + 2u * ((buf ^ 1) * TP_BUF), r1_ = fa_addr[1] + 2u * ((buf ^ 1) * TP_BUF); \
      const bool cs_ = want_cs && n1 && kt + 1 == cs_next;                                               \
      TW_MMA(fa1, 2, Y_, 2,                                                                              \
             TW_FRAG(fa0[0][0], r0_, 0, 0), TW_FRAG(fa0[1][0], r1_, 0, 0), TW_FRAG(fa0[0][1], r0_, 0, 1), \
             TW_FRAG(fa0[1][1], r1_, 0, 1), TW_FRAG(fa0[0][2], r0_, 0, 2), TW_FRAG(fa0[1][2], r1_, 0, 2), \
             TW_FRAG(fa0[0][3], r0_, 0, 3), TW_FRAG(fa0[1][3], r1_, 0, 3),                     \
             if (cs_) TW_CS_READ(buf ^ 1, 0); if (n2) issue_piece(3, kt + 2, 0), if (n2) issue_piece(3, kt + 2, 1), \
             if (n2) issue_piece(3, kt + 2, 2), if (n2) issue_piece(3, kt + 2, 3),                       \
             if (n2) advance(3); if (cs_) { TW_CS_WAIT() TW_CS_ADD(cs0, 0) }, if (cs_) { TW_CS_ADD(cs0, 1) }, \
             if (cs_) { TW_CS_ADD(cs0, 2) }, if (cs_) { TW_CS_ADD(cs0, 3) })                             \
      TW_END(n2)                                                                                         \
      TW_PIN(fa0);                                                                                       \
    }                                                                                                    \
    /* P4: A1 x B0; reads B0(kt + 1) -> Y_; requests A0(kt + 3) */                                       \
    {                                                                                                    \
      const unsigned r0_ = fb_addr[0] + 2u * ((buf ^ 1) * TP_BUF), r1_ = fb_addr[1] + 2u * ((buf ^ 1) * TP_BUF); \
      TW_MMA(fa1, 2, X_, 0,                                                                              \
             TW_FRAG(Y_[0][0], r0_, 1, 0), TW_FRAG(Y_[1][0], r1_, 1, 0), TW_FRAG(Y_[0][1], r0_, 1, 1), \
             TW_FRAG(Y_[1][1], r1_, 1, 1), TW_FRAG(Y_[0][2], r0_, 1, 2), TW_FRAG(Y_[1][2], r1_, 1, 2), \
             TW_FRAG(Y_[0][3], r0_, 1, 3), TW_FRAG(Y_[1][3], r1_, 1, 3),                       \
             if (n3) issue_piece(0, kt + 3, 0), if (n3) issue_piece(0, kt + 3, 1), if (n3) issue_piece(0, kt + 3, 2), \
             if (n3) issue_piece(0, kt + 3, 3), if (n3) advance(0), , , )                                \
      TW_END(n3)                                   /* the sixth request behind B1(kt + 1) is this phase's A0(kt + 3) */ \
      TW_PIN(Y_);                                                                                        \
    }                                                                                                    \
  }
  // (pairs of K tiles, then the odd one: with `if (kt2 + 1 < nk)` around the second tile inside the loop the fragment sets that
  // cross the back-edge come out of a merge and hipcc carries them through scratch)
  int kt2 = 0;
  for (; kt2 + 1 < nk; kt2 += 2) {
    TW_KTILE(kt2, fbx, fby)
    TW_KTILE(kt2 + 1, fby, fbx)
  }
  if (kt2 < nk) TW_KTILE(kt2, fbx, fby)
#undef TW_KTILE
#undef TW_PIN_ACC
#undef TW_END
#undef TW_MMA
#undef TW_MF
#undef TW_PIN
#undef TW_FRAG
  if (want_cs) {
    // logical chunk of this thread = cs_pc ^ (((cs_row >> 1) & 1) << 2) (rows cs_row + 16 k share bit 1);
    // sub-block s of A0 = tile columns s*128 + [0,64), of A1 = s*128 + 64 + [0,64)
    float* red = reinterpret_cast<float*>(smem);            // [16][256] floats = 16 KB (the ring is free: last barrier above)
    const int tc = cs_sub * 128 + (cs_pc ^ (((cs_row >> 1) & 1) << 2)) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[cs_row * 256 + tc + j] = cs0[j];
      red[cs_row * 256 + tc + 64 + j] = cs1[j];
    }
    __syncthreads();
    if (r0 + tid < out.N1) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += red[r * 256 + tid];
      out.cslab[(long)split * out.slab_stride + (long)t2 * out.N1 + r0 + tid] = a;
    }
    __syncthreads();
  }
#undef TW_CS_READ
#undef TW_CS_WAIT
#undef TW_CS_ADD
  float* stg = reinterpret_cast<float*>(smem + TP_RING_BYTES) + wave * 16 * TP_STG_LD;
  float* dst = out.slab + (long)split * out.slab_stride;
  // 16 rows x 64 contiguous columns per pass: rows r0 + wr*128 + mi*32 + half*16 + [0,16), columns c0 + jp*128 + wc*64 + [0,64)
#define TW_EPI(mi_, half_, jp_)                                                                          \
  {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);               /* one pass at a time: hoisted together, the 256 accumulator reads spill */ \
    const int col = lane & 31, rhalf = (lane >> 5) * 4;                                                  \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int r = 0; r < 8; ++r)       \
        stg[((r & 3) + 8 * (r >> 2) + rhalf) * TP_STG_LD + ni * 32 + col] = acc[mi_][2 * (jp_) + ni][8 * (half_) + r]; \
    tp_lgkm0();                                                                                          \
    __builtin_amdgcn_wave_barrier();                                                                     \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                      \
      const int rw = e * 8 + (lane >> 3);                                                                \
      const int r = r0 + wr * 128 + (mi_) * 32 + (half_) * 16 + rw;                                      \
      const int c = c0 + (jp_) * 128 + wc * 64 + (lane & 7) * 8;                                         \
      if (r < out.N1 && c < out.N2) {                                                                    \
        const float* sp_ = stg + rw * TP_STG_LD + (lane & 7) * 8;                                        \
        float* d = dst + (long)r * out.N2 + c;                                                           \
        *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(sp_);                           \
        *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(sp_ + 4);                   \
      }                                                                                                  \
    }                                                                                                    \
    tp_lgkm0();                                                                                          \
    __builtin_amdgcn_wave_barrier();                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
  TW_EPI(0, 0, 0) TW_EPI(0, 1, 0) TW_EPI(1, 0, 0) TW_EPI(1, 1, 0) TW_EPI(2, 0, 0) TW_EPI(2, 1, 0) TW_EPI(3, 0, 0) TW_EPI(3, 1, 0)
  TW_EPI(0, 0, 1) TW_EPI(0, 1, 1) TW_EPI(1, 0, 1) TW_EPI(1, 1, 1) TW_EPI(2, 0, 1) TW_EPI(2, 1, 1) TW_EPI(3, 0, 1) TW_EPI(3, 1, 1)
#undef TW_EPI
}

static int tp_splits(int M, int N1, int N2) {
  const int tiles = cdiv(N1, 256) * cdiv(N2, 256);
  int s = options().tn_cus / tiles;             // one workgroup per CU (option tn_cus: 256 unless the caller reserves CUs, common.h)
  const int max_s = M / 128;                    // every split spans >= 2 K tiles
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : s;
}
static bool tp_eligible(int M, int N1, int N2) { return M >= 4096 && N1 % 256 == 0 && N2 % 256 == 0; }
static bool tp_map_ok(const vtx_rowmap& m) { return m.grp == 0 || m.grp > TP_BK; }

__global__ __launch_bounds__(NT_THREADS) void gemm_tn_f32_kernel(
    int M, int m_per_split, const float* __restrict__ A, long lda, vtx_rowmap amap,
    const float* __restrict__ B, long ldb, vtx_rowmap bmap, int tiles2, int tiles12, TnOut out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);              // [16][128]
  float* Bs = As + TN_BKM32 * TN_LD32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int split = blockIdx.x / tiles12;
  const int tile = blockIdx.x - split * tiles12;
  const int t1 = tile / tiles2, t2 = tile - t1 * tiles2;
  const int r0 = t1 * 128, c0 = t2 * 128;
  const int m_begin = split * m_per_split;
  const int m_end = min(M, m_begin + m_per_split);

  // loader: 16 rows x 32 float4 chunks = 512 chunks -> 2 per thread per operand
  const int lc = tid & 31, lr = tid >> 5;
  const bool a_col_ok = (r0 + lc * 4) < out.N1;
  const bool b_col_ok = (c0 + lc * 4) < out.N2;
  const int a_col = a_col_ok ? r0 + lc * 4 : 0;
  const int b_col = b_col_ok ? c0 + lc * 4 : 0;
  float4 ra[2], rb[2];
  const float4 z4 = make_float4(0, 0, 0, 0);
#define TN_GLOAD32(mt_)                                                                                        \
  {                                                                                                            \
    const int mt__ = (mt_);                                                                                    \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                                         \
      const int m = mt__ + lr + 8 * it;                                                                        \
      const bool ok = m < m_end;                                                                               \
      const int mc = m < M ? m : M - 1;                                                                        \
      ra[it] = *reinterpret_cast<const float4*>(A + map_row(amap, mc) * lda + a_col);                          \
      rb[it] = *reinterpret_cast<const float4*>(B + map_row(bmap, mc) * ldb + b_col);                          \
      if (!(ok && a_col_ok)) ra[it] = z4;                                                                      \
      if (!(ok && b_col_ok)) rb[it] = z4;                                                                      \
    }                                                                                                          \
  }
#define TN_LSTORE32()                                                  \
  {                                                                    \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                 \
      const int off = (lr + 8 * it) * TN_LD32 + lc * 4;                \
      *reinterpret_cast<float4*>(As + off) = ra[it];                   \
      *reinterpret_cast<float4*>(Bs + off) = rb[it];                   \
    }                                                                  \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kh = lane >> 5, col = lane & 31;
  const bool do_cs = out.cslab != nullptr && t2 == 0;
  float csum[4] = {0, 0, 0, 0};
  TN_GLOAD32(m_begin);
  for (int mt = m_begin; mt < m_end; mt += TN_BKM32) {
    TN_LSTORE32();
    if (do_cs) {
#pragma unroll
      for (int it = 0; it < 2; ++it) { csum[0] += ra[it].x; csum[1] += ra[it].y; csum[2] += ra[it].z; csum[3] += ra[it].w; }
    }
    __syncthreads();
    if (mt + TN_BKM32 < m_end) TN_GLOAD32(mt + TN_BKM32);
#pragma unroll
    for (int ks = 0; ks < TN_BKM32 / 2; ++ks) {
      float af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = As[(ks * 2 + kh) * TN_LD32 + wm * 64 + i * 32 + col];
        bfr[i] = Bs[(ks * 2 + kh) * TN_LD32 + wn * 64 + i * 32 + col];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  if (do_cs) {
    float* red = reinterpret_cast<float*>(smem);            // [8][128]
#pragma unroll
    for (int j = 0; j < 4; ++j) red[lr * 128 + lc * 4 + j] = csum[j];
    __syncthreads();
    if (tid < 128 && r0 + tid < out.N1) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) a += red[r * 128 + tid];
      out.cslab[(long)split * out.slab_stride + r0 + tid] = a;
    }
    __syncthreads();
  }
  float* stage = reinterpret_cast<float*>(smem) + wave * 64 * STAGE_LD;
  stage_acc(stage, acc, lane);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  tn_store(out, stage, split, r0 + wm * 64, c0 + wn * 64, lane);
}

// ---- column sums (bias gradients): part[blk][N] then reduce -------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, int rows_per_blk, const T* __restrict__ A,
                                                     long lda, vtx_rowmap amap, float* __restrict__ part) {
  // block = 256 threads: 32 column-chunks (8 elements) x 8 row lanes; grid.x = column groups of 256, grid.y = row blocks
  __shared__ float red[8][256 + 8];
  const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = blockIdx.x * 256 + cc * 8;
  const long m0 = (long)blockIdx.y * rows_per_blk;
  const long m1 = min((long)M, m0 + rows_per_blk);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < N) {
    for (long m = m0 + rl; m < m1; m += 8) {
      float v[8];
      load8(A + map_row(amap, m) * lda + n, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cc * 8 + j] = acc[j];
  __syncthreads();
  const int c = threadIdx.x;
  const int ncol = blockIdx.x * 256 + c;
  if (ncol < N) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) a += red[r][c];
    part[(long)blockIdx.y * N + ncol] = a;
  }
}

static int tn_splits(int M, int N1, int N2) {
  const int tiles = cdiv(N1, 128) * cdiv(N2, 128);
  int s = cdiv(768, tiles);                    // ~3 workgroups per CU over 256 CUs
  const int max_s = M / 256 > 0 ? M / 256 : 1;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}
static int colsum_blocks(int M) {
  int b = cdiv(M, 256);
  return b > 256 ? 256 : (b < 1 ? 1 : b);
}

}  // namespace vtx

using namespace vtx;

extern "C" size_t vtx_gemm_tn_workspace(int M, int N1, int N2) {
  size_t s = (size_t)tn_splits(M, N1, N2);
  if (tp_eligible(M, N1, N2) && (size_t)tp_splits(M, N1, N2) > s) s = (size_t)tp_splits(M, N1, N2);
  // slabs + column-sum slabs (the ring and ping-pong kernels keep one partial copy per column tile)
  return s * (size_t)N1 * (size_t)N2 * sizeof(float) + s * (size_t)N1 * (size_t)cdiv(N2, 128) * sizeof(float);
}

extern "C" int vtx_gemm_tn(const vtx_gemm_tn_desc* d, void* stream) {
  VTX_REQUIRE(d != nullptr, VTX_EINVAL, "gemm_tn: null descriptor");
  VTX_REQUIRE(d->M > 0 && d->N1 > 0 && d->N2 > 0, VTX_EINVAL, "gemm_tn: bad shape");
  VTX_REQUIRE(d->N1 % 8 == 0 && d->N2 % 8 == 0, VTX_EINVAL, "gemm_tn: N1=%d, N2=%d must be multiples of 8", d->N1, d->N2);
  VTX_REQUIRE(d->A && d->B && d->C && d->workspace, VTX_EINVAL, "gemm_tn: null pointer");
  VTX_REQUIRE(d->ldc == d->N2, VTX_EINVAL, "gemm_tn: C must be contiguous (ldc == N2)");
  VTX_REQUIRE(closed_form(d->amap) && closed_form(d->bmap), VTX_EINVAL, "gemm_tn: row maps must be in closed form (no table)");
  VTX_REQUIRE(d->dtype == VTX_F32 || d->dtype == VTX_BF16, VTX_EINVAL, "gemm_tn: bad dtype");
  const long vec = d->dtype == VTX_BF16 ? 8 : 4;
  VTX_REQUIRE(aligned16(d->A) && aligned16(d->B) && aligned16(d->C) && aligned16(d->workspace) &&
                  d->lda % vec == 0 && d->ldb % vec == 0, VTX_EALIGN, "gemm_tn: 16-byte alignment required");
  VTX_REQUIRE(d->ws_bytes >= vtx_gemm_tn_workspace(d->M, d->N1, d->N2), VTX_EWS, "gemm_tn: workspace too small");

  const int splits = tn_splits(d->M, d->N1, d->N2);
  const int tile_m = d->dtype == VTX_BF16 ? TN_BKM : TN_BKM32;
  int m_per = cdiv(d->M, splits);
  m_per = cdiv(m_per, tile_m) * tile_m;
  const int tiles1 = cdiv(d->N1, 128), tiles2 = cdiv(d->N2, 128);
  TnOut out;
  // per-split slab = [N1*N2 weight partials | N1 column-sum partials]
  const long w_elems = (long)d->N1 * d->N2;
  out.slab = (float*)d->workspace; out.slab_stride = w_elems + d->N1; out.N1 = d->N1; out.N2 = d->N2;
  out.cslab = d->colsum ? out.slab + w_elems : nullptr; out.cs_fold = 1;
  dim3 grid(tiles1 * tiles2 * splits), block(NT_THREADS);
  hipStream_t st = as_stream(stream);
  if (d->dtype == VTX_BF16) {
    const size_t need = (size_t)2 * TN_BKM * TN_LD * 2;
    const size_t lds = STAGE_BYTES > need ? STAGE_BYTES : need;
    const Options& o = options();
    const bool safe = o.tn_safe != 0, nodma = o.gemm_nodma != 0;
    const bool want_ring = o.gemm_tn != TN_DMA2 && d->M >= 1024;   // "pp256" falls back to the ring when ineligible
    // Default: the 256x256 ping-pong kernel wherever it is eligible (M >= 4096, N1 and N2 multiples of 256, row-map
    // groups > 64 rows), the 256x128 ring (2 workgroups per CU) otherwise.  tools/tn_compare.py at M = 150528 with the
    // fused bias-gradient sums, ring / ping-pong: 768x3072 974 / 731 us, 3072x768 791 / 708, 2304x768 603 / 551,
    // 768x768 219 / 208.  gemm_tn=pp256 / ring force one of them.
    const bool pp_fits = tp_eligible(d->M, d->N1, d->N2) && tp_map_ok(d->amap) && tp_map_ok(d->bmap);
    const bool want_pp = pp_fits && (o.gemm_tn == TN_PP256 || o.gemm_tn == TN_AUTO);
    if (!safe && !nodma && pp_fits && o.gemm_tn == TN_W4) {
      // one wave per SIMD, 128 x 128 wave tiles: the same tile / slab partition as the ping-pong kernel (bit-identical slabs)
      const int t1p = cdiv(d->N1, 256), t2p = cdiv(d->N2, 256);
      const int s_p = tp_splits(d->M, d->N1, d->N2);
      const int m_per_p = (d->M / s_p) / TP_BK * TP_BK;
      static std::atomic<unsigned long long> attr_set_w{0};
      if (first_launch_on_device(attr_set_w))
      {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_bf16_w4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_bf16_w4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES);
      }
      if (m_per_p >= 2 * TP_BK) {
        out.slab_stride = w_elems + (long)t2p * d->N1; out.cs_fold = t2p;
        if (out.cslab)
          hipLaunchKernelGGL(gemm_tn_bf16_w4_kernel<true>, dim3(t1p * t2p * s_p), dim3(TW_THREADS), TW_LDS_BYTES, st, d->M, m_per_p,
                             (const bf16raw*)d->A, d->lda, d->amap, (const bf16raw*)d->B, d->ldb, d->bmap, t2p, t1p * t2p, out);
        else
          hipLaunchKernelGGL(gemm_tn_bf16_w4_kernel<false>, dim3(t1p * t2p * s_p), dim3(TW_THREADS), TW_LDS_BYTES, st, d->M, m_per_p,
                             (const bf16raw*)d->A, d->lda, d->amap, (const bf16raw*)d->B, d->ldb, d->bmap, t2p, t1p * t2p, out);
        int rc_w = check_launch("gemm_tn_w4");
        if (rc_w) return rc_w;
        if (!d->colsum) return launch_reduce_partials(out.slab, s_p, out.slab_stride, w_elems, d->C, d->accumulate, 1.0f, st);
        return launch_reduce_partials(out.slab, s_p, out.slab_stride, w_elems + d->N1, d->C, d->accumulate, 1.0f, st,
                                      d->colsum, w_elems, d->colsum_accumulate, t2p, d->N1);
      }
    }
    if (!safe && !nodma && want_pp) {
      const int t1p = cdiv(d->N1, 256), t2p = cdiv(d->N2, 256);
      const int s_p = tp_splits(d->M, d->N1, d->N2);
      const int m_per_p = (d->M / s_p) / TP_BK * TP_BK;        // the last split also takes the remainder
      const int s_eff = s_p;
      static std::atomic<unsigned long long> attr_set_p{0};
      if (first_launch_on_device(attr_set_p)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_bf16_pp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TP_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_bf16_pp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TP_LDS_BYTES);
      }
      if (m_per_p >= 2 * TP_BK) {
        out.slab_stride = w_elems + (long)t2p * d->N1; out.cs_fold = t2p;
        long long* trace_p = reinterpret_cast<long long*>(o.pp_trace);
        if (trace_p)                                 // diagnostic instantiation with the phase stamps (tools/tn_timeline.py)
          hipLaunchKernelGGL(gemm_tn_bf16_pp_kernel<true>, dim3(t1p * t2p * s_eff), dim3(TP_THREADS), TP_LDS_BYTES, st, d->M, m_per_p,
                             (const bf16raw*)d->A, d->lda, d->amap, (const bf16raw*)d->B, d->ldb, d->bmap, t2p, t1p * t2p, out, trace_p);
        else
          hipLaunchKernelGGL(gemm_tn_bf16_pp_kernel<false>, dim3(t1p * t2p * s_eff), dim3(TP_THREADS), TP_LDS_BYTES, st, d->M, m_per_p,
                             (const bf16raw*)d->A, d->lda, d->amap, (const bf16raw*)d->B, d->ldb, d->bmap, t2p, t1p * t2p, out, nullptr);
        int rc_p = check_launch("gemm_tn_pp");
        if (rc_p) return rc_p;
        if (!d->colsum) return launch_reduce_partials(out.slab, s_eff, out.slab_stride, w_elems, d->C, d->accumulate, 1.0f, st);
        return launch_reduce_partials(out.slab, s_eff, out.slab_stride, w_elems + d->N1, d->C, d->accumulate, 1.0f, st,
                                      d->colsum, w_elems, d->colsum_accumulate, t2p, d->N1);
      }
      out.slab_stride = w_elems + d->N1; out.cs_fold = 1;
    }
    if (!safe && !nodma && want_ring) {
      const int tiles1r = cdiv(d->N1, 256);
      // one resident round: 2 workgroups per CU x 256 CUs = 512 slots, as full as the workspace allows
      int s_r = 512 / (tiles1r * tiles2);
      const int ws_splits = (tp_eligible(d->M, d->N1, d->N2) && tp_splits(d->M, d->N1, d->N2) > splits)
                                ? tp_splits(d->M, d->N1, d->N2) : splits;   // slabs vtx_gemm_tn_workspace sized
      if (s_r > ws_splits) s_r = ws_splits;
      if (s_r > d->M / (2 * TR_BKM)) s_r = d->M / (2 * TR_BKM);
      if (s_r < 1) s_r = 1;
      int m_per_r = cdiv(cdiv(d->M, s_r), TR_BKM) * TR_BKM;
      const size_t ring_bytes = (size_t)TR_NBUF * TR_STAGE * 2;
      const size_t lds_r = ring_bytes > (size_t)8 * 32 * STAGE_LD * 4 ? ring_bytes : (size_t)8 * 32 * STAGE_LD * 4;
      static std::atomic<unsigned long long> attr_set{0};
      if (first_launch_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_bf16_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
      }
      out.slab_stride = w_elems + (long)tiles2 * d->N1; out.cs_fold = tiles2;
      hipLaunchKernelGGL(gemm_tn_bf16_ring_kernel, dim3(tiles1r * tiles2 * s_r), dim3(512), lds_r, st, d->M, m_per_r,
                         (const bf16raw*)d->A, d->lda, d->amap, (const bf16raw*)d->B, d->ldb, d->bmap, tiles2, tiles1r * tiles2, out);
      int rc_r = check_launch("gemm_tn_ring");
      if (rc_r) return rc_r;
      if (!d->colsum) return launch_reduce_partials(out.slab, s_r, out.slab_stride, w_elems, d->C, d->accumulate, 1.0f, st);
      return launch_reduce_partials(out.slab, s_r, out.slab_stride, w_elems + d->N1, d->C, d->accumulate, 1.0f, st,
                                    d->colsum, w_elems, d->colsum_accumulate, tiles2, d->N1);
    } else if (!safe && !nodma) {
      const size_t need_d = (size_t)4 * TN_BKM * TN_DLD * 2;
      const size_t lds_d = STAGE_BYTES > need_d ? STAGE_BYTES : need_d;
      hipLaunchKernelGGL(gemm_tn_bf16_dma_kernel, grid, block, lds_d, st, d->M, m_per, (const bf16raw*)d->A, d->lda,
                         d->amap, (const bf16raw*)d->B, d->ldb, d->bmap, tiles2, tiles1 * tiles2, out);
    } else if (safe)
      hipLaunchKernelGGL(gemm_tn_bf16_kernel<true>, grid, block, lds, st, d->M, m_per, (const bf16raw*)d->A, d->lda,
                         d->amap, (const bf16raw*)d->B, d->ldb, d->bmap, tiles2, tiles1 * tiles2, out);
    else
      hipLaunchKernelGGL(gemm_tn_bf16_kernel<false>, grid, block, lds, st, d->M, m_per, (const bf16raw*)d->A, d->lda,
                         d->amap, (const bf16raw*)d->B, d->ldb, d->bmap, tiles2, tiles1 * tiles2, out);
  } else {
    const size_t need = (size_t)2 * TN_BKM32 * TN_LD32 * 4;
    const size_t lds = STAGE_BYTES > need ? STAGE_BYTES : need;
    hipLaunchKernelGGL(gemm_tn_f32_kernel, grid, block, lds, st, d->M, m_per, (const float*)d->A, d->lda, d->amap,
                       (const float*)d->B, d->ldb, d->bmap, tiles2, tiles1 * tiles2, out);
  }
  int rc = check_launch("gemm_tn");
  if (rc) return rc;
  if (!d->colsum) return launch_reduce_partials(out.slab, splits, out.slab_stride, w_elems, d->C, d->accumulate, 1.0f, st);
  return launch_reduce_partials(out.slab, splits, out.slab_stride, w_elems + d->N1, d->C, d->accumulate, 1.0f, st,
                                d->colsum, w_elems, d->colsum_accumulate);
}

extern "C" size_t vtx_colsum_workspace(int M, int N) {
  return (size_t)colsum_blocks(M) * (size_t)N * sizeof(float);
}

extern "C" int vtx_colsum(int dtype, int M, int N, const void* A, long lda, vtx_rowmap amap, float* out,
                          int accumulate, void* workspace, size_t ws_bytes, void* stream) {
  VTX_REQUIRE(M > 0 && N > 0 && N % 8 == 0, VTX_EINVAL, "colsum: bad shape M=%d N=%d", M, N);
  VTX_REQUIRE(A && out && workspace, VTX_EINVAL, "colsum: null pointer");
  VTX_REQUIRE(ws_bytes >= vtx_colsum_workspace(M, N), VTX_EWS, "colsum: workspace too small");
  const long vec = dtype == VTX_BF16 ? 8 : 4;
  VTX_REQUIRE(aligned16(A) && lda % vec == 0, VTX_EALIGN, "colsum: 16-byte alignment required");
  const int nb = colsum_blocks(M);
  const int rows_per = cdiv(M, nb);
  dim3 grid(cdiv(N, 256), nb), block(256);
  hipStream_t st = as_stream(stream);
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, st, M, N, rows_per, (const float*)A, lda, amap, (float*)workspace);
  else if (dtype == VTX_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16raw>, grid, block, 0, st, M, N, rows_per, (const bf16raw*)A, lda, amap, (float*)workspace);
  else
    VTX_REQUIRE(false, VTX_EINVAL, "colsum: bad dtype %d", dtype);
  int rc = check_launch("colsum");
  if (rc) return rc;
  return launch_reduce_partials((const float*)workspace, nb, N, N, out, accumulate, 1.0f, st);
}
