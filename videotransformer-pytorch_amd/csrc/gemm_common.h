// gemm_common.h -- pieces shared by the NT (forward / dgrad) and TN (wgrad) GEMMs.
#pragma once
#include "common.h"

namespace vtx {

constexpr int BM = 128, BN = 128;
constexpr int STAGE_LD = 68;                               // floats per staged row (64 + 4 pad)
constexpr int STAGE_BYTES = 4 * 64 * STAGE_LD * 4;         // 69632
constexpr int NT_THREADS = 256;

struct EpiParams {
  int M, N;
  void* C; long ldc; vtx_rowmap cmap;
  const float* bias;
  int act; void* C2; long ldc2;
  const void* dgelu_in; long ld_dgelu; int dgelu_kind;
  const float* row_scale; int rs_d1, rs_m1, rs_d2, rs_m2;
  const void* R; long ldr; vtx_rowmap rmap; int r_period;
  int split_row; void* Csplit; long ldsplit;
  unsigned rs_magic1, rs_shift1, rs_magic2, rs_shift2;   // fast_div constants of rs_d1 / rs_d2 (persistent kernel)
};

// Division of n < 2^31 by a launch constant d >= 1 without the ~25-instruction VALU sequence:
// q = umulhi(n, magic) >> shift with magic = floor(2^(31+s) / d) + 1, s = ceil(log2 d), shift = s - 1 (d = 1: magic 0).
struct FastDiv { unsigned magic, shift; };
inline FastDiv make_fast_div(unsigned d) {
  FastDiv f;
  if (d <= 1) { f.magic = 0; f.shift = 0; return f; }
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.magic = (unsigned)(((1ull << (31 + s)) / d) + 1);
  f.shift = s - 1;
  return f;
}
__device__ inline unsigned fast_div(unsigned n, unsigned magic, unsigned shift) {
  return magic ? (__umulhi(n, magic) >> shift) : n;
}

// Row map restricted to one 256-row tile (first row wave-uniform): the group quotient changes at most once
// inside the tile when grp >= 256, so a row costs a compare and an add instead of a division.
struct TileMap { int grp, skip; long base_q; int bound; bool fast; };
__device__ inline TileMap make_tile_map(const vtx_rowmap& m, int first_row) {
  TileMap t;
  t.grp = m.grp; t.skip = m.skip;
  t.fast = m.grp <= 0 || m.grp >= 256;
  const unsigned q0 = m.grp > 0 ? (unsigned)first_row / (unsigned)m.grp : 0u;
  t.base_q = (long)m.base + (long)q0 * (long)m.skip;
  t.bound = m.grp > 0 ? (int)((q0 + 1) * (unsigned)m.grp) : 0x7fffffff;
  if (m.tab != nullptr) {                        // table form: the offsets of the tile's first group and of the one behind it
    // (the entry behind the last group is never applied -- no row of the tile lies beyond `bound` then -- but is read:
    // the table carries one spare entry)
    // (first_row is wave-uniform; the explicit readfirstlane keeps the offsets in scalar registers: the persistent GEMM
    // builds scalar base addresses from them)
    const int o0 = __builtin_amdgcn_readfirstlane(m.tab[q0]), o1 = __builtin_amdgcn_readfirstlane(m.tab[q0 + 1]);
    t.base_q = (long)m.base + o0;
    t.skip = o1 - o0;
  }
  return t;
}
__device__ inline long tile_map_row(const TileMap& t, const vtx_rowmap& m, int r) {     // r in [first_row, first_row + 256)
  if (t.fast) return t.base_q + r + (r >= t.bound ? (long)t.skip : 0L);
  return map_row(m, r);
}
// the same for a WAVE-UNIFORM row whose result feeds a scalar base address (a table offset comes out of a vector load)
__device__ inline long tile_map_row_u(const TileMap& t, const vtx_rowmap& m, int r) {
  if (t.fast) return t.base_q + r + (r >= t.bound ? (long)t.skip : 0L);
  if (m.tab != nullptr) return (long)m.base + r + (long)__builtin_amdgcn_readfirstlane(m.tab[(unsigned)r / (unsigned)m.grp]);
  return map_row(m, r);
}

// XCD-aware bijective remap of the linear block id (8 XCDs, block b runs on XCD b%8):
// gives every XCD a contiguous run of logical tiles so neighbouring tiles share L2.
__device__ inline int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// Store the wave's 2x2 MFMA 32x32 accumulators (C/D layout: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) into its private fp32 staging tile.
__device__ inline void stage_acc(float* stage, const f32x16 (&acc)[2][2], int lane) {
  const int col = lane & 31, rhalf = (lane >> 5) * 4;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
        stage[row * STAGE_LD + ni * 32 + col] = acc[mi][ni][r];
      }
}

// Half-tile variants (32 rows per pass): 8.7 KB of staging per wave instead of 17.4 KB, for kernels
// whose LDS budget is sized for two co-resident workgroups per CU.
__device__ inline void stage_acc_half(float* stage, const f32x16 (&acc)[2], int lane) {
  const int col = lane & 31, rhalf = (lane >> 5) * 4;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + rhalf) * STAGE_LD + ni * 32 + col] = acc[ni][r];
}

// Fused epilogue over a staged [ROWS x 64] fp32 block: lane -> (row e*8 + lane/8, 8 columns (lane%8)*8).
// Rows are handled four at a time with every global load of the batch (dgelu input, row scale, residual)
// issued before the first use, and the bias -- identical for all rows -- loaded once: a row-at-a-time
// loop pays one full memory latency per row (measured ~8 us per 256x128 tile, 40 % of a K = 768 tile).
// Loads of rows >= M read row M-1 (always valid) instead of branching; only the stores are predicated.
template <typename T, int ROWS = 64, int UBMAX = 4, int LD = STAGE_LD>
__device__ inline void epilogue(const EpiParams& p, const float* stage, int m_base, int n_base, int lane) {
  constexpr int IT = ROWS / 8, UB = IT < UBMAX ? IT : UBMAX;
  const int n = n_base + (lane & 7) * 8;
  if (n >= p.N) return;
  const float* srow = stage + (lane >> 3) * LD + (lane & 7) * 8;
  float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    b8[0] = b0.x; b8[1] = b0.y; b8[2] = b0.z; b8[3] = b0.w; b8[4] = b1.x; b8[5] = b1.y; b8[6] = b1.z; b8[7] = b1.w;
  }
#pragma unroll 1
  for (int e0 = 0; e0 < IT; e0 += UB) {
    int m[UB], ml[UB];
    bool ok[UB], split[UB];
    float v[UB][8];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      m[u] = m_base + (e0 + u) * 8 + (lane >> 3);
      ok[u] = m[u] < p.M;
      ml[u] = ok[u] ? m[u] : p.M - 1;
      split[u] = p.split_row > 0 && ml[u] >= p.split_row;
      const float4 s0 = *reinterpret_cast<const float4*>(srow + (e0 + u) * 8 * LD);
      const float4 s1 = *reinterpret_cast<const float4*>(srow + (e0 + u) * 8 * LD + 4);
      v[u][0] = s0.x + b8[0]; v[u][1] = s0.y + b8[1]; v[u][2] = s0.z + b8[2]; v[u][3] = s0.w + b8[3];
      v[u][4] = s1.x + b8[4]; v[u][5] = s1.y + b8[5]; v[u][6] = s1.z + b8[6]; v[u][7] = s1.w + b8[7];
    }
    if (p.act == 1) {
      if (p.C2) {
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (ok[u]) store8(reinterpret_cast<T*>(p.C2) + (long)m[u] * p.ldc2 + n, v[u]);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = gelu_erf(v[u][j]);
    } else if (p.act == 2) {                     // GELU with its derivative as the second output
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        float gp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) gelu_erf_both(v[u][j], v[u][j], gp[j]);
        if (ok[u]) store8(reinterpret_cast<T*>(p.C2) + (long)m[u] * p.ldc2 + n, gp);
      }
    }
    if (p.dgelu_in) {
      float h[UB][8];
#pragma unroll
      for (int u = 0; u < UB; ++u) load8(reinterpret_cast<const T*>(p.dgelu_in) + (long)ml[u] * p.ld_dgelu + n, h[u]);
      if (p.dgelu_kind == 1) {
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] *= h[u][j];
      } else {
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] *= gelu_erf_grad(h[u][j]);
      }
    }
    if (p.row_scale) {
      float sc[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        sc[u] = p.row_scale[split[u] ? (ml[u] - p.split_row) : (ml[u] / p.rs_d1) * p.rs_m1 + (ml[u] % p.rs_d2) * p.rs_m2];
#pragma unroll
      for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] *= sc[u];
    }
    if (p.R) {
      float r8[UB][8];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        // split rows take no residual: do not form their (out-of-range) residual address, read row 0
        const long rr = split[u] ? 0 : (p.r_period > 0 ? (long)(ml[u] % p.r_period) : map_row(p.rmap, ml[u]));
        load8(reinterpret_cast<const T*>(p.R) + rr * p.ldr + n, r8[u]);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (!split[u]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] += r8[u][j];
        }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (!ok[u]) continue;
      if (split[u]) store8(reinterpret_cast<T*>(p.Csplit) + (long)(m[u] - p.split_row) * p.ldsplit + n, v[u]);
      else store8(reinterpret_cast<T*>(p.C) + map_row(p.cmap, m[u]) * p.ldc + n, v[u]);
    }
  }
}


// One 64-deep K tile of the wave's 64x64 block: 4 k-steps x (2 A + 2 B fragment reads, 4 MFMAs),
// with the fragment reads of step ks+1 issued BEFORE the MFMAs of step ks (register double
// buffer) so LDS latency hides behind the matrix pipe instead of serialising with it.
// Expects in scope: acc[2][2], a_row_off[2], a_sw[2], b_row_off[2], b_sw[2], khalf.
#define VTX_LD_FRAGS_BF16(Ab_, Bb_, ks_, af_, bf_)                                                              \
  {                                                                                                             \
    const int c__ = (ks_) * 2 + khalf;                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                             \
      af_[i] = *reinterpret_cast<const bf16x8*>((Ab_) + a_row_off[i] + ((c__ ^ a_sw[i]) << 3));                 \
      bf_[i] = *reinterpret_cast<const bf16x8*>((Bb_) + b_row_off[i] + ((c__ ^ b_sw[i]) << 3));                 \
    }                                                                                                           \
  }
#define VTX_MMA_TILE_BF16(Ab_, Bb_)                                                                             \
  {                                                                                                             \
    bf16x8 fa__[2][2], fb__[2][2];                                                                              \
    VTX_LD_FRAGS_BF16(Ab_, Bb_, 0, fa__[0], fb__[0]);                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                          \
      if (ks < 3) VTX_LD_FRAGS_BF16(Ab_, Bb_, ks + 1, fa__[(ks + 1) & 1], fb__[(ks + 1) & 1]);                  \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa__[ks & 1][i], fb__[ks & 1][j], acc[i][j], 0, 0, 0); \
    }                                                                                                           \
  }

}  // namespace vtx
