// gemm_nt_w4.hip -- bf16 NT GEMM, 256x256 tile, ONE wave per SIMD (4 waves, 128x128 per wave).
//
// Why: in the 8-wave ping-pong kernel (gemm_nt.hip) the two waves of a SIMD alternate MFMA and load sections between
// barriers; with both operands cache-resident its main loop still takes 1.45 us per 64-deep K tile against 1.0 us of
// MFMA time (tools/gemm_hot.py) -- the on-chip schedule (8 barriers per K tile, 12-fragment load sections that outlast
// the 8-MFMA sections, 0.75 fragment reads per MFMA) costs more than HBM does.  Here a wave owns a 128x128 block:
//   * 256 accumulator registers (the unified 512-entry file of a 1-wave-per-SIMD kernel), 0.5 fragment reads per MFMA;
//   * no MFMA / load role split: each wave streams 16 MFMAs per 16-deep k step with the 8 ds_read_b128 of the NEXT step
//     and its share of the LDS-DMA requests placed between them (sched_group_barrier), so the matrix pipe is fed from one
//     instruction stream and the only workgroup barrier is the one that publishes a freshly landed stage: one per
//     32-deep stage (32 MFMAs per wave);
//   * four 32-deep stages of [A 256x32 | B 256x32] bf16 (32 KB each) in LDS: two stages (64 KB) of requests in flight
//     behind the one being published, waits are counted (vmcnt(16)), never zero in steady state.
// Persistent over the tiles of its XCD with the same atomic tile counters and column-group-major walk as the ping-pong
// kernel.  Epilogue: the generic fused row-vector epilogue of gemm_common.h through a wave-private staging tile.
#include "gemm_common.h"

namespace vtx {

constexpr int W4_BK = 32, W4_NBUF = 4;
constexpr int W4_STAGE = 512 * W4_BK;            // elements per stage: A rows 0..255 then B rows 0..255, 32 wide
constexpr int W4_RING_BYTES = W4_NBUF * W4_STAGE * 2;          // 131072
constexpr int W4_STG_LD = 128 + 4;               // staging row (fp32), padded
constexpr int W4_STG_BYTES = 4 * 16 * W4_STG_LD * 4;           // 4 waves x [16][132] fp32 = 33792
constexpr int W4_LDS_BYTES = W4_RING_BYTES + 64;               // staging reuses the ring after the main loop

template <int N> __device__ inline void w4_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline void w4_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ inline void w4_dma16(const bf16raw* src, bf16raw* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(256, 1) void gemm_nt_bf16_w4_kernel(
    int M, int N, int K, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap, const bf16raw* __restrict__ B, long ldb,
    int tiles_n, int tiles_total, int CG, int* __restrict__ tile_ctr, long long* __restrict__ trace, EpiParams ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* lds = reinterpret_cast<bf16raw*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ns = K / W4_BK;                      // stages per tile (>= 4, host-checked)
  int trace_tile = 0;
  auto stamp = [&](int e) {
    if (trace != nullptr && tid == 0 && trace_tile < 8)
      trace[((long)blockIdx.x * 8 + trace_tile) * 8 + e] = (long long)__builtin_amdgcn_s_memrealtime();
  };

  // ---- tile walk: identical to the ping-pong kernel (per-XCD atomic counter, column-group-major order) ----
  const int xcd = blockIdx.x & 7;
  const int tiles_m = tiles_total / tiles_n;
  const int rlo = (int)((long)tiles_m * xcd / 8), rhi = (int)((long)tiles_m * (xcd + 1) / 8);
  const int nrow = rhi - rlo;
  const int xcount = nrow * tiles_n;
  int* const my_ctr = tile_ctr + xcd * 16;
  auto check_out = [&]() {
    if (tid == 0) {
      __threadfence();
      if (atomicAdd(tile_ctr + 8 * 16, 1) == (int)gridDim.x - 1) {
#pragma unroll
        for (int x = 0; x < 9; ++x) tile_ctr[x * 16] = 0;
        __threadfence();
      }
    }
  };
  if (xcount == 0) { check_out(); return; }
  typedef __attribute__((address_space(3))) int lds_int;
  lds_int* const bcast = (lds_int*)(smem + W4_RING_BYTES);
  auto next_tile = [&]() -> int {
    if (tid == 0) *bcast = atomicAdd(my_ctr, 1);
    w4_lgkm0();
    __builtin_amdgcn_s_barrier();
    const int t = *bcast;
    w4_lgkm0();
    __builtin_amdgcn_s_barrier();
    return __builtin_amdgcn_readfirstlane(t);
  };

  // ---- LDS-DMA sources: per stage a wave requests A pieces 4w..4w+3 and B pieces 4w..4w+3 (16 rows x 64 B each) ----
  const bf16raw* src[8];
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int t) {
    const int grp_tiles = nrow * CG;
    const int g = t / grp_tiles;
    const int wg = min(CG, tiles_n - g * CG);
    const int r = t - g * grp_tiles;
    const int rr = r / wg, cc = r - rr * wg;
    m0 = (rlo + rr) * 256; n0 = (g * CG + cc) * 256;
    const TileMap am = make_tile_map(amap, m0);
    const int m_last = M - 1;
    const long a_last = map_row(amap, m_last);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 16 + (lane >> 2);
      const int c = ((lane & 3) ^ ((row >> 2) & 3)) * 8;
      const int ma = m0 + row;
      int nb = n0 + row;
      if (nb >= N) nb = N - 1;
      src[j] = A + (ma > m_last ? a_last : tile_map_row(am, amap, ma)) * lda + c;
      src[4 + j] = B + (long)nb * ldb + c;
    }
  };
  auto issue_piece = [&](int s, int j) {         // piece j (0..7) of this wave for stage s
    bf16raw* dst = lds + (s & (W4_NBUF - 1)) * W4_STAGE + (j >> 2) * (256 * W4_BK) + (wave * 4 + (j & 3)) * 512;
    w4_dma16(src[j] + s * W4_BK, dst);
  };

  // ---- fragment addresses: row (lane&31) of a 32-row group, chunk (2*ks + lane>>5) ^ ((row>>2)&3) ----
  const int l31 = lane & 31;
  int fr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fr[ks] = l31 * W4_BK + (((2 * ks + (lane >> 5)) ^ ((l31 >> 2) & 3)) << 3);
  const int a_grp = wr * 128 * W4_BK, b_grp = 256 * W4_BK + wc * 128 * W4_BK;

#define W4_READ(dstA_, dstB_, s_, ks_)                                                                     \
  {                                                                                                       \
    const bf16raw* base__ = lds + ((s_) & (W4_NBUF - 1)) * W4_STAGE;                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
      dstA_[i] = *reinterpret_cast<const bf16x8*>(base__ + a_grp + i * 32 * W4_BK + fr[ks_]);             \
      dstB_[i] = *reinterpret_cast<const bf16x8*>(base__ + b_grp + i * 32 * W4_BK + fr[ks_]);             \
    }                                                                                                     \
  }
#define W4_MMA(fa_, fb_)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i], fb_[j], acc[i][j], 0, 0, 0);
  // 16 MFMAs with 8 LDS reads and 4 DMA requests spread between them
#define W4_SCHED(ISSUE_)                                                                                   \
  _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                    \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                    \
  }                                                                                                       \
  if (ISSUE_) {                                                                                           \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                       \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                  \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                  \
    }                                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                    \
  } else {                                                                                                \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                    \
  }

  int t = next_tile();
  const int tend = xcount;
  if (t >= tend) { check_out(); return; }
  while (true) {
    set_tile(t);
    // prologue: stages 0..2 requested; stage 0 published
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) issue_piece(s, j);
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    stamp(0);
    w4_wait_vmcnt<16>();                           // stage 0 landed (this wave's pieces); 1 and 2 in flight
    __builtin_amdgcn_s_barrier();
    stamp(1);
    if (trace != nullptr && tid == 0 && trace_tile < 8) trace[((long)blockIdx.x * 8 + trace_tile) * 8 + 3] = (long long)__builtin_amdgcn_s_memtime();
    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
    W4_READ(fa0, fb0, 0, 0);
    w4_lgkm0();
    // One stage: k step 0 on (fa0, fb0) with the reads of k step 1 and 4 requests of stage s+3 between the MFMAs (its buffer
    // held stage s-1: free since the barrier that published stage s); then stage s+1 is waited for and published; k step 1
    // on (fa1, fb1) with the k-step-0 reads of stage s+1 and the other 4 requests.  ISSUE_/NEXT_ are literals so that the
    // steady-state body is one basic block (sched_group_barrier only orders within one).
#define W4_STAGE_BODY(s_, ISSUE_, WAIT_, NEXT_)                                                            \
    {                                                                                                     \
      W4_READ(fa1, fb1, s_, 1);                                                                           \
      if (ISSUE_) { _Pragma("unroll") for (int j = 0; j < 4; ++j) issue_piece((s_) + 3, j); }             \
      W4_MMA(fa0, fb0);                                                                                   \
      W4_SCHED(ISSUE_);                                                                                   \
      w4_wait_vmcnt<WAIT_>();                                                                             \
      w4_lgkm0();                                                                                         \
      __builtin_amdgcn_s_barrier();                                                                       \
      if (NEXT_) { W4_READ(fa0, fb0, (s_) + 1, 0); }                                                      \
      if (ISSUE_) { _Pragma("unroll") for (int j = 4; j < 8; ++j) issue_piece((s_) + 3, j); }             \
      W4_MMA(fa1, fb1);                                                                                   \
      W4_SCHED(ISSUE_);                                                                                   \
      w4_lgkm0();                                                                                         \
    }
    for (int s = 0; s < ns - 3; ++s) W4_STAGE_BODY(s, true, 12, true)
    W4_STAGE_BODY(ns - 3, false, 8, true)
    W4_STAGE_BODY(ns - 2, false, 0, true)
    W4_STAGE_BODY(ns - 1, false, 0, false)
#undef W4_STAGE_BODY
    stamp(2);
    if (trace != nullptr && tid == 0 && trace_tile < 8) trace[((long)blockIdx.x * 8 + trace_tile) * 8 + 4] = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();                  // every wave is done with the ring: it becomes the epilogue staging
    // ---- epilogue (generic fused row-vector epilogue, 16 rows x 128 columns per pass) ----
    float* stg = reinterpret_cast<float*>(smem) + wave * 16 * W4_STG_LD;
    const int em0 = m0 + wr * 128, en0 = n0 + wc * 128;
#define W4_EPI(mi_, half_)                                                                               \
    {                                                                                                    \
      const int col = lane & 31, rhalf = (lane >> 5) * 4;                                                \
      _Pragma("unroll") for (int nj = 0; nj < 4; ++nj) _Pragma("unroll") for (int r = 0; r < 8; ++r)     \
          stg[((r & 3) + 8 * (r >> 2) + rhalf) * W4_STG_LD + nj * 32 + col] = acc[mi_][nj][8 * (half_) + r]; \
      w4_lgkm0();                                                                                        \
      __builtin_amdgcn_wave_barrier();                                                                   \
      epilogue<bf16raw, 16, 2, W4_STG_LD>(ep, stg, em0 + (mi_) * 32 + (half_) * 16, en0, lane);          \
      epilogue<bf16raw, 16, 2, W4_STG_LD>(ep, stg + 64, em0 + (mi_) * 32 + (half_) * 16, en0 + 64, lane); \
      w4_lgkm0();                                                                                        \
      __builtin_amdgcn_wave_barrier();                                                                   \
    }
    W4_EPI(0, 0) W4_EPI(0, 1) W4_EPI(1, 0) W4_EPI(1, 1) W4_EPI(2, 0) W4_EPI(2, 1) W4_EPI(3, 0) W4_EPI(3, 1)
#undef W4_EPI
    stamp(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // staging (the ring) is free again
    stamp(7);
    ++trace_tile;
    t = next_tile();
    if (t >= tend) break;
  }
  check_out();
#undef W4_READ
#undef W4_MMA
#undef W4_SCHED
}

int launch_gemm_nt_w4(const vtx_gemm_desc* d, const EpiParams& ep, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES);
    attr_set = true;
  }
  const Options& cfg = options();
  const int tiles_m = cdiv(d->M, 256), tiles_n = cdiv(d->N, 256);
  int cg = cfg.pp_cg ? cfg.pp_cg : (int)(6291456L / (512L * d->K));
  if (!cfg.pp_cg && cg < 3) cg = 3;
  if (!cfg.pp_cg && cg > 6) cg = 6;
  if (cg < 1) cg = 1;
  hipLaunchKernelGGL(gemm_nt_bf16_w4_kernel, dim3(cfg.pp_grid), dim3(256), W4_LDS_BYTES, st, d->M, d->N, d->K, (const bf16raw*)d->A,
                     d->lda, d->amap, (const bf16raw*)d->B, d->ldb, tiles_n, tiles_m * tiles_n, cg, (int*)d->workspace,
                     reinterpret_cast<long long*>(cfg.pp_trace), ep);
  return check_launch("gemm_nt_w4");
}

}  // namespace vtx
