// gemm_nt_dual.hip -- bf16 NT GEMM, 256x128 tile, 4 waves (one per SIMD, 128x64 each), TWO workgroups per CU.
//
// Why: the persistent 256x256 kernel of gemm_nt.hip leaves the matrix pipes idle for the ~7 us a tile spends outside its
// main loop (epilogue passes, refill of the operand ring) -- a quarter of a K = 768 tile -- and an in-kernel overlap needs a
// second accumulator set that a 2-wave-per-SIMD kernel does not have registers for.  Here the overlap comes from the
// hardware scheduler: two independent 4-wave workgroups share a CU (256 registers per wave: 128 accumulators, 72 KB of
// LDS each), each wave runs MFMAs, LDS reads and its share of the LDS-DMA requests from one instruction stream, and while
// one workgroup is in its epilogue the other has the matrix pipes to itself.
//   * three 32-deep stages of [A 256x32 | B 128x32] bf16 (24 KB each); one barrier per stage publishes the next one;
//   * waits are counted (vmcnt(3) in steady state: half of stage s+2 stays in flight while stage s+1 is published).
// Persistent over the tiles of its XCD with the atomic tile counters and column-group-major walk of the ping-pong kernel.
#include "gemm_common.h"

namespace vtx {

constexpr int DU_BK = 32, DU_NBUF = 3;
constexpr int DU_STAGE = 384 * DU_BK;            // elements per stage: A rows 0..255 then B rows 0..127, 32 wide
constexpr int DU_RING_BYTES = DU_NBUF * DU_STAGE * 2;          // 73728
constexpr int DU_STG_LD = 64 + 4;                // staging row (fp32), padded
constexpr int DU_LDS_BYTES = DU_RING_BYTES + 64;               // the 17 KB epilogue staging reuses the ring

template <int N> __device__ inline void du_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline void du_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ inline void du_dma16(const bf16raw* src, bf16raw* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_dual_kernel(
    int M, int N, int K, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap, const bf16raw* __restrict__ B, long ldb,
    int tiles_n, int tiles_total, int CG, int* __restrict__ tile_ctr, long long* __restrict__ trace, EpiParams ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* lds = reinterpret_cast<bf16raw*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ns = K / DU_BK;                      // stages per tile (>= 3, host-checked)
  int trace_tile = 0;
  auto stamp = [&](int e) {
    if (trace != nullptr && tid == 0 && trace_tile < 8 && blockIdx.x < 256)
      trace[((long)blockIdx.x * 8 + trace_tile) * 8 + e] = (long long)__builtin_amdgcn_s_memrealtime();
  };

  // ---- tile walk: as in the ping-pong kernel (per-XCD atomic counter, column-group-major order), 128-wide tiles ----
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
  lds_int* const bcast = (lds_int*)(smem + DU_RING_BYTES);
  auto next_tile = [&]() -> int {
    if (tid == 0) *bcast = atomicAdd(my_ctr, 1);
    du_lgkm0();
    __builtin_amdgcn_s_barrier();
    const int t = *bcast;
    du_lgkm0();
    __builtin_amdgcn_s_barrier();
    return __builtin_amdgcn_readfirstlane(t);
  };

  // ---- LDS-DMA sources: per stage a wave requests A pieces 4w..4w+3 and B pieces 2w, 2w+1 (16 rows x 64 B each) ----
  const bf16raw* src[6];
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int t) {
    const int grp_tiles = nrow * CG;
    const int g = t / grp_tiles;
    const int wg = min(CG, tiles_n - g * CG);
    const int r = t - g * grp_tiles;
    const int rr = r / wg, cc = r - rr * wg;
    m0 = (rlo + rr) * 256; n0 = (g * CG + cc) * 128;
    const TileMap am = make_tile_map(amap, m0);
    const int m_last = M - 1;
    const long a_last = map_row(amap, m_last);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 16 + (lane >> 2);
      const int c = ((lane & 3) ^ ((row >> 2) & 3)) * 8;
      const int ma = m0 + row;
      src[j] = A + (ma > m_last ? a_last : tile_map_row(am, amap, ma)) * lda + c;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave * 2 + j) * 16 + (lane >> 2);
      const int c = ((lane & 3) ^ ((row >> 2) & 3)) * 8;
      int nb = n0 + row;
      if (nb >= N) nb = N - 1;
      src[4 + j] = B + (long)nb * ldb + c;
    }
  };
  auto issue_piece = [&](int buf, int s, int j) {  // piece j (0..5) of this wave for stage s, into ring buffer buf
    bf16raw* dst = lds + buf * DU_STAGE + (j < 4 ? (wave * 4 + j) * 512 : 256 * DU_BK + (wave * 2 + j - 4) * 512);
    du_dma16(src[j] + s * DU_BK, dst);
  };

  // ---- fragment addresses: row (lane&31) of a 32-row group, chunk (2*ks + lane>>5) ^ ((row>>2)&3) ----
  const int l31 = lane & 31;
  int fr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fr[ks] = l31 * DU_BK + (((2 * ks + (lane >> 5)) ^ ((l31 >> 2) & 3)) << 3);
  const int a_grp = wr * 128 * DU_BK, b_grp = 256 * DU_BK + wc * 64 * DU_BK;

#define DU_READ(dstA_, dstB_, buf_, ks_)                                                                   \
  {                                                                                                       \
    const bf16raw* base__ = lds + (buf_) * DU_STAGE;                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
        dstA_[i] = *reinterpret_cast<const bf16x8*>(base__ + a_grp + i * 32 * DU_BK + fr[ks_]);           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
        dstB_[i] = *reinterpret_cast<const bf16x8*>(base__ + b_grp + i * 32 * DU_BK + fr[ks_]);           \
  }
#define DU_MMA(fa_, fb_)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i], fb_[j], acc[i][j], 0, 0, 0);
  // 8 MFMAs with up to 6 LDS reads and 3 DMA requests spread between them
#define DU_SCHED(READS_, ISSUE_)                                                                           \
  if (READS_) {                                                                                           \
    _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                                       \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                  \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                  \
    }                                                                                                     \
  } else {                                                                                                \
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                                    \
  }                                                                                                       \
  if (ISSUE_) {                                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                    \
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                    \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                    \
  } else {                                                                                                \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                    \
  }

  int t = next_tile();
  const int tend = xcount;
  if (t >= tend) { check_out(); return; }
  while (true) {
    set_tile(t);
    // prologue: stages 0 and 1 requested; stage 0 published
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 6; ++j) issue_piece(s, s, j);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    stamp(0);
    du_wait_vmcnt<6>();                            // stage 0 landed (this wave's pieces); stage 1 in flight
    __builtin_amdgcn_s_barrier();
    stamp(1);
    bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
    DU_READ(fa0, fb0, 0, 0);
    du_lgkm0();
    int cur = 0;                                   // ring buffer of stage s
    // One stage: k step 0 on (fa0, fb0) with the reads of k step 1 and 3 requests of stage s+2 between the MFMAs (its buffer
    // held stage s-1: free since the barrier that published stage s); then stage s+1 is waited for and published; k step 1
    // on (fa1, fb1) with the k-step-0 reads of stage s+1 and the other 3 requests.  ISSUE_/NEXT_ are literals so that the
    // steady-state body is one basic block (sched_group_barrier only orders within one).
#define DU_STAGE_BODY(s_, ISSUE_, WAIT_, NEXT_)                                                            \
    {                                                                                                     \
      const int nxt__ = cur == 2 ? 0 : cur + 1, prv__ = cur == 0 ? 2 : cur - 1;                           \
      DU_READ(fa1, fb1, cur, 1);                                                                          \
      if (ISSUE_) { issue_piece(prv__, (s_) + 2, 0); issue_piece(prv__, (s_) + 2, 1); issue_piece(prv__, (s_) + 2, 4); } \
      DU_MMA(fa0, fb0);                                                                                   \
      DU_SCHED(true, ISSUE_);                                                                             \
      if (NEXT_) du_wait_vmcnt<WAIT_>();                                                                  \
      du_lgkm0();                                                                                         \
      __builtin_amdgcn_s_barrier();                                                                       \
      if (NEXT_) { DU_READ(fa0, fb0, nxt__, 0); }                                                         \
      if (ISSUE_) { issue_piece(prv__, (s_) + 2, 2); issue_piece(prv__, (s_) + 2, 3); issue_piece(prv__, (s_) + 2, 5); } \
      DU_MMA(fa1, fb1);                                                                                   \
      DU_SCHED(NEXT_, ISSUE_);                                                                            \
      du_lgkm0();                                                                                         \
      cur = nxt__;                                                                                        \
    }
    for (int s = 0; s < ns - 2; ++s) DU_STAGE_BODY(s, true, 3, true)
    DU_STAGE_BODY(ns - 2, false, 0, true)
    DU_STAGE_BODY(ns - 1, false, 0, false)
#undef DU_STAGE_BODY
    stamp(2);
    __builtin_amdgcn_s_barrier();                  // every wave is done with the ring: it becomes the epilogue staging
    // ---- epilogue (generic fused row-vector epilogue, 16 rows x 64 columns per pass) ----
    float* stg = reinterpret_cast<float*>(smem) + wave * 16 * DU_STG_LD;
    const int em0 = m0 + wr * 128, en0 = n0 + wc * 64;
#define DU_EPI(mi_, half_)                                                                               \
    {                                                                                                    \
      const int col = lane & 31, rhalf = (lane >> 5) * 4;                                                \
      _Pragma("unroll") for (int nj = 0; nj < 2; ++nj) _Pragma("unroll") for (int r = 0; r < 8; ++r)     \
          stg[((r & 3) + 8 * (r >> 2) + rhalf) * DU_STG_LD + nj * 32 + col] = acc[mi_][nj][8 * (half_) + r]; \
      du_lgkm0();                                                                                        \
      __builtin_amdgcn_wave_barrier();                                                                   \
      epilogue<bf16raw, 16, 2, DU_STG_LD>(ep, stg, em0 + (mi_) * 32 + (half_) * 16, en0, lane);          \
      du_lgkm0();                                                                                        \
      __builtin_amdgcn_wave_barrier();                                                                   \
    }
    DU_EPI(0, 0) DU_EPI(0, 1) DU_EPI(1, 0) DU_EPI(1, 1) DU_EPI(2, 0) DU_EPI(2, 1) DU_EPI(3, 0) DU_EPI(3, 1)
#undef DU_EPI
    stamp(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // staging (the ring) is free again
    stamp(7);
    ++trace_tile;
    t = next_tile();
    if (t >= tend) break;
  }
  check_out();
#undef DU_READ
#undef DU_MMA
#undef DU_SCHED
}

int launch_gemm_nt_dual(const vtx_gemm_desc* d, const EpiParams& ep, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_dual_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DU_LDS_BYTES);
    attr_set = true;
  }
  const Options& cfg = options();
  const int tiles_m = cdiv(d->M, 256), tiles_n = cdiv(d->N, 128);
  int cg = cfg.pp_cg ? cfg.pp_cg : (int)(6291456L / (512L * d->K));
  if (!cfg.pp_cg && cg < 3) cg = 3;
  if (!cfg.pp_cg && cg > 6) cg = 6;
  if (cg < 1) cg = 1;
  hipLaunchKernelGGL(gemm_nt_bf16_dual_kernel, dim3(2 * cfg.pp_grid), dim3(256), DU_LDS_BYTES, st, d->M, d->N, d->K, (const bf16raw*)d->A,
                     d->lda, d->amap, (const bf16raw*)d->B, d->ldb, tiles_n, tiles_m * tiles_n, 2 * cg, (int*)d->workspace,
                     reinterpret_cast<long long*>(cfg.pp_trace), ep);
  return check_launch("gemm_nt_dual");
}

}  // namespace vtx
