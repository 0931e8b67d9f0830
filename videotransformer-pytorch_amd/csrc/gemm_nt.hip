// gemm_nt.hip -- C[M,N] = epilogue(A[M,K] * B[N,K]^T), both operands K-contiguous.
//
// Serves every nn.Linear forward and input-gradient on the path (reference
// transformer.py:160-175 qkv/proj, :225,267 temporal_fc, :498-507 FFN, the
// Conv2d/Conv3d patch projection :116-126 after vtx_patch_rows, and
// video_transformer.py:855,878 decoder_pred).  Input gradients use the same
// kernel with the transposed weight copy made by vtx_cast_transpose.
//
// MFMA-bound.  128x128 output tile per 256-thread workgroup (4 waves, 2x2, each
// 64x64 = 2x2 MFMA 32x32 tiles).
//   bf16: BK=64, v_mfma_f32_32x32x16_bf16; LDS tiles [128][64] bf16 with the
//         16-B chunk index XOR-swizzled by (row>>1)&7 so ds_read_b128 fragment
//         reads are bank-conflict free; register-staged global loads, double-
//         buffered LDS, one barrier per K tile.
//   fp32: BK=16, v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain); LDS [128][17].
// Epilogue: accumulators are staged through LDS (wave-private 64x68 fp32) so the
// fused bias / GELU / GELU' / DropPath-scale / residual / row-scatter run on
// row-contiguous 8-element vectors, independent of the MFMA register layout.
// Algorithmic FLOPs per launch: 2*M*N*K.
#include "gemm_common.h"

namespace vtx {

// ------------------------------------------------------------------ bf16 kernel
constexpr int BK16 = 64;  // K elements per tile (bf16)

__global__ __launch_bounds__(NT_THREADS) void gemm_nt_bf16_kernel(
    int M, int N, int K, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap,
    const bf16raw* __restrict__ B, long ldb, int tiles_n, EpiParams ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* As = reinterpret_cast<bf16raw*>(smem);                    // [2][128][64]
  bf16raw* Bs = As + 2 * BM * BK16;                                  // [2][128][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // loader assignment: 4 chunks of 16 B per operand per thread; rows tid/8 + 32*it, chunk tid%8
  const int lc = tid & 7, lr = tid >> 3;
  const bf16raw* ap[4];
  const bf16raw* bp[4];
  int lds_off[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = lr + 32 * it;
    int ma = m0 + row; if (ma >= M) ma = M - 1;
    int nb = n0 + row; if (nb >= N) nb = N - 1;
    ap[it] = A + map_row(amap, ma) * lda;
    bp[it] = B + (long)nb * ldb;
    lds_off[it] = row * BK16 + ((lc ^ ((row >> 1) & 7)) << 3);
  }
  uint4 ra[4], rb[4];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  // chunks beyond K read the row's first chunk instead (always valid: K >= 8) and are zeroed.  (They used to read
  // chunk lc of K tile 0, which lies beyond the row -- and, in the last row, beyond the tensor -- when K < 64.)
#define GLOAD16(k0_)                                                                  \
  {                                                                                   \
    const int k0__ = (k0_);                                                           \
    const bool ok = (k0__ + lc * 8) < K;                                              \
    const int ko__ = ok ? k0__ + lc * 8 : 0;                                          \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                \
      ra[it] = *reinterpret_cast<const uint4*>(ap[it] + ko__);                        \
      rb[it] = *reinterpret_cast<const uint4*>(bp[it] + ko__);                        \
      if (!ok) { ra[it] = zero4; rb[it] = zero4; }                                    \
    }                                                                                 \
  }
#define LSTORE16(buf_)                                                                \
  {                                                                                   \
    const int b__ = (buf_);                                                           \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                \
      *reinterpret_cast<uint4*>(As + b__ * BM * BK16 + lds_off[it]) = ra[it];         \
      *reinterpret_cast<uint4*>(Bs + b__ * BN * BK16 + lds_off[it]) = rb[it];         \
    }                                                                                 \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (elements) for this lane; chunk = ks*2 + (lane>>5)
  int a_row_off[2], b_row_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ar = wm * 64 + i * 32 + (lane & 31);
    const int br = wn * 64 + i * 32 + (lane & 31);
    a_row_off[i] = ar * BK16; a_sw[i] = (ar >> 1) & 7;
    b_row_off[i] = br * BK16; b_sw[i] = (br >> 1) & 7;
  }
  const int khalf = lane >> 5;

  const int nk = (K + BK16 - 1) / BK16;
  GLOAD16(0);
  LSTORE16(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) GLOAD16((kt + 1) * BK16);
    const bf16raw* Ab = As + buf * BM * BK16;
    const bf16raw* Bb = Bs + buf * BN * BK16;
    VTX_MMA_TILE_BF16(Ab, Bb);
    if (kt + 1 < nk) LSTORE16(buf ^ 1);
    __syncthreads();
  }
#undef GLOAD16
#undef LSTORE16
  float* stage = reinterpret_cast<float*>(smem) + wave * 64 * STAGE_LD;
  stage_acc(stage, acc, lane);
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): wave-private staging, no barrier needed
  __builtin_amdgcn_wave_barrier();
  epilogue<bf16raw>(ep, stage, m0 + wm * 64, n0 + wn * 64, lane);
}

// ------------------------------------------------------- bf16 kernel, LDS-DMA staging
// Same tile / fragment / epilogue code as above, but the operand tiles go HBM -> LDS directly
// with global_load_lds_dwordx4 (no VGPR round trip, no ds_write).  A wave-instruction lands
// 64 x 16 B = 1 KiB = 8 tile rows at a wave-uniform LDS base in lane order, so the XOR swizzle
// is applied on the SOURCE side: lane (row, physical chunk pc) fetches logical chunk
// pc ^ ((row>>1)&7) of its row -- still one full 128-B line per 8 lanes.  Requires K % 64 == 0.
__device__ inline void dma16(const bf16raw* src, bf16raw* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Result stores of the persistent kernel are nontemporal: a result is never read back by the launch, and as ordinary
// stores the 0.2 - 0.9 GB of it pushed the A panels out of L2 before the other column tiles of the group had re-read
// them.  Same box, back to back: 6114 -> 5911 us per layer over the twelve NT GEMMs (qkv forward 548 -> 520, fc1
// 959 -> 937), 692 -> 700 clips/s in the step.  (The same policy on the LayerNorm and attention outputs measured
// -0.6 % in the step: LayerNorm stand-alone 96 -> 99 / 183 -> 195 us; restricting the policy to results wider than 2048
// columns -- so that the 231-MB ones may stay in the memory-side cache for their consumer -- measured -0.6 % as well.)
#define PP_ST8 store8_nt
// the same request with the nontemporal policy (aux = 2): for blocks that are read exactly once (the epilogue's residual /
// multiplier block), never for operands -- those are re-read out of L2 by the other column tiles
__device__ inline void dma16_nt(const bf16raw* src, bf16raw* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

// A 16-byte nontemporal store at a SCALAR base + a 32-bit lane offset (global_store_dwordx4 voffset, data, saddr).  The base
// passes through an empty asm as in issue() below: hipcc otherwise folds base + offset into a per-lane 64-bit pointer.
__device__ inline void st16_nt_s(const char* sbase, unsigned voff, u32x4 v) {
  asm volatile("" : "+s"(sbase));
  typedef __attribute__((address_space(1))) u32x4 g_u32x4;       // (the empty asm hides the pointer's origin: name the address space,
  __builtin_nontemporal_store(v, (g_u32x4*)(sbase + voff));      //  or the store becomes a FLAT one, which also counts on lgkmcnt)
}
// the same store for a base that was laundered already (under a lane predicate the asm would define a scalar register in
// divergent control flow: "illegal VGPR to SGPR copy")
__device__ inline void st16_nt_b(const char* sbase, unsigned voff, u32x4 v) {
  typedef __attribute__((address_space(1))) u32x4 g_u32x4;
  __builtin_nontemporal_store(v, (g_u32x4*)(sbase + voff));
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ inline f32x2 mk2(float a, float b) { f32x2 r; r[0] = a; r[1] = b; return r; }
__device__ inline u32x4 pack8(const float (&v)[8]) {          // eight floats -> eight bf16 (v_cvt_pk_bf16_f32, round to nearest even)
  union { bf16x8 b; u32x4 u; } o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.b[i] = (__bf16)v[i];
  return o.u;
}

__global__ __launch_bounds__(NT_THREADS) void gemm_nt_bf16_dma_kernel(
    int M, int N, int K, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap,
    const bf16raw* __restrict__ B, long ldb, int tiles_n, EpiParams ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* As = reinterpret_cast<bf16raw*>(smem);                    // [2][128][64]
  bf16raw* Bs = As + 2 * BM * BK16;                                  // [2][128][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging assignment: wave w issues pieces j = 0..3 of each operand; piece p = w*4+j covers
  // tile rows [8p, 8p+8): lane -> row 8p + (lane>>3), physical chunk lane&7
  const bf16raw* ap[4];
  const bf16raw* bp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int ma = m0 + row; if (ma >= M) ma = M - 1;
    int nb = n0 + row; if (nb >= N) nb = N - 1;
    ap[j] = A + map_row(amap, ma) * lda + c * 8;
    bp[j] = B + (long)nb * ldb + c * 8;
  }
#define STAGE_DMA(buf_, k0_)                                                     \
  {                                                                              \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                              \
      dma16(ap[j] + (k0_), As + (buf_) * BM * BK16 + (wave * 4 + j) * 8 * BK16); \
      dma16(bp[j] + (k0_), Bs + (buf_) * BN * BK16 + (wave * 4 + j) * 8 * BK16); \
    }                                                                            \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int a_row_off[2], b_row_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ar = wm * 64 + i * 32 + (lane & 31);
    const int br = wn * 64 + i * 32 + (lane & 31);
    a_row_off[i] = ar * BK16; a_sw[i] = (ar >> 1) & 7;
    b_row_off[i] = br * BK16; b_sw[i] = (br >> 1) & 7;
  }
  const int khalf = lane >> 5;
  const int nk = K / BK16;
  STAGE_DMA(0, 0);
  __syncthreads();                         // carries the vmcnt(0) for the LDS-DMA
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) STAGE_DMA(buf ^ 1, (kt + 1) * BK16);
    const bf16raw* Ab = As + buf * BM * BK16;
    const bf16raw* Bb = Bs + buf * BN * BK16;
    VTX_MMA_TILE_BF16(Ab, Bb);
    __syncthreads();
  }
#undef STAGE_DMA
  float* stage = reinterpret_cast<float*>(smem) + wave * 64 * STAGE_LD;
  stage_acc(stage, acc, lane);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  epilogue<bf16raw>(ep, stage, m0 + wm * 64, n0 + wn * 64, lane);
}

// ------------------------------------------- bf16 kernel, LDS-DMA ring with counted vmcnt
// (WM*64) x 128 output tile, WM x 2 waves (each 64x64), NBUF-deep ring of BK-deep K tiles staged
// by LDS-DMA.  Tile kt+NBUF-1 is issued while tile kt is consumed; a wave waits only for ITS OWN
// oldest tile with a counted `s_waitcnt vmcnt(PW*(NBUF-2))` and the workgroup meets at ONE raw
// s_barrier per K tile (no vmcnt(0) drain), so NBUF-1 tiles of memory latency are in flight per
// workgroup.  Measured on the 2-buffer kernel above: one full memory latency (~1.2 us) per K tile.
//   <4,3,64>: 144 KB of LDS, one workgroup (8 waves) per CU.
//   <4,3,32>:  72 KB, TWO workgroups per CU, so one's prologue / epilogue (measured ~8 us per
//              tile, ~40 % of a K=768 tile) overlaps the other's main loop.
// LDS rows are BK elements; 16-B chunk swizzle: BK=64 -> ^((row>>1)&7), BK=32 -> ^((row>>2)&3).
template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BK> __device__ inline int ring_sw(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

template <int WM, int NBUF, int BK>
__global__ __launch_bounds__(WM * 128, (BK == 32 ? 1024 : 512) / (WM * 128) * (WM * 128) / 256) void gemm_nt_bf16_ring_kernel(
    int M, int N, int K, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap,
    const bf16raw* __restrict__ B, long ldb, int tiles_n, EpiParams ep) {
  constexpr int NW = WM * 2;                 // waves
  constexpr int RBM = WM * 64;               // tile rows
  constexpr int RPP = 512 / BK;              // tile rows per 1-KiB DMA piece
  constexpr int LPR = BK / 8;                // lanes (16-B chunks) per row
  constexpr int APW = RBM / RPP / NW;        // A pieces per wave per stage
  constexpr int BPW = BN / RPP / NW;         // B pieces per wave per stage
  constexpr int PW = APW + BPW;              // DMA instructions per wave per stage
  constexpr int KS = BK / 16;                // MFMA k-steps per stage
  constexpr int STAGE_ELEMS = (RBM + BN) * BK;
  static_assert(APW >= 1 && BPW >= 1, "tile too small for this many waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* ring = reinterpret_cast<bf16raw*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * RBM, n0 = tn * BN;

  const bf16raw* ap[APW];
  const bf16raw* bp[BPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int row = (wave * APW + j) * RPP + lane / LPR;
    const int c = (lane % LPR) ^ ring_sw<BK>(row);
    int ma = m0 + row; if (ma >= M) ma = M - 1;
    ap[j] = A + map_row(amap, ma) * lda + c * 8;
  }
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int row = (wave * BPW + j) * RPP + lane / LPR;
    const int c = (lane % LPR) ^ ring_sw<BK>(row);
    int nb = n0 + row; if (nb >= N) nb = N - 1;
    bp[j] = B + (long)nb * ldb + c * 8;
  }
  auto stage = [&](int buf, int k0) {
    bf16raw* Ab = ring + buf * STAGE_ELEMS;
    bf16raw* Bb = Ab + RBM * BK;
#pragma unroll
    for (int j = 0; j < APW; ++j) dma16(ap[j] + k0, Ab + (wave * APW + j) * 512);
#pragma unroll
    for (int j = 0; j < BPW; ++j) dma16(bp[j] + k0, Bb + (wave * BPW + j) * 512);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int a_row_off[2], b_row_off[2], a_sw[2], b_sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ar = wm * 64 + i * 32 + (lane & 31);
    const int br = wn * 64 + i * 32 + (lane & 31);
    a_row_off[i] = ar * BK; a_sw[i] = ring_sw<BK>(ar);
    b_row_off[i] = br * BK; b_sw[i] = ring_sw<BK>(br);
  }
  const int khalf = lane >> 5;
  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s)
    if (s < nk) stage(s, s * BK);
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NBUF - 2 < nk) wait_vmcnt<PW * (NBUF - 2)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    {
      const int nxt = kt + NBUF - 1;
      int nbuf = buf + NBUF - 1; if (nbuf >= NBUF) nbuf -= NBUF;
      if (nxt < nk) stage(nbuf, nxt * BK);
    }
    const bf16raw* Ab = ring + buf * STAGE_ELEMS;
    const bf16raw* Bb = Ab + RBM * BK;
    bf16x8 fa[2][2], fb[2][2];
    VTX_LD_FRAGS_BF16(Ab, Bb, 0, fa[0], fb[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) VTX_LD_FRAGS_BF16(Ab, Bb, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
    }
    if (++buf == NBUF) buf = 0;
  }
  __syncthreads();                              // every wave is done with the ring -> reuse it for staging
  float* stg = reinterpret_cast<float*>(smem) + wave * 32 * STAGE_LD;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    stage_acc_half(stg, acc[mi], lane);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    epilogue<bf16raw, 32, (BK == 32 ? 1 : 4)>(ep, stg, m0 + wm * 64 + mi * 32, n0 + wn * 64, lane);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

template <int WM, int NBUF, int BK>
static int launch_ring(const vtx_gemm_desc* d, const EpiParams& ep, hipStream_t st) {
  constexpr int RBM = WM * 64;
  const size_t ring_bytes = (size_t)NBUF * (RBM + BN) * BK * 2;
  const size_t stage_bytes = (size_t)WM * 2 * 32 * STAGE_LD * 4;
  const size_t lds = ring_bytes > stage_bytes ? ring_bytes : stage_bytes;
  static std::atomic<unsigned long long> attr_set{0};
  if (first_launch_on_device(attr_set)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_ring_kernel<WM, NBUF, BK>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int tiles_m = cdiv(d->M, RBM), tiles_n = cdiv(d->N, BN);
  hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<WM, NBUF, BK>), dim3(tiles_m * tiles_n), dim3(WM * 128), lds, st, d->M, d->N,
                     d->K, (const bf16raw*)d->A, d->lda, d->amap, (const bf16raw*)d->B, d->ldb, tiles_n, ep);
  return check_launch("gemm_nt_ring");
}

// ------------------------------------------------------------------ bf16, 256x256 tile, two staggered wave groups
// 8 waves as 2(M) x 4(N), each owning a 128x64 block of the tile (8 accumulators of 32x32).  Waves
// w and w+4 share a SIMD and sit in different groups (wr = w>>2); group 1 runs ONE barrier behind
// group 0, so between any two barriers one wave of every SIMD is in an MFMA section (8 MFMAs =
// 256 matrix-pipe cycles) while its partner is in a load section (LDS fragment reads + 2 LDS-DMA
// pieces) -- the matrix pipe of every SIMD always has exactly one client.
//
// A K tile (64 deep) takes four phases per wave; the wave's block is cut into quadrants
//   P1: read A0 (rows 0..63 of the block), B0 (cols 0..31)   MFMA A0 x B0
//   P2: read B1 (cols 32..63)                                 MFMA A0 x B1
//   P3: read A1 (rows 64..127, reusing A0's registers)        MFMA A1 x B1
//   P4: --                                                    MFMA A1 x B0
// The LDS image of a K tile is grouped the same way -- region A0 holds the "A0 rows" of BOTH wave
// rows, B0 the "B0 columns" of all four wave columns, ... (16 KB each, [128][64] bf16, 16-B chunks
// XOR-swizzled with (row>>1)&7 on the DMA source side) -- so a region is read in exactly one phase
// and can be refilled right after it: two K-tile buffers (128 KB) carry 7 regions of lookahead.
// Phase q of tile t issues region (A1,t+1) / (A0,t+2) / (B0,t+2) / (B1,t+2) for q = 1..4.
//   WAR: the overwritten copy was last read >= 1 full phase earlier and every load section retires
//        its reads (lgkmcnt(0)) BEFORE its barrier, so both groups are past them.
//   RAW: the load section BEFORE the phase that reads a region waits for it with a counted vmcnt that
//        leaves every younger region (up to 5 = 80 KB per CU) in flight, then its barrier publishes it.
constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64, PP_THREADS = 512;
constexpr int PP_REGION = 128 * PP_BK;          // elements per region
constexpr int PP_BUF = 4 * PP_REGION;           // elements per K-tile buffer: [A0 | B0 | B1 | A1]
constexpr int PP_RING_BYTES = 2 * PP_BUF * 2;   // 131072
constexpr int PP_STG_LD = 64;                   // staging rows are unpadded: 8 waves x [16][64] fp32 = 32 KB
constexpr int PP_LDS_BYTES = PP_RING_BYTES + 8 * 16 * PP_STG_LD * 4;   // 163840 = all of the CU's LDS
constexpr int PP_GRID = 256;                    // persistent: one workgroup per CU

__device__ inline void lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Persistent: workgroup b (XCD b%8, slot b/8) walks the tiles of its XCD's contiguous tile range in
// steps of gridDim/8.  The C tile is 128 KB, and a non-persistent launch exposes its write-out
// (measured at K = 768: 345 us with stores, 247 us without, 206 us without the epilogue at all):
// here the next tile's first 7 regions are requested BEFORE the epilogue of the finished tile, the
// epilogue stages through its own 32 KB of LDS, and its stores drain under the next main loop.
// (vmcnt counts loads and stores together; a counted wait can therefore over-wait on stores but never
// under-wait on the DMA, because loads retire in order among themselves.)
// PRE: what the epilogue reads per element: 0 nothing, 1 the residual, 2 the GELU' input x, 3 a plain multiplier
enum { PRE_NONE = 0, PRE_RES = 1, PRE_DGELU = 2, PRE_MUL = 3 };
template <bool EPI2, int PRE, bool HAS_SC, bool HAS_ACT, bool CONT, bool ROLL = false>
__global__ __launch_bounds__(PP_THREADS, 2) void gemm_nt_bf16_pp_kernel(
    int M, int N, int K, const bf16raw* __restrict__ A, long lda, vtx_rowmap amap,
    const bf16raw* __restrict__ B, long ldb, int tiles_n, int tiles_total, int CG, int* __restrict__ tile_ctr,
    long long* __restrict__ trace, int dbg, EpiParams ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16raw* lds = reinterpret_cast<bf16raw*>(smem);
  amap.tab = nullptr;                            // the A map is in closed form (vtx_gemm_nt checks it): no table code below
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  float* stg = reinterpret_cast<float*>(smem + PP_RING_BYTES) + wave * 16 * PP_STG_LD;
  const int nk = K / PP_BK;
  // timeline of the first 8 tiles of every workgroup (tools/pp_timeline.py): 16 words per tile -- 100 MHz ticks at 8
  // points, and the shader-clock counter at points 1 and 2 (main loop start / end: cycles / time = the clock the chip
  // sustains inside the main loop) in words 8 and 9
  int trace_tile = 0;
  auto stamp = [&](int e) {
    if (trace != nullptr && tid == 0 && trace_tile < 8) {
      long long* slot = trace + ((long)blockIdx.x * 8 + trace_tile) * 16;
      slot[e] = (long long)__builtin_amdgcn_s_memrealtime();
      if (e == 1 || e == 2) slot[7 + e] = (long long)__builtin_amdgcn_s_memtime();
    }
  };

  // Tile walk.  XCD x owns the row tiles [rlo, rhi) for ALL column tiles and walks them column-group
  // major: CG column tiles at a time over all its row tiles, so the 32 workgroups of an XCD work on
  // 32/CG row tiles x CG column tiles at any moment: the CG weight panels (CG x 256 x K) are re-read by
  // every row tile while they are hot in the XCD's 4 MB L2 / the Infinity Cache, and every activation
  // line is shared by CG workgroups.
  // (The loop is bound by the ~64 outstanding 128-B misses of a CU's vector L1 times the L2 latency --
  // TCP_PENDING_STALL 40 % of the time, average TCP->TCC read latency 394 cycles at a 45 % L2 miss rate
  // with the old row-major order of 2.7 row tiles x 12 column tiles -- so the L2 hit rate IS the speed.)
  // The workgroups of an XCD draw local tile indices from the XCD's counter (one atomicAdd per tile,
  // broadcast through an LDS word): a static slot -> tile assignment doubles the kernel time as soon as
  // fewer than 256 workgroups are resident, e.g. while an RCCL collective of the data-parallel gradient
  // exchange holds a few CUs.
  const int per = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tiles_m = tiles_total / tiles_n;
  const int rlo = (int)((long)tiles_m * xcd / 8), rhi = (int)((long)tiles_m * (xcd + 1) / 8);
  const int nrow = rhi - rlo;
  const int xcount = nrow * tiles_n;
  int* const my_ctr = tile_ctr + xcd * 16;       // one 64-B line per XCD
  // The counters clean up after themselves: every workgroup checks out through `done` when it leaves, and
  // the last one of the grid zeroes the set for the next launch (stream order makes that visible).
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
  lds_int* const bcast = (lds_int*)(smem + PP_RING_BYTES);              // staging is idle between epilogues
  auto publish = [&](int v) -> int {             // lane 0 of the workgroup -> every wave (all waves are between tiles)
    if (tid == 0) *bcast = v;
    lgkm0();
    __builtin_amdgcn_s_barrier();
    const int t = *bcast;
    lgkm0();
    __builtin_amdgcn_s_barrier();                // the word may be overwritten (epilogue staging) only after all have read it
    return __builtin_amdgcn_readfirstlane(t);
  };
  auto next_tile = [&]() -> int {                // draw and publish in one step: the atomic's round trip is exposed
    int v = 0;
    if (tid == 0) v = atomicAdd(my_ctr, 1);
    return publish(v);
  };
  static_assert(!CONT || EPI2, "the continuous flows belong to the read-ahead epilogue");
  constexpr bool CF = CONT && PRE == PRE_NONE;   // continuous operand flow: the epilogue leaves the ring alone
  constexpr bool PF = CONT && PRE != PRE_NONE;   // residual-block flow: the ring receives the epilogue's block (see below)
  // Request addresses = a wave-uniform 64-bit base per operand (first byte of the tile's A rows / B rows, in scalar
  // registers, advanced by scalar adds) + a 32-bit byte offset per lane and piece: the DMA instruction takes both
  // (saddr + voffset), so a request costs no vector ALU work inside the MFMA sections and the eight per-lane
  // pointers take 8 registers instead of 16.  (vtx_gemm_nt checks that a tile's rows span < 2 GB.)
  const char* abase = nullptr;
  const char* bbase = nullptr;
  unsigned off[4][2];                            // region kind: 0 = A0, 1 = B0, 2 = B1, 3 = A1
  int m0 = 0, n0 = 0;                            // origin of the tile being computed
  int m0s = 0, n0s = 0;                          // origin of the tile set_tile() was last called for
  int par = 0;                                   // CONT: ring buffer of the current tile's K tile 0
  auto set_tile = [&](int t, long koff) {       // t = local tile index of this XCD; bases are biased by -koff bytes
    const int grp_tiles = nrow * CG;             // tiles in a full column group
    const int g = t / grp_tiles;
    const int wg = min(CG, tiles_n - g * CG);    // width of this group (the last one may be narrower)
    const int r = t - g * grp_tiles;
    const int rr = r / wg, cc = r - rr * wg;
    const int tm = rlo + rr, tn = g * CG + cc;
    m0s = tm * PP_BM; n0s = tn * PP_BN;
    const TileMap am = make_tile_map(amap, m0s);
    const int m_last = M - 1;
    const long first = tile_map_row(am, amap, m0s);          // physical row of the tile's first row
    abase = reinterpret_cast<const char*>(A + first * lda) - koff;
    bbase = reinterpret_cast<const char*>(B + (long)n0s * ldb) - koff;
    // Row offsets relative to the tile's first physical row, in 32-bit arithmetic (the launcher guarantees that a
    // tile's rows span < 2 GB).  Row maps with grp >= 256 (or none) cross a group boundary at most once inside a
    // tile: local row l sits at l + (l >= bl ? skip : 0).  The continuous kernels are only launched for those.
    const int ml = min(m_last - m0s, PP_BM - 1);              // last valid local row (ragged last row tile)
    const int bl = am.fast ? am.bound - m0s : 0x7fffffff;    // local row at which the skip starts
    const unsigned lda2 = (unsigned)lda * 2u, ldb2 = (unsigned)ldb * 2u;
    const int nl = min(N - 1 - n0s, PP_BN - 1);               // last valid local column (ragged last column tile)
    // (the lane's row / chunk terms are recomputed per call from an opaque copy of the lane id: hoisted out of the tile loop
    // they are eight more registers live through the main loop -- the ones hipcc spilled first)
    int lq = lane;
    asm volatile("" : "+v"(lq));
#pragma unroll
    for (int j = 0; j < 2; ++j) {                // this wave owns pieces 2*wave, 2*wave+1 (8 region rows each)
      const int rho = (wave * 2 + j) * 8 + (lq >> 3);
      const unsigned c2 = (unsigned)(((lq & 7) ^ ((rho >> 1) & 7)) * 16);
      const int la0 = min((rho >> 6) * 128 + (rho & 63), ml), la1 = min((rho >> 6) * 128 + (rho & 63) + 64, ml);
      const int lb0 = min((rho >> 5) * 64 + (rho & 31), nl), lb1 = min((rho >> 5) * 64 + (rho & 31) + 32, nl);
      unsigned ra0, ra1;
      if (CONT || am.fast) {
        ra0 = (unsigned)(la0 + (la0 >= bl ? am.skip : 0));
        ra1 = (unsigned)(la1 + (la1 >= bl ? am.skip : 0));
      } else {
        ra0 = (unsigned)(map_row(amap, m0s + la0) - first);
        ra1 = (unsigned)(map_row(amap, m0s + la1) - first);
      }
      off[0][j] = ra0 * lda2 + c2;
      off[3][j] = ra1 * lda2 + c2;
      off[1][j] = (unsigned)lb0 * ldb2 + c2;
      off[2][j] = (unsigned)lb1 * ldb2 + c2;
    }
  };
  auto issue = [&](int kind, int kt) {           // region `kind` of K tile kt -> buffer (kt&1)^par
    bf16raw* dst = lds + ((kt & 1) ^ par) * PP_BUF + kind * PP_REGION + wave * 1024;
    const char* base = ((kind == 0 || kind == 3) ? abase : bbase) + (long)kt * (PP_BK * 2);
    asm volatile("" : "+s"(base));               // keep it a scalar base (saddr form): hipcc would otherwise strength-reduce
                                                 // base + offset into 64-bit per-lane pointers again
    dma16(reinterpret_cast<const bf16raw*>(base + off[kind][0]), dst);
    dma16(reinterpret_cast<const bf16raw*>(base + off[kind][1]), dst + 512);
  };
  auto prologue = [&]() {                        // tile 0 complete + 3 regions of tile 1 (nk >= 2)
    issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
    issue(0, 1); issue(1, 1); issue(2, 1);
  };

  // fragment addresses (elements): row (lane&31) of a 32-row group, chunk (2*ks + lane>>5) ^ swizzle;
  // the swizzle (row>>1)&7 only depends on lane&31 because every group starts at a multiple of 32.
  const int l31 = lane & 31;
  int fr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fr[ks] = l31 * PP_BK + (((2 * ks + (lane >> 5)) ^ ((l31 >> 1) & 7)) << 3);
  const int a_grp = wr * 64 * PP_BK;             // first row of this wave's rows inside region A0 / A1
  const int b_grp = wc * 32 * PP_BK;             // ... inside region B0 / B1

#define PP_READ_A(buf_, kind_)                                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)       \
      fa[i][ks] = *reinterpret_cast<const bf16x8*>(lds + (buf_) * PP_BUF + (kind_) * PP_REGION + a_grp + i * 32 * PP_BK + fr[ks]);
#define PP_READ_B(buf_, kind_, fb_)                                                                    \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                     \
      fb_[ks] = *reinterpret_cast<const bf16x8*>(lds + (buf_) * PP_BUF + (kind_) * PP_REGION + b_grp + fr[ks]);
// HW_: inside the fence right behind the section's request (waits of a rolling pass); HB_: behind the fence, in the same
// scheduling region as the section's remaining six MFMAs -- the compiler mixes it into their shadow
#define PP_MMA(i0_, j_, fb_, ISSUE_) PP_MMA_H(i0_, j_, fb_, ISSUE_, , )
#ifndef VTX_PP_PRIO
#define VTX_PP_PRIO 1   // 1 = s_setprio 1 around every MFMA section (default); experiments: 0 = none, 3 = the load sections at 1
#endif
#define PP_MMA_H(i0_, j_, fb_, ISSUE_, HW_, HB_)                                                       \
  if (VTX_PP_PRIO == 1) __builtin_amdgcn_s_setprio(1); else if (VTX_PP_PRIO == 3) __builtin_amdgcn_s_setprio(0); \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                      \
        acc[(i0_) + i][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb_[ks], acc[(i0_) + i][j_], 0, 0, 0); \
    if (ks == 0) { __builtin_amdgcn_sched_barrier(0); ISSUE_; HW_ __builtin_amdgcn_sched_barrier(0); HB_ }     \
  }                                                                                                    \
  if (VTX_PP_PRIO == 1) __builtin_amdgcn_s_setprio(0); else if (VTX_PP_PRIO == 3) __builtin_amdgcn_s_setprio(1);
#define PP_BAR() __builtin_amdgcn_s_barrier()

  // One 64-deep K tile = four phases.  ISS_: what the request slots of the four MFMA sections issue; H1_ .. H4_:
  // hooks inside the load sections of P1 .. P4.
  // The LDS-DMA requests are issued in the shadow of the MFMAs (after the first two of a section): inside a load
  // section each costs the wave 100+ cycles on the critical path, among MFMAs ~60.  (Round 3, same box A/B of three
  // builds that issued P2's and / or P4's request in that phase's load section instead -- 4 and 0 fragment reads there
  // -- behind the section's wait, i.e. at the same place of the request order: +0.5 / +0.8 / +1.1 % over the twelve GEMMs
  // of a layer, profiles/round3_nt_variants.txt.  The weight-gradient kernel, whose sections are longer, gains from it.)
#ifndef VTX_PP_BF16_STAGE
#define VTX_PP_BF16_STAGE 1          // packed row-pair staging of the plain epilogue (see PAIRS below); 0 = fp32 staging
#endif
  // X1_ / X2_ / X4_ (wave-uniform conditions): the wait in P1 / P2 / P4 names a region that was requested BEFORE the NST
  // result stores of this wave's previous epilogue, so its steady-state count is raised by NST (relaxed first K tiles of a
  // tile, see `relax` in the continuous flow below)
#define PP_KTILE(E1_, E2_, W1_, W2_, W4_, ISS_, H1_, H2_, H3_, H4_) PP_KTILE_X(E1_, E2_, W1_, W2_, W4_, ISS_, H1_, H2_, H3_, H4_, false, false, false)
#define PP_KTILE_X(E1_, E2_, W1_, W2_, W4_, ISS_, H1_, H2_, H3_, H4_, X1_, X2_, X4_) \
    PP_KTILE_XH(E1_, E2_, W1_, W2_, W4_, ISS_, H1_, H2_, H3_, H4_, X1_, X2_, X4_, false, false, , , , )
// X1R_ (with X1_): the previous tile's epilogue ROLLED -- the NSR stores of its first two passes went out inside its last K
// tile, ahead of the request this wait names; R4_: stores this section's predecessors issued behind the request P4's wait
// names (a rolling last K tile: the two stores of F(0), run-time flag); M3W_ / M3B_ / M4W_ / M4B_: hooks of the MFMA sections of P3 / P4 (see PP_MMA_H)
#define PP_KTILE_XH(E1_, E2_, W1_, W2_, W4_, ISS_, H1_, H2_, H3_, H4_, X1_, X2_, X4_, X1R_, R4_, M3W_, M3B_, M4W_, M4B_) \
    {                                                                                                  \
      const int buf = (kt & 1) ^ par;                                                                  \
      const bool e1 = (E1_), e2 = (E2_);           /* K tile kt+1 / kt+2 exists in this tile */        \
      const bool w1 = (W1_), w2 = (W2_);           /* ... or its slots carry other requests: steady-state wait counts */ \
      /* P1: reads A0, B0; P2 will read B1(kt) */                                                      \
      PP_READ_A(buf, 0);                                                                               \
      PP_READ_B(buf, 1, fb0);                                                                          \
      if (w1) { if (X1_) { if (X1R_) wait_vmcnt<8 + NST - NSR>(); else wait_vmcnt<8 + NST>(); } else wait_vmcnt<8>(); } else wait_vmcnt<2>(); \
      H1_                                                                                              \
      lgkm0();                                                                                         \
      PP_BAR();                                                                                        \
      PP_MMA(0, 0, fb0, ISS_(3, kt + 1, e1));                                                          \
      PP_BAR();                                                                                        \
      /* P2: reads B1; P3 will read A1(kt) */                                                          \
      PP_READ_B(buf, 2, fb1);                                                                          \
      if (w1) { if (X2_) wait_vmcnt<8 + NST>(); else wait_vmcnt<8>(); } else wait_vmcnt<0>();          \
      H2_                                                                                              \
      lgkm0();                                                                                         \
      PP_BAR();                                                                                        \
      PP_MMA(0, 1, fb1, ISS_(0, kt + 2, e2));                                                          \
      PP_BAR();                                                                                        \
      /* P3: reads A1 */                                                                               \
      PP_READ_A(buf, 3);                                                                               \
      H3_                                                                                              \
      lgkm0();                                                                                         \
      PP_BAR();                                                                                        \
      PP_MMA_H(2, 1, fb1, ISS_(1, kt + 2, e2), M3W_, M3B_);                                            \
      PP_BAR();                                                                                        \
      /* P4: no reads; P1 of the next K tile will read A0(kt+1), B0(kt+1) */                           \
      if (W4_) { if (w2) { if (X4_) wait_vmcnt<8 + NST>(); else if (R4_) wait_vmcnt<10>(); else wait_vmcnt<8>(); } else if (w1) wait_vmcnt<4>(); } \
      H4_                                                                                              \
      PP_BAR();                                                                                        \
      PP_MMA_H(2, 0, fb0, ISS_(2, kt + 2, e2), M4W_, M4B_);                                            \
      PP_BAR();                                                                                        \
    }
#define PP_ISS_COND(kind_, ktv_, ex_) if (ex_) issue(kind_, ktv_)          /* only K tiles of this tile */
#define PP_ISS_ALWAYS(kind_, ktv_, ex_) issue(kind_, ktv_)                 /* continuous flow: the next tile's follow */
#define PP_ISS_PRE(kind_, ktv_, ex_) if (ex_) issue(kind_, ktv_); else pre_slot(kind_, ktv_)   /* ... or the residual block */

  // Continuous flow (CONT: epilogues that leave the operand ring alone).  The K tiles of successive tiles form ONE
  // stream through the ring: the requests that the last two K tiles of a tile have no use for -- exactly the seven
  // regions of the old per-tile prologue -- fetch K tiles 0 and 1 of the NEXT tile under the same WAR / RAW rules as
  // any other K tile, so a tile starts with its operands on chip and nothing is requested between two main loops.
  // That needs the next tile's index two K tiles before the end of the current one:
  //   t      tile being computed            t_next  tile after it (known when the tile starts)
  //   t_nn   tile after t_next: drawn by lane 0 at K tile 0 (one atomic), handed to the other waves through an LDS
  //          word at K tile 1 -- both inside the main loop, under its counted waits, at no barrier of their own.
  // Nothing may wait for "all outstanding requests" any more (there always are next-tile requests in flight), so
  // the bias reaches the epilogue through LDS as well: one 256-B LDS-DMA per wave at K tile 0 into its idle
  // staging slice, read back into two registers before the first pass.
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned slot_rd = (unsigned)(unsigned long)(lds_char*)(smem + PP_RING_BYTES);
  const unsigned bias_rd = (unsigned)(unsigned long)(lds_char*)(reinterpret_cast<char*>(stg) + 256 + (lane & 31) * 4);
  // result stores one wave issues per tile when none of its 16 (pass, half-row) groups is empty: one per group, two with a
  // second output (the activation kernels are only instantiated for the FFN: GELU + GELU', or GELU + pre-activation copy)
  constexpr int NST = HAS_ACT ? 32 : 16;
  constexpr int NSR = NST / 4;                   // ... of them by its first two passes (the ones a rolling epilogue issues before P4's request of the last K tile)
  bool relax = false;                            // continuous operand flow only; set at the end of every epilogue
  bool prev_rolled = false;                      // ... which rolled: the stores of its first two passes precede the last K tile's P4 request
  int t = next_tile();
  const int tend = xcount;
  if (t >= tend) { check_out(); return; }
  int t_next = tend, t_nn = tend;
  if constexpr (CF) t_next = next_tile();
  int pending = 0;                               // index drawn ahead by lane 0 of the workgroup
  if (EPI2 && !CF && tid == 0) pending = atomicAdd(my_ctr, 1);
  set_tile(t, 0);
  m0 = m0s; n0 = n0s;
  prologue();
  while (true) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][4], fb0[4], fb1[4];
    const bool more_c = CF && t_next < tend;     // continuous flow: another tile follows this one
    // ---- epilogue state of this tile.  epi_head() fills the store addressing (after the main loop -- or in front of its last K
    // tile when the epilogue ROLLS: the first four passes of a lean tile, rows 0 .. 63 of the wave's block, which are final
    // after P2 of the last K tile, then run inside P3 / P4 of that K tile, in the shadow of its last sixteen MFMAs)
    const int em0 = m0 + wr * 128, en0 = n0 + wc * 64;
    const int tile_m0 = m0;                        // first row of the whole 256-row tile (wave-uniform)
    TileMap cm;
    unsigned ldcb = 0, lean_skipb = 0;
    bool lean = false;
    [[maybe_unused]] bool rolled = false;
    int lean_lb = 0;
    const char* lean_cb = nullptr;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t lean_rsrc;   // ROLL: base = lean_cb, bytes up to the end of C's last valid row
    float bcol[2] = {0.f, 0.f};
    auto epi_head = [&]() {
      cm = make_tile_map(ep.cmap, tile_m0);
      ldcb = (unsigned)ep.ldc * 2u;                // bytes per C row
      const long skipb64 = (long)cm.skip * (long)ldcb;
      lean = CONT && (dbg == 0 || dbg == 5 || dbg == 6) && em0 + 127 < ep.M && en0 + 63 < ep.N && (ep.split_row <= 0 || em0 + 127 < ep.split_row) &&
             cm.fast && cm.skip >= 0 && skipb64 < (1L << 30) && ep.ldc < (1L << 22) && PRE != PRE_DGELU;
      if constexpr (HAS_ACT) lean = lean && ep.act == 2 && ep.ldc2 < (1L << 22);
      if constexpr (ROLL) lean = true;             // what the host checked (launch_pp); ragged tiles by store predicates
      // first physical row of the wave's block, and the local row from which the map's skip applies (none if the whole
      // block lies behind the boundary: the skip is in the base then)
      lean_lb = em0 < cm.bound ? cm.bound - em0 : 0x7fffffff;
      lean_cb = reinterpret_cast<const char*>(ep.C) + ((cm.base_q + em0 + (em0 >= cm.bound ? (long)cm.skip : 0L)) * ep.ldc + en0) * 2;
      lean_skipb = (unsigned)skipb64;
      if constexpr (ROLL) {
        // C ends behind the physical row of logical row M - 1 (closed-form map: the host checked it)
        const long last = (long)ep.cmap.base + (ep.M - 1) + (ep.cmap.grp > 0 ? (long)((unsigned)(ep.M - 1) / (unsigned)ep.cmap.grp) * ep.cmap.skip : 0L);
        // (a wave whose 64 columns lie beyond N -- a ragged last column tile -- gets an empty range: it stores nothing)
        const long left = en0 < ep.N ? (reinterpret_cast<const char*>(ep.C) + (last + 1) * ep.ldc * 2) - lean_cb : 0L;
        lean_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(lean_cb), 0, (int)(left < 0 ? 0 : (left > 0x7fffffffL ? 0x7fffffffL : left)), 0x00020000);
      }
    };
    // Lean passes of the plain epilogue (packed row-pair staging; see PAIRS below), as stages: W(p) stages a pass, R(p) reads it
    // back, F(p) permutes and stores.  All LDS accesses are inline asm with a known instruction count; the LDS executes a
    // wave's instructions in issue order, so the stages form a software pipeline on ONE staging buffer and ONE register set.
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    [[maybe_unused]] unsigned pl_wa0 = 0, pl_wa1 = 0, pl_rd = 0, pl_v0 = 0;
    [[maybe_unused]] int pl_p2 = 0;
    [[maybe_unused]] u32x4 pl_w0, pl_w1;
    auto pl_setup = [&]() {                        // lane constants from an opaque lane id: never hoisted out of the tile loop
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const unsigned stg0 = (unsigned)(unsigned long)(lds_char*)(reinterpret_cast<char*>(stg));
      pl_p2 = ln >> 3;
      // staging words as in PP_EPI2B: lane (col, hi) writes row pairs 2 hi + {0, 1, 4, 5} at word 132 hi + {0, 64, 264, 328} + 32 ni + col
      pl_wa0 = stg0 + (unsigned)(((ln >> 5) * 132 + (ln & 31)) * 4);
      pl_wa1 = pl_wa0 + 264 * 4;
      pl_rd = stg0 + (unsigned)((pl_p2 * 64 + 4 * ((pl_p2 >> 1) & 1) + 8 * (pl_p2 >> 2) + (ln & 7) * 8) * 4);
      pl_v0 = (unsigned)(2 * pl_p2) * ldcb + (unsigned)(ln & 7) * 16u;
    };
#define PP_LWAIT(n_, ...) asm volatile("s_waitcnt lgkmcnt(%[cnt])" : __VA_ARGS__ : [cnt] "n"(n_) : "memory"); __builtin_amdgcn_sched_barrier(0)
#define PP_PWAIT(n_) PP_LWAIT(n_, "+v"(pl_w0), "+v"(pl_w1))
// (the bias add of a register pair is ONE v_pk_add_f32 -- same IEEE add per element --, and a launch without a bias skips it:
// acc + 0.0f is acc bit for bit, an accumulator that started from +0 never holds -0)
#define PP_PW(mi_, half_) if (ep.bias) { PP_PW_(mi_, half_, true) } else { PP_PW_(mi_, half_, false) }
#define PP_PW_(mi_, half_, B_)                                                                          \
    {                                                                                                   \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) { \
        f32x2 lo = mk2(acc[mi_][ni][8 * (half_) + 4 * kk], acc[mi_][ni][8 * (half_) + 4 * kk + 1]);     \
        f32x2 hi = mk2(acc[mi_][ni][8 * (half_) + 4 * kk + 2], acc[mi_][ni][8 * (half_) + 4 * kk + 3]); \
        if (B_) { const f32x2 b2 = mk2(bcol[ni], bcol[ni]); lo = lo + b2; hi = hi + b2; }               \
        union { bf16x2 v; unsigned u; } x;     /* row pairs 2 hi + 4 kk (registers 4 kk, 4 kk + 1) and + 1 (4 kk + 2, + 3) */ \
        union { bf16x2 v; unsigned u; } y;     /* (no comma outside parentheses in here: the stages are macro ARGUMENTS of the K tile) */ \
        x.v[0] = (__bf16)lo[0]; x.v[1] = (__bf16)lo[1];                                                 \
        y.v[0] = (__bf16)hi[0]; y.v[1] = (__bf16)hi[1];                                                 \
        if (ni == 0) asm volatile("ds_write2_b32 %0, %1, %2 offset1:64" :: "v"(kk ? pl_wa1 : pl_wa0), "v"(x.u), "v"(y.u) : "memory"); \
        else asm volatile("ds_write2_b32 %0, %1, %2 offset0:32 offset1:96" :: "v"(kk ? pl_wa1 : pl_wa0), "v"(x.u), "v"(y.u) : "memory"); \
      }                                                                                                 \
    }
#define PP_PR() asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(pl_w0), "=&v"(pl_w1) : "v"(pl_rd) : "memory")
#define PP_PF(p_)                                                                                       \
    {                                                                                                   \
      const char* base = lean_cb + (long)(16 * (p_)) * ldcb;                                            \
      if constexpr (!ROLL) asm volatile("" : "+s"(base));   /* a scalar base (see st16_nt_s) */         \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                   \
        const unsigned sel = u ? 0x07060302u : 0x05040100u;                                             \
        u32x4 o;                                                                                        \
        o[0] = __builtin_amdgcn_perm(pl_w0[1], pl_w0[0], sel);                                          \
        o[1] = __builtin_amdgcn_perm(pl_w0[3], pl_w0[2], sel);                                          \
        o[2] = __builtin_amdgcn_perm(pl_w1[1], pl_w1[0], sel);                                          \
        o[3] = __builtin_amdgcn_perm(pl_w1[3], pl_w1[2], sel);                                          \
        const unsigned voff = pl_v0 + (u ? ldcb : 0u) + ((2 * pl_p2 + u >= lean_lb - 16 * (p_)) ? lean_skipb : 0u); \
        /* ROLL: every tile comes through here, also the ragged last row tile -- the descriptor ends behind the last valid */ \
        /* row of C, the range check of the buffer store drops the lanes beyond it (no predicate, no divergent branch)     */ \
        if constexpr (ROLL) __builtin_amdgcn_raw_buffer_store_b128(o, lean_rsrc, voff + (unsigned)(16 * (p_)) * ldcb, 0, 2); \
        else st16_nt_b(base, voff, o);                                                                  \
      }                                                                                                 \
    }
    // ROLL kernels (launch_pp: the plain flow whose C map, leading dimension and K the host has checked): EVERY tile takes the lean
    // stages -- rows / columns beyond M x N are handled by store predicates -- and the first four passes of every tile that has a
    // successor in this workgroup run inside its last K tile.  (One code path on purpose: with a rolling AND a non-rolling copy of
    // the last K tile behind the same loop hipcc splits the accumulator tuples at the join and spills 160 - 220 registers.)
    constexpr bool ROLLK = ROLL;
    static_assert(!ROLL || (CF && VTX_PP_BF16_STAGE && PRE == PRE_NONE && !HAS_SC && !HAS_ACT), "rolling: the plain continuous flow only");
    // Waits are per region and counted: each names the region the NEXT phase reads and leaves every
    // younger request in flight (2 DMA instructions per region; issue order A0 B0 B1 A1 per K tile).
    stamp(0);
    // A0(0), B0(0) landed; B1(0) A1(0) A0(1) B0(1) B1(1) -- and, relaxed (see the continuous flow below), the previous
    // epilogue's stores -- in flight
    if (relax) wait_vmcnt<10 + NST>(); else wait_vmcnt<10>();
    PP_BAR();
    stamp(1);
    if (wr == 1) PP_BAR();                        // group 1 runs one barrier behind
    if constexpr (CF) {
      // bias of this wave's 64 columns -> staging slice (+256 B): lane l fetches column en0 + l;
      // DropPath scales of its 128 rows -> staging slice (+512 B): lane l fetches the scales of rows l and 64 + l
      auto bias_dma = [&]() {
        if (ep.bias) {
          const int c = n0 + wc * 64 + lane;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ep.bias + (c < ep.N ? c : 0)),
                                           (__attribute__((address_space(3))) void*)(stg + 64), 4, 0, 0);
        }
        if constexpr (HAS_SC) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            int ml = m0 + wr * 128 + h * 64 + lane;
            if (ml >= ep.M) ml = ep.M - 1;
            const bool split = ep.split_row > 0 && ml >= ep.split_row;
            const unsigned q1 = fast_div((unsigned)ml, ep.rs_magic1, ep.rs_shift1);
            const unsigned q2 = fast_div((unsigned)ml, ep.rs_magic2, ep.rs_shift2);
            const int idx = split ? (ml - ep.split_row) : (int)q1 * ep.rs_m1 + (ml - (int)q2 * ep.rs_d2) * ep.rs_m2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ep.row_scale + idx),
                                             (__attribute__((address_space(3))) void*)(stg + 128 + h * 64), 4, 0, 0);
          }
        }
      };
      // The draw.  hipcc cannot be allowed to see it: (a) its atomic optimizer aggregates a uniform-address atomic over
      // the wave and reads the result back at once; (b) its wait-count pass treats every LDS-DMA as a FLAT access that
      // may return out of order, so ANY vector-memory result it tracks is waited for with vmcnt(0) while a DMA is in
      // flight -- and here one always is.  So: the atomic and the LDS store of its result are inline asm, both inside
      // K tile 1 of the main loop (straight-line code between them), and the store sits behind an explicit counted
      // wait: lane 0's wave issues six operand requests between the two, so vmcnt(6) means the atomic has returned.  `drawn` must not be touched by anything else (checked in the ISA: one def, one use).
      int drawn;                                   // deliberately NOT initialised: with `= 0` hipcc shares the register with a
                                                   // zero it keeps for 64-bit address arithmetic (seen in the ISA: v_mad_u64_u32
                                                   // reading the pair between the atomic and its hand-over)
      asm volatile("" : "=v"(drawn));              // an opaque definition for the paths that never draw
      const int one = 1;
      int kt = 0;
      if (more_c) {
        // every request of every K tile is unconditional here.  Draw and hand-over sit in K tile 1 (the other waves read
        // the word in K tile 2: nk >= 3, the launcher sends shorter K to the per-tile flow).  They used to sit in K tile 0,
        // where the hand-over's wait made wave 0 wait for the previous tile's result stores as well (vector memory
        // operations retire in issue order, and the atomic is younger than those stores).
#define PP_CF_H1 if (kt == 0) bias_dma();                                                                        \
                 if (kt == 1 && tid == 0)                                                                        \
                   asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=&v"(drawn) : "v"(my_ctr), "v"(one) : "memory");
#define PP_CF_H2 if (kt == nk - 2) set_tile(t_next, (long)nk * (PP_BK * 2));      /* after P1's A1(nk-1): the last in-tile request */
#define PP_CF_H3 if (kt == 2) {                                                                                  \
                   int v;                                                                                        \
                   asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(slot_rd) : "memory"); \
                   t_nn = __builtin_amdgcn_readfirstlane(v);                                                     \
                 }
#define PP_CF_H4 if (kt == 1 && wave == 0) {                                                                     \
                   wait_vmcnt<6>();                                                                              \
                   if (tid == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(slot_rd), "v"(drawn) : "memory");     \
                   lgkm0();                                                                                      \
                 }
        // Relaxed first K tiles.  The regions K tiles 0 and 1 of this tile read were requested inside the PREVIOUS main loop,
        // i.e. before the NST result stores of the previous epilogue; vmcnt counts loads and stores alike and retires
        // them in issue order, so a wait that names one of those regions only has to leave the stores (and what was
        // requested since) outstanding: its count grows by NST.  With the steady-state counts every wait of K tile 0 doubled
        // as a wait for the stores -- 0.5 us per tile for a plain epilogue, 4.7 us for the FFN's two GELU outputs
        // (profiles/round2_pp_timeline_flows.txt, segment 0-1).  From P2 of K tile 1 on the waited-for regions are younger than
        // the stores.  `relax` (wave-uniform): this wave's last epilogue issued exactly NST stores.
        for (; kt < nk - 1; ++kt) {
          PP_KTILE_XH(true, true, true, true, true, PP_ISS_ALWAYS, PP_CF_H1, PP_CF_H2;, PP_CF_H3, PP_CF_H4,
                      relax && kt < 2, relax && kt == 0, relax && kt == 0, prev_rolled && kt == 1, false, , , , )
        }
        // The last K tile (nk >= 4 in ROLL kernels: none of the hooks above fires in K tile nk - 1, and the hand-over word in wave
        // 0's staging slice -- written in K tile 1, read in K tile 2 -- is dead).  Rows 0 .. 63 of the wave's block are final after P2:
        //   P3 load section   W(0) R(0)                  (behind the A1 fragment reads; the section's own lgkmcnt(0) covers R(0))
        //   P3 MFMA section   F(0) W(1) R(1)             (behind the section's request: its two stores are YOUNGER than that request;
        //                                                 straight-line code in the scheduling region of the last six MFMAs)
        //   P4 load section   wait F(1) W(2) R(2)        (behind the section's counted wait, which gains the two stores of F(0) --
        //                                                 when they were issued: a full block; a ragged one keeps the plain count)
        //   P4 MFMA section   wait F(2) W(3) R(3)
        // and the rest -- F(3), passes 4 .. 7 -- runs behind the loop as before.
        if constexpr (ROLLK) {
          epi_head();
          rolled = true;
          const bool full = em0 + 127 < ep.M && en0 + 63 < ep.N && trace == nullptr;
          if (ep.bias) asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:128\n\ts_waitcnt lgkmcnt(0)"
                                    : "=&v"(bcol[0]), "=&v"(bcol[1]) : "v"(bias_rd) : "memory");
          pl_setup();
          PP_KTILE_XH(true, true, true, true, true, PP_ISS_ALWAYS, , , PP_PW(0, 0) PP_PR();,
                      PP_PWAIT(0); PP_PF(1) PP_PW(1, 0) PP_PR();,
                      false, false, false, false, full,
                      , PP_PF(0) PP_PW(0, 1) PP_PR();,
                      PP_PWAIT(0);, PP_PF(2) PP_PW(1, 1) PP_PR();)
        } else {
          PP_KTILE_XH(true, true, true, true, true, PP_ISS_ALWAYS, PP_CF_H1, PP_CF_H2;, PP_CF_H3, PP_CF_H4,
                      relax && kt < 2, relax && kt == 0, relax && kt == 0, prev_rolled && kt == 1, false, , , , )
        }
#undef PP_CF_H1
#undef PP_CF_H2
#undef PP_CF_H3
#undef PP_CF_H4
      } else {
        for (; kt < nk; ++kt) {
          PP_KTILE(kt + 1 < nk, kt + 2 < nk, kt + 1 < nk, kt + 2 < nk, true, PP_ISS_COND, if (kt == 0) bias_dma();, , , )
        }
      }
    } else if constexpr (PF) {
      // Residual-block flow.  The epilogue's 128 x 64 block per wave (residual, GELU' input or multiplier: 128 KB per
      // workgroup) used to be requested after the main loop into the idle ring -- every CU of the chip asking for its
      // 128 KB at the same moment and waiting ~5.6 us for it -- and the next tile's operands could only be requested
      // after the passes had read it.  Now the block takes the request slots the last two K tiles have no use for:
      // slot (kind, K tile nk or nk + 1) = ring region s = 4 (ktv - nk) + kind, freed by the main loop in exactly that
      // order, receives rows [16 s, 16 s + 16) of every wave's block (each wave its own 2 KB of the region, through its
      // own two DMA instructions); regions 6 and 7 share the last slot.  Pass p of the epilogue reads region p, and the
      // wave then requests ITS part of the next tile's operand region into the 2 KB it has just read: the standard
      // seven-region prologue, spread over the passes, with no barrier (no wave touches another wave's part).
      // Addresses as in set_tile(): a wave-uniform 64-bit base (the block row that holds the tile's first row) + 32-bit
      // lane offsets; row maps with at most one group boundary per tile (the launcher routes others, and periodic
      // residuals, to the per-tile flow).  Split rows (no residual) and rows beyond M read the tile's first row.
      constexpr bool pres = PRE == PRE_RES;
      const int pen = n0 + wc * 64 + (lane & 7) * 8;
      const unsigned penc2 = pen < ep.N ? (unsigned)pen * 2u : 0u;
      const unsigned pld2 = (unsigned)(pres ? ep.ldr : ep.ld_dgelu) * 2u;
      // rows without a block row (beyond M; split rows, which take no residual) read the last row that has one
      const int lastv = min(ep.M - 1, (pres && ep.split_row > 0) ? ep.split_row - 1 : 0x7fffffff);
      const int m0c = min(m0, lastv);              // a tile of split rows only: everything reads row `lastv`
      const TileMap prm = make_tile_map(ep.rmap, m0c);
      const long pfirst = pres ? tile_map_row_u(prm, ep.rmap, m0c) : (long)m0c;
      const char* pbase = reinterpret_cast<const char*>(pres ? ep.R : ep.dgelu_in) + pfirst * (long)pld2;
      const int pbl = (pres && ep.rmap.grp > 0) ? prm.bound - m0c : 0x7fffffff;         // local row at which the skip starts
      const int pskip = pres ? prm.skip : 0;       // (the tile's own step in the table form)
      const int plast = lastv - m0c;               // >= 0
      const int pl0 = (m0 - m0c) + wr * 128 + (lane >> 3);
      auto pre_piece = [&](int j, bf16raw* dst) {  // rows 8j .. 8j+7 of the block: lane -> row 8j + lane/8, chunk lane%8
        const int l = min(pl0 + 8 * j, plast);
        const int lr = l + (l >= pbl ? pskip : 0);
        const char* b = pbase;
        asm volatile("" : "+s"(b));
        dma16_nt(reinterpret_cast<const bf16raw*>(b + ((unsigned)lr * pld2 + penc2)), dst);
      };
      auto pre_slot = [&](int kind, int ktv) {
        const int s = (ktv - nk) * 4 + kind;
        bf16raw* dst = lds + ((ktv & 1) ^ par) * PP_BUF + kind * PP_REGION + wave * 1024;
        pre_piece(2 * s, dst);
        pre_piece(2 * s + 1, dst + 512);
        if (s == 6) {                              // region 7 = (A1, K tile nk + 1): read in P3, free from P4 on
          bf16raw* dst7 = lds + ((ktv & 1) ^ par) * PP_BUF + 3 * PP_REGION + wave * 1024;
          pre_piece(14, dst7);
          pre_piece(15, dst7 + 512);
        }
      };
      // Every slot carries two requests as in the steady state: steady-state wait counts -- except P4 of the last K
      // tile, whose steady-state wait names the regions of "K tile nk": block pieces the main loop does not need, which
      // come from HBM with every CU asking for its 128 KB in the same few microseconds (requests return in order, so the
      // wait would stall on them: measured +2.8 us per tile).  All operand requests precede all block requests.
      for (int kt = 0; kt < nk; ++kt) {
        PP_KTILE(kt + 1 < nk, kt + 2 < nk, true, true, kt + 1 < nk, PP_ISS_PRE, , , , )
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        PP_KTILE(kt + 1 < nk, kt + 2 < nk, kt + 1 < nk, kt + 2 < nk, true, PP_ISS_COND, , , , )
      }
    }
    if (wr == 0) PP_BAR();                        // pairs with group 1's extra barrier: every wave is past its last LDS read
    stamp(2);
    if constexpr (!EPI2) {
      t = next_tile();
      const bool more = t < tend;
      if (more) { set_tile(t, 0); m0 = m0s; n0 = n0s; prologue(); }   // ring is free: request the next tile before the epilogue
      // eight 16-row passes, expanded by hand: a loop here makes the compiler index acc[] dynamically
      // (= the whole accumulator goes through scratch)
#define PP_EPI(mi_, half_)                                                                              \
      {                                                                                                 \
        const int col = lane & 31, rhalf = (lane >> 5) * 4;                                             \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int r = 0; r < 8; ++r)  \
            stg[((r & 3) + 8 * (r >> 2) + rhalf) * PP_STG_LD + ni * 32 + col] = acc[mi_][ni][8 * (half_) + r]; \
        lgkm0();                                                                                        \
        __builtin_amdgcn_wave_barrier();                                                                \
        epilogue<bf16raw, 16, 2, PP_STG_LD>(ep, stg, em0 + (mi_) * 32 + (half_) * 16, en0, lane);       \
        lgkm0();                                                                                        \
        __builtin_amdgcn_wave_barrier();                                                                \
      }
      PP_EPI(0, 0) PP_EPI(0, 1) PP_EPI(1, 0) PP_EPI(1, 1) PP_EPI(2, 0) PP_EPI(2, 1) PP_EPI(3, 0) PP_EPI(3, 1)
#undef PP_EPI
      if (!more) break;
    } else {
      // The index of the next tile was drawn one tile ago (`pending`): publishing it costs two barriers,
      // not an atomic round trip; the draw for the tile after it goes out below and returns under the
      // epilogue's own load latency.
      bool more = more_c;
      if constexpr (!CF) {
        t = publish(pending);
        more = t < tend;
      }
      stamp(3);
      // Every global READ of the epilogue happens HERE, before the tile's first store.  vmcnt counts loads and
      // stores together and a wait can only name "all but the N youngest": a load waited for inside the store
      // passes drains every store issued before it.  (The per-pass epilogue above does that eight times per
      // tile -- `s_waitcnt vmcnt(0)` ahead of each pass's bias / scale / residual use -- so the 128-KB C tile
      // leaves at store LATENCY: measured 14 us per tile, 24..28 us with a residual, against a 17 us main loop at
      // K = 768.)
      //   bias: 8 registers; DropPath scales of the lane's 16 rows: 16 registers;
      //   residual / GELU' input of the wave's 128 x 64 block: 16 KB -- too many registers beside the 128
      //   accumulators, so it goes through LDS: 16 LDS-DMA pieces into the wave's slice of the (idle) operand
      //   ring.  The next tile's prologue then has to wait for the passes to finish reading it (one extra
      //   barrier; ~1.5 us of prologue latency exposed instead of ~10 us of store latency).
      // an opaque copy of the lane id for everything below: the epilogue's lane constants (staging / block read addresses, row and
      // column of the lane, ...) are then recomputed per tile -- some twenty vector instructions -- instead of being hoisted out
      // of the tile loop and kept in registers through the main loop (which spilled once the lean passes were added)
      int le = lane;
      asm volatile("" : "+v"(le));
      const int en = en0 + (le & 7) * 8;
      const bool col_ok = en < ep.N;
      const int enc = col_ok ? en : 0;
      const int er = em0 + (le >> 3);            // row of pass p, half-row u: er + 16 p + 8 u
      constexpr bool HAS_PRE = PRE != PRE_NONE, pre_res = PRE == PRE_RES;
      bf16raw* const pre_lds = lds + wave * (128 * 64);          // [128][64] bf16, row r at r * 64: le-linear per piece
      if (!rolled) epi_head();                       // (a rolling epilogue did it in front of the last K tile)
      if constexpr (HAS_PRE && !PF) {
        const bf16raw* const pre_base = reinterpret_cast<const bf16raw*>(pre_res ? ep.R : ep.dgelu_in);
        const long pre_ld = pre_res ? ep.ldr : ep.ld_dgelu;
        const TileMap rm = make_tile_map(ep.rmap, tile_m0);
        const int row_last = ep.M - 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) {             // piece j = rows 8j .. 8j+7 of the block; le -> row 8j + le/8, chunk le%8
          int ml = er + 8 * j;
          const bool past = ml > row_last;         // always-valid addresses; only the stores are predicated
          if (past) ml = row_last;
          long rr = ml;
          if (pre_res) {
            const bool split = ep.split_row > 0 && ml >= ep.split_row;      // split rows take no residual: read row 0
            rr = split ? 0 : (ep.r_period > 0 ? (long)(ml % ep.r_period) : (past ? map_row(ep.rmap, ml) : tile_map_row(rm, ep.rmap, ml)));
          }
          dma16(pre_base + rr * pre_ld + enc, pre_lds + j * 512);
        }
      }
      // bias in the ACCUMULATOR layout (a le owns one column of each 32-column half: 2 registers instead of the 8 a
      // row-vector le needs, and 64 adds per tile instead of 128); same fp32 add, same result
      if (ep.bias && !rolled) {
        if constexpr (CF) {                        // landed in the staging slice during K tile 0 (see above)
          asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:128\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(bcol[0]), "=&v"(bcol[1]) : "v"(bias_rd) : "memory");
        } else {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int c = en0 + ni * 32 + (le & 31);
            bcol[ni] = ep.bias[c < ep.N ? c : 0];
          }
        }
      }
      float sc[8][2];
      // lean passes: the 128 row scales of the wave's block stay in TWO registers (le l: rows l and 64 + l); a pass fetches its
      // two values per le with ds_bpermute_b32 (no LDS storage: the staging slice is full) inside its read batch
      float scl[2] = {0.f, 0.f};
      if constexpr (HAS_SC && CF) {                // landed in the staging slice during K tile 0: row r of the block at word 128 + r
        if (lean) {
          const unsigned scl_rd = (unsigned)(unsigned long)(lds_char*)(reinterpret_cast<char*>(stg) + 512 + le * 4);
          asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(scl[0]), "=&v"(scl[1]) : "v"(scl_rd) : "memory");
        } else {
        const unsigned sc_rd = (unsigned)(unsigned long)(lds_char*)(reinterpret_cast<char*>(stg) + 512 + (le >> 3) * 4);
        asm volatile("ds_read_b32 %0, %16\n\tds_read_b32 %1, %16 offset:32\n\tds_read_b32 %2, %16 offset:64\n\t"
                     "ds_read_b32 %3, %16 offset:96\n\tds_read_b32 %4, %16 offset:128\n\tds_read_b32 %5, %16 offset:160\n\t"
                     "ds_read_b32 %6, %16 offset:192\n\tds_read_b32 %7, %16 offset:224\n\tds_read_b32 %8, %16 offset:256\n\t"
                     "ds_read_b32 %9, %16 offset:288\n\tds_read_b32 %10, %16 offset:320\n\tds_read_b32 %11, %16 offset:352\n\t"
                     "ds_read_b32 %12, %16 offset:384\n\tds_read_b32 %13, %16 offset:416\n\tds_read_b32 %14, %16 offset:448\n\t"
                     "ds_read_b32 %15, %16 offset:480\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(sc[0][0]), "=&v"(sc[0][1]), "=&v"(sc[1][0]), "=&v"(sc[1][1]), "=&v"(sc[2][0]), "=&v"(sc[2][1]),
                       "=&v"(sc[3][0]), "=&v"(sc[3][1]), "=&v"(sc[4][0]), "=&v"(sc[4][1]), "=&v"(sc[5][0]), "=&v"(sc[5][1]),
                       "=&v"(sc[6][0]), "=&v"(sc[6][1]), "=&v"(sc[7][0]), "=&v"(sc[7][1])
                     : "v"(sc_rd) : "memory");
        }
      }
      if constexpr (HAS_SC && !CF) {
        auto scale_of = [&](int ml) -> float {
          if (ml >= ep.M) ml = ep.M - 1;
          const bool split = ep.split_row > 0 && ml >= ep.split_row;
          const unsigned q1 = fast_div((unsigned)ml, ep.rs_magic1, ep.rs_shift1);
          const unsigned q2 = fast_div((unsigned)ml, ep.rs_magic2, ep.rs_shift2);
          return ep.row_scale[split ? (ml - ep.split_row) : (int)q1 * ep.rs_m1 + (ml - (int)q2 * ep.rs_d2) * ep.rs_m2];
        };
        if (lean) {
          scl[0] = scale_of(em0 + le);
          scl[1] = scale_of(em0 + 64 + le);
        } else {
#pragma unroll
          for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int u = 0; u < 2; ++u) sc[p][u] = scale_of(er + 16 * p + 8 * u);
        }
      }
      if constexpr (!CF) {
        if (more && tid == 0) pending = atomicAdd(my_ctr, 1);
        // one wait for all of it; the empty asm statements make the loaded registers "used" here, so that hipcc's own
        // wait for them lands before the prologue's DMA requests and not (as vmcnt(0)) behind them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(4);
        asm volatile("" : "+v"(bcol[0]));
        asm volatile("" : "+v"(bcol[1]));
        if constexpr (HAS_SC) {
#pragma unroll
          for (int p = 0; p < 8; ++p) { asm volatile("" : "+v"(sc[p][0])); asm volatile("" : "+v"(sc[p][1])); }
          asm volatile("" : "+v"(scl[0])); asm volatile("" : "+v"(scl[1]));
        }
        if constexpr (!HAS_PRE) {
          if (more) { set_tile(t, 0); m0 = m0s; n0 = n0s; prologue(); }   // ring is free: the next tile's first 7 regions land under the passes
        }
        if constexpr (PF) {
          if (more) set_tile(t, (long)nk * (PP_BK * 2));   // the passes below request its K tiles 0 and 1 as stream positions nk, nk + 1
        }
      }
      stamp(5);
      // The staged rows (and the residual rows) are read back with inline-asm ds_read + lgkmcnt(0) in ONE statement:
      // hipcc orders any ds_read IT emits behind every LDS-DMA in flight (`s_waitcnt vmcnt(0)`), i.e. pass 0 would
      // wait for the 14 prologue requests issued just above.
      const unsigned stg_rd = (unsigned)(unsigned long)(lds_char*)(reinterpret_cast<char*>(stg) +
                                                                    ((le >> 3) * PP_STG_LD + (le & 7) * 8) * 4);
      const unsigned pre_rd = (unsigned)(unsigned long)(lds_char*)(reinterpret_cast<char*>(pre_lds) + le * 16);
      // residual-block flow: pass p reads this wave's 2 KB of ring region p (buffer pa for p < 4, the other one after)
      const unsigned pf_rd = (unsigned)(unsigned long)(lds_char*)(reinterpret_cast<char*>(lds + wave * 1024) + le * 16);
      const int pa = (nk & 1) ^ par;
#define PP_EPI2(p_, mi_, half_)                                                                         \
      {                                                                                                 \
        const int col = le & 31, rhalf = (le >> 5) * 4;                                             \
        if (dbg != 3) {                                                                                 \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int r = 0; r < 8; ++r)  \
            stg[((r & 3) + 8 * (r >> 2) + rhalf) * PP_STG_LD + ni * 32 + col] = acc[mi_][ni][8 * (half_) + r] + bcol[ni]; \
        } else { stg[le] = acc[mi_][0][8 * (half_)] + acc[mi_][1][8 * (half_) + 7]; }                 \
        lgkm0();                                                                                        \
        __builtin_amdgcn_wave_barrier();                                                                \
        f32x4 s00, s01, s10, s11;                                                                       \
        u32x4 pre[2];                                                                                   \
        if constexpr (HAS_PRE) {                                                                        \
          const unsigned prd__ = PF ? pf_rd + (unsigned)((((p_) < 4 ? pa : pa ^ 1) * PP_BUF + ((p_) & 3) * PP_REGION) * 2) \
                                    : pre_rd + 2 * (p_) * 1024;                                         \
          asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\t"                       \
                       "ds_read_b128 %2, %6 offset:2048\n\tds_read_b128 %3, %6 offset:2064\n\t"         \
                       "ds_read_b128 %4, %7\n\tds_read_b128 %5, %7 offset:1024\n\t"                     \
                       "s_waitcnt lgkmcnt(0)"                                                           \
                       : "=&v"(s00), "=&v"(s01), "=&v"(s10), "=&v"(s11), "=&v"(pre[0]), "=&v"(pre[1])   \
                       : "v"(stg_rd), "v"(prd__) : "memory");                                           \
        } else                                                                                            \
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"                       \
                       "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:2064\n\t"         \
                       "s_waitcnt lgkmcnt(0)"                                                           \
                       : "=&v"(s00), "=&v"(s01), "=&v"(s10), "=&v"(s11) : "v"(stg_rd) : "memory");      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        float v[2][8];                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
          v[0][j] = s00[j]; v[0][4 + j] = s01[j]; v[1][j] = s10[j]; v[1][4 + j] = s11[j];               \
        }                                                                                               \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                 \
          const int m = er + 16 * (p_) + 8 * u;                                                         \
          const bool ok = col_ok && m < ep.M && dbg != 2;                                               \
          const bool split = ep.split_row > 0 && m >= ep.split_row;                                     \
          if constexpr (HAS_ACT) {                                                                      \
            if (ep.act == 2) {                     /* GELU, second output = its derivative */          \
              float gp[8];                                                                              \
              _Pragma("unroll") for (int j = 0; j < 8; ++j) gelu_erf_both(v[u][j], v[u][j], gp[j]);     \
              if (ok) PP_ST8(reinterpret_cast<bf16raw*>(ep.C2) + (long)m * ep.ldc2 + en, gp);        \
            } else {                                                                                    \
              if (ep.C2 && ok) store8(reinterpret_cast<bf16raw*>(ep.C2) + (long)m * ep.ldc2 + en, v[u]); \
              _Pragma("unroll") for (int j = 0; j < 8; ++j) v[u][j] = gelu_erf(v[u][j]);                \
            }                                                                                           \
          }                                                                                             \
          if constexpr (PRE == PRE_MUL) {          /* the block holds gelu'(x) itself */                \
            const uint32_t w[4] = {pre[u][0], pre[u][1], pre[u][2], pre[u][3]};                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
              v[u][2 * j] *= __uint_as_float(w[j] << 16);                                               \
              v[u][2 * j + 1] *= __uint_as_float(w[j] & 0xffff0000u);                                   \
            }                                                                                           \
          }                                                                                             \
          if constexpr (PRE == PRE_DGELU) {                                                             \
            const uint32_t w[4] = {pre[u][0], pre[u][1], pre[u][2], pre[u][3]};                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
              v[u][2 * j] *= gelu_erf_grad(__uint_as_float(w[j] << 16));                                \
              v[u][2 * j + 1] *= gelu_erf_grad(__uint_as_float(w[j] & 0xffff0000u));                    \
            }                                                                                           \
          }                                                                                             \
          if constexpr (HAS_SC) { _Pragma("unroll") for (int j = 0; j < 8; ++j) v[u][j] *= sc[p_][u]; } \
          if (pre_res && !split) {                                                                      \
            const uint32_t w[4] = {pre[u][0], pre[u][1], pre[u][2], pre[u][3]};                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
              v[u][2 * j] += __uint_as_float(w[j] << 16);                                               \
              v[u][2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);                                   \
            }                                                                                           \
          }                                                                                             \
          if (ok) {                                                                                     \
            if (split) store8(reinterpret_cast<bf16raw*>(ep.Csplit) + (long)(m - ep.split_row) * ep.ldsplit + en, v[u]); \
            else PP_ST8(reinterpret_cast<bf16raw*>(ep.C) + tile_map_row(cm, ep.cmap, m) * ep.ldc + en, v[u]); \
          }                                                                                             \
        }                                                                                               \
        if constexpr (PF) {                        /* region p_ is read: it takes the next tile's operands */ \
          if ((p_) < 7 && more) issue((p_) & 3, nk + ((p_) >> 2));                                      \
        }                                                                                               \
      }
      // Plain epilogue (nothing to read, no activation, no row scale: result = bf16(acc + bias)): the pass stages PAIRS OF
      // ROWS as packed bf16 -- registers 2k, 2k+1 of an accumulator are the same column of two adjacent rows, so one
      // v_cvt_pk_bf16_f32 + one ds_write_b32 stage two elements -- instead of fp32 words: 8 LDS writes and 2 LDS reads
      // per le and pass instead of 16 and 4 (the fp32 staging moved 512 KB through the LDS per tile, ~2.3 us of the
      // 4.7 us the eight passes of a plain tile take, most of it on the 64-B-per-clock ds_write_b32 path).  A le then
      // holds rows 2p, 2p+1 (p = le / 8) of 8 columns as 8 words {row 2p | row 2p+1}: eight v_perm_b32 separate them.
      // Row-pair r sits at word r * 64 + 4 * ((r >> 1) & 1) + 8 * (r >> 2): the 4-word shift keeps the 16-le groups of the
      // ds_read_b128 on 64 distinct banks, the 8-word one keeps row-pair 3's shifted tail off row-pair 4.  Same value, same rounding (one fp32 add, one round-to-nearest-even) as the fp32 staging.
      constexpr bool PAIRS = VTX_PP_BF16_STAGE && PRE == PRE_NONE && !HAS_SC && !HAS_ACT;
      // Lean passes (round 4).  Wave-uniform test per tile: the wave's whole 128 x 64 block lies inside M x N, none of its rows
      // is a split row, the C row map is in tile form (at most one group boundary inside the tile) and its skip fits 30 bits
      // of bytes.  Then (a) a store address is a SCALAR base (first row of the pass, s_add per pass) + a 32-bit le offset
      // (global_store saddr form): no predicate, no branch, three vector instructions per row where the general pass above
      // spends ~45 (64-bit multiplies, the split / table / slow-map branches); (b) every LDS access of a pass is inline asm
      // with a known instruction count, which lets the passes run as a software pipeline on ONE staging buffer and ONE
      // register set: the LDS executes a wave's instructions in issue order, so W(p+1) may be issued right behind R(p) with no
      // wait in between, and the wait in front of F(p) is lgkmcnt(<number of W instructions>): R(p)'s latency is covered by
      // the issue of W(p+1), W(p+1)'s by F(p)'s arithmetic and stores.
      //     W(0) R(0) | W(1) wait F(0) R(1) | W(2) wait F(1) R(2) | ... | W(7) wait F(6) R(7) | wait F(7)
      // Same values in the same order as the general passes (bit-identical: tests/test_gpu_kernels.py::test_gemm_nt_pp_lean_passes).
      const unsigned stg_wr0 = (unsigned)(unsigned long)(lds_char*)(reinterpret_cast<char*>(stg)) ;
      if (lean) {
        int ln = le;                               // opaque copy: the le constants below are recomputed per tile (a dozen vector
        asm volatile("" : "+v"(ln));                 // instructions) instead of living in registers through the main loop
        if constexpr (PAIRS) {
          // the stages are defined in front of the main loop (PP_PW / PP_PR / PP_PF): a rolling epilogue has run
          // W(0) R(0) F(0) ... W(3) R(3) inside the last K tile and continues here with F(3)
          if (rolled) {
            PP_PWAIT(0); PP_PF(3)
          } else {
            pl_setup();
            PP_PW(0, 0) PP_PR();
            PP_PW(0, 1) PP_PWAIT(4); PP_PF(0) PP_PR();
            PP_PW(1, 0) PP_PWAIT(4); PP_PF(1) PP_PR();
            PP_PW(1, 1) PP_PWAIT(4); PP_PF(2) PP_PR();
            PP_PWAIT(0); PP_PF(3)
          }
          PP_PW(2, 0) PP_PR();
          PP_PW(2, 1) PP_PWAIT(4); PP_PF(4) PP_PR();
          PP_PW(3, 0) PP_PWAIT(4); PP_PF(5) PP_PR();
          PP_PW(3, 1) PP_PWAIT(4); PP_PF(6) PP_PR();
          PP_PWAIT(0); PP_PF(7)
        } else {
          // fp32 staging as in PP_EPI2: le (col, hi) writes rows 4 hi + {0, 1, 2, 3, 8, 9, 10, 11} at word 256 hi + {0, 64, 128, 192, 512, ...} + 32 ni + col
          const unsigned wa0 = stg_wr0 + (unsigned)(((ln >> 5) * 256 + (ln & 31)) * 4), wa1 = wa0 + 512 * 4;
          const unsigned vf0 = (unsigned)(ln >> 3) * ldcb + (unsigned)(ln & 7) * 16u;
          const unsigned ldc2b = (unsigned)ep.ldc2 * 2u;               // second output (GELU'): identity row map
          const unsigned vg0 = (unsigned)(ln >> 3) * ldc2b + (unsigned)(ln & 7) * 16u;
          const char* const lean_c2b = reinterpret_cast<const char*>(ep.C2) + ((long)em0 * ep.ldc2 + en0) * 2;
          f32x4 s00, s01, s10, s11;
          u32x4 pre[2];
          float scu[2] = {1.f, 1.f};                   // the pass's two row scales (rows le / 8 and + 8 of its 16)
          const unsigned bp_addr = (unsigned)(ln >> 3) * 4u;
#define PP_LW(mi_, half_) if (ep.bias) { PP_LW_(mi_, half_, true) } else { PP_LW_(mi_, half_, false) }
#define PP_LW_(mi_, half_, B_)                                                                          \
          {                                                                                             \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) \
            _Pragma("unroll") for (int r2 = 0; r2 < 2; ++r2) {                                          \
              f32x2 xy = mk2(acc[mi_][ni][8 * (half_) + 4 * kk + 2 * r2], acc[mi_][ni][8 * (half_) + 4 * kk + 2 * r2 + 1]); \
              if (B_) { const f32x2 b2 = mk2(bcol[ni], bcol[ni]); xy = xy + b2; }                       \
              const float x = xy[0];                                                                    \
              const float y = xy[1];                                                                    \
              const unsigned wa = kk ? wa1 : wa0;                                                       \
              if (ni == 0 && r2 == 0) asm volatile("ds_write2_b32 %0, %1, %2 offset1:64" :: "v"(wa), "v"(x), "v"(y) : "memory"); \
              if (ni == 0 && r2 == 1) asm volatile("ds_write2_b32 %0, %1, %2 offset0:128 offset1:192" :: "v"(wa), "v"(x), "v"(y) : "memory"); \
              if (ni == 1 && r2 == 0) asm volatile("ds_write2_b32 %0, %1, %2 offset0:32 offset1:96" :: "v"(wa), "v"(x), "v"(y) : "memory"); \
              if (ni == 1 && r2 == 1) asm volatile("ds_write2_b32 %0, %1, %2 offset0:160 offset1:224" :: "v"(wa), "v"(x), "v"(y) : "memory"); \
            }                                                                                           \
          }
#define PP_LR(p_)                                                                                       \
          if constexpr (HAS_PRE) {                                                                      \
            const unsigned prd__ = PF ? pf_rd + (unsigned)((((p_) < 4 ? pa : pa ^ 1) * PP_BUF + ((p_) & 3) * PP_REGION) * 2) \
                                      : pre_rd + 2 * (p_) * 1024;                                       \
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\t"                     \
                         "ds_read_b128 %2, %6 offset:2048\n\tds_read_b128 %3, %6 offset:2064\n\t"       \
                         "ds_read_b128 %4, %7\n\tds_read_b128 %5, %7 offset:1024"                       \
                         : "=&v"(s00), "=&v"(s01), "=&v"(s10), "=&v"(s11), "=&v"(pre[0]), "=&v"(pre[1]) \
                         : "v"(stg_rd), "v"(prd__) : "memory");                                         \
          } else                                                                                        \
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"                     \
                         "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:2064"           \
                         : "=&v"(s00), "=&v"(s01), "=&v"(s10), "=&v"(s11) : "v"(stg_rd) : "memory");      \
          if constexpr (HAS_SC)                        /* source le = block row mod 64, source register by the block half */ \
            asm volatile("ds_bpermute_b32 %0, %2, %3 offset:%4\n\tds_bpermute_b32 %1, %2, %3 offset:%5"   \
                         : "=&v"(scu[0]), "=&v"(scu[1]) : "v"(bp_addr), "v"((p_) < 4 ? scl[0] : scl[1]),   \
                           "n"(((16 * (p_)) & 63) * 4), "n"(((16 * (p_) + 8) & 63) * 4) : "memory")
#define PP_LWAITF(n_)                                                                                   \
          if constexpr (HAS_PRE) { PP_LWAIT(n_, "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11), "+v"(pre[0]), "+v"(pre[1]), "+v"(scu[0]), "+v"(scu[1])); } \
          else { PP_LWAIT(n_, "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11), "+v"(scu[0]), "+v"(scu[1])); }
#define PP_LF(p_)                                                                                       \
          {                                                                                             \
            _Pragma("clang fp contract(off)")        /* scale, then residual: two roundings as in the general passes (bit-identical) */ \
            float v[2][8];                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
              v[0][j] = s00[j]; v[0][4 + j] = s01[j]; v[1][j] = s10[j]; v[1][4 + j] = s11[j];           \
            }                                                                                           \
            const char* const base = lean_cb + (long)(16 * (p_)) * ldcb;                                \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                             \
              if constexpr (HAS_ACT) {               /* GELU, second output = its derivative */        \
                float gp[8];                                                                            \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) gelu_erf_both(v[u][j], v[u][j], gp[j]);   \
                st16_nt_s(lean_c2b + (long)(16 * (p_)) * ldc2b, vg0 + (u ? 8u * ldc2b : 0u), pack8(gp)); \
              }                                                                                         \
              if constexpr (PRE == PRE_MUL) {        /* the block holds gelu'(x) itself */              \
                const uint32_t w[4] = {pre[u][0], pre[u][1], pre[u][2], pre[u][3]};                     \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
                  v[u][2 * j] *= __uint_as_float(w[j] << 16);                                           \
                  v[u][2 * j + 1] *= __uint_as_float(w[j] & 0xffff0000u);                               \
                }                                                                                       \
              }                                                                                         \
              if constexpr (HAS_SC) { _Pragma("unroll") for (int j = 0; j < 8; ++j) v[u][j] *= scu[u]; } \
              if constexpr (pre_res) {               /* no split rows in a lean block */                \
                const uint32_t w[4] = {pre[u][0], pre[u][1], pre[u][2], pre[u][3]};                     \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
                  v[u][2 * j] += __uint_as_float(w[j] << 16);                                           \
                  v[u][2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);                               \
                }                                                                                       \
              }                                                                                         \
              const unsigned voff = vf0 + (u ? 8u * ldcb : 0u) +                                        \
                                    (((ln >> 3) + 8 * u >= lean_lb - 16 * (p_)) ? lean_skipb : 0u);   \
              st16_nt_s(base, voff, pack8(v[u]));                                                       \
            }                                                                                           \
            if constexpr (PF) {                      /* region p_ is read (the wait above): it takes the next tile's operands */ \
              if ((p_) < 7 && more) issue((p_) & 3, nk + ((p_) >> 2));                                  \
            }                                                                                           \
          }
          PP_LW(0, 0) PP_LR(0);
          PP_LW(0, 1) PP_LWAITF(8) PP_LF(0) PP_LR(1);
          PP_LW(1, 0) PP_LWAITF(8) PP_LF(1) PP_LR(2);
          PP_LW(1, 1) PP_LWAITF(8) PP_LF(2) PP_LR(3);
          PP_LW(2, 0) PP_LWAITF(8) PP_LF(3) PP_LR(4);
          PP_LW(2, 1) PP_LWAITF(8) PP_LF(4) PP_LR(5);
          PP_LW(3, 0) PP_LWAITF(8) PP_LF(5) PP_LR(6);
          PP_LW(3, 1) PP_LWAITF(8) PP_LF(6) PP_LR(7);
          PP_LWAITF(0) PP_LF(7)
#undef PP_LW
#undef PP_LW_
#undef PP_LR
#undef PP_LWAITF
#undef PP_LF
        }
      } else if constexpr (PAIRS) {
        unsigned* const stgw = reinterpret_cast<unsigned*>(stg);
        const int p2 = le >> 3;
        const unsigned pair_rd = (unsigned)(unsigned long)(lds_char*)(reinterpret_cast<char*>(stg) +
                                                                       (p2 * 64 + 4 * ((p2 >> 1) & 1) + 8 * (p2 >> 2) + (le & 7) * 8) * 4);
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define PP_EPI2B(p_, mi_, half_)                                                                        \
        {                                                                                               \
          const int col = le & 31, hi = le >> 5;                                                    \
          _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int k = 0; k < 4; ++k) { \
            /* registers 2k, 2k+1 of this half: rows {0,2,8,10}[k] + 4 hi (+1) of the pass's 16 */     \
            const int rp = ((2 * k) & 3) / 2 + 4 * ((2 * k) >> 2) + 2 * hi;                             \
            union { bf16x2 v; unsigned u; } w;                                                          \
            w.v[0] = (__bf16)(acc[mi_][ni][8 * (half_) + 2 * k] + bcol[ni]);                            \
            w.v[1] = (__bf16)(acc[mi_][ni][8 * (half_) + 2 * k + 1] + bcol[ni]);                        \
            stgw[rp * 64 + 4 * ((rp >> 1) & 1) + 8 * (rp >> 2) + ni * 32 + col] = w.u;                  \
          }                                                                                             \
          lgkm0();                                                                                      \
          __builtin_amdgcn_wave_barrier();                                                              \
          u32x4 w0, w1;                                                                                 \
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"  \
                       : "=&v"(w0), "=&v"(w1) : "v"(pair_rd) : "memory");                               \
          __builtin_amdgcn_sched_barrier(0);                                                            \
          _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                               \
            const unsigned sel = u ? 0x07060302u : 0x05040100u;    /* high / low halves of (second, first) */ \
            u32x4 o;                                                                                    \
            o[0] = __builtin_amdgcn_perm(w0[1], w0[0], sel);                                            \
            o[1] = __builtin_amdgcn_perm(w0[3], w0[2], sel);                                            \
            o[2] = __builtin_amdgcn_perm(w1[1], w1[0], sel);                                            \
            o[3] = __builtin_amdgcn_perm(w1[3], w1[2], sel);                                            \
            const int m = em0 + 16 * (p_) + 2 * p2 + u;                                                 \
            const bool ok = col_ok && m < ep.M && dbg != 2;                                             \
            const bool split = ep.split_row > 0 && m >= ep.split_row;                                   \
            if (ok) {                                                                                   \
              if (split) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16raw*>(ep.Csplit) + (long)(m - ep.split_row) * ep.ldsplit + en) = o; \
              else __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(reinterpret_cast<bf16raw*>(ep.C) + tile_map_row(cm, ep.cmap, m) * ep.ldc + en)); \
            }                                                                                           \
          }                                                                                             \
          __builtin_amdgcn_wave_barrier();                                                              \
        }
        PP_EPI2B(0, 0, 0) PP_EPI2B(1, 0, 1) PP_EPI2B(2, 1, 0) PP_EPI2B(3, 1, 1)
        PP_EPI2B(4, 2, 0) PP_EPI2B(5, 2, 1) PP_EPI2B(6, 3, 0) PP_EPI2B(7, 3, 1)
#undef PP_EPI2B
      } else {
        PP_EPI2(0, 0, 0) PP_EPI2(1, 0, 1) PP_EPI2(2, 1, 0) PP_EPI2(3, 1, 1)
        PP_EPI2(4, 2, 0) PP_EPI2(5, 2, 1) PP_EPI2(6, 3, 0) PP_EPI2(7, 3, 1)
      }
#undef PP_EPI2
      stamp(6);
      if constexpr (HAS_PRE && !PF) {
        __builtin_amdgcn_s_barrier();               // every wave has read its residual block: the ring may be refilled
        if (more) { set_tile(t, 0); m0 = m0s; n0 = n0s; prologue(); }
      }
      stamp(7);
      ++trace_tile;
      if (!more) break;
      if constexpr (CONT) {                         // the next tile's K tiles 0 and 1 are in the ring / on their way
        abase += (long)nk * (PP_BK * 2); bbase += (long)nk * (PP_BK * 2);
        if constexpr (CF) {
          // every (pass, half-row) group of this wave had a row inside M and the wave a column inside N: NST stores went
          // out (rows em0 + 8 g + [0, 8), g = 0 .. 15; a store with an empty execution mask is not issued); the timeline and
          // the store-free diagnostic mode add or drop vector memory operations
          const bool second = !HAS_ACT || ep.act == 2 || ep.C2 != nullptr;
          relax = em0 + 120 < ep.M && en0 < ep.N && second && trace == nullptr && (dbg == 0 || dbg == 4 || dbg == 5 || dbg == 6) && nk >= 3;
          prev_rolled = rolled;
        }
        m0 = m0s; n0 = n0s;
        par ^= nk & 1;
        if constexpr (CF) { t = t_next; t_next = t_nn; }
      }
    }
  }
  check_out();
#undef PP_KTILE
#undef PP_LWAIT
#undef PP_PWAIT
#undef PP_PW
#undef PP_PW_
#undef PP_PR
#undef PP_PF
#undef PP_ISS_COND
#undef PP_ISS_ALWAYS
#undef PP_ISS_PRE
#undef PP_READ_A
#undef PP_READ_B
#undef PP_MMA
#undef PP_BAR
}

template <bool EPI2, int PRE, bool HAS_SC, bool HAS_ACT, bool CONT, bool ROLL = false>
static int launch_pp_t(const vtx_gemm_desc* d, const EpiParams& ep, hipStream_t st, const Options& cfg) {
  static std::atomic<unsigned long long> attr_set{0};
  if (first_launch_on_device(attr_set)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_pp_kernel<EPI2, PRE, HAS_SC, HAS_ACT, CONT, ROLL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        PP_LDS_BYTES);
  }
  const int tiles_m = cdiv(d->M, PP_BM), tiles_n = cdiv(d->N, PP_BN);
  // column tiles per group (tools/gemm_cg.py, M = 100352: 1 is 15 % slower at N = 3072, 3..8 are within noise;
  // K = 3072 wants >= 3): about 6 MB of weight panels, between 3 and 6 tiles
  int cg = cfg.pp_cg ? cfg.pp_cg : (int)(6291456L / (512L * d->K));
  if (!cfg.pp_cg && cg < 3) cg = 3;
  if (!cfg.pp_cg && cg > 6) cg = 6;
  if (cg < 1) cg = 1;
  hipLaunchKernelGGL((gemm_nt_bf16_pp_kernel<EPI2, PRE, HAS_SC, HAS_ACT, CONT, ROLL>), dim3(cfg.pp_grid), dim3(PP_THREADS), PP_LDS_BYTES, st, d->M, d->N, d->K,
                     (const bf16raw*)d->A, d->lda, d->amap, (const bf16raw*)d->B, d->ldb, tiles_n, tiles_m * tiles_n, cg,
                     (int*)d->workspace, EPI2 ? reinterpret_cast<long long*>(cfg.pp_trace) : nullptr, cfg.pp_epi, ep);
  return check_launch("gemm_nt_pp");
}

static int launch_pp(const vtx_gemm_desc* d, const EpiParams& ep, hipStream_t st) {
  const Options& cfg = options();
  if (cfg.pp_epi == 1) return launch_pp_t<false, PRE_NONE, false, false, false>(d, ep, st, cfg);   // generic epilogue: any combination
  // specialised on what the epilogue has to read, so that a plain GEMM carries no prefetch registers and no branch
  // sits inside the passes; the activation and the GELU' / multiplier inputs are only instantiated without
  // residual / row scale (the FFN's Linears: vtx_gemm_nt routes other combinations to the non-persistent kernels)
  const bool sc = d->row_scale != nullptr;
  // continuous flow: row maps that cross a group boundary at most once per 256-row tile (what its 32-bit row
  // offsets assume)
  // ... and at least three K tiles: the continuous operand flow draws and hands over the tile index in K tile 1 and reads
  // it in K tile 2
  const bool cont = cfg.pp_cont != 0 && (d->amap.grp <= 0 || d->amap.grp >= 256) && d->K / PP_BK >= 3;
  if (d->act) return cont ? launch_pp_t<true, PRE_NONE, false, true, true>(d, ep, st, cfg) : launch_pp_t<true, PRE_NONE, false, true, false>(d, ep, st, cfg);
  if (d->dgelu_in) {
    if (d->dgelu_kind == 1)
      return cont && 256L * d->ld_dgelu * 2 < (1L << 31) ? launch_pp_t<true, PRE_MUL, false, false, true>(d, ep, st, cfg) : launch_pp_t<true, PRE_MUL, false, false, false>(d, ep, st, cfg);
    return launch_pp_t<true, PRE_DGELU, false, false, false>(d, ep, st, cfg);
  }
  if (d->R) {
    // residual-block flow: no periodic residual, a row map with at most one group boundary per tile, 32-bit offsets
    const bool pf = cont && d->r_period <= 0 && (d->rmap.grp <= 0 || d->rmap.grp >= 256) && d->rmap.skip >= 0 &&
                    (256L + (d->rmap.skip > 0 ? d->rmap.skip : 0)) * d->ldr * 2 < (1L << 31);
    if (!pf) return sc ? launch_pp_t<true, PRE_RES, true, false, false>(d, ep, st, cfg) : launch_pp_t<true, PRE_RES, false, false, false>(d, ep, st, cfg);
    if (sc) return cont ? launch_pp_t<true, PRE_RES, true, false, true>(d, ep, st, cfg) : launch_pp_t<true, PRE_RES, true, false, false>(d, ep, st, cfg);
    return cont ? launch_pp_t<true, PRE_RES, false, false, true>(d, ep, st, cfg) : launch_pp_t<true, PRE_RES, false, false, false>(d, ep, st, cfg);
  }
  if (sc) return cont ? launch_pp_t<true, PRE_NONE, true, false, true>(d, ep, st, cfg) : launch_pp_t<true, PRE_NONE, true, false, false>(d, ep, st, cfg);
  // rolling epilogue (pp_epi = 6; NOT the default: measured on the three plain shapes of a layer, interleaved same-process A/B,
  // profiles/round4_nt_roll_ab.txt: +0.3 / +0.6 / +1.0 % over the lean passes behind the loop, GRBM_GUI_ACTIVE unchanged -- the
  // passes are bound by instruction issue, and the issue slots they take inside the last K tile stretch its sections by what
  // the shorter epilogue saves): every tile takes the lean stages, so what they assume is checked here for the whole launch -- a closed-form C
  // map with at most one group boundary per tile and a byte skip below 2^30, no split rows, four K tiles or more, whole 64-column
  // blocks (rows beyond M are dropped by the range check of the stores' buffer descriptor)
  const bool roll = cont && cfg.pp_epi == 6 && d->K / PP_BK >= 4 && d->N % 64 == 0 && d->split_row <= 0 && closed_form(d->cmap) &&
                    (d->cmap.grp <= 0 || d->cmap.grp >= 256) && d->cmap.skip >= 0 && d->ldc < (1L << 22) &&
                    (long)(d->cmap.grp > 0 ? d->cmap.skip : 0) * d->ldc * 2 < (1L << 30);
  if (roll) return launch_pp_t<true, PRE_NONE, false, false, true, true>(d, ep, st, cfg);
  return cont ? launch_pp_t<true, PRE_NONE, false, false, true>(d, ep, st, cfg) : launch_pp_t<true, PRE_NONE, false, false, false>(d, ep, st, cfg);
}

// ------------------------------------------------------------------ fp32 kernel
constexpr int BK32 = 16;
constexpr int LD32 = BK32 + 1;

__global__ __launch_bounds__(NT_THREADS) void gemm_nt_f32_kernel(
    int M, int N, int K, const float* __restrict__ A, long lda, vtx_rowmap amap,
    const float* __restrict__ B, long ldb, int tiles_n, EpiParams ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);        // [2][128][17]
  float* Bs = As + 2 * BM * LD32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // loader: 128 rows x 4 float4 chunks = 512 chunks -> 2 per thread per operand
  const int lc = tid & 3, lr = tid >> 2;
  const float* ap[2];
  const float* bp[2];
  int lds_off[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = lr + 64 * it;
    int ma = m0 + row; if (ma >= M) ma = M - 1;
    int nb = n0 + row; if (nb >= N) nb = N - 1;
    ap[it] = A + map_row(amap, ma) * lda;
    bp[it] = B + (long)nb * ldb;
    lds_off[it] = row * LD32 + lc * 4;
  }
  float4 ra[2], rb[2];
  // chunks beyond K read the row's first chunk (always valid) and are zeroed
#define GLOAD32(k0_)                                                                          \
  {                                                                                           \
    const int k0__ = (k0_);                                                                   \
    const bool ok = (k0__ + lc * 4) < K;                                                      \
    const int ko__ = ok ? k0__ + lc * 4 : 0;                                                  \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                        \
      ra[it] = *reinterpret_cast<const float4*>(ap[it] + ko__);                               \
      rb[it] = *reinterpret_cast<const float4*>(bp[it] + ko__);                               \
      if (!ok) { ra[it] = make_float4(0, 0, 0, 0); rb[it] = make_float4(0, 0, 0, 0); }        \
    }                                                                                         \
  }
#define LSTORE32(buf_)                                                                        \
  {                                                                                           \
    const int b__ = (buf_);                                                                   \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                        \
      float* a = As + b__ * BM * LD32 + lds_off[it];                                          \
      a[0] = ra[it].x; a[1] = ra[it].y; a[2] = ra[it].z; a[3] = ra[it].w;                     \
      float* b = Bs + b__ * BN * LD32 + lds_off[it];                                          \
      b[0] = rb[it].x; b[1] = rb[it].y; b[2] = rb[it].z; b[3] = rb[it].w;                     \
    }                                                                                         \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_off[i] = (wm * 64 + i * 32 + (lane & 31)) * LD32 + (lane >> 5);
    b_off[i] = (wn * 64 + i * 32 + (lane & 31)) * LD32 + (lane >> 5);
  }

  const int nk = (K + BK32 - 1) / BK32;
  GLOAD32(0);
  LSTORE32(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) GLOAD32((kt + 1) * BK32);
    const float* Ab = As + buf * BM * LD32;
    const float* Bb = Bs + buf * BN * LD32;
#pragma unroll
    for (int ks = 0; ks < BK32 / 2; ++ks) {
      float af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { af[i] = Ab[a_off[i] + ks * 2]; bfr[i] = Bb[b_off[i] + ks * 2]; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) LSTORE32(buf ^ 1);
    __syncthreads();
  }
#undef GLOAD32
#undef LSTORE32
  float* stage = reinterpret_cast<float*>(smem) + wave * 64 * STAGE_LD;
  stage_acc(stage, acc, lane);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  epilogue<float>(ep, stage, m0 + wm * 64, n0 + wn * 64, lane);
}

}  // namespace vtx


using namespace vtx;

extern "C" size_t vtx_gemm_nt_workspace(void) { return 9 * 64; }   // 8 per-XCD tile counters + 1 check-out counter, one 64-B line each

extern "C" int vtx_gemm_nt(const vtx_gemm_desc* d, void* stream) {
  VTX_REQUIRE(d != nullptr, VTX_EINVAL, "gemm_nt: null descriptor");
  VTX_REQUIRE(d->M >= 0 && d->N > 0 && d->K > 0, VTX_EINVAL, "gemm_nt: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  if (d->M == 0) return VTX_OK;
  VTX_REQUIRE(d->N % 8 == 0 && d->K % 8 == 0, VTX_EINVAL, "gemm_nt: N=%d and K=%d must be multiples of 8", d->N, d->K);
  VTX_REQUIRE(d->A && d->B && d->C, VTX_EINVAL, "gemm_nt: null operand");
  VTX_REQUIRE(d->dtype == VTX_F32 || d->dtype == VTX_BF16, VTX_EINVAL, "gemm_nt: bad dtype %d", d->dtype);
  const long vec = d->dtype == VTX_BF16 ? 8 : 4;
  VTX_REQUIRE(aligned16(d->A) && aligned16(d->B) && aligned16(d->C) && d->lda % vec == 0 && d->ldb % vec == 0 &&
                  d->ldc % vec == 0, VTX_EALIGN, "gemm_nt: operands must be 16-byte aligned");
  VTX_REQUIRE(!d->bias || aligned16(d->bias), VTX_EALIGN, "gemm_nt: bias not aligned");
  VTX_REQUIRE(!d->R || (aligned16(d->R) && d->ldr % vec == 0), VTX_EALIGN, "gemm_nt: residual not aligned");
  VTX_REQUIRE(!(d->act && d->C2) || (aligned16(d->C2) && d->ldc2 % vec == 0), VTX_EALIGN, "gemm_nt: C2 not aligned");
  VTX_REQUIRE(d->act != 2 || d->C2, VTX_EINVAL, "gemm_nt: act 2 needs C2 (the derivative output)");
  VTX_REQUIRE(d->dgelu_kind == 0 || d->dgelu_kind == 1, VTX_EINVAL, "gemm_nt: bad dgelu_kind %d", d->dgelu_kind);
  VTX_REQUIRE(!d->dgelu_in || (aligned16(d->dgelu_in) && d->ld_dgelu % vec == 0), VTX_EALIGN, "gemm_nt: dgelu_in not aligned");
  VTX_REQUIRE(d->split_row <= 0 || (d->Csplit && aligned16(d->Csplit) && d->ldsplit % vec == 0), VTX_EINVAL,
              "gemm_nt: split_row needs an aligned Csplit");
  VTX_REQUIRE(!d->row_scale || (d->rs_d1 > 0 && d->rs_d2 > 0), VTX_EINVAL, "gemm_nt: row_scale divisors must be > 0");
  VTX_REQUIRE(d->act >= 0 && d->act <= 2, VTX_EINVAL, "gemm_nt: bad act %d", d->act);
  VTX_REQUIRE(closed_form(d->amap), VTX_EINVAL, "gemm_nt: the A row map must be in closed form (no table)");
  VTX_REQUIRE((closed_form(d->cmap) || d->cmap.grp > 0) && (closed_form(d->rmap) || d->rmap.grp > 0), VTX_EINVAL,
              "gemm_nt: a table row map needs grp > 0");

  EpiParams ep;
  ep.M = d->M; ep.N = d->N;
  ep.C = d->C; ep.ldc = d->ldc; ep.cmap = d->cmap;
  ep.bias = d->bias; ep.act = d->act; ep.C2 = d->C2; ep.ldc2 = d->ldc2;
  ep.dgelu_in = d->dgelu_in; ep.ld_dgelu = d->ld_dgelu; ep.dgelu_kind = d->dgelu_kind;
  ep.row_scale = d->row_scale; ep.rs_d1 = d->rs_d1; ep.rs_m1 = d->rs_m1; ep.rs_d2 = d->rs_d2; ep.rs_m2 = d->rs_m2;
  ep.R = d->R; ep.ldr = d->ldr; ep.rmap = d->rmap; ep.r_period = d->r_period;
  ep.split_row = d->split_row; ep.Csplit = d->Csplit; ep.ldsplit = d->ldsplit;
  {
    const FastDiv f1 = make_fast_div(d->row_scale ? (unsigned)d->rs_d1 : 1u), f2 = make_fast_div(d->row_scale ? (unsigned)d->rs_d2 : 1u);
    ep.rs_magic1 = f1.magic; ep.rs_shift1 = f1.shift; ep.rs_magic2 = f2.magic; ep.rs_shift2 = f2.shift;
  }

  const int tiles_m = cdiv(d->M, BM), tiles_n = cdiv(d->N, BN);
  dim3 grid(tiles_m * tiles_n), block(NT_THREADS);
  hipStream_t st = as_stream(stream);
  if (d->dtype == VTX_BF16) {
    const size_t lds = STAGE_BYTES > 4 * BM * BK16 * 2 ? STAGE_BYTES : 4 * BM * BK16 * 2;
    const Options& o = options();
    const bool dma_ok = d->K % BK16 == 0 && !o.gemm_nodma;
    // default: the persistent 256x256 ping-pong kernel for the big activations GEMMs; the 256x128 ring (72 KB of
    // LDS, two co-resident workgroups per CU) for 1024 <= M < 2048; the two-buffer 128x128 kernel below that
    const int variant = o.gemm_nt != NT_AUTO ? o.gemm_nt : (d->M >= 2048 ? NT_PP256 : d->M >= 1024 ? NT_RING256X3K32 : NT_DMA2);
    const int nkt = d->K / BK16;
    // the ping-pong kernel prefetches the residual and the GELU' input into the same registers, and keeps its
    // tile counters in the caller's workspace
    const bool combo_ok = o.pp_epi == 1 || ((!d->act || !(d->R || d->dgelu_in || d->row_scale)) &&
                                            (!d->dgelu_in || !(d->R || d->row_scale)));
    // 32-bit byte offsets inside a tile: its 256 rows (plus the rows a row map skips inside it) must span < 2 GB
    const long span_rows = 256 + (d->amap.grp > 0 ? (256 / d->amap.grp + 2) * (long)(d->amap.skip > 0 ? d->amap.skip : 0) : 0);
    const bool span_ok = span_rows * d->lda * 2 < (1L << 31) && 256L * d->ldb * 2 < (1L << 31) && d->amap.skip >= 0;
    const bool pp_ok = dma_ok && nkt >= 2 && combo_ok && span_ok && d->workspace && d->ws_bytes >= vtx_gemm_nt_workspace();
    if (variant == NT_PP256 && pp_ok) return launch_pp(d, ep, st);
    if (dma_ok && nkt >= 3 && (variant == NT_RING256X3)) return launch_ring<4, 3, 64>(d, ep, st);
    if (dma_ok && nkt >= 3 && (variant == NT_RING256X3K32 || variant == NT_PP256)) return launch_ring<4, 3, 32>(d, ep, st);
    if (dma_ok && nkt >= 3 && variant == NT_RING256X4K32) return launch_ring<4, 4, 32>(d, ep, st);
    if (dma_ok && nkt >= 3 && variant == NT_RING128X3) return launch_ring<2, 3, 64>(d, ep, st);
    if (dma_ok && nkt >= 3 && variant == NT_RING128X4K32) return launch_ring<2, 4, 32>(d, ep, st);
    if (dma_ok)
      hipLaunchKernelGGL(gemm_nt_bf16_dma_kernel, grid, block, lds, st, d->M, d->N, d->K, (const bf16raw*)d->A, d->lda,
                         d->amap, (const bf16raw*)d->B, d->ldb, tiles_n, ep);
    else
      hipLaunchKernelGGL(gemm_nt_bf16_kernel, grid, block, lds, st, d->M, d->N, d->K, (const bf16raw*)d->A, d->lda,
                         d->amap, (const bf16raw*)d->B, d->ldb, tiles_n, ep);
  } else {
    const size_t need = (size_t)4 * BM * LD32 * 4;
    const size_t lds = STAGE_BYTES > need ? STAGE_BYTES : need;
    hipLaunchKernelGGL(gemm_nt_f32_kernel, grid, block, lds, st, d->M, d->N, d->K, (const float*)d->A, d->lda,
                       d->amap, (const float*)d->B, d->ldb, tiles_n, ep);
  }
  return check_launch("gemm_nt");
}
