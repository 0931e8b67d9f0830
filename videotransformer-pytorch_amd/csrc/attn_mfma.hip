// attn_mfma.hip -- bf16 MFMA attention core for sequences of 33..256 tokens, head_dim 64
// (the divided *spatial* attention of TimeSformer: L = 197; ViViT's spatial encoder).
//
// Reference: transformer.py:167-174 (q k^T * scale -> softmax -> @ v) plus the
// '(b t) p d' regrouping / cls replication of :352-356,:375 done by row arithmetic
// (in_row / out_row, shared with the VALU kernels in attn.hip).
//
// One workgroup (4 waves) per (sequence, head).  Whole K and V ([Lp][64] bf16, Lp = L
// rounded up to 32) sit in LDS; each wave owns 32-row query tiles.  All products are
// v_mfma_f32_32x32x16_bf16.  The score tile is computed TRANSPOSED (S^T = K Q^T) so that
// in the MFMA C/D layout a lane owns one query column: softmax row statistics are
// in-lane reductions plus one cross-half shuffle, and the probabilities are already
// laid out as the B operand of O^T = V^T P^T -- no LDS round trip for P.  The 8-key
// groups a lane holds are {0-3, 8-11} (lower half) / {4-7, 12-15} (upper half) of each
// 16-key step; the V^T / K^T / Q^T / dO^T operands are fetched with the hardware
// transpose read ds_read_b64_tr_b16 using the SAME key permutation, so no cross-lane
// exchange is needed (a contraction is invariant to a consistent permutation of k).
//
// LDS tiles are [rows][64] bf16 with the 16-byte chunk index XOR-ed by
// rev3((row>>1)&7): conflict-free both for the row-wise ds_read_b128 operand reads and
// for the 4-row x 16-column blocks of the transpose reads.
//
// Backward = two phases, mirroring the two contractions:
//   dq : lanes = queries (S^T layout);  dQ^T += K^T dS^T          (+ delta)
//   dkv: lanes = keys    (S layout);    dV^T += dO^T P, dK^T += Q^T dS
// run by ONE persistent kernel for up to 224 tokens (all four operand tiles stay in LDS: one pass over HBM,
// attn_bwd_fused_mfma_kernel) and by two kernels (each with two tiles in LDS) for 225..256 tokens or option attn_fused=0;
// the two forms are bit-identical.  FLOPs per (sequence, head): fwd 4*Lp^2*64, bwd 14*Lp^2*64 (incl. recompute).
#include "attn_common.h"

namespace vtx {

__device__ inline long m_in_row(const AttnP& p, int s, int i) { return in_row(p, s, i); }
__device__ inline long m_out_row(const AttnP& p, int s, int i) { return out_row(p, s, i); }

constexpr int MA_THREADS = 256;
constexpr int MA_MAXT = 8;            // up to 8 tiles of 32 rows (Lp <= 256)
constexpr int MA_KB = 4;              // key tiles per online-softmax block in the forward kernel
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ inline int sw_of(int row) {
  const int x = (row >> 1) & 7;
  return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1);
}
// element offset of (row, col) in a swizzled [rows][64] bf16 tile
__device__ inline int sw_off(int row, int col) { return row * 64 + ((((col >> 3) ^ sw_of(row))) << 3) + (col & 7); }

// Fill TWO swizzled LDS tiles (rows gathered through rowfn0 / rowfn1, zero rows beyond nvalid, up to
// Lp <= 256).  All global loads of both tiles are issued before the first LDS store: a load->store
// loop serialises one memory latency per iteration (measured: ~25 us of a 50 us workgroup).
template <int NTHR = 256, typename RowFn0, typename RowFn1>
__device__ inline void fill_tiles2(bf16raw* lds0, bf16raw* lds1, int Lp, int nvalid, const bf16raw* base0, long ld0, int col0,
                                   RowFn0 rowfn0, const bf16raw* base1, long ld1, int col1, RowFn1 rowfn1) {
  constexpr int NI = 2048 / NTHR;
  uint4 v0[NI], v1[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int id = threadIdx.x + i * NTHR;
    const int r = id >> 3, c = id & 7;
    v0[i] = make_uint4(0, 0, 0, 0);
    v1[i] = make_uint4(0, 0, 0, 0);
    if (id < Lp * 8 && r < nvalid) {
      v0[i] = *reinterpret_cast<const uint4*>(base0 + rowfn0(r) * ld0 + col0 + c * 8);
      v1[i] = *reinterpret_cast<const uint4*>(base1 + rowfn1(r) * ld1 + col1 + c * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int id = threadIdx.x + i * NTHR;
    const int r = id >> 3, c = id & 7;
    if (id < Lp * 8) {
      const int off = r * 64 + ((c ^ sw_of(r)) << 3);
      *reinterpret_cast<uint4*>(lds0 + off) = v0[i];
      *reinterpret_cast<uint4*>(lds1 + off) = v1[i];
    }
  }
}

// Row-wise operand fragment from LDS: lane (l&31) -> row, 8 consecutive columns ks*16 + 8*(l>>5)..
__device__ inline bf16x8 frag_rows(const bf16raw* lds, int row0, int ks, int lane) {
  const int row = row0 + (lane & 31);
  const int c = ks * 2 + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(lds + row * 64 + ((c ^ sw_of(row)) << 3));
}

// Transposed operand fragment: A[i = column col0 + (l&31)][k] with k running over the 16 rows
// row0..row0+15 in the permuted order {0-3, 8-11 | 4-7, 12-15} (lower | upper half-wave).
__device__ inline bf16x8 frag_cols(const bf16raw* lds, int row0, int col0, int lane) {
  const int r = row0 + 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int c = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + sw_off(r, c)));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + sw_off(r + 8, c)));
  return u.v;
}

// The same two readers with their per-lane part precomputed.  For a row0 that is a multiple of 16 the swizzle term only
// depends on the lane, so a fragment address is (lane constant) + row0 * 128 B: the kernels keep the lane constants in
// registers (FragOff) and the compiler turns row0 into an immediate / one scalar add -- the generic readers above cost
// 3 (rows) to 6 (columns) vector instructions of index arithmetic per LDS read, in kernels bound by vector-ALU issue.
struct FragOff {
  int rows[4];        // frag_rows: element offset of (row = lane&31, ks)
  int cols[2][2];     // frag_cols: element offset of (n2, half = rows r / r + 8), col0 = n2 * 32
};
__device__ inline FragOff make_frag_off(int lane) {
  FragOff f;
  const int row = lane & 31;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f.rows[ks] = row * 64 + (((ks * 2 + (lane >> 5)) ^ sw_of(row)) << 3);
  const int r = 4 * (lane >> 5) + ((lane & 15) >> 2);
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2) {
    const int c = n2 * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    f.cols[n2][0] = sw_off(r, c);
    f.cols[n2][1] = sw_off(r + 8, c);
  }
  return f;
}
__device__ inline bf16x8 frag_rows_o(const bf16raw* lds, int row0, int ks, const FragOff& f) {     // row0 % 32 == 0
  return *reinterpret_cast<const bf16x8*>(lds + row0 * 64 + f.rows[ks]);
}
__device__ inline bf16x8 frag_cols_o(const bf16raw* lds, int row0, int n2, const FragOff& f) {     // row0 % 16 == 0
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + row0 * 64 + f.cols[n2][0]));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + row0 * 64 + f.cols[n2][1]));
  return u.v;
}

// Row addressing of one sequence in closed form: row(i) = i == 0 ? row0 : base + i * stride (contiguous sequences:
// row0 = base, stride 1; divided spatial attention: row 0 is the clip's cls row / its per-frame copy, token i >= 1 sits
// i * T rows further).  in_row / out_row (attn_common.h) compute the same rows with a division per call; per lane and row
// that was ~40 vector instructions around every fragment load and row store of kernels bound by vector-ALU issue.
struct RowLin { long row0, base, stride; };
__device__ inline RowLin lin_in(const AttnP& p, int s) {
  RowLin r;
  if (p.mode == VTX_ATTN_CONTIG) { r.base = (long)s * p.L; r.stride = 1; r.row0 = r.base; return r; }
  const int b = s / p.T, t = s - b * p.T;
  r.row0 = (long)b * (1 + (long)p.P * p.T);
  r.stride = p.T;
  r.base = r.row0 + 1 + t - r.stride;           // row(i) = row0 + 1 + (i - 1) * T + t
  return r;
}
__device__ inline RowLin lin_out(const AttnP& p, int s) {
  RowLin r;
  if (p.mode == VTX_ATTN_CONTIG) { r.base = (long)s * p.L; r.stride = 1; r.row0 = r.base; return r; }
  const int b = s / p.T, t = s - b * p.T;
  r.row0 = (long)p.B * p.P * p.T + s;
  r.stride = p.T;
  r.base = (long)b * p.P * p.T + t - r.stride;  // row(i) = b * P * T + (i - 1) * T + t
  return r;
}
__device__ inline long lin_row(const RowLin& r, int i) { return i == 0 ? r.row0 : r.base + (long)i * r.stride; }

// The four row-wise fragments (64 columns) of row `row` of a [.., ld] matrix, column origin col0; rows beyond nvalid read
// row nvalid - 1 instead (no branch, one address per row): every kernel below keeps a padded query / key in its own lane
// and never stores its results, so its operands only have to be finite.
__device__ inline void load_row_frags(bf16x8 (&f)[4], const bf16raw* base, long ld, int col0, const RowLin& rl, int row, int nvalid,
                                      int lane) {
  const int rc = row < nvalid ? row : nvalid - 1;
  const bf16raw* src = base + lin_row(rl, rc) * ld + col0 + 8 * (lane >> 5);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    union { bf16x8 v; uint4 u; } x;
    x.u = *reinterpret_cast<const uint4*>(src + ks * 16);
    f[ks] = x.v;
  }
}
// sum_j a[j] * b[j] over the 8 bf16 pairs of two fragments, accumulated into acc (v_dot2c_f32_bf16: exact products, fp32 sum)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
__device__ inline float frag_dot2(const bf16x8& a, const bf16x8& b, float acc) {
  union { bf16x8 v; bf16x2_ h[4]; } ua, ub;
  ua.v = a; ub.v = b;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_fdot2_f32_bf16(ua.h[j], ub.h[j], acc, false);
  return acc;
}

// Row-wise operand fragment straight from global memory (rows beyond nvalid read as zero).
template <typename RowFn>
__device__ inline bf16x8 frag_global(const bf16raw* base, long ld, int col0, int row, int nvalid, int ks, int lane, RowFn rowfn) {
  union { bf16x8 v; uint4 u; } x;
  x.u = make_uint4(0, 0, 0, 0);
  if (row < nvalid) x.u = *reinterpret_cast<const uint4*>(base + rowfn(row) * ld + col0 + ks * 16 + 8 * (lane >> 5));
  return x.v;
}

__device__ inline bf16x8 pack8(const float* f) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)f[j];
  return v;
}
__device__ inline float frag_dot(const bf16x8& a, const bf16x8& b) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)b[j];
  return s;
}
__device__ inline void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// row index (within a 32-row tile) that accumulator register r of this lane holds
__device__ inline int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

constexpr int MA_STAGE_ELEMS = 32 * 64;   // per-wave [32][64] bf16 staging tile for row stores (4 KB)

__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

// Store a [32 x 64] result held transposed (lane&31 = row, registers = 64 columns in two C tiles).
// The accumulator layout gives each lane 8-byte pieces of 32 different rows -- stored directly that
// is 16 B per row per instruction (measured: 47 us of a 153 us forward).  Stage through a wave-private
// swizzled LDS tile instead and write whole 128-B rows: 8 lanes x 16 B, 8 rows per instruction.
// ptr_of_row(r) -> destination of tile row r (64 bf16), or nullptr for a padded row.
template <typename PtrFn>
__device__ inline void store_rows_T(bf16raw* stg, const f32x16 (&acc)[2], float mul, int lane, PtrFn ptr_of_row) {
  const int row = lane & 31;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = nt * 32 + 8 * g + 4 * (lane >> 5);
      union { bf16x4 v; uint2 u; } w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w.v[j] = (__bf16)(acc[nt][4 * g + j] * mul);
      *reinterpret_cast<uint2*>(stg + sw_off(row, col)) = w.u;
    }
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = i * 8 + (lane >> 3), c = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(stg + r * 64 + ((c ^ sw_of(r)) << 3));
    bf16raw* dst = ptr_of_row(r);
    if (dst) *reinterpret_cast<uint4*>(dst + c * 8) = v;
  }
  wave_lds_sync();
}

// Direct variant (8-byte pieces, no staging) for the packed short-sequence kernels below.
__device__ inline void store_rows_direct(bf16raw* dst_row, const f32x16 (&acc)[2], float mul, int lane) {
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      union { bf16x4 v; uint2 u; } w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w.v[j] = (__bf16)(acc[nt][4 * g + j] * mul);
      *reinterpret_cast<uint2*>(dst_row + nt * 32 + 8 * g + 4 * (lane >> 5)) = w.u;
    }
}

// ------------------------------------------------------------------------------ forward
// One workgroup per (sequence, head): K and V tiles live in LDS, each wave owns query tiles
// qt = wave, wave+4, ...  Scores stay in registers (S^T = K Q^T so that every lane owns ONE query
// row: row statistics need a single cross-lane exchange).  The kernel is bound by HBM (128 flop/B)
// and then by the VALU softmax (v_exp_f32 is quarter rate), so per score element the loop spends
// one max, one fma, one exp2 and one add: the scale is folded into the fma and the padding mask is
// applied only to the key tile that actually holds padded keys.
// NT_ > 0: the number of 32-row tiles is a compile-time constant (7 for the L = 197 of every model here): the key / query
// tile loops inside a query / key tile are fully unrolled, so every LDS fragment address is a lane-constant register plus an
// immediate offset.  With a run-time trip count each of the twelve address registers of the loop body was copied and stepped
// by 4 KB per iteration (24 of ~88 vector instructions per key tile in the dq kernel, which is bound by vector-ALU issue).
template <int NT_>
__global__ __launch_bounds__(MA_THREADS, 2) void attn_fwd_mfma_kernel(AttnP p, const bf16raw* __restrict__ qkv,
                                                                   bf16raw* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int s = blockIdx.x, h = blockIdx.y, D = p.H * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nt = NT_ > 0 ? NT_ : (p.L + 31) >> 5, Lp = nt * 32;
  bf16raw* Ks = reinterpret_cast<bf16raw*>(sm_raw);
  bf16raw* Vs = Ks + Lp * 64;
  bf16raw* stg = Vs + Lp * 64 + wave * MA_STAGE_ELEMS;
  const RowLin li = lin_in(p, s), lo = lin_out(p, s);
  auto rowfn = [&](int i) { return lin_row(li, i); };
  bf16x8 qf[4];
  load_row_frags(qf, qkv, p.ld_qkv, h * 64, li, wave * 32 + (lane & 31), p.L, lane);
  fill_tiles2(Ks, Vs, Lp, p.L, qkv, p.ld_qkv, D + h * 64, rowfn, qkv, p.ld_qkv, 2 * D + h * 64, rowfn);
  __syncthreads();
  const float c2 = p.scale * LOG2E;
  const int ragged = (p.L & 31) ? nt - 1 : -1;     // the key tile that holds padded keys
  const FragOff fo = make_frag_off(lane);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int qt = wave; qt < nt; qt += 4) {
    bf16x8 qn[4];                                  // next query tile's fragments, in flight during this one
    load_row_frags(qn, qkv, p.ld_qkv, h * 64, li, (qt + 4) * 32 + (lane & 31), p.L, lane);
    f32x16 acc[2];
    zero16(acc[0]);
    zero16(acc[1]);
    float m = -1e30f, l = 0.f;                     // running max of the RAW scores (scale > 0)
#pragma unroll(NT_ > 0 ? (NT_ + MA_KB - 1) / MA_KB : 1)
    for (int kb = 0; kb < nt; kb += MA_KB) {
      f32x16 st[MA_KB];                            // (tiles beyond nt stay undefined: every use below is guarded)
#pragma unroll
      for (int t = 0; t < MA_KB; ++t) {
        if (kb + t < nt) {
          st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Ks, (kb + t) * 32, 0, fo), qf[0], zero, 0, 0, 0);
#pragma unroll
          for (int ks = 1; ks < 4; ++ks)
            st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Ks, (kb + t) * 32, ks, fo), qf[ks], st[t], 0, 0, 0);
        }
      }
      float bm = -1e30f;
#pragma unroll
      for (int t = 0; t < MA_KB; ++t)
        if (kb + t < nt) {
          if (kb + t == ragged) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if ((kb + t) * 32 + crow(r, lane) >= p.L) st[t][r] = -1e30f;
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
      for (int t = 0; t < MA_KB; ++t)
        if (kb + t < nt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(st[t][r], c2, -mc)); st[t][r] = e; bl += e; }
        }
      bl += __shfl_xor(bl, 32, 64);
      l = l * alpha + bl;
      if (kb > 0) {
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[n2][r] *= alpha;
      }
#pragma unroll
      for (int t = 0; t < MA_KB; ++t)
        if (kb + t < nt) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            float pf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = st[t][8 * s2 + j];
            const bf16x8 pb = pack8(pf);
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
              acc[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Vs, (kb + t) * 32 + 16 * s2, n2, fo), pb, acc[n2], 0, 0, 0);
          }
        }
    }
    const int q = qt * 32 + (lane & 31);
    store_rows_T(stg, acc, 1.0f / l, lane, [&](int r) -> bf16raw* {
      const int qq = qt * 32 + r;
      return qq < p.L ? out + lin_row(lo, qq) * p.ld_out + h * 64 : nullptr;
    });
    if (q < p.L && lane < 32) lse[((long)s * p.H + h) * p.L + q] = (m * c2) * LN2 + __logf(l);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
  }
}

// The four row-wise fragments of a 32-row tile (one row per lane) written into a swizzled [32][64] LDS tile.
__device__ inline void put_tile(bf16raw* lds, const bf16x8 (&f)[4], int lane) {
  const int row = lane & 31;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c = ks * 2 + (lane >> 5);
    *reinterpret_cast<bf16x8*>(lds + row * 64 + ((c ^ sw_of(row)) << 3)) = f[ks];
  }
}

// --------------------------------------------------------------------------- backward: dq
template <int NT_>
__global__ __launch_bounds__(MA_THREADS, 2) void attn_bwd_dq_mfma_kernel(AttnP p, const bf16raw* __restrict__ qkv,
                                                                      const bf16raw* __restrict__ o, const bf16raw* __restrict__ dout,
                                                                      const float* __restrict__ lse, float* __restrict__ delta,
                                                                      bf16raw* __restrict__ dqkv, bf16raw* __restrict__ dqkv_cls) {
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int s = blockIdx.x, h = blockIdx.y, D = p.H * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nt = NT_ > 0 ? NT_ : (p.L + 31) >> 5, Lp = nt * 32;
  bf16raw* Ks = reinterpret_cast<bf16raw*>(sm_raw);
  bf16raw* Vs = Ks + Lp * 64;
  bf16raw* stg = Vs + Lp * 64 + wave * MA_STAGE_ELEMS;
  const RowLin li = lin_in(p, s), lo = lin_out(p, s);
  auto rin = [&](int i) { return lin_row(li, i); };
  bf16x8 qf[4], df[4], of[4];                      // this wave's first query tile: Q, dO, O rows (one row per lane)
  load_row_frags(qf, qkv, p.ld_qkv, h * 64, li, wave * 32 + (lane & 31), p.L, lane);
  load_row_frags(df, dout, p.ld_dout, h * 64, lo, wave * 32 + (lane & 31), p.L, lane);
  load_row_frags(of, o, p.ld_out, h * 64, lo, wave * 32 + (lane & 31), p.L, lane);
  fill_tiles2(Ks, Vs, Lp, p.L, qkv, p.ld_qkv, D + h * 64, rin, qkv, p.ld_qkv, 2 * D + h * 64, rin);
  __syncthreads();
  const float c2 = p.scale * LOG2E;
  const int ragged = (p.L & 31) ? nt - 1 : -1;
  const FragOff fo = make_frag_off(lane);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int qt = wave; qt < nt; qt += 4) {
    const int q = qt * 32 + (lane & 31);
    bf16x8 qn[4], dn[4], on[4];                    // the next query tile's rows, in flight during this one
    load_row_frags(qn, qkv, p.ld_qkv, h * 64, li, q + 128, p.L, lane);
    load_row_frags(dn, dout, p.ld_dout, h * 64, lo, q + 128, p.L, lane);
    load_row_frags(on, o, p.ld_out, h * 64, lo, q + 128, p.L, lane);
    float dl = 0.f;                                // delta = rowsum(dO * O): this lane's 32 of the row's 64 columns
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dl = frag_dot2(df[ks], of[ks], dl);
    dl += __shfl_xor(dl, 32, 64);
    const long lidx = ((long)s * p.H + h) * p.L + q;
    float l2 = 0.f;
    if (q < p.L) {
      l2 = lse[lidx] * LOG2E;
      if (lane < 32) delta[lidx] = dl;
    }
    f32x16 acc[2];
    zero16(acc[0]);
    zero16(acc[1]);
#pragma unroll(NT_ > 0 ? NT_ : 1)
    for (int kt = 0; kt < nt; ++kt) {
      f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Ks, kt * 32, 0, fo), qf[0], zero, 0, 0, 0);
      f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Vs, kt * 32, 0, fo), df[0], zero, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Ks, kt * 32, ks, fo), qf[ks], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Vs, kt * 32, ks, fo), df[ks], dp, 0, 0, 0);
      }
      // dS = P (dP - delta); the softmax scale is applied once to dq at the store.  Padded keys have
      // zero K rows, so only their probability needs masking (they would otherwise poison nothing but
      // cost accuracy in the bf16 pack), and only in the ragged tile.
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(fmaf(st[r], c2, -l2)) * (dp[r] - dl);
      if (kt == ragged) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 32 + crow(r, lane) >= p.L) ds[r] = 0.f;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 db = pack8(ds + 8 * s2);
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
          acc[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Ks, kt * 32 + 16 * s2, n2, fo), db, acc[n2], 0, 0, 0);
      }
    }
    store_rows_T(stg, acc, p.scale, lane, [&](int r) -> bf16raw* {
      const int qq = qt * 32 + r;
      if (qq >= p.L) return nullptr;
      return (p.mode == VTX_ATTN_SPACE && qq == 0) ? dqkv_cls + (long)s * p.ld_dqkv + h * 64
                                                    : dqkv + lin_row(li, qq) * p.ld_dqkv + h * 64;
    });
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qf[ks] = qn[ks]; df[ks] = dn[ks]; of[ks] = on[ks]; }
  }
}

// -------------------------------------------------------------------------- backward: dk, dv
// VAR_ (tuning variants of the unrolled kernel, option "attn_dkv"): bit 0 = scheduling barrier at every query tile, bit 1 = no
// prefetch of the wave's next key tile
template <int NT_, int VAR_>
__global__ __launch_bounds__(MA_THREADS, 2) void attn_bwd_dkv_mfma_kernel(AttnP p, const bf16raw* __restrict__ qkv,
                                                                       const bf16raw* __restrict__ dout, const float* __restrict__ lse,
                                                                       const float* __restrict__ delta, bf16raw* __restrict__ dqkv,
                                                                       bf16raw* __restrict__ dqkv_cls) {
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int s = blockIdx.x, h = blockIdx.y, D = p.H * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nt = NT_ > 0 ? NT_ : (p.L + 31) >> 5, Lp = nt * 32;
  bf16raw* Qs = reinterpret_cast<bf16raw*>(sm_raw);
  bf16raw* Os = Qs + Lp * 64;
  bf16raw* stg = Os + Lp * 64 + wave * MA_STAGE_ELEMS;
  float* Ls = reinterpret_cast<float*>(Os + Lp * 64 + 4 * MA_STAGE_ELEMS);   // lse * log2(e), +huge on padded rows
  float* Ds = Ls + Lp;
  const RowLin li = lin_in(p, s), lo = lin_out(p, s);
  auto rin = [&](int i) { return lin_row(li, i); };
  auto rout = [&](int i) { return lin_row(lo, i); };
  const int key = wave * 32 + (lane & 31);
  bf16x8 kf[4], vf[4];
  load_row_frags(kf, qkv, p.ld_qkv, D + h * 64, li, key, p.L, lane);
  load_row_frags(vf, qkv, p.ld_qkv, 2 * D + h * 64, li, key, p.L, lane);
  fill_tiles2(Qs, Os, Lp, p.L, qkv, p.ld_qkv, h * 64, rin, dout, p.ld_dout, h * 64, rout);
  for (int i = threadIdx.x; i < Lp; i += MA_THREADS) {
    const long li = ((long)s * p.H + h) * p.L + i;
    Ls[i] = i < p.L ? lse[li] * LOG2E : 1e30f;
    Ds[i] = i < p.L ? delta[li] : 0.f;
  }
  __syncthreads();
  const float c2 = p.scale * LOG2E;
  const FragOff fo = make_frag_off(lane);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int kt = wave; kt < nt; kt += 4) {
    bf16x8 kn[4], vn[4];                           // next key tile's fragments, in flight during this one
    if (!(VAR_ & 2)) {
      load_row_frags(kn, qkv, p.ld_qkv, D + h * 64, li, key + (kt - wave + 4) * 32, p.L, lane);
      load_row_frags(vn, qkv, p.ld_qkv, 2 * D + h * 64, li, key + (kt - wave + 4) * 32, p.L, lane);
    }
    f32x16 dk[2], dv[2];
    zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
#pragma unroll(NT_ > 0 ? NT_ : 1)
    for (int qt = 0; qt < nt; ++qt) {
      // unrolled: keep the instruction scheduler from hoisting the next query tile's fragment reads over this one's tail
      // (32 more live registers: the kernel sits at the 256-register limit of two waves per SIMD and spilled)
      if (VAR_ & 1) __builtin_amdgcn_sched_barrier(0);
      f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Qs, qt * 32, 0, fo), kf[0], zero, 0, 0, 0);
      f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Os, qt * 32, 0, fo), vf[0], zero, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Qs, qt * 32, ks, fo), kf[ks], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Os, qt * 32, ks, fo), vf[ks], dp, 0, 0, 0);
      }
      // padded query rows: Ls = +huge -> P = 0; padded keys only feed dk/dv rows that are never stored
      float pr[16], ds[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qrow = qt * 32 + 8 * g + 4 * (lane >> 5);
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
        const bf16x8 db = pack8(ds + 8 * s2);
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          dv[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Os, qt * 32 + 16 * s2, n2, fo), pb, dv[n2], 0, 0, 0);
          dk[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Qs, qt * 32 + 16 * s2, n2, fo), db, dk[n2], 0, 0, 0);
        }
      }
    }
    auto base_of = [&](int r) -> bf16raw* {
      const int kk = kt * 32 + r;
      if (kk >= p.L) return nullptr;
      return (p.mode == VTX_ATTN_SPACE && kk == 0) ? dqkv_cls + (long)s * p.ld_dqkv : dqkv + lin_row(li, kk) * p.ld_dqkv;
    };
    store_rows_T(stg, dk, p.scale, lane, [&](int r) -> bf16raw* { bf16raw* b = base_of(r); return b ? b + D + h * 64 : nullptr; });
    store_rows_T(stg, dv, 1.0f, lane, [&](int r) -> bf16raw* { bf16raw* b = base_of(r); return b ? b + 2 * D + h * 64 : nullptr; });
    if (VAR_ & 2) {
      if (kt + 4 < nt) {
        load_row_frags(kf, qkv, p.ld_qkv, D + h * 64, li, key + (kt - wave + 4) * 32, p.L, lane);
        load_row_frags(vf, qkv, p.ld_qkv, 2 * D + h * 64, li, key + (kt - wave + 4) * 32, p.L, lane);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
    }
  }
}

// ------------------------------------------------------------------- backward: dq, dk, dv in one pass over HBM
// The two kernels above each read q, k, v and dO (and o) of the (sequence, head).  For up to 7 tiles (L <= 224: the 197
// of every model here) ONE workgroup of 8 waves keeps all four operands on chip: K, V, Q, dO tiles in LDS (4 x 28 KB) +
// per-wave staging, one workgroup per CU, persistent over the (sequence, head) items (heads fastest: the CUs of a round
// read whole 4.6-KB rows of qkv between them).  Wave w < nt owns query tile w in the dq phase and key tile w in the
// dk / dv phase; wave 7 never owns a tile and is the LOADER.  Per item:
//   top:    the worker's Q / dO / O row fragments (already in registers: loaded behind the previous item's last
//           products) -> delta = rowsum(dO * O), Q and dO fragments copied into the LDS tiles, lse * log2(e) and delta
//           into LDS;                                                          barrier A
//   dq:     the dq kernel's tile body on the K / V tiles; then the wave takes its own K / V fragments out of the tiles
//           (same values and lane layout as the dkv kernel's global loads);     barrier B (K / V tiles are free)
//   dk/dv:  the dkv kernel's tile body on the Q / dO tiles, WHILE the loader brings the next item's K and V tiles from
//           HBM into the free K / V space (LDS-DMA); behind its last product a worker requests
//           the next item's row fragments, then stores dk, dv;                  barrier C (Q / dO tiles are free)
// so the only exposed memory time per item is what the loader / the fragment loads do not finish under the dk/dv phase.
// The dq phase is software-pipelined by hand: the score / dP products of tile pair i + 1 are issued BEFORE the exp / pack
// work of pair i (same products, same sums, same order per accumulator; the same in the dk / dv phase costs 30 registers
// the kernel does not have and measured 1.5 %).  Results are bit-identical to the two-kernel
// path (tests/test_gpu_kernels.py::test_attention_backward_one_pass_equals_two_kernels).
constexpr int MF_THREADS = 512;
constexpr int MF_LOADER = 7;

// the loader wave: rows [0, Lp) of the K and V tiles of (li, h) HBM -> LDS with global_load_lds_dwordx4 (no registers: all
// 2 x Lp / 8 requests of 1 KB are in flight at once; a wave-instruction lands 8 tile rows in lane order, so the chunk swizzle
// is applied to the SOURCE address).  Rows beyond nvalid must read as zero: their lanes fetch from a 16-byte block of zeros.
__device__ __attribute__((aligned(16))) unsigned g_attn_zero16[4];
__device__ inline void loader_fill_kv(bf16raw* Ks, bf16raw* Vs, int Lp, int nvalid, const bf16raw* qkv, long ld, int D, int h,
                                      const RowLin& li, int lane) {
  const bf16raw* zeros = reinterpret_cast<const bf16raw*>(g_attn_zero16);
  const int rl = lane >> 3, pc = lane & 7;
#pragma unroll 4
  for (int g = 0; g < Lp / 8; ++g) {
    const int r = g * 8 + rl;
    const bf16raw* src = qkv + lin_row(li, r < nvalid ? r : 0) * ld + h * 64 + ((pc ^ sw_of(r)) << 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(r < nvalid ? src + D : zeros),
                                     (__attribute__((address_space(3))) void*)(Ks + g * 512), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(r < nvalid ? src + 2 * D : zeros),
                                     (__attribute__((address_space(3))) void*)(Vs + g * 512), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NT_>
__global__ __launch_bounds__(MF_THREADS, 1) void attn_bwd_fused_mfma_kernel(AttnP p, const bf16raw* __restrict__ qkv,
                                                                         const bf16raw* __restrict__ o, const bf16raw* __restrict__ dout,
                                                                         const float* __restrict__ lse, bf16raw* __restrict__ dqkv,
                                                                         bf16raw* __restrict__ dqkv_cls) {
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int D = p.H * 64, items = p.S * p.H;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nt = NT_ > 0 ? NT_ : (p.L + 31) >> 5, Lp = nt * 32;
  bf16raw* Ks = reinterpret_cast<bf16raw*>(sm_raw);
  bf16raw* Vs = Ks + Lp * 64;
  bf16raw* Qs = Vs + Lp * 64;
  bf16raw* Os = Qs + Lp * 64;                       // dO
  bf16raw* stg = Os + Lp * 64 + wave * MA_STAGE_ELEMS;
  float* Ls = reinterpret_cast<float*>(Os + Lp * 64 + 8 * MA_STAGE_ELEMS);   // lse * log2(e), +huge on padded rows
  float* Ds = Ls + Lp;
  const bool worker = wave < nt;
  const int q = wave * 32 + (lane & 31);            // the wave's query row (dq phase) = its key row (dk / dv phase)
  const float c2 = p.scale * LOG2E;
  const int ragged = (p.L & 31) ? nt - 1 : -1;
  FragOff fo = make_frag_off(lane);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifndef VTX_FUSED_ABLATE
#define VTX_FUSED_ABLATE 0                          // timing experiments only: 1 = no dq phase, 2 = no dk / dv phase, 3 = neither
#endif

  // The three roles run their own copies of the item loop (same item sequence, same number of barriers): the register
  // allocator sees the loader's addresses and the workers' accumulators in different code, not live together.
  const int stride = gridDim.x;
  if (blockIdx.x >= items) return;
  if (wave == MF_LOADER) {
    int item = blockIdx.x;
    {
      const int s = item / p.H;
      loader_fill_kv(Ks, Vs, Lp, p.L, qkv, p.ld_qkv, D, item - s * p.H, lin_in(p, s), lane);
    }
    while (true) {
      __syncthreads();                              // A
      __syncthreads();                              // B: nobody reads the K / V tiles any more
      const int next = item + stride;
      if (next >= items) break;
      const int sn = next / p.H;
      loader_fill_kv(Ks, Vs, Lp, p.L, qkv, p.ld_qkv, D, next - sn * p.H, lin_in(p, sn), lane);
      __syncthreads();                              // C
      item = next;
    }
    return;
  }
  if (!worker) {                                    // fewer than 7 tiles: waves without a tile only keep the barrier count
    for (int item = blockIdx.x;; item += stride) {
      __syncthreads();
      __syncthreads();
      if (item + stride >= items) break;
      __syncthreads();
    }
    return;
  }

  int item = blockIdx.x;
  int s = item / p.H, h = item - s * p.H;
  RowLin li = lin_in(p, s), lo = lin_out(p, s);
  bf16x8 qf[4], df[4], of[4];
  load_row_frags(qf, qkv, p.ld_qkv, h * 64, li, q, p.L, lane);
  load_row_frags(df, dout, p.ld_dout, h * 64, lo, q, p.L, lane);
  load_row_frags(of, o, p.ld_out, h * 64, lo, q, p.L, lane);
  float l2 = q < p.L ? lse[((long)s * p.H + h) * p.L + q] * LOG2E : 0.f;

  auto scores_dq = [&](int kt, f32x16& st, f32x16& dp) {
    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Ks, kt * 32, 0, fo), qf[0], zero, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Vs, kt * 32, 0, fo), df[0], zero, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Ks, kt * 32, ks, fo), qf[ks], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Vs, kt * 32, ks, fo), df[ks], dp, 0, 0, 0);
    }
  };

  while (true) {
    // the lane's fragment offsets are made opaque once per item: every LDS address of the unrolled phases is (offset +
    // constant), loop-invariant, and hoisted out of the item loop they would need a few hundred registers
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fo.rows[i]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fo.cols[i >> 1][i & 1]));
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dl = frag_dot2(df[ks], of[ks], dl);
    dl += __shfl_xor(dl, 32, 64);
    put_tile(Qs + wave * 32 * 64, qf, lane);
    put_tile(Os + wave * 32 * 64, df, lane);
    if (lane < 32) {
      Ls[q] = q < p.L ? l2 : 1e30f;
      Ds[q] = q < p.L ? dl : 0.f;
    }
    __syncthreads();                                // A: this item's four tiles, lse and delta are in LDS
    if (!(VTX_FUSED_ABLATE & 1)) {                  // ---- dq of query tile `wave` (attn_bwd_dq_mfma_kernel's tile body)
      f32x16 acc[2];
      zero16(acc[0]);
      zero16(acc[1]);
      f32x16 st, dp;
      scores_dq(0, st, dp);
#pragma unroll(NT_ > 0 ? NT_ : 1)
      for (int kt = 0; kt < nt; ++kt) {
        f32x16 stn = zero, dpn = zero;
        if (kt + 1 < nt) scores_dq(kt + 1, stn, dpn);
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(fmaf(st[r], c2, -l2)) * (dp[r] - dl);
        if (kt == ragged) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt * 32 + crow(r, lane) >= p.L) ds[r] = 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 db = pack8(ds + 8 * s2);
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2)
            acc[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Ks, kt * 32 + 16 * s2, n2, fo), db, acc[n2], 0, 0, 0);
        }
        st = stn;
        dp = dpn;
      }
      store_rows_T(stg, acc, p.scale, lane, [&](int r) -> bf16raw* {
        const int qq = wave * 32 + r;
        if (qq >= p.L) return nullptr;
        return (p.mode == VTX_ATTN_SPACE && qq == 0) ? dqkv_cls + (long)s * p.ld_dqkv + h * 64
                                                      : dqkv + lin_row(li, qq) * p.ld_dqkv + h * 64;
      });
    }
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag_rows_o(Ks, wave * 32, ks, fo);
      vf[ks] = frag_rows_o(Vs, wave * 32, ks, fo);
    }
    __syncthreads();                                // B: nobody reads the K / V tiles any more (the loader refills them)
    const int next = item + stride;
    const bool more = next < items;
    const int nx = more ? next : item;              // last item: the prefetch below re-reads the current rows (unused)
    const int sn = nx / p.H, hn = nx - sn * p.H;
    const RowLin lin = lin_in(p, sn), lon = lin_out(p, sn);
    f32x16 dk[2], dv[2];
    zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
    if (!(VTX_FUSED_ABLATE & 2)) {                  // ---- dk, dv of key tile `wave` (attn_bwd_dkv_mfma_kernel's tile body)
      auto scores_dkv = [&](int qt, f32x16& st, f32x16& dp) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Qs, qt * 32, 0, fo), kf[0], zero, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Os, qt * 32, 0, fo), vf[0], zero, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Qs, qt * 32, ks, fo), kf[ks], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Os, qt * 32, ks, fo), vf[ks], dp, 0, 0, 0);
        }
      };
#pragma unroll(NT_ > 0 ? NT_ : 1)
      for (int qt = 0; qt < nt; ++qt) {
        f32x16 st, dp;
        scores_dkv(qt, st, dp);
        // padded query rows: Ls = +huge -> P = 0; padded keys only feed dk/dv rows that are never stored
        float pr[16], ds[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int qrow = qt * 32 + 8 * g + 4 * (lane >> 5);
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
          const bf16x8 db = pack8(ds + 8 * s2);
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2) {
            dv[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Os, qt * 32 + 16 * s2, n2, fo), pb, dv[n2], 0, 0, 0);
            dk[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Qs, qt * 32 + 16 * s2, n2, fo), db, dk[n2], 0, 0, 0);
          }
        }
      }
    }
    // the next item's row fragments, in flight under the dk / dv stores
    load_row_frags(qf, qkv, p.ld_qkv, hn * 64, lin, q, p.L, lane);
    load_row_frags(df, dout, p.ld_dout, hn * 64, lon, q, p.L, lane);
    load_row_frags(of, o, p.ld_out, hn * 64, lon, q, p.L, lane);
    l2 = q < p.L ? lse[((long)sn * p.H + hn) * p.L + q] * LOG2E : 0.f;
    if (!(VTX_FUSED_ABLATE & 2)) {
      auto base_of = [&](int r) -> bf16raw* {
        const int kk = wave * 32 + r;
        if (kk >= p.L) return nullptr;
        return (p.mode == VTX_ATTN_SPACE && kk == 0) ? dqkv_cls + (long)s * p.ld_dqkv : dqkv + lin_row(li, kk) * p.ld_dqkv;
      };
      store_rows_T(stg, dk, p.scale, lane, [&](int r) -> bf16raw* { bf16raw* b = base_of(r); return b ? b + D + h * 64 : nullptr; });
      store_rows_T(stg, dv, 1.0f, lane, [&](int r) -> bf16raw* { bf16raw* b = base_of(r); return b ? b + 2 * D + h * 64 : nullptr; });
    }
    if (!more) break;
    __syncthreads();                                // C: nobody reads the Q / dO tiles, lse or delta any more
    item = next; s = sn; h = hn; li = lin; lo = lon;
  }
}

// ------------------------------------------------------------------- backward: ONE phase, operands streamed (round 4)
// The one-pass kernel above still computes every score twice (once per phase, in the layout that phase needs), keeps all
// four operand tiles in LDS (nothing of the next item can be brought in before the current one is done) and showed that its
// load time, dq phase and dk / dv phase ADD.  This kernel runs ONE phase per (sequence, head):
//   * worker w (waves 0 .. 6) owns key tile w for the whole item: K_w, V_w as row fragments in registers, dK^T / dV^T
//     accumulators in registers; it walks the query tiles 0 .. 6 exactly like the dk / dv kernel (S = Q K^T with lanes =
//     keys, P, dS, dV^T += dO^T P, dK^T += Q^T dS: same products, same order -- dk and dv are bit-identical to the kernels
//     above) -- one exp, one score product and one dP product per score instead of two;
//   * dq needs dS with lanes = queries and a sum over ALL key tiles.  Each worker drops its packed bf16 dS tile (2 KB, as
//     [key][query]) into a double-buffered LDS exchange area; after the step's barrier WAVE 7 reads the seven tiles back
//     through `ds_read_b64_tr_b16` as B operands and runs the whole dQ^T tile of the step, 28 MFMAs against the item's K^T
//     fragments (112 registers, taken from the workers' staging tiles at the top of the item), accumulating over the key
//     tiles in the matrix instruction like the dq kernel does (same order), and stores it as whole 128-byte rows.  (The
//     first version let every worker form its own fp32 partial dQ^T and summed the seven partials through LDS: 56 KB of
//     LDS writes per step at ~80 B/clk, two barriers per step, 16 registers of K^T fragments per worker.)
//   * Q / dO live in LDS as a RING of seven 32-row tiles: query tile i is dead after step i, and wave 7 puts the NEXT
//     item's tile i there right away by LDS-DMA (with its O tile into a staging tile and its lse values), a whole item
//     ahead of the first reader; two steps later it turns dO and O into delta = rowsum(dO * O) and lse into lse * log2(e)
//     (double-buffered by item parity).  The workers fetch their next K / V rows a step before the item ends.
// One barrier per step (`s_waitcnt lgkmcnt(0); s_barrier`: no vmcnt, prefetches and result stores stay in flight across it),
// one more at the top of an item.  The next query tile's score products are issued before the barrier.
// LDS: 2 x 28 KB ring + 2 x 15.75 KB dS exchange + 8 x 4 KB per-wave staging + 3.5 KB lse / delta + 12 KB O staging = 135 KB.
constexpr int MG_LP = 224;
constexpr int MG_TILE = 32 * 64;                // elements of one [32][64] bf16 tile
constexpr int MG_SCR_LD = 36;                   // a dS tile in the exchange area: [32 keys][36] bf16 (72-byte rows: conflict-free 8-byte stores)
constexpr int MG_SCR_BYTES = 32 * MG_SCR_LD * 2;             // 2304
constexpr int MG_XCH_BYTES = 7 * 2 * MG_SCR_BYTES;           // [worker][buffer][2304]: a worker's two buffers also hold its V tile (4 KB) at the top of an item
constexpr size_t MG_LDS_BYTES = (size_t)2 * 7 * MG_TILE * 2 + MG_XCH_BYTES + 8 * MA_STAGE_ELEMS * 2 + 2 * 2 * MG_LP * 4 + 4 * MG_TILE * 2 + 2 * 32 * 64 * 4;

__device__ inline void mg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// n / d for 0 <= n < 2^24 with inv = 1.0f / d (an integer division is ~40 vector instructions; the item -> (sequence, head) and
// sequence -> (clip, frame) splits are on the path between two items)
__device__ inline int mg_div(int n, int d, float inv) {
  int q = (int)((float)n * inv);
  const int r = n - q * d;
  q += r >= d ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}
__device__ inline RowLin mg_lin_in(const AttnP& p, int s, float invT) {
  RowLin r;
  if (p.mode == VTX_ATTN_CONTIG) { r.base = (long)s * p.L; r.stride = 1; r.row0 = r.base; return r; }
  const int b = mg_div(s, p.T, invT), t = s - b * p.T;
  r.row0 = (long)b * (1 + (long)p.P * p.T);
  r.stride = p.T;
  r.base = r.row0 + 1 + t - r.stride;
  return r;
}
__device__ inline RowLin mg_lin_out(const AttnP& p, int s, float invT) {
  RowLin r;
  if (p.mode == VTX_ATTN_CONTIG) { r.base = (long)s * p.L; r.stride = 1; r.row0 = r.base; return r; }
  const int b = mg_div(s, p.T, invT), t = s - b * p.T;
  r.row0 = (long)p.B * p.P * p.T + s;
  r.stride = p.T;
  r.base = (long)b * p.P * p.T + t - r.stride;
  return r;
}
// One [32][64] bf16 tile, rows row0 .. row0 + 31 of a sequence, fetched as whole 128-byte rows: lane -> row 8 g + lane / 8,
// 16-byte chunk lane % 8 (8 rows = 8 full lines per wave-instruction; the row-per-lane fragment pattern of load_row_frags
// touches 32 lines for 32 bytes each, and the vector memory pipeline works line by line: 8 such loads cost ~3000 cycles).
// Rows are linear in the token index except the sequence's row 0; rows beyond nvalid read row nvalid - 1.
__device__ inline void mg_load_tile_raw(u32x4 (&raw)[4], const bf16raw* base, long ld, int col0, const RowLin& rl, int row0, int nvalid,
                                        int lane) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int r = row0 + g * 8 + (lane >> 3);
    const int rc = r < nvalid ? r : nvalid - 1;
    raw[g] = *reinterpret_cast<const u32x4*>(base + lin_row(rl, rc) * ld + col0 + (lane & 7) * 8);
  }
}
__device__ inline void mg_raw_to_tile(bf16raw* tile, const u32x4 (&raw)[4], int lane) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int r = g * 8 + (lane >> 3);
    *reinterpret_cast<u32x4*>(tile + r * 64 + (((lane & 7) ^ sw_of(r)) << 3)) = raw[g];
  }
}
// store_rows_T with the destination rows in closed form: tile row r = 8 i + lane / 8 goes to p_lin + i * step8 (elements), tile
// row 0 to p_row0 (the sequence's row 0 is not on the line), rows >= nrows are not stored.  (The general form calls a row
// function per store: ~20 vector instructions of 64-bit arithmetic each.)
__device__ inline void mg_store_rows_lin(bf16raw* stg, const f32x16 (&acc)[2], float mul, int lane, bf16raw* p_lin, long step8,
                                         bf16raw* p_row0, int nrows) {
  const int row = lane & 31;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = nt * 32 + 8 * g + 4 * (lane >> 5);
      union { bf16x4 v; uint2 u; } w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w.v[j] = (__bf16)(acc[nt][4 * g + j] * mul);
      *reinterpret_cast<uint2*>(stg + sw_off(row, col)) = w.u;
    }
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = i * 8 + (lane >> 3), c = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(stg + r * 64 + ((c ^ sw_of(r)) << 3));
    bf16raw* dst = (i == 0 && r == 0) ? p_row0 : p_lin + i * step8;
    if (r < nrows) *reinterpret_cast<uint4*>(dst + c * 8) = v;
  }
  wave_lds_sync();
}

#ifndef VTX_STREAM_KV_AHEAD
#define VTX_STREAM_KV_AHEAD 2 // the workers request their next K rows this many steps (+ 1) and their next V rows this many steps before the item ends
#endif
#ifndef VTX_STREAM_ABLATE
#define VTX_STREAM_ABLATE 0   // timing experiments only (wrong results): 1 = the feeder requests nothing after item 0, 2 = no products /
#endif                        // softmax in the workers, 4 = no dq products / stores, 8 = no K / V refetch and no dk / dv stores

template <int NT_>
__global__ __launch_bounds__(MF_THREADS, 1) void attn_bwd_stream_mfma_kernel(AttnP p, const bf16raw* __restrict__ qkv,
                                                                          const bf16raw* __restrict__ o, const bf16raw* __restrict__ dout,
                                                                          const float* __restrict__ lse, bf16raw* __restrict__ dqkv,
                                                                          bf16raw* __restrict__ dqkv_cls, long long* __restrict__ trace) {
  static_assert(NT_ == 7, "seven 32-row tiles (193 .. 224 tokens): seven workers + wave 7");
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int D = p.H * 64, items = p.S * p.H;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16raw* Qs = reinterpret_cast<bf16raw*>(sm_raw);
  bf16raw* Os = Qs + NT_ * MG_TILE;                 // dO
  char* xch = reinterpret_cast<char*>(Os + NT_ * MG_TILE);                 // dS exchange: [worker][buffer][32 keys][36]
  bf16raw* stg0 = reinterpret_cast<bf16raw*>(xch + MG_XCH_BYTES);           // [8 waves][32][64] staging tiles
  bf16raw* stg = stg0 + wave * MA_STAGE_ELEMS;
  float* LD = reinterpret_cast<float*>(stg0 + 8 * MA_STAGE_ELEMS);          // [parity][lse2 | delta][224]
  const float c2 = p.scale * LOG2E;
  const int stride = gridDim.x;
  if (blockIdx.x >= items) return;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // VTX_STREAM_TRACE (variant builds, tools/attn_timeline.py): shader-clock stamps of wave 0 (a worker) and wave 7
  // of workgroup 0 for its first four items: trace[((role * 4 + item) * 8 + step) * 8 + point]
#ifdef VTX_STREAM_TRACE
  int tr_item = 0;
#define MG_STAMP(role_, step_, pt_)                                                                        \
  if (trace && blockIdx.x == 0 && tr_item < 4) {                                                           \
    long long t_;                                                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
    if (lane == 0) trace[(((role_) * 4 + tr_item) * 8 + (step_)) * 8 + (pt_)] = t_;                        \
  }
#else
#define MG_STAMP(role_, step_, pt_)
#endif
  const float invH = 1.0f / (float)p.H, invT = p.mode == VTX_ATTN_SPACE ? 1.0f / (float)p.T : 1.0f;
  // transpose-read address (bytes) of a dS tile in the exchange area: rows = keys 16 s2 + 4 half + .. (+ 8), columns = queries
  const unsigned scr_r = (unsigned)((4 * (lane >> 5) + ((lane & 15) >> 2)) * (MG_SCR_LD * 2) + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);

  // O staging tiles [4][32][64]: query tile t of local item m sits in tile (t - m) & 3 (requested at step t of item m - 1, turned into
  // delta three steps later while the requests of the three steps in between are in flight or pending)
  bf16raw* stO = reinterpret_cast<bf16raw*>(reinterpret_cast<char*>(LD) + 2 * 2 * MG_LP * 4);
  // delta = rowsum(dO * O) and lse * log2(e) of query tile t: dO from ring tile t, O from staging tile `ost`, into the arrays of
  // parity `par`; the lanes of row group `grp` (rows 8 grp .. + 7; -1: all) write.  Row-per-lane arithmetic of the other kernels
  // (same delta bit for bit).
  auto finish_tile = [&](int t, const bf16raw* ost, int par, int grp, const FragOff& f) {
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dl = frag_dot2(frag_rows_o(Os, t * 32, ks, f), frag_rows_o(ost, 0, ks, f), dl);
    dl += __shfl_xor(dl, 32, 64);
    if (lane < 32 && (grp < 0 || (lane >> 3) == grp)) {
      const int row = t * 32 + lane;
      float* lp = LD + par * 2 * MG_LP + row;
      const float raw = *lp;
      *lp = row < p.L ? raw * LOG2E : 1e30f;
      lp[MG_LP] = row < p.L ? dl : 0.f;
    }
  };
  // dq tiles leave wave 7 as fp32 [32 queries][64] in one of two LDS tiles (by the parity of the running step count; 16-byte chunk
  // index XOR-ed with the query's low bits: conflict-free 16-byte writes by query-per-lane and reads by row) and are scaled, rounded
  // and stored by workers 0 .. 3 two steps later, 8 whole rows each: wave 7 is the busiest wave of a step, the conversion (48 vector
  // instructions) costs it ~500 cycles
  char* dqf = reinterpret_cast<char*>(LD) + 2 * 2 * MG_LP * 4 + 4 * MG_TILE * 2;
  constexpr int dqf_bytes = 32 * 64 * 4;

  if (wave == MF_LOADER) {                          // ------------------------------------------------- wave 7: feeds, and owns dq
    // Everything this wave brings in travels HBM -> LDS by LDS-DMA (no registers, any number of requests in flight): the
    // Q / dO tile of query tile i of the NEXT item goes straight into ring slot i right after the barrier of step i, its O
    // tile into one of three staging tiles, its lse values into the other parity's lse array -- 13 wave-instructions per
    // step, a whole item ahead of their first reader.  Two steps later (`s_waitcnt vmcnt(13)`: requests retire in issue
    // order, the 13 of the step in between may still be in flight; the dq stores in between only make the wait stricter)
    // dO (ring) and O (staging) become delta = rowsum(dO * O) and lse becomes lse * log2(e).  The schedule rolls over the
    // item boundary: tile j = i - 2 (mod 7), so tiles 5 and 6 of an item are finished during steps 0 and 1 of that item
    // (their first readers are steps 5 and 6).  The requests are inline asm: hipcc waits `vmcnt(0)` before LDS reads while
    // a DMA it knows of is in flight, which would park this wave -- and with it every barrier -- on what it has just issued.
    typedef __attribute__((address_space(3))) char lds_char;
    FragOff ffo = make_frag_off(lane);
    // Requests without vector-ALU work: row r = 8 G + lane / 8 of a sequence sits r * rs bytes behind the sequence's (virtual) row
    // 0 on the token line, so a request is `buffer_load_dwordx4 voff, desc, soff offen lds` with a per-lane constant voff (the
    // lane's row of the first 8 + the source chunk that the destination swizzle asks for: it only depends on the parity of G), a
    // scalar soff = 8 G rs and a per-(item, tensor) descriptor whose range ends behind row L - 1 (rows beyond it read as zeros: their
    // lse is +huge).  Only the first 8 rows use per-lane pointers: the sequence's row 0 is not on the line.  (A 64-bit multiply-add
    // per request: 1600 cycles for the 13 requests of a step; pointer increments: 850.)
    struct Src { const bf16raw* p1; const bf16raw* p0; u32x4 desc; };
    const int rl8 = lane >> 3, pc = lane & 7;
    const int swz[2] = {(pc ^ sw_of(rl8)) << 3, (pc ^ sw_of(8 + rl8)) << 3};
    const long tok = p.mode == VTX_ATTN_SPACE ? p.T : 1;          // rows between consecutive tokens of a sequence
    auto make_src = [&](const bf16raw* base, long ld, int col0, const RowLin& rl) {
      Src x;
      x.p1 = base + (rl.base + (long)rl8 * rl.stride) * ld + col0;
      x.p0 = base + rl.row0 * ld + col0;
      const unsigned long a = reinterpret_cast<unsigned long>(base + rl.base * ld + col0);
      x.desc[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
      x.desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
      x.desc[2] = __builtin_amdgcn_readfirstlane((unsigned)((p.L - 1) * tok * ld * 2 + 128));
      x.desc[3] = 0x00020000u;
      return x;
    };
    auto dma_tile = [&](const Src& x, long ld, int t, bf16raw* dst) {
      const unsigned rs = (unsigned)(tok * ld * 2);               // bytes per token
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int G = 4 * t + g;
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_char*)(dst + g * 512));
        if (G == 0) {
          const bf16raw* src = (rl8 == 0 ? x.p0 : x.p1) + swz[0];
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
        } else {
          const unsigned voff = (unsigned)rl8 * rs + 2u * (unsigned)swz[g & 1];
          const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(8 * G) * rs);
          asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(x.desc), "s"(soff), "s"(m0v)
                       : "memory", "m0");
        }
      }
    };
    Src sq, sd, so;
    const float* slse = nullptr;                    // lse row of the item the requests are for
    auto set_item = [&](int s, int h, const RowLin& li, const RowLin& lo) {
      sq = make_src(qkv, p.ld_qkv, h * 64, li);
      sd = make_src(dout, p.ld_dout, h * 64, lo);
      so = make_src(o, p.ld_out, h * 64, lo);
      slse = lse + ((long)s * p.H + h) * p.L;
    };
    auto dma_item_tile = [&](int t, bf16raw* ost, float* lsd) {
      dma_tile(sq, p.ld_qkv, t, Qs + t * MG_TILE);
      dma_tile(sd, p.ld_dout, t, Os + t * MG_TILE);
      dma_tile(so, p.ld_out, t, ost);
      const int row = t * 32 + (lane & 31);
      const float* src = slse + (row < p.L ? row : p.L - 1);
      const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_char*)(lsd + t * 32));
      if (lane < 32) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
    };
    int item = blockIdx.x, par = 0;
    int s = mg_div(item, p.H, invH), h = item - s * p.H;
    RowLin li = mg_lin_in(p, s, invT), lo = mg_lin_out(p, s, invT);
    // item 0: all seven tiles at once, O tiles parked in the (still unused) exchange area
    {
      bf16raw* park = reinterpret_cast<bf16raw*>(xch);
      set_item(s, h, li, lo);
#pragma unroll
      for (int t = 0; t < NT_; ++t) dma_item_tile(t, park + t * MG_TILE, LD);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < NT_; ++t) finish_tile(t, park + t * MG_TILE, 0, -1, ffo);
    }
    int nn = 1;                                     // local index of the item the requests are for (mod 4)
    int q2 = 0;                                     // (7 n + i) & 1: dq staging tile of step i
    mg_barrier();                                   // P: item 0 is in the ring
    while (true) {
      const int next = item + stride;
      const bool more = next < items;
      const int sn = more ? mg_div(next, p.H, invH) : s, hn = more ? next - sn * p.H : h;
      const RowLin lin = mg_lin_in(p, sn, invT), lon = mg_lin_out(p, sn, invT);
      if (more) set_item(sn, hn, lin, lon);
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(ffo.cols[i >> 1][i & 1]));
      MG_STAMP(1, 7, 0)
      mg_barrier();                                 // T: the workers' K tiles are in their staging tiles
      // the item's K^T fragments: A operands of dQ^T += K_kt^T dS_kt^T for all seven key tiles
      bf16x8 ktf[NT_][2][2];
#pragma unroll
      for (int kt = 0; kt < NT_; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2) ktf[kt][s2][n2] = frag_cols_o(stg0 + kt * MA_STAGE_ELEMS, 16 * s2, n2, ffo);
      MG_STAMP(1, 7, 1)
#pragma unroll
      for (int i = 0; i < NT_; ++i) {
        MG_STAMP(1, i, 0)
        mg_barrier();                               // A(i): the dS tiles of step i are in buffer i & 1; nobody reads ring tile i any more
        MG_STAMP(1, i, 1)
        f32x16 acc[2];
        zero16(acc[0]); zero16(acc[1]);
        if (!(VTX_STREAM_ABLATE & 4)) {
#pragma unroll
          for (int kt = 0; kt < NT_; ++kt) {
            const char* t0 = xch + (kt * 2 + (i & 1)) * MG_SCR_BYTES + scr_r;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              union { bf16x8 v; s16x4 hh[2]; } u;
              u.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(t0 + s2 * 16 * MG_SCR_LD * 2));
              u.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (s16x4 __attribute__((address_space(3)))*)(t0 + s2 * 16 * MG_SCR_LD * 2 + 8 * MG_SCR_LD * 2));
#pragma unroll
              for (int n2 = 0; n2 < 2; ++n2) acc[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[kt][s2][n2], u.v, acc[n2], 0, 0, 0);
            }
          }
        }
        MG_STAMP(1, i, 2)
        // the step's requests go out between the products and their conversion: the last matrix instructions drain meanwhile
        if (!(VTX_STREAM_ABLATE & 1)) {
          // the requests of step i - 2 are complete behind this wait (those of step i - 1 may be in flight); the barrier of the next
          // step publishes that, and workers 0 .. 3 turn the tile into delta / lse * log2(e) behind it
          asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
          MG_STAMP(1, i, 3)
          if (more) dma_item_tile(i, stO + ((i - nn) & 3) * MG_TILE, LD + (par ^ 1) * 2 * MG_LP);
          MG_STAMP(1, i, 4)
        }
        if (!(VTX_STREAM_ABLATE & 4)) {
          // dQ^T tile (lane & 31 = query row, registers = 64 columns) -> fp32 tile
          char* dst = dqf + q2 * dqf_bytes + (lane & 31) * 256;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int ch = nt * 8 + 2 * g + (lane >> 5);
              float4 v;
              v.x = acc[nt][4 * g]; v.y = acc[nt][4 * g + 1]; v.z = acc[nt][4 * g + 2]; v.w = acc[nt][4 * g + 3];
              *reinterpret_cast<float4*>(dst + ((ch ^ (lane & 15)) << 4)) = v;
            }
        }
        q2 ^= 1;
        MG_STAMP(1, i, 5)
        MG_STAMP(1, i, 6)
      }
      nn = (nn + 1) & 3;
      if (!more) break;
      item = next; s = sn; h = hn; li = lin; lo = lon; par ^= 1;
#ifdef VTX_STREAM_TRACE
      ++tr_item;
#endif
    }
    mg_barrier();                                   // D: the last dq tile is in its staging tile
    return;
  }

  // ------------------------------------------------------------------------------------------------------------ workers
  FragOff fo = make_frag_off(lane);
  int item = blockIdx.x, par = 0;
  int s = mg_div(item, p.H, invH), h = item - s * p.H;
  RowLin li = mg_lin_in(p, s, invT);
  const int key = wave * 32 + (lane & 31);
  // K_w / V_w arrive as whole rows (kr / vr: row 8 g + lane / 8, chunk lane % 8) and become fragments through LDS
  u32x4 kr[4], vr[4];
  mg_load_tile_raw(kr, qkv, p.ld_qkv, D + h * 64, li, wave * 32, p.L, lane);
  mg_load_tile_raw(vr, qkv, p.ld_qkv, 2 * D + h * 64, li, wave * 32, p.L, lane);
  bf16x8 kf[4], vf[4];
  const bool ragged_wave = wave == NT_ - 1 && (p.L & 31) != 0;
  // this worker's part of the exchange area: two dS buffers ([key][query], 8-byte stores of 4 queries), also its V tile at the top
  char* myx = xch + wave * 2 * MG_SCR_BYTES;
  const unsigned scr_w = (unsigned)((lane & 31) * (MG_SCR_LD * 2) + 8 * (lane >> 5));
  auto scores = [&](int qt, f32x16& st, f32x16& dp) {
    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Qs, qt * 32, 0, fo), kf[0], zero, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Os, qt * 32, 0, fo), vf[0], zero, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Qs, qt * 32, ks, fo), kf[ks], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Os, qt * 32, ks, fo), vf[ks], dp, 0, 0, 0);
    }
  };
  // dq rows: staging tile row r = 8 g + lane / 8 of query tile t -> dq_lin + (4 t + g) * dq_step8, row 0 of the sequence -> dq_row0
  bf16raw *dq_lin = nullptr, *dq_row0 = nullptr, *dq_lin_prev = nullptr, *dq_row0_prev = nullptr;
  long dq_step8 = 0, dq_step8_prev = 0;
  bool have_prev = false;
  int nloc = 0;                                     // local item index mod 4
  int rp = 0;                                       // (7 n + k) & 1 at step k: staging tile of the dq tile written two steps ago
  auto store_dq = [&](int t, bool prev) {           // workers 0 .. 3: 8 rows each of query tile t (of the previous item: prev)
    if (!(VTX_STREAM_ABLATE & 4)) {                 // (no branch around the LDS read: it is scheduled into the step's other work)
      const int r = (wave & 3) * 8 + (lane >> 3), c = lane & 7;
      const char* src = dqf + rp * dqf_bytes + r * 256;
      const float4 a = *reinterpret_cast<const float4*>(src + (((2 * c) ^ (r & 15)) << 4));
      const float4 b = *reinterpret_cast<const float4*>(src + (((2 * c + 1) ^ (r & 15)) << 4));
      union { bf16x8 v8; uint4 u; } w;
      w.v8[0] = (__bf16)(a.x * p.scale); w.v8[1] = (__bf16)(a.y * p.scale); w.v8[2] = (__bf16)(a.z * p.scale); w.v8[3] = (__bf16)(a.w * p.scale);
      w.v8[4] = (__bf16)(b.x * p.scale); w.v8[5] = (__bf16)(b.y * p.scale); w.v8[6] = (__bf16)(b.z * p.scale); w.v8[7] = (__bf16)(b.w * p.scale);
      const uint4 v = w.u;
      bf16raw* lin = prev ? dq_lin_prev : dq_lin;
      bf16raw* dst = lin + (4 * t + (wave & 3)) * (prev ? dq_step8_prev : dq_step8);
      if (t == 0 && r == 0) dst = prev ? dq_row0_prev : dq_row0;
      if (wave < 4 && t * 32 + r < p.L) *reinterpret_cast<uint4*>(dst + c * 8) = v;
    }
  };
  // (the first item's K / V rows are complete before the loop is entered: the loop header then needs no wait for them on
  // either path -- a wait there would, on the back edge, drain the dk / dv stores of the item before)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(kr[ks]), "+v"(vr[ks]));
  mg_barrier();                                     // P
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fo.rows[i]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fo.cols[i >> 1][i & 1]));
    MG_STAMP(0, 7, 0)
    dq_lin = dqkv + (li.base + (long)(lane >> 3) * li.stride) * p.ld_dqkv + h * 64;
    dq_row0 = (p.mode == VTX_ATTN_SPACE ? dqkv_cls + (long)s * p.ld_dqkv : dqkv + li.row0 * p.ld_dqkv) + h * 64;
    dq_step8 = 8 * li.stride * p.ld_dqkv;
    // K_w into the staging tile (wave 7 takes its K^T fragments from there behind barrier T); V_w into this worker's exchange
    // buffers -- behind T as well: until then wave 7 may still be reading the dS tiles of the previous item's last step
    mg_raw_to_tile(stg, kr, lane);
    mg_barrier();                                   // T
    mg_raw_to_tile(reinterpret_cast<bf16raw*>(myx), vr, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = frag_rows_o(stg, 0, ks, fo);
    wave_lds_sync();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vf[ks] = frag_rows_o(reinterpret_cast<bf16raw*>(myx), 0, ks, fo);
    const float* Lc = LD + par * 2 * MG_LP;
    const float* Dc = Lc + MG_LP;
    const int next = item + stride;
    const bool more = next < items;
    const int nx = more ? next : item;              // last item: the prefetch re-reads the current rows (unused)
    f32x16 dk[2], dv[2];
    zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
    f32x16 st = zero, dp = zero;
    if (!(VTX_STREAM_ABLATE & 2)) scores(0, st, dp);
    wave_lds_sync();                                // V fragments are in registers before step 0 writes dS over the V tile
    MG_STAMP(0, 7, 1)
#pragma unroll
    for (int i = 0; i < NT_; ++i) {
      MG_STAMP(0, i, 0)
      if (i >= 2) store_dq(i - 2, false);
      else if (have_prev) store_dq(i + 5, true);
      rp ^= 1;
      f32x16 stn = zero, dpn = zero;
      if (!(VTX_STREAM_ABLATE & 2)) {
        float pr[16], ds[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int qrow = i * 32 + 8 * g + 4 * (lane >> 5);
          const float4 l4 = *reinterpret_cast<const float4*>(Lc + qrow);
          const float4 d4 = *reinterpret_cast<const float4*>(Dc + qrow);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            const float e = __builtin_amdgcn_exp2f(fmaf(st[r], c2, -lv[j]));
            pr[r] = e;
            ds[r] = e * (dp[r] - dvv[j]);            // the softmax scale is applied once, at the stores
          }
        }
        bf16x8 pb[2], db[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) db[s2] = pack8(ds + 8 * s2);
        {
          // dS^T for wave 7: padded keys (only the last key tile has any) must not reach dq -- their dk / dv columns are never
          // stored, but a dq row sums over all keys
          bf16x8 dm[2] = {db[0], db[1]};
          if (ragged_wave && key >= p.L) { dm[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; dm[1] = dm[0]; }
          union { bf16x8 v; uint2 u[2]; } x0, x1;
          x0.v = dm[0]; x1.v = dm[1];
          char* dst = myx + (i & 1) * MG_SCR_BYTES + scr_w;
          *reinterpret_cast<uint2*>(dst) = x0.u[0];          // queries  0 ..  3 (+ 4 half)
          *reinterpret_cast<uint2*>(dst + 16) = x0.u[1];     //          8 .. 11
          *reinterpret_cast<uint2*>(dst + 32) = x1.u[0];     //         16 .. 19
          *reinterpret_cast<uint2*>(dst + 48) = x1.u[1];     //         24 .. 27
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) pb[s2] = pack8(pr + 8 * s2);
        MG_STAMP(0, i, 1)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2) {
            dv[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Os, i * 32 + 16 * s2, n2, fo), pb[s2], dv[n2], 0, 0, 0);
            dk[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Qs, i * 32 + 16 * s2, n2, fo), db[s2], dk[n2], 0, 0, 0);
          }
        MG_STAMP(0, i, 2)
        if (i + 1 < NT_) scores(i + 1, stn, dpn);   // the next query tile's products are on the matrix pipe across the barrier
      }
      if ((i == NT_ - 1 - VTX_STREAM_KV_AHEAD || i == NT_ - VTX_STREAM_KV_AHEAD) && !(VTX_STREAM_ABLATE & 8)) {
        // the next item's K_w rows are requested two steps, its V_w rows one step before the item ends (no gain from more; 8 requests per worker in
        // one step keep the vector memory pipeline of the CU busy for 1500 cycles with every worker waiting to issue)
        int nxo = nx;                               // opaque: the addresses are built here, not at the top of the item (spills)
        asm volatile("" : "+s"(nxo));
        const int sn = mg_div(nxo, p.H, invH), hn = nxo - sn * p.H;
        const RowLin lin = mg_lin_in(p, sn, invT);
        if (i == NT_ - 1 - VTX_STREAM_KV_AHEAD) mg_load_tile_raw(kr, qkv, p.ld_qkv, D + hn * 64, lin, wave * 32, p.L, lane);
        else mg_load_tile_raw(vr, qkv, p.ld_qkv, 2 * D + hn * 64, lin, wave * 32, p.L, lane);
      }
      MG_STAMP(0, i, 3)
      mg_barrier();                                 // A(i)
      MG_STAMP(0, i, 4)
      // delta / lse of the tile whose requests wave 7 saw complete before this barrier: query tile i - 3 of the next item (i >= 3) or
      // tile i + 4 of this one (requested during the item before); workers 0 .. 3 write eight rows each
      // (every worker runs the ~45 instructions -- no branch, so they are scheduled into the next step's work instead of standing
      // alone behind the barrier as an 850-cycle latency chain; no lane of workers 4 .. 6 belongs to a row group 4 .. 6)
      if (!(VTX_STREAM_ABLATE & 1)) {
        if (i >= 3) {
          if (more) finish_tile(i - 3, stO + ((i - 3 - (nloc + 1)) & 3) * MG_TILE, par ^ 1, wave, fo);
        } else if (have_prev) {
          finish_tile(i + 4, stO + ((i + 4 - nloc) & 3) * MG_TILE, par, wave, fo);
        }
      }
      st = stn;
      dp = dpn;
    }
    // The next item's K / V rows are waited for HERE (requested a step ago), in front of the dk / dv stores: requests retire in
    // order, so a wait at the top of the next item would also drain these eight stores.
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(kr[ks]), "+v"(vr[ks]));
    MG_STAMP(0, 7, 2)
    if (!(VTX_STREAM_ABLATE & 8)) {
      // tile row r = key wave * 32 + r: linear in r except the sequence's row 0 (wave 0, r = 0)
      const long krow = wave * 32 + (lane >> 3);
      bf16raw* p_lin = dqkv + (li.base + krow * li.stride) * p.ld_dqkv + D + h * 64;
      bf16raw* p_row0 = wave == 0 ? (p.mode == VTX_ATTN_SPACE ? dqkv_cls + (long)s * p.ld_dqkv : dqkv + li.row0 * p.ld_dqkv) + D + h * 64
                                  : p_lin;
      const long step8 = 8 * li.stride * p.ld_dqkv;
      const int nrows = p.L - wave * 32;
      // (the staging tile still holds K_w, but wave 7 took its fragments right behind barrier T, seven barriers ago)
      mg_store_rows_lin(stg, dk, p.scale, lane, p_lin, step8, p_row0, nrows);
      mg_store_rows_lin(stg, dv, 1.0f, lane, p_lin + D, step8, p_row0 + D, nrows);
    }
    MG_STAMP(0, 7, 3)
    if (!more) break;
    dq_lin_prev = dq_lin; dq_row0_prev = dq_row0; dq_step8_prev = dq_step8; have_prev = true;
    nloc = (nloc + 1) & 3;
    item = next; s = mg_div(item, p.H, invH); h = item - s * p.H; li = mg_lin_in(p, s, invT); par ^= 1;
#ifdef VTX_STREAM_TRACE
    ++tr_item;
#endif
  }
  store_dq(5, false);                               // the last item's last two dq tiles
  rp ^= 1;
  mg_barrier();                                     // D
  store_dq(6, false);
}

// ------------------------------------------------------------------- forward with streamed K / V (round 4, option attn_fwd_stream)
// The forward kernel at the top of this file is a workgroup per (sequence, head): fetch K and V (global -> registers -> LDS),
// barrier, seven query tiles on four waves (2 + 2 + 2 + 1), stores -- load, compute and store phases of a workgroup follow each
// other and only the second workgroup of the CU covers them.  Here one workgroup per CU is persistent over the items like the
// backward kernel above: waves 0 .. 6 own one query tile each (the tile body of the kernel above, unchanged: bit-identical
// results), wave 7 brings the NEXT item's K and V tiles into the other half of a double buffer by LDS-DMA (56 requests, range-
// checked buffer loads: padded key rows read as zeros) while the workers run; one barrier per item.
constexpr size_t MGF_LDS_BYTES = (size_t)2 * 2 * 7 * MG_TILE * 2 + 8 * MA_STAGE_ELEMS * 2;

template <int NT_>
__global__ __launch_bounds__(MF_THREADS, 1) void attn_fwd_stream_mfma_kernel(AttnP p, const bf16raw* __restrict__ qkv,
                                                                          bf16raw* __restrict__ out, float* __restrict__ lse) {
  static_assert(NT_ == 7, "seven 32-row tiles (193 .. 224 tokens)");
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int D = p.H * 64, items = p.S * p.H;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16raw* KV = reinterpret_cast<bf16raw*>(sm_raw);               // [buffer][K | V][7 tiles][32][64]
  bf16raw* stg = KV + 2 * 2 * NT_ * MG_TILE + wave * MA_STAGE_ELEMS;
  const int stride = gridDim.x;
  if (blockIdx.x >= items) return;
  const float invH = 1.0f / (float)p.H, invT = p.mode == VTX_ATTN_SPACE ? 1.0f / (float)p.T : 1.0f;

  if (wave == MF_LOADER) {                          // ------------------------------------------------------------- wave 7
    typedef __attribute__((address_space(3))) char lds_char;
    const int rl8 = lane >> 3, pc = lane & 7;
    const int swz[2] = {(pc ^ sw_of(rl8)) << 3, (pc ^ sw_of(8 + rl8)) << 3};
    const long tok = p.mode == VTX_ATTN_SPACE ? p.T : 1;
    const unsigned rs = (unsigned)(tok * p.ld_qkv * 2);           // bytes per token
    const unsigned voff[2] = {(unsigned)rl8 * rs + 2u * (unsigned)swz[0], (unsigned)rl8 * rs + 2u * (unsigned)swz[1]};
    auto request = [&](int item, int buf) {
      const int s = mg_div(item, p.H, invH), h = item - s * p.H;
      const RowLin li = mg_lin_in(p, s, invT);
#pragma unroll
      for (int kv = 0; kv < 2; ++kv) {
        const int col0 = (1 + kv) * D + h * 64;
        const bf16raw* p1 = qkv + (li.base + (long)rl8 * li.stride) * p.ld_qkv + col0;
        const bf16raw* p0 = qkv + li.row0 * p.ld_qkv + col0;
        const unsigned long a = reinterpret_cast<unsigned long>(qkv + li.base * p.ld_qkv + col0);
        u32x4 desc;
        desc[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        desc[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
        desc[2] = __builtin_amdgcn_readfirstlane((unsigned)((p.L - 1) * tok * p.ld_qkv * 2 + 128));
        desc[3] = 0x00020000u;
        bf16raw* dst = KV + (buf * 2 + kv) * NT_ * MG_TILE;
#pragma unroll
        for (int G = 0; G < 4 * NT_; ++G) {
          const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_char*)(dst + G * 512));
          if (G == 0) {
            const bf16raw* src = (rl8 == 0 ? p0 : p1) + swz[0];
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
          } else {
            const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(8 * G) * rs);
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff[G & 1]), "s"(desc), "s"(soff), "s"(m0v)
                         : "memory", "m0");
          }
        }
      }
    };
    int item = blockIdx.x, buf = 0;
    request(item, 0);
    while (true) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      mg_barrier();                                 // E: K / V of `item` are in buffer `buf`; the workers are done with the other buffer
      const int next = item + stride;
      if (next >= items) break;
      request(next, buf ^ 1);
      item = next;
      buf ^= 1;
    }
    mg_barrier();                                   // the workers' last item
    return;
  }

  // ---------------------------------------------------------------------------------------------------------------- workers
  const float c2 = p.scale * LOG2E;
  const int ragged = (p.L & 31) ? NT_ - 1 : -1;    // the key tile that holds padded keys
  const FragOff fo0 = make_frag_off(lane);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int item = blockIdx.x, buf = 0;
  int s = mg_div(item, p.H, invH), h = item - s * p.H;
  RowLin li = mg_lin_in(p, s, invT);
  const int q = wave * 32 + (lane & 31);
  bf16x8 qf[4];
  load_row_frags(qf, qkv, p.ld_qkv, h * 64, li, q, p.L, lane);
  while (true) {
    mg_barrier();                                   // E
    FragOff fo = fo0;
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fo.rows[i]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fo.cols[i >> 1][i & 1]));
    const bf16raw* Ks = KV + buf * 2 * NT_ * MG_TILE;
    const bf16raw* Vs = Ks + NT_ * MG_TILE;
    const int next = item + stride;
    const bool more = next < items;
    const int nx = more ? next : item;
    const int sn = mg_div(nx, p.H, invH), hn = nx - sn * p.H;
    const RowLin lin = mg_lin_in(p, sn, invT);
    bf16x8 qn[4];                                   // the next item's query rows, in flight during this one
    load_row_frags(qn, qkv, p.ld_qkv, hn * 64, lin, q, p.L, lane);
    f32x16 acc[2];
    zero16(acc[0]);
    zero16(acc[1]);
    float m = -1e30f, l = 0.f;                      // running max of the RAW scores (scale > 0)
#pragma unroll
    for (int kb = 0; kb < NT_; kb += MA_KB) {
      f32x16 st[MA_KB];
#pragma unroll
      for (int t = 0; t < MA_KB; ++t) {
        if (kb + t < NT_) {
          st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Ks, (kb + t) * 32, 0, fo), qf[0], zero, 0, 0, 0);
#pragma unroll
          for (int ks = 1; ks < 4; ++ks)
            st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_o(Ks, (kb + t) * 32, ks, fo), qf[ks], st[t], 0, 0, 0);
        }
      }
      float bm = -1e30f;
#pragma unroll
      for (int t = 0; t < MA_KB; ++t)
        if (kb + t < NT_) {
          if (kb + t == ragged) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if ((kb + t) * 32 + crow(r, lane) >= p.L) st[t][r] = -1e30f;
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
      for (int t = 0; t < MA_KB; ++t)
        if (kb + t < NT_) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(st[t][r], c2, -mc)); st[t][r] = e; bl += e; }
        }
      bl += __shfl_xor(bl, 32, 64);
      l = l * alpha + bl;
      if (kb > 0) {
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[n2][r] *= alpha;
      }
#pragma unroll
      for (int t = 0; t < MA_KB; ++t)
        if (kb + t < NT_) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            float pf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = st[t][8 * s2 + j];
            const bf16x8 pb = pack8(pf);
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
              acc[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_o(Vs, (kb + t) * 32 + 16 * s2, n2, fo), pb, acc[n2], 0, 0, 0);
          }
        }
    }
    {
      const RowLin lo = mg_lin_out(p, s, invT);
      const long orow = wave * 32 + (lane >> 3);
      bf16raw* p_lin = out + (lo.base + orow * lo.stride) * p.ld_out + h * 64;
      bf16raw* p_row0 = wave == 0 ? out + lo.row0 * p.ld_out + h * 64 : p_lin;
      mg_store_rows_lin(stg, acc, 1.0f / l, lane, p_lin, 8 * lo.stride * p.ld_out, p_row0, p.L - wave * 32);
      if (q < p.L && lane < 32) lse[((long)s * p.H + h) * p.L + q] = (m * c2) * LN2 + __logf(l);
    }
    if (!more) break;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    item = next; s = sn; h = hn; li = lin; buf ^= 1;
  }
  mg_barrier();
}

// =====================================================================================
// Short sequences (L <= 32, contiguous rows): temporal attention of the divided block
// (L = T = 8) and ViViT's temporal encoder (L = 9).  G = 32/L sequences are packed into one
// 32-row MFMA tile (block-diagonal mask); ONE WAVE handles a (tile, head) pair end to end, with
// operands loaded straight from HBM into MFMA fragments (the row-wise fragment pattern covers
// the [32][64] tile exactly once) and a wave-private 4 KB LDS tile per transposed operand.
// HBM-bound: forward moves 4 and backward 8 row-blocks of [rows][64] per head.
// =====================================================================================
constexpr int SM_WAVE_LDS_FWD = 32 * 64 * 2;                   // V tile
constexpr int SM_WAVE_LDS_BWD = 2 * 32 * 64 * 2 + 2 * 32 * 4;   // tile A (K, then Q), tile B (dO) + lse + delta

__device__ inline void load_frags(bf16x8 (&f)[4], const bf16raw* base, long ld, int col0, long row, bool valid, int lane) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    union { bf16x8 v; uint4 u; } x;
    x.u = make_uint4(0, 0, 0, 0);
    if (valid) x.u = *reinterpret_cast<const uint4*>(base + row * ld + col0 + ks * 16 + 8 * (lane >> 5));
    f[ks] = x.v;
  }
}

// HW (heads in the workgroup): wave w of a workgroup works on head w of ONE row tile, so that the workgroup as a whole
// reads the tile's rows contiguously (32 x 3 H x 128 B) at one moment -- instead of one 128-B line out of every
// 4.6 KB row per workgroup, with the other heads' lines of the same rows requested by other CUs at other times.
template <bool HW>
__global__ __launch_bounds__(HW ? 1024 : MA_THREADS) void attn_fwd_small_kernel(AttnP p, int ntiles, const bf16raw* __restrict__ qkv,
                                                                    bf16raw* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // HW: the head groups of a row tile are CONSECUTIVE workgroups (blockIdx.x = head group): they run at the same time and read
  // neighbouring 128-byte segments of the same rows -- with the tile index fastest the groups of one tile were a whole grid row
  // apart (temporal backward, 4 heads per workgroup: 356 -> 332 us)
  // (head group, tile) folded into blockIdx.x, head group fastest -- grid.y stops at 65535 tiles, grid.x at 2^31
  const int hw_w = blockDim.x >> 6, hw_ng = HW ? (p.H + hw_w - 1) / hw_w : 1;
  const int tile = HW ? (int)(blockIdx.x / hw_ng) : blockIdx.x * 4 + wave, h = HW ? wave + (int)(blockIdx.x % hw_ng) * hw_w : blockIdx.y, D = p.H * 64;
  if (tile >= ntiles || h >= p.H) return;
  bf16raw* Vs = reinterpret_cast<bf16raw*>(sm_raw + wave * SM_WAVE_LDS_FWD);
  const int L = p.L, G = 32 / L, used = G * L;
  const long row0 = (long)tile * used, total = (long)p.S * L;
  const int i = lane & 31;
  const bool valid = i < used && row0 + i < total;
  bf16x8 qf[4], kf[4], vf[4];
  load_frags(qf, qkv, p.ld_qkv, h * 64, row0 + i, valid, lane);
  load_frags(kf, qkv, p.ld_qkv, D + h * 64, row0 + i, valid, lane);
  load_frags(vf, qkv, p.ld_qkv, 2 * D + h * 64, row0 + i, valid, lane);
  put_tile(Vs, vf, lane);
  f32x16 st;
  zero16(st);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], st, 0, 0, 0);
  const float c2 = p.scale * LOG2E;
  const int qseq = i / L;
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = crow(r, lane);
    const bool ok = valid && key < used && key / L == qseq && row0 + key < total;
    st[r] = ok ? st[r] * c2 : -INFINITY;
    m = fmaxf(m, st[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (!valid) m = 0.f;
  float l = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - m); l += st[r]; }
  l += __shfl_xor(l, 32, 64);
  wave_lds_sync();
  f32x16 acc[2];
  zero16(acc[0]);
  zero16(acc[1]);
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    float pf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[j] = st[8 * s2 + j];
    const bf16x8 pb = pack8(pf);
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
      acc[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Vs, 16 * s2, n2 * 32, lane), pb, acc[n2], 0, 0, 0);
  }
  // whole 128-B rows through the (now idle) V tile: the direct store writes 8-byte pieces of 32 different rows per
  // instruction, and the vector memory pipeline handles those one line at a time
  store_rows_T(Vs, acc, valid ? 1.0f / l : 0.f, lane, [&](int r) -> bf16raw* {
    return (r < used && row0 + r < total) ? out + (row0 + r) * p.ld_out + h * 64 : nullptr;
  });
  if (valid && lane < 32) {
    const long sq = (row0 + i) / L;
    lse[(sq * p.H + h) * L + (row0 + i - sq * L)] = m * LN2 + __logf(l);
  }
}

template <bool HW>
__global__ __launch_bounds__(HW ? 1024 : MA_THREADS, HW ? 1 : 4) void attn_bwd_small_kernel(AttnP p, int ntiles, const bf16raw* __restrict__ qkv,
                                                                    const bf16raw* __restrict__ o, const bf16raw* __restrict__ dout,
                                                                    const float* __restrict__ lse, bf16raw* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // HW: the head groups of a row tile are CONSECUTIVE workgroups (blockIdx.x = head group): they run at the same time and read
  // neighbouring 128-byte segments of the same rows -- with the tile index fastest the groups of one tile were a whole grid row
  // apart (temporal backward, 4 heads per workgroup: 356 -> 332 us)
  // (head group, tile) folded into blockIdx.x, head group fastest -- grid.y stops at 65535 tiles, grid.x at 2^31
  const int hw_w = blockDim.x >> 6, hw_ng = HW ? (p.H + hw_w - 1) / hw_w : 1;
  const int tile = HW ? (int)(blockIdx.x / hw_ng) : blockIdx.x * 4 + wave, h = HW ? wave + (int)(blockIdx.x % hw_ng) * hw_w : blockIdx.y, D = p.H * 64;
  if (tile >= ntiles || h >= p.H) return;
  // two wave-private tiles: A holds K for phase 1 (and stages dQ), then Q for phase 2 (and stages dK); B holds dO
  // (and stages dV) -- 8.25 KB per wave instead of 12.25 KB: LDS is what bounds the waves per CU of this kernel
  char* wl = sm_raw + wave * SM_WAVE_LDS_BWD;
  bf16raw* Ks = reinterpret_cast<bf16raw*>(wl);
  bf16raw* Qs = Ks;
  bf16raw* Os = Ks + 32 * 64;
  float* Ls = reinterpret_cast<float*>(Os + 32 * 64);
  float* Ds = Ls + 32;
  const int L = p.L, G = 32 / L, used = G * L;
  const long row0 = (long)tile * used, total = (long)p.S * L;
  const int i = lane & 31;
  const bool valid = i < used && row0 + i < total;
  bf16x8 qf[4], kf[4], vf[4], df[4];
  load_frags(qf, qkv, p.ld_qkv, h * 64, row0 + i, valid, lane);
  load_frags(kf, qkv, p.ld_qkv, D + h * 64, row0 + i, valid, lane);
  load_frags(vf, qkv, p.ld_qkv, 2 * D + h * 64, row0 + i, valid, lane);
  load_frags(df, dout, p.ld_dout, h * 64, row0 + i, valid, lane);
  float dl = 0.f;
  {
    bf16x8 of[4];
    load_frags(of, o, p.ld_out, h * 64, row0 + i, valid, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dl += frag_dot(df[ks], of[ks]);
  }
  dl += __shfl_xor(dl, 32, 64);
  float l2 = 1e30f;
  if (valid) {
    const long sq = (row0 + i) / L;
    l2 = lse[(sq * p.H + h) * L + (row0 + i - sq * L)] * LOG2E;
  }
  put_tile(Ks, kf, lane);
  put_tile(Os, df, lane);
  if (lane < 32) { Ls[i] = l2; Ds[i] = dl; }
  const float c2 = p.scale * LOG2E;
  const int myseq = i / L;
  // ---- phase 1: lanes = queries.  dQ^T = K^T dS^T
  {
    f32x16 st, dp;
    zero16(st);
    zero16(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], df[ks], dp, 0, 0, 0);
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = crow(r, lane);
      const bool ok = valid && key < used && key / L == myseq && row0 + key < total;
      const float pr = ok ? __builtin_amdgcn_exp2f(st[r] * c2 - l2) : 0.f;
      ds[r] = pr * (dp[r] - dl) * p.scale;
    }
    wave_lds_sync();
    f32x16 acc[2];
    zero16(acc[0]);
    zero16(acc[1]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 db = pack8(ds + 8 * s2);
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
        acc[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Ks, 16 * s2, n2 * 32, lane), db, acc[n2], 0, 0, 0);
    }
    // K tile: its last reader was the MFMA chain above
    store_rows_T(Ks, acc, 1.0f, lane, [&](int r) -> bf16raw* {
      return (r < used && row0 + r < total) ? dqkv + (row0 + r) * p.ld_dqkv + h * 64 : nullptr;
    });
  }
  // ---- phase 2: lanes = keys.  dV^T = dO^T P, dK^T = Q^T dS
  {
    put_tile(Qs, qf, lane);                        // tile A is free again: store_rows_T above ends with a wave-level sync
    wave_lds_sync();
    f32x16 st, dp;
    zero16(st);
    zero16(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[ks], st, 0, 0, 0);
      // dO row fragments come back from tile B (it holds dO until dv is staged there): kept in registers across the two phases they
      // put the kernel at 138 registers = three waves per SIMD instead of four (124 now: 415 -> 393 us at 96 clips)
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Os, 0, ks, lane), vf[ks], dp, 0, 0, 0);
    }
    float pr[16], ds[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int qrow = 8 * g + 4 * (lane >> 5);
      const float4 l4 = *reinterpret_cast<const float4*>(Ls + qrow);
      const float4 d4 = *reinterpret_cast<const float4*>(Ds + qrow);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * g + j;
        const int q = qrow + j;
        const bool ok = valid && q < used && q / L == myseq;
        const float e = ok ? __builtin_amdgcn_exp2f(st[r] * c2 - lv[j]) : 0.f;
        pr[r] = e;
        ds[r] = e * (dp[r] - dvv[j]) * p.scale;
      }
    }
    f32x16 dk[2], dv[2];
    zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pb = pack8(pr + 8 * s2);
      const bf16x8 db = pack8(ds + 8 * s2);
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        dv[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Os, 16 * s2, n2 * 32, lane), pb, dv[n2], 0, 0, 0);
        dk[n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(Qs, 16 * s2, n2 * 32, lane), db, dk[n2], 0, 0, 0);
      }
    }
    auto dst = [&](int r, int col) -> bf16raw* {
      return (r < used && row0 + r < total) ? dqkv + (row0 + r) * p.ld_dqkv + col + h * 64 : nullptr;
    };
    store_rows_T(Qs, dk, 1.0f, lane, [&](int r) -> bf16raw* { return dst(r, D); });
    store_rows_T(Os, dv, 1.0f, lane, [&](int r) -> bf16raw* { return dst(r, 2 * D); });
  }
}

// host-side launchers used by attn.hip's entry points --------------------------------------
bool attn_mfma_eligible(int dtype, int L, int hd) {
  return dtype == VTX_BF16 && hd == 64 && L > 32 && L <= 32 * MA_MAXT;
}
bool attn_small_eligible(int dtype, int mode, int L, int hd) {
  return dtype == VTX_BF16 && hd == 64 && mode == VTX_ATTN_CONTIG && L >= 1 && L <= 32;
}
// heads per workgroup in the HW variants: option value n > 0, capped at 16 waves (1024 threads) and at H
static int hw_waves(int n, int H) { n = n < H ? n : H; return n < 16 ? n : 16; }

int attn_fwd_small_launch(const AttnP& p, const void* qkv, void* out, float* lse, hipStream_t st) {
  const int G = 32 / p.L;
  const int ntiles = (p.S + G - 1) / G;
  if (options().attn_hw_fwd > 0) {
    const int w = hw_waves(options().attn_hw_fwd, p.H);
    hipLaunchKernelGGL(attn_fwd_small_kernel<true>, dim3(((p.H + w - 1) / w) * ntiles), dim3(64 * w), w * SM_WAVE_LDS_FWD, st, p, ntiles,
                       (const bf16raw*)qkv, (bf16raw*)out, lse);
  } else {
    hipLaunchKernelGGL(attn_fwd_small_kernel<false>, dim3((ntiles + 3) / 4, p.H), dim3(MA_THREADS), 4 * SM_WAVE_LDS_FWD, st, p, ntiles,
                       (const bf16raw*)qkv, (bf16raw*)out, lse);
  }
  return check_launch("attn_fwd_small");
}
int attn_bwd_small_launch(const AttnP& p, const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv,
                          hipStream_t st) {
  const int G = 32 / p.L;
  const int ntiles = (p.S + G - 1) / G;
  if (options().attn_hw_bwd > 0) {
    const int w = hw_waves(options().attn_hw_bwd, p.H);
    const size_t lds = (size_t)w * SM_WAVE_LDS_BWD;
    static std::atomic<unsigned long long> attr_set{0};
    if (first_launch_on_device(attr_set)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_small_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(attn_bwd_small_kernel<true>, dim3(((p.H + w - 1) / w) * ntiles), dim3(64 * w), lds, st, p, ntiles,
                       (const bf16raw*)qkv, (const bf16raw*)o, (const bf16raw*)dout, lse, (bf16raw*)dqkv);
  } else {
    hipLaunchKernelGGL(attn_bwd_small_kernel<false>, dim3((ntiles + 3) / 4, p.H), dim3(MA_THREADS), 4 * SM_WAVE_LDS_BWD, st, p, ntiles,
                       (const bf16raw*)qkv, (const bf16raw*)o, (const bf16raw*)dout, lse, (bf16raw*)dqkv);
  }
  return check_launch("attn_bwd_small");
}

template <auto Kernel>
static void allow_lds(size_t lds) {
  // > 64 KB of dynamic LDS needs an explicit opt-in (Lp = 224: 56 KB of tiles + 16 KB of staging), once per kernel
  // instantiation (the template argument: one static per kernel) and device
  static std::atomic<unsigned long long> seen{0};
  if (lds > 65536 && first_launch_on_device(seen))
    hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

int attn_fwd_mfma_launch(const AttnP& p, const void* qkv, void* out, float* lse, hipStream_t st) {
  const int nt = (p.L + 31) >> 5, Lp = nt * 32;
  // (mg_div splits item -> (sequence, head) -> (clip, frame) with reciprocal multiplies that are exact below 2^24 items)
  if (nt == 7 && options().attn_fwd_stream && (long)p.S * p.H < (1L << 24)) {      // one persistent workgroup per CU, K / V of the next item streamed in by LDS-DMA
    allow_lds<attn_fwd_stream_mfma_kernel<7>>(MGF_LDS_BYTES);
    const int items = p.S * p.H, cus = device_cus();
    hipLaunchKernelGGL(attn_fwd_stream_mfma_kernel<7>, dim3(items < cus ? items : cus), dim3(MF_THREADS), MGF_LDS_BYTES, st, p,
                       (const bf16raw*)qkv, (bf16raw*)out, lse);
    return check_launch("attn_fwd_stream_mfma");
  }
  const size_t lds = (size_t)2 * Lp * 64 * 2 + 4 * MA_STAGE_ELEMS * 2;
  if (nt == 7) {                                   // L in 193 .. 224: the 197 tokens of every 224^2 / patch 16 model
    allow_lds<attn_fwd_mfma_kernel<7>>(lds);
    hipLaunchKernelGGL(attn_fwd_mfma_kernel<7>, dim3(p.S, p.H), dim3(MA_THREADS), lds, st, p, (const bf16raw*)qkv, (bf16raw*)out, lse);
  } else {
    allow_lds<attn_fwd_mfma_kernel<0>>(lds);
    hipLaunchKernelGGL(attn_fwd_mfma_kernel<0>, dim3(p.S, p.H), dim3(MA_THREADS), lds, st, p, (const bf16raw*)qkv, (bf16raw*)out, lse);
  }
  return check_launch("attn_fwd_mfma");
}

template <int NT_, int NTK_, int VAR_>
static int attn_bwd_mfma_launch_t(const AttnP& p, const void* qkv, const void* o, const void* dout, const float* lse,
                                  float* delta, void* dqkv, void* dqkv_cls, hipStream_t st) {
  const int Lp = ((p.L + 31) >> 5) * 32;
  const size_t lds = (size_t)2 * Lp * 64 * 2 + 4 * MA_STAGE_ELEMS * 2;
  allow_lds<attn_bwd_dq_mfma_kernel<NT_>>(lds);
  allow_lds<attn_bwd_dkv_mfma_kernel<NTK_, VAR_>>(lds + (size_t)2 * Lp * 4);
  hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<NT_>, dim3(p.S, p.H), dim3(MA_THREADS), lds, st, p, (const bf16raw*)qkv,
                     (const bf16raw*)o, (const bf16raw*)dout, lse, delta, (bf16raw*)dqkv, (bf16raw*)dqkv_cls);
  int rc = check_launch("attn_bwd_dq_mfma");
  if (rc) return rc;
  hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<NTK_, VAR_>), dim3(p.S, p.H), dim3(MA_THREADS), lds + (size_t)2 * Lp * 4, st, p,
                     (const bf16raw*)qkv, (const bf16raw*)dout, lse, delta, (bf16raw*)dqkv, (bf16raw*)dqkv_cls);
  return check_launch("attn_bwd_dkv_mfma");
}

template <int NT_>
static int attn_bwd_fused_launch_t(const AttnP& p, const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv,
                                   void* dqkv_cls, hipStream_t st) {
  const int Lp = ((p.L + 31) >> 5) * 32;
  const size_t lds = (size_t)4 * Lp * 64 * 2 + 8 * MA_STAGE_ELEMS * 2 + (size_t)2 * Lp * 4;
  allow_lds<attn_bwd_fused_mfma_kernel<NT_>>(lds);
  const int items = p.S * p.H, cus = device_cus();          // one workgroup per CU (its LDS holds one item), persistent
  hipLaunchKernelGGL(attn_bwd_fused_mfma_kernel<NT_>, dim3(items < cus ? items : cus), dim3(MF_THREADS), lds, st, p, (const bf16raw*)qkv,
                     (const bf16raw*)o, (const bf16raw*)dout, lse, (bf16raw*)dqkv, (bf16raw*)dqkv_cls);
  return check_launch("attn_bwd_fused_mfma");
}

static int attn_bwd_stream_launch(const AttnP& p, const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv,
                                  void* dqkv_cls, hipStream_t st) {
  allow_lds<attn_bwd_stream_mfma_kernel<7>>(MG_LDS_BYTES);
  const int items = p.S * p.H, cus = device_cus();          // one workgroup per CU, persistent over the items (heads fastest)
  hipLaunchKernelGGL(attn_bwd_stream_mfma_kernel<7>, dim3(items < cus ? items : cus), dim3(MF_THREADS), MG_LDS_BYTES, st, p,
                     (const bf16raw*)qkv, (const bf16raw*)o, (const bf16raw*)dout, lse, (bf16raw*)dqkv, (bf16raw*)dqkv_cls,
                     reinterpret_cast<long long*>(options().pp_trace));
  return check_launch("attn_bwd_stream_mfma");
}

int attn_bwd_mfma_launch(const AttnP& p, const void* qkv, const void* o, const void* dout, const float* lse,
                         float* delta, void* dqkv, void* dqkv_cls, hipStream_t st) {
  const int nt = (p.L + 31) >> 5;
  if (options().attn_fused >= 2 && nt == 7 && (long)p.S * p.H < (1L << 24)) return attn_bwd_stream_launch(p, qkv, o, dout, lse, dqkv, dqkv_cls, st);
  if (options().attn_fused && nt <= 7) {           // one pass over HBM: all four operand tiles fit the LDS of one workgroup
    if (nt == 7) return attn_bwd_fused_launch_t<7>(p, qkv, o, dout, lse, dqkv, dqkv_cls, st);
    return attn_bwd_fused_launch_t<0>(p, qkv, o, dout, lse, dqkv, dqkv_cls, st);
  }
  if (nt == 7) {
    switch (options().attn_dkv) {                  // dk / dv kernel: 0 = run-time tile loop, 1..4 = unrolled variants
      case 1: return attn_bwd_mfma_launch_t<7, 7, 1>(p, qkv, o, dout, lse, delta, dqkv, dqkv_cls, st);
      case 2: return attn_bwd_mfma_launch_t<7, 7, 0>(p, qkv, o, dout, lse, delta, dqkv, dqkv_cls, st);
      case 3: return attn_bwd_mfma_launch_t<7, 7, 2>(p, qkv, o, dout, lse, delta, dqkv, dqkv_cls, st);
      case 4: return attn_bwd_mfma_launch_t<7, 7, 3>(p, qkv, o, dout, lse, delta, dqkv, dqkv_cls, st);
      default: return attn_bwd_mfma_launch_t<7, 0, 0>(p, qkv, o, dout, lse, delta, dqkv, dqkv_cls, st);
    }
  }
  return attn_bwd_mfma_launch_t<0, 0, 0>(p, qkv, o, dout, lse, delta, dqkv, dqkv_cls, st);
}

}  // namespace vtx
