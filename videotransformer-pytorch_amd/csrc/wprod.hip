// wprod.hip -- small fp32 "weights x weights" products on the exact-fp32 matrix instruction.
//
// The merged attn.proj o temporal_fc GEMM of the divided temporal attention (reference transformer.py:268-275;
// vtx/functions.py TimeAttnFn) needs, per layer and step, three products of 768 x 768 fp32 matrices that no
// activation passes through:
//     forward    W_c   = W_tfc W_proj                    and  b_c = W_tfc b_proj (+ b_tfc / c)
//     backward   dW_tfc = c G W_proj^T + u b_proj^T
//                dW_proj = c W_tfc^T G                   and  db_proj = W_tfc^T u
// One kernel serves all of them:  C[N1,N2] (+)= alpha * sum_k A(i,k) B(k,j) (+ u[i] v[j]),  with strided element
// access A(i,k) = A[i*a_rs + k*a_ks], B(k,j) = B[k*b_ks + j*b_cs] (one stride of each is 1: that axis is loaded with
// 16-byte vectors), and optionally the vector product that shares the A tiles already on chip:
//     y[i] (+)= alpha_y * sum_k A(i,k) x[k] + beta_z * z[i].
//
// 32 x 32 output tile per 256-thread workgroup (round 6; 64 x 64 tiles staged through LDS before: 144 workgroups on 256 CUs, twelve
// barrier-separated K tiles, 33 - 38 us per 768^3 product against a matrix-pipe floor of 10): 4 waves = 4 quarters of K, every wave
// runs the whole 32 x 32 tile over its quarter (blocks h, h + 4, ... of 8 k) with v_mfma_f32_32x32x2_f32 (an fp32 fma chain) and takes its operands STRAIGHT from
// L2 into the instruction's registers -- no LDS staging, no barrier inside the K loop; the quarters are folded through LDS in a
// fixed order (deterministic).  Operand layout of the instruction: lane l holds A(i = l % 32, k = l / 32) and B(k = l / 32,
// j = l % 32).  A block of 8 consecutive k is four instructions; instruction s of a block takes k0 + s from lanes 0 .. 31 and
// k0 + 4 + s from lanes 32 .. 63, so that a lane's four values are ONE 16-byte load when the operand is contiguous along k (four
// 4-byte loads of 128 contiguous bytes per half wave otherwise).  768^3: 576 workgroups (2 - 3 per CU), 96 matrix instructions of
// 64 cycles per wave; operands are L2-resident (2.4 MB each, 113 MB of L2 reads per product).  WP_D blocks of look-ahead per wave.
// Algorithmic work per launch: 2*N1*N2*K flop, (N1*K + K*N2 + N1*N2) * 4 bytes.
#include "common.h"

namespace vtx {

#ifndef VTX_WP_PAD_KB
#define VTX_WP_PAD_KB 0
#endif
#ifndef VTX_WP_NACC
#define VTX_WP_NACC 2
#endif
#ifndef VTX_WP_D
#define VTX_WP_D 4
#endif
constexpr int WP_T = 32, WP_THREADS = 256, WP_D = VTX_WP_D;

struct WprodParams {
  int N1, N2, K;
  const float* A; long a_rs, a_ks;
  const float* B; long b_ks, b_cs;
  float alpha;
  float* C; long ldc; int accumulate;
  const float* u; const float* v;
  const float* x; float* y; float alpha_y; const float* z; float beta_z; int y_accumulate;
  long long* trace;                                    // VTX_WP_TRACE builds (tools/micro/wprod_timeline.py): 8 stamps per wave
};

#ifdef VTX_WP_TRACE
#define WP_STAMP(e_) if (p.trace != nullptr && lane == 0) { \
    long long* sl__ = p.trace + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + h) * 8; \
    sl__[e_] = (long long)__builtin_amdgcn_s_memrealtime(); if ((e_) == 0 || (e_) == 3) sl__[(e_) == 0 ? 6 : 7] = (long long)__builtin_amdgcn_s_memtime(); }
#else
#define WP_STAMP(e_)
#endif

// FULL: N1, N2 multiples of 32 and K of 32 * WP_D (no lane ever holds a row / column / k beyond the extents: no selects in the K loop);
// WITH_X: the vector product rides along (x loads and four multiply-adds per block)
template <bool A_KCONT, bool B_KCONT, bool FULL, bool WITH_X>
__global__ __launch_bounds__(WP_THREADS) void wprod_kernel(WprodParams p) {
  __shared__ float red[3][16][64];                     // accumulators of K quarters 1 .. 3
  __shared__ float yred[4][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);          // K quarter
  const int li = lane & 31, kh = lane >> 5;
  WP_STAMP(0)
  const int i0 = blockIdx.y * WP_T, j0 = blockIdx.x * WP_T;
  const int row = i0 + li, col = j0 + li;              // the lane's row of A / column of B
  const bool row_ok = row < p.N1, col_ok = col < p.N2;
  const bool with_y = WITH_X && blockIdx.x == 0;
  const int nb = (p.K + 7) >> 3;                       // blocks of 8 k (K is a multiple of 4: the last block may hold 4)
  // blocks per wave, a multiple of the look-ahead: the K loop below is straight-line code per trip (a wave-uniform `if` around a
  // slot makes hipcc wait for every load in flight at the top of each trip); blocks beyond K are all zeros
  const int per = (((nb + 3) >> 2) + WP_D - 1) / WP_D * WP_D;
  // wave h takes blocks h, h + 4, h + 8, ...: the four waves of a workgroup then walk the SAME 128-byte lines of a k-contiguous
  // operand at the same time (a line = four blocks of a row), so a line comes out of L2 once per workgroup and the workgroup's
  // window is 64 lines.  (With a contiguous quarter of K per wave every wave had its own 64-line window -- 9 .. 12 of them per
  // CU against a 32-KB vector L1 -- and every line was fetched four times: TCP_TCC_READ_REQ 4x the operand bytes.)
  const int b_lo = 0, b_hi = per;
  const float* const arow = p.A + (A_KCONT ? (long)(row_ok ? row : 0) * p.a_rs : (long)(row_ok ? row : 0));
  const float* const bcol = p.B + (B_KCONT ? (long)(col_ok ? col : 0) * p.b_cs : (long)(col_ok ? col : 0));

  // x rides along in EVERY workgroup of a launch that has one (one broadcast 16-byte load per block): a load under a run-time
  // branch makes hipcc wait for ALL loads in flight at the join, which would empty the look-ahead
  const float* const xbase = WITH_X ? p.x : p.A;
  float a[WP_D][4], b[WP_D][4], xv[WP_D][4];
  auto load = [&](int blk, float (&va)[4], float (&vb)[4], float (&vx)[4]) {      // the lane's four k of block blk: kk .. kk + 3
    const int kk = (blk * 4 + h) * 8 + kh * 4;
    const int ks = kk < p.K ? kk : 0;                  // blocks / vectors beyond K read the first one (a whole vector is inside or
                                                       // outside K); nothing is zeroed here -- a select on a value that has just been
                                                       // requested is a wait for it
    if (A_KCONT) {
      const float4 t = *reinterpret_cast<const float4*>(arow + ks);
      va[0] = t.x; va[1] = t.y; va[2] = t.z; va[3] = t.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) va[s] = arow[(long)(ks + s) * p.a_ks];
    }
    if (B_KCONT) {
      const float4 t = *reinterpret_cast<const float4*>(bcol + ks);
      vb[0] = t.x; vb[1] = t.y; vb[2] = t.z; vb[3] = t.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) vb[s] = bcol[(long)(ks + s) * p.b_ks];
    }
    if constexpr (WITH_X) {
      const float4 t = *reinterpret_cast<const float4*>(xbase + ks);
      vx[0] = t.x; vx[1] = t.y; vx[2] = t.z; vx[3] = t.w;
    }
  };

  // VTX_WP_NACC accumulators take the matrix instructions of a block in turn (2: instructions 0, 2 and 1, 3 -- no instruction
  // waits for the one right in front of it; the two are added at the end)
  f32x16 accs[VTX_WP_NACC];
#pragma unroll
  for (int c = 0; c < VTX_WP_NACC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[c][r] = 0.f;
  float yacc = 0.f;
#pragma unroll
  for (int d = 0; d < WP_D; ++d) { load(b_lo + d, a[d], b[d], xv[d]); __builtin_amdgcn_sched_barrier(0); }
  WP_STAMP(1)
  bool first_trip = true;
  for (int blk = b_lo; blk < b_hi; blk += WP_D) {
#pragma unroll
    for (int d = 0; d < WP_D; ++d) {
      const bool kok = FULL || ((blk + d) * 4 + h) * 8 + kh * 4 < p.K;   // rows / columns / k beyond the extents contribute zeros
      const bool aok = FULL || (kok && row_ok), bok = FULL || (kok && col_ok);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = aok ? a[d][s] : 0.f, bv = bok ? b[d][s] : 0.f;
        accs[s % VTX_WP_NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accs[s % VTX_WP_NACC], 0, 0, 0);
        if constexpr (WITH_X) yacc = fmaf(av, xv[d][s], yacc);
      }
      load(blk + d + WP_D, a[d], b[d], xv[d]);         // the slot is free again (beyond the wave's range: loaded, never used)
      __builtin_amdgcn_sched_barrier(0);               // keep the request order: hipcc otherwise sinks the loads to their uses (no look-ahead left)
    }
#ifdef VTX_WP_TRACE
    if (first_trip) { WP_STAMP(2) first_trip = false; }
#endif
  }
  WP_STAMP(3)
  f32x16 acc = accs[0];
#pragma unroll
  for (int c = 1; c < VTX_WP_NACC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += accs[c][r];
  // fold the four K quarters in the order 0, 1, 2, 3: waves 1 .. 3 hand their accumulators to wave 0 through LDS
  if (h > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[h - 1][r][lane] = acc[r];
  }
  if (with_y) {                                        // the lane pair (l, l + 32) holds the two k halves of row l % 32
    const float other = __shfl_xor(yacc, 32, 64);
    if (kh == 0) yred[h][li] = yacc + other;
  }
  __syncthreads();
  WP_STAMP(4)
  if (h == 0) {
    const float vj = (p.u && col_ok) ? p.v[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int orow = i0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (orow < p.N1 && col_ok) {
        float val = p.alpha * (((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane]);
        if (p.u) val = fmaf(p.u[orow], vj, val);
        float* dst = p.C + (long)orow * p.ldc + col;
        if (p.accumulate) val += *dst;
        *dst = val;
      }
    }
    if (with_y && kh == 0 && row_ok) {
      float val = p.alpha_y * (((yred[0][li] + yred[1][li]) + yred[2][li]) + yred[3][li]);
      if (p.z) val = fmaf(p.beta_z, p.z[row], val);
      if (p.y_accumulate) val += p.y[row];
      p.y[row] = val;
    }
  }
  WP_STAMP(5)
}

}  // namespace vtx

using namespace vtx;

extern "C" int vtx_wprod(const vtx_wprod_desc* d, void* stream) {
  VTX_REQUIRE(d != nullptr, VTX_EINVAL, "wprod: null descriptor");
  VTX_REQUIRE(d->N1 > 0 && d->N2 > 0 && d->K > 0, VTX_EINVAL, "wprod: bad shape N1=%d N2=%d K=%d", d->N1, d->N2, d->K);
  VTX_REQUIRE(d->N1 % 4 == 0 && d->N2 % 4 == 0 && d->K % 4 == 0, VTX_EINVAL, "wprod: N1, N2, K must be multiples of 4");
  VTX_REQUIRE(d->A && d->B && d->C, VTX_EINVAL, "wprod: null operand");
  VTX_REQUIRE((d->a_rs == 1) != (d->a_ks == 1), VTX_EINVAL, "wprod: exactly one stride of A must be 1 (a_rs=%ld a_ks=%ld)",
              d->a_rs, d->a_ks);
  VTX_REQUIRE((d->b_ks == 1) != (d->b_cs == 1), VTX_EINVAL, "wprod: exactly one stride of B must be 1 (b_ks=%ld b_cs=%ld)",
              d->b_ks, d->b_cs);
  const long a_ld = d->a_ks == 1 ? d->a_rs : d->a_ks, b_ld = d->b_ks == 1 ? d->b_cs : d->b_ks;
  VTX_REQUIRE(aligned16(d->A) && aligned16(d->B) && a_ld % 4 == 0 && b_ld % 4 == 0 && (!d->x || aligned16(d->x)), VTX_EALIGN,
              "wprod: operands (and x) must be 16-byte aligned with leading dimensions that are multiples of 4");
  VTX_REQUIRE(d->ldc >= d->N2, VTX_EINVAL, "wprod: ldc=%ld < N2=%d", d->ldc, d->N2);
  VTX_REQUIRE((d->u == nullptr) == (d->v == nullptr), VTX_EINVAL, "wprod: the rank-1 term needs both u and v");
  VTX_REQUIRE((d->y == nullptr) == (d->x == nullptr), VTX_EINVAL, "wprod: the vector product needs both x and y");
  WprodParams p;
  p.N1 = d->N1; p.N2 = d->N2; p.K = d->K;
  p.A = d->A; p.a_rs = d->a_rs; p.a_ks = d->a_ks;
  p.B = d->B; p.b_ks = d->b_ks; p.b_cs = d->b_cs;
  p.alpha = d->alpha; p.C = d->C; p.ldc = d->ldc; p.accumulate = d->accumulate;
  p.u = d->u; p.v = d->v;
  p.trace = nullptr;
#ifdef VTX_WP_TRACE
  p.trace = reinterpret_cast<long long*>(options().pp_trace);
#endif
  p.x = d->x; p.y = d->y; p.alpha_y = d->alpha_y; p.z = d->z; p.beta_z = d->beta_z; p.y_accumulate = d->y_accumulate;
  const dim3 grid(cdiv(d->N2, WP_T), cdiv(d->N1, WP_T)), block(WP_THREADS);
  hipStream_t st = as_stream(stream);
  const bool ak = d->a_ks == 1, bk = d->b_ks == 1;
  // (VTX_WP_PAD_KB: unused dynamic LDS as an occupancy limiter, an experiment: the dispatcher was suspected of filling CUs one by
  // one; tools/micro/census.hip shows that it places the 576 workgroups 2 - 3 per CU whatever the limit, and 0 / 36 / 56 KB time the same)
  const size_t pad = (size_t)VTX_WP_PAD_KB * 1024;
  // FULL: whole tiles, and K a whole number of look-ahead trips of all four waves (no block beyond K is ever walked)
  const bool full = d->N1 % 32 == 0 && d->N2 % 32 == 0 && d->K % (32 * WP_D) == 0, wx = d->x != nullptr;
#define WP_LAUNCH(AK_, BK_)                                                                                          \
  {                                                                                                                  \
    if (full && wx) hipLaunchKernelGGL((wprod_kernel<AK_, BK_, true, true>), grid, block, pad, st, p);               \
    else if (full) hipLaunchKernelGGL((wprod_kernel<AK_, BK_, true, false>), grid, block, pad, st, p);               \
    else if (wx) hipLaunchKernelGGL((wprod_kernel<AK_, BK_, false, true>), grid, block, pad, st, p);                 \
    else hipLaunchKernelGGL((wprod_kernel<AK_, BK_, false, false>), grid, block, pad, st, p);                        \
  }
  if (ak && bk) WP_LAUNCH(true, true)
  else if (ak) WP_LAUNCH(true, false)
  else if (bk) WP_LAUNCH(false, true)
  else WP_LAUNCH(false, false)
#undef WP_LAUNCH
  return check_launch("wprod");
}
