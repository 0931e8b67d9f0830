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
// 64 x 64 output tile per 512-thread workgroup: 8 waves = 4 quadrants of 32 x 32 (v_mfma_f32_32x32x2_f32, an fp32 fma
// chain) x 2 halves of every 64-deep K tile, folded through LDS in a fixed order (deterministic).  768^3: 144
// workgroups, 192 matrix instructions of 64 cycles per wave = 12.3k cycles; operands are L2-resident (2.4 MB each).
// LDS tiles are k-major [64][65]: fragment reads are 32 consecutive words per lane group, and both store patterns
// (vector along k / vector along i) are bank-conflict free with the odd row stride.
// Algorithmic work per launch: 2*N1*N2*K flop, (N1*K + K*N2 + N1*N2) * 4 bytes.
#include "common.h"

namespace vtx {

constexpr int WP_T = 64, WP_BK = 64, WP_LD = 65, WP_THREADS = 512, WP_V = WP_T * WP_BK / 4 / WP_THREADS;   // float4 per thread, operand and K tile

struct WprodParams {
  int N1, N2, K;
  const float* A; long a_rs, a_ks;
  const float* B; long b_ks, b_cs;
  float alpha;
  float* C; long ldc; int accumulate;
  const float* u; const float* v;
  const float* x; float* y; float alpha_y; const float* z; float beta_z; int y_accumulate;
};

template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(WP_THREADS) void wprod_kernel(WprodParams p) {
  __shared__ __attribute__((aligned(16))) float As[2][WP_BK][WP_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][WP_BK][WP_LD];
  __shared__ float xs[2][WP_BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = wave & 3, qi = q >> 1, qj = q & 1, h = wave >> 2;
  const int i0 = blockIdx.y * WP_T, j0 = blockIdx.x * WP_T;
  const bool with_y = p.y != nullptr && blockIdx.x == 0;

  // loader coordinates of float4 v of this thread (index tid + v * 512 of the tile's 1024): k-contiguous operands are cut into
  // float4 along k (16 per row of the other axis), the others along i / j (16 per k)
  auto a_k = [&](int v) { const int id = tid + v * WP_THREADS; return A_KCONT ? (id & 15) * 4 : id >> 4; };
  auto a_i = [&](int v) { const int id = tid + v * WP_THREADS; return A_KCONT ? id >> 4 : (id & 15) * 4; };
  auto b_k = [&](int v) { const int id = tid + v * WP_THREADS; return B_KCONT ? (id & 15) * 4 : id >> 4; };
  auto b_j = [&](int v) { const int id = tid + v * WP_THREADS; return B_KCONT ? id >> 4 : (id & 15) * 4; };
  // Operand tiles are L2-resident but an L2 round trip (~1 us under load) is several times the matrix instructions of a
  // K tile: three K tiles of register prefetch keep the loop on the matrix pipe instead of on that latency
  // (one tile of look-ahead measured 45 us per 768^3 product, latency-bound).  64-deep K tiles (round 4; 32 before): half as many
  // barrier-separated iterations, each of which is bound by its latency chain, not by its 16 matrix instructions.
  constexpr int PF = 3;
  float4 ra[PF][WP_V], rb[PF][WP_V];
  float rx[PF] = {0.f, 0.f, 0.f};
  auto gload = [&](int k0, float4 (&va)[WP_V], float4 (&vb)[WP_V], float& vx) {
#pragma unroll
    for (int v = 0; v < WP_V; ++v) {
      {
        const int i = i0 + a_i(v), k = k0 + a_k(v);
        const bool ok = i < p.N1 && k < p.K;         // extents are multiples of 4: a vector is inside or outside as a whole
        const float* src = p.A + (ok ? (long)i * p.a_rs + (long)k * p.a_ks : 0L);
        va[v] = *reinterpret_cast<const float4*>(src);
        if (!ok) va[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      {
        const int j = j0 + b_j(v), k = k0 + b_k(v);
        const bool ok = j < p.N2 && k < p.K;
        const float* src = p.B + (ok ? (long)k * p.b_ks + (long)j * p.b_cs : 0L);
        vb[v] = *reinterpret_cast<const float4*>(src);
        if (!ok) vb[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (with_y && tid < WP_BK) vx = (k0 + tid) < p.K ? p.x[k0 + tid] : 0.f;
  };
  auto lstore = [&](int buf, const float4 (&va)[WP_V], const float4 (&vb)[WP_V], float vx) {
#pragma unroll
    for (int v = 0; v < WP_V; ++v) {
      const int ak = a_k(v), ai = a_i(v), bk = b_k(v), bj = b_j(v);
      if (A_KCONT) { As[buf][ak][ai] = va[v].x; As[buf][ak + 1][ai] = va[v].y; As[buf][ak + 2][ai] = va[v].z; As[buf][ak + 3][ai] = va[v].w; }
      else { As[buf][ak][ai] = va[v].x; As[buf][ak][ai + 1] = va[v].y; As[buf][ak][ai + 2] = va[v].z; As[buf][ak][ai + 3] = va[v].w; }
      if (B_KCONT) { Bs[buf][bk][bj] = vb[v].x; Bs[buf][bk + 1][bj] = vb[v].y; Bs[buf][bk + 2][bj] = vb[v].z; Bs[buf][bk + 3][bj] = vb[v].w; }
      else { Bs[buf][bk][bj] = vb[v].x; Bs[buf][bk][bj + 1] = vb[v].y; Bs[buf][bk][bj + 2] = vb[v].z; Bs[buf][bk][bj + 3] = vb[v].w; }
    }
    if (with_y && tid < WP_BK) xs[buf][tid] = vx;
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float yacc = 0.f;
  const int fa = qi * 32 + (lane & 31), fb = qj * 32 + (lane & 31), fk = h * (WP_BK / 2) + (lane >> 5);
  const int nk = (p.K + WP_BK - 1) / WP_BK;
  // register slot of K tile t = t % PF; LDS buffer = t & 1.  Tile t is consumed while tiles t+1 .. t+PF are in flight.
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (s < nk) gload(s * WP_BK, ra[s], rb[s], rx[s]);
  lstore(0, ra[0], rb[0], rx[0]);
  if (PF < nk) gload(PF * WP_BK, ra[0], rb[0], rx[0]);
  __syncthreads();
  auto ktile = [&](int kt, float4 (&va)[WP_V], float4 (&vb)[WP_V], float& vx) {   // va/vb/vx: the slot that holds tile kt + 1
    const int buf = kt & 1;
#pragma unroll
    for (int s = 0; s < WP_BK / 4; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][fk + 2 * s][fa], Bs[buf][fk + 2 * s][fb], acc, 0, 0, 0);
    if (with_y && wave == 0) {                        // rows i0 .. i0+63 of A times x, k ascending: lane = row
#pragma unroll
      for (int k = 0; k < WP_BK; ++k) yacc = fmaf(As[buf][k][lane], xs[buf][k], yacc);
    }
    if (kt + 1 < nk) {
      lstore(buf ^ 1, va, vb, vx);
      if (kt + 1 + PF < nk) gload((kt + 1 + PF) * WP_BK, va, vb, vx);   // the slot is free again: tile kt + 1 + PF
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += PF) {
    ktile(kt, ra[1], rb[1], rx[1]);
    if (kt + 1 < nk) ktile(kt + 1, ra[2], rb[2], rx[2]);
    if (kt + 2 < nk) ktile(kt + 2, ra[0], rb[0], rx[0]);
  }
  // fold the two K halves: waves 4..7 hand their accumulators to waves 0..3 through LDS (the operand tiles are dead)
  float* red = &As[0][0][0];                          // [4][16][64] floats = 16 KB <= sizeof(As)
  if (h == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(q * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (h == 0) {
    const int col = j0 + qj * 32 + (lane & 31);
    const float vj = (p.u && col < p.N2) ? p.v[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < p.N1 && col < p.N2) {
        float val = p.alpha * (acc[r] + red[(q * 16 + r) * 64 + lane]);
        if (p.u) val = fmaf(p.u[row], vj, val);
        float* dst = p.C + (long)row * p.ldc + col;
        if (p.accumulate) val += *dst;
        *dst = val;
      }
    }
    if (with_y && wave == 0) {
      const int row = i0 + lane;
      if (row < p.N1) {
        float val = p.alpha_y * yacc;
        if (p.z) val = fmaf(p.beta_z, p.z[row], val);
        if (p.y_accumulate) val += p.y[row];
        p.y[row] = val;
      }
    }
  }
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
  VTX_REQUIRE(aligned16(d->A) && aligned16(d->B) && a_ld % 4 == 0 && b_ld % 4 == 0, VTX_EALIGN,
              "wprod: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  VTX_REQUIRE(d->ldc >= d->N2, VTX_EINVAL, "wprod: ldc=%ld < N2=%d", d->ldc, d->N2);
  VTX_REQUIRE((d->u == nullptr) == (d->v == nullptr), VTX_EINVAL, "wprod: the rank-1 term needs both u and v");
  VTX_REQUIRE((d->y == nullptr) == (d->x == nullptr), VTX_EINVAL, "wprod: the vector product needs both x and y");
  WprodParams p;
  p.N1 = d->N1; p.N2 = d->N2; p.K = d->K;
  p.A = d->A; p.a_rs = d->a_rs; p.a_ks = d->a_ks;
  p.B = d->B; p.b_ks = d->b_ks; p.b_cs = d->b_cs;
  p.alpha = d->alpha; p.C = d->C; p.ldc = d->ldc; p.accumulate = d->accumulate;
  p.u = d->u; p.v = d->v;
  p.x = d->x; p.y = d->y; p.alpha_y = d->alpha_y; p.z = d->z; p.beta_z = d->beta_z; p.y_accumulate = d->y_accumulate;
  const dim3 grid(cdiv(d->N2, WP_T), cdiv(d->N1, WP_T)), block(WP_THREADS);
  hipStream_t st = as_stream(stream);
  const bool ak = d->a_ks == 1, bk = d->b_ks == 1;
  if (ak && bk) hipLaunchKernelGGL((wprod_kernel<true, true>), grid, block, 0, st, p);
  else if (ak) hipLaunchKernelGGL((wprod_kernel<true, false>), grid, block, 0, st, p);
  else if (bk) hipLaunchKernelGGL((wprod_kernel<false, true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((wprod_kernel<false, false>), grid, block, 0, st, p);
  return check_launch("wprod");
}
