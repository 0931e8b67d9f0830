// common.h -- shared device/host helpers for libvtx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include "../../include/vtx.h"

namespace vtx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short bf16raw;  // storage type of a bf16 element

// ---- bf16 <-> fp32 (round-to-nearest-even, NaN preserved) ------------------
__host__ __device__ inline float bf2f(bf16raw v) {
  union { uint32_t u; float f; } c; c.u = ((uint32_t)v) << 16; return c.f;
}
__host__ __device__ inline bf16raw f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  // gfx950 converts in hardware (v_cvt_pk_bf16_f32, RNE, quiet NaN); the bit-twiddling below compiles
  // to a ~12-instruction exec-masked sequence per element, which dominated the GEMM epilogues.
  union { __bf16 b; bf16raw r; } c; c.b = (__bf16)f; return c.r;
#else
  union { uint32_t u; float f; } c; c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16raw)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16raw)(u >> 16);
#endif
}

// Element-type traits: T is `float` (VTX_F32) or `bf16raw` (VTX_BF16).
template <typename T> struct ET;
template <> struct ET<float> {
  static constexpr int VEC = 4;               // elements per 16-byte vector
  __device__ static inline float ld(const float* p) { return *p; }
  __device__ static inline void st(float* p, float v) { *p = v; }
  __device__ static inline void st4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct ET<bf16raw> {
  static constexpr int VEC = 8;
  __device__ static inline float ld(const bf16raw* p) { return bf2f(*p); }
  __device__ static inline void st(bf16raw* p, float v) { *p = f2bf(v); }
  __device__ static inline void st4(bf16raw* p, const float (&v)[4]) {       // 8-byte aligned
    uint2 r;
    r.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    r.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = r;
  }
};

// 8 consecutive elements <-> 8 floats (bf16: one 16-B access; f32: two).
__device__ inline void load8(const float* p, float (&v)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ inline void load8(const bf16raw* p, float (&v)[8]) {
  uint4 r = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ inline void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ inline void store8(bf16raw* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = o;
}

// streaming form (nontemporal): results that the launch itself never reads back
__device__ inline void store8_nt(bf16raw* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
  __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p));
}

// ---- row maps ---------------------------------------------------------------
// Row indices fit 32 bits: use a 32-bit unsigned division (a 64-bit one costs ~100 instructions
// and this sits in GEMM loaders / epilogues).
__host__ __device__ inline long map_row(const vtx_rowmap& m, long r) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (m.tab != nullptr) return (long)m.base + r + (long)m.tab[(unsigned)r / (unsigned)m.grp];     // table form (grp > 0)
#endif
  const long q = (m.grp > 0) ? (long)((unsigned)r / (unsigned)m.grp) * (long)m.skip : 0;
  return (long)m.base + r + q;
}
inline vtx_rowmap ident_map() { vtx_rowmap m; m.grp = 0; m.skip = 0; m.base = 0; m.tab = nullptr; return m; }
inline bool closed_form(const vtx_rowmap& m) { return m.tab == nullptr; }

// ---- wave reductions (wave = 64) ---------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf-GELU (the reference's nn.GELU()) and its derivative, branch-free.  libm's erff compiles to two
// exec-masked branches per element, which made the FFN epilogues VALU-bound (fc1 forward 370 us vs 264 us
// for the same GEMM without the activation).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, five
// FMAs + one v_rcp_f32 + one v_exp_f32); exp(-x^2/2) is shared between erf(x/sqrt2) and the Gaussian term.
// Every multiply-add below is an explicit fmaf and contraction is off, so the value is the same in every
// kernel the epilogue is inlined into (the clip-independence test compares outputs across GEMM variants).
__device__ inline void gelu_parts(float x, float& cdf, float& gauss) {
#pragma clang fp contract(off)
  const float u = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float uu = u * u;
  gauss = __builtin_amdgcn_exp2f(uu * -1.4426950408889634f);        // exp(-x^2/2)
  const float q = (0.5f * p) * t;
  const float half_tail = q * gauss;                                 // 0.5 * erfc(|x|/sqrt2)
  const float upper = 1.0f - half_tail;
  cdf = x >= 0.f ? upper : half_tail;                                // 0.5 * (1 + erf(x/sqrt2))
}
__device__ inline float gelu_erf(float x) {
#pragma clang fp contract(off)
  float cdf, g;
  gelu_parts(x, cdf, g);
  return x * cdf;
}
__device__ inline float gelu_erf_grad(float x) {
#pragma clang fp contract(off)
  // d/dx [x Phi(x)] = Phi(x) + x * exp(-x^2/2) / sqrt(2 pi)
  float cdf, g;
  gelu_parts(x, cdf, g);
  const float xs = x * 0.3989422804014327f;
  return fmaf(xs, g, cdf);
}
// value and derivative from ONE evaluation of the shared parts (the FFN forward writes both: the backward's
// epilogue is then a plain multiply); each equals gelu_erf / gelu_erf_grad of the same x bit for bit
__device__ inline void gelu_erf_both(float x, float& val, float& grad) {
#pragma clang fp contract(off)
  float cdf, g;
  gelu_parts(x, cdf, g);
  const float xs = x * 0.3989422804014327f;
  grad = fmaf(xs, g, cdf);
  val = x * cdf;
}

// ---- tuning / diagnostic switches ----------------------------------------------------
// Process-wide, set through vtx_set_option() (initial values come from the VTX_* environment
// variables, read ONCE when the library is first used -- never on the launch path).
enum { NT_AUTO = 0, NT_PP256, NT_DMA2, NT_RING128X3, NT_RING128X4K32, NT_RING256X3, NT_RING256X3K32, NT_RING256X4K32 };
enum { TN_AUTO = 0, TN_PP256, TN_RING, TN_DMA2, TN_W4 };
struct Options {
  int gemm_nt = NT_AUTO;     // VTX_GEMM_NT: kernel family override of vtx_gemm_nt (bf16)
  int gemm_tn = TN_AUTO;     // VTX_GEMM_TN: ... of vtx_gemm_tn (bf16)
  int gemm_nodma = 0;        // VTX_GEMM_NODMA: register-staged GEMM kernels (no LDS-DMA)
  int tn_safe = 0;           // VTX_TN_SAFE: bounds-checked TN loader (diagnostic)
  int tn_cus = 256;          // VTX_TN_CUS: compute units the weight-gradient kernel's one-round slab split is sized for.  Its workgroups
                             // own a CU each; with fewer CUs free than workgroups (an RCCL collective in flight holds some) the launch
                             // runs a second, nearly empty round (1.4 - 1.6x, profiles/round6_cu_contention.txt).  240 keeps it to one
                             // round with up to 16 CUs held, at +4 .. 7 % per launch when none is.  Changes the slab partition, i.e. the
                             // fp32 summation order of the weight gradients (still fixed for a given value: bit-reproducible).
  int attn_valu = 0;         // VTX_ATTN_VALU: fp32-VALU attention kernels also for bf16
  int attn_hw_fwd = 16;      // VTX_ATTN_HW_FWD / _BWD: short-sequence attention with n heads of a row tile in one workgroup
  int attn_hw_bwd = 4;       //   (0: one head per workgroup, four row tiles; backward: 4 heads -- 512 contiguous bytes per row -- measured best)
  int attn_fused = 2;        // VTX_ATTN_FUSED: backward of the 33..224-token attention: 2 = one phase per (sequence, head), operands streamed (193..224
                             // tokens; other lengths as 1), 1 = one kernel with a dq and a dk / dv phase (one pass over HBM), 0 = dq + dkv kernels
  int attn_fwd_stream = 1;   // VTX_ATTN_FWD_STREAM: forward of the 193..224-token attention as a persistent kernel with streamed K / V (bit-identical results)
  int attn_dkv = 3;          // VTX_ATTN_DKV: dk / dv kernel of the 197-token attention: 0 = run-time query-tile loop, 1..4 = unrolled variants
                             // (3 = unrolled, next key tile loaded behind the current one: profiles/round3_attn_dkv_variants.txt)
  int ln_rows = 3;           // VTX_LN_ROWS: rows per trip of the LayerNorm forward kernel (1 = one row per wave, the round-1 kernel; 2 .. 4: ln_fwd2_kernel)
  int pp_grid = 256;         // VTX_GEMM_PP_GRID: resident workgroups of the persistent NT GEMM
  int pp_cg = 0;             // VTX_GEMM_PP_CG: column tiles per group (0: from K)
  int pp_epi = 0;            // VTX_GEMM_PP_EPI: 1 = per-pass epilogue (A/B timing); 2 / 3 = diagnostics: no stores / no staging
  int pp_cont = 1;           // VTX_GEMM_PP_CONT: continuous flow of the persistent NT GEMM (next tile's first K tiles requested
                             // inside the current main loop) for epilogues that leave the operand ring alone; 0 = per-tile prologue
  unsigned long long pp_trace = 0;   // device address of a long long[256][8][8] timeline buffer (tools/pp_timeline.py), 0 = off
};
Options& options();

// ---- host-side error plumbing --------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// out[n] (+)= scale * sum_s part[s*stride + n]; columns >= split optionally go to out2, summed over `fold` copies per
// slab (ln.hip)
int launch_reduce_partials(const float* part, int nslabs, long stride, long N, float* out, int accumulate, float scale,
                           hipStream_t st, float* out2 = nullptr, long split = 0, int accumulate2 = 0, int fold = 1,
                           long fold_stride = 0);
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define VTX_REQUIRE(cond, code, ...)        \
  do {                                      \
    if (!(cond)) {                          \
      ::vtx::set_error(__VA_ARGS__);        \
      return (code);                        \
    }                                       \
  } while (0)

// hipFuncSetAttribute applies to the CURRENT device only: a kernel that needs more than 64 KB of dynamic LDS is
// configured once per (instantiation, device).  `seen` is the call site's static mask; true = configure it now.
inline bool first_launch_on_device(std::atomic<unsigned long long>& seen) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  return (seen.fetch_or(bit, std::memory_order_relaxed) & bit) == 0;
}

// compute units of the current device (asked once per device)
inline int device_cus() {
  static std::atomic<int> cached[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int v = cached[dev & 63].load(std::memory_order_relaxed);
  if (v <= 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cached[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace vtx
