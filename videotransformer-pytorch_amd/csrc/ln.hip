// ln.hip -- LayerNorm forward / backward (HBM-bound; one wave64 per row).
//
// Reference: nn.LayerNorm instances transformer.py:215,321,418,495 (eps 1e-5)
// and video_transformer.py:119,401 (eps 1e-6).  Statistics in fp32, biased
// variance.  Algorithmic bytes per row: read D + write D elements (+8 B stats).
//
// Layout: lane l owns elements {4*(l + 64*c) .. +3 : c < NCH}; a wave load is a
// fully coalesced 64 x (4 elements) segment.  Row reductions are xor-shuffles.
#include "common.h"

namespace vtx {

template <typename T> __device__ inline void ld4(const T* p, float (&v)[4]);
template <> __device__ inline void ld4<float>(const float* p, float (&v)[4]) {
  float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ inline void ld4<bf16raw>(const bf16raw* p, float (&v)[4]) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
// streaming (nontemporal) form for rows that are read once: the forward kernel has one row per wave in flight (8 waves
// per SIMD x 1.5 KB = 48 KB per CU against a 32-KB vector L1); with plain loads it ran at 4.4 TB/s and got SLOWER with
// more rows in flight (second row prefetched: 150 us, with `nt` 142 us; two rows per trip: 145 us); `nt` loads:
// 104.8 -> 95.6 us.  The backward kernel (two rows per trip already): 205 -> 183 us.  NOT for the attention kernels'
// fragment loads: those touch a 128-B line in four 32-B pieces and need the L1 to merge them (temporal forward 182 -> 275 us).
typedef unsigned ln_u32x2 __attribute__((ext_vector_type(2)));
typedef float ln_f32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ inline void ld4_nt(const T* p, float (&v)[4]);
template <> __device__ inline void ld4_nt<float>(const float* p, float (&v)[4]) {
  ln_f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const ln_f32x4*>(p)); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ inline void ld4_nt<bf16raw>(const bf16raw* p, float (&v)[4]) {
  ln_u32x2 r = __builtin_nontemporal_load(reinterpret_cast<const ln_u32x2*>(p));
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <typename T> __device__ inline void st4(T* p, const float (&v)[4]);
template <> __device__ inline void st4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ inline void st4<bf16raw>(bf16raw* p, const float (&v)[4]) {
  uint2 r;
  r.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  r.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = r;
}

constexpr int LN_WAVES = 4;

// raw 4-element vectors (kept packed while a prefetched row waits in registers)
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef float4 type; };
template <> struct Raw4<bf16raw> { typedef uint2 type; };
__device__ inline float4 ld_raw_nt(const float4* p) {
  const ln_f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const ln_f32x4*>(p));
  return make_float4(a.x, a.y, a.z, a.w);
}
__device__ inline uint2 ld_raw_nt(const uint2* p) {
  const ln_u32x2 a = __builtin_nontemporal_load(reinterpret_cast<const ln_u32x2*>(p));
  return make_uint2(a.x, a.y);
}
__device__ inline void unpack4(const float4& r, float (&v)[4]) { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
__device__ inline void unpack4(const uint2& r, float (&v)[4]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}


template <typename T, int NCH>
__global__ __launch_bounds__(LN_WAVES * 64) void ln_fwd_kernel(
    int rows, int D, const T* __restrict__ x, long ldx, vtx_rowmap xmap,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    T* __restrict__ y, long ldy, vtx_rowmap ymap, float* __restrict__ mean_out,
    float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long total_waves = (long)gridDim.x * LN_WAVES;
  const float invD = 1.0f / (float)D;
  // gamma / beta of this lane's columns stay in registers across the row loop (re-reading them per row
  // is 4x the L1 traffic of the bf16 row itself)
  float gm[NCH][4], bt[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = 4 * (lane + 64 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gm[c][j] = (col < D) ? gamma[col + j] : 0.f;
      bt[c][j] = (col < D) ? beta[col + j] : 0.f;
    }
  }
  for (long r = (long)blockIdx.x * LN_WAVES + wave; r < rows; r += total_waves) {
    const T* xr = x + map_row(xmap, r) * ldx;
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (col < D) {
        ld4_nt<T>(xr + col, v[c]);
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
      } else {
        v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
      }
    }
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (col < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[c][j] - mu; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invD + eps);
    T* yr = y + map_row(ymap, r) * ldy;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (col < D) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mu) * rstd * gm[c][j] + bt[c][j];
        st4<T>(yr + col, o);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[r] = mu;
      if (rstd_out) rstd_out[r] = rstd;
    }
  }
}

// NR rows per trip (option ln_rows = 2 .. 4, default 3; round 4): the raw segments of ALL NR rows are requested before the
// first one is reduced, as in the backward kernel below.  FULL (D == 256 NCH) is a template flag so that the row loop carries no
// exec-masked load -- with one in it hipcc waits vmcnt(0) per load and the later rows' requests do not overlap the first row's
// reductions, which is why round 2's "next row prefetched" / "two rows per trip" experiments on the kernel above (whose `col < D`
// is a run-time predicate) measured SLOWER.  Same arithmetic per row: bit-identical outputs.  150 624 rows of 768 bf16, same box,
// interleaved: 95.5 us (one row per wave) -> 79.8 / 76.3 / 77.3 us for NR = 2 / 3 / 4 = 6.06 TB/s at NR = 3 (0.76 of 8 TB/s,
// the streaming-copy rate of this chip is ~6.3); 12 552 rows (8 clips): 18.5 -> 11.2 us; D = 1024: 202 -> 115 us.  A variant with
// 16 bytes per lane and access (1.5 accesses per 768-wide row) measured 124 us and was dropped.
template <typename T, int NCH, bool FULL, int NR>
__global__ __launch_bounds__(LN_WAVES * 64) void ln_fwd2_kernel(
    int rows, int D, const T* __restrict__ x, long ldx, vtx_rowmap xmap,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    T* __restrict__ y, long ldy, vtx_rowmap ymap, float* __restrict__ mean_out,
    float* __restrict__ rstd_out) {
  typedef typename Raw4<T>::type raw_t;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long total_waves = (long)gridDim.x * LN_WAVES;
  const float invD = 1.0f / (float)D;
  float gm[NCH][4], bt[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = 4 * (lane + 64 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gm[c][j] = (FULL || col < D) ? gamma[col + j] : 0.f;
      bt[c][j] = (FULL || col < D) ? beta[col + j] : 0.f;
    }
  }
  struct Row { raw_t x[NCH]; };
  auto fetch = [&](long r, Row& w) {
    const T* xr = x + map_row(xmap, r) * ldx;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) w.x[c] = ld_raw_nt(reinterpret_cast<const raw_t*>(xr + col));
    }
  };
  auto finish = [&](long r, const Row& w) {
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) {
        unpack4(w.x[c], v[c]);
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
      } else {
        v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
      }
    }
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[c][j] - mu; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invD + eps);
    T* yr = y + map_row(ymap, r) * ldy;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mu) * rstd * gm[c][j] + bt[c][j];
        st4<T>(yr + col, o);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[r] = mu;
      if (rstd_out) rstd_out[r] = rstd;
    }
  };
  for (long r = (long)blockIdx.x * LN_WAVES + wave; r < rows; r += NR * total_waves) {
    Row w[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) fetch(r + i * total_waves < rows ? r + i * total_waves : r, w[i]);
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i == 0 || r + i * total_waves < rows) finish(r + i * total_waves, w[i]);
  }
}

// ---- the exact residual stream (round 6) ------------------------------------------------------------------------------
// The bf16 path stores the residual stream as bf16: every sub-block's `x + f(x)` is rounded, 36 (TimeSformer-B) to 72
// (TimeSformer-L) roundings in series, and the error of the SUM grows with sqrt(depth) -- at 24 layers the outputs deviate 3x as
// much from the fp32 reference as the reference's own autocast run, which keeps the stream in float32
// (tests/golden/make_golden_r6.py, tools/precision_study_l96.py).  Under vtx.set_stream('fp32') a sub-block hands on its
// CONTRIBUTION d = f(x) (bf16, the GEMM epilogue's output without the residual) and the stream itself lives in float32, touched
// by one kernel per sub-block -- this one, which the LayerNorm of the NEXT sub-block would have been anyway:
//     xo[omap(r)] = xs[smap(r)] + d[smap(r)]            (float32; xs == nullptr: the first sub-block, the stream starts at d)
//     y[ymap(r)]  = LayerNorm(xo row) * gamma + beta     (T = bf16; statistics of the float32 row)
// Two rows per trip as in the backward kernel.  Bytes per row: 4 D + 2 D read, 4 D + 2 D written (1.4 GB per 150 528 x 768
// launch against 0.46 GB for the bf16 LayerNorm, and the residual GEMM epilogues read and write 0.23 GB less each).
template <int NCH, bool FULL>
__global__ __launch_bounds__(LN_WAVES * 64) void ln_acc_fwd_kernel(
    int rows, int D, const float* __restrict__ xs, const bf16raw* __restrict__ d, long lds_, vtx_rowmap smap,
    float* __restrict__ xo, long ldo, vtx_rowmap omap, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    bf16raw* __restrict__ y, long ldy, vtx_rowmap ymap, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long total_waves = (long)gridDim.x * LN_WAVES;
  const float invD = 1.0f / (float)D;
  float gm[NCH][4], bt[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = 4 * (lane + 64 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gm[c][j] = (FULL || col < D) ? gamma[col + j] : 0.f;
      bt[c][j] = (FULL || col < D) ? beta[col + j] : 0.f;
    }
  }
  const bool has_xs = xs != nullptr;                   // uniform over the launch
  struct Row { float4 x[NCH]; uint2 d[NCH]; };
  auto fetch = [&](long r, Row& w) {
    const long pr = map_row(smap, r);
    const bf16raw* dr = d + pr * lds_;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) w.d[c] = ld_raw_nt(reinterpret_cast<const uint2*>(dr + col));
    }
    if (has_xs) {
      const float* xr = xs + pr * lds_;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = 4 * (lane + 64 * c);
        if (FULL || col < D) w.x[c] = ld_raw_nt(reinterpret_cast<const float4*>(xr + col));
      }
    }
  };
  auto finish = [&](long r, const Row& w) {
    float v[NCH][4];
    float s = 0.f;
    float* xor_ = xo + map_row(omap, r) * ldo;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) {
        float dv[4];
        unpack4(w.d[c], dv);
        if (has_xs) {
          unpack4(w.x[c], v[c]);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[c][j] += dv[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[c][j] = dv[j];
        }
        st4<float>(xor_ + col, v[c]);
        s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
      } else {
        v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
      }
    }
    if (y == nullptr) return;                          // accumulate only (no LayerNorm of these rows)
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float dd = v[c][j] - mu; q += dd * dd; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invD + eps);
    bf16raw* yr = y + map_row(ymap, r) * ldy;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mu) * rstd * gm[c][j] + bt[c][j];
        st4<bf16raw>(yr + col, o);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[r] = mu;
      if (rstd_out) rstd_out[r] = rstd;
    }
  };
  for (long r = (long)blockIdx.x * LN_WAVES + wave; r < rows; r += 2 * total_waves) {
    Row ra, rb;
    const bool two = r + total_waves < rows;
    fetch(r, ra);
    fetch(two ? r + total_waves : r, rb);
    finish(r, ra);
    if (two) finish(r + total_waves, rb);
  }
}

// Backward.  Each wave walks rows r = w, w + W, ...; per-lane column partials of
// dgamma/dbeta stay in registers, are combined across the block's 4 waves in
// LDS and written to part[block][2][D]; reduce_partials_kernel finishes.
// The x / dy / residual-gradient segments of the NEXT row are requested before the current row is reduced (one memory
// round trip per row instead of two -- the residual used to be fetched after the reductions -- and overlapped with the
// arithmetic): the kernel is bound by bytes in flight per wave, not by arithmetic.
// FULL: D == 256 NCH (no column predicate); RES: a residual gradient is added.  Both are compile-time so that the row loop
// is branch free -- with exec-masked loads in it hipcc falls back to vmcnt(0) waits and the prefetch is lost.
// TX: the type x is stored in -- T, or float for the bf16 kernels under the exact residual stream (vtx_layernorm_acc_fwd below keeps
// the stream in float32; gradients stay T)
// G32 (with RES): the GRADIENT of the residual stream is float32 too -- the residual gradient is read from `dres32`, the sum
// dres32 + LayerNorm-backward goes to `dx32` in float32 and, rounded ONCE, to `dx` (what the GEMMs of the sub-block before read):
// the running sum of the stream's gradient is never rounded to T (vtx_layernorm_bwd_g32)
template <typename T, int NCH, bool FULL, bool RES, typename TX = T, bool G32 = false>
__global__ __launch_bounds__(LN_WAVES * 64, (sizeof(T) == 2 && sizeof(TX) == 2 && NCH <= 3) ? 3 : 1) void ln_bwd_kernel(
    int rows, int D, const T* __restrict__ dy, long lddy, vtx_rowmap dymap,
    const TX* __restrict__ x, long ldx, vtx_rowmap xmap, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const T* __restrict__ dres,
    T* __restrict__ dx, long lddx, float* __restrict__ part, const float* __restrict__ dres32 = nullptr,
    float* __restrict__ dx32 = nullptr) {
  static_assert(!G32 || RES, "G32: a residual gradient is always present");
  typedef typename Raw4<T>::type raw_t;
  typedef typename Raw4<TX>::type rawx_t;
  typedef typename Raw4<typename std::conditional<G32, float, T>::type>::type rawr_t;
  __shared__ float red[LN_WAVES][2][NCH * 256];
  __shared__ float gsm[NCH * 256];                 // gamma: read per row from LDS instead of held in 4*NCH registers
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long total_waves = (long)gridDim.x * LN_WAVES;
  const float invD = 1.0f / (float)D;
  for (int i = threadIdx.x; i < NCH * 256; i += LN_WAVES * 64) gsm[i] = i < D ? gamma[i] : 0.f;
  __syncthreads();
  float dg[NCH][4], db[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) { dg[c][j] = 0.f; db[c][j] = 0.f; }
  struct Row { rawx_t x[NCH]; raw_t dy[NCH]; rawr_t dr[NCH]; float mu, rs; long pr; };
  auto fetch = [&](long r, Row& w) {
    w.pr = map_row(xmap, r);
    const TX* xr = x + w.pr * ldx;
    const T* dyr = dy + map_row(dymap, r) * lddy;
    const T* drr = dres + w.pr * lddx;
    w.mu = mean[r]; w.rs = rstd[r];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) {
        w.x[c] = ld_raw_nt(reinterpret_cast<const rawx_t*>(xr + col));
        w.dy[c] = ld_raw_nt(reinterpret_cast<const raw_t*>(dyr + col));
        if constexpr (G32) w.dr[c] = ld_raw_nt(reinterpret_cast<const rawr_t*>(dres32 + w.pr * lddx + col));
        else if (RES) w.dr[c] = ld_raw_nt(reinterpret_cast<const raw_t*>(drr + col));
      }
    }
  };
  // pass 1: row sums and the dgamma / dbeta partials; pass 2 recomputes xhat and g from the packed row (a few VALU ops
  // per element, the kernel is HBM-bound) instead of keeping 8 NCH floats per row alive next to the second row
  auto pass1 = [&](const Row& w, float& c1, float& c2) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) {
        float xv[4], dv[4], gm[4];
        unpack4(w.x[c], xv);
        unpack4(w.dy[c], dv);
        unpack4(*reinterpret_cast<const float4*>(gsm + col), gm);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[j] - w.mu) * w.rs, g = dv[j] * gm[j];
          s1 += g;
          s2 += g * xh;
          dg[c][j] += dv[j] * xh;
          db[c][j] += dv[j];
        }
      }
    }
    c1 = wave_sum(s1) * invD;
    c2 = wave_sum(s2) * invD;
  };
  auto pass2 = [&](const Row& w, float c1, float c2) {
    T* dxr = dx + w.pr * lddx;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = 4 * (lane + 64 * c);
      if (FULL || col < D) {
        float xv[4], dv[4], gm[4], o[4];
        unpack4(w.x[c], xv);
        unpack4(w.dy[c], dv);
        unpack4(*reinterpret_cast<const float4*>(gsm + col), gm);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[j] - w.mu) * w.rs, g = dv[j] * gm[j];
          o[j] = w.rs * (g - c1 - xh * c2);
        }
        if (RES) {
          float rv[4];
          unpack4(w.dr[c], rv);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += rv[j];
        }
        if constexpr (G32) st4<float>(dx32 + w.pr * lddx + col, o);
        st4<T>(dxr + col, o);
      }
    }
  };
  // two rows per trip: both rows' x / dy / residual segments are requested before either is reduced
  for (long r = (long)blockIdx.x * LN_WAVES + wave; r < rows; r += 2 * total_waves) {
    Row ra, rb;
    const bool two = r + total_waves < rows;
    fetch(r, ra);
    fetch(two ? r + total_waves : r, rb);
    float a1, a2, b1, b2;
    pass1(ra, a1, a2);
    if (two) pass1(rb, b1, b2);
    pass2(ra, a1, a2);
    if (two) pass2(rb, b1, b2);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[wave][0][4 * (lane + 64 * c) + j] = dg[c][j];
      red[wave][1][4 * (lane + 64 * c) + j] = db[c][j];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += LN_WAVES * 64) {
    const int which = i / D, col = i - which * D;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < LN_WAVES; ++w) a += red[w][which][col];
    part[((long)blockIdx.x * 2 + which) * D + col] = a;
  }
}

// out[n] (+)= scale * sum_s part[s*stride + n], n < N.  64 columns x 4 slab lanes per block:
// coalesced 256-B row segments, independent loads in flight, fixed summation order.
// Columns n >= split (when out2 != nullptr) go to out2[n - split] (two results, one launch); with fold > 1 every slab
// holds `fold` partial copies of those columns, fold_stride apart, which are summed too (slab-major, copy-minor order).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nslabs, long stride,
                                                              long N, float* __restrict__ out, int accumulate, float scale,
                                                              float* __restrict__ out2, long split, int accumulate2,
                                                              int fold, long fold_stride) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, sy = threadIdx.x >> 6;
  const long n = (long)blockIdx.x * 64 + cx;
  const bool tail = out2 != nullptr && n >= split;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (n < N && !(tail && fold > 1)) {
    int s = sy;
    for (; s + 12 < nslabs; s += 16) {
      a0 += part[(long)s * stride + n];
      a1 += part[(long)(s + 4) * stride + n];
      a2 += part[(long)(s + 8) * stride + n];
      a3 += part[(long)(s + 12) * stride + n];
    }
    for (; s < nslabs; s += 4) a0 += part[(long)s * stride + n];
  } else if (n < N) {
    // (slab, copy) pairs sy, sy+4, ... in slab-major order; four independent loads per round
    const int total = nslabs * fold;
    int v = sy, s = sy / fold, f = sy - (sy / fold) * fold;
    auto next = [&]() { v += 4; f += 4; while (f >= fold) { f -= fold; ++s; } };
    auto at = [&]() { return v < total ? part[(long)s * stride + (long)f * fold_stride + n] : 0.f; };
    while (v < total) {
      const float x0 = at(); next();
      const float x1 = at(); next();
      const float x2 = at(); next();
      const float x3 = at(); next();
      a0 += x0; a1 += x1; a2 += x2; a3 += x3;
    }
  }
  red[sy][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sy == 0 && n < N) {
    const float a = ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx])) * scale;
    if (tail) {
      float* o = out2 + (n - split);
      *o = accumulate2 ? *o + a : a;
    } else {
      out[n] = accumulate ? out[n] + a : a;
    }
  }
}

// The same sums, four columns per lane (16-byte loads: 1 KB per wave instruction instead of 256 B) for the first `nvec`
// columns (the weight-gradient part: no second output, no folded copies there); per column the additions are the
// ones of the kernel above in the same order, so the result is bit-identical.  Blocks beyond the vector part run the
// scalar code on columns [nvec, N).
__global__ __launch_bounds__(256) void reduce_partials_v4_kernel(const float* __restrict__ part, int nslabs, long stride,
                                                                 long N, long nvec, int vec_blocks, float* __restrict__ out,
                                                                 int accumulate, float scale, float* __restrict__ out2, long split,
                                                                 int accumulate2, int fold, long fold_stride) {
  __shared__ float red[4][64 * 4];
  const int cx = threadIdx.x & 63, sy = threadIdx.x >> 6;
  if ((int)blockIdx.x < vec_blocks) {
    const long n = ((long)blockIdx.x * 64 + cx) * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    if (n < nvec) {
      int s = sy;
      for (; s + 12 < nslabs; s += 16) {
        const float4 x0 = ld_raw_nt(reinterpret_cast<const float4*>(part + (long)s * stride + n));
        const float4 x1 = ld_raw_nt(reinterpret_cast<const float4*>(part + (long)(s + 4) * stride + n));
        const float4 x2 = ld_raw_nt(reinterpret_cast<const float4*>(part + (long)(s + 8) * stride + n));
        const float4 x3 = ld_raw_nt(reinterpret_cast<const float4*>(part + (long)(s + 12) * stride + n));
        add(a0, x0); add(a1, x1); add(a2, x2); add(a3, x3);
      }
      for (; s < nslabs; s += 4) add(a0, ld_raw_nt(reinterpret_cast<const float4*>(part + (long)s * stride + n)));
    }
    float* r = &red[sy][cx * 4];
    r[0] = (a0.x + a1.x) + (a2.x + a3.x); r[1] = (a0.y + a1.y) + (a2.y + a3.y);
    r[2] = (a0.z + a1.z) + (a2.z + a3.z); r[3] = (a0.w + a1.w) + (a2.w + a3.w);
    __syncthreads();
    if (sy == 0 && n < nvec) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v[j] = ((red[0][cx * 4 + j] + red[1][cx * 4 + j]) + (red[2][cx * 4 + j] + red[3][cx * 4 + j])) * scale;
      float4* o = reinterpret_cast<float4*>(out + n);
      if (accumulate) { const float4 c = *o; v[0] = c.x + v[0]; v[1] = c.y + v[1]; v[2] = c.z + v[2]; v[3] = c.w + v[3]; }
      *o = make_float4(v[0], v[1], v[2], v[3]);
    }
    return;
  }
  // scalar part: columns nvec .. N-1 (the bias-gradient tail and any remainder)
  const long n = nvec + (long)((int)blockIdx.x - vec_blocks) * 64 + cx;
  const bool tail = out2 != nullptr && n >= split;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (n < N && !(tail && fold > 1)) {
    int s = sy;
    for (; s + 12 < nslabs; s += 16) {
      a0 += part[(long)s * stride + n];
      a1 += part[(long)(s + 4) * stride + n];
      a2 += part[(long)(s + 8) * stride + n];
      a3 += part[(long)(s + 12) * stride + n];
    }
    for (; s < nslabs; s += 4) a0 += part[(long)s * stride + n];
  } else if (n < N) {
    const int total = nslabs * fold;
    int v = sy, s = sy / fold, f = sy - (sy / fold) * fold;
    auto next = [&]() { v += 4; f += 4; while (f >= fold) { f -= fold; ++s; } };
    auto at = [&]() { return v < total ? part[(long)s * stride + (long)f * fold_stride + n] : 0.f; };
    while (v < total) {
      const float x0 = at(); next();
      const float x1 = at(); next();
      const float x2 = at(); next();
      const float x3 = at(); next();
      a0 += x0; a1 += x1; a2 += x2; a3 += x3;
    }
  }
  red[sy][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sy == 0 && n < N) {
    const float a = ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx])) * scale;
    if (tail) {
      float* o = out2 + (n - split);
      *o = accumulate2 ? *o + a : a;
    } else {
      out[n] = accumulate ? out[n] + a : a;
    }
  }
}

// Few columns, many slabs (the LayerNorm backward: 2 D = 1536 columns of up to 1024 per-block partial rows): the kernel above would
// run 24 blocks, each lane adding 256 values one behind the other (17.5 us for 6 MB, latency-bound on 24 CUs).  Here a block is 64
// columns x 16 slab lanes: a lane adds nslabs / 16 values (four independent chains), the 16 lane sums are folded through LDS in a
// fixed order.  Deterministic; another (equally fixed) summation order than the 4-lane kernel.  No folded copies (fold == 1).
__global__ __launch_bounds__(1024) void reduce_partials_wide_kernel(const float* __restrict__ part, int nslabs, long stride,
                                                                    long N, float* __restrict__ out, int accumulate, float scale,
                                                                    float* __restrict__ out2, long split, int accumulate2) {
  __shared__ float red[16][64];
  const int cx = threadIdx.x & 63, sy = threadIdx.x >> 6;
  const long n = (long)blockIdx.x * 64 + cx;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (n < N) {
    int s = sy;
    for (; s + 48 < nslabs; s += 64) {
      a0 += part[(long)s * stride + n];
      a1 += part[(long)(s + 16) * stride + n];
      a2 += part[(long)(s + 32) * stride + n];
      a3 += part[(long)(s + 48) * stride + n];
    }
    for (; s < nslabs; s += 16) a0 += part[(long)s * stride + n];
  }
  red[sy][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sy == 0 && n < N) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) a += red[w][cx];
    a *= scale;
    if (out2 != nullptr && n >= split) {
      float* o = out2 + (n - split);
      *o = accumulate2 ? *o + a : a;
    } else {
      out[n] = accumulate ? out[n] + a : a;
    }
  }
}

int launch_reduce_partials(const float* part, int nslabs, long stride, long N, float* out,
                           int accumulate, float scale, hipStream_t st, float* out2, long split, int accumulate2,
                           int fold, long fold_stride) {
  // vector part: columns below the second output's start (all of them without one), in whole groups of four
  const long lim = out2 != nullptr ? (split < N ? split : N) : N;
  const long nvec = lim & ~3L;
  const bool vec_ok = nvec >= 4096 && stride % 4 == 0 && ((uintptr_t)part % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (vec_ok) {
    const int vec_blocks = (int)cdiv(nvec, 256);
    const int tail_blocks = (int)cdiv(N - nvec, 64);
    hipLaunchKernelGGL(reduce_partials_v4_kernel, dim3(vec_blocks + tail_blocks), dim3(256), 0, st, part, nslabs, stride, N, nvec,
                       vec_blocks, out, accumulate, scale, out2, split, accumulate2, fold, fold_stride);
    return check_launch("reduce_partials_v4");
  }
  if (fold <= 1 && nslabs >= 128 && N <= 4096) {       // few columns, many slabs: 16 slab lanes per column
    hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3(cdiv(N, 64)), dim3(1024), 0, st, part, nslabs, stride, N, out, accumulate,
                       scale, out2, split, accumulate2);
    return check_launch("reduce_partials_wide");
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(N, 64)), dim3(256), 0, st, part, nslabs,
                     stride, N, out, accumulate, scale, out2, split, accumulate2, fold, fold_stride);
  return check_launch("reduce_partials");
}

static int ln_blocks(int rows) {
  int b = cdiv(rows, LN_WAVES);
  return b > 2048 ? 2048 : (b < 1 ? 1 : b);    // 8 waves per SIMD: the kernel is bound by bytes in flight
}
// backward keeps per-block partial sums of dgamma/dbeta: fewer, fatter blocks
static int ln_bwd_blocks(int rows) {
  // 106 VGPRs -> 4 waves per SIMD -> 4 resident workgroups per CU: 1024 workgroups fill the chip in one round, and
  // every extra one only adds a [2][D] partial row for reduce_partials to read
  int b = cdiv(rows, LN_WAVES * 4);
  return b > 1024 ? 1024 : (b < 1 ? 1 : b);
}

template <typename T>
static int ln_fwd_t(int rows, int D, const void* x, long ldx, vtx_rowmap xmap, const float* gamma,
                    const float* beta, float eps, void* y, long ldy, vtx_rowmap ymap, float* mean,
                    float* rstd, hipStream_t st) {
  const int nch = cdiv(D, 256);
  dim3 g(ln_blocks(rows)), b(LN_WAVES * 64);
  if (options().ln_rows >= 2 && nch <= 4) {            // NR rows per trip, all requested before the first is reduced
    const int nr = options().ln_rows;
    dim3 g2(ln_blocks((rows + nr - 1) / nr));
#define LN_FWD2_(N, R)                                                                             \
    { if (D == N * 256) hipLaunchKernelGGL((ln_fwd2_kernel<T, N, true, R>), g2, b, 0, st, rows, D, (const T*)x, ldx, xmap, gamma, beta, eps, (T*)y, ldy, ymap, mean, rstd); \
      else hipLaunchKernelGGL((ln_fwd2_kernel<T, N, false, R>), g2, b, 0, st, rows, D, (const T*)x, ldx, xmap, gamma, beta, eps, (T*)y, ldy, ymap, mean, rstd); }
#define LN_FWD2(N) { if (nr == 2) LN_FWD2_(N, 2) else if (nr == 3) LN_FWD2_(N, 3) else LN_FWD2_(N, 4) }
    if (nch == 1) LN_FWD2(1) else if (nch == 2) LN_FWD2(2) else if (nch == 3) LN_FWD2(3) else LN_FWD2(4)
#undef LN_FWD2
#undef LN_FWD2_
    return check_launch("layernorm_fwd2");
  }
#define LN_FWD(N)                                                                                  \
  hipLaunchKernelGGL((ln_fwd_kernel<T, N>), g, b, 0, st, rows, D, (const T*)x, ldx, xmap, gamma,   \
                     beta, eps, (T*)y, ldy, ymap, mean, rstd)
  switch (nch) {
    case 1: LN_FWD(1); break;
    case 2: LN_FWD(2); break;
    case 3: LN_FWD(3); break;
    case 4: LN_FWD(4); break;
    case 5: case 6: LN_FWD(6); break;
    default: LN_FWD(8); break;
  }
#undef LN_FWD
  return check_launch("layernorm_fwd");
}

template <typename T, typename TX = T>
static int ln_bwd_t(int rows, int D, const void* dy, long lddy, vtx_rowmap dymap, const void* x,
                    long ldx, vtx_rowmap xmap, const float* mean, const float* rstd,
                    const float* gamma, const void* dres, void* dx, long lddx, float* part,
                    int nblocks, int* launched, hipStream_t st, const float* dres32 = nullptr, float* dx32 = nullptr) {
  const int nch = cdiv(D, 256);
  dim3 b(LN_WAVES * 64);
  // one resident round: as many workgroups as the instantiation's occupancy holds (never more than the workspace rows)
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
#define LN_BWD_(N, F, R)                                                                           \
  {                                                                                                \
    static int per_cu = 0;                                                                         \
    if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ln_bwd_kernel<T, N, F, R, TX>, LN_WAVES * 64, 0) \
                        != hipSuccess || per_cu <= 0)) per_cu = 2;                                 \
    dim3 g(nblocks < per_cu * n_cu ? nblocks : per_cu * n_cu);                                     \
    hipLaunchKernelGGL((ln_bwd_kernel<T, N, F, R, TX>), g, b, 0, st, rows, D, (const T*)dy, lddy, dymap, \
                       (const TX*)x, ldx, xmap, mean, rstd, gamma, (const T*)dres, (T*)dx, lddx, part); \
    *launched = (int)g.x;                                                                          \
  }
#define LN_BWD_G_(N, F)                                                                            \
  {                                                                                                \
    static int per_cu = 0;                                                                         \
    if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ln_bwd_kernel<T, N, F, true, TX, true>, LN_WAVES * 64, 0) \
                        != hipSuccess || per_cu <= 0)) per_cu = 1;                                 \
    dim3 g(nblocks < per_cu * n_cu ? nblocks : per_cu * n_cu);                                     \
    hipLaunchKernelGGL((ln_bwd_kernel<T, N, F, true, TX, true>), g, b, 0, st, rows, D, (const T*)dy, lddy, dymap, \
                       (const TX*)x, ldx, xmap, mean, rstd, gamma, (const T*)nullptr, (T*)dx, lddx, part, dres32, dx32); \
    *launched = (int)g.x;                                                                          \
  }
  if constexpr (sizeof(T) == 2 && sizeof(TX) == 4) {
    if (dres32) {                                  // float32 gradient stream (vtx_layernorm_bwd_g32): D <= 1024 like the forward kernel
#define LN_BWD_G(N) { if (D == N * 256) LN_BWD_G_(N, true) else LN_BWD_G_(N, false) }
      if (nch == 1) LN_BWD_G(1) else if (nch == 2) LN_BWD_G(2) else if (nch == 3) LN_BWD_G(3) else LN_BWD_G(4)
#undef LN_BWD_G
      return check_launch("layernorm_bwd_g32");
    }
  }
#undef LN_BWD_G_
#define LN_BWD(N)                                                                                  \
  if (D == N * 256) { if (dres) LN_BWD_(N, true, true) else LN_BWD_(N, true, false) }              \
  else { if (dres) LN_BWD_(N, false, true) else LN_BWD_(N, false, false) }
  switch (nch) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    case 4: LN_BWD(4); break;
    case 5: case 6: LN_BWD(6); break;
    default: LN_BWD(8); break;
  }
#undef LN_BWD
#undef LN_BWD_
  return check_launch("layernorm_bwd");
}

}  // namespace vtx

using namespace vtx;

extern "C" int vtx_layernorm_fwd(int dtype, int rows, int D, const void* x, long ldx,
                                 vtx_rowmap xmap, const float* gamma, const float* beta, float eps,
                                 void* y, long ldy, vtx_rowmap ymap, float* mean, float* rstd,
                                 void* stream) {
  VTX_REQUIRE(rows >= 0 && D > 0 && D % 4 == 0 && D <= 2048, VTX_EINVAL,
              "layernorm_fwd: D=%d must be a multiple of 4 and <= 2048", D);
  if (rows == 0) return VTX_OK;
  VTX_REQUIRE(x && y && gamma && beta, VTX_EINVAL, "layernorm_fwd: null pointer");
  VTX_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) && ldx % 4 == 0 &&
                  ldy % 4 == 0, VTX_EALIGN, "layernorm_fwd: 16-byte alignment required");
  if (dtype == VTX_F32)
    return ln_fwd_t<float>(rows, D, x, ldx, xmap, gamma, beta, eps, y, ldy, ymap, mean, rstd, as_stream(stream));
  if (dtype == VTX_BF16)
    return ln_fwd_t<bf16raw>(rows, D, x, ldx, xmap, gamma, beta, eps, y, ldy, ymap, mean, rstd, as_stream(stream));
  VTX_REQUIRE(false, VTX_EINVAL, "layernorm_fwd: bad dtype %d", dtype);
}

extern "C" int vtx_layernorm_acc_fwd(int rows, int D, const float* xs, const void* d, long lds, vtx_rowmap smap, float* xo, long ldo,
                                    vtx_rowmap omap, const float* gamma, const float* beta, float eps, void* y, long ldy,
                                    vtx_rowmap ymap, float* mean, float* rstd, void* stream) {
  VTX_REQUIRE(rows >= 0 && D > 0 && D % 4 == 0 && D <= 1024, VTX_EINVAL, "layernorm_acc_fwd: D=%d must be a multiple of 4 and <= 1024", D);
  if (rows == 0) return VTX_OK;
  VTX_REQUIRE(d && xo && (y == nullptr || (gamma && beta)), VTX_EINVAL, "layernorm_acc_fwd: null pointer");
  VTX_REQUIRE(aligned16(d) && aligned16(xo) && (!xs || aligned16(xs)) && (!y || (aligned16(y) && aligned16(gamma) && aligned16(beta))) &&
                  lds % 4 == 0 && ldo % 4 == 0 && ldy % 4 == 0, VTX_EALIGN, "layernorm_acc_fwd: 16-byte alignment required");
  if (y == nullptr) { gamma = xo; beta = xo; }       // never applied; the kernel loads its register copies from valid memory
  const int nch = cdiv(D, 256);
  dim3 g(ln_blocks((rows + 1) / 2)), b(LN_WAVES * 64);
  hipStream_t st = as_stream(stream);
#define LN_ACC(N)                                                                                                         \
  { if (D == N * 256) hipLaunchKernelGGL((ln_acc_fwd_kernel<N, true>), g, b, 0, st, rows, D, xs, (const bf16raw*)d, lds, smap, xo, ldo, omap, gamma, beta, eps, (bf16raw*)y, ldy, ymap, mean, rstd); \
    else hipLaunchKernelGGL((ln_acc_fwd_kernel<N, false>), g, b, 0, st, rows, D, xs, (const bf16raw*)d, lds, smap, xo, ldo, omap, gamma, beta, eps, (bf16raw*)y, ldy, ymap, mean, rstd); }
  if (nch == 1) LN_ACC(1) else if (nch == 2) LN_ACC(2) else if (nch == 3) LN_ACC(3) else LN_ACC(4)
#undef LN_ACC
  return check_launch("layernorm_acc_fwd");
}

extern "C" size_t vtx_layernorm_bwd_workspace(int rows, int D) {
  return (size_t)ln_bwd_blocks(rows) * 2 * (size_t)D * sizeof(float);
}

extern "C" int vtx_layernorm_bwd(int dtype, int rows, int D, const void* dy, long lddy,
                                 vtx_rowmap dymap, const void* x, long ldx, vtx_rowmap xmap,
                                 const float* mean, const float* rstd, const float* gamma,
                                 const void* dres, void* dx, long lddx, float* dgamma, float* dbeta,
                                 void* workspace, size_t ws_bytes, void* stream) {
  VTX_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 2048, VTX_EINVAL, "layernorm_bwd: bad shape");
  VTX_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && workspace, VTX_EINVAL,
              "layernorm_bwd: null pointer");
  VTX_REQUIRE(ws_bytes >= vtx_layernorm_bwd_workspace(rows, D), VTX_EWS, "layernorm_bwd: workspace too small");
  VTX_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dx) && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0,
              VTX_EALIGN, "layernorm_bwd: 16-byte alignment required");
  const int nb = ln_bwd_blocks(rows);
  float* part = (float*)workspace;
  hipStream_t st = as_stream(stream);
  int rc, launched = nb;
  if (dtype == VTX_F32)
    rc = ln_bwd_t<float>(rows, D, dy, lddy, dymap, x, ldx, xmap, mean, rstd, gamma, dres, dx, lddx, part, nb, &launched, st);
  else if (dtype == VTX_BF16)
    rc = ln_bwd_t<bf16raw>(rows, D, dy, lddy, dymap, x, ldx, xmap, mean, rstd, gamma, dres, dx, lddx, part, nb, &launched, st);
  else if (dtype == VTX_BF16_X32)                    // gradients bf16, x float32 (the exact residual stream)
    rc = ln_bwd_t<bf16raw, float>(rows, D, dy, lddy, dymap, x, ldx, xmap, mean, rstd, gamma, dres, dx, lddx, part, nb, &launched, st);
  else
    VTX_REQUIRE(false, VTX_EINVAL, "layernorm_bwd: bad dtype %d", dtype);
  if (rc) return rc;
  // part layout: [block][2][D] -> dgamma += sum_b part[b][0], dbeta += sum_b part[b][1] (one launch)
  return launch_reduce_partials(part, launched, 2L * D, 2L * D, dgamma, 1, 1.0f, st, dbeta, D, 1);
}

// The float32 GRADIENT stream (bf16 kernels, exact residual stream): dx32 = dres32 + LayerNorm-backward(dy) on the mapped rows in
// float32, dx = bf16(dx32).  x is the float32 stream vtx_layernorm_acc_fwd stored; dres32 / dx32 / dx share xmap and lddx.
extern "C" int vtx_layernorm_bwd_g32(int rows, int D, const void* dy, long lddy, vtx_rowmap dymap, const float* x, long ldx,
                                     vtx_rowmap xmap, const float* mean, const float* rstd, const float* gamma,
                                     const float* dres32, float* dx32, void* dx, long lddx, float* dgamma, float* dbeta,
                                     void* workspace, size_t ws_bytes, void* stream) {
  VTX_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 1024, VTX_EINVAL, "layernorm_bwd_g32: D=%d must be a multiple of 4 and <= 1024", D);
  VTX_REQUIRE(dy && x && mean && rstd && gamma && dres32 && dx32 && dx && dgamma && dbeta && workspace, VTX_EINVAL,
              "layernorm_bwd_g32: null pointer");
  VTX_REQUIRE(ws_bytes >= vtx_layernorm_bwd_workspace(rows, D), VTX_EWS, "layernorm_bwd_g32: workspace too small");
  VTX_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(dres32) && aligned16(dx32) && lddy % 4 == 0 && ldx % 4 == 0 &&
                  lddx % 4 == 0, VTX_EALIGN, "layernorm_bwd_g32: 16-byte alignment required");
  const int nb = ln_bwd_blocks(rows);
  float* part = (float*)workspace;
  hipStream_t st = as_stream(stream);
  int launched = nb;
  const int rc = ln_bwd_t<bf16raw, float>(rows, D, dy, lddy, dymap, x, ldx, xmap, mean, rstd, gamma, nullptr, dx, lddx, part, nb, &launched, st,
                                          dres32, dx32);
  if (rc) return rc;
  return launch_reduce_partials(part, launched, 2L * D, 2L * D, dgamma, 1, 1.0f, st, dbeta, D, 1);
}
