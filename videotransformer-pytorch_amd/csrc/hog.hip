// hog.hip -- MaskFeat HOG target extractor and the MaskFeat head's small kernels.
//
// HOG: reference dataset.py:39-45 -> skimage.feature.hog(orientations=9,
// pixels_per_cell=(8,8), cells_per_block=(1,1), block_norm='L2') per channel.
// Bit-exact contract (see oracle/hog_oracle.py for the derivation):
//   * gradients are integer central differences (0 on the image border);
//   * the orientation bin is decided without atan2 and without floating point: skimage's interval test equals the eight
//     sign tests  g_row cos(20k) - g_col sin(20k) >= 0  (proven for all 261 121 integer gradient pairs, oracle tests), and
//     those equal  |g_col| <= floor(g_row cot(20j))  for j = 1..4 with the sign of g_col choosing the half -- four
//     fixed-point products (g cot is never within 2.6e-4 of an integer for g <= 255: 16 fraction bits decide it);
//     vtx_selftest checks the integer rule against the double rule for every pair ON THE DEVICE;
//   * magnitude = libm hypot(g_col, g_row) of the HOST (what numpy calls): for the 65 536 integer pairs it differs from
//     the correctly rounded sqrt(g_row^2 + g_col^2) in 300 entries by one ulp (glibc 2.35), so the kernel computes the
//     correctly rounded square root (device sqrt) and adds a 2-bit correction looked up in a 16-KB table that
//     vtx_hog_build_table derives from the host's hypot (round 4 gathered the 512-KB double table per pixel: the
//     vector L1 fills, not the arithmetic, bounded it); vtx_selftest compares all 65 536 magnitudes with the host table;
//   * per (cell, bin): the magnitudes of that bin's pixels are accumulated in row-major order into a float32
//     accumulator with double adds, divided by 64 in float32 (pixels with magnitude 0 leave the accumulator unchanged);
//   * per cell: L2 norm in float64 with numpy's pairwise order for 9 terms.
// HBM-bound by bytes: 150 528 B read + 169 344 B written per 224x224 frame.
//
// Round 5 kernel.  One persistent workgroup per (frame, cell row) item for ALL THREE channels: the ten pixel rows of a
// strip are one contiguous 30 W-byte block of the interleaved RGB frame, fetched with 16-byte loads (every byte once per
// strip; round 4: one workgroup per channel, stride-3 byte loads, each row three times).  Per channel: (A) a thread per
// pixel column walks the eight rows -- gradient, integer bin, magnitude -> LDS, and ONE LDS atomic OR that sets the
// pixel's bit in the 64-bit mask of its (cell, bin); (B) a thread per (cell, bin) walks only the SET bits of its mask in
// ascending (= row-major) order -- on average 7 double adds instead of 64 compare-and-maybe-add steps; then (C) the
// norms, and the 27 features of a cell leave as 16-byte stores (two horizontally adjacent cells = 432 contiguous bytes).
#include <math.h>
#include <string.h>
#include <vector>
#include "common.h"

namespace vtx {

constexpr int HOG_EXC_WORDS = 4096;          // 65 536 gradient pairs x 2 bits: 0 = hypot is the rounded sqrt, 1 = one ulp above, 3 = one below

__constant__ double HOG_BC[8] = {0.93969262078590838, 0.76604444311897804, 0.5, 0.17364817766693035,
                                 -0.17364817766693035, -0.5, -0.76604444311897804, -0.93969262078590838};
__constant__ double HOG_BS[8] = {0.34202014332566873, 0.64278760968653933, 0.8660254037844386, 0.98480775301220806,
                                 0.98480775301220806, 0.8660254037844386, 0.64278760968653933, 0.34202014332566873};

// the eight sign tests in float64 (round 1 - 4 kernel; kept as the device-side cross-check of the integer rule)
__device__ inline int hog_bin_f64(int gr, int gc) {
  if (gr == 0 && gc == 0) return 0;
  if (gr < 0 || (gr == 0 && gc < 0)) { gr = -gr; gc = -gc; }
  int b = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) b += (HOG_BC[k] * (double)gr - HOG_BS[k] * (double)gc >= 0.0) ? 1 : 0;
  return b;
}

// integer rule: with (g, c) = (g_row, g_col) flipped into the upper half plane, the orientation is at least 20 j degrees
// iff |c| <= floor(g cot(20 j)) on the c >= 0 side (bins 0..4), and the mirror image on the c < 0 side (bins 4..8).
// floor(g cot) = (g * round(cot * 2^16)) >> 16 for every g in 0..255 (checked offline and by vtx_selftest).
__device__ inline int hog_bin(int gr, int gc) {
  const bool flip = gr < 0 || (gr == 0 && gc < 0);
  const int g = flip ? -gr : gr, c = flip ? -gc : gc;
  const int a = c < 0 ? -c : c;
  // (g <= 255 and the constants are below 2^18: the 24-bit multiply is exact and full rate, v_mul_lo_u32 is quarter rate)
  const int cnt = (a <= (__mul24(g, 180059) >> 16) ? 1 : 0) + (a <= (__mul24(g, 78103) >> 16) ? 1 : 0) +
                  (a <= (__mul24(g, 37837) >> 16) ? 1 : 0) + (a <= (__mul24(g, 11556) >> 16) ? 1 : 0);
  const int b = c >= 0 ? cnt : 8 - cnt;
  return (gr | gc) == 0 ? 0 : b;
}

// host hypot(ac, ar) for 0 <= ar, ac <= 255: correctly rounded sqrt of the exact integer n = ar^2 + ac^2 <= 130 050, moved by
// the table's correction (`word` = exc[(ar * 256 + ac) >> 4]).  The square root: n is exact in float32, s = v_sqrt_f32(n) is
// good to ~2^-23, h = 0.5 / s from v_rcp_f32; two Newton corrections in float64 -- r = fma(-m, m, n) is the exact residual of
// the 24-bit s in the first, m += r * h -- leave an error of ~2^-69 m, far below the half-ulp of the result (the generic
// float64 lowering costs twice the float64 instructions: range scaling, v_rsq_f64, three iterations).  "Far below" is not a
// proof of correct rounding: vtx_selftest compares ALL 65 536 magnitudes with the host table on the device at hand.
__device__ inline double hog_mag(int ar, int ac, unsigned word) {
  const int idx = ar * 256 + ac;
  const int n = __mul24(ar, ar) + __mul24(ac, ac);
#if defined(HOG_ABLATE) && HOG_ABLATE == 1           // diagnostic builds (csrc/build.py --variant): no square root
  double m = (double)n;
#elif defined(HOG_ABLATE) && HOG_ABLATE == 6         // the compiler's float64 square root
  double m = sqrt((double)n);
#else
  const float nf = (float)n;
  const float s = __builtin_amdgcn_sqrtf(nf);
  const double x = (double)nf;
  double m = (double)s;
  const double h = (double)(0.5f * __builtin_amdgcn_rcpf(s));
  m = fma(fma(-m, m, x), h, m);
  m = fma(fma(-m, m, x), h, m);
  m = n == 0 ? 0.0 : m;                              // rcp(0) = inf
#endif
#if defined(HOG_ABLATE) && HOG_ABLATE == 2           // no correction
  return m + (double)(word & 0u);
#else
  const int delta = ((int)(word << (30 - 2 * (idx & 15)))) >> 30;          // 2-bit field, sign-extended: 0, +1, -1
  // one ulp up or down = the low word +- 1: no square root of these integers has a low word of all zeros or all ones next to a
  // correction (vtx_selftest would show it), so the carry into the high word is never needed
  const long long b = __double_as_longlong(m);
  return __longlong_as_double((b & ~0xffffffffLL) | (unsigned)((int)b + delta));
#endif
}

// frames [F,H,W,3] u8; grid = persistent workgroups over the F * (H/8) strips; block = 256 * CP threads.
// CP = channels in parallel: 3 (W <= 256: threads 256 c .. 256 c + 255 work on channel c, every phase once per strip, 12 waves per
// workgroup and two workgroups per CU) or 1 (wider frames: the three channels one after the other through one magnitude tile).
template <int CP>
__global__ __launch_bounds__(256 * CP, CP == 3 ? 6 : 1) void hog_kernel(const uint8_t* __restrict__ frames, int F, int H, int W,
                                                       const uint32_t* __restrict__ exc, double* __restrict__ out,
                                                       int32_t* __restrict__ bins) {
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  constexpr int T = 256 * CP;
  const int nc = W / 8, W3 = 3 * W;
  uint32_t* excs = reinterpret_cast<uint32_t*>(hsm);                                    // [4096]
  uint8_t* px = reinterpret_cast<uint8_t*>(hsm + HOG_EXC_WORDS * 4);                    // [10][3 W]  (30 W bytes, W % 16 == 0)
  double* mag_all = reinterpret_cast<double*>(px + 10 * W3);                            // [CP][nc][8 rows][8 columns]
  uint32_t* msk_all = reinterpret_cast<uint32_t*>(mag_all + CP * 8 * W);                // [CP][nc][9][2]
  double* hist = reinterpret_cast<double*>(msk_all + CP * nc * 18);                     // [nc][27]
  const int tid_all = threadIdx.x;
  const int grp = CP == 1 ? 0 : tid_all >> 8, tid = CP == 1 ? tid_all : tid_all & 255;   // channel group of this thread
  double* mag = mag_all + grp * 8 * W;
  uint32_t* msk = msk_all + grp * nc * 18;
  for (int i = tid_all; i < HOG_EXC_WORDS; i += T) excs[i] = exc[i];
  const int strips = H / 8;
  const long items = (long)F * strips;
  for (long item = blockIdx.x; item < items; item += gridDim.x) {
    const int f = (int)(item / strips), cr = (int)(item - (long)f * strips);
    const int y0 = cr * 8;
    const uint8_t* img = frames + (long)f * H * W3;
    __syncthreads();                                 // the previous strip's feature stores have read hist; excs is in place
    // rows y0-1 .. y0+8: one contiguous block of the frame, 16 bytes per load; rows outside the frame read as 0
    {
      const long first = (long)(y0 - 1) * W3, last = (long)H * W3;
      for (int v = tid_all; v < (10 * W3) / 16; v += T) {
        const long g0 = first + (long)v * 16;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (g0 >= 0 && g0 < last) val = *reinterpret_cast<const uint4*>(img + g0);
        *reinterpret_cast<uint4*>(px + v * 16) = val;
      }
    }
#pragma unroll 1
    for (int chi = 0; chi < 3 / CP; ++chi) {
      const int ch = CP == 1 ? chi : grp;
      for (int i = tid; i < nc * 18; i += 256) msk[i] = 0u;
      __syncthreads();                               // pixels in place / masks cleared; phase B of the previous channel is done
      // (A) one pixel column per thread, eight rows
      for (int x = tid; x < W; x += 256) {
        const bool xin = x > 0 && x < W - 1;
        const int xo = 3 * x + ch;
#pragma unroll
        for (int ry = 0; ry < 8; ++ry) {
          const int y = y0 + ry;
          int gr = 0, gc = 0;
          if (y > 0 && y < H - 1) gr = (int)px[(ry + 2) * W3 + xo] - (int)px[ry * W3 + xo];
          if (xin) gc = (int)px[(ry + 1) * W3 + xo + 3] - (int)px[(ry + 1) * W3 + xo - 3];
#if defined(HOG_ABLATE) && HOG_ABLATE == 4           // no bin rule
          const int b = (gr + gc) & 7;
#else
          const int b = hog_bin(gr, gc);
#endif
          const int ar = gr < 0 ? -gr : gr, ac = gc < 0 ? -gc : gc;
#if defined(HOG_ABLATE) && HOG_ABLATE == 5           // no magnitude at all
          const double m = (double)(ar + ac);
#else
          const double m = hog_mag(ar, ac, excs[(ar * 256 + ac) >> 4]);
#endif
          mag[(x >> 3) * 64 + ry * 8 + (x & 7)] = m;      // cell-major: phase B addresses a pixel by its bit position alone
          if ((gr | gc) != 0) atomicOr(&msk[((x >> 3) * 9 + b) * 2 + (ry >> 2)], 1u << ((ry & 3) * 8 + (x & 7)));
          if (bins) bins[(((long)f * 3 + ch) * H + y) * W + x] = b;
        }
      }
      __syncthreads();
      // (B) one (cell, bin) per thread: the set bits of its mask in ascending order = the cell's pixels of that bin, row-major
#if defined(HOG_ABLATE) && HOG_ABLATE == 3           // no phase B
      for (int i = tid; i < nc * 9; i += 256) hist[(i / 9) * 27 + ch * 9 + i % 9] = mag[i];     // (diagnostic)
      if (false)
#endif
      for (int i = tid; i < nc * 9; i += 256) {
        const int cc = i / 9, ob = i - cc * 9;
        uint32_t m0 = msk[i * 2], m1 = msk[i * 2 + 1];
        const double* mg = mag + cc * 64;
        float tot = 0.0f;
        // four set bits per trip: the four magnitudes are requested together, the adds stay sequential and in ascending bit
        // order (= row-major in the cell); a bit position beyond the mask's population reads pixel 0 and is not added
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          uint32_t m = half ? m1 : m0;
          const double* mh = mg + half * 32;
          while (m) {
            int p[4];
            bool v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[k] = m != 0;
              p[k] = v[k] ? __builtin_ctz(m) : 0;
              m &= m - 1;
            }
            double a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = mh[p[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (v[k]) tot = (float)((double)tot + a[k]);
          }
        }
        hist[cc * 27 + ch * 9 + ob] = (double)(tot / 64.0f);
      }
      __syncthreads();
    }
    // (C) L2 norm per (cell, channel), in place
    for (int i = tid_all; i < nc * 3; i += T) {
      double* h = hist + (i / 3) * 27 + (i % 3) * 9;
      double s = ((h[0] * h[0] + h[1] * h[1]) + (h[2] * h[2] + h[3] * h[3])) +
                 ((h[4] * h[4] + h[5] * h[5]) + (h[6] * h[6] + h[7] * h[7]));
      s += h[8] * h[8];
      const double nrm = sqrt(s + 1e-5 * 1e-5);
#pragma unroll
      for (int k = 0; k < 9; ++k) h[k] = h[k] / nrm;
    }
    __syncthreads();
    // features: cells 2 pw and 2 pw + 1 of this strip are 54 contiguous doubles of out (dw = 0, 1 at dh = cr & 1)
    {
      const int ph = cr >> 1, dh = cr & 1;
      double* orow = out + (((long)f * (H / 16) + ph) * (W / 16)) * 108 + dh * 54;
      for (int i = tid_all; i < (nc / 2) * 27; i += T) {
        const int pw = i / 27, j = i - pw * 27;
        *reinterpret_cast<double2*>(orow + (long)pw * 108 + 2 * j) = *reinterpret_cast<const double2*>(hist + pw * 54 + 2 * j);
      }
    }
  }
}

// vtx_selftest: every gradient pair through the device functions above -- magnitudes [256*256] and both bin rules [511*511]
__global__ void hog_probe_kernel(const uint32_t* __restrict__ exc, double* __restrict__ mags, int* __restrict__ bin_int,
                                 int* __restrict__ bin_f64) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 65536) mags[i] = hog_mag(i >> 8, i & 255, exc[i >> 4]);
  if (i < 511 * 511) {
    const int gr = i / 511 - 255, gc = i % 511 - 255;
    bin_int[i] = hog_bin(gr, gc);
    bin_f64[i] = hog_bin_f64(gr, gc);
  }
}

// ---- MaskFeat mask-token blend (video_transformer.py:914-919) ------------------------
template <typename T>
__global__ void mf_blend_fwd_kernel(long rows, int Tq, int Hq, int Wq, int C, int g, const T* __restrict__ x,
                                    const uint8_t* __restrict__ mask, const float* __restrict__ tok, T* __restrict__ out) {
  const long per = C / 8;
  const int r = Hq / g;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < rows * per; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / per;
    const int c = (int)(idx - row * per) * 8;
    long q = row;
    const int w = (int)(q % Wq); q /= Wq;
    const int h = (int)(q % Hq); q /= Hq;
    const int t = (int)(q % Tq); q /= Tq;
    const uint8_t mk = mask[((q * Tq + t) * g + h / r) * g + w / r];
    float v[8];
    if (mk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tok[c + j];
    } else {
      load8(x + row * C + c, v);
    }
    store8(out + row * C + c, v);
  }
}

template <typename T>
__global__ void mf_blend_bwd_kernel(long rows, int Tq, int Hq, int Wq, int C, int g, const T* __restrict__ dy,
                                    const uint8_t* __restrict__ mask, T* __restrict__ dx, float* __restrict__ dtok) {
  // grid.x covers column chunks of 8, grid.y strides rows; per-block partial of dtoken via atomics (C is tiny: 96)
  const int c = blockIdx.x * 8;
  const int r = Hq / g;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long row = (long)blockIdx.y * blockDim.x + threadIdx.x; row < rows; row += (long)gridDim.y * blockDim.x) {
    long q = row;
    const int w = (int)(q % Wq); q /= Wq;
    const int h = (int)(q % Hq); q /= Hq;
    const int t = (int)(q % Tq); q /= Tq;
    const uint8_t mk = mask[((q * Tq + t) * g + h / r) * g + w / r];
    float v[8];
    load8(dy + row * C + c, v);
    if (mk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc[j] += v[j]; v[j] = 0.f; }
    }
    store8(dx + row * C + c, v);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float s = wave_sum(acc[j]);
    if ((threadIdx.x & 63) == 0 && s != 0.f) atomicAdd(dtok + c + j, s);
  }
}

// ---- masked MSE (video_transformer.py:882-901) ----------------------------------------
// pred row (b, tq, h, w) holds ts*Cf features: frame tq*ts+dt uses features [dt*Cf, (dt+1)*Cf).
// One wave per (b, frame, h, w) cell of the target grid; skips cells with cmask == 0.
template <typename T>
__global__ __launch_bounds__(256) void mf_loss_fwd_kernel(long cells, int Tq, int ts, int g, int Cf, const T* __restrict__ pred,
                                                          long ldp, const double* __restrict__ target,
                                                          const uint8_t* __restrict__ cmask, double* __restrict__ acc2) {
  const int lane = threadIdx.x & 63;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  double lsum = 0.0, msum = 0.0;
  for (long cell = wave0; cell < cells; cell += nwaves) {
    if (!cmask[cell]) continue;
    long q = cell;
    const int w = (int)(q % g); q /= g;
    const int h = (int)(q % g); q /= g;
    const int fr = (int)(q % (Tq * ts)); q /= (Tq * ts);
    const int tq = fr / ts, dt = fr - tq * ts;
    const T* pr = pred + (((q * Tq + tq) * g + h) * g + w) * ldp + dt * Cf;
    const double* tg = target + cell * Cf;
    double e = 0.0;
    for (int k = lane; k < Cf; k += 64) {
      const double d = (double)ET<T>::ld(pr + k) - tg[k];
      e += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
    lsum += e / (double)Cf;
    msum += 1.0;
  }
  if (lane == 0 && msum != 0.0) {
    atomicAdd(acc2, lsum);
    atomicAdd(acc2 + 1, msum);
  }
}
// The reference computes `mask.sum() + 1e-5` on an int32 mask: the integer sum is promoted to
// float32 (video_transformer.py:901), so the denominator is float32(sum + 1e-5).
__device__ inline double mf_denominator(double msum) { return (double)((float)msum + 1e-5f); }
__global__ void mf_loss_finish_kernel(double* acc2) { acc2[0] = acc2[0] / mf_denominator(acc2[1]); }

template <typename T>
__global__ __launch_bounds__(256) void mf_loss_bwd_kernel(long rows, int Tq, int ts, int g, int Cf, const T* __restrict__ pred,
                                                          long ldp, const double* __restrict__ target,
                                                          const uint8_t* __restrict__ cmask, const double* __restrict__ acc2,
                                                          float gloss, T* __restrict__ dpred, long lddp) {
  // one thread per element of pred [rows, ts*Cf]
  const long total = rows * ts * Cf;
  const double coef = (double)gloss * 2.0 / ((double)Cf * mf_denominator(acc2[1]));
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / (ts * Cf);
    const int col = (int)(idx - row * (ts * Cf));
    const int dt = col / Cf, k = col - dt * Cf;
    long q = row;
    const int w = (int)(q % g); q /= g;
    const int h = (int)(q % g); q /= g;
    const int tq = (int)(q % Tq); q /= Tq;
    const long cell = ((q * (Tq * ts) + tq * ts + dt) * g + h) * g + w;
    float gv = 0.f;
    if (cmask[cell]) gv = (float)(coef * ((double)ET<T>::ld(pred + row * ldp + col) - target[cell * Cf + k]));
    ET<T>::st(dpred + row * lddp + col, gv);
  }
}

}  // namespace vtx

using namespace vtx;

extern "C" size_t vtx_hog_table_bytes(void) { return (size_t)256 * 256 * sizeof(double) + (size_t)HOG_EXC_WORDS * sizeof(uint32_t); }

// [65 536 doubles: the host's hypot(c, r)] [4096 words: per pair, 2 bits = hypot minus the correctly rounded sqrt(r^2 + c^2) in ulps]
extern "C" int vtx_hog_build_table(double* host_table) {
  VTX_REQUIRE(host_table != nullptr, VTX_EINVAL, "hog_build_table: null pointer");
  uint32_t* exc = reinterpret_cast<uint32_t*>(host_table + 256 * 256);
  for (int w = 0; w < HOG_EXC_WORDS; ++w) exc[w] = 0u;
  for (int r = 0; r < 256; ++r)
    for (int c = 0; c < 256; ++c) {
      const double h = hypot((double)c, (double)r);
      const double q = sqrt((double)(r * r + c * c));          // the integer is exact, the host square root correctly rounded
      host_table[r * 256 + c] = h;
      long long hb, qb;
      memcpy(&hb, &h, 8);
      memcpy(&qb, &q, 8);
      const long long d = hb - qb;
      VTX_REQUIRE(d >= -1 && d <= 1, VTX_EINVAL, "hog_build_table: hypot(%d, %d) is %lld ulps from the rounded square root", c, r, d);
      const int idx = r * 256 + c;
      exc[idx >> 4] |= (uint32_t)(d & 3) << (2 * (idx & 15));
    }
  return VTX_OK;
}

namespace vtx {
// part of vtx_selftest (api.hip): 0 = the device reproduces the host's hypot for all 65 536 gradient pairs and the integer
// bin rule equals the float64 sign tests for all 511 x 511 pairs
int hog_selftest(int* bad_mag, int* bad_bin) {
  // (defined counts on every path -- 0 mismatches can only come from a completed comparison --, every device buffer
  // freed on every path, every runtime call checked)
  *bad_mag = 65536; *bad_bin = 511 * 511;
  std::vector<double> tab(vtx_hog_table_bytes() / 8);
  if (vtx_hog_build_table(tab.data()) != VTX_OK) return -1;
  uint32_t* d_exc = nullptr; double* d_mag = nullptr; int *d_bi = nullptr, *d_bf = nullptr;
  const int NB = 511 * 511;
  std::vector<double> mags(65536);
  std::vector<int> bi(NB), bf(NB);
  bool ok = hipMalloc(&d_exc, HOG_EXC_WORDS * 4) == hipSuccess && hipMalloc(&d_mag, 65536 * 8) == hipSuccess &&
            hipMalloc(&d_bi, NB * 4) == hipSuccess && hipMalloc(&d_bf, NB * 4) == hipSuccess;
  ok = ok && hipMemcpy(d_exc, tab.data() + 65536, HOG_EXC_WORDS * 4, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(hog_probe_kernel, dim3((NB + 255) / 256), dim3(256), 0, 0, d_exc, d_mag, d_bi, d_bf);
    ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  }
  ok = ok && hipMemcpy(mags.data(), d_mag, 65536 * 8, hipMemcpyDeviceToHost) == hipSuccess &&
       hipMemcpy(bi.data(), d_bi, NB * 4, hipMemcpyDeviceToHost) == hipSuccess &&
       hipMemcpy(bf.data(), d_bf, NB * 4, hipMemcpyDeviceToHost) == hipSuccess;
  if (d_exc) (void)hipFree(d_exc);
  if (d_mag) (void)hipFree(d_mag);
  if (d_bi) (void)hipFree(d_bi);
  if (d_bf) (void)hipFree(d_bf);
  if (!ok) return -1;
  *bad_mag = 0; *bad_bin = 0;
  for (int i = 0; i < 65536; ++i) *bad_mag += memcmp(&mags[i], &tab[i], 8) != 0;
  for (int i = 0; i < NB; ++i) *bad_bin += bi[i] != bf[i];
  return 0;
}
}  // namespace vtx

extern "C" int vtx_hog_fwd(const uint8_t* frames, int F, int H, int W, const double* table, size_t table_bytes, double* out,
                           int32_t* bins, void* stream) {
  VTX_REQUIRE(F >= 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0 && W <= 1024, VTX_EINVAL,
              "hog_fwd: H=%d, W=%d must be multiples of 16 (W <= 1024)", H, W);
  if (F == 0) return VTX_OK;                       // empty batch: nothing to do (pointers may be null)
  VTX_REQUIRE(frames && table && out, VTX_EINVAL, "hog_fwd: null pointer");
  // the blob is 512 KB of magnitudes + the 16 KB of correction words the kernel reads: a caller that still uploads the
  // magnitude table of library version 200 and earlier (512 KB) would have the kernel read 16 KB beyond it
  VTX_REQUIRE(table_bytes >= vtx_hog_table_bytes(), VTX_EINVAL, "hog_fwd: the table blob holds %zu bytes, vtx_hog_table_bytes() = %zu "
              "(magnitudes + correction words: build it with vtx_hog_build_table)", table_bytes, vtx_hog_table_bytes());
  VTX_REQUIRE(aligned16(frames) && aligned16(out), VTX_EALIGN, "hog_fwd: frames and out must be 16-byte aligned");
  const uint32_t* exc = reinterpret_cast<const uint32_t*>(table + 256 * 256);   // the correction words behind the 65 536 doubles
  // W <= 256: the three channels side by side (768 threads); wider frames: one channel at a time (256 threads, one magnitude tile)
  const int cp = W <= 256 ? 3 : 1;
  const size_t lds = (size_t)HOG_EXC_WORDS * 4 + (size_t)30 * W + (size_t)cp * 8 * W * 8 + (size_t)cp * (W / 8) * 18 * 4 + (size_t)(W / 8) * 27 * 8;
  static std::atomic<unsigned long long> attr_set{0};
  if (first_launch_on_device(attr_set)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&hog_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&hog_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  // persistent: as many workgroups as fit at once (LDS- and wave-limited), each walks strips with stride gridDim
  const long items = (long)F * (H / 8);
  long per_cu = (long)(160 * 1024) / (long)lds; if (per_cu < 1) per_cu = 1;
  const long wave_cap = 32 / (4 * cp); if (per_cu > wave_cap) per_cu = wave_cap;
  long grid = per_cu * device_cus(); if (grid > items) grid = items;
  if (cp == 3)
    hipLaunchKernelGGL(hog_kernel<3>, dim3((unsigned)grid), dim3(768), lds, as_stream(stream), frames, F, H, W, exc, out, bins);
  else
    hipLaunchKernelGGL(hog_kernel<1>, dim3((unsigned)grid), dim3(256), lds, as_stream(stream), frames, F, H, W, exc, out, bins);
  return check_launch("hog_fwd");
}

extern "C" int vtx_maskfeat_blend_fwd(int dtype, int B, int Tq, int Hq, int Wq, int C, int g, const void* x,
                                      const uint8_t* mask, const float* mask_token, void* out, void* stream) {
  VTX_REQUIRE(B > 0 && C % 8 == 0 && g > 0 && Hq % g == 0 && Wq % g == 0 && Hq / g == Wq / g && x && mask && mask_token && out,
              VTX_EINVAL, "maskfeat_blend_fwd: bad arguments");
  const long rows = (long)B * Tq * Hq * Wq;
  long gsz = (rows * (C / 8) + 255) / 256; if (gsz > 8192) gsz = 8192;
  dim3 grid((unsigned)gsz), block(256);
  hipStream_t st = as_stream(stream);
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(mf_blend_fwd_kernel<float>, grid, block, 0, st, rows, Tq, Hq, Wq, C, g, (const float*)x, mask, mask_token, (float*)out);
  else if (dtype == VTX_BF16)
    hipLaunchKernelGGL(mf_blend_fwd_kernel<bf16raw>, grid, block, 0, st, rows, Tq, Hq, Wq, C, g, (const bf16raw*)x, mask, mask_token, (bf16raw*)out);
  else
    VTX_REQUIRE(false, VTX_EINVAL, "maskfeat_blend_fwd: bad dtype");
  return check_launch("maskfeat_blend_fwd");
}

extern "C" int vtx_maskfeat_blend_bwd(int dtype, int B, int Tq, int Hq, int Wq, int C, int g, const void* dy,
                                      const uint8_t* mask, void* dx, float* dtoken, void* stream) {
  VTX_REQUIRE(B > 0 && C % 8 == 0 && g > 0 && Hq % g == 0 && Wq % g == 0 && dy && mask && dx && dtoken, VTX_EINVAL,
              "maskfeat_blend_bwd: bad arguments");
  const long rows = (long)B * Tq * Hq * Wq;
  long gy = (rows + 255) / 256; if (gy > 1024) gy = 1024;
  dim3 grid(C / 8, (unsigned)gy), block(256);
  hipStream_t st = as_stream(stream);
  hipMemsetAsync(dtoken, 0, (size_t)C * sizeof(float), st);
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(mf_blend_bwd_kernel<float>, grid, block, 0, st, rows, Tq, Hq, Wq, C, g, (const float*)dy, mask, (float*)dx, dtoken);
  else if (dtype == VTX_BF16)
    hipLaunchKernelGGL(mf_blend_bwd_kernel<bf16raw>, grid, block, 0, st, rows, Tq, Hq, Wq, C, g, (const bf16raw*)dy, mask, (bf16raw*)dx, dtoken);
  else
    VTX_REQUIRE(false, VTX_EINVAL, "maskfeat_blend_bwd: bad dtype");
  return check_launch("maskfeat_blend_bwd");
}

extern "C" int vtx_maskfeat_loss_fwd(int dtype, int B, int Tq, int ts, int g, int Cf, const void* pred, long ldp,
                                     const double* target, const uint8_t* cmask, double* loss_out, void* stream) {
  VTX_REQUIRE(B > 0 && Tq > 0 && ts > 0 && g > 0 && Cf > 0 && pred && target && cmask && loss_out, VTX_EINVAL,
              "maskfeat_loss_fwd: bad arguments");
  const long cells = (long)B * Tq * ts * g * g;
  hipStream_t st = as_stream(stream);
  hipMemsetAsync(loss_out, 0, 2 * sizeof(double), st);
  long gsz = (cells * 64 + 255) / 256; if (gsz > 2048) gsz = 2048;
  dim3 grid((unsigned)gsz), block(256);
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(mf_loss_fwd_kernel<float>, grid, block, 0, st, cells, Tq, ts, g, Cf, (const float*)pred, ldp, target, cmask, loss_out);
  else if (dtype == VTX_BF16)
    hipLaunchKernelGGL(mf_loss_fwd_kernel<bf16raw>, grid, block, 0, st, cells, Tq, ts, g, Cf, (const bf16raw*)pred, ldp, target, cmask, loss_out);
  else
    VTX_REQUIRE(false, VTX_EINVAL, "maskfeat_loss_fwd: bad dtype");
  hipLaunchKernelGGL(mf_loss_finish_kernel, dim3(1), dim3(1), 0, st, loss_out);
  return check_launch("maskfeat_loss_fwd");
}

extern "C" int vtx_maskfeat_loss_bwd(int dtype, int B, int Tq, int ts, int g, int Cf, const void* pred, long ldp,
                                     const double* target, const uint8_t* cmask, const double* loss_out, float gloss,
                                     void* dpred, long lddp, void* stream) {
  VTX_REQUIRE(B > 0 && pred && target && cmask && loss_out && dpred, VTX_EINVAL, "maskfeat_loss_bwd: bad arguments");
  const long rows = (long)B * Tq * g * g;
  long gsz = (rows * ts * Cf + 255) / 256; if (gsz > 8192) gsz = 8192;
  dim3 grid((unsigned)gsz), block(256);
  hipStream_t st = as_stream(stream);
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(mf_loss_bwd_kernel<float>, grid, block, 0, st, rows, Tq, ts, g, Cf, (const float*)pred, ldp, target, cmask, loss_out, gloss, (float*)dpred, lddp);
  else if (dtype == VTX_BF16)
    hipLaunchKernelGGL(mf_loss_bwd_kernel<bf16raw>, grid, block, 0, st, rows, Tq, ts, g, Cf, (const bf16raw*)pred, ldp, target, cmask, loss_out, gloss, (bf16raw*)dpred, lddp);
  else
    VTX_REQUIRE(false, VTX_EINVAL, "maskfeat_loss_bwd: bad dtype");
  return check_launch("maskfeat_loss_bwd");
}
