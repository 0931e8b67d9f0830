// hog.hip -- MaskFeat HOG target extractor and the MaskFeat head's small kernels.
//
// HOG: reference dataset.py:39-45 -> skimage.feature.hog(orientations=9,
// pixels_per_cell=(8,8), cells_per_block=(1,1), block_norm='L2') per channel.
// Bit-exact contract (see oracle/hog_oracle.py for the derivation):
//   * gradients are integer central differences (0 on the image border);
//   * the orientation bin is decided by exact sign tests against the eight
//     interior bin boundaries (integer gradients only ever touch the 0-degree
//     boundary), so no atan2 is evaluated;
//   * magnitude = libm hypot(g_col, g_row), taken from a host-built 256x256
//     table indexed by |g_row|,|g_col| (libm hypot is not always the correctly
//     rounded sqrt, and bit-exactness vs skimage needs the same function);
//   * per (cell, bin): the 64 magnitudes are accumulated in row-major order into
//     a float32 accumulator with double adds, divided by 64 in float32;
//   * per cell: L2 norm in float64 with numpy's pairwise order for 9 terms.
// HBM-bound: 150 528 B read + 169 344 B written per 224x224 frame.
// One workgroup per (frame, channel, cell row): 28 cells x 9 bins = 252 threads.
#include <math.h>
#include "common.h"

namespace vtx {

__constant__ double HOG_BC[8] = {0.93969262078590838, 0.76604444311897804, 0.5, 0.17364817766693035,
                                 -0.17364817766693035, -0.5, -0.76604444311897804, -0.93969262078590838};
__constant__ double HOG_BS[8] = {0.34202014332566873, 0.64278760968653933, 0.8660254037844386, 0.98480775301220806,
                                 0.98480775301220806, 0.8660254037844386, 0.64278760968653933, 0.34202014332566873};

__device__ inline int hog_bin(int gr, int gc) {
  if (gr == 0 && gc == 0) return 0;
  if (gr < 0 || (gr == 0 && gc < 0)) { gr = -gr; gc = -gc; }
  int b = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) b += (HOG_BC[k] * (double)gr - HOG_BS[k] * (double)gc >= 0.0) ? 1 : 0;
  return b;
}

// frames [F,H,W,3] u8; grid = (H/8 cell rows, 3 channels, F); block = 256 threads.
__global__ __launch_bounds__(256) void hog_kernel(const uint8_t* __restrict__ frames, int H, int W,
                                                  const double* __restrict__ table, double* __restrict__ out,
                                                  int32_t* __restrict__ bins) {
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  const int nc = W / 8;
  // LDS: pixels of rows [y0-1, y0+8] for this channel (10 x W u8), then per-pixel bin (u8) and magnitude (f64)
  uint8_t* px = reinterpret_cast<uint8_t*>(hsm);                       // [10][W]
  double* mag = reinterpret_cast<double*>(hsm + ((10 * W + 15) & ~15)); // [8][W]
  uint8_t* pb = reinterpret_cast<uint8_t*>(mag + 8 * W);               // [8][W]
  double* hist = reinterpret_cast<double*>(pb + ((8 * W + 15) & ~15)); // [nc][9]
  const int cr = blockIdx.x, ch = blockIdx.y, f = blockIdx.z;
  const int y0 = cr * 8;
  const uint8_t* img = frames + (long)f * H * W * 3;
  for (int i = threadIdx.x; i < 10 * W; i += blockDim.x) {
    const int ry = i / W, x = i - ry * W;
    const int y = y0 - 1 + ry;
    px[i] = (y >= 0 && y < H) ? img[((long)y * W + x) * 3 + ch] : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * W; i += blockDim.x) {
    const int ry = i / W, x = i - ry * W;
    const int y = y0 + ry;
    int gr = 0, gc = 0;
    if (y > 0 && y < H - 1) gr = (int)px[(ry + 2) * W + x] - (int)px[ry * W + x];
    if (x > 0 && x < W - 1) gc = (int)px[(ry + 1) * W + x + 1] - (int)px[(ry + 1) * W + x - 1];
    const int b = hog_bin(gr, gc);
    pb[i] = (uint8_t)b;
    mag[i] = table[abs(gr) * 256 + abs(gc)];
    if (bins) bins[(((long)f * 3 + ch) * H + y) * W + x] = b;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc * 9; i += blockDim.x) {
    const int cc = i / 9, ob = i - cc * 9;
    float tot = 0.0f;
    for (int py = 0; py < 8; ++py)
      for (int pxx = 0; pxx < 8; ++pxx) {
        const int o = py * W + cc * 8 + pxx;
        if (pb[o] == ob) tot = (float)((double)tot + mag[o]);
      }
    hist[i] = (double)(tot / 64.0f);
  }
  __syncthreads();
  for (int cc = threadIdx.x; cc < nc; cc += blockDim.x) {
    const double* h = hist + cc * 9;
    double s = ((h[0] * h[0] + h[1] * h[1]) + (h[2] * h[2] + h[3] * h[3])) +
               ((h[4] * h[4] + h[5] * h[5]) + (h[6] * h[6] + h[7] * h[7]));
    s += h[8] * h[8];
    const double nrm = sqrt(s + 1e-5 * 1e-5);
    const int ph = cr >> 1, dh = cr & 1, pw = cc >> 1, dw = cc & 1;
    double* o = out + (((long)f * (H / 16) + ph) * (W / 16) + pw) * 108 + dh * 54 + dw * 27 + ch * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = h[k] / nrm;
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

extern "C" size_t vtx_hog_table_bytes(void) { return (size_t)256 * 256 * sizeof(double); }

extern "C" int vtx_hog_build_table(double* host_table) {
  VTX_REQUIRE(host_table != nullptr, VTX_EINVAL, "hog_build_table: null pointer");
  for (int r = 0; r < 256; ++r)
    for (int c = 0; c < 256; ++c) host_table[r * 256 + c] = hypot((double)c, (double)r);
  return VTX_OK;
}

extern "C" int vtx_hog_fwd(const uint8_t* frames, int F, int H, int W, const double* table, double* out, int32_t* bins,
                           void* stream) {
  VTX_REQUIRE(F >= 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0 && W <= 1024, VTX_EINVAL,
              "hog_fwd: H=%d, W=%d must be multiples of 16 (W <= 1024)", H, W);
  if (F == 0) return VTX_OK;                       // empty batch: nothing to do (pointers may be null)
  VTX_REQUIRE(frames && table && out, VTX_EINVAL, "hog_fwd: null pointer");
  const size_t lds = ((10 * W + 15) & ~15) + (size_t)8 * W * 8 + ((8 * W + 15) & ~15) + (size_t)(W / 8) * 9 * 8;
  dim3 grid(H / 8, 3, F), block(256);
  hipLaunchKernelGGL(hog_kernel, grid, block, lds, as_stream(stream), frames, H, W, table, out, bins);
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
