// mvit.hip -- the operators of the MViT-B backbone that the transformer kernels do not cover
// (reference video_transformer.py:621-800 builds the backbone from
// pytorchvideo -- semantics restated in oracle/mvit_oracle.py, parity unpinned by the reference):
//
//   pool_conv_ln   pooling of q / k / v inside MultiScaleAttention: per-head depthwise Conv3d(3x3x3, stride
//                  (1,s,s), padding 1, no bias) over the [T,H,W] token grid + LayerNorm(head_dim); the cls token
//                  bypasses the convolution but not the norm.  Tokens stay in [B, 1+T*H*W, heads*hd] layout
//                  (heads interleaved in the feature axis): the conv weight of column c is w[c % hd].
//   maxpool_skip   MaxPool3d(kernel (1,3,3), stride (1,2,2), padding (0,1,1)) on the residual path, cls kept.
//   xattn          softmax(q k^T hd^-0.5) v with Lq != Lk (queries and keys pooled by different strides),
//                  head_dim 96 (MViT-B) or 64.  bf16: the MFMA kernels of xattn_mfma.hip; fp32 (and option attn_valu):
//                  the VALU kernels below (K/V tiles in LDS, online softmax).
//   pos_encoding   separable spatial + temporal position embedding and the cls token.
//   im2col3d       rows of the overlapping Conv3d(3 -> 96, kernel (3,7,7), stride (2,4,4), padding (1,3,3)) stem.
//
// The pooling kernels work on 8 channels per thread (16-byte accesses, 32-bit index arithmetic, conv weights in LDS as
// [tap][channel]); maxpool / pos_encoding / im2col3d are one-element-per-thread kernels (the backbone is 70 GFLOP per clip
// against TimeSformer-B's 392; DESIGN.md 4.6 has the step breakdown).
#include "common.h"

namespace vtx {

// ---------------------------------------------------------------------------------------------------
// pool_conv_ln.  A group of LPU lanes owns one (output token, head); lane l < hd/8 of the group holds channels
// 8l .. 8l+7 of the head (one 16-byte load per tap; hd 96 leaves 4 of 16 lanes idle).  Conv weights in LDS as [tap][channel].
template <typename T, int HD>
__global__ __launch_bounds__(256) void pool_conv_ln_fwd_kernel(int B, int Tn, int H, int W, int Ho, int Wo, int sh, int sw, int heads,
                                                               const T* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                               T* __restrict__ pre, T* __restrict__ y, float* __restrict__ mean,
                                                               float* __restrict__ rstd) {
  constexpr int LPU = HD == 96 ? 16 : 8, UPB = 256 / LPU;
  __shared__ float s_w[27 * HD];
  for (int i = threadIdx.x; i < 27 * HD; i += 256) s_w[i] = w[(i % HD) * 27 + i / HD];
  __syncthreads();
  const int C = heads * HD;
  const unsigned HoWo = (unsigned)Ho * Wo;
  const unsigned n_in = 1 + (unsigned)Tn * H * W, n_out = 1 + (unsigned)Tn * HoWo;
  const unsigned units = (unsigned)B * n_out * heads;                     // < 2^31 (checked by the launcher)
  const unsigned unit = blockIdx.x * UPB + threadIdx.x / LPU;             // (b, out token, head)
  const int l = threadIdx.x % LPU;
  const bool act = l < HD / 8 && unit < units;
  const unsigned bo = unit / heads, hd_i = unit - bo * heads;
  const unsigned b = bo / n_out, o = bo - b * n_out;
  const int c0 = (int)hd_i * HD + l * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  const long orow = ((long)b * n_out + o) * C + c0;
  if (act) {
    const T* xb = x + (long)b * n_in * C + c0;
    if (o == 0) {
      load8(xb, v);
    } else {
      const unsigned r = o - 1;
      const unsigned to = r / HoWo, rr = r - to * HoWo, ho = rr / (unsigned)Wo, wo = rr - ho * Wo;
      for (int kt = 0; kt < 3; ++kt) {
        const int t = (int)to + kt - 1;
        if (t < 0 || t >= Tn) continue;
        for (int kh = 0; kh < 3; ++kh) {
          const int hh = (int)ho * sh + kh - 1;
          if (hh < 0 || hh >= H) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int ww = (int)wo * sw + kw - 1;
            if (ww < 0 || ww >= W) continue;
            float xv[8];
            load8(xb + (1 + ((long)t * H + hh) * W + ww) * C, xv);
            const float* wt = s_w + ((kt * 3 + kh) * 3 + kw) * HD + l * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(wt[j], xv[j], v[j]);
          }
        }
      }
    }
    store8(pre + orow, v);
    load8(pre + orow, v);                       // the LayerNorm sees the stored (rounded) value, as backward will
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
#pragma unroll
  for (int m = LPU / 2; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
  const float mu = s / HD;
  float q = 0.f;
  if (act) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = v[j] - mu; q += d * d; }
  }
#pragma unroll
  for (int m = LPU / 2; m > 0; m >>= 1) q += __shfl_xor(q, m, 64);
  const float rs = rsqrtf(q / HD + eps);
  if (act) {
    float g8[8], b8[8];
    load8(gamma + l * 8, g8);
    load8(beta + l * 8, b8);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (v[j] - mu) * rs * g8[j] + b8[j];
    store8(y + orow, v);
    if (l == 0) { mean[unit] = mu; rstd[unit] = rs; }
  }
}

// LayerNorm backward per (token, head): dpre, and per-block partial sums of dgamma / dbeta
template <typename T, int HD>
__global__ __launch_bounds__(256) void pool_ln_bwd_kernel(long units, int heads, const T* __restrict__ dy, const T* __restrict__ pre,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, T* __restrict__ dpre, float* __restrict__ part) {
  constexpr int CPL = HD / 32;
  __shared__ float red[8][2][HD];
  const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
  float dg[CPL], db[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) { dg[j] = 0.f; db[j] = 0.f; }
  for (long unit = (long)blockIdx.x * 8 + hw; unit < units; unit += (long)gridDim.x * 8) {
    const long off = (unit / heads) * ((long)heads * HD) + (unit % heads) * HD;
    const float mu = mean[unit], rs = rstd[unit];
    float g[CPL], xh[CPL], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const float d = ET<T>::ld(dy + off + l + 32 * j);
      xh[j] = (ET<T>::ld(pre + off + l + 32 * j) - mu) * rs;
      g[j] = d * gamma[l + 32 * j];
      s1 += g[j]; s2 += g[j] * xh[j];
      dg[j] += d * xh[j]; db[j] += d;
    }
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
    s1 /= HD; s2 /= HD;
#pragma unroll
    for (int j = 0; j < CPL; ++j) ET<T>::st(dpre + off + l + 32 * j, rs * (g[j] - s1 - xh[j] * s2));
  }
#pragma unroll
  for (int j = 0; j < CPL; ++j) { red[hw][0][l + 32 * j] = dg[j]; red[hw][1][l + 32 * j] = db[j]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * HD; i += 256) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += red[k][i / HD][i % HD];
    part[(long)blockIdx.x * 2 * HD + i] = a;
  }
}

// dx of the depthwise conv (gather over the outputs whose window holds the input token); cls row passes through.
// One block per (clip, t, h) row of the input grid (+ one per clip for the cls row): which kt / kh taps reach an output
// is block-uniform, only the kw test is per thread.  A thread owns 8 consecutive channels of one token of the row
// (one 16-byte load per live tap); the conv weights sit in LDS as [tap][channel].  Tap order (kt, kh, kw) is fixed.
template <typename T, int HD>
__global__ __launch_bounds__(256) void pool_conv_bwd_data_kernel(int B, int Tn, int H, int W, int Ho, int Wo, int sh, int sw, int C,
                                                                 const T* __restrict__ dpre, const float* __restrict__ w,
                                                                 T* __restrict__ dx) {
  __shared__ float s_w[27 * HD];
  const int rows = Tn * H + 1;
  const int b = blockIdx.x / rows, row = blockIdx.x - b * rows;
  const long n_in = 1 + (long)Tn * H * W, n_out = 1 + (long)Tn * Ho * Wo;
  const int C8 = C / 8;
  const T* db = dpre + (long)b * n_out * C;
  T* dxb = dx + (long)b * n_in * C;
  float a[8];
  if (row == rows - 1) {                                  // cls
    for (int i = threadIdx.x; i < C8; i += 256) { load8(db + i * 8, a); store8(dxb + i * 8, a); }
    return;
  }
  const int t = row / H, h = row - t * H;
  int hq[3];                                              // output row reached through tap kh, or -1
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hn = h - kh + 1, q = hn >= 0 ? hn / sh : -1;
    hq[kh] = (q >= 0 && q * sh == hn && q < Ho) ? q : -1;
  }
  T* dxr = dxb + (1 + (long)row * W) * C;
  const int items = W * C8;
  if ((hq[0] & hq[1] & hq[2]) < 0) {                      // no output row looks at this input row
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    for (int i = threadIdx.x; i < items; i += 256) store8(dxr + (long)i * 8, a);
    return;
  }
  for (int i = threadIdx.x; i < 27 * HD; i += 256) s_w[i] = w[(i % HD) * 27 + i / HD];
  __syncthreads();
  for (int i = threadIdx.x; i < items; i += 256) {
    const int ww = i / C8, c0 = (i - ww * C8) * 8;
    const float* wc = s_w + c0 % HD;
    int wq[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wn = ww - kw + 1, q = wn >= 0 ? wn / sw : -1;
      wq[kw] = (q >= 0 && q * sw == wn && q < Wo) ? q : -1;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int to = t - kt + 1;
      if (to < 0 || to >= Tn) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        if (hq[kh] < 0) continue;
        const T* dr = db + (1 + ((long)to * Ho + hq[kh]) * Wo) * C + c0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          if (wq[kw] < 0) continue;
          float v[8];
          load8(dr + (long)wq[kw] * C, v);
          const float* wt = wc + ((kt * 3 + kh) * 3 + kw) * HD;
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = fmaf(wt[j], v[j], a[j]);
        }
      }
    }
    store8(dxr + (long)i * 8, a);
  }
}

// dW[c][tap] partials: each block covers a slice of (b, output token) pairs.  A thread owns one tap of one 8-channel
// chunk of the head (32 lanes per chunk, 27 of them live): per pair it reads the chunk of dpre and the chunk of x at
// its tap's position (two 16-byte loads) for 8 multiply-adds.  The pairs' rows and grid positions are worked out once
// per batch of PB pairs into LDS.  One accumulator set per thread, summed over heads and pairs in a fixed order.
template <typename T, int HD>
__global__ __launch_bounds__((HD / 8) * 32) void pool_conv_bwd_weight_kernel(int B, int Tn, int H, int W, int Ho, int Wo, int sh, int sw,
                                                                             int heads, const T* __restrict__ dpre,
                                                                             const T* __restrict__ x, float* __restrict__ part) {
  constexpr int PB = 64;
  __shared__ int4 s_pair[PB];      // {dpre row, x row of the window's corner (may lie off the grid), t+1 | (h+1)<<8 | (w+1)<<20 of it, -}
  const int C = heads * HD;
  const unsigned HoWo = (unsigned)Ho * Wo;
  const unsigned n_in = 1 + (unsigned)Tn * H * W, per_clip = (unsigned)Tn * HoWo, n_out = 1 + per_clip;
  const long pairs = (long)B * per_clip;
  const long per = (pairs + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per, p1 = min(pairs, p0 + per);
  const int tap = threadIdx.x & 31, chunk = threadIdx.x >> 5;
  const bool live = tap < 27;
  const int kt = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
  const int tap_off = (kt * H + kh) * W + kw;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (long q0 = p0; q0 < p1; q0 += PB) {
    const int nb = (int)min((long)PB, p1 - q0);
    __syncthreads();
    if ((int)threadIdx.x < nb) {
      const unsigned p = (unsigned)(q0 + threadIdx.x);
      const unsigned b = p / per_clip, r = p - b * per_clip;
      const unsigned to = r / HoWo, rr = r - to * HoWo, ho = rr / (unsigned)Wo, wo = rr - ho * Wo;
      const int t0 = (int)to - 1, h0 = (int)(ho * sh) - 1, w0 = (int)(wo * sw) - 1;
      s_pair[threadIdx.x] = make_int4((int)(b * n_out + 1 + r), (int)(b * n_in) + 1 + (t0 * H + h0) * W + w0,
                                      (t0 + 1) | ((h0 + 1) << 8) | ((w0 + 1) << 20), 0);
    }
    __syncthreads();
    if (live) {
      for (int hh = 0; hh < heads; ++hh) {
        const int c0 = hh * HD + chunk * 8;
        for (int i = 0; i < nb; ++i) {
          const int4 pr = s_pair[i];
          const int t = (pr.z & 255) - 1 + kt, h = ((pr.z >> 8) & 4095) - 1 + kh, ww = (int)((unsigned)pr.z >> 20) - 1 + kw;
          if ((unsigned)t < (unsigned)Tn && (unsigned)h < (unsigned)H && (unsigned)ww < (unsigned)W) {
            float d8[8], x8[8];
            load8(dpre + (long)pr.x * C + c0, d8);
            load8(x + (long)(pr.y + tap_off) * C + c0, x8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(d8[j], x8[j], acc[j]);
          }
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < 8; ++j) part[(long)blockIdx.x * 27 * HD + (chunk * 8 + j) * 27 + tap] = acc[j];
  }
}

// ---------------------------------------------------------------------------------------------------
// MaxPool3d (1,3,3)/(1,2,2)/(0,1,1) on the residual path; arg = winning tap (kh*3+kw), 255 for the cls row.
// A thread owns 8 consecutive channels of one token (16-byte accesses, 8 argument bytes at a time, 32-bit index arithmetic).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(int B, int Tn, int H, int W, int Ho, int Wo, int C, const T* __restrict__ x,
                                                          T* __restrict__ y, uint8_t* __restrict__ arg) {
  const unsigned C8 = (unsigned)C / 8, HoWo = (unsigned)Ho * Wo;
  const unsigned n_in = 1 + (unsigned)Tn * H * W, n_out = 1 + (unsigned)Tn * HoWo;
  const unsigned total = (unsigned)B * n_out * C8;                      // < 2^31 (checked by the launcher)
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const unsigned bo = i / C8, c0 = (i - bo * C8) * 8;
    const unsigned b = bo / n_out, o = bo - b * n_out;
    const T* xb = x + (long)b * n_in * C + c0;
    const long oi = ((long)b * n_out + o) * C + c0;
    float best[8];
    unsigned bi[8];
    if (o == 0) {
      load8(xb, best);
      store8(y + oi, best);
      *reinterpret_cast<uint2*>(arg + oi) = make_uint2(0xffffffffu, 0xffffffffu);
      continue;
    }
    const unsigned r = o - 1;
    const unsigned t = r / HoWo, rr = r - t * HoWo, ho = rr / (unsigned)Wo, wo = rr - ho * Wo;
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = (int)ho * 2 + kh - 1;
      if (hh < 0 || hh >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = (int)wo * 2 + kw - 1;
        if (ww < 0 || ww >= W) continue;
        float v[8];
        load8(xb + (1 + ((long)t * H + hh) * W + ww) * C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = kh * 3 + kw; }   // strictly greater (or NaN): the first maximum wins, as ATen
      }
    }
    store8(y + oi, best);
    *reinterpret_cast<uint2*>(arg + oi) = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24),
                                                     bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(int B, int Tn, int H, int W, int Ho, int Wo, int C, const T* __restrict__ dy,
                                                          const uint8_t* __restrict__ arg, T* __restrict__ dx) {
  const unsigned C8 = (unsigned)C / 8, HW = (unsigned)H * W;
  const unsigned n_in = 1 + (unsigned)Tn * HW, n_out = 1 + (unsigned)Tn * Ho * Wo;
  const unsigned total = (unsigned)B * n_in * C8;                       // < 2^31 (checked by the launcher)
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const unsigned bn = i / C8, c0 = (i - bn * C8) * 8;
    const unsigned b = bn / n_in, n = bn - b * n_in;
    const long ob = (long)b * n_out * C + c0;
    float a[8];
    if (n == 0) {
      load8(dy + ob, a);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      const unsigned r = n - 1;
      const unsigned t = r / HW, rr = r - t * HW, hu = rr / (unsigned)W;
      const int h = (int)hu, w = (int)(rr - hu * W);
      for (int kh = 0; kh < 3; ++kh) {
        const int hn = h - kh + 1;
        if (hn < 0 || (hn & 1) || hn / 2 >= Ho) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int wn = w - kw + 1;
          if (wn < 0 || (wn & 1) || wn / 2 >= Wo) continue;
          const long o = ob + (1 + ((long)t * Ho + hn / 2) * Wo + wn / 2) * C;
          const uint2 ar = *reinterpret_cast<const uint2*>(arg + o);
          float v[8];
          load8(dy + o, v);
          const unsigned tap = kh * 3 + kw;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if ((((j < 4 ? ar.x : ar.y) >> (8 * (j & 3))) & 255u) == tap) a[j] += v[j];
        }
      }
    }
    store8(dx + ((long)b * n_in + n) * C + c0, a);
  }
}

// ---------------------------------------------------------------------------------------------------
// out[b,0] = cls + pos_class;  out[b,1+n] = x[b,n] + spatial[n % HW] + temporal[n / HW]
template <typename T>
__global__ __launch_bounds__(256) void pos_encoding_kernel(int B, int Tn, int HW, int C, const T* __restrict__ x, const float* __restrict__ cls,
                                                           const float* __restrict__ pos_class, const float* __restrict__ spatial,
                                                           const float* __restrict__ temporal, T* __restrict__ out) {
  const long n1 = 1 + (long)Tn * HW, total = (long)B * n1 * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long bn = i / C;
    const long n = bn % n1;
    const long b = bn / n1;
    float v;
    if (n == 0) v = cls[c] + pos_class[c];
    else {
      const long r = n - 1;
      v = ET<T>::ld(x + (b * (n1 - 1) + r) * C + c) + (spatial[(r % HW) * C + c] + temporal[(r / HW) * C + c]);
    }
    ET<T>::st(out + i, v);
  }
}

// rows[(b,to,ho,wo)][c*kt*kh*kw ...] of the stem convolution; clip is [B, Tc, Cc, H, W] fp32 (the module's input layout);
// row width Kp >= Cc*KT*KH*KW, zero padded
// A thread writes 8 consecutive columns of one row (one 16-byte store for bf16): the row's (b, to, ho, wo) and the first
// column's (c, kt, kh, kw) are decoded once with 32-bit divisions, the other seven columns by carrying.
template <typename T>
__global__ __launch_bounds__(256) void im2col3d_kernel(int B, int Tc, int Cc, int H, int W, int KT, int KH, int KW, int st, int sh, int sw,
                                                       int pt, int ph, int pw, int To, int Ho, int Wo, int Kp, const float* __restrict__ clip,
                                                       T* __restrict__ rows) {
  const unsigned K8 = (unsigned)Kp / 8;
  const unsigned total = (unsigned)B * To * Ho * Wo * K8;               // < 2^31 (checked by the launcher)
  const int K = Cc * KT * KH * KW;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const unsigned row = i / K8, k0 = (i - row * K8) * 8;
    const unsigned r1 = row / (unsigned)Wo, wo = row - r1 * Wo;
    const unsigned r2 = r1 / (unsigned)Ho, ho = r1 - r2 * Ho;
    const unsigned b = r2 / (unsigned)To, to = r2 - b * To;
    const unsigned q1 = k0 / (unsigned)KW;
    int kw = (int)(k0 - q1 * KW);
    const unsigned q2 = q1 / (unsigned)KH;
    int kh = (int)(q1 - q2 * KH);
    int c = (int)(q2 / (unsigned)KT), kt = (int)(q2 - (unsigned)c * KT);
    const int t0 = (int)to * st - pt, h0 = (int)ho * sh - ph, w0 = (int)wo * sw - pw;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = t0 + kt, h = h0 + kh, w = w0 + kw;
      v[j] = 0.f;
      if ((int)k0 + j < K && t >= 0 && t < Tc && h >= 0 && h < H && w >= 0 && w < W)
        v[j] = clip[((((long)b * Tc + t) * Cc + c) * H + h) * W + w];
      if (++kw == KW) { kw = 0; if (++kh == KH) { kh = 0; if (++kt == KT) { kt = 0; ++c; } } }
    }
    store8(rows + (long)row * Kp + k0, v);
  }
}

// ---------------------------------------------------------------------------------------------------
// Cross attention.  q [B, Lq, H*HD], k / v [B, Lk, H*HD] (heads interleaved), out [B, Lq, H*HD], lse [B, H, Lq].
// One thread per query row (forward, dq) or key row (dk/dv); the other side streams through LDS in 32-row tiles.
constexpr int XA_THREADS = 128, XA_TILE = 32;

template <typename T, int HD> __device__ inline void xa_load(const T* p, float (&v)[HD]) {
#pragma unroll
  for (int c = 0; c < HD / 8; ++c) {
    float t8[8];
    load8(p + c * 8, t8);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[c * 8 + j] = t8[j];
  }
}
template <typename T, int HD> __device__ inline void xa_store(T* p, const float (&v)[HD]) {
#pragma unroll
  for (int c = 0; c < HD / 8; ++c) {
    float t8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t8[j] = v[c * 8 + j];
    store8(p + c * 8, t8);
  }
}
template <typename T, int HD>
__device__ inline void xa_tile(float* lds, const T* base, long ld, int row0, int nrows) {      // [XA_TILE][HD] fp32
  for (int id = threadIdx.x; id < XA_TILE * (HD / 8); id += XA_THREADS) {
    const int r = id / (HD / 8), c = id % (HD / 8);
    float t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < nrows) load8(base + (long)(row0 + r) * ld + c * 8, t8);
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[r * HD + c * 8 + j] = t8[j];
  }
}

template <typename T, int HD>
__global__ __launch_bounds__(XA_THREADS) void xattn_fwd_kernel(int Lq, int Lk, int heads, float scale, const T* __restrict__ q,
                                                               const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ out,
                                                               float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) float Ks[XA_TILE * HD];
  __shared__ __attribute__((aligned(16))) float Vs[XA_TILE * HD];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * HD;
  const int i = blockIdx.x * XA_THREADS + threadIdx.x;
  const bool active = i < Lq;
  float qv[HD], acc[HD];
#pragma unroll
  for (int e = 0; e < HD; ++e) { qv[e] = 0.f; acc[e] = 0.f; }
  if (active) {
    xa_load<T, HD>(q + ((long)b * Lq + i) * C + h * HD, qv);
#pragma unroll
    for (int e = 0; e < HD; ++e) qv[e] *= scale;
  }
  float m = -INFINITY, l = 0.f;
  const T* kb = k + (long)b * Lk * C + h * HD;
  const T* vb = v + (long)b * Lk * C + h * HD;
  for (int k0 = 0; k0 < Lk; k0 += XA_TILE) {
    const int nk = min(XA_TILE, Lk - k0);
    __syncthreads();
    xa_tile<T, HD>(Ks, kb, C, k0, nk);
    xa_tile<T, HD>(Vs, vb, C, k0, nk);
    __syncthreads();
    if (!active) continue;
    for (int j = 0; j < nk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < HD; ++e) s = fmaf(qv[e], Ks[j * HD + e], s);
      if (s > m) {                                       // rescale on a new maximum (online softmax)
        const float corr = __expf(m - s);
        l *= corr;
#pragma unroll
        for (int e = 0; e < HD; ++e) acc[e] *= corr;
        m = s;
      }
      const float p = __expf(s - m);
      l += p;
#pragma unroll
      for (int e = 0; e < HD; ++e) acc[e] = fmaf(p, Vs[j * HD + e], acc[e]);
    }
  }
  if (active) {
    const float inv = 1.f / l;
#pragma unroll
    for (int e = 0; e < HD; ++e) acc[e] *= inv;
    xa_store<T, HD>(out + ((long)b * Lq + i) * C + h * HD, acc);
    lse[((long)b * heads + h) * Lq + i] = m + __logf(l);
  }
}

// dq (thread per query) and delta = rowsum(do * o)
template <typename T, int HD>
__global__ __launch_bounds__(XA_THREADS) void xattn_bwd_dq_kernel(int Lq, int Lk, int heads, float scale, const T* __restrict__ q,
                                                                  const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ o,
                                                                  const T* __restrict__ dout, const float* __restrict__ lse,
                                                                  float* __restrict__ delta, T* __restrict__ dq) {
  __shared__ __attribute__((aligned(16))) float Ks[XA_TILE * HD];
  __shared__ __attribute__((aligned(16))) float Vs[XA_TILE * HD];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * HD;
  const int i = blockIdx.x * XA_THREADS + threadIdx.x;
  const bool active = i < Lq;
  float qv[HD], dov[HD], acc[HD];
  float dl = 0.f, my_lse = 0.f;
#pragma unroll
  for (int e = 0; e < HD; ++e) { qv[e] = 0.f; dov[e] = 0.f; acc[e] = 0.f; }
  if (active) {
    const long off = ((long)b * Lq + i) * C + h * HD;
    xa_load<T, HD>(q + off, qv);
    xa_load<T, HD>(dout + off, dov);
    float ov[HD];
    xa_load<T, HD>(o + off, ov);
#pragma unroll
    for (int e = 0; e < HD; ++e) { dl = fmaf(dov[e], ov[e], dl); qv[e] *= scale; }
    my_lse = lse[((long)b * heads + h) * Lq + i];
    delta[((long)b * heads + h) * Lq + i] = dl;
  }
  const T* kb = k + (long)b * Lk * C + h * HD;
  const T* vb = v + (long)b * Lk * C + h * HD;
  for (int k0 = 0; k0 < Lk; k0 += XA_TILE) {
    const int nk = min(XA_TILE, Lk - k0);
    __syncthreads();
    xa_tile<T, HD>(Ks, kb, C, k0, nk);
    xa_tile<T, HD>(Vs, vb, C, k0, nk);
    __syncthreads();
    if (!active) continue;
    for (int j = 0; j < nk; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int e = 0; e < HD; ++e) { s = fmaf(qv[e], Ks[j * HD + e], s); dp = fmaf(dov[e], Vs[j * HD + e], dp); }
      const float ds = __expf(s - my_lse) * (dp - dl);
#pragma unroll
      for (int e = 0; e < HD; ++e) acc[e] = fmaf(ds, Ks[j * HD + e], acc[e]);
    }
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < HD; ++e) acc[e] *= scale;
    xa_store<T, HD>(dq + ((long)b * Lq + i) * C + h * HD, acc);
  }
}

// dk, dv (thread per key); queries, dout stream through LDS with their lse / delta
template <typename T, int HD>
__global__ __launch_bounds__(XA_THREADS) void xattn_bwd_dkv_kernel(int Lq, int Lk, int heads, float scale, const T* __restrict__ q,
                                                                   const T* __restrict__ k, const T* __restrict__ v,
                                                                   const T* __restrict__ dout, const float* __restrict__ lse,
                                                                   const float* __restrict__ delta, T* __restrict__ dk, T* __restrict__ dv,
                                                                   int kblocks, int q_per_split, float* __restrict__ part_k,
                                                                   float* __restrict__ part_v, long part_stride) {
  // A key row is one thread; with few keys (393 after pooling) and many queries (25 089) the query range is cut into
  // splits (blockIdx.x / kblocks) whose fp32 partial dk / dv are summed by xattn_reduce_kernel in split order.
  __shared__ __attribute__((aligned(16))) float Qs[XA_TILE * HD];
  __shared__ __attribute__((aligned(16))) float Ds[XA_TILE * HD];
  __shared__ float Ls[XA_TILE], Dl[XA_TILE];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * HD;
  const int split = blockIdx.x / kblocks;
  const int j = (blockIdx.x - split * kblocks) * XA_THREADS + threadIdx.x;
  const int q_begin = split * q_per_split, q_end = min(Lq, q_begin + q_per_split);
  const bool active = j < Lk;
  float kv[HD], vv[HD], dkv[HD], dvv[HD];
#pragma unroll
  for (int e = 0; e < HD; ++e) { kv[e] = 0.f; vv[e] = 0.f; dkv[e] = 0.f; dvv[e] = 0.f; }
  if (active) {
    const long off = ((long)b * Lk + j) * C + h * HD;
    xa_load<T, HD>(k + off, kv);
    xa_load<T, HD>(v + off, vv);
  }
  const T* qb = q + (long)b * Lq * C + h * HD;
  const T* db = dout + (long)b * Lq * C + h * HD;
  const float* lb = lse + ((long)b * heads + h) * Lq;
  const float* dlb = delta + ((long)b * heads + h) * Lq;
  for (int q0 = q_begin; q0 < q_end; q0 += XA_TILE) {
    const int nq = min(XA_TILE, q_end - q0);
    __syncthreads();
    xa_tile<T, HD>(Qs, qb, C, q0, nq);
    xa_tile<T, HD>(Ds, db, C, q0, nq);
    if (threadIdx.x < XA_TILE) {
      Ls[threadIdx.x] = threadIdx.x < nq ? lb[q0 + threadIdx.x] : 0.f;
      Dl[threadIdx.x] = threadIdx.x < nq ? dlb[q0 + threadIdx.x] : 0.f;
    }
    __syncthreads();
    if (!active) continue;
    for (int i = 0; i < nq; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int e = 0; e < HD; ++e) { s = fmaf(Qs[i * HD + e], kv[e], s); dp = fmaf(Ds[i * HD + e], vv[e], dp); }
      const float p = __expf(s * scale - Ls[i]);
      const float ds = p * (dp - Dl[i]);
#pragma unroll
      for (int e = 0; e < HD; ++e) { dvv[e] = fmaf(p, Ds[i * HD + e], dvv[e]); dkv[e] = fmaf(ds, Qs[i * HD + e], dkv[e]); }
    }
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < HD; ++e) dkv[e] *= scale;
    const long off = ((long)b * Lk + j) * C + h * HD;
    if (part_k == nullptr) {
      xa_store<T, HD>(dk + off, dkv);
      xa_store<T, HD>(dv + off, dvv);
    } else {
      xa_store<float, HD>(part_k + split * part_stride + off, dkv);
      xa_store<float, HD>(part_v + split * part_stride + off, dvv);
    }
  }
}

// out[i] = sum over splits (in order) of part[s][i], converted to T
template <typename T>
__global__ __launch_bounds__(256) void xattn_reduce_kernel(const float* __restrict__ part, int nsplit, long stride, long n8, T* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int sp = 0; sp < nsplit; ++sp) {
      float t[8];
      load8(part + sp * stride + i * 8, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += t[e];
    }
    store8(out + i * 8, a);
  }
}

static int grid_for(long work, int cap = 65536) { long g = (work + 255) / 256; return (int)(g > cap ? cap : (g < 1 ? 1 : g)); }

}  // namespace vtx

using namespace vtx;

#define MV_DISPATCH(dtype_, HD_, F32_96, BF_96, F32_64, BF_64, who_)                                    \
  do {                                                                                                  \
    if ((HD_) == 96) { if ((dtype_) == VTX_F32) { F32_96; } else { BF_96; } }                          \
    else if ((HD_) == 64) { if ((dtype_) == VTX_F32) { F32_64; } else { BF_64; } }                     \
    else VTX_REQUIRE(false, VTX_EINVAL, "%s: head_dim %d unsupported (64 or 96)", who_, (HD_));         \
  } while (0)

static int pool_check(const vtx_pool_desc* d, const char* who) {
  VTX_REQUIRE(d != nullptr, VTX_EINVAL, "%s: null descriptor", who);
  VTX_REQUIRE(d->dtype == VTX_F32 || d->dtype == VTX_BF16, VTX_EINVAL, "%s: bad dtype", who);
  VTX_REQUIRE(d->B > 0 && d->T > 0 && d->H > 0 && d->W > 0 && d->heads > 0 && d->sh > 0 && d->sw > 0, VTX_EINVAL, "%s: bad shape", who);
  return VTX_OK;
}
constexpr int POOL_W_BLOCKS = 2048;                                            // slices of the conv weight-gradient sum
static inline int pooled(int n, int s) { return (n + 2 - 3) / s + 1; }        // kernel 3, padding 1

extern "C" int vtx_pool_conv_ln_fwd(const vtx_pool_desc* d, const void* x, const float* w, const float* gamma, const float* beta,
                                    float eps, void* pre, void* y, float* mean, float* rstd, void* stream) {
  int rc = pool_check(d, "pool_conv_ln_fwd");
  if (rc) return rc;
  VTX_REQUIRE(x && w && gamma && beta && pre && y && mean && rstd, VTX_EINVAL, "pool_conv_ln_fwd: null pointer");
  const int Ho = pooled(d->H, d->sh), Wo = pooled(d->W, d->sw);
  const long units = (long)d->B * (1 + (long)d->T * Ho * Wo) * d->heads;
  VTX_REQUIRE(units < (1L << 31) && (long)d->B * (1 + (long)d->T * d->H * d->W) < (1L << 31), VTX_EINVAL,
              "pool_conv_ln_fwd: token grid too large for the 32-bit index arithmetic");
  dim3 g(cdiv(units, d->hd == 96 ? 16 : 32)), blk(256);
  hipStream_t st = as_stream(stream);
#define L_(T_, HD_) hipLaunchKernelGGL((pool_conv_ln_fwd_kernel<T_, HD_>), g, blk, 0, st, d->B, d->T, d->H, d->W, Ho, Wo, d->sh, d->sw, d->heads, \
                                       (const T_*)x, w, gamma, beta, eps, (T_*)pre, (T_*)y, mean, rstd)
  MV_DISPATCH(d->dtype, d->hd, L_(float, 96), L_(bf16raw, 96), L_(float, 64), L_(bf16raw, 64), "pool_conv_ln_fwd");
#undef L_
  return check_launch("pool_conv_ln_fwd");
}

extern "C" size_t vtx_pool_conv_ln_bwd_workspace(const vtx_pool_desc* d) {
  if (!d) return 0;
  return ((size_t)1024 * 2 * d->hd + (size_t)POOL_W_BLOCKS * 27 * d->hd) * sizeof(float);
}

extern "C" int vtx_pool_conv_ln_bwd(const vtx_pool_desc* d, const void* dy, const void* x, const void* pre, const float* mean,
                                    const float* rstd, const float* w, const float* gamma, void* dpre, void* dx, float* dw,
                                    float* dgamma, float* dbeta, void* workspace, size_t ws_bytes, void* stream) {
  int rc = pool_check(d, "pool_conv_ln_bwd");
  if (rc) return rc;
  VTX_REQUIRE(dy && x && pre && mean && rstd && w && gamma && dpre && dx && dw && dgamma && dbeta && workspace, VTX_EINVAL,
              "pool_conv_ln_bwd: null pointer");
  VTX_REQUIRE(ws_bytes >= vtx_pool_conv_ln_bwd_workspace(d), VTX_EWS, "pool_conv_ln_bwd: workspace too small");
  const int Ho = pooled(d->H, d->sh), Wo = pooled(d->W, d->sw), C = d->heads * d->hd;
  const long n_out = 1 + (long)d->T * Ho * Wo, units = (long)d->B * n_out * d->heads;
  hipStream_t st = as_stream(stream);
  float* part_ln = (float*)workspace;                        // [1024][2][hd]
  float* part_w = part_ln + (size_t)1024 * 2 * d->hd;        // [POOL_W_BLOCKS][27][hd]
  const int nb_ln = (int)(cdiv(units, 8) > 1024 ? 1024 : cdiv(units, 8));
  const long pairs = (long)d->B * (n_out - 1);
  const long want_w = cdiv(pairs, 32);                       // >= 32 pairs per block
  const int nb_w = (int)(want_w < 1 ? 1 : want_w > POOL_W_BLOCKS ? POOL_W_BLOCKS : want_w);
  VTX_REQUIRE(d->T < 255 && d->H < 4095 && d->W < 4095 && (long)d->B * (1 + (long)d->T * d->H * d->W) < (1L << 31),
              VTX_EINVAL, "pool_conv_ln_bwd: token grid too large for the 32-bit index arithmetic");
#define LN_(T_, HD_) hipLaunchKernelGGL((pool_ln_bwd_kernel<T_, HD_>), dim3(nb_ln), dim3(256), 0, st, units, d->heads, (const T_*)dy, (const T_*)pre, \
                                        mean, rstd, gamma, (T_*)dpre, part_ln)
  MV_DISPATCH(d->dtype, d->hd, LN_(float, 96), LN_(bf16raw, 96), LN_(float, 64), LN_(bf16raw, 64), "pool_conv_ln_bwd");
#undef LN_
  rc = check_launch("pool_ln_bwd");
  if (rc) return rc;
  rc = launch_reduce_partials(part_ln, nb_ln, 2L * d->hd, 2L * d->hd, dgamma, 0, 1.0f, st, dbeta, d->hd, 0);
  if (rc) return rc;
#define BD_(T_, HD_) hipLaunchKernelGGL((pool_conv_bwd_data_kernel<T_, HD_>), dim3(d->B * (d->T * d->H + 1)), dim3(256), 0, st, d->B, d->T, d->H, d->W, \
                                        Ho, Wo, d->sh, d->sw, C, (const T_*)dpre, w, (T_*)dx)
  MV_DISPATCH(d->dtype, d->hd, BD_(float, 96), BD_(bf16raw, 96), BD_(float, 64), BD_(bf16raw, 64), "pool_conv_ln_bwd");
#undef BD_
  rc = check_launch("pool_conv_bwd_data");
  if (rc) return rc;
#define BW_(T_, HD_) hipLaunchKernelGGL((pool_conv_bwd_weight_kernel<T_, HD_>), dim3(nb_w), dim3((HD_ / 8) * 32), 0, st, d->B, d->T, d->H, d->W, Ho, Wo, \
                                        d->sh, d->sw, d->heads, (const T_*)dpre, (const T_*)x, part_w)
  MV_DISPATCH(d->dtype, d->hd, BW_(float, 96), BW_(bf16raw, 96), BW_(float, 64), BW_(bf16raw, 64), "pool_conv_ln_bwd");
#undef BW_
  rc = check_launch("pool_conv_bwd_weight");
  if (rc) return rc;
  return launch_reduce_partials(part_w, nb_w, 27L * d->hd, 27L * d->hd, dw, 0, 1.0f, st);
}

extern "C" int vtx_maxpool_skip_fwd(int dtype, int B, int T, int H, int W, int C, const void* x, void* y, uint8_t* arg, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && x && y && arg, VTX_EINVAL, "maxpool_skip_fwd: bad arguments (C must be a multiple of 8)");
  const int Ho = pooled(H, 2), Wo = pooled(W, 2);
  VTX_REQUIRE((long)B * (1 + (long)T * H * W) * (C / 8) < (1L << 31), VTX_EINVAL, "maxpool_skip_fwd: tensor too large for the 32-bit index arithmetic");
  const long total = (long)B * (1 + (long)T * Ho * Wo) * (C / 8);
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, T, H, W, Ho, Wo, C, (const float*)x, (float*)y, arg);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16raw>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, T, H, W, Ho, Wo, C, (const bf16raw*)x, (bf16raw*)y, arg);
  return check_launch("maxpool_skip_fwd");
}

extern "C" int vtx_maxpool_skip_bwd(int dtype, int B, int T, int H, int W, int C, const void* dy, const uint8_t* arg, void* dx, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && dy && dx && arg, VTX_EINVAL, "maxpool_skip_bwd: bad arguments (C must be a multiple of 8)");
  const int Ho = pooled(H, 2), Wo = pooled(W, 2);
  VTX_REQUIRE((long)B * (1 + (long)T * H * W) * (C / 8) < (1L << 31), VTX_EINVAL, "maxpool_skip_bwd: tensor too large for the 32-bit index arithmetic");
  const long total = (long)B * (1 + (long)T * H * W) * (C / 8);
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, T, H, W, Ho, Wo, C, (const float*)dy, arg, (float*)dx);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16raw>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, T, H, W, Ho, Wo, C, (const bf16raw*)dy, arg, (bf16raw*)dx);
  return check_launch("maxpool_skip_bwd");
}

extern "C" int vtx_pos_encoding_fwd(int dtype, int B, int T, int HW, int C, const void* x, const float* cls, const float* pos_class,
                                    const float* spatial, const float* temporal, void* out, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && HW > 0 && C > 0 && x && cls && pos_class && spatial && temporal && out, VTX_EINVAL, "pos_encoding_fwd: bad arguments");
  const long total = (long)B * (1 + (long)T * HW) * C;
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(pos_encoding_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, T, HW, C, (const float*)x, cls, pos_class, spatial, temporal, (float*)out);
  else
    hipLaunchKernelGGL(pos_encoding_kernel<bf16raw>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, T, HW, C, (const bf16raw*)x, cls, pos_class, spatial, temporal, (bf16raw*)out);
  return check_launch("pos_encoding_fwd");
}

extern "C" int vtx_im2col3d(int dtype, int B, int T, int C, int H, int W, const int* k3, const int* s3, const int* p3, int Kp,
                            const float* clip, void* rows, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && C > 0 && H > 0 && W > 0 && k3 && s3 && p3 && clip && rows, VTX_EINVAL, "im2col3d: bad arguments");
  VTX_REQUIRE(Kp >= C * k3[0] * k3[1] * k3[2] && Kp % 8 == 0, VTX_EINVAL, "im2col3d: row width must cover C*kt*kh*kw and be a multiple of 8");
  const int To = (T + 2 * p3[0] - k3[0]) / s3[0] + 1, Ho = (H + 2 * p3[1] - k3[1]) / s3[1] + 1, Wo = (W + 2 * p3[2] - k3[2]) / s3[2] + 1;
  VTX_REQUIRE((long)B * To * Ho * Wo * (Kp / 8) < (1L << 31), VTX_EINVAL, "im2col3d: tensor too large for the 32-bit index arithmetic");
  const long total = (long)B * To * Ho * Wo * (Kp / 8);
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(im2col3d_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, T, C, H, W, k3[0], k3[1], k3[2], s3[0], s3[1], s3[2],
                       p3[0], p3[1], p3[2], To, Ho, Wo, Kp, clip, (float*)rows);
  else
    hipLaunchKernelGGL(im2col3d_kernel<bf16raw>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), B, T, C, H, W, k3[0], k3[1], k3[2], s3[0], s3[1], s3[2],
                       p3[0], p3[1], p3[2], To, Ho, Wo, Kp, clip, (bf16raw*)rows);
  return check_launch("im2col3d");
}

// xattn_mfma.hip: the bf16 MFMA kernels (head_dim 96 / 64)
namespace vtx {
bool xattn_mfma_eligible(int dtype, int hd);
int xattn_mfma_splits(int B, int Lq, int Lk, int heads);
int xattn_fwd_mfma_launch(const vtx_xattn_desc* d, hipStream_t st);
int xattn_bwd_mfma_launch(const vtx_xattn_desc* d, const void* dout, float* delta, void* dq, int nsplit, float* part_k, float* part_v,
                          long part_stride, hipStream_t st);
}
static bool xattn_use_mfma(const vtx_xattn_desc* d) { return xattn_mfma_eligible(d->dtype, d->hd) && !options().attn_valu; }

static int xattn_check(const vtx_xattn_desc* d, const char* who) {
  VTX_REQUIRE(d != nullptr, VTX_EINVAL, "%s: null descriptor", who);
  VTX_REQUIRE(d->dtype == VTX_F32 || d->dtype == VTX_BF16, VTX_EINVAL, "%s: bad dtype", who);
  VTX_REQUIRE(d->B > 0 && d->Lq > 0 && d->Lk > 0 && d->heads > 0, VTX_EINVAL, "%s: bad shape", who);
  VTX_REQUIRE(d->q && d->k && d->v && d->out && d->lse, VTX_EINVAL, "%s: null pointer", who);
  VTX_REQUIRE(aligned16(d->q) && aligned16(d->k) && aligned16(d->v) && aligned16(d->out), VTX_EALIGN, "%s: 16-byte alignment required", who);
  return VTX_OK;
}

extern "C" int vtx_xattn_fwd(const vtx_xattn_desc* d, void* stream) {
  int rc = xattn_check(d, "xattn_fwd");
  if (rc) return rc;
  dim3 g(cdiv(d->Lq, XA_THREADS), d->heads, d->B), blk(XA_THREADS);
  hipStream_t st = as_stream(stream);
  if (xattn_use_mfma(d)) return xattn_fwd_mfma_launch(d, st);
#define L_(T_, HD_) hipLaunchKernelGGL((xattn_fwd_kernel<T_, HD_>), g, blk, 0, st, d->Lq, d->Lk, d->heads, d->scale, (const T_*)d->q, (const T_*)d->k, \
                                       (const T_*)d->v, (T_*)d->out, d->lse)
  MV_DISPATCH(d->dtype, d->hd, L_(float, 96), L_(bf16raw, 96), L_(float, 64), L_(bf16raw, 64), "xattn_fwd");
#undef L_
  return check_launch("xattn_fwd");
}

static int xattn_splits(const vtx_xattn_desc* d) {
  const int kblocks = cdiv(d->Lk, XA_THREADS);
  const long blocks = (long)kblocks * d->heads * d->B;
  int s = (int)(1024 / (blocks > 0 ? blocks : 1));             // aim at ~4 workgroups per CU
  const int max_s = cdiv(d->Lq, 4 * XA_TILE);                  // at least 4 query tiles per split
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}

extern "C" size_t vtx_xattn_bwd_workspace(const vtx_xattn_desc* d) {
  if (!d) return 0;
  if (xattn_use_mfma(d))                        // the MFMA dk / dv kernel always leaves fp32 partials
    return (size_t)2 * xattn_mfma_splits(d->B, d->Lq, d->Lk, d->heads) * d->B * d->Lk * d->heads * d->hd * sizeof(float);
  const int s = xattn_splits(d);
  return s > 1 ? (size_t)2 * s * d->B * d->Lk * d->heads * d->hd * sizeof(float) : 16;
}

extern "C" int vtx_xattn_bwd(const vtx_xattn_desc* d, const void* dout, float* delta, void* dq, void* dk, void* dv, void* workspace,
                             size_t ws_bytes, void* stream) {
  int rc = xattn_check(d, "xattn_bwd");
  if (rc) return rc;
  VTX_REQUIRE(dout && delta && dq && dk && dv && workspace, VTX_EINVAL, "xattn_bwd: null pointer");
  VTX_REQUIRE(ws_bytes >= vtx_xattn_bwd_workspace(d), VTX_EWS, "xattn_bwd: workspace too small");
  hipStream_t st = as_stream(stream);
  if (xattn_use_mfma(d)) {
    const int ns = xattn_mfma_splits(d->B, d->Lq, d->Lk, d->heads);
    const long stride = (long)d->B * d->Lk * d->heads * d->hd;
    float* pk = (float*)workspace;
    float* pv = pk + (size_t)ns * stride;
    rc = xattn_bwd_mfma_launch(d, dout, delta, dq, ns, pk, pv, stride, st);
    if (rc) return rc;
    const long n8 = stride / 8;
    hipLaunchKernelGGL(xattn_reduce_kernel<bf16raw>, dim3(grid_for(n8)), dim3(256), 0, st, pk, ns, stride, n8, (bf16raw*)dk);
    hipLaunchKernelGGL(xattn_reduce_kernel<bf16raw>, dim3(grid_for(n8)), dim3(256), 0, st, pv, ns, stride, n8, (bf16raw*)dv);
    return check_launch("xattn_reduce");
  }
  const int kblocks = cdiv(d->Lk, XA_THREADS), nsplit = xattn_splits(d);
  const int q_per = cdiv(cdiv(d->Lq, nsplit), XA_TILE) * XA_TILE;
  const long part_stride = (long)d->B * d->Lk * d->heads * d->hd;
  float* part_k = nsplit > 1 ? (float*)workspace : nullptr;
  float* part_v = nsplit > 1 ? part_k + (size_t)nsplit * part_stride : nullptr;
  dim3 gq(cdiv(d->Lq, XA_THREADS), d->heads, d->B), gk(kblocks * nsplit, d->heads, d->B), blk(XA_THREADS);
#define Q_(T_, HD_) hipLaunchKernelGGL((xattn_bwd_dq_kernel<T_, HD_>), gq, blk, 0, st, d->Lq, d->Lk, d->heads, d->scale, (const T_*)d->q, (const T_*)d->k, \
                                       (const T_*)d->v, (const T_*)d->out, (const T_*)dout, d->lse, delta, (T_*)dq)
  MV_DISPATCH(d->dtype, d->hd, Q_(float, 96), Q_(bf16raw, 96), Q_(float, 64), Q_(bf16raw, 64), "xattn_bwd");
#undef Q_
  rc = check_launch("xattn_bwd_dq");
  if (rc) return rc;
#define K_(T_, HD_) hipLaunchKernelGGL((xattn_bwd_dkv_kernel<T_, HD_>), gk, blk, 0, st, d->Lq, d->Lk, d->heads, d->scale, (const T_*)d->q, (const T_*)d->k, \
                                       (const T_*)d->v, (const T_*)dout, d->lse, delta, (T_*)dk, (T_*)dv, kblocks, q_per, part_k, part_v, part_stride)
  MV_DISPATCH(d->dtype, d->hd, K_(float, 96), K_(bf16raw, 96), K_(float, 64), K_(bf16raw, 64), "xattn_bwd");
#undef K_
  rc = check_launch("xattn_bwd_dkv");
  if (rc || nsplit == 1) return rc;
  const long n8 = part_stride / 8;
  if (d->dtype == VTX_F32) {
    hipLaunchKernelGGL(xattn_reduce_kernel<float>, dim3(grid_for(n8)), dim3(256), 0, st, part_k, nsplit, part_stride, n8, (float*)dk);
    hipLaunchKernelGGL(xattn_reduce_kernel<float>, dim3(grid_for(n8)), dim3(256), 0, st, part_v, nsplit, part_stride, n8, (float*)dv);
  } else {
    hipLaunchKernelGGL(xattn_reduce_kernel<bf16raw>, dim3(grid_for(n8)), dim3(256), 0, st, part_k, nsplit, part_stride, n8, (bf16raw*)dk);
    hipLaunchKernelGGL(xattn_reduce_kernel<bf16raw>, dim3(grid_for(n8)), dim3(256), 0, st, part_v, nsplit, part_stride, n8, (bf16raw*)dv);
  }
  return check_launch("xattn_reduce");
}
