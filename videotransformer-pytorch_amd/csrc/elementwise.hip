// elementwise.hip -- HBM-bound glue kernels of the divided space-time block:
// weight staging (cast + transpose), cls-token mean / replication gradients,
// DropPath row scaling, strided row reductions (embedding gradients), the patch
// gather of PatchEmbed and the positional/time embedding table.
// All of them move each byte once with 16-byte vector accesses.
#include "common.h"

namespace vtx {

// ---- W fp32 [R,C] -> Wc (T) [R,C] and WcT (T) [C,R] via a 64x64 LDS tile -------
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(int R, int C, const float* __restrict__ W,
                                                             T* __restrict__ Wc, T* __restrict__ WcT) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;     // 64 x 4
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) {
      v = W[(long)r * C + c];
      if (Wc) ET<T>::st(Wc + (long)r * C + c, v);
    }
    tile[rr][tx] = v;
  }
  if (!WcT) return;
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, r = r0 + tx;
    if (r < R && c < C) ET<T>::st(WcT + (long)c * R + r, tile[tx][cc]);
  }
}

// ---- the same over a table of matrices: block b -> (tensor, 64x64 tile) by binary search in the tile prefix sums.
// 16-byte loads and 8-byte stores where the shape allows (rows, cols multiples of 4), element-wise otherwise.
template <typename T>
__global__ __launch_bounds__(256) void mt_cast_transpose_kernel(const vtx_ct_tensor* __restrict__ tab, const int* __restrict__ tile_start,
                                                                int n_tensors) {
  __shared__ float tile[64][65];
  int lo = 0, hi = n_tensors;                  // tile_start[lo] <= b < tile_start[hi]
  const int b = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= b) lo = mid; else hi = mid;
  }
  const vtx_ct_tensor t = tab[lo];
  const int R = t.rows, C = t.cols;
  const int l = b - tile_start[lo], tiles_c = (C + 63) >> 6;
  const int r0 = (l / tiles_c) * 64, c0 = (l % tiles_c) * 64;
  T* Wc = reinterpret_cast<T*>(t.dst_c);
  T* WcT = reinterpret_cast<T*>(t.dst_t);
  const bool vec = (R % 4 == 0) && (C % 4 == 0);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 x 16: four columns x (rows ty, ty+16, ...)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = ty + 16 * i, r = r0 + rr, c = c0 + 4 * tx;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      if (vec && c < C) {
        const float4 a = *reinterpret_cast<const float4*>(t.src + (long)r * C + c);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        if (Wc) ET<T>::st4(Wc + (long)r * C + c, v);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < C) {
            v[j] = t.src[(long)r * C + c + j];
            if (Wc) ET<T>::st(Wc + (long)r * C + c + j, v[j]);
          }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[rr][4 * tx + j] = v[j];
  }
  if (!WcT) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cc = ty + 16 * i, c = c0 + cc, r = r0 + 4 * tx;      // four consecutive rows of column c
    if (c < C) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = tile[4 * tx + j][cc];
      if (vec && r < R) {
        ET<T>::st4(WcT + (long)c * R + r, v);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (r + j < R) ET<T>::st(WcT + (long)c * R + r + j, v[j]);
      }
    }
  }
}

template <typename T>
__global__ void cast_from_f32_kernel(size_t n, const float* __restrict__ src, T* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    ET<T>::st(dst + i, src[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(size_t n, const T* __restrict__ src, float* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = ET<T>::ld(src + i);
}

// ---- out[b,0,:] = x[b,0,:] + mean_t a_cls[b*T+t,:] ------------------------------
template <typename T>
__global__ void cls_mean_fwd_kernel(int B, int T_, int D, const T* __restrict__ a_cls, long lda, const T* __restrict__ x,
                                    T* __restrict__ out, long ld, long rows_per_clip) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per 8 columns
  const int per = D / 8;
  if (idx >= B * per) return;
  const int b = idx / per, c = (idx - b * per) * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = 0; t < T_; ++t) {
    float v[8];
    load8(a_cls + (long)(b * T_ + t) * lda + c, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += v[j];
  }
  float xv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (x != nullptr) load8(x + (long)b * rows_per_clip * ld + c, xv);     // x == nullptr: the contribution alone (exact residual stream)
  const float inv = 1.0f / (float)T_;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = xv[j] + acc[j] * inv;
  store8(out + (long)b * rows_per_clip * ld + c, acc);
}

// ---- ViViT fact_encoder: glue between the spatial and the temporal encoder (reference video_transformer.py:511-525) ----
// x [(b t), 1 + P, D] -> h [b, 1 + T, D]:  h[i, 0] = x[i, 0] + e[0]   (the reference reads `x[:b, 0]` of the FLATTENED (b t) axis:
// row i < b, i.e. frame i % T of clip i / T -- kept literal), h[i, 1 + t] = mean_p x[i T + t, 1 + p] + e[1 + t].
// One workgroup per output row; the patch rows are summed in fp32 in two interleaved halves (p even / odd, fixed order) by
// 16-byte loads.  e = time_embed [1 + T, D] fp32.
template <typename T>
__global__ __launch_bounds__(256) void fact_glue_fwd_kernel(int B, int T_, int P, int D, const T* __restrict__ x, const float* __restrict__ e,
                                                            T* __restrict__ h) {
  __shared__ float part[2][2048];
  const int row = blockIdx.x;                       // output row i * (1 + T) + s
  const int i = row / (1 + T_), sidx = row - i * (1 + T_);
  const int per = D / 8, half = threadIdx.x / per, cc = (threadIdx.x - half * per) * 8;
  const long rs = (long)(1 + P) * D;                // elements per (b t) row block of x
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (half < 2) {
    if (sidx == 0) {
      if (half == 0) load8(x + (long)i * rs + cc, acc);
    } else {
      const T* src = x + ((long)i * T_ + (sidx - 1)) * rs + D + cc;      // patch 0 of frame (i, s - 1)
      for (int p = half; p < P; p += 2) {
        float v[8];
        load8(src + (long)p * D, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[half][cc + j] = acc[j];
  }
  __syncthreads();
  if (half == 0) {
    float o[8];
    const float inv = sidx == 0 ? 1.0f : 1.0f / (float)P;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (part[0][cc + j] + part[1][cc + j]) * inv + e[(long)sidx * D + cc + j];
    store8(h + (long)row * D + cc, o);
  }
}
// backward: dx[(i T + t), 1 + p] = dh[i, 1 + t] / P;  dx[r, 0] = r < B ? dh[r, 0] : 0;  de[s] (+)= sum_i dh[i, s] (fp32, i ascending)
template <typename T>
__global__ __launch_bounds__(256) void fact_glue_bwd_kernel(int B, int T_, int P, int D, const T* __restrict__ dh, T* __restrict__ dx) {
  const long per = D / 8;
  const long total = (long)B * T_ * (1 + P) * per;
  const float inv = 1.0f / (float)P;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / per;                       // row of x: (frame f = i T + t, token q)
    const int cc = (int)(idx - r * per) * 8;
    const long f = r / (1 + P);
    const int q = (int)(r - f * (1 + P));
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (q == 0) {
      if (f < B) load8(dh + f * (1 + T_) * D + cc, v);
    } else {
      const long i = f / T_, t = f - i * T_;
      load8(dh + (i * (1 + T_) + 1 + t) * D + cc, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= inv;
    }
    store8(dx + r * D + cc, v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void fact_glue_de_kernel(int B, int T_, int D, const T* __restrict__ dh, float* __restrict__ de, int accumulate) {
  const long n = (long)(1 + T_) * D;
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  float a = 0.f;
  for (int i = 0; i < B; ++i) a += ET<T>::ld(dh + (long)i * n + k);
  de[k] = accumulate ? de[k] + a : a;
}

// ---- da rows for the spatial projection backward ----------------------------------
template <typename T>
__global__ void space_grad_prep_kernel(int B, int T_, int P, int D, const T* __restrict__ dout, long ld,
                                       const float* __restrict__ s, T* __restrict__ da, long ldda) {
  const long per = D / 8;
  const long N = (long)P * T_;
  const long total = ((long)B * N + (long)B * T_) * per;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / per;
    const int c = (int)(idx - row * per) * 8;
    float v[8];
    float sc;
    if (row < (long)B * N) {
      const long b = row / N, n = row - b * N;
      load8(dout + (b * (N + 1) + 1 + n) * ld + c, v);
      sc = s ? s[b * T_ + (n % T_)] : 1.0f;
    } else {
      const long bt = row - (long)B * N;
      const long b = bt / T_;
      load8(dout + (b * (N + 1)) * ld + c, v);
      sc = (s ? s[bt] : 1.0f) / (float)T_;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= sc;
    store8(da + row * ldda + c, v);
  }
}

// ---- dqkv[b,0,:] = sum_t dqkv_cls[b*T+t,:] ----------------------------------------
template <typename T>
__global__ void cls_qkv_reduce_kernel(int B, int T_, int W, const T* __restrict__ src, long lds_, T* __restrict__ dst,
                                      long ld, long rows_per_clip) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int per = W / 8;
  if (idx >= B * per) return;
  const int b = idx / per, c = (idx - b * per) * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = 0; t < T_; ++t) {
    float v[8];
    load8(src + (long)(b * T_ + t) * lds_ + c, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += v[j];
  }
  store8(dst + (long)b * rows_per_clip * ld + c, acc);
}

// ---- dst[dmap(m)] = src[smap(m)] * s[idx(m)] ----------------------------------------
template <typename T>
__global__ void row_scale_copy_kernel(int rows, int D, const T* __restrict__ src, long lds_, vtx_rowmap smap,
                                      T* __restrict__ dst, long ldd, vtx_rowmap dmap, const float* __restrict__ s,
                                      int d1, int m1, int d2, int m2) {
  const long per = D / 8;
  const long total = (long)rows * per;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / per;
    const int c = (int)(idx - m * per) * 8;
    float v[8];
    load8(src + map_row(smap, m) * lds_ + c, v);
    if (s) {
      const float sc = s[(m / d1) * m1 + (m % d2) * m2];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= sc;
    }
    store8(dst + map_row(dmap, m) * ldd + c, v);
  }
}

// ---- out[j,:] (+)= scale * sum_i in[base + i*si + j*sj, :] -----------------------------
// grid.x = column chunks, grid.y = j.  256 threads = 32 column lanes (x4 cols) x 8 row lanes.
template <typename T>
__global__ __launch_bounds__(256) void reduce_rows_kernel(int ni, int D, const T* __restrict__ in, long ld, long base,
                                                          long si, long sj, float* __restrict__ out, long ldo,
                                                          float scale, int accumulate) {
  __shared__ float red[8][128 + 4];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 128 + cl * 4;
  const long j = blockIdx.y;
  float a[4] = {0, 0, 0, 0};
  if (c < D) {
    for (int i = rl; i < ni; i += 8) {
      const T* p = in + (base + (long)i * si + j * sj) * ld + c;
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += ET<T>::ld(p + e);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][cl * 4 + e] = a[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int col = blockIdx.x * 128 + threadIdx.x;
    if (col < D) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += red[r][threadIdx.x];
      t *= scale;
      float* o = out + j * ldo + col;
      *o = accumulate ? *o + t : t;
    }
  }
}

// ---- patch gather: clip [B,T,C,H,W] fp32 -> rows [(b,p,t') or (b,t',p)][K] -----------
// One thread per (row, c, kt, kh) run of ps contiguous pixels (ps = 16 -> 64 B read).
template <typename T>
__global__ void patch_rows_kernel(int B, int Tn, int C, int H, int W, int ps, int ts, const float* __restrict__ clip,
                                  T* __restrict__ rows, long ldr, int frame_major) {
  const int gh = H / ps, gw = W / ps, P = gh * gw, Tq = Tn / ts;
  const long runs_per_row = (long)C * ts * ps;                 // (c, kt, kh)
  const long total = (long)B * Tq * P * runs_per_row;
  const bool vec8 = (ps & 7) == 0 && (ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(rows) & 15u) == 0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    // idx -> (b, tq, ph, c, kt, kh, pw): pw fastest so that neighbouring threads read neighbouring pixels
    long r = idx;
    const int pw = (int)(r % gw); r /= gw;
    const int kh = (int)(r % ps); r /= ps;
    const int kt = (int)(r % ts); r /= ts;
    const int c = (int)(r % C); r /= C;
    const int ph = (int)(r % gh); r /= gh;
    const int tq = (int)(r % Tq); r /= Tq;
    const int b = (int)r;
    const float* src = clip + ((((long)b * Tn + (tq * ts + kt)) * C + c) * H + (ph * ps + kh)) * W + pw * ps;
    const int p = ph * gw + pw;
    const long row = frame_major ? ((long)b * Tq + tq) * P + p : ((long)b * P + p) * Tq + tq;
    T* dst = rows + row * ldr + ((long)(c * ts + kt) * ps + kh) * ps;
    if (vec8) {
      // eight pixels per store: one 16-byte (bf16) or two 16-byte (fp32) stores instead of eight 2- / 4-byte ones (round 5:
      // the run's 64 B were read with four loads and written with sixteen 2-byte stores, 2.3 TB/s)
      for (int kw = 0; kw < ps; kw += 8) {
        float v[8];
        load8(src + kw, v);
        store8(dst + kw, v);
      }
    } else {
      for (int kw = 0; kw < ps; kw += 4) {
        const float4 v = *reinterpret_cast<const float4*>(src + kw);
        ET<T>::st(dst + kw, v.x); ET<T>::st(dst + kw + 1, v.y); ET<T>::st(dst + kw + 2, v.z); ET<T>::st(dst + kw + 3, v.w);
      }
    }
  }
}

// ---- patch gather straight from decoded video: clip [B,T,H,W,3] uint8 (channels last) --------------
// Fuses the reference's ToTensor (x.float().div(255), data_transform.py:52-63) and
// transforms.Normalize ((x - mean[c]) / std[c], data_transform.py:534-539) into the gather: the three
// IEEE operations in the reference's order, so the fp32 rows are bit-identical to patch_rows() of the
// normalised fp32 clip, and the clip crosses HBM once as 1 byte per sample instead of 4 + 4 + 4.
// One thread per (row, kt, kh): reads ps pixels x 3 channels = 3*ps contiguous bytes (48 B at ps = 16).
template <typename T>
__global__ void patch_rows_u8_kernel(int B, int Tn, int H, int W, int ps, int ts, const uint8_t* __restrict__ clip,
                                     float m0, float m1, float m2, float s0, float s1, float s2,
                                     T* __restrict__ rows, long ldr, int frame_major) {
#pragma clang fp contract(off)
  const int gh = H / ps, gw = W / ps, P = gh * gw, Tq = Tn / ts;
  const long total = (long)B * Tq * P * ts * ps;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;                                   // idx -> (b, tq, ph, kt, kh, pw): pw fastest
    const int pw = (int)(r % gw); r /= gw;
    const int kh = (int)(r % ps); r /= ps;
    const int kt = (int)(r % ts); r /= ts;
    const int ph = (int)(r % gh); r /= gh;
    const int tq = (int)(r % Tq); r /= Tq;
    const int b = (int)r;
    const uint8_t* src = clip + ((((long)b * Tn + (tq * ts + kt)) * H + (ph * ps + kh)) * W + pw * ps) * 3;
    const int p = ph * gw + pw;
    const long row = frame_major ? ((long)b * Tq + tq) * P + p : ((long)b * P + p) * Tq + tq;
    T* dst = rows + row * ldr + ((long)kt * ps + kh) * ps;
    for (int kw = 0; kw < ps; kw += 4) {            // 12 bytes = 4 pixels x 3 channels
      const uint32_t w0 = *reinterpret_cast<const uint32_t*>(src + kw * 3);
      const uint32_t w1 = *reinterpret_cast<const uint32_t*>(src + kw * 3 + 4);
      const uint32_t w2 = *reinterpret_cast<const uint32_t*>(src + kw * 3 + 8);
      const uint32_t by[12] = {w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255, w0 >> 24, w1 & 255, (w1 >> 8) & 255,
                               (w1 >> 16) & 255, w1 >> 24, w2 & 255, (w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24};
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = (float)by[3 * j + c] / 255.0f;
          ET<T>::st(dst + (long)c * ts * ps * ps + kw + j, (x - mean[c]) / sd[c]);
        }
    }
  }
}

// ---- E[(p,t) or (t,p)] = bias + pos[1+p] + time[t];  cls_row = cls + pos[0] -------------
template <typename T>
__global__ void embed_table_kernel(int P, int Tn, int D, const float* __restrict__ bias, const float* __restrict__ pos,
                                   const float* __restrict__ time_embed, const float* __restrict__ cls,
                                   T* __restrict__ E, T* __restrict__ cls_row, int frame_major) {
  const long total = ((long)P * Tn + 1) * D;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / D;
    const int d = (int)(idx - row * D);
    if (row == (long)P * Tn) {
      if (cls_row) ET<T>::st(cls_row + d, cls[d] + pos[d]);
      continue;
    }
    int p, t;
    if (frame_major) { t = (int)(row / P); p = (int)(row - (long)t * P); }
    else { p = (int)(row / Tn); t = (int)(row - (long)p * Tn); }
    float v = pos[(long)(1 + p) * D + d];
    if (bias) v += bias[d];
    if (time_embed) v += time_embed[(long)t * D + d];
    ET<T>::st(E + row * D + d, v);
  }
}

static inline int grid_for(long work, int block, int cap = 4096) {
  long g = (work + block - 1) / block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}


// ---- rows of dropped DropPath groups (s[m / group_rows] == 0) --------------------------------------
// The merged attn.proj + temporal_fc GEMM (vtx/functions.py TimeAttnFn) computes s * (acc + b') + x; for a dropped
// group the reference's value is x + b_tfc (transformer.py:268-275: the DropPath sits BETWEEN the two Linear layers),
// and the group's rows must not reach the merged weight gradient.  One wave per group; a kept group's wave exits.
template <typename T>
__global__ __launch_bounds__(256) void dropped_rows_fix_kernel(long M, int D, int group_rows, const float* __restrict__ s,
                                                               const T* __restrict__ x, long ldx, vtx_rowmap xmap,
                                                               const float* __restrict__ bias, T* __restrict__ out, long ldo,
                                                               vtx_rowmap omap, T* __restrict__ zero, long ldz) {
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long m0 = g * group_rows;
  if (m0 >= M || s[g] != 0.f) return;
  const int lane = threadIdx.x & 63;
  const float z8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < group_rows && m0 + r < M; ++r) {
    const long m = m0 + r;
    for (int c = lane * 8; c < D; c += 512) {
      if (out) {
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8];
        if (x != nullptr) load8(x + map_row(xmap, m) * ldx + c, v);      // x == nullptr: the row's value is the bias alone
        load8(bias + c, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += b[j];
        store8(out + map_row(omap, m) * ldo + c, v);
      }
      if (zero) store8(zero + m * ldz + c, z8);
    }
  }
}

// part[w, :] = sum of src[smap(m), :] over the rows m of the dropped groups among block w's run of consecutive groups
// (fixed partition, fixed order: deterministic; the caller folds the nparts rows with vtx_reduce_rows).
// grid = (column chunks of 128, nparts); 256 threads = 32 column lanes (x4 columns) x 8 row lanes.
template <typename T>
__global__ __launch_bounds__(256) void dropped_rows_colsum_kernel(long M, int D, int group_rows, const float* __restrict__ s,
                                                                  const T* __restrict__ src, long lds_, vtx_rowmap smap,
                                                                  float* __restrict__ part, long n_groups) {
  __shared__ float red[8][128 + 4];
  __shared__ float flag[256];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 128 + cl * 4;
  const long per = (n_groups + gridDim.y - 1) / gridDim.y;
  const long g0 = (long)blockIdx.y * per, g1 = g0 + per < n_groups ? g0 + per : n_groups;
  float a[4] = {0, 0, 0, 0};
  // the scales of 256 groups per coalesced load (one dependent global load per group made the scan the whole cost)
  for (long gb = g0; gb < g1; gb += 256) {
    __syncthreads();
    flag[threadIdx.x] = gb + threadIdx.x < g1 ? s[gb + threadIdx.x] : 1.f;
    __syncthreads();
    const int n = g1 - gb < 256 ? (int)(g1 - gb) : 256;
    for (int i = 0; i < n; ++i) {
      if (flag[i] != 0.f) continue;                // uniform over the block
      if (c < D)
        for (int r = rl; r < group_rows; r += 8) {
          const long m = (gb + i) * group_rows + r;
          if (m >= M) break;
          const T* p = src + map_row(smap, m) * lds_ + c;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] += ET<T>::ld(p + e);
        }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][cl * 4 + e] = a[e];
  __syncthreads();
  if (rl == 0 && c < D) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) t += red[i][cl * 4 + e];
      part[(long)blockIdx.y * D + c + e] = t;
    }
  }
}

}  // namespace vtx

using namespace vtx;

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16, name)              \
  if ((dtype) == VTX_F32) { CALL_F32; }                           \
  else if ((dtype) == VTX_BF16) { CALL_BF16; }                    \
  else VTX_REQUIRE(false, VTX_EINVAL, name ": bad dtype %d", (int)(dtype))

// out = dy * gelu'(h)  (backward of a GELU whose consumer is not a GEMM epilogue)
template <typename T>
__global__ __launch_bounds__(256) void gelu_grad_mul_kernel(size_t n8, const T* __restrict__ dy, const T* __restrict__ h, T* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float a[8], b[8];
    load8(dy + i * 8, a);
    load8(h + i * 8, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] *= gelu_erf_grad(b[j]);
    store8(out + i * 8, a);
  }
}

extern "C" int vtx_gelu_grad_mul(int dtype, size_t n, const void* dy, const void* h, void* out, void* stream) {
  VTX_REQUIRE(n % 8 == 0 && dy && h && out && aligned16(dy) && aligned16(h) && aligned16(out), VTX_EINVAL,
              "gelu_grad_mul: n must be a multiple of 8 and the pointers 16-byte aligned");
  if (n == 0) return VTX_OK;
  const size_t n8 = n / 8;
  const int grid = (int)((n8 + 255) / 256 > 16384 ? 16384 : (n8 + 255) / 256);
  hipStream_t st = as_stream(stream);
  if (dtype == VTX_F32)
    hipLaunchKernelGGL(gelu_grad_mul_kernel<float>, dim3(grid), dim3(256), 0, st, n8, (const float*)dy, (const float*)h, (float*)out);
  else if (dtype == VTX_BF16)
    hipLaunchKernelGGL(gelu_grad_mul_kernel<bf16raw>, dim3(grid), dim3(256), 0, st, n8, (const bf16raw*)dy, (const bf16raw*)h, (bf16raw*)out);
  else
    VTX_REQUIRE(false, VTX_EINVAL, "gelu_grad_mul: bad dtype %d", dtype);
  return check_launch("gelu_grad_mul");
}

extern "C" int vtx_cast_transpose(int dtype, int R, int C, const float* W, void* Wc, void* WcT, void* stream) {
  VTX_REQUIRE(R > 0 && C > 0 && W && (Wc || WcT), VTX_EINVAL, "cast_transpose: bad arguments");
  dim3 grid(cdiv(C, 64), cdiv(R, 64)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, block, 0, st, R, C, W, (float*)Wc, (float*)WcT),
             hipLaunchKernelGGL(cast_transpose_kernel<bf16raw>, grid, block, 0, st, R, C, W, (bf16raw*)Wc, (bf16raw*)WcT),
             "cast_transpose");
  return check_launch("cast_transpose");
}

extern "C" int vtx_mt_cast_transpose(int dtype, const vtx_ct_tensor* tab, const int* tile_start, int n_tensors, int n_tiles,
                                     void* stream) {
  VTX_REQUIRE(tab && tile_start && n_tensors > 0 && n_tiles >= 0, VTX_EINVAL, "mt_cast_transpose: bad arguments");
  if (n_tiles == 0) return VTX_OK;
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(mt_cast_transpose_kernel<float>, dim3(n_tiles), dim3(256), 0, st, tab, tile_start, n_tensors),
             hipLaunchKernelGGL(mt_cast_transpose_kernel<bf16raw>, dim3(n_tiles), dim3(256), 0, st, tab, tile_start, n_tensors),
             "mt_cast_transpose");
  return check_launch("mt_cast_transpose");
}

extern "C" int vtx_cast_from_f32(int dtype, size_t n, const float* src, void* dst, void* stream) {
  VTX_REQUIRE(src && dst, VTX_EINVAL, "cast_from_f32: null pointer");
  if (n == 0) return VTX_OK;
  dim3 grid(grid_for((long)n, 256)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_from_f32_kernel<float>, grid, block, 0, st, n, src, (float*)dst),
             hipLaunchKernelGGL(cast_from_f32_kernel<bf16raw>, grid, block, 0, st, n, src, (bf16raw*)dst), "cast_from_f32");
  return check_launch("cast_from_f32");
}

extern "C" int vtx_cast_to_f32(int dtype, size_t n, const void* src, float* dst, void* stream) {
  VTX_REQUIRE(src && dst, VTX_EINVAL, "cast_to_f32: null pointer");
  if (n == 0) return VTX_OK;
  dim3 grid(grid_for((long)n, 256)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_to_f32_kernel<float>, grid, block, 0, st, n, (const float*)src, dst),
             hipLaunchKernelGGL(cast_to_f32_kernel<bf16raw>, grid, block, 0, st, n, (const bf16raw*)src, dst), "cast_to_f32");
  return check_launch("cast_to_f32");
}

extern "C" int vtx_cls_mean_fwd(int dtype, int B, int T, int D, const void* a_cls, long lda, const void* x, void* out,
                                long ld_tok, long rows_per_clip, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && D > 0 && D % 8 == 0 && a_cls && out, VTX_EINVAL, "cls_mean_fwd: bad arguments");
  dim3 grid(cdiv((long)B * D / 8, 256)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(cls_mean_fwd_kernel<float>, grid, block, 0, st, B, T, D, (const float*)a_cls, lda, (const float*)x, (float*)out, ld_tok, rows_per_clip),
             hipLaunchKernelGGL(cls_mean_fwd_kernel<bf16raw>, grid, block, 0, st, B, T, D, (const bf16raw*)a_cls, lda, (const bf16raw*)x, (bf16raw*)out, ld_tok, rows_per_clip),
             "cls_mean_fwd");
  return check_launch("cls_mean_fwd");
}

extern "C" int vtx_fact_glue_fwd(int dtype, int B, int T, int P, int D, const void* x, const float* time_embed, void* h, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && P > 0 && D > 0 && D % 8 == 0 && D <= 1024 && x && time_embed && h, VTX_EINVAL, "fact_glue_fwd: bad arguments");
  VTX_REQUIRE(aligned16(x) && aligned16(h), VTX_EALIGN, "fact_glue_fwd: 16-byte alignment required");
  dim3 grid(B * (1 + T)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(fact_glue_fwd_kernel<float>, grid, block, 0, st, B, T, P, D, (const float*)x, time_embed, (float*)h),
             hipLaunchKernelGGL(fact_glue_fwd_kernel<bf16raw>, grid, block, 0, st, B, T, P, D, (const bf16raw*)x, time_embed, (bf16raw*)h),
             "fact_glue_fwd");
  return check_launch("fact_glue_fwd");
}

extern "C" int vtx_fact_glue_bwd(int dtype, int B, int T, int P, int D, const void* dh, void* dx, float* d_time_embed, int accumulate,
                                 void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && P > 0 && D > 0 && D % 8 == 0 && dh && dx, VTX_EINVAL, "fact_glue_bwd: bad arguments");
  VTX_REQUIRE(aligned16(dh) && aligned16(dx), VTX_EALIGN, "fact_glue_bwd: 16-byte alignment required");
  const long total = (long)B * T * (1 + P) * (D / 8);
  dim3 grid(grid_for(total, 256)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(fact_glue_bwd_kernel<float>, grid, block, 0, st, B, T, P, D, (const float*)dh, (float*)dx),
             hipLaunchKernelGGL(fact_glue_bwd_kernel<bf16raw>, grid, block, 0, st, B, T, P, D, (const bf16raw*)dh, (bf16raw*)dx),
             "fact_glue_bwd");
  int rc = check_launch("fact_glue_bwd");
  if (rc || !d_time_embed) return rc;
  dim3 g2(cdiv((long)(1 + T) * D, 256));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(fact_glue_de_kernel<float>, g2, block, 0, st, B, T, D, (const float*)dh, d_time_embed, accumulate),
             hipLaunchKernelGGL(fact_glue_de_kernel<bf16raw>, g2, block, 0, st, B, T, D, (const bf16raw*)dh, d_time_embed, accumulate),
             "fact_glue_de");
  return check_launch("fact_glue_de");
}

extern "C" int vtx_space_grad_prep(int dtype, int B, int T, int P, int D, const void* dout, long ld, const float* s,
                                   void* da, long ldda, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && P > 0 && D % 8 == 0 && dout && da, VTX_EINVAL, "space_grad_prep: bad arguments");
  const long work = ((long)B * P * T + (long)B * T) * (D / 8);
  dim3 grid(grid_for(work, 256)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(space_grad_prep_kernel<float>, grid, block, 0, st, B, T, P, D, (const float*)dout, ld, s, (float*)da, ldda),
             hipLaunchKernelGGL(space_grad_prep_kernel<bf16raw>, grid, block, 0, st, B, T, P, D, (const bf16raw*)dout, ld, s, (bf16raw*)da, ldda),
             "space_grad_prep");
  return check_launch("space_grad_prep");
}

extern "C" int vtx_cls_qkv_reduce(int dtype, int B, int T, int W, const void* dqkv_cls, long ldc, void* dqkv, long ld,
                                  long rows_per_clip, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && W % 8 == 0 && dqkv_cls && dqkv, VTX_EINVAL, "cls_qkv_reduce: bad arguments");
  dim3 grid(cdiv((long)B * W / 8, 256)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(cls_qkv_reduce_kernel<float>, grid, block, 0, st, B, T, W, (const float*)dqkv_cls, ldc, (float*)dqkv, ld, rows_per_clip),
             hipLaunchKernelGGL(cls_qkv_reduce_kernel<bf16raw>, grid, block, 0, st, B, T, W, (const bf16raw*)dqkv_cls, ldc, (bf16raw*)dqkv, ld, rows_per_clip),
             "cls_qkv_reduce");
  return check_launch("cls_qkv_reduce");
}

extern "C" int vtx_row_scale_copy(int dtype, int rows, int D, const void* src, long lds, vtx_rowmap smap, void* dst,
                                  long ldd, vtx_rowmap dmap, const float* s, int rs_d1, int rs_m1, int rs_d2, int rs_m2,
                                  void* stream) {
  VTX_REQUIRE(rows > 0 && D > 0 && D % 8 == 0 && src && dst, VTX_EINVAL, "row_scale_copy: bad arguments");
  VTX_REQUIRE(!s || (rs_d1 > 0 && rs_d2 > 0), VTX_EINVAL, "row_scale_copy: divisors must be > 0");
  dim3 grid(grid_for((long)rows * (D / 8), 256)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(row_scale_copy_kernel<float>, grid, block, 0, st, rows, D, (const float*)src, lds, smap, (float*)dst, ldd, dmap, s, rs_d1, rs_m1, rs_d2, rs_m2),
             hipLaunchKernelGGL(row_scale_copy_kernel<bf16raw>, grid, block, 0, st, rows, D, (const bf16raw*)src, lds, smap, (bf16raw*)dst, ldd, dmap, s, rs_d1, rs_m1, rs_d2, rs_m2),
             "row_scale_copy");
  return check_launch("row_scale_copy");
}

extern "C" int vtx_dropped_rows_fix(int dtype, long M, int D, int group_rows, const float* s, const void* x, long ldx,
                                    vtx_rowmap xmap, const float* bias, void* out, long ldo, vtx_rowmap omap, void* zero,
                                    long ldz, void* stream) {
  VTX_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && group_rows > 0 && s, VTX_EINVAL, "dropped_rows_fix: bad arguments");
  VTX_REQUIRE(out || zero, VTX_EINVAL, "dropped_rows_fix: nothing to do");
  VTX_REQUIRE(!out || bias, VTX_EINVAL, "dropped_rows_fix: out needs the bias");
  const long groups = (M + group_rows - 1) / group_rows;
  dim3 grid((unsigned)((groups + 3) / 4)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(dropped_rows_fix_kernel<float>, grid, block, 0, st, M, D, group_rows, s, (const float*)x, ldx, xmap, bias, (float*)out, ldo, omap, (float*)zero, ldz),
             hipLaunchKernelGGL(dropped_rows_fix_kernel<bf16raw>, grid, block, 0, st, M, D, group_rows, s, (const bf16raw*)x, ldx, xmap, bias, (bf16raw*)out, ldo, omap, (bf16raw*)zero, ldz),
             "dropped_rows_fix");
  return check_launch("dropped_rows_fix");
}

extern "C" int vtx_dropped_rows_colsum(int dtype, long M, int D, int group_rows, const float* s, const void* src, long lds,
                                       vtx_rowmap smap, float* part, int nparts, void* stream) {
  VTX_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && group_rows > 0 && s && src && part && nparts > 0 && nparts <= 65535, VTX_EINVAL,
              "dropped_rows_colsum: bad arguments");
  const long groups = (M + group_rows - 1) / group_rows;
  dim3 grid((unsigned)((D + 127) / 128), (unsigned)nparts), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(dropped_rows_colsum_kernel<float>, grid, block, 0, st, M, D, group_rows, s, (const float*)src, lds, smap, part, groups),
             hipLaunchKernelGGL(dropped_rows_colsum_kernel<bf16raw>, grid, block, 0, st, M, D, group_rows, s, (const bf16raw*)src, lds, smap, part, groups),
             "dropped_rows_colsum");
  return check_launch("dropped_rows_colsum");
}

extern "C" int vtx_reduce_rows(int in_dtype, int nj, int ni, int D, const void* in, long ld, long base, long si, long sj,
                               float* out, long ldo, float scale, int accumulate, void* stream) {
  VTX_REQUIRE(nj > 0 && ni > 0 && D > 0 && D % 4 == 0 && in && out, VTX_EINVAL, "reduce_rows: bad arguments");
  dim3 grid(cdiv(D, 128), nj), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(in_dtype,
             hipLaunchKernelGGL(reduce_rows_kernel<float>, grid, block, 0, st, ni, D, (const float*)in, ld, base, si, sj, out, ldo, scale, accumulate),
             hipLaunchKernelGGL(reduce_rows_kernel<bf16raw>, grid, block, 0, st, ni, D, (const bf16raw*)in, ld, base, si, sj, out, ldo, scale, accumulate),
             "reduce_rows");
  return check_launch("reduce_rows");
}

extern "C" int vtx_patch_rows(int dtype, int B, int T, int C, int H, int W, int ps, int ts, const float* clip, void* rows,
                              long ldr, int frame_major, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && C > 0 && ps > 0 && ts > 0 && clip && rows, VTX_EINVAL, "patch_rows: bad arguments");
  VTX_REQUIRE(H % ps == 0 && W % ps == 0 && T % ts == 0 && ps % 4 == 0 && W % 4 == 0, VTX_EINVAL,
              "patch_rows: H,W must be multiples of ps (itself a multiple of 4), T of ts");
  const long work = (long)B * (T / ts) * (H / ps) * (W / ps) * C * ts * ps;
  dim3 grid(grid_for(work, 256, 16384)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(patch_rows_kernel<float>, grid, block, 0, st, B, T, C, H, W, ps, ts, clip, (float*)rows, ldr, frame_major),
             hipLaunchKernelGGL(patch_rows_kernel<bf16raw>, grid, block, 0, st, B, T, C, H, W, ps, ts, clip, (bf16raw*)rows, ldr, frame_major),
             "patch_rows");
  return check_launch("patch_rows");
}

extern "C" int vtx_patch_rows_u8(int dtype, int B, int T, int H, int W, int ps, int ts, const unsigned char* clip,
                                 const float* mean3, const float* std3, void* rows, long ldr, int frame_major, void* stream) {
  VTX_REQUIRE(B > 0 && T > 0 && ps > 0 && ts > 0 && clip && rows && mean3 && std3, VTX_EINVAL, "patch_rows_u8: bad arguments");
  VTX_REQUIRE(H % ps == 0 && W % ps == 0 && T % ts == 0 && ps % 4 == 0, VTX_EINVAL,
              "patch_rows_u8: H,W must be multiples of ps (itself a multiple of 4), T of ts");
  VTX_REQUIRE((reinterpret_cast<uintptr_t>(clip) & 3u) == 0, VTX_EALIGN, "patch_rows_u8: clip must be 4-byte aligned");
  VTX_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, VTX_EINVAL, "patch_rows_u8: zero std");
  const long work = (long)B * (T / ts) * (H / ps) * (W / ps) * ts * ps;
  dim3 grid(grid_for(work, 256, 16384)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(patch_rows_u8_kernel<float>, grid, block, 0, st, B, T, H, W, ps, ts, clip, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (float*)rows, ldr, frame_major),
             hipLaunchKernelGGL(patch_rows_u8_kernel<bf16raw>, grid, block, 0, st, B, T, H, W, ps, ts, clip, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (bf16raw*)rows, ldr, frame_major),
             "patch_rows_u8");
  return check_launch("patch_rows_u8");
}

extern "C" int vtx_embed_table(int dtype, int P, int T, int D, const float* bias, const float* pos, const float* time_embed,
                               const float* cls, void* E, void* cls_row, int frame_major, void* stream) {
  VTX_REQUIRE(P > 0 && T > 0 && D > 0 && pos && E, VTX_EINVAL, "embed_table: bad arguments");
  VTX_REQUIRE(!cls_row || cls, VTX_EINVAL, "embed_table: cls_row needs cls");
  dim3 grid(grid_for(((long)P * T + 1) * D, 256)), block(256);
  hipStream_t st = as_stream(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(embed_table_kernel<float>, grid, block, 0, st, P, T, D, bias, pos, time_embed, cls, (float*)E, (float*)cls_row, frame_major),
             hipLaunchKernelGGL(embed_table_kernel<bf16raw>, grid, block, 0, st, P, T, D, bias, pos, time_embed, cls, (bf16raw*)E, (bf16raw*)cls_row, frame_major),
             "embed_table");
  return check_launch("embed_table");
}
