// head.hip -- the steps either side of the backbone in a supervised training step:
// on-device Mixup / CutMix of the clip batch with soft targets (before), and the classification
// loss + top-k accuracy on the logits (after).  SURVEY.md section 8(f) ranks 3 and 4.
//
// Reference:
//   mixup.py:102-126      Mixup._mix_batch / __call__: batch-mode mixup  x = x*lam + x.flip(0)*(1-lam)  or CutMix
//                         x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]  on [B, T*C, H, W] clips; the
//                         random draws (numpy) stay on the host, in the reference's order (mixup.py: rows 81-100, 33-49)
//   mixup.py:16-25        one_hot / mixup_target: label-smoothed one-hot rows mixed with the flipped batch's
//   model_trainer.py:85-91,207-215  loss = SoftTargetCrossEntropy (timm: mean_b sum_c -t log_softmax(x)) with mixup,
//                         nn.CrossEntropyLoss otherwise; torchmetrics Accuracy(top_k) on softmax(preds)
// All HBM-bound, one pass each; arithmetic is ordered as the reference's ATen calls round it (separate fp32
// multiply, multiply, add -- no fma contraction), so mixed clips and targets are bit-identical.
#include "common.h"

namespace vtx {

// x[b] <- x[b]*lam + x[B-1-b]*oml for every b (pairs swapped in one pass; B even)
__global__ __launch_bounds__(256) void mixup_kernel(float* __restrict__ x, int half_b, long per_clip, int B, float lam, float oml) {
#pragma clang fp contract(off)
  const long n4 = per_clip / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)half_b * n4; i += (long)gridDim.x * 256) {
    const long b = i / n4, o = (i - b * n4) * 4;
    float4* pa = reinterpret_cast<float4*>(x + b * per_clip + o);
    float4* pc = reinterpret_cast<float4*>(x + (long)(B - 1 - b) * per_clip + o);
    const float4 a = *pa, c = *pc;
    float4 ra, rc;
    ra.x = a.x * lam + c.x * oml; ra.y = a.y * lam + c.y * oml; ra.z = a.z * lam + c.z * oml; ra.w = a.w * lam + c.w * oml;
    rc.x = c.x * lam + a.x * oml; rc.y = c.y * lam + a.y * oml; rc.z = c.z * lam + a.z * oml; rc.w = c.w * lam + a.w * oml;
    *pa = ra;
    *pc = rc;
  }
}

// swap the box [yl,yh) x [xl,xh) of every plane between clips b and B-1-b
__global__ __launch_bounds__(256) void cutmix_kernel(float* __restrict__ x, int half_b, int planes, int H, int W, int B,
                                                     int yl, int yh, int xl, int xh) {
  const int bw = xh - xl, bh = yh - yl;
  const long box = (long)bw * bh, total = (long)half_b * planes * box;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long bp = i / box, r = i - bp * box;
    const int b = (int)(bp / planes), p = (int)(bp - (long)b * planes);
    const int yy = yl + (int)(r / bw), xx = xl + (int)(r % bw);
    float* pa = x + (((long)b * planes + p) * H + yy) * W + xx;
    float* pc = x + (((long)(B - 1 - b) * planes + p) * H + yy) * W + xx;
    const float a = *pa;
    *pa = *pc;
    *pc = a;
  }
}

__global__ __launch_bounds__(256) void mixup_target_kernel(const long* __restrict__ labels, int B, int C, float on, float off,
                                                           float lam, float oml, float* __restrict__ out) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)B * C; i += (long)gridDim.x * 256) {
    const int b = (int)(i / C), c = (int)(i - (long)b * C);
    const float v1 = labels[b] == c ? on : off, v2 = labels[B - 1 - b] == c ? on : off;
    out[i] = v1 * lam + v2 * oml;
  }
}

// ---- softmax cross-entropy: one wave per row ------------------------------------------------
// loss_row[b] = sum_c -t[b][c] * log_softmax(x[b])[c]   (soft targets)   or   -log_softmax(x[b])[label[b]]
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ x, const float* __restrict__ soft,
                                                       const long* __restrict__ labels, int B, int C, float* __restrict__ loss_row,
                                                       float* __restrict__ lse_out) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* xr = x + (long)row * C;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, xr[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(xr[c] - mx);
  const float lse = mx + logf(wave_sum(se));
  float acc = 0.f;
  if (soft) {
    const float* tr = soft + (long)row * C;
    for (int c = lane; c < C; c += 64) acc += tr[c] * (lse - xr[c]);
    acc = wave_sum(acc);
  } else {
    const long lab = labels[row];                // outside [0, C) (nn.CrossEntropyLoss's ignore_index = -100 included): not counted
    acc = (lab >= 0 && lab < C) ? lse - xr[lab] : 0.f;
  }
  if (lane == 0) { loss_row[row] = acc; lse_out[row] = lse; }
}

// out[0] = mean of the counted rows of rows[0..B) in a fixed order (one workgroup), out[1] = their number: with labels,
// rows whose label lies outside [0, C) are not counted (torch's mean over the non-ignored rows; all ignored: 0 / 0 = nan)
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ rows, const long* __restrict__ labels, int B, int C,
                                                        float* __restrict__ out) {
  __shared__ float red[256];
  __shared__ int cnt[256];
  float a = 0.f;
  int n = 0;
  for (int i = threadIdx.x; i < B; i += 256) {
    a += rows[i];
    n += labels ? (labels[i] >= 0 && labels[i] < C) : 1;
  }
  red[threadIdx.x] = a;
  cnt[threadIdx.x] = n;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { red[threadIdx.x] += red[threadIdx.x + s]; cnt[threadIdx.x] += cnt[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = red[0] / (float)cnt[0]; out[1] = (float)cnt[0]; }
}

// dx[b][c] = g * (softmax(x[b])[c] * sum_c t[b][c] - t[b][c]),  g = dloss / B  (mean reduction)
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ x, const float* __restrict__ soft,
                                                       const long* __restrict__ labels, const float* __restrict__ lse, int B, int C,
                                                       float g, const float* __restrict__ gdev, const float* __restrict__ count,
                                                       float* __restrict__ dx) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  if (gdev) g *= gdev[0];                       // upstream gradient of the scalar loss, read on the device (no host sync)
  if (count) g /= count[0];                     // mean over the counted rows (forward's loss_mean[1])
  const float* xr = x + (long)row * C;
  float ts = 1.f;
  if (soft) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += soft[(long)row * C + c];
    ts = wave_sum(a);
  }
  const float l = lse[row];
  const long lab = soft ? -1 : labels[row];
  if (!soft && (lab < 0 || lab >= C)) {         // not counted: no gradient
    for (int c = lane; c < C; c += 64) dx[(long)row * C + c] = 0.f;
    return;
  }
  for (int c = lane; c < C; c += 64) {
    const float t = soft ? soft[(long)row * C + c] : (c == lab ? 1.f : 0.f);
    dx[(long)row * C + c] = g * (expf(xr[c] - l) * ts - t);
  }
}

// correct[0] += #rows whose label is among the k largest entries (ties: entries strictly greater count first, then
// equal entries at a smaller index -- the order torch.topk breaks ties in)
__global__ __launch_bounds__(256) void topk_correct_kernel(const float* __restrict__ x, const long* __restrict__ labels, int B, int C,
                                                           int k, int* __restrict__ correct) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* xr = x + (long)row * C;
  const long lab = labels[row];
  if (lab < 0 || lab >= C) return;              // no such class: never correct
  const float v = xr[lab];
  int ahead = 0;
  for (int c = lane; c < C; c += 64) ahead += (xr[c] > v) || (xr[c] == v && c < lab);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ahead += __shfl_xor(ahead, o, 64);
  if (lane == 0 && ahead < k) atomicAdd(correct, 1);
}

}  // namespace vtx

using namespace vtx;

static int grid_for(long work) { long g = (work + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

extern "C" int vtx_mixup_batch(float* x, int B, long per_clip, float lam, float one_minus_lam, void* stream) {
  VTX_REQUIRE(x && B > 0 && B % 2 == 0 && per_clip > 0 && per_clip % 4 == 0 && aligned16(x), VTX_EINVAL,
              "mixup_batch: need an even batch, 16-byte aligned clips of a multiple of 4 floats");
  hipLaunchKernelGGL(mixup_kernel, dim3(grid_for((long)(B / 2) * (per_clip / 4))), dim3(256), 0, as_stream(stream), x, B / 2, per_clip, B,
                     lam, one_minus_lam);
  return check_launch("mixup_batch");
}

extern "C" int vtx_cutmix_batch(float* x, int B, int planes, int H, int W, int yl, int yh, int xl, int xh, void* stream) {
  VTX_REQUIRE(x && B > 0 && B % 2 == 0 && planes > 0 && H > 0 && W > 0, VTX_EINVAL, "cutmix_batch: bad shape");
  VTX_REQUIRE(0 <= yl && yl <= yh && yh <= H && 0 <= xl && xl <= xh && xh <= W, VTX_EINVAL, "cutmix_batch: box outside the frame");
  if (yl == yh || xl == xh) return VTX_OK;
  hipLaunchKernelGGL(cutmix_kernel, dim3(grid_for((long)(B / 2) * planes * (yh - yl) * (xh - xl))), dim3(256), 0, as_stream(stream), x,
                     B / 2, planes, H, W, B, yl, yh, xl, xh);
  return check_launch("cutmix_batch");
}

extern "C" int vtx_mixup_target(const long* labels, int B, int C, float on_value, float off_value, float lam, float one_minus_lam,
                                float* out, void* stream) {
  VTX_REQUIRE(labels && out && B > 0 && C > 0, VTX_EINVAL, "mixup_target: bad arguments");
  hipLaunchKernelGGL(mixup_target_kernel, dim3(grid_for((long)B * C)), dim3(256), 0, as_stream(stream), labels, B, C, on_value, off_value,
                     lam, one_minus_lam, out);
  return check_launch("mixup_target");
}

extern "C" int vtx_softmax_xent_fwd(const float* logits, const float* soft_targets, const long* labels, int B, int C, float* loss_rows,
                                    float* lse, float* loss_mean, void* stream) {
  VTX_REQUIRE(logits && loss_rows && lse && B > 0 && C > 0 && ((soft_targets != nullptr) != (labels != nullptr)), VTX_EINVAL,
              "softmax_xent_fwd: give either soft targets or labels");
  hipLaunchKernelGGL(xent_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), logits, soft_targets, labels, B, C, loss_rows, lse);
  int rc = check_launch("softmax_xent_fwd");
  if (rc || !loss_mean) return rc;
  hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, as_stream(stream), loss_rows, labels, B, C, loss_mean);
  return check_launch("softmax_xent_mean");
}

extern "C" int vtx_softmax_xent_bwd(const float* logits, const float* soft_targets, const long* labels, const float* lse, int B, int C,
                                    float grad_scale, const float* grad_loss, const float* count, float* dlogits, void* stream) {
  VTX_REQUIRE(logits && lse && dlogits && B > 0 && C > 0 && ((soft_targets != nullptr) != (labels != nullptr)), VTX_EINVAL,
              "softmax_xent_bwd: give either soft targets or labels");
  hipLaunchKernelGGL(xent_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), logits, soft_targets, labels, lse, B, C, grad_scale,
                     grad_loss, count, dlogits);
  return check_launch("softmax_xent_bwd");
}

extern "C" int vtx_topk_correct(const float* scores, const long* labels, int B, int C, int k, int* correct, void* stream) {
  VTX_REQUIRE(scores && labels && correct && B > 0 && C > 0 && k > 0, VTX_EINVAL, "topk_correct: bad arguments");
  hipLaunchKernelGGL(topk_correct_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), scores, labels, B, C, k, correct);
  return check_launch("topk_correct");
}
