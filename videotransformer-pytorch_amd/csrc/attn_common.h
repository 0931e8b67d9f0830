// attn_common.h -- launch parameters and row addressing shared by attn.hip (VALU) and attn_mfma.hip.
#pragma once
#include "common.h"

namespace vtx {

struct AttnP {
  int mode, S, L, H, B, T, P;
  long ld_qkv, ld_out, ld_dout, ld_dqkv;
  float scale;
  int G;                        // sequences per workgroup
};

__device__ inline long in_row(const AttnP& p, int s, int i) {
  if (p.mode == VTX_ATTN_CONTIG) return (long)s * p.L + i;
  const int b = s / p.T, t = s - b * p.T;
  return (long)b * (1 + (long)p.P * p.T) + (i == 0 ? 0 : 1 + (long)(i - 1) * p.T + t);
}
__device__ inline long out_row(const AttnP& p, int s, int i) {
  if (p.mode == VTX_ATTN_CONTIG) return (long)s * p.L + i;
  const int b = s / p.T, t = s - b * p.T;
  return i == 0 ? (long)p.B * p.P * p.T + s : (long)b * p.P * p.T + (long)(i - 1) * p.T + t;
}


// attn_mfma.hip
bool attn_mfma_eligible(int dtype, int L, int hd);
bool attn_small_eligible(int dtype, int mode, int L, int hd);
int attn_fwd_small_launch(const AttnP& p, const void* qkv, void* out, float* lse, hipStream_t st);
int attn_bwd_small_launch(const AttnP& p, const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv,
                          hipStream_t st);
int attn_fwd_mfma_launch(const AttnP& p, const void* qkv, void* out, float* lse, hipStream_t st);
int attn_bwd_mfma_launch(const AttnP& p, const void* qkv, const void* o, const void* dout, const float* lse,
                         float* delta, void* dqkv, void* dqkv_cls, hipStream_t st);

}  // namespace vtx
