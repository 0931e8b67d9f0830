/* vtx.h -- C ABI of libvtx.so, the MI355X (gfx950) video-transformer hot path.
 *
 * The reference (mx-mark/VideoTransformer-pytorch) has no FFI layer: its hot
 * path is ATen calls made from transformer.py / video_transformer.py.  This
 * header is the boundary a maintainer binds instead of those ATen calls; every
 * entry point names the reference lines it replaces.  Python binds it with
 * ctypes (videotransformer-pytorch_amd/vtx/_lib.py); see INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every device pointer is BORROWED for the call; the library never
 *     allocates, frees or synchronises (scratch memory comes from the caller:
 *     vtx_*_workspace() queries); it keeps no mutable state besides the
 *     tuning switches of vtx_set_option(), so calls may run concurrently on
 *     different streams / devices as long as their buffers (workspaces
 *     included) are distinct; work is enqueued on `stream` (a hipStream_t
 *     passed as void*; NULL = the null stream) of the CURRENT device.
 *   - returns VTX_OK (0) or a negative VTX_E* code; never throws.
 *     vtx_last_error_string() gives the reason for the last failure on the
 *     calling thread.
 *   - dtype VTX_F32 = exact-fp32 path (f32 MFMA / fp32 VALU), VTX_BF16 = bf16
 *     storage with fp32 accumulation (bf16 MFMA).  Parameters that stay fp32 in
 *     both modes (bias, LayerNorm gamma/beta, all gradients of parameters) are
 *     typed float* here.
 *   - "row map": a logical row m of an operand lives at physical row
 *         base + m + (m / grp) * skip            (or base + m + tab[m / grp], see vtx_rowmap)
 *     which expresses "skip the cls row of every clip" (grp = P*T, skip = 1,
 *     base = 1) and "one row per clip" (grp = 1, skip = P*T) without copies.
 *     grp <= 0 means identity (+ base).
 */
#ifndef VTX_H_
#define VTX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTX_OK 0
#define VTX_EINVAL (-1)   /* bad shape / null pointer / unsupported combination */
#define VTX_EALIGN (-2)   /* pointer or leading dimension not 16-byte aligned  */
#define VTX_ELAUNCH (-3)  /* hipGetLastError() after launch != hipSuccess      */
#define VTX_EWS (-4)      /* workspace too small                               */

#define VTX_F32 0
#define VTX_BF16 1
#define VTX_BF16_X32 3    /* vtx_layernorm_bwd only: bf16 gradients, x stored as float32 (the exact residual stream, vtx_layernorm_acc_fwd) */

typedef struct {
  int grp;   /* rows per group (<=0: no groups) */
  int skip;  /* extra physical rows at the start of every group (table form: an upper bound of tab[g + 1] - tab[g]) */
  int base;  /* physical row of logical row 0 */
  const int* tab; /* NULL, or a DEVICE array of one row offset per group (non-decreasing) + one spare entry: the table form
                        base + m + tab[m / grp]
                     -- "the kept clips only" of a DropPath'ed FFN (transformer.py:34-42,543: group g = the g-th kept clip,
                     tab[g] = (its clip index - g) * rows per clip).  Accepted by LayerNorm, the element-wise kernels and
                     the C / residual maps of vtx_gemm_nt; vtx_gemm_nt's A map and vtx_gemm_tn take the closed form only. */
} vtx_rowmap;

int vtx_version(void);
const char* vtx_last_error_string(void);
/* Tuning / diagnostic switches (process-wide; initial values from the VTX_* environment variables,
 * read once): "gemm_nt" = auto|pp256|dma2|ring128x3|ring128x4k32|ring256x3|ring256x3k32|ring256x4k32,
 * "gemm_tn" = auto|pp256|ring|dma2|w4, "gemm_nodma", "tn_safe", "attn_valu" = 0|1, "tn_cus" = n (compute units the weight-gradient
 * kernel's one-round slab split is sized for: 256; 240 leaves room for 16 CUs held by a collective in flight -- another, equally fixed
 * summation order of the slabs), "attn_hw_fwd", "attn_hw_bwd" = n
 * (short-sequence attention: n heads of a row tile per workgroup, 0 = one head and four row tiles; defaults 16 / 4), "pp_grid", "pp_cg",
 * "pp_epi" = integers ("pp_epi": 1 = per-pass epilogue of the persistent GEMM; 2 / 3 = timing diagnostics that skip
 * its stores / its LDS staging and produce WRONG output; 4 = the general passes instead of the lean ones of the
 * continuous-flow kernels, 5 = lean passes (the default, same as 0), 6 = lean passes with the first four of every tile rolled
 * into its last K tile -- 4 / 5 / 6 give identical results), "pp_cont" = 0|1 (continuous flow of the persistent GEMM: the
 * next tile's first K tiles are requested inside the current main loop; 1 by default, 0 = per-tile prologue; identical
 * results), "ln_rows" = 1 .. 4 (rows per trip of the LayerNorm forward kernel, default 3; identical results),
 * "attn_fused" = 0|1|2 (backward of the 33..224-token attention: kernel pair / one pass with two phases / one phase with streamed
 * operands (193..224 tokens; default); dk and dv identical, dq of 2 equal to fp32 rounding), "attn_fwd_stream" = 0|1 (forward of
 * the 193..224-token attention: workgroup per (sequence, head) / persistent with streamed K and V (default); identical results),
 * "attn_dkv" = 0 .. 4 (variants of the pair's dk / dv kernel; identical results),
 * "pp_trace" = device address of a timeline buffer (tools/pp_timeline.py, tools/attn_timeline.py).  Returns VTX_EINVAL for an
 * unknown name or value. */
int vtx_set_option(const char* name, const char* value);

/* ------------------------------------------------------------------ LayerNorm
 * Replaces nn.LayerNorm in the blocks (transformer.py:215,257 / :321,359 /
 * :418,439 / :495,519; eps 1e-5) and the final norm (video_transformer.py:119,
 * 251 / :401,527; eps 1e-6).  fp32 statistics; saves mean / rstd per row. */
int vtx_layernorm_fwd(int dtype, int rows, int D, const void* x, long ldx, vtx_rowmap xmap,
                      const float* gamma, const float* beta, float eps,
                      void* y, long ldy, vtx_rowmap ymap, float* mean, float* rstd, void* stream);
/* The exact residual stream (library version 220; vtx.set_stream('fp32')).  The bf16 path stores the stream x as bf16 and every
 * sub-block's x + f(x) is rounded: 3 roundings per layer in series, an error of the SUM that grows with sqrt(depth) (at 24 layers 3x
 * the deviation of the reference's own torch.autocast run, which keeps the stream in float32: transformer.py:275,380,522 add in
 * float32 under autocast).  In this mode a sub-block hands on its contribution d = f(x) (bf16: the residual GEMM's epilogue without
 * the residual) and the stream lives in float32, advanced by the LayerNorm of the NEXT sub-block:
 *     xo[omap(m)] = (xs ? xs[smap(m)] : 0) + d[smap(m)]          float32
 *     y[ymap(m)]  = LayerNorm(row xo) * gamma + beta              bf16; y == NULL: accumulate only (rows no LayerNorm covers)
 * xs, d share the leading dimension lds (the stream layout); mean / rstd as in vtx_layernorm_fwd.  bf16 compute only. */
int vtx_layernorm_acc_fwd(int rows, int D, const float* xs, const void* d, long lds, vtx_rowmap smap, float* xo, long ldo,
                          vtx_rowmap omap, const float* gamma, const float* beta, float eps, void* y, long ldy, vtx_rowmap ymap,
                          float* mean, float* rstd, void* stream);
/* dx[xmap(m)] = (dres ? dres[xmap(m)] : 0) + LN'(dy[m]);  dgamma/dbeta += column sums.  dtype VTX_BF16_X32: x is float32.
 * workspace: vtx_layernorm_bwd_workspace(rows, D) bytes. */
size_t vtx_layernorm_bwd_workspace(int rows, int D);
int vtx_layernorm_bwd(int dtype, int rows, int D, const void* dy, long lddy, vtx_rowmap dymap,
                      const void* x, long ldx, vtx_rowmap xmap, const float* mean, const float* rstd,
                      const float* gamma, const void* dres, void* dx, long lddx,
                      float* dgamma, float* dbeta, void* workspace, size_t ws_bytes, void* stream);
/* The float32 GRADIENT stream of the bf16 kernels (round 6; with vtx_layernorm_acc_fwd the residual stream is float32 in both
 * directions, as it is under the reference's torch.autocast where `x = x + drop_path(f(norm(x)))` (transformer.py:281,381,455,522)
 * adds bf16 branch outputs to a float32 x and autograd sums the branch gradients in float32):
 *   dx32[xmap(m)] = dres32[xmap(m)] + LN'(dy[m])  in float32,   dx[xmap(m)] = bf16(dx32[xmap(m)])
 * dy, dx bf16; x the float32 stream; dres32 / dx32 / dx share xmap and lddx; D <= 1024; same workspace as vtx_layernorm_bwd. */
int vtx_layernorm_bwd_g32(int rows, int D, const void* dy, long lddy, vtx_rowmap dymap, const float* x, long ldx, vtx_rowmap xmap,
                          const float* mean, const float* rstd, const float* gamma, const float* dres32, float* dx32,
                          void* dx, long lddx, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes, void* stream);

/* ----------------------------------------------------------------------- GEMM
 * C[M,N] = epilogue(A[M,K] * B[N,K]^T).  Replaces every nn.Linear forward and
 * input-gradient on the path: qkv / proj (transformer.py:160-175), temporal_fc
 * (:225,267), FFN (:498-507,520-521), the patch/tubelet projection after the
 * patch gather (:116-126,142,146), MaskFeat decoder_pred
 * (video_transformer.py:855,878).
 * Epilogue order:  v = acc (+bias[n]);  act 1 (erf GELU): C2 = v, v = gelu(v);  act 2: C2 = gelu'(v), v = gelu(v);
 *                  v *= gelu'(dgelu_in[m][n])  (dgelu_kind 1: v *= dgelu_in[m][n]);  v *= row_scale[idx(m)];
 *                  v += R[rmap(m) or m % r_period][n];  C[cmap(m)][n] = v.
 * Rows m >= split_row (if split_row > 0) are stored to Csplit[m - split_row]
 * with bias/act/scale applied but no residual (the per-frame cls rows of the
 * divided spatial attention, transformer.py:371-373). */
typedef struct {
  int dtype;
  int M, N, K;
  const void* A; long lda; vtx_rowmap amap;
  const void* B; long ldb;            /* [N,K], K contiguous */
  void* C; long ldc; vtx_rowmap cmap;
  const float* bias;                  /* [N] or NULL */
  int act;                            /* 0 none, 1 GELU(erf), 2 GELU(erf) with the derivative as second output */
  void* C2; long ldc2;                /* act 1: pre-activation copy (may be NULL); act 2: gelu'(pre-activation), required
                                         -- what the FFN backward multiplies by, so that its epilogue needs no erf / exp */
  const void* dgelu_in; long ld_dgelu;/* multiply by gelu'(x) (FFN backward), or NULL */
  int dgelu_kind;                     /* 0: dgelu_in holds x (the act-1 copy); 1: it holds gelu'(x) itself (the act-2 output) */
  const float* row_scale;             /* DropPath keep-scale per row group, or NULL */
  int rs_d1, rs_m1, rs_d2, rs_m2;     /* idx(m) = (m / rs_d1) * rs_m1 + (m % rs_d2) * rs_m2 */
  const void* R; long ldr; vtx_rowmap rmap; int r_period; /* residual; r_period>0: row m % r_period */
  int split_row; void* Csplit; long ldsplit;
  void* workspace; size_t ws_bytes;   /* vtx_gemm_nt_workspace() bytes, ZEROED once by the caller and private to the
                                         stream the call runs on: tile counters of the persistent bf16 kernel, which
                                         leaves them zeroed.  NULL selects the non-persistent kernels. */
} vtx_gemm_desc;
size_t vtx_gemm_nt_workspace(void);
int vtx_gemm_nt(const vtx_gemm_desc* d, void* stream);

/* Weight gradient: C[N1,N2] (fp32) (+)= sum_m A[amap(m)][n1] * B[bmap(m)][n2].
 * Replaces autograd's mm(grad^T, input) for every Linear above.  Split over M
 * into `splits` slabs reduced by a second kernel (deterministic). */
typedef struct {
  int dtype;
  int M, N1, N2;
  const void* A; long lda; vtx_rowmap amap;
  const void* B; long ldb; vtx_rowmap bmap;
  float* C; long ldc; int accumulate;
  void* workspace; size_t ws_bytes;
  float* colsum; int colsum_accumulate; /* optional: colsum[n1] (+)= sum_m A[amap(m)][n1] (the bias gradient,
                                           fused: A tiles are already on chip) */
} vtx_gemm_tn_desc;
size_t vtx_gemm_tn_workspace(int M, int N1, int N2);
int vtx_gemm_tn(const vtx_gemm_tn_desc* d, void* stream);

/* Small fp32 products of weights with weights (exact-fp32 matrix instruction, fixed summation order):
 *   C[N1,N2] (+)= alpha * sum_k A(i,k) B(k,j)  (+ u[i] v[j]),     A(i,k) = A[i*a_rs + k*a_ks], B(k,j) = B[k*b_ks + j*b_cs]
 * (exactly one stride of each operand is 1), and optionally, from the same A tiles,
 *   y[i] (+)= alpha_y * sum_k A(i,k) x[k] + beta_z * z[i].
 * The merged attn.proj o temporal_fc GEMM of the divided temporal attention (transformer.py:268-275: two Linear layers
 * with only a DropPath row scale between them) forms W_c = W_tfc W_proj, b_c = W_tfc b_proj with it and maps the merged
 * weight gradient G back: dW_tfc = c G W_proj^T + u b_proj^T, dW_proj = c W_tfc^T G, db_proj = W_tfc^T u.
 * N1, N2, K multiples of 4; 16-byte aligned operands. */
typedef struct {
  int N1, N2, K;
  const float* A; long a_rs, a_ks;
  const float* B; long b_ks, b_cs;
  float alpha;
  float* C; long ldc; int accumulate;
  const float* u; const float* v;     /* rank-1 term, added unscaled; both or neither */
  const float* x; float* y;           /* vector product; both or neither */
  float alpha_y; const float* z; float beta_z; int y_accumulate;
} vtx_wprod_desc;
int vtx_wprod(const vtx_wprod_desc* d, void* stream);

/* out[n] (+)= sum_m A[amap(m)][n]   (bias gradients). */
size_t vtx_colsum_workspace(int M, int N);
int vtx_colsum(int dtype, int M, int N, const void* A, long lda, vtx_rowmap amap,
               float* out, int accumulate, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ Attention
 * softmax(q k^T * scale) v per (sequence, head).  Replaces transformer.py:
 * 167-174 and the rearranges around it (:250, :352-356, :375).  qkv feature
 * order [3][head][hd] (:167).
 * mode VTX_ATTN_CONTIG: sequence s = rows [s*L, (s+1)*L) of qkv / out.
 * mode VTX_ATTN_SPACE : divided spatial attention.  qkv is in natural token
 *   order [B, 1+P*T, 3D]; sequence s=(b,t) has L = 1+P tokens: i=0 is the cls
 *   row of clip b, i>=1 is token 1+(i-1)*T+t.  out is [B*P*T + B*T, D]: token
 *   rows first (clip-major, (p t) order, no cls), then one cls row per (b,t).
 * lse: [S,H,L] fp32 log-sum-exp (saved for backward).  probs (optional,
 *   fp32 [S,H,L,L]) materialises the softmax for get_last_selfattention
 *   (video_transformer.py:258-261). */
#define VTX_ATTN_CONTIG 0
#define VTX_ATTN_SPACE 1
typedef struct {
  int dtype, mode;
  int S, L, H, hd;
  int B, T, P;                  /* VTX_ATTN_SPACE only */
  const void* qkv; long ld_qkv;
  void* out; long ld_out;
  float* lse;
  float* probs;
  float scale;
} vtx_attn_desc;
int vtx_attn_fwd(const vtx_attn_desc* d, void* stream);

/* dqkv from (qkv, out, dout, lse).  dout/out share the forward `out` layout.
 * VTX_ATTN_SPACE: token rows of dqkv in natural order; the T per-frame
 * gradients of each clip's cls row go to dqkv_cls [B*T, 3D] (same dtype) and
 * are summed over t by vtx_cls_qkv_reduce.  delta: [S,H,L] fp32 scratch. */
typedef struct {
  vtx_attn_desc f;
  const void* dout; long ld_dout;
  void* dqkv; long ld_dqkv;
  void* dqkv_cls;
  float* delta;
} vtx_attn_bwd_desc;
int vtx_attn_bwd(const vtx_attn_bwd_desc* d, void* stream);

/* ------------------------------------------------- divided-attention glue ops */
/* out[b,0,:] = x[b,0,:] + mean_t a_cls[b*T+t,:]   (transformer.py:371-377); x == NULL: the mean alone (exact residual stream) */
int vtx_cls_mean_fwd(int dtype, int B, int T, int D, const void* a_cls, long lda,
                     const void* x, void* out, long ld_tok, long rows_per_clip, void* stream);
/* ViViT fact_encoder glue (video_transformer.py:511-525): x [(B T), 1 + P, D] -> h [B, 1 + T, D],
 * h[i, 0] = x[i, 0] + e[0] (row i of the FLATTENED (b t) axis, as the reference's `x[:b, 0]` reads it),
 * h[i, 1 + t] = mean_p x[i T + t, 1 + p] + e[1 + t];  e = time_embed [1 + T, D] fp32.  Backward: dx, d_time_embed (+)=. */
int vtx_fact_glue_fwd(int dtype, int B, int T, int P, int D, const void* x, const float* time_embed, void* h, void* stream);
int vtx_fact_glue_bwd(int dtype, int B, int T, int P, int D, const void* dh, void* dx, float* d_time_embed, int accumulate,
                      void* stream);
/* da[0:B*N]   = dout[b,1+n] * s[b*T + n%T];  da[B*N + b*T+t] = dout[b,0] * s[b*T+t] / T
 * (backward of transformer.py:367-377; s may be NULL = 1) */
int vtx_space_grad_prep(int dtype, int B, int T, int P, int D, const void* dout, long ld,
                        const float* s, void* da, long ldda, void* stream);
/* dqkv[b,0,:] = sum_t dqkv_cls[b*T+t,:]  (cls row replicated per frame, transformer.py:354-356) */
int vtx_cls_qkv_reduce(int dtype, int B, int T, int W, const void* dqkv_cls, long ldc,
                       void* dqkv, long ld, long rows_per_clip, void* stream);
/* dst[dmap(m)] = src[smap(m)] * (s ? s[(m/rs_d1)*rs_m1 + (m%rs_d2)*rs_m2] : 1) */
int vtx_row_scale_copy(int dtype, int rows, int D, const void* src, long lds, vtx_rowmap smap,
                       void* dst, long ldd, vtx_rowmap dmap, const float* s,
                       int rs_d1, int rs_m1, int rs_d2, int rs_m2, void* stream);
/* Rows of dropped DropPath groups (s[m / group_rows] == 0.0f; a group = group_rows consecutive rows m of [0, M)):
 *   out[omap(m)] = x[xmap(m)] + bias   (out may be NULL; x == NULL: the bias alone -- exact residual stream),   zero[m] = 0   (zero may be NULL; compact [M, D] rows).
 * Fix-up behind the merged attn.proj + temporal_fc GEMM: the reference's DropPath sits between the two Linear layers
 * (transformer.py:268-275), so a dropped sequence leaves the block as x + temporal_fc.bias. */
int vtx_dropped_rows_fix(int dtype, long M, int D, int group_rows, const float* s, const void* x, long ldx,
                         vtx_rowmap xmap, const float* bias, void* out, long ldo, vtx_rowmap omap, void* zero,
                         long ldz, void* stream);
/* part[w, :] (fp32, [nparts, D]) = column sums of src[smap(m), :] over the rows of the dropped groups inside the w-th of
 * nparts runs of consecutive groups; fold the rows of `part` with vtx_reduce_rows (fixed order, no atomics). */
int vtx_dropped_rows_colsum(int dtype, long M, int D, int group_rows, const float* s, const void* src, long lds,
                            vtx_rowmap smap, float* part, int nparts, void* stream);
/* out[j,:] (+)= scale * sum_{i<ni} in[base + i*si + j*sj, :]   (fp32 out).  in_dtype: VTX_F32/BF16. */
int vtx_reduce_rows(int in_dtype, int nj, int ni, int D, const void* in, long ld, long base,
                    long si, long sj, float* out, long ldo, float scale, int accumulate, void* stream);
/* out = dy * gelu'(h), n elements (n % 8 == 0): backward of an erf-GELU whose gradient does not come out of a
 * GEMM epilogue (MViT Mlp, video_transformer.py:763-786 via pytorchvideo). */
int vtx_gelu_grad_mul(int dtype, size_t n, const void* dy, const void* h, void* out, void* stream);
/* Weight staging: W fp32 [R,C] -> Wc (dtype, [R,C], optional) and WcT (dtype, [C,R], optional). */
int vtx_cast_transpose(int dtype, int R, int C, const float* W, void* Wc, void* WcT, void* stream);
/* The same for a whole table of weights in ONE launch (what the step after an optimizer update needs: ~86 matrices
 * of TimeSformer-B).  tab / tile_start are DEVICE arrays: tile_start[i] = number of 64x64 tiles of tensors 0..i-1
 * (n_tensors + 1 entries, tiles of tensor i = ceil(rows/64) * ceil(cols/64)), n_tiles = tile_start[n_tensors]. */
typedef struct {
  const float* src;   /* [rows, cols] fp32 */
  void* dst_c;        /* [rows, cols] dtype, or NULL */
  void* dst_t;        /* [cols, rows] dtype, or NULL */
  int rows, cols;
} vtx_ct_tensor;
int vtx_mt_cast_transpose(int dtype, const vtx_ct_tensor* tab, const int* tile_start, int n_tensors, int n_tiles, void* stream);
/* dst (dtype) = src (fp32), n elements; and the reverse. */
int vtx_cast_from_f32(int dtype, size_t n, const float* src, void* dst, void* stream);
int vtx_cast_to_f32(int dtype, size_t n, const void* src, float* dst, void* stream);

/* ------------------------------------------------------ patch / tubelet embed
 * Gather of PatchEmbed.forward (transformer.py:138-151): clip [B,T,C,H,W]
 * (fp32) -> rows[(b, p, t'), K] in `dtype`, already in token order (p major,
 * t' minor), K index = c*ts*ps*ps + kt*ps*ps + kh*ps + kw (ts = 1 for Conv2d).
 * frame_major != 0 keeps the reference's (b t') p row order instead (ViViT
 * fact_encoder, space_only). */
int vtx_patch_rows(int dtype, int B, int T, int C, int H, int W, int ps, int ts,
                   const float* clip, void* rows, long ldr, int frame_major, void* stream);
/* Same rows straight from decoded video: clip [B,T,H,W,3] uint8, channels last (what
 * dataset.py:171 holds before its permute), with the reference's ToTensor
 * (x.float().div(255), data_transform.py:52-63) and transforms.Normalize
 * ((x - mean[c]) / std[c], data_transform.py:534-539) applied on the fly in that
 * order (fp32 rows bit-identical to vtx_patch_rows of the normalised clip).
 * mean3 / std3 are HOST pointers to 3 floats. */
int vtx_patch_rows_u8(int dtype, int B, int T, int H, int W, int ps, int ts,
                      const unsigned char* clip, const float* host_mean3, const float* host_std3,
                      void* rows, long ldr, int frame_major, void* stream);
/* E[p*T+t,:] = bias + pos[1+p,:] + (time ? time[t,:] : 0);  cls_row = cls + pos[0]
 * (prepare_tokens, video_transformer.py:199-237).  E, cls_row in `dtype`. */
int vtx_embed_table(int dtype, int P, int T, int D, const float* bias, const float* pos,
                    const float* time_embed, const float* cls, void* E, void* cls_row,
                    int frame_major, void* stream);

/* ------------------------------------------------------------------------ HOG
 * extract_hog_features (dataset.py:39-45) for F frames [F,H,W,3] uint8 ->
 * [F, H/16, W/16, 108] float64, bit-exact vs skimage 0.18.3 when `table` is
 * the blob built by vtx_hog_build_table on the host and uploaded by the caller
 * (vtx_hog_table_bytes() bytes: the 256*256 doubles hypot(|g_col|, |g_row|) of
 * the host's libm, then 4096 words holding, per gradient pair, how many ulps
 * that hypot lies from the correctly rounded square root -- the kernel computes
 * the square root and applies the correction).  frames and out 16-byte aligned.
 * bins (optional): [F,3,H,W] int32 orientation-bin ids. */
size_t vtx_hog_table_bytes(void);
int vtx_hog_build_table(double* host_table);
/* table_bytes: size of the uploaded blob (vtx_version() >= 210).  The kernel reads the correction words BEHIND the 512 KB of
 * magnitudes, so a blob of the pre-210 layout (magnitudes only) is rejected with VTX_EINVAL instead of being read 16 KB beyond
 * its end. */
int vtx_hog_fwd(const uint8_t* frames, int F, int H, int W, const double* table, size_t table_bytes,
                double* out, int32_t* bins, void* stream);

/* ------------------------------------------------------------ MaskFeat head
 * Mask-token blend (video_transformer.py:914-919): x [B, Tq*Hq*Wq, C], mask
 * [B,Tq,g,g] uint8 upsampled by r = Hq/g (nearest). */
int vtx_maskfeat_blend_fwd(int dtype, int B, int Tq, int Hq, int Wq, int C, int g,
                           const void* x, const uint8_t* mask, const float* mask_token,
                           void* out, void* stream);
/* dx = dy * (1-w);  dtoken[C] (+)= sum dy * w  (fp32). */
int vtx_maskfeat_blend_bwd(int dtype, int B, int Tq, int Hq, int Wq, int C, int g,
                           const void* dy, const uint8_t* mask, void* dx, float* dtoken,
                           void* stream);
/* Masked MSE (video_transformer.py:882-901).  pred: decoder output without the
 * cls row, [B, Tq*g*g, ts*Cf] in `dtype` (Cf = 108 HOG features, ts = temporal
 * stride); target f64 [B, Tq*ts, g, g, Cf]; cmask uint8 [B, Tq*ts, g, g] = mask
 * repeated over ts with non-centre frames zeroed (host-built, :889-896).
 * loss_out[0] = loss (f64), loss_out[1] = sum(cmask).  scratch: f64[2] zeroed
 * by the call. */
int vtx_maskfeat_loss_fwd(int dtype, int B, int Tq, int ts, int g, int Cf, const void* pred, long ldp,
                          const double* target, const uint8_t* cmask, double* loss_out, void* stream);
/* dpred = gloss * 2 (pred - target) * cmask / (Cf * float32(sum(cmask) + 1e-5)) */
int vtx_maskfeat_loss_bwd(int dtype, int B, int Tq, int ts, int g, int Cf, const void* pred, long ldp,
                          const double* target, const uint8_t* cmask, const double* loss_out,
                          float gloss, void* dpred, long lddp, void* stream);

/* ------------------------------------------------------------------ MViT backbone operators
 * The operators of the MViT-B backbone of MaskFeat that the blocks above do not cover (the reference
 * builds it from pytorchvideo: video_transformer.py:621-800; semantics restated in oracle/mvit_oracle.py).
 * Token tensors are [B, 1 + T*H*W, heads*hd] (cls first, heads interleaved in the feature axis), head_dim 64 or 96.
 *
 * Pooling of q / k / v (MultiScaleAttention): per-head depthwise Conv3d(3x3x3, stride (1,sh,sw), padding 1, no
 * bias; w [hd,27] = the [hd,1,3,3,3] weight) on the non-cls tokens + LayerNorm(hd) on every token (cls included).
 * Output grid H' = (H - 1) / sh + 1.  fwd saves `pre` (conv output, before the norm) and mean / rstd [B*(1+N')*heads]. */
typedef struct { int dtype, B, T, H, W, heads, hd, sh, sw; } vtx_pool_desc;
int vtx_pool_conv_ln_fwd(const vtx_pool_desc* d, const void* x, const float* w, const float* gamma, const float* beta,
                         float eps, void* pre, void* y, float* mean, float* rstd, void* stream);
size_t vtx_pool_conv_ln_bwd_workspace(const vtx_pool_desc* d);
/* dpre: scratch of the output's shape; dx of the input's; dw [hd,27], dgamma / dbeta [hd] are overwritten. */
int vtx_pool_conv_ln_bwd(const vtx_pool_desc* d, const void* dy, const void* x, const void* pre, const float* mean,
                         const float* rstd, const float* w, const float* gamma, void* dpre, void* dx, float* dw,
                         float* dgamma, float* dbeta, void* workspace, size_t ws_bytes, void* stream);
/* Residual-path MaxPool3d(kernel (1,3,3), stride (1,2,2), padding (0,1,1)), cls row kept; arg [B,1+N',C] uint8. */
int vtx_maxpool_skip_fwd(int dtype, int B, int T, int H, int W, int C, const void* x, void* y, uint8_t* arg, void* stream);
int vtx_maxpool_skip_bwd(int dtype, int B, int T, int H, int W, int C, const void* dy, const uint8_t* arg, void* dx, void* stream);
/* Separable position encoding + cls token: out [B,1+T*HW,C] from x [B,T*HW,C]; cls, pos_class [C], spatial [HW,C],
 * temporal [T,C] fp32. */
int vtx_pos_encoding_fwd(int dtype, int B, int T, int HW, int C, const void* x, const float* cls, const float* pos_class,
                         const float* spatial, const float* temporal, void* out, void* stream);
/* Rows of an overlapping, padded Conv3d on the clip [B,T,C,H,W] fp32: rows[(b,to,ho,wo)][c*kt*kh*kw + ...] (weight
 * order), zero padded to Kp columns; k3 / s3 / p3 = HOST (t,h,w) kernel / stride / padding triples. */
int vtx_im2col3d(int dtype, int B, int T, int C, int H, int W, const int* k3, const int* s3, const int* p3, int Kp,
                 const float* clip, void* rows, void* stream);
/* softmax(q k^T * scale) v with separate q [B,Lq,heads*hd] and k, v [B,Lk,heads*hd]; lse [B,heads,Lq] fp32.
 * bwd: delta [B,heads,Lq] fp32 scratch; workspace (vtx_xattn_bwd_workspace bytes): fp32 partial dk / dv of the query
 * splits that keep the chip busy when the keys are few (393 pooled keys against 25 089 queries in MViT block 0). */
typedef struct {
  int dtype, B, Lq, Lk, heads, hd;
  float scale;
  const void* q; const void* k; const void* v;
  void* out; float* lse;
} vtx_xattn_desc;
int vtx_xattn_fwd(const vtx_xattn_desc* d, void* stream);
size_t vtx_xattn_bwd_workspace(const vtx_xattn_desc* d);
int vtx_xattn_bwd(const vtx_xattn_desc* d, const void* dout, float* delta, void* dq, void* dk, void* dv, void* workspace,
                  size_t ws_bytes, void* stream);

/* ------------------------------------------- clip-batch mixing, classification loss, accuracy
 * Mixup / CutMix of the clip batch in place (mixup.py:102-126; the numpy draws stay on the host): x is
 * [B, per_clip] fp32 (a [B,T,C,H,W] batch), B even; clip b is mixed with clip B-1-b.
 *   mixup : x[b] = x[b]*lam + x[B-1-b]*one_minus_lam  (both passed as the fp32 values ATen uses; separate
 *           multiply / multiply / add roundings: bit-identical to the reference's mul_().add_())
 *   cutmix: the box rows [yl,yh) x columns [xl,xh) of every one of the `planes` = T*C planes is swapped. */
int vtx_mixup_batch(float* x, int B, long per_clip, float lam, float one_minus_lam, void* stream);
int vtx_cutmix_batch(float* x, int B, int planes, int H, int W, int yl, int yh, int xl, int xh, void* stream);
/* mixup_target (mixup.py:20-25): out[b][c] = v(b,c)*lam + v(B-1-b,c)*one_minus_lam, v = on_value at the label, else off_value. */
int vtx_mixup_target(const long* labels, int B, int C, float on_value, float off_value, float lam, float one_minus_lam,
                     float* out, void* stream);
/* Softmax cross-entropy on fp32 logits [B,C] with EITHER soft targets [B,C] (timm SoftTargetCrossEntropy,
 * model_trainer.py:87-88) or int64 labels (nn.CrossEntropyLoss, :91).  loss_rows[b] = sum_c -t log_softmax(x);
 * lse[b] saved for backward.  loss_mean (optional, float[2]): [0] = mean of loss_rows over the COUNTED rows, summed in
 * a fixed order, [1] = their number.  Every row counts, except rows whose label lies outside [0, C) -- that is
 * nn.CrossEntropyLoss's default ignore_index = -100 (loss 0, no gradient, not in the denominator; torch raises a device
 * assert for other out-of-range labels, here they are ignored the same way).
 * bwd: dlogits = grad_scale * grad_loss[0] / count[0] * (softmax * sum_c t - t); grad_loss = the upstream gradient of
 * the scalar loss ON THE DEVICE (NULL = 1); count (device, NULL = 1) = forward's loss_mean + 1 with grad_scale = 1 for
 * the mean; or grad_scale = 1/B with count = NULL when every row counts. */
int vtx_softmax_xent_fwd(const float* logits, const float* soft_targets, const long* labels, int B, int C, float* loss_rows,
                         float* lse, float* loss_mean, void* stream);
int vtx_softmax_xent_bwd(const float* logits, const float* soft_targets, const long* labels, const float* lse, int B, int C,
                         float grad_scale, const float* grad_loss, const float* count, float* dlogits, void* stream);
/* correct[0] += number of rows whose label is among the k largest scores (torchmetrics Accuracy(top_k),
 * model_trainer.py:83-84,213-214); ties resolved like torch.topk (lower index first); a label outside [0, C) is never correct. */
int vtx_topk_correct(const float* scores, const long* labels, int B, int C, int k, int* correct, void* stream);

/* ------------------------------------------------- optimizer step / gradient clipping
 * The step right after backward (model_trainer.py:155-170 clip_gradients, :218-231; optimizer.py:31-38):
 * every parameter is one row of a DEVICE table; three launches per step whatever the parameter count.
 * chunk_start: device int[n_tensors + 1], exclusive prefix sum of vtx_mt_chunks(numel) per tensor;
 * n_chunks = chunk_start[n_tensors].  All tensors fp32, contiguous. */
typedef struct {
  void* p;         /* parameter                                   */
  const void* g;   /* gradient                                    */
  void* s1;        /* SGD momentum buffer / AdamW exp_avg         */
  void* s2;        /* AdamW exp_avg_sq (NULL for SGD)             */
  long n;          /* elements                                    */
  float lr, wd;    /* learning rate and weight decay of its group */
} vtx_mt_tensor;
int vtx_mt_chunks(long numel);
/* norms[t] = ||g_t||_2 (t < n_tensors), norms[n_tensors] = ||(norms[0..n))||_2 -- the value the reference's
 * clip_gradients returns.  partial: float[n_chunks] scratch.  Deterministic (fixed summation order). */
int vtx_mt_grad_norms(const vtx_mt_tensor* tab, const int* chunk_start, int n_tensors, int n_chunks,
                      float* partial, float* norms, void* stream);
/* torch.optim.SGD(momentum, nesterov, dampening 0, weight_decay = tab[t].wd): g += wd p; buf = first_step ? g :
 * momentum buf + g; p -= lr (nesterov ? g + momentum buf : buf).  clip > 0: g is first scaled by
 * min(1, clip / (norms[t] + 1e-6)) (per parameter, model_trainer.py:165-168); the stored gradient is not modified. */
int vtx_mt_sgd_step(const vtx_mt_tensor* tab, const int* chunk_start, int n_tensors, int n_chunks, const float* norms,
                    float clip, float momentum, int nesterov, int first_step, void* stream);
/* torch.optim.AdamW(betas, eps, weight_decay = tab[t].wd), step counted from 1. */
int vtx_mt_adamw_step(const vtx_mt_tensor* tab, const int* chunk_start, int n_tensors, int n_chunks, const float* norms,
                      float clip, float beta1, float beta2, float eps, int step, void* stream);

/* ------------------------------------------------- data-parallel gradient exchange: NOT in this library
 * SURVEY.md section 8(b2) lists vtx_dp_{init, allreduce_bucket, finalize} in the minimum export set.  They are deliberately
 * absent: the exchange (the reference gets it from Lightning's DDP plugin, model_pretrain.py:200-204) is host-side
 * orchestration of RCCL collectives, and this library's rules -- no allocation, no synchronisation, no state -- exclude
 * owning communicators, bucket memory and side streams.  It lives in Python over torch.distributed (backend "nccl" = RCCL
 * on ROCm; SURVEY.md section 7 step 8 allows exactly this): videotransformer-pytorch_amd/vtx/dp.py
 *     init             -> torch.distributed.init_process_group (model_pretrain.single_run, bench.py)
 *     allreduce_bucket -> GradBuckets: flat fp32 buckets in reverse layer order, one async all-reduce(mean) per bucket,
 *                         issued from the post-accumulate hook of its last gradient (or directly behind the kernel that
 *                         wrote it: this library's weight-gradient kernels accumulate straight into the bucket views)
 *     finalize         -> GradBuckets.finish() (wait; fold 1/world when the backend has no AVG op)
 * What the kernels contribute is only that they tolerate the collective's CUs (dynamic tile scheduling, grid-agnostic
 * results) and write gradients where the collective reads them. */

/* MFMA / LDS-transpose layout self-test: runs one-hot probes through the
 * instructions the GEMM kernels rely on and writes a report; returns the
 * number of layout assumptions that failed (0 = all hold). */
int vtx_selftest(char* report, size_t report_bytes);

#ifdef __cplusplus
}
#endif
#endif /* VTX_H_ */
