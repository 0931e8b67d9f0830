"""CPU restatement of the MViT-B backbone the reference builds from pytorchvideo (TEST INFRASTRUCTURE ONLY).

**Parity unpinned.**  The arithmetic lives in the third-party package ``pytorchvideo`` (requirements.txt:5,
unpinned; imported at video_transformer.py:15-17; call sites :693-698 SpatioTemporalClsPositionalEncoding,
:763-786 MultiScaleBlock, :792-800 MultiscaleVisionTransformers), which is on no disk of this build: there is
no reference run and no golden vector to pin this file against.  It restates, from the MViT paper
(arXiv:2104.11227) and the published pytorchvideo 0.1.3 sources as recalled, the semantics selected by the
keyword arguments the reference passes (no ``residual_pool`` / ``bias_on`` / ``depthwise_conv`` /
``separate_qkv`` arguments exist yet in that version):

  MultiScaleAttention   q, k, v = three Linear(dim, dim, bias=qkv_bias); heads split of the feature axis;
                        pooling of the non-cls tokens of q / k / v: per-head depthwise
                        Conv3d(hd, hd, kernel, stride, padding = kernel // 2, groups = hd, bias = False)
                        on the [T, H, W] token grid, cls token re-attached, then LayerNorm(hd) (eps 1e-5:
                        the block passes the bare nn.LayerNorm class); softmax(q k^T hd^-0.5) v; Linear proj.
  MultiScaleBlock       x_res = MaxPool3d(kernel = stride + 1 where stride > 1, stride, padding = kernel // 2)
                        skip on the non-cls tokens; x = x_res + attn(norm1(x)); x_norm = norm2(x);
                        x = (proj(x_norm) if dim != dim_out else x) + mlp(x_norm); Mlp = fc1, GELU, fc2(dim_out).
  positional encoding   separable: pos_embed_spatial [1, H*W, D] repeated over T + pos_embed_temporal [1, T, D]
                        repeat-interleaved over H*W, pos_embed_class for the cls token (prepended).
  schedule              create_multiscale_vision_transformers (video_transformer.py:621-800), norms eps 1e-6.

The module / parameter names are pytorchvideo's, i.e. the ``state_dict`` keys a MaskFeat checkpoint of the
reference holds (``mvit.blocks.3.attn.pool_q.weight`` ...), which the drop-in backbone keeps.
"""
import math

import torch
import torch.nn as nn


def round_width(width, multiplier, min_width=8, divisor=8):
    """pytorchvideo.layers.utils.round_width."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if width_out < 0.9 * width:
        width_out += divisor
    return int(width_out)


def attention_pool(tensor, pool, thw, norm=None):
    """[B, heads, 1 + T*H*W, hd] (or [B, 1 + THW, C]) -> pooled, cls token kept apart and re-attached."""
    if pool is None:
        return tensor, thw
    nd = tensor.ndim
    if nd == 3:
        tensor = tensor.unsqueeze(1)
    cls_tok, tensor = tensor[:, :, :1], tensor[:, :, 1:]
    B, N, L, C = tensor.shape
    T, H, W = thw
    tensor = tensor.reshape(B * N, T, H, W, C).permute(0, 4, 1, 2, 3).contiguous()
    tensor = pool(tensor)
    thw = [tensor.shape[2], tensor.shape[3], tensor.shape[4]]
    tensor = tensor.reshape(B, N, C, thw[0] * thw[1] * thw[2]).transpose(2, 3)
    tensor = torch.cat((cls_tok, tensor), dim=2)
    if norm is not None:
        tensor = norm(tensor)
    if nd == 3:
        tensor = tensor.squeeze(1)
    return tensor, thw


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, out_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class MultiScaleAttention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, kernel_q, kernel_kv, stride_q, stride_kv):
        super().__init__()
        self.num_heads = num_heads
        hd = dim // num_heads
        self.scale = hd ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

        def pool(kernel, stride):
            if not kernel or (math.prod(kernel) == 1 and math.prod(stride) == 1):
                return None, None
            conv = nn.Conv3d(hd, hd, tuple(kernel), stride=tuple(stride), padding=tuple(k // 2 for k in kernel), groups=hd,
                             bias=False)
            return conv, nn.LayerNorm(hd)
        self.pool_q, self.norm_q = pool(kernel_q, stride_q)
        self.pool_k, self.norm_k = pool(kernel_kv, stride_kv)
        self.pool_v, self.norm_v = pool(kernel_kv, stride_kv)

    def forward(self, x, thw):
        B, N, C = x.shape
        split = lambda t: t.reshape(B, N, self.num_heads, C // self.num_heads).permute(0, 2, 1, 3)   # noqa: E731
        q, q_thw = attention_pool(split(self.q(x)), self.pool_q, thw, self.norm_q)
        k, _ = attention_pool(split(self.k(x)), self.pool_k, thw, self.norm_k)
        v, _ = attention_pool(split(self.v(x)), self.pool_v, thw, self.norm_v)
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, q.shape[2], C)
        return self.proj(x), q_thw


class MultiScaleBlock(nn.Module):
    def __init__(self, dim, dim_out, num_heads, mlp_ratio, qkv_bias, kernel_q, kernel_kv, stride_q, stride_kv, norm_eps=1e-6):
        super().__init__()
        self.dim, self.dim_out = dim, dim_out
        self.norm1 = nn.LayerNorm(dim, eps=norm_eps)
        self.attn = MultiScaleAttention(dim, num_heads, qkv_bias, kernel_q, kernel_kv, stride_q, stride_kv)
        self.norm2 = nn.LayerNorm(dim, eps=norm_eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), dim_out)
        if dim != dim_out:
            self.proj = nn.Linear(dim, dim_out)
        kernel_skip = [s + 1 if s > 1 else s for s in stride_q]
        self.pool_skip = (nn.MaxPool3d(kernel_skip, list(stride_q), [k // 2 for k in kernel_skip], ceil_mode=False)
                          if len(kernel_skip) > 0 else None)

    def forward(self, x, thw):
        x_block, thw_new = self.attn(self.norm1(x), thw)
        x_res, _ = attention_pool(x, self.pool_skip, thw)
        x = x_res + x_block
        x_norm = self.norm2(x)
        x_mlp = self.mlp(x_norm)
        if self.dim != self.dim_out:
            x = self.proj(x_norm)
        return x + x_mlp, thw_new


class SpatioTemporalClsPositionalEncoding(nn.Module):
    def __init__(self, embed_dim, patch_embed_shape):
        super().__init__()
        self.patch_embed_shape = list(patch_embed_shape)
        T, H, W = patch_embed_shape
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed_spatial = nn.Parameter(torch.zeros(1, H * W, embed_dim))
        self.pos_embed_temporal = nn.Parameter(torch.zeros(1, T, embed_dim))
        self.pos_embed_class = nn.Parameter(torch.zeros(1, 1, embed_dim))

    def forward(self, x):
        T, H, W = self.patch_embed_shape
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        pos = self.pos_embed_spatial.repeat(1, T, 1) + torch.repeat_interleave(self.pos_embed_temporal, H * W, dim=1)
        return x + torch.cat([self.pos_embed_class, pos], 1)


def mvit_schedule(depth, patch_embed_dim, num_heads, embed_dim_mul, atten_head_mul, pool_q_stride_size,
                  pool_kv_stride_adaptive, pool_kvq_kernel):
    """Per-block (dim, dim_out, heads, kernel_q, kernel_kv, stride_q, stride_kv): the bookkeeping of
    create_multiscale_vision_transformers (reference video_transformer.py:700-762)."""
    dim_mul, head_mul = [1.0] * (depth + 1), [1.0] * (depth + 1)
    for i, m in (embed_dim_mul or []):
        dim_mul[i] = m
    for i, m in (atten_head_mul or []):
        head_mul[i] = m
    pool_q = [[] for _ in range(depth)]
    pool_kv = [[] for _ in range(depth)]
    stride_q = [[] for _ in range(depth)]
    stride_kv = [[] for _ in range(depth)]
    for item in (pool_q_stride_size or []):
        stride_q[item[0]] = list(item[1:])
        pool_q[item[0]] = list(pool_kvq_kernel) if pool_kvq_kernel is not None else [s + 1 if s > 1 else s for s in item[1:]]
    if pool_kv_stride_adaptive is not None:
        cur = list(pool_kv_stride_adaptive)
        for i in range(depth):
            if len(stride_q[i]) > 0:
                cur = [max(cur[d] // stride_q[i][d], 1) for d in range(len(cur))]
            stride_kv[i] = list(cur)
            pool_kv[i] = list(pool_kvq_kernel) if pool_kvq_kernel is not None else [s + 1 if s > 1 else s for s in cur]
    out = []
    for i in range(depth):
        num_heads = round_width(num_heads, head_mul[i], min_width=1, divisor=1)
        patch_embed_dim = round_width(patch_embed_dim, dim_mul[i], divisor=num_heads)
        dim_out = round_width(patch_embed_dim, dim_mul[i + 1], divisor=round_width(num_heads, head_mul[i + 1]))
        out.append(dict(dim=patch_embed_dim, dim_out=dim_out, num_heads=num_heads, kernel_q=pool_q[i], kernel_kv=pool_kv[i],
                        stride_q=stride_q[i], stride_kv=stride_kv[i]))
    return out


class MultiscaleVisionTransformers(nn.Module):
    def __init__(self, spatial_size=224, temporal_size=16, depth=16, patch_embed_dim=96, conv_patch_embed_stride=(2, 4, 4),
                 num_heads=1, mlp_ratio=4.0, qkv_bias=True, embed_dim_mul=None, atten_head_mul=None, pool_q_stride_size=None,
                 pool_kv_stride_adaptive=None, pool_kvq_kernel=None):
        super().__init__()
        dims = [temporal_size // conv_patch_embed_stride[0], spatial_size // conv_patch_embed_stride[1],
                spatial_size // conv_patch_embed_stride[2]]
        self.cls_positional_encoding = SpatioTemporalClsPositionalEncoding(patch_embed_dim, dims)
        sched = mvit_schedule(depth, patch_embed_dim, num_heads, embed_dim_mul, atten_head_mul, pool_q_stride_size,
                              pool_kv_stride_adaptive, pool_kvq_kernel)
        self.blocks = nn.ModuleList([MultiScaleBlock(mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, **s) for s in sched])
        self.norm_embed = nn.LayerNorm(sched[-1]['dim_out'], eps=1e-6)

    def forward(self, x):
        x = self.cls_positional_encoding(x)
        thw = self.cls_positional_encoding.patch_embed_shape
        for blk in self.blocks:
            x, thw = blk(x, thw)
        return self.norm_embed(x)
