"""Native MViT-B backbone for MaskFeat (the reference builds it from pytorchvideo:
video_transformer.py:15-17 imports, :621-800 ``create_multiscale_vision_transformers``, :844-852).

pytorchvideo is not a dependency here: the classes below carry pytorchvideo's module / parameter names
(``cls_positional_encoding.pos_embed_spatial``, ``blocks.N.attn.pool_q.weight``, ``blocks.N.mlp.fc1`` ...),
so a MaskFeat checkpoint trained with the reference loads key for key, and run on libvtx kernels
(csrc/mvit.hip + the GEMM / LayerNorm kernels of the transformer path).  Semantics: the keyword
arguments the reference passes select pytorchvideo 0.1.3 behaviour -- restated with citations in
oracle/mvit_oracle.py, against which tests/test_gpu_mvit.py checks every operator and the whole
backbone.  **Parity is unpinned by the reference** (pytorchvideo is absent from every disk of this build).

Supported: what the reference constructs -- conv pooling with 3x3x3 kernels, spatial strides, cls token,
separable position embedding, no dropout / DropPath (droppath_rate_block = 0 in the reference builder).
"""
from functools import partial

import torch
import torch.nn as nn

import vtx
from vtx import functions as F_


def round_width(width, multiplier, min_width=8, divisor=8):
    """pytorchvideo.layers.utils.round_width."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


def _compute(x):
    return F_.CastFn.apply(x, vtx.compute_dtype())


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, out_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x, res=None):
        return F_.MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, res)


class MultiScaleAttention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, kernel_q, kernel_kv, stride_q, stride_kv):
        super().__init__()
        self.num_heads = num_heads
        hd = dim // num_heads
        if hd not in (64, 96):
            raise NotImplementedError(f'vtx: MViT head_dim {hd} has no attention kernel (64 or 96)')
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

        def pool(kernel, stride):
            if not kernel or (all(k == 1 for k in kernel) and all(s == 1 for s in stride)):
                return None, None, None
            conv = nn.Conv3d(hd, hd, tuple(kernel), stride=tuple(stride), padding=tuple(k // 2 for k in kernel), groups=hd,
                             bias=False)
            return conv, nn.LayerNorm(hd), tuple(stride)
        self.pool_q, self.norm_q, self._stride_q = pool(kernel_q, stride_q)
        self.pool_k, self.norm_k, self._stride_kv = pool(kernel_kv, stride_kv)
        self.pool_v, self.norm_v, _ = pool(kernel_kv, stride_kv)

    def _pooled(self, x, lin, conv, norm, stride, thw):
        y = F_.LinearActResFn.apply(x, lin.weight, lin.bias, False, None)
        if conv is None:
            return y, thw
        y = F_.PoolConvLNFn.apply(y, conv.weight, norm.weight, norm.bias, thw, self.num_heads, stride, norm.eps)
        return y, F_.pooled_thw(thw, stride)

    def forward(self, x, thw, res):
        q, q_thw = self._pooled(x, self.q, self.pool_q, self.norm_q, self._stride_q, thw)
        k, _ = self._pooled(x, self.k, self.pool_k, self.norm_k, self._stride_kv, thw)
        v, _ = self._pooled(x, self.v, self.pool_v, self.norm_v, self._stride_kv, thw)
        o = F_.XAttnFn.apply(q, k, v, self.num_heads)
        return F_.LinearActResFn.apply(o, self.proj.weight, self.proj.bias, False, res), q_thw


class MultiScaleBlock(nn.Module):
    def __init__(self, dim, dim_out, num_heads, mlp_ratio=4.0, qkv_bias=False, dropout_rate=0.0, droppath_rate=0.0,
                 norm_layer=nn.LayerNorm, kernel_q=(1, 1, 1), kernel_kv=(1, 1, 1), stride_q=(1, 1, 1), stride_kv=(1, 1, 1),
                 pool_mode='conv', has_cls_embed=True, pool_first=False):
        super().__init__()
        if dropout_rate or droppath_rate or pool_mode != 'conv' or not has_cls_embed or pool_first:
            raise NotImplementedError('vtx: MultiScaleBlock supports conv pooling with a cls token, no dropout / DropPath '
                                      '(what the reference builds)')
        self.dim, self.dim_out = dim, dim_out
        self.norm1 = norm_layer(dim)
        self.attn = MultiScaleAttention(dim, num_heads, qkv_bias, list(kernel_q), list(kernel_kv), list(stride_q), list(stride_kv))
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), dim_out)
        if dim != dim_out:
            self.proj = nn.Linear(dim, dim_out)
        kernel_skip = [s + 1 if s > 1 else s for s in stride_q]
        # parameter-free; kept as a module so that printing / hooks look like pytorchvideo's block
        self.pool_skip = (nn.MaxPool3d(kernel_skip, list(stride_q), [k // 2 for k in kernel_skip], ceil_mode=False)
                          if len(kernel_skip) > 0 else None)
        if self.pool_skip is not None and (list(kernel_skip) != [1, 3, 3] or list(stride_q) != [1, 2, 2]):
            raise NotImplementedError('vtx: the skip-path max pool is implemented for kernel (1,3,3), stride (1,2,2)')

    def forward(self, x, thw):
        xn = F_.LayerNormFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, False)
        x_res = F_.MaxPoolSkipFn.apply(x, thw) if self.pool_skip is not None else x
        x, thw_new = self.attn(xn, thw, x_res)                   # x_res + attn(norm1(x)): the add rides in proj's epilogue
        x_norm = F_.LayerNormFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, False)
        if self.dim != self.dim_out:
            x = F_.LinearActResFn.apply(x_norm, self.proj.weight, self.proj.bias, False, None)
        return self.mlp(x_norm, x), thw_new                      # (proj(x_norm) | x) + mlp(x_norm)


class SpatioTemporalClsPositionalEncoding(nn.Module):
    def __init__(self, embed_dim, patch_embed_shape, sep_pos_embed=True, has_cls=True):
        super().__init__()
        if not (sep_pos_embed and has_cls):
            raise NotImplementedError('vtx: separable position embedding with a cls token only (what the reference builds)')
        self.patch_embed_shape = list(patch_embed_shape)
        T, H, W = patch_embed_shape
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed_spatial = nn.Parameter(torch.zeros(1, H * W, embed_dim))
        self.pos_embed_temporal = nn.Parameter(torch.zeros(1, T, embed_dim))
        self.pos_embed_class = nn.Parameter(torch.zeros(1, 1, embed_dim))

    def forward(self, x):
        return F_.PosEncodingFn.apply(x, self.cls_token, self.pos_embed_class, self.pos_embed_spatial, self.pos_embed_temporal)


class MultiscaleVisionTransformers(nn.Module):
    def __init__(self, *, patch_embed, cls_positional_encoding, pos_drop, norm_patch_embed, blocks, norm_embed, head):
        super().__init__()
        if pos_drop is not None or norm_patch_embed is not None or head is not None or patch_embed is not None:
            raise NotImplementedError('vtx: MultiscaleVisionTransformers as the reference builds it (no patch_embed / dropout / head)')
        self.cls_positional_encoding = cls_positional_encoding
        self.blocks = blocks
        self.norm_embed = norm_embed
        self._init_vit_weights()

    def _init_vit_weights(self, std=0.02):
        """What pytorchvideo's constructor ends with -- ``init_net_weights(self, init_std=0.02, style='vit')`` (restated
        from the package, which is on no disk here: parity unpinned like the rest of the backbone): truncated-normal Linear
        weights with zero biases, LayerNorm at (1, 0), truncated-normal cls token and separable position embeddings.
        MaskFeat pretraining always starts from this state (reference video_transformer.py:844-864 then re-initialises
        only patch_embed, decoder_pred and mask_token); the depthwise pooling convolutions keep nn.Conv3d's default."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=std)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
            elif isinstance(m, SpatioTemporalClsPositionalEncoding):
                for w in m.parameters():
                    nn.init.trunc_normal_(w, std=std)

    def forward(self, x):
        x = self.cls_positional_encoding(_compute(x))
        thw = self.cls_positional_encoding.patch_embed_shape
        for blk in self.blocks:
            x, thw = blk(x, thw)
        x = F_.LayerNormFn.apply(x, self.norm_embed.weight, self.norm_embed.bias, self.norm_embed.eps, False)
        return F_.CastFn.apply(x, torch.float32)


def create_multiscale_vision_transformers(*, spatial_size, temporal_size, cls_embed_on=True, sep_pos_embed=True, depth=16,
                                          norm='layernorm', input_channels=3, patch_embed_dim=96,
                                          conv_patch_embed_kernel=(3, 7, 7), conv_patch_embed_stride=(2, 4, 4),
                                          conv_patch_embed_padding=(1, 3, 3), enable_patch_embed_norm=False, use_2d_patch=False,
                                          num_heads=1, mlp_ratio=4.0, qkv_bias=True, dropout_rate_block=0.0,
                                          droppath_rate_block=0.0, pooling_mode='conv', pool_first=False, residual_pool=False,
                                          depthwise_conv=True, bias_on=True, separate_qkv=True, embed_dim_mul=None,
                                          atten_head_mul=None, pool_q_stride_size=None, pool_kv_stride_size=None,
                                          pool_kv_stride_adaptive=None, pool_kvq_kernel=None, head=None):
    """Same signature and bookkeeping as the reference builder (video_transformer.py:621-800): per-block width / heads
    from the multiplier tables, Q pooling at the listed blocks, K/V stride shrinking adaptively with the Q strides."""
    if use_2d_patch or enable_patch_embed_norm or norm != 'layernorm' or dropout_rate_block or droppath_rate_block:
        raise NotImplementedError('vtx: 3-D patches, LayerNorm, no dropout (the configuration the reference uses)')
    if pool_kv_stride_adaptive is not None:
        assert pool_kv_stride_size is None, 'pool_kv_stride_size should be none if pool_kv_stride_adaptive is set.'
    norm_layer = partial(nn.LayerNorm, eps=1e-6)
    if isinstance(spatial_size, int):
        spatial_size = (spatial_size, spatial_size)
    dims = [temporal_size, spatial_size[0], spatial_size[1]]
    shape = [dims[i] // conv_patch_embed_stride[i] for i in range(3)]
    pos = SpatioTemporalClsPositionalEncoding(embed_dim=patch_embed_dim, patch_embed_shape=shape, sep_pos_embed=sep_pos_embed,
                                              has_cls=cls_embed_on)
    dim_mul, head_mul = [1.0] * (depth + 1), [1.0] * (depth + 1)
    for i, m in (embed_dim_mul or []):
        dim_mul[i] = m
    for i, m in (atten_head_mul or []):
        head_mul[i] = m
    pool_q = [[] for _ in range(depth)]
    pool_kv = [[] for _ in range(depth)]
    stride_q = [[] for _ in range(depth)]
    stride_kv = [[] for _ in range(depth)]
    own_kernel = lambda strides: [s + 1 if s > 1 else s for s in strides]   # noqa: E731
    for item in (pool_q_stride_size or []):
        stride_q[item[0]] = list(item[1:])
        pool_q[item[0]] = list(pool_kvq_kernel) if pool_kvq_kernel is not None else own_kernel(item[1:])
    if pool_kv_stride_adaptive is not None:
        cur = list(pool_kv_stride_adaptive)
        pool_kv_stride_size = []
        for i in range(depth):
            if len(stride_q[i]) > 0:
                cur = [max(cur[d] // stride_q[i][d], 1) for d in range(len(cur))]
            pool_kv_stride_size.append([i] + cur)
    for item in (pool_kv_stride_size or []):
        stride_kv[item[0]] = list(item[1:])
        pool_kv[item[0]] = list(pool_kvq_kernel) if pool_kvq_kernel is not None else own_kernel(item[1:])
    blocks = nn.ModuleList()
    for i in range(depth):
        num_heads = round_width(num_heads, head_mul[i], min_width=1, divisor=1)
        patch_embed_dim = round_width(patch_embed_dim, dim_mul[i], divisor=num_heads)
        dim_out = round_width(patch_embed_dim, dim_mul[i + 1], divisor=round_width(num_heads, head_mul[i + 1]))
        blocks.append(MultiScaleBlock(dim=patch_embed_dim, dim_out=dim_out, num_heads=num_heads, mlp_ratio=mlp_ratio,
                                      qkv_bias=qkv_bias, norm_layer=norm_layer, kernel_q=pool_q[i], kernel_kv=pool_kv[i],
                                      stride_q=stride_q[i], stride_kv=stride_kv[i], pool_mode=pooling_mode,
                                      has_cls_embed=cls_embed_on, pool_first=pool_first))
    return MultiscaleVisionTransformers(patch_embed=None, cls_positional_encoding=pos, pos_drop=None, norm_patch_embed=None,
                                        blocks=blocks, norm_embed=norm_layer(dim_out), head=None)


class PatchEmbeding(nn.Module):
    """Conv3d patch embedding of the MViT stem (reference video_transformer.py:563-584): [B,C,T,H,W] -> [B, T'H'W', D].
    The Conv3d module holds the parameters (key ``patch_model.{weight,bias}``); the arithmetic is im2col + GEMM."""

    def __init__(self, *, patch_model=None):
        super().__init__()
        assert patch_model is not None
        self.patch_model = patch_model

    def forward(self, x):
        # the reference hands over clip.transpose(1, 2) ([B,C,T,H,W] view of the [B,T,C,H,W] batch): undo the view
        clip = x.transpose(1, 2)
        pm = self.patch_model
        return F_.ConvStemFn.apply(clip, pm.weight, pm.bias, tuple(pm.stride), tuple(pm.padding), vtx.compute_dtype())


def create_conv_patch_embed(*, in_channels, out_channels, conv_kernel_size=(1, 16, 16), conv_stride=(1, 4, 4),
                            conv_padding=(1, 7, 7), conv_bias=True, conv=nn.Conv3d):
    return PatchEmbeding(patch_model=conv(in_channels=in_channels, out_channels=out_channels, kernel_size=conv_kernel_size,
                                          stride=conv_stride, padding=conv_padding, bias=conv_bias))
