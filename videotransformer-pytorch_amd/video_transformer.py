"""Drop-in replacement for the reference ``video_transformer.py`` models
(TimeSformer, ViViT, MaskFeat head) running on libvtx.so HIP kernels (gfx950).

Same constructor arguments, methods, attributes and ``state_dict`` keys as the
reference.  The token preparation is fused: one patch
gather kernel + one GEMM whose epilogue adds bias / positional / time embeddings
and writes tokens straight into ``b (p t) d`` order (reference
video_transformer.py:193-240), instead of the reference's four full-tensor copies.
"""
import math

import torch
import torch.nn as nn

import vtx
from vtx import functions as F_
from vtx import ops
from transformer import PatchEmbed, TransformerContainer, get_sine_cosine_pos_emb, stream_value
from weight_init import (trunc_normal_, init_from_vit_pretrain_, init_from_mae_pretrain_,
                         init_from_kinetics_pretrain_)
from mvit import (PatchEmbeding, create_conv_patch_embed, create_multiscale_vision_transformers)  # noqa: F401


def _embed(param_or_tensor, device):
    """Learnable embeddings are Parameters; sine/cosine ones are plain tensors that the
    reference moves per call with .type_as(x).detach() (video_transformer.py:204,226)."""
    if isinstance(param_or_tensor, nn.Parameter):
        return param_or_tensor
    return param_or_tensor.detach().to(device=device, dtype=torch.float32)


class _VideoTransformerBase(nn.Module):
    def _setup_embeddings(self, num_patches, num_frames, embed_dims, use_learnable_pos_emb, dropout_p,
                          with_time):
        if dropout_p:
            raise NotImplementedError('vtx: dropout_p > 0 is not supported by the HIP path')
        if use_learnable_pos_emb:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dims))
        else:
            self.pos_embed = get_sine_cosine_pos_emb(num_patches, embed_dims)
        self.drop_after_pos = nn.Dropout(p=dropout_p)
        if with_time:
            if use_learnable_pos_emb:
                self.time_embed = nn.Parameter(torch.zeros(1, num_frames, embed_dims))
            else:
                self.time_embed = get_sine_cosine_pos_emb(num_frames, embed_dims)
            self.drop_after_time = nn.Dropout(p=dropout_p)

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'pos_embed', 'cls_token', 'mask_token'}

    def _tokens(self, x, layout, time_embed, pos_embed=None):
        proj = self.patch_embed.projection
        dev = x.device
        pos = _embed(self.pos_embed, dev) if pos_embed is None or pos_embed is self.pos_embed else pos_embed
        dtype = vtx.compute_dtype()
        exact = F_.exact_stream() and dtype == torch.bfloat16
        out = F_.TokensFn.apply(x, proj.weight, proj.bias, self.cls_token, pos,
                                None if time_embed is None else _embed(time_embed, dev), dtype, layout, exact)
        if exact:                                    # the stream starts in float32 (transformer._stream_of reads the attribute)
            out, xs0 = out
            out._vtx_xs = xs0
        return out

    def _readout(self, x):
        # under vtx.set_stream('fp32') x is the last sub-block's contribution and carries the float32 stream (transformer._stream_of)
        exact = F_.exact_stream() and x.dtype == torch.bfloat16 and getattr(x, '_vtx_xs', None) is not None
        xs = x._vtx_xs if exact else None
        if self.return_cls_token:
            y = F_.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, True, xs, exact)
            return F_.CastFn.apply(y, torch.float32)
        y = F_.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, False, xs, exact)
        return F_.CastFn.apply(y, torch.float32)[:, 1:].mean(1)


class TimeSformer(_VideoTransformerBase):
    """TimeSformer (reference video_transformer.py:20-261)."""
    supported_attention_types = ['divided_space_time', 'space_only', 'joint_space_time']

    def __init__(self, num_frames, img_size=224, patch_size=16, pretrain_pth=None, weights_from='imagenet',
                 embed_dims=768, num_heads=12, num_transformer_layers=12, in_channels=3, conv_type='Conv2d',
                 dropout_p=0., attention_type='divided_space_time', norm_layer=nn.LayerNorm,
                 copy_strategy='repeat', use_learnable_pos_emb=True, return_cls_token=True, **kwargs):
        super().__init__()
        assert attention_type in self.supported_attention_types, f'Unsupported Attention Type {attention_type}!'
        self.num_frames = num_frames
        self.pretrain_pth = pretrain_pth
        self.weights_from = weights_from
        self.embed_dims = embed_dims
        self.num_transformer_layers = num_transformer_layers
        self.attention_type = attention_type
        self.copy_strategy = copy_strategy
        self.conv_type = conv_type
        self.use_learnable_pos_emb = use_learnable_pos_emb
        self.return_cls_token = return_cls_token

        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_channels=in_channels,
                                      embed_dims=embed_dims, conv_type=conv_type)
        num_patches = self.patch_embed.num_patches
        if attention_type == 'divided_space_time':
            operator_order = ['time_attn', 'space_attn', 'ffn']
        else:
            operator_order = ['self_attn', 'ffn']
        self.transformer_layers = TransformerContainer(
            num_transformer_layers=num_transformer_layers, embed_dims=embed_dims, num_heads=num_heads,
            num_frames=num_frames, norm_layer=norm_layer, hidden_channels=embed_dims * 4,
            operator_order=operator_order)
        self.norm = norm_layer(embed_dims, eps=1e-6)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dims))
        self.use_cls_token_temporal = operator_order[-2] == 'time_attn'      # always False (App. A)
        if self.use_cls_token_temporal:
            num_frames = num_frames + 1
        else:
            num_patches = num_patches + 1
        self._setup_embeddings(num_patches, num_frames, embed_dims, use_learnable_pos_emb, dropout_p,
                               with_time=attention_type != 'space_only')
        self.init_weights()

    def init_weights(self):
        if self.use_learnable_pos_emb:
            nn.init.trunc_normal_(self.pos_embed, std=.02)
            if self.attention_type != 'space_only':
                nn.init.trunc_normal_(self.time_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        if self.pretrain_pth is not None:
            if self.weights_from == 'imagenet':
                init_from_vit_pretrain_(self, self.pretrain_pth, self.conv_type, self.attention_type,
                                        self.copy_strategy)
            elif self.weights_from == 'kinetics':
                init_from_kinetics_pretrain_(self, self.pretrain_pth)
            else:
                raise TypeError(f'not support the pretrained weight {self.pretrain_pth}')

    def interpolate_pos_encoding(self, x, w, h):
        """The positional table for a clip of width w and height h whose token tensor is x ([., 1 + patches, D]; only its
        shape is read): pos_embed itself at the trained square resolution, otherwise its patch part resized bicubically to
        the clip's patch grid (reference video_transformer.py:171-191, with its quirks: both sides divided by
        patch_size[0], the width ratio applied to the first grid axis, +0.1 on each side before the ratio).  The resize is a
        torch op on the [1, D, s, s] parameter view -- outside the per-layer path, differentiable, and never run at the
        trained resolution."""
        npatch = x.shape[1] - 1
        n = self.pos_embed.shape[1] - 1
        if npatch == n and w == h:
            return self.pos_embed
        side = int(math.sqrt(n))
        ps = self.patch_embed.patch_size[0]
        w0, h0 = w // ps + 0.1, h // ps + 0.1
        table = self.pos_embed if isinstance(self.pos_embed, nn.Parameter) else self.pos_embed.detach()
        grid = table[:, 1:].reshape(1, side, side, -1).permute(0, 3, 1, 2)
        grid = nn.functional.interpolate(grid, scale_factor=(w0 / math.sqrt(n), h0 / math.sqrt(n)), mode='bicubic')
        if (int(w0), int(h0)) != tuple(grid.shape[-2:]):
            raise ValueError(f'pos_embed resize produced a {tuple(grid.shape[-2:])} grid for a {int(w0)} x {int(h0)} clip')
        return torch.cat([table[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, table.shape[-1])], dim=1)

    def prepare_tokens(self, x):
        # x: float [B,T,C,H,W] (the reference's input) or, beyond the reference, the decoded uint8 clip
        # [B,T,H,W,3] with vtx.set_input_normalization(mean, std) (ToTensor + Normalize fused into the gather)
        b, t, c, h, w = vtx.ops.clip_dims(x)
        ps = self.patch_embed.patch_size
        grid = (h // ps[0]) * (w // ps[1])
        pos = self.interpolate_pos_encoding(torch.empty(0, 1 + grid, 0), w, h)
        if pos is not self.pos_embed:
            pos = pos.to(device=x.device, dtype=torch.float32)
        if self.attention_type == 'space_only':
            return self._tokens(x, 'tp', None, pos), b
        return self._tokens(x, 'pt', self.time_embed, pos), b

    def forward(self, x):
        x, b = self.prepare_tokens(x)
        x = self.transformer_layers(x)
        if self.attention_type == 'space_only':
            # mean over the frames of each clip before the norm (reference :247-249)
            x = stream_value(x)
            n1, d = x.shape[1], x.shape[2]
            x32 = F_.CastFn.apply(x, torch.float32).reshape(b, -1, n1, d).mean(1)
            x = F_.CastFn.apply(x32, vtx.compute_dtype())
        return self._readout(x)

    def get_last_selfattention(self, x):
        x, b = self.prepare_tokens(x)
        return self.transformer_layers(x, return_attention=True)


def get_vit_base_patch16_224(**kwargs):
    return TimeSformer(num_frames=kwargs['num_frames'], pretrain_pth=kwargs['pretrain_pth'],
                       weights_from=kwargs['weights_from'], img_size=kwargs['img_size'],
                       attention_type=kwargs['attention_type'], patch_size=16, embed_dims=768, num_heads=12,
                       in_channels=3, num_transformer_layers=12, conv_type='Conv2d', dropout_p=0.,
                       norm_layer=nn.LayerNorm, copy_strategy='repeat', use_learnable_pos_emb=True,
                       return_cls_token=True)


class ViViT(_VideoTransformerBase):
    """ViViT (reference video_transformer.py:270-556)."""
    supported_attention_types = ['fact_encoder', 'joint_space_time', 'divided_space_time']

    def __init__(self, num_frames, img_size=224, patch_size=16, pretrain_pth=None, weights_from='imagenet',
                 embed_dims=768, num_heads=12, num_transformer_layers=12, in_channels=3, dropout_p=0.,
                 tube_size=2, conv_type='Conv3d', attention_type='fact_encoder', norm_layer=nn.LayerNorm,
                 copy_strategy='repeat', extend_strategy='temporal_avg', use_learnable_pos_emb=True,
                 return_cls_token=True, **kwargs):
        super().__init__()
        assert attention_type in self.supported_attention_types, f'Unsupported Attention Type {attention_type}!'
        num_frames = num_frames // tube_size
        self.num_frames = num_frames
        self.pretrain_pth = pretrain_pth
        self.weights_from = weights_from
        self.embed_dims = embed_dims
        self.num_transformer_layers = num_transformer_layers
        self.attention_type = attention_type
        self.conv_type = conv_type
        self.copy_strategy = copy_strategy
        self.extend_strategy = extend_strategy
        self.tube_size = tube_size
        self.num_time_transformer_layers = 0
        self.use_learnable_pos_emb = use_learnable_pos_emb
        self.return_cls_token = return_cls_token

        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_channels=in_channels,
                                      embed_dims=embed_dims, tube_size=tube_size, conv_type=conv_type)
        num_patches = self.patch_embed.num_patches

        def container(layers, order):
            return TransformerContainer(num_transformer_layers=layers, embed_dims=embed_dims,
                                        num_heads=num_heads, num_frames=num_frames, norm_layer=norm_layer,
                                        hidden_channels=embed_dims * 4, operator_order=order)

        operator_order = None
        if attention_type == 'divided_space_time':
            operator_order = ['time_attn', 'space_attn', 'ffn']
            transformer_layers = container(num_transformer_layers, operator_order)
        elif attention_type == 'joint_space_time':
            operator_order = ['self_attn', 'ffn']
            transformer_layers = container(num_transformer_layers, operator_order)
        else:
            self.num_time_transformer_layers = 4
            transformer_layers = nn.ModuleList([
                container(num_transformer_layers, ['self_attn', 'ffn']),
                container(self.num_time_transformer_layers, ['self_attn', 'ffn'])])
        self.transformer_layers = transformer_layers
        self.norm = norm_layer(embed_dims, eps=1e-6)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dims))
        if attention_type == 'fact_encoder':
            num_frames = num_frames + 1
            num_patches = num_patches + 1
            self.use_cls_token_temporal = False
        else:
            self.use_cls_token_temporal = operator_order[-2] == 'time_attn'
            if self.use_cls_token_temporal:
                num_frames = num_frames + 1
            else:
                num_patches = num_patches + 1
        self._setup_embeddings(num_patches, num_frames, embed_dims, use_learnable_pos_emb, dropout_p,
                               with_time=True)
        self.init_weights()

    def init_weights(self):
        if self.use_learnable_pos_emb:
            nn.init.trunc_normal_(self.pos_embed, std=.02)
            nn.init.trunc_normal_(self.time_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        if self.pretrain_pth is not None:
            if self.weights_from == 'imagenet':
                init_from_vit_pretrain_(self, self.pretrain_pth, self.conv_type, self.attention_type,
                                        self.copy_strategy, self.extend_strategy, self.tube_size,
                                        self.num_time_transformer_layers)
            elif self.weights_from == 'kinetics':
                init_from_kinetics_pretrain_(self, self.pretrain_pth)
            else:
                raise TypeError(f'not support the pretrained weight {self.pretrain_pth}')

    def prepare_tokens(self, x):
        b = x.shape[0]
        if self.attention_type == 'fact_encoder':
            tok = self._tokens(x, 'tp', None)
        else:
            tok = self._tokens(x, 'pt', self.time_embed)
        # the reference also returns its (pre-embedding) cls tokens; nothing downstream reads them
        cls_tokens = self.cls_token.expand(tok.shape[0], -1, -1)
        return tok, cls_tokens, b

    def _fact_temporal_tokens(self, x, b):
        """Glue between the spatial and temporal encoders (reference :515-523), kept
        literal: the cls rows are the first b rows of the flattened (b t) axis."""
        x = stream_value(x)                          # (the exact residual stream: the spatial encoder's stream as one tensor)
        D = x.shape[2]
        if D % 8 == 0 and D <= 1024 and x.dtype == vtx.compute_dtype():
            return F_.FactGlueFn.apply(x, _embed(self.time_embed, x.device), b)
        # widths vtx_fact_glue_fwd does not take (16-byte vectors of a row, one row per workgroup pass): the same expressions
        # as device-side ATen ops between the two encoders (round 3's form; not on any BASELINE configuration)
        x32 = F_.CastFn.apply(x, torch.float32)
        cls_b = x32[:b, 0:1]
        frames = x32[:, 1:].reshape(b, -1, x32.shape[1] - 1, x32.shape[2]).mean(2)
        h = torch.cat([cls_b, frames], dim=1) + _embed(self.time_embed, x.device)
        return F_.CastFn.apply(h.contiguous(), vtx.compute_dtype())

    def forward(self, x):
        x, cls_tokens, b = self.prepare_tokens(x)
        if self.attention_type != 'fact_encoder':
            x = self.transformer_layers(x)
        else:
            spatial_transformer, temporal_transformer = self.transformer_layers
            x = spatial_transformer(x)
            x = self._fact_temporal_tokens(x, b)
            x = temporal_transformer(x)
        return self._readout(x)

    def get_last_selfattention(self, x):
        x, cls_tokens, b = self.prepare_tokens(x)
        if self.attention_type != 'fact_encoder':
            return self.transformer_layers(x, return_attention=True)
        spatial_transformer, temporal_transformer = self.transformer_layers
        x = spatial_transformer(x)
        x = self._fact_temporal_tokens(x, b)
        return temporal_transformer(x, return_attention=True)


# ------------------------------------------------------------------------------------
# MaskFeat (reference video_transformer.py:803-922): MViT-B backbone (mvit.py -- native replacement of
# the pytorchvideo modules the reference imports) + mask-token blend, decoder and HOG-target masked MSE,
# all on libvtx kernels.
# ------------------------------------------------------------------------------------
class _MaskBlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask_u8, mask_token, dims):
        B, Tq, Hq, Wq, C, g = dims
        x = x.contiguous()
        out = torch.empty_like(x)
        vtx._lib.call('vtx_maskfeat_blend_fwd', ops.dt(x), B, Tq, Hq, Wq, C, g, ops.ptr(x), ops.ptr(mask_u8),
                      ops.ptr(mask_token.reshape(-1).contiguous()), ops.ptr(out), ops.stream())
        ctx.save_for_backward(mask_u8)
        ctx.dims = dims
        ctx.tok_shape = mask_token.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        (mask_u8,) = ctx.saved_tensors
        B, Tq, Hq, Wq, C, g = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        dtok = torch.empty(C, dtype=torch.float32, device=dy.device)
        vtx._lib.call('vtx_maskfeat_blend_bwd', ops.dt(dy), B, Tq, Hq, Wq, C, g, ops.ptr(dy), ops.ptr(mask_u8),
                      ops.ptr(dx), ops.ptr(dtok), ops.stream())
        return dx, None, dtok.reshape(ctx.tok_shape), None


class _MaskedMSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, cmask, dims):
        B, Tq, ts, g, Cf = dims
        pred = pred.contiguous()
        acc = torch.empty(2, dtype=torch.float64, device=pred.device)
        vtx._lib.call('vtx_maskfeat_loss_fwd', ops.dt(pred), B, Tq, ts, g, Cf, ops.ptr(pred), pred.shape[-1],
                      ops.ptr(target), ops.ptr(cmask), ops.ptr(acc), ops.stream())
        ctx.save_for_backward(pred, target, cmask, acc)
        ctx.dims = dims
        return acc[0]

    @staticmethod
    def backward(ctx, gloss):
        pred, target, cmask, acc = ctx.saved_tensors
        B, Tq, ts, g, Cf = ctx.dims
        dpred = torch.empty_like(pred)
        vtx._lib.call('vtx_maskfeat_loss_bwd', ops.dt(pred), B, Tq, ts, g, Cf, ops.ptr(pred), pred.shape[-1],
                      ops.ptr(target), ops.ptr(cmask), ops.ptr(acc), float(gloss), ops.ptr(dpred),
                      pred.shape[-1], ops.stream())
        return dpred, None, None, None


def center_frame_mask(mask, cube_marker, num_frames, tstride):
    """Host-side replacement of the reference's per-sample device loop
    (video_transformer.py:889-896): mask [B,T',h,w] -> uint8 [B,T,h,w] keeping only
    the centre frame (start*ts + span*ts//2) of every cube, plus the keep table."""
    keep = torch.zeros(mask.shape[0], num_frames, dtype=torch.bool)
    for i, markers in enumerate(cube_marker):
        for start, span in markers:
            keep[i, int(start) * tstride + int(span) * tstride // 2] = True
    keep = keep.to(mask.device)
    m = mask.repeat_interleave(tstride, 1) * keep[:, :, None, None].to(mask.dtype)
    return m.to(torch.uint8).contiguous(), keep


class MaskFeat(nn.Module):
    """MaskFeat pretraining head over an MViT backbone (reference :803-922)."""

    def __init__(self, img_size=224, num_frames=16, input_channels=3, feature_dim=10, patch_embed_dim=96,
                 conv_patch_embed_kernel=(3, 7, 7), conv_patch_embed_stride=(2, 4, 4),
                 conv_patch_embed_padding=(1, 3, 3), embed_dim_mul=[[1, 2.0], [3, 2.0], [14, 2.0]],
                 atten_head_mul=[[1, 2.0], [3, 2.0], [14, 2.0]],
                 pool_q_stride_size=[[1, 1, 2, 2], [3, 1, 2, 2], [14, 1, 2, 2]],
                 pool_kv_stride_adaptive=[1, 8, 8], pool_kvq_kernel=[3, 3, 3], head=None, pretrain_pth=None,
                 backbone=None, **kwargs):
        super().__init__()
        self.num_frames = num_frames
        self.img_size = img_size
        self.stride = conv_patch_embed_stride
        self.downsample_rate = 2 ** len(pool_q_stride_size)
        self.embed_dims = 2 ** len(embed_dim_mul) * patch_embed_dim
        self.patch_embed = create_conv_patch_embed(in_channels=input_channels, out_channels=patch_embed_dim,
                                                   conv_kernel_size=conv_patch_embed_kernel,
                                                   conv_stride=conv_patch_embed_stride,
                                                   conv_padding=conv_patch_embed_padding, conv=nn.Conv3d)
        if backbone is not None:                  # beyond the reference: any module mapping [B, L, 96] -> [B, 1+L/16, 768]
            self.mvit = backbone
        else:
            self.mvit = create_multiscale_vision_transformers(
                spatial_size=img_size, temporal_size=num_frames, embed_dim_mul=embed_dim_mul, atten_head_mul=atten_head_mul,
                pool_q_stride_size=pool_q_stride_size, pool_kv_stride_adaptive=pool_kv_stride_adaptive,
                pool_kvq_kernel=pool_kvq_kernel, head=head)
        self.decoder_pred = nn.Linear(self.embed_dims, feature_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, patch_embed_dim))
        w = self.patch_embed.patch_model.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.xavier_uniform_(self.decoder_pred.weight)
        nn.init.constant_(self.decoder_pred.bias, 0)
        nn.init.trunc_normal_(self.mask_token, std=.02)
        if pretrain_pth is not None:
            self.init_weights(pretrain_pth)

    def init_weights(self, pretrain_pth):
        init_from_kinetics_pretrain_(self, pretrain_pth)

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'pos_embed', 'cls_token', 'mask_token'}

    def blend_mask_tokens(self, tokens, mask):
        """x*(1-w) + mask_token*w with the mask upsampled by downsample_rate (:914-919)."""
        B, L, Cc = tokens.shape
        tq = self.num_frames // self.stride[0]
        hq = self.img_size // self.stride[1]
        wq = self.img_size // self.stride[2]
        g = hq // self.downsample_rate
        tokens = F_.CastFn.apply(tokens, vtx.compute_dtype())
        return _MaskBlendFn.apply(tokens, mask.to(torch.uint8).contiguous(), self.mask_token,
                                  (B, tq, hq, wq, Cc, g))

    def forward_features(self, x, mask=None):
        x = self.patch_embed(x.transpose(1, 2))
        if mask is not None:
            x = self.blend_mask_tokens(x, mask)
        return self.mvit(x)

    def head_loss(self, feat, target_x, mask, cube_marker, visualize=False):
        """decoder_pred + reshape + centre-frame masked MSE (:878-909) on features
        [B, 1+T'*g*g, embed_dims]."""
        ts = self.stride[0]
        tq = self.num_frames // ts
        g = self.img_size // (self.stride[1] * self.downsample_rate)
        feat = F_.CastFn.apply(feat, vtx.compute_dtype())
        pred = F_.LinearFn.apply(feat, self.decoder_pred.weight, self.decoder_pred.bias)
        pred = pred[:, 1:, :].contiguous()                          # [B, tq*g*g, ts*Cf]
        B = pred.shape[0]
        cf = pred.shape[-1] // ts
        cmask, keep = center_frame_mask(mask, cube_marker, self.num_frames, ts)
        target = target_x.to(device=pred.device, dtype=torch.float64).contiguous()
        loss = _MaskedMSEFn.apply(pred, target, cmask, (B, tq, ts, g, cf))
        x = F_.CastFn.apply(pred, torch.float32).reshape(B, tq, g, g, ts, cf).permute(0, 1, 4, 2, 3, 5)
        x = x.reshape(B, self.num_frames, g, g, cf)
        if visualize:
            # reference :904-907, kept literal: `center_index` is reset after every sample of the loop above it (:892), so what
            # the reference returns is the EMPTY selection x[:, no frame] rearranged to 'b t (h dh) (w dw) c o' and the all-False
            # index ("need to update" in the reference); consumers get the same shapes and values here
            center_index = torch.zeros(self.num_frames, dtype=torch.bool, device=x.device)
            mp = x[:, center_index]
            mp = mp.reshape(B, mp.shape[1], g, g, 2, 2, 3, 9).permute(0, 1, 2, 4, 3, 5, 6, 7).reshape(B, mp.shape[1], 2 * g, 2 * g, 3, 9)
            return x, loss, mp, center_index
        return x, loss

    def forward(self, x, target_x, mask, cube_marker, visualize=False):
        feat = self.forward_features(x, mask)
        return self.head_loss(feat, target_x, mask, cube_marker, visualize)
