"""Drop-in replacement for the reference ``transformer.py`` (block / operator
library) whose arithmetic runs in libvtx.so HIP kernels for MI355X (gfx950).

Same class names, constructor arguments, ``forward`` signatures and
``state_dict`` keys as the reference; the bodies are new:
each module's forward is one ``torch.autograd.Function`` from ``vtx.functions``
issuing a short chain of HIP kernels.  There is no PyTorch/CPU fallback -- CPU
tensors raise (the CPU reference lives in oracle/, used by the tests only).

Not carried over (raise ``NotImplementedError`` when exercised): non-zero
dropout probabilities (the reference models always pass 0), norm / activation
classes other than nn.LayerNorm / nn.GELU, FFN num_layers != 2, and the
``use_cls_token`` variants that the shipped operator_order never instantiates
(temporal op with cls / spatial op without cls).
"""
import numpy as np

import torch
import torch.nn as nn
import torch.utils.checkpoint
from torch.nn.modules.utils import _pair

import vtx
from vtx import functions as F_
from weight_init import trunc_normal_, constant_init_, kaiming_init_


def get_sine_cosine_pos_emb(n_position, d_hid):
    """Sinusoid position table [1, n_position, d_hid] (reference transformer.py:12-22):
    angle(pos, j) = pos / 10000^(2*(j//2)/d_hid); sin on even j, cos on odd j."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table).unsqueeze(0)


def _to_compute(x):
    """Bring an activation into the active compute dtype (fp32 <-> bf16 boundary)."""
    return F_.CastFn.apply(x, vtx.compute_dtype())


def _build_norm(norm_layer, embed_dims):
    """norm_layer is nn.LayerNorm or a callable producing one (e.g. functools.partial(nn.LayerNorm, eps=1e-6),
    the timm idiom); anything else has no HIP kernel."""
    norm = norm_layer(embed_dims)
    if type(norm) is not nn.LayerNorm or not norm.elementwise_affine or tuple(norm.normalized_shape) != (embed_dims,):
        raise NotImplementedError('vtx: only nn.LayerNorm (elementwise affine, over embed_dims) is supported as norm_layer')
    return norm


def _no_dropout(p, what):
    if p:
        raise NotImplementedError(f'vtx: {what} > 0 is not supported by the HIP path '
                                  '(the reference models always use 0)')


class DropPath(nn.Module):
    """Stochastic depth per sample (reference transformer.py:25-42).

    Inside the fused blocks the mask is applied in the GEMM epilogue; this module
    supplies it through ``scale_vector`` -- drawn from the CPU default generator
    with the reference's shape and order, so training trajectories can be
    reproduced draw for draw.  Nothing is drawn when dropout_p == 0 or in eval."""

    def __init__(self, dropout_p=None):
        super().__init__()
        self.dropout_p = dropout_p

    def scale_vector(self, rows, ndim, device):
        p = self.dropout_p
        if not p or not self.training:
            return None
        keep = 1 - p
        u = torch.rand((rows,) + (1,) * (ndim - 1))          # the reference's draw (shape and generator)
        # mask arithmetic in NumPy, not ATen: on many-core hosts with a cgroup CPU quota the ATen
        # intra-op pool (128 threads on the bench box, quota 16 CPUs) gets the whole process
        # CFS-throttled for ~90 ms at a time when tiny CPU ops run between kernel launches
        scale = np.floor(np.float32(keep) + u.numpy().reshape(rows)) / np.float32(keep)
        scale = torch.from_numpy(scale.astype(np.float32, copy=False))
        if device.type != 'cuda':
            return scale
        # no hipMemcpy: a copy kernel reads the mask from pinned host memory (see vtx.ops.upload_f32)
        dev = vtx.ops.upload_f32(scale, device)
        dev._vtx_host = scale                                # the host copy: lets the FFN skip the dropped clips' rows
        return dev

    def forward(self, x):
        s = self.scale_vector(x.shape[0], x.ndim, x.device)
        if s is None:
            return x
        rows_per = x.numel() // x.shape[-1] // x.shape[0]
        return F_.RowScaleFn.apply(x, s, rows_per)


def _stream_of(t):
    """(float32 stream, exact?) for a block input under vtx.set_stream('fp32'): the stream rides on the contribution tensor as an
    attribute (it is a side buffer, not an autograd tensor -- vtx/functions.py, "the exact residual stream").  Exact mode is the
    bf16 path only; float32 activations are their own exact stream."""
    exact = F_.exact_stream() and t.dtype == torch.bfloat16
    return (getattr(t, '_vtx_xs', None) if exact else None), exact


def _with_stream(res, exact):
    """What a block hands on: the contribution with the new float32 stream attached (exact), or the bf16 stream itself."""
    if not exact:
        return res
    out, x32 = res
    out._vtx_xs = x32
    return out


def stream_value(t):
    """The stream as one tensor for consumers outside the blocks: t itself, or bf16(float32 stream + contribution) under the
    exact stream (vtx.functions.StreamValueFn)."""
    xs, exact = _stream_of(t)
    if not exact or xs is None:
        return t
    return F_.StreamValueFn.apply(t, xs)


def _no_exact(what):
    if F_.exact_stream() and vtx.compute_dtype() == torch.bfloat16:
        raise NotImplementedError(f"vtx.set_stream('fp32') does not cover {what}")


def _drop_scale(layer_drop, rows, ndim, device):
    return layer_drop.scale_vector(rows, ndim, device) if isinstance(layer_drop, DropPath) else None


def _keep_scale(layer_drop):
    """The non-zero value of scale_vector(): 1 / keep, in the same float32 arithmetic."""
    return float(np.float32(1.0) / np.float32(1 - layer_drop.dropout_p))


def _build_layer_drop(layer_drop):
    # the reference pops both keys from the caller's dict (transformer.py:221-222)
    p = layer_drop.pop('dropout_p')
    kind = layer_drop.pop('type')
    return kind(p) if kind else nn.Identity()


class ClassificationHead(nn.Module):
    """Linear classifier on the clip feature (reference transformer.py:45-80)."""

    def __init__(self, num_classes, in_channels, init_std=0.02, eval_metrics='finetune', **kwargs):
        super().__init__()
        self.init_std = init_std
        self.eval_metrics = eval_metrics
        self.cls_head = nn.Linear(in_channels, num_classes)
        self.init_weights(self.cls_head)

    def init_weights(self, module):
        if getattr(module, 'weight', None) is not None:
            if self.eval_metrics == 'finetune':
                trunc_normal_(module.weight, std=self.init_std)
            else:
                module.weight.data.normal_(mean=0.0, std=0.01)
        if getattr(module, 'bias', None) is not None:
            constant_init_(module.bias, constant_value=0)

    def forward(self, x):
        y = F_.LinearFn.apply(_to_compute(x), self.cls_head.weight, self.cls_head.bias)
        return F_.CastFn.apply(y, torch.float32)


class PatchEmbed(nn.Module):
    """Non-overlapping patch / tubelet projection (reference transformer.py:83-151).
    The Conv2d / Conv3d module only holds the parameters (state_dict key
    ``projection.{weight,bias}``); the arithmetic is vtx_patch_rows + vtx_gemm_nt."""

    def __init__(self, img_size, patch_size, tube_size=2, in_channels=3, embed_dims=768, conv_type='Conv2d'):
        super().__init__()
        self.img_size = _pair(img_size)
        self.patch_size = _pair(patch_size)
        self.num_patches = (self.img_size[1] // self.patch_size[1]) * (self.img_size[0] // self.patch_size[0])
        if conv_type == 'Conv2d':
            self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size=patch_size, stride=patch_size)
        elif conv_type == 'Conv3d':
            self.projection = nn.Conv3d(in_channels, embed_dims,
                                        kernel_size=(tube_size, patch_size, patch_size),
                                        stride=(tube_size, patch_size, patch_size))
        else:
            raise TypeError(f'Unsupported conv layer type {conv_type}')
        self.init_weights(self.projection)

    def init_weights(self, module):
        if getattr(module, 'weight', None) is not None:
            kaiming_init_(module.weight, mode='fan_in', nonlinearity='relu')
        if getattr(module, 'bias', None) is not None:
            constant_init_(module.bias, constant_value=0)

    def forward(self, x):
        if type(self.projection) not in (nn.Conv2d, nn.Conv3d):
            raise TypeError(f'Unsupported conv layer type {type(self.projection)}')
        return F_.PatchEmbedFn.apply(x, self.projection.weight, self.projection.bias, vtx.compute_dtype())


class Attention(nn.Module):
    """qkv Linear -> softmax(q k^T * scale) v -> proj Linear; returns (x, attn)
    (reference transformer.py:153-177)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        if qk_scale is not None and qk_scale != head_dim ** -0.5:
            raise NotImplementedError('vtx: custom qk_scale is not supported')

    def forward(self, x):
        _no_dropout(self.attn_drop.p if self.training else 0, 'attn_drop')
        _no_dropout(self.proj_drop.p if self.training else 0, 'proj_drop')
        x = _to_compute(x)
        qkv = F_.LinearFn.apply(x, self.qkv.weight, self.qkv.bias)
        ctx, probs = F_.AttnCoreFn.apply(qkv, self.num_heads, True)
        return F_.LinearFn.apply(ctx, self.proj.weight, self.proj.bias), probs


class _DividedBase(nn.Module):
    def __init__(self, embed_dims, num_heads, num_frames, use_cls_token, attn_drop, proj_drop, layer_drop,
                 norm_layer):
        super().__init__()
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.num_frames = num_frames
        self.use_cls_token = use_cls_token
        self.norm = _build_norm(norm_layer, embed_dims)
        self.attn = Attention(embed_dims, num_heads, qkv_bias=True, attn_drop=attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self.layer_drop = _build_layer_drop(layer_drop)

    def _guard(self):
        if self.training:
            _no_dropout(self.proj_drop.p, 'proj_drop')
            _no_dropout(self.attn.attn_drop.p, 'attn_drop')


class DividedTemporalAttentionWithPreNorm(_DividedBase):
    """Temporal half of divided space-time attention (reference transformer.py:179-282)."""

    def __init__(self, embed_dims, num_heads, num_frames, use_cls_token, attn_drop=0., proj_drop=0.,
                 layer_drop=dict(type=DropPath, dropout_p=0.1), norm_layer=nn.LayerNorm, **kwargs):
        super().__init__(embed_dims, num_heads, num_frames, use_cls_token, attn_drop, proj_drop, layer_drop,
                         norm_layer)
        if not use_cls_token:
            self.temporal_fc = nn.Linear(self.embed_dims, self.embed_dims)
            self.init_weights(self.temporal_fc)

    def init_weights(self, module):
        if getattr(module, 'weight', None) is not None:
            constant_init_(module.weight, constant_value=0)
        if getattr(module, 'bias', None) is not None:
            constant_init_(module.bias, constant_value=0)

    def forward(self, query, key=None, value=None, residual=None, return_attention=False, **kwargs):
        assert residual is None, 'Always adding the shortcut in the forward function'
        if self.use_cls_token:
            raise NotImplementedError('vtx: temporal attention over the cls token (use_cls_token=True) is a '
                                      'dead branch of the reference models and is not implemented')
        self._guard()
        x = _to_compute(query)
        xs, exact = _stream_of(query)
        b, n1, d = x.shape
        t = self.num_frames
        if (n1 - 1) % t:
            raise ValueError(f'{n1 - 1} tokens per clip are not a multiple of num_frames={t}')
        p = (n1 - 1) // t
        if return_attention:
            _no_exact('the attention map of a temporal block')
            tok = x[:, 1:].reshape(b * p, t, d)
            out = F_.SelfAttnFn.apply(tok, self.norm.weight, self.norm.bias, self.attn.qkv.weight,
                                      self.attn.qkv.bias, self.attn.proj.weight, self.attn.proj.bias,
                                      self.num_heads, None, True, self.norm.eps)
            return out
        s = _drop_scale(self.layer_drop, b * p, 3, x.device)
        return _with_stream(F_.TimeAttnFn.apply(
            x, self.norm.weight, self.norm.bias, self.attn.qkv.weight, self.attn.qkv.bias, self.attn.proj.weight,
            self.attn.proj.bias, self.temporal_fc.weight, self.temporal_fc.bias, t, self.num_heads, s, self.norm.eps,
            _keep_scale(self.layer_drop) if s is not None else None, xs, exact), exact)


class DividedSpatialAttentionWithPreNorm(_DividedBase):
    """Spatial half of divided space-time attention (reference transformer.py:285-382)."""

    def __init__(self, embed_dims, num_heads, num_frames, use_cls_token, attn_drop=0., proj_drop=0.,
                 layer_drop=dict(type=DropPath, dropout_p=0.1), norm_layer=nn.LayerNorm, **kwargs):
        super().__init__(embed_dims, num_heads, num_frames, use_cls_token, attn_drop, proj_drop, layer_drop,
                         norm_layer)
        self.init_weights()

    def init_weights(self):
        pass

    def forward(self, query, key=None, value=None, residual=None, return_attention=False, **kwargs):
        assert residual is None, 'Always adding the shortcut in the forward function'
        if not self.use_cls_token:
            raise NotImplementedError('vtx: spatial attention without the cls token (use_cls_token=False) is a '
                                      'dead branch of the reference models and is not implemented')
        self._guard()
        x = _to_compute(query)
        xs, exact = _stream_of(query)
        b = x.shape[0]
        t = self.num_frames
        if (x.shape[1] - 1) % t:
            raise ValueError(f'{x.shape[1] - 1} tokens per clip are not a multiple of num_frames={t}')
        s = None if return_attention else _drop_scale(self.layer_drop, b * t, 3, x.device)
        res = F_.SpaceAttnFn.apply(x, self.norm.weight, self.norm.bias, self.attn.qkv.weight, self.attn.qkv.bias,
                                   self.attn.proj.weight, self.attn.proj.bias, t, self.num_heads, s,
                                   bool(return_attention), self.norm.eps, xs, exact)
        return res if return_attention else _with_stream(res, exact)


class MultiheadAttentionWithPreNorm(nn.Module):
    """Pre-norm self attention with residual (reference transformer.py:385-456)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., norm_layer=nn.LayerNorm,
                 layer_drop=dict(type=DropPath, dropout_p=0.), batch_first=False, **kwargs):
        super().__init__()
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.norm = _build_norm(norm_layer, embed_dims)
        self.attn = Attention(embed_dims, num_heads, qkv_bias=True, attn_drop=attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self.layer_drop = _build_layer_drop(layer_drop)

    def forward(self, query, key=None, value=None, residual=None, attn_mask=None, key_padding_mask=None,
                return_attention=False, **kwargs):
        if self.training:
            _no_dropout(self.proj_drop.p, 'proj_drop')
            _no_dropout(self.attn.attn_drop.p, 'attn_drop')
        x = _to_compute(query)
        xs, exact = _stream_of(query)
        s = None if return_attention else _drop_scale(self.layer_drop, x.shape[0], 3, x.device)
        res = F_.SelfAttnFn.apply(x, self.norm.weight, self.norm.bias, self.attn.qkv.weight, self.attn.qkv.bias,
                                  self.attn.proj.weight, self.attn.proj.bias, self.num_heads, s,
                                  bool(return_attention), self.norm.eps, xs, exact)
        return res if return_attention else _with_stream(res, exact)


class FFNWithPreNorm(nn.Module):
    """Pre-norm MLP with residual (reference transformer.py:459-523)."""

    def __init__(self, embed_dims=256, hidden_channels=1024, num_layers=2, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, dropout_p=0., layer_drop=None, **kwargs):
        super().__init__()
        assert num_layers >= 2, f'num_layers should be no less than 2. got {num_layers}.'
        if num_layers != 2 or act_layer is not nn.GELU:
            raise NotImplementedError('vtx: FFN supports num_layers=2 and nn.GELU only')
        self.embed_dims = embed_dims
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self.norm = _build_norm(norm_layer, embed_dims)
        layers = []
        in_channels = embed_dims
        for _ in range(num_layers - 1):
            layers.append(nn.Sequential(nn.Linear(in_channels, hidden_channels), act_layer(), nn.Dropout(dropout_p)))
            in_channels = hidden_channels
        layers.append(nn.Linear(hidden_channels, embed_dims))
        layers.append(nn.Dropout(dropout_p))
        self.layers = nn.ModuleList(layers)
        self.dropout_p = dropout_p
        self.layer_drop = _build_layer_drop(layer_drop) if layer_drop else nn.Identity()

    def forward(self, x):
        if self.training:
            _no_dropout(self.dropout_p, 'dropout_p')
        xs, exact = _stream_of(x)
        x = _to_compute(x)
        s = _drop_scale(self.layer_drop, x.shape[0], x.ndim, x.device)
        fc1, fc2 = self.layers[0][0], self.layers[1]
        return _with_stream(F_.FFNFn.apply(x, self.norm.weight, self.norm.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, s,
                                           self.norm.eps, xs, exact), exact)


class BasicTransformerBlock(nn.Module):
    """One block = the attention operators in ``operator_order`` then the FFNs
    (reference transformer.py:568-636)."""

    def __init__(self, embed_dims, num_heads, num_frames, hidden_channels, operator_order,
                 norm_layer=nn.LayerNorm, act_layer=nn.GELU, num_layers=2, dpr=0):
        super().__init__()
        self.attentions = nn.ModuleList([])
        self.ffns = nn.ModuleList([])
        last_attn = len(operator_order) - 2
        for i, operator in enumerate(operator_order):
            drop = dict(type=DropPath, dropout_p=dpr)
            if operator == 'self_attn':
                self.attentions.append(MultiheadAttentionWithPreNorm(
                    embed_dims=embed_dims, num_heads=num_heads, batch_first=True, norm_layer=nn.LayerNorm,
                    layer_drop=drop))
            elif operator == 'time_attn':
                self.attentions.append(DividedTemporalAttentionWithPreNorm(
                    embed_dims=embed_dims, num_heads=num_heads, num_frames=num_frames, norm_layer=norm_layer,
                    use_cls_token=(i == last_attn), layer_drop=drop))
            elif operator == 'space_attn':
                self.attentions.append(DividedSpatialAttentionWithPreNorm(
                    embed_dims=embed_dims, num_heads=num_heads, num_frames=num_frames, norm_layer=norm_layer,
                    use_cls_token=(i == last_attn), layer_drop=drop))
            elif operator == 'ffn':
                self.ffns.append(FFNWithPreNorm(
                    embed_dims=embed_dims, hidden_channels=hidden_channels, num_layers=num_layers,
                    act_layer=act_layer, norm_layer=norm_layer, layer_drop=drop))
            else:
                raise TypeError(f'Unsupported operator type {operator}')

    def forward(self, x, return_attention=False):
        n = len(self.attentions)
        for idx, layer in enumerate(self.attentions):
            if return_attention and idx >= n - 1:
                return layer(x, return_attention=True)
            x = layer(x)
        for layer in self.ffns:
            x = layer(x)
        return x


class TransformerContainer(nn.Module):
    """Stack of blocks with linearly increasing DropPath (reference transformer.py:526-565)."""

    def __init__(self, num_transformer_layers, embed_dims, num_heads, num_frames, hidden_channels,
                 operator_order, drop_path_rate=0.1, norm_layer=nn.LayerNorm, act_layer=nn.GELU, num_layers=2):
        super().__init__()
        self.layers = nn.ModuleList([])
        self.num_transformer_layers = num_transformer_layers
        dpr = np.linspace(0, drop_path_rate, num_transformer_layers)
        for i in range(num_transformer_layers):
            self.layers.append(BasicTransformerBlock(
                embed_dims=embed_dims, num_heads=num_heads, num_frames=num_frames,
                hidden_channels=hidden_channels, operator_order=operator_order, norm_layer=norm_layer,
                act_layer=act_layer, num_layers=num_layers, dpr=dpr[i]))

    def forward(self, x, return_attention=False):
        last = self.num_transformer_layers - 1
        recompute = vtx.recompute_enabled() and torch.is_grad_enabled() and not return_attention
        for idx, layer in enumerate(self.layers):
            if return_attention and idx >= last:
                x = layer(x, return_attention=True)
            elif recompute:
                # the CPU generator state is saved and restored around the re-run: DropPath draws the same masks.  (Under the exact
                # stream the block's input carries its float32 stream as an attribute; the re-run reads it from the same object.)
                x = torch.utils.checkpoint.checkpoint(layer, x, use_reentrant=False)
            else:
                x = layer(x)
        return x
