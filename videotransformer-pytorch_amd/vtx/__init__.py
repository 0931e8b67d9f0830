"""vtx -- Python binding of libvtx.so, the MI355X video-transformer hot path.

Precision policy (the reference trains under Lightning ``precision=16`` autocast,
model_pretrain.py:203; BASELINE asks for bf16):
    'auto' (default): bfloat16 kernels inside ``torch.autocast('cuda')``, exact
                      float32 kernels otherwise;
    'fp32' / 'bf16' : force one path.
"""
import torch

from . import _lib, ops, functions  # noqa: F401
from ._lib import VtxError, load, set_option  # noqa: F401
from .ops import set_input_normalization  # noqa: F401

_precision = 'auto'


def set_precision(p):
    global _precision
    if p not in ('auto', 'fp32', 'bf16'):
        raise ValueError(p)
    _precision = p


def get_precision():
    return _precision


def compute_dtype():
    if _precision == 'fp32':
        return torch.float32
    if _precision == 'bf16':
        return torch.bfloat16
    return torch.bfloat16 if torch.is_autocast_enabled() else torch.float32


_recompute = False


def set_recompute(on):
    """Re-run every transformer block in backward instead of keeping its activations (torch.utils.checkpoint
    around BasicTransformerBlock): trades ~1/3 more block compute for the ~0.8 GB per clip and layer that
    TimeSformer-L on 96-frame clips saves for backward (BASELINE.json configs[4]; 288 GB holds ~12 such clips
    without it).  The reference has no equivalent; off by default."""
    global _recompute
    _recompute = bool(on)


def recompute_enabled():
    return _recompute


def set_stream(kind):
    """How the bf16 path keeps the residual stream: 'bf16' (default: every sub-block's x + f(x) is rounded to bf16 -- the fastest
    form, whose rounding of the running sum grows with sqrt(depth): 2x the reference's own autocast deviation on outputs at 12
    layers, 3x at 24) or 'fp32' (the exact stream: sub-blocks hand on their contribution, the running sum lives in float32 and is
    advanced inside the next LayerNorm kernel -- what torch.autocast does in the reference; +0.7 GB of HBM traffic per sub-block at
    96 clips, +4 % step time) or 'fp32+grad' (the exact stream AND its gradient in float32 through the backward: the LayerNorm
    backward of every sub-block adds its term to the float32 gradient of the stream and hands on the sum and its one bf16 rounding;
    +7 % step time in all; parameter gradients at 12 layers: worst 1.4e-2 -> 1.2e-2, median 8.5e-3 -> 7.2e-3 off the fp32 reference).  Every attention type of
    TimeSformer / ViViT; the float32 precision mode is its own exact stream."""
    if kind not in ('bf16', 'fp32', 'fp32+grad'):
        raise ValueError(kind)
    functions.set_exact_stream(kind != 'bf16')
    # 'fp32+grad': the stream's GRADIENT stays float32 through the backward as well (vtx_layernorm_bwd_g32): both directions of
    # `x = x + f(norm(x))` as torch.autocast runs them; +6 bytes per element and sub-block on the LayerNorm backward
    functions.set_exact_grad_stream(kind == 'fp32+grad')


def get_stream():
    return ('fp32+grad' if functions.exact_grad_stream() else 'fp32') if functions.exact_stream() else 'bf16'


import os as _os  # noqa: E402

if _os.environ.get('VTX_STREAM', 'bf16') in ('fp32', 'fp32+grad'):      # initial value for entry points that keep the reference's flag list (model_pretrain.py)
    set_stream(_os.environ['VTX_STREAM'])
