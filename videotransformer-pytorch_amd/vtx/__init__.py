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
