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
