"""Parameter initialisers used by the drop-in modules (reference weight_init.py:31-103).

Only the init helpers the hot path needs are provided.  The pretrained-checkpoint
importers of the reference (``init_from_vit_pretrain_`` & co., weight_init.py:107-315)
are a "next" row of the scope table and raise here.
The ``state_dict`` key contract those importers target is kept by transformer.py /
video_transformer.py, so a checkpoint saved by the reference loads with
``load_state_dict(strict=True)``.
"""
import math
import warnings

import torch
import torch.nn as nn


@torch.no_grad()
def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """Truncated normal by inverse-CDF sampling of a uniform on [cdf(a), cdf(b)]."""
    if (mean < a - 2 * std) or (mean > b + 2 * std):
        warnings.warn('mean is more than 2 std from [a, b] in trunc_normal_; the distribution may be off.',
                      stacklevel=2)
    cdf = lambda v: (1. + math.erf(v / math.sqrt(2.))) / 2.
    lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
    tensor.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_()
    tensor.mul_(std * math.sqrt(2.)).add_(mean).clamp_(min=a, max=b)
    return tensor


@torch.no_grad()
def constant_init_(tensor, constant_value=0):
    nn.init.constant_(tensor, constant_value)


@torch.no_grad()
def kaiming_init_(tensor, a=0, mode='fan_out', nonlinearity='relu', distribution='normal'):
    assert distribution in ['uniform', 'normal']
    fn = nn.init.kaiming_uniform_ if distribution == 'uniform' else nn.init.kaiming_normal_
    fn(tensor, a=a, mode=mode, nonlinearity=nonlinearity)


def _not_yet(name):
    def f(*args, **kwargs):
        raise NotImplementedError(
            f'vtx: {name} (pretrained-checkpoint import, reference weight_init.py) is outside the round-1 '
            'hot-path scope; load a reference-format state_dict with load_state_dict instead')
    f.__name__ = name
    return f


init_from_vit_pretrain_ = _not_yet('init_from_vit_pretrain_')
init_from_mae_pretrain_ = _not_yet('init_from_mae_pretrain_')
init_from_kinetics_pretrain_ = _not_yet('init_from_kinetics_pretrain_')
