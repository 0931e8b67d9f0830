"""Parameter initialisers and pretrained-checkpoint importers of the drop-in modules
(reference weight_init.py:17-315).

The importers are host-side ``state_dict`` key surgery: each ``init_from_*_pretrain_`` is a pure
``remap_*`` function (old dict -> new dict, unit-tested against the reference's own functions in
tests/test_weight_import.py) followed by ``module.load_state_dict(strict=False)`` -- including the
reference's quirks, which decide what a published checkpoint initialises:
  * divided space-time: the image ViT's attention lands in ``attentions.0`` (the TEMPORAL operator)
    and is copied to / zeroed for ``attentions.1`` (weight_init.py:166-174);
  * MAE checkpoints: the attention qkv / proj WEIGHTS keep their ``...layers.N.attn.*`` names (the
    renames are commented out in the reference, weight_init.py:244-247) and therefore end up in
    ``unexpected_keys``; only the q/v biases are imported.
"""
import math
import re
import warnings

import torch
import torch.nn as nn

from utils import print_on_rank_zero


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




# ------------------------------------------------------------------------------------------
# pretrained-checkpoint import
# ------------------------------------------------------------------------------------------
def show_state_dict(state_dict):
    for name in state_dict:
        print(name)


def _rename_mha(key):
    """nn.MultiheadAttention parameter names -> this repo's Attention (reference weight_init.py:22-25,148-151)."""
    if 'in_proj' in key:
        return key.replace('in_proj_', 'qkv.')
    if 'out_proj' in key:
        return key.replace('out_proj', 'proj')
    return key


def replace_state_dict(state_dict):
    """Lightning checkpoint of this repo's trainer -> bare model keys, IN PLACE (reference :17-28):
    'model.<k>' -> '<k>' (+ MultiheadAttention renames); everything else is the classification head,
    saved as 'cls_head.<k>', and loses its first 9 characters."""
    for old in list(state_dict.keys()):
        new = _rename_mha(old[6:]) if old.startswith('model') else old[9:]
        state_dict[new] = state_dict.pop(old)


def _inflate_conv2d(weight, tube_size, extend_strategy):
    """[D,C,h,w] image patch kernel -> [D,C,t,h,w] tubelet kernel (reference :128-136).
    'temporal_avg': every frame gets weight / t.  'center_frame': the reference zeroes einops' stride-0
    expanded view IN PLACE, which also zeroes the source kernel it then copies into the centre frame
    (:134-135) -- what it really imports is an all-zero kernel; reproduced, not repaired."""
    w3 = weight.unsqueeze(2).repeat(1, 1, tube_size, 1, 1)
    if extend_strategy == 'temporal_avg':
        w3 = w3 / tube_size
    elif extend_strategy == 'center_frame':
        w3.zero_()
    return w3


_LAYER_IDX = re.compile(r'(?<=layers.)\d+')


def _seed_second_operator(sd, attention_type, copy_strategy, num_time_transformer_layers):
    """Second pass of both image-checkpoint importers (reference :160-181 / :278-299): initialise the
    operators an image model does not have from the ones it has."""
    def seeded(t):
        if copy_strategy == 'repeat':
            return t.clone()
        if copy_strategy == 'set_zero':
            return t.clone().zero_()
        return None
    for key in list(sd.keys()):
        new = None
        if attention_type == 'divided_space_time':
            if 'attentions.0' in key:
                new = key.replace('attentions.0', 'attentions.1')
        elif attention_type == 'fact_encoder':
            idx = _LAYER_IDX.findall(key)
            if len(idx) > 1 and int(idx[1]) < num_time_transformer_layers:
                new = key.replace('transformer_layers.0.layers', 'transformer_layers.1.layers')
        if new is not None:
            t = seeded(sd[key])
            if t is not None:
                sd[new] = t


def remap_vit_state_dict(state_dict, conv_type, attention_type, copy_strategy, extend_strategy='temporal_avg',
                         tube_size=2, num_time_transformer_layers=4):
    """ImageNet ViT checkpoint (mmaction-style keys) -> this repo's keys (reference :107-181)."""
    out = {}
    for key, val in state_dict.items():
        if conv_type == 'Conv3d' and 'patch_embed.projection.weight' in key:
            out[key] = _inflate_conv2d(val, tube_size, extend_strategy)
            continue
        new = key.replace('transformer_layers.layers', 'transformer_layers.0.layers') if attention_type == 'fact_encoder' else key
        new = _rename_mha(new)
        if 'norms' in new:
            new = new.replace('norms.0', 'attentions.0.norm').replace('norms.1', 'ffns.0.norm')
        out[new] = val
    _seed_second_operator(out, attention_type, copy_strategy, num_time_transformer_layers)
    return out


def remap_mae_state_dict(state_dict, conv_type, attention_type, copy_strategy, extend_strategy='temporal_avg',
                         tube_size=2, num_time_transformer_layers=4):
    """MAE / VideoMAE-style encoder checkpoint -> this repo's keys (reference :186-299), quirks included."""
    blocks = 'transformer_layers.0.layers' if attention_type == 'fact_encoder' else 'transformer_layers.layers'
    out = {}
    consumed = set()
    for key, val in state_dict.items():
        if key in consumed or 'decoder' in key:
            continue
        if 'encoder.patch_embed.proj' in key:
            new = key.replace('encoder.patch_embed.proj', 'patch_embed.projection')
            out[new] = _inflate_conv2d(val, tube_size, extend_strategy) if (conv_type == 'Conv3d' and 'weight' in key) else val
            continue
        new = key.replace('encoder.blocks', blocks)
        if 'norm' in new:
            new = new.replace('norm1', 'attentions.0.norm').replace('norm2', 'ffns.0.norm')
        elif 'attn' in new:
            if 'q_bias' in new:                    # (q_bias, 0, v_bias) -> the fused qkv bias; k has none
                prefix = key[:key.index('attn.q_bias')]
                q, v = state_dict[prefix + 'attn.q_bias'], state_dict[prefix + 'attn.v_bias']
                out[new.replace('attn.q_bias', 'attentions.0.attn.qkv.bias')] = torch.cat((q, torch.zeros_like(q), v))
                consumed.add(prefix + 'attn.v_bias')
                continue
            if 'v_bias' in new:                    # consumed with its q_bias; an orphan keeps its old name (reference :262)
                if (key[:key.index('attn.v_bias')] + 'attn.q_bias') not in state_dict:
                    out[key] = val
                continue
        elif 'mlp' in new:
            new = new.replace('mlp.fc1', 'ffns.0.layers.0.0').replace('mlp.fc2', 'ffns.0.layers.1')
        if 'encoder.norm' in key:
            new = key.replace('encoder.norm', 'norm')
        out[new] = val
    _seed_second_operator(out, attention_type, copy_strategy, num_time_transformer_layers)
    return out


def _load_checkpoint(path, inner_key):
    sd = torch.load(path) if torch.cuda.is_available() else torch.load(path, map_location=torch.device('cpu'))
    return sd[inner_key] if inner_key in sd else sd


@torch.no_grad()
def init_from_vit_pretrain_(module, pretrained, conv_type, attention_type, copy_strategy,
                            extend_strategy='temporal_avg', tube_size=2, num_time_transformer_layers=4):
    if not isinstance(pretrained, str):
        return
    sd = remap_vit_state_dict(_load_checkpoint(pretrained, 'state_dict'), conv_type, attention_type, copy_strategy,
                              extend_strategy, tube_size, num_time_transformer_layers)
    missing_keys, unexpected_keys = module.load_state_dict(sd, strict=False)
    print_on_rank_zero(f'missing_keys:{missing_keys}\n unexpected_keys:{unexpected_keys}')


@torch.no_grad()
def init_from_mae_pretrain_(module, pretrained, conv_type, attention_type, copy_strategy,
                            extend_strategy='temporal_avg', tube_size=2, num_time_transformer_layers=4):
    if not isinstance(pretrained, str):
        return
    sd = remap_mae_state_dict(_load_checkpoint(pretrained, 'model'), conv_type, attention_type, copy_strategy,
                              extend_strategy, tube_size, num_time_transformer_layers)
    missing_keys, unexpected_keys = module.load_state_dict(sd, strict=False)
    print_on_rank_zero(f'missing_keys:{missing_keys}\n unexpected_keys:{unexpected_keys}')


def init_from_kinetics_pretrain_(module, pretrain_pth):
    sd = _load_checkpoint(pretrain_pth, 'state_dict')
    replace_state_dict(sd)
    print_on_rank_zero(module.load_state_dict(sd, strict=False))
