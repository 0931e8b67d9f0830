"""Synthetic checkpoints and cases shared by tests/golden/make_golden_import.py and
tests/test_weight_import.py (TEST INFRASTRUCTURE)."""
import zlib

import torch

D, L = 16, 6          # embed dim / layers of the toy checkpoints (key structure is what matters)


def _t(key, *shape):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    return torch.randn(*shape, generator=g)


def synth_checkpoint(kind):
    sd = {}
    if kind == 'vit':               # mmaction2-style ImageNet ViT (nn.MultiheadAttention names, norms.N)
        sd['cls_token'] = _t('cls', 1, 1, D)
        sd['pos_embed'] = _t('pos', 1, 5, D)
        sd['patch_embed.projection.weight'] = _t('pe.w', D, 3, 4, 4)
        sd['patch_embed.projection.bias'] = _t('pe.b', D)
        for i in range(L):
            p = f'transformer_layers.layers.{i}.'
            sd[p + 'attentions.0.attn.in_proj_weight'] = _t(p + 'ipw', 3 * D, D)
            sd[p + 'attentions.0.attn.in_proj_bias'] = _t(p + 'ipb', 3 * D)
            sd[p + 'attentions.0.attn.out_proj.weight'] = _t(p + 'opw', D, D)
            sd[p + 'attentions.0.attn.out_proj.bias'] = _t(p + 'opb', D)
            sd[p + 'ffns.0.layers.0.0.weight'] = _t(p + 'f1w', 4 * D, D)
            sd[p + 'ffns.0.layers.0.0.bias'] = _t(p + 'f1b', 4 * D)
            sd[p + 'ffns.0.layers.1.weight'] = _t(p + 'f2w', D, 4 * D)
            sd[p + 'ffns.0.layers.1.bias'] = _t(p + 'f2b', D)
            for j in (0, 1):
                sd[p + f'norms.{j}.weight'] = _t(p + f'n{j}w', D)
                sd[p + f'norms.{j}.bias'] = _t(p + f'n{j}b', D)
        sd['norm.weight'] = _t('nw', D)
        sd['norm.bias'] = _t('nb', D)
        return {'state_dict': sd}
    if kind == 'mae':               # MAE / BEiT-style encoder + decoder
        sd['encoder.cls_token'] = _t('cls', 1, 1, D)
        sd['encoder.patch_embed.proj.weight'] = _t('pe.w', D, 3, 4, 4)
        sd['encoder.patch_embed.proj.bias'] = _t('pe.b', D)
        sd['mask_token'] = _t('mt', 1, 1, D)
        for i in range(L):
            p = f'encoder.blocks.{i}.'
            sd[p + 'norm1.weight'] = _t(p + 'n1w', D)
            sd[p + 'norm1.bias'] = _t(p + 'n1b', D)
            sd[p + 'attn.q_bias'] = _t(p + 'qb', D)
            sd[p + 'attn.v_bias'] = _t(p + 'vb', D)
            sd[p + 'attn.qkv.weight'] = _t(p + 'qkvw', 3 * D, D)
            sd[p + 'attn.proj.weight'] = _t(p + 'pw', D, D)
            sd[p + 'attn.proj.bias'] = _t(p + 'pb', D)
            sd[p + 'norm2.weight'] = _t(p + 'n2w', D)
            sd[p + 'norm2.bias'] = _t(p + 'n2b', D)
            sd[p + 'mlp.fc1.weight'] = _t(p + 'f1w', 4 * D, D)
            sd[p + 'mlp.fc1.bias'] = _t(p + 'f1b', 4 * D)
            sd[p + 'mlp.fc2.weight'] = _t(p + 'f2w', D, 4 * D)
            sd[p + 'mlp.fc2.bias'] = _t(p + 'f2b', D)
        sd['encoder.norm.weight'] = _t('nw', D)
        sd['encoder.norm.bias'] = _t('nb', D)
        sd['decoder.blocks.0.norm1.weight'] = _t('dec', D)
        sd['encoder_to_decoder.weight'] = _t('e2d', D, D)
        return {'model': sd}
    # 'kinetics': a Lightning checkpoint of this repo's own trainer
    sd['model.cls_token'] = _t('cls', 1, 1, D)
    sd['model.transformer_layers.layers.0.attentions.0.attn.in_proj_weight'] = _t('ipw', 3 * D, D)
    sd['model.transformer_layers.layers.0.attentions.0.attn.out_proj.bias'] = _t('opb', D)
    sd['model.transformer_layers.layers.0.attentions.1.attn.qkv.weight'] = _t('qkv', 3 * D, D)
    sd['model.norm.weight'] = _t('nw', D)
    sd['cls_head.cls_head.weight'] = _t('hw', 7, D)
    sd['cls_head.cls_head.bias'] = _t('hb', 7)
    return {'state_dict': sd}


def _kw(conv, att, copy='repeat', extend='temporal_avg'):
    return dict(conv_type=conv, attention_type=att, copy_strategy=copy, extend_strategy=extend, tube_size=2,
                num_time_transformer_layers=4)


CASES = [(f'{kind}_{conv}_{att}_{copy}_{extend}', kind, _kw(conv, att, copy, extend))
         for kind in ('vit', 'mae')
         for (conv, att, copy, extend) in [('Conv2d', 'divided_space_time', 'repeat', 'temporal_avg'),
                                           ('Conv2d', 'divided_space_time', 'set_zero', 'temporal_avg'),
                                           ('Conv2d', 'space_only', 'repeat', 'temporal_avg'),
                                           ('Conv2d', 'joint_space_time', 'repeat', 'temporal_avg'),
                                           ('Conv3d', 'fact_encoder', 'repeat', 'temporal_avg'),
                                           ('Conv3d', 'fact_encoder', 'set_zero', 'center_frame'),
                                           ('Conv3d', 'joint_space_time', 'repeat', 'center_frame')]]
CASES.append(('kinetics', 'kinetics', {}))


def summarize(sd):
    return {k: [list(v.shape), round(float(v.double().sum()), 6), round(float(v.double().abs().sum()), 6)]
            for k, v in sd.items()}
