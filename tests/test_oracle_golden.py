"""The CPU oracle (oracle/vt_oracle.py) against the golden vectors generated from the
running reference (tests/golden/make_golden.py).  Runs anywhere, no GPU, no reference."""
import json

import numpy as np
import pytest
import torch

from helpers import gold, relerr
from oracle import synth, vt_oracle as O
from oracle.synth import synth_tensor

SMALL = dict(heads=2, layers=2)


def _shapes_tsf(keys, T, D=768, P=196, tube=None):
    return keys


def _sd(shapes, seed, requires_grad=False):
    sd = synth.synth_state_dict(shapes, seed)
    if requires_grad:
        for v in sd.values():
            v.requires_grad_(True)
    return sd


def _small_shapes(model_kind, at, T):
    """state_dict shapes of the SMALL reference config, derived without the reference."""
    D, L, P = 128, 2, 16
    sh = {'cls_token': (1, 1, D), 'pos_embed': (1, P + 1, D), 'norm.weight': (D,), 'norm.bias': (D,)}
    if model_kind == 'tsf':
        sh['patch_embed.projection.weight'] = (D, 3, 16, 16)
        if at != 'space_only':
            sh['time_embed'] = (1, T, D)
    else:
        sh['patch_embed.projection.weight'] = (D, 3, 2, 16, 16)
        sh['time_embed'] = (1, T + 1 if at == 'fact_encoder' else T, D)
    sh['patch_embed.projection.bias'] = (D,)

    def block(pre, ops):
        ai = 0
        for op in ops:
            if op == 'ffn':
                f = pre + 'ffns.0.'
                sh.update({f + 'norm.weight': (D,), f + 'norm.bias': (D,), f + 'layers.0.0.weight': (4 * D, D),
                           f + 'layers.0.0.bias': (4 * D,), f + 'layers.1.weight': (D, 4 * D), f + 'layers.1.bias': (D,)})
                continue
            a = f'{pre}attentions.{ai}.'
            sh.update({a + 'norm.weight': (D,), a + 'norm.bias': (D,), a + 'attn.qkv.weight': (3 * D, D),
                       a + 'attn.qkv.bias': (3 * D,), a + 'attn.proj.weight': (D, D), a + 'attn.proj.bias': (D,)})
            if op == 'time_attn':
                sh.update({a + 'temporal_fc.weight': (D, D), a + 'temporal_fc.bias': (D,)})
            ai += 1
    ops = ['time_attn', 'space_attn', 'ffn'] if at == 'divided_space_time' else ['self_attn', 'ffn']
    if model_kind == 'vivit' and at == 'fact_encoder':
        for i in range(L):
            block(f'transformer_layers.0.layers.{i}.', ops)
        for i in range(4):
            block(f'transformer_layers.1.layers.{i}.', ops)
    else:
        for i in range(L):
            block(f'transformer_layers.layers.{i}.', ops)
    return sh


def _check_grads(sd, g, tol=2e-4):
    for k in g.files:
        if k.startswith('g:'):
            assert relerr(sd[k[2:]].grad, g[k]) < tol, k
        elif k.startswith('gh:'):
            name = k[3:]
            got = sd[name].grad
            assert relerr(got.flatten()[:256], g[k]) < tol * 20, k     # head only: relative to its own max
            assert abs(got.double().norm().item() - g['gn:' + name][0]) / g['gn:' + name][0] < tol, k


@pytest.mark.parametrize('at', ['divided_space_time', 'space_only', 'joint_space_time'])
def test_timesformer_small(at):
    g = gold(f'tsf_small_{at}.npz')
    sd = _sd(_small_shapes('tsf', at, 4), 3, True)
    x = synth.synth_clip(3, 4, 3, 64, 64, seed=2)
    torch.manual_seed(11)
    y = O.timesformer_forward(sd, x, 4, attention_type=at, training=True, **SMALL)
    assert relerr(y, g['out']) < 1e-5
    (y * (synth_tensor('loss_w', (128,), 0) * 10.0)).sum().backward()
    _check_grads(sd, g)
    with torch.no_grad():
        ye = O.timesformer_forward(sd, x, 4, attention_type=at, **SMALL)
        att = O.timesformer_forward(sd, x, 4, attention_type=at, return_attention=True, **SMALL)
    assert relerr(ye, g['out_eval']) < 1e-5
    assert relerr(att, g['attn']) < 1e-5


@pytest.mark.parametrize('at', ['fact_encoder', 'joint_space_time', 'divided_space_time'])
def test_vivit_small(at):
    g = gold(f'vivit_small_{at}.npz')
    sd = _sd(_small_shapes('vivit', at, 4), 4, True)
    x = synth.synth_clip(3, 8, 3, 64, 64, seed=5)
    torch.manual_seed(13)
    y = O.vivit_forward(sd, x, 8, attention_type=at, training=True, **SMALL)
    assert relerr(y, g['out']) < 1e-5
    (y * (synth_tensor('loss_w', (128,), 0) * 10.0)).sum().backward()
    _check_grads(sd, g)


def test_timesformer_b_cfg1():
    """BASELINE.json configs[0]: TimeSformer-B, 2 frames, batch 2, CPU forward."""
    from helpers import gold_keys
    shapes = {k: tuple(v) for k, v in gold_keys()['timesformer_b_t8'].items()}
    shapes['time_embed'] = (1, 2, 768)
    sd = _sd(shapes, 0)
    with torch.no_grad():
        y = O.timesformer_forward(sd, synth.synth_clip(2, 2, seed=0), 2)
    assert relerr(y, gold('tsf_b_cfg1.npz')['out']) < 1e-5


def test_vivit_b_eval():
    from helpers import gold_keys
    shapes = {k: tuple(v) for k, v in gold_keys()['vivit_b_t16'].items()}
    sd = _sd(shapes, 0)
    with torch.no_grad():
        y = O.vivit_forward(sd, synth.synth_clip(2, 16, seed=3), 16)
    assert relerr(y, gold('vivit_b_t16_eval.npz')['out']) < 1e-5


def test_maskfeat_head():
    g = gold('maskfeat_head.npz')
    mask = torch.from_numpy(g['mask'])
    markers = json.loads(str(g['markers']))
    shapes = {'decoder_pred.weight': (216, 768), 'decoder_pred.bias': (216,), 'mask_token': (1, 1, 96),
              'patch_embed.patch_model.weight': (96, 3, 3, 7, 7), 'patch_embed.patch_model.bias': (96,)}
    sd = _sd(shapes, 6, True)
    x = synth.synth_clip(2, 16, seed=8)
    tokens = torch.nn.functional.conv3d(x.transpose(1, 2), sd['patch_embed.patch_model.weight'],
                                        sd['patch_embed.patch_model.bias'], stride=(2, 4, 4), padding=(1, 3, 3))
    tokens = tokens.flatten(2).transpose(1, 2)
    blended = O.maskfeat_blend(tokens, mask, sd['mask_token'], 4)
    cs = g['blend_checksum']
    assert abs(blended.double().sum().item() - cs[0]) / abs(cs[1]) < 1e-6
    w = synth_tensor('standin.w', (768, 96), 0)
    v = blended.reshape(2, 8, 14, 4, 14, 4, 96).mean(dim=(3, 5)).reshape(2, 1568, 96) @ w.t()
    feat = torch.cat([v.mean(1, keepdim=True), v], dim=1)
    target = torch.rand(2, 16, 14, 14, 108, generator=torch.Generator().manual_seed(99), dtype=torch.float64)
    pred, loss = O.maskfeat_head(feat, sd['decoder_pred.weight'], sd['decoder_pred.bias'], target, mask, markers)
    assert abs(loss.item() - float(g['loss'])) / float(g['loss']) < 1e-6
    assert relerr(pred[:, :, :2, :2], g['pred_head']) < 1e-5
    loss.backward()
    assert relerr(sd['decoder_pred.bias'].grad, g['d_decoder_b']) < 1e-4
    assert relerr(sd['decoder_pred.weight'].grad[:8], g['d_decoder_w_head']) < 1e-4
    assert relerr(sd['mask_token'].grad, g['d_mask_token']) < 1e-4
