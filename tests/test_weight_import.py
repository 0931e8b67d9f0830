"""Pretrained-checkpoint importers (weight_init.py) against golden key maps produced by the reference's own
functions (tests/golden/make_golden_import.py), and -- in the dev container -- against the reference live."""
import json
import os

import pytest
import torch

from helpers import GOLD
from import_cases import CASES, summarize, synth_checkpoint
from oracle import ref_loader


class Recorder:
    def load_state_dict(self, sd, strict=True):
        self.sd = dict(sd)
        return [], []


def _run(W, name, kind, kw, tmp_path):
    path = str(tmp_path / (name + '.pth'))
    torch.save(synth_checkpoint(kind), path)
    rec = Recorder()
    if kind == 'vit':
        W.init_from_vit_pretrain_(rec, path, **kw)
    elif kind == 'mae':
        W.init_from_mae_pretrain_(rec, path, **kw)
    else:
        W.init_from_kinetics_pretrain_(rec, path)
    return summarize(rec.sd)


@pytest.mark.parametrize('name,kind,kw', CASES, ids=[c[0] for c in CASES])
def test_importer_matches_reference_golden(name, kind, kw, tmp_path):
    import weight_init as W
    gold = json.load(open(os.path.join(GOLD, 'weight_import.json')))[name]
    got = _run(W, name, kind, kw, tmp_path)
    assert sorted(got) == sorted(gold), (set(got) ^ set(gold))
    for k in gold:
        assert got[k][0] == gold[k][0], k
        assert abs(got[k][1] - gold[k][1]) < 1e-4 and abs(got[k][2] - gold[k][2]) < 1e-4, k


@pytest.mark.skipif(not ref_loader.available(), reason='needs /root/reference (dev container)')
@pytest.mark.parametrize('name,kind,kw', CASES[::3], ids=[c[0] for c in CASES[::3]])
def test_importer_matches_reference_live(name, kind, kw, tmp_path):
    import weight_init as W
    ref = _run(ref_loader.load().weight_init, name, kind, kw, tmp_path)
    assert _run(W, name, kind, kw, tmp_path) == ref


def test_vit_checkpoint_initialises_a_model(tmp_path):
    """End to end on the host: TimeSformer(pretrain_pth=...) with a ViT-format checkpoint loads every block weight
    (only the embeddings of the video model are missing, nothing unexpected)."""
    import video_transformer as V
    m0 = V.TimeSformer(num_frames=2, img_size=32, patch_size=16, embed_dims=16, num_heads=2, num_transformer_layers=2,
                       attention_type='space_only')
    sd = {}
    for k, v in m0.state_dict().items():
        k = k.replace('attentions.0.norm', 'norms.0').replace('ffns.0.norm', 'norms.1')
        k = k.replace('attn.qkv.', 'attn.in_proj_').replace('attn.proj', 'attn.out_proj')
        sd[k] = torch.randn_like(v)
    path = str(tmp_path / 'vit.pth')
    torch.save({'state_dict': sd}, path)
    m = V.TimeSformer(num_frames=2, img_size=32, patch_size=16, embed_dims=16, num_heads=2, num_transformer_layers=2,
                      attention_type='space_only', pretrain_pth=path)
    w = m.transformer_layers.layers[1].attentions[0].attn.qkv.weight
    assert torch.equal(w, sd['transformer_layers.layers.1.attentions.0.attn.in_proj_weight'])
    assert torch.equal(m.transformer_layers.layers[0].ffns[0].norm.bias, sd['transformer_layers.layers.0.norms.1.bias'])
