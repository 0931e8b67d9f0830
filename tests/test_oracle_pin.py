"""Pin the oracle restatement against the RUNNING reference (dev container only)."""
import pytest
import torch

from helpers import relerr
from oracle import ref_loader, synth, vt_oracle as O

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason='/root/reference not present')
SM = dict(img_size=64, patch_size=16, embed_dims=128, num_heads=2, num_transformer_layers=3)


@pytest.mark.parametrize('at', ['divided_space_time', 'space_only', 'joint_space_time'])
@pytest.mark.parametrize('train', [False, True])
def test_timesformer(at, train):
    R = ref_loader.load()
    m = R.video_transformer.TimeSformer(num_frames=4, attention_type=at, **SM)
    sd = synth.synth_state_dict(synth.shapes_of(m), 1)
    m.load_state_dict(sd)
    m.train(train)
    x = synth.synth_clip(2, 4, 3, 64, 64)
    torch.manual_seed(5)
    y = m(x)
    torch.manual_seed(5)
    yo = O.timesformer_forward(sd, x, 4, heads=2, layers=3, attention_type=at, training=train)
    assert relerr(yo, y) < 1e-5
    m.eval()
    a = m.get_last_selfattention(x)
    ao = O.timesformer_forward(sd, x, 4, heads=2, layers=3, attention_type=at, return_attention=True)
    assert relerr(ao, a) < 1e-5


@pytest.mark.parametrize('at', ['fact_encoder', 'joint_space_time', 'divided_space_time'])
def test_vivit(at):
    R = ref_loader.load()
    m = R.video_transformer.ViViT(num_frames=8, attention_type=at, **SM)
    sd = synth.synth_state_dict(synth.shapes_of(m), 2)
    m.load_state_dict(sd)
    m.train()
    x = synth.synth_clip(3, 8, 3, 64, 64)
    torch.manual_seed(7)
    y = m(x)
    torch.manual_seed(7)
    yo = O.vivit_forward(sd, x, 8, heads=2, layers=3, attention_type=at, training=True)
    assert relerr(yo, y) < 1e-5


@pytest.mark.parametrize('hw', [(96, 96), (64, 96), (96, 64), (32, 32)])
@pytest.mark.parametrize('at', ['divided_space_time', 'space_only'])
def test_timesformer_other_resolution(at, hw):
    """Clips of another resolution than img_size go through interpolate_pos_encoding (reference video_transformer.py:171-191,
    :209): bicubic resize of the positional table, with the reference's width / height quirk for non-square clips."""
    R = ref_loader.load()
    m = R.video_transformer.TimeSformer(num_frames=2, attention_type=at, **SM)
    sd = synth.synth_state_dict(synth.shapes_of(m), 6)
    m.load_state_dict(sd)
    m.eval()
    x = synth.synth_clip(2, 2, 3, hw[0], hw[1])
    y = m(x)
    yo = O.timesformer_forward(sd, x, 2, heads=2, layers=3, attention_type=at)
    assert relerr(yo, y) < 1e-5
    pe = m.interpolate_pos_encoding(torch.empty(1, 1 + (hw[0] // 16) * (hw[1] // 16), 128), hw[1], hw[0])
    po = O.interpolated_pos_embed(sd['pos_embed'], (hw[0] // 16) * (hw[1] // 16), hw[1], hw[0], 16)
    assert torch.equal(pe.detach(), po)


def test_droppath_rng_stream():
    """one training forward of TimeSformer-B (B=2,T=2) makes 33 torch.rand calls
    (layer 0 draws nothing); the oracle must consume the generator identically."""
    R = ref_loader.load()
    m = R.video_transformer.TimeSformer(num_frames=2, **SM)
    sd = synth.synth_state_dict(synth.shapes_of(m), 1)
    m.load_state_dict(sd)
    m.train()
    x = synth.synth_clip(2, 2, 3, 64, 64)
    torch.manual_seed(3)
    m(x)
    after_ref = torch.rand(4)
    torch.manual_seed(3)
    O.timesformer_forward(sd, x, 2, heads=2, layers=3, training=True)
    assert torch.equal(after_ref, torch.rand(4))
