"""Shared pieces of the model-level GPU parity tests (tests/test_gpu_00_baseline_configs.py, tests/test_gpu_models.py)."""
import pytest
import torch

from helpers import TOL_BF16, TOL_BF16_GRAD, TOL_F32
from oracle import synth
from oracle.synth import synth_tensor

DEV = 'cuda:0'
SMALL = dict(img_size=64, patch_size=16, embed_dims=128, num_heads=2, num_transformer_layers=2)
PRECS = [('fp32', TOL_F32, TOL_F32), ('bf16', TOL_BF16, TOL_BF16_GRAD)]


@pytest.fixture(autouse=True)
def _reset_precision():
    import vtx
    yield
    vtx.set_precision('auto')


def _build(cls, seed, **kw):
    m = cls(**kw)
    sd = synth.synth_state_dict(synth.shapes_of(m), seed)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV), sd


def _train_step(m, x, seed, d):
    m.train()
    m.zero_grad()
    torch.manual_seed(seed)
    y = m(x.to(DEV))
    w = (synth_tensor('loss_w', (d,), 0) * 10.0).to(DEV)
    (y * w).sum().backward()
    torch.cuda.synchronize()
    return y, {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
