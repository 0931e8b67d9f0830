"""Model-level parity of the HIP path (through the drop-in nn.Module API and the C ABI)
against (a) the committed golden vectors generated from the reference and (b) the CPU oracle.

Bars (tests/helpers.py): fp32 path 1e-3 relative (north_star); bf16 path 1.5e-2 on outputs and 2e-2 relative L2
on parameter gradients versus the same fp32 reference, calibrated against the reference's own bf16-autocast run
(tsf_b_t8_autocast.npz: 9.6e-3 / 1.76e-2 worst), which the bf16 test of that configuration is also compared with.
All weights come from oracle/synth.py: temporal_fc is NOT zero, so the temporal kernels matter.
"""
import json

import numpy as np
import pytest
import torch

from helpers import TOL_BF16, TOL_BF16_GRAD, TOL_F32, check, compare_grads, gold, relerr, report
from oracle import synth, vt_oracle as O
from oracle.synth import synth_tensor

pytestmark = pytest.mark.gpu
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


@pytest.mark.parametrize('prec,tol,gtol', PRECS)
@pytest.mark.parametrize('at', ['divided_space_time', 'space_only', 'joint_space_time'])
def test_timesformer_small_vs_golden(at, prec, tol, gtol):
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    g = gold(f'tsf_small_{at}.npz')
    m, _ = _build(V.TimeSformer, 3, num_frames=4, attention_type=at, **SMALL)
    x = synth.synth_clip(3, 4, 3, 64, 64, seed=2)
    y, grads = _train_step(m, x, 11, 128)
    assert y.dtype == torch.float32 and y.shape == (3, 128)
    check(f'tsf_small {at} {prec} train out', y.cpu(), g['out'], tol)
    compare_grads(f'tsf_small {at} {prec}', grads, g, gtol, exact_elements=(prec == 'fp32'))
    m.eval()
    with torch.no_grad():
        check(f'tsf_small {at} {prec} eval out', m(x.to(DEV)).cpu(), g['out_eval'], tol)
        att = m.get_last_selfattention(x.to(DEV))
    assert tuple(att.shape) == g['attn'].shape
    check(f'tsf_small {at} {prec} last attention', att.cpu(), g['attn'], tol)
    assert abs(att.sum(-1).cpu() - 1).max().item() < 1e-3


@pytest.mark.parametrize('prec,tol,gtol', PRECS)
@pytest.mark.parametrize('hw', [(96, 96), (64, 96), (32, 32)])
def test_timesformer_other_resolution_vs_oracle(hw, prec, tol, gtol):
    """A clip of another resolution than img_size: the positional table is resized (reference video_transformer.py:171-191,
    :209, non-square quirk included) and its gradient flows back through the resize to pos_embed.  Oracle = the
    reference-pinned restatement (tests/test_oracle_pin.py::test_timesformer_other_resolution)."""
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    m, sd = _build(V.TimeSformer, 6, num_frames=2, **SMALL)
    x = synth.synth_clip(2, 2, 3, hw[0], hw[1], seed=3)
    m.eval()
    m.zero_grad()
    y = m(x.to(DEV))
    w = (synth_tensor('loss_w', (128,), 0) * 10.0)
    (y * w.to(DEV)).sum().backward()
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = O.timesformer_forward(ps, x, 2, heads=2, layers=2)
    (yo * w).sum().backward()
    check(f'tsf other resolution {hw} {prec} out', y.detach().cpu(), yo.detach(), tol)
    for k in ('pos_embed', 'time_embed', 'cls_token', 'patch_embed.projection.weight'):
        got, ref = dict(m.named_parameters())[k].grad.cpu(), ps[k].grad
        e = (got.double() - ref.double()).norm().item() / ref.double().norm().item()
        assert e <= gtol, (k, hw, prec, e)


@pytest.mark.parametrize('prec,tol,gtol', PRECS)
@pytest.mark.parametrize('at', ['fact_encoder', 'joint_space_time', 'divided_space_time'])
def test_vivit_small_vs_golden(at, prec, tol, gtol):
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    g = gold(f'vivit_small_{at}.npz')
    m, _ = _build(V.ViViT, 4, num_frames=8, attention_type=at, **SMALL)
    x = synth.synth_clip(3, 8, 3, 64, 64, seed=5)
    y, grads = _train_step(m, x, 13, 128)
    check(f'vivit_small {at} {prec} train out', y.cpu(), g['out'], tol)
    compare_grads(f'vivit_small {at} {prec}', grads, g, gtol, exact_elements=(prec == 'fp32'))


@pytest.mark.parametrize('prec,tol', [('fp32', TOL_F32), ('bf16', TOL_BF16)])
def test_timesformer_b_cfg1_forward(prec, tol):
    """BASELINE.json configs[0]: TimeSformer-B divided_space_time, 2 frames, batch 2, forward."""
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    m, _ = _build(V.TimeSformer, 0, num_frames=2)
    m.eval()
    with torch.no_grad():
        y = m(synth.synth_clip(2, 2, seed=0).to(DEV))
    check(f'TimeSformer-B cfg1 {prec}', y.cpu(), gold('tsf_b_cfg1.npz')['out'], tol)


@pytest.mark.parametrize('prec,tol,gtol', PRECS)
def test_timesformer_b_t8_train_vs_golden(prec, tol, gtol):
    """BASELINE.json configs[1] shape (TimeSformer-B, 8x224^2), train mode with DropPath, fwd+bwd."""
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    g = gold('tsf_b_t8_autocast.npz')                 # the fp32 run of the reference ('out', gradients) + its autocast run
    m, _ = _build(V.TimeSformer, 0, num_frames=8)
    y, grads = _train_step(m, synth.synth_clip(1, 8, seed=1), 7, 768)
    check(f'TimeSformer-B T=8 train {prec} out', y.cpu(), g['out'], tol)
    compare_grads(f'TimeSformer-B T=8 train {prec}', grads, g, gtol, exact_elements=(prec == 'fp32'))
    ge = gold('tsf_b_t8_eval.npz')
    m.eval()
    with torch.no_grad():
        x = synth.synth_clip(1, 8, seed=1).to(DEV)
        check(f'TimeSformer-B T=8 eval {prec} out', m(x).cpu(), ge['out'], tol)
        att = m.get_last_selfattention(x)
    assert list(att.shape) == list(ge['attn_shape'])                 # [8, 12, 197, 197]
    check(f'TimeSformer-B T=8 {prec} attention', att[:2, :, :8, :8].cpu(), ge['attn_head'], tol)


@pytest.mark.parametrize('prec,tol,gtol', PRECS)
def test_timesformer_b_t16_train_vs_golden(prec, tol, gtol):
    """The north_star's second clip shape: TimeSformer-B on 16x3x224x224, train mode with DropPath,
    fwd+bwd against the reference's own run (tests/golden/make_golden_r2.py).  Temporal attention
    here packs two 16-token sequences per 32-row MFMA tile."""
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    g = gold('tsf_b_t16_train.npz')
    m, _ = _build(V.TimeSformer, 0, num_frames=16)
    y, grads = _train_step(m, synth.synth_clip(1, 16, seed=21), 9, 768)
    check(f'TimeSformer-B T=16 train {prec} out', y.cpu(), g['out'], tol)
    compare_grads(f'TimeSformer-B T=16 train {prec}', grads, g, gtol, exact_elements=(prec == 'fp32'))


def test_timesformer_b_t8_bf16_vs_reference_autocast():
    """The benchmarked configuration and precision (BASELINE cfg 2, bf16) against the reference's fp32 run, with the
    reference's OWN bf16-autocast run of the same step as the yardstick: every parameter gradient within 2x of the
    deviation the reference's AMP shows for that tensor (floor 1e-2), the median within 1.25x, outputs within 1.3x."""
    import vtx
    import video_transformer as V
    vtx.set_precision('bf16')
    g = gold('tsf_b_t8_autocast.npz')
    m, _ = _build(V.TimeSformer, 0, num_frames=8)
    y, grads = _train_step(m, synth.synth_clip(1, 8, seed=1), 7, 768)
    e = check('TimeSformer-B T=8 train bf16 out (r2 golden)', y.cpu(), g['out'], TOL_BF16)
    ref_dev = relerr(g['out_autocast'], g['out'])
    report(f'     reference autocast output deviation {ref_dev:.3e}, this path {e:.3e}')
    assert e <= 1.3 * ref_dev, f'bf16 outputs deviate {e:.3e}, more than 1.3x the reference autocast run ({ref_dev:.3e})'
    compare_grads('TimeSformer-B T=8 train bf16 vs reference autocast', grads, g, TOL_BF16_GRAD, autocast_cal=True)


@pytest.mark.parametrize('prec,tol,gtol', PRECS)
def test_vivit_b_t16_train_vs_golden(prec, tol, gtol):
    """BASELINE.json configs[2] at full size, TRAIN mode fwd+bwd: ViViT-B fact_encoder, Conv3d tubelets, 16x224^2,
    batch 2 (so that the reference's `x[:b, 0]` cls quirk between the two encoders matters), all 231 gradients."""
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    g = gold('vivit_b_t16_train.npz')
    m, _ = _build(V.ViViT, 0, num_frames=16)
    y, grads = _train_step(m, synth.synth_clip(2, 16, seed=3), 17, 768)
    check(f'ViViT-B T=16 train {prec} out', y.cpu(), g['out'], tol)
    compare_grads(f'ViViT-B T=16 train {prec}', grads, g, gtol, exact_elements=(prec == 'fp32'))


@pytest.mark.parametrize('prec,tol,gtol', PRECS)
def test_timesformer_l_t96_train_vs_golden(prec, tol, gtol):
    """BASELINE.json configs[4] geometry: TimeSformer-L (D 1024, 16 heads, hidden 4096) on 96x224^2 clips -- 18 817
    tokens per clip, temporal attention over 96 frames, 96-frame cls mean -- at depth 2, train mode fwd+bwd."""
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    g = gold('tsf_l_t96_d2_train.npz')
    m, _ = _build(V.TimeSformer, 0, num_frames=96, embed_dims=1024, num_heads=16, num_transformer_layers=2)
    y, grads = _train_step(m, synth.synth_clip(1, 96, seed=5), 19, 1024)
    check(f'TimeSformer-L T=96 depth 2 train {prec} out', y.cpu(), g['out'], tol)
    compare_grads(f'TimeSformer-L T=96 depth 2 train {prec}', grads, g, gtol, exact_elements=(prec == 'fp32'))


@pytest.mark.parametrize('prec,tol', [('fp32', TOL_F32), ('bf16', TOL_BF16)])
def test_vivit_b_forward(prec, tol):
    """BASELINE.json configs[2] shape: ViViT-B fact_encoder, Conv3d tubelets, 16 frames."""
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    m, _ = _build(V.ViViT, 0, num_frames=16)
    m.eval()
    with torch.no_grad():
        y = m(synth.synth_clip(2, 16, seed=3).to(DEV))
    check(f'ViViT-B fact_encoder {prec}', y.cpu(), gold('vivit_b_t16_eval.npz')['out'], tol)


def test_block_recompute_gives_identical_gradients():
    """vtx.set_recompute(True): every transformer block is re-run in backward instead of keeping its activations
    (long-clip sizing, BASELINE cfg 5) -- same DropPath draws, bit-identical outputs and gradients."""
    import vtx
    import video_transformer as V
    vtx.set_precision('bf16')
    res = []
    try:
        for rc in (False, True):
            vtx.set_recompute(rc)
            m, _ = _build(V.TimeSformer, 3, num_frames=4, **SMALL)
            y, grads = _train_step(m, synth.synth_clip(3, 4, 3, 64, 64, seed=2), 11, 128)
            res.append((y.detach().clone(), {k: v.clone() for k, v in grads.items()}))
    finally:
        vtx.set_recompute(False)
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize('prec,tol,gtol', [('fp32', 2e-5, 2e-4), ('bf16', TOL_BF16, TOL_BF16_GRAD)])
def test_merged_temporal_fc_matches_the_two_linear_layers(prec, tol, gtol):
    """attn.proj + DropPath + temporal_fc as one GEMM with the product weight (vtx.functions.TimeAttnFn, the default)
    against the two GEMMs of the reference (transformer.py:268-275), same DropPath draws, drop rate high enough that
    dropped and kept sequences both occur: fp32 agrees to re-association error, bf16 within the parity bars (and the
    goldens above bound each path against the reference separately)."""
    import vtx
    import transformer as T_
    import video_transformer as V
    from vtx import functions
    vtx.set_precision(prec)
    res = []
    try:
        for merged in (True, False):
            functions.set_merge_temporal_fc(merged)
            m, _ = _build(V.TimeSformer, 3, num_frames=4, **SMALL)
            for mod in m.modules():
                if isinstance(mod, T_.DropPath):
                    mod.dropout_p = 0.4
            y, grads = _train_step(m, synth.synth_clip(3, 4, 3, 64, 64, seed=2), 11, 128)
            res.append((y.detach().float().cpu(), {k: v.float().cpu() for k, v in grads.items()}))
    finally:
        functions.set_merge_temporal_fc(True)
    check(f'merged temporal_fc {prec} out', res[0][0], res[1][0], tol)
    assert set(res[0][1]) == set(res[1][1])
    for k in res[1][1]:
        e = relerr(res[0][1][k], res[1][1][k])
        assert e <= gtol, f'{prec} {k}: {e:.3e} > {gtol}'


def test_direct_parameter_gradients_match_autograd_accumulation():
    """vtx.dp.GradBuckets(direct=True): the weight-gradient reductions, bias column sums and LayerNorm backward
    accumulate straight into the bucket views and fire the bucket hooks themselves (vtx.functions.set_direct_grads).
    Same gradients bit for bit as autograd's accumulation, every bucket hook fires exactly once per backward, a
    second backward into the same buffers accumulates (2x), and the recompute path works the same way."""
    import vtx
    import video_transformer as V
    from vtx import dp, functions
    for prec in ('bf16', 'fp32'):
        vtx.set_precision(prec)
        m, _ = _build(V.TimeSformer, 3, num_frames=4, **SMALL)
        x = synth.synth_clip(3, 4, 3, 64, 64, seed=2)
        _, ref = _train_step(m, x, 11, 128)
        ref = {k: v.clone() for k, v in ref.items()}
        for recompute in (False, True):
            vtx.set_recompute(recompute)
            buckets = dp.GradBuckets(list(m.parameters()), bucket_bytes=64 << 10, direct=True)
            try:
                assert functions.direct_grads_enabled()
                buckets.zero()
                m.train()
                torch.manual_seed(11)
                w = (synth_tensor('loss_w', (128,), 0) * 10.0).to(DEV)
                (m(x.to(DEV)) * w).sum().backward()
                torch.cuda.synchronize()
                assert all(b['pending'] == 0 for b in buckets.buckets), [b['pending'] for b in buckets.buckets]
                for k, p in m.named_parameters():
                    assert torch.equal(p.grad, ref[k]), f'{prec} recompute={recompute}: {k}'
            finally:
                buckets.remove()
                vtx.set_recompute(False)
            assert not functions.direct_grads_enabled()
        # accumulation over two backward passes into caller-owned buffers (no buckets, no hooks)
        functions.set_direct_grads(True)
        try:
            for p in m.parameters():
                p.grad = torch.zeros_like(p)
            for _ in range(2):
                torch.manual_seed(11)
                (m(x.to(DEV)) * w).sum().backward()
            for k, p in m.named_parameters():
                check(f'{prec} two backward passes {k}', p.grad.cpu(), 2 * ref[k].cpu(), 1e-6)
        finally:
            functions.set_direct_grads(False)


def _ffn_drop_pattern(seed, B, P, T, layers, rate=0.1):
    """The FFN DropPath masks a training forward draws after torch.manual_seed(seed): per layer with p > 0 the reference draws
    (B*P) temporal, (B*T) spatial and B FFN values from the CPU generator, in that order (transformer.py:34-42, 268-275, 371-377, 543)."""
    torch.manual_seed(seed)
    out = []
    for p in np.linspace(0, rate, layers):
        if p == 0:
            out.append([False] * B)
            continue
        torch.rand(B * P, 1, 1)
        torch.rand(B * T, 1, 1)
        u = torch.rand(B, 1, 1).reshape(B)
        out.append([bool(np.floor(np.float32(1 - p) + np.float32(v)) == 0) for v in u.tolist()])
    return out


@pytest.mark.parametrize('prec,tol,gtol', PRECS)
def test_bench_stack_vs_oracle_with_partial_ffn_drop(prec, tol, gtol):
    """The stack bench.py times -- GradBuckets(direct=True) (kernels accumulate into the bucket views), DropPath-aware
    compaction of the FFN with SOME BUT NOT ALL clips of a layer dropped (table row maps: >= 256 rows per clip), the merged
    attn.proj o temporal_fc projection -- against the reference restatement (oracle.timesformer_forward with the same CPU
    draws) at 8 clips: outputs and every parameter gradient.  The committed goldens are B <= 3 and never drop part of a batch."""
    import vtx
    import video_transformer as V
    from vtx import dp, functions
    B, T, L = 8, 16, 4
    cfg = dict(img_size=64, patch_size=16, embed_dims=128, num_heads=2, num_transformer_layers=L)
    P = (64 // 16) ** 2
    assert 1 + P * T >= 256, 'the compact path needs >= 256 rows per clip'
    seed = next(s for s in range(200) if any(0 < sum(l) < B for l in _ffn_drop_pattern(s, B, P, T, L)))
    pattern = _ffn_drop_pattern(seed, B, P, T, L)
    report(f'bench-stack test: seed {seed}, dropped clips per layer {[sum(l) for l in pattern]}')
    vtx.set_precision(prec)
    assert functions._compact and functions._merge_tfc, 'defaults: compaction and merged projection on'
    m, sd = _build(V.TimeSformer, 5, num_frames=T, **cfg)
    x = synth.synth_clip(B, T, 3, 64, 64, seed=4)
    w = synth_tensor('loss_w', (128,), 0) * 10.0
    buckets = dp.GradBuckets(list(m.parameters()), bucket_bytes=256 << 10, direct=True)
    try:
        buckets.zero()
        m.train()
        torch.manual_seed(seed)
        y = m(x.to(DEV))
        (y * w.to(DEV)).sum().backward()
        buckets.finish()
        torch.cuda.synchronize()
        assert all(b['pending'] == 0 for b in buckets.buckets)
        grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    finally:
        buckets.remove()
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.manual_seed(seed)
    yo = O.timesformer_forward(ps, x, T, heads=2, layers=L, training=True)
    (yo * w).sum().backward()
    check(f'bench stack {prec} out', y.detach().cpu(), yo.detach(), tol)
    worst = 0.0
    for k, g in grads.items():
        ref = ps[k].grad
        e = (g.double() - ref.double()).norm().item() / max(ref.double().norm().item(), 1e-30)
        worst = max(worst, e)
        assert e <= gtol, f'{prec} {k}: {e:.3e} > {gtol}'
    report(f'bench stack {prec}: worst parameter gradient l2-rel {worst:.3e}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,rows_per,D,Hd', [(5, 37, 128, 256), (6, 1569, 768, 3072)])
def test_ffn_skips_dropped_clips(dtype, B, rows_per, D, Hd):
    """DropPath at the FFN drops whole clips (reference transformer.py:34-42,543).  FFNFn runs the block on the kept clips
    only -- table row maps in LayerNorm, the fc2 epilogue's residual / result rows and the gradient gather (the large case
    goes through the persistent GEMM's residual-block flow) -- and must give what computing every clip and multiplying by
    zero gives: the same stream values bit for bit, the same input gradient, parameter gradients up to the fp32 summation
    order of the weight-gradient reduction."""
    from vtx import functions as F_
    g = torch.Generator().manual_seed(11)
    r = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc)              # noqa: E731
    x0 = r(B, rows_per, D).to(dtype)
    dout = r(B, rows_per, D).to(dtype)
    params0 = [1 + 0.1 * r(D), 0.1 * r(D), r(Hd, D, sc=D ** -0.5), 0.1 * r(Hd), r(D, Hd, sc=Hd ** -0.5), 0.1 * r(D)]
    c = float(np.float32(1.0) / np.float32(0.9))
    patterns = [[1, 3], [0], [B - 1], list(range(B)), list(range(1, B)), []]
    for dropped in patterns:
        host = torch.tensor([0.0 if i in dropped else c for i in range(B)], dtype=torch.float32)
        res = []
        for compact in (False, True, True):               # the compact path twice: it must be deterministic
            F_.set_compact_droppath(compact)
            try:
                x = x0.to(DEV).requires_grad_(True)
                ps = [p.clone().to(DEV).requires_grad_(True) for p in params0]
                sv = host.to(DEV)
                sv._vtx_host = host
                y = F_.FFNFn.apply(x, *ps, sv, 1e-5)
                y.backward(dout.to(DEV))
                torch.cuda.synchronize()
                res.append((y.detach().cpu(), x.grad.cpu(), [p.grad.cpu() for p in ps]))
            finally:
                F_.set_compact_droppath(True)
        (y0, dx0, g0), (y1, dx1, g1), (y2, dx2, g2) = res
        tag = f'ffn compaction {dtype} B={B} rows={rows_per} dropped={dropped}'
        assert torch.equal(y1, y2) and torch.equal(dx1, dx2) and all(torch.equal(a, b) for a, b in zip(g1, g2)), tag + ': not deterministic'
        assert torch.equal(y0, y1), tag + ': outputs differ'
        assert torch.equal(dx0[dropped], dx1[dropped]) and torch.equal(y1[dropped], x0[dropped]), tag
        if len(dropped) < B:
            # bf16: every intermediate (LayerNorm output and statistics, both GEMMs, dh, dxn) is bit-identical between the
            # two paths (tools/micro/ffn_compact_debug.py); the LayerNorm backward kernel handles two rows per trip in two
            # inlined copies of its row code whose fp32 contraction differs in the last bit, and which copy a row gets depends
            # on the row count -- a handful of elements per tensor land on the other side of a bf16 rounding boundary
            check(tag + ' dx', dx1.float(), dx0.float(), 1e-6 if dtype == torch.float32 else 4e-3)
            if dtype == torch.bfloat16:                  # (fp32: that last bit is visible in a few per cent of the elements)
                assert (dx1 != dx0).sum().item() <= max(4, dx0.numel() // 100000), tag + ': too many elements of dx differ'
        for name, a, b in zip(('ln_w', 'ln_b', 'w1', 'b1', 'w2', 'b2'), g1, g0):
            if len(dropped) == B:
                assert torch.count_nonzero(a) == 0 and torch.count_nonzero(b) == 0, tag
            else:
                check(f'{tag} d{name}', a, b, 2e-5)


def test_batch_and_length_properties():
    """Size-independent properties at full width: clips are independent (a clip's output does not
    depend on its batch neighbours) and eval forward is deterministic."""
    import vtx
    import video_transformer as V
    vtx.set_precision('bf16')
    m, _ = _build(V.TimeSformer, 0, num_frames=8)
    m.eval()
    x = synth.synth_clip(3, 8, seed=4).to(DEV)
    with torch.no_grad():
        y3 = m(x)
        y1 = m(x[1:2])
        y3b = m(x)
    assert torch.equal(y3, y3b), 'eval forward must be deterministic'
    check('clip independence (bf16, B=3 vs B=1)', y1.cpu(), y3[1:2].cpu(), 1e-6)


def test_uint8_clip_input_equals_float_input():
    """A decoded uint8 [B,T,H,W,3] clip fed to the model (vtx.set_input_normalization) gives exactly the
    output of the reference-style float [B,T,C,H,W] input produced by ToTensor + Normalize."""
    import vtx
    import video_transformer as V
    mean, std = [0.45, 0.45, 0.45], [0.225, 0.225, 0.225]
    g = torch.Generator().manual_seed(8)
    u8 = torch.randint(0, 256, (2, 4, 64, 64, 3), generator=g, dtype=torch.uint8)
    xf = u8.permute(0, 1, 4, 2, 3).float().div(255)
    xf = xf.sub(torch.tensor(mean).view(1, 1, 3, 1, 1)).div(torch.tensor(std).view(1, 1, 3, 1, 1))
    for cls, kw in ((V.TimeSformer, dict(num_frames=4)), (V.ViViT, dict(num_frames=4))):
        for prec in ('fp32', 'bf16'):
            vtx.set_precision(prec)
            m, _ = _build(cls, 5, **kw, **SMALL)
            m.eval()
            vtx.set_input_normalization(mean, std)
            try:
                with torch.no_grad():
                    y8 = m(u8.to(DEV))
                    yf = m(xf.to(DEV))
            finally:
                vtx.set_input_normalization(None, None)
            assert torch.equal(y8, yf), f'{cls.__name__} {prec}: uint8 and float inputs disagree'


def test_autocast_selects_bf16_path():
    import vtx
    import video_transformer as V
    m, _ = _build(V.TimeSformer, 3, num_frames=4, **SMALL)
    m.eval()
    x = synth.synth_clip(2, 4, 3, 64, 64, seed=2).to(DEV)
    with torch.no_grad():
        y32 = m(x)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y16 = m(x)
        vtx.set_precision('bf16')
        yb = m(x)
    assert torch.equal(y16, yb) and not torch.equal(y16, y32)


@pytest.mark.parametrize('prec,tol', [('fp32', TOL_F32), ('bf16', TOL_BF16)])
def test_maskfeat_head_vs_golden(prec, tol):
    """MaskFeat head (blend + decoder + centre-frame masked MSE) against the reference's own
    MaskFeat.forward run on a stand-in backbone (tests/golden/make_golden.py)."""
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    g = gold('maskfeat_head.npz')
    mask = torch.from_numpy(g['mask'])
    markers = json.loads(str(g['markers']))

    class StandIn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer('w', synth_tensor('standin.w', (768, 96), 0))

        def forward(self, t):
            t = t.float()
            b = t.shape[0]
            v = t.reshape(b, 8, 14, 4, 14, 4, 96).mean(dim=(3, 5)).reshape(b, 1568, 96) @ self.w.t()
            return torch.cat([v.mean(1, keepdim=True), v], dim=1)

    m = V.MaskFeat(pool_q_stride_size=[[1, 1, 2, 2], [3, 1, 2, 2]], feature_dim=2 * 2 * 2 * 3 * 9, backbone=StandIn())
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if 'mvit' not in k}, seed=6)
    m.load_state_dict(sd, strict=False)
    m.to(DEV).train()
    x = synth.synth_clip(2, 16, seed=8).to(DEV)
    target = torch.rand(2, 16, 14, 14, 108, generator=torch.Generator().manual_seed(99), dtype=torch.float64)
    pred, loss = m(x, target.to(DEV), mask.to(DEV), markers)
    loss.backward()
    assert pred.shape == (2, 16, 14, 14, 108) and loss.dtype == torch.float64
    rel = abs(loss.item() - float(g['loss'])) / float(g['loss'])
    report(f'maskfeat {prec}: loss {loss.item():.9f} vs reference {float(g["loss"]):.9f} (rel {rel:.2e})')
    assert rel < tol
    check(f'maskfeat {prec} pred', pred[:, :, :2, :2].cpu(), g['pred_head'], tol)
    check(f'maskfeat {prec} d decoder bias', m.decoder_pred.bias.grad.cpu(), g['d_decoder_b'], 2 * tol)
    check(f'maskfeat {prec} d decoder weight', m.decoder_pred.weight.grad[:8].cpu(), g['d_decoder_w_head'], 2 * tol)
    check(f'maskfeat {prec} d mask_token', m.mask_token.grad.cpu(), g['d_mask_token'], 4 * tol)


def test_hip_vs_oracle_fresh_seed():
    """Independent of the goldens: a new seed, oracle run on this host."""
    import vtx
    import video_transformer as V
    vtx.set_precision('fp32')
    m, sd = _build(V.TimeSformer, 21, num_frames=8, img_size=96, patch_size=16, embed_dims=192, num_heads=3,
                   num_transformer_layers=3)
    x = synth.synth_clip(2, 8, 3, 96, 96, seed=22)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.manual_seed(5)
    yo = O.timesformer_forward(sdo, x, 8, heads=3, layers=3, training=True)
    (yo * synth_tensor('loss_w', (192,), 0) * 10.0).sum().backward()
    y, grads = _train_step(m, x, 5, 192)
    check('fresh-seed fp32 out vs oracle', y.cpu(), yo.detach(), TOL_F32)
    worst = max(relerr(grads[k].cpu(), sdo[k].grad) for k in grads)
    report(f'fresh-seed fp32: {len(grads)} grads, worst rel {worst:.3e}')
    assert worst < TOL_F32
