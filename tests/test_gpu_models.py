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

from helpers import AUTOCAST_FACTOR, TOL_BF16, TOL_BF16_GRAD, TOL_F32, WIDEN_CAP, cal_entry, check, compare_grads, gold, relerr, report
from oracle import synth, vt_oracle as O
from oracle.synth import synth_tensor

from model_common import DEV, PRECS, SMALL, _build, _reset_precision, _train_step  # noqa: F401

pytestmark = pytest.mark.gpu


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
    check(f'tsf_small {at} {prec} train out', y.cpu(), g['out'], tol, cal=(f'tsf_small {at} train' if prec == 'bf16' else None))
    compare_grads(f'tsf_small {at} {prec}', grads, g, gtol, exact_elements=(prec == 'fp32'), cal=f'tsf_small {at} train')
    m.eval()
    with torch.no_grad():
        check(f'tsf_small {at} {prec} eval out', m(x.to(DEV)).cpu(), g['out_eval'], tol, cal=(f'tsf_small {at} eval' if prec == 'bf16' else None))
        att = m.get_last_selfattention(x.to(DEV))
    assert tuple(att.shape) == g['attn'].shape
    check(f'tsf_small {at} {prec} last attention', att.cpu(), g['attn'], tol, cal=(f'tsf_small {at} last attention' if prec == 'bf16' else None))
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
    check(f'tsf other resolution {hw} {prec} out', y.detach().cpu(), yo.detach(), tol, cal=(f'tsf other resolution {hw}' if prec == 'bf16' else None))
    for k in ('pos_embed', 'time_embed', 'cls_token', 'patch_embed.projection.weight'):
        got, ref = dict(m.named_parameters())[k].grad.cpu(), ps[k].grad
        e = (got.double() - ref.double()).norm().item() / ref.double().norm().item()
        bar = gtol if prec == 'fp32' else max(gtol, min(AUTOCAST_FACTOR * cal_entry(f'tsf other resolution {hw}')['grad'][k], WIDEN_CAP * gtol))
        report(f'{"ok  " if e <= bar else "FAIL"} tsf other resolution {hw} {prec} grad {k}: l2-rel={e:.3e} (tol {bar:g})')
        assert e <= bar, (k, hw, prec, e)


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
    check(f'vivit_small {at} {prec} train out', y.cpu(), g['out'], tol, cal=(f'vivit_small {at} train' if prec == 'bf16' else None))
    compare_grads(f'vivit_small {at} {prec}', grads, g, gtol, exact_elements=(prec == 'fp32'), cal=f'vivit_small {at} train')


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
def test_vivit_fact_encoder_width_above_1024(prec, tol):
    """ViViT fact_encoder with embed_dims 1152: vtx_fact_glue_fwd takes rows up to 1024 wide, wider models run the glue between the
    two encoders (reference video_transformer.py:515-523) as device-side ATen ops instead of failing."""
    import vtx
    import video_transformer as V
    vtx.set_precision(prec)
    m, sd = _build(V.ViViT, 8, num_frames=4, img_size=32, patch_size=16, embed_dims=1152, num_heads=18, num_transformer_layers=1)
    x = synth.synth_clip(2, 4, 3, 32, 32, seed=9)
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = O.vivit_forward(ps, x, 4, heads=18, layers=1)
    (yo * synth_tensor('loss_w', (1152,), 0)).sum().backward()
    m.eval()
    m.zero_grad()
    y = m(x.to(DEV))
    (y * synth_tensor('loss_w', (1152,), 0).to(DEV)).sum().backward()
    check(f'vivit fact_encoder D=1152 {prec} out', y.detach().cpu(), yo.detach(), tol)
    gtol = TOL_F32 if prec == 'fp32' else TOL_BF16_GRAD
    for k in ('time_embed', 'cls_token', 'pos_embed'):
        got, ref = dict(m.named_parameters())[k].grad.cpu(), ps[k].grad
        e = (got.double() - ref.double()).norm().item() / ref.double().norm().item()
        assert e <= gtol, (k, prec, e)


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
