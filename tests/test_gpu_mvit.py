"""MViT-B backbone (mvit.py + csrc/mvit.hip) against the CPU restatement oracle/mvit_oracle.py: every new operator
on small grids, a reduced backbone with all gradients, and MaskFeat end to end exactly as the reference's trainer
constructs it (model_trainer.py:53-54).  PARITY UNPINNED BY THE REFERENCE: pytorchvideo, which holds this arithmetic
in the reference, is on no disk of this build -- the oracle is a restatement from the paper / the 0.1.3 sources."""
import numpy as np
import pytest
import torch

from helpers import TOL_BF16, TOL_F32, check
from oracle import mvit_oracle as MO
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def q(t, dtype):
    return t.to(dtype).double()


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('hd,heads,stride,thw', [(96, 2, (1, 2, 2), (2, 8, 8)), (96, 1, (1, 8, 8), (2, 16, 16)),
                                                 (64, 3, (1, 4, 4), (3, 8, 12)), (96, 4, (1, 1, 1), (2, 5, 7))])
def test_pool_conv_ln(dtype, hd, heads, stride, thw):
    from vtx import functions as F_
    B, C = 2, heads * hd
    T, H, W = thw
    x = rnd(B, 1 + T * H * W, C, seed=1)
    conv = torch.nn.Conv3d(hd, hd, 3, stride=stride, padding=1, groups=hd, bias=False).double()
    norm = torch.nn.LayerNorm(hd).double()
    with torch.no_grad():
        conv.weight.copy_(rnd(hd, 1, 3, 3, 3, seed=2) * 0.3)
        norm.weight.copy_(1 + 0.1 * rnd(hd, seed=3))
        norm.bias.copy_(0.1 * rnd(hd, seed=4))
    xr = q(x, dtype).requires_grad_(True)
    t4 = xr.reshape(B, 1 + T * H * W, heads, hd).permute(0, 2, 1, 3)
    ref, thw2 = MO.attention_pool(t4, conv, list(thw), norm)
    ref = ref.permute(0, 2, 1, 3).reshape(B, -1, C)
    dy = rnd(*ref.shape, seed=5)
    ref.backward(q(dy, dtype))
    xg = x.to(DEV).to(dtype).requires_grad_(True)
    cw = conv.weight.detach().float().to(DEV).requires_grad_(True)
    gw = norm.weight.detach().float().to(DEV).requires_grad_(True)
    gb = norm.bias.detach().float().to(DEV).requires_grad_(True)
    y = F_.PoolConvLNFn.apply(xg, cw, gw, gb, list(thw), heads, stride, norm.eps)
    assert F_.pooled_thw(list(thw), stride) == thw2
    y.backward(dy.to(DEV).to(dtype))
    tag = f'pool {dtype} hd={hd} s={stride}'
    check(f'{tag} y', y.float().cpu(), ref.detach(), TOL[dtype])
    check(f'{tag} dx', xg.grad.float().cpu(), xr.grad, 2 * TOL[dtype])
    check(f'{tag} dw', cw.grad.cpu(), conv.weight.grad, 2 * TOL[dtype])
    check(f'{tag} dgamma', gw.grad.cpu(), norm.weight.grad, 2 * TOL[dtype])
    check(f'{tag} dbeta', gb.grad.cpu(), norm.bias.grad, 2 * TOL[dtype])


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('thw', [(2, 8, 8), (3, 7, 9)])
def test_maxpool_skip(dtype, thw):
    from vtx import functions as F_
    B, C = 2, 40
    T, H, W = thw
    x = rnd(B, 1 + T * H * W, C, seed=1)
    xr = q(x, dtype).requires_grad_(True)
    pool = torch.nn.MaxPool3d([1, 3, 3], [1, 2, 2], [0, 1, 1])
    ref, _ = MO.attention_pool(xr, pool, list(thw))
    dy = rnd(*ref.shape, seed=2)
    ref.backward(q(dy, dtype))
    xg = x.to(DEV).to(dtype).requires_grad_(True)
    y = F_.MaxPoolSkipFn.apply(xg, list(thw))
    y.backward(dy.to(DEV).to(dtype))
    assert torch.equal(y.float().cpu().double(), ref.detach()), 'max pooling is exact'
    check(f'maxpool {dtype} dx', xg.grad.float().cpu(), xr.grad, 1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('hd,heads,Lq,Lk', [(96, 2, 130, 37), (96, 1, 300, 393), (64, 3, 33, 200), (96, 2, 2500, 393), (96, 4, 1569, 128),
                                           (64, 2, 700, 161)])
def test_cross_attention(dtype, hd, heads, Lq, Lk):
    from vtx import functions as F_
    B, C = 2, heads * hd
    qq, kk, vv = rnd(B, Lq, C, seed=1), rnd(B, Lk, C, seed=2), rnd(B, Lk, C, seed=3)
    ref_in = [q(t, dtype).requires_grad_(True) for t in (qq, kk, vv)]
    sp = lambda t, L: t.reshape(B, L, heads, hd).permute(0, 2, 1, 3)   # noqa: E731
    att = ((sp(ref_in[0], Lq) @ sp(ref_in[1], Lk).transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    ref = (att @ sp(ref_in[2], Lk)).transpose(1, 2).reshape(B, Lq, C)
    do = rnd(B, Lq, C, seed=4)
    ref.backward(q(do, dtype))
    g = [t.to(DEV).to(dtype).requires_grad_(True) for t in (qq, kk, vv)]
    o = F_.XAttnFn.apply(g[0], g[1], g[2], heads)
    o.backward(do.to(DEV).to(dtype))
    tag = f'xattn {dtype} hd={hd} {Lq}x{Lk}'
    check(f'{tag} out', o.float().cpu(), ref.detach(), TOL[dtype])
    for name, a, b in zip('qkv', g, ref_in):
        check(f'{tag} d{name}', a.grad.float().cpu(), b.grad, 2 * TOL[dtype])
    if dtype == torch.bfloat16:                      # the MFMA kernels (xattn_mfma.hip): deterministic, and equal to the VALU ones
        import vtx                                   # of mvit.hip within bf16 rounding
        g2 = [t.to(DEV).to(dtype).requires_grad_(True) for t in (qq, kk, vv)]
        o2 = F_.XAttnFn.apply(g2[0], g2[1], g2[2], heads)
        o2.backward(do.to(DEV).to(dtype))
        assert torch.equal(o2, o) and all(torch.equal(a.grad, b.grad) for a, b in zip(g, g2)), tag + ': not deterministic'
        vtx.set_option('attn_valu', 1)
        try:
            g3 = [t.to(DEV).to(dtype).requires_grad_(True) for t in (qq, kk, vv)]
            o3 = F_.XAttnFn.apply(g3[0], g3[1], g3[2], heads)
            o3.backward(do.to(DEV).to(dtype))
        finally:
            vtx.set_option('attn_valu', 0)
        check(f'{tag} out mfma vs valu', o.float().cpu(), o3.float().cpu(), TOL[dtype])
        for name, a, b in zip('qkv', g, g3):
            check(f'{tag} d{name} mfma vs valu', a.grad.float().cpu(), b.grad.float().cpu(), 2 * TOL[dtype])


@pytest.mark.parametrize('dtype', DT)
def test_pos_encoding_and_stem(dtype):
    from vtx import functions as F_
    B, T, H, W, C = 2, 2, 4, 4, 96
    enc = MO.SpatioTemporalClsPositionalEncoding(C, [T, H, W]).double()
    with torch.no_grad():
        for i, p in enumerate(enc.parameters()):
            p.copy_(rnd(*p.shape, seed=10 + i))
    x = rnd(B, T * H * W, C, seed=1)
    xr = q(x, dtype).requires_grad_(True)
    ref = enc(xr)
    dy = rnd(*ref.shape, seed=2)
    ref.backward(q(dy, dtype))
    ps = [p.detach().float().to(DEV).requires_grad_(True) for p in (enc.cls_token, enc.pos_embed_class, enc.pos_embed_spatial,
                                                                      enc.pos_embed_temporal)]
    xg = x.to(DEV).to(dtype).requires_grad_(True)
    y = F_.PosEncodingFn.apply(xg, *ps)
    y.backward(dy.to(DEV).to(dtype))
    check(f'pos enc {dtype}', y.float().cpu(), ref.detach(), TOL[dtype])
    check(f'pos enc dx {dtype}', xg.grad.float().cpu(), xr.grad, TOL[dtype])
    for name, a, b in zip(('cls', 'class', 'spatial', 'temporal'), ps, (enc.cls_token, enc.pos_embed_class,
                                                                         enc.pos_embed_spatial, enc.pos_embed_temporal)):
        check(f'pos enc d{name} {dtype}', a.grad.cpu(), b.grad, 2 * TOL[dtype])
    # stem: Conv3d(3 -> 96, (3,7,7), stride (2,4,4), padding (1,3,3)) on a [B,T,C,H,W] clip
    conv = torch.nn.Conv3d(3, 96, (3, 7, 7), stride=(2, 4, 4), padding=(1, 3, 3)).double()
    clip = rnd(2, 4, 3, 16, 24, seed=3)
    wq = q(conv.weight.detach().float(), dtype)
    refs = torch.nn.functional.conv3d(q(clip, dtype).transpose(1, 2), wq, conv.bias, stride=(2, 4, 4), padding=(1, 3, 3))
    refs = refs.flatten(2).transpose(1, 2)
    cw = conv.weight.detach().float().to(DEV).requires_grad_(True)
    cb = conv.bias.detach().float().to(DEV).requires_grad_(True)
    ys = F_.ConvStemFn.apply(clip.to(DEV), cw, cb, (2, 4, 4), (1, 3, 3), dtype)
    check(f'conv stem {dtype}', ys.float().cpu(), refs.detach(), TOL[dtype])
    dy = rnd(*ys.shape, seed=4)
    ys.backward(dy.to(DEV).to(dtype))
    conv.zero_grad()
    wr = conv.weight.detach().clone().requires_grad_(True)
    out = torch.nn.functional.conv3d(q(clip, dtype).transpose(1, 2), wr, conv.bias, stride=(2, 4, 4), padding=(1, 3, 3))
    out.flatten(2).transpose(1, 2).backward(q(dy, dtype))
    check(f'conv stem dw {dtype}', cw.grad.cpu(), wr.grad, 2 * TOL[dtype])


def _share(ours, oracle, seed):
    sd = synth.synth_state_dict(synth.shapes_of(oracle), seed)
    for k in list(sd):                              # depthwise pooling kernels: fan-in 27, keep them O(1)
        if 'pool_' in k:
            sd[k] = sd[k] * 3.0
        if 'pos_embed' in k:
            sd[k] = synth.synth_tensor(k, tuple(sd[k].shape), seed) * 0.02
    oracle.load_state_dict(sd, strict=True)
    missing, unexpected = ours.load_state_dict(sd, strict=True), None
    return sd


@pytest.mark.parametrize('prec,tol', [('fp32', TOL_F32), ('bf16', 2 * TOL_BF16)])
def test_reduced_mvit_all_gradients(prec, tol):
    """A 5-block MViT (two Q-pooling stages, adaptive K/V strides, widening, head doubling) on a 4x32x32 token grid:
    outputs and every parameter gradient against the oracle with shared weights."""
    import vtx
    import mvit
    vtx.set_precision(prec)
    try:
        kw = dict(depth=5, patch_embed_dim=96, num_heads=1, embed_dim_mul=[[1, 2.0], [3, 2.0]], atten_head_mul=[[1, 2.0], [3, 2.0]],
                  pool_q_stride_size=[[1, 1, 2, 2], [3, 1, 2, 2]], pool_kv_stride_adaptive=[1, 4, 4], pool_kvq_kernel=[3, 3, 3])
        ours = mvit.create_multiscale_vision_transformers(spatial_size=128, temporal_size=8, **kw)
        oracle = MO.MultiscaleVisionTransformers(spatial_size=128, temporal_size=8, **kw)
        _share(ours, oracle, 5)
        oracle.double()                             # an fp32 CPU run is itself ~5e-3 off on the thinly-summed gradients
        ours.to(DEV)
        x = rnd(2, 4 * 32 * 32, 96, seed=7)
        yo = oracle(x.double())
        w = rnd(*yo.shape, seed=8)
        (yo * w.double()).sum().backward()
        y = ours(x.to(DEV))
        (y * w.to(DEV)).sum().backward()
        assert y.dtype == torch.float32 and y.shape == yo.shape
        check(f'reduced MViT {prec} out', y.cpu(), yo.detach(), tol)
        go = dict(oracle.named_parameters())
        # bf16 yardstick: the oracle itself under torch.autocast(bfloat16) (what Lightning precision=16 does to the
        # reference's backbone): some gradients are sums over very few token rows (pos_embed_spatial: batch x frames)
        # and inherit the element-wise bf16 noise of the stream gradient un-averaged
        ac = {}
        if prec == 'bf16':
            import copy
            o32 = copy.deepcopy(oracle).float()
            o32.zero_grad()
            with torch.autocast('cpu', dtype=torch.bfloat16):
                ya = o32(x)
            (ya.float() * w).sum().backward()
            ac = {k: (p.grad.double() - go[k].grad).norm().item() / max(go[k].grad.norm().item(), 1e-30)
                  for k, p in o32.named_parameters()}
        worst = 0.0
        # norm_k.bias shifts every key by the same vector, which softmax ignores: its exact gradient is 0 -- errors are
        # measured against the larger of a tensor's own gradient norm and 1e-3 of the typical one
        typical = sorted(g.grad.norm().item() for g in go.values())[len(go) // 2]
        for k, p in ours.named_parameters():
            ref = go[k].grad
            if ref.norm().item() < 1e-3 * typical:     # an exactly-zero gradient (norm_k.bias): only noise to bound
                assert p.grad.norm().item() <= 5e-3 * typical, (k, p.grad.norm().item(), typical)
                continue
            e = (p.grad.cpu().double() - ref).norm().item() / max(ref.norm().item(), 1e-3 * typical)
            worst = max(worst, e)
            bar = 1e-3 if prec == 'fp32' else max(2.0 * ac[k] * ref.norm().item() / max(ref.norm().item(), 1e-3 * typical), 3e-2)
            assert e <= bar, (k, e, bar)
        from helpers import report
        report(f'ok   reduced MViT {prec}: {len(go)} parameter gradients, worst l2-rel={worst:.3e}')
    finally:
        vtx.set_precision('auto')


def _maskfeat_full(seed=9):
    """MaskFeat as the reference's trainer builds it (model_trainer.py:53-54) with the synthetic weights of
    oracle/mvit_cases.py, on the device."""
    import video_transformer as V
    from oracle import mvit_cases as MC
    m = V.MaskFeat(**MC.MASKFEAT_KW)
    assert m.embed_dims == 768 and m.downsample_rate == 4
    oracle = MC.make_oracle()
    assert sorted(k for k in m.mvit.state_dict()) == sorted(oracle.state_dict())
    assert abs(sum(p.numel() for p in m.mvit.parameters()) - 36.26e6) < 0.05e6
    sd = {'mvit.' + k: v for k, v in MC.backbone_state(oracle, seed).items()}
    sd.update(MC.head_state(seed))
    m.load_state_dict(sd, strict=True)
    oracle.load_state_dict(MC.backbone_state(oracle, seed), strict=True)
    return m.to(DEV), oracle, MC


def test_maskfeat_as_the_reference_trainer_builds_it():
    """MaskFeat(pool_q_stride_size=[[1,1,2,2],[3,1,2,2]], feature_dim=216) -- model_trainer.py:53-54 -- on one 16x224^2
    clip: MViT-B (16 blocks, 36.3 M parameters, 25 089 -> 6 273 -> 1 569 tokens), decoder, HOG-target masked MSE; loss and
    gradients on the exact-fp32 path against the float64 oracle run live on the host (oracle/mvit_cases.reference: oracle
    backbone + the reference's head arithmetic), and that run against the committed golden."""
    import vtx
    from helpers import gold
    vtx.set_precision('fp32')
    try:
        m, oracle, MC = _maskfeat_full()
        x, target, mask, markers = MC.inputs()
        pred, loss = m(x.to(DEV), target.to(DEV), mask.to(DEV), markers)
        loss.backward()
        p, ref_loss, go = MC.reference(oracle, MC.head_state(9), x, target, mask, markers)
        g = gold('maskfeat_mvit_b_full.npz')
        assert abs(float(ref_loss) - float(g['loss'])) <= 1e-9 * abs(float(g['loss'])), 'the live oracle run is not the golden one'
        check('MaskFeat/MViT-B pred', pred.cpu(), p, 1e-3)
        lg, lr_ = float(loss.detach()), float(ref_loss)
        assert abs(lg - lr_) <= 1e-4 * abs(lr_), (lg, lr_)
        ours = dict(m.named_parameters())
        for k in ('decoder_pred.weight', 'mask_token', 'patch_embed.patch_model.weight', 'mvit.blocks.0.attn.pool_k.weight',
                  'mvit.blocks.1.attn.pool_q.weight', 'mvit.blocks.3.attn.q.weight', 'mvit.blocks.13.proj.weight',
                  'mvit.cls_positional_encoding.pos_embed_spatial', 'mvit.blocks.15.mlp.fc2.bias', 'mvit.blocks.0.norm1.weight'):
            a, b = ours[k].grad.cpu().double(), go[k]
            e = (a - b).norm().item() / max(b.norm().item(), 1e-30)
            # pos_embed_spatial sums 8 token rows of an fp32 gradient that went through 16 blocks of LayerNorm
            # backward on a sparse (masked) loss: fp32 round-off is not averaged there (an fp32 CPU run of the oracle
            # is itself ~1e-2 off its float64 run on this tensor); the reduced model above pins it to 1e-6 in fp32
            assert e < (2e-2 if 'pos_embed_spatial' in k else 1e-3), (k, e)
    finally:
        vtx.set_precision('auto')


def test_maskfeat_mvit_b_full_size_bf16():
    """The same full-size case on the bf16 path -- the precision BASELINE cfg 4 is specified in -- with FIXED bars:
    prediction within 2e-2 of max|ref|, loss within 1e-2, and EVERY one of the 364 parameter gradients within
    max(2 x ae, 1e-2) relative L2 of the float64 oracle, where ae is what the oracle graph itself deviates under
    torch.autocast(bfloat16) on that tensor (committed in the golden, tests/golden/make_golden_mvit.py; the same rule and
    floor as the TimeSformer bars in helpers.py) -- no wider escape; the median no worse than 1.25 x the autocast median.
    Gradients that vanish identically (norm_k.bias) only have noise to bound.  Backbone parity is UNPINNED (see module
    docstring): this measures the bf16 path against the restatement, not against pytorchvideo."""
    import vtx
    from helpers import AUTOCAST_FACTOR, AUTOCAST_FLOOR, gold, report, sample_idx
    vtx.set_precision('bf16')
    try:
        m, _, MC = _maskfeat_full()
        x, target, mask, markers = MC.inputs()
        pred, loss = m(x.to(DEV), target.to(DEV), mask.to(DEV), markers)
        loss.backward()
        g = gold('maskfeat_mvit_b_full.npz')
        pf = pred.detach().double().cpu().flatten()
        e_pred = (pf[sample_idx(pf.numel())] - torch.as_tensor(g['pred_s']).double()).abs().max().item() / float(g['pred_n'][1])
        assert e_pred <= 2e-2, e_pred
        assert abs(pf.norm().item() - float(g['pred_n'][0])) <= 1e-2 * float(g['pred_n'][0])
        assert abs(float(loss.detach()) - float(g['loss'])) <= 1e-2 * float(g['loss'])
        typical = float(g['typical_norm'])
        ours, theirs, n, large = [], [], 0, []
        for k, p in m.named_parameters():
            got = p.grad.detach().double().cpu().flatten()
            if 'zero:' + k in g.files:
                assert got.norm().item() <= 5e-3 * typical, (k, got.norm().item())
                continue
            ae = float(g['ae:' + k])
            if 'g:' + k in g.files:
                ref = torch.as_tensor(g['g:' + k]).double().flatten()
                e = (got - ref).norm().item() / ref.norm().item()
            else:
                ref, gn = torch.as_tensor(g['gs:' + k]).double(), float(g['gn:' + k][0])
                part = got[sample_idx(got.numel())]
                rms_norm = gn / got.numel() ** 0.5 * ref.numel() ** 0.5      # the norm that many typical elements would have
                e = max((part - ref).norm().item() / max(ref.norm().item(), rms_norm), abs(got.norm().item() - gn) / gn)
            bar = max(AUTOCAST_FACTOR * ae, AUTOCAST_FLOOR)
            if e > bar:
                report(f'FAIL MaskFeat/MViT-B bf16 grad {k}: l2-rel={e:.3e} bar={bar:.3e} (autocast {ae:.3e})')
            assert e <= bar, (k, e, bar, ae)
            if e > 5e-2:
                # every tensor whose RELATIVE error is large is named, with the size of the reference gradient next to the
                # typical parameter gradient of the model: a relative figure only means something when the tensor has a gradient
                rn = (ref.norm().item() if 'g:' + k in g.files else gn)
                report(f'note MaskFeat/MViT-B bf16 grad {k}: l2-rel {e:.3e} (oracle under autocast {ae:.3e}); |ref| = {rn:.3e} = '
                       f'{rn / typical:.2e} x the typical gradient norm; absolute error {e * rn / typical:.2e} x typical')
                large.append((k, e, ae, rn / typical))
            ours.append(e)
            theirs.append(ae)
            n += 1
        mo, mt = sorted(ours)[n // 2], sorted(theirs)[n // 2]
        report(f'ok   MaskFeat/MViT-B bf16 full size: pred {e_pred:.3e}, {n} gradients, median l2 {mo:.3e} vs oracle-autocast median '
               f'{mt:.3e}, worst {max(ours):.3e} (oracle-autocast worst {max(theirs):.3e})')
        assert mo <= 1.25 * mt, (mo, mt)
        report(f'MaskFeat/MViT-B bf16: {len(large)} of {n} gradients above 5e-2 relative: ' + ', '.join(f'{k} ({e:.2e}; |ref| {r:.1e} x typical)' for k, e, _, r in large))
        # a 34 % gradient error must not pass on the strength of an unpinned oracle alone.  What it is (and is not):
        # exactly ONE tensor is above 5e-2, pos_embed_spatial, and its gradient is LARGE (11 x the typical norm), so this is no
        # vanishing / cancelling gradient.  It is d(loss)/d(backbone input) summed over the 8 frames only (one clip), i.e. an
        # 8-term sum per element of a quantity that 16 blocks of bf16 arithmetic with this case's synthetic weights have
        # already made ~35 % wrong ELEMENT-WISE -- in every bf16 implementation: the oracle graph under torch.autocast shows
        # 0.359 on the same tensor, and the check below measures it inside THIS library, between its own exact-fp32 path (which
        # matches the float64 oracle to 1e-6, test_maskfeat_as_the_reference_trainer_builds_it) and its bf16 path.  Every
        # parameter that averages the same input gradient over thousands of tokens (stem convolution, temporal / class position
        # embeddings, mask token) is inside 5e-2.  So: the flagged tensors must be sums of FEW input-gradient terms and must
        # not be worse than the element-wise input gradient itself.
        if large:
            def input_grad(prec):
                vtx.set_precision(prec)
                mm, _, _ = _maskfeat_full()
                cap = {}
                def fwd_hook(mod, inp, out):            # (returns None: a forward hook's return value would replace the output)
                    out.register_hook(lambda gr: cap.__setitem__('g', gr.detach().double().cpu()))
                h = mm.mvit.cls_positional_encoding.register_forward_hook(fwd_hook)
                _, ls = mm(x.to(DEV), target.to(DEV), mask.to(DEV), markers)
                ls.backward()
                h.remove()
                return cap['g'], mm.mvit.cls_positional_encoding.pos_embed_spatial.grad.detach().double().cpu()
            dx16, ps16 = input_grad('bf16')
            dx32, ps32 = input_grad('fp32')
            e_dx = (dx16 - dx32).norm().item() / dx32.norm().item()
            e_ps = (ps16 - ps32).norm().item() / ps32.norm().item()
            report(f'MaskFeat/MViT-B: input gradient d(loss)/d(tokens) bf16 path vs fp32 path of this library: l2-rel {e_dx:.3e}; '
                   f'pos_embed_spatial (its sum over 8 frames) bf16 vs fp32 path: {e_ps:.3e}')
            for k, e, ae, r in large:
                assert k.endswith('pos_embed_spatial'), f'{k}: {e:.3e} relative with |ref| = {r:.2e} x typical -- not a known short sum of input gradients'
                assert e <= 1.25 * e_dx, (k, e, e_dx)
    finally:
        vtx.set_precision('auto')
