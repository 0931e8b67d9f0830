"""Kernel-level parity: every libvtx entry point against a plain fp32/fp64 PyTorch-CPU
restatement of the same op on seeded inputs (called through the C ABI via vtx.ops).

fp32 path bar: 1e-3 of max|ref| (BASELINE north_star); in practice ~1e-6.
bf16 path: inputs are rounded to bf16 first, the reference is computed in fp64 from the
rounded inputs, bar 1e-2 (one bf16 output rounding is 2^-9 = 2e-3 of the element).
Bit-exact: patch gather (fp32), HOG features and bins.
"""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from helpers import check, gold, relerr, report

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dtype):
    """round to the compute dtype and come back to fp64 (what the kernel actually sees)."""
    return t.to(dtype).double()


def dev(t, dtype=None):
    return (t if dtype is None else t.to(dtype)).to(DEV).contiguous()


def phys(m, r):
    return m.base + r + ((r // m.grp) * m.skip if m.grp > 0 else 0)


def test_selftest():
    import vtx
    buf = ctypes.create_string_buffer(16384)
    fails = vtx.load().vtx_selftest(buf, len(buf))
    report('selftest:\n' + buf.value.decode())
    assert fails == 0, buf.value.decode()


# --------------------------------------------------------------------------- weights x weights (fp32)
@pytest.mark.parametrize('ta', [False, True])
@pytest.mark.parametrize('tb', [False, True])
@pytest.mark.parametrize('N1,N2,K', [(768, 768, 768), (128, 192, 96), (68, 132, 36)])
def test_wprod(ta, tb, N1, N2, K):
    """vtx_wprod = alpha * op(A) op(B) + u v^T (+ C), and y = alpha_y * op(A) x + beta_z * z (+ y), every transposition,
    full and ragged tiles -- the five rocBLAS products of the merged attn.proj o temporal_fc path (reference
    transformer.py:268-275) as three launches of this library."""
    from vtx import ops
    A = rnd(*((K, N1) if ta else (N1, K)), seed=1)
    B = rnd(*((N2, K) if tb else (K, N2)), seed=2)
    u, v, x, z = rnd(N1, seed=3), rnd(N2, seed=4), rnd(K, seed=5), rnd(N1, seed=6)
    C0, y0 = rnd(N1, N2, seed=7), rnd(N1, seed=8)
    Ad, Bd = (A.t() if ta else A).double(), (B.t() if tb else B).double()
    ref = 0.75 * Ad @ Bd
    got = ops.wprod(dev(A), dev(B), ta=ta, tb=tb, alpha=0.75)
    check(f'wprod[{ta},{tb},{N1}x{N2}x{K}] plain', got, ref, 1e-5)
    C, y = dev(C0), dev(y0)
    ops.wprod(dev(A), dev(B), ta=ta, tb=tb, alpha=-1.5, out=C, accumulate=True, u=dev(u), v=dev(v), x=dev(x), y=y,
              alpha_y=2.0, z=dev(z), beta_z=0.25, y_accumulate=True)
    check(f'wprod[{ta},{tb},{N1}x{N2}x{K}] C', C, C0.double() - 1.5 * Ad @ Bd + torch.outer(u, v).double(), 1e-5)
    check(f'wprod[{ta},{tb},{N1}x{N2}x{K}] y', y, y0.double() + 2.0 * Ad @ x.double() + 0.25 * z.double(), 1e-5)
    _, y2 = ops.wprod(dev(A), dev(B), ta=ta, tb=tb, x=dev(x))
    check(f'wprod[{ta},{tb},{N1}x{N2}x{K}] y plain', y2, Ad @ x.double(), 1e-5)
    again = ops.wprod(dev(A), dev(B), ta=ta, tb=tb, alpha=0.75)
    assert torch.equal(got, again), 'wprod is not deterministic'


# --------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('D', [128, 768, 1024])
def test_layernorm_fwd_bwd(dtype, D):
    from vtx import ops
    B, N = 3, 37
    x = rnd(B, 1 + N, D, seed=1) * 2 + 0.5
    gamma, beta = 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    dy = rnd(B * N, D, seed=4)
    dres = rnd(B, 1 + N, D, seed=5)
    xq = q(x, dtype).requires_grad_(True)
    gq, bq = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xq[:, 1:], (D,), gq, bq, 1e-5).reshape(B * N, D)
    ref.backward(q(dy, dtype))
    dx_ref = xq.grad + q(dres, dtype)
    dx_ref[:, 0] = 0
    tm = ops.tokmap(N)
    xd = dev(x, dtype)
    y = torch.empty(B * N, D, dtype=dtype, device=DEV)
    mean = torch.empty(B * N, device=DEV)
    rstd = torch.empty(B * N, device=DEV)
    ops.layernorm_fwd(xd, B * N, D, D, tm, dev(gamma), dev(beta), 1e-5, y, D, mean=mean, rstd=rstd)
    check(f'ln_fwd {dtype} D={D}', y.float().cpu(), ref.detach(), TOL[dtype])
    check(f'ln_mean {dtype} D={D}', mean.cpu(), xq[:, 1:].detach().mean(-1).reshape(-1), 1e-4)
    dx = torch.zeros(B, 1 + N, D, dtype=dtype, device=DEV)
    dg = torch.zeros(D, device=DEV)
    db = torch.zeros(D, device=DEV)
    ops.layernorm_bwd(dev(dy, dtype), D, ops.IDENT, xd, D, tm, B * N, D, mean, rstd, dev(gamma), dev(dres, dtype), dx, D, dg, db)
    check(f'ln_bwd dx {dtype} D={D}', dx.float().cpu(), dx_ref, TOL[dtype])
    check(f'ln_bwd dgamma {dtype} D={D}', dg.cpu(), gq.grad, TOL[dtype])
    check(f'ln_bwd dbeta {dtype} D={D}', db.cpu(), bq.grad, TOL[dtype])


@pytest.mark.parametrize('dtype', DTYPES)
def test_integration_stub_layer_norm(dtype):
    """The ctypes stub INTEGRATION.md section 2 documents, extracted from the document and executed as written (its own
    CDLL handle, its own RowMap), against torch.nn.functional.layer_norm in float64 -- reference transformer.py:519."""
    from helpers import exec_integration_stub
    ns = exec_integration_stub()
    rows, D = 1569, 768
    x = rnd(rows, D, seed=11) * 2 + 0.5
    gamma, beta = 1 + 0.1 * rnd(D, seed=12), 0.1 * rnd(D, seed=13)
    y = ns['layer_norm_hip'](dev(x, dtype), dev(gamma), dev(beta), 1e-5)
    torch.cuda.synchronize()
    assert y.dtype == dtype and y.shape == (rows, D)
    ref = torch.nn.functional.layer_norm(q(x, dtype), (D,), gamma.double(), beta.double(), 1e-5)
    check(f'INTEGRATION.md stub layer_norm_hip {dtype}', y.float().cpu(), ref, TOL[dtype])


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('D', [128, 200, 768, 1024])
def test_layernorm_fwd_rows_per_trip(dtype, D, vtx_opts):
    """The forward kernel with 2 / 3 / 4 rows per trip (all rows requested before the first is reduced; the default is 3)
    against the one-row-per-wave kernel: the same arithmetic per row -- bit-identical outputs and statistics --, row counts
    that do not divide by the rows per trip, row maps on input and output, D that is and is not a multiple of 256."""
    from vtx import ops
    B, N = 5, 43                                    # 215 rows
    x = dev(rnd(B, 1 + N, D, seed=1) * 2 + 0.5, dtype)
    gamma, beta = dev(1 + 0.1 * rnd(D, seed=2)), dev(0.1 * rnd(D, seed=3))
    tm = ops.tokmap(N)
    outs = []
    for nr in ('1', '2', '3', '4'):
        vtx_opts('ln_rows', nr)
        y = torch.zeros(B, 1 + N, D, dtype=dtype, device=DEV)
        mean, rstd = torch.zeros(B * N, device=DEV), torch.zeros(B * N, device=DEV)
        ops.layernorm_fwd(x, B * N, D, D, tm, gamma, beta, 1e-5, y, D, mean=mean, rstd=rstd, ymap=tm)
        outs.append((y, mean, rstd))
    xq = x.double().cpu()[:, 1:]
    ref = torch.nn.functional.layer_norm(xq, (D,), gamma.double().cpu(), beta.double().cpu(), 1e-5)
    check(f'ln_fwd rows-per-trip {dtype} D={D}', outs[2][0].float().cpu()[:, 1:], ref, TOL[dtype])
    assert not outs[2][0][:, 0].any(), 'a cls row (skipped by the output map) was written'
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o)), f'{dtype} D={D}'


@pytest.mark.parametrize('dtype', DTYPES)
def test_fact_glue_fwd_bwd(dtype):
    """vtx_fact_glue_fwd / _bwd against the reference's expressions (video_transformer.py:515-523), the `x[:b, 0]` quirk
    included: the cls rows of the temporal encoder's input are the first b rows of the flattened (b t) axis."""
    from vtx import ops
    b, T, P, D = 3, 5, 14, 128
    x = rnd(b * T, 1 + P, D, seed=1)
    e = rnd(1, 1 + T, D, seed=2)
    dh = rnd(b, 1 + T, D, seed=3)
    xq = q(x, dtype).requires_grad_(True)
    eq = e.double().requires_grad_(True)
    cls_tokens = xq[:b, 0, :].unsqueeze(1)
    frames = xq[:, 1:, :].reshape(b, T, P, D).mean(2)
    ref = torch.cat((cls_tokens, frames), dim=1) + eq
    ref.backward(q(dh, dtype))
    h = ops.fact_glue_fwd(dev(x, dtype), dev(e).reshape(1 + T, D), b, T, P, D)
    check(f'fact_glue fwd {dtype}', h.float().cpu(), ref.detach(), TOL[dtype])
    de = torch.full((1 + T, D), float('nan'), device=DEV)
    dx = ops.fact_glue_bwd(dev(dh, dtype), b, T, P, D, d_time_embed=de)
    check(f'fact_glue dx {dtype}', dx.float().cpu(), xq.grad, TOL[dtype])
    check(f'fact_glue d_time_embed {dtype}', de.cpu(), eq.grad.reshape(1 + T, D), 1e-5)
    de2 = torch.ones(1 + T, D, device=DEV)
    ops.fact_glue_bwd(dev(dh, dtype), b, T, P, D, d_time_embed=de2, accumulate=True)
    check(f'fact_glue d_time_embed accumulate {dtype}', de2.cpu(), eq.grad.reshape(1 + T, D) + 1, 1e-5)


# --------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('variant', ['pp256', 'ring256x3', 'ring256x3k32', 'ring256x4k32', 'ring128x3', 'ring128x4k32', 'dma2'])
def test_gemm_nt_bf16_variants(variant, vtx_opts):
    """Every staging variant of the bf16 NT GEMM (2-buffer DMA, DMA rings with counted vmcnt)."""
    from vtx import ops
    vtx_opts('gemm_nt', variant)
    for (M, N, K) in [(1568, 2304, 768), (3000, 216, 3072), (1030, 768, 192), (12544, 768, 768), (777, 1000, 128)]:
        A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2) * K ** -0.5, rnd(N, seed=3)
        ref = q(A, torch.bfloat16) @ q(W, torch.bfloat16).t() + b.double()
        C = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=DEV)
        ops.gemm_nt(dev(A, torch.bfloat16), dev(W, torch.bfloat16), C, M, N, K, bias=dev(b))
        check(f'gemm_nt {variant} {M}x{N}x{K}', C.float().cpu(), ref, 1e-2)
        if variant == 'pp256':
            # the persistent kernel draws tiles from per-XCD counters: any grid (= any number of resident
            # workgroups) must produce the same result, and the counters must be left clean for the next launch
            for grid in ('40', '8', '256'):
                vtx_opts('pp_grid', grid)
                C2 = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=DEV)
                ops.gemm_nt(dev(A, torch.bfloat16), dev(W, torch.bfloat16), C2, M, N, K, bias=dev(b))
                assert torch.equal(C2, C), f'pp256 grid {grid}: result depends on the grid size'
            vtx_opts('pp_grid', '256')


@pytest.mark.parametrize('nodma', ['0', '1'])
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K', [(300, 256, 128), (1568, 2304, 768), (130, 216, 768), (257, 768, 96), (64, 8, 8),
                                   (12544, 768, 3072)])
def test_gemm_nt_plain(dtype, M, N, K, nodma, vtx_opts):
    """nodma=0: LDS-DMA staged kernel when K % 64 == 0 (bf16); nodma=1: register-staged kernel."""
    from vtx import ops
    if nodma == '1' and dtype == torch.float32:
        pytest.skip('fp32 has a single kernel')
    vtx_opts('gemm_nodma', nodma)
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2) * K ** -0.5, rnd(N, seed=3)
    ref = q(A, dtype) @ q(W, dtype).t() + b.double()
    C = torch.full((M, N), float('nan'), dtype=dtype, device=DEV)
    ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, N, K, bias=dev(b))
    check(f'gemm_nt {dtype} {M}x{N}x{K}', C.float().cpu(), ref, TOL[dtype])


@pytest.mark.parametrize('dtype', DTYPES)
def test_gemm_nt_epilogues(dtype):
    from vtx import ops
    B, N, T, D, Hd = 2, 24, 4, 128, 256            # N = P*T tokens per clip, P = 6
    M = B * N
    tm = ops.tokmap(N)
    A, W, bias = rnd(M, D, seed=1), rnd(Hd, D, seed=2) * D ** -0.5, rnd(Hd, seed=3)
    Aq, Wq = q(A, dtype), q(W, dtype)
    # (1) bias + GELU with pre-activation copy
    pre = Aq @ Wq.t() + bias.double()
    C = torch.empty(M, Hd, dtype=dtype, device=DEV)
    C2 = torch.empty(M, Hd, dtype=dtype, device=DEV)
    ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, Hd, D, bias=dev(bias), act=1, C2=C2)
    check(f'gemm gelu {dtype}', C.float().cpu(), torch.nn.functional.gelu(pre), TOL[dtype])
    check(f'gemm preact {dtype}', C2.float().cpu(), pre, TOL[dtype])
    # (2) GELU' multiply (FFN backward)
    h = rnd(M, Hd, seed=4)
    hq = q(h, dtype).requires_grad_(True)
    torch.nn.functional.gelu(hq).sum().backward()
    ref = (Aq @ Wq.t()) * hq.grad
    ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, Hd, D, dgelu_in=dev(h, dtype))
    check(f'gemm dgelu {dtype}', C.float().cpu(), ref, TOL[dtype])
    # (2b) act 2: GELU with its derivative as second output; dgelu_kind 1: multiply by that output as stored
    preq = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(preq).sum().backward()
    ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, Hd, D, bias=dev(bias), act=2, C2=C2)
    check(f'gemm gelu (act 2) {dtype}', C.float().cpu(), torch.nn.functional.gelu(pre), TOL[dtype])
    check(f"gemm gelu' out {dtype}", C2.float().cpu(), preq.grad, TOL[dtype])
    ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, Hd, D, dgelu_in=dev(h, dtype), dgelu_kind=1)
    check(f'gemm mul {dtype}', C.float().cpu(), (Aq @ Wq.t()) * q(h, dtype), TOL[dtype])
    # (3) A rows through the token map, per-(b,p) row scale, residual + output through the map
    X = rnd(B, 1 + N, D, seed=5)
    R = rnd(B, 1 + N, Hd, seed=6)
    s = (torch.rand(M // T, generator=torch.Generator().manual_seed(7)) > 0.3).float() / 0.7
    Xq = q(X, dtype)
    ref = (Xq[:, 1:].reshape(M, D) @ Wq.t() + bias.double()) * s.double().repeat_interleave(T)[:, None]
    ref = ref.reshape(B, N, Hd) + q(R, dtype)[:, 1:]
    out = torch.zeros(B, 1 + N, Hd, dtype=dtype, device=DEV)
    ops.gemm_nt(dev(X, dtype), dev(W, dtype), out, M, Hd, D, amap=tm, cmap=tm, bias=dev(bias), row_scale=dev(s),
                rs=(T, 1, 1, 0), R=dev(R, dtype), rmap=tm)
    check(f'gemm map+scale+residual {dtype}', out.float().cpu()[:, 1:], ref, TOL[dtype])
    assert out[:, 0].abs().max().item() == 0, 'cls rows must not be touched'
    # (4) periodic residual (embedding table)
    E = rnd(N, Hd, seed=8)
    ref = (Aq @ Wq.t()).reshape(B, N, Hd) + q(E, dtype)[None]
    ops.gemm_nt(dev(A, dtype), dev(W, dtype), out, M, Hd, D, cmap=tm, R=dev(E, dtype), r_period=N)
    check(f'gemm periodic residual {dtype}', out.float().cpu()[:, 1:], ref, TOL[dtype])
    # (5) split output region + spatial scale indexing (tokens (b, p*T+t) -> s[b*T+t]; tail rows -> s[row])
    Mo = M + B * T
    A2 = rnd(Mo, D, seed=9)
    s2 = (torch.rand(B * T, generator=torch.Generator().manual_seed(10)) > 0.3).float() / 0.7
    full = q(A2, dtype) @ Wq.t() + bias.double()
    n_idx = torch.arange(M)
    tok_scale = s2.double()[(n_idx // N) * T + (n_idx % T)]
    ref_tok = (full[:M] * tok_scale[:, None]).reshape(B, N, Hd) + q(R, dtype)[:, 1:]
    ref_cls = full[M:] * s2.double()[:, None]
    out.zero_()
    a_cls = torch.empty(B * T, Hd, dtype=dtype, device=DEV)
    ops.gemm_nt(dev(A2, dtype), dev(W, dtype), out, Mo, Hd, D, cmap=tm, bias=dev(bias), row_scale=dev(s2),
                rs=(N, T, T, 1), R=dev(R, dtype), rmap=tm, split_row=M, Csplit=a_cls)
    check(f'gemm split tokens {dtype}', out.float().cpu()[:, 1:], ref_tok, TOL[dtype])
    check(f'gemm split cls rows {dtype}', a_cls.float().cpu(), ref_cls, TOL[dtype])


@pytest.mark.parametrize('epi', ['0', '1'])
@pytest.mark.parametrize('Hd', [256, 320])
def test_gemm_nt_pp_epilogues(Hd, epi, vtx_opts):
    """Every fused epilogue of the persistent 256x256 bf16 kernel (M >= 2048: the path the benchmark runs), with
    full and ragged column tiles, many tiles per workgroup (grid 8) and both epilogue structures
    (pp_epi=0: all epilogue reads ahead of the tile's stores, residual block staged through LDS;
    pp_epi=1: per-pass reads).  Reference: float64 on the CPU from the bf16-rounded operands."""
    from vtx import ops
    dtype = torch.bfloat16
    vtx_opts('gemm_nt', 'pp256')
    vtx_opts('pp_epi', epi)
    B, P, T, D = 3, 196, 4, 128                    # N = 784 tokens per clip, M = 2352 rows (10 row tiles, last ragged)
    N = P * T
    M = B * N
    tm = ops.tokmap(N)
    A, W, bias = rnd(M, D, seed=1), rnd(Hd, D, seed=2) * D ** -0.5, rnd(Hd, seed=3)
    Aq, Wq = q(A, dtype), q(W, dtype)
    for grid in ('256', '8'):
        vtx_opts('pp_grid', grid)
        tag = f'pp epi={epi} N={Hd} grid={grid}'
        # (1) bias + GELU with pre-activation copy
        pre = Aq @ Wq.t() + bias.double()
        C = torch.full((M, Hd), float('nan'), dtype=dtype, device=DEV)
        C2 = torch.full((M, Hd), float('nan'), dtype=dtype, device=DEV)
        ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, Hd, D, bias=dev(bias), act=1, C2=C2)
        check(f'{tag} gelu', C.float().cpu(), torch.nn.functional.gelu(pre), 1e-2)
        check(f'{tag} preact', C2.float().cpu(), pre, 1e-2)
        # (2) GELU' multiply (FFN backward)
        h = rnd(M, Hd, seed=4)
        hq = q(h, dtype).requires_grad_(True)
        torch.nn.functional.gelu(hq).sum().backward()
        C.fill_(float('nan'))
        ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, Hd, D, dgelu_in=dev(h, dtype))
        check(f'{tag} dgelu', C.float().cpu(), (Aq @ Wq.t()) * hq.grad, 1e-2)
        # (2b) act 2 (derivative as second output) and dgelu_kind 1 (multiply by it as stored)
        preq = pre.clone().requires_grad_(True)
        torch.nn.functional.gelu(preq).sum().backward()
        C.fill_(float('nan')); C2.fill_(float('nan'))
        ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, Hd, D, bias=dev(bias), act=2, C2=C2)
        check(f'{tag} gelu (act 2)', C.float().cpu(), torch.nn.functional.gelu(pre), 1e-2)
        check(f"{tag} gelu' out", C2.float().cpu(), preq.grad, 1e-2)
        C.fill_(float('nan'))
        ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, Hd, D, dgelu_in=dev(h, dtype), dgelu_kind=1)
        check(f'{tag} mul', C.float().cpu(), (Aq @ Wq.t()) * q(h, dtype), 1e-2)
        # (3) A rows through the token map, per-(b,p) row scale, residual + output through the map
        X, R = rnd(B, 1 + N, D, seed=5), rnd(B, 1 + N, Hd, seed=6)
        s = (torch.rand(M // T, generator=torch.Generator().manual_seed(7)) > 0.3).float() / 0.7
        ref = (q(X, dtype)[:, 1:].reshape(M, D) @ Wq.t() + bias.double()) * s.double().repeat_interleave(T)[:, None]
        ref = ref.reshape(B, N, Hd) + q(R, dtype)[:, 1:]
        out = torch.zeros(B, 1 + N, Hd, dtype=dtype, device=DEV)
        ops.gemm_nt(dev(X, dtype), dev(W, dtype), out, M, Hd, D, amap=tm, cmap=tm, bias=dev(bias), row_scale=dev(s),
                    rs=(T, 1, 1, 0), R=dev(R, dtype), rmap=tm)
        check(f'{tag} map+scale+residual', out.float().cpu()[:, 1:], ref, 1e-2)
        assert out[:, 0].abs().max().item() == 0, 'cls rows must not be touched'
        # (3b) row scale without a residual, residual without a row scale
        C.fill_(float('nan'))
        ops.gemm_nt(dev(A, dtype), dev(W, dtype), C, M, Hd, D, bias=dev(bias), row_scale=dev(s), rs=(T, 1, 1, 0))
        check(f'{tag} scale only', C.float().cpu(), (Aq @ Wq.t() + bias.double()) * s.double().repeat_interleave(T)[:, None], 1e-2)
        out.zero_()
        ops.gemm_nt(dev(A, dtype), dev(W, dtype), out, M, Hd, D, cmap=tm, bias=dev(bias), R=dev(R, dtype), rmap=tm)
        check(f'{tag} residual only', out.float().cpu()[:, 1:], (Aq @ Wq.t() + bias.double()).reshape(B, N, Hd) + q(R, dtype)[:, 1:], 1e-2)
        # (4) periodic residual (embedding table)
        E = rnd(N, Hd, seed=8)
        out.zero_()
        ops.gemm_nt(dev(A, dtype), dev(W, dtype), out, M, Hd, D, cmap=tm, R=dev(E, dtype), r_period=N)
        check(f'{tag} periodic residual', out.float().cpu()[:, 1:], (Aq @ Wq.t()).reshape(B, N, Hd) + q(E, dtype)[None], 1e-2)
        # (5) split output region + spatial scale indexing (tokens (b, p*T+t) -> s[b*T+t]; tail rows -> s[row])
        Mo = M + B * T
        A2 = rnd(Mo, D, seed=9)
        s2 = (torch.rand(B * T, generator=torch.Generator().manual_seed(10)) > 0.3).float() / 0.7
        full = q(A2, dtype) @ Wq.t() + bias.double()
        n_idx = torch.arange(M)
        tok_scale = s2.double()[(n_idx // N) * T + (n_idx % T)]
        ref_tok = (full[:M] * tok_scale[:, None]).reshape(B, N, Hd) + q(R, dtype)[:, 1:]
        out.zero_()
        a_cls = torch.full((B * T, Hd), float('nan'), dtype=dtype, device=DEV)
        ops.gemm_nt(dev(A2, dtype), dev(W, dtype), out, Mo, Hd, D, cmap=tm, bias=dev(bias), row_scale=dev(s2),
                    rs=(N, T, T, 1), R=dev(R, dtype), rmap=tm, split_row=M, Csplit=a_cls)
        check(f'{tag} split tokens', out.float().cpu()[:, 1:], ref_tok, 1e-2)
        check(f'{tag} split cls rows', a_cls.float().cpu(), full[M:] * s2.double()[:, None], 1e-2)


def test_gemm_nt_pp_epilogue_structures_agree(vtx_opts):
    """The two epilogue structures of the persistent kernel are the same arithmetic in the same order: bit-identical
    outputs; and a second stream (its own tile-counter workspace) gives the same result concurrently."""
    from vtx import ops
    dtype = torch.bfloat16
    vtx_opts('gemm_nt', 'pp256')
    M, N, K = 5000, 768, 256
    A, W, b = dev(rnd(M, K, seed=1), dtype), dev(rnd(N, K, seed=2) * K ** -0.5, dtype), dev(rnd(N, seed=3))
    R = dev(rnd(M, N, seed=4), dtype)
    s = dev((torch.rand(M, generator=torch.Generator().manual_seed(5)) > 0.3).float() / 0.7)
    outs = []
    for epi in ('0', '1'):
        vtx_opts('pp_epi', epi)
        C = torch.empty(M, N, dtype=dtype, device=DEV)
        ops.gemm_nt(A, W, C, M, N, K, bias=b, row_scale=s, R=R)
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    vtx_opts('pp_epi', '0')
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    C_main = torch.empty(M, N, dtype=dtype, device=DEV)
    C_side = torch.empty(M, N, dtype=dtype, device=DEV)
    for _ in range(3):
        ops.gemm_nt(A, W, C_main, M, N, K, bias=b, row_scale=s, R=R)
        with torch.cuda.stream(side):
            ops.gemm_nt(A, W, C_side, M, N, K, bias=b, row_scale=s, R=R)
    torch.cuda.synchronize()
    assert torch.equal(C_main, outs[0]) and torch.equal(C_side, outs[0])


@pytest.mark.parametrize('K', [128, 192, 768])
@pytest.mark.parametrize('N', [256, 320, 768])
def test_gemm_nt_pp_continuous_flow(N, K, vtx_opts):
    """Continuous flow of the persistent kernel (pp_cont=1: the next tile's first K tiles are requested inside the
    current main loop, tile indices and bias travel through LDS) against the per-tile-prologue flow (pp_cont=0):
    bit-identical outputs, for an even and an odd number of K tiles (ring parity), 2 K tiles (the minimum), ragged row
    and column tiles, one and many tiles per workgroup, identity and cls-skipping row maps on A, with and without bias /
    GELU; and both against the float64 reference."""
    from vtx import ops
    dtype = torch.bfloat16
    vtx_opts('gemm_nt', 'pp256')
    B, P, T = 3, 196, 4
    Ntok = P * T
    M = B * Ntok                                   # 2352 rows: 10 row tiles, the last one ragged
    tm = ops.tokmap(Ntok)
    X = rnd(B, 1 + Ntok, K, seed=1)
    W, bias = rnd(N, K, seed=2) * K ** -0.5, rnd(N, seed=3)
    Xd, Wd, bd = dev(X, dtype), dev(W, dtype), dev(bias)
    Xq, Wq = q(X, dtype), q(W, dtype)
    sv = (torch.rand(M // T, generator=torch.Generator().manual_seed(7)) > 0.3).float() / 0.7
    cases = {
        'plain': (dict(), Xq.reshape(-1, K)[:M] @ Wq.t()),
        'bias+map': (dict(amap=tm, bias=bd), Xq[:, 1:].reshape(M, K) @ Wq.t() + bias.double()),
        'bias+map+scale': (dict(amap=tm, bias=bd, row_scale=dev(sv), rs=(T, 1, 1, 0)),
                           (Xq[:, 1:].reshape(M, K) @ Wq.t() + bias.double()) * sv.double().repeat_interleave(T)[:, None]),
        'scale': (dict(row_scale=dev(sv), rs=(T, 1, 1, 0)),
                  (Xq.reshape(-1, K)[:M] @ Wq.t()) * sv.double().repeat_interleave(T)[:, None]),
    }
    pre = Xq[:, 1:].reshape(M, K) @ Wq.t() + bias.double()
    preq = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(preq).sum().backward()
    for grid in ('256', '8'):
        vtx_opts('pp_grid', grid)
        for name, (kw, ref) in cases.items():
            outs = []
            for cont in ('1', '0'):
                vtx_opts('pp_cont', cont)
                C = torch.full((M, N), float('nan'), dtype=dtype, device=DEV)
                ops.gemm_nt(Xd, Wd, C, M, N, K, **kw)
                outs.append(C)
            check(f'cont {name} N={N} K={K} grid={grid}', outs[0].float().cpu(), ref, 1e-2)
            assert torch.equal(outs[0], outs[1]), f'{name} N={N} K={K} grid={grid}: continuous flow changes the result'
        outs = []
        for cont in ('1', '0'):
            vtx_opts('pp_cont', cont)
            C = torch.full((M, N), float('nan'), dtype=dtype, device=DEV)
            C2 = torch.full((M, N), float('nan'), dtype=dtype, device=DEV)
            ops.gemm_nt(Xd, Wd, C, M, N, K, amap=tm, bias=bd, act=2, C2=C2)
            outs.append((C, C2))
        check(f'cont gelu N={N} K={K} grid={grid}', outs[0][0].float().cpu(), torch.nn.functional.gelu(pre), 1e-2)
        check(f"cont gelu' N={N} K={K} grid={grid}", outs[0][1].float().cpu(), preq.grad, 1e-2)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # residual-block flow (pp_cont=1 for epilogues that read a 128 x 64 block per wave: the block is requested inside
    # the last two K tiles, the next tile's operands inside the passes) against the per-tile flow
    R = rnd(B, 1 + Ntok, N, seed=6)
    Rd, Rq = dev(R, dtype), q(R, dtype)
    Hm = rnd(M, N, seed=8)
    full = Xq[:, 1:].reshape(M, K) @ Wq.t() + bias.double()
    sT = (torch.rand(B * T, generator=torch.Generator().manual_seed(9)) > 0.3).float() / 0.7
    A2 = rnd(M + B * T, K, seed=10)                # rows [M, M + B*T): split rows (no residual, own output)
    full2 = q(A2, dtype) @ Wq.t() + bias.double()
    sc2 = torch.cat([sT.double().repeat_interleave(Ntok // T).reshape(B, T, P).transpose(1, 2).reshape(-1), sT.double()])
    for grid in ('256', '8'):
        vtx_opts('pp_grid', grid)
        res = {}
        for cont in ('1', '0'):
            vtx_opts('pp_cont', cont)
            o1 = torch.zeros(B, 1 + Ntok, N, dtype=dtype, device=DEV)
            ops.gemm_nt(Xd, Wd, o1, M, N, K, amap=tm, cmap=tm, bias=bd, R=Rd, rmap=tm)
            o2 = torch.full((M, N), float('nan'), dtype=dtype, device=DEV)
            ops.gemm_nt(Xd, Wd, o2, M, N, K, amap=tm, bias=bd, R=Rd.view(-1, N)[:M].contiguous())
            o3 = torch.full((M, N), float('nan'), dtype=dtype, device=DEV)
            ops.gemm_nt(Xd, Wd, o3, M, N, K, amap=tm, dgelu_in=dev(Hm, dtype), dgelu_kind=1)
            o4 = torch.zeros(B, 1 + Ntok, N, dtype=dtype, device=DEV)
            c4 = torch.full((B * T, N), float('nan'), dtype=dtype, device=DEV)
            ops.gemm_nt(dev(A2, dtype), Wd, o4, M + B * T, N, K, cmap=tm, bias=bd, row_scale=dev(sT), rs=(Ntok, T, T, 1),
                        R=Rd, rmap=tm, split_row=M, Csplit=c4)
            res[cont] = (o1, o2, o3, o4, c4)
        tag = f'residual flow N={N} K={K} grid={grid}'
        check(f'{tag} map+residual', res['1'][0].float().cpu()[:, 1:], full.reshape(B, Ntok, N) + Rq[:, 1:], 1e-2)
        check(f'{tag} residual', res['1'][1].float().cpu(), full + Rq.reshape(-1, N)[:M], 1e-2)
        check(f'{tag} multiplier', res['1'][2].float().cpu(), (Xq[:, 1:].reshape(M, K) @ Wq.t()) * q(Hm, dtype), 1e-2)
        check(f'{tag} split tokens', res['1'][3].float().cpu()[:, 1:],
              (full2[:M] * sc2[:M, None]).reshape(B, Ntok, N) + Rq[:, 1:], 1e-2)
        check(f'{tag} split rows', res['1'][4].float().cpu(), full2[M:] * sc2[M:, None], 1e-2)
        for a, b_ in zip(res['1'], res['0']):
            assert torch.equal(a, b_), f'{tag}: the residual-block flow changes the result'
    # repeated launches on one stream (self-resetting counters, LDS hand-over words) stay identical
    vtx_opts('pp_cont', '1')
    vtx_opts('pp_grid', '256')
    first = None
    for _ in range(5):
        C = torch.empty(M, N, dtype=dtype, device=DEV)
        ops.gemm_nt(Xd, Wd, C, M, N, K, amap=tm, bias=bd)
        first = C if first is None else first
        assert torch.equal(C, first)


@pytest.mark.parametrize('K', [192, 256, 768])
def test_gemm_nt_pp_lean_passes(K, vtx_opts):
    """Lean epilogue passes of the continuous-flow kernels (scalar store base + 32-bit lane offset, software-pipelined LDS
    staging, row scales through ds_bpermute) against the general passes of the same kernels (pp_epi=4): bit-identical for
    every fused epilogue; row maps whose group boundary falls INSIDE a 16-row pass (closed form and table form), full and
    ragged tiles, split rows (general passes inside a lean launch); and against the float64 reference."""
    from vtx import ops
    dtype = torch.bfloat16
    vtx_opts('gemm_nt', 'pp256')
    vtx_opts('pp_cont', '1')
    N, grp, skip, G = 512, 300, 3, 9                # 2700 logical rows; group boundaries at local rows 44, 88, 132, ... of their tiles
    M = grp * G
    phys = M + skip * G + 1
    X, W, bias = rnd(M, K, seed=1), rnd(N, K, seed=2) * K ** -0.5, rnd(N, seed=3)
    Xd, Wd, bd = dev(X, dtype), dev(W, dtype), dev(bias)
    full = q(X, dtype) @ q(W, dtype).t() + bias.double()
    cm = ops.rowmap(grp, skip, 1)
    rows = torch.arange(M) + 1 + (torch.arange(M) // grp) * skip                      # physical row of every logical row
    tab_vals = [5 * g for g in range(G)] + [5 * G]                                    # table form: m + 5 * (m // grp), one spare entry
    tab = ops.upload_i32(tab_vals, DEV)
    tmap = ops.tabmap(grp, tab, 5)
    trows = torch.arange(M) + (torch.arange(M) // grp) * 5
    R = rnd(phys + 5 * G, N, seed=6)
    Rd, Rq = dev(R, dtype), q(R, dtype)
    Hm = rnd(M, N, seed=8)
    sv = (torch.rand(M // 4, generator=torch.Generator().manual_seed(7)) > 0.3).float() / 0.7
    svr = sv.double().repeat_interleave(4)[:, None]

    def run(**kw):
        outs = []
        for epi in ('0', '4', '6'):                 # lean passes, general passes, lean passes with the first four rolled into the last K tile
            vtx_opts('pp_epi', epi)
            C = torch.zeros(phys + 5 * G, N, dtype=dtype, device=DEV)
            C2 = torch.zeros(M, N, dtype=dtype, device=DEV)
            if kw.get('act'):
                ops.gemm_nt(Xd, Wd, C, M, N, K, C2=C2, **kw)
            else:
                ops.gemm_nt(Xd, Wd, C, M, N, K, **kw)
            outs.append((C, C2))
        for o in outs[1:]:
            assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]), f'lean passes differ: {sorted(kw)} K={K}'
        return outs[0][0].float().cpu(), outs[0][1].float().cpu()

    for grid in ('256', '8'):
        vtx_opts('pp_grid', grid)
        c, _ = run(bias=bd, cmap=cm)
        check(f'lean plain map K={K}', c[rows], full, 1e-2)
        untouched = torch.ones(c.shape[0], dtype=torch.bool)
        untouched[rows] = False
        assert not c[untouched].any(), 'a skipped physical row was written'
        c, _ = run(bias=bd, cmap=tmap)
        check(f'lean plain table K={K}', c[trows], full, 1e-2)
        c, _ = run(bias=bd, cmap=cm, row_scale=dev(sv), rs=(4, 1, 1, 0))
        check(f'lean scale map K={K}', c[rows], full * svr, 1e-2)
        c, _ = run(bias=bd, cmap=cm, R=Rd, rmap=cm)
        check(f'lean residual map K={K}', c[rows], full + Rq[rows], 1e-2)
        c, _ = run(bias=bd, cmap=tmap, R=Rd, rmap=tmap, row_scale=dev(sv), rs=(4, 1, 1, 0))
        check(f'lean residual+scale table K={K}', c[trows], full * svr + Rq[trows], 1e-2)
        c, _ = run(dgelu_in=dev(Hm, dtype), dgelu_kind=1)
        check(f'lean multiplier K={K}', c[:M], (q(X, dtype) @ q(W, dtype).t()) * q(Hm, dtype), 1e-2)
        pre = full.clone().requires_grad_(True)
        torch.nn.functional.gelu(pre).sum().backward()
        c, c2 = run(bias=bd, act=2)
        check(f'lean gelu K={K}', c[:M], torch.nn.functional.gelu(full), 1e-2)
        check(f"lean gelu' K={K}", c2, pre.grad, 1e-2)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N1,N2', [(1000, 256, 128), (3136, 768, 768), (500, 216, 768), (70, 8, 2304), (12552, 768, 768),
                                     (1031, 216, 768), (4099, 3072, 768), (1024, 8, 136), (4131, 1024, 1280)])
def test_gemm_tn(dtype, M, N1, N2, vtx_opts):
    from vtx import ops
    A, Bm = rnd(M, N1, seed=1), rnd(M, N2, seed=2)
    ref = q(A, dtype).t() @ q(Bm, dtype)
    for safe, nodma, tnv in ([('0', '0', 'pp256'), ('0', '0', 'ring'), ('0', '0', 'dma2'), ('0', '1', 'ring'), ('1', '1', 'ring')]
                             if dtype == torch.bfloat16 else [('0', '0', 'ring')]):
        vtx_opts('tn_safe', safe)
        vtx_opts('gemm_nodma', nodma)
        vtx_opts('gemm_tn', tnv)
        C, cs = ops.gemm_tn(dev(A, dtype), dev(Bm, dtype), M, N1, N2, want_colsum=True)
        check(f'gemm_tn {dtype} safe={safe} nodma={nodma} {tnv} {M}x{N1}x{N2}', C.cpu(), ref, 2e-3 if dtype == torch.bfloat16 else 1e-3)
        check(f'gemm_tn colsum {dtype} safe={safe} nodma={nodma} {M}x{N1}x{N2}', cs.cpu(), q(A, dtype).sum(0), 1e-3)
    vtx_opts('tn_safe', '0')
    vtx_opts('gemm_nodma', '0')
    vtx_opts('gemm_tn', 'auto')
    C0 = torch.ones(N1, N2, device=DEV)
    ops.gemm_tn(dev(A, dtype), dev(Bm, dtype), M, N1, N2, out=C0, accumulate=True)
    check(f'gemm_tn accumulate {dtype}', C0.cpu(), ref + 1, 2e-3)


@pytest.mark.parametrize('M,N1,N2', [(12552, 768, 768), (12544, 2304, 768), (150528 // 8 + 37, 768, 3072), (4096, 256, 256),
                                      (33000, 256, 512), (8 * 1569, 3072, 768)])
@pytest.mark.parametrize('cs', [False, True])
def test_gemm_tn_one_wave_per_simd_equals_ping_pong(M, N1, N2, cs, vtx_opts):
    """gemm_tn=w4 (round 5: four waves per workgroup, one per SIMD, 128 x 128 wave tiles, accumulators in AGPRs, fragment reads
    and LDS-DMA requests in the MFMA gaps) against gemm_tn=pp256: the same tile / slab partition, the same products summed in
    the same order per accumulator -- weight gradients BIT-identical; the fused bias-gradient column sums associate differently
    (four rows per thread and K tile instead of two: equal to fp32 rounding, 1e-6 of max); and against float64.
    Shapes: K-tile counts per slab odd / even / 2 (the pairs-then-odd-tile loop, the ragged last tile), token-row maps with the
    group boundary inside a K tile, 9 ... 36 output tiles."""
    from vtx import ops
    dtype = torch.bfloat16
    A, Bm = rnd(M, N1, seed=1), rnd(M, N2, seed=2)
    Ad, Bd = dev(A, dtype), dev(Bm, dtype)
    res = {}
    for v in ('pp256', 'w4', 'w4'):
        vtx_opts('gemm_tn', v)
        r = ops.gemm_tn(Ad, Bd, M, N1, N2, want_colsum=cs)
        torch.cuda.synchronize()
        res.setdefault(v, []).append(r if cs else (r, None))
    C_pp, cs_pp = res['pp256'][0]
    for C_w, cs_w in res['w4']:
        assert torch.equal(C_w, C_pp), f'w4 != pp256: max diff {(C_w - C_pp).abs().max().item():.3e}'
        if cs:
            check(f'gemm_tn w4 colsum {M}x{N1}x{N2} vs pp256', cs_w.cpu(), cs_pp.cpu(), 1e-6)
            assert torch.equal(cs_w, res['w4'][0][1]), 'run-to-run'
    if M <= 20000:
        check(f'gemm_tn w4 {M}x{N1}x{N2} vs f64', C_pp.cpu(), q(A, dtype).t() @ q(Bm, dtype), 2e-3)
    if N1 == 256:                                     # token-row maps (cls rows skipped) on both operands
        Bp, Np = 4, M // 4 - 1
        Xp, Yp = rnd(Bp, 1 + Np, N1, seed=11), rnd(Bp, 1 + Np, N2, seed=12)
        tm = ops.tokmap(Np)
        out = []
        for v in ('pp256', 'w4'):
            vtx_opts('gemm_tn', v)
            out.append(ops.gemm_tn(dev(Xp, dtype), dev(Yp, dtype), Bp * Np, N1, N2, amap=tm, bmap=tm, want_colsum=True))
        assert torch.equal(out[0][0], out[1][0])
        check(f'gemm_tn w4 rowmaps colsum {M}x{N1}x{N2} vs pp256', out[1][1].cpu(), out[0][1].cpu(), 1e-6)
        if M <= 20000:
            check(f'gemm_tn w4 rowmaps {M}x{N1}x{N2} vs f64', out[1][0].cpu(),
                  q(Xp, dtype)[:, 1:].reshape(Bp * Np, N1).t() @ q(Yp, dtype)[:, 1:].reshape(Bp * Np, N2), 2e-3)


@pytest.mark.parametrize('dtype', DTYPES)
def test_gemm_tn_rowmap_and_colsum(dtype):
    from vtx import ops
    B, N, D1, D2 = 3, 50, 128, 64
    X = rnd(B, 1 + N, D1, seed=1)
    Y = rnd(B * N, D2, seed=2)
    tm = ops.tokmap(N)
    ref = q(X, dtype)[:, 1:].reshape(B * N, D1).t() @ q(Y, dtype)
    C = ops.gemm_tn(dev(X, dtype), dev(Y, dtype), B * N, D1, D2, amap=tm)
    check(f'gemm_tn rowmap {dtype}', C.cpu(), ref, 2e-3)
    cs = ops.colsum(dev(X, dtype), B * N, D1, amap=tm)
    check(f'colsum rowmap {dtype}', cs.cpu(), q(X, dtype)[:, 1:].reshape(-1, D1).sum(0), 1e-3)
    C2, cs2 = ops.gemm_tn(dev(X, dtype), dev(Y, dtype), B * N, D1, D2, amap=tm, want_colsum=True)
    check(f'gemm_tn fused colsum {dtype}', cs2.cpu(), q(X, dtype)[:, 1:].reshape(-1, D1).sum(0), 1e-3)
    check(f'gemm_tn with fused colsum {dtype}', C2.cpu(), ref, 2e-3)
    A3, B3 = rnd(5000, 216, seed=7), rnd(5000, 768, seed=8)
    C3, cs3 = ops.gemm_tn(dev(A3, dtype), dev(B3, dtype), 5000, 216, 768, want_colsum=True)
    check(f'gemm_tn fused colsum N1 tail {dtype}', cs3.cpu(), q(A3, dtype).sum(0), 1e-3)
    check(f'gemm_tn N1 tail {dtype}', C3.cpu(), q(A3, dtype).t() @ q(B3, dtype), 2e-3)
    # token-row maps through the 256x256 ping-pong kernel (M >= 4096, N1 and N2 multiples of 256)
    Bp, Np = 4, 1570
    Xp, Yp = rnd(Bp, 1 + Np, 256, seed=11), rnd(Bp, 1 + Np, 512, seed=12)
    tmp_ = ops.tokmap(Np)
    refp = q(Xp, dtype)[:, 1:].reshape(Bp * Np, 256).t() @ q(Yp, dtype)[:, 1:].reshape(Bp * Np, 512)
    Cp, csp = ops.gemm_tn(dev(Xp, dtype), dev(Yp, dtype), Bp * Np, 256, 512, amap=tmp_, bmap=tmp_, want_colsum=True)
    check(f'gemm_tn pp rowmaps {dtype}', Cp.cpu(), refp, 2e-3)
    check(f'gemm_tn pp rowmaps colsum {dtype}', csp.cpu(), q(Xp, dtype)[:, 1:].reshape(-1, 256).sum(0), 1e-3)
    big = rnd(5000, 2304, seed=3)
    cs = ops.colsum(dev(big, dtype), 5000, 2304)
    check(f'colsum big {dtype}', cs.cpu(), q(big, dtype).sum(0), 1e-3)


# ---------------------------------------------------------------------------- attention
def _attn_ref(qkv, heads):
    Bn, L, D3 = qkv.shape
    D = D3 // 3
    hd = D // heads
    t = qkv.reshape(Bn, L, 3, heads, hd)
    qq, kk, vv = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    p = torch.softmax(qq @ kk.transpose(-1, -2) * hd ** -0.5, dim=-1)
    return (p @ vv).permute(0, 2, 1, 3).reshape(Bn, L, D), p


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('S,L', [(37, 8), (5, 9), (3, 17), (2, 64), (2, 65), (2, 197), (1, 300)])
def test_attention_contig(dtype, S, L):
    from vtx import ops
    from vtx._lib import ATTN_CONTIG
    H, hd = 2, 64
    D = H * hd
    qkv = rnd(S, L, 3 * D, seed=L) * 1.5
    do = rnd(S, L, D, seed=L + 1)
    qq = q(qkv, dtype).requires_grad_(True)
    ref, p_ref = _attn_ref(qq, H)
    ref.backward(q(do, dtype))
    qd = dev(qkv, dtype)
    o = torch.empty(S, L, D, dtype=dtype, device=DEV)
    lse = torch.empty(S * H * L, device=DEV)
    probs = torch.empty(S, H, L, L, device=DEV)
    ops.attn_fwd(qd, o, lse, ATTN_CONTIG, S, L, H, hd, hd ** -0.5, probs=probs)
    check(f'attn fwd {dtype} S={S} L={L}', o.float().cpu(), ref.detach(), TOL[dtype])
    check(f'attn probs {dtype} S={S} L={L}', probs.cpu(), p_ref.detach(), 1e-3)
    dqkv = torch.full((S, L, 3 * D), float('nan'), dtype=dtype, device=DEV)
    ops.attn_bwd(qd, o, lse, dev(do, dtype), dqkv, ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
    check(f'attn bwd {dtype} S={S} L={L}', dqkv.float().cpu(), qq.grad, 2 * TOL[dtype])


@pytest.mark.parametrize('S,L,H', [(401, 8, 12), (77, 9, 12), (50, 8, 5), (9, 4, 20)])
def test_attention_short_sequences_workgroup_layouts(S, L, H, vtx_opts):
    """Short-sequence (temporal) attention: one head and four row tiles per workgroup (attn_hw_* = 0) or n heads of one
    row tile per workgroup -- same arithmetic per (tile, head): bit-identical outputs, log-sum-exp and gradients, for
    head counts that do and do not divide into the groups, ragged last tiles, and against the float64 reference."""
    from vtx import ops
    from vtx._lib import ATTN_CONTIG
    dtype = torch.bfloat16
    hd = 64
    D = H * hd
    qkv = rnd(S, L, 3 * D, seed=L) * 1.5
    do = rnd(S, L, D, seed=L + 1)
    qq = q(qkv, dtype).requires_grad_(True)
    ref, _ = _attn_ref(qq, H)
    ref.backward(q(do, dtype))
    qd, dd = dev(qkv, dtype), dev(do, dtype)
    res = []
    for n in ('0', '3', '16'):
        vtx_opts('attn_hw_fwd', n)
        vtx_opts('attn_hw_bwd', n)
        o = torch.full((S, L, D), float('nan'), dtype=dtype, device=DEV)
        lse = torch.full((S * H * L,), float('nan'), device=DEV)
        ops.attn_fwd(qd, o, lse, ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
        dqkv = torch.full((S, L, 3 * D), float('nan'), dtype=dtype, device=DEV)
        ops.attn_bwd(qd, o, lse, dd, dqkv, ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
        res.append((o, lse, dqkv))
    check(f'attn fwd S={S} L={L} H={H}', res[0][0].float().cpu(), ref.detach(), TOL[dtype])
    check(f'attn bwd S={S} L={L} H={H}', res[0][2].float().cpu(), qq.grad, 2 * TOL[dtype])
    for r in res[1:]:
        for a, b_ in zip(r, res[0]):
            assert torch.equal(a, b_)


def test_attention_short_sequences_more_than_65535_row_tiles(vtx_opts):
    """17..32-token sequences occupy a 32-row tile each: 66 000 sequences = 66 000 tiles.  The heads-per-workgroup layouts used
    to put the tile index on grid.y (limit 65 535: the launch failed); (head group, tile) now share blockIdx.x.
    Same arithmetic per (tile, head) as the one-head layout: bit-identical outputs, log-sum-exp and gradients."""
    from vtx import ops
    from vtx._lib import ATTN_CONTIG
    S, L, H, hd = 66000, 17, 2, 64
    D = H * hd
    bf = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = (torch.randn(S, L, 3 * D, generator=g, device=DEV) * 1.5).to(bf)
    do = torch.randn(S, L, D, generator=g, device=DEV).to(bf)
    res = []
    for n in ('0', '4'):
        vtx_opts('attn_hw_fwd', n)
        vtx_opts('attn_hw_bwd', n)
        o = torch.full((S, L, D), float('nan'), dtype=bf, device=DEV)
        lse = torch.full((S * H * L,), float('nan'), device=DEV)
        ops.attn_fwd(qkv, o, lse, ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
        dqkv = torch.full((S, L, 3 * D), float('nan'), dtype=bf, device=DEV)
        ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
        torch.cuda.synchronize()
        res.append((o, lse, dqkv))
    assert all(torch.isfinite(t.float()).all() for t in res[0])
    for a, b_ in zip(res[1], res[0]):
        assert torch.equal(a, b_)
    # the last sequence against float64 (the tile beyond 65 535)
    qq = qkv[-1:].double().cpu().requires_grad_(True)
    ref, _ = _attn_ref(qq, H)
    ref.backward(do[-1:].double().cpu())
    check('attn fwd tile 65999', res[1][0][-1:].float().cpu(), ref.detach(), TOL[bf])
    check('attn bwd tile 65999', res[1][2][-1:].float().cpu(), qq.grad, 2 * TOL[bf])


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,T,P', [(2, 4, 9), (2, 3, 36), (1, 2, 196)])
def test_attention_space_mode(dtype, B, T, P):
    """Divided spatial attention addressing: natural-order qkv in, [tokens | per-frame cls] out.
    P+1 <= 32 runs the VALU kernels, 33..256 the MFMA kernels (bf16)."""
    from vtx import ops
    from vtx._lib import ATTN_SPACE
    H, hd = 2, 64
    D = H * hd
    N = P * T
    qkv = rnd(B, 1 + N, 3 * D, seed=1) * 1.5
    do_tok = rnd(B, N, D, seed=2)
    do_cls = rnd(B * T, D, seed=3)
    qq = q(qkv, dtype).requires_grad_(True)
    tok = qq[:, 1:].reshape(B, P, T, 3 * D).permute(0, 2, 1, 3).reshape(B * T, P, 3 * D)
    cls = qq[:, :1].expand(B, T, 3 * D).reshape(B * T, 1, 3 * D)
    ref, _ = _attn_ref(torch.cat([cls, tok], 1), H)                    # [(b t), 1+P, D]
    ref_tok = ref[:, 1:].reshape(B, T, P, D).permute(0, 2, 1, 3).reshape(B, N, D)
    ref_cls = ref[:, 0]
    (ref_tok * q(do_tok, dtype)).sum().backward(retain_graph=True)
    (ref_cls * q(do_cls, dtype)).sum().backward()
    qd = dev(qkv, dtype)
    o = torch.empty(B * N + B * T, D, dtype=dtype, device=DEV)
    lse = torch.empty(B * T * H * (P + 1), device=DEV)
    ops.attn_fwd(qd, o, lse, ATTN_SPACE, B * T, P + 1, H, hd, hd ** -0.5, B, T, P)
    check(f'attn space tokens {dtype}', o[:B * N].float().cpu().reshape(B, N, D), ref_tok.detach(), TOL[dtype])
    check(f'attn space cls {dtype}', o[B * N:].float().cpu(), ref_cls.detach(), TOL[dtype])
    do = torch.cat([do_tok.reshape(B * N, D), do_cls], 0)
    dqkv = torch.zeros(B, 1 + N, 3 * D, dtype=dtype, device=DEV)
    dqkv_cls = torch.empty(B * T, 3 * D, dtype=dtype, device=DEV)
    ops.attn_bwd(qd, o, lse, dev(do, dtype), dqkv, ATTN_SPACE, B * T, P + 1, H, hd, hd ** -0.5, B, T, P, dqkv_cls=dqkv_cls)
    ops.cls_qkv_reduce(dqkv_cls, dqkv, B, T, 3 * D, 1 + N)
    check(f'attn space bwd {dtype}', dqkv.float().cpu(), qq.grad, 2 * TOL[dtype])


@pytest.mark.parametrize('mode,S,L,H', [('contig', 70, 197, 12), ('space', 0, 197, 12)])
def test_attention_streamed_kernels_vs_float64_at_bench_scale(mode, S, L, H):
    """The DEFAULT bf16 kernels of 193..224-token attention (attn_fwd_stream_mfma_kernel<7>, attn_bwd_stream_mfma_kernel<7>)
    against the float64 restatement of reference transformer.py:165-177 (spatial form: :352-377) with SEVERAL items per
    persistent workgroup: 840 / 576 (sequence, head) items on 256 CUs -- K / V double buffer and Q / dO ring refills, both
    parities of the lse / delta buffers, the rolling delta schedule across the item boundary.  Checked: outputs, the saved
    log-sum-exp, dqkv and (spatial form) the per-frame cls gradient rows -- the other tests of these kernels at this item count
    compare with the older kernels only."""
    from vtx import ops
    from vtx._lib import ATTN_CONTIG, ATTN_SPACE
    dtype, hd = torch.bfloat16, 64
    D = H * hd
    scale = hd ** -0.5

    def ref_lse(qq3):                                                   # [S, L, 3D] float64 -> [S, H, L]
        t = qq3.reshape(qq3.shape[0], L, 3, H, hd)
        qh, kh = t[:, :, 0].permute(0, 2, 1, 3), t[:, :, 1].permute(0, 2, 1, 3)
        return torch.logsumexp(qh @ kh.transpose(-1, -2) * scale, dim=-1)

    if mode == 'contig':
        qkv = rnd(S, L, 3 * D, seed=L) * 1.5
        do = rnd(S, L, D, seed=L + 1)
        qq = q(qkv, dtype).requires_grad_(True)
        ref, _ = _attn_ref(qq, H)
        ref.backward(q(do, dtype))
        lse_ref = ref_lse(qq.detach())
        qd = dev(qkv, dtype)
        o = torch.full((S, L, D), float('nan'), dtype=dtype, device=DEV)
        lse = torch.full((S * H * L,), float('nan'), device=DEV)
        ops.attn_fwd(qd, o, lse, ATTN_CONTIG, S, L, H, hd, scale)
        dqkv = torch.full((S, L, 3 * D), float('nan'), dtype=dtype, device=DEV)
        ops.attn_bwd(qd, o, lse, dev(do, dtype), dqkv, ATTN_CONTIG, S, L, H, hd, scale)
        check(f'attn streamed fwd contig {S}x{H}x{L} vs f64', o.float().cpu(), ref.detach(), TOL[dtype])
        check(f'attn streamed lse contig {S}x{H}x{L} vs f64', lse.cpu().reshape(S, H, L), lse_ref, 1e-4)
        check(f'attn streamed bwd contig {S}x{H}x{L} vs f64', dqkv.float().cpu(), qq.grad, 2 * TOL[dtype])
        return
    B, T, P = 6, 8, L - 1
    N = P * T
    qkv = rnd(B, 1 + N, 3 * D, seed=1) * 1.5
    do_tok = rnd(B, N, D, seed=2)
    do_cls = rnd(B * T, D, seed=3)
    qq = q(qkv, dtype).requires_grad_(True)
    tok = qq[:, 1:].reshape(B, P, T, 3 * D).permute(0, 2, 1, 3).reshape(B * T, P, 3 * D)
    cls = qq[:, :1].expand(B, T, 3 * D).reshape(B * T, 1, 3 * D)
    seqs = torch.cat([cls, tok], 1)                                     # [(b t), 1+P, 3D]
    ref, _ = _attn_ref(seqs, H)
    ref_tok = ref[:, 1:].reshape(B, T, P, D).permute(0, 2, 1, 3).reshape(B, N, D)
    ref_cls = ref[:, 0]
    (ref_tok * q(do_tok, dtype)).sum().backward(retain_graph=True)
    (ref_cls * q(do_cls, dtype)).sum().backward()
    lse_ref = ref_lse(seqs.detach())
    qd = dev(qkv, dtype)
    o = torch.full((B * N + B * T, D), float('nan'), dtype=dtype, device=DEV)
    lse = torch.full((B * T * H * L,), float('nan'), device=DEV)
    ops.attn_fwd(qd, o, lse, ATTN_SPACE, B * T, L, H, hd, scale, B, T, P)
    check(f'attn streamed fwd space tokens {B}x{T}x{P} H={H} vs f64', o[:B * N].float().cpu().reshape(B, N, D), ref_tok.detach(), TOL[dtype])
    check(f'attn streamed fwd space cls {B}x{T}x{P} H={H} vs f64', o[B * N:].float().cpu(), ref_cls.detach(), TOL[dtype])
    check(f'attn streamed lse space {B}x{T}x{P} H={H} vs f64', lse.cpu().reshape(B * T, H, L), lse_ref, 1e-4)
    dout = torch.cat([do_tok.reshape(B * N, D), do_cls], 0)
    dqkv = torch.zeros(B, 1 + N, 3 * D, dtype=dtype, device=DEV)
    dqkv_cls = torch.full((B * T, 3 * D), float('nan'), dtype=dtype, device=DEV)
    ops.attn_bwd(qd, o, lse, dev(dout, dtype), dqkv, ATTN_SPACE, B * T, L, H, hd, scale, B, T, P, dqkv_cls=dqkv_cls)
    # per-frame cls rows: d(loss)/d(the cls copy of frame (b, t)) -- the reference's replicated cls token, transformer.py:354-356
    seqs2 = seqs.detach().clone().requires_grad_(True)
    r2, _ = _attn_ref(seqs2, H)
    r2_tok = r2[:, 1:].reshape(B, T, P, D).permute(0, 2, 1, 3).reshape(B, N, D)
    ((r2_tok * q(do_tok, dtype)).sum() + (r2[:, 0] * q(do_cls, dtype)).sum()).backward()
    check(f'attn streamed bwd space cls rows {B}x{T}x{P} H={H} vs f64', dqkv_cls.float().cpu(), seqs2.grad[:, 0], 2 * TOL[dtype])
    ops.cls_qkv_reduce(dqkv_cls, dqkv, B, T, 3 * D, 1 + N)
    check(f'attn streamed bwd space {B}x{T}x{P} H={H} vs f64', dqkv.float().cpu(), qq.grad, 2 * TOL[dtype])


@pytest.mark.parametrize('S,L', [(3, 197), (37, 8), (5, 9), (7, 16), (4, 32), (3, 33), (1, 256), (9, 1), (6, 5)])
def test_attention_mfma_matches_valu(S, L, vtx_opts):
    """The bf16 MFMA kernels (packed short sequences / 33..256 tokens) and the VALU kernels are
    two implementations of the same op."""
    from vtx import ops
    from vtx._lib import ATTN_CONTIG
    H, hd = 3, 64
    D = H * hd
    qkv = dev(rnd(S, L, 3 * D, seed=5) * 1.5, torch.bfloat16)
    do = dev(rnd(S, L, D, seed=6), torch.bfloat16)
    res = {}
    for mode in ('1', '0'):
        vtx_opts('attn_valu', mode)
        o = torch.empty(S, L, D, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(S * H * L, device=DEV)
        ops.attn_fwd(qkv, o, lse, ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
        dqkv = torch.full((S, L, 3 * D), float('nan'), dtype=torch.bfloat16, device=DEV)
        ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
        res[mode] = (o.float().cpu(), lse.cpu(), dqkv.float().cpu())
    check(f'attn mfma vs valu out S={S} L={L}', res['0'][0], res['1'][0], 1e-2)
    check(f'attn mfma vs valu lse S={S} L={L}', res['0'][1], res['1'][1], 1e-4)
    check(f'attn mfma vs valu dqkv S={S} L={L}', res['0'][2], res['1'][2], 2e-2)


@pytest.mark.parametrize('mode,S,L,H', [('contig', 3, 197, 3), ('contig', 2, 224, 2), ('contig', 1, 193, 1), ('contig', 70, 197, 12),
                                        ('space', 0, 197, 3), ('space', 0, 211, 12)])
def test_attention_forward_streamed_equals_workgroup_per_item(mode, S, L, H, vtx_opts):
    """attn_fwd_stream=1 (193..224 tokens): one persistent workgroup per CU, a wave per query tile, K / V of the next item brought
    in by LDS-DMA while the current one is computed.  The tile body is the other kernel's: outputs and lse bit-identical; 840 items =
    several per workgroup (double buffer, padded key rows zero-filled by the range check)."""
    from vtx import ops
    from vtx._lib import ATTN_CONTIG, ATTN_SPACE
    hd = 64
    D = H * hd
    bf = torch.bfloat16
    if mode == 'contig':
        qkv = dev(rnd(S, L, 3 * D, seed=L) * 1.5, bf)
        args = (ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
        new_out = lambda: torch.full((S, L, D), float('nan'), dtype=bf, device=DEV)       # noqa: E731
        nlse = S * H * L
    else:
        B, T, P = (2, 3, L - 1) if H == 3 else (6, 8, L - 1)
        N = P * T
        qkv = dev(rnd(B, 1 + N, 3 * D, seed=L) * 1.5, bf)
        args = (ATTN_SPACE, B * T, L, H, hd, hd ** -0.5, B, T, P)
        new_out = lambda: torch.full((B * N + B * T, D), float('nan'), dtype=bf, device=DEV)   # noqa: E731
        nlse = B * T * H * L
    res = []
    for st in ('0', '1', '1'):
        vtx_opts('attn_fwd_stream', st)
        o = new_out()
        lse = torch.full((nlse,), float('nan'), device=DEV)
        ops.attn_fwd(qkv, o, lse, *args)
        torch.cuda.synchronize()
        res.append((o, lse))
    assert torch.isfinite(res[0][0].float()).all() and torch.isfinite(res[0][1]).all()
    for o, lse in res[1:]:
        assert torch.equal(o, res[0][0])
        assert torch.equal(lse, res[0][1])


@pytest.mark.parametrize('mode,S,L,H', [('contig', 3, 197, 3), ('contig', 2, 224, 2), ('contig', 1, 193, 1), ('contig', 70, 197, 12),
                                        ('space', 0, 197, 3), ('space', 0, 211, 12)])
def test_attention_backward_streamed_one_phase(mode, S, L, H, vtx_opts):
    """attn_fused=2 (193..224 tokens): one phase per (sequence, head) -- a wave owns a key tile, dS goes through LDS once for
    the dq product, the seven partial dq tiles are summed in fixed order.  dk / dv: the same products in the same order as the
    other kernels (bit-identical); dq: equal up to the fp32 rounding of a different summation order (then one bf16 rounding);
    run-to-run bit-reproducible; 840 items = several per workgroup (ring refill, parity of the lse / delta buffers)."""
    from vtx import ops
    from vtx._lib import ATTN_CONTIG, ATTN_SPACE
    hd = 64
    D = H * hd
    bf = torch.bfloat16
    if mode == 'contig':
        qkv = dev(rnd(S, L, 3 * D, seed=L) * 1.5, bf)
        do = dev(rnd(S, L, D, seed=L + 1), bf)
        args = (ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
        new_out = lambda: torch.empty(S, L, D, dtype=bf, device=DEV)                      # noqa: E731
        nlse = S * H * L
        new_dqkv = lambda: (torch.full((S, L, 3 * D), float('nan'), dtype=bf, device=DEV), None)   # noqa: E731
    else:
        B, T, P = (2, 3, L - 1) if H == 3 else (6, 8, L - 1)
        N = P * T
        qkv = dev(rnd(B, 1 + N, 3 * D, seed=L) * 1.5, bf)
        do = dev(rnd(B * N + B * T, D, seed=L + 1), bf)
        args = (ATTN_SPACE, B * T, L, H, hd, hd ** -0.5, B, T, P)
        new_out = lambda: torch.empty(B * N + B * T, D, dtype=bf, device=DEV)             # noqa: E731
        nlse = B * T * H * L
        new_dqkv = lambda: (torch.zeros(B, 1 + N, 3 * D, dtype=bf, device=DEV),           # noqa: E731
                            torch.full((B * T, 3 * D), float('nan'), dtype=bf, device=DEV))
    o = new_out()
    lse = torch.empty(nlse, device=DEV)
    ops.attn_fwd(qkv, o, lse, *args)
    res = []
    for fused in ('1', '2', '2'):
        vtx_opts('attn_fused', fused)
        dqkv, dcls = new_dqkv()
        if dcls is None:
            ops.attn_bwd(qkv, o, lse, do, dqkv, *args)
        else:
            ops.attn_bwd(qkv, o, lse, do, dqkv, *args, dqkv_cls=dcls)
        torch.cuda.synchronize()
        res.append((dqkv, dcls))
    (ref, rcls), (got, gcls), (again, acls) = res
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, again) and (gcls is None or torch.equal(gcls, acls))
    for a, b in ((got, ref),) + (((gcls, rcls),) if gcls is not None else ()):
        a3, b3 = a.view(-1, 3, D), b.view(-1, 3, D)
        assert torch.equal(a3[:, 1:], b3[:, 1:]), 'dk / dv must be bit-identical'
        dq, rq = a3[:, 0].float(), b3[:, 0].float()
        # one bf16 ulp where the fp32 sums round differently: |diff| <= 2^-7 |ref| (+ tiny absolute), on few elements
        diff = (dq - rq).abs()
        assert (diff <= rq.abs() * 2 ** -7 + 1e-6).all(), float(diff.max())
        assert (diff > 0).float().mean() < 0.05


@pytest.mark.parametrize('mode,S,L,H', [('contig', 3, 197, 3), ('contig', 5, 33, 2), ('contig', 2, 224, 2), ('contig', 4, 130, 12),
                                        ('space', 0, 197, 3), ('space', 0, 37, 2), ('contig', 2, 256, 2)])
def test_attention_backward_one_pass_equals_two_kernels(mode, S, L, H, vtx_opts):
    """Backward of the 33..224-token attention: the one-pass kernel (dq, dk, dv from tiles that stay in LDS) and the dq + dk/dv
    kernel pair (all four unrolled / rolled variants) do the same products and sums in the same order: bit-identical
    gradients, ragged last tiles, both row addressings; 225..256 tokens only have the pair."""
    from vtx import ops
    from vtx._lib import ATTN_CONTIG, ATTN_SPACE
    hd = 64
    D = H * hd
    bf = torch.bfloat16
    if mode == 'contig':
        qkv = dev(rnd(S, L, 3 * D, seed=L) * 1.5, bf)
        do = dev(rnd(S, L, D, seed=L + 1), bf)
        args = (ATTN_CONTIG, S, L, H, hd, hd ** -0.5)
        new_out = lambda: torch.empty(S, L, D, dtype=bf, device=DEV)                      # noqa: E731
        nlse = S * H * L
        new_dqkv = lambda: (torch.full((S, L, 3 * D), float('nan'), dtype=bf, device=DEV), None)   # noqa: E731
    else:
        B, T, P = 2, 3, L - 1
        N = P * T
        qkv = dev(rnd(B, 1 + N, 3 * D, seed=L) * 1.5, bf)
        do = dev(rnd(B * N + B * T, D, seed=L + 1), bf)
        args = (ATTN_SPACE, B * T, L, H, hd, hd ** -0.5, B, T, P)
        new_out = lambda: torch.empty(B * N + B * T, D, dtype=bf, device=DEV)             # noqa: E731
        nlse = B * T * H * L
        new_dqkv = lambda: (torch.zeros(B, 1 + N, 3 * D, dtype=bf, device=DEV),           # noqa: E731
                            torch.full((B * T, 3 * D), float('nan'), dtype=bf, device=DEV))
    o = new_out()
    lse = torch.empty(nlse, device=DEV)
    ops.attn_fwd(qkv, o, lse, *args)
    res = []
    for fused, dkv in (('1', '3'), ('0', '3'), ('0', '0'), ('0', '1')):
        vtx_opts('attn_fused', fused)
        vtx_opts('attn_dkv', dkv)
        dqkv, dcls = new_dqkv()
        if dcls is None:
            ops.attn_bwd(qkv, o, lse, do, dqkv, *args)
        else:
            ops.attn_bwd(qkv, o, lse, do, dqkv, *args, dqkv_cls=dcls)
        torch.cuda.synchronize()
        res.append((dqkv, dcls))
    assert torch.isfinite(res[0][0].float()).all()
    for dqkv, dcls in res[1:]:
        assert torch.equal(dqkv, res[0][0])
        if dcls is not None:
            assert torch.equal(dcls, res[0][1])


# ----------------------------------------------------------------------------- glue ops
@pytest.mark.parametrize('dtype', DTYPES)
def test_glue_ops(dtype):
    from vtx import ops
    B, T, P, D = 2, 4, 5, 64
    N = P * T
    a_cls, x = rnd(B * T, D, seed=1), rnd(B, 1 + N, D, seed=2)
    out = torch.zeros(B, 1 + N, D, dtype=dtype, device=DEV)
    ops.cls_mean_fwd(dev(a_cls, dtype), dev(x, dtype), out, B, T, D, 1 + N)
    ref = q(x, dtype)[:, 0] + q(a_cls, dtype).reshape(B, T, D).mean(1)
    check(f'cls_mean {dtype}', out[:, 0].float().cpu(), ref, TOL[dtype])
    s = torch.rand(B * T, generator=torch.Generator().manual_seed(3))
    dout = rnd(B, 1 + N, D, seed=4)
    da = torch.empty(B * N + B * T, D, dtype=dtype, device=DEV)
    ops.space_grad_prep(dev(dout, dtype), dev(s), da, B, T, P, D)
    dq = q(dout, dtype)
    n = torch.arange(N)
    ref_tok = dq[:, 1:] * s.double().reshape(B, T)[:, n % T][:, :, None]
    ref_cls = (dq[:, :1] * s.double().reshape(B, T, 1) / T).reshape(B * T, D)
    check(f'space_grad_prep tok {dtype}', da[:B * N].float().cpu().reshape(B, N, D), ref_tok, TOL[dtype])
    check(f'space_grad_prep cls {dtype}', da[B * N:].float().cpu(), ref_cls, TOL[dtype])
    # strided row reductions
    r = ops.reduce_rows(dev(dout, dtype), N, B, D, D, 1, 1 + N, 1)
    check(f'reduce_rows batch {dtype}', r.cpu(), dq[:, 1:].sum(0), 1e-3)
    r2 = ops.reduce_rows(r, P, T, D, D, 0, 1, T)
    check(f'reduce_rows pos {dtype}', r2.cpu(), dq[:, 1:].sum(0).reshape(P, T, D).sum(1), 1e-3)
    r3 = ops.reduce_rows(r, T, P, D, D, 0, T, 1)
    check(f'reduce_rows time {dtype}', r3.cpu(), dq[:, 1:].sum(0).reshape(P, T, D).sum(0), 1e-3)
    # weight staging
    W = rnd(96, 200, seed=5)
    wc, wt = ops.cast_transpose(dev(W), dtype)
    check(f'cast {dtype}', wc.float().cpu(), q(W, dtype), 1e-6)
    check(f'cast_transpose {dtype}', wt.float().cpu(), q(W, dtype).t(), 1e-6)
    # row broadcast / cls-row copy
    src = rnd(D, seed=6)
    dst = torch.zeros(B, 1 + N, D, dtype=dtype, device=DEV)
    ops.row_scale_copy(dev(src, dtype), dst, B, D, smap=ops.rowmap(1, -1, 0), dmap=ops.clsmap(N))
    check(f'cls row broadcast {dtype}', dst[:, 0].float().cpu(), q(src, dtype)[None].expand(B, D), 1e-6)
    assert dst[:, 1:].abs().max().item() == 0


@pytest.mark.parametrize('dtype', DTYPES)
def test_dropped_rows_kernels(dtype):
    """vtx_dropped_rows_fix / _colsum: the fix-ups behind the merged attn.proj + temporal_fc GEMM act on exactly the
    rows of the groups whose DropPath scale is 0 (token rows through the row map, cls rows untouched)."""
    from vtx import ops
    B, T, P, D = 3, 4, 7, 200
    N = P * T
    M = B * N
    tm = ops.tokmap(N)
    g = torch.Generator().manual_seed(5)
    s = (torch.rand(M // T, generator=g) > 0.3).float() * 1.25
    assert 0 < (s == 0).sum() < s.numel()
    x, out0, o0, bias = rnd(B, 1 + N, D, seed=1), rnd(B, 1 + N, D, seed=2), rnd(M, D, seed=3), rnd(D, seed=4)
    out, o = dev(out0, dtype), dev(o0, dtype)
    ops.dropped_rows_fix(dev(s), M, D, T, x=dev(x, dtype), xmap=tm, bias=dev(bias), out=out, omap=tm, zero=o)
    drop = (s == 0).repeat_interleave(T).reshape(B, N)
    ref_out = q(out0, dtype).clone()
    ref_out[:, 1:][drop] = (q(x, dtype)[:, 1:] + bias.double())[drop]
    ref_o = q(o0, dtype).clone()
    ref_o[drop.reshape(M)] = 0
    check(f'dropped_rows_fix out {dtype}', out.float().cpu(), ref_out, TOL[dtype])
    assert torch.equal(out[:, 0].cpu(), out0.to(dtype)[:, 0])                     # cls rows untouched
    assert torch.equal(out[:, 1:].cpu()[~drop], out0.to(dtype)[:, 1:][~drop])     # kept rows untouched
    assert torch.equal(o.float().cpu().double(), ref_o)
    for nparts in (1, 5, 32):
        part = ops.dropped_rows_colsum(dev(x, dtype), dev(s), M, D, T, smap=tm, nparts=nparts)
        assert tuple(part.shape) == (nparts, D)
        check(f'dropped_rows_colsum {dtype} nparts={nparts}', part.sum(0).cpu(), q(x, dtype)[:, 1:][drop].sum(0), 1e-4)
    # ragged last group, nothing dropped / everything dropped
    M2 = M - 2
    for sv in (torch.ones(M // T), torch.zeros(M // T)):
        part = ops.dropped_rows_colsum(dev(o0, dtype), dev(sv), M2, D, T, nparts=3)
        ref = q(o0, dtype)[:M2].sum(0) if sv[0] == 0 else torch.zeros(D, dtype=torch.float64)
        check(f'dropped_rows_colsum ragged {dtype}', part.sum(0).cpu(), ref, 1e-4)


@pytest.mark.parametrize('ts', [1, 2])
def test_patch_rows_bit_exact(ts):
    """Patch / tubelet indexing must be bit-exact (fp32 path: pure data movement)."""
    from oracle import vt_oracle as O
    from vtx import ops
    B, T, C, H, W, ps = 2, 4, 3, 64, 48, 16
    x = rnd(B, T, C, H, W, seed=1)
    want = (O.patch_rows_2d(x, ps) if ts == 1 else O.patch_rows_3d(x, ps, ts))      # [(b t'), P, K]
    Tq, P = T // ts, (H // ps) * (W // ps)
    got = ops.patch_rows(dev(x), torch.float32, ps, ts, frame_major=True).cpu()
    assert torch.equal(got, want.reshape(B * Tq * P, -1)), 'frame-major patch rows differ'
    got = ops.patch_rows(dev(x), torch.float32, ps, ts, frame_major=False).cpu()
    want_pt = want.reshape(B, Tq, P, -1).permute(0, 2, 1, 3).reshape(B * P * Tq, -1)
    assert torch.equal(got, want_pt), 'token-order patch rows differ'
    report(f'ok   patch_rows bit-exact ts={ts}')


@pytest.mark.parametrize('ts', [1, 2])
def test_patch_rows_uint8_ingest_bit_exact(ts):
    """uint8 [B,T,H,W,3] clip -> normalised patch rows in one pass: bit-identical (fp32) to the reference's
    own order of operations -- permute to [T,C,H,W] (dataset.py:171), ToTensor = x.float().div(255)
    (data_transform.py:52-63), transforms.Normalize = sub(mean).div(std) (:534-539) -- followed by the
    fp32 patch gather; every uint8 value occurs."""
    from oracle import vt_oracle as O
    import vtx
    from vtx import ops
    B, T, H, W, ps = 2, 4, 64, 48, 16
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (B, T, H, W, 3), generator=g, dtype=torch.uint8)
    u8[0, 0, 0, :, 0] = torch.arange(48, dtype=torch.uint8) * 5
    u8[0, 0, 1].view(-1)[:144] = torch.arange(144, dtype=torch.uint8)
    u8[0, 0, 2].view(-1)[:112] = torch.arange(144, 256, dtype=torch.uint8)
    mean, std = [0.45, 0.456, 0.406], [0.225, 0.224, 0.229]
    x = u8.permute(0, 1, 4, 2, 3).float().div(255)                                  # ToTensor
    x = x.sub(torch.tensor(mean).view(1, 1, 3, 1, 1)).div(torch.tensor(std).view(1, 1, 3, 1, 1))   # Normalize
    want = (O.patch_rows_2d(x, ps) if ts == 1 else O.patch_rows_3d(x, ps, ts))
    Tq, P = T // ts, (H // ps) * (W // ps)
    vtx.set_input_normalization(mean, std)
    try:
        got = ops.patch_rows(u8.to(DEV), torch.float32, ps, ts, frame_major=True).cpu()
        assert torch.equal(got, want.reshape(B * Tq * P, -1)), 'uint8 ingest: frame-major rows differ'
        got = ops.patch_rows(u8.to(DEV), torch.float32, ps, ts, frame_major=False).cpu()
        want_pt = want.reshape(B, Tq, P, -1).permute(0, 2, 1, 3).reshape(B * P * Tq, -1)
        assert torch.equal(got, want_pt), 'uint8 ingest: token-order rows differ'
        got16 = ops.patch_rows(u8.to(DEV), torch.bfloat16, ps, ts, frame_major=False).cpu()
        assert torch.equal(got16, want_pt.to(torch.bfloat16)), 'uint8 ingest: bf16 rows differ from the rounded fp32 rows'
    finally:
        vtx.set_input_normalization(None, None)
    with pytest.raises(ValueError):
        ops.patch_rows(u8.to(DEV), torch.float32, ps, ts, frame_major=True)           # no normalisation set
    report(f'ok   patch_rows uint8 ingest bit-exact ts={ts}')


# ---------------------------------------------------------------------------------- HOG
def _hog_frames():
    fr = [np.random.RandomState(s).randint(0, 256, (224, 224, 3)).astype(np.uint8) for s in (1234, 7)]
    yy, xx = np.mgrid[0:224, 0:224]
    fr.append(np.stack([(yy * 255 // 223), (xx * 255 // 223), ((xx * 3 + yy * 5) // 8 % 256)], -1).astype(np.uint8))
    return np.stack(fr)


def test_hog_bit_exact_vs_skimage_golden():
    from oracle import hog_oracle as H
    from vtx import ops
    frames = _hog_frames()
    feats, bins = ops.hog_fwd(torch.from_numpy(frames).to(DEV), want_bins=True)
    feats, bins = feats.cpu().numpy(), bins.cpu().numpy()
    ref = gold('hog_skimage.npz')['feats']
    for i in range(3):
        assert np.array_equal(bins[i], H.hog_bin_map(frames[i])), f'frame {i}: orientation bins differ'
        nbad = int((feats[i] != ref[i]).sum())
        report(f'hog frame {i}: {nbad} of {ref[i].size} feature values differ, max abs {np.abs(feats[i] - ref[i]).max():.3e}')
        assert nbad == 0, f'frame {i}: HOG features not bit-exact ({nbad} differ)'


def test_hog_other_sizes_and_edges():
    from oracle import hog_oracle as H
    from vtx import ops
    rs = np.random.RandomState(3)
    for shape in [(32, 48), (16, 16), (224, 224), (32, 272)]:      # W > 256: the one-channel-at-a-time kernel
        fr = rs.randint(0, 256, (2,) + shape + (3,)).astype(np.uint8)
        fr[1] = 0 if shape != (224, 224) else 255                    # constant frame -> all zeros
        got = ops.hog_fwd(torch.from_numpy(fr).to(DEV)).cpu().numpy()
        for i in range(2):
            want = H.extract_hog_features(fr[i]) if shape == (224, 224) else _hog_any(fr[i])
            assert np.array_equal(got[i], want), (shape, i)
    assert ops.hog_fwd(torch.zeros(0, 32, 32, 3, dtype=torch.uint8, device=DEV)).shape == (0, 2, 2, 108)
    report('ok   hog sizes/edges bit-exact')


def _hog_any(img):
    """oracle for sizes other than 224 (extract_hog_features hard-codes ph=pw=14 like the reference)."""
    from oracle import hog_oracle as H
    per = [H.hog_channel(img[:, :, c]) for c in range(3)]
    f = np.concatenate(per, axis=-1)
    nr, nc, k = f.shape
    return f.reshape(nr // 2, 2, nc // 2, 2, k).transpose(0, 2, 1, 3, 4).reshape(nr // 2, nc // 2, 4 * k)


# ------------------------------------------------------------------------ MaskFeat head
@pytest.mark.parametrize('dtype', DTYPES)
def test_maskfeat_kernels(dtype):
    import vtx
    from oracle import vt_oracle as O
    from vtx import ops
    B, Tq, g, r, C = 2, 4, 3, 2, 32
    Hq = g * r
    L = Tq * Hq * Hq
    x = rnd(B, L, C, seed=1)
    mask = (torch.rand(B, Tq, g, g, generator=torch.Generator().manual_seed(2)) > 0.5).to(torch.int32)
    tok = rnd(1, 1, C, seed=3)
    xq = q(x, dtype).requires_grad_(True)
    tq_ = tok.double().requires_grad_(True)
    ref = O.maskfeat_blend(xq, mask, tq_, r)
    dy = rnd(B, L, C, seed=4)
    ref.backward(q(dy, dtype))
    out = torch.empty(B, L, C, dtype=dtype, device=DEV)
    m8 = dev(mask.to(torch.uint8))
    vtx._lib.call('vtx_maskfeat_blend_fwd', ops.dt(out), B, Tq, Hq, Hq, C, g, ops.ptr(dev(x, dtype)), ops.ptr(m8),
                  ops.ptr(dev(tok.reshape(-1))), ops.ptr(out), ops.stream())
    check(f'maskfeat blend {dtype}', out.float().cpu(), ref.detach(), TOL[dtype])
    dx = torch.empty(B, L, C, dtype=dtype, device=DEV)
    dtok = torch.empty(C, device=DEV)
    vtx._lib.call('vtx_maskfeat_blend_bwd', ops.dt(out), B, Tq, Hq, Hq, C, g, ops.ptr(dev(dy, dtype)), ops.ptr(m8),
                  ops.ptr(dx), ops.ptr(dtok), ops.stream())
    check(f'maskfeat blend dx {dtype}', dx.float().cpu(), xq.grad, TOL[dtype])
    check(f'maskfeat blend dtoken {dtype}', dtok.cpu(), tq_.grad.reshape(-1), 1e-3)
    # masked MSE
    ts, Cf = 2, 24
    pred = rnd(B, Tq * g * g, ts * Cf, seed=5)
    target = torch.rand(B, Tq * ts, g, g, Cf, generator=torch.Generator().manual_seed(6), dtype=torch.float64)
    cm = (torch.rand(B, Tq * ts, g, g, generator=torch.Generator().manual_seed(7)) > 0.6)
    pq = q(pred, dtype).requires_grad_(True)
    p5 = pq.reshape(B, Tq, g, g, ts, Cf).permute(0, 1, 4, 2, 3, 5).reshape(B, Tq * ts, g, g, Cf)
    err = ((p5 - target) ** 2).mean(-1)
    loss_ref = (err * cm).sum() / (cm.to(torch.int32).sum() + 1e-5)     # int32 sum + 1e-5 -> float32, as in the reference
    loss_ref.backward()
    acc = torch.empty(2, dtype=torch.float64, device=DEV)
    pd = dev(pred, dtype)
    vtx._lib.call('vtx_maskfeat_loss_fwd', ops.dt(pd), B, Tq, ts, g, Cf, ops.ptr(pd), ts * Cf, ops.ptr(dev(target)),
                  ops.ptr(dev(cm.to(torch.uint8))), ops.ptr(acc), ops.stream())
    assert abs(acc[0].item() - loss_ref.item()) / loss_ref.item() < 1e-12, (acc, loss_ref)
    dp = torch.empty_like(pd)
    vtx._lib.call('vtx_maskfeat_loss_bwd', ops.dt(pd), B, Tq, ts, g, Cf, ops.ptr(pd), ts * Cf, ops.ptr(dev(target)),
                  ops.ptr(dev(cm.to(torch.uint8))), ops.ptr(acc), 1.0, ops.ptr(dp), ts * Cf, ops.stream())
    check(f'maskfeat loss grad {dtype}', dp.float().cpu(), pq.grad, TOL[dtype])
    report(f'ok   maskfeat loss {dtype}: {acc[0].item():.12f} vs {loss_ref.item():.12f}')
