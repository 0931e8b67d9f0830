"""Per-shape table of the 14 vtx_gemm_nt and 7 vtx_gemm_tn launches of one TimeSformer-B layer
(fwd + dgrad + wgrad) with the epilogue each one really runs (GPU box only).

    python tools/gemm_shapes.py [clips] [frames] [--stream-f32] [--old-gelu]

Per launch: time (HIP events, 20 launches), TFLOP/s, ALGORITHMIC bytes (every operand the launch
must read or write once: A, the weight, C, and the epilogue's residual / pre-activation copy /
GELU' input), the bound that byte and FLOP count imply on MI355X (2.5 PFLOP/s dense bf16, 8 TB/s)
and the fraction of that bound reached.  `--stream-f32` uses the fp32 residual-stream epilogues.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from vtx import ops  # noqa: E402

DEV = 'cuda:0'
PEAK_TF, PEAK_TB = 2500.0, 8.0


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    B = int(args[0]) if args else 64
    T = int(args[1]) if len(args) > 1 else 8
    sdt = torch.float32 if '--stream-f32' in sys.argv else torch.bfloat16
    P, D, Hd = 196, 768, 3072
    N = P * T
    Mt, Ms, Mo = B * N, B * (N + 1), B * N + B * T
    bf = torch.bfloat16
    r = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(bf)
    tm = ops.tokmap(N)
    x = (torch.randn(B, N + 1, D, device=DEV) * 0.5).to(sdt)          # residual stream
    out = torch.empty_like(x)
    bias = {n: torch.randn(n, device=DEV) * 0.1 for n in (D, 3 * D, Hd)}
    sv_t = torch.rand(B * P, device=DEV).round() / 0.9
    sv_s = torch.rand(B * T, device=DEV).round() / 0.9
    sv_f = torch.rand(B, device=DEV).round() / 0.9
    W = {(n, k): r(n, k) for (n, k) in ((3 * D, D), (D, D), (Hd, D), (D, Hd), (D, 3 * D))}
    es, ss = 2, x.element_size()
    rows = []

    def nt(tag, M, Nn, K, extra_bytes, **kw):
        A = kw.pop('A', None)
        if A is None:
            A = r(M, K)
        C = kw.pop('C', None)
        if C is None:
            C = torch.empty(M, Nn, device=DEV, dtype=bf)
        t = timeit(lambda: ops.gemm_nt(A, W[(Nn, K)], C, M, Nn, K, **kw))
        fl = 2.0 * M * Nn * K
        by = M * K * es + Nn * K * es + extra_bytes
        rows.append(('NT ' + tag, M, Nn, K, t, fl, by))

    h = r(Ms, Hd)
    g = torch.empty(Ms, Hd, device=DEV, dtype=bf)
    a_cls = torch.empty(B * T, D, device=DEV, dtype=sdt)
    # ---- forward
    nt('qkv_t fwd', Mt, 3 * D, D, Mt * 3 * D * es, bias=bias[3 * D])
    # attn.proj and temporal_fc run as ONE GEMM with the product weight (vtx/functions.py TimeAttnFn)
    nt('proj.tfc fwd (+x)', Mt, D, D, 2 * Mt * D * ss, C=out, cmap=tm, bias=bias[D], R=x, rmap=tm, row_scale=sv_t, rs=(T, 1, 1, 0))
    nt('qkv_s fwd', Ms, 3 * D, D, Ms * 3 * D * es, bias=bias[3 * D])
    nt('proj_s fwd (+x)', Mo, D, D, 2 * Mt * D * ss + B * T * D * ss, C=out, cmap=tm, bias=bias[D], row_scale=sv_s,
       rs=(N, T, T, 1), R=x, rmap=tm, split_row=B * N, Csplit=a_cls)
    nt("fc1 fwd (g,g')", Ms, Hd, D, 2 * Ms * Hd * es, C=g, bias=bias[Hd], act=2, C2=h)
    if '--old-gelu' in sys.argv:
        nt('fc1 fwd (h,g) old', Ms, Hd, D, 2 * Ms * Hd * es, C=g, bias=bias[Hd], act=1, C2=h)
    nt('fc2 fwd (+x)', Ms, D, Hd, 2 * Ms * D * ss, C=out.view(Ms, D), bias=bias[D], row_scale=sv_f, rs=(N + 1, 1, 1, 0),
       R=x.view(Ms, D))
    # ---- input gradients
    dout = r(B, N + 1, D)
    nt('proj.tfc dgrad', Mt, D, D, Mt * D * es, A=dout, amap=tm, row_scale=sv_t, rs=(T, 1, 1, 0))
    nt('qkv_t dgrad', Mt, D, 3 * D, Mt * D * es)
    nt('proj_s dgrad', Mo, D, D, Mo * D * es)
    nt('qkv_s dgrad', Ms, D, 3 * D, Ms * D * es)
    nt("fc2 dgrad (*g')", Ms, Hd, D, 2 * Ms * Hd * es, dgelu_in=h, dgelu_kind=1)
    if '--old-gelu' in sys.argv:
        nt("fc2 dgrad (gelu') old", Ms, Hd, D, 2 * Ms * Hd * es, dgelu_in=h)
    nt('fc1 dgrad', Ms, D, Hd, Ms * D * es)

    def tn(tag, M, N1, N2, **kw):
        A, Bm = kw.pop('A', None), r(M, N2)
        if A is None:
            A = r(M, N1)
        o = torch.empty(N1, N2, device=DEV)
        t = timeit(lambda: ops.gemm_tn(A, Bm, M, N1, N2, out=o, want_colsum=True, **kw))
        rows.append(('TN ' + tag, M, N1, N2, t, 2.0 * M * N1 * N2, M * (N1 + N2) * es + N1 * N2 * 4))

    tn('proj.tfc wgrad', Mt, D, D, A=dout, amap=tm)
    tn('qkv_t wgrad', Mt, 3 * D, D)
    tn('proj_s wgrad', Mo, D, D)
    tn('qkv_s wgrad', Ms, 3 * D, D)
    tn('fc2 wgrad', Ms, D, Hd)
    tn('fc1 wgrad', Ms, Hd, D)

    print(f'clips {B}, frames {T}, stream dtype {sdt}; env ' +
          ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('VTX_')))
    print(f'{"launch":22s} {"M":>7s} {"N":>5s} {"K":>5s} {"us":>8s} {"TF/s":>7s} {"alg MB":>8s} {"TB/s":>6s} bound  frac')
    tot = {'NT': [0.0, 0.0], 'TN': [0.0, 0.0]}
    for tag, M, Nn, K, t, fl, by in rows:
        t_m, t_h = fl / (PEAK_TF * 1e12), by / (PEAK_TB * 1e12)
        bound = 'mfma' if t_m >= t_h else 'hbm'
        frac = max(t_m, t_h) / t
        print(f'{tag:22s} {M:7d} {Nn:5d} {K:5d} {t * 1e6:8.1f} {fl / t / 1e12:7.1f} {by / 1e6:8.1f} {by / t / 1e12:6.2f} '
              f'{bound:5s} {frac:5.3f}', flush=True)
        tot[tag[:2]][0] += t
        tot[tag[:2]][1] += fl
    for k, (t, fl) in tot.items():
        print(f'{k} sum: {t * 1e6:8.1f} us per layer, {fl / t / 1e12:7.1f} TF/s = {fl / t / 1e12 / PEAK_TF:5.3f} of the MFMA peak')


if __name__ == '__main__':
    main()
