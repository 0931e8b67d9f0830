"""Interleaved same-process A/B of the LayerNorm kernels (GPU box only): option ln_rows = 1 .. 4, forward and backward.

    python tools/ln_ab.py [clips] [D]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vtx  # noqa: E402
from vtx import ops  # noqa: E402

DEV = 'cuda:0'


def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    rows = B * 1569
    x = (torch.randn(rows, D, device=DEV) * 0.5).bfloat16()
    y, dx = torch.empty_like(x), torch.empty_like(x)
    g, b_ = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV) * 0.1
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    fw = lambda: ops.layernorm_fwd(x, rows, D, D, ops.IDENT, g, b_, 1e-5, y, D, mean=mean, rstd=rstd)          # noqa: E731
    bw = lambda: ops.layernorm_bwd(y, D, ops.IDENT, x, D, ops.IDENT, rows, D, mean, rstd, g, x, dx, D, dg, db)  # noqa: E731
    res = {}
    for r in range(10):
        for v in ('1', '2', '3', '4'):
            vtx.set_option('ln_rows', v)
            for name, fn in (('fwd', fw), ('bwd', bw)):
                t = timeit(fn)
                if r >= 2:
                    res.setdefault((name, v), []).append(t)
    es = 2
    for name, nb in (('fwd', 2), ('bwd', 4)):
        for v in ('1', '2', '3', '4'):
            t = sorted(res[(name, v)])
            med = t[len(t) // 2]
            print(f'ln_{name} rows {rows} D {D} ln_rows={v}: median {med:7.1f} us  min {t[0]:7.1f} us  {nb * rows * D * es / med / 1e6:6.2f} TB/s')
    # the two forward kernels against each other and against float64
    outs = []
    for v in ('1', '2', '3', '4'):
        vtx.set_option('ln_rows', v)
        fw()
        outs.append((y.float().cpu(), mean.cpu().clone(), rstd.cpu().clone()))
    xd = x.double().cpu()
    ref = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5) * g.double().cpu() + b_.double().cpu()
    for v, (o, m, r) in zip(('1', '2', '3', '4'), outs):
        print(f'ln_rows={v}: max |y - ref| / max|ref| = {(o.double() - ref).abs().max().item() / ref.abs().max().item():.3e}')
    print('outputs equal:', [torch.equal(outs[0][0], o[0]) for o in outs[1:]])


if __name__ == '__main__':
    main()
