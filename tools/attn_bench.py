"""Attention / LayerNorm kernels at the benchmark shapes (GPU box only): time and algorithmic TB/s per launch.

    python tools/attn_bench.py [clips] [option=value ...]       e.g.  python tools/attn_bench.py 96 attn_hw=1
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vtx  # noqa: E402
from vtx import ops  # noqa: E402
from vtx._lib import ATTN_CONTIG, ATTN_SPACE, IDENT  # noqa: E402

DEV = 'cuda:0'


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
    args = [a for a in sys.argv[1:] if '=' not in a]
    for a in sys.argv[1:]:
        if '=' in a:
            k, v = a.split('=')
            vtx.set_option(k, v)
    B = int(args[0]) if args else 96
    T, P, D, H = 8, 196, 768, 12
    hd = D // H
    N = P * T
    bf = torch.bfloat16
    r = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(bf)
    rows = []
    # temporal: sequences of T contiguous rows
    M = B * N
    qkv, o, do = r(M, 3 * D), torch.empty(M, D, device=DEV, dtype=bf), r(M, D)
    S = M // T
    lse = torch.empty(S * H * T, device=DEV)
    dqkv = torch.empty(M, 3 * D, device=DEV, dtype=bf)
    t = timeit(lambda: ops.attn_fwd(qkv, o, lse, ATTN_CONTIG, S, T, H, hd, hd ** -0.5))
    rows.append(('attn fwd temporal', t, M * D * 4 * 2))
    t = timeit(lambda: ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_CONTIG, S, T, H, hd, hd ** -0.5))
    rows.append(('attn bwd temporal', t, M * D * 8 * 2))
    # spatial: [B, 1+N, .] rows, sequences (b, t) of 1 + P tokens
    M1, Mo = B * (N + 1), B * N + B * T
    qkv, o, do = r(M1, 3 * D), torch.empty(Mo, D, device=DEV, dtype=bf), r(Mo, D)
    S, L = B * T, P + 1
    lse = torch.empty(S * H * L, device=DEV)
    dqkv = torch.empty(M1, 3 * D, device=DEV, dtype=bf)
    dcls = torch.empty(B * T, 3 * D, device=DEV, dtype=bf)
    t = timeit(lambda: ops.attn_fwd(qkv, o, lse, ATTN_SPACE, S, L, H, hd, hd ** -0.5, B, T, P))
    rows.append(('attn fwd spatial', t, Mo * D * 4 * 2))
    t = timeit(lambda: ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_SPACE, S, L, H, hd, hd ** -0.5, B, T, P, dqkv_cls=dcls))
    rows.append(('attn bwd spatial', t, Mo * D * 8 * 2))
    # LayerNorm
    x, y, dy, dres = r(M1, D), torch.empty(M1, D, device=DEV, dtype=bf), r(M1, D), r(M1, D)
    g, b_ = torch.randn(D, device=DEV), torch.randn(D, device=DEV)
    mean, rstd = torch.empty(M1, device=DEV), torch.empty(M1, device=DEV)
    t = timeit(lambda: ops.layernorm_fwd(x, M1, D, D, IDENT, g, b_, 1e-5, y, D, IDENT, mean, rstd))
    rows.append(('layernorm fwd', t, M1 * D * 2 * 2))
    dx = torch.empty_like(x)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    t = timeit(lambda: ops.layernorm_bwd(dy, D, IDENT, x, D, IDENT, M1, D, mean, rstd, g, dres, dx, D, dg, db))
    rows.append(('layernorm bwd (+res)', t, M1 * D * 4 * 2))
    print(f'clips {B}; options ' + ' '.join(a for a in sys.argv[1:] if '=' in a))
    for name, t, by in rows:
        print(f'{name:22s} {t * 1e6:8.1f} us  {by / t / 1e12:6.2f} TB/s', flush=True)


if __name__ == '__main__':
    main()
