"""Micro-benchmarks of the individual libvtx kernels at the TimeSformer-B shapes (GPU box only).
Prints one line per kernel: time, achieved TFLOP/s or GB/s.  Used for tuning; not part of the tests."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vtx  # noqa: E402
from vtx import ops  # noqa: E402
from vtx._lib import ATTN_CONTIG, ATTN_SPACE  # noqa: E402

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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T, P, D, H = 8, 196, 768, 12
    N = P * T
    for dtype in (torch.bfloat16, torch.float32):
        name = 'bf16' if dtype == torch.bfloat16 else 'fp32'
        r = lambda *s: torch.randn(*s, device=DEV).to(dtype)
        for (M, Nn, K, tag) in [(B * N, 3 * D, D, 'qkv'), (B * N, D, D, 'proj'), (B * (N + 1), 4 * D, D, 'ffn1'),
                                (B * (N + 1), D, 4 * D, 'ffn2'), (B * N, D, 3 * D, 'dqkv')]:
            if dtype == torch.float32 and tag not in ('qkv', 'ffn2'):
                continue
            A, W, C = r(M, K), r(Nn, K), torch.empty(M, Nn, device=DEV, dtype=dtype)
            bias = torch.randn(Nn, device=DEV)
            for variant in (['dma2', 'ring128x4k32', 'ring256x3', 'ring256x3k32', 'ring256x4k32'] if dtype == torch.bfloat16 else ['-']):
                vtx.set_option('gemm_nt', variant)
                t = timeit(lambda: ops.gemm_nt(A, W, C, M, Nn, K, bias=bias))
                print(f'gemm_nt {name} {tag:5s} {variant:9s} M={M} N={Nn} K={K}: {t*1e6:8.1f} us  {2*M*Nn*K/t/1e12:7.1f} TFLOP/s', flush=True)
            vtx.set_option('gemm_nt', 'auto')
        for (M, N1, N2, tag) in [(B * N, 3 * D, D, 'dWqkv'), (B * N, D, D, 'dWproj'), (B * (N + 1), 4 * D, D, 'dWffn1'),
                                 (B * (N + 1), D, 4 * D, 'dWffn2')]:
            if dtype == torch.float32 and tag != 'dWqkv':
                continue
            A, Bm = r(M, N1), r(M, N2)
            out = torch.empty(N1, N2, device=DEV)
            for variant in (['dma2', 'ring'] if dtype == torch.bfloat16 else ['-']):
                vtx.set_option('gemm_tn', variant)
                t = timeit(lambda: ops.gemm_tn(A, Bm, M, N1, N2, out=out, want_colsum=True))
                print(f'gemm_tn {name} {tag:6s} {variant:5s} M={M} N1={N1} N2={N2}: {t*1e6:8.1f} us  {2*M*N1*N2/t/1e12:7.1f} TFLOP/s', flush=True)
            vtx.set_option('gemm_tn', 'auto')
        es = 2 if dtype == torch.bfloat16 else 4
        # attention cores
        qkv = r(B * N, 3 * D)
        o = torch.empty(B * N, D, device=DEV, dtype=dtype)
        lse = torch.empty(B * P * H * T, device=DEV)
        t = timeit(lambda: ops.attn_fwd(qkv, o, lse, ATTN_CONTIG, B * P, T, H, 64, 0.125))
        print(f'attn_fwd time  {name}: {t*1e6:8.1f} us  {(B*N*4*D*es)/t/1e9:7.1f} GB/s', flush=True)
        do, dqkv = r(B * N, D), torch.empty(B * N, 3 * D, device=DEV, dtype=dtype)
        t = timeit(lambda: ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_CONTIG, B * P, T, H, 64, 0.125))
        print(f'attn_bwd time  {name}: {t*1e6:8.1f} us  {(B*N*9*D*es)/t/1e9:7.1f} GB/s', flush=True)
        qkv = r(B * (N + 1), 3 * D)
        o = torch.empty(B * N + B * T, D, device=DEV, dtype=dtype)
        lse = torch.empty(B * T * H * (P + 1), device=DEV)
        fl = 4.0 * B * T * H * (P + 1) ** 2 * 64
        t = timeit(lambda: ops.attn_fwd(qkv, o, lse, ATTN_SPACE, B * T, P + 1, H, 64, 0.125, B, T, P))
        print(f'attn_fwd space {name}: {t*1e6:8.1f} us  {fl/t/1e12:7.2f} TFLOP/s', flush=True)
        do = r(B * N + B * T, D)
        dqkv = torch.empty(B * (N + 1), 3 * D, device=DEV, dtype=dtype)
        dcls = torch.empty(B * T, 3 * D, device=DEV, dtype=dtype)
        t = timeit(lambda: ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_SPACE, B * T, P + 1, H, 64, 0.125, B, T, P, dqkv_cls=dcls), iters=5)
        print(f'attn_bwd space {name}: {t*1e6:8.1f} us  {2.5*fl/t/1e12:7.2f} TFLOP/s', flush=True)
        # LayerNorm
        x = r(B * (N + 1), D)
        y = torch.empty_like(x)
        g, b_ = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
        mean, rstd = torch.empty(B * (N + 1), device=DEV), torch.empty(B * (N + 1), device=DEV)
        t = timeit(lambda: ops.layernorm_fwd(x, B * (N + 1), D, D, ops.IDENT, g, b_, 1e-5, y, D, mean=mean, rstd=rstd))
        print(f'ln_fwd {name}: {t*1e6:8.1f} us  {2*x.numel()*es/t/1e9:7.1f} GB/s', flush=True)
        dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        t = timeit(lambda: ops.layernorm_bwd(y, D, ops.IDENT, x, D, ops.IDENT, B * (N + 1), D, mean, rstd, g, x, y, D, dg, db))
        print(f'ln_bwd {name}: {t*1e6:8.1f} us  {4*x.numel()*es/t/1e9:7.1f} GB/s', flush=True)
        t = timeit(lambda: ops.colsum(qkv, B * (N + 1), 3 * D))
        print(f'colsum {name}: {t*1e6:8.1f} us  {qkv.numel()*es/t/1e9:7.1f} GB/s', flush=True)
    frames = torch.randint(0, 256, (64, 224, 224, 3), dtype=torch.uint8, device=DEV)
    t = timeit(lambda: ops.hog_fwd(frames))
    print(f'hog_fwd 64 frames: {t*1e6:8.1f} us  {64/t:9.0f} frames/s  {64*(150528+169344)/t/1e9:7.1f} GB/s', flush=True)


if __name__ == '__main__':
    main()
