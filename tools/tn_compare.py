"""Weight-gradient GEMM (TN) kernels on the training shapes: ring (256x128, default) against the 256x256 ping-pong
kernel, with and without the fused bias-gradient column sums.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import vtx
from vtx import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 150528
dev = 'cuda:0'
for (N1, N2) in ((768, 3072), (3072, 768), (2304, 768), (768, 768)):
    x = torch.randn(M, N1, device=dev).bfloat16()
    y = torch.randn(M, N2, device=dev).bfloat16()
    ref = None
    for variant in ('ring', 'pp256'):
        vtx.set_option('gemm_tn', variant)
        for cs in (True, False):
            for _ in range(3):
                r = ops.gemm_tn(x, y, M, N1, N2, want_colsum=cs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                r = ops.gemm_tn(x, y, M, N1, N2, want_colsum=cs)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            out = r[0] if cs else r
            if ref is None:
                ref = out
            print(f'{M}x{N1}x{N2} {variant:6s} colsum={int(cs)}: {us:8.1f} us {2.0 * M * N1 * N2 / us * 1e-6:7.1f} TF/s   '
                  f'max diff vs first {(out - ref).abs().max().item():.2e}', flush=True)
vtx.set_option('gemm_tn', 'auto')
