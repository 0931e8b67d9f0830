"""Weight-gradient GEMM (TN) kernels on the training shapes, interleaved A/B: by default the 256x256 ping-pong kernel
(gemm_tn=pp256, two waves per SIMD, 128x64 wave tiles) against the one-wave-per-SIMD kernel (gemm_tn=w4, 128x128 wave
tiles), with and without the fused bias-gradient column sums; launch + slab reduction per call.  GPU box only.

    python tools/tn_compare.py [M] [variant ...]          e.g.  python tools/tn_compare.py 150528 pp256 w4 ring
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import vtx
from vtx import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 150528
variants = sys.argv[2:] or ['pp256', 'w4']
dev = 'cuda:0'
ROUNDS, REPS = 6, 10
for (N1, N2) in ((768, 3072), (3072, 768), (2304, 768), (768, 768)):
    x = torch.randn(M, N1, device=dev).bfloat16()
    y = torch.randn(M, N2, device=dev).bfloat16()
    for cs in (True, False):
        ref, best, tot = None, {}, {}
        for v in variants:                                   # warm-up + bit comparison
            vtx.set_option('gemm_tn', v)
            for _ in range(3):
                r = ops.gemm_tn(x, y, M, N1, N2, want_colsum=cs)
            out = r[0] if cs else r
            ref = out if ref is None else ref
            tot[v] = (0.0, (out - ref).abs().max().item())
        for _ in range(ROUNDS):                              # interleaved: every variant sees the same clock / thermal state
            for v in variants:
                vtx.set_option('gemm_tn', v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(REPS):
                    ops.gemm_tn(x, y, M, N1, N2, want_colsum=cs)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1000 / REPS
                tot[v] = (tot[v][0] + us, tot[v][1])
                best[v] = min(best.get(v, 1e9), us)
        for v in variants:
            us = tot[v][0] / ROUNDS
            print(f'{M}x{N1}x{N2} {v:6s} colsum={int(cs)}: mean {us:8.1f} us  best {best[v]:8.1f} us  {2.0 * M * N1 * N2 / us * 1e-6:7.1f} TF/s '
                  f'= {2.0 * M * N1 * N2 / us * 1e-6 / 2500:.3f} of 2.5 PF   max diff vs {variants[0]} {tot[v][1]:.2e}', flush=True)
vtx.set_option('gemm_tn', 'auto')
