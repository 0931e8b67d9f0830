"""The one-wave-per-SIMD NT GEMM (gemm_nt=w4) against the ping-pong kernel: same outputs, launch time, main-loop time per
64-deep K tile from the device timeline.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import vtx
from vtx import ops

dev = 'cuda:0'
torch.manual_seed(0)
for (M, N, K) in ((12544, 768, 768), (100352, 768, 768), (100352, 2304, 768), (100352, 3072, 768), (100352, 768, 3072), (3000, 216, 3072)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    outs = {}
    for variant in ('pp256', 'w4', 'dual'):
        vtx.set_option('gemm_nt', variant)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm_nt(A, W, C, M, N, K, bias=bias)
        trace = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
        vtx.set_option('pp_trace', str(trace.data_ptr()))
        ops.gemm_nt(A, W, C, M, N, K, bias=bias)
        torch.cuda.synchronize()
        vtx.set_option('pp_trace', '0')
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_nt(A, W, C, M, N, K, bias=bias)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        t = trace.cpu().reshape(256, 8, 8).double() * 0.01
        ok = (t[:, :, 1] > 0) & (t[:, :, 2] > 0) & (t[:, :, 7] > 0)
        ml = (t[:, :, 2] - t[:, :, 1])[ok].mean().item()
        tile = (t[:, :, 7] - t[:, :, 0])[ok].mean().item()
        if variant == 'w4':
            cyc = (t[:, :, 4] - t[:, :, 3])[ok].mean().item() * 100
            print(f'    shader clock in the main loop: {cyc / ml:.0f} MHz ({cyc / (K / 64):.0f} cycles per 64-deep K tile; MFMA floor 2048)')
        outs[variant] = C.float()
        print(f'{M}x{N}x{K} {variant:6s}: {us:8.1f} us {2.0 * M * N * K / us * 1e-6:7.1f} TF/s   main loop {ml:6.2f} us '
              f'({ml / (K / 64):.3f} per 64-deep K tile)  tile {tile:6.2f} us', flush=True)
    ref = (A.float() @ W.float().t() + bias)
    for v, c in outs.items():
        err = ((c - ref).norm() / ref.norm()).item()
        print(f'    {v}: rel l2 vs fp32 = {err:.3e}  max|v-pp| = {(c - outs["pp256"]).abs().max().item():.3e}')
