"""Main-loop bound of the persistent NT GEMM: the same launch with the A and / or B operand collapsed onto one row
(leading dimension 0: every tile row is the same 128-byte line, always cache resident).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import vtx
from vtx import ops

M, N, K = 100352, 768, 3072
dev = 'cuda:0'
A = torch.randn(M, K, device=dev).bfloat16()
W = torch.randn(N, K, device=dev).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
vtx.set_option('gemm_nt', 'pp256')
for name, kw in (('normal', {}), ('A hot', dict(lda=0)), ('B hot', dict(ldb=0)), ('A+B hot', dict(lda=0, ldb=0))):
    for _ in range(3):
        ops.gemm_nt(A, W, C, M, N, K, **kw)
    trace = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
    vtx.set_option('pp_trace', str(trace.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_nt(A, W, C, M, N, K, **kw); e1.record()
    torch.cuda.synchronize()
    vtx.set_option('pp_trace', '0')
    t = trace.cpu().reshape(256, 8, 16).double()
    ok = (t[:, :, 1] > 0) & (t[:, :, 2] > 0)
    ml = (t[:, :, 2] - t[:, :, 1])[ok] * 0.01
    cyc = (t[:, :, 9] - t[:, :, 8])[ok]
    print(f'{name:8s}: launch {e0.elapsed_time(e1)*1e3:7.1f} us  main loop {ml.mean():6.2f} us = {ml.mean() / (K // 64):.3f} us per K tile, '
          f'{cyc.mean() / (K // 64):.0f} shader cycles per K tile (MFMA floor 2048) at {cyc.mean() / ml.mean():.0f} MHz', flush=True)
