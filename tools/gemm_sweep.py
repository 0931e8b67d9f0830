"""K sweep of the bf16 NT / TN GEMMs: t = T_fixed + K * t_k per tile (latency model of DESIGN.md)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch
from vtx import ops
from kernel_bench import timeit
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50176
for N in (768, 3072):
    for K in (128, 256, 512, 768, 1536, 3072):
        a = torch.randn(M, K, device='cuda').bfloat16()
        w = torch.randn(N, K, device='cuda').bfloat16()
        c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm_nt(a, w, c, M, N, K))
        print(f'NT M={M} N={N} K={K}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF', flush=True)
for (K1, K2) in ((768, 768), (768, 3072), (3072, 768), (2304, 768)):
    for M2 in ((M // 4, M // 2, M) if len(sys.argv) < 3 else (M,)):
        x = torch.randn(M2, K1, device='cuda').bfloat16()
        y = torch.randn(M2, K2, device='cuda').bfloat16()
        t = timeit(lambda: ops.gemm_tn(x, y, M2, K1, K2))
        print(f'TN M={M2} N1={K1} N2={K2}: {t*1e6:8.1f} us  {2.0*M2*K1*K2/t/1e12:7.1f} TF', flush=True)
