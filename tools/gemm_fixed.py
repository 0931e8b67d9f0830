import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch
from vtx import ops
from kernel_bench import timeit
M, N = 50176, 3072
for var in ('pp256',):
    os.environ['VTX_GEMM_NT'] = var
    for dbg in (0, 1, 4):
        os.environ['VTX_GEMM_DBG'] = str(dbg)
        for K in (128, 768, 3072):
            a = torch.randn(M, K, device='cuda').bfloat16()
            w = torch.randn(N, K, device='cuda').bfloat16()
            c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
            t = timeit(lambda: ops.gemm_nt(a, w, c, M, N, K))
            print(f'{var} dbg={dbg} K={K}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF', flush=True)
