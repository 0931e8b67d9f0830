import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch
import vtx
from vtx import ops
from kernel_bench import timeit
M = 50176
vtx.set_option('gemm_nt', 'pp256')
for skew in ('0', '0.5', '1.0'):
    vtx.set_option('pp_skew', skew)
    for (N, K) in ((3072, 768), (2304, 768), (768, 768), (768, 3072), (3072, 3072)):
        a = torch.randn(M, K, device='cuda').bfloat16()
        w = torch.randn(N, K, device='cuda').bfloat16()
        c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm_nt(a, w, c, M, N, K))
        print(f'skew={skew} N={N} K={K}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF', flush=True)
