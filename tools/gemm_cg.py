import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch
import vtx
from vtx import ops
from kernel_bench import timeit
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100352
for (N, K) in ((3072, 768), (2304, 768), (768, 3072), (768, 2304), (768, 768)):
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = torch.randn(N, K, device='cuda').bfloat16()
    c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    res = []
    for cg in (1, 2, 3, 4, 6, 8, 12):
        vtx.set_option('pp_cg', str(cg))
        t = timeit(lambda: ops.gemm_nt(a, w, c, M, N, K))
        res.append(f'cg{cg}={t*1e6:.0f}us/{2.0*M*N*K/t/1e12:.0f}TF')
    print(f'N={N} K={K}: ' + ' '.join(res), flush=True)
