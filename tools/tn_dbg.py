import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch
from vtx import ops
from kernel_bench import timeit
M = 50176
for dbg in (0, 1, 2, 3):
    os.environ['VTX_TN_DBG'] = str(dbg)
    for (N1, N2) in ((768, 3072), (768, 768)):
        x = torch.randn(M, N1, device='cuda').bfloat16()
        y = torch.randn(M, N2, device='cuda').bfloat16()
        t = timeit(lambda: ops.gemm_tn(x, y, M, N1, N2))
        print(f'dbg={dbg} TN {N1}x{N2}: {t*1e6:8.1f} us  {2.0*M*N1*N2/t/1e12:7.1f} TF', flush=True)
