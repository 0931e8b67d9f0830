"""Is the NT main loop bound by the L2 latency of the streamed A operand?  Same GEMM with every row tile
mapped onto the same 256 rows of A (row map grp=256, skip=-256): A is then always an L2 hit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch
from vtx import ops
from vtx._lib import RowMap
from kernel_bench import timeit
M = 100352
for (N, K) in ((3072, 768), (768, 3072), (3072, 3072), (768, 768)):
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = torch.randn(N, K, device='cuda').bfloat16()
    c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.gemm_nt(a, w, c, M, N, K))
    t1 = timeit(lambda: ops.gemm_nt(a, w, c, M, N, K, amap=RowMap(256, -256, 0)))
    print(f'N={N} K={K}: streamed A {t0*1e6:7.1f} us {2.0*M*N*K/t0/1e12:6.0f} TF | hot A {t1*1e6:7.1f} us {2.0*M*N*K/t1/1e12:6.0f} TF', flush=True)
