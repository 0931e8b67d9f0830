"""Per-tile time of the persistent ping-pong NT GEMM with 8 / 64 / 256 resident workgroups (one CU per XCD
in isolation vs the whole chip): separates what a CU costs alone from what the shared L2 / HBM add."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch
import vtx
from vtx import ops
from kernel_bench import timeit
vtx.set_option('gemm_nt', 'pp256')
N, K = 3072, 768
for grid, M in ((8, 2048), (64, 2048 * 8), (256, 2048 * 32)):
    vtx.set_option('pp_grid', str(grid))
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = torch.randn(N, K, device='cuda').bfloat16()
    c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm_nt(a, w, c, M, N, K))
    ntile = (M // 256) * (N // 256) / grid
    print(f'grid={grid} M={M}: {t*1e6:8.1f} us  per tile {t*1e6/ntile:6.2f} us ({ntile:.0f} tiles/WG)', flush=True)
