"""Run a few bf16 GEMM launches (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
from vtx import ops
M = 50176
for (N, K) in ((3072, 3072), (768, 768)):
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = torch.randn(N, K, device='cuda').bfloat16()
    c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt(a, w, c, M, N, K)
x = torch.randn(M, 768, device='cuda').bfloat16()
y = torch.randn(M, 3072, device='cuda').bfloat16()
for _ in range(3):
    ops.gemm_tn(x, y, M, 768, 3072)
torch.cuda.synchronize()
