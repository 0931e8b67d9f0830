import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
from vtx import ops
M = 50176
a = torch.randn(M, 768, device='cuda').bfloat16()
w = torch.randn(3072, 768, device='cuda').bfloat16()
c = torch.empty(M, 3072, device='cuda', dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm_nt(a, w, c, M, 3072, 768)
for _ in range(3):
    ops.gemm_tn(a, c, M, 768, 3072)
torch.cuda.synchronize()
