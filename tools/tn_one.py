import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
from vtx import ops
M = 50176
for (N1, N2) in ((768, 3072), (768, 768), (2304, 768)):
    x = torch.randn(M, N1, device='cuda').bfloat16()
    y = torch.randn(M, N2, device='cuda').bfloat16()
    for _ in range(3):
        ops.gemm_tn(x, y, M, N1, N2, want_colsum=True)
torch.cuda.synchronize()
