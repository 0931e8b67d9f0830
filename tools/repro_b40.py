import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
from vtx import ops
M, N, K = 62720, 768, 768
a = torch.randn(M, K, device='cuda').bfloat16()
w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
b = torch.randn(N, device='cuda')
c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for bias in (None, b):
    print('bias', bias is not None, flush=True)
    ops.gemm_nt(a, w, c, M, N, K, bias=bias)
    torch.cuda.synchronize()
    ref = a[:4096].float() @ w.float().t() + (bias if bias is not None else 0)
    print('err', (c[:4096].float() - ref).abs().max().item(), flush=True)
    ref = a[-4096:].float() @ w.float().t() + (bias if bias is not None else 0)
    print('err tail', (c[-4096:].float() - ref).abs().max().item(), flush=True)
