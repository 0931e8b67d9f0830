import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch, vtx
from vtx import ops
M = 150528; dev = 'cuda:0'
vtx.set_option('gemm_tn', 'pp256')
x = torch.randn(M, 768, device=dev).bfloat16(); y = torch.randn(M, 3072, device=dev).bfloat16()
for cs in (True, False):
    for _ in range(6): ops.gemm_tn(x, y, M, 768, 3072, want_colsum=cs)
    torch.cuda.synchronize()
