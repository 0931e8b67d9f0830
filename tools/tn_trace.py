import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch
from vtx import ops
M, N1, N2 = 50176, 768, 3072
x = torch.randn(M, N1, device='cuda').bfloat16()
y = torch.randn(M, N2, device='cuda').bfloat16()
for _ in range(3):
    ops.gemm_tn(x, y, M, N1, N2)
tr = torch.zeros(256, dtype=torch.int64, device='cuda')
os.environ['VTX_TN_TRACE'] = str(tr.data_ptr())
ops.gemm_tn(x, y, M, N1, N2)
torch.cuda.synchronize()
t = tr.cpu().numpy()
for g in range(2):
    v = t[g * 128:(g + 1) * 128]
    d = (v[1:] - v[:-1])
    print('group', g, 'deltas between consecutive barriers (cycles):')
    print(' '.join(str(int(z)) for z in d[:96]))
