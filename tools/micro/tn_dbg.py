import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch, vtx
from vtx import ops
M = 150528; dev = 'cuda:0'
vtx.set_option('gemm_tn', 'pp256')
for (N1, N2) in ((768, 3072), (768, 768)):
    x = torch.randn(M, N1, device=dev).bfloat16(); y = torch.randn(M, N2, device=dev).bfloat16()
    for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
        vtx.set_option('pp_epi', str(dbg))
        for _ in range(3): ops.gemm_tn(x, y, M, N1, N2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_tn(x, y, M, N1, N2)
        e1.record(); torch.cuda.synchronize()
        print(f'{N1}x{N2} dbg={dbg} (noDMA={dbg&1} halfreads={(dbg>>1)&1} noMFMA={(dbg>>2)&1}): {e0.elapsed_time(e1)*100:8.1f} us', flush=True)
vtx.set_option('pp_epi', '0')
