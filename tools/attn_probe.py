import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch
from vtx import ops
from vtx._lib import ATTN_SPACE, ATTN_CONTIG
from kernel_bench import timeit
B, T, P, H, D = 32, 8, 196, 12, 768
N = P * T
qkv = torch.randn(B * (N + 1), 3 * D, device='cuda').bfloat16()
o = torch.empty(B * N + B * T, D, device='cuda', dtype=torch.bfloat16)
lse = torch.empty(B * T * H * (P + 1), device='cuda')
for dbg in (0, 16, 1, 2, 4, 8, 1|2|4, 31, 32, 31|64, 31|128):
    os.environ['VTX_ATTN_DBG'] = str(dbg)
    t = timeit(lambda: ops.attn_fwd(qkv, o, lse, ATTN_SPACE, B * T, P + 1, H, 64, 0.125, B, T, P))
    torch.cuda.synchronize(); print('sum', o.float().abs().sum().item())
    print(f'dbg mask {dbg:2d}: {t*1e6:7.1f} us', flush=True)
os.environ['VTX_ATTN_DBG'] = '0'
qc = torch.randn(B * T * 197, 3 * D, device='cuda').bfloat16()
oc = torch.empty(B * T * 197, D, device='cuda', dtype=torch.bfloat16)
t = timeit(lambda: ops.attn_fwd(qc, oc, lse, ATTN_CONTIG, B * T, 197, H, 64, 0.125))
print(f'contig L=197: {t*1e6:7.1f} us')
