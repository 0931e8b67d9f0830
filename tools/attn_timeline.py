"""Device timeline of the streamed attention backward (attn_fused=2), GPU box only.

    python videotransformer-pytorch_amd/csrc/build.py --variant trace VTX_STREAM_TRACE=1
    VTX_LIB=videotransformer-pytorch_amd/libvtx_trace.so python tools/attn_timeline.py [clips]

Workgroup 0 stamps the shader clock (s_memtime, 100 MHz-independent: core clock counts) at fixed points of its first four
items: wave 0 (a worker) and wave 7 (the feeder).  Prints per step the cycles between consecutive points (items 1..3: item 0
has the prologue in front)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vtx  # noqa: E402
from vtx import ops  # noqa: E402
from vtx._lib import ATTN_SPACE  # noqa: E402

DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
T, P, D, H = 8, 196, 768, 12
hd, N = 64, P * T
bf = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(bf)   # noqa: E731
M1, Mo = B * (N + 1), B * N + B * T
qkv, o, do = r(M1, 3 * D), torch.empty(Mo, D, device=DEV, dtype=bf), r(Mo, D)
S, L = B * T, P + 1
lse = torch.empty(S * H * L, device=DEV)
dqkv = torch.empty(M1, 3 * D, device=DEV, dtype=bf)
dcls = torch.empty(B * T, 3 * D, device=DEV, dtype=bf)
ops.attn_fwd(qkv, o, lse, ATTN_SPACE, S, L, H, hd, hd ** -0.5, B, T, P)
vtx.set_option('attn_fused', '2')
trace = torch.zeros(2 * 4 * 8 * 8, dtype=torch.int64, device=DEV)
for _ in range(3):
    ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_SPACE, S, L, H, hd, hd ** -0.5, B, T, P, dqkv_cls=dcls)
vtx.set_option('pp_trace', str(trace.data_ptr()))
ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_SPACE, S, L, H, hd, hd ** -0.5, B, T, P, dqkv_cls=dcls)
torch.cuda.synchronize()
vtx.set_option('pp_trace', '0')
t = trace.cpu().view(2, 4, 8, 8).numpy()
if not t.any():
    sys.exit('no stamps: load a VTX_STREAM_TRACE build (VTX_LIB=...)')
wn = ['top', 'packed + dS written', 'dv / dk issued', 'next scores issued', 'barrier passed']
print('worker (wave 0): cycles from the previous point; columns = steps 0..6')
for it in (1, 2, 3):
    print(f' item {it}: item top (K / V fragments + scores 0) {t[0, it, 7, 1] - t[0, it, 7, 0]}; previous item end: kv wait '
          f'{t[0, it - 1, 7, 2] - t[0, it - 1, 6, 4]}, dk/dv stores {t[0, it - 1, 7, 3] - t[0, it - 1, 7, 2]}, '
          f'to this top {t[0, it, 7, 0] - t[0, it - 1, 7, 3]};  whole item {t[0, it, 7, 3] - t[0, it - 1, 7, 3]} cycles')
    for pt in range(1, 5):
        print(f'   {wn[pt]:20s}', ' '.join(f'{int(t[0, it, i, pt] - t[0, it, i, pt - 1]):6d}' for i in range(7)))
    print('   step total          ', ' '.join(f'{int(t[0, it, i, 4] - t[0, it, i, 0]):6d}' for i in range(7)))
fn = ['(arrive)', 'barrier passed', 'dq products', 'vmcnt(13)', 'requests issued', 'dq tile staged', '-']
print('wave 7')
for it in (1, 2):
    print(f' item {it}: wait at T {t[1, it, 7, 0] - t[1, it - 1, 6, 6]} (from the last request of the item before), K^T fragments {t[1, it, 7, 1] - t[1, it, 7, 0]}')
    for pt in range(1, 7):
        print(f'   {fn[pt]:20s}', ' '.join(f'{int(t[1, it, i, pt] - t[1, it, i, pt - 1]):6d}' for i in range(7)))
    print('   to next barrier     ', ' '.join(f'{int((t[1, it, i + 1, 0] if i < 6 else t[1, it + 1, 7, 0]) - t[1, it, i, 6]):6d}' for i in range(7)))
