"""Phase timeline of the ping-pong weight-gradient kernel (GPU box only): wave 0 of every workgroup stamps the shader
clock at 15 points of K tiles 8..15 (`vtx.set_option('pp_trace', <device address>)`); prints mean cycles per segment.

    python tools/tn_timeline.py [M N1 N2]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import vtx
from vtx import ops

SEG = ['P1 16 transpose reads (A0) + lgkmcnt(0)', 'P1 (stamp only)', 'P1 barrier', 'P1 8 MFMAs (+ 8 reads of B1 in the gaps) + barrier',
       'P2 request A1(kt+1) + lgkmcnt(0)', 'P2 vmcnt wait (A1 of this tile)', 'P2 barrier', 'P2 8 MFMAs (+ A0(kt+2) request in the gaps) + barrier',
       'P3 16 transpose reads (A1) + vmcnt wait (A0, B0 of next tile)', 'P3 barrier', 'P3 8 MFMAs + barrier',
       'P4 request B0(kt+2) + vmcnt wait (B1 of next tile)', 'P4 barrier', 'P4 8 MFMAs (+ B1(kt+2) request, 8 reads of next B0 in the gaps) + barrier']
a = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else [150528, 768, 3072]
M, N1, N2 = a
dev = 'cuda:0'
x = torch.randn(M, N1, device=dev).bfloat16()
y = torch.randn(M, N2, device=dev).bfloat16()
vtx.set_option('gemm_tn', 'pp256')
for _ in range(3):
    ops.gemm_tn(x, y, M, N1, N2, want_colsum=True)
trace = torch.zeros(256 * 8 * 24, dtype=torch.int64, device=dev)
vtx.set_option('pp_trace', str(trace.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm_tn(x, y, M, N1, N2, want_colsum=True); e1.record()
torch.cuda.synchronize()
vtx.set_option('pp_trace', '0')
vtx.set_option('gemm_tn', 'auto')
t = trace.cpu().reshape(256, 8, 24).double()
ok = (t[:, :, 0] > 0) & (t[:, :, 14] > 0)
print(f'{M}x{N1}x{N2}: launch {e0.elapsed_time(e1) * 1e3:.1f} us; {int(ok.sum())} traced K tiles')
tot = 0.0
for i, name in enumerate(SEG):
    d = (t[:, :, i + 1] - t[:, :, i])[ok]
    tot += d.mean().item()
    print(f'  {name:48s} mean {d.mean():7.0f}  p50 {d.median():7.0f}  p90 {d.quantile(0.9):7.0f} cycles')
for ph, (a, b, c) in enumerate(((3, 15, 4), (7, 16, 8), (10, 17, 11), (13, 18, 14))):
    mma = (t[:, :, b] - t[:, :, a])[ok]; bar = (t[:, :, c] - t[:, :, b])[ok]
    print(f'  P{ph + 1}: 8 MFMAs + request (+ prefetch) {mma.mean():6.0f} cycles, closing barrier {bar.mean():6.0f}')
print(f'  K tile total {tot:.0f} cycles (MFMA floor 2048; stamps cost ~50 cycles each)')
