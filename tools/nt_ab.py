"""Interleaved same-process A/B of vtx_gemm_nt option values on one shape (GPU box only).

    python tools/nt_ab.py M N K kind option=v1,v2[,v3] [rounds] [launches]

kind: plain | scale | residual | mul | gelu2.  Every round times `launches` back-to-back launches of each option value
(HIP events on the launch stream), values interleaved; prints median / min per value and the ratio to the first one.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vtx  # noqa: E402
from vtx import ops  # noqa: E402


def main():
    M, N, K = (int(x) for x in sys.argv[1:4])
    kind = sys.argv[4]
    opt, vals = sys.argv[5].split('=')
    vals = vals.split(',')
    rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 12
    L = int(sys.argv[7]) if len(sys.argv) > 7 else 20
    dev = 'cuda:0'
    A = torch.randn(M, K, device=dev).bfloat16()
    W = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(bias=torch.randn(N, device=dev))
    if kind == 'residual':
        kw.update(R=torch.randn(M, N, device=dev).bfloat16(), row_scale=torch.ones(M, device=dev))
    elif kind == 'mul':
        kw = dict(dgelu_in=torch.randn(M, N, device=dev).bfloat16(), dgelu_kind=1)
    elif kind == 'gelu2':
        kw.update(act=2, C2=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    elif kind == 'scale':
        kw.update(row_scale=torch.ones(M, device=dev))
    vtx.set_option('gemm_nt', 'pp256')
    times = {v: [] for v in vals}
    for r in range(rounds + 2):
        for v in vals:
            vtx.set_option(opt, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(L):
                ops.gemm_nt(A, W, C, M, N, K, **kw)
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[v].append(e0.elapsed_time(e1) / L * 1e3)
    base = None
    for v in vals:
        t = sorted(times[v])
        med, mn = t[len(t) // 2], t[0]
        base = base or med
        print(f'{M}x{N}x{K} {kind:8s} {opt}={v}: median {med:8.1f} us  min {mn:8.1f} us  ({2.0 * M * N * K / med / 1e6:7.1f} TF/s)  x{med / base:.4f}')


if __name__ == '__main__':
    main()
