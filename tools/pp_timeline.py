"""Where does a tile of the persistent NT GEMM spend its time?  (GPU box only)

    python tools/pp_timeline.py [M N K] [residual|dgelu|mul|gelu2|scale|plain] [option=value ...]

The kernel stamps the 100 MHz constant clock at 8 points of each of the first 8 tiles of every workgroup
(wave 0, lane 0; `vtx.set_option('pp_trace', <device address>)`); this prints the mean / p90 duration of
every segment over all workgroups and tiles 1..6 (steady state), in microseconds.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vtx  # noqa: E402
from vtx import ops  # noqa: E402

SEG = ['0-1 wait first regions + barrier', '1-2 main loop', '2-3 publish next tile', '3-4 epilogue loads (wait)',
       '4-5 prologue issue (no-residual kernels)', '5-6 eight store passes', '6-7 barrier + prologue (residual kernels)',
       '7-0 to next tile top']


def main():
    a = [x for x in sys.argv[1:] if x.isdigit()]
    M, N, K = (int(a[0]), int(a[1]), int(a[2])) if len(a) == 3 else (100352, 768, 768)
    kind = next((x for x in sys.argv[1:] if not x.isdigit() and '=' not in x), 'plain')
    for x in sys.argv[1:]:
        if '=' in x:
            vtx.set_option(*x.split('='))
    dev = 'cuda:0'
    A = torch.randn(M, K, device=dev).bfloat16()
    W = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    kw = dict(bias=bias)
    if kind == 'residual':
        kw.update(R=torch.randn(M, N, device=dev).bfloat16(), row_scale=torch.ones(M, device=dev))
    elif kind == 'dgelu':
        kw = dict(dgelu_in=torch.randn(M, N, device=dev).bfloat16())
    elif kind == 'mul':
        kw = dict(dgelu_in=torch.randn(M, N, device=dev).bfloat16(), dgelu_kind=1)
    elif kind == 'gelu2':
        kw.update(act=2, C2=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    elif kind == 'scale':
        kw.update(row_scale=torch.ones(M, device=dev))
    vtx.set_option('gemm_nt', 'pp256')
    for _ in range(3):
        ops.gemm_nt(A, W, C, M, N, K, **kw)
    trace = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
    vtx.set_option('pp_trace', str(trace.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_nt(A, W, C, M, N, K, **kw)
    e1.record()
    torch.cuda.synchronize()
    vtx.set_option('pp_trace', '0')
    raw = trace.cpu().reshape(256, 8, 16).double()
    t = raw[:, :, :8] * 0.01                                     # us
    skipped = (t[:, :, 4] == 0) & (t[:, :, 3] > 0)              # continuous operand flow: no epilogue-load wait, stamp 4 is not taken
    t[:, :, 4] = torch.where(skipped, t[:, :, 3], t[:, :, 4])
    print(f'M={M} N={N} K={K} {kind} {" ".join(x for x in sys.argv[1:] if "=" in x)}: launch {e0.elapsed_time(e1) * 1e3:.1f} us, {2.0 * M * N * K / e0.elapsed_time(e1) / 1e9:.0f} TF/s')
    valid = (t[:, :, 0] > 0) & (t[:, :, 7] > 0)
    ntiles = valid.sum(1)
    print(f'tiles traced per workgroup: min {int(ntiles.min())} max {int(ntiles.max())}; kernel span '
          f'{(t[valid][:, 7].max() - t[t[:, :, 0] > 0][:, 0].min()):.1f} us')
    for s in range(8):
        if s < 7:
            d = (t[:, :, s + 1] - t[:, :, s])[valid]
            sel = d
        else:
            nxt = t[:, 1:, 0] - t[:, :-1, 7]
            sel = nxt[valid[:, 1:] & valid[:, :-1]]
        if sel.numel():
            print(f'  {SEG[s]:45s} mean {sel.mean():7.2f}  p50 {sel.median():7.2f}  p90 {sel.quantile(0.9):7.2f}  max {sel.max():7.2f}')
    whole = (t[:, 1:, 0] - t[:, :-1, 0])[valid[:, 1:] & valid[:, :-1]]
    print(f'  tile period (top to top)                      mean {whole.mean():7.2f}  p50 {whole.median():7.2f}  p90 {whole.quantile(0.9):7.2f}')
    cyc = (raw[:, :, 9] - raw[:, :, 8])[valid]
    ml = (t[:, :, 2] - t[:, :, 1])[valid]
    print(f'  main loop: {cyc.mean() / (K // 64):.0f} shader cycles per 64-deep K tile (MFMA floor 2048) at {cyc.mean() / ml.mean():.0f} MHz')
    start = t[:, 0, 0][t[:, 0, 0] > 0]
    print(f'  first-tile start spread across workgroups: {start.max() - start.min():.1f} us')


if __name__ == '__main__':
    main()
