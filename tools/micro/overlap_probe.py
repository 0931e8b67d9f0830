"""Small batches: does running a layer's weight-gradient GEMM (TN) BESIDE its input-gradient GEMM (NT) on a second stream fill the CUs
the under-filled launches leave idle?  (the 8-clip row.)  Both consume the same `dout`; they are independent.

    python tools/micro/overlap_probe.py [clips ...]          # default 8 32 96
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from vtx import ops
    dev = torch.device('cuda', 0)
    clips = [int(v) for v in sys.argv[1:]] or [8, 32, 96]
    side = torch.cuda.Stream()
    shapes = [('proj', 768, 768), ('qkv', 2304, 768), ('fc1', 3072, 768), ('fc2', 768, 3072)]       # (name, N_out, K_in) of y = x W^T
    for nc in clips:
        M = nc * 1568
        print(f'{nc} clips (M = {M}); us per pair: input-gradient NT [M x K_in] = dY [M x N_out] W, weight-gradient TN [N_out x K_in] = dY^T X')
        tot = [0.0, 0.0, 0.0, 0.0]
        for name, no, ki in shapes:
            dy = torch.randn(M, no, device=dev).bfloat16()
            x = torch.randn(M, ki, device=dev).bfloat16()
            wt = torch.randn(ki, no, device=dev).bfloat16()          # W^T: the B operand of the NT launch
            dx = torch.empty(M, ki, device=dev, dtype=torch.bfloat16)
            dw = torch.empty(no, ki, device=dev, dtype=torch.float32)

            def nt():
                ops.gemm_nt(dy, wt, dx, M, ki, no)

            def tn():
                ops.gemm_tn(dy, x, M, no, ki, out=dw)

            def serial():
                nt(); tn()

            def overlapped():
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    tn()
                    ev2 = torch.cuda.Event(); ev2.record()
                nt()
                torch.cuda.current_stream().wait_event(ev2)

            def timeit(fn, n=40):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record(); e1.synchronize()
                return e0.elapsed_time(e1) / n * 1e3
            r = [timeit(nt), timeit(tn), timeit(serial), timeit(overlapped)]
            for i in range(4):
                tot[i] += r[i]
            print(f'   {name:5s} N_out {no:4d} K_in {ki:4d}: NT alone {r[0]:7.1f}  TN alone {r[1]:7.1f}  serial {r[2]:7.1f}  two streams {r[3]:7.1f}  ({r[3] / r[2]:.3f})')
        print(f'   sum  : NT {tot[0]:7.1f}  TN {tot[1]:7.1f}  serial {tot[2]:7.1f}  two streams {tot[3]:7.1f}  ({tot[3] / tot[2]:.3f})')


if __name__ == '__main__':
    main()
