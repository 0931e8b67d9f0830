"""Does a weight-gradient GEMM on a side stream fill the tail of the input-gradient GEMM that runs beside it?
One layer's GEMM pairs (qkv, fc1, fc2: dgrad = vtx_gemm_nt, wgrad = vtx_gemm_tn, same dY) back to back, wgrad on the
launch stream vs on a second stream that waits for dY.  Prints time per pair sequence."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch

import vtx  # noqa: F401
from vtx import ops

DEV = torch.device('cuda', 0)
bf = torch.bfloat16


def main(B=96):
    M = B * 1569
    r = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(bf)            # noqa: E731
    pairs = []
    for (N, K) in ((2304, 768), (3072, 768), (768, 3072), (2304, 768), (768, 768)):
        dY, X, WT = r(M, N), r(M, K), r(K, N)
        dX = torch.empty(M, K, device=DEV, dtype=bf)
        dW = torch.empty(N, K, device=DEV)
        pairs.append((N, K, dY, X, WT, dX, dW))
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()

    def run(two):
        for (N, K, dY, X, WT, dX, dW) in pairs:
            if two:
                side.wait_stream(main_s)
                with torch.cuda.stream(side):
                    ops.gemm_tn(dY, X, M, N, K, out=dW)
                ops.gemm_nt(dY, WT, dX, M, K, N)
            else:
                ops.gemm_tn(dY, X, M, N, K, out=dW)
                ops.gemm_nt(dY, WT, dX, M, K, N)
        if two:
            main_s.wait_stream(side)

    for two in (False, True, False, True):
        for _ in range(2):
            run(two)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            run(two)
        e1.record()
        torch.cuda.synchronize()
        print(f'{"two streams" if two else "one stream ":12s}: {e0.elapsed_time(e1) / 6 * 1e3:9.1f} us per sequence of {len(pairs)} (dgrad, wgrad) pairs', flush=True)


if __name__ == '__main__':
    main()
