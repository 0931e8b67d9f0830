"""What the vendor library reaches on this library's training GEMM shapes, timed beside vtx on the same box (GPU box only;
a MEASUREMENT TOOL -- nothing under videotransformer-pytorch_amd/ calls torch.mm / hipBLASLt / rocBLAS, and this file is not
imported by the product, bench.py or the tests).  SURVEY.md section 7 step 3 / VERDICT r5 item 4.

    python tools/blaslt_ref.py [clips] [--rounds R] [--launches L]

For each of the plain (no fused epilogue) training shapes of TimeSformer-B 8 x 224^2: R rounds of {L launches of vtx, L launches
of torch.mm} INTERLEAVED in one process on one stream, random bf16 operands (the power-limited case: MI355X_MICROARCH.md, DVFS
give-back), HIP events around each group.  NT = C[M,N] = A[M,K] B[N,K]^T (forward / input gradient), TN = C[N1,N2] = A[M,N1]^T
B[M,N2] (weight gradient; vtx adds its fixed-order fp32 slab reduction, torch.mm returns bf16 from the library's own split).
`frac` = FLOP / time / 2.5 PFLOP/s.  Run it under `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` (tools/micro/r6a.sh) for the
shader clock of each kernel: clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from vtx import ops  # noqa: E402

DEV = 'cuda:0'
PEAK_TF = 2500.0


def group(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    return e0, e1


def ab(fa, fb, rounds, launches):
    for _ in range(3):
        fa()
        fb()
    torch.cuda.synchronize()
    ev = []
    for _ in range(rounds):
        ev.append((group(fa, launches), group(fb, launches)))
    torch.cuda.synchronize()
    ta = [a[0].elapsed_time(a[1]) / launches * 1e3 for a, _ in ev]
    tb = [b[0].elapsed_time(b[1]) / launches * 1e3 for _, b in ev]
    med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
    return med(ta), med(tb), min(ta), min(tb)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    B = int(args[0]) if args else 96
    opt = lambda name, d: int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d      # noqa: E731
    rounds, launches = opt('--rounds', 6), opt('--launches', 20)
    T, P, D, Hd = 8, 196, 768, 3072
    Mt, Ms = B * P * T, B * (P * T + 1)
    bf = torch.bfloat16
    r = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(bf)        # noqa: E731
    print(f'clips {B}: {rounds} rounds x {launches} launches per library, interleaved, random bf16 operands; torch {torch.__version__}; '
          f'preferred BLAS backend: {torch.backends.cuda.preferred_blas_library()}')
    print(f'{"kind":3s} {"shape (M x N x K | M: N1 x N2)":32s} {"vtx us":>8s} {"lib us":>8s} {"vtx frac":>8s} {"lib frac":>8s} {"vtx/lib":>7s}  (best of rounds: vtx, lib)')
    nt_shapes = [('qkv fwd', Mt, 3 * D, D), ('qkv dgrad', Mt, D, 3 * D), ('fc1 fwd (plain: no GELU)', Ms, Hd, D), ('fc1 dgrad / fc2 fwd (plain)', Ms, D, Hd),
                 ('proj fwd / dgrad (plain)', Mt, D, D)]
    tot = {'NT': [0.0, 0.0, 0.0], 'TN': [0.0, 0.0, 0.0]}
    for tag, M, N, K in nt_shapes:
        A, W, C, C2 = r(M, K), r(N, K), torch.empty(M, N, device=DEV, dtype=bf), torch.empty(M, N, device=DEV, dtype=bf)
        Wt = W.t()
        tv, tl, bv, bl = ab(lambda: ops.gemm_nt(A, W, C, M, N, K), lambda: torch.mm(A, Wt, out=C2), rounds, launches)
        fl = 2.0 * M * N * K
        err = (C.float() - C2.float()).abs().max().item() / C2.float().abs().max().item()
        print(f'NT  {f"{M}x{N}x{K}  {tag}":32s} {tv:8.1f} {tl:8.1f} {fl / tv / 1e6 / PEAK_TF:8.3f} {fl / tl / 1e6 / PEAK_TF:8.3f} {tv / tl:7.3f}  '
              f'({bv:.1f}, {bl:.1f}; max |vtx - lib| / max |lib| = {err:.1e})', flush=True)
        for i, v in enumerate((tv, tl, fl)):
            tot['NT'][i] += v
        del A, W, C, C2
    tn_shapes = [('fc2 wgrad', Ms, D, Hd), ('fc1 wgrad', Ms, Hd, D), ('qkv wgrad', Mt, 3 * D, D), ('proj wgrad', Mt, D, D)]
    for tag, M, N1, N2 in tn_shapes:
        A, Bm = r(M, N1), r(M, N2)
        o = torch.empty(N1, N2, device=DEV)
        o2 = torch.empty(N1, N2, device=DEV, dtype=bf)
        At = A.t()
        tv, tl, bv, bl = ab(lambda: ops.gemm_tn(A, Bm, M, N1, N2, out=o), lambda: torch.mm(At, Bm, out=o2), rounds, launches)
        fl = 2.0 * M * N1 * N2
        err = (o - o2.float()).abs().max().item() / o.abs().max().item()
        print(f'TN  {f"{M}: {N1}x{N2}  {tag}":32s} {tv:8.1f} {tl:8.1f} {fl / tv / 1e6 / PEAK_TF:8.3f} {fl / tl / 1e6 / PEAK_TF:8.3f} {tv / tl:7.3f}  '
              f'({bv:.1f}, {bl:.1f}; max |vtx - lib| / max |vtx| = {err:.1e}: the library returns bf16)', flush=True)
        for i, v in enumerate((tv, tl, fl)):
            tot['TN'][i] += v
        del A, Bm
    for k, (tv, tl, fl) in tot.items():
        print(f'{k} sum over the shapes above: vtx {tv:.1f} us = {fl / tv / 1e6 / PEAK_TF:.3f}, library {tl:.1f} us = {fl / tl / 1e6 / PEAK_TF:.3f} of 2.5 PFLOP/s')


if __name__ == '__main__':
    main()
