"""Compare two parity reports (gpurun_out/parity_report.txt of two runs; profiles/round<N>_parity_report.txt): which margins moved?

    python tools/parity_diff.py OLD NEW [--moved 0.15] [--close 0.85]

Prints (1) every check whose value changed by more than `--moved` (relative), old -> new with the bar, (2) every check of NEW that sits
above `--close` of its bar, (3) checks that exist in only one of the two.  VERDICT r5: "consult the parity report before committing
anything that changes bf16 roundings" -- this is the consultation.  CPU only."""
import re
import sys


def parse(path):
    out = {}
    for ln in open(path):
        m = re.match(r'^(ok  |FAIL) (.*?): (?:rel=|.*?l2-rel=)([0-9.e+-]+).*?\(tol ([0-9.e+-]+)', ln)
        if m:
            out[m.group(2)] = (float(m.group(3)), float(m.group(4)), m.group(1).strip())
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    opt = lambda name, d: float(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d      # noqa: E731
    moved, close = opt('--moved', 0.15), opt('--close', 0.85)
    a, b = parse(args[0]), parse(args[1])
    print(f'{len(a)} checks in {args[0]}, {len(b)} in {args[1]}, {len(set(a) & set(b))} in both')
    print(f'-- moved by more than {moved:.0%} (bf16 checks and anything above 1e-4; old -> new, bar):')
    for k in sorted(set(a) & set(b)):
        (va, ta, _), (vb, tb, sb) = a[k], b[k]
        if max(va, vb) > 1e-4 and abs(vb - va) > moved * max(va, 1e-30):
            print(f'   {k}: {va:.3e} -> {vb:.3e}  (bar {tb:g}; {vb / tb:.0%} of it){"  FAIL" if sb == "FAIL" else ""}')
    print(f'-- above {close:.0%} of the bar in {args[1]}:')
    for k, (v, t, st) in sorted(b.items(), key=lambda kv: -kv[1][0] / kv[1][1]):
        if v > close * t:
            print(f'   {k}: {v:.3e} of {t:g} ({v / t:.0%}){"  FAIL" if st == "FAIL" else ""}')
    only_a, only_b = sorted(set(a) - set(b)), sorted(set(b) - set(a))
    print(f'-- only in {args[0]}: {len(only_a)}; only in {args[1]}: {len(only_b)}')
    for k in only_b[:80]:
        print(f'   new: {k}: {b[k][0]:.3e} (bar {b[k][1]:g})')


if __name__ == '__main__':
    main()
