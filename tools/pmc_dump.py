"""Print per-kernel PMC counter totals from a rocprofv3 rocpd database directory.

    python tools/pmc_dump.py <dir> [name,filters] [--clock]

Second argument: comma-separated substrings a kernel name must contain one of (default: the vtx GEMM / attention / LayerNorm /
wprod kernels).  With --clock and a GRBM_GUI_ACTIVE pass collected together with --kernel-trace: the shader clock of every kernel
= GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the kernel's duration, when the database carries dispatch times."""
import sqlite3, glob, sys, collections
args = [a for a in sys.argv[1:] if not a.startswith('--')]
filters = args[1].split(',') if len(args) > 1 else ['gemm', 'attn', 'ln_', 'wprod']
for db in glob.glob(args[0] + '/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
    view = 'counters_collection' if 'counters_collection' in tabs else None
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    kn = 'kernel_name' if 'kernel_name' in cols else 'name'
    has_t = 'start' in cols and 'end' in cols
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
    q = f"select {kn}, counter_name, value" + (", end - start" if has_t else ", 0") + f" from {view}"
    for name, cn, v, dur in c.execute(q):
        if not any(k in name for k in filters):
            continue
        short = name.split('(')[0]
        short = short[short.index('vtx::') + 5:] if 'vtx::' in short else short      # kernel name with its template arguments
        a = acc[(short[-72:], cn)]
        a[0] += v; a[1] += 1; a[2] += dur
    # values are summed over dimensions per dispatch row; report mean per dispatch
    for (k, cn), (s, n, d) in sorted(acc.items()):
        extra = ''
        if '--clock' in sys.argv and cn == 'GRBM_GUI_ACTIVE' and d > 0:
            extra = f' avg_us={d / n / 1e3:.1f} clock_GHz={s / 8.0 / d:.3f}'
        print(f'{k:74s} {cn:32s} total={s:.4g} rows={n}{extra}')
    if '--schema' in sys.argv:
        print('# counters_collection columns:', ', '.join(cols))
