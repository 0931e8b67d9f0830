"""Print per-kernel PMC counter means from a rocprofv3 rocpd database directory."""
import sqlite3, glob, sys, collections
for db in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
    view = 'counters_collection' if 'counters_collection' in tabs else None
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    kn = 'kernel_name' if 'kernel_name' in cols else 'name'
    acc = collections.defaultdict(lambda: [0.0, 0])
    for name, cn, v in c.execute(f"select {kn}, counter_name, value from {view}"):
        if not any(k in name for k in ('gemm', 'attn', 'ln_', 'wprod')):
            continue
        short = name.split('(')[0]
        short = short[short.index('vtx::') + 5:] if 'vtx::' in short else short      # kernel name with its template arguments
        a = acc[(short[-72:], cn)]
        a[0] += v; a[1] += 1
    # values are summed over dimensions per dispatch row; report mean per dispatch
    disp = collections.defaultdict(int)
    for (k, cn), (s, n) in sorted(acc.items()):
        print(f'{k:74s} {cn:32s} total={s:.4g} rows={n}')
