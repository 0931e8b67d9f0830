"""Dump per-kernel stats (calls, total, avg, %) from a rocprofv3 rocpd SQLite output directory as CSV."""
import sqlite3, glob, sys, collections
rows = collections.defaultdict(lambda: [0, 0])
for db in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    for name, dur in c.execute("select name, duration from kernels"):
        r = rows[name]
        r[0] += 1; r[1] += dur
tot = sum(v[1] for v in rows.values()) or 1
print('kernel,calls,total_ns,avg_ns,percent')
for name, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f'"{name}",{n},{t},{t / n:.0f},{100.0 * t / tot:.2f}')
