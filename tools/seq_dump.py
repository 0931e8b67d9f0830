"""Print the in-order durations of one kernel family within one training step (from a rocpd DB)."""
import sqlite3, glob, sys
pat = sys.argv[2]
for db in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, duration from kernels order by start").fetchall()
    sel = [(n, s, d) for (n, s, d) in rows if pat in n]
    per_step = int(sys.argv[3])
    last = sel[-per_step:]
    print(' '.join(f'{d/1000:.0f}' for (_, _, d) in last))
