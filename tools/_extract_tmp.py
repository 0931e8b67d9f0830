import sqlite3, glob, sys
for db in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print(cols)
    q = "select * from kernels order by start"
    rows = c.execute(q).fetchall()
    for r in rows:
        d = dict(zip(cols, r))
        if 'attn' in str(d.get('name', '')):
            print({k: d[k] for k in cols if k in ('name','duration','grid_x','grid_y','workgroup_x','lds_size','scratch_size','vgpr_count','accum_vgpr_count','sgpr_count','lds_block_size','private_segment_size','group_segment_size')} , d.get('end',0)-d.get('start',0))
