"""Per-kernel durations of a rocprofv3 run of tools/bench_blk.py, grouped by kernel and grid (rocpd database)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'").fetchall()]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
q = f"""select s.kernel_name, d.grid_size_x, d.workgroup_size_x, count(*), avg(d.end-d.start), min(d.end-d.start), min(d.id) from {kd} d join {ks} s on d.kernel_id=s.id
 where s.kernel_name like '%blk3%' or s.kernel_name like '%conv_px%' or s.kernel_name like '%conv_ws%' or s.kernel_name like '%conv_tile%' group by s.kernel_name, d.grid_size_x order by 7"""
for r in cur.execute(q).fetchall():
    name = r[0].replace("_ZN4cgen", "").replace("EEvNS_3B3PE.kd", "").replace("EEvNS_5ConvPENS_3PxPE.kd", "").replace("EEvNS_5ConvPENS_3WsPE.kd", "")
    print("%-40s grid %6d calls %4d avg %7.1f min %7.1f us" % (name[:40], r[1] // r[2], r[3], r[4] / 1e3, r[5] / 1e3))
