#!/usr/bin/env python3
"""Per-kernel, per-grid-size durations from a rocprofv3 --kernel-trace sqlite database (rocpd):
the launches of k_step differ in the number of segments they carry."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
like = sys.argv[2] if len(sys.argv) > 2 else "k_step"
q = f"""select s.kernel_name, d.grid_size_x, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
        from {disp} d join {sym} s on d.kernel_id=s.id where s.kernel_name like ? group by s.kernel_name, d.grid_size_x order by 2"""
print(f"{'kernel':40s} {'grid_x':>9s} {'calls':>6s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s}")
for r in cur.execute(q, (f"%{like}%",)):
    print(f"{r[0][:40]:40s} {r[1]:9d} {r[2]:6d} {r[3]:9.2f} {r[4]:8.2f} {r[5]:9.2f}")
