#!/usr/bin/env python3
"""Every dispatch of the kernels whose name contains NAME in a rocprofv3 --kernel-trace database (rocpd): grid, duration.
usage: klaunches.py DB NAME"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"""select s.kernel_name, d.grid_size_x, d.workgroup_size_x, (d.end - d.start) / 1e3, d.start
        from {disp} d join {sym} s on d.kernel_id = s.id where s.kernel_name like ? order by d.start"""
rows = list(cur.execute(q, (f"%{sys.argv[2]}%",)))
t0 = rows[0][4] if rows else 0
print(f"{'kernel':40s} {'workgroups':>10s} {'duration_us':>14s} {'start_ms':>10s}")
for name, gx, wx, us, st in rows:
    print(f"{name[:40]:40s} {gx // max(wx, 1):10d} {us:14.2f} {(st - t0) / 1e6:10.1f}")
