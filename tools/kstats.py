#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 --kernel-trace sqlite database (rocpd)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"""select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3,
        min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
        from {disp} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"""
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
print(f"{'kernel':52s} {'calls':>7s} {'total_us':>12s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'pct':>6s}")
for r in rows:
    print(f"{r[0][:52]:52s} {r[1]:7d} {r[2]:12.1f} {r[3]:9.2f} {r[4]:8.2f} {r[5]:9.2f} {100*r[2]/tot:6.1f}")
if len(sys.argv) > 2:
    # only the big launches of one kernel (grid >= threshold)
    name, thr = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1
    q = f"""select count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
            from {disp} d join {sym} s on d.kernel_id=s.id where s.kernel_name like ? and d.grid_size_x >= ?"""
    print(name, "grid>=", thr, list(cur.execute(q, (f"%{name}%", thr))))
