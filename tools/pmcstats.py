#!/usr/bin/env python3
"""Per-kernel PMC averages from a rocprofv3 --pmc sqlite database (rocpd)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
disp, sym, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
cols = [r[1] for r in cur.execute(f"pragma table_info({pe})")]
icol = [r[1] for r in cur.execute(f"pragma table_info({pi})")]
if "--schema" in sys.argv:
    print(cols, icol)
q = f"""select s.kernel_name, i.name, count(*), avg(e.value), max(e.value), d.grid_size_x
        from {pe} e join {pi} i on e.pmc_id = i.id
        join {disp} d on e.event_id = d.event_id
        join {sym} s on d.kernel_id = s.id
        group by s.kernel_name, i.name, (d.grid_size_x >= 100000) order by 1, 6"""
for r in cur.execute(q):
    print(f"{r[0][:44]:44s} {r[1]:12s} n={r[2]:5d} avg={r[3]:14.1f} max={r[4]:14.1f} grid_x~{r[5]}")
