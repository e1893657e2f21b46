#!/usr/bin/env python3
"""Duration percentiles per kernel from a rocprofv3 kernel-trace rocpd database."""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select s.kernel_name, (d.end-d.start)/1e3, d.start, d.end from {disp} d join {sym} s on d.kernel_id=s.id order by d.start"))
by = {}
for n, d, st, en in rows: by.setdefault(n[:40], []).append(d)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    print(f"{n:40s} n={len(v):5d} sum={v.sum():9.1f} p10={np.percentile(v,10):7.2f} p50={np.percentile(v,50):7.2f} p90={np.percentile(v,90):7.2f} max={v.max():7.2f}")
# gaps between consecutive kernels (same process), over the steady part
st = np.array([r[2] for r in rows]); en = np.array([r[3] for r in rows])
gaps = (st[1:] - en[:-1]) / 1e3
g = gaps[(gaps > 0) & (gaps < 50)]
print(f"inter-kernel gaps: n={len(g)} p50={np.percentile(g,50):.2f} p90={np.percentile(g,90):.2f} mean={g.mean():.2f} us")
