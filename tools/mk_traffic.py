#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of `python bench.py --config C ...`.
usage: mk_traffic.py CONFIG FETCH.db WRITE.db BENCH_LINE.json OUT.json
The bench line of the same command gives the entries per step and the number of steps executed;
bytes per entry = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over every launch of the kernel,
divided by the entries those launches appended (gfx950: FETCH_SIZE counts half of a wide coalesced
read, WRITE_SIZE is exact -- the guide's HBM section)."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernel_sum(path, counter, like):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    disp, sym, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    q = f"""select count(*), sum(e.value) from {pe} e join {pi} i on e.pmc_id = i.id
            join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id
            where i.name = ? and s.kernel_name like ?"""
    n, tot = list(cur.execute(q, (counter, f"%{like}%")))[0]
    return int(n or 0), float(tot or 0.0)


def main():
    cfg, fdb, wdb, line_path, out = sys.argv[1:6]
    line = json.loads([l for l in open(line_path) if l.startswith("{")][-1])
    kern = line["roofline"]["kernel"]
    nf, fetch_kb = kernel_sum(fdb, "FETCH_SIZE", kern)
    nw, write_kb = kernel_sum(wdb, "WRITE_SIZE", kern)
    entries = line["entries_per_step"] * line["steps_executed"]
    total = (2.0 * fetch_kb + write_kb) * 1024.0
    try:
        doc = json.load(open(out))
    except Exception:
        doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes of "
                         "`python bench.py --config C --steps 2 --warmup 1 --no-cpu --eager --no-latency --no-ack-path` "
                         "(tools/gpu_profile.sh, tools/mk_traffic.py)",
               "correction": "gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> doubled; "
                             "WRITE_SIZE is taken as is",
               "configs": {}}
    doc["configs"][cfg] = {"kernel": kern, "launches": nf, "launches_write_pass": nw,
                           "FETCH_SIZE_KB_total": fetch_kb, "WRITE_SIZE_KB_total": write_kb,
                           "entries": entries, "bytes_total": total, "bytes_per_entry": total / entries,
                           "workload": line["config"]["workload"]}
    from bench import kernel_source_hash
    doc["kernel_source_sha256"] = kernel_source_hash()       # bench.py quotes these counters only for the build they were taken on
    json.dump(doc, open(out, "w"), indent=1)
    print(cfg, doc["configs"][cfg])


if __name__ == "__main__":
    main()
