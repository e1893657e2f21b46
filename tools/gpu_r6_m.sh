#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_m; mkdir -p $O
export SWEEP_STEPS=6
timeout 600 python tools/rep_sweep.py "m:3:0:0:0" "m:3:0:0:0" "m:1:0:0:0" "m.t:3:0:0:512" "m.t:1:0:0:512" "m.t:5:0:0:512" > $O/sweep.txt 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/r06_m/sweep.txt"):
    try:
        i=line.index("{"); d=json.loads(line[i:])
    except Exception: print(line[:300]); continue
    print(line[:i], d["Meps"], d["ok"], "lat",d["lat"],d["lat_app"], "seq_us", d["seq_us"])
    if "dbg=512" in line:
        for k in ("seq","com","app","f0r","f0a","seq_more","seq_prune_us"): print("   ",k,d.get(k))
PY
