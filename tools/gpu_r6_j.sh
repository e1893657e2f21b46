#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_path.py tests/test_gpu_replica.py -m gpu -q -x --timeout=600 > $O/tests.txt 2>&1
echo "tests exit: $?"; tail -5 $O/tests.txt
export SWEEP_STEPS=6
timeout 300 python tools/rep_sweep.py "q:3:0:0:0" "q:3:0:0:0" "q:3:0:0:0" "q:1:0:0:0" "q:1:0:0:0" "q:5:0:0:0" "q:7:0:0:0" "q.t:3:0:0:768" "q.t:1:0:0:768" > $O/sweep.txt 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/r06_j/sweep.txt"):
    try:
        i=line.index("{"); d=json.loads(line[i:])
    except Exception: print(line[:300]); continue
    print(line[:i], d["Meps"], d["ok"], "lat",d["lat"],d["lat_app"], "seq_us", d["seq_us"])
    if "dbg=768" in line:
        for k in ("seq","com","app","f0r","f0a","append","f0_work","seq_more","seq_prune_us"): print("   ",k,d.get(k))
PY
