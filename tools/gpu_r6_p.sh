#!/bin/bash
# Round 6, last call: workgroup grids around the defaults on the final kernels (each twice, interleaved), 3 / 5 / 7 replicas
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_p; mkdir -p $O
export SWEEP_STEPS=8
timeout 1500 python tools/rep_sweep.py "d:3:384:96:0" "a:3:448:96:0" "b:3:384:128:0" "c:3:320:128:0" "e:3:512:64:0" "f:3:416:112:0" \
   "d:3:384:96:0" "a:3:448:96:0" "b:3:384:128:0" "c:3:320:128:0" "e:3:512:64:0" "f:3:416:112:0" \
   "d5:5:384:64:0" "a5:5:320:80:0" "b5:5:448:64:0" "d5:5:384:64:0" "a5:5:320:80:0" "b5:5:448:64:0" \
   "d7:7:384:48:0" "a7:7:320:56:0" "b7:7:448:48:0" "d7:7:384:48:0" "a7:7:320:56:0" "b7:7:448:48:0" > $O/sweep.txt 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/r06_p/sweep.txt"):
    try:
        i=line.index("{"); d=json.loads(line[i:])
    except Exception: print(line[:200]); continue
    print(line[:i], d["Meps"], d["ok"], "lat", d["lat"], d["lat_app"])
PY
