#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_path.py tests/test_gpu_replica.py -m gpu -q -x --timeout=600 > $O/tests.txt 2>&1
echo "tests exit: $?"; tail -5 $O/tests.txt
timeout 300 python tools/rep_bench.py --steps 2 --grid 0:0 > $O/hostfed.txt 2>&1; APUS_REP_DBG=512 timeout 300 python tools/rep_bench.py --steps 2 --grid 0:0 >> $O/hostfed.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06_h/hostfed.txt"):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    if d.get("mode")=="host-fed": print(json.dumps(d)[:900])
    else: print(d.get("replicas"), d.get("entries_per_s"), d.get("verified"), d.get("lat_us_p50"), d.get("lat_appended_us_p50"))
PY
