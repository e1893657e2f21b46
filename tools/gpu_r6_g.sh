#!/bin/bash
# Round 6, call G: window words of the request ring (host-fed), one-size flag of the wordless passes (configs[2]), serial roles that
# load as many chunks as there is work (lone-round latency)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_replica.py tests/test_gpu_host_path.py -m gpu -q -x --timeout=600 > $O/tests.txt 2>&1
echo "tests exit: $?"; tail -5 $O/tests.txt
export SWEEP_STEPS=6
timeout 300 python tools/rep_sweep.py "d:3:0:0:0" "d:3:0:0:0" "d:1:0:0:0" "d:5:0:0:0" "d:7:0:0:0" > $O/sweep.txt 2>&1
cut -c1-130 $O/sweep.txt
for c in c3 c4; do timeout 200 python tools/rep_profile_run.py $c; done > $O/c34.txt 2>&1
cat $O/c34.txt | cut -c1-400
for g in "352:30" "448:30" "512:24" "384:48"; do APUS_REP_DEFAULT_APPEND=${g%%:*} APUS_REP_DEFAULT_FWORK=${g##*:} timeout 200 python tools/rep_profile_run.py c3 | cut -c1-300; done > $O/c3_grid.txt 2>&1
for g in "352:21" "448:21" "512:16" "384:32"; do APUS_REP_DEFAULT_APPEND=${g%%:*} APUS_REP_DEFAULT_FWORK=${g##*:} timeout 200 python tools/rep_profile_run.py c4 | cut -c1-300; done >> $O/c3_grid.txt 2>&1
cat $O/c3_grid.txt
timeout 300 python tools/rep_bench.py --steps 2 --grid 0:0 > $O/hostfed.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06_g/hostfed.txt"):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    if d.get("mode")=="host-fed": print(json.dumps(d)[:900])
    else: print(d.get("replicas"), d.get("entries_per_s"), d.get("verified"), d.get("lat_us_p50"), d.get("lat_appended_us_p50"))
PY
