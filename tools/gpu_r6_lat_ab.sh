#!/bin/bash
# Round 6, the lone request's path: A/B of builds (apus_amd/variants/libapus_gpu_<v>.so, tools/build_variants.sh) in ONE call --
#  (1) per build, PASSES times alternating: lone-request / lone-round latencies (request ring behind the BAR), host-fed with 2 producers
#  (2) per build: staged throughput at 3 / 1 / 3 replicas (rep_sweep.py)
#  (3) on the FIRST variant (the candidate): the replica-kernel, host-path and redis end-to-end tests
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lat_ab.txt; : > $O
VARIANTS=${VARIANTS:-new old}
for rep in $(seq 1 ${PASSES:-3}); do for v in $VARIANTS; do
  echo "## $v pass $rep" >> $O
  APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so APUS_REQ_RING=bar timeout 200 python - >> $O 2>&1 <<'PY'
import os, sys, json
import numpy as np
sys.path.insert(0, ".")
from apus_amd import trace as T
from apus_amd.engine import Engine
out = {}
for g in (3,):
    tr = T.steady_trace(g, 1 << 14, 64, 16, 64, log_len=T.DEFAULT_LOG)
    eng = Engine(g, tr.log_len)
    try:
        eng.elect(0); eng.sync()
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        reqs = np.ascontiguousarray(tr.reqs[16:16 + 64])
        h64 = eng.rep_roundtrip_ns(reqs, tr.arena, 400) / 1e3
        h1 = eng.rep_roundtrip_ns(reqs[:1], tr.arena, 600) / 1e3
        h8 = eng.rep_roundtrip_ns(reqs[:8], tr.arena, 300) / 1e3
        eng.rep_drain(); code = eng.rep_park()
        la, ls = eng.rep_latency_appended_ns(), eng.rep_latency_ns()
        out = {"host64_p50": round(float(np.percentile(h64[40:], 50)), 2), "host1_p50": round(float(np.percentile(h1[40:], 50)), 2), "host8_p50": round(float(np.percentile(h8[40:], 50)), 2),
               "host1_p99": round(float(np.percentile(h1[40:], 99)), 2), "seq_to_applied_p50": round(float(np.percentile(ls[20:], 50)) / 1e3, 2),
               "appended_to_applied_p50": round(float(np.percentile(la[20:], 50)) / 1e3, 2), "exit": code}
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        hr0 = eng.rep_highest_rec()
        n, sec = eng.rep_feed(np.ascontiguousarray(tr.reqs[16:16 + 4096]), tr.arena, 2, 0.3, prune_every_reqs=(8 << 20) // 128)
        ok = eng.rep_highest_rec() == hr0 + n
        eng.rep_park()
        out["host_fed2_Meps"] = round(n / sec / 1e6); out["host_fed_ok"] = bool(ok)
    finally:
        eng.close()
print(json.dumps(out))
PY
done; done
for srep in $(seq 1 ${SPASSES:-1}); do for v in $VARIANTS; do
  echo "## $v staged" >> $O
  specs=""; for sp in ${SPECS:-3 1 3 5}; do specs="$specs $v:$sp:0:0:0"; done
  APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so SWEEP_STEPS=8 timeout 300 python tools/rep_sweep.py $specs 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    try:
        i = line.index('{'); d = json.loads(line[i:]); print(line[:i], d['Meps'], d['ok'], d['lat'], d['lat_app'])
    except Exception: print(line[:200].rstrip())
" >> $O
done; done
grep -v "amdgpu.ids\|^W0" $O
if [ -z "$NO_TESTS" ]; then
  set -- $VARIANTS
  APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$1.so timeout 900 python -m pytest tests/test_gpu_replica.py tests/test_gpu_host_path.py tests/test_gpu_e2e_redis.py tests/test_gpu_peers.py -m gpu -q -x --timeout 600 2>&1 | grep -v "^W0\|amdgpu.ids" | tail -6 | cut -c1-300
fi
