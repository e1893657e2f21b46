#!/bin/bash
# One short GPU call: the request path's latency (host submit -> highest_rec) with the request ring behind the BAR and in pinned
# host memory, the pinned-source replica tests, the redis end-to-end test
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for m in bar host; do
  APUS_REQ_RING=$m timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import os, sys, json
import numpy as np
sys.path.insert(0, ".")
from apus_amd import trace as T
from apus_amd.engine import Engine
out = {"ring": os.environ["APUS_REQ_RING"]}
for g in (3,):
    tr = T.steady_trace(g, 1 << 14, 64, 16, 64, log_len=T.DEFAULT_LOG)
    eng = Engine(g, tr.log_len)
    try:
        eng.elect(0); eng.sync()
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        reqs = np.ascontiguousarray(tr.reqs[16:16 + 64])
        h64 = eng.rep_roundtrip_ns(reqs, tr.arena, 400) / 1e3
        h1 = eng.rep_roundtrip_ns(reqs[:1], tr.arena, 400) / 1e3
        eng.rep_drain(); code = eng.rep_park()
        la, ls = eng.rep_latency_appended_ns(), eng.rep_latency_ns()
        out[str(g)] = {"kind": eng.rep_req_ring_kind(), "host64_p50": float(np.percentile(h64[40:], 50)), "host1_p50": float(np.percentile(h1[40:], 50)),
                       "host1_p99": float(np.percentile(h1[40:], 99)), "seq_to_applied_p50": float(np.percentile(ls[20:], 50)) / 1e3,
                       "appended_to_applied_p50": float(np.percentile(la[20:], 50)) / 1e3, "exit": code}
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        hr0 = eng.rep_highest_rec()
        n, sec = eng.rep_feed(np.ascontiguousarray(tr.reqs[16:16 + 4096]), tr.arena, 2, 0.3, prune_every_reqs=(8 << 20) // 128)
        ok = eng.rep_highest_rec() == hr0 + n
        eng.rep_park()
        out[str(g)]["host_fed_Meps"] = n / sec / 1e6; out[str(g)]["host_fed_ok"] = bool(ok)
    finally:
        eng.close()
print(json.dumps(out))
PY
done
[ -n "$LAT_ONLY" ] || timeout 400 python -m pytest tests/test_gpu_replica.py tests/test_gpu_e2e_redis.py tests/test_gpu_host_path.py -m gpu -q -x --timeout 300 -s 2>&1 | grep -v "^W0\|amdgpu.ids" | grep -i "passed\|failed\|error\|SET\|req/s\|k/s" | tail -12 | cut -c1-300
