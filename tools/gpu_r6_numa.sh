#!/bin/bash
# Round 6: the request path from the device's own NUMA node against wherever the scheduler puts the threads (two-socket hosts):
# lone-request round trips and host-fed throughput, unbound and bound (apus_gpu_bind_near), three processes each, interleaved
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_numa; mkdir -p $O; : > $O/numa.txt
for rep in 1 2 3; do for bind in ${MODES:-0 1}; do
BIND=$bind APUS_FEED_PIN=$( [ $bind = 2 ] && echo node || echo 0 ) timeout 300 python - <<'PY' >> $O/numa.txt 2>&1
import os, sys, json
import numpy as np
sys.path.insert(0, ".")
from apus_amd import trace as T
from apus_amd.engine import Engine
bind = int(os.environ["BIND"])
tr = T.steady_trace(3, 1 << 14, 64, 16, 64, log_len=T.DEFAULT_LOG)
eng = Engine(3, tr.log_len)
try:
    node = eng.L.apus_gpu_numa_node(0)
    rc = eng.L.apus_gpu_bind_near(eng.h, 1) if bind == 1 else -1
    eng.elect(0); eng.sync()
    eng.rep_start(idle_ms=5000, peer_ms=1000)
    reqs = np.ascontiguousarray(tr.reqs[16:16 + 64])
    h64 = eng.rep_roundtrip_ns(reqs, tr.arena, 400) / 1e3
    h1 = eng.rep_roundtrip_ns(reqs[:1], tr.arena, 400) / 1e3
    eng.rep_drain(); eng.rep_park()
    out = {"bind": bind, "rc": rc, "gpu_node": node, "cpus": len(os.sched_getaffinity(0)), "host1_p50": round(float(np.percentile(h1[40:], 50)), 2), "host1_p99": round(float(np.percentile(h1[40:], 99)), 2),
           "host64_p50": round(float(np.percentile(h64[40:], 50)), 2), "fed": {}}
    big = np.ascontiguousarray(tr.reqs[16:16 + 4096])
    for nt in (1, 2, 4, 8):
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        hr0 = eng.rep_highest_rec()
        n, sec = eng.rep_feed(big, tr.arena, nt, 0.4, prune_every_reqs=(8 << 20) // 128)
        ok = eng.rep_highest_rec() == hr0 + n
        eng.rep_park()
        out["fed"][nt] = [round(n / sec / 1e6), bool(ok)]
    print(json.dumps(out))
finally:
    eng.close()
PY
done; done
grep "^{" $O/numa.txt; grep -v "^{" $O/numa.txt | grep -v amdgpu.ids | tail -5
