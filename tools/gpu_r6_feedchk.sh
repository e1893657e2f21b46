#!/bin/bash
# the bench's host-fed leg alone, several times: 1 / 2 / 4 / 8 producers, 0.4 s each, blocks of 4096 requests (rep_feed's default placement)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in $(seq 1 ${RUNS:-6}); do
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -4
import os, sys, json
import numpy as np
sys.path.insert(0, ".")
from apus_amd import trace as T
from apus_amd.engine import Engine, EngineError
tr = T.steady_trace(3, 1 << 14, 64, 16, 64, log_len=T.DEFAULT_LOG)
eng = Engine(3, tr.log_len)
out = {}
try:
    eng.elect(0); eng.sync()
    blk = np.ascontiguousarray(tr.reqs[16:16 + 4096])
    for nt in (1, 2, 4, 8):
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        hr0 = eng.rep_highest_rec()
        try:
            n, sec = eng.rep_feed(blk, tr.arena, nt, float(os.environ.get("SECS", "0.4")), prune_every_reqs=(8 << 20) // 128)
            good = eng.rep_highest_rec() == hr0 + n
            out[nt] = [round(n / sec / 1e6), bool(good)]
        except EngineError as ex:
            w = (C := __import__("ctypes")).c_uint32 * 8; words = w()
            eng.L.apus_gpu_status_words(eng.h, words)
            out[nt] = ["FAILED", str(ex)[:120], [hex(x) for x in words]]
            break
        code = eng.rep_park()
    print(json.dumps(out))
finally:
    eng.close()
PY
done
