#!/bin/bash
# Where a producer's time goes (APUS_FEED_PROF=1: phase clocks inside apus_gpu_rep_submit, include/apus_gpu.h): 1 / 2 / 4 / 8 producers,
# 0.4 s each, blocks of 4096 requests per call = 16 blocks of 256 slots, 64-byte payloads, three replicas
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
APUS_FEED_PROF=1 timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_feed_profile.txt
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, ".")
from apus_amd import trace as T
from apus_amd.engine import Engine
tr = T.steady_trace(3, 1 << 14, 64, 16, 64, log_len=T.DEFAULT_LOG)
eng = Engine(3, tr.log_len)
names = ["reserve", "stores", "fence", "slot_words", "windows", "last_fence"]
try:
    eng.elect(0); eng.sync()
    blk = np.ascontiguousarray(tr.reqs[16:16 + 4096])
    for nt in (1, 2, 4, 8):
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        hr0 = eng.rep_highest_rec()
        out = (C.c_uint64 * 9)()
        eng.L.apus_gpu_rep_feed_profile(eng.h, out)
        n, sec = eng.rep_feed(blk, tr.arena, nt, 0.4, prune_every_reqs=(8 << 20) // 128)
        ok = eng.rep_highest_rec() == hr0 + n
        eng.L.apus_gpu_rep_feed_profile(eng.h, out)
        eng.rep_park()
        tpu = float(out[8]); blocks = max(1, int(out[0]))
        per = {k: round(int(out[2 + i]) / tpu / blocks * 1000) for i, k in enumerate(names)}        # ns per block of 256 slots, per producer
        tot = sum(per.values())
        print(json.dumps({"producers": nt, "M_entries_per_s": round(n / sec / 1e6), "ok": bool(ok), "slots_per_block": round(int(out[1]) / blocks),
                          "ns_per_block": per, "ns_per_block_total": tot, "share": {k: round(v / tot, 2) for k, v in per.items()},
                          "GBps_of_128B_slots_per_producer": round(int(out[1]) / blocks * 128 / tot, 2), "tsc_per_us": round(tpu)}))
finally:
    eng.close()
PY
done
