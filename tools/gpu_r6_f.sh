#!/bin/bash
# Round 6, call F: the whole GPU suite on the new defaults, host issue A/B, a short bench
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_f; mkdir -p $O
export SWEEP_STEPS=6
{
timeout 300 python tools/rep_sweep.py "batch:3:0:0:0" "batch:3:0:0:0" "batch:1:0:0:0" "batch:1:0:0:0" "batch:5:0:0:0" "batch:7:0:0:0"
SWEEP_ONE_BY_ONE=1 timeout 300 python tools/rep_sweep.py "onebyone:3:0:0:0" "onebyone:1:0:0:0"
} > $O/sweep.txt 2>&1
cut -c1-110 $O/sweep.txt
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=900 > $O/tests.txt 2>&1
echo "tests exit: $?"; tail -8 $O/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit: $?"; tail -3 $O/bench.err; python - <<'PY'
import json
l=json.loads(open("gpurun_out/r06_f/bench.json").read().strip().splitlines()[-1])
rk=l.get("replica_kernels",{})
print("value",l["value"],"ms/step",l["ms_per_step"],"headline",l.get("headline_kernel"))
print("dr", {k:rk.get("device_resident",{}).get(k) for k in ("value","verified","bit_exact_vs_oracle","oracle_check","launch_ms")})
print("by_group", {k:(v.get("entries_per_s"),v.get("verified")) for k,v in rk.get("by_group_size",{}).items()})
print("latency", rk.get("latency"))
print("host_fed", rk.get("host_fed",{}).get("by_producer_threads"))
print("other", {k:{kk:(vv.get("entries_per_s") or vv.get("ms")) for kk,vv in v.items() if isinstance(vv,dict)} for k,v in l.get("other_configs",{}).items()})
print("configs0", l.get("configs0_redis"))
print("cpu", {k:l.get("cpu_baseline",{}).get(k) for k in ("value","kind","cores")})
PY
