#!/bin/bash
# `python bench.py` with no flags (what a default run is) on a fresh box: its wall clock and the line's main figures
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1000 python bench.py > gpurun_out/r06_bench_line_default_flags.json 2> gpurun_out/r06_bench_default.err
echo "exit $? wall $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r06_bench_line_default_flags.json") if x.startswith("{")][-1])
r = l["roofline"]; rk = l["replica_kernels"]
print("value %.3f G ms/step %.4f steps %s frac_moved %.3f bit_exact %s" % (l["value"]/1e9, l["ms_per_step"], l["steps"], r["frac_moved"], rk["device_resident"].get("bit_exact_vs_oracle")))
print({g: round(v["entries_per_s"]/1e9, 2) for g, v in rk["by_group_size"].items()}, rk["latency"]["host_submit_to_highest_rec_us_p50_1_entry"], {k: round(v["entries_per_s"]/1e6) for k, v in rk["host_fed"]["by_producer_threads"].items()})
print(l["configs0_redis"]["requests_per_s"], l["cpu_baseline"]["configs0"]["requests_per_s"])
PY
