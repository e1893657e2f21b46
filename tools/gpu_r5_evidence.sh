#!/bin/bash
# One GPU-box call: the round's profile evidence (tools/gpu_profile_r5.sh), then -- with the fresh traffic file in place, as the
# committed one will be -- the bench line exactly as the driver runs it.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${ROUND:-r05}
bash tools/gpu_profile_r5.sh
cp gpurun_out/${R}_replica_pmc_traffic.json profiles/${R}_replica_pmc_traffic.json
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench.err
echo "bench exit: $?"
python - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r05_bench_line.json") if x.startswith("{")][-1])
r = l["roofline"]
print("value %.3f G  ms/step %.4f  frac_moved %s  frac %.3f  moved B/entry %s  kernel %s" % (l["value"] / 1e9, l["ms_per_step"], r.get("frac_moved"), r["frac"], r.get("moved_bytes_per_entry"), l.get("headline_kernel")))
rk = l.get("replica_kernels", {})
for g, v in rk.get("by_group_size", {}).items():
    print(" N=%s %.3f G frac_moved %s" % (g, v["entries_per_s"] / 1e9, v["roofline"].get("frac_moved")))
for c, v in l.get("other_configs", {}).items():
    if "replica_kernels" in v and "roofline" in v["replica_kernels"]:
        print(" %s %.3f G frac_moved %s" % (c, v["replica_kernels"]["entries_per_s"] / 1e9, v["replica_kernels"]["roofline"].get("frac_moved")))
print(" latency", json.dumps(rk.get("latency", {}))[:400])
print(" host_fed", json.dumps(rk.get("host_fed", {}).get("by_producer_threads", {}))[:300])
print(" redis", json.dumps(l.get("configs0_redis", {}))[:300])
print(" fused %.3f G" % (l.get("fused_step_path", {}).get("value", 0) / 1e9))
PY
tail -5 gpurun_out/${R}_bench.err | cut -c1-300
