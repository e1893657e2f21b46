#!/bin/bash
# Round 6, the evidence call: on ONE box, one build --
#  (1) k_step's kernel trace + PMC passes (tools/gpu_profile.sh; the fused step path's roofline)
#  (2) the replica kernels': kernel trace around the bench's command, PMC passes per configuration of the line, SQ counters
#      (tools/gpu_profile_rep.sh)
#  (3) with both traffic files in place as they will be committed: the bench line exactly as the driver runs it
#  (4) the GPU suite
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r06
ROUND=$R CONFIGS=c2 bash tools/gpu_profile.sh > gpurun_out/${R}_kstep_profile.log 2>&1
cp gpurun_out/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json
cp gpurun_out/prof_c2/${R}_c2_kernel_stats.txt gpurun_out/ 2>/dev/null
ROUND=$R SQ=1 bash tools/gpu_profile_rep.sh > gpurun_out/${R}_rep_profile.log 2>&1
cp gpurun_out/${R}_replica_pmc_traffic.json profiles/${R}_replica_pmc_traffic.json
tail -20 gpurun_out/${R}_rep_profile.log | cut -c1-200
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench.err
echo "bench exit: $?"
python - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r06_bench_line.json") if x.startswith("{")][-1])
r = l["roofline"]
print("value %.3f G  ms/step %.4f  frac_moved %s  frac %.3f  survey %.3f moved B/entry %s  kernel %s" % (l["value"] / 1e9, l["ms_per_step"], r.get("frac_moved"), r["frac"], r.get("frac_survey_formula") or 0, r.get("moved_bytes_per_entry"), l.get("headline_kernel")))
rk = l.get("replica_kernels", {})
print(" bit exact:", rk.get("device_resident", {}).get("bit_exact_vs_oracle"), rk.get("device_resident", {}).get("oracle_check"))
for g, v in rk.get("by_group_size", {}).items():
    print(" N=%s %.3f G frac_moved %s verified %s" % (g, v["entries_per_s"] / 1e9, v["roofline"].get("frac_moved"), v.get("verified")))
for c, v in l.get("other_configs", {}).items():
    if "replica_kernels" in v and "roofline" in v["replica_kernels"]:
        print(" %s %.3f G frac_moved %s" % (c, v["replica_kernels"]["entries_per_s"] / 1e9, v["replica_kernels"]["roofline"].get("frac_moved")))
print(" latency", json.dumps(rk.get("latency", {}))[:500])
print(" host_fed", json.dumps(rk.get("host_fed", {}).get("by_producer_threads", {}))[:400])
print(" redis", json.dumps(l.get("configs0_redis", {}))[:300])
print(" cpu", json.dumps({k: l.get("cpu_baseline", {}).get(k) for k in ("value", "kind", "cores", "configs0")})[:300])
print(" fused %.3f G" % (l.get("fused_step_path", {}).get("value", 0) / 1e9))
PY
tail -3 gpurun_out/${R}_bench.err | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/${R}_gpu_tests.txt 2>&1
echo "gpu suite exit: $?"; tail -4 gpurun_out/${R}_gpu_tests.txt
