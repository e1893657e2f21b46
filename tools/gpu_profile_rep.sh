#!/bin/bash
# One GPU-box call: the round's profile evidence for the replica kernels.
#  (1) rocprofv3 --kernel-trace --stats around the SAME command the bench line's headline comes from (bench.py --steps K):
#      the line printed under the profiler and every k_replica* dispatch with its duration side by side
#  (2) per configuration of the line (CFGS): two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of ONE resident launch
#      (tools/rep_profile_run.py) -> gpurun_out/${R}_replica_pmc_traffic.json (tools/mk_rep_traffic.py)
#  (3) SQ=1: the SQ counter sets on configs[1] at 3 replicas
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${ROUND:-r06}
d=$PWD/gpurun_out/prof_$R
rm -rf $d; mkdir -p $d
HERE=$PWD
if [ -z "$NO_KT" ]; then
  ARGS="--steps ${KT_STEPS:-20} --warmup 5 --no-cpu --no-latency --no-ack-path --no-other --no-configs0"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $d/kt -o kt -- python $HERE/bench.py $ARGS > $d/kt_run.log 2> $d/kt_run.err)
  f=$(find $d/kt -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS"; python tools/kstats.py $f | head -14; echo; echo "# every resident launch of that run"; python tools/klaunches.py $f k_replica; echo; echo "# the line that run printed (roofline of the headline launch)";
    python - $d/kt_run.log <<'PY'
import json, sys
l = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print(json.dumps({"value": l["value"], "ms_per_step": l["ms_per_step"], "roofline": l["roofline"],
                  "by_group_size": {k: {kk: v.get(kk) for kk in ("entries_per_s", "launch_ms", "entries_in_launch")} for k, v in l.get("replica_kernels", {}).get("by_group_size", {}).items()}}, indent=1))
PY
  } > $d/${R}_replica_kernel_stats.txt 2>&1
  tail -1 $d/kt_run.log > $d/${R}_bench_line_under_rocprof.json
  head -30 $d/${R}_replica_kernel_stats.txt | cut -c1-200
fi
[ -n "$NO_PMC" ] && CFGS=" "
for c in ${CFGS-c2x3 c2x1 c2x5 c2x7 c3 c4}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d $d/pmc_${c}_$ctr -o pmc -- python $HERE/tools/rep_profile_run.py $c > $d/pmc_${c}_$ctr.log 2>&1)
    echo "pmc $c $ctr exit: $?"
  done
  python tools/mk_rep_traffic.py $c $(find $d/pmc_${c}_FETCH_SIZE -name "*.db" | head -1) $(find $d/pmc_${c}_WRITE_SIZE -name "*.db" | head -1) \
      $d/pmc_${c}_FETCH_SIZE.log $d/pmc_${c}_WRITE_SIZE.log gpurun_out/${R}_replica_pmc_traffic.json 2>&1 | tail -2
done
if [ -n "$SQ" ]; then
  : > gpurun_out/${R}_replica_sq_counters.txt
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES"; do
    n=$(echo $set | tr ' ' '_' | cut -c1-40)
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $d/sq_$n -o pmc -- python $HERE/tools/rep_profile_run.py c2x3 > $d/sq_$n.log 2>&1)
    echo "# $set  ($(tail -1 $d/sq_$n.log | cut -c1-160))" >> gpurun_out/${R}_replica_sq_counters.txt
    for c in $set; do python tools/pmcstats.py $(find $d/sq_$n -name "*.db" | head -1) $c k_replica 2>&1 | grep k_replica >> gpurun_out/${R}_replica_sq_counters.txt; done
  done
  cat gpurun_out/${R}_replica_sq_counters.txt | cut -c1-200
fi
cp $d/${R}_replica_kernel_stats.txt $d/${R}_bench_line_under_rocprof.json gpurun_out/ 2>/dev/null
find $d -name "*.db" -size +8M -delete
