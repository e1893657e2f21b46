#!/bin/bash
# One GPU-box call: the round's profile artifacts for bench.py (per config): rocprofv3 kernel-trace
# statistics and the two PMC passes HBM traffic is computed from.  Output under gpurun_out/prof_<cfg>/;
# the summaries are copied into profiles/ by hand (tools/README.md).
#   CONFIGS="c2 c3 c4" ROUND=r03 bash tools/gpu_profile.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${ROUND:-r03}
mkdir -p gpurun_out
for c in ${CONFIGS:-c2}; do
  d=$PWD/gpurun_out/prof_$c
  rm -rf $d; mkdir -p $d
  ARGS="--config $c --steps 5 --warmup 2 --no-cpu --no-latency --no-ack-path --no-replica --no-other --no-configs0"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $d/kt -o kt -- python $OLDPWD/bench.py $ARGS > $d/kt_run.log 2>&1)
  f=$(find $d/kt -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS"; python tools/kstats.py $f; echo; python tools/kgrid.py $f k_step; } > $d/${R}_${c}_kernel_stats.txt
  tail -1 $d/kt_run.log > $d/${R}_${c}_bench_line_under_rocprof.json
  PARGS="--config $c --steps 2 --warmup 1 --no-cpu --eager --no-latency --no-ack-path --no-replica --no-other --no-configs0"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $d/pmc_$ctr -o pmc -- python $OLDPWD/bench.py $PARGS > $d/pmc_$ctr.log 2>&1)
    echo "pmc $c $ctr exit: $?"
  done
  python tools/mk_traffic.py $c $(find $d/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $d/pmc_WRITE_SIZE -name "*.db" | head -1) $d/pmc_FETCH_SIZE.log gpurun_out/${R}_pmc_traffic.json
  cat $d/${R}_${c}_kernel_stats.txt | head -12
  find $d -name "*.db" -size +20M -delete
done

# the replica kernels (k_replica: one resident launch per run): kernel-trace statistics and the two PMC passes
if [ -n "$REPLICA" ]; then
  d=$PWD/gpurun_out/prof_rep
  rm -rf $d; mkdir -p $d
  RARGS="--grid ${REP_GRID:-192:128} --steps 3 --no-hostfed --brief"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $d/kt -o kt -- python $OLDPWD/tools/rep_bench.py $RARGS > $d/kt_run.log 2>&1)
  f=$(find $d/kt -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python tools/rep_bench.py $RARGS"; python tools/kstats.py $f; echo; tail -1 $d/kt_run.log | cut -c1-600; } > $d/${R}_replica_kernel_stats.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $d/pmc_$ctr -o pmc -- python $OLDPWD/tools/rep_bench.py $RARGS > $d/pmc_$ctr.log 2>&1)
    echo "pmc replica $ctr exit: $?"
    python tools/pmcstats.py $(find $d/pmc_$ctr -name "*.db" | head -1) $ctr k_replica >> $d/${R}_replica_pmc.txt 2>&1
    tail -1 $d/pmc_$ctr.log | cut -c1-300 >> $d/${R}_replica_pmc.txt
  done
  python tools/mk_rep_traffic.py $(find $d/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $d/pmc_WRITE_SIZE -name "*.db" | head -1) $d/pmc_WRITE_SIZE.log gpurun_out/${R}_replica_pmc_traffic.json
  cat $d/${R}_replica_kernel_stats.txt | head -12; cat $d/${R}_replica_pmc.txt
  find $d -name "*.db" -size +20M -delete
fi
