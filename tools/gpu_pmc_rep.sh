#!/bin/bash
# One GPU-box call: SQ counters of the replica kernels (are the wavefronts executing or waiting?).  CTRS="A B C" GRID=192:96
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
d=$PWD/gpurun_out/prof_sq; rm -rf $d; mkdir -p $d
RARGS="--grid ${GRID:-192:96} --steps 3 --no-hostfed --brief"
: > gpurun_out/sq_pmc.txt
for set in "${CTRSETS[@]:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY}" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $d/$n -o pmc -- python $OLDPWD/tools/rep_bench.py $RARGS > $d/$n.log 2>&1)
  echo "pmc $set exit: $?" >> gpurun_out/sq_pmc.txt
  for c in $set; do python tools/pmcstats.py $(find $d/$n -name "*.db" | head -1) $c k_replica 2>&1 | grep k_replica >> gpurun_out/sq_pmc.txt; done
  tail -1 $d/$n.log | cut -c1-200 >> gpurun_out/sq_pmc.txt
done
find $d -name "*.db" -size +5M -delete
cat gpurun_out/sq_pmc.txt
