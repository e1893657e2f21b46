#!/bin/bash
# Round 6, call A: (1) the new parity tests of the replica kernels, (2) the N = 3 regression bisect: the same sweep on
# library variants built from this tree (tools/build_variants.sh), several launches each, in ONE call on ONE box.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_a
mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/box.txt
timeout 1500 python -m pytest tests/test_gpu_replica.py -m gpu -q -x --timeout=600 \
   -k "every_group_size or single_replica_host_fed or bench_shaped or random_traces or doorbell_metadata_against or inherited_entries" > $O/new_tests.txt 2>&1
echo "new tests exit: $?"; tail -15 $O/new_tests.txt
export SWEEP_STEPS=4
for v in head rwin8 ackb nopad wg3 head; do
  for i in 1 2 3; do
    APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so timeout 120 python tools/rep_sweep.py "$v.$i:3:0:0:0" "$v.$i:1:0:0:0" 2>&1 | cut -c1-260
  done
done > $O/bisect.txt 2>&1
APUS_GPU_LIB=apus_amd/variants/libapus_gpu_head.so timeout 120 python tools/rep_sweep.py "head.timers:3:0:0:768" "head.timers:1:0:0:768" > $O/timers.txt 2>&1
cut -c1-200 $O/bisect.txt
cat $O/timers.txt | cut -c1-1500
