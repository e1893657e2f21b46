#!/bin/bash
# Round 6, call D: passes without ticket words (the append wavefronts find their pass in the record ring)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_replica.py -m gpu -q -x --timeout=600 > $O/tests.txt 2>&1
echo "tests exit: $?"; tail -5 $O/tests.txt
export SWEEP_STEPS=4
run() { v=$1; shift; APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so timeout 300 python tools/rep_sweep.py "$@" 2>&1; }
{
run g2 "g2:3:0:0:0" "g2:3:0:0:0" "g2:3:0:0:0" "g2.words:3:0:0:128" "g2:1:0:0:0" "g2:1:0:0:0" "g2.words:1:0:0:128" "g2:5:0:0:0" "g2:7:0:0:0" "g2:3:192:96:0" "g2:3:192:64:0" "g2:3:224:64:0" "g2:3:160:128:0"
run g3 "g3:3:0:0:0" "g3:3:256:128:0" "g3:3:320:128:0" "g3:3:384:96:0" "g3:3:448:64:0" "g3:1:256:0:0" "g3:1:320:0:0" "g3:1:448:0:0" "g3:1:640:0:0" "g3:5:320:96:0" "g3:7:320:64:0"
} > $O/sweep.txt 2>&1
run g2 "g2.t:3:0:0:768" "g2.t:1:0:0:768" > $O/timers.txt 2>&1
run g3 "g3.t:3:384:96:768" "g3.t:1:448:0:768" >> $O/timers.txt 2>&1
cut -c1-150 $O/sweep.txt
