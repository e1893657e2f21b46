#!/bin/bash
# Round 6, call B: the new sequencer (passes of up to 4096 rounds) at 2 / 3 / 4 workgroups per CU, R_SUB 4 / 8, grids swept.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_replica.py -m gpu -q -x --timeout=600 > $O/tests.txt 2>&1
echo "tests exit: $?"; tail -5 $O/tests.txt
export SWEEP_STEPS=4
run() { v=$1; shift; APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so timeout 300 python tools/rep_sweep.py "$@" 2>&1 | cut -c1-330; }
{
run w2s4 "w2s4:3:0:0:0" "w2s4:3:0:0:0" "w2s4:3:0:0:0" "w2s4:1:0:0:0" "w2s4:1:0:0:0" "w2s4:5:0:0:0" "w2s4:7:0:0:0"
run w3s4 "w3s4:3:0:0:0" "w3s4:3:0:0:0" "w3s4:3:256:128:0" "w3s4:3:320:128:0" "w3s4:3:384:128:0" "w3s4:3:384:96:0" "w3s4:3:448:96:0" "w3s4:1:320:0:0" "w3s4:1:448:0:0" "w3s4:1:640:0:0"
run w3s8 "w3s8:3:0:0:0" "w3s8:3:256:128:0" "w3s8:3:320:128:0" "w3s8:3:384:128:0" "w3s8:3:384:96:0" "w3s8:3:448:96:0" "w3s8:1:320:0:0" "w3s8:1:448:0:0" "w3s8:1:640:0:0" "w3s8:5:320:96:0" "w3s8:7:320:64:0"
run w4s8 "w4s8:3:0:0:0" "w4s8:3:320:128:0" "w4s8:3:448:128:0" "w4s8:3:512:128:0" "w4s8:3:640:128:0" "w4s8:3:640:96:0" "w4s8:1:448:0:0" "w4s8:1:640:0:0" "w4s8:1:896:0:0" "w4s8:5:448:96:0" "w4s8:7:448:64:0"
} > $O/sweep.txt 2>&1
run w3s8 "w3s8.t:3:384:128:768" "w3s8.t:1:448:0:768" > $O/timers.txt 2>&1
cut -c1-150 $O/sweep.txt
cut -c1-1600 $O/timers.txt
