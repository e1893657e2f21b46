#!/bin/bash
# Round 6, call E: R_SUB 8 + serial roles at high priority; 2 / 3 workgroups per CU
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_e; mkdir -p $O
export SWEEP_STEPS=4
run() { v=$1; shift; APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so timeout 300 python tools/rep_sweep.py "$@" 2>&1; }
{
run p2 "p2:3:0:0:0" "p2:3:0:0:0" "p2:1:0:0:0" "p2:1:0:0:0" "p2:5:0:0:0" "p2:7:0:0:0" "p2:3:224:64:0"
run p3 "p3:3:256:128:0" "p3:3:320:128:0" "p3:3:384:96:0" "p3:3:384:64:0" "p3:3:448:64:0" "p3:3:512:64:0" "p3:1:256:0:0" "p3:1:320:0:0" "p3:1:448:0:0" "p3:1:640:0:0" "p3:5:320:96:0" "p3:5:384:64:0" "p3:7:320:64:0" "p3:7:384:48:0"
run np3 "np3:3:384:96:0" "np3:1:448:0:0"
run s4p3 "s4p3:3:384:96:0" "s4p3:1:448:0:0"
} > $O/sweep.txt 2>&1
run p3 "p3.t:3:384:96:768" "p3.t:1:448:0:768" "p3.t:5:320:96:768" > $O/timers.txt 2>&1
cut -c1-120 $O/sweep.txt
