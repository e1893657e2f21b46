#!/bin/bash
# Round 6, call C: chunk summaries (closer wavefronts; committer / applier by chunk) on top of the new sequencer.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_replica.py -m gpu -q -x --timeout=600 > $O/tests.txt 2>&1
echo "tests exit: $?"; tail -5 $O/tests.txt
export SWEEP_STEPS=4
run() { v=$1; shift; APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so timeout 300 python tools/rep_sweep.py "$@" 2>&1 | cut -c1-330; }
{
run c2 "c2:3:0:0:0" "c2:3:0:0:0" "c2:3:0:0:0" "c2.nochunk:3:0:0:65536" "c2:1:0:0:0" "c2:1:0:0:0" "c2.nochunk:1:0:0:65536" "c2:5:0:0:0" "c2:7:0:0:0"
run c3 "c3:3:0:0:0" "c3:3:256:128:0" "c3:3:320:128:0" "c3:3:384:128:0" "c3:3:384:96:0" "c3:3:448:96:0" "c3:1:320:0:0" "c3:1:448:0:0" "c3:1:640:0:0" "c3:5:320:96:0" "c3:7:320:64:0"
} > $O/sweep.txt 2>&1
run c2 "c2.t:3:0:0:768" "c2.t:1:0:0:768" > $O/timers.txt 2>&1
run c3 "c3.t:3:384:96:768" "c3.t:1:448:0:768" >> $O/timers.txt 2>&1
cut -c1-150 $O/sweep.txt
cut -c1-1700 $O/timers.txt
