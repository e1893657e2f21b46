#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_q; mkdir -p $O; : > $O/ab2.txt
export SWEEP_STEPS=8
for rep in 1 2 3 4 5; do for v in ${VARIANTS:-old new}; do
  APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so timeout 600 python tools/rep_sweep.py "$v:3:0:0:0" "$v:7:0:0:0" "$v:3:0:0:0" 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    try:
        i = line.index('{'); d = json.loads(line[i:]); print(line[:i], d['Meps'], d['ok'])
    except Exception: print(line[:200].rstrip())
" >> $O/ab2.txt
done; done
cat $O/ab2.txt
