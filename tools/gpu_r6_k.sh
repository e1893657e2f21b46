#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_k; mkdir -p $O
export SWEEP_STEPS=6
timeout 600 python tools/rep_sweep.py "k:3:0:0:0" "k:3:0:0:0" "k:1:0:0:0" "k:1:0:0:0" "k:5:0:0:0" "k:7:0:0:0" "k.t:1:0:0:768" > $O/sweep.txt 2>&1
cut -c1-100 $O/sweep.txt; tail -1 $O/sweep.txt | cut -c1-1500; timeout 600 python -m pytest tests/test_gpu_replica.py -m gpu -q -x --timeout=600 2>&1 | tail -2
