#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_i; mkdir -p $O
timeout 300 python tools/rep_spread.py 3 10 8 > $O/spread3.txt 2>&1
timeout 300 python tools/rep_spread.py 1 10 8 > $O/spread1.txt 2>&1
cat $O/spread3.txt $O/spread1.txt | cut -c1-260
