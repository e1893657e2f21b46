#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_e2e_partition.py -m gpu -q -x --timeout=500 > $O/partition.txt 2>&1
echo "partition exit: $?"; tail -30 $O/partition.txt | cut -c1-300


ls gpurun_out/*postmortem* 2>/dev/null
