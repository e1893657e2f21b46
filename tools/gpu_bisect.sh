#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for dbg in ${DBGS:-0 32 64 96}; do
  APUS_REP_DBG=$dbg timeout 200 python -m pytest tests/test_gpu_peers.py -m gpu -q -x --timeout 180 -k "${K:-replica_kernels and steady7}" 2>&1 | grep -v "^W0\|Gloo\|amdgpu.ids" > gpurun_out/bisect_$dbg.log
  echo "dbg=$dbg: $(tail -1 gpurun_out/bisect_$dbg.log) $(grep -o "rank [0-9]*: AssertionError([^)]\{0,300\}" gpurun_out/bisect_$dbg.log | head -2 | cut -c1-500)"
done
