#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_replica.py ${EXTRA_TESTS} -m gpu -q --timeout 600 2>&1 | grep -v "amdgpu.ids\|^W0\|Gloo" | tail -25 > gpurun_out/rep_tests.log
tail -12 gpurun_out/rep_tests.log | cut -c1-400
{
timeout 400 python tools/rep_sweep.py ${SWEEP:-base:3:96:48:0 base:3:128:64:0 base:3:192:96:0 timers:3:96:48:256 one:1:96:1:0 one:1:192:1:0 five:5:0:0:0 seven:7:0:0:0}
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/rep_sweep.log | cut -c1-900
