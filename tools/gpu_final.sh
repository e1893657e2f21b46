#!/bin/bash
# The round's last GPU-box call when minutes are short: the replica kernels' profile (the traffic file must belong to the build),
# the bench line, then as much of the GPU suite as the time allows (results so far are kept in gpurun_out/pytest_gpu.log).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CONFIGS="" REPLICA=1 ROUND=${ROUND:-r04} bash tools/gpu_profile.sh 2>&1 | grep "k_replica\|bytes_per_entry" | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_line.json 2> gpurun_out/bench.err
echo "bench exit: $?"
timeout ${SUITE_TIMEOUT:-200} python -m pytest tests -m gpu -q --timeout 150 -p no:cacheprovider 2>&1 | grep -v "^W0\|Gloo\|amdgpu.ids" | tail -15 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
