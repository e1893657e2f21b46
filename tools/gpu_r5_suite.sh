#!/bin/bash
# One GPU-box call: the whole GPU suite (every failure, not just the first), then a short sweep of the replica kernels
# (tools/rep_sweep.py) on the default library and on the variants named in $VARIANTS.  Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout ${SUITE_TIMEOUT:-420} python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider --durations=8 ${PYTEST_ARGS} 2>&1 | grep -v "^W0\|Gloo\|amdgpu.ids" | tail -${TAIL:-120} > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log | cut -c1-300
if [ -n "$SWEEP" ]; then OUT=${OUT:-rep_sweep.log} CUT=${CUT:-330} bash tools/gpu_sweep.sh; fi
