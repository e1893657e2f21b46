#!/bin/bash
# One short GPU call: tools/rep_sweep.py over the specs in $SWEEP (see rep_sweep.py), log in gpurun_out/$OUT
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout ${SWEEP_TIMEOUT:-400} python tools/rep_sweep.py $SWEEP 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/${OUT:-rep_sweep.log} | cut -c1-900
