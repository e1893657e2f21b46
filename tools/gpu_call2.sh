#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gpu_dbg.py c2_full 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/dbg_c2.log
SOAK=16 SOAK_K="join_soak and c5_rejoin and replica" SOAK_TIMEOUT=400 bash tools/gpu_soak.sh
