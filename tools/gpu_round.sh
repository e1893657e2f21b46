#!/bin/bash
# One GPU-box call at the end of a round: the whole GPU suite, the full bench line, the profile artifacts.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s --timeout 400 2>&1 | grep -v "^W0\|Gloo\|amdgpu.ids" | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_line.json 2> gpurun_out/bench.err
echo "bench exit: $?"; tail -2 gpurun_out/bench.err | cut -c1-300
CONFIGS="${CONFIGS:-c2}" REPLICA=1 ROUND=${ROUND:-r04} bash tools/gpu_profile.sh 2>&1 | tail -30
# the reference-as-is cluster on this box's host cores (does process_vm_writev work here, or does the NIC-thread transport take over?)
timeout 200 python -m oracle.procref --opt O0 -n 20000 -c 1,50 --plain > gpurun_out/procref.json 2> gpurun_out/procref.err; echo "procref exit: $?"; cut -c1-600 gpurun_out/procref.json
