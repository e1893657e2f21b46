#!/bin/bash
# Round 4, GPU call 1: (a) JOIN soak + reproduction, (b) the new full-size replica tests + the replica suite,
# (c) where the replica kernels' time goes on the current build: grids, measurement knobs, R_SUB.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
SOAK=${SOAK:-12} SOAK_TIMEOUT=700 bash tools/gpu_soak.sh
echo "soak took $(( $(date +%s) - t0 )) s"
t1=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_replica.py -m gpu -q --timeout 500 -x 2>&1 | tail -8 > gpurun_out/rep_tests.log
tail -4 gpurun_out/rep_tests.log
echo "replica tests took $(( $(date +%s) - t1 )) s"
t2=$(date +%s)
{
timeout 400 python tools/rep_sweep.py base:3:96:48:0 base:3:128:64:0 base:3:160:80:0 base:3:192:96:0 base:3:224:112:0 \
    timers:3:96:48:256 timers:3:160:80:256 nohost:3:96:48:2048 nohost:3:160:80:2048 nt:3:96:48:1024 nt:3:160:80:1024 nt+nohost:3:160:80:3072 \
    noreply:3:96:48:1 nofderived:3:96:48:2 nolderived:3:96:48:8192 nothing:3:96:48:8195 nothing+nohost:3:160:80:10243 \
    one:1:96:1:0 one:1:192:1:0 five:5:0:0:0 seven:7:0:0:0
APUS_GPU_LIB=apus_amd/variants/libapus_gpu_rsub8.so timeout 200 python tools/rep_sweep.py rsub8:3:96:48:0 rsub8:3:120:60:0 rsub8+nohost:3:120:60:2048 rsub8+nt+nohost:3:120:60:3072
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/rep_sweep1.log | cut -c1-700
echo "sweep took $(( $(date +%s) - t2 )) s"
