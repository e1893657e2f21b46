#!/bin/bash
# per-role pass statistics of several builds at SPECS group sizes (rep_sweep.py with APUS_REP_DBG=512: per-pass clocks)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_roles_ab.txt; : > $O
for srep in 1 2; do for v in ${VARIANTS:-new old}; do
  specs=""; for sp in ${SPECS:-1}; do specs="$specs $v:$sp:0:0:${DBG:-512}"; done
  APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so SWEEP_STEPS=8 timeout 300 python tools/rep_sweep.py $specs 2>&1 | grep -v "amdgpu.ids\|^W0" | cut -c1-1500 >> $O
done; done
cat $O
