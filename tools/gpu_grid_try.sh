#!/bin/bash
# One GPU-box call: the default workgroup grids of the replica kernels against alternatives, per configuration of the bench line
# (APUS_REP_DEFAULT_APPEND / _FWORK are read by apus_gpu_rep_start when the caller passes 0).  GRIDS="cfg:append:fwork ..."
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/grid_try.log
for spec in $GRIDS; do
  IFS=: read cfg na nf <<< "$spec"
  r=$(APUS_REP_DEFAULT_APPEND=$na APUS_REP_DEFAULT_FWORK=$nf timeout 60 python tools/rep_profile_run.py $cfg --steps ${STEPS:-4} 2>&1 | grep "^{" | tail -1)
  echo "$cfg $na $nf $(echo $r | python -c 'import json,sys; d=json.loads(sys.stdin.read() or "{}"); print(round(d.get("entries_per_s",0)/1e6), d.get("verified"))')" >> gpurun_out/grid_try.log
done
cat gpurun_out/grid_try.log
