#!/bin/bash
# One GPU call: tools/rep_sweep.py over the specs in $SWEEP (see rep_sweep.py) on the default library, then $VSWEEP on every
# tuning variant named in $VARIANTS (apus_amd/variants/libapus_gpu_<name>.so, tools/build_variants.sh); log in gpurun_out/$OUT
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/${OUT:-rep_sweep.log}
: > $OUT
if [ -n "$SWEEP" ]; then
  echo "# default library" >> $OUT
  timeout ${SWEEP_TIMEOUT:-400} python tools/rep_sweep.py $SWEEP 2>&1 | grep -v "amdgpu.ids" >> $OUT
fi
for v in $VARIANTS; do
  echo "# variant $v" >> $OUT
  APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so timeout ${SWEEP_TIMEOUT:-400} python tools/rep_sweep.py ${VSWEEP:-$SWEEP} 2>&1 | grep -v "amdgpu.ids" >> $OUT
done
cut -c1-${CUT:-260} $OUT
