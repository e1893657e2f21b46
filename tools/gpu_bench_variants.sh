#!/bin/bash
# bench only, on the default library and on every tuning variant in $VARIANTS
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/variants.log
run() {
  local label=$1; shift
  local out
  out=$(env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-latency --no-ack-path 2>&1 | tail -1)
  echo "$label $(echo "$out" | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); r = d['roofline']
    print('value=%.3fG ms/step=%.4f k_step_us=%.2f launches=%d' % (d['value'] / 1e9, d['ms_per_step'], r['avg_launch_us'], r['launches']))
except Exception as e:
    print('FAILED', e)
")" | tee -a gpurun_out/variants.log
}
run "default"
for v in ${VARIANTS}; do run "$v" APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so; done
