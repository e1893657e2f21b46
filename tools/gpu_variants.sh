#!/bin/bash
# One GPU-box call: parity tests on the default library, then the bench on every tuning variant.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; fi
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
: > gpurun_out/variants.log
run() {  # label, env...
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
run "default(gp4,nolb)"
run "gp1(baseline)" APUS_GP_ROUNDS=1
for v in ${VARIANTS}; do
  run "$v" APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so
done
