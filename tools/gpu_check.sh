#!/bin/bash
# One GPU-box call: parity tests, smoke, a short bench, kernel-trace profile.  Logs land in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_X--x} --timeout 600 2>&1 | tail -80 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 --cpu-seconds ${CPU_SECONDS:-5} > gpurun_out/bench.log 2>&1
echo "bench exit: $?"
tail -3 gpurun_out/bench.log
if [ -n "$PROFILE" ]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o kt -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu > $OLDPWD/gpurun_out/prof_run.log 2>&1)
  echo "prof exit: $?"
  find gpurun_out/prof -name "*stats*" | head
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -20 "$f"
fi
if [ -n "$PMC" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $OLDPWD/gpurun_out/pmc_$c -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu --eager --no-latency > $OLDPWD/gpurun_out/pmc_$c.log 2>&1)
    echo "pmc $c exit: $?"
    ls gpurun_out/pmc_$c | head
  done
fi
if [ -n "$GROUP" ]; then
  APUS_DIST_BACKEND=gloo APUS_DIST_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 3 --steps 2 --warmup 1 --entries 262144 > gpurun_out/bench_group.log 2>&1
  echo "group bench exit: $?"; tail -3 gpurun_out/bench_group.log | cut -c1-600
fi
if [ -n "$EXTRA_CONFIGS" ]; then
  for c in c3 c4; do
    timeout 600 python bench.py --config $c --steps 5 --warmup 2 --no-cpu --no-latency > gpurun_out/bench_$c.log 2>&1
    echo "bench $c exit: $?"; tail -1 gpurun_out/bench_$c.log | cut -c1-400
  done
fi
