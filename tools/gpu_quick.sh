#!/bin/bash
# quick perf probe: bench (no cpu/latency legs) under rocprofv3 kernel trace
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OLDPWD/gpurun_out/prof -o kt -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu --no-latency > $OLDPWD/gpurun_out/b.log 2>&1)
tail -1 gpurun_out/b.log | cut -c1-200
