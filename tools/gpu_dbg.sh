#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for d in 0 1 2 4 8 16; do
  rm -rf gpurun_out/prof_$d
  (cd /tmp && APUS_DBG=$d timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/gpurun_out/prof_$d -o kt -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu --no-latency > $OLDPWD/gpurun_out/b_$d.log 2>&1)
  echo "dbg $d: $(grep -c Traceback gpurun_out/b_$d.log)"
done
