#!/bin/bash
# Round 6: A/B of two builds of the library in ONE call (apus_amd/variants/libapus_gpu_{old,new}.so, tools/build_variants.sh):
# default grids at 3 / 1 / 5 replicas twice each, interleaved; the lone round's latencies; then the replica tests on the tree's build
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_q; mkdir -p $O; : > $O/ab.txt
export SWEEP_STEPS=8
for rep in 1 2; do for v in ${VARIANTS:-old new}; do
  echo "== $v (pass $rep)" >> $O/ab.txt
  APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so timeout 600 python tools/rep_sweep.py "$v:3:0:0:0" "$v:1:0:0:0" "$v:5:0:0:0" "$v:7:0:0:0" 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    try:
        i = line.index('{'); d = json.loads(line[i:]); print(line[:i], d['Meps'], d['ok'], 'lat', d['lat'], d['lat_app'])
    except Exception: print(line[:200].rstrip())
" >> $O/ab.txt
  APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so LAT_ONLY=1 bash tools/gpu_lat.sh 2>&1 | grep "^{" | cut -c1-600 >> $O/ab.txt
done; done
cat $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_replica.py tests/test_gpu_host_path.py tests/test_gpu_persistent.py -m gpu -x -q --timeout=600 2>&1 | tail -3
