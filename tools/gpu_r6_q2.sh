#!/bin/bash
# Round 6: A/B of builds (apus_amd/variants/libapus_gpu_<v>.so, tools/build_variants.sh) in ONE call: PASSES passes over VARIANTS, a process per
# build and pass (SPECS = rep_sweep.py's "label:replicas:n_append:n_fwork:dbg" without the label), then the runs by build in time order
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_q; mkdir -p $O; : > $O/ab2.txt
export SWEEP_STEPS=${SWEEP_STEPS:-8}
for rep in $(seq 1 ${PASSES:-5}); do for v in ${VARIANTS:-old new}; do
  specs=""; for s in ${SPECS:-3:0:0:0 7:0:0:0 3:0:0:0}; do specs="$specs $v:$s"; done
  APUS_GPU_LIB=apus_amd/variants/libapus_gpu_$v.so timeout 600 python tools/rep_sweep.py $specs 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    try:
        i = line.index('{'); d = json.loads(line[i:]); print(line[:i], d['Meps'], d['ok'], d['lat'], d['lat_app'])
    except Exception: print(line[:200].rstrip())
" >> $O/ab2.txt
done; done
python - <<'PY'
import collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r06_q/ab2.txt"):
    p = l.split()
    if len(p) >= 7: d[(p[0], p[1])].append(int(p[5]))
for k in sorted(d): print(k, d[k], "mean", sum(d[k]) // len(d[k]), "mean of the last half", sum(d[k][len(d[k]) // 2:]) // max(1, len(d[k]) - len(d[k]) // 2))
PY
