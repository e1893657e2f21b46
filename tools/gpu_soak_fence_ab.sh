#!/bin/bash
# One GPU-box call: a cross-process trace of tests/_peer_worker.py SOAK times over in one process group, the workers' whole
# stderr kept (APUS_DEBUG=1: the engine says what it maps, unmaps and refuses).  NAME / MODE / WORLD / SOAK
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp APUS_DEBUG=1 HSA_ENABLE_IPC_MODE_LEGACY=0 APUS_DIST_BACKEND=gloo APUS_DIST_ONE_DEVICE=1
mkdir -p gpurun_out
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=${WORLD:-5} --master-addr 127.0.0.1 --master-port 29617 \
    tests/_peer_worker.py /tmp/soak_res ${NAME:-c5_rejoin} ${MODE:-replica} 0 ${SOAK:-8} > gpurun_out/soak_diag.log 2>&1
echo "exit $?"
grep -E "apus_gpu\]|Error|rank" gpurun_out/soak_diag.log | grep -v "workgroups per CU\|replica launch\|k_step" | cut -c1-200 | sort | uniq -c | sort -rn | head -${HEAD:-25}
for r in 0 1 2 3 4; do [ -f /tmp/soak_res.$r ] && python -c "import json;d=json.load(open('/tmp/soak_res.$r'));print($r,d['ok'],d['runs'],d['checks'],str(d.get('error'))[:150])"; done
