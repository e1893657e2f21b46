#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_n; mkdir -p $O; rm -f gpurun_out/failover_postmortem.txt
for i in $(seq 1 ${RUNS:-12}); do
  APUS_DEBUG=1 timeout 900 python -m pytest tests/test_gpu_e2e_failover.py -m gpu -q --timeout=800 > $O/fo_$i.txt 2>&1
  rc=$?
  echo "failover $i exit $rc : $(tail -1 $O/fo_$i.txt)"
  if [ $rc -ne 0 ]; then cp gpurun_out/failover_postmortem.txt $O/postmortem_$i.txt; grep -v "^ *[|(\`._-]" gpurun_out/failover_postmortem.txt | grep "\[T\|\[apus\|====" | cut -c1-230; break; fi
done
