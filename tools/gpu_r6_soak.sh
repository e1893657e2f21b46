#!/bin/bash
# Round 6 soak: the whole GPU suite three times over (a flaky test shows here, not in the driver's run), the replica-kernel
# tests ten times, the cross-process peers tests five times, the kill -9 fail-over and reconfiguration groups twenty times
# (SUITES / REPS / PEERS / FAILOVERS)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_soak; mkdir -p $O
: > $O/summary.txt
for i in $(seq 1 ${SUITES:-3}); do
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/suite_$i.txt 2>&1
  echo "suite $i exit $? : $(tail -1 $O/suite_$i.txt)" >> $O/summary.txt
done
for i in $(seq 1 ${REPS:-10}); do
  timeout 600 python -m pytest tests/test_gpu_replica.py tests/test_gpu_host_path.py -m gpu -q --timeout=600 > $O/rep_$i.txt 2>&1
  echo "replica $i exit $? : $(tail -1 $O/rep_$i.txt)" >> $O/summary.txt
done
for i in $(seq 1 ${PEERS:-5}); do
  timeout 900 python -m pytest tests/test_gpu_peers.py tests/test_gpu_peers_kill.py tests/test_gpu_peers_deposed.py tests/test_gpu_e2e_partition.py -m gpu -q --timeout=800 > $O/peers_$i.txt 2>&1
  echo "peers $i exit $? : $(tail -1 $O/peers_$i.txt)" >> $O/summary.txt
done
for i in $(seq 1 ${FAILOVERS:-20}); do
  timeout 900 python -m pytest tests/test_gpu_e2e_failover.py tests/test_gpu_e2e_reconf.py -m gpu -q --timeout=800 > $O/failover_$i.txt 2>&1
  rc=$?
  echo "failover+reconf $i exit $rc : $(tail -1 $O/failover_$i.txt)" >> $O/summary.txt
  if [ $rc -ne 0 ]; then cp gpurun_out/failover_postmortem.txt $O/failover_postmortem_$i.txt 2>/dev/null; fi
done
cat $O/summary.txt
grep -l "failed\|error" $O/*.txt | head
