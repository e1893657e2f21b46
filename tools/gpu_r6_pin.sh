#!/bin/bash
# Round 6: host-fed throughput with the producers where the scheduler puts them against one producer per core (APUS_FEED_PIN=<stride>), three times each
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_pin; mkdir -p $O
{ nproc; lscpu | grep -i "model name\|thread(s) per core\|socket\|numa node\|^CPU(s)"; taskset -cp $$; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo; } > $O/box.txt 2>&1
: > $O/feed.txt
for rep in 1 2 3; do for pin in 0 1 2 4; do
  APUS_FEED_PIN=$pin timeout 300 python - <<PY >> $O/feed.txt 2>&1
import sys, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from rep_bench import hostfed
d = hostfed(3, 64, 0.4, 0, 0)
print("pin=$pin", {k: round(v["entries_per_s"] / 1e6) for k, v in d["by_threads"].items()}, all(v["verified"] for v in d["by_threads"].values()))
PY
done; done
cat $O/box.txt; grep "^pin" $O/feed.txt; grep -v "^pin" $O/feed.txt | tail -5
