#!/bin/bash
# Round 6: does the device clock down under the replica kernels' sustained load?  rocm-smi sampled twice a second beside six launches of 16 steps
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_clk; mkdir -p $O
( for i in $(seq 1 80); do echo "t=$i $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|Socket Power\|junction\|fclk" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';' | cut -c1-400)"; sleep 0.5; done ) > $O/smi.txt 2>&1 &
SMI=$!
sleep 2
SWEEP_STEPS=16 timeout 600 python tools/rep_sweep.py "c:3:0:0:0" "c:3:0:0:0" "c:3:0:0:0" "c:3:0:0:0" "c:3:0:0:0" "c:3:0:0:0" 2>&1 | python -c "
import sys, json, time
for line in sys.stdin:
    try:
        i = line.index('{'); d = json.loads(line[i:]); print(line[:i], d['Meps'], d['ok'])
    except Exception: print(line[:200].rstrip())
" > $O/runs.txt
kill $SMI 2>/dev/null
cat $O/runs.txt; awk 'NR%4==1' $O/smi.txt | cut -c1-330 | head -40
