#!/bin/bash
# Round 6, last question: the two states of a box (5.1-5.3 G entries/s for the first launches of a call, 4.5-4.6 G behind them).
# Does an idle pause bring the first state back?  Three launches, 20 s idle, three launches, 40 s idle, three launches -- with
# rocm-smi's power / clocks / temperatures (junction AND memory) sampled beside them.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_state; mkdir -p $O
( for i in $(seq 1 400); do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power\|junction\|memory" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';' | cut -c1-500)"; sleep 0.4; done ) > $O/smi.txt 2>&1 &
SMI=$!
run3() {
  echo "## $1 $(date +%s.%N | cut -c1-14)"
  SWEEP_STEPS=16 timeout 300 python tools/rep_sweep.py "c:3:0:0:0" "c:3:0:0:0" "c:3:0:0:0" 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    try:
        i = line.index('{'); d = json.loads(line[i:]); print(line[:i], d['Meps'], d['ok'])
    except Exception: pass
"
}
{ run3 "cold"; run3 "at once behind it"; sleep 20; run3 "after 20 s idle"; sleep 40; run3 "after 40 s idle"; } > $O/runs.txt 2>&1
kill $SMI 2>/dev/null
cat $O/runs.txt
awk 'NR%5==1' $O/smi.txt | cut -c1-420 | head -60
