mkdir -p gpurun_out
show() { python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): print(l); continue
    d=json.loads(l)
    if 'error' in d: print(d); continue
    r=d['roles']
    f=lambda k: (r[k]['moved'], round(r[k]['rounds']/max(1,r[k]['moved'])), round(r[k]['busy_us']/max(1,r[k]['moved']),1), round(r[k]['busy_us']/r[k]['us'],2)) if k in r else None
    print(d['replicas'], d['n_append'], d['n_fwork'], round(d['entries_per_s']/1e6), d['verified'], d['lat_us_p50'], d['lat_appended_us_p50'], 'seq(x)', r['sequencer']['x'], round(r['sequencer']['busy_us']/269,1), 'com', f('committer'), 'app', f('applier'), 'f0r', f('f0_retire'), 'f0a', f('f0_apply'), r.get('append'))
"; }
for d in $DBGS; do echo dbg=$d; APUS_REP_DBG=$d timeout 100 python tools/rep_bench.py --replicas ${REPL:-3} --grid $GRID --steps 3 --no-hostfed --brief; done 2>&1 | tee -a gpurun_out/rep_exp3.log | show
