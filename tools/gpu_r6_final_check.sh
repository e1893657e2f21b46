#!/bin/bash
# Round 6, last tree: what the driver runs in front of the bench (smoke()), the bench with its default flags, and the
# --gpus N launch line of the driver as a dry run on one device (APUS_DIST_ONE_DEVICE=1, gloo: every rank on cuda:0).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06_final
export TMPDIR=/tmp
O=gpurun_out/r06_final
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.txt | cut -c1-300
timeout 600 python bench.py > $O/bench_default_flags.json 2> $O/bench_default_flags.err; echo "bench rc=$?"
python - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r06_final/bench_default_flags.json") if x.startswith("{")][-1])
r = l["roofline"]
print("value %.3f G ms/step %.4f frac_moved %s bit_exact %s" % (l["value"] / 1e9, l["ms_per_step"], r.get("frac_moved"), l["replica_kernels"]["device_resident"].get("bit_exact_vs_oracle")))
PY
for n in 2 3 4 8; do
APUS_DIST_BACKEND=gloo APUS_DIST_ONE_DEVICE=1 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 2 --warmup 1 --entries 131072 --cpu-seconds 1 --watchdog 100 > $O/group_dry_$n.json 2> $O/group_dry_$n.err; echo "n=$n rc=$?"
grep "\[bench\]\|EngineError\|Timeout (" $O/group_dry_$n.err | head -5 | cut -c1-300
grep "^{" $O/group_dry_$n.json | cut -c1-600
done
