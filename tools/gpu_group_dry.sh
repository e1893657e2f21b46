mkdir -p gpurun_out
for n in 2 4; do
APUS_DIST_BACKEND=gloo APUS_DIST_ONE_DEVICE=1 timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 2 --warmup 1 --entries 131072 --cpu-seconds 1 --watchdog 60 > gpurun_out/g$n.out 2> gpurun_out/g$n.err; echo n=$n rc=$?
grep "\[bench\]\|EngineError\|Timeout (" gpurun_out/g$n.err | head -5
grep "^{" gpurun_out/g$n.out | cut -c1-3000
done
