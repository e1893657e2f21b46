"""In-kernel timing of one k_step launch: the bookkeeping chain (row 7) and an append group (row 8).
python -m apus_amd.build --trace; APUS_GPU_LIB=apus_amd/libapus_gpu_trace.so python tools/chain_probe.py"""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
os.environ.setdefault('APUS_GPU_LIB', os.path.join('apus_amd', 'libapus_gpu_trace.so'))
import bench
from apus_amd import trace as T
from apus_amd.engine import Engine
tr = T.config_c2()
eng = Engine(3, tr.log_len)
eng.stage_trace(tr); eng.elect(0)
calls = bench.step_calls(tr, eng)
for rep in range(3):
    bench.issue(eng, calls)
eng.sync()
L = eng.L
L.apus_gpu_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
buf = np.zeros(64 * 16, dtype=np.uint64)
L.apus_gpu_trace(eng.h, buf.ctypes.data, len(buf))
row = buf[7 * 64:8 * 64].astype(np.int64)
t0 = int(row[0])
print("chain block (us from its start): per segment [iteration start, inputs staged, decided, epoch raised]")
for k in range(16):
    v = row[4 * k:4 * k + 4]
    if not v[3]: break
    print("  seg %2d:" % k, [round((int(x) - t0) / 100.0, 2) if x else None for x in v])
g = buf[8 * 64:9 * 64].astype(np.int64)
names = ["start", "descriptors in (barrier 1)", "layout + prefetch issued", "SeqOut known (barrier 2)", "placed + directory", "stores issued", "apply records", "decide lane done"]
print("append group 0 of the last segment (us):", {n: round((int(g[i]) - int(g[0])) / 100.0, 2) for i, n in enumerate(names) if g[i]})
