"""Per-segment timeline of one k_step launch (diagnostics build).
python -m apus_amd.build --trace; APUS_GPU_LIB=apus_amd/libapus_gpu_trace.so python tools/timeline_probe.py"""
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
# one launch only: the first 7 calls of a step (the stamps of later launches would overwrite)
first = []
n = 0
for c in calls:
    first.append(c)
    if c[0] == "rounds":
        n += 1
        if n == 6: break
eng.set_timing(True)
for rep in range(3):
    eng.batch_begin()
    for c in first:
        if c[0] == "rounds": eng.run_rounds(c[1], c[2])
        elif c[0] == "prune": eng.tick_prune()
    eng.batch_end()
    eng.sync()
print('launch (HIP events): total ms, launches =', eng.kernel_time(0))
L = eng.L
L.apus_gpu_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
buf = np.zeros(64 * 16, dtype=np.uint64)
L.apus_gpu_trace(eng.h, buf.ctypes.data, len(buf))
row = lambda r: buf[r * 64:(r + 1) * 64].astype(np.int64)
t0 = int(row(10)[0])
us = lambda v: round((int(v) - t0) / 100.0, 2) if v else None
print("seg | record published | first group: start, record seen | last group: start, stores issued, stores done")
for k in range(8):
    if not row(9)[k]: break
    print(" %2d | %7s | %7s %7s | %7s %7s %7s" % (k, us(row(9)[k]), us(row(10)[k]), us(row(11)[k]), us(row(14)[k]), us(row(12)[k]), us(row(13)[k])))

c = row(15)
print("janitor (record block 0) per segment (us): started, records written, append blocks done, hashes folded, all signed off")
for k in range(8):
    v = c[8 * k:8 * k + 5]
    if not v[0]: break
    print(" %2d |" % k, " ".join("%8s" % us(x) for x in v))
