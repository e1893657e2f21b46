"""In-kernel timing probe.  Build the diagnostics library first (python -m apus_amd.build --trace),
then: APUS_GPU_LIB=apus_amd/libapus_gpu_trace.so python tools/trace_probe.py"""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
os.environ.setdefault('APUS_GPU_LIB', os.path.join('apus_amd', 'libapus_gpu_trace.so'))
from apus_amd import trace as T
from apus_amd.engine import Engine
tr = T.config_c2()
eng = Engine(3, tr.log_len)
eng.stage_trace(tr); eng.elect(0)
ev = tr.events; i = 0; calls = []
while i < len(ev):
    if ev[i][0] == "ROUND":
        j = i
        while j < len(ev) and ev[j][0] == "ROUND": j += 1
        calls.append((eng.round_of_g0[ev[i][1]], j - i)); i = j; continue
    if ev[i][0] == "PRUNE": calls.append(None)
    i += 1
batch = "--batch" in sys.argv        # the calls as one multi-segment launch (k_step): stamps of its last segment
for rep in range(2):
    if batch: eng.batch_begin()
    for c in calls[:9]:
        if c is None: eng.tick_prune()
        else: eng.run_rounds(*c)
    if batch: eng.batch_end()
eng.sync()
L = eng.L
L.apus_gpu_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
buf = np.zeros(64 * 16, dtype=np.uint64)
L.apus_gpu_trace(eng.h, buf.ctypes.data, len(buf))
names = {0: "k_sequence", 1: "append block 0", 6: "append block R-1", 4: "scan block 0", 2: "applier 0", 5: "recorder 0", 3: "keeper"}
base = int(buf[0])
for k in (0, 1, 6, 4, 2, 5, 3):
    row = buf[k * 64:k * 64 + 8].astype(np.int64)
    if False: print("   applier probes:", [round((int(b) - int(a)) / 100.0, 2) for a, b in zip(buf[k*64+10:k*64+13], buf[k*64+11:k*64+14])], "barrier+setup:", round((int(buf[k*64+10]) - int(buf[k*64+1])) / 100.0, 2))
    st = [int(v) for v in row if v]
    if not st: continue
    if k == 1:
        o = [int(buf[64 + i]) for i in (0, 3, 4, 1, 5, 2)]
        print("append block 0: start, staged, prefetch issued, sequenced, placed, stored (us):", [round((v - o[0]) / 100.0, 2) for v in o]); continue
    print("%-22s" % names[k], "start@%.2f" % ((st[0] - base) / 100.0), "deltas(us):", [round((b - a) / 100.0, 2) for a, b in zip(st, st[1:])])
