import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
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
for rep in range(2):
    for c in calls[:9]:
        if c is None: eng.tick_prune()
        else: eng.run_rounds(*c)
eng.sync()
L = eng.L
L.apus_gpu_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
buf = np.zeros(64 * 8, dtype=np.uint64)
L.apus_gpu_trace(eng.h, buf.ctypes.data, len(buf))
names = {0: "k_sequence", 2: "k_apply b00", 4: "k_persist_commit b0"}
base = int(buf[0])
for k in (0, 2, 4):
    row = buf[k * 64:k * 64 + 8].astype(np.int64)
    st = [int(v) for v in row if v]
    print(names[k], "start@%.2f" % ((st[0] - base) / 100.0), "deltas(us):", [round((b - a) / 100.0, 2) for a, b in zip(st, st[1:])])
