"""The rate at which one MI355X takes the data path's write-through stores when the launch does nothing else
(apus_gpu_calib_store_multi): 8 KiB chunks at the same offset into 1 / 3 / 5 / 7 rings of 64 MiB, by 192 ... 1024 workgroups."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apus_amd.engine import Engine  # noqa: E402

eng = Engine(7, 1 << 26)
g = C.c_float(0)
for skew in (0, 64, 16):
    for n in (1, 3, 7):
        for wgs in (192, 768):
            rc = eng.L.apus_gpu_calib_store_multi(eng.h, (1 << n) - 1, (8 if n > 1 else 16) | (skew << 16), wgs, C.byref(g))
            print(f"offset in line {skew:3d}  rings {n} workgroups {wgs:4d}: rc {rc} {g.value / 1e3:.2f} TB/s written", flush=True)
eng.close()
