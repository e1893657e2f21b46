"""apus_gpu_selftest on one device, both roles in one launch: plain, with a release behind the pusher's stores (16), with an
invalidate in front of the checker's loads (32), both.  APUS_DEBUG=1 prints the first unit that differed."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["APUS_DEBUG"] = "1"
from apus_amd.engine import Engine  # noqa: E402

eng = Engine(3, 1 << 26)
out = (C.c_uint64 * 4)()
for mode in (0, 16, 32, 48):
    for regions in (64, 1024):
        rc = eng.L.apus_gpu_selftest(eng.h, 0, 1, 3 | mode, int(sys.argv[1]) if len(sys.argv) > 1 else 50000, regions, 20000, out)
        print("mode", mode, "regions", regions, "rc", rc, list(out), flush=True)
eng.close()
