"""First contact (apus_amd/csrc/apus_selftest.h, apus_gpu_selftest): a pusher's write-through stores + doorbells into a replica's
ring and mailbox against that replica's resident checker, both roles in one launch on the one device; a checker without a
pusher reports its timeout instead of hanging; the log rings in fine-grained memory (APUS_RING_ALLOC, bench.py's fall-back)
carry the same bit-exact path.  The two-process form runs under tests/test_gpu_group.py (bench.py --gpus N, dry run)."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_pusher_and_resident_checker_agree_on_every_byte():
    from apus_amd.engine import Engine
    eng = Engine(3, 1 << 26)
    try:
        out = (C.c_uint64 * 4)()
        for pusher, owner, regions in ((0, 1, 256), (0, 2, 1024), (2, 0, 64)):
            rc = eng.L.apus_gpu_selftest(eng.h, pusher, owner, 3, 300000, regions, 20000, out)
            assert rc == 0 and list(out) == [300000, 0, 0, 0], (pusher, owner, regions, rc, list(out))
        assert eng.L.apus_gpu_ring_alloc_kind(eng.h) == 0
        # round 6: the checker's system-scope atomic max into the pusher's mailbox (what REP_FAST_ACK sends) always got there in
        # front of the store issued behind its drain
        miss = C.c_uint64(7)
        assert eng.L.apus_gpu_selftest_atomic_misses(eng.h, C.byref(miss)) == 0 and miss.value == 0, miss.value
        # what the test touched is cleared: the engine still walks a trace bit for bit
        assert not eng.ring(1, 0, 1 << 16).any() and not eng.ring(0, 0, 1 << 16).any()
    finally:
        eng.close()


def test_a_checker_without_a_pusher_times_out_and_says_so():
    from apus_amd.engine import Engine
    eng = Engine(2, 1 << 26)
    try:
        out = (C.c_uint64 * 4)()
        rc = eng.L.apus_gpu_selftest(eng.h, 0, 1, 2, 1000, 64, 50, out)
        assert rc == 0 and out[0] == 0 and out[3] > 0, (rc, list(out))
        assert eng.L.apus_gpu_selftest(eng.h, 0, 0, 3, 1000, 64, 50, out) != 0          # pusher == owner
        assert eng.L.apus_gpu_selftest(eng.h, 0, 1, 3, 1000, 32, 50, out) != 0          # fewer regions than wavefronts
    finally:
        eng.close()


def test_fine_grained_rings_carry_the_same_path():
    """APUS_RING_ALLOC=finegrained (what bench.py --gpus N falls back to): smoke() -- one small invocation of the hot path
    checked against the oracle -- and the self-test, in a process of its own"""
    code = ("import torch, ctypes as C, __graft_entry__ as g\n"
            "from apus_amd.engine import Engine\n"
            "e = Engine(3, 1 << 26)\n"
            "assert e.L.apus_gpu_ring_alloc_kind(e.h) == 1\n"
            "out = (C.c_uint64 * 4)()\n"
            "assert e.L.apus_gpu_selftest(e.h, 0, 1, 3, 200000, 512, 20000, out) == 0 and list(out) == [200000, 0, 0, 0], list(out)\n"
            "e.close()\n"
            "g.smoke()\nprint('fine-grained ok')\n")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, APUS_RING_ALLOC="finegrained"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "fine-grained ok" in p.stdout, p.stdout[-800:] + p.stderr[-2500:]
