"""SURVEY.md 8 f4, CPU part: the durability side channel.

The records persist_new_entries hands to proxy_store_cmd (BerkeleyDB's input, = the snapshot a joiner's
donor ships) are pinned on the reference itself (tests/test_oracle_vs_refloops.py compares the streams
byte for byte; tests/golden/cluster_ref.json holds their SHA-256).  Here: the host layer's replay of such
a snapshot (apus_snapshot_replay = stablestorage_load_records, src/proxy/proxy.c:306-339) walks the
oracle's stream exactly as the records were laid down -- host logic in C, no GPU needed."""
import ctypes as C

import numpy as np
import pytest

from apus_amd import build
from oracle import oracle as orc
from tests import traces

DO_ACTION = C.CFUNCTYPE(None, C.c_uint16, C.c_uint8, C.c_size_t, C.c_void_p, C.c_void_p)
STORE = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p)


def host_lib():
    L = C.CDLL(build.build())
    L.apus_snapshot_replay.restype = C.c_int
    L.apus_snapshot_replay.argtypes = [C.c_char_p, C.c_uint32, STORE, DO_ACTION, C.c_void_p]
    return L


def expected_records(cl, r):
    """(clt_id, type) of every client entry replica r holds, in log order (no wrap in these traces)"""
    o = cl.log(r).offsets()
    ring = cl.log(r).ring()
    out, off = [], 0
    while off != o["end"]:
        typ = int(ring[off + 26])
        ln = 64 if typ in (0, 2, 3) else 64 + int(ring[off + 48]) + (int(ring[off + 49]) << 8)
        if typ in (4, 5, 6):
            out.append((int(ring[off + 24]) | int(ring[off + 25]) << 8, typ))
        off += ln
    return out


@pytest.mark.parametrize("name,n", [("steady3", 3), ("steady7_mixed", 7)])
def test_replay_walks_the_reference_format(name, n):
    tr = traces.CATALOGUE[name]()
    tr.log_len = 1 << 22                      # no wrap: the stream covers the whole log
    tr.events = [e for e in tr.events if e[0] != "PRUNE"]
    cl = orc.run_trace(tr)
    L = host_lib()
    for r in (0, n - 1):
        snap = cl.store_stream(r)
        assert len(snap) == cl.records_len(r)
        seen, stored = [], []
        do = DO_ACTION(lambda clt, typ, ln, data, arg: seen.append((clt, typ, ln)))
        st = STORE(lambda rec, nbytes, arg: stored.append(nbytes))
        assert L.apus_snapshot_replay(snap, len(snap), st, do, None) == len(seen)
        want = expected_records(cl, r)
        assert [(c, t) for c, t, _ in seen] == want
        # the overlay: a SEND record is 24 bytes + reply[4] | reply[5] << 8 -- no command byte in step
        assert all(ln == 0 for c, t, ln in seen) and sum(stored) == len(snap)
        assert set(stored) <= {4, 24}


def test_replay_refuses_a_malformed_stream():
    L = host_lib()
    do = DO_ACTION(lambda *a: None)
    st = STORE(lambda *a: None)
    assert L.apus_snapshot_replay(bytes([1, 0, 9, 0]), 4, st, do, None) == -1          # action 9
    assert L.apus_snapshot_replay(bytes([1, 0, 5, 0]) + bytes(8), 12, st, do, None) == -1     # SEND cut short
    assert L.apus_snapshot_replay(b"", 0, st, do, None) == 0


def test_overlay_takes_the_length_from_reply_bytes():
    """hold_release: server 2 is caught up after servers 3 and 4 acknowledged -- reply[4] is set in the
    bytes it persists, so its SEND records are 25 bytes long, pinned on the reference"""
    tr = traces.hold_release()
    cl = orc.run_trace(tr)
    lens = {r: cl.records_len(r) for r in range(5)}
    assert lens[2] > lens[0] == lens[1] == lens[3] == lens[4]
    L = host_lib()
    sizes = []
    st = STORE(lambda rec, nbytes, arg: sizes.append(nbytes))
    snap = cl.store_stream(2)
    assert L.apus_snapshot_replay(snap, len(snap), st, DO_ACTION(lambda *a: None), None) > 0
    assert 25 in sizes and sum(sizes) == lens[2]
