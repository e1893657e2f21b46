"""Pin the oracle's restatement of dare_log.h against the reference header itself
(oracle/_ref/libapus_ref.so = /root/reference/src/include/dare/dare_log.h compiled
unchanged).  Covers the known answers of SURVEY.md section 10 and a randomized
differential over every log function."""
import random

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built and /root/reference absent")

L64M = orc.DEFAULT_LOG


def both(length):
    return orc.OracleLog(length), orc.RefLog(length)


def same_state(o, r, ring=True):
    assert o.offsets() == r.offsets()
    assert o.prev_head == r.prev_head
    if ring:
        assert np.array_equal(o.ring(), r.ring())


def test_layout_matches_survey_probe():
    lay = orc.ref_layout()
    assert lay == dict(sizeof_entry=64, idx=0, term=8, req_id=16, clt_id=24, type=26, sender=27,
                       reply=28, data=48, sizeof_cid=16, entries=319656, LOG_SIZE=67108864)


def test_fresh_log_and_two_appends():
    o, r = both(L64M)
    for lg in (o, r):
        assert lg.offsets() == dict(head=0, apply=0, commit=0, end=L64M, tail=L64M, old_end=L64M,
                                    old_commit=0, len=L64M)
        assert lg.append(1, 7, 3, orc.SEND, bytes(range(64))) == 1
        assert lg.append(1, 8, 3, orc.SEND, bytes(range(64))) == 2
        off = lg.offsets()
        assert (off["tail"], off["end"]) == (128, 256)
        assert lg.entry_len_at(0) == 128
    same_state(o, r, ring=False)
    assert np.array_equal(o.ring()[:512], r.ring()[:512])


def test_circular_compare_known_answers():
    o, r = both(4096)
    for lg in (o, r):
        lg.append(1, 1, 1, orc.SEND, b"x" * 264)      # end = 328
        assert lg.offsets()["end"] == 328
        assert lg.is_larger(10, 20) == 0
        assert lg.is_larger(20, 10) == 1
        assert lg.end_distance(10) == 318


def test_case1_wrap_header_does_not_fit():
    length = 4096
    o, r = both(length)
    for lg in (o, r):
        lg.append(1, 1, 1, orc.SEND, b"a" * 100)
        lg.append(1, 2, 1, orc.SEND, b"b" * 100)
        # consume so that the log is not full, then force end = len-40
        lg.set_offsets(head=164, apply=328, commit=328, end=length - 40, tail=164)
        assert lg.append(1, 3, 1, orc.SEND, b"c" * 100) == 3
        off = lg.offsets()
        assert (off["tail"], off["end"]) == (0, 164)
        assert lg.get_entry(length - 40) == 0        # reader redirected to offset 0
    same_state(o, r)


def test_case2_wrap_leaves_stale_header():
    length = 1024
    o, r = both(length)
    for lg in (o, r):
        assert lg.append(2, 0, 0, orc.NOOP) == 1              # 64 B
        for k in range(6):                                     # 6 x 128 -> end = 832
            assert lg.append(2, k + 1, 9, orc.SEND, bytes([k]) * 64) == k + 2
        lg.set_offsets(head=448, apply=832, commit=832)
        assert lg.offsets()["end"] == 832
        # remaining 192: header fits, a 200-byte payload does not
        assert lg.append(2, 99, 9, orc.SEND, b"z" * 200) == 8
        off = lg.offsets()
        assert (off["tail"], off["end"]) == (0, 264)
        ring = lg.ring()
        # stale header left at 832 with the new idx / type / len
        assert int.from_bytes(ring[832:840].tobytes(), "little") == 8
        assert ring[832 + 26] == orc.SEND
        assert int.from_bytes(ring[832 + 48:832 + 50].tobytes(), "little") == 200
        # reader: entry at 768 is idx 7; at 832 the stale header says "does not fit"
        assert lg.get_entry(832) == 832 and length - 832 < lg.entry_len_at(832)
    same_state(o, r)


def test_exact_fit_restarts_index_q13():
    length = 1024
    o, r = both(length)
    for lg in (o, r):
        for k in range(8):                                     # 8 x 128 = 1024 exactly
            assert lg.append(1, k + 1, 5, orc.SEND, b"q" * 64) == k + 1
        assert lg.offsets()["end"] == length                   # reads as "empty"
        lg.set_offsets(head=512, apply=512, commit=512)
        assert lg.append(1, 9, 5, orc.SEND, b"q" * 64) == 1    # index restarts
        assert lg.append(1, 10, 5, orc.SEND, b"q" * 64) == 2
        off = lg.offsets()
        assert (off["tail"], off["end"]) == (128, 256)
    same_state(o, r)


def test_log_full_drops_entry():
    length = 1024
    o, r = both(length)
    for lg in (o, r):
        for k in range(4):
            lg.append(1, k + 1, 5, orc.SEND, b"f" * 64)        # end = 512
        lg.set_offsets(head=512)
        before = lg.offsets()
        assert lg.append(1, 5, 5, orc.SEND, b"f" * 64) == 0
        assert lg.offsets()["end"] == before["end"]
    same_state(o, r)


def test_control_entries_are_64_bytes():
    o, r = both(4096)
    cid = orc.cid_bytes(3, 5, 0, 0, 0b11111)
    for lg in (o, r):
        assert lg.append(4, 0, 0, orc.CONFIG, cid) == 1
        assert lg.append(4, 0, 0, orc.HEAD, 0x1234) == 2
        assert lg.append(4, 0, 0, orc.NOOP) == 3
        assert lg.offsets()["end"] == 192
        for off in (0, 64, 128):
            assert lg.entry_len_at(off) == 64
        ring = lg.ring()
        assert ring[48:64].tobytes() == cid
        assert int.from_bytes(ring[64 + 48:64 + 56].tobytes(), "little") == 0x1234
    same_state(o, r)


def test_prev_head_flag_rules():
    o, r = both(4096)
    for lg in (o, r):
        lg.prev_head = 1
        lg.append(1, 0, 0, orc.HEAD, 0)
        assert lg.prev_head == 1          # a HEAD append does not clear it
        lg.append(1, 1, 1, orc.CONNECT)
        assert lg.prev_head == 0          # anything else does (dare_log.h:477-480)


@pytest.mark.parametrize("seed", range(12))
def test_random_differential(seed):
    rng = random.Random(seed)
    length = rng.choice([1024, 2048, 4096, 8192, 65536])
    o, r = both(length)
    sizes = [0, 1, 13, 14, 15, 16, 40, 64, 100, 107, 128, 200, 256, 500]
    term = 1
    consumed = None
    for step in range(rng.randrange(200, 600)):
        act = rng.random()
        if act < 0.70:
            t = rng.choice([orc.SEND] * 6 + [orc.CONNECT, orc.CLOSE, orc.NOOP, orc.HEAD, orc.CONFIG, orc.CSM])
            if t in (orc.SEND, orc.CSM):
                n = rng.choice(sizes)
                n = min(n, length // 4)
                data = bytes(rng.randrange(256) for _ in range(n))
            elif t == orc.CONFIG:
                data = orc.cid_bytes(rng.randrange(4), 5, 0, 0, rng.randrange(32))
            elif t == orc.HEAD:
                data = rng.randrange(length)
            else:
                data = None
            elen = 64 + (len(data) if t in (orc.SEND, orc.CSM) else 0)
            off = o.offsets()
            used = 0 if off["end"] == off["len"] else o.end_distance(off["head"])
            if used + 2 * elen + 64 > length:
                # never let an append run over the head (the reference only detects
                # end == head exactly, dare_log.h:168): consume everything first
                for lg in (o, r):
                    e = lg.offsets()["end"]
                    lg.set_offsets(head=lg.get_tail(), apply=e, commit=e)
            args = (term, rng.randrange(1 << 40), rng.randrange(1 << 16), t, data)
            assert o.append(*args) == r.append(*args)
        elif act < 0.85:
            # consume: move head/apply/commit forward to an entry boundary the way the
            # state machine would (walk from commit with the reader rule)
            off = o.offsets()
            if off["end"] != off["len"]:
                nc = r.to_ncbuf(0)
                assert nc == o.to_ncbuf()
                if nc:
                    k = rng.randrange(len(nc))
                    tgt = nc[k][2]
                    move_head = rng.random() < 0.5
                    for lg in (o, r):
                        lg.set_offsets(head=tgt if move_head else lg.offsets()["head"],
                                       apply=tgt, commit=tgt)
        elif act < 0.90:
            for lg in (o, r):
                lg.set_offsets(tail=lg.offsets()["len"])        # hb_receive_cb resets the tail
            assert o.get_tail() == r.get_tail()
        elif act < 0.95:
            term += 1
        else:
            a, b = rng.randrange(length + 1), rng.randrange(length + 1)
            assert o.end_distance(a) == r.end_distance(a)
            assert o.is_larger(a, b) == r.is_larger(a, b)
            nc = r.to_ncbuf(0)
            assert nc == o.to_ncbuf()
            if nc:
                # perturb a determinant the way a diverged follower would
                cut = rng.randrange(len(nc) + 1)
                dets = list(nc[:cut])
                if cut < len(nc) and rng.random() < 0.5:
                    d = nc[cut]
                    dets.append((d[0], d[1] + 1, d[2]))
                assert o.find_remote_end(dets) == r.find_remote_end(dets)
        same_state(o, r, ring=(step % 16 == 0))
    same_state(o, r)


def test_full_size_case2_wrap_known_answer():
    """1 NOOP + 524287 x 128 B leaves 64 B; the next 128-B
    append wraps (case 2) and gets idx 524289 at offset 0."""
    o, r = both(L64M)
    payload = bytes(range(64))
    for lg in (o, r):
        assert lg.append(1, 0, 0, orc.NOOP) == 1
        last = 0
        for k in range(524287):
            last = lg.append(1, k, 1, orc.SEND, payload)
            if k == 400000:
                lg.set_offsets(head=1 << 20, apply=1 << 20, commit=1 << 20)
        assert last == 524288
        assert lg.offsets()["end"] == 67108800
        assert lg.append(1, 7, 1, orc.SEND, payload) == 524289
        off = lg.offsets()
        assert (off["tail"], off["end"]) == (0, 128)
        ring = lg.ring()
        assert int.from_bytes(ring[67108800:67108808].tobytes(), "little") == 524289
        assert int.from_bytes(ring[0:8].tobytes(), "little") == 524289
    assert o.offsets() == r.offsets()
