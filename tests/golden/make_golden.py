#!/usr/bin/env python3
"""Generate tests/golden/digests.json.

log_cases     : op sequences replayed on the REFERENCE's own dare_log.h (oracle/_ref,
                compiled unchanged from /root/reference) -- return values, final
                offsets and SHA-256 of the whole ring are the golden answers.
cluster_cases : small consensus traces; answers come from the oracle (the loops
                cannot be built from the reference here).

Run in the build container (needs /root/reference): python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from apus_amd import trace as T      # noqa: E402
from oracle import oracle as orc     # noqa: E402


def build_log_case(seed, length):
    rng = random.Random(seed)
    r = orc.RefLog(length)
    ops, results = [], []
    term = 1
    for _ in range(rng.randrange(120, 260)):
        off = r.offsets()
        used = 0 if off["end"] == off["len"] else r.end_distance(off["head"])
        t = rng.choice([orc.SEND] * 6 + [orc.CONNECT, orc.CLOSE, orc.NOOP, orc.HEAD, orc.CONFIG])
        if t == orc.SEND:
            data = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 14, 16, 40, 64, 107, 200])))
        elif t == orc.CONFIG:
            data = orc.cid_bytes(rng.randrange(3), 5, 0, 0, rng.randrange(32))
        elif t == orc.HEAD:
            data = rng.randrange(length)
        else:
            data = None
        elen = 64 + (len(data) if t == orc.SEND else 0)
        if used + 2 * elen + 64 > length:
            e = off["end"]
            kw = dict(head=r.get_tail(), apply=e, commit=e)
            r.set_offsets(**kw)
            ops.append(["set", kw])
        if rng.random() < 0.05:
            term += 1
        req_id, clt_id = rng.randrange(1 << 40), rng.randrange(1 << 16)
        results.append(r.append(term, req_id, clt_id, t, data))
        ops.append(["append", term, req_id, clt_id, t, data.hex() if isinstance(data, bytes) else data])
    return dict(len=length, ops=ops, results=results, final_offsets=r.offsets(),
                ring_sha256=hashlib.sha256(r.ring().tobytes()).hexdigest())


def cluster_case(n, n_send, payload, conns, batch, log_len, seed):
    tr = T.steady_trace(n, n_send, tuple(payload), conns, tuple(batch), log_len=log_len, seed=seed)
    cl = orc.run_trace(tr)
    o = cl.log(0).offsets()
    stream, cnt = orc.canon(cl.log(0).ring(), o["end"], o["head"], o["commit"])
    return dict(n=n, n_send=n_send, payload=list(payload), conns=conns, batch=list(batch), log_len=log_len,
                seed=seed, leader_offsets=o, n_entries=cnt, canon_sha256=hashlib.sha256(stream).hexdigest(),
                follower_apply_hash=cl.apply_hash(1))


def main():
    assert orc.have_ref(), "needs /root/reference (oracle/_ref)"
    out = {"generator": "tests/golden/make_golden.py",
           "log_cases": [build_log_case(s, L) for s, L in [(1, 1024), (2, 4096), (3, 65536), (4, 2048)]],
           "cluster_cases": [cluster_case(3, 700, (64,), 4, (1, 32), 1 << 14, 11),
                             cluster_case(5, 900, (64, 107, 40, 1024), 8, (1, 64), 1 << 16, 12),
                             cluster_case(7, 500, (4096, 64), 8, (1, 16), 1 << 18, 13)]}
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(HERE, "digests.json"))


if __name__ == "__main__":
    main()
