"""Comparison of two clusters that duck-type oracle.Cluster (the restated oracle, or the
reference-as-is cluster of oracle/refloops.py), and the state record that is committed as a
golden fixture (tests/golden/cluster_ref.json, written from the REFERENCE's outputs)."""
from __future__ import annotations

import hashlib

import numpy as np

from oracle import oracle as orc

APPLY_FIELDS = ("off", "idx", "len", "clt_id", "type", "kind")


def replica_record(cl, r) -> dict:
    """Everything that is compared for one server, hashed where it is big."""
    lg = cl.log(r)
    o = lg.offsets()
    ring = lg.ring()
    rec = {"offsets": o, "sid": cl.sid(r), "highest_rec": cl.highest_rec(r), "store_count": cl.store_count(r),
           "apply_count": cl.apply_count(r), "prev_head": int(lg.prev_head)}
    if o["end"] != o["len"]:
        mask = orc.defined_mask(ring, o["end"], o["head"], o["end"])
        rec["defined_bytes"] = int(mask.sum())
        rec["defined_sha256"] = hashlib.sha256(np.where(mask, ring, 0).astype(np.uint8).tobytes()).hexdigest()
        stream, n = orc.canon(ring, o["end"], o["head"], o["commit"])
        rec["canon_entries"] = n
        rec["canon_sha256"] = hashlib.sha256(stream).hexdigest()
    al = cl.apply_log(r)
    h = hashlib.sha256()
    for f in APPLY_FIELDS:
        h.update(np.ascontiguousarray(al[f]).astype("<u8").tobytes())
    rec["apply_records"] = len(al)
    rec["apply_sha256"] = h.hexdigest()
    # the durability side channel: what proxy_store_cmd handed to BerkeleyDB (proxy.c:268-291), back to back
    st = cl.store_stream(r)
    rec["store_bytes"] = len(st)
    rec["records_len"] = cl.records_len(r)
    rec["store_sha256"] = hashlib.sha256(st).hexdigest()
    return rec


def cluster_record(cl, n) -> dict:
    com, end = cl.round_record()
    return {"leader": cl.leader, "replicas": [replica_record(cl, r) for r in range(n)],
            "rounds": len(com),
            "round_sha256": hashlib.sha256(np.ascontiguousarray(com).astype("<u8").tobytes() +
                                           np.ascontiguousarray(end).astype("<u8").tobytes()).hexdigest()}


def assert_same_state(a, b, n, tag="", names=("oracle", "reference")):
    """Field-by-field comparison with readable failures (offsets, every defined ring byte,
    ids, counters, apply upcalls, per-round end/commit record of the leader)."""
    for r in range(n):
        if getattr(b, "gone", lambda r: False)(r):
            continue            # the reference's server shut itself down ("Somebody removed me") and freed its log
        oa, ob = a.log(r).offsets(), b.log(r).offsets()
        assert oa == ob, f"{tag} server {r}: offsets differ\n {names[0]}={oa}\n {names[1]}={ob}"
        if oa["end"] != oa["len"]:
            ra, rb = a.log(r).ring(), b.log(r).ring()
            ma = orc.defined_mask(ra, oa["end"], oa["head"], oa["end"])
            mb = orc.defined_mask(rb, ob["end"], ob["head"], ob["end"])
            assert np.array_equal(ma, mb), f"{tag} server {r}: entry boundaries differ"
            d = np.nonzero((ra != rb) & ma)[0]
            assert len(d) == 0, (f"{tag} server {r}: {len(d)} defined ring bytes differ, first at {d[:8].tolist()} "
                                 f"{names[0]}={ra[d[:8]].tolist()} {names[1]}={rb[d[:8]].tolist()}")
        for f in ("sid", "highest_rec", "store_count", "apply_count"):
            va, vb = getattr(a, f)(r), getattr(b, f)(r)
            assert va == vb, f"{tag} server {r}: {f} {va} vs {vb}"
        assert int(a.log(r).prev_head) == int(b.log(r).prev_head), f"{tag} server {r}: prev_log_entry_head"
        sa, sb = a.store_stream(r), b.store_stream(r)
        assert a.records_len(r) == b.records_len(r), f"{tag} server {r}: records_len {a.records_len(r)} vs {b.records_len(r)}"
        assert sa == sb, f"{tag} server {r}: the bytes handed to the storage callback differ ({len(sa)} vs {len(sb)} bytes)"
        aa, ab = a.apply_log(r), b.apply_log(r)
        assert len(aa) == len(ab), f"{tag} server {r}: {len(aa)} vs {len(ab)} apply upcalls"
        for f in APPLY_FIELDS:
            assert np.array_equal(aa[f], ab[f]), f"{tag} server {r}: apply upcall field {f} differs"
    ca, ea = a.round_record()
    cb, eb = b.round_record()
    assert len(ca) == len(cb), f"{tag}: {len(ca)} vs {len(cb)} leader passes recorded"
    bad = np.nonzero((ca != cb) | (ea != eb))[0]
    assert len(bad) == 0, f"{tag}: per-pass end/commit differs at passes {bad[:8].tolist()}"
    assert a.leader == b.leader, f"{tag}: leader {a.leader} vs {b.leader}"
