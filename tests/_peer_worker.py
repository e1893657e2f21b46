"""Worker of tests/test_gpu_peers.py: one rank of a peer-mapped replica group (apus_amd/peers.py).
Every rank hosts its replica in its own allocation and maps the others' over HIP IPC; the GPU box
has one device, so all ranks use GPU 0 (the peer stores then stay on the device; on a multi-GPU
node they cross xGMI).  torch.distributed (gloo here) carries the handle exchange and barriers.

Each rank walks its own copy of the oracle next to the trace and compares at every check point:
 * its OWN replica, read from its own memory (offsets, every defined ring byte, canonical
   digest, sid, counters, apply-stream hash) -- nothing on the host told it what to expect;
 * if it leads: every reachable replica through the mappings, and the per-pass end/commit record
   of the terms it led."""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch.distributed as dist


def main():
    out_path, name, mode = sys.argv[1], sys.argv[2], sys.argv[3]
    flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    repeat = int(sys.argv[5]) if len(sys.argv) > 5 else 1      # the whole walk, `repeat` times in one process group (soak runs)
    from apus_amd import peers
    from oracle import oracle as orc
    from tests import traces
    from tests.parity import compare_replica
    rank = int(os.environ.get("RANK", "0"))
    res = {"rank": rank, "ok": False, "checks": 0, "runs": 0}
    m = None
    try:
        rank, world, local, backend = peers.init_process_group_from_env(0)
        for it in range(repeat):
            m = run_once(peers, orc, traces, compare_replica, name, mode, flags, rank, world, local, res, it)
            res["runs"] = it + 1
            if it + 1 < repeat:
                m.close()
        res["ok"] = True
        res["end"] = m.eng.offsets(rank)["end"]
        res["led"] = len(m.led)
    except BaseException as e:      # noqa: BLE001
        res["error"] = repr(e) + "\n" + traceback.format_exc()[-1500:]
        with open(f"{out_path}.{rank}", "w") as f:
            json.dump(res, f)
        os._exit(1)                 # peers may sit in a barrier: let the launcher tear the group down
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(res, f)
    m.close()
    dist.destroy_process_group()


def run_once(peers, orc, traces, compare_replica, name, mode, flags, rank, world, local, res, it):
    tr = {**traces.CATALOGUE, **traces.EXTRA}[name]()
    assert tr.group_size <= world       # the other ranks are machines that JOIN later
    m = peers.PeerMember(world, rank, local, tr.log_len, flags=flags, configured=tr.group_size)
    cl = orc.Cluster(tr.group_size, tr.log_len, record_apply=True)
    reqs = np.ascontiguousarray(tr.reqs, dtype=orc.REQ_DTYPE)
    pos = 0
    led = []                 # [first pass, last pass) of the oracle's record for the terms this rank led

    def oracle_to(i):
        nonlocal pos
        while pos <= i and pos < len(tr.events):
            ev = tr.events[pos]
            op = ev[0]
            if op == "ROUND":
                cl.round(reqs[ev[1]:ev[1] + ev[2]], tr.arena)
            elif op == "ELECT":
                n0 = len(cl.round_record()[0])
                if led and led[-1][1] is None:
                    led[-1][1] = n0
                cl.elect(ev[1])
                if ev[1] == rank:
                    led.append([n0, None])
            elif op == "KILL":
                if led and led[-1][1] is None and ev[1] == rank:
                    led[-1][1] = len(cl.round_record()[0])
                cl.kill(ev[1])
            else:
                getattr(cl, {"PRUNE": "tick_prune", "QUIESCE": "quiesce", "HOLD": "hold", "RELEASE": "release",
                             "JOIN": "join"}[op])(*ev[1:])
            pos += 1

    def record_expected():
        oc, oe = cl.round_record()
        parts = [(a, len(oc) if b is None else b) for a, b in led]
        if not parts:
            return oc[:0], oe[:0]
        return (np.concatenate([oc[a:b] for a, b in parts]), np.concatenate([oe[a:b] for a, b in parts]))

    def check(i, ev, mm):
        oracle_to(i)
        e = mm.eng
        tag = f"{name} run {it} rank {rank} event {i} {ev}"
        alive = [r for r in range(cl.n) if (e.reachable >> r) & 1 and (e.bitmask >> r) & 1]
        if os.environ.get("APUS_PEER_SLOW_RANK") == str(rank):
            import time
            time.sleep(0.2)          # (diagnostic: a rank that looks late; see PeerMember.check_done)
        if rank in alive:
            compare_replica(e, cl, rank, tag=tag)
        if mm.is_leader:
            e.check_status()
            for r in alive:
                compare_replica(e, cl, r, tag=tag + " (leader's view)")
        if mm.led and mode != "replica":          # (the replica kernels keep no per-pass record)
            gc, ge = e.round_record()
            oc, oe = record_expected()
            assert len(gc) == len(oc), f"{tag}: {len(gc)} passes recorded, oracle {len(oc)}"
            bad = np.nonzero((gc != oc) | (ge != oe))[0]
            assert len(bad) == 0, f"{tag}: per-pass end/commit differs at {bad[:8].tolist()}"
        res["checks"] += 1

    peers.walk_trace(m, tr, on_check=check, check_at=("QUIESCE", "PRUNE") if mode == "per-call" else ("QUIESCE",),
                     batch=(mode == "batched"), replica=(mode == "replica"), rep_grid=(24, 12))
    # settle both sides like tests/parity.py does at the end, then the final comparison
    oracle_to(len(tr.events) - 1)
    cl.quiesce(); m.quiesce(); m.settle()
    pos = len(tr.events)
    check(len(tr.events), ("END",), m)
    assert cl.force_prunes == 0
    m.check_done()
    return m


if __name__ == "__main__":
    main()
