"""Worker of tests/test_gpu_peers_deposed.py: one rank of a peer-mapped group whose LEADER IS DEPOSED BUT LIVES and keeps pushing.

Rank 0 leads term 2: the replica kernels run `ra` rounds, everything is committed everywhere -- and its resident kernel STAYS
(the host never parks it: a leader behind a partition does not know that it was voted out).  Ranks 1 and 2 ask their own
workgroups to leave, hold an election among themselves (rank 1 wins term 4), run `rb` rounds through the replica kernels and
park.  THEN the deposed leader issues `rc` more rounds: its append wavefronts push them through the mappings it holds -- at
the very offsets the new term's entries occupy in the survivors' logs.

The reference's voters reset the old leader's QPs, its WRITEs bounce (rc_revoke_log_access, dare_ibv_rc.c:2156-2243).  Here a
server that adopts a newer term LEAVES the ring and the mailbox the old leader has mapped (apus_gpu_fence_replica, called
by PeerMember.elect): the stale stores land in the allocation that was left.  Every survivor checks
 * its replica against the oracle's (elect 0, ra rounds, kill 0, elect 1, rb rounds): offsets, every defined ring byte, SID;
 * that the deposed leader's first stale entry IS in the ring it left (term 2, the request id rank 0 appended there): the
   stores arrived -- somewhere nobody reads.
APUS_PEER_NO_RING_FENCE=1 (diagnostic) skips the fence: the same walk then damages the survivors' logs, and the test that
sets it expects exactly that."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ctypes as C

import numpy as np
import torch.distributed as dist


def main():
    out_path, n_send, ra, rb, rc_ = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    from apus_amd import peers
    from apus_amd import trace as T
    from apus_amd.engine import EngineError
    from oracle import oracle as orc
    rank = int(os.environ.get("RANK", "0"))
    res = {"rank": rank, "ok": False}
    try:
        import datetime
        rank, world, local, backend = peers.init_process_group_from_env(0, timeout=datetime.timedelta(seconds=120))
        assert world == 3
        sub = dist.new_group(ranks=[1, 2])
        log_len = 1 << 26
        tr = T.steady_trace(3, n_send, 64, 64, 64, log_len=log_len, prune_bytes=1 << 60, name="deposed_leader")
        rounds = [(e[1], e[2]) for e in tr.events if e[0] == "ROUND"]
        assert all(n == 64 for _, n in rounds) and ra + rb + rc_ <= len(rounds)
        m = peers.PeerMember(3, rank, local, log_len)
        e = m.eng
        e.stage_trace(tr)
        m.elect(0)
        m.rep_begin(8, 4, idle_ms=60000, peer_ms=1500)
        if rank == 0:
            e.rep_run(0, ra)
            e.rep_drain(timeout_ms=30000)                   # term 2's rounds: committed by majority, applied
            # (the blank CONFIG entry of the election, then the requests in order: no wrap in this walk)
            end_a = 64 + int(np.sum(64 + tr.reqs["len"][:rounds[ra][0]].astype(np.int64)))
        dist.barrier()                                      # ---- the partition: rank 0 hears nothing from here on
        if rank == 0:
            dist.barrier()                                  # (the survivors hold their election and run term 4)
            # ---- the deposed leader pushes on: its kernel never left
            stale_first = int(tr.reqs["req_id"][rounds[ra + rb][0]])
            e.rep_run(ra + rb, rc_)
            t0 = time.time()
            pushed = False
            while time.time() - t0 < 20 and not pushed:     # until its own ring shows the stale rounds (its append wavefronts have stored)
                hdr = e.ring(0, end_a, 24).view(np.uint64)
                pushed = int(hdr[2]) == stale_first
                time.sleep(0.01)
            try:
                e.rep_drain(timeout_ms=4000)                # nobody acknowledges: no majority, no commit
            except EngineError:
                pass
            try:
                code = e.rep_park()
            except EngineError as exc:
                code = repr(exc)
            res.update(ok=bool(pushed), pushed=pushed, end_a=end_a, stale_first_req_id=stale_first, park_code=code)
            dist.barrier()
        else:
            e._chk(e.L.apus_gpu_rep_follower_stop(e.h, rank), "follower_stop")
            e.rep_park()
            m.rep_running = m.rep_here = False
            m.pg = sub
            m.kill(0)
            m.elect(1)                                      # <- the fence: ring + mailbox of ranks 1 and 2 move, both map the new ones
            m.rep_begin(8, 4, idle_ms=60000, peer_ms=1500)
            m.rep_rounds(ra, rb)
            m.rep_end()
            m.quiesce()
            m.settle()
            dist.barrier()                                  # ---- now the deposed leader pushes
            dist.barrier()                                  # ---- ... and has parked
            cl = orc.Cluster(3, log_len)
            reqs = np.ascontiguousarray(tr.reqs, dtype=orc.REQ_DTYPE)
            cl.elect(0)
            for k in range(ra):
                cl.round(reqs[rounds[k][0]:rounds[k][0] + 64], tr.arena)
            end_a = cl.log(rank).offsets()["end"]
            cl.kill(0)
            cl.elect(1)
            for k in range(ra, ra + rb):
                cl.round(reqs[rounds[k][0]:rounds[k][0] + 64], tr.arena)
            cl.quiesce()
            go, oo = e.offsets(rank), cl.log(rank).offsets()
            res.update(fenced=m.fenced, end_a=end_a, end=go["end"])
            assert go == oo, f"rank {rank}: offsets differ\n gpu={go}\n orc={oo}"
            ring_g, ring_o = e.ring(rank), cl.log(rank).ring()
            mask = orc.defined_mask(ring_o, oo["end"], oo["head"], oo["end"])
            d = np.nonzero((ring_g != ring_o) & mask)[0]
            res["ring_bytes_differing"] = int(len(d))
            assert len(d) == 0, f"rank {rank}: {len(d)} defined ring bytes differ from the oracle, first at {d[:8].tolist()} gpu={ring_g[d[:8]].tolist()} orc={ring_o[d[:8]].tolist()}"
            assert e.counters(rank)["sid"] == cl.sid(rank)
            e.check_status()
            if m.fenced:
                # where the stale stores went: the ring this replica left at the election
                buf = np.zeros(24, dtype=np.uint8)
                e._chk(e.L.apus_gpu_read_retired_ring(e.h, rank, 1, end_a, 24, buf.ctypes.data), "read_retired_ring")
                w = buf.view(np.uint64)
                res["retired_at_end_a"] = [int(x) for x in w]
                stale_first = int(tr.reqs["req_id"][rounds[ra + rb][0]])
                assert int(w[1]) == 2 and int(w[2]) == stale_first, \
                    f"rank {rank}: the ring left at the election does not hold the deposed leader's entry at {end_a}: idx/term/req_id = {w.tolist()} (want term 2, req_id {stale_first})"
                live = ring_g[end_a:end_a + 24].view(np.uint64)
                assert int(live[1]) == 4, f"rank {rank}: the live ring's entry at {end_a} is not the new term's: {live.tolist()}"
            res["ok"] = True
    except BaseException as exc:      # noqa: BLE001
        res["ok"] = False
        res["error"] = repr(exc) + "\n" + traceback.format_exc()[-2500:]
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(res, f)
    os._exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
