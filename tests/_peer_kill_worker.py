"""Worker of tests/test_gpu_peers_kill.py: one rank of a peer-mapped group whose LEADER PROCESS dies with tickets in flight.

Rank 0 leads, starts the replica kernels together with the others, issues a long stretch of device-resident rounds as a
queue of host commands -- and `os._exit`s as soon as the first command has been carried out, the rest still queued, rounds
in every stage of the pipeline, doorbells rung out of order by its append wavefronts, ACKs on their way.  Nobody drains,
nobody parks.  The survivors notice that the process is gone, ask their own workgroups to leave, and carry on among
themselves: the one that holds more wins the next term on its device (k_elect casts the votes, log adjustment + catch-up
through the mappings), a second stretch of rounds runs, everything is compared.

What every survivor checks, from its own memory:
 * contiguity: the entries it holds are idx 1 .. n without a gap, its end is a round boundary of the dead leader's;
 * the commit the dead leader rang (R4 doorbell, as rung) never exceeded what the most up-to-date survivor holds in order:
   with cumulative in-order ACKs an entry commits only behind a majority that holds everything in front of it
   (round 3's per-round ACKs could promise a round no survivor had behind a contiguous log);
 * its replica against an oracle that walks the schedule the crash really had (tests/_cluster.py:oracle_replay's rule: the
   server that held less is cut off behind its last round, released for the election, caught up by the winner)."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def main():
    out_path, n_send, ra, rb, grid_a, grid_f = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    from apus_amd import peers
    from apus_amd import trace as T
    from oracle import oracle as orc
    rank = int(os.environ.get("RANK", "0"))
    res = {"rank": rank, "ok": False}
    try:
        import datetime
        rank, world, local, backend = peers.init_process_group_from_env(0, timeout=datetime.timedelta(seconds=120))
        assert world == 3
        sub = dist.new_group(ranks=[1, 2])                  # (made while everybody lives: the survivors' barriers run on it)
        log_len = 1 << 27
        tr = T.steady_trace(3, n_send, 64, 64, 64, log_len=log_len, prune_bytes=1 << 60, name="kill_in_flight")
        rounds = [(e[1], e[2]) for e in tr.events if e[0] == "ROUND"]
        assert all(n == 64 for _, n in rounds) and ra + rb <= len(rounds)
        m = peers.PeerMember(3, rank, local, log_len)
        e = m.eng
        e.stage_trace(tr)
        pids = [torch.zeros(1, dtype=torch.int64) for _ in range(3)]
        dist.all_gather(pids, torch.tensor([os.getpid()], dtype=torch.int64))
        leader_pid = int(pids[0].item())
        m.elect(0)
        m.rep_begin(grid_a, grid_f, idle_ms=60000, peer_ms=5000)
        if rank == 0:
            # ---- the leader: a queue of commands, then gone
            chunk = max(1, ra // 64)
            first_cmd = e.rep_stats()["cmd_head"]
            for r0 in range(0, ra, chunk):
                e.rep_run(r0, min(chunk, ra - r0))
            t0 = time.time()
            while e.rep_stats()["cmd_head"] < first_cmd + 1 and time.time() - t0 < 20:
                pass
            os._exit(0)                                     # no drain, no park, no goodbye
        # ---- a survivor
        t0 = time.time()
        while time.time() - t0 < 60:
            try:
                os.kill(leader_pid, 0)
                st = open(f"/proc/{leader_pid}/stat").read().rsplit(")", 1)[1].split()[0]
                if st in ("Z", "X"):
                    break
            except OSError:
                break
            time.sleep(0.001)
        else:
            raise AssertionError("the leader's process did not go away")
        e._chk(e.L.apus_gpu_rep_follower_stop(e.h, rank), "follower_stop")
        code = e.rep_park()
        m.rep_running = m.rep_here = False
        m.pg = sub
        w = (C.c_uint64 * 8)()
        e._chk(e.L.apus_gpu_rep_box_words(e.h, rank, 0, w), "box_words")
        bell = int(w[0])
        le = (C.c_uint64 * 4)()
        e._chk(e.L.apus_gpu_last_entry(e.h, rank, le), "last_entry")
        mine = torch.tensor([int(le[2]), bell, code], dtype=torch.int64)
        both = [torch.zeros(3, dtype=torch.int64) for _ in range(2)]
        dist.all_gather(both, mine, group=sub)
        held = {1: int(both[0][0]), 2: int(both[1][0])}
        bells = {1: int(both[0][1]), 2: int(both[1][1])}
        res.update(held=held, bells=bells, park_code=code)
        # contiguity of what this survivor holds: 1 CONFIG entry + whole rounds of 64, idx 1 .. n
        n_mine = held[rank]
        assert (n_mine - 1) % 64 == 0, f"rank {rank}: holds {n_mine} entry slots: not a round boundary of the leader's"
        o = e.offsets(rank)
        ring = e.ring(rank)
        sizes = 64 + tr.reqs["len"][:n_mine - 1].astype(np.int64)
        pos = np.concatenate([[0, 64], 64 + np.cumsum(sizes)])          # the blank CONFIG entry, then the requests in order
        assert o["end"] == int(pos[-1]), (o, n_mine, int(pos[-1]))
        idx = np.array([int(ring[p:p + 8].view(np.uint64)[0]) for p in pos[:-1][:: max(1, (n_mine) // 4096)]], dtype=np.uint64)
        want = np.arange(1, n_mine + 1, dtype=np.uint64)[:: max(1, (n_mine) // 4096)]
        assert np.array_equal(idx, want), f"rank {rank}: the idx sequence has a gap"
        # the commit the dead leader promised vs what a survivor holds in order
        top = max(held.values())
        assert all(b <= top for b in bells.values()), f"a commit doorbell ({bells}) beyond what the most up-to-date survivor holds ({held})"
        winner = 1 if held[1] >= held[2] else 2
        lag = 3 - winner
        k_w, k_l = (held[winner] - 1) // 64, (held[lag] - 1) // 64
        res.update(winner=winner, rounds_winner=k_w, rounds_lag=k_l, in_flight=bool(k_w < ra))
        # ---- the survivors carry on
        m.kill(0)
        m.elect(winner)
        m.rep_begin(grid_a, grid_f, idle_ms=60000, peer_ms=5000)
        m.rep_rounds(ra, rb)
        m.rep_end()
        m.quiesce()
        m.settle()
        # ---- the oracle under the schedule the crash had
        cl = orc.Cluster(3, log_len)
        reqs = np.ascontiguousarray(tr.reqs, dtype=orc.REQ_DTYPE)
        cl.elect(0)
        for k in range(k_l):
            cl.round(reqs[rounds[k][0]:rounds[k][0] + 64], tr.arena)
        if k_l < k_w:
            cl.hold(lag)
        for k in range(k_l, k_w):
            cl.round(reqs[rounds[k][0]:rounds[k][0] + 64], tr.arena)
        cl.kill(0)
        if k_l < k_w:
            cl.release(lag)
        cl.elect(winner)
        for k in range(ra, ra + rb):
            cl.round(reqs[rounds[k][0]:rounds[k][0] + 64], tr.arena)
        cl.quiesce()
        go, oo = e.offsets(rank), cl.log(rank).offsets()
        assert go == oo, f"rank {rank}: offsets differ\n gpu={go}\n orc={oo}"
        ring_g, ring_o = e.ring(rank), cl.log(rank).ring()
        mask = orc.defined_mask(ring_o, oo["end"], oo["head"], oo["end"])
        d = np.nonzero((ring_g != ring_o) & mask)[0]
        assert len(d) == 0, f"rank {rank}: {len(d)} defined ring bytes differ from the oracle, first at {d[:8].tolist()} gpu={ring_g[d[:8]].tolist()} orc={ring_o[d[:8]].tolist()}"
        assert e.counters(rank)["sid"] == cl.sid(rank)
        e.check_status()
        res["ok"] = True
        res["end"] = go["end"]
        m.check_done()
    except BaseException as exc:      # noqa: BLE001
        res["error"] = repr(exc) + "\n" + traceback.format_exc()[-2500:]
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(res, f)
    os._exit(0 if res["ok"] else 1)       # (the default process group still names a rank that is gone: no orderly teardown)


if __name__ == "__main__":
    main()
