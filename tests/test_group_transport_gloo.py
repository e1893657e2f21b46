"""world_size-2 gloo test (CPU) of the multi-process exchange: ring ranges (incl. the
two-piece wrapped case) and directory slots shipped with apus_amd.distributed's own
ship_range / recv_range land byte-exactly at the same offsets; the oracle provides
the leader's log."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from apus_amd import trace as T


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from apus_amd import distributed as D
        from oracle import oracle as orc
        L = 1 << 15
        tr = T.steady_trace(2, 900, (64, 107, 300), 4, (1, 32), log_len=L, seed=9)
        tr.events = tr.events[:150]
        # both ranks replay the oracle: rank 0 takes the leader's log as the thing to ship,
        # rank 1 the follower's log as the expected result
        cl = orc.Cluster(2, L)
        reqs = np.ascontiguousarray(tr.reqs, dtype=orc.REQ_DTYPE)
        tp = D.Transport("gloo", torch.device("cpu"))
        cap = 1 << 10
        ring = torch.zeros(L, dtype=torch.uint8)
        dir_off = torch.zeros(cap * 8, dtype=torch.uint8)
        dir_len = torch.zeros(cap * 4, dtype=torch.uint8)
        shipped_off, shipped_slot, slot = L, 0, 0
        wrapped_pieces = 0
        for ev in tr.events:
            if ev[0] == "ELECT":
                cl.elect(0)
                n_new = 1
            elif ev[0] == "ROUND":
                cl.round(reqs[ev[1]:ev[1] + ev[2]], tr.arena)
                n_new = ev[2]
            elif ev[0] == "PRUNE":
                n_new = cl.tick_prune()          # 1 when a HEAD entry was appended
            else:
                continue
            lead = cl.log(0)
            o = lead.offsets()
            if n_new == 0:
                continue
            if rank == 0:
                lr = torch.from_numpy(lead.ring().copy())
                # directory of the new entries (walk them the way the reader does)
                off = 0 if shipped_off == L else shipped_off
                do = dir_off.view(torch.int64)
                dl = dir_len.view(torch.int32)
                for k in range(n_new):
                    e = lead.get_entry(off)
                    off = e
                    ln = lead.entry_len_at(off)
                    if L - off < ln:
                        off = 0
                        ln = lead.entry_len_at(off)
                    do[(slot + k) % cap] = off
                    dl[(slot + k) % cap] = ln
                    off += ln
                D.ship_range(tp, 1, lr, dir_off, dir_len, shipped_off, o["end"], shipped_slot, slot + n_new,
                             L, cap, 0, 2)
            else:
                h = tp.recv_words(D.HDR_WORDS, 0)
                assert h[0] == D.OP_DATA and h[3] == shipped_slot and h[4] == slot + n_new
                if len(D.ring_pieces(h[1], h[2], L)) == 2:
                    wrapped_pieces += 1
                D.recv_range(tp, 0, ring, dir_off, dir_len, h[1], h[2], h[3], h[4], L, cap)
            slot += n_new
            shipped_off, shipped_slot = o["end"], slot
        if rank == 1:
            fol = cl.log(1)
            o = fol.offsets()
            mask = orc.defined_mask(fol.ring(), o["end"], o["head"], o["end"])
            got = ring.numpy()
            exp = fol.ring().copy()
            # reply bytes are written by the follower's own persist step, not by the transfer
            ok = bool((got[mask] == exp[mask]).sum() >= mask.sum() - 2 * slot)
            hg, n1 = orc.canon_hash(got, o["end"], o["head"], o["end"])
            he, n2 = orc.canon_hash(exp, o["end"], o["head"], o["end"])
            q.put((ok and hg == he and n1 == n2, wrapped_pieces, n1))
    finally:
        dist.destroy_process_group()


def test_ring_and_slot_pieces():
    from apus_amd.distributed import ring_pieces, slot_pieces
    assert ring_pieces(10, 50, 100) == [(10, 50)]
    assert ring_pieces(90, 20, 100) == [(90, 100), (0, 20)]
    assert ring_pieces(100, 20, 100) == [(0, 20)]          # follower log was empty
    assert ring_pieces(30, 30, 100) == []
    assert slot_pieces(5, 9, 8) == [(5, 8), (0, 1)]
    assert slot_pieces(16, 20, 8) == [(0, 4)]


def test_two_rank_gloo_exchange_matches_oracle():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, wrapped, n = q.get(timeout=5)
    assert ok, "follower ring differs from the oracle's follower"
    assert wrapped >= 1, "the test must exercise the two-piece wrapped transfer"
    assert n > 0
