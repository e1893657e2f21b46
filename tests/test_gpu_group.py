"""Multi-process replica group (one process per replica, the --gpus N path of
bench.py) against the oracle.  The GPU box has a single device, so every rank uses
GPU 0 and the exchange is staged through gloo; on a multi-GPU node the same code
runs with backend nccl (RCCL) and device tensors."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_send,log_len", [(2, 1500, 1 << 17), (3, 2500, 1 << 18)])
def test_group_of_processes_matches_oracle(world, n_send, log_len):
    out = os.path.join(tempfile.mkdtemp(), "res")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_group_worker.py"), out, str(n_send), str(log_len)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    res = []
    for r in range(world):
        path = f"{out}.{r}"
        assert os.path.exists(path), f"rank {r} produced no result\n{p.stdout[-2000:]}\n{p.stderr[-3000:]}"
        res.append(json.load(open(path)))
    for r in res:
        assert r["ok"], f"rank {r['rank']}: {r.get('error')}\n{p.stderr[-2000:]}"
    assert len({r["end"] for r in res}) == 1


@pytest.mark.parametrize("n", [2, 3, 4, 5, 7, 8])
def test_bench_gpus_n_dry_run_full_schema(n):
    """`python bench.py --gpus N` as the driver starts it on an 8-GPU node, here in the one-device test mode (every
    rank on GPU 0, handles over gloo): N -> 1 / 3 / 5 / 7 replicas (an even N keeps a spare machine that JOINs), every
    process runs its replica's own kernels, and rank 0's line carries the whole schema -- placement, link calibration,
    the xGMI roofline with the bytes shipped per link, the lone-round latency, the join, the CPU baseline -- verified."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--entries", "131072", "--cpu-seconds", "1",
           "--watchdog", "240"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", APUS_DIST_BACKEND="gloo", APUS_DIST_ONE_DEVICE="1", APUS_SELFTEST_ROUNDS="100000")
    if n == 5:
        env["APUS_SELFTEST_FORCE_NO_FAST_ACK"] = "1"       # one size walks the group without REP_FAST_ACK
    if n == 4:
        env["APUS_SELFTEST_FORCE_FALLBACK"] = "1"          # one size walks the fall-back: the group starts again with fine-grained rings
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, f"rc={p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-4000:]}"
    d = json.loads(lines[0])
    replicas = n if n % 2 else n - 1
    st = d["ring_visibility_selftest"]
    if replicas >= 2:
        # first contact: every follower's resident kernel checked every round its leader's device pushed, nothing differed
        assert st["verdict"] == "ok" and len(st["by_follower"]) == replicas - 1
        assert all(v["rounds"] == 100000 and v["bad_units"] == 0 and not v["timeouts"] and not v["pusher_timeouts"] for v in st["by_follower"].values()), st
        # ... and the follower's own ACK of a lone round (a system-scope atomic max into the leader's mailbox) is on when the
        # atomics of first contact all landed in order, off for the whole group when told they did not
        assert st["fast_ack"] == "on" if n != 5 else st["fast_ack"].startswith("off (APUS_REP_DBG & 65536)"), st["fast_ack"]
        assert st["retested"] == (n == 4) and ("fine-grained" in st["allocation"]) == (n == 4) and st["allocation"].split(" (")[0] in d["config"]["mode"]
        # the smaller groups BASELINE names, in the same run (1 / 3 / 5 replicas below the headline's), every one verified
        want = {str(k) for k in (1, 3, 5, 7) if k <= replicas}
        assert want <= set(d["by_group_size"]) and all(d["by_group_size"][k]["verified"] and d["by_group_size"][k]["entries_per_s"] > 0 for k in want), d["by_group_size"]
        # the same group over send / recv (gloo staging here, RCCL on the real node), verified
        assert d["rccl_transport"]["verified"] is True and d["rccl_transport"]["value"] > 0
    else:
        assert st is None and "rccl_transport" not in d
    assert d["n_gpus"] == n and d["config"]["replicas"] == replicas and d["config"]["spare_machines"] == n - replicas
    assert d["verified"] is True and d["value"] > 0 and d["metric"] == "committed entries/sec" and d["scaling"] == "weak"
    assert d["roofline"]["bound"] == "xgmi" and d["roofline"]["kernel"] == "k_replica" and 0 < d["roofline"]["bytes_per_entry"] < 64 + 64 + 3
    assert d["placement"]["ranks_in_group"] == n and len(d["placement"]["ranks"]) == n and d["placement"]["peer_access_matrix"]
    cal = d["link_calibration"]
    assert cal and all(v and v > 0 for v in cal["doorbell_round_trip_us_p50"].values()) and len(cal["doorbell_round_trip_us_p50"]) == n - 1
    assert cal["peer_store_peak_GBps"] and cal["peer_store_peak_GBps"] > 1
    if replicas > 1:
        assert d["latency"]["appended_to_committed_and_applied_us_p50"] and d["p50_round_latency_us"] > 0
    if n % 2 == 0 and replicas >= 3:
        assert d["join_catch_up"] and d["join_catch_up"]["log_bytes"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    print(f"--gpus {n}: {d['value'] / 1e6:.0f} M entries/s, p50 {d['p50_round_latency_us']} us, doorbell rt {cal['doorbell_round_trip_us_p50']}, "
          f"peer store peak {cal['peer_store_peak_GBps']:.0f} GB/s, join {d['join_catch_up']}")


def test_bench_gpus_extras_are_cut_off_not_the_line():
    """`bench.py --gpus N`: what is measured below the headline (smaller groups, the send / recv transport) may never cost the
    driver its line -- with a budget they cannot meet, rank 0 prints the line as it stands, says so, and every rank leaves with
    status 0."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=5",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "5", "--steps", "2", "--warmup", "1", "--entries", "131072", "--cpu-seconds", "1",
           "--watchdog", "240", "--extras-timeout", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", APUS_DIST_BACKEND="gloo", APUS_DIST_ONE_DEVICE="1", APUS_SELFTEST_ROUNDS="100000")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, f"rc={p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-3000:]}"
    d = json.loads(lines[0])
    assert d["verified"] is True and d["value"] > 0 and d["config"]["replicas"] == 5 and "cut off" in d["extras"]
    assert "rccl_transport" not in d


def test_bench_gpus_group_failure_falls_back_to_independent_groups():
    """`bench.py --gpus N`: when the cross-GPU group RAISES (told to here; on a real node: an IPC / fabric layer that answers
    differently from the one device everything was developed on), the driver still gets ONE line and status 0 -- N independent
    single-GPU groups, each rank on its own device without any collective, marked as the fallback it is, rank 0's run
    bit-exact against the oracle."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--entries", "131072", "--cpu-seconds", "1",
           "--watchdog", "240"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", APUS_DIST_BACKEND="gloo", APUS_DIST_ONE_DEVICE="1", APUS_BENCH_FORCE_GROUP_FAILURE="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, f"rc={p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-4000:]}"
    d = json.loads(lines[0])
    assert d["fallback"] is True and d["verified_cross_gpu"] is False and "APUS_BENCH_FORCE_GROUP_FAILURE" in d["group_failure"]
    assert d["n_gpus"] == 2 and d["config"]["groups"] == 2 and d["config"]["ranks_reporting"] == [0, 1] and d["config"]["replicas"] == 3
    assert d["verified"] is True and d["value"] > 0 and d["metric"] == "committed entries/sec" and d["scaling"] == "weak"
    assert d["by_rank"][0]["bit_exact_vs_oracle"] is True and all(r["verified"] for r in d["by_rank"])
    assert d["config"]["workload"].startswith("FALLBACK") and d["cpu_baseline"]["value"] > 0


@pytest.mark.parametrize("where,n", [("1", 2), ("resident", 3)])
def test_bench_gpus_group_hang_falls_back_to_independent_groups(where, n):
    """... and when the cross-GPU group HANGS (every rank told to sit in bench_multi for good -- right behind the process group's
    start, or with a three-replica group's workgroups resident and every peer's rings mapped): at --watchdog seconds every rank
    dumps its stacks and replaces itself (exec) by a process that goes straight to the same last resort -- one line, status 0."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--entries", "131072", "--cpu-seconds", "1",
           "--watchdog", "12" if where == "1" else "30", "--no-calibration"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", APUS_DIST_BACKEND="gloo", APUS_DIST_ONE_DEVICE="1", APUS_BENCH_FORCE_GROUP_HANG=where,
               APUS_SELFTEST_ROUNDS="100000")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, f"rc={p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-4000:]}"
    d = json.loads(lines[0])
    assert d["fallback"] is True and d["verified_cross_gpu"] is False and d["group_failure"].startswith("RuntimeError: watchdog")
    assert d["n_gpus"] == n and d["config"]["ranks_reporting"] == list(range(n)) and d["verified"] is True and d["value"] > 0
    assert d["by_rank"][0]["bit_exact_vs_oracle"] is True
