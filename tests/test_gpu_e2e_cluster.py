"""One PROCESS per server in C, end to end (benchmarks/run.sh + reconf_bench.sh:92-175,249-343 in small): three redis-server
2.8.17 processes, each under LD_PRELOAD=libapus_interpose.so with its own server_idx, its own engine hosting ITS replica,
the others' mapped over HIP IPC -- handles exchanged through a directory (APUS_GROUP_DIR), no Python, no torch.  The
leader's process runs the leader's replica kernels, each follower's process its follower kernels and replays what is
applied into ITS redis (do_action_to_server, proxy.c:341-439).  Then the leader's PROCESS is killed: the survivors notice,
park, the lowest live index wins the next term on the device (k_elect, log adjustment, the dead server removed), its redis
takes the clients over, the other survivor follows it.

Checked: every redis holds the same keys after each phase; the survivors' logs, offsets and commit indices equal an
oracle that is fed the request sequence read back from the new leader's log (ELECT 0, phase A, KILL 0, ELECT 1, phase B)."""
import os
import signal
import socket
import subprocess
import tempfile
import time

import numpy as np
import pytest

from apus_amd import trace as T
from tests.test_gpu_e2e_redis import REF, ROOT, _free_port, _wait_port, parse_dump

pytestmark = pytest.mark.gpu


def _dbsize(port, clean):
    r = subprocess.run([os.path.join(REF, "redis-cli"), "-p", str(port), "dbsize"], env=clean, capture_output=True, text=True, timeout=20)
    try:
        return int(r.stdout.strip().split()[-1])
    except (ValueError, IndexError):
        return -1


def _wait_equal(ports, clean, seconds=15.0):
    t0 = time.time()
    while time.time() - t0 < seconds:
        sizes = [_dbsize(p, clean) for p in ports]
        if len(set(sizes)) == 1 and sizes[0] > 0:
            return sizes
        time.sleep(0.2)
    return sizes


def _wait_log(path, needle, seconds):
    t0 = time.time()
    while time.time() - t0 < seconds:
        try:
            if needle in open(path, errors="replace").read():
                return True
        except OSError:
            pass
        time.sleep(0.1)
    return False


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "redis-server")), reason="oracle/_ref/redis-server not built (make -C oracle redis)")
def test_three_redis_processes_replicate_and_fail_over():
    from oracle import oracle as orc
    n, LOG = 3, 1 << 24
    tmp = tempfile.mkdtemp()
    gdir = os.path.join(tmp, "group")
    os.makedirs(gdir)
    hook = os.path.join(ROOT, "apus_amd", "libapus_interpose.so")
    ports = []
    while len(ports) < n:                     # (distinct: the kernel hands a port that was just released out again)
        p = _free_port()
        if p not in ports:
            ports.append(p)
    clean = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    procs, logs, dumps = [], [], []
    try:
        for i in range(n):
            d = os.path.join(tmp, f"r{i}")
            os.makedirs(d)
            cfg = os.path.join(d, "node.cfg")
            open(cfg, "w").write(f'db_name = "node_{i}";\nreq_log = 0;\nip_address = "127.0.0.1";\nport = {ports[i]};\n')
            logs.append(os.path.join(d, "dare.log"))
            dumps.append(os.path.join(d, "replicas.bin"))
            env = dict(os.environ, server_idx=str(i), group_size=str(n), APUS_GROUP_DIR=gdir, APUS_GPU_LOG_LEN=str(LOG), APUS_PRUNE_PERIOD_MS="100000000",
                       config_path=cfg, LD_PRELOAD=hook, dare_log_file=logs[i], APUS_PROXY_DUMP=dumps[i], APUS_REP_APPEND="8", APUS_REP_FWORK="4",
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([os.path.join(REF, "redis-server"), "--port", str(ports[i]), "--save", "", "--appendonly", "no"],
                                          cwd=d, env=env, stdout=open(os.path.join(d, "redis.out"), "w"), stderr=subprocess.STDOUT))
        for i in range(n):
            assert _wait_port(ports[i], procs[i], timeout=150), f"redis-server {i} did not come up\n" + open(os.path.join(tmp, f"r{i}", "redis.out")).read()[-3000:]
        assert _wait_log(logs[0], "[T2] LEADER", 60), "server 0 did not announce itself as the leader\n" + \
            "\n".join(f"--- server {i}:\n" + open(os.path.join(tmp, f"r{i}", "redis.out"), errors="replace").read()[-1500:] for i in range(n))

        def bench(port, n_req):
            b = subprocess.run([os.path.join(REF, "redis-benchmark"), "-p", str(port), "-t", "set", "-d", "16", "-r", "5000", "-n", str(n_req), "-c", "4", "-q"],
                               env=clean, capture_output=True, text=True, timeout=240)
            assert b.returncode == 0 and "requests per second" in b.stdout, b.stdout + b.stderr
            return b.stdout.strip().splitlines()[-1]

        # ---- phase A: clients at server 0; every follower's redis gets every SET through its own kernel + do_action
        line_a = bench(ports[0], 6000)
        sizes_a = _wait_equal(ports, clean)
        assert len(set(sizes_a)) == 1 and sizes_a[0] > 100, f"the followers' redis instances did not catch up: {sizes_a}"
        # ---- the leader's PROCESS dies
        procs[0].send_signal(signal.SIGKILL)
        procs[0].wait(timeout=30)
        assert _wait_log(logs[1], "[T4] LEADER", 60), "server 1 did not take over:\n" + open(logs[1], errors="replace").read()[-2000:] + \
            open(os.path.join(tmp, "r1", "redis.out")).read()[-2000:]
        # ---- phase B: clients at server 1
        line_b = bench(ports[1], 4000)
        sizes_b = _wait_equal(ports[1:], clean)
        assert len(set(sizes_b)) == 1 and sizes_b[0] >= sizes_a[0], f"server 2 does not follow the new leader: {sizes_b}"
        # ---- SHUTDOWN at the leader: a client request like any other -- replicated, committed by the two survivors, applied
        #      by the leader's redis (which exits and dumps every replica it has mapped) and replayed into the follower's
        #      (which exits too: a replicated state machine)
        subprocess.run([os.path.join(REF, "redis-cli"), "-p", str(ports[1]), "shutdown", "nosave"], env=clean, capture_output=True, text=True, timeout=60)
        procs[1].wait(timeout=120)
        try:
            procs[2].wait(timeout=20)
        except subprocess.TimeoutExpired:
            subprocess.run([os.path.join(REF, "redis-cli"), "-p", str(ports[2]), "shutdown", "nosave"], env=clean, capture_output=True, text=True, timeout=60)
            procs[2].wait(timeout=120)
    except BaseException:
        # post-mortem for the GPU box: what every server said, and where the threads of the ones still running sit
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "cluster_postmortem.txt"), "w") as f:
            for i, p in enumerate(procs):
                f.write(f"==== server {i} pid {p.pid} returncode {p.poll()}\n")
                for name in ("redis.out", "dare.log"):
                    try:
                        f.write(f"-- {name}\n" + open(os.path.join(tmp, f"r{i}", name), errors="replace").read()[-3000:] + "\n")
                    except OSError:
                        pass
                if p.poll() is None:
                    try:
                        for tid in sorted(os.listdir(f"/proc/{p.pid}/task")):
                            rd = lambda n: open(f"/proc/{p.pid}/task/{tid}/{n}", errors="replace").read().strip()
                            f.write(f"   thread {tid} {rd('comm')} wchan={rd('wchan')} syscall={rd('syscall')[:40]}\n")
                    except OSError as e:
                        f.write(f"   /proc: {e}\n")
            f.write("group dir: " + " ".join(sorted(os.listdir(gdir))) + "\n")
        raise
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert os.path.exists(dumps[1]), "no replica dump from the new leader\n" + open(os.path.join(tmp, "r1", "redis.out")).read()[-3000:]
    reps, rings = parse_dump(dumps[1], n)
    lead = reps[1]
    assert lead["status"] == 0, reps
    assert lead["commit"] == lead["end"] == lead["apply"], lead
    assert (reps[2]["commit"], reps[2]["end"]) == (lead["commit"], lead["end"]), (reps[1], reps[2])

    # the new leader's log, entry by entry: CONFIG of term 2, phase A (term 2), the two CONFIG entries of term 4, phase B
    ring = rings[1]
    off, entries = 0, []
    while off < lead["end"]:
        typ = int(ring[off + 26])
        term = int(ring[off + 8:off + 16].view(np.uint64)[0])
        ln = int(ring[off + 48:off + 50].view(np.uint16)[0]) if typ not in (T.CONFIG, T.HEAD, T.NOOP) else 0
        entries.append((typ, term, int(ring[off + 16:off + 24].view(np.uint64)[0]), int(ring[off + 24:off + 26].view(np.uint16)[0]),
                        ring[off + 50:off + 50 + ln].tobytes() if ln else b"", ln))
        off += 64 + ln
    assert off == lead["end"]
    assert {e[1] for e in entries} == {2, 4}
    client = [e for e in entries if e[0] not in (T.CONFIG, T.HEAD, T.NOOP)]
    a = [e for e in client if e[1] == 2]
    b = [e for e in client if e[1] == 4]
    assert sum(e[4].count(b"SET") for e in a) >= 6000 and sum(e[4].count(b"SET") for e in b) >= 4000
    assert all((e[3] >> 8) == 0 for e in a) and all((e[3] >> 8) == 1 for e in b)          # clt_id = leader's index << 8 | connection
    assert [e[0] for e in entries if e[1] == 4][:2] == [T.CONFIG, T.CONFIG]                 # blank CONFIG + removal of the dead server

    # the same history through the oracle
    reqs = np.zeros(len(client), dtype=orc.REQ_DTYPE)
    arena = bytearray(16)
    for g, (typ, term, rid, cid, body, ln) in enumerate(client):
        reqs[g] = (rid, len(arena), cid, ln, typ, (0, 0, 0))
        arena += body + bytes((-ln) % 16)
    arena = np.frombuffer(bytes(arena) + bytes(32), dtype=np.uint8)
    cl = orc.Cluster(n, LOG)
    cl.elect(0)
    for g0 in range(0, len(a), 64):
        cl.round(reqs[g0:min(g0 + 64, len(a))], arena)
    cl.quiesce()
    cl.kill(0)
    cl.elect(1)
    for g0 in range(len(a), len(reqs), 64):
        cl.round(reqs[g0:g0 + 64], arena)
    cl.quiesce()
    for r in (1, 2):
        o = cl.log(r).offsets()
        for k in ("head", "apply", "commit", "end"):
            assert reps[r][k] == o[k], f"replica {r}: {k} {reps[r][k]} vs oracle {o[k]}"
        ro = cl.log(r).ring()
        mask = orc.defined_mask(ro, o["end"], o["head"], o["end"])
        d = np.nonzero((rings[r] != ro) & mask)[0]
        assert len(d) == 0, f"replica {r}: {len(d)} defined ring bytes differ from the oracle, first at {d[:8].tolist()}"
    print(f"3 redis processes, one replica each: phase A {line_a}; leader killed; phase B at server 1 {line_b}; keys {sizes_a[0]} -> {sizes_b[0]}; "
          f"{len(client)} client entries, survivors' logs = oracle replay")
