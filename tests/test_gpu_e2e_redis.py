"""BASELINE configs[0] end to end: the application the reference benchmarks -- redis 2.8.17, built from
the tarball the reference vendors (oracle/Makefile `redis`, binaries under oracle/_ref/) -- runs
UNMODIFIED under LD_PRELOAD=libapus_interpose.so, redis-benchmark drives SET requests at it
(benchmarks/run.sh:71-88,127), and every socket read / accept / close of the server goes through
proxy_on_* -> the persistent consensus kernel before redis sees it (src/spec_hooks.cpp:102-178).

Checked: redis answered every request (redis-benchmark completes, DBSIZE > 0), the leader's log holds
exactly the hooked calls in order, all three replicas hold the same bytes, and an oracle fed with
the request sequence read back from the leader's log ends in the same state, bit for bit."""
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import pytest

from apus_amd import trace as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _wait_port(port, proc, timeout=90.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if proc.poll() is not None:
            return False
        try:
            socket.create_connection(("127.0.0.1", port), timeout=0.5).close()
            return True
        except OSError:
            time.sleep(0.2)
    return False


def parse_dump(path, n):
    raw = open(path, "rb").read()
    head, _, rest = raw.partition(b"rings\n")
    reps = []
    for line in head.decode().strip().splitlines():
        w = line.split()
        reps.append({w[i]: int(w[i + 1]) for i in range(0, len(w), 2)})
    assert len(reps) == n
    L = max(r["len"] for r in reps)        # (a place whose mapping was dropped -- a dead server -- reads as zeros)
    rings = [np.frombuffer(rest[i * L:(i + 1) * L], dtype=np.uint8) for i in range(n)]
    return reps, rings


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "redis-server")), reason="oracle/_ref/redis-server not built (make -C oracle redis)")
@pytest.mark.parametrize("n_req,n_conn,dsize", [(20000, 1, 3), (100000, 50, 64)])
def test_redis_under_ld_preload_replicates_every_request(n_req, n_conn, dsize):
    """redis-benchmark -t set -d <dsize> -n <n_req> -c <n_conn> (benchmarks/run.sh:71-88,127: one client, and fifty)"""
    from oracle import oracle as orc
    n, LOG = 3, 1 << 26
    port = _free_port()
    tmp = tempfile.mkdtemp()
    dump = os.path.join(tmp, "replicas.bin")
    hook = os.path.join(ROOT, "apus_amd", "libapus_interpose.so")
    assert os.path.exists(hook), "libapus_interpose.so is not built"
    env = dict(os.environ, server_idx="0", group_size=str(n), APUS_GPU_LOG_LEN=str(LOG), APUS_PRUNE_PERIOD_MS="100000000",
               APUS_PROXY_DUMP=dump, LD_PRELOAD=hook, dare_log_file=os.path.join(tmp, "dare.log"))
    srv = subprocess.Popen([os.path.join(REF, "redis-server"), "--port", str(port), "--save", "", "--appendonly", "no"],
                           cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        assert _wait_port(port, srv), f"redis-server did not come up\n{srv.stdout.read()[-3000:] if srv.poll() is not None else ''}"
        clean = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
        t0 = time.time()
        b = subprocess.run([os.path.join(REF, "redis-benchmark"), "-p", str(port), "-t", "set", "-d", str(dsize), "-n", str(n_req), "-c", str(n_conn), "-q"],
                           env=clean, capture_output=True, text=True, timeout=300)
        dt = time.time() - t0
        assert b.returncode == 0 and "requests per second" in b.stdout, b.stdout + b.stderr
        size = subprocess.run([os.path.join(REF, "redis-cli"), "-p", str(port), "dbsize"], env=clean, capture_output=True, text=True, timeout=30)
        assert size.returncode == 0 and int(size.stdout.strip().split()[-1]) >= 1, size.stdout
        subprocess.run([os.path.join(REF, "redis-cli"), "-p", str(port), "shutdown", "nosave"], env=clean, capture_output=True, text=True, timeout=60)
        srv.wait(timeout=120)
    finally:
        if srv.poll() is None:
            srv.kill()
    assert os.path.exists(dump), f"no replica dump: the server did not shut down through the interposer\n{srv.stdout.read()[-3000:]}"
    reps, rings = parse_dump(dump, n)
    assert all(r["status"] == 0 for r in reps), reps
    lead = reps[0]
    assert lead["commit"] == lead["end"] == lead["apply"] and lead["head"] == 0

    # the leader's log, entry by entry (no prune tick, no wrap: it starts at 0 with the blank CONFIG entry)
    ring = rings[0]
    off, entries = 0, []
    while off < lead["end"]:
        typ = int(ring[off + 26])
        ln = int(ring[off + 48:off + 50].view(np.uint16)[0]) if typ not in (T.CONFIG, T.HEAD, T.NOOP) else 0
        entries.append((typ, int(ring[off + 16:off + 24].view(np.uint64)[0]), int(ring[off + 24:off + 26].view(np.uint16)[0]),
                        ring[off + 50:off + 50 + ln].tobytes() if ln else b"", ln))
        off += 64 + ln
    assert off == lead["end"]
    assert entries[0][0] == T.CONFIG
    client = entries[1:]
    sets = sum(e[3].count(b"SET") for e in client if e[0] == T.SEND)
    connects = sum(1 for e in client if e[0] == T.CONNECT)
    assert sets >= n_req, f"{sets} SET commands in the log, {n_req} were sent"
    assert connects >= n_conn + 2            # the benchmark's connections + the two redis-cli calls
    assert lead["highest_rec"] == len(client)
    per_conn = {}
    for typ, rid, cid, body, ln in client:
        assert rid == per_conn.get(cid, 0) + 1, f"connection {cid:#x}: req_id {rid} after {per_conn.get(cid, 0)}"
        per_conn[cid] = rid

    # the same request sequence through the oracle
    reqs = np.zeros(len(client), dtype=orc.REQ_DTYPE)
    arena = bytearray(16)
    for g, (typ, rid, cid, body, ln) in enumerate(client):
        reqs[g] = (rid, len(arena), cid, ln, typ, (0, 0, 0))
        arena += body + bytes((-ln) % 16)
    arena = np.frombuffer(bytes(arena) + bytes(32), dtype=np.uint8)
    cl = orc.Cluster(n, LOG)
    cl.elect(0)
    for g0 in range(0, len(reqs), 64):
        cl.round(reqs[g0:g0 + 64], arena)
    cl.quiesce()
    for r in range(n):
        o = cl.log(r).offsets()
        for k in ("head", "apply", "commit", "end"):
            assert reps[r][k] == o[k], f"replica {r}: {k} {reps[r][k]} vs oracle {o[k]}"
        ro = cl.log(r).ring()
        mask = orc.defined_mask(ro, o["end"], o["head"], o["end"])
        d = np.nonzero((rings[r] != ro) & mask)[0]
        assert len(d) == 0, f"replica {r}: {len(d)} defined ring bytes differ from the oracle, first at {d[:8].tolist()}"
    print(f"redis under LD_PRELOAD: {n_req} SETs over {n_conn} connections in {dt:.2f} s "
          f"({n_req / dt:.0f} req/s host-observed), {len(client)} log entries; {b.stdout.strip().splitlines()[-1]}")
