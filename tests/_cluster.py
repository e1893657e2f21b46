"""Helpers of the one-process-per-server end-to-end tests (tests/test_gpu_e2e_cluster.py, test_gpu_e2e_failover.py,
test_gpu_e2e_reconf.py): redis-server 2.8.17 processes under LD_PRELOAD=libapus_interpose.so in group mode (APUS_GROUP_DIR),
a small RESP client that knows which of its requests were ANSWERED, the control files of the group directory, and the
oracle replay of a survivor's log under the schedule the elections really had."""
from __future__ import annotations

import os
import signal
import socket
import struct
import atexit
import shutil
import subprocess
import tempfile
import threading
import time

import numpy as np

from apus_amd import trace as T
from tests.test_gpu_e2e_redis import REF, ROOT, _free_port, _wait_port, parse_dump

CFG_FMT = "<QQIIIIQ"          # g_cfg_t: seq, term, leader, bitmask, size, kind, epoch (apus_amd/host/apus_proxy.c)
CFG_LEN = struct.calcsize(CFG_FMT)


class Group:
    """n redis-server processes, one replica each; `capacity` > n leaves room for machines that JOIN."""

    _old_tmp = []

    def __init__(self, n, log_len=1 << 24, capacity=None, rep_append=8, rep_fwork=4):
        self.n, self.log_len, self.capacity = n, log_len, capacity or n
        # (a group's directory holds its servers' replica dumps -- tens of MiB each -- and is read after close(): the one before
        #  this one has been looked at by now; the last one goes at exit.  A soak of thirty runs filled /tmp without this.)
        while Group._old_tmp:
            shutil.rmtree(Group._old_tmp.pop(), ignore_errors=True)
        self.tmp = tempfile.mkdtemp()
        Group._old_tmp.append(self.tmp)
        self.gdir = os.path.join(self.tmp, "group")
        os.makedirs(self.gdir)
        self.hook = os.path.join(ROOT, "apus_amd", "libapus_interpose.so")
        self.clean = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
        self.ports, self.procs, self.logs, self.dumps, self.dirs = {}, {}, {}, {}, {}
        self.grid = (rep_append, rep_fwork)
        self.nproc = 0

    def start(self, idx, join=False):
        """server `idx` (a joiner's place is worked out by the host layer itself: `idx` is only this test's name for it)"""
        d = os.path.join(self.tmp, f"p{self.nproc}")
        self.nproc += 1
        os.makedirs(d)
        port = _free_port()
        while port in self.ports.values():          # (the kernel hands a port that was just released out again: two servers of one group on one port)
            port = _free_port()
        cfg = os.path.join(d, "node.cfg")
        open(cfg, "w").write(f'db_name = "node_{idx}";\nreq_log = 0;\nip_address = "127.0.0.1";\nport = {port};\n')
        env = dict(os.environ, group_size=str(self.n), APUS_GROUP_DIR=self.gdir, APUS_GROUP_CAPACITY=str(self.capacity),
                   APUS_GPU_LOG_LEN=str(self.log_len), APUS_PRUNE_PERIOD_MS="100000000", config_path=cfg, LD_PRELOAD=self.hook,
                   dare_log_file=os.path.join(d, "dare.log"), APUS_PROXY_DUMP=os.path.join(d, "replicas.bin"),
                   APUS_REP_APPEND=str(self.grid[0]), APUS_REP_FWORK=str(self.grid[1]), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if join:
            env["server_type"] = "join"
            env.pop("server_idx", None)
        else:
            env["server_idx"] = str(idx)
        p = subprocess.Popen([os.path.join(REF, "redis-server"), "--port", str(port), "--save", "", "--appendonly", "no"],
                             cwd=d, env=env, stdout=open(os.path.join(d, "redis.out"), "w"), stderr=subprocess.STDOUT)
        self.ports[idx], self.procs[idx], self.logs[idx], self.dumps[idx], self.dirs[idx] = port, p, os.path.join(d, "dare.log"), os.path.join(d, "replicas.bin"), d
        return p

    def start_all(self):
        for i in range(self.n):
            self.start(i)
        for i in range(self.n):
            assert _wait_port(self.ports[i], self.procs[i], timeout=180), f"redis-server {i} did not come up\n" + self.tail(i)
        assert self.wait_log(0, "[T2] LEADER", 90), "server 0 did not announce itself as the leader\n" + self.all_tails()

    def tail(self, i, n=2500):
        out = ""
        for name in ("redis.out", "dare.log"):
            try:
                out += f"-- server {i} {name}\n" + open(os.path.join(self.dirs[i], name), errors="replace").read()[-n:] + "\n"
            except OSError:
                pass
        return out

    def all_tails(self):
        return "\n".join(self.tail(i) for i in sorted(self.dirs))

    def wait_log(self, i, needle, seconds):
        t0 = time.time()
        while time.time() - t0 < seconds:
            try:
                if needle in open(self.logs[i], errors="replace").read():
                    return True
            except OSError:
                pass
            time.sleep(0.05)
        return False

    def alive(self):
        return [i for i, p in self.procs.items() if p.poll() is None]

    def kill(self, i):
        self.procs[i].send_signal(signal.SIGKILL)
        self.procs[i].wait(timeout=30)

    def cfg_latest(self):
        """the newest announcement of the group directory -> dict, or None"""
        try:
            raw = open(os.path.join(self.gdir, "cfg_latest"), "rb").read()
        except OSError:
            return None
        if len(raw) < CFG_LEN:
            return None
        seq, term, leader, bitmask, size, kind, epoch = struct.unpack(CFG_FMT, raw[:CFG_LEN])
        return dict(seq=seq, term=term, leader=leader, bitmask=bitmask, size=size, kind=kind, epoch=epoch)

    def wait_cfg(self, pred, seconds=60.0):
        t0 = time.time()
        while time.time() - t0 < seconds:
            c = self.cfg_latest()
            if c is not None and pred(c):
                return c
            time.sleep(0.02)
        return None

    def parked(self, term):
        """the vote requests of the election of `term`: {server: (last_term, last_idx, n_end, end)}"""
        out = {}
        for i in range(self.capacity):
            try:
                raw = open(os.path.join(self.gdir, f"parked_{term}_{i}"), "rb").read()
            except OSError:
                continue
            if len(raw) >= 32:
                out[i] = struct.unpack("<QQQQ", raw[:32])
        return out

    def cli(self, i, *args, timeout=30):
        r = subprocess.run([os.path.join(REF, "redis-cli"), "-p", str(self.ports[i])] + list(args), env=self.clean, capture_output=True, text=True, timeout=timeout)
        return r.stdout.strip()

    def dbsize(self, i):
        try:
            return int(self.cli(i, "dbsize").split()[-1])
        except (ValueError, IndexError, subprocess.TimeoutExpired):
            return -1

    def shutdown(self, leader, wait_others=20):
        """SHUTDOWN at the leader: a client request like any other -- replicated, committed, applied by the leader's redis (which
        exits and dumps every replica it has mapped) and replayed into the followers' (which exit too)"""
        try:
            self.cli(leader, "shutdown", "nosave", timeout=60)
        except subprocess.TimeoutExpired:
            pass
        self.procs[leader].wait(timeout=120)
        for i in self.alive():
            try:
                self.procs[i].wait(timeout=wait_others)
            except subprocess.TimeoutExpired:
                try:
                    self.cli(i, "shutdown", "nosave", timeout=60)
                except subprocess.TimeoutExpired:
                    pass
                self.procs[i].wait(timeout=120)

    def postmortem(self, name="cluster_postmortem.txt"):
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            for i, p in sorted(self.procs.items()):
                f.write(f"==== server {i} pid {p.pid} returncode {p.poll()}\n" + self.tail(i, 4000))
                if p.poll() is None:
                    try:
                        for tid in sorted(os.listdir(f"/proc/{p.pid}/task")):
                            rd = lambda n: open(f"/proc/{p.pid}/task/{tid}/{n}", errors="replace").read().strip()
                            f.write(f"   thread {tid} {rd('comm')} wchan={rd('wchan')} syscall={rd('syscall')[:40]}\n")
                    except OSError as e:
                        f.write(f"   /proc: {e}\n")
            f.write("group dir: " + " ".join(sorted(os.listdir(self.gdir))) + "\n")

    def close(self):
        for p in self.procs.values():
            if p.poll() is None:
                p.kill()
        for p in self.procs.values():
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass


atexit.register(lambda: [shutil.rmtree(t, ignore_errors=True) for t in Group._old_tmp])


class Load:
    """`n_conn` connections that SET distinct keys one request at a time and remember which requests were ANSWERED
    (a reply the client has seen = an entry that was committed by a majority and applied, proxy.c:160)."""

    def __init__(self, port, n_conn, tag, value_bytes=16):
        self.port, self.n_conn, self.tag, self.vb = port, n_conn, tag, value_bytes
        self.acked = [[] for _ in range(n_conn)]          # per connection: (key, value) answered with +OK
        self.pending = [None] * n_conn                    # the request that was on the wire when the connection broke
        self.stop = threading.Event()
        self.threads = [threading.Thread(target=self._run, args=(c,), daemon=True) for c in range(n_conn)]

    def _run(self, c):
        try:
            sk = socket.create_connection(("127.0.0.1", self.port), timeout=10)
            sk.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            sk.settimeout(20)
        except OSError:
            return
        i = 0
        try:
            while not self.stop.is_set():
                key = f"{self.tag}:{c}:{i}"
                val = f"{i:0{self.vb}d}"
                self.pending[c] = (key, val)
                sk.sendall(f"SET {key} {val}\r\n".encode())
                buf = b""
                while not buf.endswith(b"\r\n"):
                    chunk = sk.recv(64)
                    if not chunk:
                        raise OSError("closed")
                    buf += chunk
                if not buf.startswith(b"+OK"):
                    raise OSError(f"unexpected reply {buf!r}")
                self.acked[c].append((key, val))
                self.pending[c] = None
                i += 1
        except OSError:
            pass
        finally:
            try:
                sk.close()
            except OSError:
                pass

    def start(self):
        for t in self.threads:
            t.start()
        return self

    def n_acked(self):
        return sum(len(a) for a in self.acked)

    def finish(self, timeout=30):
        self.stop.set()
        for t in self.threads:
            t.join(timeout)
        return [kv for a in self.acked for kv in a]


def mget(port, keys, batch=500):
    """{key: value or None} read from the redis at `port` (a follower's redis answers reads itself)"""
    out = {}
    sk = socket.create_connection(("127.0.0.1", port), timeout=20)
    sk.settimeout(30)
    f = sk.makefile("rb")
    try:
        for b0 in range(0, len(keys), batch):
            ks = keys[b0:b0 + batch]
            sk.sendall(("MGET " + " ".join(ks) + "\r\n").encode())
            head = f.readline()
            assert head.startswith(b"*"), head
            for k in ks:
                ln = f.readline()
                assert ln.startswith(b"$"), ln
                n = int(ln[1:])
                if n < 0:
                    out[k] = None
                else:
                    out[k] = f.read(n + 2)[:n].decode()
    finally:
        sk.close()
    return out


def wait_keys(port, kv, seconds=20.0):
    """every (key, value) of `kv` readable at `port` -> list of the ones that are not, after `seconds`"""
    t0 = time.time()
    missing = list(kv)
    while missing and time.time() - t0 < seconds:
        got = mget(port, [k for k, _ in missing])
        missing = [(k, v) for k, v in missing if got.get(k) != v]
        if missing:
            time.sleep(0.1)
    return missing


def log_entries(ring, end):
    """the entries of a log that starts at offset 0 and has not wrapped: (idx, term, type, req_id, clt_id, body)"""
    off, out = 0, []
    while off < end:
        typ = int(ring[off + 26])
        ln = int(ring[off + 48:off + 50].view(np.uint16)[0]) if typ not in (T.CONFIG, T.HEAD, T.NOOP) else 0
        out.append((int(ring[off:off + 8].view(np.uint64)[0]), int(ring[off + 8:off + 16].view(np.uint64)[0]), typ,
                    int(ring[off + 16:off + 24].view(np.uint64)[0]), int(ring[off + 24:off + 26].view(np.uint16)[0]),
                    ring[off + 50:off + 50 + ln].tobytes() if ln else b""))
        off += 64 + ln
    assert off == end, f"the log does not end on an entry boundary: {off} vs {end}"
    return out


def oracle_replay(n, log_len, entries, elections, start_leader=0):
    """The history of a group through the oracle, under the schedule the elections really had.

    entries: log_entries() of the final leader's log.  elections: [(term, winner, dead_leader, {server: n_end when it parked})]
    in order.  A server that held FEWER entries than the winner when the old leader died is cut off from the old leader
    (HOLD) exactly behind the last entry it held -- from there on the old leader's rounds did not reach it -- and released
    for the election; the winner's first pass catches it up (log adjustment + update_remote_logs).  Entries are fed in rounds
    of <= 64 (the bytes of a log do not depend on how its entries were grouped into polling() passes)."""
    from oracle import oracle as orc
    client = [e for e in entries if e[2] not in (T.CONFIG, T.HEAD, T.NOOP)]
    reqs = np.zeros(len(client), dtype=orc.REQ_DTYPE)
    arena = bytearray(16)
    for g, (idx, term, typ, rid, cid, body) in enumerate(client):
        reqs[g] = (rid, len(arena), cid, len(body), typ, (0, 0, 0))
        arena += body + bytes((-len(body)) % 16)
    arena = np.frombuffer(bytes(arena) + bytes(32), dtype=np.uint8)
    cl = orc.Cluster(n, log_len)
    cl.elect(start_leader)
    g = 0
    terms = sorted({e[1] for e in entries})
    el = {t: (w, dead, parked) for t, w, dead, parked in elections}
    for ti, t in enumerate(terms):
        seg = [(k, e) for k, e in enumerate(client) if e[1] == t]
        nxt = el.get(terms[ti + 1]) if ti + 1 < len(terms) else None
        cuts = {}
        if nxt is not None:
            w, dead, parked = nxt
            top = parked[w]
            for i, n_end in parked.items():
                if i != w and n_end < top:
                    cuts.setdefault(n_end, []).append(i)
        held = []
        buf = []

        def flush():
            nonlocal buf
            if buf:
                cl.round(reqs[buf[0]:buf[-1] + 1], arena)
                buf = []
        # (an entry's idx = the number of entry slots up to and including it: the numbering has not restarted in these logs)
        prev_idx = None
        for k, e in seg:
            for cut in sorted(cuts):
                if e[0] > cut and (prev_idx is None or prev_idx <= cut):
                    flush()
                    for i in cuts[cut]:
                        cl.hold(i)
                        held.append(i)
            buf.append(k)
            prev_idx = e[0]
            if len(buf) == 64:
                flush()
        flush()
        if nxt is not None:
            w, dead, parked = nxt
            for cut in sorted(cuts):            # (servers that held everything the segment's client entries brought, but not the winner's tail)
                for i in cuts[cut]:
                    if i not in held:
                        cl.hold(i)
                        held.append(i)
            cl.kill(dead)
            for i in held:
                cl.release(i)
            cl.elect(w)
    cl.quiesce()
    return cl


def compare_with_oracle(cl, reps, rings, servers):
    from oracle import oracle as orc
    for r in servers:
        o = cl.log(r).offsets()
        for k in ("head", "apply", "commit", "end"):
            assert reps[r][k] == o[k], f"replica {r}: {k} {reps[r][k]} vs oracle {o[k]}"
        ro = cl.log(r).ring()
        mask = orc.defined_mask(ro, o["end"], o["head"], o["end"])
        d = np.nonzero((rings[r] != ro) & mask)[0]
        assert len(d) == 0, (f"replica {r}: {len(d)} defined ring bytes differ from the oracle, first at {d[:8].tolist()} "
                             f"(entry offsets mod 64+len: gpu={rings[r][d[:8]].tolist()} orc={ro[d[:8]].tolist()})")
