"""BASELINE configs[0], the REFERENCE side (TEST INFRASTRUCTURE, like everything under oracle/): three redis-server
processes on this host, each under the reference's OWN interposer -- spec_hooks.cpp, proxy.c, db-interface.c, libdare,
compiled unmodified from /root/reference into oracle/_ref/interpose_ref_{O0,O2}.so (recipe: oracle/Makefile
`procref`) -- talking through the verbs stand-in in its one-server-per-process mode (shared-memory arena +
process_vm_writev: "CPU loopback (no RDMA/GPU)"), and redis-benchmark driving SETs at the leader
(/root/reference/benchmarks/run.sh:71-88,127).  What SURVEY.md section 8(d)(ii) asks for.

    python -m oracle.procref [--opt O0|O2] [-n 100000] [-c 1,50] [-d 64]

Prints one JSON object: requests/s per client count, the leader, the key counts of all three redis instances
afterwards (the followers replay through do_action_to_server), the unreplicated redis on the same box beside it."""
from __future__ import annotations

import argparse
import json
import os
import re
import shutil
import signal
import socket
import subprocess
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")

CFG = """#configuration of one replicated-state-machine node (the reference's target/nodes.local.cfg, one port per server:
#the reference assumes one server per host)
db_name = "node_test_{i}";
req_log = 0;
ip_address = "127.0.0.1";
port       = {port};
dare_global_config = {{
    hb_period = 0.01;
    elec_timeout_low = 100000;
    elec_timeout_high = 300000;
    retransmit_period = 0.04;
    rc_info_period = 0.05;
    log_pruning_period = 0.05;
}};
"""


def available(opt: str = "O0") -> bool:
    return all(os.path.exists(os.path.join(REF, f)) for f in (f"interpose_ref_{opt}.so", "redis-server", "redis-benchmark", "redis-cli"))


def _free_ports(n):
    socks, ports = [], []
    for _ in range(n):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        socks.append(s)
        ports.append(s.getsockname()[1])
    for s in socks:
        s.close()
    return ports


def _wait_port(port, procs, timeout=30.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if any(p.poll() is not None for p in procs):
            return False
        try:
            socket.create_connection(("127.0.0.1", port), timeout=0.5).close()
            return True
        except OSError:
            time.sleep(0.1)
    return False


def _bench(port, n_req, conns, dsize, timeout):
    clean = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    try:
        b = subprocess.run([os.path.join(REF, "redis-benchmark"), "-p", str(port), "-t", "set", "-d", str(dsize), "-n", str(n_req), "-c", str(conns), "-q"],
                           env=clean, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return None, f"redis-benchmark did not finish within {timeout} s (the server's hooked read never returned)"
    m = re.search(r"SET:\s*([0-9.]+) requests per second", b.stdout)
    return float(m.group(1)) if (b.returncode == 0 and m) else None, b.stdout[-300:] + b.stderr[-300:]


def _dbsize(port):
    clean = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    r = subprocess.run([os.path.join(REF, "redis-cli"), "-p", str(port), "dbsize"], env=clean, capture_output=True, text=True, timeout=20)
    try:
        return int(r.stdout.strip().split()[-1])
    except (ValueError, IndexError):
        return None


def unreplicated(n_req=100000, conns=(1, 50), dsize=64, timeout=120):
    """the same redis-server without any hook, on this box: the context number"""
    port = _free_ports(1)[0]
    tmp = tempfile.mkdtemp(prefix="apus_plain_")
    srv = subprocess.Popen([os.path.join(REF, "redis-server"), "--port", str(port), "--save", "", "--appendonly", "no"], cwd=tmp,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    try:
        if _wait_port(port, [srv]):
            for c in conns:
                out[str(c)] = _bench(port, n_req, c, dsize, timeout)[0]
    finally:
        srv.kill()
        srv.wait()
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def run(opt="O0", n=3, n_req=100000, conns=(1, 50), dsize=64, timeout=180, keep=False, n_req_by_conns=None, attempts=3):
    """n_req_by_conns: {clients: requests} overrides n_req per client count (a bounded sample for bench.py).  A start in
    which a follower did not get connected in time (the reference's leader then removes it for good: check_failure_count,
    dare_server.c:1189-1230, and runs on with the others) is repeated: the figure wanted is the one with every server
    replicating."""
    res = None
    for _ in range(attempts):
        res = _run_once(opt, n, n_req, conns, dsize, timeout, keep, n_req_by_conns)
        if res.get("ok") and res.get("replicated"):
            break
    res["attempts"] = _ + 1
    return res


def _run_once(opt, n, n_req, conns, dsize, timeout, keep, n_req_by_conns):
    assert available(opt), f"oracle/_ref/interpose_ref_{opt}.so or the redis binaries are missing (make -C oracle procref redis)"
    tmp = tempfile.mkdtemp(prefix="apus_procref_")
    shm = f"/apus_fab_{os.getpid()}_{int(time.time() * 1000) % 100000}"
    ports = _free_ports(n)
    procs, logs = [], []
    res = {"opt": opt, "servers": n, "requests": n_req, "payload": dsize, "ok": False}
    try:
        for i in range(n):
            d = os.path.join(tmp, f"r{i}")
            os.makedirs(os.path.join(d, ".db"))
            cfg = os.path.join(d, "node.cfg")
            open(cfg, "w").write(CFG.format(i=i, port=ports[i]))
            log = os.path.join(d, "dare.log")
            logs.append(log)
            env = dict(os.environ, LD_PRELOAD=os.path.join(REF, f"interpose_ref_{opt}.so"), server_idx=str(i), group_size=str(n),
                       server_type="start", config_path=cfg, dare_log_file=log, APUS_FAB_SHM=shm)
            env.pop("PYTHONPATH", None)
            procs.append(subprocess.Popen([os.path.join(REF, "redis-server"), "--port", str(ports[i]), "--save", "", "--appendonly", "no"],
                                          cwd=d, env=env, stdout=open(os.path.join(d, "redis.out"), "w"), stderr=subprocess.STDOUT))
            time.sleep(0.3)
        for i in range(n):
            if not _wait_port(ports[i], procs):
                res["error"] = f"redis-server {i} did not come up: " + open(os.path.join(tmp, f"r{i}", "redis.out")).read()[-800:]
                return res
        # the leader announces itself in its log (dare_server.c:1396; benchmarks/run.sh:44-69 greps for it)
        leader, t0 = None, time.time()
        while leader is None and time.time() - t0 < 30:
            for i, lg in enumerate(logs):
                try:
                    if "] LEADER" in open(lg, errors="replace").read():
                        leader = i
                except OSError:
                    pass
            if any(p.poll() is not None for p in procs):
                break
            time.sleep(0.2)
        if leader is None:
            res["error"] = "no leader was elected: " + "".join(open(lg, errors="replace").read()[-400:] for lg in logs if os.path.exists(lg))
            return res
        res["leader"] = leader
        # the followers connect to the new leader (RC_SYN / SYNACK): "New connection: #i" in its log, one per follower
        t0 = time.time()
        while time.time() - t0 < 8:
            txt = open(logs[leader], errors="replace").read()
            if len(set(re.findall(r"New connection: #(\d+)", txt))) >= n - 1:
                break
            time.sleep(0.2)
        time.sleep(1.0)
        res["requests_per_s"] = {}
        for c in conns:
            rps, tail = _bench(ports[leader], (n_req_by_conns or {}).get(c, n_req), c, dsize, timeout)
            res["requests_per_s"][str(c)] = rps
            if rps is None:
                res["error"] = f"redis-benchmark -c {c} failed: {tail}"
                return res
        time.sleep(0.5)
        res["dbsize"] = [_dbsize(p) for p in ports]
        res["replicated"] = bool(res["dbsize"][leader]) and all(s == res["dbsize"][leader] for s in res["dbsize"])
        res["ok"] = True
        return res
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGKILL)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        try:
            os.unlink("/dev/shm" + shm)
        except OSError:
            pass
        if keep:
            res["dir"] = tmp
        else:
            shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt", default="O0", choices=["O0", "O2"])
    ap.add_argument("-n", type=int, default=100000)
    ap.add_argument("-c", default="1,50")
    ap.add_argument("-d", type=int, default=64)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--plain", action="store_true", help="also time the same redis without any hook")
    a = ap.parse_args()
    conns = tuple(int(x) for x in a.c.split(","))
    out = run(a.opt, 3, a.n, conns, a.d, keep=a.keep)
    if a.plain:
        out["unreplicated_redis_requests_per_s"] = unreplicated(a.n, conns, a.d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
