"""The bench line's contract (CPU): the committed line of the round's build (profiles/r06_bench_line.json, written by
`python bench.py --gpus 1 --steps 20 --warmup 5` on one MI355X right after the PMC passes of the same build) carries every key the
driver and the judge read, the headline is the replica kernels' figure and is consistent with its own parts, and EVERY
configuration in it has a counter-backed roofline."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = "r06_bench_line.json"


def _line():
    return json.loads([l for l in open(os.path.join(ROOT, "profiles", LINE)) if l.startswith("{")][-1])


def _check_roofline(r, kernel="k_replica"):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "frac_moved", "moved_bytes_per_entry", "bytes_per_entry"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["kernel"] == kernel and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # algorithmic bytes x the launch's entries / the launch's duration
    assert abs(r["achieved"] - r["bytes_per_entry"] * r["entries_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved"]
    # counter-backed: traffic = what the PMC passes say this configuration moves per entry x the launch's entries
    assert r["traffic"] is not None and r["frac_moved"] is not None, "the line was taken on a build without PMC passes"
    assert abs(r["traffic"] - r["moved_bytes_per_entry"] * r["entries_per_launch"]) <= 1.0 + 1e-9 * r["traffic"]
    assert abs(r["frac_moved"] - r["traffic"] / (r["avg_launch_us"] * 1e-6) / 1e9 / r["peak"]) < 1e-6
    # what is moved is never less than what has to move; nothing is quoted above the measured copy ceiling
    assert r["moved_bytes_per_entry"] >= r["bytes_per_entry"] and r["frac"] <= r["frac_moved"] < r["copy_ceiling"] / r["peak"]


def test_committed_bench_line_has_the_contract_keys():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "committed entries/sec" and d["unit"] == "entries/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["steps"] == 20 and d["warmup"] == 5
    assert d["headline_kernel"] == "k_replica" and d["roofline"]["lead"] == "frac_moved"
    _check_roofline(d["roofline"])
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    # the headline is the replica kernels' device-resident figure, the fused launches ride along
    assert abs(d["value"] - d["replica_kernels"]["device_resident"]["value"]) < 1e-6 * d["value"]
    assert d["replica_kernels"]["device_resident"]["verified"] is True
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] - d["entries_per_step"]) < 1e-3 * d["entries_per_step"]
    assert "fused_step_path" in d and d["fused_step_path"]["roofline"]["kernel"] == "k_step"
    # no dead fields (round 4's busy_us = 0.0 / launches_per_step = 0)
    assert "launches_per_step" not in d["config"] and "busy_us" not in json.dumps(d["replica_kernels"]["device_resident"])
    assert "ack_aggregation_path" not in d                      # retired (round 6)
    # "verified" means bit-exact: the timed run itself was replayed by the oracle and compared, outside the timed regions
    assert d["replica_kernels"]["device_resident"]["bit_exact_vs_oracle"] is True


def test_every_configuration_of_the_line_has_a_counter_backed_roofline():
    d = _line()
    for g in ("1", "5", "7"):
        e = d["replica_kernels"]["by_group_size"][g]
        assert e["verified"] is True
        _check_roofline(e["roofline"])
    for cfg in ("c3", "c4"):
        e = d["other_configs"][cfg]
        assert e["replica_kernels"]["verified"] and e["fused_step_path"]["verified"]
        _check_roofline(e["replica_kernels"]["roofline"])
    assert d["other_configs"]["c5_failover_rejoin"]["replica_kernels"]["verified"]
    # host-fed: monotonic in producers
    hf = d["replica_kernels"]["host_fed"]["by_producer_threads"]
    assert all(v["verified"] for v in hf.values()) and hf["1"]["entries_per_s"] <= hf["2"]["entries_per_s"] <= hf["4"]["entries_per_s"]


def test_first_contact_verdicts():
    """bench.py --gpus N: how the self-test's per-follower results are read (no GPU): ok / mismatch (the fall-back is worth a try) /
    stuck (nothing to fall back to)"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    good = {"rounds": 1000, "bad_units": 0, "first_bad_round": 0, "timeouts": 0, "pusher_timeouts": 0, "rc": [0, 0]}
    assert bench._selftest_verdict({"1": dict(good), "2": dict(good)}, 1000) == "ok"
    assert bench._selftest_verdict({"1": dict(good), "2": dict(good, bad_units=3, first_bad_round=17)}, 1000) == "mismatch"
    assert bench._selftest_verdict({"1": dict(good, rounds=400, timeouts=1)}, 1000) == "stuck"
    assert bench._selftest_verdict({"1": dict(good, rc=[0, -6])}, 1000) == "stuck"
    # a side that timed out AND saw differences: the differences decide (fine-grained rings may cure both)
    assert bench._selftest_verdict({"1": dict(good, rounds=400, timeouts=1, bad_units=9)}, 1000) == "mismatch"


def test_algorithmic_bytes():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_bytes(3, 128) == 3 * 128 + 80 + 16 and bench.algorithmic_bytes(1, 128) == 208
    assert bench.algorithmic_bytes(3, 128, followers_look=False) == 464
    # always below SURVEY's (3N-1)E + 64 for N >= 2, and never below what the N rings alone take
    for n in (2, 3, 5, 7):
        for e in (104, 128, 1088, 4160):
            assert n * e < bench.algorithmic_bytes(n, e) < (3 * n - 1) * e + 64


def test_independent_groups_fallback_line():
    """`bench.py --gpus N`, last resort (the cross-GPU group raised on a fabric it has never met): the line of N independent
    single-GPU groups is the contract's -- the entries ALL reporting ranks committed over the SLOWEST rank's region -- and
    cannot be taken for the one-replica-per-GPU group's: marked in workload, mode, `fallback`, `verified_cross_gpu`."""
    import argparse
    import sys
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(replicas=3, payload=64, batch=64, warmup=5, gpus=4)
    parts = [{"rank": r, "device": r, "entries": 1000, "steps": 20, "seconds": 0.004 + 0.001 * r, "entries_per_s": 1000 * 20 / (0.004 + 0.001 * r),
              "verified": True, "bit_exact_vs_oracle": True if r == 0 else None} for r in (2, 0, 3, 1)]
    d = bench.independent_groups_line(args, 4, parts, "RuntimeError: hipIpcOpenMemHandle: invalid argument")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "committed entries/sec" and d["n_gpus"] == 4 and d["steps"] == 20 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert abs(d["value"] - 4 * 1000 * 20 / 0.007) < 1e-6 and abs(d["ms_per_step"] - 0.007 / 20 * 1e3) < 1e-12
    assert d["value"] <= sum(p["entries_per_s"] for p in parts)           # never more than the ranks' own figures add up to
    assert d["fallback"] is True and d["verified_cross_gpu"] is False and d["verified"] is True
    assert d["config"]["workload"].startswith("FALLBACK") and d["config"]["mode"].startswith("FALLBACK") and "hipIpcOpenMemHandle" in d["config"]["mode"]
    assert d["config"]["ranks_reporting"] == [0, 1, 2, 3] and [p["rank"] for p in d["by_rank"]] == [0, 1, 2, 3] and "model" not in d["config"]
    # a rank that did not report: the line exists, and is not verified
    d3 = bench.independent_groups_line(args, 4, parts[:3], "x")
    assert d3["verified"] is False and d3["config"]["groups"] == 3 and d3["n_gpus"] == 4
    json.dumps(d)


def test_group_watchdog_prints_the_line_once_the_headline_exists():
    """`bench.py --gpus N`: a rank still inside the group's measurement at --watchdog seconds.  With the headline measured (the
    phase is "extras"), rank 0's watchdog prints the line as it stands and the rank leaves with status 0 -- it never gives a
    measured group up for the fallback."""
    import subprocess
    import sys
    code = ("import sys, time, argparse; sys.path.insert(0, %r); import bench\n"
            "bench._WATCH['phase'], bench._WATCH['out'] = 'extras', {'metric': 'committed entries/sec', 'value': 1.0, 'n_gpus': 3}\n"
            "bench._arm_group_watchdog(argparse.Namespace(watchdog=-1))\n"
            "time.sleep(30)\nsys.exit(7)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout, p.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["n_gpus"] == 3 and "cut off by the watchdog" in d["extras"]


def test_independent_groups_fallback_two_ranks_meet_without_a_process_group():
    """The last resort's exchange, world size 2, on CPU: two ranks (children of one launcher, no process group between them) each
    "measure" their own device -- the measurement itself is replaced here, it needs an MI355X and has its GPU test -- leave their
    figures in the launcher-keyed directory, and rank 0 alone prints the one line: both ranks in it, the slower rank's time."""
    import subprocess
    import sys
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = ("import sys, json, argparse; sys.path.insert(0, %r); import bench, torch, os\n"
            "torch.cuda.set_device = lambda d: None\n"
            "torch.cuda.synchronize = lambda d=None: None\n"
            "rank = int(os.environ['RANK'])\n"
            "def fake(args, tr, n_rep, steps=None, device=0, **kw):\n"
            "    assert device == rank and n_rep == 3\n"
            "    return {'device_resident': {'value': len(tr.reqs) * steps / (0.01 * (rank + 1)), 'ms_per_step': 10.0 * (rank + 1) / steps, 'steps': steps,\n"
            "                                'verified': True, 'bit_exact_vs_oracle': True if kw.get('oracle_check') else None}}\n"
            "bench.measure_replica_kernels = fake\n"
            "args = argparse.Namespace(replicas=3, payload=64, batch=64, warmup=1, steps=4, gpus=2, entries=4096, watchdog=60, no_cpu=True, cpu_seconds=1.0, config='c2')\n"
            "out = bench.independent_groups_fallback(args, RuntimeError('the group fell apart'))\n"
            "if out is not None: print(json.dumps(out))\n" % ROOT)
    procs = []
    for r in (1, 0):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_PORT=str(port))
        env.pop("APUS_DIST_ONE_DEVICE", None)
        procs.append((r, subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    outs = {r: p.communicate(timeout=120) + (p.returncode,) for r, p in procs}
    assert outs[0][2] == 0 and outs[1][2] == 0, outs
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]            # rank 1 prints nothing
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0]
    d = json.loads(lines[0])
    assert d["fallback"] is True and d["n_gpus"] == 2 and d["config"]["ranks_reporting"] == [0, 1] and d["verified"] is True
    n = d["by_rank"][0]["entries"]
    assert abs(d["value"] - 2 * n * 4 / 0.02) < 1e-6 * d["value"] and "the group fell apart" in d["group_failure"]
    assert [p["device"] for p in d["by_rank"]] == [0, 1] and d["by_rank"][0]["bit_exact_vs_oracle"] is True and d["by_rank"][1]["bit_exact_vs_oracle"] is None
