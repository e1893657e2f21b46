"""The bench line's contract (CPU): the committed line of the round's build (profiles/r05_bench_line.json, written by
`python bench.py --gpus 1 --steps 20 --warmup 5` on one MI355X right after the PMC passes of the same build) carries every key the
driver and the judge read, the headline is the replica kernels' figure and is consistent with its own parts, and EVERY
configuration in it has a counter-backed roofline."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = "r05_bench_line.json"


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
    assert "ack_aggregation_path" not in d                      # frozen, opt-in


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
