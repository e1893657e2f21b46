"""BASELINE configs[0], the reference side: three redis-server processes under the reference's OWN interposer
(spec_hooks.cpp + proxy.c + db-interface.c + libdare, unmodified, oracle/_ref/interpose_ref_O0.so) on this host, the
verbs stand-in in its one-server-per-process mode, redis-benchmark at the leader (oracle/procref.py).  CPU only."""
import os

import pytest

from oracle import procref


@pytest.mark.skipif(not procref.available("O0"), reason="oracle/_ref/interpose_ref_O0.so or redis not built (make -C oracle procref redis)")
def test_reference_as_is_cluster_replicates_redis_sets():
    r = procref.run("O0", 3, 3000, (1, 8), 64, timeout=90)
    assert r["ok"], r
    assert r["leader"] in (0, 1, 2)
    assert all(v and v > 100 for v in r["requests_per_s"].values()), r
    # every follower replayed what the leader committed into its own redis (do_action_to_server, proxy.c:341-439)
    assert r["replicated"] and r["dbsize"][r["leader"]] >= 1, r


def test_fabric_builds_in_both_modes(tmp_path):
    """the verbs stand-in compiles as the in-process fabric of the lock-step oracle pin and as the one-server-per-process
    fabric (-DFAB_PROC)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "oracle", "refshim", "fabric.c")
    inc = os.path.join(root, "oracle", "refshim")
    for flags in ([], ["-DFAB_PROC"]):
        subprocess.check_call(["gcc", "-O1", "-fPIC", "-std=gnu11", "-Wall", "-Werror", "-c", "-I", inc, *flags, src, "-o", str(tmp_path / "f.o")])
