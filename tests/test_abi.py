"""The C-ABI library builds, loads without a GPU, and exports every symbol that
include/apus_gpu.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for hdr in ("apus_gpu.h", "apus_smr.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"typedef[^;]*;", "", txt)
        names += re.findall(r"\b((?:apus_gpu_|apus_tailq_|apus_snapshot_|apus_proxy_|dare_ib_|dare_server_|proxy_(?:init|on_)|is_leader|get_node_id)\w*)\s*\(", txt)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from apus_amd import build
    lib = build.build()
    L = ctypes.CDLL(lib)
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, f"declared in include/apus_gpu.h but not exported: {missing}"


def test_python_mirror_binds_every_engine_symbol():
    from apus_amd import _lib
    L = _lib.load()
    for name in _lib.SIGNATURES:
        assert getattr(L, name) is not None


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device engine creation must fail."""
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from apus_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(3, 1 << 16)


def test_struct_sizes_match_header():
    from apus_amd import trace as T
    from apus_amd.engine import APPLY_DTYPE
    assert T.REQ_DTYPE.itemsize == 24        # apus_req_t
    assert APPLY_DTYPE.itemsize == 32        # apus_apply_t


def test_headers_are_plain_c_and_cxx(tmp_path):
    """the boundary is a C ABI: both headers compile on their own as C99 and as C++17, warnings as errors"""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "apus_gpu.h"\n#include "apus_smr.h"\n'
                   'int main(void) { apus_cfg_t c; apus_req_t r; apus_apply_t a; (void)c; (void)r; (void)a; return 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, "-x", "c++", str(src)])


def test_reference_submission_queue_symbols(tmp_path):
    """message.h:20-22: `tailhead` and `tailq_lock` are exported with the reference's layout, so the reference's own
    proxy.c binds without a source change.  Where the reference tree is present: a C program that includes the
    reference's message.h (its tentative definitions become references to this library's objects), queues three
    requests exactly as proxy.c:147-158 does, and lets the library drain them (get_tailq_message)."""
    import subprocess
    from apus_amd import build
    lib = build.build()
    L = ctypes.CDLL(lib)
    for sym in ("tailhead", "tailq_lock"):
        assert ctypes.c_char.in_dll(L, sym) is not None
    ref = "/root/reference/src/include/dare/message.h"
    if not os.path.exists(ref):
        return
    src = tmp_path / "q.c"
    src.write_text(r"""
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "message.h"
int apus_tailq_drain(void);
int main(void)
{
    if (sizeof(tailq_entry_t) != 87416) return 2;
    TAILQ_INIT(&tailhead);                                                          /* proxy.c:486 */
    if (pthread_spin_init(&tailq_lock, PTHREAD_PROCESS_PRIVATE)) return 3;          /* proxy.c:494 */
    for (int i = 0; i < 3; i++) {
        tailq_entry_t *n2 = (tailq_entry_t *)malloc(sizeof(tailq_entry_t));      /* proxy.c:147-158 */
        n2->req_id = (uint64_t)i + 1; n2->connection_id = 7; n2->type = 5; n2->cmd.len = 4;
        memcpy(n2->cmd.cmd, "abcd", 4);
        pthread_spin_lock(&tailq_lock);
        TAILQ_INSERT_TAIL(&tailhead, n2, entries);
        pthread_spin_unlock(&tailq_lock);
    }
    const int n = apus_tailq_drain();
    printf("%d %d\n", n, TAILQ_EMPTY(&tailhead) ? 1 : 0);
    return (n == 3 && TAILQ_EMPTY(&tailhead)) ? 0 : 1;
}
""")
    exe = tmp_path / "q"
    pkg = os.path.dirname(lib)
    subprocess.check_call(["gcc", "-O1", "-fcommon", "-I", os.path.dirname(ref), str(src), "-o", str(exe), "-L", pkg, "-lapus_gpu",
                           "-lpthread", f"-Wl,-rpath,{pkg}"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.split() == ["3", "1"], (r.returncode, r.stdout, r.stderr)


def test_pmc_traffic_file_belongs_to_this_build():
    """bench.py quotes roofline.traffic only for the build the PMC passes were taken on: the committed
    profiles/r03_pmc_traffic.json must carry the hash of the kernel sources as they are now (apus_device.h + apus_kernels.h) --
    a later edit of those files without new PMC passes would silently null the figure in the driver's bench line."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    path = os.path.join(ROOT, "profiles", bench.PMC_FILE)
    assert os.path.exists(path), f"{path} is missing"
    doc = json.load(open(path))
    assert doc.get("kernel_source_sha256") == bench.kernel_source_hash(), \
        "apus_device.h / apus_kernels.h changed after the PMC passes: re-run tools/gpu_profile.sh and commit profiles/r03_pmc_traffic.json"
    assert doc["configs"]["c2"]["kernel"] == "k_step" and 400 < doc["configs"]["c2"]["bytes_per_entry"] < 1088


def test_replica_pmc_traffic_file_belongs_to_this_build():
    """the line's roofline.traffic / frac_moved (k_replica) are quoted only for the build the PMC passes were taken on:
    profiles/r05_replica_pmc_traffic.json must carry the hash of apus_device.h + apus_persistent.h + apus_replica.h as they are
    now, and one entry per configuration of the line (configs[1] at 1 / 3 / 5 / 7 replicas, configs[2], configs[3])"""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    path = os.path.join(ROOT, "profiles", bench.REP_PMC_FILE)
    assert os.path.exists(path), f"{path} is missing"
    doc = json.load(open(path))
    assert doc.get("kernel_source_sha256") == bench.replica_source_hash(), \
        "the replica kernels' sources changed after the PMC passes: re-run tools/gpu_profile_r5.sh and commit profiles/r05_replica_pmc_traffic.json"
    assert doc["kernel"] == "k_replica"
    for cfg, n in (("c2x1", 1), ("c2x3", 3), ("c2x5", 5), ("c2x7", 7), ("c3", 5), ("c4", 7)):
        c = doc["configs"][cfg]
        assert c["replicas"] == n and c["launches"] == 1 and c["launches_write_pass"] == 1
        # at least every replica's copy is written; nothing near SURVEY's (3N-1)E + 64 is moved
        assert n * c["mean_entry_bytes"] < c["written_bytes_per_entry"] * 1.02 and c["bytes_per_entry"] < (3 * n - 1) * c["mean_entry_bytes"] + 64
        assert bench.replica_pmc(cfg) == c
    r = bench.replica_roofline("c2x3", 3, 128, 10 ** 8, 25.0)
    assert r["lead"] == "frac_moved" and r["traffic"] == int(doc["configs"]["c2x3"]["bytes_per_entry"] * 10 ** 8)
    assert r["bytes_per_entry"] == 3 * 128 + 80 + 2 * 8 and abs(r["frac"] - 480 * 10 ** 8 / 25e-3 / 1e9 / 8000.0) < 1e-9
    # what is moved is never less than what has to move -- for every configuration
    for cfg, n in (("c2x1", 1), ("c2x3", 3), ("c2x5", 5), ("c2x7", 7), ("c3", 5), ("c4", 7)):
        c = doc["configs"][cfg]
        assert c["bytes_per_entry"] >= bench.algorithmic_bytes(n, c["mean_entry_bytes"]), cfg
