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
