"""Build libapus_gpu.so (hand-written HIP for gfx950 + the C host layer) in-tree."""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB = os.path.join(PKG, "libapus_gpu.so")
HOOK_LIB = os.path.join(PKG, "libapus_interpose.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = [os.path.join(CSRC, "apus_engine.hip")]
def _headers():
    """every header the library is made of: csrc/*.h and include/*.h"""
    inc = os.path.join(ROOT, "include")
    return sorted(os.path.join(d, f) for d in (CSRC, inc) if os.path.isdir(d) for f in os.listdir(d) if f.endswith(".h"))


DEPS = _headers()


def _host_c_sources():
    if not os.path.isdir(HOST):
        return []
    return sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".c") and not f.startswith("hook_"))


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    hook = os.path.join(HOST, "hook_interpose.c")
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in SOURCES + DEPS + _host_c_sources() + [hook])


TRACE_LIB = os.path.join(PKG, "libapus_gpu_trace.so")


def build_trace(verbose: bool = False) -> str:
    """Diagnostics build: the same engine with -DAPUS_TRACE (in-kernel wall-clock stamps,
    apus_gpu_trace()); a separate library, the product library is never instrumented."""
    objs = []
    for c in _host_c_sources():
        o = c[:-2] + ".trace.o"
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=gnu11", "-I", os.path.join(ROOT, "include"), "-c", c, "-o", o])
        objs.append(o)
    for src in SOURCES:
        o = src.rsplit(".", 1)[0] + ".trace.o"
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DAPUS_TRACE", "-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(o)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TRACE_LIB] + objs + ["-lpthread"])
    return TRACE_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    objs = []
    for c in _host_c_sources():
        o = c[:-2] + ".o"
        cmd = ["gcc", "-O2", "-g", "-fPIC", "-std=gnu11", "-Wall", "-I", os.path.join(ROOT, "include"), "-c", c, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(o)
    for src in SOURCES:
        o = src.rsplit(".", 1)[0] + ".o"
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    # the LD_PRELOAD interposer (plain C), linked against the engine library
    hook = os.path.join(HOST, "hook_interpose.c")
    if os.path.exists(hook):
        cmd = ["gcc", "-O2", "-g", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-I", os.path.join(ROOT, "include"),
               hook, "-o", HOOK_LIB, "-L", PKG, "-lapus_gpu", "-ldl", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--trace" in sys.argv:
        print(build_trace(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
