"""ctypes loader of libapus_gpu.so.  Fails loudly: there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

u64, u32, u16, u8 = C.c_uint64, C.c_uint32, C.c_uint16, C.c_uint8
vp = C.c_void_p


class IpcReplica(C.Structure):
    """apus_ipc_replica_t (include/apus_gpu.h)"""
    _fields_ = [("handle", (u8 * 64) * 8), ("log_len", u64), ("dir_cap", u32), ("replica", u32),
                ("device", C.c_int32), ("fences", u32), ("pair", ((u8 * 64) * 2) * 4)]


class Cfg(C.Structure):
    _fields_ = [("group_size", u32), ("n_local", u32), ("local_ids", u8 * 13), ("pad", u8 * 3),
                ("log_len", u64), ("device", C.c_int32), ("flags", u32), ("stream", vp)]


_lib = None

SIGNATURES = {
    "apus_gpu_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(vp)]),
    "apus_gpu_destroy": (None, [vp]),
    "apus_gpu_reset": (C.c_int, [vp]),
    "apus_gpu_sync": (C.c_int, [vp]),
    "apus_gpu_stage": (C.c_int, [vp, vp, u64, vp, u64, vp, u64]),
    "apus_gpu_become_leader": (C.c_int, [vp, u32, u64, u32]),
    "apus_gpu_become_leader_ex": (C.c_int, [vp, u32, u64, u32, u32]),
    "apus_gpu_set_reachable": (C.c_int, [vp, u32]),
    "apus_gpu_append_control": (C.c_int, [vp, u8, vp]),
    "apus_gpu_run_rounds": (C.c_int, [vp, u64, u64]),
    "apus_gpu_tick_prune": (C.c_int, [vp]),
    "apus_gpu_quiesce": (C.c_int, [vp]),
    "apus_gpu_capture_begin": (C.c_int, [vp]),
    "apus_gpu_capture_end": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "apus_gpu_graph_launch": (C.c_int, [vp, C.c_int]),
    "apus_gpu_offsets": (C.c_int, [vp, u32, C.POINTER(u64)]),
    "apus_gpu_counters": (C.c_int, [vp, u32, C.POINTER(u64)]),
    "apus_gpu_hdr_words": (C.c_int, [vp, u32, C.POINTER(u64), u32]),
    "apus_gpu_read_ring": (C.c_int, [vp, u32, u64, u64, vp]),
    "apus_gpu_round_record": (C.c_int, [vp, u64, u64, vp, vp]),
    "apus_gpu_round_count": (u64, [vp]),
    "apus_gpu_apply_records": (C.c_int, [vp, u32, u64, u64, vp]),
    "apus_gpu_status": (u32, [vp]),
    "apus_gpu_clear_status": (None, [vp]),
    "apus_gpu_status_words": (C.c_int, [vp, C.POINTER(u32)]),
    "apus_gpu_device_ptr": (vp, [vp, u32, C.c_int, C.POINTER(u64)]),
    "apus_gpu_set_timing": (C.c_int, [vp, C.c_int]),
    "apus_gpu_kernel_time": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(u64)]),
    "apus_gpu_stream": (vp, [vp]),
    "apus_gpu_bind_global": (C.c_int, [vp]),
    "apus_gpu_global": (vp, []),
    "apus_gpu_device_arch": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "apus_gpu_follow": (C.c_int, [vp, u32, u32, u64, u32]),
    "apus_gpu_append_rounds": (C.c_int, [vp, u64, u64]),
    "apus_gpu_commit_rounds": (C.c_int, [vp, u64, u64]),
    "apus_gpu_ship_info": (C.c_int, [vp, C.POINTER(u64)]),
    "apus_gpu_ingest": (C.c_int, [vp, u32, u64, u64]),
    "apus_gpu_ack_merge": (C.c_int, [vp, u32, u64, u64]),
    "apus_gpu_follower_commit": (C.c_int, [vp, u32, u64, u64]),
    "apus_gpu_persist_start": (C.c_int, [vp, u32, u32]),
    "apus_gpu_persist_submit": (C.c_int, [vp, vp, u32, vp, u64]),
    "apus_gpu_persist_prune": (C.c_int, [vp]),
    "apus_gpu_persist_drain": (C.c_int, [vp, u32]),
    "apus_gpu_persist_highest_rec": (u64, [vp]),
    "apus_gpu_persist_stop": (C.c_int, [vp]),
    "apus_gpu_persist_latency_phase": (C.c_int, [vp, C.c_int, vp, u32, C.POINTER(u32)]),
    "apus_gpu_persist_roundtrip": (C.c_int, [vp, vp, u32, vp, u64, u32, vp]),
    "apus_gpu_persist_latency": (C.c_int, [vp, vp, u32, C.POINTER(u32)]),
    "apus_gpu_submit": (C.c_int, [vp, vp, u32, vp, u64]),
    "apus_gpu_append_live": (C.c_int, [vp, vp, u32, vp, u64]),
    "apus_gpu_commit_live": (C.c_int, [vp, C.c_int]),
    "apus_gpu_elect": (C.c_int, [vp, u32, u32, u32, C.POINTER(u64)]),
    "apus_gpu_export_replica": (C.c_int, [vp, u32, C.POINTER(IpcReplica)]),
    "apus_gpu_import_replica": (C.c_int, [vp, C.POINTER(IpcReplica)]),
    "apus_gpu_unmap_peers": (C.c_int, [vp]),
    "apus_gpu_store_stream": (C.c_int, [vp, u32, u64, u64, vp, u64, C.POINTER(u64), C.POINTER(u64)]),
    "apus_gpu_force_prune": (C.c_int, [vp, C.POINTER(u64)]),
    "apus_gpu_adopt_sid": (C.c_int, [vp, u32, u64]),
    "apus_gpu_clear_replica": (C.c_int, [vp, u32]),
    "apus_gpu_set_config": (C.c_int, [vp, u32, u64]),
    "apus_gpu_set_group_size": (C.c_int, [vp, u32]),
    "apus_gpu_join": (C.c_int, [vp, u32, u16, u32, u32, C.POINTER(u64)]),
    "apus_gpu_batch_begin": (C.c_int, [vp]),
    "apus_gpu_batch_end": (C.c_int, [vp]),
    "apus_gpu_calib_pingpong": (C.c_int, [vp, u32, u32, u32, u32, u64, vp, u32]),
    "apus_gpu_calib_store_bw": (C.c_int, [vp, u32, u64, u32, vp]),
    "apus_gpu_set_leader": (C.c_int, [vp, u32]),
    "apus_gpu_rep_start": (C.c_int, [vp, u32, u32, u32, u32]),
    "apus_gpu_rep_park": (C.c_int, [vp]),
    "apus_gpu_rep_reserve": (C.c_int, [vp, u32, C.POINTER(u64), C.POINTER(vp)]),
    "apus_gpu_rep_publish": (C.c_int, [vp, u64, vp, u64, u16, u8, u16]),
    "apus_gpu_rep_submit": (C.c_int, [vp, vp, u32, vp, u64]),
    "apus_gpu_rep_run": (C.c_int, [vp, u64, u64]),
    "apus_gpu_rep_prune": (C.c_int, [vp]),
    "apus_gpu_rep_cmds": (C.c_int, [vp, C.POINTER(u64), u32, u32]),
    "apus_gpu_rep_push_info": (C.c_int, [vp, C.POINTER(u32)]),
    "apus_gpu_rep_drain": (C.c_int, [vp, u32]),
    "apus_gpu_rep_full": (C.c_int, [vp]),
    "apus_gpu_rep_highest_rec": (u64, [vp]),
    "apus_gpu_rep_stats": (C.c_int, [vp, C.POINTER(u64)]),
    "apus_gpu_rep_latency": (C.c_int, [vp, vp, u32, C.POINTER(u32)]),
    "apus_gpu_rep_latency_appended": (C.c_int, [vp, vp, u32, C.POINTER(u32)]),
    "apus_gpu_rep_feed": (C.c_int, [vp, vp, u32, vp, u64, u32, C.c_double, u64, C.POINTER(u64)]),
    "apus_gpu_rep_follower_progress": (C.c_int, [vp, u32, C.POINTER(u64)]),
    "apus_gpu_rep_follower_stop": (C.c_int, [vp, u32]),
    "apus_gpu_rep_follower_replayed": (C.c_int, [vp, C.c_uint32, C.c_uint64]),
    "apus_gpu_rep_role_stats": (C.c_int, [vp, vp]),
    "apus_gpu_rep_launch_ms": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "apus_gpu_rep_req_ring_kind": (C.c_int, [vp]),
    "apus_gpu_rep_roundtrip": (C.c_int, [vp, vp, u32, vp, u64, u32, vp]),
    "apus_gpu_unmap_replica": (C.c_int, [vp, u32]),
    "apus_gpu_selftest": (C.c_int, [vp, u32, u32, u32, u64, u32, u32, C.POINTER(u64)]),
    "apus_gpu_ring_alloc_kind": (C.c_int, [vp]),
    "apus_gpu_calib_store_multi": (C.c_int, [vp, u32, u32, u32, C.POINTER(C.c_float)]),
    "apus_gpu_fence_replica": (C.c_int, [vp, u32, C.POINTER(IpcReplica)]),
    "apus_gpu_remap_fenced": (C.c_int, [vp, C.POINTER(IpcReplica)]),
    "apus_gpu_read_retired_ring": (C.c_int, [vp, u32, u32, u64, u64, vp]),
    "apus_gpu_last_entry": (C.c_int, [vp, u32, C.POINTER(u64)]),
    "apus_gpu_rep_box_words": (C.c_int, [vp, u32, u32, C.POINTER(u64)]),
    "apus_gpu_selftest_atomic_misses": (C.c_int, [vp, C.POINTER(u64)]),
    "apus_gpu_rep_feed_profile": (C.c_int, [vp, C.POINTER(u64)]),
    "apus_gpu_numa_node": (C.c_int, [C.c_int]),
    "apus_gpu_bind_near": (C.c_int, [vp, C.c_int]),
}


def lib_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load the HIP extension; raise if it cannot be built/loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("APUS_GPU_LIB") or _build.LIB     # APUS_GPU_LIB: e.g. the -DAPUS_TRACE diagnostics build
    if not os.path.exists(path):
        if not build_if_missing or path != _build.LIB:
            raise RuntimeError(f"{path} is missing: run python -m apus_amd.build (no CPU fallback exists)")
        _build.build()
    L = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the ABI symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
