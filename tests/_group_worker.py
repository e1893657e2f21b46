"""Worker of tests/test_gpu_group.py: one rank of a multi-process replica group.
All ranks share GPU 0 here (the GPU box has one device); the exchange goes through
gloo with host staging, the device halves are the product's own kernels."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def main():
    out_path, n_send, log_len = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from apus_amd import trace as T
    from apus_amd.distributed import GroupMember, run_trace_group
    from oracle import oracle as orc
    from tests.parity import compare_replica
    tr = T.steady_trace(world, n_send, (64, 107, 1024), 8, (1, 64), log_len=log_len, seed=21)
    m = GroupMember(world, rank, 0, 0, "gloo", log_len)
    res = {"rank": rank, "ok": False}
    try:
        run_trace_group(m, tr)
        m.eng.sync()
        cl = orc.run_trace(tr)
        compare_replica(m.eng, cl, rank, tag=f"group rank {rank}")
        res["ok"] = True
        res["end"] = m.eng.offsets(rank)["end"]
    except Exception as e:      # noqa: BLE001
        res["error"] = repr(e)
        try:
            if m.is_leader:
                m.leader_stop()
        except Exception:
            pass
    finally:
        with open(f"{out_path}.{rank}", "w") as f:
            json.dump(res, f)
        m.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
