"""apus_amd -- MI355X-native Paxos log replication / quorum aggregation engine
behind the APUS (hku-systems/apus) proxy and DARE SMR interfaces.

The compute path is hand-written HIP for gfx950 (apus_amd/csrc) behind the C ABI
declared in include/apus_gpu.h; this package is the thin host-side mirror of it."""

__all__ = ["trace"]
