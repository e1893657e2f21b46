#!/bin/bash
# One GPU-box call: the cross-process JOIN traces N times each (default 50) in all three data-plane modes, no retries
# (tests/test_gpu_peers.py::test_peer_mapped_group_join_soak), then the reproduction of round 3's mismatch.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp APUS_PEER_SOAK=${SOAK:-50}
timeout ${SOAK_TIMEOUT:-1500} python -m pytest tests/test_gpu_peers.py -m gpu -q --timeout 900 \
    -k "${SOAK_K:-join_soak or closing_barrier}" 2>&1 | grep -v "^W0\|Gloo\|amdgpu.ids" > gpurun_out/join_soak_full.log
echo "soak=$APUS_PEER_SOAK pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/join_soak_full.log
grep -E "rank [0-9]+: |rank\(s\) reported|Error\(|passed|failed|pytest exit" gpurun_out/join_soak_full.log | cut -c1-1200 | tail -40 > gpurun_out/join_soak.log
cat gpurun_out/join_soak.log
