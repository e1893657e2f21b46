/* APUS_F_REF_QUIRKS: the reference's behaviour at a commit pointer parked on a wrap position, bit for bit.
 *
 * Situation ("Deviation 1"): everything up to offset X is committed and applied, the next
 * entry did not fit behind X and was appended at offset 0, and it has no majority yet.  The reference's leader
 * pass then
 *   - "commits" offset 0: the scan of update_remote_logs starts at X, log_get_entry redirects it to 0, the entry
 *     there has too few replies, the loop leaves with min_offset = 0 -- and log_is_offset_larger(0, X) holds
 *     (src/dare/dare_ibv_rc.c:1725-1758), so log->commit = 0: the same position, another number;
 *   - APPLIES the entry at 0 when the header did not fit behind X (a case-1 wrap): apply_committed_entries runs
 *     while commit is "larger" than apply (0 against X: true), log_get_entry redirects log->apply to 0 IN PLACE
 *     (src/include/dare/dare_log.h:327-330) and the entry found there is applied although nobody else has it
 *     (src/dare/dare_server.c:1815-1974): one upcall, highest_rec + 1, apply = end of that entry -- ahead of
 *     commit.  When the header fits behind X but the payload does not (case 2), the stale header there sends
 *     apply to 0 and the loop ends: nothing is applied.
 * With a quorum the same happens one pass early and nobody can tell; without one the client is released by an
 * entry only the leader holds.  The engine does not do that unless asked to: under APUS_F_REF_QUIRKS this
 * kernel runs behind every pass of the call-per-pass path (apus_gpu_run_rounds outside a batch, the live
 * calls, control rounds, quiesce) and brings the leader's control block to the reference's state.  The later
 * passes need nothing special: they commit by slot number, and apply from the leader's own apply slot.
 * A CONFIG entry in that position is left alone (the leader's apply of a CONFIG entry appends the next one,
 * dare_server.c:1859-1932: that belongs to the control rounds). */
#ifndef APUS_F_REF_QUIRKS
#define APUS_F_REF_QUIRKS 4u      /* apus_cfg_t.flags: the parity harness's diagnostic, not part of the public ABI (include/apus_gpu.h) */
#endif

#pragma once
#include "apus_kernels.h"

/* The per-pass record across calls.  A pass that starts with the commit pointer parked on the wrap position and
 * HAS its majority still ends with commit == 0 in the reference: the first scan ran before any ACK, "committed" offset
 * 0, and rc_write_remote_logs returned before the followers' end doorbell went out (dare_ibv_rc.c:1744-1758).  The pass
 * behind it finds the wrapped round acknowledged, commits exactly that and returns as early: it ends with commit ==
 * the END OF THE PASS BEFORE; the one after catches up (pinned on the reference: tests/traces.py:
 * wrap_quirk_second_round).  Inside one call finish_records (apus_kernels.h) writes the records that way; when the
 * two passes are two calls, this kernel does: q[0] = index of the record that is to read "end of the pass before"
 * (0 = none), left by the pass that recorded commit 0 with everything committed.  kind: 0 = a call of staged / live
 * rounds without a fused prune tick, 1 = anything else (control round, prune tick, quiesce): forgets q[0]. */
__device__ static inline void ref_quirk_records(const EngDev &E, const uint64_t *lh, uint64_t *q, int kind)
{
    const uint64_t n_rec = *E.rec_count, want = q[0];
    q[0] = 0;
    if (kind != 0 || n_rec == 0 || n_rec > E.rec_cap) return;
    if (want && want < n_rec) E.rec_commit[want] = E.rec_end[want - 1];
    const uint64_t last = n_rec - 1;
    if (last != want && E.rec_commit[last] == 0 && E.rec_end[last] != 0 && E.rec_end[last] != E.log_len &&
        lh[H_COMMIT] != 0 && lh[H_N_COMMIT] == lh[H_N_END])
        q[0] = n_rec;
}

__global__ __launch_bounds__(64) void k_ref_quirk_wrap(const EngDev E, uint64_t *q, int pass_kind)
{
    if (threadIdx.x) return;
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *lh = Ld.hdr;
    const uint64_t L = E.log_len;
    ref_quirk_records(E, lh, q, pass_kind);
    const uint64_t n_commit = lh[H_N_COMMIT], n_end = lh[H_N_END], n_apply = lh[H_N_APPLY];
    const uint64_t commit = lh[H_COMMIT], apply = lh[H_APPLY], end = lh[H_END];
    if (n_end <= n_commit || end == L) return;           /* nothing behind the commit point / the log reads as empty (Q13) */
    const uint32_t di = (uint32_t)n_commit & E.dir_mask;
    if (Ld.dir_off[di] != 0 || commit == 0) return;      /* the commit pointer is not parked on a wrap position */
    lh[H_COMMIT] = 0;
    if (n_apply != n_commit || apply != commit) return;
    if (L - commit >= APUS_HDR) { lh[H_APPLY] = 0; return; }       /* case 2: the stale header behind X sends apply to 0, no more */
    const uint32_t T = Ld.dir_len[di] & 0xFFFFFFu;
    const uint4 u0 = ld16u(Ld.ring), u1 = ld16u(Ld.ring + 16);
    const uint64_t idx = (uint64_t)u0.x | ((uint64_t)u0.y << 32);
    const uint32_t type = (u1.z >> 16) & 0xFF;
    const uint16_t clt = (uint16_t)(u1.z & 0xFFFF);
    if (type == APUS_CONFIG) return;
    const uint32_t client = (type != APUS_NOOP && type != APUS_HEAD);
    const uint32_t kind = client ? 1u : 0u;
    uint4 *rp = (uint4 *)&Ld.apply[(uint32_t)n_commit & E.dir_mask];
    rp[0] = make_uint4((uint32_t)n_commit, (uint32_t)(n_commit >> 32), 0u, 0u);
    rp[1] = make_uint4(u0.x, u0.y, T - APUS_HDR, (uint32_t)clt | (type << 16) | (kind << 24));
    if (client) {
        lh[H_APPLY_HASH] += apus_apply_mix(n_commit, 0, idx, T - APUS_HDR, clt, (uint8_t)type, (uint8_t)kind);
        lh[H_APPLY_COUNT] += 1;
        lh[H_HIGHEST_REC] += 1;
    }
    lh[H_APPLY] = T;
    lh[H_N_APPLY] = n_commit + 1;
}
