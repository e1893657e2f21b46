/* Per-member configuration views: which configuration does server i ITSELF hold?
 *
 * In the reference every server keeps its own dare_cid_t and moves it forward in poll_config_entries
 * (src/dare/dare_server.c:2133-2187): it walks the CONFIG entries of its log in order and takes each one whose
 * idx is above its cid_idx -- 0 for a server of the initial group, the idx of the CONFIG entry that admitted it for a
 * server that joined (the join reply carries it, dare_ibv_ud.c:1451-1490).  cid_idx never moves afterwards, and the
 * index sequence restarts at 1 after an exact-fit wrap: from then on a server that joined ignores every
 * CONFIG entry and keeps a stale configuration for good.  A voter also takes the candidate's configuration with its
 * vote (dare_server.c:1697).  It matters for the next JOIN: a member answers the joiner's RC_SYN only if its OWN
 * configuration has the joiner's bit ON and does not still show the slot's former holder (handle_rc_syn,
 * dare_ibv_ud.c), the joiner needs more than half of the group it joins, and retries for ever otherwise
 * (oracle/apus_oracle.c:orc_join, -6).
 *
 * The engine keeps ONE configuration (the leader's) on the device.  What the members hold is derived on the host when a
 * JOIN needs it, from a journal of every CONFIG entry a leader appended -- (slot, idx, bitmask), written on the device by
 * k_cfg_journal behind every launch that can append one, in stream order, so no launch path waits for it -- and of every
 * vote (k_cfg_note).  apus_gpu_join reads the journal once (it synchronises anyway) and refuses with APUS_E_NOANSWER
 * where the reference's joiner would retry for ever. */
#pragma once
#include "apus_kernels.h"

#include "apus_members_host.h"

/* the CONFIG entries among the last (at most 8) entries the leader appended that are not in the journal yet */
__global__ __launch_bounds__(64) void k_cfg_journal(const EngDev E, CfgJournal *J)
{
    if (threadIdx.x || E.leader >= E.group_size) return;
    const RepDev &Ld = E.rep[E.leader];
    const uint64_t n_end = Ld.hdr[H_N_END];
    uint64_t s = J->next_slot;
    if (s > n_end) s = 0;                                  /* a new leader whose log was cut: look again, duplicates are dropped below */
    if (n_end > 8 && s < n_end - 8) s = n_end - 8;
    for (; s < n_end; s++) {
        const uint64_t off = Ld.dir_off[(uint32_t)s & E.dir_mask];
        const uint4 u0 = ld16u(Ld.ring + off), u1 = ld16u(Ld.ring + off + 16), u3 = ld16u(Ld.ring + off + 48);
        if (((u1.z >> 16) & 0xFF) != APUS_CONFIG) continue;
        const uint64_t idx = (uint64_t)u0.x | ((uint64_t)u0.y << 32);
        bool dup = false;
        for (uint64_t k = J->n > 8 ? J->n - 8 : 0; k < J->n; k++) {
            const CfgItem &o = J->it[k % CFGJ_CAP];
            if (o.who == 0 && o.slot == s && o.idx == idx) dup = true;
        }
        if (dup) continue;
        CfgItem &it = J->it[J->n % CFGJ_CAP];
        it.slot = s; it.idx = idx; it.bitmask = u3.w; it.who = 0;
        J->n = J->n + 1;
    }
    J->next_slot = n_end;
}

/* the voters of an election take the winner's configuration with their vote (dare_server.c:1697) */
__global__ __launch_bounds__(64) void k_cfg_note(CfgJournal *J, uint32_t voters, uint32_t bitmask)
{
    if (threadIdx.x || !voters) return;
    CfgItem &it = J->it[J->n % CFGJ_CAP];
    it.slot = ~0ull; it.idx = ~0ull; it.bitmask = bitmask; it.who = voters;
    J->n = J->n + 1;
}
