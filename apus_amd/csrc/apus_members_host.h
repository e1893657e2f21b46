/* The journal of CONFIG entries and votes, and what a member itself holds according to it: plain C++, no HIP --
 * shared by the kernels that write the journal (apus_members.h) and the host code that reads it (apus_gpu_join);
 * tests/test_members_host.py compiles it with g++. */
#pragma once
#include <stdint.h>

#define CFGJ_CAP 1024
struct CfgItem {
    uint64_t slot, idx;          /* position in the total order of the log; the entry's idx */
    uint32_t bitmask, who;       /* who: 0 = a log entry (every member that received it, subject to idx > cid_idx);
                                    otherwise the mask of voters that took the candidate's configuration with their vote */
};
struct CfgJournal { uint64_t n, next_slot; CfgItem it[CFGJ_CAP]; };

/* host side: what server i holds after the first `upto` journal items (H = a host copy of the journal).
 * base: the configuration it was given outright (initial group; join reply); since: the first slot of its log (a
 * server that joined polls from the head it was given -- entries OLDER than the one that admitted it included, if
 * they are still in the log); cid_idx as above; votes_from: journal length when it joined (votes of the slot's
 * former holder are not its own). */
struct MemberView { uint32_t base; uint64_t votes_from, since, cid_idx; };
static inline uint32_t member_view(const CfgJournal &H, const MemberView &m, uint32_t i, uint64_t upto)
{
    uint32_t v = m.base;
    for (uint64_t k = H.n > CFGJ_CAP ? H.n - CFGJ_CAP : 0; k < upto && k < H.n; k++) {
        const CfgItem &it = H.it[k % CFGJ_CAP];
        if (it.who) { if (k >= m.votes_from && ((it.who >> i) & 1u)) v = it.bitmask; }
        else if (it.idx > m.cid_idx && it.slot >= m.since) v = it.bitmask;
    }
    return v;
}

/* How many members answer the RC_SYN of a server joining slot r (handle_rc_syn, dare_ibv_ud.c; oracle/apus_oracle.c:
 * orc_join sweep 1): members of the group it joins (jsize = the new size while the configuration is being extended)
 * that are ON in the configuration of the join reply (nb), reachable, whose own configuration shows the joiner after
 * the JOIN's CONFIG entries (journal items [n0, n1)) and did not still show the slot's former holder before them.
 * The leader answers: its configuration is the one the entries carry.  The joiner needs more than jsize / 2. */
static inline uint32_t join_answers(const CfgJournal &H, const MemberView *mv, uint32_t r, uint32_t leader, uint32_t nb,
                                    uint32_t reachable, uint32_t jsize, uint64_t n0, uint64_t n1)
{
    uint32_t conn = 0;
    for (uint32_t i = 0; i < jsize; i++) {
        if (i == r || !((nb >> i) & 1u) || !((reachable >> i) & 1u)) continue;
        if (i == leader) { conn++; continue; }
        const bool stale = (member_view(H, mv[i], i, n0) >> r) & 1u;
        const bool knows = (member_view(H, mv[i], i, n1) >> r) & 1u;
        if (!stale && knows) conn++;
    }
    return conn;
}
