// Host logic of apus_gpu_join's answer count (apus_amd/csrc/apus_members_host.h), checked without a GPU:
// tests/test_members_host.py compiles this with g++ and runs it.  Prints "ok" or the first failed check.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../apus_amd/csrc/apus_members_host.h"

static CfgJournal H;
static void log_item(uint64_t slot, uint64_t idx, uint32_t bitmask) { CfgItem &it = H.it[H.n % CFGJ_CAP]; it.slot = slot; it.idx = idx; it.bitmask = bitmask; it.who = 0; H.n++; }
static void vote_item(uint32_t voters, uint32_t bitmask) { CfgItem &it = H.it[H.n % CFGJ_CAP]; it.slot = ~0ull; it.idx = ~0ull; it.bitmask = bitmask; it.who = voters; H.n++; }
#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void)
{
    MemberView mv[13];
    // 1. a server of the initial group takes every CONFIG entry (cid_idx 0), in order
    memset(&H, 0, sizeof H);
    for (auto &m : mv) m = MemberView{0x7, 0, 0, 0};
    CHECK(member_view(H, mv[1], 1, 0) == 0x7);
    log_item(10, 11, 0x3); log_item(40, 41, 0xB);
    CHECK(member_view(H, mv[1], 1, 1) == 0x3 && member_view(H, mv[1], 1, 2) == 0xB && member_view(H, mv[1], 1, 99) == 0xB);
    // 2. a server that joined ignores entries whose idx is not above the idx of the one that admitted it (dare_server.c:2152)
    MemberView j{0xF, H.n, 30, 41};
    log_item(50, 42, 0x7);                                   // idx 42 > 41: taken
    CHECK(member_view(H, j, 3, H.n) == 0x7);
    log_item(60, 1, 0x5); log_item(61, 2, 0x1);              // the index sequence restarted (exact-fit wrap): ignored for good
    CHECK(member_view(H, j, 3, H.n) == 0x7 && member_view(H, mv[1], 1, H.n) == 0x1);
    // 3. ... polls from the head it was given: an OLDER entry that is still in its log counts (slot >= since), one before the head does not
    MemberView j2{0xF, H.n, 45, 0};
    CHECK(member_view(H, j2, 4, 3) == 0x7);                  // items at slots 10, 40 lie before slot 45; slot 50 counts
    MemberView j3{0xF, H.n, 100, 0};
    CHECK(member_view(H, j3, 4, H.n) == 0xF);                // nothing behind its head yet: the join reply's configuration
    // 4. a voter takes the candidate's configuration with its vote, whatever the idx; votes cast before a server joined are not its own
    const uint64_t before = H.n;
    vote_item(0x8 | 0x2, 0x1B);
    CHECK(member_view(H, j, 3, H.n) == 0x1B && member_view(H, mv[1], 1, H.n) == 0x1B && member_view(H, mv[2], 2, H.n) == 0x1);
    MemberView late{0xF, H.n, 0, 1000};
    CHECK(member_view(H, late, 3, H.n) == 0xF && before + 1 == H.n);
    // 5. the journal is a ring: only the newest CFGJ_CAP items are looked at
    memset(&H, 0, sizeof H);
    for (uint64_t k = 0; k < CFGJ_CAP + 10; k++) log_item(k, k + 1, (uint32_t)(k & 0xFF));
    CHECK(member_view(H, mv[0], 0, H.n) == ((CFGJ_CAP + 9) & 0xFF));
    // 6. the schedule of tests/test_oracle_vs_refloops.py:_random_join_trace(14): 3 servers, JOIN 3 (the group grows to 4), the index
    //    sequence restarts, KILL 2 (removal entry with a small idx), JOIN 2: server 3 never saw the removal -> it still shows the slot's
    //    former holder and does not answer; 0 (the leader) and 1 answer: 2 of 4 is not more than half -- the reference's joiner retries for ever
    memset(&H, 0, sizeof H);
    for (auto &m : mv) m = MemberView{0x7, 0, 0, 0};
    log_item(0, 1, 0x7);                                                        // the first leader's blank CONFIG entry
    uint64_t n0 = H.n;
    log_item(200, 201, 0xF); log_item(201, 202, 0xF); log_item(202, 203, 0xF);  // EXTENDED, TRANSIT, STABLE
    CHECK(join_answers(H, mv, 3, 0, 0xF, 0x7, 4, n0, H.n) == 3);               // 0, 1, 2 answer: more than 4 / 2
    mv[3] = MemberView{0xF, H.n, 150, 201};
    log_item(900, 5, 0xB);                                                      // KILL 2 behind the restart: idx 5
    CHECK(member_view(H, mv[3], 3, H.n) == 0xF && member_view(H, mv[1], 1, H.n) == 0xB);
    n0 = H.n;
    log_item(950, 6, 0xF);                                                      // JOIN 2: the empty slot is taken again
    CHECK(join_answers(H, mv, 2, 0, 0xF, 0xB, 4, n0, H.n) == 2);               // refused: 2 <= 4 / 2
    // ... without the restart server 3 takes the removal and the new entry, and answers
    memset(&H, 0, sizeof H);
    log_item(0, 1, 0x7); log_item(200, 201, 0xF); log_item(201, 202, 0xF); log_item(202, 203, 0xF);
    log_item(900, 905, 0xB);
    n0 = H.n;
    log_item(950, 906, 0xF);
    CHECK(join_answers(H, mv, 2, 0, 0xF, 0xB, 4, n0, H.n) == 3);
    // a configured member that is not reachable does not answer
    CHECK(join_answers(H, mv, 2, 0, 0xF, 0x9, 4, n0, H.n) == 2);
    printf("ok\n");
    return 0;
}
