/* First contact between two devices, checked before anything is measured (bench.py --gpus N; VERDICT r4, next #5a).
 *
 * The replica kernels rest on one assumption nothing on a one-GPU box can exercise: bytes a PEER's kernel stores into this
 * device's log ring (write-through, system scope: st16_wt) and the doorbell it stores behind them into this device's mailbox
 * are what this device's RESIDENT kernel reads with its system-scope loads (ld32_sys) -- never an older copy out of one of
 * this device's caches.  The rings are ordinary device memory (hipMalloc), the mailboxes uncached.  k_selftest replays
 * exactly that hand-off, with the data path's own instructions, and verifies every byte:
 *
 *   pusher (the leader's part, runs on the peer)   round r -> region r % regions of the owner's ring: 512 16-byte units of
 *       a pattern of (seed, r, unit), s_waitcnt vmcnt(0), then the four self-tagged granules of a doorbell into the owner's
 *       mailbox slot r % regions -- a round of 64 entries of 128 bytes, as rep_append_wave does it;
 *   checker (the follower's part, resident on the owner)   polls the doorbell, reads the region with ld32_sys, compares
 *       every unit, then frees the region: one word into the pusher's mailbox (the seqdone flow control).
 *
 * Regions are reused every `regions` rounds, so a stale line of an earlier lap is a mismatch.  Both roles can run in one
 * launch (roles = 3: the one-device unit test) or in two processes (roles = 1 on the pusher's, 2 on the owner's).
 * Round 6: the cumulative ACK a follower's work wavefront sends itself (REP_FAST_ACK) is a system-scope ATOMIC MAX into the
 * leader's mailbox through the mapping; the checker does the same -- round + 1 into the pusher's persisted_fast_by[15], a
 * drain, then the word that frees the region -- and the pusher, when it finds a region free, expects the atomic's word to
 * have got at least that far ([10]: times it had not: bench.py then runs the group with APUS_REP_DBG & 65536, the ACKs
 * through the retire wavefronts' plain stores alone).
 * Result words: [0] rounds checked, [1] units that differed, [2] first round that differed + 1, [3] waits that timed out.
 * A mismatch is not the end: bench.py re-creates the group with the rings in fine-grained memory (APUS_RING_ALLOC) and
 * tests again; the line says which allocation the numbers were taken on. */
#pragma once
#include "apus_replica.h"

#define ST_ROUND_BYTES 8192u
#define ST_UNITS (ST_ROUND_BYTES / 16u)

__device__ static inline uint4 st_pattern(uint64_t seed, uint64_t r, uint32_t u)
{
    const uint64_t x = (seed ^ (r + 1) * 0x9E3779B97F4A7C15ull) + (uint64_t)u * 0xD1B54A32D192ED03ull;
    return make_uint4((uint32_t)x, (uint32_t)(x >> 32), u ^ (uint32_t)r, ~(uint32_t)(x >> 17));
}

__global__ __launch_bounds__(256) void k_selftest(const EngDev E, uint32_t pusher, uint32_t owner, uint32_t roles_mode, uint64_t rounds,
                                                  uint32_t regions, uint64_t seed, uint64_t max_polls, unsigned long long *res)
{
    const uint32_t roles = roles_mode & 3u;
    const bool m_wbl2 = roles_mode & 16u, m_inv = roles_mode & 32u;      /* experiments: a release behind the stores / an invalidate in front of the loads */
    const uint32_t lane = lane_id();
    const uint32_t wgs = (roles == 3u) ? gridDim.x / 2 : gridDim.x;
    const bool push = (roles == 3u) ? blockIdx.x < wgs : roles == 1u;
    const uint32_t wg = (roles == 3u && !push) ? blockIdx.x - wgs : blockIdx.x;
    const uint32_t g = wg * 4 + (threadIdx.x >> 6), G = wgs * 4;
    uint8_t *ring = E.rep[owner].ring;
    RepBox *obox = E.box[owner], *pbox = E.box[pusher];
    if (push) {
        for (uint64_t r = g; r < rounds; r += G) {
            const uint32_t s = (uint32_t)(r % regions);
            if (r >= regions) {
                /* the region is free once the checker has said so: the round that used it last */
                uint64_t polls = 0;
                while (ld_sys(&pbox->rnd[s][7]) != r - regions + 1) {
                    if (++polls > max_polls) { if (lane == 0) atomicAdd(&res[3], 1ull); return; }
                    __builtin_amdgcn_s_sleep(2);
                }
                /* (the checker's atomic max went out, and was drained, in front of that word) */
                if (lane == 0 && ld_sys(&pbox->persisted_fast_by[15]) < r - regions + 1) atomicAdd(&res[10], 1ull);
            }
            uint8_t *dst = ring + (uint64_t)s * ST_ROUND_BYTES;
#pragma unroll
            for (uint32_t k = 0; k < ST_UNITS / WAVE; k++) {
                const uint32_t u = k * WAVE + lane;
                st16_wt(dst + 16u * u, st_pattern(seed, r, u));
            }
            if (m_wbl2) rep_release(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane < 4) st_sys(&obox->rnd[s][lane], rep_gran(r, lane));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    unsigned long long checked = 0, bad = 0, first_bad = 0;
    for (uint64_t r = g; r < rounds; r += G) {
        const uint32_t s = (uint32_t)(r % regions);
        uint64_t polls = 0;
        for (;;) {
            uint64_t gr = 0;
            if (lane < 4) gr = ld_sys(&obox->rnd[s][lane]);
            if (__ballot(lane < 4 && rep_gran_ok(gr, r)) == 0xFull) break;
            if (++polls > max_polls) { if (lane == 0) { atomicAdd(&res[3], 1ull); atomicAdd(&res[0], checked); atomicAdd(&res[1], bad); if (first_bad) atomicMin(&res[2], first_bad); } return; }
            __builtin_amdgcn_s_sleep(1);
        }
        const uint8_t *src = ring + (uint64_t)s * ST_ROUND_BYTES;
        if (m_inv) asm volatile("buffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        uint32_t wrong = 0;
        uint32_t bu = 0xFFFFFFFFu; uint4 bgot = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t k = 0; k < ST_UNITS / (2 * WAVE); k++) {
            const uint32_t u = 2 * (k * WAVE + lane);
            uint4 a, b;
            ld32_sys(src + 16u * u, a, b);
            const uint4 wa = st_pattern(seed, r, u), wb = st_pattern(seed, r, u + 1);
            const bool ba = (a.x != wa.x || a.y != wa.y || a.z != wa.z || a.w != wa.w), bb = (b.x != wb.x || b.y != wb.y || b.z != wb.z || b.w != wb.w);
            wrong += (ba ? 1u : 0u) + (bb ? 1u : 0u);
            if (ba && bu == 0xFFFFFFFFu) { bu = u; bgot = a; }
            if (bb && bu == 0xFFFFFFFFu) { bu = u + 1; bgot = b; }
        }
        const uint32_t nw = wsum32(wrong);
        checked++;
        if (nw) {
            bad += nw; if (!first_bad) first_bad = r + 1;
            /* the first unit that differed, once: which one, and what was there instead */
            const unsigned long long bl = __ballot(bu != 0xFFFFFFFFu);
            if (lane == (uint32_t)__builtin_ctzll(bl) && atomicCAS(&res[4], 0ull, r + 1) == 0ull) {
                res[5] = bu; res[6] = (unsigned long long)bgot.x | ((unsigned long long)bgot.y << 32); res[7] = (unsigned long long)bgot.z | ((unsigned long long)bgot.w << 32);
                const uint4 w = st_pattern(seed, r, bu);
                res[8] = (unsigned long long)w.x | ((unsigned long long)w.y << 32); res[9] = (unsigned long long)w.z | ((unsigned long long)w.w << 32);
            }
        }
        if (lane == 0) {
            __hip_atomic_fetch_max((APUS_GLOBAL uint64_t *)(uintptr_t)&pbox->persisted_fast_by[15], (uint64_t)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_sys(&pbox->rnd[s][7], r + 1);
        }
    }
    if (lane == 0) { atomicAdd(&res[0], checked); atomicAdd(&res[1], bad); if (first_bad) atomicMin(&res[2], first_bad); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


/* The data path's store pattern with nothing around it (round 5, DESIGN 9 "what bounds it"): every wavefront takes 8 KiB chunks
 * -- a round of 64 entries of 128 bytes -- and writes each chunk, write-through (st16_wt), at the same offset into EVERY ring of
 * `rings`, chunk after chunk through the whole ring, `passes` times.  With three or more 64 MiB rings the footprint is past the
 * 256 MiB of memory-side cache: what comes out is the rate at which this device takes write-through stores in this pattern --
 * the ceiling the replica kernels' ring stores sit under, whatever else they do. */
struct CalibRings { uint8_t *r[APUS_DEV_MAX_SERVERS]; uint32_t n; };
__global__ __launch_bounds__(256) void k_calib_store_multi(const CalibRings R, uint64_t bytes, uint32_t passes, uint32_t seed, uint32_t skew)
{
    const uint32_t lane = lane_id();
    const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), G = (uint64_t)gridDim.x * 4;
    const uint64_t chunks = bytes / ST_ROUND_BYTES - 1;
    for (uint32_t p = 0; p < passes; p++)
        for (uint64_t c = g; c < chunks; c += G) {
            const uint4 v = make_uint4(seed + p, (uint32_t)c, lane, 0x9E3779B9u);
            for (uint32_t i = 0; i < R.n; i++) {
                uint8_t *dst = R.r[i] + c * ST_ROUND_BYTES + 16u * lane + skew;          /* (skew: where a round really starts within a 128-byte line) */
#pragma unroll
                for (uint32_t k = 0; k < ST_UNITS / WAVE; k++) st16_wt(dst + 1024u * k, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          /* (a round's drain in front of its doorbell) */
        }
}
