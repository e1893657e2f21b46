/*
 * apus_persistent.h -- the persistent consensus kernel: the dare_server polling()
 * loop (/root/reference/src/dare/dare_server.c:1012-1125) as ONE resident HIP
 * kernel, one workgroup per replica, for the live (latency) path.
 *
 *   leader workgroup, per event from the host command ring:
 *     ROUND(n)  get_tailq_message + log_append_entry for n <= 64 queued requests
 *               (one lane per entry, wave scan for the offsets, both wrap rules),
 *               R1: the round's bytes go to its own ring and to every in-sync
 *               follower ring (write-through stores), R2: end doorbell,
 *               then the ACK scan of update_remote_logs (dare_ibv_rc.c:1725-1758):
 *               the window's ACK words and the ballot (SID) table staged in LDS,
 *               lane k checks popcount(ack_k | self) >= size/2+1, __ballot, commit
 *               prefix = count-trailing-ones; R4 commit doorbell; apply; the
 *               host-visible highest_rec releases the blocked submitters.
 *     PRUNE     log_pruning (dare_server.c:1996-2067)
 *   follower workgroup: wait for the end doorbell, persist_new_entries +
 *     rc_send_entries_reply (ACK byte + ACK bit), wait for the commit doorbell,
 *     apply_committed_entries (apply stream records).
 *
 * Hand-offs follow cdna_hip_programming.md G16: payload with agent-scope
 * (write-through) stores, s_waitcnt vmcnt(0) on every storing wave, barrier, ONE
 * lane publishes a monotonic 8-byte doorbell; the consumer polls that one word
 * relaxed, then ONE agent acquire, then plain loads.  Host <-> kernel words live in
 * coherent pinned memory and use system scope.  Every spin is bounded: on timeout
 * the kernel sets APUS_ST_SPIN_TIMEOUT and exits.
 */
#pragma once
#include "apus_kernels.h"

#define P_EV_CAP     1024u          /* events in the host command ring          */
#define P_REQ_CAP    (1u << 16)     /* request descriptors in the pinned ring    */
#define P_ARENA_CAP  (32u << 20)    /* pinned payload ring bytes                 */
#define P_LAT_CAP    (1u << 16)

enum { P_OP_ROUND = 1, P_OP_PRUNE = 2, P_OP_STOP = 3 };
#define P_BURST      16u            /* rounds whose commit pass may be shared when the host keeps the ring filled */

struct PEvent { uint32_t op; uint32_t n; uint64_t req_first;
                uint32_t arena_off;      /* the round's payloads are contiguous in the pinned arena: [off, off+bytes) */
                uint32_t arena_bytes; };
#define P_PREFETCH_BYTES (32u << 10)    /* rounds whose payloads fit are prefetched into LDS in one PCIe round trip */

/* host-coherent control block (hipHostMalloc, mapped) */
struct PersistHost {
    volatile uint64_t ev_tail;       /* host -> kernel: events published              */
    volatile uint64_t stop;          /* host -> kernel                                */
    volatile uint64_t ev_head;       /* kernel -> host: events consumed               */
    volatile uint64_t highest_rec;   /* kernel -> host: proxy->highest_rec (proxy.c:263) */
    volatile uint64_t commit_slot;   /* kernel -> host                                */
    volatile uint64_t alive;         /* kernel -> host: 1 running, 2 exited           */
    volatile uint64_t exit_code;     /* 0 stop, 1 idle limit, 2 spin timeout          */
    volatile uint64_t rounds_done;
    volatile uint64_t full;          /* kernel -> host: a round was refused, the log is full (dare_log.h:168,492-495) */
    PEvent   ev[P_EV_CAP];
    ReqDev   req[P_REQ_CAP];
    uint16_t req_len[P_REQ_CAP];
    uint8_t  arena[P_ARENA_CAP + 64];
};

/* device-side doorbells (agent scope) + latency samples */
struct PersistDev {
    uint64_t end_bell[APUS_DEV_MAX_SERVERS];     /* visible entry slots, per follower  */
    uint64_t commit_bell[APUS_DEV_MAX_SERVERS];  /* committed entry slots              */
    uint64_t applied_bell[APUS_DEV_MAX_SERVERS]; /* follower -> leader: applied slots  */
    uint64_t quit;                               /* leader -> followers                */
    uint32_t lat_n;
    uint32_t pad;
    uint32_t lat_ticks[P_LAT_CAP];               /* append -> commit, wall_clock64 ticks */
    uint32_t lat_seq[P_LAT_CAP];                 /* event seen -> offsets/headers computed   */
    uint32_t lat_push[P_LAT_CAP];                /* ... -> bytes pushed + end doorbell rung   */
};

#define RLX_AGENT  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define RLX_SYSTEM __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

/* Every word these touch lies in device or pinned host memory: the pointers are cast to the global address space, so
 * that the compiler emits global_load / global_store.  Through a generic pointer they are FLAT instructions, which count in
 * vmcnt AND lgkmcnt: every wait for an LDS word then also waits for the write-through stores in flight (2-4 us each). */
#define APUS_G64(p) ((APUS_GLOBAL uint64_t *)(uintptr_t)(p))
__device__ static inline uint64_t ld_agent(const uint64_t *p) { return __hip_atomic_load(APUS_G64(p), RLX_AGENT); }
__device__ static inline void st_agent(uint64_t *p, uint64_t v) { __hip_atomic_store(APUS_G64(p), v, RLX_AGENT); }
__device__ static inline uint64_t ld_sys(const volatile uint64_t *p) { return __hip_atomic_load(APUS_G64(p), RLX_SYSTEM); }
__device__ static inline void st_sys(volatile uint64_t *p, uint64_t v) { __hip_atomic_store(APUS_G64(p), v, RLX_SYSTEM); }
/* words in LDS that wavefronts of one workgroup hand each other: volatile, and typed as LDS -- a volatile access through a
 * generic pointer is a FLAT instruction followed by s_waitcnt vmcnt(0) (the address space of a volatile access is never
 * inferred): every look at such a word waited for every store the wavefront had in flight. */
#define APUS_LDS __attribute__((address_space(3)))
typedef volatile uint64_t APUS_LDS *lds_u64;
#define APUS_LDS64(p) ((lds_u64)(p))

/* 16 bytes to another workgroup's view of memory: ONE write-through store
 * (global_store_dwordx4 sc0 sc1 = the R1 store of cdna_hip_programming.md G16); byte-wise
 * agent-scope stores when the address is not dword aligned (unaligned entries) */
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
__device__ static inline void st16_agent(uint8_t *p, uint4 v)
{
    if ((((uintptr_t)p) & 3) == 0) {
        v4u_t d = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (int i = 0; i < 16; i++)
            __hip_atomic_store((APUS_GLOBAL uint8_t *)(uintptr_t)(p + i), (uint8_t)(w[i >> 2] >> (8 * (i & 3))), RLX_AGENT);
    }
}

/* 16 bytes at an arbitrary byte offset of an LDS buffer (dword reads + funnel shift) */
__device__ static inline uint4 lds_ld16u(const uint32_t *buf, uint32_t byte_off)
{
    const uint32_t i = byte_off >> 2, sh = (byte_off & 3) * 8;
    const uint32_t d0 = buf[i], d1 = buf[i + 1], d2 = buf[i + 2], d3 = buf[i + 3], d4 = buf[i + 4];
    if (sh == 0) return make_uint4(d0, d1, d2, d3);
    return make_uint4((d0 >> sh) | (d1 << (32 - sh)), (d1 >> sh) | (d2 << (32 - sh)),
                      (d2 >> sh) | (d3 << (32 - sh)), (d3 >> sh) | (d4 << (32 - sh)));
}

/* payload_unit() on a payload that sits in LDS at byte offset `src` (bytes src-2 .. readable) */
__device__ static inline uint4 payload_unit_lds(const uint32_t *buf, uint32_t src, uint32_t so, uint32_t P, uint32_t len16)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (P == 0 && so >= 50) return v;
    uint64_t lo = 0, hi = 0;
    if (P != 0) {
        v = lds_ld16u(buf, src + so - 50);
        lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
        hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
        const int vb = so < 50 ? (int)(50 - so) : 0;
        const int ve = (int)min(16u, 50u + P - so);
        lo &= byte_mask64(vb, ve);
        hi &= byte_mask64(vb - 8, ve - 8);
    }
    if (so == 48)      lo |= (uint64_t)(len16 & 0xFFFFu);
    else if (so == 49) lo |= (uint64_t)((len16 >> 8) & 0xFFu);
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

/* bounded wait until *p (agent scope) reaches `want`; lane 0 of wave 0 polls */
__device__ static inline bool wait_ge_agent(const uint64_t *p, uint64_t want, uint64_t max_polls,
                                            const volatile uint64_t *stop, uint64_t *seen)
{
    for (uint64_t i = 0; i < max_polls; i++) {
        const uint64_t v = ld_agent(p);
        if (v >= want) { *seen = v; return true; }
        if ((i & 255) == 255 && ld_sys(stop)) { *seen = v; return false; }
        __builtin_amdgcn_s_sleep(2);
    }
    *seen = ld_agent(p);
    return false;
}

/* ------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_consensus_persistent(const EngDev E, PersistHost *H, PersistDev *D,
                                                              uint32_t local_mask, uint32_t push_mask,
                                                              uint64_t idle_polls, uint64_t peer_polls)
{
    __shared__ AppendLds lds;
    __shared__ SeqOut s_seq;
    __shared__ uint64_t s_word[8];
    __shared__ uint32_t s_ack[WAVE];                  /* ACK bitmaps of the window               */
    __shared__ uint64_t s_sid[APUS_DEV_MAX_SERVERS];  /* ballot numbers (SIDs) of the group      */
    __shared__ unsigned long long s_acc[2];
    __shared__ uint32_t s_pay[(P_PREFETCH_BYTES + 64) / 4];   /* the round's payload bytes (leader) */
    const uint32_t tid = threadIdx.x, lane = lane_id();
    const uint64_t L = E.log_len;

    /* which replica this workgroup is */
    int me = -1;
    for (int i = 0, k = 0; i < APUS_DEV_MAX_SERVERS; i++)
        if (local_mask & (1u << i)) { if (k == (int)blockIdx.x) { me = i; break; } k++; }
    if (me < 0) return;
    const RepDev &Md = E.rep[me];
    uint64_t *mh = Md.hdr;

    if ((uint32_t)me != E.leader) {
        /* ============================ follower ============================ */
        if (!((push_mask >> me) & 1u)) return;            /* not reachable: nothing will arrive */
        uint64_t n_persist = mh[H_N_PERSIST], n_apply = mh[H_N_APPLY];
        for (;;) {
            /* wait for news: more visible entries, a newer commit, or quit */
            if (tid == 0) {
                uint64_t code = 0, vis = n_persist, com = n_apply;
                for (uint64_t i = 0;; i++) {
                    vis = ld_agent(&D->end_bell[me]);
                    com = ld_agent(&D->commit_bell[me]);
                    if (vis > n_persist || com > n_apply) break;
                    if (ld_agent(&D->quit)) { code = 1; break; }
                    if (i >= 16 * (idle_polls + peer_polls)) { code = 2; break; }   /* safety net only: the leader's quit bell ends us */
                    __builtin_amdgcn_s_sleep(1);
                }
                s_word[0] = code; s_word[1] = vis; s_word[2] = com;
            }
            __syncthreads();
            const uint64_t code = s_word[0], vis = s_word[1];
            uint64_t com = s_word[2];
            if (code) break;
            if (vis > n_persist) {
                /* persist_new_entries + rc_send_entries_reply for slots [n_persist, vis) */
                /* the directory words were published with write-through stores: L1-bypassing
                 * loads read them without an acquire fence (G16: sc1 loads pair with sc1 stores) */
                for (uint64_t s = n_persist + tid; s < vis; s += blockDim.x) {
                    const uint32_t di = (uint32_t)s & E.dir_mask;
                    const uint64_t off = ld_agent(&Md.dir_off[di]);
                    const uint32_t sender = __hip_atomic_load(&Md.dir_len[di], RLX_AGENT) >> 24;
                    __hip_atomic_store(Md.ring + off + 28 + me, (uint8_t)1, RLX_AGENT);
                    if (sender < APUS_DEV_MAX_SERVERS && E.rep[sender].ring) {
                        __hip_atomic_store(E.rep[sender].ring + off + 28 + me, (uint8_t)1, RLX_AGENT);   /* R3 */
                        atomicOr(&E.rep[sender].ack[di], 1u << me);
                    }
                }
                if (tid == 0) {
                    /* R2: the new end = the byte after the last visible entry (own directory) */
                    const uint32_t dl = (uint32_t)(vis - 1) & E.dir_mask;
                    const uint64_t lend = ld_agent(&Md.dir_off[dl]) + (__hip_atomic_load(&Md.dir_len[dl], RLX_AGENT) & 0xFFFFFFu);
                    mh[H_STORE_COUNT] += vis - n_persist;
                    mh[H_END] = lend; mh[H_OLD_END] = lend; mh[H_N_END] = vis; mh[H_N_PERSIST] = vis;
                }
                n_persist = vis;
            }
            if (com > n_persist) com = n_persist;
            if (com > n_apply) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       /* ring headers were written by the leader WG */
                apply_range(E, me, n_apply, com, 0, blockDim.x, s_acc);
                if (tid == 0) {
                    const uint64_t coff = (com == mh[H_N_END]) ? mh[H_END] : Md.dir_off[(uint32_t)com & E.dir_mask];
                    mh[H_COMMIT] = coff; mh[H_N_COMMIT] = com; mh[H_N_APPLY] = com;
                    st_agent(&mh[H_APPLY], coff);
                    const uint64_t hs = ld_agent(&mh[H_HEAD_SLOT]);
                    if (hs) {
                        const uint64_t hv = ld8u(Md.ring + Md.dir_off[(uint32_t)(hs - 1) & E.dir_mask] + 48);
                        if (apus_is_larger(mh[H_END], L, hv, mh[H_HEAD])) mh[H_HEAD] = hv;
                        mh[H_HEAD_SLOT] = 0;
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_agent(&D->applied_bell[me], com);
                }
                n_apply = com;
            }
            __syncthreads();
        }
        return;
    }

    /* ================================ leader ================================ */
    uint64_t ev_head = 0;
    uint64_t exit_code = 0;
    uint32_t burst = 0;                               /* rounds appended since the last commit pass */
    if (tid < APUS_DEV_MAX_SERVERS) s_sid[tid] = (tid < E.group_size && E.rep[tid].ring) ? E.rep[tid].hdr[H_SID] : 0;
    if (tid == 0) st_sys(&H->alive, 1);
    __syncthreads();
    const uint64_t my_term = s_sid[me] >> 9;

    for (;;) {
        /* ---- wait for the next event from the host ---- */
        if (tid == 0) {
            uint64_t code = 0;
            for (uint64_t i = 0;; i++) {
                if (ld_sys(&H->ev_tail) > ev_head) break;
                if (ld_sys(&H->stop)) { code = 1; break; }
                if (i >= idle_polls) { code = 2; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            s_word[0] = code;
            if (!code) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");       /* system scope: the event and its requests */
                const PEvent ev = H->ev[ev_head % P_EV_CAP];
                s_word[1] = ev.op; s_word[2] = ev.n; s_word[3] = ev.req_first;
                s_word[5] = ev.arena_off; s_word[6] = ev.arena_bytes;
            }
        }
        __syncthreads();
        if (s_word[0]) { exit_code = (s_word[0] == 1) ? 0 : 1; break; }
        const uint32_t op = (uint32_t)s_word[1];
        const uint32_t nr = (uint32_t)s_word[2];
        const uint64_t req_first = s_word[3];
        const uint32_t ar_off = (uint32_t)s_word[5], ar_bytes = (uint32_t)s_word[6];
        __syncthreads();
        if (op == P_OP_STOP) { exit_code = 0; ev_head++; break; }

        const uint64_t t_start = wall_clock64();
        uint64_t n_end0 = mh[H_N_END];
        bool appended = false;

        if (op == P_OP_ROUND && nr > 0 && nr <= WAVE) {
            /* the round's payload bytes start their way across PCIe now, in parallel with the
             * descriptors: [ar_off - 16, ar_off + ar_bytes + 16) of the pinned arena -> LDS */
            const bool pre = ar_bytes != 0 && ar_bytes <= P_PREFETCH_BYTES;
            constexpr int PRE_PER_THREAD = (P_PREFETCH_BYTES + 32) / 16 / 256 + 1;
            uint4 pf[PRE_PER_THREAD];
            const uint32_t pre_units = pre ? (ar_bytes + 32 + 15) / 16 : 0;
            if (pre) {
#pragma unroll
                for (int k = 0; k < PRE_PER_THREAD; k++) {
                    const uint32_t u = tid + k * 256;
                    if (u < pre_units) pf[k] = *(const uint4 *)(H->arena + (ar_off - 16) + 16 * u);
                }
            }
            /* ---- get_tailq_message + log_append_entry: one lane per entry ---- */
            if (tid < WAVE) {
                const bool active = lane < nr;
                const uint64_t g = (req_first + lane) % P_REQ_CAP;
                ReqDev d; d.req_id = 0; d.pay16_type = 0; d.len = 0; d.clt_id = 0;
                if (active) d = H->req[g];
                const uint32_t T = active ? APUS_HDR + d.len : 0;
                const uint64_t incl = wave_incl_scan((uint64_t)T);
                const uint64_t e0 = mh[H_END];
                const uint64_t a = e0 + incl - T;
                /* the entry that does not fit before len wraps (dare_log.h:502-538) */
                const unsigned long long over = __ballot(active && a + T > L);
                SeqOut s;
                s.e0 = e0; s.idx0 = mh[H_LAST_IDX] + 1; s.n_end0 = n_end0; s.term = my_term;
                s.kstar = -1; s.estar = -1; s.stale = 0; s.w = 0; s.n = nr;
                s.first_fail = ~0ull; s.commit_before = mh[H_COMMIT]; s.n_commit_before = mh[H_N_COMMIT];
                if (over) {
                    const int ks = __builtin_ctzll(over);
                    s.kstar = ks;
                    s.w = __shfl(a, ks, WAVE);
                    if (s.w == L) s.estar = ks; else if (L - s.w >= APUS_HDR) s.stale = 1;
                }
                /* log_append_entry refuses a full log (end == head, dare_log.h:168,492-495); the engine refuses
                 * the whole round when it does not fit into the free part of the ring (the reference would
                 * run over un-pruned entries): nothing is stored, the host is told (H->full) */
                bool refuse = false;
                if (e0 != L) {
                    const uint64_t head = mh[H_HEAD];
                    const uint64_t total = __shfl(incl, (int)nr - 1, WAVE), waste = over ? L - s.w : 0;
                    const uint64_t used = e0 >= head ? e0 - head : L - (head - e0);
                    refuse = e0 == head || total + waste > L - used;
                }
                if (refuse) {
                    if (lane == 0) { set_status(E, 1u << 1); st_sys(&H->full, 1); s_word[0] = 8; }
                } else {
                const int64_t gk = lane;
                const uint64_t pos = apus_place(s, gk, a);
                const uint64_t idx = apus_entry_idx(s, gk);
                const uint32_t nu = active ? (T + 15) / 16 : 0;
                const uint32_t uincl = wave_incl_scan(nu);
                const uint32_t type = d.pay16_type >> 28;
                const uint4 h0 = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)s.term, (uint32_t)(s.term >> 32));
                const uint4 h1 = make_uint4((uint32_t)d.req_id, (uint32_t)(d.req_id >> 32),
                                            (uint32_t)d.clt_id | (type << 16) | ((uint32_t)E.leader << 24), 0);
                lds.pos[lane] = pos;
                lds.src[lane] = (uint64_t)(d.pay16_type & 0x0FFFFFFFu) * 16;
                lds.T[lane] = T;
                lds.ubase[lane] = uincl - nu;
                lds.h0[lane] = h0; lds.h1[lane] = h1;
                if (lane == WAVE - 1) { lds.ubase[WAVE] = uincl; lds.uniform_nu = 0; }
                if (active) {
                    const uint32_t di = (uint32_t)(n_end0 + lane) & E.dir_mask;
                    const uint32_t dl = T | ((uint32_t)E.leader << 24);
                    for (uint32_t m = push_mask | (1u << me); m; m &= m - 1) {
                        const RepDev &Fd = E.rep[__builtin_ctz(m)];
                        st_agent(&Fd.dir_off[di], pos);
                        __hip_atomic_store(&Fd.dir_len[di], dl, RLX_AGENT);
                    }
                    __hip_atomic_store(&Md.ack[di], 0u, RLX_AGENT);
                    if (s.stale && gk == s.kstar) {
                        const uint4 z = make_uint4(0, 0, 0, 0), l = make_uint4((uint32_t)d.len, 0, 0, 0);
                        for (uint32_t m = push_mask | (1u << me); m; m &= m - 1) {
                            uint8_t *rg = E.rep[__builtin_ctz(m)].ring;
                            st16_agent(rg + a, h0); st16_agent(rg + a + 16, h1);
                            st16_agent(rg + a + 32, z); st16_agent(rg + a + 48, l);
                        }
                    }
                }
                if (lane == nr - 1) {
                    /* leader control words: end, tail, persist (dare_log.h:547-549, dare_server.c:1792) */
                    const uint64_t end_new = pos + T;
                    mh[H_END] = end_new; mh[H_TAIL] = pos; mh[H_N_END] = n_end0 + nr;
                    mh[H_LAST_IDX] = idx; mh[H_PREV_HEAD] = 0;
                    mh[H_OLD_END] = end_new; mh[H_N_PERSIST] = n_end0 + nr; mh[H_STORE_COUNT] += nr;
                    s_word[4] = end_new;
                }
                if (lane == 0) s_seq = s;
                }
            }
            if (pre) {
#pragma unroll
                for (int k = 0; k < PRE_PER_THREAD; k++) {
                    const uint32_t u = tid + k * 256;
                    if (u < pre_units) ((uint4 *)s_pay)[u] = pf[k];
                }
            }
            __syncthreads();
            const bool refused = s_word[0] == 8;
            if (tid == 0) s_word[7] = wall_clock64();
            /* ---- the round's bytes: own ring + R1 to every in-sync follower ---- */
            const uint32_t utotal = refused ? 0u : lds.ubase[WAVE];
            constexpr int PUSH_ILP = 4;           /* units per thread and pass: the PCIe payload reads overlap */
            for (uint32_t u0 = tid; u0 < utotal; u0 += blockDim.x * PUSH_ILP) {
                uint4 v[PUSH_ILP];
                uint64_t p[PUSH_ILP];
                bool on[PUSH_ILP];
#pragma unroll
                for (int k = 0; k < PUSH_ILP; k++) {
                    const uint32_t u = u0 + k * blockDim.x;
                    on[k] = u < utotal;
                    v[k] = make_uint4(0, 0, 0, 0); p[k] = 0;
                    if (!on[k]) continue;
                    uint32_t lo = 0, hi = nr - 1;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi + 1) >> 1;
                        if (lds.ubase[mid] <= u) lo = mid; else hi = mid - 1;
                    }
                    const uint32_t e = lo, Te = lds.T[e], j = u - lds.ubase[e];
                    const uint32_t so = min(16u * j, Te - 16u);
                    if (so == 0) v[k] = lds.h0[e];
                    else if (so == 16) v[k] = lds.h1[e];
                    else if (so == 32) v[k] = make_uint4(0, 0, 0, 0);
                    else if (pre) v[k] = payload_unit_lds(s_pay, (uint32_t)(lds.src[e] - (ar_off - 16)), so, Te - APUS_HDR, Te - APUS_HDR);
                    else v[k] = payload_unit(H->arena + lds.src[e], so, Te - APUS_HDR, Te - APUS_HDR);
                    p[k] = lds.pos[e] + so;
                }
#pragma unroll
                for (int k = 0; k < PUSH_ILP; k++)
                    if (on[k])
                        for (uint32_t m = push_mask | (1u << me); m; m &= m - 1) st16_agent(E.rep[__builtin_ctz(m)].ring + p[k], v[k]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          /* every storing wave drains */
            __syncthreads();
            appended = !refused;
            if (tid == 0) s_word[0] = 0;
        } else if (op == P_OP_PRUNE) {
            /* log_pruning: wait until the reachable followers applied what is committed
             * (the timer fires between polling() passes), then decide */
            if (tid == 0) {
                const uint64_t want = mh[H_N_COMMIT];
                uint64_t seen;
                for (uint32_t m = push_mask; m; m &= m - 1)
                    if (!wait_ge_agent(&D->applied_bell[__builtin_ctz(m)], want, peer_polls, &H->stop, &seen)) s_word[0] = 9;
                const uint64_t end = mh[H_END];
                const uint32_t bitmask = (uint32_t)mh[H_CID_BITMASK];
                uint64_t min_off = mh[H_APPLY];
                for (uint32_t i = 0; i < E.group_size; i++) {
                    if (!((bitmask >> i) & 1u)) mh[H_APPLY_OFFSETS + i] = mh[H_APPLY];
                    if (apus_is_larger(end, L, min_off, mh[H_APPLY_OFFSETS + i])) min_off = mh[H_APPLY_OFFSETS + i];
                }
                if (apus_end_distance(end, L, min_off) == 0) min_off = mh[H_TAIL];
                bool do_append = apus_is_larger(end, L, min_off, mh[H_HEAD]) && !mh[H_PREV_HEAD];
                s_word[5] = 0;
                if (do_append) {
                    mh[H_HEAD] = min_off;
                    const uint64_t idx = (end == L) ? 1 : mh[H_LAST_IDX] + 1;
                    const uint64_t pos = (end == L || L - end < APUS_HDR) ? 0 : end;
                    const uint32_t di = (uint32_t)n_end0 & E.dir_mask;
                    const uint4 h0 = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)my_term, (uint32_t)(my_term >> 32));
                    const uint4 h1 = make_uint4(0, 0, (3u << 16) | ((uint32_t)E.leader << 24), 0);
                    const uint4 h3 = make_uint4((uint32_t)min_off, (uint32_t)(min_off >> 32), 0, 0);
                    for (uint32_t m = push_mask | (1u << me); m; m &= m - 1) {
                        const RepDev &Fd = E.rep[__builtin_ctz(m)];
                        st16_agent(Fd.ring + pos, h0); st16_agent(Fd.ring + pos + 16, h1);
                        st16_agent(Fd.ring + pos + 32, make_uint4(0, 0, 0, 0)); st16_agent(Fd.ring + pos + 48, h3);
                        st_agent(&Fd.dir_off[di], pos);
                        __hip_atomic_store(&Fd.dir_len[di], APUS_HDR | ((uint32_t)E.leader << 24), RLX_AGENT);
                    }
                    __hip_atomic_store(&Md.ack[di], 0u, RLX_AGENT);
                    mh[H_PREV_HEAD] = 1;
                    mh[H_TAIL] = pos; mh[H_END] = pos + APUS_HDR; mh[H_N_END] = n_end0 + 1; mh[H_LAST_IDX] = idx;
                    mh[H_OLD_END] = pos + APUS_HDR; mh[H_N_PERSIST] = n_end0 + 1; mh[H_STORE_COUNT] += 1;
                    s_word[4] = pos + APUS_HDR;
                    s_word[5] = 1;
                }
                /* READ the apply offsets for the next tick (rc_get_remote_apply_offsets) */
                for (uint32_t i = 0; i < E.group_size; i++) {
                    if (i == E.leader || !((bitmask >> i) & 1u)) { mh[H_APPLY_OFFSETS + i] = mh[H_APPLY]; continue; }
                    if (!((push_mask >> i) & 1u)) continue;
                    mh[H_APPLY_OFFSETS + i] = ld_agent(&E.rep[i].hdr[H_APPLY]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            appended = s_word[5] != 0;
        }

        const uint64_t n_end = appended ? ((op == P_OP_ROUND) ? n_end0 + nr : n_end0 + 1) : n_end0;
        const uint64_t end_off = appended ? s_word[4] : mh[H_END];
        /* the log reads as empty when end sits on len: nothing of this round is visible yet */
        const uint64_t vis = (end_off == L) ? n_end0 : n_end;

        /* A burst: when the host has already queued the next ROUND, this round's doorbell / ACK
         * aggregation / commit / apply ride with the next one's (up to P_BURST rounds): the followers
         * then persist and ACK the burst in one go and the ballot loop below commits it in windows of
         * 64.  Polling() does the same when requests queue up (one pass takes the whole tailq); batch
         * boundaries never change the logs.  A lone round is never held back. */
        if (tid == 0) {
            uint32_t defer = 0;
            if (op == P_OP_ROUND && appended && burst < P_BURST && ld_sys(&H->ev_tail) > ev_head + 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                if (H->ev[(ev_head + 1) % P_EV_CAP].op == P_OP_ROUND) defer = 1;
            }
            s_word[2] = defer;
        }
        __syncthreads();
        const bool defer = s_word[2] != 0;
        burst = defer ? burst + 1 : 0;
        __syncthreads();

        if (!defer && vis > mh[H_N_COMMIT]) {
            /* ---- R2: end doorbell to every in-sync follower ---- */
            uint64_t t_bell = 0;
            if (tid == 0) {
                for (uint32_t m = push_mask; m; m &= m - 1) st_agent(&D->end_bell[__builtin_ctz(m)], vis);
                t_bell = wall_clock64();
            }

            /* ---- ACK aggregation: window of <= 64 entries per pass, bitmaps in LDS ---- */
            uint64_t cs = mh[H_N_COMMIT];
            const uint32_t size = E.group_size, quorum = size / 2 + 1, size_mask = (1u << size) - 1;
            const bool can_commit = (uint32_t)__popc((push_mask | (1u << me)) & size_mask) >= quorum;
            if (tid < WAVE) {
                uint64_t polls = 0;
                while (cs < vis) {
                    const uint64_t s = cs + lane;
                    const bool in = s < vis;
                    s_ack[lane] = in ? __hip_atomic_load(&Md.ack[(uint32_t)s & E.dir_mask], RLX_AGENT) : 0xFFFFFFFFu;
                    const uint32_t m = (s_ack[lane] | (1u << me)) & size_mask;
                    const bool ok = !in || (uint32_t)__popc(m) >= quorum;       /* replies >= size/2+1 */
                    const unsigned long long bal = __ballot(ok);
                    const uint32_t prefix = (~bal) ? (uint32_t)__builtin_ctzll(~bal) : WAVE;   /* trailing ones */
                    cs += min((uint64_t)prefix, vis - cs);
                    if (prefix < WAVE && cs < vis) {
                        if (!can_commit || ++polls > peer_polls) break;          /* like the 1000-pass threshold, dare_ibv_rc.c:1939 */
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (lane == 0) s_word[6] = cs;
            }
            __syncthreads();
            cs = s_word[6];
            if (cs > mh[H_N_COMMIT]) {
                /* ---- commit, R4 commit doorbell, apply ---- */
                if (tid == 0) {
                    const uint64_t coff = (cs == n_end) ? end_off : Md.dir_off[(uint32_t)cs & E.dir_mask];
                    mh[H_COMMIT] = coff; mh[H_N_COMMIT] = cs; mh[H_N_VISIBLE] = vis;
                    for (uint32_t m = push_mask; m; m &= m - 1) st_agent(&D->commit_bell[__builtin_ctz(m)], cs);
                    const uint32_t k = D->lat_n;
                    if (appended && k < P_LAT_CAP) {
                        D->lat_ticks[k] = (uint32_t)(wall_clock64() - t_start);
                        D->lat_seq[k] = (op == P_OP_ROUND) ? (uint32_t)(s_word[7] - t_start) : 0;
                        D->lat_push[k] = (uint32_t)(t_bell - t_start);
                        D->lat_n = k + 1;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                apply_range(E, me, mh[H_N_APPLY], cs, 0, blockDim.x, s_acc);
                if (tid == 0) {
                    const uint64_t coff = mh[H_COMMIT];
                    mh[H_APPLY] = coff; mh[H_N_APPLY] = cs;
                    /* highest_rec: what the blocked submitters spin on (proxy.c:160) */
                    st_sys(&H->highest_rec, ld_agent(&mh[H_HIGHEST_REC]));
                    st_sys(&H->commit_slot, cs);
                }
            }
        }
        ev_head++;
        if (tid == 0) { st_sys(&H->ev_head, ev_head); st_sys(&H->rounds_done, ev_head); }
        __syncthreads();
    }

    if (tid == 0) {
        st_agent(&D->quit, 1);
        if (exit_code == 2) atomicOr(E.status, 1u << 4);
        st_sys(&H->exit_code, exit_code);
        st_sys(&H->ev_head, ev_head);
        __threadfence_system();
        st_sys(&H->alive, 2);
    }
}
