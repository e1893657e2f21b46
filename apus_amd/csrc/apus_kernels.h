/*
 * apus_kernels.h -- the HIP kernels of the consensus hot path (gfx950, wave64).
 *
 * One leader polling() pass of the reference (dare_server.c:1012-1125) over a
 * batch of R rounds is four launches; every launch is wide (one workgroup per
 * round / one lane per entry) and the launches are the only global syncs:
 *
 *   k_sequence        get_tailq_message -> log_append_entry offsets
 *                     (dare_ibv_ud.c:780, dare_log.h:466-558): where every entry of
 *                     the batch goes, incl. both wrap rules and the end == len
 *                     "empty" encoding; leader persist bookkeeping; a log_pruning
 *                     tick that was due right before the batch is fused in
 *   k_append_push     writes the entries into the leader ring (log_append_entry
 *                     body + persist_new_entries' sender stamp, dare_server.c:1803)
 *                     AND into every in-sync follower ring at the same offsets
 *                     (R1 of update_remote_logs, fused so the payload is read once)
 *   k_persist_commit  follower persist_new_entries + rc_send_entries_reply
 *                     (dare_server.c:1792, dare_ibv_rc.c:1828): reply byte in both
 *                     logs + ACK bit in the leader's slot word, and the ACK scan of
 *                     update_remote_logs (dare_ibv_rc.c:1725-1758):
 *                     popcount(ack | self) >= size/2+1 per lane, wave ballot, first
 *                     slot without a majority   (k_commit: the scan alone, used when
 *                     the ACK bits were merged from followers in other processes)
 *   k_apply           apply_committed_entries on every replica
 *                     (dare_server.c:1815-1974): apply-stream records, HEAD adoption;
 *                     every block computes a slice of the per-round commit record,
 *                     the block that finishes last does the scalar bookkeeping
 *                     (commit/apply offsets, R2/R4 doorbells)
 *
 *   k_control_round   one workgroup: a whole pass that carries at most one control
 *                     entry (CONFIG / HEAD / NOOP, prune tick, quiesce)
 *   k_catchup         update_remote_logs step I for followers that lag by a lot
 *                     (dare_ibv_rc.c:1507-1547): copy [remote_end, end)
 *   k_mp_*            device halves of the multi-process (one replica per GPU) exchange
 *
 * All arithmetic is integer / byte work; the binding roofline is HBM.
 */
#pragma once
#include "apus_device.h"

#define WAVE 64

/* Diagnostics (python -m apus_amd.build --trace): thread 0 of the calling block drains its
 * memory queue and records the 100 MHz wall clock; tools/trace_probe.py prints the deltas.
 * Compiles to nothing in the product library. */
#ifdef APUS_TRACE
#define STAMP(K, k) do { if (threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); E.trace[(K) * 64 + (k)] = wall_clock64(); } } while (0)
#define STAMPW(K, k, T) do { if (threadIdx.x == (T)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); E.trace[(K) * 64 + (k)] = wall_clock64(); } } while (0)
/* STAMPN: no drain -- the time the code got here, not the time its memory operations finished */
#define STAMPN(K, k) do { if (threadIdx.x == 0) E.trace[(K) * 64 + (k)] = wall_clock64(); } while (0)
#else
#define STAMP(K, k) do { } while (0)
#define STAMPW(K, k, T) do { } while (0)
#define STAMPN(K, k) do { } while (0)
#endif

/* a control word (uncached device memory) read past L1 / L2: blocks of one launch may read words
 * an earlier part of the same launch wrote */
__device__ static inline uint64_t ldw(const uint64_t *p)
{
    return __hip_atomic_load((const APUS_GLOBAL unsigned long long *)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* where a call's blocks meet: arrival counters and the per-round hash scratch.  One set per
 * engine for k_call / the phased kernels; one per segment in a multi-segment launch (k_step). */
struct CallEnv {
    uint32_t *lines;      /* [32 lines x 32 words] word 0 append arrivals, word 1 the sequencer's flag, word 2 "inputs fetched" */
    uint32_t *ticket;     /* [8] T_PASS, T_SCAN, T_DONE */
    uint64_t *hash;       /* [2 x rounds] fast path: per-round apply-stream sums */
};
#define APUS_ENV_OF(E) CallEnv{(E).tick_lines, (E).ticket, (E).round_hash}

/* E.ticket words */
enum { T_APPLY = 1, T_PASS = 2, T_SCAN = 3, T_DONE = 4, T_JANITORS = 8 /* segment 0's block only: janitors of a launch that are through */ };

__device__ static inline uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }

/* inclusive scan across the 64 lanes of a wavefront */
template <typename T>
__device__ static inline T wave_incl_scan(T v)
{
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        T o = __shfl_up(v, d, WAVE);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

template <typename T>
__device__ static inline T wave_sum(T v)
{
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

#define FENCE_WORD 2      /* status[2]: raised by k_fence_check, every fenced launch leaves when it is set */
__device__ static inline void set_status(const EngDev &E, uint32_t bit) { atomicOr(E.status, bit); }
/* a bounded spin ran out: status bit 4; the first site to do so leaves its source line in status word 1 */
__device__ static inline void spin_timeout(const EngDev &E, uint32_t site) { atomicOr(E.status, 1u << 4); atomicCAS(E.status + 1, 0u, site); }

/* bytes [b0, b1) of a little-endian u64 set to 0xFF (0 <= b0, b1 <= 8) */
__device__ static inline uint64_t byte_mask64(int b0, int b1)
{
    if (b0 < 0) b0 = 0;
    if (b1 > 8) b1 = 8;
    if (b1 <= b0) return 0;
    const uint64_t hi = (b1 == 8) ? ~0ull : ((1ull << (8 * b1)) - 1);
    const uint64_t lo = (b0 == 0) ? 0ull : ((1ull << (8 * b0)) - 1);
    return hi & ~lo;
}

/* 16 bytes [so, so+16) of the byte stream of a client entry, so >= 48:
 *   [48,50) cmd.len   [50,50+P) payload   [50+P, 64+P) unused (written as 0)
 * src points at payload byte 0 (readable from src-2 to src+P+14). */
__device__ static inline uint4 payload_unit(const uint8_t *src, uint32_t so, uint32_t P, uint32_t len16)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (P == 0 && so >= 50) return v;
    uint64_t lo = 0, hi = 0;
    if (P != 0) {
        v = ld16u(src + (int64_t)so - 50);
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

/* ------------------------------------------------------------------------- */
/* R1 for a follower whose log is behind the leader's (released after a HOLD, or
 * left one round behind by an exact-fit wrap): copy [end_f, end_l) and the
 * matching directory slots (update_remote_logs step I, dare_ibv_rc.c:1507-1547;
 * a wrapped range is two pieces, :1538-1545).  Cooperative over (tid, nth).   */
__device__ static inline void catchup_range(const EngDev &E, int f, uint64_t end_l, uint64_t n_l,
                                            uint64_t tid, uint64_t nth)
{
    const RepDev &Ld = E.rep[E.leader];
    const RepDev &Fd = E.rep[f];
    const uint64_t L = E.log_len;
    const uint64_t end_f = Fd.hdr[H_END];
    const uint64_t n_f = Fd.hdr[H_N_END];
    if (n_f >= n_l || end_l == L) return;            /* in sync, or nothing visible yet */
    for (uint64_t s = n_f + tid; s < n_l; s += nth) {
        const uint32_t i = (uint32_t)s & E.dir_mask;
        Fd.dir_off[i] = Ld.dir_off[i];
        Fd.dir_len[i] = Ld.dir_len[i];
    }
    const uint64_t from = (end_f == L) ? 0 : end_f;
    uint64_t seg0_to, seg1_to = 0;
    if (end_l > from || end_f == L) seg0_to = end_l;
    else { seg0_to = L; seg1_to = end_l; }
    for (int seg = 0; seg < 2; seg++) {
        const uint64_t a = seg ? 0 : from, b = seg ? seg1_to : seg0_to;
        if (b <= a) continue;
        const uint64_t nunit = (b - a) / 16;
        for (uint64_t u = tid; u < nunit; u += nth) st16u(Fd.ring + a + 16 * u, ld16u(Ld.ring + a + 16 * u));
        for (uint64_t x = a + 16 * nunit + tid; x < b; x += nth) Fd.ring[x] = Ld.ring[x];
    }
}

/* k_catchup: wide version, launched by the host when it knows a follower may lag
 * by a lot (RELEASE after a HOLD).  grid.y = follower ordinal in fmask.        */
__global__ __launch_bounds__(256) void k_catchup(const EngDev E, uint32_t fmask)
{
    if (E.status[FENCE_WORD]) return;                 /* the term fence (k_fence_check) */
    int f = -1;
    for (int i = 0, k = 0; i < APUS_DEV_MAX_SERVERS; i++)
        if (fmask & (1u << i)) { if (k == (int)blockIdx.y) { f = i; break; } k++; }
    if (f < 0) return;
    const uint64_t *lh = E.rep[E.leader].hdr;
    catchup_range(E, f, lh[H_END], lh[H_N_END], (uint64_t)blockIdx.x * blockDim.x + threadIdx.x,
                  (uint64_t)gridDim.x * blockDim.x);
}

template <bool FX>
__device__ static inline SeqOut control_append(const EngDev &E, int mode, uint32_t type, uint64_t d0, uint64_t d1,
                                               uint32_t push_mask, uint64_t *s_lh, uint64_t rec_base, uint32_t ack_mask,
                                               bool apply_now, bool note_head_slot = true, uint64_t req_id = 0, uint32_t clt_id = 0);
__device__ static inline void sample_apply_offsets(const EngDev &E, const uint64_t *s_lh, uint32_t sample_mask, uint32_t i,
                                                   const uint64_t *staged_apply);

/* block-wide inclusive scan of one u64 per thread (blockDim.x <= 1024), returns
 * the inclusive value; *total gets the block sum                              */
__device__ static inline uint64_t block_incl_scan(uint64_t v, uint64_t *s_tot /*[16]*/, uint64_t *total)
{
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint64_t incl = wave_incl_scan(v);
    __syncthreads();
    if (lane == WAVE - 1) s_tot[wv] = incl;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (uint32_t w = 0; w < nw; w++) { if (w < wv) base += s_tot[w]; tot += s_tot[w]; }
    *total = tot;
    return base + incl;
}

/* ------------------------------------------------------------------------- */
/* k_sequence: where every entry of rounds [r0, r0+R) goes -- one workgroup: exclusive
 * scan over the rounds' byte totals (staged with the batch: the admission side knows every
 * request's size, so a round arrives with its total), the wrap point, the leader's
 * control words and the per-round end record.                                  */
/* A control-state snapshot: everything a block stages before it sequences a call -- the leader's
 * control block, eight words per follower, the record count.  In a multi-segment launch (k_step)
 * the bookkeeper of segment k writes snapshot k+1 (write-once, uncached memory), so later
 * segments never read words an earlier one is still changing. */
#define SNAP_FW        64
#define SNAP_REC       (SNAP_FW + 8 * APUS_DEV_MAX_SERVERS)
#define SNAP_WORDS     (SNAP_REC + 1)
#define SNAP_STRIDE    192            /* u64 words per snapshot */
/* follower words in snapshot / bookkeeper order */
enum { FW_END = 0, FW_N_END, FW_APPLY, FW_N_PERSIST, FW_N_APPLY, FW_N_COMMIT, FW_STORE_COUNT, FW_HEAD };
__device__ static inline int fw_hdr_word(uint32_t j)
{
    switch (j) {
    case FW_END: return H_END;            case FW_N_END: return H_N_END;        case FW_APPLY: return H_APPLY;
    case FW_N_PERSIST: return H_N_PERSIST; case FW_N_APPLY: return H_N_APPLY;    case FW_N_COMMIT: return H_N_COMMIT;
    case FW_STORE_COUNT: return H_STORE_COUNT; default: return H_HEAD;
    }
}
/* a staged control word: from a snapshot, or (snap == nullptr) from the live control blocks */
__device__ static inline uint64_t stage_leader_word(const EngDev &E, const uint64_t *snap, uint32_t i)
{
    return snap ? ldw(&snap[i]) : ldw(&E.rep[E.leader].hdr[i]);
}
__device__ static inline uint64_t stage_follower_word(const EngDev &E, const uint64_t *snap, uint32_t f, uint32_t j)
{
    return snap ? ldw(&snap[SNAP_FW + 8 * f + j]) : ldw(&E.rep[f].hdr[fw_hdr_word(j)]);
}
__device__ static inline uint64_t stage_rec_count(const EngDev &E, const uint64_t *snap)
{
    return snap ? ldw(&snap[SNAP_REC]) : ldw(E.rec_count);
}

/* what a batch of n entries does to the leader's control block (log_append_entry's offsets and
 * the leader side of persist_new_entries, dare_server.c:1792-1810) */
__device__ static inline void seq_apply_batch(uint64_t *lh, uint64_t end_new, uint64_t t_last, uint64_t n_end0, uint32_t n,
                                              uint64_t idx0, int64_t estar)
{
    lh[H_END] = end_new;
    lh[H_TAIL] = end_new - t_last;
    lh[H_N_END] = n_end0 + n;
    lh[H_LAST_IDX] = (estar < 0) ? idx0 + n - 1 : 1 + (uint64_t)(n - 1 - estar);
    lh[H_PREV_HEAD] = 0;
    lh[H_OLD_END] = end_new;
    lh[H_N_PERSIST] = n_end0 + n;
    lh[H_STORE_COUNT] += n;
}

struct SeqLds {
    uint64_t lh[64];                      /* the leader's control block, kept current */
    uint64_t fw[APUS_DEV_MAX_SERVERS][5]; /* followers: end, n_end, apply, n_persist, n_apply */
    uint64_t misc[2];                     /* rec_count, len of the batch's last request */
    uint64_t tot[16];
    uint64_t w;
    int64_t  kstar;
    uint32_t head_round;
    unsigned int rstar;
    uint64_t virt[1025];                  /* exclusive scan of the round totals when R <= 1024 */
    uint32_t bytes0[1024];                /* byte totals of the first 1024 rounds */
    uint64_t virt_rstar;                  /* exclusive prefix of the round that crosses len */
    uint64_t my_virt;                     /* FX = false: where round my_r starts */
    uint64_t pfx[3];                      /* wave-0 variant: staged byte prefix at round 0, my_r, R of the call */
    uint32_t rfx[2];                      /* wave-0 variant: first request of round 0 and of round R */
    uint32_t ok;                          /* wave-0 variant: 1 = SeqOut worked out, 0 = take the block-wide path */
    uint32_t chain_did;                   /* rec_wait: the chain block does this segment's sequencer effects itself */
    uint32_t refused;                     /* rec_wait: the segment was refused (segment_refused): nothing is stored */
    uint64_t end_new;                     /* the leader's end offset after the batch */
    SeqOut   out;                         /* FX = false: the call's SeqOut */
};

/* which pushed followers persist + ACK with the push itself (they have persisted everything so
 * far), and whether every replica on the device has persisted, committed and applied everything */
__device__ static inline void seq_in_step(const EngDev &E, const uint64_t *lh, const uint64_t (*fw)[5], uint32_t push_mask,
                                          uint32_t &fuse_mask, bool &in_step)
{
    const uint64_t L = E.log_len;
    const uint64_t e_pre = lh[H_END], n_pre = lh[H_N_END];
    fuse_mask = 0;
    for (uint32_t m = push_mask; m; m &= m - 1)
        if (fw[__builtin_ctz(m)][3] == n_pre && e_pre != L) fuse_mask |= 1u << __builtin_ctz(m);
    const uint32_t size = E.group_size, size_mask = (1u << size) - 1;
    in_step = fuse_mask == push_mask && (uint32_t)__popc((fuse_mask | (1u << E.leader)) & size_mask) >= size / 2 + 1
           && lh[H_N_COMMIT] == n_pre && lh[H_N_APPLY] == n_pre;
    for (uint32_t m = push_mask; m; m &= m - 1) in_step = in_step && fw[__builtin_ctz(m)][4] == n_pre;
}

/* sample_apply_offsets, applied to an LDS copy of the leader's control block (the stores to
 * the control block itself are the sequencer's) */
__device__ static inline void sample_into_copy(const EngDev &E, uint64_t *lh, uint32_t sample_mask, uint32_t i, uint64_t staged_apply)
{
    const uint32_t bitmask = (uint32_t)lh[H_CID_BITMASK];
    if (i >= E.group_size) return;
    if (i == E.leader || !((bitmask >> i) & 1u)) lh[H_APPLY_OFFSETS + i] = lh[H_APPLY];
    else if ((sample_mask >> i) & 1u) lh[H_APPLY_OFFSETS + i] = staged_apply;
}

/* the batch-level switches of SeqOut; hidden = the batch ends exactly on len (its last round
 * stays invisible and is not pushed as acknowledged) */
struct SeqFlags { uint32_t fuse_batch, tail_needed, fast; };
__device__ static inline SeqFlags seq_flags(const EngDev &E, uint32_t push_mask, uint32_t fuse_mask, bool in_step, bool hidden,
                                            uint32_t n, uint64_t n_commit_before, uint64_t n_end0, uint32_t head_round)
{
    SeqFlags f;
    f.fuse_batch = hidden ? 0u : fuse_mask;
    const uint32_t size = E.group_size, size_mask = (1u << size) - 1;
    const bool quorum_fused = (uint32_t)__popc((f.fuse_batch | (1u << E.leader)) & size_mask) >= size / 2 + 1;
    f.tail_needed = ((push_mask & ~f.fuse_batch) != 0 || !quorum_fused || n_commit_before < n_end0 - head_round) ? 1u : 0u;
    f.fast = (in_step && !f.tail_needed && !hidden && n) ? 1u : 0u;
    return f;
}

/* The sequencer, for one workgroup of 256 .. 1024 threads, in two steps.
 * seq_stage: everything it needs from HBM, requested by different lanes in one round trip.
 * seq_body<FX>: FX = true is the sequencer proper (control words, SeqOut, round_virt, the
 * per-round end record, a due prune tick's <HEAD> entry, catch-up of lagging followers).
 * FX = false computes the same SeqOut from the same inputs with NO store to HBM: in k_call every
 * append block does that for itself instead of waiting for the sequencer block (q.out, and
 * q.my_virt = where round my_r starts). */
__device__ static inline void seq_stage(const EngDev &E, uint64_t r0, uint32_t R, uint32_t push_mask,
                                        uint32_t sample_mask, SeqLds &q, bool need_tail = true, const uint64_t *snap = nullptr)
{
    uint64_t (&s_lh)[64] = q.lh;
    uint64_t (&s_fw)[APUS_DEV_MAX_SERVERS][5] = q.fw;
    uint64_t (&s_misc)[2] = q.misc;
    uint32_t &s_head_round = q.head_round;
    unsigned int &s_rstar = q.rstar;
    const uint32_t tid = threadIdx.x;
    const uint32_t *rf = E.round_first + r0;
    const uint32_t *rb = E.round_bytes + r0;

    /* Everything the block needs from HBM is requested NOW, by different lanes: one round
     * trip instead of a chain of dependent loads later. */
    if (blockIdx.x == 0) STAMP(0, 0);
    uint64_t st0 = 0, st1 = 0, st2 = 0, st3 = ~0ull, st4 = 0;
    for (uint32_t i = tid; i < R && i < 1024; i += blockDim.x) q.bytes0[i] = rb[i];   /* the rounds' byte totals */
    if (tid < 64) st0 = stage_leader_word(E, snap, tid);
    else if (tid < 64 + APUS_DEV_MAX_SERVERS) {
        const uint32_t f_ = tid - 64;
        if (((push_mask | sample_mask) >> f_) & 1u) {
            st0 = stage_follower_word(E, snap, f_, FW_END); st1 = stage_follower_word(E, snap, f_, FW_N_END);
            st2 = stage_follower_word(E, snap, f_, FW_APPLY);
            if ((push_mask >> f_) & 1u) { st3 = stage_follower_word(E, snap, f_, FW_N_PERSIST); st4 = stage_follower_word(E, snap, f_, FW_N_APPLY); }
        }
    } else if (tid == 96) st0 = stage_rec_count(E, snap);
    else if (tid == 97 && need_tail) st0 = (rf[R] > rf[0]) ? E.req_len[rf[R] - 1] : 0;   /* two dependent loads: only the real sequencer */
    if (tid < 64) s_lh[tid] = st0;
    else if (tid < 64 + APUS_DEV_MAX_SERVERS) { s_fw[tid - 64][0] = st0; s_fw[tid - 64][1] = st1; s_fw[tid - 64][2] = st2; s_fw[tid - 64][3] = st3; s_fw[tid - 64][4] = st4; }
    else if (tid == 96) s_misc[0] = st0;
    else if (tid == 97) s_misc[1] = st0;
    else if (tid == 98) { q.pfx[0] = E.round_prefix ? E.round_prefix[r0] : 0; q.pfx[1] = q.pfx[0]; }      /* for seq_w0_decide */
    else if (tid == 99) q.pfx[2] = E.round_prefix ? E.round_prefix[r0 + R] : ~0ull;
    else if (tid == 100) { q.rfx[0] = rf[0]; q.rfx[1] = rf[R]; }
    if (tid == 0) { s_rstar = 0xFFFFFFFFu; s_head_round = 0; }
    __syncthreads();
    if (blockIdx.x == 0) STAMP(0, 1);
}

template <bool FX>
__device__ static inline void seq_body(const EngDev &E, uint64_t r0, uint32_t R, uint32_t push_mask,
                                       uint32_t tick, uint32_t sample_mask, SeqLds &q, uint32_t my_r, bool write_rec = true)
{
    uint64_t (&s_lh)[64] = q.lh;
    uint64_t (&s_fw)[APUS_DEV_MAX_SERVERS][5] = q.fw;
    uint64_t (&s_misc)[2] = q.misc;
    uint32_t &s_head_round = q.head_round;
    uint64_t (&s_tot)[16] = q.tot;
    unsigned int &s_rstar = q.rstar;
    int64_t &s_kstar = q.kstar;
    uint64_t &s_w = q.w;
    uint64_t (&s_virt)[1025] = q.virt;
    const uint32_t BD = blockDim.x;
    const uint32_t tid = threadIdx.x;
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *hdr = Ld.hdr;
    const uint64_t L = E.log_len;
    const uint32_t *rf = E.round_first + r0;
    const uint32_t *rb = E.round_bytes + r0;

    const uint32_t g0 = rf[0];
    const uint32_t n = rf[R] - g0;
    uint32_t fuse_mask = 0;
    bool in_step = false;       /* every replica on the device has persisted, committed and applied everything so far */
    {
        const uint64_t e_pre = s_lh[H_END], n_pre = s_lh[H_N_END];
        seq_in_step(E, s_lh, s_fw, push_mask, fuse_mask, in_step);
        /* followers that silently fell behind (hidden exact-fit round) are caught up here */
        bool any_lag = false;
        for (uint32_t m = push_mask; m; m &= m - 1) any_lag |= (s_fw[__builtin_ctz(m)][1] < n_pre) && e_pre != L;
        if (FX && any_lag) {
            for (uint32_t m = push_mask; m; m &= m - 1) catchup_range(E, __builtin_ctz(m), e_pre, n_pre, tid, blockDim.x);
            __syncthreads();
            if (tid == 0)
                for (uint32_t m = push_mask; m; m &= m - 1) {
                    const int f = __builtin_ctz(m);
                    if (s_fw[f][1] < n_pre) { uint64_t *fh = E.rep[f].hdr; fh[H_END] = e_pre; fh[H_N_END] = n_pre; }
                }
        }
        /* a log_pruning tick that was due right before this batch (the timer fired between two
         * polling() passes): decision + <HEAD> entry happen here, its persist / ACK / commit /
         * apply ride with the batch's own tail kernels */
        if (tick) {
            if (FX && tid >= 64 && tid < 64 + APUS_DEV_MAX_SERVERS)
                sample_apply_offsets(E, s_lh, sample_mask, tid - 64, &s_fw[tid - 64][2]);
            __syncthreads();                             /* the sampling lanes read the pre-tick block */
            if (tid == 0) s_head_round = control_append<FX>(E, 1, 3, 0, 0, push_mask, s_lh, s_misc[0], fuse_mask, in_step, write_rec).n;
            __syncthreads();
            /* the offsets sampled above are what the NEXT tick decides from: note them in the LDS copy too */
            if (tid >= 64 && tid < 64 + APUS_DEV_MAX_SERVERS) sample_into_copy(E, s_lh, sample_mask, tid - 64, s_fw[tid - 64][2]);
            __syncthreads();
        }
    }
    if (blockIdx.x == 0) STAMP(0, 2);
    const uint32_t head_round = s_head_round;
    const uint64_t e0 = s_lh[H_END];
    const uint64_t n_end0 = s_lh[H_N_END];

    /* exclusive scan of the round totals: every thread takes a run of consecutive rounds, one
     * block scan over the runs; up to 1024 rounds stay in LDS for the passes below */
    const uint32_t per = (R + BD - 1) / BD;
    const uint32_t r_lo = min(R, tid * per), r_hi = min(R, r_lo + per);
    uint64_t run = 0;
    for (uint32_t r = r_lo; r < r_hi; r++) run += (r < 1024) ? q.bytes0[r] : rb[r];
    uint64_t vtot;
    {
        const uint64_t incl = block_incl_scan(run, s_tot, &vtot);
        uint64_t v = incl - run;
        for (uint32_t r = r_lo; r < r_hi; r++) {
            const uint64_t bytes = (r < 1024) ? q.bytes0[r] : rb[r];
            if (FX && write_rec) E.round_virt[r] = v;      /* k_append_push of a later launch reads it; k_call does not */
            if (R <= 1024) s_virt[r] = v;
            if (!FX && r == my_r) q.my_virt = v;
            /* the first round that does not fit before len (the totals are positive: exactly one) */
            if (e0 + v + bytes > L && e0 + v <= L) { s_rstar = r; q.virt_rstar = v; }
            v += bytes;
        }
    }
    if (tid == 0) { if (FX && write_rec) E.round_virt[R] = vtot; if (R <= 1024) s_virt[R] = vtot; }
    __syncthreads();
    const uint32_t rstar = s_rstar;
    if (blockIdx.x == 0) STAMP(0, 3);

    if (tid == 0) {
        int64_t kstar = -1, estar = -1;
        uint64_t w = 0;
        uint32_t stale = 0;
        if (rstar < R) {
            uint64_t a = e0 + q.virt_rstar;
            for (uint32_t g = rf[rstar]; g < rf[rstar + 1]; g++) {
                const uint64_t T = APUS_HDR + E.req_len[g];
                if (a + T > L) {
                    kstar = (int64_t)(g - g0);
                    w = a;
                    if (a == L) estar = kstar;              /* empty encoding: idx restarts, dare_log.h:486-488 */
                    else if (L - a >= APUS_HDR) stale = 1;  /* header fitted, payload did not, dare_log.h:521-537 */
                    break;
                }
                a += T;
            }
        }
        const uint64_t end_new = (kstar < 0) ? e0 + vtot : e0 + vtot - w;
        if (FX && kstar >= 0 && end_new > L) set_status(E, 1u << 0);      /* second wrap */
        /* free space: the reference only notices end == head exactly (dare_log.h:168) */
        {
            const uint64_t head = s_lh[H_HEAD];
            const uint64_t used = (e0 == L) ? 0 : (e0 >= head ? e0 - head : L - (head - e0));
            const uint64_t waste = (kstar >= 0) ? L - w : 0;
            if (FX && n && e0 != L && vtot + waste >= L - used) set_status(E, 1u << 1);
            if (FX && n && e0 == L && vtot > L) set_status(E, 1u << 1);
        }
        const uint64_t idx0 = s_lh[H_LAST_IDX] + 1;
        SeqOut s;
        s.e0 = e0; s.idx0 = idx0; s.w = w; s.n_end0 = n_end0;
        s.term = s_lh[H_SID] >> 9;
        s.kstar = kstar; s.estar = estar; s.stale = stale; s.n = n;
        s.head_round = head_round;
        s.first_fail = ~0ull;
        s.commit_before = s_lh[H_COMMIT];
        s.n_commit_before = s_lh[H_N_COMMIT];
        /* what the followers may see / the scan may commit (visible_slots): a batch that ends
         * exactly on len reads as empty, its last round stays hidden */
        {
            const uint64_t end_after = n ? end_new : e0, n_end_after = n_end0 + n;
            s.vis = (end_after != L) ? n_end_after
                  : (n == 0 ? s_lh[H_N_VISIBLE] : n_end0 + (rf[R - 1] - g0));
            const SeqFlags fl = seq_flags(E, push_mask, fuse_mask, in_step, end_after == L, n, s.n_commit_before, n_end0, head_round);
            const uint32_t fuse_batch = fl.fuse_batch;
            uint64_t lo = s.n_commit_before;
            for (uint32_t f = 0; f < APUS_DEV_MAX_SERVERS; f++) {
                uint64_t npf = s_fw[f][3];                              /* ~0 for servers not pushed to */
                if ((fuse_mask >> f) & 1u) npf = ((fuse_batch >> f) & 1u) ? ~0ull : npf + head_round;
                s.np[f] = npf;
                lo = min(lo, npf);
            }
            s.scan_lo = lo;
            s.fuse_mask = fuse_batch;
            s.tail_needed = fl.tail_needed;
            s.fast = fl.fast;
            s.pad1 = 0;
            s.rec_base = s_misc[0];
        }
        q.out = s;
        q.end_new = n ? end_new : e0;
        if (FX) *E.seq = s;
        if (n) {
            /* the LDS copy becomes the control block as it is after the batch; the sequencer
             * proper stores the words that changed */
            seq_apply_batch(s_lh, end_new, APUS_HDR + s_misc[1], n_end0, n, idx0, estar);
            if (FX) {
                hdr[H_END] = s_lh[H_END]; hdr[H_TAIL] = s_lh[H_TAIL]; hdr[H_N_END] = s_lh[H_N_END];
                hdr[H_LAST_IDX] = s_lh[H_LAST_IDX]; hdr[H_PREV_HEAD] = 0; hdr[H_OLD_END] = s_lh[H_OLD_END];
                hdr[H_N_PERSIST] = s_lh[H_N_PERSIST]; hdr[H_STORE_COUNT] = s_lh[H_STORE_COUNT];
            }
        }
        s_kstar = kstar; s_w = w;
    }
    __syncthreads();

    if (blockIdx.x == 0) STAMP(0, 4);
    if (!FX || !write_rec) return;
    /* end offset after every round (the leader's per-round record) */
    const uint64_t rec_base = s_misc[0];
    const int64_t kstar = s_kstar;
    const uint64_t w = s_w;
    for (uint32_t r = tid; r < R; r += BD) {
        const uint64_t a_end = e0 + (R <= 1024 ? s_virt[r + 1] : E.round_virt[r + 1]);
        const int64_t last = (int64_t)(rf[r + 1] - g0) - 1;   /* batch index of the round's last entry */
        const uint64_t end_r = (kstar < 0 || last < kstar) ? a_end : a_end - w;
        if (rec_base + head_round + r < E.rec_cap) E.rec_end[rec_base + head_round + r] = end_r;
    }
    if (blockIdx.x == 0) STAMP(0, 5);
}

__global__ __launch_bounds__(1024) void k_sequence(const EngDev E, uint64_t r0, uint32_t R, uint32_t push_mask,
                                                    uint32_t tick, uint32_t sample_mask)
{
    __shared__ SeqLds q;
    seq_stage(E, r0, R, push_mask, sample_mask, q);
    seq_body<true>(E, r0, R, push_mask, tick, sample_mask, q, 0);
}

/* ------------------------------------------------------------------------- */
/* k_append_push: one workgroup per round.  Wave 0: one lane per entry (offset
 * scan, index, header words, directory).  All 256 threads: the round's bytes as
 * 16-byte units, each stored to the leader ring and to every in-sync follower
 * ring at the same offset, so that the payload is read from HBM once.          */
/* k_call's append blocks, common case -- the batch stays clear of the end of the ring: wave 0
 * alone works out the SeqOut fields an append block needs (no block-wide scan: the host staged
 * the byte prefix of the rounds next to their totals), while the other waves fetch payload.
 * seq_w0_stage: the inputs, one round trip, by the 64 lanes.  seq_w0_decide: one lane, the same
 * decision code as the sequencer (seq_in_step, control_append<false>, seq_flags).  If the batch
 * could reach len (q.ok = 0) the block falls back to seq_body<false> on the same staged inputs. */
__device__ static inline void seq_w0_stage(const EngDev &E, uint64_t r0, uint32_t R, uint32_t push_mask, uint32_t my_r, SeqLds &q,
                                           const uint64_t *snap = nullptr, bool need_tail = false)
{
    const uint32_t lane = lane_id();
    /* q.fw[f][j], j = FW_END .. FW_N_APPLY: word k = 5 f + j by lane k (and word 64 by lane 0).  The
     * follower's control block is addressed through the pointer table in HBM-resident kernarg
     * memory, never through a private copy of E */
    const uint32_t kf = lane / 5, kj = lane - kf * 5;
    const uint64_t v0 = stage_leader_word(E, snap, lane);
    uint64_t f0 = kj == FW_N_PERSIST ? ~0ull : 0ull;
    if ((push_mask >> kf) & 1u) f0 = stage_follower_word(E, snap, kf, kj);
    uint64_t x = 0;
    if (lane == 0) { x = 0; if ((push_mask >> 12) & 1u) x = stage_follower_word(E, snap, 12, FW_N_APPLY); }
    else if (lane == 1) x = stage_rec_count(E, snap);
    else if (lane == 2) x = E.round_prefix[r0];
    else if (lane == 3) x = E.round_prefix[r0 + my_r];
    else if (lane == 4) x = E.round_prefix[r0 + R];
    else if (lane == 5) x = E.round_first[r0];
    else if (lane == 6) x = E.round_first[r0 + R];
    else if (lane == 7 && need_tail) {                              /* len of the batch's last request (for H_TAIL) */
        const uint32_t a = E.round_first[r0], b = E.round_first[r0 + R];
        x = (b > a) ? E.req_len[b - 1] : 0;
    }
    q.lh[lane] = v0;
    (&q.fw[0][0])[lane] = f0;
    if (lane == 0) { (&q.fw[0][0])[64] = x; q.rstar = 0xFFFFFFFFu; q.head_round = 0; }
    else if (lane == 1) q.misc[0] = x;
    else if (lane >= 2 && lane <= 4) q.pfx[lane - 2] = x;
    else if (lane == 5 || lane == 6) q.rfx[lane - 5] = (uint32_t)x;
    else if (lane == 7) q.misc[1] = x;
}

/* FX = true: the sequencer block's own use of it -- the same decisions, plus every effect
 * (the tick's <HEAD> entry, the sampled apply offsets, the control words, SeqOut); it declines
 * (q.ok = 0, nothing stored) whenever the block-wide sequencer is needed: a possible wrap, or a
 * pushed follower that lags. */
/* LEAN: only what an append / record block needs of SeqOut (no follower table, no post-call
 * control block): this lane's work sits on every append block's critical path */
template <bool FX, bool LEAN = false>
__device__ static inline void seq_w0_decide(const EngDev &E, uint32_t push_mask, uint32_t tick, SeqLds &q)
{
    const uint64_t L = E.log_len;
    const uint64_t vtot = q.pfx[2] - q.pfx[0];
    const uint64_t e_pre = q.lh[H_END];
    /* clear of len even if a <HEAD> entry (64 bytes) goes in front of the batch? */
    if (e_pre == L || e_pre + APUS_HDR + vtot >= L) { q.ok = 0; return; }
    if (FX) {
        const uint64_t n_pre = q.lh[H_N_END];
        for (uint32_t m = push_mask; m; m &= m - 1)
            if (q.fw[__builtin_ctz(m)][1] < n_pre) { q.ok = 0; return; }       /* catch-up first: the block-wide path */
    }
    uint64_t *hdr = E.rep[E.leader].hdr;
    uint32_t fuse_mask; bool in_step;
    seq_in_step(E, q.lh, q.fw, push_mask, fuse_mask, in_step);
    uint32_t head_round = 0;
    if (tick) {
        if (FX) for (uint32_t i = 0; i < APUS_DEV_MAX_SERVERS; i++) sample_apply_offsets(E, q.lh, push_mask, i, &q.fw[i][2]);
        head_round = control_append<FX>(E, 1, 3, 0, 0, push_mask, q.lh, q.misc[0], fuse_mask, in_step, false).n;
        if (!LEAN) for (uint32_t i = 0; i < APUS_DEV_MAX_SERVERS; i++) sample_into_copy(E, q.lh, push_mask, i, q.fw[i][2]);
    }
    const uint32_t n = q.rfx[1] - q.rfx[0];
    SeqOut &s = q.out;              /* built in LDS: 31 words of it in registers cost occupancy */
    s.e0 = q.lh[H_END]; s.idx0 = q.lh[H_LAST_IDX] + 1; s.w = 0; s.n_end0 = q.lh[H_N_END];
    s.term = q.lh[H_SID] >> 9; s.kstar = -1; s.estar = -1; s.stale = 0; s.n = n; s.head_round = head_round; s.pad0 = 0;
    s.first_fail = ~0ull; s.commit_before = q.lh[H_COMMIT]; s.n_commit_before = q.lh[H_N_COMMIT];
    const SeqFlags fl = seq_flags(E, push_mask, fuse_mask, in_step, false, n, s.n_commit_before, s.n_end0, head_round);
    s.vis = s.n_end0 + n;
    if (!LEAN) {
        uint64_t lo = s.n_commit_before;
        for (uint32_t f = 0; f < APUS_DEV_MAX_SERVERS; f++) {
            uint64_t npf = q.fw[f][3];                                  /* ~0 for servers not pushed to */
            if ((fuse_mask >> f) & 1u) npf = ((fl.fuse_batch >> f) & 1u) ? ~0ull : npf + head_round;
            s.np[f] = npf;
            lo = min(lo, npf);
        }
        s.scan_lo = lo;
    }
    s.fuse_mask = fl.fuse_batch; s.tail_needed = fl.tail_needed; s.fast = fl.fast; s.pad1 = 0; s.rec_base = q.misc[0];
    q.end_new = n ? s.e0 + vtot : s.e0;
    if (FX) {
        /* free space: the reference only notices end == head exactly (dare_log.h:168) */
        const uint64_t head = q.lh[H_HEAD], e0 = s.e0;
        const uint64_t used = e0 >= head ? e0 - head : L - (head - e0);
        if (n && vtot >= L - used) set_status(E, 1u << 1);
        /* (the caller copies q.out to E.seq with a wave) */
    }
    /* q.lh becomes the control block as the sequencer leaves it (H_TAIL is only right when the
     * length of the batch's last request was staged: the bookkeeper and the sequencer do) */
    if (n && !LEAN) {
        seq_apply_batch(q.lh, q.end_new, APUS_HDR + q.misc[1], s.n_end0, n, s.idx0, -1);
        if (FX) {
            hdr[H_END] = q.lh[H_END]; hdr[H_TAIL] = q.lh[H_TAIL]; hdr[H_N_END] = q.lh[H_N_END];
            hdr[H_LAST_IDX] = q.lh[H_LAST_IDX]; hdr[H_PREV_HEAD] = 0; hdr[H_OLD_END] = q.lh[H_OLD_END];
            hdr[H_N_PERSIST] = q.lh[H_N_PERSIST]; hdr[H_STORE_COUNT] = q.lh[H_STORE_COUNT];
        }
    }
    q.my_virt = q.pfx[1] - q.pfx[0];
    q.ok = 1;
}

/* The chain block's sequencing of a segment that is IN STEP and clear of the end of the ring,
 * by one wavefront instead of one lane: lane i holds server i's words, the leader's words are
 * fetched from LDS in one batch, the per-server loops of seq_in_step / control_append (mode 1:
 * log_pruning, dare_server.c:1996-2067) / sample_into_copy become wave operations.  Produces
 * exactly what seq_w0_decide<false> leaves in q for such a segment (q.lh = the leader's control
 * block after the call, q.out, q.end_new, q.ok = 1) and returns 1; returns 0 with q untouched when
 * any precondition fails -- then the general single-lane path decides.  The single-lane path costs
 * ~2.6 us of dependent LDS round trips per segment on the launch's critical chain. */
/* The same sequencing on REGISTERS: the chain block's first wavefront runs it for every segment of
 * a launch before anything else (the record pass), so it must not queue behind the append blocks
 * that share its CU -- no LDS, no memory access except the staged byte prefix of a batch that wraps.
 * Scalars are wave-uniform; apoff / f_* are per lane (lane = server). */
/* Admission of a segment of a multi-segment launch, on the device (between two prune ticks of one
 * launch the host cannot know where head is).  log_append_entry refuses a request when the log is
 * full (end == head, dare_log.h:168,492-495) and the reference runs over un-pruned entries when a
 * request merely crosses head; here a segment whose bytes (+ a due tick's <HEAD> entry + the bytes
 * a wrap may skip, at most one entry: max_T) do not fit into the free part of the ring is refused
 * AS A WHOLE: its record says so, nobody stores anything, the state stays as it is, status bit
 * LOG_FULL tells the host.  (Calls outside a batch and the live path are admitted on the host,
 * apus_engine.hip:admit_bytes.) */
__device__ static inline bool segment_refused(uint64_t L, uint64_t end, uint64_t head, uint64_t vtot, uint32_t tick, uint32_t max_T)
{
    if (end == L) return vtot + (tick ? APUS_HDR : 0) > L;             /* reads as empty */
    if (end == head) return true;                                      /* log_is_full */
    const uint64_t used = end > head ? end - head : L - (head - end);
    return vtot + (tick ? APUS_HDR : 0) + max_T > L - used;
}

struct ChainRegs {
    uint64_t end, n_end, last_idx, sid, commit, n_commit, n_apply, apply, head, tail, prev_head, store_count, rec_base;
    uint32_t bitmask;
    uint64_t apoff, f_apply, f_np, f_na;         /* per lane */
};
struct ChainSeg { uint64_t r0, pfx0, pfx2, len_last; uint32_t rf0, rf1, R, tick; };
struct ChainOut {
    uint64_t e0, idx0, n_end0, w, end_new, hd, sc, my_apoff /* per lane */;
    int64_t kstar, estar;
    uint32_t stale, head_round, n;
};
__device__ static inline uint64_t rl64(uint64_t v, uint32_t l)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l) << 32);
}
__device__ static inline int chain_core(const EngDev &E, uint32_t push_mask, const ChainRegs &c, const ChainSeg &g, ChainOut &o)
{
    const uint32_t lane = lane_id();
    const uint64_t L = E.log_len;
    const uint32_t size = E.group_size, leader = E.leader;
    const bool srv = lane < APUS_DEV_MAX_SERVERS;
    const uint64_t end = c.end, n_end = c.n_end;
    const uint64_t vtot = g.pfx2 - g.pfx0;
    const uint32_t n = g.rf1 - g.rf0, R = g.R;

    /* in step (seq_in_step), a fused majority (seq_flags), clear of len, not full */
    const bool pushed = srv && ((push_mask >> lane) & 1u);
    const uint32_t size_mask = (1u << size) - 1;
    const bool quorum = (uint32_t)__popc((push_mask | (1u << leader)) & size_mask) >= size / 2 + 1;
    /* (a batch that may reach the end of the ring is placed below; one that is longer than the
     * ring, or an empty log, goes the general way) */
    const bool pre = !(E.flags & 1u) && end != L && L - end >= APUS_HDR && vtot + APUS_HDR < L && n > 0 && R <= 1024 && quorum &&
                     c.n_commit == n_end && c.n_apply == n_end && end != c.head;
    if (!pre || !__all(!pushed || (c.f_np == n_end && c.f_na == n_end))) return 0;

    uint64_t e = end, ne = n_end, li = c.last_idx, hd = c.head, tl = c.tail, sc = c.store_count;
    uint32_t head_round = 0;
    uint64_t my_apoff = c.apoff;
    if (g.tick) {
        /* log_pruning: the offset that lags most (largest distance to end); a server that is OFF counts as apply */
        const bool in_grp = lane < size;
        const bool on = in_grp && ((c.bitmask >> lane) & 1u);
        if (in_grp && !on) my_apoff = c.apply;
        const uint64_t dist = in_grp ? apus_end_distance(e, L, my_apoff) : 0;
        uint64_t D = apus_end_distance(e, L, c.apply);
        for (uint32_t i = 0; i < size; i++) { const uint64_t di = rl64(dist, i); if (di > D) D = di; }
        uint64_t min_off = (D <= e) ? e - D : e + L - D;
        if (D == 0) min_off = tl;                                        /* leave one entry, :2038-2041 */
        if (apus_is_larger(e, L, min_off, hd) && !c.prev_head) {
            hd = min_off;                                                /* <HEAD, head>: 64 bytes at end (no wrap: pre) */
            tl = e; e += APUS_HDR; ne += 1; li += 1; sc += 1; head_round = 1;
        }
        /* rc_get_remote_apply_offsets for the next tick (sample_into_copy) */
        if (in_grp) { if (lane == leader || !on) my_apoff = c.apply; else if (pushed) my_apoff = c.f_apply; }
    }
    if (e == L) return 0;            /* the <HEAD> entry ended exactly on len: the log reads as empty, general path */
    const uint64_t e0 = e, idx0 = li + 1, n_end0 = ne;
    /* where the batch wraps, if it does (log_append_entry's two wrap rules, dare_log.h:502-538; what
     * seq_body's block-wide scan finds): the round whose bytes cross len, by a 64-ary search of the
     * staged byte prefix, then the entry inside it */
    int64_t kstar = -1, estar = -1;
    uint64_t w = 0;
    uint32_t stale = 0;
    if (e0 + vtot >= L) {
        if (e0 + vtot == L) return 0;                                    /* ends exactly on len: hidden round, general path */
        const uint64_t *pf = E.round_prefix + g.r0;
        const uint64_t p0 = g.pfx0;
        const uint32_t step = (R + WAVE - 1) / WAVE;                     /* <= 16 */
        const uint32_t ra = min(R, lane * step);
        const uint64_t va = gld(&pf[ra]) - p0;
        const uint32_t blk = (uint32_t)__popcll(__ballot(ra < R && e0 + va <= L)) - 1u;   /* monotone: a prefix of lanes */
        const uint32_t rb = blk * step + lane;
        const bool inb = lane <= step && rb <= R;
        const uint64_t vb = inb ? gld(&pf[rb]) - p0 : ~0ull;
        const uint32_t cnt = (uint32_t)__popcll(__ballot(inb && rb < R && e0 + vb <= L));
        const uint32_t rstar = blk * step + cnt - 1u;                    /* last round that starts at or before len */
        const uint64_t vstar = rl64(vb, cnt - 1u);
        const uint32_t g_lo = gld(&E.round_first[g.r0 + rstar]), g_hi = gld(&E.round_first[g.r0 + rstar + 1]);
        const uint32_t g0 = g.rf0;
        const bool act = lane < g_hi - g_lo;
        const uint64_t Te = act ? APUS_HDR + (uint64_t)gld(&E.req_len[g_lo + lane]) : 0;
        const uint64_t a = e0 + vstar + wave_incl_scan(Te) - Te;
        const unsigned long long hit = __ballot(act && a + Te > L);
        if (!hit) return 0;                                              /* (cannot happen: the totals say it wraps) */
        const int hl = __builtin_ctzll(hit);
        const uint64_t aw = rl64(a, (uint32_t)hl);
        kstar = (int64_t)(g_lo - g0) + hl; w = aw;
        if (aw == L) estar = kstar;                                      /* empty encoding: idx restarts, dare_log.h:486-488 */
        else if (L - aw >= APUS_HDR) stale = 1;                          /* header fitted, payload did not, :521-537 */
        if (e0 + vtot - w >= L) return 0;                                /* a second wrap, or the batch ends on len: general path */
    }
    o.e0 = e0; o.idx0 = idx0; o.n_end0 = n_end0; o.w = w; o.kstar = kstar; o.estar = estar; o.stale = stale;
    o.end_new = (kstar < 0) ? e0 + vtot : e0 + vtot - w;
    o.hd = hd; o.sc = sc; o.my_apoff = my_apoff; o.head_round = head_round; o.n = n;
    return 1;
}
/* the state after a segment that chain_core sequenced: the control words as the sequencer leaves
 * them, then what keeper_publish does to them in step (everything visible is committed and applied
 * on every pushed replica) */
__device__ static inline void chain_advance(const EngDev &E, uint32_t push_mask, ChainRegs &c, const ChainSeg &g, const ChainOut &o)
{
    const uint32_t lane = lane_id();
    const uint64_t vis = o.n_end0 + o.n;
    c.end = o.end_new; c.tail = o.end_new - (APUS_HDR + g.len_last); c.n_end = vis;
    c.last_idx = (o.estar < 0) ? o.idx0 + o.n - 1 : 1 + (uint64_t)(o.n - 1 - o.estar);
    c.prev_head = 0; c.store_count = o.sc + o.n; c.head = o.hd;
    if (g.tick && lane < E.group_size) c.apoff = o.my_apoff;
    c.commit = o.end_new; c.n_commit = vis; c.apply = o.end_new; c.n_apply = vis;
    if (lane < APUS_DEV_MAX_SERVERS && ((push_mask >> lane) & 1u)) { c.f_np = vis; c.f_na = vis; c.f_apply = o.end_new; }
    c.rec_base += g.R + o.head_round;
}
/* LDS form (the chain block's second pass, and the single source of the logic above) */
__device__ static inline int chain_decide_fast(const EngDev &E, uint32_t push_mask, uint32_t tick, SeqLds &q, uint64_t r0, uint32_t R)
{
    const uint32_t lane = lane_id();
    const bool srv = lane < APUS_DEV_MAX_SERVERS;
    ChainRegs c;
    /* one batch of LDS reads */
    c.end = q.lh[H_END]; c.n_end = q.lh[H_N_END]; c.last_idx = q.lh[H_LAST_IDX]; c.sid = q.lh[H_SID];
    c.commit = q.lh[H_COMMIT]; c.n_commit = q.lh[H_N_COMMIT]; c.n_apply = q.lh[H_N_APPLY]; c.apply = q.lh[H_APPLY];
    c.head = q.lh[H_HEAD]; c.tail = q.lh[H_TAIL]; c.prev_head = q.lh[H_PREV_HEAD]; c.store_count = q.lh[H_STORE_COUNT];
    c.bitmask = (uint32_t)q.lh[H_CID_BITMASK];
    c.apoff = srv ? q.lh[H_APPLY_OFFSETS + lane] : 0;
    c.f_apply = srv ? q.fw[lane][2] : 0; c.f_np = srv ? q.fw[lane][3] : 0; c.f_na = srv ? q.fw[lane][4] : 0;
    c.rec_base = q.misc[0];
    ChainSeg g;
    g.r0 = r0; g.pfx0 = q.pfx[0]; g.pfx2 = q.pfx[2]; g.len_last = q.misc[1]; g.rf0 = q.rfx[0]; g.rf1 = q.rfx[1]; g.R = R; g.tick = tick;
    ChainOut o;
    if (!chain_core(E, push_mask, c, g, o)) return 0;
    if (lane == 0) {
        q.lh[H_END] = o.end_new; q.lh[H_TAIL] = o.end_new - (APUS_HDR + g.len_last); q.lh[H_N_END] = o.n_end0 + o.n;
        q.lh[H_LAST_IDX] = (o.estar < 0) ? o.idx0 + o.n - 1 : 1 + (uint64_t)(o.n - 1 - o.estar); q.lh[H_PREV_HEAD] = 0; q.lh[H_OLD_END] = o.end_new;
        q.lh[H_N_PERSIST] = o.n_end0 + o.n; q.lh[H_STORE_COUNT] = o.sc + o.n; q.lh[H_HEAD] = o.hd;
        SeqOut &s = q.out;
        s.e0 = o.e0; s.idx0 = o.idx0; s.w = o.w; s.n_end0 = o.n_end0; s.term = c.sid >> 9; s.kstar = o.kstar; s.estar = o.estar; s.stale = o.stale;
        s.n = o.n; s.head_round = o.head_round; s.pad0 = 0; s.first_fail = ~0ull; s.commit_before = c.commit; s.n_commit_before = c.n_commit;
        s.vis = o.n_end0 + o.n; s.scan_lo = c.n_commit; s.fuse_mask = push_mask; s.tail_needed = 0; s.fast = 1; s.pad1 = 0; s.rec_base = c.rec_base;
        q.end_new = o.end_new; q.my_virt = q.pfx[1] - q.pfx[0]; q.ok = 1;
    }
    if (srv) { q.out.np[lane] = ~0ull; if (tick && lane < E.group_size) q.lh[H_APPLY_OFFSETS + lane] = o.my_apoff; }
    return 1;
}

/* The effects of a segment that chain_decide_fast sequenced -- what the sequencer block does
 * with FX = true (seq_w0_decide<true>: the leader's live control words, the sampled apply
 * offsets, a due prune tick's <HEAD> entry pushed to every replica, SeqOut) -- by the chain
 * block's first wavefront, from the post-call state in q: the sequencers of in-step segments
 * have nothing left to do (they used to form a second serial chain through memory, ~12 us per
 * segment, that the launch could not end before). */
__device__ static inline void chain_effects_fast(const EngDev &E, uint32_t push_mask, uint32_t tick, const SeqLds &q)
{
    const uint32_t lane = lane_id();
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *hdr = Ld.hdr;
    const uint64_t L = E.log_len;
    const SeqOut &s = q.out;
    if (lane == 0) {
        gst(&hdr[H_END], (uint64_t)(q.lh[H_END])); gst(&hdr[H_TAIL], (uint64_t)(q.lh[H_TAIL])); gst(&hdr[H_N_END], (uint64_t)(q.lh[H_N_END])); gst(&hdr[H_LAST_IDX], (uint64_t)(q.lh[H_LAST_IDX]));
        gst(&hdr[H_PREV_HEAD], (uint64_t)(0)); gst(&hdr[H_OLD_END], (uint64_t)(q.lh[H_OLD_END])); gst(&hdr[H_N_PERSIST], (uint64_t)(q.lh[H_N_PERSIST]));
        gst(&hdr[H_STORE_COUNT], (uint64_t)(q.lh[H_STORE_COUNT]));
        /* free space: the reference only notices end == head exactly (dare_log.h:168) */
        const uint64_t head = q.lh[H_HEAD], e0 = s.e0;
        const uint64_t used = e0 >= head ? e0 - head : L - (head - e0);
        const uint64_t vt = (s.kstar >= 0) ? q.end_new + s.w - e0 : q.end_new - e0, waste = (s.kstar >= 0) ? L - s.w : 0;
        if (s.n && vt + waste >= L - used) set_status(E, 1u << 1);
    }
    if (tick && lane < E.group_size) gst(&hdr[H_APPLY_OFFSETS + lane], (uint64_t)(q.lh[H_APPLY_OFFSETS + lane]));
    if (s.head_round) {
        /* <HEAD, head> right in front of the batch (log_append_entry of a 64-byte entry; everybody is in
         * step: pushed with its reply bytes, committed and applied as it lands) */
        const uint64_t pos = s.e0 - APUS_HDR, idx = s.idx0 - 1, slot = s.n_end0 - 1, hv = q.lh[H_HEAD], term = s.term;
        const uint32_t di = (uint32_t)slot & E.dir_mask, fuse = s.fuse_mask;
        if (lane == 0) { gst(&hdr[H_HEAD], (uint64_t)(hv)); if (s.rec_base < E.rec_cap) gst(&E.rec_end[s.rec_base], (uint64_t)s.e0); }
        if (lane < APUS_DEV_MAX_SERVERS && (((push_mask | (1u << E.leader)) >> lane) & 1u)) {
            const RepDev &Rd = E.rep[lane];
            const ReplyWords rw = apus_reply_words(lane == E.leader ? fuse : (fuse & (1u << lane)));
            uint8_t *rg = Rd.ring;
            st16u(rg + pos, make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)term, (uint32_t)(term >> 32)));
            st16u(rg + pos + 16, make_uint4(0, 0, (3u << 16) | ((uint32_t)E.leader << 24), rw.w28));
            st16u(rg + pos + 32, make_uint4(rw.x32, rw.y36, rw.z40, 0));
            st16u(rg + pos + 48, make_uint4((uint32_t)hv, (uint32_t)(hv >> 32), 0, 0));
            gst(&Rd.dir_off[di], pos); gst(&Rd.dir_len[di], (uint32_t)(APUS_HDR | ((uint32_t)E.leader << 24)));
            uint8_t *rp = (uint8_t *)&Rd.apply[di];
            st16u(rp, make_uint4((uint32_t)slot, (uint32_t)(slot >> 32), (uint32_t)pos, (uint32_t)(pos >> 32)));
            st16u(rp + 16, make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), 0, 3u << 16));
            if (lane == E.leader) __hip_atomic_store(&Ld.ack[di], fuse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lane < sizeof(SeqOut) / 8) gst(&((uint64_t *)E.seq)[lane], ((const uint64_t *)&q.out)[lane]);
}

/* The books of a launch whose segments were ALL sequenced by the record pass (in step, chain_core):
 * one wavefront walks the segments once more on registers and does, per segment, only what is not
 * overwritten by the next one -- a due prune tick's <HEAD> entry on every replica (chain_effects_fast),
 * its per-round record, the log-full check, the segment's sign-off -- and then stores the control
 * words of the leader and of every pushed follower ONCE, as the last segment leaves them
 * (chain_effects_fast + keeper_publish of every segment in turn write the same words; nobody can
 * observe the intermediate values: no block of the launch reads them, the host sees the launch's
 * end).  Per-segment stores from this block queue behind the append blocks that share its CU
 * (~5.5 us per segment, the launch could not end before 7 x that; profiles/README.md).
 * cr = the state before the launch; fw_sc / fw_head / fw_nc: the followers' store count, head and
 * commit slot (lane = server). */
template <typename SEG>      /* CallArgs (defined below) */
__device__ static inline void chain_books_fast(const EngDev &E, uint32_t push_mask, ChainRegs cr, uint64_t fw_sc, uint64_t fw_head,
                                               uint32_t S, const uint64_t (*pre_pfx)[2], const uint32_t (*pre_rf)[2], const uint64_t *pre_last,
                                               const SEG *segs, SeqOut &last_out, uint32_t max_T)
{
    const uint32_t lane = lane_id();
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *hdr = Ld.hdr;
    const uint64_t L = E.log_len;
    const bool srv = lane < APUS_DEV_MAX_SERVERS;
    const bool pushed = srv && ((push_mask >> lane) & 1u);
    const bool target = srv && (((push_mask | (1u << E.leader)) >> lane) & 1u);
    const uint64_t n_end_before = cr.n_end;
    uint64_t my_pfx0 = 0, my_pfx2 = 0, my_last = 0, my_r0 = 0;
    uint32_t my_rf0 = 0, my_rf1 = 0, my_R = 0, my_tick = 0;
    if (lane < S) {
        my_pfx0 = pre_pfx[lane][0]; my_pfx2 = pre_pfx[lane][1]; my_last = pre_last[lane];
        my_rf0 = pre_rf[lane][0]; my_rf1 = pre_rf[lane][1];
        my_r0 = segs[lane].r0; my_R = segs[lane].R; my_tick = segs[lane].tick;
    }
    uint64_t n_total = 0;
    for (uint32_t k = 0; k < S; k++) {
        ChainSeg g;
        g.r0 = rl64(my_r0, k); g.pfx0 = rl64(my_pfx0, k); g.pfx2 = rl64(my_pfx2, k); g.len_last = rl64(my_last, k);
        g.rf0 = (uint32_t)__builtin_amdgcn_readlane((int)my_rf0, (int)k); g.rf1 = (uint32_t)__builtin_amdgcn_readlane((int)my_rf1, (int)k);
        g.R = (uint32_t)__builtin_amdgcn_readlane((int)my_R, (int)k); g.tick = (uint32_t)__builtin_amdgcn_readlane((int)my_tick, (int)k);
        ChainOut o;
        if (segment_refused(L, cr.end, cr.head, g.pfx2 - g.pfx0, g.tick, max_T)) {
            /* refused by the record pass: nothing was stored, nothing changes; its blocks sign off on their own */
            if (lane == 0) {
                set_status(E, 1u << 1);
                __hip_atomic_fetch_add(E.step_tickets + (size_t)k * 32 + T_PASS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (k + 1 == S && lane == 0) {
                SeqOut &s = last_out;
                s.e0 = cr.end; s.idx0 = cr.last_idx + 1; s.w = 0; s.n_end0 = cr.n_end; s.term = cr.sid >> 9; s.kstar = -1; s.estar = -1; s.stale = 0;
                s.n = 0; s.head_round = 0; s.pad0 = 0; s.first_fail = ~0ull; s.commit_before = cr.commit; s.n_commit_before = cr.n_commit;
                s.vis = cr.n_end; s.scan_lo = cr.n_commit; s.fuse_mask = push_mask; s.tail_needed = 0; s.fast = 1; s.pad1 = 0; s.rec_base = cr.rec_base;
            }
            if (k + 1 == S && srv) last_out.np[lane] = ~0ull;
            continue;
        }
        if (!chain_core(E, push_mask, cr, g, o)) { if (lane == 0) set_status(E, 1u << 4); break; }      /* (the record pass took it: cannot happen) */
        const uint64_t term = cr.sid >> 9;
        if (lane == 0) {
            /* free space: the reference only notices end == head exactly (dare_log.h:168) */
            const uint64_t used = o.e0 >= o.hd ? o.e0 - o.hd : L - (o.hd - o.e0);
            const uint64_t vt = (o.kstar >= 0) ? o.end_new + o.w - o.e0 : o.end_new - o.e0, waste = (o.kstar >= 0) ? L - o.w : 0;
            if (o.n && vt + waste >= L - used) set_status(E, 1u << 1);
        }
        if (o.head_round) {
            /* <HEAD, head> right in front of the batch, pushed with its reply bytes, committed and applied as it lands */
            const uint64_t pos = o.e0 - APUS_HDR, idx = o.idx0 - 1, slot = o.n_end0 - 1, hv = o.hd;
            const uint32_t di = (uint32_t)slot & E.dir_mask, fuse = push_mask;
            if (lane == 0 && cr.rec_base < E.rec_cap) gst(&E.rec_end[cr.rec_base], (uint64_t)o.e0);
            if (target) {
                const RepDev &Rd = E.rep[lane];
                const ReplyWords rw = apus_reply_words(lane == E.leader ? fuse : (fuse & (1u << lane)));
                uint8_t *rg = Rd.ring;
                st16u(rg + pos, make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)term, (uint32_t)(term >> 32)));
                st16u(rg + pos + 16, make_uint4(0, 0, (3u << 16) | ((uint32_t)E.leader << 24), rw.w28));
                st16u(rg + pos + 32, make_uint4(rw.x32, rw.y36, rw.z40, 0));
                st16u(rg + pos + 48, make_uint4((uint32_t)hv, (uint32_t)(hv >> 32), 0, 0));
                gst(&Rd.dir_off[di], pos); gst(&Rd.dir_len[di], (uint32_t)(APUS_HDR | ((uint32_t)E.leader << 24)));
                uint8_t *rp = (uint8_t *)&Rd.apply[di];
                st16u(rp, make_uint4((uint32_t)slot, (uint32_t)(slot >> 32), (uint32_t)pos, (uint32_t)(pos >> 32)));
                st16u(rp + 16, make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), 0, 3u << 16));
                if (lane == E.leader) __hip_atomic_store(&Ld.ack[di], fuse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            /* a follower adopts a committed <HEAD> (keeper_publish, pure_head) */
            if (pushed && apus_is_larger(o.end_new, L, hv, fw_head)) fw_head = hv;
        }
        if (k + 1 == S && lane == 0) {
            SeqOut &s = last_out;
            s.e0 = o.e0; s.idx0 = o.idx0; s.w = o.w; s.n_end0 = o.n_end0; s.term = term; s.kstar = o.kstar; s.estar = o.estar; s.stale = o.stale;
            s.n = o.n; s.head_round = o.head_round; s.pad0 = 0; s.first_fail = ~0ull; s.commit_before = cr.commit; s.n_commit_before = cr.n_commit;
            s.vis = o.n_end0 + o.n; s.scan_lo = cr.n_commit; s.fuse_mask = push_mask; s.tail_needed = 0; s.fast = 1; s.pad1 = 0; s.rec_base = cr.rec_base;
        }
        if (k + 1 == S && srv) last_out.np[lane] = ~0ull;
        n_total += o.n;
        chain_advance(E, push_mask, cr, g, o);
        /* the segment's sign-off (its janitor waits for it) */
        if (lane == 0) __hip_atomic_fetch_add(E.step_tickets + (size_t)k * 32 + T_PASS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    /* ---- the control words, once ---- */
    const uint64_t vis = cr.n_end;
    if (lane == 0) {
        gst(&hdr[H_END], cr.end); gst(&hdr[H_TAIL], cr.tail); gst(&hdr[H_N_END], cr.n_end); gst(&hdr[H_LAST_IDX], cr.last_idx);
        gst(&hdr[H_PREV_HEAD], (uint64_t)0); gst(&hdr[H_OLD_END], cr.end); gst(&hdr[H_N_PERSIST], cr.n_end); gst(&hdr[H_STORE_COUNT], cr.store_count);
        gst(&hdr[H_HEAD], cr.head);
        gst(&hdr[H_N_VISIBLE], vis); gst(&hdr[H_COMMIT], cr.commit); gst(&hdr[H_N_COMMIT], cr.n_commit);
        gst(&hdr[H_APPLY], cr.apply); gst(&hdr[H_N_APPLY], cr.n_apply);
        gst(E.rec_count, (uint64_t)cr.rec_base);
        atomicAdd((unsigned long long *)&hdr[H_APPLY_COUNT], (unsigned long long)n_total);
        atomicAdd((unsigned long long *)&hdr[H_HIGHEST_REC], (unsigned long long)n_total);
    }
    if (lane < E.group_size) gst(&hdr[H_APPLY_OFFSETS + lane], cr.apoff);
    if (pushed) {
        uint64_t *fh = E.rep[lane].hdr;
        gst(&fh[H_STORE_COUNT], fw_sc + (vis - n_end_before));
        gst(&fh[H_END], cr.end); gst(&fh[H_OLD_END], cr.end); gst(&fh[H_N_END], vis); gst(&fh[H_N_PERSIST], vis);
        gst(&fh[H_COMMIT], cr.end); gst(&fh[H_N_COMMIT], vis); gst(&fh[H_APPLY], cr.end); gst(&fh[H_N_APPLY], vis);
        gst(&fh[H_HEAD], fw_head);
        atomicAdd((unsigned long long *)&fh[H_APPLY_COUNT], (unsigned long long)n_total);
    }
}

/* a block of k_call works the call's SeqOut out for itself: wave 0's variant, or the block-wide
 * scan when the batch could reach len; posts its "inputs fetched" ticket on tick line read_line */
__device__ static inline void seq_local(const EngDev &E, const CallEnv &X, uint64_t r0, uint32_t R, uint32_t push_mask, uint32_t tick,
                                        uint32_t my_r, uint32_t read_line, SeqLds &q, const uint64_t *snap = nullptr,
                                        bool need_tail = false, bool post_read = true)
{
    const uint32_t tid = threadIdx.x;
    if (tid < WAVE) {
        seq_w0_stage(E, r0, R, push_mask, my_r, q, snap, need_tail);
        if (tid == 0) {
            if (post_read) __hip_atomic_fetch_add(X.lines + (read_line & 31u) * 32 + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seq_w0_decide<false>(E, push_mask, tick, q);
        }
    }
    __syncthreads();
    if (!q.ok) {
        const uint32_t *rb = E.round_bytes + r0;
        for (uint32_t i = tid; i < R && i < 1024; i += blockDim.x) q.bytes0[i] = rb[i];
        __syncthreads();
        seq_body<false>(E, r0, R, push_mask, tick, push_mask, q, my_r);
    }
}

/* ------------------------------------------------------------------------- */
/* Multi-segment launches: the sequencing RECORD of a segment.  The chain block (segment 0's
 * bookkeeper, which keeps every segment's books) works the call's SeqOut out once and publishes
 * what an append block needs of it as ten self-tagged 8-byte granules {data, tag} in uncached
 * memory -- no flag, no fence, no drain on the writer's side (MI355X_MICROARCH.md, hand-off by
 * data-tagged granules); the append blocks poll the granules instead of staging 100+ control
 * words and repeating the single-lane sequencing themselves.  ok = 0 (the batch could reach the
 * end of the ring): the block falls back to the block-wide scan on snapshot `seg`.
 * The records are cleared by the launch's last janitor, so a stale record never matches. */
__device__ static inline void wait_count(const EngDev &E, const uint32_t *lines32, uint32_t b, uint32_t want);
#define REC_WORDS   32
#ifndef APUS_REC_SLEEP
#define APUS_REC_SLEEP 2        /* x 64 clocks between two polls of a segment's record */
#endif
#define REC_GRAN    20
#define REC_TAG(seg) (0x5E000000u | ((seg) + 1u))
enum { RECF_REFUSED = 1u << 19,      /* the segment does not fit into the free part of the ring: nobody stores anything */
       RECF_FAST = 1u << 13, RECF_OK = 1u << 14, RECF_HEAD = 1u << 15, RECF_CHAIN = 1u << 16, RECF_STALE = 1u << 17, RECF_ESTAR = 1u << 18 };   /* CHAIN: the chain block does the sequencer's effects */

__device__ static inline void rec_store(uint64_t *rec, uint32_t g, uint32_t data, uint32_t tag)
{
    __hip_atomic_store((APUS_GLOBAL unsigned long long *)(uintptr_t)&rec[g], ((unsigned long long)tag << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* the fields of a record, 8 bytes each = two granules: e0, idx0, n_end0, term, flags | n << 32, kstar, w,
 * rec_base, commit_before, n_commit_before */
struct RecFields { uint64_t e0, idx0, n_end0, term, flags_n, kstar, w, rec_base, commit_before, n_commit_before; };
__device__ static inline uint32_t rec_flags(uint32_t ok, uint32_t fuse_mask, uint32_t fast, uint32_t head_round, uint32_t stale, bool estar, bool chain_did)
{
    return (ok ? (fuse_mask & 0x1FFFu) | (fast ? RECF_FAST : 0u) | (head_round ? RECF_HEAD : 0u) | RECF_OK |
                 (stale ? RECF_STALE : 0u) | (estar ? RECF_ESTAR : 0u) : 0u) | (chain_did ? RECF_CHAIN : 0u);
}
__device__ static inline void rec_publish_fields(const EngDev &E, uint32_t seg, const RecFields &f)
{
    const uint32_t lane = lane_id();
    if (lane >= REC_GRAN) return;
    uint64_t v = 0;
    switch (lane >> 1) {
    case 0: v = f.e0; break;
    case 1: v = f.idx0; break;
    case 2: v = f.n_end0; break;
    case 3: v = f.term; break;
    case 4: v = f.flags_n; break;
    case 5: v = f.kstar; break;                    /* -1: the batch does not wrap */
    case 6: v = f.w; break;
    case 7: v = f.rec_base; break;
    case 8: v = f.commit_before; break;
    default: v = f.n_commit_before; break;
    }
    rec_store(E.step_rec + (size_t)seg * REC_WORDS, lane, (uint32_t)(lane & 1 ? v >> 32 : v), REC_TAG(seg));
}
/* lanes 0 .. REC_GRAN-1 of the calling wave publish segment seg's record from q.out / q.ok */
__device__ static inline void rec_publish(const EngDev &E, uint32_t seg, const SeqLds &q, bool chain_did)
{
    const SeqOut &s = q.out;
    const uint32_t ok = q.ok ? 1u : 0u;
    RecFields f;
    f.e0 = s.e0; f.idx0 = s.idx0; f.n_end0 = s.n_end0; f.term = s.term;
    f.flags_n = (uint64_t)rec_flags(ok, s.fuse_mask, s.fast, s.head_round, s.stale, s.estar >= 0, chain_did) | ((uint64_t)(ok ? s.n : 0u) << 32);
    f.kstar = (uint64_t)s.kstar; f.w = s.w; f.rec_base = s.rec_base; f.commit_before = s.commit_before; f.n_commit_before = s.n_commit_before;
    rec_publish_fields(E, seg, f);
}
/* wave-wide: poll segment seg's record, fill the SeqOut fields an append block uses (q.out, q.ok);
 * bounded like every spin of the engine */
__device__ static inline void rec_wait(const EngDev &E, uint32_t seg, SeqLds &q)
{
    const uint32_t lane = lane_id();
    const uint64_t *rec = E.step_rec + (size_t)seg * REC_WORDS;
    const uint32_t tag = REC_TAG(seg);
    uint64_t v = 0;
    unsigned long long spins = 0;
    for (;;) {
        if (lane < REC_GRAN) v = ldw(&rec[lane]);
        if (__all(lane >= REC_GRAN || (uint32_t)(v >> 32) == tag)) break;
        __builtin_amdgcn_s_sleep(APUS_REC_SLEEP);
        if (++spins > (1ull << 22)) { if (lane == 0) spin_timeout(E, 896); break; }      /* bounded */
    }
    const uint32_t d = (uint32_t)v;
    const uint64_t e0 = (uint64_t)__shfl(d, 0, WAVE) | ((uint64_t)__shfl(d, 1, WAVE) << 32);
    const uint64_t idx0 = (uint64_t)__shfl(d, 2, WAVE) | ((uint64_t)__shfl(d, 3, WAVE) << 32);
    const uint64_t n_end0 = (uint64_t)__shfl(d, 4, WAVE) | ((uint64_t)__shfl(d, 5, WAVE) << 32);
    const uint64_t term = (uint64_t)__shfl(d, 6, WAVE) | ((uint64_t)__shfl(d, 7, WAVE) << 32);
    const uint32_t flags = __shfl(d, 8, WAVE), n = __shfl(d, 9, WAVE);
    const uint64_t kst = (uint64_t)__shfl(d, 10, WAVE) | ((uint64_t)__shfl(d, 11, WAVE) << 32);
    const uint64_t wv_ = (uint64_t)__shfl(d, 12, WAVE) | ((uint64_t)__shfl(d, 13, WAVE) << 32);
    const uint64_t rb_ = (uint64_t)__shfl(d, 14, WAVE) | ((uint64_t)__shfl(d, 15, WAVE) << 32);
    const uint64_t cb_ = (uint64_t)__shfl(d, 16, WAVE) | ((uint64_t)__shfl(d, 17, WAVE) << 32);
    const uint64_t ncb_ = (uint64_t)__shfl(d, 18, WAVE) | ((uint64_t)__shfl(d, 19, WAVE) << 32);
    if (lane == 0) {
        SeqOut &s = q.out;
        s.rec_base = rb_; s.commit_before = cb_; s.n_commit_before = ncb_; s.vis = n_end0 + n;
        s.e0 = e0; s.idx0 = idx0; s.w = wv_; s.n_end0 = n_end0; s.term = term; s.kstar = (int64_t)kst;
        s.estar = (flags & RECF_ESTAR) ? (int64_t)kst : -1; s.stale = (flags & RECF_STALE) ? 1u : 0u;
        s.n = n; s.head_round = (flags & RECF_HEAD) ? 1u : 0u; s.fuse_mask = flags & 0x1FFFu; s.fast = (flags & RECF_FAST) ? 1u : 0u;
        q.ok = (flags & RECF_OK) ? 1u : 0u;
        q.chain_did = (flags & RECF_CHAIN) ? 1u : 0u;
        q.refused = (flags & RECF_REFUSED) ? 1u : 0u;
    }
}

struct AppendLds {
    uint64_t pos[WAVE];
    uint64_t src[WAVE];
    uint32_t T[WAVE];
    uint32_t ubase[WAVE + 1];
    uint4    h0[WAVE];
    uint4    h1[WAVE];
    uint32_t uniform_nu;      /* units per entry when every entry of the round has the same size, else 0 */
    uint32_t fuse_mask;       /* SeqOut::fuse_mask */
    uint32_t fast;            /* SeqOut::fast */
    uint64_t slot0;           /* slot of the round's first entry */
};

/* which entry of the round owns 16-byte unit u, and which of its units it is */
__device__ static inline void locate_unit(const AppendLds &lds, uint32_t u, uint32_t unu, uint32_t nr, uint32_t &e, uint32_t &j)
{
    if (unu) { e = u / unu; j = u - e * unu; return; }
    uint32_t lo = 0, hi = nr - 1;                       /* largest e with ubase[e] <= u */
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (lds.ubase[mid] <= u) lo = mid; else hi = mid - 1;
    }
    e = lo; j = u - lds.ubase[e];
}

/* the sequencer of the same launch (k_call) has published SeqOut / round_virt: its flag is
 * replicated on 32 cache lines (word 1 of every tick line) so that a thousand pollers do not
 * queue up on one word; then an agent-scope acquire for the whole block */
__device__ static inline uint32_t wait_sequenced(const EngDev &E, const CallEnv &X, uint32_t b, uint32_t *s_flag)
{
    if (threadIdx.x == 0) {
        unsigned long long spins = 0;
        uint32_t f;
        while ((f = __hip_atomic_load(X.lines + (b & 31u) * 32 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1ull << 22)) { spin_timeout(E, 951); f = 1; break; }     /* bounded */
        }
        *s_flag = f;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return *s_flag;
}

/* One round by one workgroup of 256.  Everything that does not depend on the sequencer's
 * result -- the request descriptors, sizes, unit layout, and the first payload units -- is
 * fetched first; with IN_LAUNCH the block then waits for the sequencer block of the same launch
 * (k_call), so that fetch overlaps the sequencing.  Then: positions, indices, header words,
 * directory, (in step) the apply records, and the stores. */
template <bool IN_LAUNCH>
__device__ static inline void append_round(const EngDev &E, const CallEnv &X, uint64_t r0, uint32_t R, uint32_t push_mask, uint32_t r,
                                           AppendLds &lds, SeqLds *sq, uint32_t tick, uint32_t slice = 0, uint32_t n_slices = 1,
                                           const uint64_t *snap = nullptr, bool post_read = true, int rec_seg = -1)
{
    const uint32_t tid = threadIdx.x, lane = lane_id();
    const RepDev &Ld = E.rep[E.leader];
    const uint32_t *rf = E.round_first + r0;
    const uint32_t g0 = rf[0];
    if (r == 0) STAMP(1, 0);
    const uint32_t first = rf[r] - g0, nr = rf[r + 1] - rf[r];

    /* ---- phase 1 (wave 0): descriptors, sizes, unit layout ---- */
    const bool active = lane < nr;
    ReqDev d; d.req_id = 0; d.pay16_type = 0; d.len = 0; d.clt_id = 0;
    uint32_t T = 0;
    uint64_t incl = 0;
    if (tid < WAVE && active) d = E.req[g0 + first + lane];
    uint64_t pfx_r = 0, pfx_0 = 0;
    if (IN_LAUNCH && rec_seg >= 0 && tid < WAVE) { pfx_r = E.round_prefix[r0 + r]; pfx_0 = E.round_prefix[r0]; }
    if (IN_LAUNCH && rec_seg < 0) {
        if (tid < WAVE) {
            /* the sequencer's inputs, in the same round trip as the descriptors; once they are in LDS
             * the sequencer block may start changing the control words (it waits for these tickets) */
            seq_w0_stage(E, r0, R, push_mask, r, *sq, snap);
            if (r == 0) STAMP(1, 3);
            if (tid == 0 && post_read) __hip_atomic_fetch_add(X.lines + ((r * n_slices + slice) & 31u) * 32 + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        /* one lane of wave 1 works the call's SeqOut out while wave 0 lays the round out */
        if (tid == WAVE) seq_w0_decide<false, true>(E, push_mask, tick, *sq);
    }
    if (tid < WAVE) {
        T = active ? APUS_HDR + d.len : 0;
        incl = wave_incl_scan((uint64_t)T);
        const uint32_t nu = active ? (T + 15) / 16 : 0;
        const uint32_t uincl = wave_incl_scan(nu);
        const uint32_t T0 = __shfl(T, 0, WAVE);
        const bool uni = __all(!active || T == T0);
        lds.src[lane] = (uint64_t)(d.pay16_type & 0x0FFFFFFFu) * 16;
        lds.T[lane] = T;
        lds.ubase[lane] = uincl - nu;
        if (lane == WAVE - 1) { lds.ubase[WAVE] = uincl; lds.uniform_nu = uni ? (T0 + 15) / 16 : 0; }
    }
    __syncthreads();

    /* ---- phase 2 (all): the first payload units into registers ---- */
    /* a round with many units is shared by n_slices workgroups: each lays the round out for
     * itself and copies a contiguous share of the units; slice 0 also writes the directory, the
     * ACK words and (in step) the apply records */
    const uint32_t uall = lds.ubase[WAVE];
    const uint32_t ushare = ((uall + n_slices - 1) / n_slices + 255u) & ~255u;
    const uint32_t ubeg = min(uall, slice * ushare), utotal = min(uall, ubeg + ushare);
    const uint32_t unu = lds.uniform_nu;
    constexpr int PF = 2;
    uint4 pv[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        pv[k] = make_uint4(0, 0, 0, 0);
        const uint32_t u = ubeg + tid + (uint32_t)k * 256;
        if (u < utotal) {
            uint32_t e, j;
            locate_unit(lds, u, unu, nr, e, j);
            const uint32_t Te = lds.T[e];
            const uint32_t so = min(16u * j, Te - 16u);
            if (so >= 48) pv[k] = payload_unit(E.arena + lds.src[e], so, Te - APUS_HDR, Te - APUS_HDR);
        }
    }
    /* the call's SeqOut: worked out here, from the same inputs by the same code as the sequencer
     * block's, while the payload loads are in flight -- nobody waits for the sequencer */
    if (r == 0) STAMP(1, 4);
    if (IN_LAUNCH) {
        if (rec_seg >= 0) {            /* a segment of a multi-segment launch: the SeqOut comes from its record */
            if (tid < WAVE) {
                rec_wait(E, (uint32_t)rec_seg, *sq);
                if (tid == 0) { sq->pfx[0] = pfx_0; sq->my_virt = pfx_r - pfx_0; }
            }
            __syncthreads();
            if (sq->refused) { if (tid == 0) lds.fast = 1; __syncthreads(); return; }     /* segment_refused: nothing is stored */
            if (!sq->ok) {             /* the block-wide path, on snapshot rec_seg */
                if (rec_seg > 0) wait_count(E, E.step_epoch, r, (uint32_t)rec_seg);
                __syncthreads();
                if (tid < WAVE) seq_w0_stage(E, r0, R, push_mask, r, *sq, E.step_snap + (size_t)rec_seg * SNAP_STRIDE);
                __syncthreads();
                if (tid == 0) seq_w0_decide<false, true>(E, push_mask, tick, *sq);
                __syncthreads();
            }
        }
        if (!sq->ok) {                 /* the batch could reach len: the block-wide scan, on the inputs staged above */
            const uint32_t *rb = E.round_bytes + r0;
            for (uint32_t i = tid; i < R && i < 1024; i += blockDim.x) sq->bytes0[i] = rb[i];
            __syncthreads();
            seq_body<false>(E, r0, R, push_mask, tick, push_mask, *sq, r);
        }
    }
    if (r == 0) STAMP(1, 1);

    /* ---- phase 3 (wave 0): where the entries go ---- */
    if (tid < WAVE) {
        const SeqOut s = IN_LAUNCH ? sq->out : *E.seq;
        const uint64_t a = s.e0 + (IN_LAUNCH ? sq->my_virt : E.round_virt[r]) + incl - T;
        const int64_t gk = (int64_t)first + lane;
        const uint64_t pos = apus_place(s, gk, a);
        const uint64_t idx = apus_entry_idx(s, gk);
        const uint64_t slot = s.n_end0 + (uint64_t)gk;
        const uint32_t type = d.pay16_type >> 28;

        const uint4 h0 = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)s.term, (uint32_t)(s.term >> 32));
        /* bytes 16..31: req_id, clt_id, type, sender (= leader: persist_new_entries), reply[0..3] = 0 */
        const uint4 h1 = make_uint4((uint32_t)d.req_id, (uint32_t)(d.req_id >> 32),
                                    (uint32_t)d.clt_id | (type << 16) | ((uint32_t)E.leader << 24), 0);
        lds.pos[lane] = pos;
        if (lane == WAVE - 1) { lds.fuse_mask = s.fuse_mask; lds.fast = s.fast; lds.slot0 = s.n_end0 + first; }
        lds.h0[lane] = h0;
        lds.h1[lane] = h1;

        if (active && slice == 0) {
            const uint32_t di = (uint32_t)slot & E.dir_mask;
            const uint32_t dl = T | ((uint32_t)E.leader << 24);     /* derived: total bytes | sender << 24 */
            Ld.dir_off[di] = pos; Ld.dir_len[di] = dl; Ld.ack[di] = s.fuse_mask;     /* ACK bits of the fused followers */
            for (uint32_t m = push_mask; m; m &= m - 1) {
                const RepDev &Fd = E.rep[__builtin_ctz(m)];
                Fd.dir_off[di] = pos; Fd.dir_len[di] = dl;
            }
            if (s.stale && gk == s.kstar) {
                /* the header that log_append_entry wrote before it found out that the
                 * payload does not fit (dare_log.h:497-504, 521-523); readers use it to
                 * detect the wrap (log_fit_entry) */
                const uint4 z = make_uint4(0, 0, 0, 0);
                const uint4 l = make_uint4((uint32_t)d.len, 0, 0, 0);
                for (uint32_t m = push_mask | (1u << E.leader); m; m &= m - 1) {
                    uint8_t *rg = E.rep[__builtin_ctz(m)].ring;
                    st16u(rg + a, h0); st16u(rg + a + 16, h1);
                    st16u(rg + a + 32, z); st16u(rg + a + 48, l);
                }
            }
        }
    }
    __syncthreads();
    if (r == 0) STAMP(1, 5);

    /* ---- phase 4 (all): the round's bytes as 16-byte units, to the leader ring and every
     * pushed follower ring at the same offset.  Reply bytes ride with the entry: fused followers
     * persist + ACK as part of the push (their own byte in their ring, every fused follower's
     * byte in the leader's ring) ---- */
    const uint32_t fuse = lds.fuse_mask;
    const ReplyWords rwl = apus_reply_words(fuse);
    auto store_unit = [&](uint32_t e, uint32_t so, uint4 v) {
        const uint64_t p = lds.pos[e] + so;
        if (fuse && so - 16u <= 16u) {                 /* the two units that hold reply[0..12] */
            const bool second = so == 16;
            st16u(Ld.ring + p, second ? make_uint4(v.x, v.y, v.z, rwl.w28) : make_uint4(rwl.x32, rwl.y36, rwl.z40, 0));
            for (uint32_t m = push_mask; m; m &= m - 1) {
                const int f = __builtin_ctz(m);
                const ReplyWords rwf = apus_reply_words(fuse & (1u << f));
                st16u(E.rep[f].ring + p, second ? make_uint4(v.x, v.y, v.z, rwf.w28) : make_uint4(rwf.x32, rwf.y36, rwf.z40, 0));
            }
        } else {
            st16u(Ld.ring + p, v);
            for (uint32_t m = push_mask; m; m &= m - 1) st16u(E.rep[__builtin_ctz(m)].ring + p, v);
        }
    };
#pragma unroll
    for (int k = 0; k < PF; k++) {
        const uint32_t u = ubeg + tid + (uint32_t)k * 256;
        if (u < utotal) {
            uint32_t e, j;
            locate_unit(lds, u, unu, nr, e, j);
            const uint32_t Te = lds.T[e];
            const uint32_t so = min(16u * j, Te - 16u);
            const uint4 v = so == 0 ? lds.h0[e] : so == 16 ? lds.h1[e] : so == 32 ? make_uint4(0, 0, 0, 0) : pv[k];
            store_unit(e, so, v);
        }
    }
    if (lds.fast && slice == 0 && tid >= 3 * WAVE) {
        /* apply_committed_entries (dare_server.c:1815-1974) for the round, by the last wave while
         * the first stores are in flight: what the replicas would read back from their logs is
         * still on chip -- record + stream hash for the leader (kind 1: proxy_update_state) and
         * every fused follower (kind 2: proxy_do_action) */
        const bool act = lane < nr;
        uint64_t mix1 = 0, mix2 = 0;
        if (act) {
            const uint64_t slot = lds.slot0 + lane, pos = lds.pos[lane];
            const uint4 h0 = lds.h0[lane], h1 = lds.h1[lane];
            const uint64_t idx = (uint64_t)h0.x | ((uint64_t)h0.y << 32);
            const uint32_t len = lds.T[lane] - APUS_HDR;
            const uint32_t tail = h1.z & 0x00FFFFFFu;                    /* clt_id | type << 16 */
            const uint32_t di = (uint32_t)slot & E.dir_mask;
            const uint4 r0v = make_uint4((uint32_t)slot, (uint32_t)(slot >> 32), (uint32_t)pos, (uint32_t)(pos >> 32));
            uint4 *rp = (uint4 *)&Ld.apply[di];
            rp[0] = r0v; rp[1] = make_uint4(h0.x, h0.y, len, tail | (1u << 24));
            for (uint32_t m = fuse; m; m &= m - 1) {
                uint4 *fp = (uint4 *)&E.rep[__builtin_ctz(m)].apply[di];
                fp[0] = r0v; fp[1] = make_uint4(h0.x, h0.y, len, tail | (2u << 24));
            }
            mix1 = apus_apply_mix(slot, pos, idx, len, (uint16_t)tail, (uint8_t)(tail >> 16), 1);
            mix2 = apus_apply_mix(slot, pos, idx, len, (uint16_t)tail, (uint8_t)(tail >> 16), 2);
        }
        /* the round's contribution to the stream hashes; the call's record blocks fold them
         * into the control blocks (a thousand workgroups adding to the same words would queue
         * up in one L2 channel), the upcall counters advance by the batch size there too */
        const uint64_t sum1 = wave_sum(mix1), sum2 = wave_sum(mix2);
        if (lane == 0) {     /* write-through: a record block of the same launch may read them */
            __hip_atomic_store(&X.hash[2 * r], sum1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&X.hash[2 * r + 1], sum2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (uint32_t u = ubeg + tid + PF * 256; u < utotal; u += 256) {
        uint32_t e, j;
        locate_unit(lds, u, unu, nr, e, j);
        const uint32_t Te = lds.T[e];
        const uint32_t so = min(16u * j, Te - 16u);
        uint4 v;
        if (so == 0) v = lds.h0[e];
        else if (so == 16) v = lds.h1[e];
        else if (so == 32) v = make_uint4(0, 0, 0, 0);
        else v = payload_unit(E.arena + lds.src[e], so, Te - APUS_HDR, Te - APUS_HDR);
        store_unit(e, so, v);
    }
    if (r == 0) STAMP(1, 2);
}

/* ------------------------------------------------------------------------- */
#define APUS_GP 4            /* wavefronts of an append workgroup */
#ifndef APUS_GD
#define APUS_GD 2            /* rounds per wavefront, one after the other; the second one's loads are issued with the first one's */
#endif
#define APUS_GR (APUS_GP * APUS_GD)      /* rounds per workgroup */
#ifndef APUS_GP_PF
#define APUS_GP_PF 4          /* payload units per lane fetched before the SeqOut is known */
#endif
struct GroupLds {
    /* per wave, reused from one round to the next */
    uint64_t pos[APUS_GP][WAVE];
    uint64_t idx[APUS_GP][WAVE];
    uint64_t req[APUS_GP][WAVE];
    uint32_t tail[APUS_GP][WAVE];         /* clt_id | type << 16 | sender << 24 */
    /* the replicas a unit is stored to (leader first), with the reply words their copy carries */
    uint8_t *tgt_ring[APUS_GP][APUS_DEV_MAX_SERVERS];
    uint4    tgt_rw[APUS_GP][APUS_DEV_MAX_SERVERS];        /* {w28, x32, y36, z40} */
    /* per round of the wave: its layout ... */
    uint32_t src16[APUS_GD][APUS_GP][WAVE];                /* payload offset in the arena / 16 */
    uint32_t T[APUS_GD][APUS_GP][WAVE];
    uint32_t ubase[APUS_GD][APUS_GP][WAVE + 1];
    /* ... and, for the rounds after the first, what was fetched ahead: descriptors and payload units */
    uint4    dsc[APUS_GD - 1][APUS_GP][WAVE];
    uint4    pay[APUS_GD - 1][APUS_GP][APUS_GP_PF][WAVE];
};

/* Grouped append (k_call / k_step, small rounds): ONE WAVEFRONT PER ROUND at a time, APUS_GD rounds
 * per wavefront, APUS_GP wavefronts per workgroup.
 *   round trip 1   the descriptors (lane = entry) and byte prefix of ALL the wave's rounds;
 *                  k_call: wave 0 also the call's control words (seq_w0_stage)
 *   on chip        the layout of every round (wave scans, no block barrier inside a round)
 *   round trip 2   the first APUS_GP_PF payload units per lane of every round: the first round's
 *                  stay in registers, the later rounds' are parked in LDS
 *   the segment's sequencing record (k_step) / the call's SeqOut (k_call)
 *   stores         round after round: lane l writes units l, l+64, ... to the leader ring and every
 *                  pushed follower ring (consecutive lanes = consecutive 16-byte units)
 * Everything a wave loads before its first store is in flight while it waits for the record, and a
 * launch's append workgroups are ALL resident from the start (5462 rounds of configs[1] = 683
 * workgroups on 768 slots): when they trickled in behind each other, every newcomer's two dependent
 * round trips queued behind the residents' store traffic and the launch ran at ~3 TB/s instead of
 * the ~7 TB/s the first, pre-loaded wave of workgroups reached (profiles/README.md, timeline).
 * Semantics are append_round<true>'s, line for line (positions, stale header, fused reply
 * bytes, directory, in-step apply records and the per-round hash words).
 * rec_seg >= 0: a segment of a multi-segment launch -- the SeqOut comes from the segment's record
 * (rec_wait); rec_seg < 0: k_call -- the block stages the control words and sequences for itself */
__device__ static inline uint32_t append_group(const EngDev &E, const CallEnv &X, uint64_t r0, uint32_t R, uint32_t push_mask,
                                               uint32_t grp, GroupLds &gl, SeqLds *sq, uint32_t tick,
                                               const uint64_t *snap, bool post_read, int rec_seg, uint32_t gd)
{
    /* gd <= APUS_GD rounds per wave (the host's choice: as few as keep all append workgroups of the launch resident) */
    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const RepDev &Ld = E.rep[E.leader];
    const uint32_t *rf = E.round_first + r0;
    const uint32_t rbase = (grp * APUS_GP + wv) * gd;          /* the wave's first round */
    const uint32_t GRr = APUS_GP * gd;                         /* rounds per workgroup */
    const uint32_t n_grp = (R + GRr - 1) / GRr;
    constexpr int PF = APUS_GP_PF;
    if (rec_seg >= 0 && rec_seg < 64) { if (grp == 0) STAMPN(10, rec_seg); if (grp + 1 == n_grp) STAMPN(14, rec_seg); }

    /* ---- round trip 0 / 1: round bounds, then descriptors, of every round of the wave ---- */
    const uint32_t g0 = gld(&rf[0]);
    const uint64_t pfx_0 = gld(&E.round_prefix[r0]);
    uint32_t rfv[APUS_GD + 1];
    uint64_t pfx_r[APUS_GD];
#pragma unroll
    for (int it = 0; it <= APUS_GD; it++) rfv[it] = gld(&rf[min(rbase + (uint32_t)it, R)]);
#pragma unroll
    for (int it = 0; it < APUS_GD; it++) pfx_r[it] = gld(&E.round_prefix[r0 + min(rbase + (uint32_t)it, R)]);
    uint4 dv[APUS_GD];
#pragma unroll
    for (int it = 0; it < APUS_GD; it++) {
        dv[it] = make_uint4(0, 0, 0, 0);
        if ((uint32_t)it < gd && rbase + it < R && lane < rfv[it + 1] - rfv[it]) dv[it] = ld16u((const uint8_t *)&E.req[rfv[it] + lane]);
    }
    if (rec_seg < 0) {
        if (wv == 0) {
            seq_w0_stage(E, r0, R, push_mask, grp * GRr, *sq, snap);
            if (tid == 0 && post_read) __hip_atomic_fetch_add(X.lines + (grp & 31u) * 32 + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (grp == 0) STAMP(8, 0);
        __syncthreads();
        if (grp == 0) STAMP(8, 1);
        /* one lane of wave 1 works the call's SeqOut out while the waves lay their rounds out */
        if (tid == WAVE) seq_w0_decide<false, true>(E, push_mask, tick, *sq);
        if (grp == 0) STAMPW(8, 7, WAVE);
    } else if (grp == 0) STAMP(8, 0);

    /* ---- the layout of a round (registers + wave scans) ---- */
    struct Lay { bool active; uint32_t T, uall, unu, T0, magic, nr; uint64_t incl; };
    auto lay_of = [&](const uint4 &d, uint32_t nr) -> Lay {
        Lay y;
        y.nr = nr;
        y.active = lane < nr;
        y.T = y.active ? APUS_HDR + (d.w & 0xFFFFu) : 0;
        y.incl = wave_incl_scan((uint64_t)y.T);
        y.T0 = __shfl(y.T, 0, WAVE);
        const bool uni = __all(!y.active || y.T == y.T0);
        y.unu = uni ? (y.T0 + 15) / 16 : 0;
        y.magic = y.unu > 1 ? (uint32_t)((1ull << 32) / y.unu + 1) : 0;   /* u / unu = mulhi(u, magic) for u < 64 * 261 */
        y.uall = 0;
        return y;
    };
    /* which entry of the round owns 16-byte unit u, at which byte offset of the entry */
    auto unit_of = [&](int it, const Lay &y, uint32_t u, uint32_t &e, uint32_t &so, uint32_t &Te) {
        uint32_t j;
        if (y.unu) { e = y.unu > 1 ? __umulhi(u, y.magic) : u; j = u - e * y.unu; }
        else {
            uint32_t lo = 0, hi = y.nr - 1;             /* largest e with ubase[e] <= u */
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (gl.ubase[it][wv][mid] <= u) lo = mid; else hi = mid - 1;
            }
            e = lo; j = u - gl.ubase[it][wv][e];
        }
        Te = gl.T[it][wv][e];
        so = min(16u * j, Te - 16u);
    };
    auto src_of = [&](int it, uint32_t e) -> const uint8_t * { return E.arena + (uint64_t)gl.src16[it][wv][e] * 16; };
    /* tables of round `it` + the number of units; returns the layout */
    auto lay_out = [&](int it, const uint4 &d, uint32_t nr) -> Lay {
        Lay y = lay_of(d, nr);
        const uint32_t nu = y.active ? (y.T + 15) / 16 : 0;
        const uint32_t uincl = wave_incl_scan(nu);
        y.uall = __shfl(uincl, WAVE - 1, WAVE);
        gl.src16[it][wv][lane] = d.z & 0x0FFFFFFFu;
        gl.T[it][wv][lane] = y.T;
        gl.ubase[it][wv][lane] = uincl - nu;
        if (lane == WAVE - 1) gl.ubase[it][wv][WAVE] = uincl;
        return y;
    };
    Lay L0;
    {
        Lay ly[APUS_GD];
#pragma unroll
        for (int it = 0; it < APUS_GD; it++) ly[it] = lay_out(it, dv[it], ((uint32_t)it < gd && rbase + it < R) ? rfv[it + 1] - rfv[it] : 0u);
        __builtin_amdgcn_wave_barrier();          /* LDS is in order within a wave; keep the compiler from reordering */
        L0 = ly[0];
        /* ---- round trip 2: the first payload units of every round; the later rounds' go to LDS ---- */
#pragma unroll
        for (int it = APUS_GD - 1; it >= 1; it--) {
            if ((uint32_t)it >= gd) continue;
            uint4 t[PF];
#pragma unroll
            for (int k = 0; k < PF; k++) {
                t[k] = make_uint4(0, 0, 0, 0);
                const uint32_t u = lane + (uint32_t)k * WAVE;
                if (u < ly[it].uall) {
                    uint32_t e, so, Te;
                    unit_of(it, ly[it], u, e, so, Te);
                    if (so >= 48) t[k] = payload_unit(src_of(it, e), so, Te - APUS_HDR, Te - APUS_HDR);
                }
            }
            gl.dsc[it - 1][wv][lane] = dv[it];
#pragma unroll
            for (int k = 0; k < PF; k++) gl.pay[it - 1][wv][k][lane] = t[k];
        }
    }
    uint4 pv[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        pv[k] = make_uint4(0, 0, 0, 0);
        const uint32_t u = lane + (uint32_t)k * WAVE;
        if (u < L0.uall) {
            uint32_t e, so, Te;
            unit_of(0, L0, u, e, so, Te);
            if (so >= 48) pv[k] = payload_unit(src_of(0, e), so, Te - APUS_HDR, Te - APUS_HDR);
        }
    }
    if (grp == 0) STAMP(8, 2);
    if (rec_seg >= 0 && wv == 0) {            /* the segment's record: normally there by now */
        rec_wait(E, (uint32_t)rec_seg, *sq);
        if (lane == 0) sq->pfx[0] = pfx_0;
    }
    __syncthreads();                          /* the SeqOut is in sq->out (or sq->ok == 0) */
    if (grp == 0) STAMP(8, 3);
    if (rec_seg >= 0 && rec_seg < 64 && grp == 0) STAMPN(11, rec_seg);
    if (rec_seg >= 0 && sq->refused) return 1;     /* the segment does not fit into the log (segment_refused): nothing is stored */
    if (!sq->ok) {                            /* the batch could reach len: the block-wide scan */
        if (rec_seg >= 0) {
            /* on snapshot rec_seg (the state before this segment): complete once the epoch says so
             * (segment 0's was written before its record) */
            if (rec_seg > 0) wait_count(E, E.step_epoch, grp, (uint32_t)rec_seg);
            __syncthreads();
            if (tid < WAVE) seq_w0_stage(E, r0, R, push_mask, grp * GRr, *sq, E.step_snap + (size_t)rec_seg * SNAP_STRIDE);
            __syncthreads();
            if (tid == 0) seq_w0_decide<false, true>(E, push_mask, tick, *sq);
            __syncthreads();
        }
        if (!sq->ok) {
            const uint32_t *rb = E.round_bytes + r0;
            for (uint32_t i = tid; i < R && i < 1024; i += blockDim.x) sq->bytes0[i] = rb[i];
            __syncthreads();
            seq_body<false>(E, r0, R, push_mask, tick, push_mask, *sq, grp * GRr);
            __syncthreads();
        }
    }

    const uint64_t e0 = sq->out.e0, n_end0 = sq->out.n_end0, term = sq->out.term;
    const uint32_t fuse = sq->out.fuse_mask, fast = sq->out.fast;
    /* the wave's target table: lane t < 13 = replica t, the leader first */
    if (lane < APUS_DEV_MAX_SERVERS && (((push_mask | (1u << E.leader)) >> lane) & 1u)) {
        const uint32_t at = lane == E.leader ? 0u : 1u + (uint32_t)__popc(push_mask & ~(1u << E.leader) & ((1u << lane) - 1u));
        const ReplyWords rw = apus_reply_words(lane == E.leader ? fuse : (fuse & (1u << lane)));
        gl.tgt_ring[wv][at] = E.rep[lane].ring;
        gl.tgt_rw[wv][at] = make_uint4(rw.w28, rw.x32, rw.y36, rw.z40);
    }

    /* ---- one round: where its entries go (lane = entry), its bytes, its apply records ---- */
    auto emit = [&](int it, const uint4 &dvv, const Lay &y) {
        const uint32_t r = rbase + (uint32_t)it;
        const bool has = r < R;          /* (it < gd: the caller's loop) */
        const uint32_t first = rfv[it] - g0;
        const bool active = y.active;
        const uint32_t T = y.T, uall = y.uall, unu = y.unu, nr = y.nr;
        const uint64_t incl = y.incl;
        ReqDev d;
        d.req_id = (uint64_t)dvv.x | ((uint64_t)dvv.y << 32); d.pay16_type = dvv.z; d.len = (uint16_t)dvv.w; d.clt_id = (uint16_t)(dvv.w >> 16);
        const SeqOut &s = sq->out;
        const uint64_t virt_r = sq->ok ? pfx_r[it] - sq->pfx[0] : (has ? sq->virt[r] : 0);
        const uint64_t a = e0 + virt_r + incl - T;
        const int64_t gk = (int64_t)first + lane;
        const uint64_t pos = apus_place(s, gk, a);
        const uint64_t idx = apus_entry_idx(s, gk);
        const uint64_t slot = n_end0 + (uint64_t)gk;
        const uint32_t type = d.pay16_type >> 28;
        const uint32_t tail = (uint32_t)d.clt_id | (type << 16) | ((uint32_t)E.leader << 24);
        gl.pos[wv][lane] = pos; gl.idx[wv][lane] = idx; gl.req[wv][lane] = d.req_id; gl.tail[wv][lane] = tail;
        if (active) {
            const uint32_t di = (uint32_t)slot & E.dir_mask;
            const uint32_t dl = T | ((uint32_t)E.leader << 24);     /* derived: total bytes | sender << 24 */
            gst_nt(&Ld.dir_off[di], pos); gst_nt(&Ld.dir_len[di], dl); gst_nt(&Ld.ack[di], fuse);     /* ACK bits of the fused followers */
            for (uint32_t m = push_mask; m; m &= m - 1) {
                const RepDev &Fd = E.rep[__builtin_ctz(m)];
                gst_nt(&Fd.dir_off[di], pos); gst_nt(&Fd.dir_len[di], dl);
            }
            if (s.stale && gk == s.kstar) {
                /* the header log_append_entry wrote before it found out that the payload does
                 * not fit (dare_log.h:497-504, 521-523) */
                const uint4 h0 = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)term, (uint32_t)(term >> 32));
                const uint4 h1 = make_uint4((uint32_t)d.req_id, (uint32_t)(d.req_id >> 32), tail, 0);
                const uint4 z = make_uint4(0, 0, 0, 0);
                const uint4 lw = make_uint4((uint32_t)d.len, 0, 0, 0);
                for (uint32_t m = push_mask | (1u << E.leader); m; m &= m - 1) {
                    uint8_t *rg = E.rep[__builtin_ctz(m)].ring;
                    st16u(rg + a, h0); st16u(rg + a + 16, h1);
                    st16u(rg + a + 32, z); st16u(rg + a + 48, lw);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (grp == 0 && it == 0) STAMP(8, 4);

        /* ---- the round's bytes: lane l stores units l, l + 64, ... ---- */
        /* (a batch that wraps: every round but the one the wrapping entry sits INSIDE is still laid out
         * in one piece -- before the wrap as sequenced, from the wrapping entry on shifted to offset 0) */
        if (unu && sq->ok && (sq->out.kstar < 0 || sq->out.kstar <= (int64_t)first || sq->out.kstar >= (int64_t)(first + nr))) {
            /* Entries of one size, no wrap inside the batch: everything about unit u follows from
             * arithmetic -- entry e = u / unu (multiply-high), its position a_r + e * T, its index
             * idx_r + e -- and ONE straight-line store sequence per target replica: the value is
             * selected, not branched on (header words, the reply words of this replica's copy, or
             * payload).  The general path below runs every divergent case of the wave one after the
             * other, ~500 VALU instructions per unit; with 12 wavefronts per CU the append blocks were
             * bound by instruction issue (~10 us per block), not by HBM (tools/timeline_probe.py). */
            const uint32_t magic = y.magic;
            const uint32_t Tu = y.T0;
            const uint64_t a_r = __shfl(pos, 0, WAVE);                   /* where the round's first entry goes */
            const uint64_t idx_r = __shfl(idx, 0, WAVE);
            const uint32_t n_tgt = 1u + (uint32_t)__popc(push_mask & ~(1u << E.leader));
            auto fast_unit = [&](uint32_t u, uint4 pay) {
                const uint32_t e = unu > 1 ? __umulhi(u, magic) : u;
                const uint32_t j = u - e * unu;
                const uint32_t so = min(16u * j, Tu - 16u);
                const uint64_t p = a_r + (uint64_t)e * Tu + so;
                const uint64_t ix = idx_r + e, rq = gl.req[wv][e];
                const uint32_t tl = gl.tail[wv][e];
                uint4 v = pay;
                if (so == 0) v = make_uint4((uint32_t)ix, (uint32_t)(ix >> 32), (uint32_t)term, (uint32_t)(term >> 32));
                if (so == 16) v = make_uint4((uint32_t)rq, (uint32_t)(rq >> 32), tl, 0);
                if (so == 32) v = make_uint4(0, 0, 0, 0);
                const bool r16 = fuse && so == 16, r32 = fuse && so == 32;
                for (uint32_t t = 0; t < n_tgt; t++) {
                    const uint4 rw = gl.tgt_rw[wv][t];
                    uint4 vt = v;
                    if (r16) vt.w = rw.x;
                    if (r32) vt = make_uint4(rw.y, rw.z, rw.w, 0);
                    st16u(gl.tgt_ring[wv][t] + p, vt);
                }
            };
            auto fast_pay = [&](uint32_t u) -> uint4 {
                const uint32_t e = unu > 1 ? __umulhi(u, magic) : u;
                const uint32_t j = u - e * unu;
                const uint32_t so = min(16u * j, Tu - 16u);
                uint4 pay = make_uint4(0, 0, 0, 0);
                if (so >= 48) pay = payload_unit(src_of(it, e), so, Tu - APUS_HDR, Tu - APUS_HDR);
                return pay;
            };
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const uint32_t u = lane + (uint32_t)k * WAVE;
                if (u < uall) fast_unit(u, pv[k]);
            }
            for (uint32_t ub = PF * WAVE; ub < uall; ub += PF * WAVE) {
#pragma unroll
                for (int k = 0; k < PF; k++) {
                    const uint32_t u = ub + lane + (uint32_t)k * WAVE;
                    pv[k] = (u < uall) ? fast_pay(u) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < PF; k++) {
                    const uint32_t u = ub + lane + (uint32_t)k * WAVE;
                    if (u < uall) fast_unit(u, pv[k]);
                }
            }
        } else {
            const ReplyWords rwl = apus_reply_words(fuse);
            auto store_unit = [&](uint32_t e, uint32_t so, uint4 v) {
                const uint64_t p = gl.pos[wv][e] + so;
                if (fuse && so - 16u <= 16u) {                 /* the two units that hold reply[0..12] */
                    const bool second = so == 16;
                    st16u(Ld.ring + p, second ? make_uint4(v.x, v.y, v.z, rwl.w28) : make_uint4(rwl.x32, rwl.y36, rwl.z40, 0));
                    for (uint32_t m = push_mask; m; m &= m - 1) {
                        const int f = __builtin_ctz(m);
                        const ReplyWords rwf = apus_reply_words(fuse & (1u << f));
                        st16u(E.rep[f].ring + p, second ? make_uint4(v.x, v.y, v.z, rwf.w28) : make_uint4(rwf.x32, rwf.y36, rwf.z40, 0));
                    }
                } else {
                    st16u(Ld.ring + p, v);
                    for (uint32_t m = push_mask; m; m &= m - 1) st16u(E.rep[__builtin_ctz(m)].ring + p, v);
                }
            };
            auto header_or = [&](uint32_t e, uint32_t so, uint4 pay) -> uint4 {
                if (so == 0) { const uint64_t ix = gl.idx[wv][e]; return make_uint4((uint32_t)ix, (uint32_t)(ix >> 32), (uint32_t)term, (uint32_t)(term >> 32)); }
                if (so == 16) { const uint64_t rq = gl.req[wv][e]; return make_uint4((uint32_t)rq, (uint32_t)(rq >> 32), gl.tail[wv][e], 0); }
                if (so == 32) return make_uint4(0, 0, 0, 0);
                return pay;
            };
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const uint32_t u = lane + (uint32_t)k * WAVE;
                if (u < uall) {
                    uint32_t e, so, Te;
                    unit_of(it, y, u, e, so, Te);
                    store_unit(e, so, header_or(e, so, pv[k]));
                }
            }
            /* the rest of the round in batches of PF units per lane: all loads of a batch are issued
             * before its first store (one load latency per batch, not per unit) */
            for (uint32_t ub = PF * WAVE; ub < uall; ub += PF * WAVE) {
#pragma unroll
                for (int k = 0; k < PF; k++) {
                    pv[k] = make_uint4(0, 0, 0, 0);
                    const uint32_t u = ub + lane + (uint32_t)k * WAVE;
                    if (u < uall) {
                        uint32_t e, so, Te;
                        unit_of(it, y, u, e, so, Te);
                        if (so >= 48) pv[k] = payload_unit(src_of(it, e), so, Te - APUS_HDR, Te - APUS_HDR);
                    }
                }
#pragma unroll
                for (int k = 0; k < PF; k++) {
                    const uint32_t u = ub + lane + (uint32_t)k * WAVE;
                    if (u < uall) {
                        uint32_t e, so, Te;
                        unit_of(it, y, u, e, so, Te);
                        store_unit(e, so, header_or(e, so, pv[k]));
                    }
                }
            }
        }
        if (grp == 0 && it == 0) STAMP(8, 5);
        /* ---- in step: apply_committed_entries for the round, from the registers that built it
         * (leader kind 1: proxy_update_state, fused followers kind 2: proxy_do_action) ---- */
        if (fast) {
            uint64_t mix1 = 0, mix2 = 0;
            if (active) {
                const uint32_t len = T - APUS_HDR;
                const uint32_t t24 = tail & 0x00FFFFFFu;                      /* clt_id | type << 16 */
                const uint32_t di = (uint32_t)slot & E.dir_mask;
                const uint4 r0v = make_uint4((uint32_t)slot, (uint32_t)(slot >> 32), (uint32_t)pos, (uint32_t)(pos >> 32));
                uint8_t *rp = (uint8_t *)&Ld.apply[di];
                st16u(rp, r0v); st16u(rp + 16, make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), len, t24 | (1u << 24)));
                for (uint32_t m = fuse; m; m &= m - 1) {
                    uint8_t *fp = (uint8_t *)&E.rep[__builtin_ctz(m)].apply[di];
                    st16u(fp, r0v); st16u(fp + 16, make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), len, t24 | (2u << 24)));
                }
                mix1 = apus_apply_mix(slot, pos, idx, len, (uint16_t)t24, (uint8_t)(t24 >> 16), 1);
                mix2 = apus_apply_mix(slot, pos, idx, len, (uint16_t)t24, (uint8_t)(t24 >> 16), 2);
            }
            const uint64_t sum1 = wave_sum(mix1), sum2 = wave_sum(mix2);
            if (has && lane == 0) {     /* write-through: a record block of the same launch reads them */
                __hip_atomic_store(&X.hash[2 * r], sum1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&X.hash[2 * r + 1], sum2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __builtin_amdgcn_wave_barrier();          /* the next round reuses pos / idx / req / tail */
    };
    emit(0, dv[0], L0);
#pragma unroll
    for (int it = 1; it < APUS_GD; it++) {
        if ((uint32_t)it >= gd) break;
        /* what was fetched ahead for this round: its descriptors, its first payload units */
        const uint4 dn = gl.dsc[it - 1][wv][lane];
#pragma unroll
        for (int k = 0; k < PF; k++) pv[k] = gl.pay[it - 1][wv][k][lane];
        Lay y = lay_of(dn, (rbase + it < R) ? rfv[it + 1] - rfv[it] : 0u);
        y.uall = gl.ubase[it][wv][WAVE];
        emit(it, dn, y);
    }
    if (grp == 0) STAMP(8, 6);
    if (rec_seg >= 0 && rec_seg < 64 && grp + 1 == n_grp) { STAMPN(12, rec_seg); STAMP(13, rec_seg); }
    return fast;
}

__global__ __launch_bounds__(256) void k_append_push(const EngDev E, uint64_t r0, uint32_t R, uint32_t push_mask)
{
    __shared__ AppendLds lds;
    const CallEnv X = APUS_ENV_OF(E);
    append_round<false>(E, X, r0, R, push_mask, blockIdx.x, lds, nullptr, 0);
}

/* ------------------------------------------------------------------------- */
/* the slot up to which the batch is visible to followers / committable: when
 * the leader's end sits exactly on len the log reads as empty (dare_log.h:158)
 * and neither update_remote_logs nor the ACK scan touch the last round.       */
__device__ static inline uint64_t visible_slots(const EngDev &E, const uint64_t *lhdr, uint64_t r0, uint32_t R)
{
    /* all loads are issued before the first use so that they overlap */
    const uint64_t n_end = lhdr[H_N_END], end = lhdr[H_END], nvis = lhdr[H_N_VISIBLE];
    const uint32_t sn = E.seq->n;
    const uint64_t n_end0 = E.seq->n_end0;
    uint32_t rfa = 0, rfb = 0;
    if (R) { rfa = E.round_first[r0]; rfb = E.round_first[r0 + R - 1]; }
    if (end != E.log_len) return n_end;
    if (sn == 0) return nvis;                         /* nothing new: what was visible stays visible */
    if (R == 0) return n_end0;                        /* a control entry landed on len */
    return n_end0 + (rfb - rfa);                      /* everything before the round that landed on len */
}

/* commit slot reached by this call on the leader */
__device__ static inline uint64_t commit_slot(const EngDev &E, uint64_t vis)
{
    const uint64_t ff = __hip_atomic_load((unsigned long long *)&E.seq->first_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t cs = min(ff, vis);
    if (cs < E.seq->n_commit_before) cs = E.seq->n_commit_before;
    return cs;
}

/* follower persist_new_entries + rc_send_entries_reply (dare_server.c:1792-1810,
 * dare_ibv_rc.c:1828-1863) for entry slots [from, vis) of follower f            */
__device__ static inline void persist_ack_range(const EngDev &E, int f, uint64_t vis, uint64_t tid, uint64_t nth)
{
    const RepDev &Fd = E.rep[f];
    const uint64_t from = Fd.hdr[H_N_PERSIST];
    for (uint64_t s = from + tid; s < vis; s += nth) {
        const uint32_t di = (uint32_t)s & E.dir_mask;
        const uint64_t off = Fd.dir_off[di];
        /* entry->sender says whose log gets the ACK (dare_server.c:1806); the directory
         * carries a copy of that byte so that the ring line is only written, not read */
        const uint32_t sender = Fd.dir_len[di] >> 24;
        Fd.ring[off + 28 + f] = 1;                               /* local reply byte, dare_ibv_rc.c:1840 */
        if (sender < APUS_DEV_MAX_SERVERS && E.rep[sender].ring) {
            E.rep[sender].ring[off + 28 + f] = 1;                /* R3: 1-byte WRITE at the same offset */
            atomicOr(&E.rep[sender].ack[di], 1u << f);           /* derived ACK word the scan reads */
        }
    }
}

/* k_persist_commit: when the followers live on this device, one lane handles one
 * entry slot for ALL of them -- follower persist + ACK (reply byte in both rings, ACK
 * bit) -- and, since the lane then holds the slot's ACK bits, the quorum test of
 * update_remote_logs (dare_ibv_rc.c:1725-1758) right away:
 * popcount(ack | self) >= size/2+1 per lane, wave ballot, first slot without a majority.
 * Latency shape: one load round trip for the call context (k_sequence left it in SeqOut),
 * one for the slot's directory entry, then only stores.  Followers in the mask hold the
 * same entries at the same offsets as the leader (that is what R1 pushed), so the lane
 * reads the leader's directory entry once instead of one copy per follower.        */
__device__ static inline void persist_commit_blocks(const EngDev &E, uint32_t fmask, uint32_t blk, uint32_t nblk,
                                                    uint64_t *s_np /*[APUS_DEV_MAX_SERVERS]*/, uint64_t *s_c /*[4]*/)
{
    const uint32_t tid = threadIdx.x;
    if (blk == 0) STAMP(4, 0);
    if (tid < APUS_DEV_MAX_SERVERS) s_np[tid] = E.seq->np[tid];
    else if (tid == 16) s_c[0] = E.seq->vis;
    else if (tid == 17) s_c[1] = E.seq->scan_lo;
    else if (tid == 18) s_c[2] = E.seq->n_commit_before;
    else if (tid == 19) s_c[3] = E.seq->tail_needed;
    __syncthreads();
    if (!s_c[3]) return;          /* every pushed follower acknowledged with the push and a majority was reached */
    const RepDev &Ld = E.rep[E.leader];
    const uint64_t vis = s_c[0], lo = s_c[1], n_commit = s_c[2];
    if (blk == 0) STAMP(4, 1);
    const uint32_t size = E.group_size, size_mask = (1u << size) - 1, quorum = size / 2 + 1;
    const uint32_t self = 1u << E.leader;
    const uint64_t nth = (uint64_t)nblk * blockDim.x;
    for (uint64_t tile = lo + (uint64_t)blk * blockDim.x; tile < vis; tile += nth) {
        const uint64_t s = tile + tid;
        const bool in = s < vis;
        const uint32_t di = (uint32_t)(in ? s : vis - 1) & E.dir_mask;     /* clamped: the loads are unconditional */
        const uint64_t off = Ld.dir_off[di];
        const uint32_t sender = Ld.dir_len[di] >> 24;                      /* entry->sender, dare_server.c:1806 */
        if (blk == 0) STAMP(4, 2);
        bool ok = true;
        if (in) {
            uint8_t *sring = (sender == E.leader) ? Ld.ring
                           : (sender < APUS_DEV_MAX_SERVERS ? E.rep[sender].ring : nullptr);
            uint32_t bits = 0;
            for (uint32_t m = fmask; m; m &= m - 1) {
                const int f = __builtin_ctz(m);
                if (s < s_np[f]) continue;                               /* this follower persisted it earlier */
                E.rep[f].ring[off + 28 + f] = 1;                         /* local reply byte, dare_ibv_rc.c:1840 */
                if (sring) sring[off + 28 + f] = 1;                      /* R3: 1-byte WRITE at the same offset */
                bits |= 1u << f;
            }
            if (s >= n_commit) {
                uint32_t word = bits;
                if (sender == E.leader && (uint32_t)__popc((bits | self) & size_mask) >= quorum) {
                    if (bits) __hip_atomic_fetch_or(&Ld.ack[di], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    /* not decided by the local followers alone: the word may hold remote ACK bits */
                    if (bits && sender == E.leader) word = atomicOr(&Ld.ack[di], bits) | bits;
                    else word = __hip_atomic_load(&Ld.ack[di], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (uint32_t)__popc((word | self) & size_mask) >= quorum;   /* replies >= size/2+1, :1738 */
                }
            } else if (bits && sender == E.leader) {
                __hip_atomic_fetch_or(&Ld.ack[di], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (blk == 0) STAMP(4, 3);
        const unsigned long long bal = __ballot(!ok);
        if (bal && lane_id() == 0) {
            const uint64_t first = s + (uint64_t)__builtin_ctzll(bal);
            atomicMin((unsigned long long *)&E.seq->first_fail, (unsigned long long)first);
        }
    }
}

__global__ __launch_bounds__(256) void k_persist_commit(const EngDev E, uint64_t r0, uint32_t R, uint32_t fmask)
{
    __shared__ uint64_t s_np[APUS_DEV_MAX_SERVERS];
    __shared__ uint64_t s_c[4];
    persist_commit_blocks(E, fmask, blockIdx.x, gridDim.x, s_np, s_c);
}

/* The ACK scan of update_remote_logs (dare_ibv_rc.c:1725-1758) over slots
 * [from, vis): ACK words staged in LDS, one lane per entry,
 * popcount(ack | self) >= size/2+1, wave ballot, first slot without a majority.
 * Must be called by all threads of the block (tile0 / tile_stride in slots).    */
__device__ static inline void commit_scan(const EngDev &E, uint64_t from, uint64_t vis, uint64_t tile0,
                                          uint64_t tile_stride, uint32_t *s_ack)
{
    const RepDev &Ld = E.rep[E.leader];
    const uint32_t size = E.group_size;                          /* cid.size[0], dare_ibv_rc.c:1656 */
    const uint32_t size_mask = (1u << size) - 1;
    for (uint64_t tile = from + tile0; tile < vis; tile += tile_stride) {
        const uint64_t s = tile + threadIdx.x;
        const bool in = s < vis;
        __syncthreads();
        s_ack[threadIdx.x] = in ? __hip_atomic_load(&Ld.ack[(uint32_t)s & E.dir_mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : 0xFFFFFFFFu;
        __syncthreads();
        const uint32_t m = (s_ack[threadIdx.x] | (1u << E.leader)) & size_mask;
        const bool ok = !in || (uint32_t)__popc(m) >= size / 2 + 1;  /* replies >= size/2+1, :1738 */
        const unsigned long long bal = __ballot(!ok);
        if (bal && lane_id() == 0) {
            const uint64_t first = (s - 0) + (uint64_t)__builtin_ctzll(bal);
            atomicMin((unsigned long long *)&E.seq->first_fail, (unsigned long long)first);
        }
    }
}

__global__ __launch_bounds__(1024) void k_commit(const EngDev E, uint64_t r0, uint32_t R)
{
    __shared__ uint32_t s_ack[1024];
    const uint64_t from = E.seq->n_commit_before, vis = E.seq->vis;      /* left there by k_sequence */
    commit_scan(E, from, vis, (uint64_t)blockIdx.x * blockDim.x, (uint64_t)gridDim.x * blockDim.x, s_ack);
}

/* apply_committed_entries (dare_server.c:1815-1974) for slots [from, cs) of
 * replica p: apply-stream records, HEAD adoption candidates; block-reduced
 * counters.  Must be called by all threads of the block.                        */
template <int APPLY_ILP = 4>
__device__ static inline void apply_range(const EngDev &E, int p, uint64_t from, uint64_t cs, uint64_t tile0,
                                          uint64_t tile_stride, unsigned long long *s_acc /*[2]*/)
{
    const RepDev &Pd = E.rep[p];
    const bool leader = (uint32_t)p == E.leader;
    if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
    __syncthreads();
    /* APPLY_ILP slots per thread and pass: the three dependent memory steps (directory ->
     * header -> record) are issued for all of them before the first result is needed */
    for (uint64_t tile = from + tile0 * APPLY_ILP; tile < cs; tile += tile_stride * APPLY_ILP) {
        uint64_t sl[APPLY_ILP], off[APPLY_ILP];
        uint32_t T[APPLY_ILP];
        bool in[APPLY_ILP];
        /* loads are unconditional (lanes past the end re-read slot cs-1): a load inside a
         * divergent branch makes the compiler wait for it before the next one is issued */
#pragma unroll
        for (int k = 0; k < APPLY_ILP; k++) {
            sl[k] = tile + (uint64_t)k * blockDim.x + threadIdx.x;
            in[k] = sl[k] < cs;
            const uint32_t di = (uint32_t)(in[k] ? sl[k] : cs - 1) & E.dir_mask;
            off[k] = Pd.dir_off[di];
            T[k] = Pd.dir_len[di];
        }
        if (blockIdx.x == 0 && blockIdx.y == 0) STAMP(2, 2);
        uint4 u0[APPLY_ILP], u1[APPLY_ILP];
#pragma unroll
        for (int k = 0; k < APPLY_ILP; k++) {
            u0[k] = ld16u(Pd.ring + off[k]);
            u1[k] = ld16u(Pd.ring + off[k] + 16);
        }
#pragma unroll
        for (int k = 0; k < APPLY_ILP; k++) T[k] &= 0xFFFFFFu;
        if (blockIdx.x == 0 && blockIdx.y == 0) STAMP(2, 3);
        uint64_t mix = 0;
        uint32_t nclient = 0;
#pragma unroll
        for (int k = 0; k < APPLY_ILP; k++) {
            if (!in[k]) continue;
            const uint64_t idx = (uint64_t)u0[k].x | ((uint64_t)u0[k].y << 32);
            const uint32_t type = (u1[k].z >> 16) & 0xFF;
            const uint16_t clt = (uint16_t)(u1[k].z & 0xFFFF);
            const uint32_t client = (type != 0 && type != 2 && type != 3);
            const uint32_t kind = client ? (leader ? 1u : 2u) : 0u;
            /* the 32-byte record as two 16-byte stores (field by field it would be four
             * partial writes per slot): slot, off | idx, len, clt_id | type << 16 | kind << 24 */
            uint4 *rp = (uint4 *)&Pd.apply[(uint32_t)sl[k] & E.dir_mask];
            rp[0] = make_uint4((uint32_t)sl[k], (uint32_t)(sl[k] >> 32), (uint32_t)off[k], (uint32_t)(off[k] >> 32));
            rp[1] = make_uint4(u0[k].x, u0[k].y, T[k] - APUS_HDR, (uint32_t)clt | (type << 16) | (kind << 24));
            if (client) { mix += apus_apply_mix(sl[k], off[k], idx, T[k] - APUS_HDR, clt, (uint8_t)type, (uint8_t)kind); nclient++; }
            if (type == 3 && !leader)                  /* poll_config_entries: committed HEAD, dare_server.c:2164 */
                atomicMax((unsigned long long *)&Pd.hdr[H_HEAD_SLOT], (unsigned long long)(sl[k] + 1));
        }
        if (blockIdx.x == 0 && blockIdx.y == 0) STAMP(2, 4);
        const uint64_t wsum = wave_sum(mix);
        const uint32_t wcnt = wave_sum(nclient);
        if (lane_id() == 0 && wcnt) {
            atomicAdd(&s_acc[0], (unsigned long long)wsum);
            atomicAdd(&s_acc[1], (unsigned long long)wcnt);
        }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) STAMP(2, 5);
    __syncthreads();
    if (threadIdx.x == 0 && s_acc[1]) {
        atomicAdd((unsigned long long *)&Pd.hdr[H_APPLY_HASH], s_acc[0]);
        atomicAdd((unsigned long long *)&Pd.hdr[H_APPLY_COUNT], s_acc[1]);
        if (leader) atomicAdd((unsigned long long *)&Pd.hdr[H_HIGHEST_REC], s_acc[1]);
    }
    __syncthreads();
}


/* per-round commit record of rounds [r0, r0+R) of this call, one thread per round;
 * every block of k_apply takes a slice (gtid over gthreads)                        */
/* virt = nullptr: the end offsets come from E.rec_end (written by k_sequence, an earlier launch);
 * virt = an exclusive prefix of the round totals, offset by vbase (k_call: the block's own scan in
 * LDS, or the host-staged E.round_prefix): they are worked out here, the same way the sequencer
 * does, and written to E.rec_end as well */
__device__ static inline void finish_records(const EngDev &E, uint64_t r0, uint32_t R, uint64_t cs,
                                             uint64_t gtid, uint64_t gthreads, const SeqOut &s, uint64_t rec_base0,
                                             const uint64_t *virt = nullptr, uint64_t vbase = 0)
{
    const RepDev &Ld = E.rep[E.leader];
    const uint64_t L = E.log_len;
    const uint32_t hr = s.head_round;
    const uint64_t rec_base = rec_base0 + hr;
    const uint32_t *rf = E.round_first + r0;
    /* the <HEAD> round of a fused prune tick committed (or not) on its own, before the batch */
    uint64_t base_commit = s.commit_before;
    if (hr) {
        const uint64_t he = virt ? s.e0 : E.rec_end[rec_base0];       /* the <HEAD> entry ends where the batch starts */
        if (cs >= s.n_end0 && cs > s.n_commit_before && he != L) base_commit = he;
        if (gtid == 0 && rec_base0 < E.rec_cap) E.rec_commit[rec_base0] = base_commit;
    }
    for (uint64_t r = gtid; r < R; r += gthreads) {
        if (rec_base + r >= E.rec_cap) break;
        const uint64_t slot_end_r = s.n_end0 + (rf[r + 1] - rf[0]);
        const uint64_t slot_start_r = s.n_end0 + (rf[r] - rf[0]);
        const uint64_t c = min(cs, slot_end_r);
        uint64_t end_r, end_prev = 0;
        if (virt) {
            const int64_t g0 = (int64_t)rf[0];
            const uint64_t a_end = s.e0 + (virt[r + 1] - vbase);
            end_r = (s.kstar < 0 || (int64_t)rf[r + 1] - g0 - 1 < s.kstar) ? a_end : a_end - s.w;
            if (r) {
                const uint64_t p_end = s.e0 + (virt[r] - vbase);
                end_prev = (s.kstar < 0 || (int64_t)rf[r] - g0 - 1 < s.kstar) ? p_end : p_end - s.w;
            }
            E.rec_end[rec_base + r] = end_r;
        } else {
            end_r = E.rec_end[rec_base + r];
        }
        uint64_t cr;
        if (c <= s.n_commit_before) cr = s.commit_before;
        else if (c <= s.n_end0) cr = base_commit;
        else if (c == slot_end_r) cr = end_r;
        else cr = Ld.dir_off[(uint32_t)c & E.dir_mask];
        /* Reference quirk (dare_ibv_rc.c:1725-1758): when a polling() pass starts with the
         * commit pointer parked at the wrap position X (everything before is committed, the
         * pass's first entry wrapped to offset 0), the first scan redirects to offset 0, finds
         * no ACKs yet, and "commits" offset 0 -- the same position, but log_is_offset_larger(0, X)
         * holds, `committed` is set and rc_write_remote_logs returns before the followers
         * were brought up to date.  That pass therefore ends with commit == 0. */
        if (s.kstar >= 0 && (int64_t)(rf[r] - rf[0]) == s.kstar && s.w < L && cs >= slot_start_r &&
            (r ? true : base_commit == s.e0))
            cr = 0;
        /* ... and the pass behind it starts with the followers one step back (the end doorbell of the
         * wrapped round is still to be sent): its first scan finds the wrapped round acknowledged, commits
         * it, `committed` is set and the pass returns before its own entries went out -- it ends with
         * commit == the END OF THE ROUND BEFORE (pinned on the reference, tests/traces.py:
         * wrap_quirk_second_round); the pass after that one catches up. */
        if (r >= 1 && s.kstar >= 0 && (int64_t)(rf[r - 1] - rf[0]) == s.kstar && s.w < L && c > slot_start_r &&
            (r - 1 ? true : base_commit == s.e0))
            cr = virt ? end_prev : E.rec_end[rec_base + r - 1];
        /* a round that ended exactly on len could not commit: the log read as empty
         * (dare_log.h:158), the leader's scan saw distance 0 (dare_ibv_rc.c:1726) */
        if (end_r == L) {
            if (r == 0) cr = base_commit;
            else {
                const uint64_t pe = virt ? end_prev : E.rec_end[rec_base + r - 1];
                const uint64_t pc = min(cs, slot_start_r);
                cr = (pc <= s.n_commit_before) ? s.commit_before
                   : (pc <= s.n_end0 ? base_commit : (pc == slot_start_r ? pe : Ld.dir_off[(uint32_t)pc & E.dir_mask]));
            }
        }
        E.rec_commit[rec_base + r] = cr;
    }
}

/* Scalar bookkeeping of a call (all threads of one block).  mode 0: R staged
 * rounds; mode 1: one control-entry round (s.n tells whether it happened);
 * mode 2: quiesce.                                                              */
__device__ static inline void finish_call(const EngDev &E, uint64_t r0, uint32_t R, int mode, uint32_t fmask)
{
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *lh = Ld.hdr;
    const uint64_t L = E.log_len;
    const SeqOut s = *E.seq;
    const uint64_t vis = visible_slots(E, lh, r0, R);
    const uint64_t cs = commit_slot(E, vis);
    const uint64_t end_l = lh[H_END];
    const uint64_t n_end_l = lh[H_N_END];
    /* byte offset that corresponds to a slot boundary b (<= n_end) */
    auto slot_off = [&](uint64_t b) -> uint64_t {
        if (b == n_end_l) return end_l;
        return Ld.dir_off[(uint32_t)b & E.dir_mask];
    };
    const uint64_t commit_off = (cs > s.n_commit_before) ? slot_off(cs) : s.commit_before;
    const uint64_t vis_off = slot_off(vis);
    const uint64_t apply_l = lh[H_N_APPLY];
    const uint32_t tid = threadIdx.x;
    const uint64_t rec_base = *E.rec_count;
    __syncthreads();                                  /* everybody sampled the control words */

    __syncthreads();
    if (tid == 0) {
        if (mode == 0) {
            *E.rec_count = rec_base + R + s.head_round;
        } else if (mode == 1 && s.n) {
            if (rec_base < E.rec_cap)
                E.rec_commit[rec_base] = (end_l == L) ? s.commit_before : commit_off;
            *E.rec_count = rec_base + 1;
        }
        /* leader: commit, apply (update_remote_logs :1744-1758, apply_committed_entries) */
        lh[H_N_VISIBLE] = vis;
        if (cs > s.n_commit_before) { lh[H_COMMIT] = commit_off; lh[H_N_COMMIT] = cs; }
        if (cs > apply_l) { lh[H_APPLY] = slot_off(cs); lh[H_N_APPLY] = cs; }
    }
    /* followers: R2 end doorbell, persist bookkeeping, R4 lazy commit, apply, HEAD adoption */
    if (tid >= 1 && tid <= APUS_DEV_MAX_SERVERS) {
        const int f = (int)tid - 1;
        if ((fmask >> f) & 1u) {
            uint64_t *fh = E.rep[f].hdr;
            /* load first, store afterwards: the loads overlap instead of queueing behind stores */
            const uint64_t f_np = fh[H_N_PERSIST], f_nc = fh[H_N_COMMIT], f_na = fh[H_N_APPLY];
            const uint64_t f_sc = fh[H_STORE_COUNT], f_head = fh[H_HEAD], f_end = fh[H_END];
            const uint64_t hs = __hip_atomic_load((unsigned long long *)&fh[H_HEAD_SLOT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t cs_off = slot_off(cs);
            uint64_t end_now = f_end;
            if (vis > f_np) {
                fh[H_STORE_COUNT] = f_sc + (vis - f_np);
                fh[H_END] = vis_off; fh[H_OLD_END] = vis_off;
                fh[H_N_END] = vis; fh[H_N_PERSIST] = vis;
                end_now = vis_off;
            }
            if (cs > f_nc) { fh[H_COMMIT] = cs_off; fh[H_N_COMMIT] = cs; }
            if (cs > f_na) { fh[H_APPLY] = cs_off; fh[H_N_APPLY] = cs; }
            if (hs) {
                const uint64_t hoff = E.rep[f].dir_off[(uint32_t)(hs - 1) & E.dir_mask];
                const uint64_t hv = ld8u(E.rep[f].ring + hoff + 48);
                if (apus_is_larger(end_now, L, hv, f_head)) fh[H_HEAD] = hv;
                fh[H_HEAD_SLOT] = 0;
            }
        }
    }
}

/* k_apply: apply on every replica in rmask (grid.y), then the block that finishes
 * last does the call's scalar bookkeeping (finish_call).                          */
/* what every block of k_apply needs before it can start, fetched in ONE round trip
 * (different lanes load different words) */
static_assert(sizeof(apus_apply_rec) == 32 && offsetof(apus_apply_rec, idx) == 16 && offsetof(apus_apply_rec, len) == 24 &&
              offsetof(apus_apply_rec, clt_id) == 28 && offsetof(apus_apply_rec, type) == 30 && offsetof(apus_apply_rec, kind) == 31,
              "apply_range stores the record as two uint4");

struct ApplyCtx {
    uint64_t lh[64];          /* leader control block (bookkeeper only) */
    SeqOut   seq;
    uint64_t rec_base, n_apply_p;
    uint64_t fw[APUS_DEV_MAX_SERVERS][8];   /* bookkeeping block: followers' control words */
    uint64_t off_cs, off_vis;
};
static_assert(sizeof(SeqOut) / 8 <= 32, "k_apply stages SeqOut with lanes 64..95");

/* what a block of the call's tail needs before it can start, fetched in ONE round trip
 * (different lanes load different words); p = replica an applier works on, -1 otherwise */
__device__ static inline void stage_apply_ctx(const EngDev &E, ApplyCtx &c, int p, bool keeper, uint32_t fmask)
{
    const uint32_t tid = threadIdx.x;
    const uint64_t *lh = E.rep[E.leader].hdr;
    if (tid < 64) { if (keeper) c.lh[tid] = lh[tid]; }
    else if (tid < 64 + sizeof(SeqOut) / 8) ((uint64_t *)&c.seq)[tid - 64] = ((const uint64_t *)E.seq)[tid - 64];
    else if (tid == 97) c.n_apply_p = (p >= 0) ? E.rep[p].hdr[H_N_APPLY] : 0;
    else if (keeper && tid >= 128 && tid < 128 + 8 * APUS_DEV_MAX_SERVERS) {
        const uint32_t f = (tid - 128) >> 3, j = (tid - 128) & 7;
        c.fw[f][j] = ((fmask >> f) & 1u) ? ldw(&E.rep[f].hdr[fw_hdr_word(j)]) : 0;      /* FW_* order */
    }
    __syncthreads();
    if (tid == 0) c.rec_base = c.seq.rec_base;      /* k_sequence noted it; the bookkeeper advances *rec_count */
    __syncthreads();
}

__device__ static inline uint64_t ctx_commit_slot(const ApplyCtx &c)
{
    uint64_t cs = min((uint64_t)c.seq.first_fail, c.seq.vis);
    if (cs < c.seq.n_commit_before) cs = c.seq.n_commit_before;
    return cs;
}

/* fast path: the per-round stream sums k_append_push left, folded into the replicas' hashes */
__device__ static inline void fold_round_hashes(const EngDev &E, const CallEnv &X, uint32_t R, uint32_t blk, uint32_t nblk, uint32_t fuse_mask)
{
    uint64_t h1 = 0, h2 = 0;
    for (uint64_t r = (uint64_t)blk * blockDim.x + threadIdx.x; r < R; r += (uint64_t)nblk * blockDim.x) {
        h1 += X.hash[2 * r]; h2 += X.hash[2 * r + 1];
    }
    h1 = wave_sum(h1); h2 = wave_sum(h2);
    if (lane_id() == 0 && (h1 | h2)) {
        atomicAdd((unsigned long long *)&E.rep[E.leader].hdr[H_APPLY_HASH], (unsigned long long)h1);
        for (uint32_t m = fuse_mask; m; m &= m - 1)
            atomicAdd((unsigned long long *)&E.rep[__builtin_ctz(m)].hdr[H_APPLY_HASH], (unsigned long long)h2);
    }
}

/* record blocks: the leader's per-round commit record, and (fast path) the per-round stream
 * sums k_append_push left folded into the replicas' hashes; blk of nblk, all threads */
__device__ static inline void recorder_body(const EngDev &E, uint64_t r0, uint32_t R, int mode, uint64_t cs,
                                            uint32_t blk, uint32_t nblk, const ApplyCtx &c)
{
    const uint32_t tid = threadIdx.x;
    if (mode == 0)
        finish_records(E, r0, R, cs, (uint64_t)blk * blockDim.x + tid, (uint64_t)nblk * blockDim.x, c.seq, c.rec_base);
    if (mode == 0 && c.seq.fast) { const CallEnv X = APUS_ENV_OF(E); fold_round_hashes(E, X, R, blk, nblk, c.seq.fuse_mask); }
}

/* the bookkeeper, once everybody else is done: per-call counters, the leader's commit / apply
 * offsets (update_remote_logs :1744-1758), and for every follower the R2 end doorbell, persist
 * bookkeeping, R4 lazy commit, apply offset and HEAD adoption.  c.off_cs / c.off_vis are set. */
/* pure_head: k_call in step -- the followers applied the call's <HEAD> entry (if any) when the
 * sequencer pushed it; slot + 1 and value are derived (head_slot1, head_value) instead of read */
__device__ static inline void keeper_publish(const EngDev &E, ApplyCtx &c, uint32_t R, int mode, uint32_t fmask,
                                             uint64_t vis, uint64_t cs, bool pure_head = false, uint64_t head_slot1 = 0,
                                             uint64_t head_value = 0, uint64_t *snap_next = nullptr)
{
    const uint32_t tid = threadIdx.x;
    uint64_t *lh = E.rep[E.leader].hdr;
    const uint64_t L = E.log_len;
    const uint64_t end_l = c.lh[H_END];
    const SeqOut &s = c.seq;
    const uint64_t commit_off = (cs > s.n_commit_before) ? c.off_cs : s.commit_before;
    /* every store is mirrored in the LDS copies (c.lh, c.fw, c.rec_base): they become the state
     * after the call, which a multi-segment launch hands to the next segment as a snapshot */
    if (tid == 0) {
        if (mode == 0) {
            c.rec_base = c.rec_base + R + s.head_round;
            gst(E.rec_count, (uint64_t)c.rec_base);
        } else if (mode == 1 && s.n) {
            if (c.rec_base < E.rec_cap) E.rec_commit[c.rec_base] = (end_l == L) ? s.commit_before : commit_off;
            c.rec_base = c.rec_base + 1;
            gst(E.rec_count, (uint64_t)c.rec_base);
        }
        gst(&lh[H_N_VISIBLE], (uint64_t)(vis)); c.lh[H_N_VISIBLE] = vis;
        if (cs > s.n_commit_before) { gst(&lh[H_COMMIT], (uint64_t)(commit_off)); gst(&lh[H_N_COMMIT], (uint64_t)(cs)); c.lh[H_COMMIT] = commit_off; c.lh[H_N_COMMIT] = cs; }
        if (cs > c.lh[H_N_APPLY]) { gst(&lh[H_APPLY], (uint64_t)(c.off_cs)); gst(&lh[H_N_APPLY], (uint64_t)(cs)); c.lh[H_APPLY] = c.off_cs; c.lh[H_N_APPLY] = cs; }
        if (mode == 0 && s.fast) {
            /* the append blocks applied the batch (every entry a client entry): one upcall each */
            atomicAdd((unsigned long long *)&lh[H_APPLY_COUNT], (unsigned long long)s.n);
            atomicAdd((unsigned long long *)&lh[H_HIGHEST_REC], (unsigned long long)s.n);
            for (uint32_t m = s.fuse_mask; m; m &= m - 1)
                atomicAdd((unsigned long long *)&E.rep[__builtin_ctz(m)].hdr[H_APPLY_COUNT], (unsigned long long)s.n);
        }
    }
    if (tid >= 1 && tid <= APUS_DEV_MAX_SERVERS) {
        const int f = (int)tid - 1;
        if ((fmask >> f) & 1u) {
            uint64_t *fh = E.rep[f].hdr;
            uint64_t *w = c.fw[f];
            const uint64_t f_np = w[FW_N_PERSIST], f_nc = w[FW_N_COMMIT], f_na = w[FW_N_APPLY], f_sc = w[FW_STORE_COUNT];
            const uint64_t f_head = w[FW_HEAD], f_end = w[FW_END];
            /* the appliers of this call may have raised the HEAD slot: read it now */
            const uint64_t hs = pure_head ? head_slot1
                              : __hip_atomic_load((unsigned long long *)&fh[H_HEAD_SLOT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint64_t end_now = f_end;
            if (vis > f_np) {
                gst(&fh[H_STORE_COUNT], (uint64_t)(f_sc + (vis - f_np)));
                gst(&fh[H_END], (uint64_t)(c.off_vis)); gst(&fh[H_OLD_END], (uint64_t)(c.off_vis));
                gst(&fh[H_N_END], (uint64_t)(vis)); gst(&fh[H_N_PERSIST], (uint64_t)(vis));
                w[FW_STORE_COUNT] = f_sc + (vis - f_np); w[FW_END] = c.off_vis; w[FW_N_END] = vis; w[FW_N_PERSIST] = vis;
                end_now = c.off_vis;
            }
            if (cs > f_nc) { gst(&fh[H_COMMIT], (uint64_t)(c.off_cs)); gst(&fh[H_N_COMMIT], (uint64_t)(cs)); w[FW_N_COMMIT] = cs; }
            if (cs > f_na) { gst(&fh[H_APPLY], (uint64_t)(c.off_cs)); gst(&fh[H_N_APPLY], (uint64_t)(cs)); w[FW_APPLY] = c.off_cs; w[FW_N_APPLY] = cs; }
            if (hs) {
                uint64_t hv = head_value;
                if (!pure_head) {
                    const uint64_t hoff = E.rep[f].dir_off[(uint32_t)(hs - 1) & E.dir_mask];
                    hv = ld8u(E.rep[f].ring + hoff + 48);
                }
                if (apus_is_larger(end_now, L, hv, f_head)) { gst(&fh[H_HEAD], (uint64_t)(hv)); w[FW_HEAD] = hv; }
                if (!pure_head) gst(&fh[H_HEAD_SLOT], (uint64_t)(0));
            }
        }
    }
    if (snap_next) {
        /* the state after this call, for the next segment of the launch (write-once, uncached) */
        __syncthreads();
        {
            if (tid < 64) gst(&snap_next[tid], c.lh[tid]);
            else if (tid < 64 + 8 * APUS_DEV_MAX_SERVERS) gst(&snap_next[SNAP_FW + (tid - 64)], (&c.fw[0][0])[tid - 64]);
            else if (tid == 64 + 8 * APUS_DEV_MAX_SERVERS) gst(&snap_next[SNAP_REC], (uint64_t)c.rec_base);
        }
    }
}

/* thread 0 spins until ticket `which` reaches `want` (bounded), then the whole block passes an
 * agent-scope acquire: what the ticketing blocks released is visible to plain loads */
__device__ static inline void wait_ticket(const EngDev &E, const CallEnv &X, int which, uint32_t want)
{
    if (threadIdx.x == 0) {
        unsigned long long spins = 0;
        while (__hip_atomic_load(X.ticket + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1ull << 22)) { spin_timeout(E, 2050); break; }     /* bounded */
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
/* the append blocks' arrivals are spread over 32 counters in 32 cache lines (a thousand
 * workgroups finishing together would queue up on one word): lane i < 32 waits for counter i */
__device__ static inline void wait_append(const EngDev &E, const CallEnv &X, uint32_t n_blocks)
{
    if (threadIdx.x < 32) {
        const uint32_t quota = n_blocks / 32 + (threadIdx.x < (n_blocks & 31u) ? 1u : 0u);
        unsigned long long spins = 0;
        while (__hip_atomic_load(X.lines + threadIdx.x * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < quota) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1ull << 22)) { spin_timeout(E, 2065); break; }     /* bounded */
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ static inline void post_append(const EngDev &E, const CallEnv &X, uint32_t b, bool release)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(X.lines + (b & 31u) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
/* every block of k_call that works the sequencing out for itself has fetched its copy of the
 * control words (tick word 2 of the 32 lines): only then may they be changed */
__device__ static inline void wait_readers(const EngDev &E, const CallEnv &X, uint32_t n_readers)
{
    if (threadIdx.x < 32) {
        const uint32_t quota = n_readers / 32 + (threadIdx.x < (n_readers & 31u) ? 1u : 0u);
        unsigned long long spins = 0;
        while (__hip_atomic_load(X.lines + threadIdx.x * 32 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < quota) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1ull << 22)) { spin_timeout(E, 2088); break; }     /* bounded */
        }
    }
    __syncthreads();
}
/* all of the block's stores are done (barrier), optionally released to the device, then the ticket */
__device__ static inline void post_ticket(const EngDev &E, const CallEnv &X, int which, bool release)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(X.ticket + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* k_apply: apply_committed_entries on every replica in rmask (grid.y).  Block roles along x:
 *   [0, nA)        appliers (grid.y = replica)
 *   [nA, nA + nR)  the leader's per-round commit record (y == 0 only)
 *   nA + nR        the call's bookkeeper (y == 0 only): it fetches everything finish needs
 *                  while the others work, waits for their arrival tickets, then publishes
 *                  commit/apply offsets and the R2/R4 doorbell words.
 * Nobody waits for the bookkeeper, so its wait cannot deadlock.  Every role starts with ONE
 * load round trip (the context k_sequence left in SeqOut + a few control words). */
__global__ __launch_bounds__(256) void k_apply(const EngDev E, uint64_t r0, uint32_t R, uint32_t rmask,
                                               int mode, uint32_t fmask, uint32_t nR)
{
    __shared__ unsigned long long s_acc[2];
    __shared__ ApplyCtx c;
    const uint32_t tid = threadIdx.x;
    const uint32_t nA = gridDim.x - 1 - nR;
    const bool keeper = blockIdx.x == gridDim.x - 1;
    const bool recorder = !keeper && blockIdx.x >= nA;
    if ((keeper || recorder) && blockIdx.y != 0) return;
    int p = -1;
    for (int i = 0, k = 0; i < APUS_DEV_MAX_SERVERS; i++)
        if (rmask & (1u << i)) { if (k == (int)blockIdx.y) { p = i; break; } k++; }
    const RepDev &Ld = E.rep[E.leader];

    const bool B0 = blockIdx.x == 0 && blockIdx.y == 0;
    if (B0) STAMP(2, 0);
    if (keeper) STAMP(3, 0);
    if (recorder && blockIdx.x == nA) STAMP(5, 0);
    stage_apply_ctx(E, c, p, keeper, fmask);
    const uint64_t vis = c.seq.vis;
    const uint64_t cs = ctx_commit_slot(c);

    if (B0) STAMP(2, 1);
    if (recorder) {
        recorder_body(E, r0, R, mode, cs, blockIdx.x - nA, nR, c);
        __syncthreads();
        if (blockIdx.x == nA) STAMP(5, 1);
        if (tid == 0) atomicAdd(E.ticket + T_APPLY, 1u);
        return;
    }
    if (!keeper) {
        /* in step (SeqOut::fast): k_append_push applied the batch as it wrote it */
        if (p >= 0 && !c.seq.fast) apply_range(E, p, c.n_apply_p, cs, (uint64_t)blockIdx.x * blockDim.x,
                                               (uint64_t)nA * blockDim.x, s_acc);
        __syncthreads();
        if (B0) STAMP(2, 6);
        if (tid == 0) atomicAdd(E.ticket + T_APPLY, 1u);    /* arrival ticket; no fence needed (see below) */
        return;
    }

    /* ---- the bookkeeper ---- */
    const uint64_t end_l = c.lh[H_END], n_end_l = c.lh[H_N_END];
    if (tid == 0) c.off_cs = (cs == n_end_l) ? end_l : Ld.dir_off[(uint32_t)cs & E.dir_mask];
    if (tid == 1) c.off_vis = (vis == n_end_l) ? end_l : Ld.dir_off[(uint32_t)vis & E.dir_mask];
    /* wait for every applier: it only consumes words they updated with device-scope atomics
     * (HEAD slot) plus control words nobody else writes, so the ticket needs no fence */
    STAMP(3, 1);
    if (tid == 0) {
        const unsigned int want = nA * gridDim.y + nR;
        unsigned long long spins = 0;
        while (__hip_atomic_load(E.ticket + T_APPLY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1ull << 24)) { spin_timeout(E, 2164); break; }     /* bounded */
        }
        __hip_atomic_store(E.ticket + T_APPLY, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    STAMP(3, 2);
    keeper_publish(E, c, R, mode, fmask, vis, cs);
    STAMP(3, 3);
}

/* k_call: a whole run_rounds call (R <= 1024 rounds) in ONE launch.  Block roles by grid index:
 *   0                         the bookkeeper
 *   1                         the sequencer
 *   [2, 2 + R * SP)           append + push (+ fused persist/ACK, + apply when in step): one round each,
 *                             or SP workgroups per round when the rounds are large (host's choice)
 *   [.., + nR)                the leader's per-round record (+ fast-path hash fold); recorder 0 is
 *                             also the janitor: last to leave, it clears the call's counters
 *   [.., + nS)                follower persist + ACK + quorum scan; idle unless SeqOut::tail_needed
 *   [.., + nA * replicas)     apply_committed_entries per replica; idle when in step (SeqOut::fast)
 * Nobody waits for the sequencer on the fast path: every append block (and record block) works
 * the sequencing out for itself -- same inputs, same code (seq_stage + seq_body<false>), no
 * stores -- while its descriptor / payload loads are in flight; so do the record blocks and the
 * bookkeeper, which in step publishes right away (nothing it writes is read by another block of
 * the launch).  The sequencer block changes the control words only after every such block has
 * fetched its copy of them (tick word 2), then does the effects (seq_body<true>) and raises the
 * flag (tick word 1): 1 = in step, the idle roles just leave; 2 = not in step, its results are
 * released (L2 write-back) first and the persist/scan -> apply -> records -> bookkeeping chain
 * runs on arrival tickets, producers releasing and consumers acquiring at agent scope.
 * A block only ever waits for blocks with a LOWER index (or, for the sequencer, for tickets the
 * append blocks post before they wait for anything); workgroups are dispatched in index order per
 * XCD, so what a waiting block needs is already running or done: no co-residency assumption. */
/* one call's parameters (k_call's arguments; one entry per segment in k_step's table) */
/* k_call / k_step: three workgroups per CU (12 wavefronts) is what the launch geometry counts on
 * (flush_batch: all append workgroups of a launch resident at once); the register allocator is told
 * so -- left alone it drifts a few registers over the 168 that allows (a handful of spills in cold
 * paths instead).  -DAPUS_LB_W=k: another target. */
#ifndef APUS_LB_W
#define APUS_LB_W 3
#endif
#define APUS_CALL_BOUNDS __launch_bounds__(256, APUS_LB_W)
struct CallArgs {
    uint64_t r0;
    uint32_t R, tick, SP, nR, nS, nA;
    uint32_t GP;           /* > 1: grouped append, GP = APUS_GP x (1 .. APUS_GD) small rounds per workgroup (then SP == 1) */
};
/* number of append blocks of a call */
__host__ __device__ static inline uint32_t call_append_blocks(const CallArgs &A)
{
    return A.GP > 1 ? (A.R + A.GP - 1) / A.GP : A.R * A.SP;
}
union CallLds {
    AppendLds app;
    GroupLds grp;
    struct { ApplyCtx c; unsigned long long acc[2]; uint64_t np[APUS_DEV_MAX_SERVERS]; uint64_t sc[4]; uint32_t flag; } t;
};

#ifndef APUS_POLL_SLEEP
#define APUS_POLL_SLEEP 8       /* x 64 clocks between two polls of a chain count */
#endif
/* multi-segment launches: segment k waits until epoch >= k (the bookkeeper of segment k-1 has
 * written snapshot k) / until the sequencer of segment k-1 is done; 32 replicas, one per cache line */
__device__ static inline void wait_count(const EngDev &E, const uint32_t *lines32, uint32_t b, uint32_t want)
{
    if (threadIdx.x == 0) {
        unsigned long long spins = 0;
        while (__hip_atomic_load(lines32 + (b & 31u) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(APUS_POLL_SLEEP);
            if (++spins > (1ull << 22)) { spin_timeout(E, 2229); break; }     /* bounded */
        }
    }
    __syncthreads();
}
/* all of the block's stores (uncached control words, snapshot) have reached memory, then the count */
__device__ static inline void bump_count(uint32_t *lines32, uint32_t value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 32) __hip_atomic_store(lines32 + threadIdx.x * 32, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define APUS_STEP_SEGS 32
struct StepTable {
    CallArgs seg[APUS_STEP_SEGS];
    uint32_t blk0[APUS_STEP_SEGS + 1];        /* order 0: first block of every segment; blk0[S] = grid size */
    uint32_t S;
    /* order 1 (append blocks first): blocks [0, S) the segments' bookkeepers, [S, 2S) their sequencers,
     * then every segment's append blocks (ab0[k] = first of segment k), then every segment's record /
     * scan / apply blocks (sv0[k]).  All append blocks of the launch are then resident from the
     * start and issue their loads before anybody stores (see append_group); the host only picks it
     * when they -- and one segment's other blocks -- fit on the device together (flush_batch), which
     * keeps "a block only waits for blocks that are running or done" true in this order as well. */
    uint32_t order;
    uint32_t max_T;                           /* largest entry of the staged requests (segment_refused) */
    uint32_t ab0[APUS_STEP_SEGS + 1];
    uint32_t sv0[APUS_STEP_SEGS + 1];
};

/* The body of a call for block b of its grid.  STEP = false: k_call (one call per launch, inputs =
 * the live control blocks).  STEP = true: segment seg of S in a k_step launch: the blocks work from
 * the segment's sequencing record (rec_wait); where the record says "not in step" they fall back to
 * snapshot seg, which the chain block completes (epoch >= seg) before it publishes that record. */
template <bool STEP>
__device__ static inline void call_block(const EngDev &E, const CallEnv &X, const CallArgs &A, uint32_t push_mask, uint32_t rmask,
                                         uint32_t b_grid, SeqLds &sq, CallLds &l, uint32_t seg, uint32_t S,
                                         const StepTable *TT = nullptr)
{
    const uint64_t r0 = A.r0;
    const uint32_t R = A.R, tick = A.tick, SP = A.SP, nR = A.nR, nS = A.nS, nA = A.nA;
    /* Order in the grid: bookkeeper, sequencer, append blocks, record blocks, scan, apply.  The two
     * single blocks come FIRST: workgroups are dispatched in index order, and in a multi-segment
     * launch the next segment's append blocks wait for this segment's bookkeeper -- it must not
     * queue behind a thousand append blocks of its own segment.  (b below is the role index the
     * rest of the function was written in: sequencer 0, append 1 .. nAB, records, bookkeeper, ...) */
    uint32_t b;
    {
        const uint32_t nAB_ = call_append_blocks(A);
        if (b_grid == 0) b = 1 + nAB_ + nR;                    /* the bookkeeper */
        else if (b_grid == 1) b = 0;                           /* the sequencer */
        else if (b_grid < 2 + nAB_ + nR) b = b_grid - 1;       /* append and record blocks */
        else b = b_grid;                                       /* scan and apply blocks */
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t ny = (uint32_t)__popc(rmask);
    const uint32_t fmask = push_mask;
    ApplyCtx &c = l.t.c;
    const uint32_t nAB = call_append_blocks(A);                /* append blocks: SP per round, or one per GP rounds */
    /* blocks that fetch the live control words themselves: append + nR record + 1 bookkeeper; in a
     * multi-segment launch the append blocks work from the segments' records instead */
    const uint32_t n_readers = STEP ? nR + 1 : nAB + nR + 1;
    const uint32_t rl_base = STEP ? 0u : nAB;             /* tick line index of the first non-append reader */
    /* blocks that sign off with T_PASS: everybody but the append blocks and the janitor */
    const uint32_t n_pass = 1 + (nR - 1) + 1 + nS + nA * ny;
    /* inputs: the live control blocks, or (later segments of a launch) the previous bookkeeper's
     * snapshot -- then nobody has to wait before changing the live words */
    const bool live = !STEP || seg == 0;
    const uint64_t *snap = live ? nullptr : E.step_snap + (size_t)seg * SNAP_STRIDE;
    uint64_t *snap_next = STEP ? E.step_snap + (size_t)(seg + 1) * SNAP_STRIDE : nullptr;

    if (b >= 1 && b <= nAB) {                                  /* ---- append + push ---- */
        const uint32_t ab = b - 1;
        /* (a segment of a multi-segment launch: no wait for the epoch -- the block polls the
         * segment's sequencing record when it needs it, append_group / append_round) */
        if (b == nAB) STAMP(6, 0);
        const int rec_seg = STEP ? (int)seg : -1;
        uint32_t fast;
#ifndef APUS_NO_GP
        if (A.GP > 1) {
            fast = append_group(E, X, r0, R, push_mask, ab, l.grp, &sq, tick, snap, live && !STEP, rec_seg, A.GP / APUS_GP);
        } else
#endif
        {
            const uint32_t r = ab / SP, slice = ab - r * SP;
            append_round<true>(E, X, r0, R, push_mask, r, l.app, &sq, tick, slice, SP, snap, live && !STEP, rec_seg);
            fast = l.app.fast;
        }
        if (b == nAB) STAMP(6, 1);
        post_append(E, X, ab, fast == 0);
        if (b == nAB) STAMP(6, 2);
        return;
    }
    if (b == 0) {                                              /* ---- the sequencer ---- */
        if (STEP) {
            /* a segment that is in step is sequenced AND carried out by the chain block; in a
             * multi-segment launch this block only works for the segments the chain block hands over
             * (not in step, or the batch may wrap) -- and the chain block waits for it then, so the
             * sequencers need no chain of their own */
            if (tid < WAVE) rec_wait(E, seg, sq);
            __syncthreads();
            if (sq.chain_did) { post_ticket(E, X, T_PASS, false); return; }
            __syncthreads();
        }
        if (!live) wait_count(E, E.step_epoch, b, seg);
        seq_stage(E, r0, R, push_mask, push_mask, sq, true, STEP ? E.step_snap + (size_t)seg * SNAP_STRIDE : snap);
        if (live) wait_readers(E, X, n_readers);               /* every reader of the live words has its copy */
        /* the common case needs no block-wide scan: one lane decides and does the effects */
        if (tid == 0) seq_w0_decide<true>(E, push_mask, tick, sq);
        __syncthreads();
        if (sq.ok) {
            if (tid < sizeof(SeqOut) / 8) ((uint64_t *)E.seq)[tid] = ((const uint64_t *)&sq.out)[tid];
        } else {                                               /* possible wrap / lagging follower: the full sequencer */
            seq_body<true>(E, r0, R, push_mask, tick, push_mask, sq, 0, false);
        }
        __syncthreads();
        const bool fast = sq.out.fast != 0;
        if (!fast && tid == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   /* SeqOut, control words, <HEAD> entry */
        __syncthreads();
        if (tid < 32) __hip_atomic_store(X.lines + tid * 32 + 1, fast ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        post_ticket(E, X, T_PASS, false);
        return;
    }
    uint32_t q = b - 1 - nAB;
    if (q < nR) {                                              /* ---- per-round records ---- */
        if (q == 0) STAMP(5, 0);
        if (STEP && q == 0 && seg < 8) STAMPN(15, 8 * seg + 0);      /* janitor block: started */
        bool from_rec = false;
        if (STEP) {
            /* a segment the chain block sequenced: everything finish_records needs is in its record */
            if (tid < WAVE) {
                rec_wait(E, seg, sq);
                if (tid == 0) sq.pfx[0] = gld(&E.round_prefix[r0]);
            }
            __syncthreads();
            from_rec = sq.ok && sq.out.fast && sq.chain_did;
            /* (segment 0: the chain block counts this block among the readers of the live words) */
            if (from_rec && live && tid == 0)
                __hip_atomic_fetch_add(X.lines + ((rl_base + q) & 31u) * 32 + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!from_rec) {
            if (STEP) __syncthreads();
            if (!live) wait_count(E, E.step_epoch, b, seg);
            /* the sequencing, worked out locally (SeqOut in sq.out) */
            seq_local(E, X, r0, R, push_mask, tick, 0, rl_base + q, sq, snap, false, live);
        }
        if (q == 0) STAMP(5, 1);
        /* the rounds' byte prefix: the host-staged one, or the block's own scan */
        const uint64_t *virt = sq.ok ? E.round_prefix + r0 : sq.virt;
        const uint64_t vbase = sq.ok ? sq.pfx[0] : 0;
        const bool refused = STEP && from_rec && sq.refused;     /* segment_refused: no rounds, no records */
        if (sq.out.fast) {
            /* in step: the commit slot is known (everything visible commits); only the rounds'
             * hash words have to be waited for */
            if (!refused)
            finish_records(E, r0, R, sq.out.vis, (uint64_t)q * blockDim.x + tid, (uint64_t)nR * blockDim.x, sq.out,
                           sq.out.rec_base, virt, vbase);
            if (q == 0) STAMP(5, 2);
            if (STEP && q == 0 && seg < 8) STAMPN(15, 8 * seg + 1);  /* records written */
            wait_append(E, X, nAB);
            if (q == 0) STAMP(5, 3);
            if (STEP && q == 0 && seg < 8) STAMPN(15, 8 * seg + 2);  /* the segment's append blocks are done */
            if (!refused) fold_round_hashes(E, X, R, q, nR, sq.out.fuse_mask);
            if (q == 0) STAMP(5, 4);
            if (STEP && q == 0 && seg < 8) STAMPN(15, 8 * seg + 3);  /* hashes folded */
        } else {
            wait_append(E, X, nAB);
            wait_ticket(E, X, T_SCAN, nS);
            if (tid == 0)
                sq.out.first_fail = __hip_atomic_load((unsigned long long *)&E.seq->first_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            uint64_t cs = min((uint64_t)sq.out.first_fail, sq.out.vis);
            if (cs < sq.out.n_commit_before) cs = sq.out.n_commit_before;
            finish_records(E, r0, R, cs, (uint64_t)q * blockDim.x + tid, (uint64_t)nR * blockDim.x, sq.out, sq.out.rec_base, virt, vbase);
            post_ticket(E, X, T_DONE, false);
        }
        if (q != 0) { post_ticket(E, X, T_PASS, false); return; }
        /* the janitor: every append block is done (wait_append above), everybody else -- the
         * sequencer with its <HEAD> entry too -- signs off with T_PASS */
        wait_ticket(E, X, T_PASS, n_pass);
        if (STEP && seg < 8) STAMPN(15, 8 * seg + 4);                    /* everybody has signed off */
        /* a multi-segment launch: the last segment's janitor clears what the whole launch shares, so it
         * goes last of all janitors (every other block of a segment has signed off with its janitor) */
        if (STEP && S > 1) {
            uint32_t *jan = E.step_tickets + T_JANITORS;
            if (seg + 1 != S) {
                if (tid == 0) __hip_atomic_fetch_add(jan, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (tid == 0) {
                    unsigned long long spins = 0;
                    while (__hip_atomic_load(jan, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S - 1) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1ull << 22)) { spin_timeout(E, 2470); break; }     /* bounded */
                    }
                    __hip_atomic_store(jan, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
        }
        /* then the call's counters and the flag are cleared for the next call */
        if (tid < 32) {
            __hip_atomic_store(X.lines + tid * 32 + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(X.lines + tid * 32 + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(X.lines + tid * 32 + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            /* the launch's last segment: its chain counts too.  Every block that reads them belongs
             * to this or an earlier segment; this segment's have all signed off (above), earlier
             * segments' were dispatched before any block of this one and read the counts first thing
             * -- a block that had not got that far by now would have stalled for a whole segment and
             * runs into its bounded spin (status bit 4), it cannot hang */
            if (STEP && seg + 1 == S) {
                __hip_atomic_store(E.step_epoch + tid * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(E.step_seq_done + tid * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                /* and the segments' sequencing records (read first thing by every block that uses them) */
                for (uint32_t i = tid; i < S * REC_WORDS; i += 32)
                    __hip_atomic_store((unsigned long long *)&E.step_rec[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (tid == 32) {
            __hip_atomic_store(X.ticket + T_PASS, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(X.ticket + T_SCAN, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(X.ticket + T_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    q -= nR;
    if (q == 0) {                                              /* ---- the bookkeeper ---- */
        /* In a multi-segment launch ONE block -- segment 0's bookkeeper, the CHAIN BLOCK -- keeps the
         * books of every segment; the other segments' bookkeeper blocks just leave.
         *   1. RECORD PASS (wave 0, registers only, chain_core): every segment that is in step is
         *      sequenced from the state the previous one leaves and its record published -- that is
         *      all the append / record blocks wait for.  It stops at the first segment it cannot take.
         *   2a. every segment was taken: chain_books_fast -- per segment only what the next one does
         *      not overwrite (a due tick's <HEAD> entry, the log-full check, the sign-off), then the
         *      control words once, as the last segment leaves them.
         *   2b. otherwise the SECOND PASS below walks the segments on the LDS copies (c.lh, c.fw,
         *      c.rec_base): records for the segments the first pass did not reach, the sequencer's
         *      effects, keeper_publish with the snapshot the fallback roles of the next segment
         *      sequence from, the waits for a segment's own blocks where it is not in step, the epoch.
         * (Chained through memory the bookkeeping cost ~9 us per segment, through LDS and block barriers
         * ~3.5-5.5 us: tools/timeline_probe.py.) */
        if (STEP && seg > 0) return;
        const uint32_t S_ = STEP ? S : 1u;
        __shared__ uint32_t chain_fast;                        /* this segment was sequenced by chain_decide_fast */
        uint32_t epoch_done = 0;                               /* epochs raised so far (snapshots <= this are complete) */
        /* every segment's staged sizes, in one round trip: bytes, first / last request, length of the last request */
        __shared__ uint64_t pre_pfx[APUS_STEP_SEGS][2];
        __shared__ uint32_t pre_rf[APUS_STEP_SEGS][2];
        __shared__ uint64_t pre_last[APUS_STEP_SEGS];
        __shared__ uint32_t n_rec_s;
        if (STEP && TT && tid < S_) {
            const CallArgs &Ap = TT->seg[tid];
            const uint32_t a = E.round_first[Ap.r0], bb = E.round_first[Ap.r0 + Ap.R];
            pre_pfx[tid][0] = E.round_prefix[Ap.r0]; pre_pfx[tid][1] = E.round_prefix[Ap.r0 + Ap.R];
            pre_rf[tid][0] = a; pre_rf[tid][1] = bb;
            pre_last[tid] = (bb > a) ? E.req_len[bb - 1] : 0;
        }
        {   /* ---- the inputs of segment 0: the live control words, one round trip ---- */
            const CallArgs &A0 = (STEP && TT) ? TT->seg[0] : A;
            const uint32_t nAB0 = call_append_blocks(A0);
            /* followers' control words: nobody else writes them while the replicas are in step */
            if (tid >= 128 && tid < 128 + 8 * APUS_DEV_MAX_SERVERS) {
                const uint32_t f = (tid - 128) >> 3, j2 = (tid - 128) & 7;
                c.fw[f][j2] = ((fmask >> f) & 1u) ? stage_follower_word(E, nullptr, f, j2) : 0;     /* FW_* order */
            }
            if (tid < WAVE) {
                seq_w0_stage(E, A0.r0, A0.R, push_mask, 0, sq, nullptr, true);
                if (tid == 0) __hip_atomic_fetch_add(X.lines + ((STEP ? A0.nR : nAB0 + A0.nR) & 31u) * 32 + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) n_rec_s = 0;
            __syncthreads();
#ifndef APUS_NO_RECORD_PASS      /* diagnostics: every segment through the second pass */
            if (STEP && TT) {
                /* ---- the RECORD PASS: wave 0 sequences every segment it can on registers (chain_core)
                 * and publishes the records, before the block does anything else -- no LDS, no barrier, no
                 * store other than the records (on a CU it shares with two append blocks every LDS or
                 * memory round trip queues behind theirs: ~3.5 us per segment when this pass worked on
                 * the LDS copies, profiles/README.md).  It stops at the first segment that is not in
                 * step or needs the general path; the second pass publishes the records from there on. */
                if (tid < WAVE) {
                    const uint32_t lane = tid;
                    const bool srv = lane < APUS_DEV_MAX_SERVERS;
                    ChainRegs cr;
                    cr.end = sq.lh[H_END]; cr.n_end = sq.lh[H_N_END]; cr.last_idx = sq.lh[H_LAST_IDX]; cr.sid = sq.lh[H_SID];
                    cr.commit = sq.lh[H_COMMIT]; cr.n_commit = sq.lh[H_N_COMMIT]; cr.n_apply = sq.lh[H_N_APPLY]; cr.apply = sq.lh[H_APPLY];
                    cr.head = sq.lh[H_HEAD]; cr.tail = sq.lh[H_TAIL]; cr.prev_head = sq.lh[H_PREV_HEAD]; cr.store_count = sq.lh[H_STORE_COUNT];
                    cr.bitmask = (uint32_t)sq.lh[H_CID_BITMASK];
                    cr.apoff = srv ? sq.lh[H_APPLY_OFFSETS + lane] : 0;
                    cr.f_apply = srv ? c.fw[lane][FW_APPLY] : 0; cr.f_np = srv ? c.fw[lane][FW_N_PERSIST] : 0; cr.f_na = srv ? c.fw[lane][FW_N_APPLY] : 0;
                    cr.rec_base = sq.misc[0];
                    /* lane k holds segment k's staged sizes */
                    uint64_t my_pfx0 = 0, my_pfx2 = 0, my_last = 0, my_r0 = 0;
                    uint32_t my_rf0 = 0, my_rf1 = 0, my_R = 0, my_tick = 0;
                    if (lane < S_) {
                        my_pfx0 = pre_pfx[lane][0]; my_pfx2 = pre_pfx[lane][1]; my_last = pre_last[lane];
                        my_rf0 = pre_rf[lane][0]; my_rf1 = pre_rf[lane][1];
                        my_r0 = TT->seg[lane].r0; my_R = TT->seg[lane].R; my_tick = TT->seg[lane].tick;
                    }
                    uint32_t nd = 0;
                    for (uint32_t k = 0; k < S_; k++) {
                        ChainSeg g;
                        g.r0 = rl64(my_r0, k); g.pfx0 = rl64(my_pfx0, k); g.pfx2 = rl64(my_pfx2, k); g.len_last = rl64(my_last, k);
                        g.rf0 = (uint32_t)__builtin_amdgcn_readlane((int)my_rf0, (int)k); g.rf1 = (uint32_t)__builtin_amdgcn_readlane((int)my_rf1, (int)k);
                        g.R = (uint32_t)__builtin_amdgcn_readlane((int)my_R, (int)k); g.tick = (uint32_t)__builtin_amdgcn_readlane((int)my_tick, (int)k);
                        ChainOut o;
                        if (segment_refused(E.log_len, cr.end, cr.head, g.pfx2 - g.pfx0, g.tick, TT->max_T)) {
                            RecFields f;
                            f.e0 = cr.end; f.idx0 = cr.last_idx + 1; f.n_end0 = cr.n_end; f.term = cr.sid >> 9;
                            f.flags_n = (uint64_t)(rec_flags(1u, push_mask, 1u, 0u, 0u, false, true) | RECF_REFUSED);
                            f.kstar = ~0ull; f.w = 0; f.rec_base = cr.rec_base; f.commit_before = cr.commit; f.n_commit_before = cr.n_commit;
                            rec_publish_fields(E, k, f);
                            nd = k + 1;
                            continue;
                        }
                        if (!chain_core(E, push_mask, cr, g, o)) break;
                        RecFields f;
                        f.e0 = o.e0; f.idx0 = o.idx0; f.n_end0 = o.n_end0; f.term = cr.sid >> 9;
                        f.flags_n = (uint64_t)rec_flags(1u, push_mask, 1u, o.head_round, o.stale, o.estar >= 0, true) | ((uint64_t)o.n << 32);
                        f.kstar = (uint64_t)o.kstar; f.w = o.w; f.rec_base = cr.rec_base; f.commit_before = cr.commit; f.n_commit_before = cr.n_commit;
                        rec_publish_fields(E, k, f);
                        if (k < 64) STAMPN(9, k);
                        chain_advance(E, push_mask, cr, g, o);
                        nd = k + 1;
                    }
                    if (lane == 0) n_rec_s = nd;
                }
                __syncthreads();
            }
#endif
            if (STEP) {
                /* snapshot 0 = the state before the launch: what a fallback path of segment 0
                 * sequences from (the live words may change as soon as the readers are through) */
                uint64_t *s0 = E.step_snap;
                const bool snap0 = !TT || n_rec_s < S_;
                if (tid < 64) { if (snap0) gst(&s0[tid], sq.lh[tid]); c.lh[tid] = sq.lh[tid]; }
                else if (tid < 64 + 8 * APUS_DEV_MAX_SERVERS) { if (snap0) gst(&s0[SNAP_FW + (tid - 64)], (&c.fw[0][0])[tid - 64]); }
                else if (tid == 64 + 8 * APUS_DEV_MAX_SERVERS) { if (snap0) gst(&s0[SNAP_REC], sq.misc[0]); c.rec_base = sq.misc[0]; }
                __syncthreads();
            }
        }
        /* The second pass: everything but the records of the segments the record pass sequenced -- the
         * live control words, the <HEAD> entries, the snapshots, the tickets, and the general path for
         * the segments that need it.  It starts over from the state before the launch (c.lh / c.fw). */
        const uint32_t n_rec = (STEP && TT) ? n_rec_s : 0u;
#ifndef APUS_NO_FAST_BOOKS
        if (STEP && TT && S_ > 0 && n_rec == S_) {
            /* every segment is in step and was sequenced on registers: so are the books (chain_books_fast) */
            if (tid < WAVE) {
                const uint32_t lane = tid;
                const bool srv = lane < APUS_DEV_MAX_SERVERS;
                ChainRegs cr;
                cr.end = sq.lh[H_END]; cr.n_end = sq.lh[H_N_END]; cr.last_idx = sq.lh[H_LAST_IDX]; cr.sid = sq.lh[H_SID];
                cr.commit = sq.lh[H_COMMIT]; cr.n_commit = sq.lh[H_N_COMMIT]; cr.n_apply = sq.lh[H_N_APPLY]; cr.apply = sq.lh[H_APPLY];
                cr.head = sq.lh[H_HEAD]; cr.tail = sq.lh[H_TAIL]; cr.prev_head = sq.lh[H_PREV_HEAD]; cr.store_count = sq.lh[H_STORE_COUNT];
                cr.bitmask = (uint32_t)sq.lh[H_CID_BITMASK];
                cr.apoff = srv ? sq.lh[H_APPLY_OFFSETS + lane] : 0;
                cr.f_apply = srv ? c.fw[lane][FW_APPLY] : 0; cr.f_np = srv ? c.fw[lane][FW_N_PERSIST] : 0; cr.f_na = srv ? c.fw[lane][FW_N_APPLY] : 0;
                cr.rec_base = sq.misc[0];
                chain_books_fast(E, push_mask, cr, srv ? c.fw[lane][FW_STORE_COUNT] : 0, srv ? c.fw[lane][FW_HEAD] : 0,
                                 S_, pre_pfx, pre_rf, pre_last, TT->seg, sq.out, TT->max_T);
                __builtin_amdgcn_wave_barrier();
                if (lane < sizeof(SeqOut) / 8) gst(&((uint64_t *)E.seq)[lane], ((const uint64_t *)&sq.out)[lane]);
            }
            bump_count(E.step_epoch, S_);          /* (nobody waits for it in such a launch; the janitor clears it) */
            return;
        }
#endif
        /* snapshots are what the fallback paths of a segment sequence from: none of them runs when the
         * record pass got through the whole launch */
        const bool snaps = STEP && n_rec < S_;
        for (uint32_t k = 0; k < S_; k++) {
            const CallArgs &Ak = (STEP && TT) ? TT->seg[k] : A;
            const CallEnv Xk = (STEP && TT) ? CallEnv{E.step_lines + (size_t)k * 1024, E.step_tickets + (size_t)k * 32,
                                                      E.step_hash + (size_t)k * 2 * 1024} : X;
            const uint64_t r0k = Ak.r0;
            const uint32_t Rk = Ak.R, tickk = Ak.tick;
            const uint32_t nABk = call_append_blocks(Ak), nRk = Ak.nR, nAk = Ak.nA;
            const uint32_t n_readers_k = STEP ? nRk + 1 : nABk + nRk + 1;
            const bool live_k = !STEP || k == 0;
            uint64_t *snap_next_k = snaps ? E.step_snap + (size_t)(k + 1) * SNAP_STRIDE : nullptr;
            if (STEP) {
                /* the inputs: the state the previous segment's bookkeeping left in LDS (segment 0: the
                 * staged live words) + this segment's staged sizes (what seq_w0_stage fetches) */
                __syncthreads();
                if (tid < 64) sq.lh[tid] = c.lh[tid];
                else if (tid < 64 + 5 * APUS_DEV_MAX_SERVERS) {
                    const uint32_t w = tid - 64, f = w / 5, j2 = w - f * 5;
                    (&sq.fw[0][0])[w] = ((push_mask >> f) & 1u) ? c.fw[f][j2] : (j2 == FW_N_PERSIST ? ~0ull : 0ull);
                } else if (tid == 192) { sq.misc[0] = c.rec_base; sq.rstar = 0xFFFFFFFFu; sq.head_round = 0; }
                else if (tid == 193) { sq.pfx[0] = pre_pfx[k][0]; sq.pfx[1] = pre_pfx[k][0]; sq.pfx[2] = pre_pfx[k][1]; }
                else if (tid == 194) { sq.rfx[0] = pre_rf[k][0]; sq.rfx[1] = pre_rf[k][1]; sq.misc[1] = pre_last[k]; }
                __syncthreads();
            }
            if (STEP && TT && segment_refused(E.log_len, sq.lh[H_END], sq.lh[H_HEAD], pre_pfx[k][1] - pre_pfx[k][0], tickk, TT->max_T)) {
                /* does not fit into the log: refused as a whole (the record pass said the same for k < n_rec) */
                if (tid == 0) { set_status(E, 1u << 1); sq.ok = 1; }
                if (k >= n_rec && tid < WAVE) {
                    RecFields f;
                    f.e0 = sq.lh[H_END]; f.idx0 = sq.lh[H_LAST_IDX] + 1; f.n_end0 = sq.lh[H_N_END]; f.term = sq.lh[H_SID] >> 9;
                    f.flags_n = (uint64_t)(rec_flags(1u, push_mask, 1u, 0u, 0u, false, true) | RECF_REFUSED);
                    f.kstar = ~0ull; f.w = 0; f.rec_base = c.rec_base; f.commit_before = sq.lh[H_COMMIT]; f.n_commit_before = sq.lh[H_N_COMMIT];
                    rec_publish_fields(E, k, f);
                }
                if (snap_next_k) {       /* the state after this segment = the state before it */
                    if (tid < 64) gst(&snap_next_k[tid], c.lh[tid]);
                    else if (tid < 64 + 8 * APUS_DEV_MAX_SERVERS) gst(&snap_next_k[SNAP_FW + (tid - 64)], (&c.fw[0][0])[tid - 64]);
                    else if (tid == 64 + 8 * APUS_DEV_MAX_SERVERS) gst(&snap_next_k[SNAP_REC], (uint64_t)c.rec_base);
                }
                if (STEP && k + 1 == S_) bump_count(E.step_epoch, S_);
                post_ticket(E, Xk, T_PASS, false);
                continue;
            }
            if (tid < WAVE) {
                const int cf = STEP ? chain_decide_fast(E, push_mask, tickk, sq, r0k, Rk) : 0;
                if (!cf && tid == 0) seq_w0_decide<false>(E, push_mask, tickk, sq);
                if (tid == 0) chain_fast = (uint32_t)cf;
            }
            __syncthreads();
            /* this segment's other blocks (its sequencer first) have work the books depend on, unless the
             * chain block sequenced it itself (chain_fast: in step, then it also does the effects) */
            const bool wait_needed = !chain_fast;
            if (STEP) {
                if (wait_needed) {
                    /* they sequence from snapshot k: complete it (drain) before the record tells them to */
                    if (k == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
                    else if (epoch_done < k) { bump_count(E.step_epoch, k); epoch_done = k; }
                }
                if (k >= n_rec) {
                    if (tid < WAVE) rec_publish(E, k, sq, chain_fast != 0);
                    if (k < 64) STAMPN(9, k);
                }
            }
            if (!sq.ok) {
                const uint32_t *rb = E.round_bytes + r0k;
                for (uint32_t i2 = tid; i2 < Rk && i2 < 1024; i2 += blockDim.x) sq.bytes0[i2] = rb[i2];
                __syncthreads();
                seq_body<false>(E, r0k, Rk, push_mask, tickk, push_mask, sq, 0);
            }
            if (sq.out.fast) {
                /* in step: everything the bookkeeping needs follows from the sequencing it just worked
                 * out; nothing it writes is read by another block of the launch: publish right away */
                __syncthreads();
                if (tid < 64) c.lh[tid] = sq.lh[tid];
                else if (tid < 64 + sizeof(SeqOut) / 8) ((uint64_t *)&c.seq)[tid - 64] = ((const uint64_t *)&sq.out)[tid - 64];
                __syncthreads();
                const uint64_t vis = sq.out.vis, cs = vis;
                if (tid == 0) {
                    c.rec_base = sq.out.rec_base;
                    c.off_cs = sq.end_new; c.off_vis = sq.end_new;       /* vis == n_end: the batch is fully visible */
                }
                __syncthreads();
                if (live_k) wait_readers(E, Xk, n_readers_k);          /* it changes words the other blocks sequence from */
                if (STEP && chain_fast && tid < WAVE) chain_effects_fast(E, push_mask, tickk, sq);
                /* in step, but sequenced by the general path: this segment's sequencer block carries it
                 * out -- it has to be through before the next segment's effects touch the same words */
                if (STEP && !chain_fast) wait_sequenced(E, Xk, b, &l.t.flag);
                keeper_publish(E, c, Rk, 0, fmask, vis, cs, true, sq.out.head_round ? sq.out.n_end0 : 0, sq.lh[H_HEAD], snap_next_k);
            } else {
                wait_sequenced(E, Xk, b, &l.t.flag);               /* the sequencer's results, released */
                stage_apply_ctx(E, c, -1, true, fmask);
                wait_ticket(E, Xk, T_DONE, nAk * ny + nRk);
                if (tid == 0)
                    c.seq.first_fail = __hip_atomic_load((unsigned long long *)&E.seq->first_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                const uint64_t vis = c.seq.vis, cs = ctx_commit_slot(c);
                const uint64_t end_l = c.lh[H_END], n_end_l = c.lh[H_N_END];
                if (tid == 0) c.off_cs = (cs == n_end_l) ? end_l : E.rep[E.leader].dir_off[(uint32_t)cs & E.dir_mask];
                if (tid == 1) c.off_vis = (vis == n_end_l) ? end_l : E.rep[E.leader].dir_off[(uint32_t)vis & E.dir_mask];
                __syncthreads();
                keeper_publish(E, c, Rk, 0, fmask, vis, cs, false, 0, 0, snap_next_k);
            }
            /* the books of the launch are closed: every snapshot is complete -- raised BEFORE this
             * segment's sign-off (its janitor clears the chain counts once everybody has signed off) */
            if (STEP && k + 1 == S_) bump_count(E.step_epoch, S_);
            post_ticket(E, Xk, T_PASS, false);
        }
        return;
    }
    q -= 1;
    /* ---- the roles that only work when the replicas are not in step ---- */
    uint32_t flag = 0;
    if (STEP) {
        /* a segment's record says whether it is in step: then these blocks leave at once instead
         * of sitting on a CU until the segment's sequencer has run */
        if (tid < WAVE) rec_wait(E, seg, sq);
        __syncthreads();
        if (sq.ok && sq.out.fast) flag = 1;
    }
    if (!flag) flag = wait_sequenced(E, X, b, &l.t.flag);
    if (flag == 2) {
        if (q < nS) {                                          /* persist + ACK + quorum scan */
            if (tid == 0) l.t.sc[3] = E.seq->tail_needed;
            __syncthreads();
            if (l.t.sc[3]) {
                wait_append(E, X, nAB);
                persist_commit_blocks(E, fmask, q, nS, l.t.np, l.t.sc);
            }
            post_ticket(E, X, T_SCAN, l.t.sc[3] != 0);
        } else {                                               /* apply */
            q -= nS;
            const uint32_t y = q / nA, x = q - y * nA;
            int p = -1;
            for (int i = 0, k = 0; i < APUS_DEV_MAX_SERVERS; i++)
                if (rmask & (1u << i)) { if (k == (int)y) { p = i; break; } k++; }
            wait_ticket(E, X, T_SCAN, nS);                     /* first_fail is final, the entries are visible */
            stage_apply_ctx(E, c, p, false, fmask);
            /* (2 slots per lane and pass: keeps the whole kernel's register count down) */
            if (p >= 0 && !c.seq.fast) apply_range<2>(E, p, c.n_apply_p, ctx_commit_slot(c), (uint64_t)x * blockDim.x,
                                                      (uint64_t)nA * blockDim.x, l.t.acc);
            post_ticket(E, X, T_DONE, false);
        }
    }
    post_ticket(E, X, T_PASS, false);
}

/* Every block works from a copy of the engine descriptor (pointers, sizes) in LDS.  Read from
 * the kernarg segment through a generic pointer each E.rep[f].ring / E.dir_mask / ... is a memory load
 * that the compiler must repeat after every 16-byte store (the stores are char-typed: they may
 * alias anything in memory) -- a dependent round trip per store, ~10 us per append block under
 * load (tools/timeline_probe.py).  LDS cannot alias the rings, so the loads are hoisted, and what is
 * left costs an LDS access. */
__device__ static inline void stage_engine(EngDev &dst)
{
    const uint64_t *src = (const uint64_t *)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
    static_assert(sizeof(EngDev) % 8 == 0, "EngDev is copied as u64 words");
    for (uint32_t i = threadIdx.x; i < sizeof(EngDev) / 8; i += blockDim.x) ((uint64_t *)&dst)[i] = src[i];
    __syncthreads();
}

__global__ APUS_CALL_BOUNDS void k_call(const EngDev E_arg, const CallArgs A, uint32_t push_mask, uint32_t rmask)
{
    (void)E_arg;
    __shared__ EngDev E_s;
    stage_engine(E_s);
    const EngDev &E = E_s;
    __shared__ SeqLds sq;
    __shared__ CallLds l;
    const CallEnv X = APUS_ENV_OF(E);
    call_block<false>(E, X, A, push_mask, rmask, blockIdx.x, sq, l, 0, 1);
}

/* k_step: several consecutive calls (segments) in ONE launch -- k_call's roles per segment on
 * segment-private counters.  What replaces the kernel boundary between two calls: the chain block
 * (segment 0's bookkeeper) sequences every in-step segment ahead on registers and publishes a
 * sequencing RECORD per segment (rec_publish_fields / rec_wait) that the segment's append, record and
 * idle blocks work from; segments that are not in step fall back to snapshots of the control state
 * (write-once, uncached memory) chained by an epoch count.  Grid order 0: segment by segment; order 1
 * (StepTable::order, chosen by flush_batch when everything fits on the device at once): the single
 * blocks, then every segment's append blocks, then the rest -- all loads of the launch are issued
 * before anybody stores.  No end-of-kernel write-back, no dispatch ramp, no graph gap between
 * segments. */
__global__ APUS_CALL_BOUNDS void k_step(const EngDev E_arg, const StepTable T, uint32_t push_mask, uint32_t rmask)
{
    (void)E_arg;
    __shared__ EngDev E_s;                 /* see stage_engine */
    stage_engine(E_s);
    const EngDev &E = E_s;
    __shared__ SeqLds sq;
    __shared__ CallLds l;
    uint32_t seg = 0, b_grid;
    if (T.order == 0) {
        for (uint32_t k = 1; k < T.S; k++) if (blockIdx.x >= T.blk0[k]) seg = k;
        b_grid = blockIdx.x - T.blk0[seg];
    } else {
        const uint32_t b = blockIdx.x;
        if (b < T.S) { seg = b; b_grid = 0; }
        else if (b < 2 * T.S) { seg = b - T.S; b_grid = 1; }
        else if (b < T.sv0[0]) {
            for (uint32_t k = 1; k < T.S; k++) if (b >= T.ab0[k]) seg = k;
            b_grid = 2 + (b - T.ab0[seg]);
        } else {
            for (uint32_t k = 1; k < T.S; k++) if (b >= T.sv0[k]) seg = k;
            b_grid = 2 + call_append_blocks(T.seg[seg]) + (b - T.sv0[seg]);
        }
    }
    const CallEnv X{E.step_lines + (size_t)seg * 1024, E.step_tickets + (size_t)seg * 32, E.step_hash + (size_t)seg * 2 * 1024};
    call_block<true>(E, X, T.seg[seg], push_mask, rmask, b_grid, sq, l, seg, T.S, &T);
}

/* ---- the term fence ---------------------------------------------------------------------------- */
/* The reference fences a deposed leader in the RECEIVER's NIC: a server that votes for, or hears from, a
 * leader of a newer term resets its LOG QPs towards everybody else (rc_revoke_log_access,
 * dare_ibv_rc.c:2156-2243), the old leader's WRITEs bounce (IBV_WC_RETRY_EXC_ERR).  Peer-mapped HBM has
 * no receiver that could refuse a store, so the check sits in front of the writer's launch, on the
 * device: k_fence_check compares the SID in every pushed follower's control block (written by its own
 * process, or by the winner of an election through the mapping: k_elect / k_set_roles) with the
 * leader's; a follower that has moved on to a newer term raises APUS_ST_TERM_FENCE and the fence word
 * (status[2]), and every workgroup of the launches behind it leaves before its first store -- the
 * deposed leader's batch lands nowhere, its own log included (the reference would still append
 * locally; those entries could never commit and are cut by the next leader's log adjustment).
 * Launched only where another process can change a follower's term (peer-mapped groups,
 * APUS_F_TERM_FENCE); the unfenced kernels are the same code without the load. */
__global__ __launch_bounds__(64) void k_fence_check(const EngDev E, uint32_t push_mask)
{
    const uint32_t i = threadIdx.x;
    bool newer = false;
    if (i < APUS_DEV_MAX_SERVERS && ((push_mask >> i) & 1u) && E.rep[i].ring && E.leader < APUS_DEV_MAX_SERVERS)
        newer = (E.rep[i].hdr[H_SID] >> 9) > (E.rep[E.leader].hdr[H_SID] >> 9);
    if (__ballot(newer) && i == 0) { atomicOr(E.status, 1u << 2); E.status[FENCE_WORD] = 1; }
}

__global__ APUS_CALL_BOUNDS void k_call_fenced(const EngDev E_arg, const CallArgs A, uint32_t push_mask, uint32_t rmask)
{
    if (E_arg.status[FENCE_WORD]) return;
    __shared__ EngDev E_s;
    stage_engine(E_s);
    const EngDev &E = E_s;
    __shared__ SeqLds sq;
    __shared__ CallLds l;
    const CallEnv X = APUS_ENV_OF(E);
    call_block<false>(E, X, A, push_mask, rmask, blockIdx.x, sq, l, 0, 1);
}

__global__ APUS_CALL_BOUNDS void k_step_fenced(const EngDev E_arg, const StepTable T, uint32_t push_mask, uint32_t rmask)
{
    if (E_arg.status[FENCE_WORD]) return;
    __shared__ EngDev E_s;
    stage_engine(E_s);
    const EngDev &E = E_s;
    __shared__ SeqLds sq;
    __shared__ CallLds l;
    uint32_t seg = 0, b_grid;
    if (T.order == 0) {
        for (uint32_t k = 1; k < T.S; k++) if (blockIdx.x >= T.blk0[k]) seg = k;
        b_grid = blockIdx.x - T.blk0[seg];
    } else {
        const uint32_t b = blockIdx.x;
        if (b < T.S) { seg = b; b_grid = 0; }
        else if (b < 2 * T.S) { seg = b - T.S; b_grid = 1; }
        else if (b < T.sv0[0]) {
            for (uint32_t k = 1; k < T.S; k++) if (b >= T.ab0[k]) seg = k;
            b_grid = 2 + (b - T.ab0[seg]);
        } else {
            for (uint32_t k = 1; k < T.S; k++) if (b >= T.sv0[k]) seg = k;
            b_grid = 2 + call_append_blocks(T.seg[seg]) + (b - T.sv0[seg]);
        }
    }
    const CallEnv X{E.step_lines + (size_t)seg * 1024, E.step_tickets + (size_t)seg * 32, E.step_hash + (size_t)seg * 2 * 1024};
    call_block<true>(E, X, T.seg[seg], push_mask, rmask, b_grid, sq, l, seg, T.S, &T);
}

/* READ the apply offset of peer i for the next prune tick (rc_get_remote_apply_offsets,
 * dare_ibv_rc.c:1970-2034); one lane per peer, s_lh = the leader's control block
 * as it was before the tick */
__device__ static inline void sample_apply_offsets(const EngDev &E, const uint64_t *s_lh, uint32_t sample_mask, uint32_t i,
                                                   const uint64_t *staged_apply)
{
    uint64_t *hdr = E.rep[E.leader].hdr;
    const uint32_t bitmask = (uint32_t)s_lh[H_CID_BITMASK];
    if (i >= E.group_size) return;
    if (i == E.leader || !((bitmask >> i) & 1u)) hdr[H_APPLY_OFFSETS + i] = s_lh[H_APPLY];
    else if ((sample_mask >> i) & 1u) hdr[H_APPLY_OFFSETS + i] = staged_apply ? *staged_apply : E.rep[i].hdr[H_APPLY];
}

/* One thread: the leader appends at most one control entry (log_append_entry,
 * dare_log.h:466-558; a 64-byte entry never hits wrap case 2).
 *   mode 0: <type, d0, d1>       mode 2: nothing
 *   mode 1: log_pruning (dare_server.c:1996-2067): decide from the sampled apply offsets
 *           whether the head moves and append <HEAD, head> if so
 * s_lh is an LDS copy of the leader's control block; returns the call's SeqOut. */
/* FX = false: decide and account only (the LDS copy s_lh is updated, nothing is stored to HBM) --
 * what every append block of k_call computes for itself; FX = true: the sequencer, with all effects */
template <bool FX>
__device__ static inline SeqOut control_append(const EngDev &E, int mode, uint32_t type, uint64_t d0, uint64_t d1,
                                               uint32_t push_mask, uint64_t *s_lh, uint64_t rec_base, uint32_t ack_mask,
                                               bool apply_now, bool note_head_slot, uint64_t req_id, uint32_t clt_id)
{
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *hdr = Ld.hdr;
    const uint64_t L = E.log_len;
    const uint64_t end = s_lh[H_END];
    uint64_t head = s_lh[H_HEAD];
    bool do_append = (mode != 2);
    if (mode == 1) {
        const uint32_t size = E.group_size;
        const uint32_t bitmask = (uint32_t)s_lh[H_CID_BITMASK];
        uint64_t min_off = s_lh[H_APPLY];
        for (uint32_t i = 0; i < size; i++) {
            if (!((bitmask >> i) & 1u)) { s_lh[H_APPLY_OFFSETS + i] = s_lh[H_APPLY]; if (FX) hdr[H_APPLY_OFFSETS + i] = s_lh[H_APPLY]; }
            if (apus_is_larger(end, L, min_off, s_lh[H_APPLY_OFFSETS + i])) min_off = s_lh[H_APPLY_OFFSETS + i];
        }
        if (apus_end_distance(end, L, min_off) == 0) min_off = s_lh[H_TAIL];   /* leave one entry, :2038-2041 */
        do_append = apus_is_larger(end, L, min_off, head) && !s_lh[H_PREV_HEAD];
        if (do_append) { if (FX) hdr[H_HEAD] = min_off; s_lh[H_HEAD] = min_off; head = min_off; d0 = min_off; d1 = 0; type = 3; }
    }
    SeqOut s;
    s.e0 = end; s.idx0 = s_lh[H_LAST_IDX] + 1; s.w = 0; s.n_end0 = s_lh[H_N_END];
    s.term = s_lh[H_SID] >> 9; s.kstar = -1; s.estar = -1; s.stale = 0; s.n = 0; s.head_round = 0;
    s.first_fail = ~0ull; s.commit_before = s_lh[H_COMMIT]; s.n_commit_before = s_lh[H_N_COMMIT];
    s.vis = 0; s.scan_lo = 0; s.fuse_mask = 0; s.tail_needed = 1; s.fast = 0; s.pad1 = 0; s.rec_base = rec_base;
    for (uint32_t f = 0; f < APUS_DEV_MAX_SERVERS; f++) s.np[f] = ~0ull;
    if (do_append && end == head && end != L) { if (FX) set_status(E, 1u << 1); do_append = false; }
    if (do_append) {
        const uint64_t idx = (end == L) ? 1 : s_lh[H_LAST_IDX] + 1;         /* dare_log.h:486-488 */
        const uint64_t pos = (end == L || L - end < APUS_HDR) ? 0 : end;   /* log_add_new_entry, :213-221 */
        if (type != 3) { if (FX) hdr[H_PREV_HEAD] = 0; s_lh[H_PREV_HEAD] = 0; } else if (mode == 1) { if (FX) hdr[H_PREV_HEAD] = 1; s_lh[H_PREV_HEAD] = 1; }
        if (FX && type == 2) hdr[H_CID_BITMASK] = (uint32_t)(d1 >> 32);     /* the leader's own cid follows its CONFIG entries */
        const uint64_t term = s.term;
        const uint64_t slot = s.n_end0;
        const uint32_t di = (uint32_t)slot & E.dir_mask;
        const uint4 h0 = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)term, (uint32_t)(term >> 32));
        /* req_id and clt_id are 0 but for the CONFIG entry that admits a joining server (its request id and LID,
         * handle_server_join_request dare_ibv_ud.c:1057-1060) */
        const uint4 h1 = make_uint4((uint32_t)req_id, (uint32_t)(req_id >> 32), (clt_id & 0xFFFFu) | (type << 16) | ((uint32_t)E.leader << 24), 0);
        const uint4 h3 = make_uint4((uint32_t)d0, (uint32_t)(d0 >> 32), (uint32_t)d1, (uint32_t)(d1 >> 32));
        for (uint32_t m = FX ? (push_mask | (1u << E.leader)) : 0u; m; m &= m - 1) {
            const int t = __builtin_ctz(m);
            uint8_t *rg = E.rep[t].ring;
            /* followers in ack_mask persist + ACK with the push: own reply byte in their ring, all of them in the leader's */
            const ReplyWords rw = apus_reply_words((uint32_t)t == E.leader ? ack_mask : (ack_mask & (1u << t)));
            st16u(rg + pos, h0); st16u(rg + pos + 16, make_uint4(h1.x, h1.y, h1.z, rw.w28));
            st16u(rg + pos + 32, make_uint4(rw.x32, rw.y36, rw.z40, 0)); st16u(rg + pos + 48, h3);
            E.rep[t].dir_off[di] = pos; E.rep[t].dir_len[di] = APUS_HDR | ((uint32_t)E.leader << 24);
            if (apply_now) {
                /* everybody is in step: the entry is committed as it lands, the replicas apply it
                 * right away (a control entry: no upcall; a follower notes a committed <HEAD>) */
                uint4 *rp = (uint4 *)&E.rep[t].apply[di];
                rp[0] = make_uint4((uint32_t)slot, (uint32_t)(slot >> 32), (uint32_t)pos, (uint32_t)(pos >> 32));
                rp[1] = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), 0, type << 16);
                /* (k_call's bookkeeper derives the adoption itself: note_head_slot = false) */
                if (type == 3 && (uint32_t)t != E.leader && note_head_slot)
                    atomicMax((unsigned long long *)&E.rep[t].hdr[H_HEAD_SLOT], (unsigned long long)(slot + 1));
            }
        }
        if (FX) __hip_atomic_store(&Ld.ack[di], ack_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (FX) hdr[H_TAIL] = pos;               s_lh[H_TAIL] = pos;
        if (FX) hdr[H_END] = pos + APUS_HDR;     s_lh[H_END] = pos + APUS_HDR;
        if (FX) hdr[H_N_END] = slot + 1;         s_lh[H_N_END] = slot + 1;
        if (FX) hdr[H_LAST_IDX] = idx;           s_lh[H_LAST_IDX] = idx;
        if (FX) hdr[H_OLD_END] = pos + APUS_HDR; s_lh[H_OLD_END] = pos + APUS_HDR;
        if (FX) hdr[H_N_PERSIST] = slot + 1;     s_lh[H_N_PERSIST] = slot + 1;
        if (FX) hdr[H_STORE_COUNT] = s_lh[H_STORE_COUNT] + 1;
        s_lh[H_STORE_COUNT] += 1;
        s.n = 1;
        if (FX && rec_base < E.rec_cap) E.rec_end[rec_base] = pos + APUS_HDR;
    }
    return s;
}

/* ------------------------------------------------------------------------- */
/* k_control_round: ONE workgroup does a whole polling() pass that carries at most
 * one control entry -- used for the prune tick, the new leader's blank CONFIG
 * entry, removed-server CONFIG entries and quiesce, where the wide pipeline's six
 * launches would be pure overhead.
 *   mode 0: append <type, d0, d1> (log_append_entry, dare_log.h:466-558; 64-byte
 *           entries never hit wrap case 2)
 *   mode 1: log_pruning (dare_server.c:1996-2067): decide from the sampled apply
 *           offsets whether the head moves, append <HEAD, head> if so, re-sample the
 *           apply offsets (rc_get_remote_apply_offsets, dare_ibv_rc.c:1970)
 *   mode 2: no append (quiesce)
 *   mode 3: append <type, d0, d1> and stop: the entry joins the next pass (the new leader's
 *           blank CONFIG when check_failure_count appends a removal behind it)
 * then followers persist + ACK, the ACK scan, apply and the bookkeeping.          */
__global__ __launch_bounds__(256) void k_control_round(const EngDev E, int mode_flags, uint32_t type,
                                                       uint64_t d0, uint64_t d1, uint32_t push_mask,
                                                       uint32_t sample_mask, uint64_t req_id, uint32_t clt_id)
{
    __shared__ uint32_t s_ack[256];
    __shared__ unsigned long long s_acc[2];
    const uint32_t tid = threadIdx.x;
    /* flags above the mode: 16 = the pass leaves no per-round record (the CONFIG entries of a JOIN: the
     * whole join is one record, as in the schedule the oracle is pinned on), 32 = record end / commit as
     * they are after this pass even though it appended nothing (the pass that closes a JOIN) */
    const int mode = mode_flags & 7;
    if (E.status[FENCE_WORD]) return;                 /* the term fence (k_fence_check) */
    const uint64_t rec0 = *E.rec_count;
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *hdr = Ld.hdr;
    const uint64_t L = E.log_len;

    /* followers that lag (RELEASE without a wide catch-up, hidden exact-fit round) */
    {
        const uint64_t e0 = hdr[H_END], n0 = hdr[H_N_END];
        for (uint32_t m = push_mask; m; m &= m - 1) catchup_range(E, __builtin_ctz(m), e0, n0, tid, blockDim.x);
        __syncthreads();
        if (tid == 0)
            for (uint32_t m = push_mask; m; m &= m - 1) {
                uint64_t *fh = E.rep[__builtin_ctz(m)].hdr;
                if (fh[H_N_END] < n0 && e0 != L) { fh[H_END] = e0; fh[H_N_END] = n0; }
            }
    }

    /* the leader's control block goes through LDS: one round trip instead of a chain */
    __shared__ uint64_t s_lh[64];
    if (tid < 64) s_lh[tid] = hdr[tid];
    __syncthreads();
    if (tid == 0) *E.seq = control_append<true>(E, mode, type, d0, d1, push_mask, s_lh, *E.rec_count, 0, false, true, req_id, clt_id);
    if (mode == 1 && tid >= 64 && tid < 64 + APUS_DEV_MAX_SERVERS) sample_apply_offsets(E, s_lh, sample_mask, tid - 64, nullptr);
    __syncthreads();
    if (mode == 3) return;     /* the followers' end words follow in the next pass's catch-up prelude */

    const uint64_t vis = visible_slots(E, hdr, 0, 0);
    for (uint32_t m = push_mask; m; m &= m - 1) persist_ack_range(E, __builtin_ctz(m), vis, tid, blockDim.x);
    __threadfence_block();
    __syncthreads();
    commit_scan(E, hdr[H_N_COMMIT], vis, 0, blockDim.x, s_ack);
    __syncthreads();
    const uint64_t cs = commit_slot(E, vis);
    for (uint32_t m = push_mask | (1u << E.leader); m; m &= m - 1)
        apply_range(E, __builtin_ctz(m), E.rep[__builtin_ctz(m)].hdr[H_N_APPLY], cs, 0, blockDim.x, s_acc);
    __syncthreads();
    finish_call(E, 0, 0, mode == 2 ? 2 : 1, push_mask);
    if (tid == 0) {                                   /* (finish_call's leader words are thread 0's stores too) */
        if (mode_flags & 16) *E.rec_count = rec0;
        if (mode_flags & 32) {
            if (rec0 < E.rec_cap) { E.rec_end[rec0] = hdr[H_END]; E.rec_commit[rec0] = hdr[H_COMMIT]; }
            *E.rec_count = rec0 + 1;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Multi-process groups (one replica per GPU): the byte range and the directory
 * slots travel by RCCL point-to-point between processes (apus_amd/distributed.py);
 * these kernels are the device halves on either side of that exchange.           */

/* follower: the entries of slots [n_persist, vis) have landed in ring + directory:
 * persist_new_entries + the local half of rc_send_entries_reply (own reply byte);
 * the ACK itself goes back to the leader as one cumulative slot number.           */
__global__ __launch_bounds__(256) void k_mp_ingest(const EngDev E, uint32_t f, uint64_t vis)
{
    const RepDev &Fd = E.rep[f];
    const uint64_t from = Fd.hdr[H_N_PERSIST];
    const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < vis; s += nth) {
        const uint32_t di = (uint32_t)s & E.dir_mask;
        Fd.ring[Fd.dir_off[di] + 28 + f] = 1;
    }
}

__global__ void k_mp_ingest_fin(const EngDev E, uint32_t f, uint64_t vis)
{
    if (threadIdx.x || blockIdx.x) return;
    uint64_t *fh = E.rep[f].hdr;
    const uint64_t np = fh[H_N_PERSIST];
    if (vis <= np) return;
    const uint32_t dl = (uint32_t)(vis - 1) & E.dir_mask;
    const uint64_t end = E.rep[f].dir_off[dl] + (E.rep[f].dir_len[dl] & 0xFFFFFFu);
    fh[H_STORE_COUNT] += vis - np;
    fh[H_END] = end; fh[H_OLD_END] = end; fh[H_N_END] = vis; fh[H_N_PERSIST] = vis;
}

/* leader: follower f acknowledged every entry below `upto` (R3 for a whole range):
 * reply byte in the leader's ring + ACK bit in the slot word                       */
__global__ __launch_bounds__(256) void k_mp_ack_merge(const EngDev E, uint32_t f, uint64_t from, uint64_t upto)
{
    const RepDev &Ld = E.rep[E.leader];
    const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < upto; s += nth) {
        const uint32_t di = (uint32_t)s & E.dir_mask;
        Ld.ring[Ld.dir_off[di] + 28 + f] = 1;
        atomicOr(&Ld.ack[di], 1u << f);
    }
}

/* follower: the leader's commit (R4) reached slot cs: apply_committed_entries */
__global__ __launch_bounds__(256) void k_mp_apply(const EngDev E, uint32_t f, uint64_t cs)
{
    __shared__ unsigned long long s_acc[2];
    apply_range(E, (int)f, E.rep[f].hdr[H_N_APPLY], cs, (uint64_t)blockIdx.x * blockDim.x, (uint64_t)gridDim.x * blockDim.x, s_acc);
}

__global__ void k_mp_apply_fin(const EngDev E, uint32_t f, uint64_t cs)
{
    if (threadIdx.x || blockIdx.x) return;
    const RepDev &Fd = E.rep[f];
    uint64_t *fh = Fd.hdr;
    if (cs > fh[H_N_PERSIST]) cs = fh[H_N_PERSIST];
    const uint64_t coff = (cs == fh[H_N_END]) ? fh[H_END] : Fd.dir_off[(uint32_t)cs & E.dir_mask];
    if (cs > fh[H_N_COMMIT]) { fh[H_COMMIT] = coff; fh[H_N_COMMIT] = cs; }
    if (cs > fh[H_N_APPLY]) { fh[H_APPLY] = coff; fh[H_N_APPLY] = cs; }
    const uint64_t hs = fh[H_HEAD_SLOT];
    if (hs) {
        const uint64_t hv = ld8u(Fd.ring + Fd.dir_off[(uint32_t)(hs - 1) & E.dir_mask] + 48);
        if (apus_is_larger(fh[H_END], E.log_len, hv, fh[H_HEAD])) fh[H_HEAD] = hv;
        fh[H_HEAD_SLOT] = 0;
    }
}

/* k_reset: log_new() (dare_log.h:120-136) without touching the ring bytes that
 * were never made visible; the host memsets rings separately when asked to.   */
__global__ void k_reset(const EngDev E)
{
    const int p = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (p == 0) {
        /* the engine's own scratch (not gated on hosting replica 0: a process of a multi-process
         * group hosts only its own replica) */
        *E.rec_count = 0; *E.status = 0; E.status[1] = 0; E.status[2] = 0;
        for (int i = 0; i < 8; i++) E.ticket[i] = 0;
        for (int i = 0; i < 32; i++) { E.tick_lines[i * 32] = 0; E.tick_lines[i * 32 + 1] = 0; E.tick_lines[i * 32 + 2] = 0; }
        for (int i = 0; i < 32; i++) { E.step_epoch[i * 32] = 0; E.step_seq_done[i * 32] = 0; }
        for (int i = 0; i < APUS_STEP_SEGS * REC_WORDS; i++) E.step_rec[i] = 0;
        for (int i = 0; i < APUS_STEP_SEGS * 1024; i++) E.step_lines[i] = 0;
        for (int i = 0; i < APUS_STEP_SEGS * 32; i++) E.step_tickets[i] = 0;
    }
    if (!E.rep[p].ring) return;
    uint64_t *h = E.rep[p].hdr;
    /* every word is stored exactly once: the control block is uncached memory, and two stores of one thread to
     * one word are not something to lean on there */
    for (int i = 0; i < H_WORDS; i++) {
        uint64_t v = 0;
        if (i == H_LEN || i == H_END || i == H_TAIL || i == H_OLD_END) v = E.log_len;
        else if (i == H_SID) v = (uint64_t)p;
        else if (i == H_CID_BITMASK) v = (1u << E.group_size) - 1;
        h[i] = v;
    }
}
