/*
 * apus_engine.hip -- host side of libapus_gpu.so: owns the HBM-resident state of
 * the local replicas and turns the C ABI of include/apus_gpu.h into kernel
 * launches on one HIP stream.  Every hot-path call is asynchronous and free of
 * host read-backs, so a sequence of calls can be captured into a hipGraph.
 *
 * There is no CPU fallback: every entry point needs a gfx950 device.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#include "../../include/apus_gpu.h"
#include "apus_kernels.h"
#include "apus_persistent.h"
#include "apus_replica.h"
#include "apus_quirks.h"
#include "apus_selftest.h"
#include "apus_members.h"
#include <pthread.h>
#include <sched.h>
#include <dirent.h>
#include <time.h>

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { \
    fprintf(stderr, "[apus_gpu] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    return APUS_E_HIP; } } while (0)

static_assert(sizeof(apus_apply_t) == 32 && sizeof(apus_apply_rec) == 32, "apply record is 32 bytes");
static_assert(sizeof(apus_req_t) == 24, "request record is 24 bytes");
static_assert(sizeof(ReqDev) == 16, "device request descriptor is 16 bytes");
static_assert(sizeof(EngDev) <= 3072, "EngDev travels as a kernel argument");

struct TimedLaunch { hipEvent_t a, b; };

struct apus_engine {
    apus_cfg_t cfg;
    EngDev d;                       /* host mirror, passed by value to every kernel */
    hipStream_t stream;
    bool own_stream;
    uint32_t dir_cap;
    uint32_t local_mask;            /* replicas hosted here, or peer-mapped (imported) */
    uint32_t imported_mask;         /* the peer-mapped ones: memory owned by another process */
    std::vector<void *> ipc_ptrs;    /* hipIpcOpenMemHandle results, closed at destroy */
    uint32_t reachable;             /* peers the leader can post to (trace KILL/HOLD/RELEASE) */
    bool lag_possible;              /* a follower may be far behind: run the wide catch-up first */
    /* APUS_FEED_PROF=1: where a producer's time goes in apus_gpu_rep_submit, by phase, summed over the producers (TSC ticks):
     * [0] blocks [1] slots [2] reserve (the fetch-and-add + waiting for room in the ring) [3] payload + descriptor stores (write-combined,
     * through the BAR) [4] the fence behind them [5] the slots' publish words [6] the windows' accounts + words [7] the call's last fence */
    uint64_t feed_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int feed_prof_on = -1;
    uint64_t st_atomic_misses = 0;  /* apus_gpu_selftest, pushing side: system-scope atomics that had not landed in front of the store behind them */
    bool tick_pending;              /* a prune tick waits to be fused into the next batch's sequencer */
    uint64_t max_rounds;
    /* staging */
    void *d_req, *d_req_len, *d_arena, *d_round_first, *d_round_prefix, *d_round_change;
    uint64_t n_reqs, n_rounds_staged;
    std::vector<uint32_t> h_round_first;
    std::vector<uint64_t> h_round_prefix;      /* byte prefix of the staged rounds */
    bool batching;                             /* apus_gpu_batch_begin .. _end */
    uint64_t *d_quirk;                         /* APUS_F_REF_QUIRKS: k_ref_quirk_wrap's word between passes */
    CfgJournal *d_cfgj;                        /* every CONFIG entry a leader appended, every vote (apus_members.h) */
    MemberView mv[APUS_MAX_SERVERS];           /* what each server's own configuration is derived from */
    bool cfg_appended;
    uint64_t sp_units;                         /* APUS_SP_UNITS (default 768), see call_args */
    uint32_t gp_rounds;                        /* APUS_GP_ROUNDS (default 4; 1 = one workgroup per round), see call_args */
    uint64_t gp_max_units;                     /* APUS_GP_MAX_UNITS: largest mean round (16-byte units) that is grouped */
    uint32_t step_slots;                       /* workgroups of k_step the device holds at once (occupancy x CUs) */
    /* admission (log_append_entry refuses a full log, dare_log.h:168,492-495): a lower bound of the free
     * bytes of the leader's ring, refreshed from the control block only when it does not cover a batch */
    uint64_t free_lb;
    uint64_t stage_max_T;                      /* largest entry (header + payload) of the staged requests */
    uint32_t host_status;                      /* status bits raised on the host side (APUS_ST_LOG_FULL) */
    /* election (apus_gpu_elect): servers that refused their vote keep their log closed to the winner
     * (no vote ACK, hb_receive_cb dare_server.c:903-910) until the next election; those that granted it
     * get their logs adjusted by the new leader's first pass (log_adjustment, dare_ibv_rc.c:1292-1451) */
    uint32_t no_access, adjust_mask;
    uint64_t *d_elect;                         /* k_elect's verdict */
    uint64_t cid_epoch;                        /* config.cid.epoch: one more with every group extension (apus_gpu_join) */
    struct BatchSeg { CallArgs a; uint32_t blocks; uint64_t bytes; bool lean; };
    std::vector<BatchSeg> batch;               /* recorded calls: arguments, blocks, bytes they append */
    /* graphs */
    std::vector<hipGraphExec_t> graphs;
    bool capturing;
    /* timing of the dominant kernel (k_append_push) */
    bool timing;
    std::vector<TimedLaunch> timed;
    size_t timed_used;
    std::vector<void *> allocs;
    /* live submission path (apus_gpu_submit): pinned host staging + device buffers */
    uint8_t *h_live;                /* pinned: [ReqDev x LIVE_REQS][u16 x LIVE_REQS][u32 x (LIVE_REQS+1)][u32 x LIVE_REQS][arena] */
    uint8_t *d_live;
    hipEvent_t live_copied;
    bool live_pending;
    uint64_t live_r0, live_R, live_n, live_bytes;   /* rounds appended by apus_gpu_append_live, not yet committed */
    /* persistent consensus kernel (live / latency path) */
    PersistHost *ph;                /* pinned, coherent */
    PersistHost *ph_dev;            /* device view of the same memory */
    PersistDev *pd;
    hipStream_t pstream;
    bool p_running;
    uint64_t p_ev_tail, p_req_tail, p_arena_pos;
    uint64_t p_req_end[P_EV_CAP];   /* cumulative request count after each published event */
    /* replica kernels (apus_gpu_rep_*): every hosted replica runs its own workgroups */
    RepHost *rh, *rh_dev;           /* pinned, coherent: the leader's progress words */
    RepReq *rq;                     /* the command + request rings the host fills: device memory behind the BAR, or pinned (rq_bar) */
    RepReq *rq_dev;
    bool rq_bar;
    RepLead *rl;                    /* leader-local hand-off state */
    RepFollow *rfs[APUS_MAX_SERVERS];
    RepFHost *rfh[APUS_MAX_SERVERS], *rfh_dev[APUS_MAX_SERVERS];   /* pinned: a hosted follower's progress / stop words */
    uint64_t r_consumer[APUS_MAX_SERVERS], r_replayed[APUS_MAX_SERVERS];   /* a host consumer of a hosted follower's apply stream (apus_gpu_rep_follower_replayed) */
    uint8_t *ss_buf; uint64_t ss_cap;       /* apus_gpu_store_stream's scratch, kept between calls */
    hipStream_t rstream;            /* the run's one resident launch (k_replica / k_replica_leader / k_replica_follower) */
    /* apus_gpu_fence_replica: the APUS_FENCE_PAIRS (log ring, mailbox) pairs a replica moves through, fence f lives in pair
     * f % APUS_FENCE_PAIRS; the ones it has LEFT stay allocated (a deposed leader's mapping still leads there: its stores hit
     * memory that exists and nobody reads) */
    uint8_t *pair_ring[APUS_MAX_SERVERS][APUS_FENCE_PAIRS];     /* hosted: allocations (pair 0 = the one the replica was created in); */
    RepBox *pair_box[APUS_MAX_SERVERS][APUS_FENCE_PAIRS];       /* mapped: the peer's, all of them open from apus_gpu_import_replica on */
    bool pairs_ready[APUS_MAX_SERVERS];
    apus_ipc_replica_t ipc_cache[APUS_MAX_SERVERS];             /* hosted: the handles, taken once (ipc_cached) */
    bool ipc_cached[APUS_MAX_SERVERS];
    int ring_alloc;                 /* 0 hipMalloc, 1 fine-grained, 2 uncached (APUS_RING_ALLOC) */
    uint32_t fences[APUS_MAX_SERVERS];          /* hosted: fences so far; imported: the exporter's count as mapped here */
    hipEvent_t rev0, rev1;          /* around the last resident launch (apus_gpu_rep_launch_ms) */
    bool rev_valid;
    bool r_running, r_lead;         /* a launch is resident; it carries the leader's workgroups */
    uint32_t r_follow_mask;
    uint32_t r_test_skip;           /* tests: followers whose workgroups are NOT launched although they are pushed to (a dead process) */
    uint64_t r_fruns[APUS_MAX_SERVERS];          /* leader: every pushed follower's f_runs when the run began (apus_gpu_rep_park waits for the next) */
    uint32_t r_push_mask, r_cand_mask;        /* the last run started here as the leader: who gets its rounds, who could be reached */
    uint64_t r_slot_tail, r_arena_tail, r_cmd_tail;   /* producer side of the pinned rings (under r_lock) */
    uint64_t r_done_seen = 0;       /* rh->slots_done as a producer last read it (rep_reserve_inline) */
    uint64_t *r_slot_aend;          /* [RQ_CAP] logical arena position behind every slot's payload */
    uint32_t *r_win_cnt, *r_win_len;/* [RQ_CAP / 64] per aligned window of 64 request slots: how many are published, and the one length they
                                     * all have (rep_win_note: whoever brings a window to 64 writes its word, RepReq.ready_win) */
    pthread_spinlock_t r_lock;
    bool r_lock_init;
};

#define LIVE_REQS   4096u
#define LIVE_ARENA  (16u << 20)
#define LIVE_OFF_LEN    (sizeof(ReqDev) * LIVE_REQS)
#define LIVE_OFF_RF     (LIVE_OFF_LEN + sizeof(uint16_t) * LIVE_REQS)
#define LIVE_OFF_RB     (LIVE_OFF_RF + sizeof(uint32_t) * (LIVE_REQS + 1))
#define LIVE_OFF_PFX    ((LIVE_OFF_RB + sizeof(uint32_t) * LIVE_REQS + 7) & ~(size_t)7)
#define LIVE_OFF_ARENA  ((LIVE_OFF_PFX + sizeof(uint64_t) * (LIVE_REQS + 1) + 31) & ~(size_t)15)
#define LIVE_BYTES      (LIVE_OFF_ARENA + LIVE_ARENA + 64)

static apus_engine *g_engine = nullptr;
static int flush_tick(apus_engine *e);
extern "C" int apus_gpu_persist_stop(apus_engine_t *e);
extern "C" int apus_gpu_rep_park(apus_engine_t *e);

static int flush_live(apus_engine *e);        /* a staged live batch runs before anything else touches the engine */

template <typename T>
static int dev_alloc(apus_engine *e, T **out, size_t bytes, bool zero = true, unsigned ext_flags = 0)
{
    void *p = nullptr;
    /* Whole 2 MiB blocks for everything that is not tiny.  The runtime carves allocations that are not a multiple of its
     * 2 MiB block out of shared blocks (a ring of 1 MiB + 4 KiB came back 28 KiB into a block), and hipIpcGetMemHandle on such
     * a fragment fails with "invalid argument" once the process has exported, freed and re-allocated a few of them: round 5's
     * 50-fold soak of the cross-process JOIN traces died at the fifth engine of one process group, on round 4's tree as well
     * (tools/gpu_soak.sh; the suite's three repetitions never got there).  A buffer that owns its blocks exports every time. */
    const size_t asked = bytes;
    if (bytes > (64u << 10)) bytes = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    if (ext_flags) { if (hipExtMallocWithFlags(&p, bytes, ext_flags) != hipSuccess) return APUS_E_NOMEM; }
    else if (hipMalloc(&p, bytes) != hipSuccess) return APUS_E_NOMEM;
    /* (on the engine's stream: a hipMemset on the null stream may still be in flight when the first kernel of the
     * non-blocking engine stream runs -- k_reset's words were seen zeroed again by a late memset) */
    if (zero && hipMemsetAsync(p, 0, asked, e->stream) != hipSuccess) return APUS_E_HIP;
    e->allocs.push_back(p);
    *out = (T *)p;
    return 0;
}

static uint32_t pow2_at_least(uint64_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

static inline uint32_t sync_mask(const apus_engine *e)
{
    /* local followers the leader can currently post to */
    if (e->d.leader >= APUS_MAX_SERVERS) return 0;
    return e->local_mask & e->reachable & ~e->no_access & ~(1u << e->d.leader);
}
static inline int popc(uint32_t v) { return __builtin_popcount(v); }
#define R_WIN_UNSET 0xFFFFFFFFu      /* rep_win_note: the window's account has no length yet / lengths that differ */
#define R_WIN_MIXED 0xFFFFFFFEu

/* the term fence (k_fence_check): in front of every launch that stores into followers, where another
 * process can move a follower to a newer term behind this leader's back */
static inline bool fence_on(const apus_engine *e) { return e->imported_mask != 0 || (e->cfg.flags & 2u); }
/* APUS_F_REF_QUIRKS: behind every pass, the reference's state at a commit pointer parked on a wrap position (apus_quirks.h) */
static inline bool ref_quirks(const apus_engine *e) { return (e->cfg.flags & APUS_F_REF_QUIRKS) != 0; }
#define QUIRK_PASS(e, kind) do { if (ref_quirks(e)) hipLaunchKernelGGL(k_ref_quirk_wrap, dim3(1), dim3(64), 0, (e)->stream, (e)->d, (e)->d_quirk, (int)(kind)); } while (0)
#define FENCE_CHECK(e, fm) do { if (fence_on(e) && (fm)) hipLaunchKernelGGL(k_fence_check, dim3(1), dim3(64), 0, (e)->stream, (e)->d, (fm)); } while (0)

extern "C" int apus_gpu_create(const apus_cfg_t *cfg, apus_engine_t **out)
{
    if (!cfg || !out || cfg->group_size < 1 || cfg->group_size > APUS_MAX_SERVERS ||
        cfg->n_local < 1 || cfg->n_local > cfg->group_size) return APUS_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[apus_gpu] no HIP device: the consensus engine has no CPU path\n");
        return APUS_E_HIP;
    }
    HIPCHK(hipSetDevice(cfg->device));
    apus_engine *e = new apus_engine();
    e->cfg = *cfg;
    memset(&e->d, 0, sizeof e->d);
    e->capturing = false; e->timing = false; e->timed_used = 0; e->lag_possible = false; e->tick_pending = false;
    e->batching = false;
    {   /* tuning knob, read once: 16-byte units of a round one append workgroup takes */
        const char *sp_env = getenv("APUS_SP_UNITS");
        e->sp_units = (sp_env && atoi(sp_env) > 0) ? (uint64_t)atoi(sp_env) : 768;
        /* grouped append (one wavefront per round): on unless APUS_GP_ROUNDS=1; for rounds of up to
         * APUS_GP_MAX_UNITS 16-byte units on average (default 1024 = 16 KiB) */
        const char *gp_env = getenv("APUS_GP_ROUNDS");
        e->gp_rounds = (gp_env && atoi(gp_env) > 0) ? (uint32_t)atoi(gp_env) : APUS_GP;
        const char *gu_env = getenv("APUS_GP_MAX_UNITS");
        e->gp_max_units = (gu_env && atoi(gu_env) > 0) ? (uint64_t)atoi(gu_env) : 1024;
    }
    {
        int per_cu = 0;
        hipDeviceProp_t prop;
        e->step_slots = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_step, 256, 0) == hipSuccess &&
            hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && per_cu > 0)
            e->step_slots = (uint32_t)per_cu * (uint32_t)prop.multiProcessorCount;
        if (getenv("APUS_DEBUG")) fprintf(stderr, "[apus_gpu] k_step: %d workgroups per CU, %u slots\n", per_cu, e->step_slots);
        const char *so_env = getenv("APUS_STEP_ORDER");          /* 0: always segment by segment */
        if (so_env && atoi(so_env) == 0) e->step_slots = 0;
    }
    e->free_lb = 0; e->stage_max_T = APUS_HDR; e->host_status = 0;
    e->no_access = 0; e->adjust_mask = 0; e->d_elect = nullptr; e->cid_epoch = 0;
    e->n_reqs = 0; e->n_rounds_staged = 0;
    e->d_req = e->d_req_len = e->d_arena = e->d_round_first = e->d_round_prefix = e->d_round_change = nullptr;
    e->h_live = nullptr; e->d_live = nullptr; e->live_pending = false; e->live_copied = nullptr;
    e->live_r0 = e->live_R = e->live_n = 0;
    e->ph = e->ph_dev = nullptr; e->pd = nullptr; e->pstream = nullptr; e->p_running = false;
    e->p_ev_tail = e->p_req_tail = e->p_arena_pos = 0;
    e->rh = e->rh_dev = nullptr; e->rq = e->rq_dev = nullptr; e->rq_bar = false; e->rl = nullptr; e->rstream = nullptr; e->rev0 = e->rev1 = nullptr; e->rev_valid = false; e->r_running = e->r_lead = false; e->r_follow_mask = 0; e->r_test_skip = 0;
    for (auto &f : e->rfs) f = nullptr;
    for (auto &f : e->fences) f = 0;
    e->r_slot_tail = e->r_arena_tail = e->r_cmd_tail = 0; e->r_slot_aend = nullptr; e->r_win_cnt = e->r_win_len = nullptr;
    pthread_spin_init(&e->r_lock, PTHREAD_PROCESS_PRIVATE); e->r_lock_init = true;
    if (cfg->stream) { e->stream = (hipStream_t)cfg->stream; e->own_stream = false; }
    else { HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)); e->own_stream = true; }

    const uint64_t L = cfg->log_len ? cfg->log_len : APUS_LOG_SIZE;
    if (L < 1024 || (L % 64)) { delete e; return APUS_E_ARG; }
    e->d.log_len = L;
    e->d.group_size = cfg->group_size;
    e->d.leader = 0xFFFFFFFFu;
    e->dir_cap = pow2_at_least(L / APUS_ENTRY_HDR < 4096 ? 4096 : L / APUS_ENTRY_HDR);
    e->d.dir_mask = e->dir_cap - 1;
    e->d.flags = cfg->flags;
    if (cfg->flags & APUS_F_REF_QUIRKS) { if (dev_alloc(e, &e->d_quirk, 16)) { delete e; return APUS_E_NOMEM; } }
    if (dev_alloc(e, &e->d_cfgj, sizeof(CfgJournal))) { delete e; return APUS_E_NOMEM; }
    e->local_mask = 0; e->imported_mask = 0;
    { const char *ra = getenv("APUS_RING_ALLOC"); e->ring_alloc = (ra && !strcmp(ra, "finegrained")) ? 1 : (ra && !strcmp(ra, "uncached")) ? 2 : 0; }
    e->reachable = (1u << cfg->group_size) - 1;
    e->d.reachable = e->reachable;
    int rc = 0;
    for (uint32_t k = 0; k < cfg->n_local; k++) {
        const uint32_t i = cfg->local_ids[k];
        if (i >= cfg->group_size || (e->local_mask >> i) & 1u) { rc = APUS_E_ARG; break; }
        e->local_mask |= 1u << i;
        RepDev &r = e->d.rep[i];
        r.idx = i;
        /* the log ring: ordinary device memory -- unless first contact between two devices showed that a peer's stores are not
         * what the owner's resident kernel reads (apus_selftest.h; bench.py then sets APUS_RING_ALLOC and starts again) */
        if ((rc = dev_alloc(e, &r.ring, L + 4096, true, e->ring_alloc == 1 ? hipDeviceMallocFinegrained : e->ring_alloc == 2 ? hipDeviceMallocUncached : 0u))) break;
        /* control blocks live in uncached device memory: stores reach memory without a release, so
         * that blocks of a LATER segment of the same launch (k_step) can read them with bypassing loads */
        if ((rc = dev_alloc(e, &r.hdr, sizeof(uint64_t) * 64, true, hipDeviceMallocUncached))) break;
        if ((rc = dev_alloc(e, &r.dir_off, sizeof(uint64_t) * e->dir_cap))) break;
        if ((rc = dev_alloc(e, &r.dir_len, sizeof(uint32_t) * e->dir_cap))) break;
        if ((rc = dev_alloc(e, &r.ack, sizeof(uint32_t) * e->dir_cap))) break;
        if ((rc = dev_alloc(e, &r.apply, sizeof(apus_apply_rec) * (size_t)e->dir_cap))) break;
        /* the replica kernels' mailbox and ACK byte maps: uncached, so that a peer's system-scope stores are
         * what the owner's polls read (and the other way round) */
        if ((rc = dev_alloc(e, &e->d.box[i], sizeof(RepBox), true, hipDeviceMallocUncached))) break;
        if ((rc = dev_alloc(e, &e->d.ackb[i], (size_t)cfg->group_size * e->dir_cap, true, hipDeviceMallocUncached))) break;
    }
    e->max_rounds = 1u << 16;
    e->d.rec_cap = 1ull << 22;
    if (!rc) rc = dev_alloc(e, &e->d.status, 64);
    if (!rc) e->d.ticket = e->d.status + 8;
    if (!rc) rc = dev_alloc(e, &e->d.tick_lines, sizeof(uint32_t) * 32 * 32);
    if (!rc) rc = dev_alloc(e, &e->d.seq, sizeof(SeqOut));
    if (!rc) rc = dev_alloc(e, &e->d.step_lines, sizeof(uint32_t) * APUS_STEP_SEGS * 1024);
    if (!rc) rc = dev_alloc(e, &e->d.step_tickets, sizeof(uint32_t) * APUS_STEP_SEGS * 32);
    if (!rc) rc = dev_alloc(e, &e->d.step_hash, sizeof(uint64_t) * APUS_STEP_SEGS * 2 * 1024);
    if (!rc) rc = dev_alloc(e, &e->d.step_snap, sizeof(uint64_t) * (APUS_STEP_SEGS + 1) * SNAP_STRIDE, true, hipDeviceMallocUncached);
    if (!rc) rc = dev_alloc(e, &e->d.step_rec, sizeof(uint64_t) * APUS_STEP_SEGS * REC_WORDS, true, hipDeviceMallocUncached);
    if (!rc) rc = dev_alloc(e, &e->d.step_epoch, sizeof(uint32_t) * 32 * 32);
    if (!rc) rc = dev_alloc(e, &e->d.step_seq_done, sizeof(uint32_t) * 32 * 32);
#ifdef APUS_TRACE
    if (!rc) rc = dev_alloc(e, &e->d.trace, 8 * 16 * 64);
#else
    e->d.trace = nullptr;
#endif
    if (!rc) rc = dev_alloc(e, &e->d.round_virt, sizeof(uint64_t) * (e->max_rounds + 1));
    if (!rc) rc = dev_alloc(e, &e->d.round_hash, sizeof(uint64_t) * 2 * e->max_rounds);
    if (!rc) rc = dev_alloc(e, &e->d.rec_end, sizeof(uint64_t) * e->d.rec_cap);
    if (!rc) rc = dev_alloc(e, &e->d.rec_commit, sizeof(uint64_t) * e->d.rec_cap);
    if (!rc) rc = dev_alloc(e, &e->d.rec_count, 64, true, hipDeviceMallocUncached);
    if (!rc) rc = dev_alloc(e, &e->d_elect, 256);
    if (rc) { apus_gpu_destroy(e); return rc; }
    *out = e;
    rc = apus_gpu_reset(e);
    if (rc) { apus_gpu_destroy(e); *out = nullptr; return rc; }
    return apus_gpu_sync(e);
}

extern "C" void apus_gpu_destroy(apus_engine_t *e)
{
    if (!e) return;
    hipStreamSynchronize(e->stream);
    for (auto g : e->graphs) hipGraphExecDestroy(g);
    for (auto &t : e->timed) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    if (e->p_running) apus_gpu_persist_stop(e);       /* before anything it reads is freed */
    if (e->r_running) apus_gpu_rep_park(e);
    if (!e->ipc_ptrs.empty() && getenv("APUS_DEBUG")) fprintf(stderr, "[apus_gpu] destroy: %zu peer mappings still open (closed now)\n", e->ipc_ptrs.size());
    for (void *p : e->ipc_ptrs) hipIpcCloseMemHandle(p);
    for (void *p : e->allocs) hipFree(p);
    if (e->d_req) hipFree(e->d_req);
    if (e->d_req_len) hipFree(e->d_req_len);
    if (e->d_arena) hipFree(e->d_arena);
    if (e->d_round_first) hipFree(e->d_round_first);
    if (e->d_round_prefix) hipFree(e->d_round_prefix);
    if (e->d_round_change) hipFree(e->d_round_change);
    if (e->rq) { if (e->rq_bar) hipFree(e->rq); else hipHostFree(e->rq); }
    if (e->rh) hipHostFree(e->rh);
    if (e->rl) hipFree(e->rl);
    for (auto f : e->rfs) if (f) hipFree(f);
    for (auto f : e->rfh) if (f) hipHostFree(f);
    if (e->ss_buf) hipFree(e->ss_buf);
    if (e->rev0) hipEventDestroy(e->rev0);
    if (e->rev1) hipEventDestroy(e->rev1);
    if (e->rstream) hipStreamDestroy(e->rstream);
    free(e->r_slot_aend); free(e->r_win_cnt); free(e->r_win_len);
    if (e->r_lock_init) pthread_spin_destroy(&e->r_lock);
    if (e->ph) hipHostFree(e->ph);
    if (e->pd) hipFree(e->pd);
    if (e->pstream) hipStreamDestroy(e->pstream);
    if (e->h_live) hipHostFree(e->h_live);
    if (e->d_live) hipFree(e->d_live);
    if (e->live_copied) hipEventDestroy(e->live_copied);
    if (e->own_stream) hipStreamDestroy(e->stream);
    if (g_engine == e) g_engine = nullptr;
    delete e;
}

extern "C" int apus_gpu_sync(apus_engine_t *e)
{
    if (e && e->live_R) { int rc_ = flush_live(e); if (rc_) return rc_; }
    if (e && e->batching) return APUS_E_STATE;      /* close the batch first (apus_gpu_batch_end) */
    if (!e) return APUS_E_ARG;
    { int frc = flush_tick(e); if (frc) return frc; }
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

/* cfg.group_size is the CAPACITY (replicas that exist); d.group_size the size of the configuration the
 * leader decides with -- cid.size[0], or cid.size[1] while a resize is in its TRANSIT phase (the `size`
 * of the commit scan, dare_ibv_rc.c:1650-1758).  They differ only when a group is meant to grow. */
extern "C" int apus_gpu_set_group_size(apus_engine_t *e, uint32_t n)
{
    if (!e || n < 1 || n > e->cfg.group_size) return APUS_E_ARG;
    if (e->batching) return APUS_E_STATE;
    if (e->d.leader < e->cfg.group_size) { int frc = flush_tick(e); if (frc) return frc; }
    e->d.group_size = n;
    if (!e->cfg_appended)           /* the configured size the group starts with: every server holds that configuration */
        for (uint32_t i = 0; i < APUS_MAX_SERVERS; i++) e->mv[i].base = (1u << n) - 1;
    return 0;
}

/* a machine that (re)joins starts from log_new() (dare_log.h:120-136): its own process zeroes the replica
 * it hosts before the leader's apus_gpu_join recovers it (a peer's memory is never cleared from here) */
extern "C" int apus_gpu_clear_replica(apus_engine_t *e, uint32_t r)
{
    if (!e || r >= e->cfg.group_size || !e->d.rep[r].ring || ((e->imported_mask >> r) & 1u)) return APUS_E_STATE;
    if (e->batching) return APUS_E_STATE;
    HIPCHK(hipMemsetAsync(e->d.rep[r].ring, 0, e->d.log_len + 4096, e->stream));
    HIPCHK(hipMemsetAsync(e->d.rep[r].ack, 0, sizeof(uint32_t) * e->dir_cap, e->stream));
    HIPCHK(hipMemsetAsync(e->d.box[r], 0, sizeof(RepBox), e->stream));       /* a new machine: its doorbell sequence restarts */
    HIPCHK(hipMemsetAsync(e->d.ackb[r], 0, (size_t)e->cfg.group_size * e->dir_cap, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

/* the configuration as another rank's leader changed it (a JOIN it carried out): size and epoch */
extern "C" int apus_gpu_set_config(apus_engine_t *e, uint32_t group_size, uint64_t epoch)
{
    int rc = apus_gpu_set_group_size(e, group_size);
    if (rc) return rc;
    e->cid_epoch = epoch;
    return 0;
}

extern "C" int apus_gpu_reset(apus_engine_t *e)
{
    if (!e) return APUS_E_ARG;
    for (uint32_t i = 0; i < e->cfg.group_size; i++)
        if (e->d.rep[i].ring && !((e->imported_mask >> i) & 1u)) {
            /* log_new() zeroes the whole log (dare_log.h:128) */
            HIPCHK(hipMemsetAsync(e->d.rep[i].ring, 0, e->d.log_len + 4096, e->stream));
            HIPCHK(hipMemsetAsync(e->d.rep[i].ack, 0, sizeof(uint32_t) * e->dir_cap, e->stream));
            HIPCHK(hipMemsetAsync(e->d.box[i], 0, sizeof(RepBox), e->stream));
            HIPCHK(hipMemsetAsync(e->d.ackb[i], 0, (size_t)e->cfg.group_size * e->dir_cap, e->stream));
        }
    if (e->d_quirk) HIPCHK(hipMemsetAsync(e->d_quirk, 0, 16, e->stream));
    HIPCHK(hipMemsetAsync(e->d_cfgj, 0, 16, e->stream));
    for (uint32_t i = 0; i < APUS_MAX_SERVERS; i++) e->mv[i] = MemberView{(1u << e->d.group_size) - 1, 0, 0, 0};
    e->cfg_appended = false;
    e->d.leader = 0xFFFFFFFFu;
    e->tick_pending = false;
    e->free_lb = 0; e->host_status = 0; e->no_access = 0; e->adjust_mask = 0; e->cid_epoch = 0;
    e->reachable = (1u << e->d.group_size) - 1;
    e->d.reachable = e->reachable;
    {
        /* a peer's replica is reset by the process that hosts it */
        EngDev own = e->d;
        for (uint32_t i = 0; i < e->cfg.group_size; i++) if ((e->imported_mask >> i) & 1u) own.rep[i].ring = nullptr;
        hipLaunchKernelGGL(k_reset, dim3(e->cfg.group_size), dim3(64), 0, e->stream, own);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

/* ---- peer-mapped replicas (one replica per GPU / process) ------------------------------- */
/* hipIpcGetMemHandle with a second and a third try.  Round 6's soak (every election of the C layer fences now: ~300 fences per
 * suite run, five processes on one device allocating and exporting side by side) saw ONE export of a freshly allocated, block-owning
 * buffer fail with "invalid argument" -- and a group of five lose a second server to it.  The call succeeds when it is repeated. */
static hipError_t ipc_handle_of(hipIpcMemHandle_t *h, void *p)
{
    hipError_t er = hipSuccess;
    for (int attempt = 0; attempt < 4; attempt++) {
        er = hipIpcGetMemHandle(h, p);
        if (er == hipSuccess) {
            if (attempt) fprintf(stderr, "[apus_gpu] hipIpcGetMemHandle(%p) succeeded at attempt %d\n", p, attempt + 1);
            return er;
        }
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        struct timespec ts = {0, 2000000L << attempt};
        nanosleep(&ts, nullptr);
    }
    return er;
}

/* the spare (ring, mailbox) pairs of a hosted replica: allocated when the replica is first exported or fenced -- an engine that
 * hosts a whole group in one process (bench.py, most tests) never pays for them */
static int ensure_pairs(apus_engine *e, uint32_t replica)
{
    if (e->pairs_ready[replica]) return 0;
    const uint32_t cur = e->fences[replica] % APUS_FENCE_PAIRS;
    e->pair_ring[replica][cur] = e->d.rep[replica].ring; e->pair_box[replica][cur] = e->d.box[replica];
    for (uint32_t k = 0; k < APUS_FENCE_PAIRS; k++) {
        if (k == cur) continue;
        int rc;
        if ((rc = dev_alloc(e, &e->pair_ring[replica][k], e->d.log_len + 4096, true, e->ring_alloc == 1 ? hipDeviceMallocFinegrained : e->ring_alloc == 2 ? hipDeviceMallocUncached : 0u))) return rc;
        if ((rc = dev_alloc(e, &e->pair_box[replica][k], sizeof(RepBox), true, hipDeviceMallocUncached))) return rc;
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    e->pairs_ready[replica] = true;
    return 0;
}

extern "C" int apus_gpu_export_replica(apus_engine_t *e, uint32_t replica, apus_ipc_replica_t *out)
{
    if (!e || !out || replica >= e->cfg.group_size) return APUS_E_ARG;
    if (!((e->local_mask >> replica) & 1u) || ((e->imported_mask >> replica) & 1u)) return APUS_E_STATE;
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "apus_ipc_replica_t carries 64 bytes per handle");
    if (!e->ipc_cached[replica]) {
        /* every handle of the replica is taken ONCE, all pairs included: nothing is allocated, exported or mapped at election
         * time (see apus_gpu_fence_replica) */
        int rc = ensure_pairs(e, replica);
        if (rc) return rc;
        const RepDev &r = e->d.rep[replica];
        apus_ipc_replica_t &c = e->ipc_cache[replica];
        memset(&c, 0, sizeof c);
        void *bufs[APUS_IPC_BUFFERS] = { nullptr, r.hdr, r.dir_off, r.dir_len, r.ack, r.apply, nullptr, e->d.ackb[replica] };
        for (uint32_t k = 0; k < APUS_IPC_BUFFERS + 2 * APUS_FENCE_PAIRS; k++) {
            void *buf = k < APUS_IPC_BUFFERS ? bufs[k] : (k - APUS_IPC_BUFFERS) % 2 ? (void *)e->pair_box[replica][(k - APUS_IPC_BUFFERS) / 2] : (void *)e->pair_ring[replica][(k - APUS_IPC_BUFFERS) / 2];
            if (!buf) continue;                                    /* (handle[0], handle[6]: copies of the current pair's, below) */
            hipIpcMemHandle_t h;
            const hipError_t er = ipc_handle_of(&h, buf);
            if (er != hipSuccess) {
                fprintf(stderr, "[apus_gpu] hipIpcGetMemHandle failed for replica %u buffer %u (%p): %s\n", replica, k, buf, hipGetErrorString(er));
                (void)hipGetLastError();
                return APUS_E_HIP;
            }
            memcpy(k < APUS_IPC_BUFFERS ? c.handle[k] : c.pair[(k - APUS_IPC_BUFFERS) / 2][(k - APUS_IPC_BUFFERS) % 2], &h, sizeof h);
        }
        c.log_len = e->d.log_len; c.dir_cap = e->dir_cap; c.replica = replica; c.device = e->cfg.device;
        e->ipc_cached[replica] = true;
    }
    *out = e->ipc_cache[replica];
    out->fences = e->fences[replica];
    const uint32_t cur = e->fences[replica] % APUS_FENCE_PAIRS;
    memcpy(out->handle[0], out->pair[cur][0], 64);
    memcpy(out->handle[6], out->pair[cur][1], 64);
    return 0;
}

/* Drop every peer mapping (hipIpcCloseMemHandle) without freeing anything of this engine's own: the first half of an
 * orderly shutdown of a peer-mapped group -- every process unmaps, a barrier, then every process frees (apus_gpu_destroy).
 * An owner that frees a buffer a peer still has open cannot export the memory it gets next (hipIpcGetMemHandle fails). */
extern "C" int apus_gpu_unmap_peers(apus_engine_t *e)
{
    if (!e) return APUS_E_ARG;
    if (e->r_running || e->p_running || e->batching) {
        fprintf(stderr, "[apus_gpu] unmap_peers refused (replica kernels %d, persistent kernel %d, batch %d): the mappings stay open\n", (int)e->r_running, (int)e->p_running, (int)e->batching);
        return APUS_E_STATE;
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    for (uint32_t r = 0; r < APUS_MAX_SERVERS; r++)
        if ((e->imported_mask >> r) & 1u) {
            e->d.rep[r] = RepDev{};
            e->d.box[r] = nullptr; e->d.ackb[r] = nullptr;
            for (uint32_t k = 0; k < APUS_FENCE_PAIRS; k++) { e->pair_ring[r][k] = nullptr; e->pair_box[r][k] = nullptr; }
        }
    e->local_mask &= ~e->imported_mask;
    e->imported_mask = 0;
    int bad = 0;
    for (void *p : e->ipc_ptrs) {
        const hipError_t er = hipIpcCloseMemHandle(p);
        if (er != hipSuccess) { bad++; fprintf(stderr, "[apus_gpu] hipIpcCloseMemHandle(%p) failed: %s\n", p, hipGetErrorString(er)); (void)hipGetLastError(); }
    }
    if (getenv("APUS_DEBUG")) fprintf(stderr, "[apus_gpu] unmap_peers: %zu mappings closed, %d refused\n", e->ipc_ptrs.size(), bad);
    e->ipc_ptrs.clear();
    return bad ? APUS_E_HIP : 0;
}

/* One peer's mapping alone: a server whose PROCESS is gone and whose slot a new machine takes over (JOIN into an empty
 * place, dare_ibv_ud.c:995-1021) -- the members drop the mapping of the dead process's memory before they map the
 * newcomer's.  The mappings are closed in the order they were opened (APUS_IPC_BUFFERS per replica). */
extern "C" int apus_gpu_unmap_replica(apus_engine_t *e, uint32_t replica)
{
    if (!e || replica >= e->cfg.group_size) return APUS_E_ARG;
    if (!((e->imported_mask >> replica) & 1u)) return APUS_E_STATE;
    if (e->r_running || e->p_running || e->batching) return APUS_E_STATE;
    /* (the LEADER's mapping is dropped like any other once its process is gone or deposed: this engine has no leader until the next
     *  election's become_leader / set_leader) */
    if (e->d.leader == replica) e->d.leader = 0xFFFFFFFFu;
    HIPCHK(hipStreamSynchronize(e->stream));
    const RepDev &r = e->d.rep[replica];
    std::vector<void *> mine = { r.hdr, r.dir_off, r.dir_len, r.ack, r.apply, e->d.ackb[replica] };
    for (uint32_t k = 0; k < APUS_FENCE_PAIRS; k++) { mine.push_back(e->pair_ring[replica][k]); mine.push_back(e->pair_box[replica][k]); }
    for (void *m : mine)
        for (size_t k = 0; m && k < e->ipc_ptrs.size(); k++)
            if (e->ipc_ptrs[k] == m) { hipIpcCloseMemHandle(m); e->ipc_ptrs.erase(e->ipc_ptrs.begin() + (long)k); break; }
    for (uint32_t k = 0; k < APUS_FENCE_PAIRS; k++) { e->pair_ring[replica][k] = nullptr; e->pair_box[replica][k] = nullptr; }
    e->d.rep[replica] = RepDev{};
    e->d.box[replica] = nullptr; e->d.ackb[replica] = nullptr;
    e->local_mask &= ~(1u << replica);
    e->imported_mask &= ~(1u << replica);
    e->reachable &= ~(1u << replica); e->d.reachable = e->reachable;
    return 0;
}

/* The last entry of a replica's log -- what a vote is decided on (poll_vote_requests, dare_server.c:1661-1673: the
 * candidate's last (term, idx) must be at least as good as the voter's): out[0] = term, out[1] = idx (both 0: the log reads
 * as empty), out[2] = entry slots it holds, out[3] = its end offset.  Synchronises; between runs of the replica kernels. */
extern "C" int apus_gpu_last_entry(apus_engine_t *e, uint32_t replica, uint64_t out[4])
{
    if (!e || !out || replica >= e->cfg.group_size || !e->d.rep[replica].ring) return APUS_E_ARG;
    if (e->batching) return APUS_E_STATE;
    HIPCHK(hipStreamSynchronize(e->stream));
    uint64_t h[16];
    HIPCHK(hipMemcpy(h, e->d.rep[replica].hdr, sizeof h, hipMemcpyDeviceToHost));
    out[0] = out[1] = 0; out[2] = h[H_N_END]; out[3] = h[H_END];
    if (h[H_END] != e->d.log_len && h[H_N_END] > 0) {
        uint64_t off = 0, w[2] = {0, 0};
        HIPCHK(hipMemcpy(&off, &e->d.rep[replica].dir_off[(uint32_t)(h[H_N_END] - 1) & e->d.dir_mask], sizeof off, hipMemcpyDeviceToHost));
        if (off + 16 <= e->d.log_len) HIPCHK(hipMemcpy(w, e->d.rep[replica].ring + off, sizeof w, hipMemcpyDeviceToHost));
        out[1] = w[0]; out[0] = w[1];
    }
    return 0;
}

/* diagnostics / tests: words of a replica's mailbox between runs -- out[0] the commit doorbell (R4: entry slots the
 * leader told it are committed, as rung: NOT clipped to what it holds), [1] ctrl, [2] f_seq_next, [3] f_runs, [4] f_exit,
 * [5] persisted_by[who], [6] applied_by[who], [7] seqdone_by[who] (what follower `who` told this replica while it led) */
extern "C" int apus_gpu_rep_box_words(apus_engine_t *e, uint32_t replica, uint32_t who, uint64_t out[8])
{
    if (!e || !out || replica >= e->cfg.group_size || who >= 16 || !e->d.box[replica]) return APUS_E_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    RepBox *b = e->d.box[replica];
    uint64_t *src[8] = { &b->commit_bell, &b->ctrl, &b->f_seq_next, &b->f_runs, &b->f_exit, &b->persisted_by[who], &b->applied_by[who], &b->seqdone_by[who] };
    for (int k = 0; k < 8; k++) HIPCHK(hipMemcpy(&out[k], src[k], sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

/* ---- the receiver's fence (rc_revoke_log_access, dare_ibv_rc.c:2156-2243) for peer-mapped groups --------------------
 * The reference's voters reset the QPs of the old leader: its WRITEs bounce.  A buffer another process has mapped cannot be
 * taken back, but it can be LEFT: the replica this engine hosts moves the two buffers peers store into during a run -- its
 * log ring and its mailbox (doorbells, commit bell, cumulative ACKs) -- to the next of its APUS_FENCE_PAIRS pairs (device
 * copies of the old ones).  Every pointer another process still runs on, a deposed leader's first of all, leads to memory
 * nobody reads any more; its kernel may go on storing for as long as it likes.  `out` = the replica's handles, fences + 1:
 * the members of the new term follow with apus_gpu_remap_fenced.  All pairs are allocated at the first export, their handles
 * taken once, and mapped by a peer together with the replica: a fence allocates, exports, opens and closes NOTHING (the first
 * cut of round 6 did all four between two runs, and the soak found the runtime handing one address range to two mappings:
 * DESIGN.md 8.1).  A pair is lived in again APUS_FENCE_PAIRS fences after it was left -- a leader deposed that many terms
 * ago whose kernel still stores would write into the live ring again: outside the failure model, and said so in
 * include/apus_gpu.h (a deposed leader's process steps down when it sees a newer term's announcement, its kernel parks within
 * peer_ms).
 * The buffers peers write only from control-plane launches (control block, directory, apply stream: log adjustment, JOIN)
 * stay where they are: those launches sit behind the sender's term check (k_fence_check).
 * Not while a resident kernel or a batch is open; graphs captured before the fence hold the old pointers (peer-mapped groups
 * capture none). */
extern "C" int apus_gpu_fence_replica(apus_engine_t *e, uint32_t replica, apus_ipc_replica_t *out)
{
    if (!e || replica >= e->cfg.group_size) return APUS_E_ARG;
    if (!((e->local_mask >> replica) & 1u) || ((e->imported_mask >> replica) & 1u) || !e->d.rep[replica].ring || !e->d.box[replica]) return APUS_E_STATE;
    if (e->r_running || e->p_running || e->batching || !e->graphs.empty()) return APUS_E_STATE;
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    {   int rc = ensure_pairs(e, replica); if (rc) return rc; }
    const uint64_t ring_bytes = e->d.log_len + 4096;
    uint8_t *oring = e->d.rep[replica].ring; RepBox *obox = e->d.box[replica];
    const uint32_t next = (e->fences[replica] + 1) % APUS_FENCE_PAIRS;
    uint8_t *nring = e->pair_ring[replica][next]; RepBox *nbox = e->pair_box[replica][next];
    HIPCHK(hipMemcpyAsync(nring, oring, ring_bytes, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(nbox, obox, sizeof(RepBox), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->d.rep[replica].ring = nring; e->d.box[replica] = nbox;
    e->fences[replica]++;
    if (getenv("APUS_DEBUG")) fprintf(stderr, "[apus_gpu] fence %u of replica %u: ring %p -> %p, mailbox %p -> %p\n", e->fences[replica], replica, (void *)oring, (void *)nring, (void *)obox, (void *)nbox);
    return out ? apus_gpu_export_replica(e, replica, out) : 0;
}

/* a member of the new term follows a peer's fence: the pair `in->fences` names has been open here since the replica was mapped
 * (apus_gpu_import_replica maps them all), the pointers the kernels of the next run get are switched -- nothing is opened or
 * closed.  The replica must be imported already. */
extern "C" int apus_gpu_remap_fenced(apus_engine_t *e, const apus_ipc_replica_t *in)
{
    if (!e || !in || in->replica >= e->cfg.group_size) return APUS_E_ARG;
    if (in->log_len != e->d.log_len || in->dir_cap != e->dir_cap) return APUS_E_ARG;
    if (!((e->imported_mask >> in->replica) & 1u)) return APUS_E_STATE;
    if (in->fences == e->fences[in->replica]) return 0;            /* nothing moved since these buffers were mapped */
    if (e->r_running || e->p_running || e->batching) return APUS_E_STATE;
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const uint32_t cur = in->fences % APUS_FENCE_PAIRS;
    void *old[2] = { e->d.rep[in->replica].ring, e->d.box[in->replica] };
    void *np[2] = { e->pair_ring[in->replica][cur], e->pair_box[in->replica][cur] };
    if (!np[0] || !np[1]) return APUS_E_STATE;
    if (getenv("APUS_DEBUG")) fprintf(stderr, "[apus_gpu] remap of replica %u (fence %u -> %u): ring %p -> %p, mailbox %p -> %p\n", in->replica, e->fences[in->replica], in->fences, old[0], np[0], old[1], np[1]);
    e->d.rep[in->replica].ring = (uint8_t *)np[0];
    e->d.box[in->replica] = (RepBox *)np[1];
    e->fences[in->replica] = in->fences;
    return 0;
}

/* tests: `n` bytes at `off` of the ring replica `replica` left `back` fences ago (1 = the last one) -- where a deposed
 * leader's stores went */
extern "C" int apus_gpu_read_retired_ring(apus_engine_t *e, uint32_t replica, uint32_t back, uint64_t off, uint64_t n, void *dst)
{
    if (!e || replica >= e->cfg.group_size || !dst || back == 0 || back >= APUS_FENCE_PAIRS || back > e->fences[replica] || !e->pairs_ready[replica] || off + n > e->d.log_len) return APUS_E_ARG;
    HIPCHK(hipMemcpy(dst, e->pair_ring[replica][(e->fences[replica] - back) % APUS_FENCE_PAIRS] + off, n, hipMemcpyDeviceToHost));
    return 0;
}

/* bytes of the k-th buffer a replica exports (apus_ipc_replica_t: handle[0..7], then the pairs) */
static size_t ipc_buffer_bytes(const apus_engine *e, uint32_t k)
{
    if (k >= APUS_IPC_BUFFERS) return (k - APUS_IPC_BUFFERS) % 2 ? sizeof(RepBox) : (size_t)e->d.log_len + 4096;
    switch (k) {
    case 1: return sizeof(uint64_t) * 64;
    case 2: return sizeof(uint64_t) * e->dir_cap;
    case 3: case 4: return sizeof(uint32_t) * e->dir_cap;
    case 5: return sizeof(apus_apply_rec) * (size_t)e->dir_cap;
    case 7: return (size_t)e->cfg.group_size * e->dir_cap;
    default: return 0;
    }
}

extern "C" int apus_gpu_import_replica(apus_engine_t *e, const apus_ipc_replica_t *in)
{
    if (!e || !in || in->replica >= e->cfg.group_size) return APUS_E_ARG;
    if (in->log_len != e->d.log_len || in->dir_cap != e->dir_cap) return APUS_E_ARG;
    if ((e->local_mask >> in->replica) & 1u) return APUS_E_STATE;          /* hosted here, or imported already */
    HIPCHK(hipSetDevice(e->cfg.device));
    /* the six buffers that never move, then every (ring, mailbox) pair the replica moves through at its fences: ALL mappings of a
     * peer are made here, once.  (Round 6's first cut mapped the pair a fence had moved to at election time; the soak found a
     * survivor's new ring mapped over another live mapping -- the runtime handed an address range out twice once mappings had
     * been closed and opened between runs -- and the log bytes pushed through it lost while the doorbells arrived.) */
    const uint32_t NB = APUS_IPC_BUFFERS + 2 * APUS_FENCE_PAIRS;
    void *p[APUS_IPC_BUFFERS + 2 * APUS_FENCE_PAIRS] = {nullptr};
    for (uint32_t k = 0; k < NB; k++) {
        if (k == 0 || k == 6) continue;                              /* (copies of the current pair's handles) */
        hipIpcMemHandle_t h;
        memcpy(&h, k < APUS_IPC_BUFFERS ? in->handle[k] : in->pair[(k - APUS_IPC_BUFFERS) / 2][(k - APUS_IPC_BUFFERS) % 2], sizeof h);
        if (hipIpcOpenMemHandle(&p[k], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            fprintf(stderr, "[apus_gpu] hipIpcOpenMemHandle failed for replica %u buffer %u: %s\n", in->replica, k, hipGetErrorString(hipGetLastError()));
            for (uint32_t j = 0; j < k; j++) if (p[j]) hipIpcCloseMemHandle(p[j]);
            return APUS_E_HIP;
        }
    }
    /* what the kernels are about to be given must not lie over one another */
    for (uint32_t a = 0; a < NB; a++)
        for (uint32_t b = 0; p[a] && b < NB; b++)
            if (b != a && p[b] && (uintptr_t)p[a] <= (uintptr_t)p[b] && (uintptr_t)p[b] < (uintptr_t)p[a] + ipc_buffer_bytes(e, a)) {
                fprintf(stderr, "[apus_gpu] replica %u: the mappings of buffers %u (%p) and %u (%p) overlap\n", in->replica, a, p[a], b, p[b]);
                for (uint32_t j = 0; j < NB; j++) if (p[j]) hipIpcCloseMemHandle(p[j]);
                return APUS_E_HIP;
            }
    for (uint32_t k = 0; k < NB; k++) if (p[k]) e->ipc_ptrs.push_back(p[k]);
    for (uint32_t k = 0; k < APUS_FENCE_PAIRS; k++) {
        e->pair_ring[in->replica][k] = (uint8_t *)p[APUS_IPC_BUFFERS + 2 * k];
        e->pair_box[in->replica][k] = (RepBox *)p[APUS_IPC_BUFFERS + 2 * k + 1];
    }
    const uint32_t cur = in->fences % APUS_FENCE_PAIRS;
    RepDev &r = e->d.rep[in->replica];
    r.ring = e->pair_ring[in->replica][cur]; r.hdr = (uint64_t *)p[1]; r.dir_off = (uint64_t *)p[2]; r.dir_len = (uint32_t *)p[3];
    r.ack = (uint32_t *)p[4]; r.apply = (apus_apply_rec *)p[5]; r.idx = in->replica;
    e->d.box[in->replica] = e->pair_box[in->replica][cur]; e->d.ackb[in->replica] = (uint8_t *)p[7];
    e->local_mask |= 1u << in->replica;
    e->imported_mask |= 1u << in->replica;
    e->fences[in->replica] = in->fences;
    return 0;
}

extern "C" int apus_gpu_stage(apus_engine_t *e, const apus_req_t *reqs, uint64_t n,
                              const uint8_t *arena, uint64_t arena_bytes,
                              const uint32_t *round_n, uint64_t n_rounds)
{
    if (!e || (n && !reqs) || (n_rounds && !round_n)) return APUS_E_ARG;
    if (n >= (1ull << 32) || arena_bytes >= (1ull << 32)) return APUS_E_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    std::vector<ReqDev> hd(n);
    std::vector<uint16_t> hl(n);
    e->stage_max_T = APUS_HDR;
    for (uint64_t i = 0; i < n; i++) e->stage_max_T = std::max<uint64_t>(e->stage_max_T, APUS_HDR + (uint64_t)reqs[i].len);
    for (uint64_t g = 0; g < n; g++) {
        const apus_req_t &q = reqs[g];
        if (q.payload_off % 16 || q.payload_off + q.len > arena_bytes) return APUS_E_ARG;
        if (q.type == APUS_NOOP || q.type == APUS_CONFIG || q.type == APUS_HEAD || q.type > 15) return APUS_E_ARG;
        if (q.len && q.payload_off < 16) return APUS_E_ARG;     /* the copy reads 2 bytes in front */
        hd[g].req_id = q.req_id;
        hd[g].pay16_type = (uint32_t)(q.payload_off / 16) | ((uint32_t)q.type << 28);
        hd[g].len = q.len;
        hd[g].clt_id = q.clt_id;
        hl[g] = q.len;
    }
    /* round descriptors: [first request of round r : n_rounds + 1][bytes round r appends : n_rounds] */
    e->h_round_first.assign(2 * n_rounds + 1, 0);
    uint64_t acc = 0;
    for (uint64_t r = 0; r < n_rounds; r++) {
        if (round_n[r] < 1 || round_n[r] > APUS_MAX_ROUND || acc + round_n[r] > n) return APUS_E_ARG;
        e->h_round_first[r] = (uint32_t)acc;
        uint32_t bytes = 0;
        for (uint64_t g = acc; g < acc + round_n[r]; g++) bytes += APUS_HDR + (uint32_t)hl[g];
        e->h_round_first[n_rounds + 1 + r] = bytes;
        acc += round_n[r];
    }
    if (acc != n) return APUS_E_ARG;
    e->h_round_first[n_rounds] = (uint32_t)acc;
    auto renew = [&](void **p, size_t bytes) -> int {
        if (*p) { hipFree(*p); *p = nullptr; }
        if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) return APUS_E_NOMEM;
        return 0;
    };
    int rc;
    if ((rc = renew(&e->d_req, sizeof(ReqDev) * n))) return rc;
    if ((rc = renew(&e->d_req_len, sizeof(uint16_t) * n + 64))) return rc;
    if ((rc = renew(&e->d_arena, arena_bytes + 64))) return rc;
    if ((rc = renew(&e->d_round_first, sizeof(uint32_t) * (2 * n_rounds + 1)))) return rc;
    if (n) {
        HIPCHK(hipMemcpy(e->d_req, hd.data(), sizeof(ReqDev) * n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_req_len, hl.data(), sizeof(uint16_t) * n, hipMemcpyHostToDevice));
    }
    if (arena_bytes) HIPCHK(hipMemcpy(e->d_arena, arena, arena_bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_round_first, e->h_round_first.data(), sizeof(uint32_t) * (2 * n_rounds + 1), hipMemcpyHostToDevice));
    {
        /* byte prefix of the rounds: lets an append block place its round without a scan */
        std::vector<uint64_t> &pfx = e->h_round_prefix;
        pfx.assign(n_rounds + 1, 0);
        for (uint64_t r = 0; r < n_rounds; r++) pfx[r + 1] = pfx[r] + e->h_round_first[n_rounds + 1 + r];
        if ((rc = renew(&e->d_round_prefix, sizeof(uint64_t) * (n_rounds + 1)))) return rc;
        HIPCHK(hipMemcpy(e->d_round_prefix, pfx.data(), sizeof(uint64_t) * (n_rounds + 1), hipMemcpyHostToDevice));
    }
    {
        /* change points of the round size: chg[r] = #{i in [1, r]: round i has another number of requests than round i - 1} --
         * rounds [a, a + n) are all of one size iff chg[a + n - 1] == chg[a] (the sequencer's passes without ticket words) */
        std::vector<uint32_t> chg(n_rounds + 1, 0);
        for (uint64_t r = 1; r < n_rounds; r++) chg[r] = chg[r - 1] + (round_n[r] != round_n[r - 1] ? 1u : 0u);
        if (n_rounds) chg[n_rounds] = chg[n_rounds - 1];
        if ((rc = renew(&e->d_round_change, sizeof(uint32_t) * (n_rounds + 1)))) return rc;
        HIPCHK(hipMemcpy(e->d_round_change, chg.data(), sizeof(uint32_t) * (n_rounds + 1), hipMemcpyHostToDevice));
    }
    e->d.round_change = (const uint32_t *)e->d_round_change;
    e->d.req = (const ReqDev *)e->d_req;
    e->d.req_len = (const uint16_t *)e->d_req_len;
    e->d.arena = (const uint8_t *)e->d_arena;
    e->d.round_first = (const uint32_t *)e->d_round_first;
    e->d.round_bytes = e->d.round_first + n_rounds + 1;
    e->d.round_prefix = (const uint64_t *)e->d_round_prefix;
    e->n_reqs = n;
    e->n_rounds_staged = n_rounds;
    return 0;
}

/* ---- launch helpers -------------------------------------------------------- */
static inline uint32_t cap_grid(uint64_t n, uint32_t per_block, uint32_t cap)
{
    uint64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

/* persist + ACK scan + apply + bookkeeping for whatever is new (mode 0: staged rounds) -- the
 * second half of the phased ABI (commit_live / commit_rounds); run_rounds uses k_call instead */
static int launch_tail_view(apus_engine *e, const EngDev &view, uint64_t r0, uint32_t R, int mode, uint64_t n_hint)
{
    const uint32_t fm = sync_mask(e);
    const uint32_t rm = fm | ((e->local_mask >> e->d.leader) & 1u ? (1u << e->d.leader) : 0);
    const uint64_t n = n_hint ? n_hint : 1;
    if (fm)     /* followers on this device: persist + ACK + quorum test in one pass */
        hipLaunchKernelGGL(k_persist_commit, dim3(cap_grid(n, 256, 2048)), dim3(256), 0, e->stream, view, r0, R, fm);
    else        /* ACK bits were merged from remote followers (k_mp_ack_merge) */
        hipLaunchKernelGGL(k_commit, dim3(cap_grid(n, 1024, 512)), dim3(1024), 0, e->stream, view, r0, R);
    /* appliers + per-round record blocks + the bookkeeping block */
    const uint32_t nR = (mode == 0 && R) ? cap_grid(R, 256, 8) : 0;
    hipLaunchKernelGGL(k_apply, dim3(cap_grid(n, 1024, 1024) + nR + 1, popc(rm)), dim3(256), 0, e->stream, view, r0, R, rm, mode,
                       fm, nR);
    QUIRK_PASS(e, 1);
    HIPCHK(hipGetLastError());
    return 0;
}

static int launch_tail(apus_engine *e, uint64_t r0, uint32_t R, int mode, uint64_t n_hint)
{
    return launch_tail_view(e, e->d, r0, R, mode, n_hint);
}

static int launch_catchup(apus_engine *e)
{
    const uint32_t fm = sync_mask(e);
    FENCE_CHECK(e, fm);                         /* every launch path comes through here first */
    if (!fm || !e->lag_possible) return 0;      /* small lags are handled inside k_sequence / k_control_round */
    hipLaunchKernelGGL(k_catchup, dim3(128, popc(fm)), dim3(256), 0, e->stream, e->d, fm);
    HIPCHK(hipGetLastError());
    e->lag_possible = false;
    return 0;
}

static int need_leader(apus_engine *e)
{
    if (!e) return APUS_E_ARG;
    if (e->d.leader >= e->d.group_size || !((e->local_mask >> e->d.leader) & 1u)) return APUS_E_STATE;
    return 0;
}

/* get_tailq_message + log_append_entry + R1: catch-up, sequence, append+push */
static int launch_append(apus_engine *e, const EngDev &view, uint64_t r0, uint32_t R)
{
    int rc;
    const uint32_t fm = sync_mask(e);
    if ((rc = launch_catchup(e))) return rc;
    const uint32_t tick = e->tick_pending ? 1u : 0u;
    e->tick_pending = false;
    hipLaunchKernelGGL(k_sequence, dim3(1), dim3(1024), 0, e->stream, view, r0, R, fm, tick, fm);
    hipLaunchKernelGGL(k_append_push, dim3(R), dim3(256), 0, e->stream, view, r0, R, fm);
    HIPCHK(hipGetLastError());
    return 0;
}

/* Admission of `bytes` of new entries (+ `slack`: <HEAD> entries of due prune ticks, the bytes a wrap
 * skips) BEFORE anything is launched.  The reference's leader refuses a request when the log is full
 * (log_add_new_entry returns NULL at end == head, dare_log.h:168,213-221, 492-495) -- and runs over
 * un-pruned entries when a request merely crosses head; here a batch that does not fit into the free
 * part of the ring is refused as a whole and nothing of it is stored.  The bound is conservative
 * (head only moves forward); the control block is read back only when the bound does not cover the
 * batch, i.e. about once per lap of the ring.  A captured graph cannot consult the host: its
 * launches rely on the device-side flag alone (apus_gpu_status, APUS_ST_LOG_FULL). */
static int admit_bytes(apus_engine *e, uint64_t bytes, uint64_t slack)
{
    if (e->capturing) return 0;
    const uint64_t need = bytes + slack;
    if (e->free_lb >= need) { e->free_lb -= need; return 0; }
    HIPCHK(hipStreamSynchronize(e->stream));
    uint64_t h[64];
    HIPCHK(hipMemcpy(h, e->d.rep[e->d.leader].hdr, sizeof h, hipMemcpyDeviceToHost));
    const uint64_t L = e->d.log_len, end = h[H_END], head = h[H_HEAD];
    uint64_t fr;
    if (end == L) fr = L;                                   /* reads as empty */
    else if (end == head) fr = 0;                           /* log_is_full */
    else fr = L - (end > head ? end - head : L - (head - end));
    e->free_lb = fr;
    if (need > fr) { e->host_status |= APUS_ST_LOG_FULL; return APUS_E_FULL; }
    e->free_lb -= need;
    return 0;
}

/* k_call sequences up to this many rounds per launch (their prefix stays in LDS) */
#define APUS_CALL_ROUNDS 1024u

/* one call's launch parameters */
static CallArgs call_args(apus_engine *e, uint64_t c0, uint32_t R, uint32_t tick, uint32_t *blocks, bool lean = false)
{
    const uint32_t fm = sync_mask(e);
    const uint32_t rm = fm | ((e->local_mask >> e->d.leader) & 1u ? (1u << e->d.leader) : 0);
    const uint64_t n = e->h_round_first[c0 + R] - e->h_round_first[c0];
    CallArgs a;
    a.r0 = c0; a.R = R; a.tick = tick;
    /* the scan / apply blocks only work when the replicas are not in step: a modest number, grid-stride */
    a.nS = cap_grid(n, 256, 32); a.nA = cap_grid(n, 1024, 16); a.nR = cap_grid(R, 256, 8);
    /* lean: every configured follower is reachable and nobody lags -- the call is expected to be in step,
     * where these blocks only look at the segment's record and leave; a few of them keep the not-in-step
     * path correct (grid-stride) and leave the device to the append workgroups (flush_batch) */
    if (lean) { a.nS = std::min(a.nS, 8u); a.nA = std::min(a.nA, 4u); a.nR = std::min(a.nR, 4u); }
    /* rounds with many 16-byte units are shared by SP workgroups each (a launch of a few hundred
     * large rounds would leave most of the 256 CUs idle) */
    const uint64_t units = (e->h_round_prefix[c0 + R] - e->h_round_prefix[c0]) / 16;
    const uint64_t sp_units = e->sp_units;
    a.SP = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(1, (units / R + sp_units * 2 / 3) / sp_units));
    /* small rounds: one wavefront per round, APUS_GP rounds per workgroup (append_group) -- the
     * block's load / sequencing chain is paid once for four rounds.  Large rounds stay one
     * workgroup (or SP) per round: a wavefront per round would leave too few workgroups. */
    a.GP = (e->gp_rounds > 1 && a.SP == 1 && R >= 4 * APUS_GP && units / R <= e->gp_max_units) ? APUS_GP : 1;
    *blocks = 1 + call_append_blocks(a) + a.nR + 1 + a.nS + a.nA * popc(rm);
    return a;
}

/* launch the recorded segments of a batch (apus_gpu_batch_begin .. _end) as k_step launches */
static int flush_batch(apus_engine *e)
{
    if (e->batch.empty()) return 0;
    const uint32_t fm = sync_mask(e);
    const uint32_t rm = fm | ((e->local_mask >> e->d.leader) & 1u ? (1u << e->d.leader) : 0);
    /* One launch must not lap the ring: the XCDs' L2s are not coherent with each other, so two
     * segments of one launch that store to the same ring line (one lap apart) could be written
     * back in either order.  A kernel boundary writes everything back; so a launch takes
     * segments while they append less than one ring (minus slack for <HEAD> entries and the bytes
     * skipped at a wrap). */
    const uint64_t lap = e->d.log_len - e->d.log_len / 8;
    /* (admission of a batch is the device's: prune ticks inside it move head -- segment_refused) */
    e->free_lb = 0;
    size_t i = 0;
    while (i < e->batch.size()) {
        StepTable T;
        memset(&T, 0, sizeof T);
        uint32_t blk = 0;
        uint64_t bytes = 0;
        uint32_t k = 0;
        /* A launch whose append workgroups are all resident at once issues every load before anybody
         * stores and then runs at the HBM roofline (append_group); one that is bigger than the device
         * does not.  So: as many segments as fit with APUS_GD rounds per wavefront, then as few rounds
         * per wavefront as still fit (more wavefronts in flight for a small launch). */
        auto grouped_blocks = [](const CallArgs &a, uint32_t gd) { return a.GP > 1 ? (a.R + APUS_GP * gd - 1) / (APUS_GP * gd) : a.R * a.SP; };
        auto svc_of = [](const apus_engine::BatchSeg &g) { return g.blocks - 2 - call_append_blocks(g.a); };
        const bool fit_ok = !e->p_running && e->step_slots > 0;
        uint32_t fit_app = 0, fit_svc = 0, consumed = 0, seg_svc[APUS_STEP_SEGS];
        apus_engine::BatchSeg split_rest;
        bool have_rest = false;
        while (i + k < e->batch.size() && k < APUS_STEP_SEGS) {
            const apus_engine::BatchSeg &g = e->batch[i + k];
            if (k && bytes + g.bytes + APUS_HDR > lap) break;
            const uint32_t nab = grouped_blocks(g.a, APUS_GD), ms = std::max(fit_svc, svc_of(g));
            /* (only launches of grouped segments are cut to the device: with a workgroup or more per round
             * the append blocks of even one segment exceed it, residency is not to be had) */
            if (k && fit_ok && g.a.GP > 1 && 2 * (k + 1) + fit_app + nab + ms > e->step_slots && 2 * k + fit_app + fit_svc <= e->step_slots) {
                /* the segment does not fit as a whole: its first rounds fill the launch (a call is a run of
                 * rounds; where it is cut never changes the logs), the rest opens the next one */
                const uint32_t used = 2 * (k + 1) + fit_app + ms;
                const uint32_t room = e->step_slots > used ? e->step_slots - used : 0;
                const uint32_t R1 = room * APUS_GR;
                if (g.a.GP > 1 && R1 >= 64 && g.a.R >= R1 + 4 * APUS_GP) {
                    uint32_t b1 = 0, b2 = 0;
                    apus_engine::BatchSeg first = g, rest = g;
                    first.a = call_args(e, g.a.r0, R1, g.a.tick, &b1, g.lean);
                    first.blocks = b1; first.bytes = e->h_round_prefix[g.a.r0 + R1] - e->h_round_prefix[g.a.r0];
                    rest.a = call_args(e, g.a.r0 + R1, g.a.R - R1, 0, &b2, g.lean);
                    rest.blocks = b2; rest.bytes = g.bytes - first.bytes;
                    split_rest = rest; have_rest = true;
                    fit_app += grouped_blocks(first.a, APUS_GD); fit_svc = std::max(fit_svc, first.blocks - 2 - call_append_blocks(first.a));
                    T.seg[k] = first.a;
                    seg_svc[k] = first.blocks - 2 - call_append_blocks(first.a);
                    bytes += first.bytes + APUS_HDR;
                    k++;
                }
                break;
            }
            fit_app += nab; fit_svc = ms;
            T.seg[k] = g.a;
            seg_svc[k] = svc_of(g);
            bytes += g.bytes + APUS_HDR;
            k++; consumed++;
        }
        T.S = k;
        T.max_T = (uint32_t)e->stage_max_T;
        uint32_t gd = 1;
        for (; gd < APUS_GD; gd++) {
            uint32_t n_app = 0;
            for (uint32_t j = 0; j < k; j++) n_app += grouped_blocks(T.seg[j], gd);
            if (fit_ok && 2 * k + n_app + fit_svc <= e->step_slots) break;
        }
        if (!fit_ok) gd = 1;
        uint32_t n_app = 0;
        for (uint32_t j = 0; j < k; j++) {
            if (T.seg[j].GP > 1) T.seg[j].GP = APUS_GP * gd;
            const uint32_t nab = call_append_blocks(T.seg[j]);
            T.blk0[j] = blk;
            blk += 2 + nab + seg_svc[j];
            n_app += nab;
        }
        T.blk0[k] = blk;
        {
            /* append blocks first, if all of them + the single blocks + any one segment's other blocks
             * fit on the device at once (and nothing else of this engine occupies it) */
            T.order = (k > 1 && fit_ok && 2 * k + n_app + fit_svc <= e->step_slots) ? 1u : 0u;
            if (getenv("APUS_DEBUG")) fprintf(stderr, "[apus_gpu] k_step launch: %u segments, %u append blocks (%u rounds per wavefront), %u blocks, order %u\n", k, n_app, gd, blk, T.order);
            uint32_t ab = 2 * k, sv = 2 * k + n_app;
            for (uint32_t j = 0; j < k; j++) {
                const uint32_t nab = call_append_blocks(T.seg[j]);
                T.ab0[j] = ab; T.sv0[j] = sv;
                ab += nab; sv += (T.blk0[j + 1] - T.blk0[j]) - 2 - nab;
            }
            T.ab0[k] = ab; T.sv0[k] = sv;
        }
        TimedLaunch *tl = nullptr;
        if (e->timing && !e->capturing) {
            if (e->timed_used == e->timed.size()) {
                TimedLaunch t;
                HIPCHK(hipEventCreate(&t.a)); HIPCHK(hipEventCreate(&t.b));
                e->timed.push_back(t);
            }
            tl = &e->timed[e->timed_used++];
            HIPCHK(hipEventRecord(tl->a, e->stream));
        }
        FENCE_CHECK(e, fm);
        if (fence_on(e)) hipLaunchKernelGGL(k_step_fenced, dim3(blk), dim3(256), 0, e->stream, e->d, T, fm, rm);
        else hipLaunchKernelGGL(k_step, dim3(blk), dim3(256), 0, e->stream, e->d, T, fm, rm);
        if (tl) HIPCHK(hipEventRecord(tl->b, e->stream));
        i += consumed;
        if (have_rest) e->batch[i] = split_rest;       /* the rest of the segment that was cut opens the next launch */
    }
    e->batch.clear();
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int apus_gpu_run_rounds(apus_engine_t *e, uint64_t r0, uint64_t n_rounds)
{
    if (e && e->live_R) { int rc_ = flush_live(e); if (rc_) return rc_; }
    int rc = need_leader(e);
    if (rc) return rc;
    if (r0 + n_rounds > e->n_rounds_staged || n_rounds > e->max_rounds) return APUS_E_ARG;
    if (n_rounds == 0) return 0;
    const uint32_t fm = sync_mask(e);
    const uint32_t rm = fm | ((e->local_mask >> e->d.leader) & 1u ? (1u << e->d.leader) : 0);
    if (!e->batching) {
        const uint64_t chunks = (n_rounds + APUS_CALL_ROUNDS - 1) / APUS_CALL_ROUNDS;
        if ((rc = admit_bytes(e, e->h_round_prefix[r0 + n_rounds] - e->h_round_prefix[r0],
                              chunks * e->stage_max_T + (e->tick_pending ? APUS_HDR : 0) + APUS_HDR))) return rc;
    }
    if (!e->batching || e->batch.empty()) { if ((rc = launch_catchup(e))) return rc; }
    for (uint64_t done = 0; done < n_rounds; done += APUS_CALL_ROUNDS) {
        const uint64_t c0 = r0 + done;
        const uint32_t R = (uint32_t)std::min<uint64_t>(APUS_CALL_ROUNDS, n_rounds - done);
        const uint32_t tick = e->tick_pending ? 1u : 0u;
        e->tick_pending = false;
        uint32_t blocks = 0;
        const bool lean = e->batching && !(e->d.flags & 1u) && !e->lag_possible &&
                          popc(fm) + 1 == (int)e->d.group_size;
        const CallArgs a = call_args(e, c0, R, tick, &blocks, lean);
        if (e->batching) {                    /* recorded; apus_gpu_batch_end launches the lot as k_step */
            e->batch.push_back(apus_engine::BatchSeg{a, blocks, e->h_round_prefix[c0 + R] - e->h_round_prefix[c0], lean});
            continue;
        }
        TimedLaunch *tl = nullptr;
        if (e->timing && !e->capturing) {
            if (e->timed_used == e->timed.size()) {
                TimedLaunch t;
                HIPCHK(hipEventCreate(&t.a)); HIPCHK(hipEventCreate(&t.b));
                e->timed.push_back(t);
            }
            tl = &e->timed[e->timed_used++];
            HIPCHK(hipEventRecord(tl->a, e->stream));
        }
        /* the whole call in one launch: sequencer, append + push, per-round records, bookkeeper,
         * persist + ACK scan, apply (k_call's block roles) */
        if (fence_on(e)) hipLaunchKernelGGL(k_call_fenced, dim3(blocks), dim3(256), 0, e->stream, e->d, a, fm, rm);
        else hipLaunchKernelGGL(k_call, dim3(blocks), dim3(256), 0, e->stream, e->d, a, fm, rm);
        if (tl) HIPCHK(hipEventRecord(tl->b, e->stream));
        QUIRK_PASS(e, tick ? 1 : 0);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

/* Batching: between _begin and _end, apus_gpu_run_rounds calls (and the prune ticks deferred into
 * them) are recorded and then issued as multi-segment launches (k_step, up to 32 calls each) --
 * the same work, without a kernel boundary between consecutive calls.  Only run_rounds and
 * tick_prune may be called while a batch is open. */
extern "C" int apus_gpu_batch_begin(apus_engine_t *e)
{
    if (e && e->live_R) { int rc_ = flush_live(e); if (rc_) return rc_; }
    int rc = need_leader(e);
    if (rc) return rc;
    if (e->batching || ref_quirks(e)) return APUS_E_STATE;      /* (the quirk pass runs behind single calls only) */
    e->batching = true;
    e->batch.clear();
    return 0;
}

extern "C" int apus_gpu_batch_end(apus_engine_t *e)
{
    if (!e || !e->batching) return APUS_E_STATE;
    e->batching = false;
    return flush_batch(e);
}

/* ---- live submission: what the proxy's DARE thread does every polling() pass ---- */
static int live_view(apus_engine *e, EngDev *view)
{
    if (!e->h_live) {
        HIPCHK(hipHostMalloc((void **)&e->h_live, LIVE_BYTES, hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&e->d_live, LIVE_BYTES));
        HIPCHK(hipEventCreateWithFlags(&e->live_copied, hipEventDisableTiming));
    }
    *view = e->d;
    view->req = (const ReqDev *)e->d_live;
    view->req_len = (const uint16_t *)(e->d_live + LIVE_OFF_LEN);
    view->round_first = (const uint32_t *)(e->d_live + LIVE_OFF_RF);
    view->round_bytes = (const uint32_t *)(e->d_live + LIVE_OFF_RB);
    view->round_prefix = (const uint64_t *)(e->d_live + LIVE_OFF_PFX);
    view->arena = e->d_live + LIVE_OFF_ARENA;
    return 0;
}

/* dare_ib_poll_tailq: n queued requests become log entries (rounds of <= 64) */
extern "C" int apus_gpu_append_live(apus_engine_t *e, const apus_req_t *reqs, uint32_t n,
                                    const uint8_t *arena, uint64_t arena_bytes)
{
    if (e && e->batching) return APUS_E_STATE;      /* close the batch first (apus_gpu_batch_end) */
    int rc = need_leader(e);
    if (rc) return rc;
    if (!reqs || n == 0 || n > LIVE_REQS || arena_bytes + 16 > LIVE_ARENA) return APUS_E_ARG;
    if (e->live_R) return APUS_E_STATE;             /* previous batch not committed yet */
    EngDev view;
    if ((rc = live_view(e, &view))) return rc;
    if (e->live_pending) { HIPCHK(hipEventSynchronize(e->live_copied)); e->live_pending = false; }
    ReqDev *hd = (ReqDev *)e->h_live;
    uint16_t *hl = (uint16_t *)(e->h_live + LIVE_OFF_LEN);
    uint32_t *rf = (uint32_t *)(e->h_live + LIVE_OFF_RF);
    uint8_t *ha = e->h_live + LIVE_OFF_ARENA;
    /* the live arena keeps the caller's offsets, shifted by 16 so that byte -2 exists */
    for (uint32_t g = 0; g < n; g++) {
        const apus_req_t &q = reqs[g];
        if (q.payload_off % 16 || q.payload_off + q.len > arena_bytes) return APUS_E_ARG;
        if (q.type == APUS_NOOP || q.type == APUS_CONFIG || q.type == APUS_HEAD || q.type > 15) return APUS_E_ARG;
        hd[g].req_id = q.req_id;
        hd[g].pay16_type = (uint32_t)(q.payload_off / 16 + 1) | ((uint32_t)q.type << 28);
        hd[g].len = q.len; hd[g].clt_id = q.clt_id; hl[g] = q.len;
    }
    uint32_t *rb = (uint32_t *)(e->h_live + LIVE_OFF_RB);
    uint64_t *pf = (uint64_t *)(e->h_live + LIVE_OFF_PFX);
    uint32_t R = 0;
    pf[0] = 0;
    for (uint32_t g = 0; g < n; g += APUS_MAX_ROUND) {
        uint32_t bytes = 0;
        for (uint32_t k = g; k < n && k < g + APUS_MAX_ROUND; k++) bytes += APUS_HDR + (uint32_t)hl[k];
        rb[R] = bytes;
        pf[R + 1] = pf[R] + bytes;
        rf[R++] = g;
    }
    rf[R] = n;
    {
        /* get_tailq_message refuses what the log cannot take (dare_ibv_ud.c:780-790 -> log_append_entry) */
        uint64_t max_T = APUS_HDR;
        for (uint32_t g = 0; g < n; g++) max_T = std::max<uint64_t>(max_T, APUS_HDR + (uint64_t)hl[g]);
        if ((rc = admit_bytes(e, pf[R], max_T + (e->tick_pending ? APUS_HDR : 0) + APUS_HDR))) return rc;
    }
    if (arena_bytes) memcpy(ha + 16, arena, arena_bytes);
    HIPCHK(hipMemcpyAsync(e->d_live, e->h_live, LIVE_OFF_ARENA + 16 + arena_bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipEventRecord(e->live_copied, e->stream));
    e->live_pending = true;
    /* The entries are staged; the consensus pass over them -- append, replication, ACKs, commit,
     * apply -- is ONE launch (k_call), issued by apus_gpu_commit_live (dare_ib_write_remote_logs,
     * which polling() calls right after dare_ib_poll_tailq) or by whatever engine call comes next. */
    e->live_r0 = 0; e->live_R = R; e->live_n = n; e->live_bytes = pf[R];
    return 0;
}

/* the staged live batch as one k_call launch over the live view */
static int flush_live(apus_engine *e)
{
    if (!e->live_R) return 0;
    int rc;
    EngDev view;
    if ((rc = live_view(e, &view))) return rc;
    if ((rc = launch_catchup(e))) return rc;
    const uint32_t fm = sync_mask(e);
    const uint32_t rm = fm | ((e->local_mask >> e->d.leader) & 1u ? (1u << e->d.leader) : 0);
    const uint32_t R = (uint32_t)e->live_R;
    CallArgs a;
    a.r0 = 0; a.R = R; a.tick = e->tick_pending ? 1u : 0u;
    e->tick_pending = false;
    a.nS = cap_grid(e->live_n, 256, 32); a.nA = cap_grid(e->live_n, 1024, 16); a.nR = cap_grid(R, 256, 8);
    a.SP = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(1, (e->live_bytes / 16 / R + e->sp_units * 2 / 3) / e->sp_units));
    a.GP = 1;                                      /* (no staged byte prefix on the live path) */
    if (fence_on(e)) hipLaunchKernelGGL(k_call_fenced, dim3(1 + R * a.SP + a.nR + 1 + a.nS + a.nA * popc(rm)), dim3(256), 0, e->stream, view, a, fm, rm);
    else hipLaunchKernelGGL(k_call, dim3(1 + R * a.SP + a.nR + 1 + a.nS + a.nA * popc(rm)), dim3(256), 0, e->stream, view, a, fm, rm);
    e->live_R = 0;
    QUIRK_PASS(e, a.tick ? 1 : 0);
    HIPCHK(hipGetLastError());
    return 0;
}

/* dare_ib_write_remote_logs: follower ACKs, ACK scan, commit, apply for the appended batch */
extern "C" int apus_gpu_commit_live(apus_engine_t *e, int wait_for_commit)
{
    int rc = need_leader(e);
    if (rc) return rc;
    EngDev view;
    if ((rc = live_view(e, &view))) return rc;
    if (e->live_R) rc = flush_live(e);
    else rc = apus_gpu_quiesce(e);
    if (rc) return rc;
    if (wait_for_commit) HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int apus_gpu_submit(apus_engine_t *e, const apus_req_t *reqs, uint32_t n,
                               const uint8_t *arena, uint64_t arena_bytes)
{
    int rc = apus_gpu_append_live(e, reqs, n, arena, arena_bytes);
    if (rc) return rc;
    return apus_gpu_commit_live(e, 0);
}

static int launch_control_round(apus_engine *e, int mode, uint32_t type, uint64_t d0, uint64_t d1,
                                uint64_t req_id = 0, uint32_t clt_id = 0);
static int flush_tick(apus_engine *e)
{
    if (!e->tick_pending) return 0;
    e->tick_pending = false;
    return launch_control_round(e, 1, APUS_HEAD, 0, 0);
}

static int launch_control_round(apus_engine *e, int mode, uint32_t type, uint64_t d0, uint64_t d1,
                                uint64_t req_id, uint32_t clt_id)
{
    int rc = launch_catchup(e);
    if (rc) return rc;
    const uint32_t fm = sync_mask(e);
    if ((mode & 7) != 2) e->free_lb = e->free_lb > 2 * APUS_HDR ? e->free_lb - 2 * APUS_HDR : 0;     /* at most one 64-byte entry (+ a skipped tail) */
    hipLaunchKernelGGL(k_control_round, dim3(1), dim3(256), 0, e->stream, e->d, mode, type, d0, d1, fm, fm, req_id, clt_id);
    QUIRK_PASS(e, 1);
    if (type == APUS_CONFIG && (mode & 7) != 2) {
        hipLaunchKernelGGL(k_cfg_journal, dim3(1), dim3(64), 0, e->stream, e->d, e->d_cfgj);
        e->cfg_appended = true;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int apus_gpu_quiesce(apus_engine_t *e)
{
    if (e && e->live_R) { int rc_ = flush_live(e); if (rc_) return rc_; }
    if (e && e->batching) return APUS_E_STATE;      /* close the batch first (apus_gpu_batch_end) */
    int rc = need_leader(e);
    if (rc) return rc;
    { int frc = flush_tick(e); if (frc) return frc; }
    return launch_control_round(e, 2, 0, 0, 0);
}

extern "C" int apus_gpu_append_control(apus_engine_t *e, uint8_t type, const void *data)
{
    if (e && e->live_R) { int rc_ = flush_live(e); if (rc_) return rc_; }
    if (e && e->batching) return APUS_E_STATE;      /* close the batch first (apus_gpu_batch_end) */
    int rc = need_leader(e);
    if (rc) return rc;
    { int frc = flush_tick(e); if (frc) return frc; }
    uint64_t d0 = 0, d1 = 0;
    if (type == APUS_CONFIG) { if (!data) return APUS_E_ARG; memcpy(&d0, data, 8); memcpy(&d1, (const uint8_t *)data + 8, 8); }
    else if (type == APUS_HEAD) { if (!data) return APUS_E_ARG; memcpy(&d0, data, 8); }
    else if (type != APUS_NOOP) return APUS_E_ARG;
    return launch_control_round(e, 0, type, d0, d1);
}

/* log_pruning timer tick.  The reference's timer fires between polling() passes,
 * i.e. with every reachable follower caught up; the pipeline above leaves them
 * caught up after every call, so no separate quiesce pass is needed. */
extern "C" int apus_gpu_tick_prune(apus_engine_t *e)
{
    if (e && e->live_R) { int rc_ = flush_live(e); if (rc_) return rc_; }
    int rc = need_leader(e);
    if (rc) return rc;
    /* an open batch: anything that has to run now goes behind what was recorded so far */
    const uint32_t all_here = (1u << e->d.group_size) - 1;
    if (e->batching && (e->tick_pending || (e->local_mask & all_here) != all_here) && (rc = flush_batch(e))) return rc;
    if ((rc = flush_tick(e))) return rc;            /* two ticks in a row: the first one runs now */
    /* When every replica lives on this device the tick is deferred and fused into the
     * sequencer of the next batch (same position in the order of events, one launch less);
     * any other call flushes it first. */
    if ((e->local_mask & all_here) == all_here) { e->tick_pending = true; return 0; }
    return launch_control_round(e, 1, APUS_HEAD, 0, 0);
}

__global__ void k_set_roles(const EngDev E, uint64_t sid, uint32_t bitmask, uint32_t follow_mask)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *lh = Ld.hdr;
    lh[H_SID] = sid;
    lh[H_CID_BITMASK] = bitmask;
    /* a former follower has no tail (hb_receive_cb resets it): log_get_tail, dare_log.h:402-457 */
    const uint64_t n_end = lh[H_N_END];
    if (n_end > 0 && lh[H_END] != E.log_len) {
        const uint64_t t = Ld.dir_off[(uint32_t)(n_end - 1) & E.dir_mask];
        lh[H_TAIL] = t;
        lh[H_LAST_IDX] = ld8u(Ld.ring + t);
    }
    lh[H_N_VISIBLE] = n_end;
    lh[H_OLD_END] = lh[H_END];
    lh[H_N_PERSIST] = n_end;
    lh[H_TERM_SLOT0] = n_end;                 /* the blank CONFIG entry of this term goes here (become_leader appends it next) */
    /* ACK words of entries this server did not append itself start empty */
    for (uint64_t s2 = lh[H_N_COMMIT]; s2 < n_end; s2++) Ld.ack[(uint32_t)s2 & E.dir_mask] = 0;
    for (uint32_t i = 0; i < E.group_size; i++) lh[H_APPLY_OFFSETS + i] = lh[H_HEAD];  /* dare_server.c:1504-1507 */
    /* A term fence raised against this engine while it led an OLDER term (k_fence_check) ends here only if no server of the
     * configuration that this engine can see holds a NEWER term than the one it has just won -- looked at BEFORE the
     * followers below are given the new SID (round 3 looked afterwards: always false), and over the configured
     * servers this leader can REACH (rc_restore_log_access, dare_ibv_rc.c:2245-2290: the voters restore the log access of
     * the server they vote for -- a server that died holding a higher term, e.g. after a failed candidacy of its own, takes
     * no part in that; round 4 scanned the unreachable ones too and such a server kept every launch of a legitimately
     * elected leader fenced for ever: ADVICE r4).  A follower that is ahead and cut off raises the fence again through
     * k_fence_check in front of the first launch that pushes to it once it is released. */
    bool ahead = false;
    for (uint32_t i = 0; i < E.group_size; i++)
        if (i != E.leader && ((bitmask >> i) & 1u) && ((follow_mask >> i) & 1u) && E.rep[i].ring && (E.rep[i].hdr[H_SID] >> 9) > (sid >> 9)) ahead = true;
    for (uint32_t i = 0; i < E.group_size; i++) {
        if (i == E.leader || !((follow_mask >> i) & 1u) || !E.rep[i].ring) continue;
        uint64_t *fh = E.rep[i].hdr;
        fh[H_SID] = sid;                       /* heartbeat from the new leader, dare_server.c:822-920 */
        fh[H_TAIL] = E.log_len;
        fh[H_CID_BITMASK] = bitmask;
    }
    if (!ahead) { E.status[FENCE_WORD] = 0; atomicAnd(E.status, ~(1u << 2)); }
}


/* ---- election on the device ------------------------------------------------------------------ */
/* ELECT(winner) of the trace, in the schedule the oracle is pinned on (oracle/apus_oracle.c:orc_elect):
 * every live server that is not cut off becomes a candidate of term t+1 (start_election,
 * dare_server.c:1264-1322), the winner's timeout fires first (t+2) and its vote request reaches
 * every live configured server.  One lane per server decides as poll_vote_requests does
 * (:1526-1743): no vote from a leader or for a SID that is not newer; the candidate's last entry
 * (term, idx) must be at least as good as the voter's own, else the voter raises its term to the
 * candidate's and refuses (:1661-1673); otherwise it adopts the SID and grants.  poll_vote_count
 * (:1327-1518): the winner needs size/2+1 votes, its own included.
 * out: [0] elected, [1] grant mask, [2] refuse mask, [3] the winner's SID as a candidate, [4] votes */
__global__ void k_elect(const EngDev E, uint32_t winner, uint32_t live_mask, uint32_t bitmask, uint64_t *out)
{
    const uint32_t i = threadIdx.x;
    const uint64_t L = E.log_len;
    const bool mine = i < E.group_size && E.rep[i].ring && ((live_mask >> i) & 1u);
    uint64_t sid = 0, last_idx = 0, last_term = 0;
    if (mine) {
        const RepDev &R = E.rep[i];
        sid = R.hdr[H_SID];
        sid = (((sid >> 9) + 1) << 9) | i;                                   /* candidate of term t+1 */
        if (i == winner) sid = (((sid >> 9) + 1) << 9) | i;                  /* its timeout fires first: t+2 */
        const uint64_t n_end = R.hdr[H_N_END], end = R.hdr[H_END];
        if (end != L && n_end > 0) {                                         /* last_entry_of: log_get_tail + the entry there */
            const uint64_t off = R.dir_off[(uint32_t)(n_end - 1) & E.dir_mask];
            last_idx = ld8u(R.ring + off); last_term = ld8u(R.ring + off + 8);
        }
    }
    const uint64_t req_sid = rl64(sid, winner), req_idx = rl64(last_idx, winner), req_term = rl64(last_term, winner);
    bool grant = false, refuse = false;
    if (mine && i != winner && ((bitmask >> i) & 1u)) {
        const uint64_t old = sid | (1ull << 8);
        if (!((sid >> 8) & 1u) && old < req_sid) {
            if (last_term > req_term || (last_term == req_term && last_idx > req_idx)) {
                sid = ((req_sid >> 9) << 9) | i;                             /* raise own term, no vote */
                refuse = true;
            } else { sid = req_sid; grant = true; }
        }
    }
    if (mine) E.rep[i].hdr[H_SID] = sid;
    const unsigned long long gm = __ballot(grant), rm = __ballot(refuse);
    if (i == 0) {
        const uint32_t votes = 1u + (uint32_t)__popcll(gm);
        out[0] = (((live_mask >> winner) & 1u) && votes >= E.group_size / 2 + 1) ? 1 : 0;
        out[1] = gm; out[2] = rm; out[3] = req_sid; out[4] = votes;
    }
}

/* log_adjustment for the followers that granted their vote (dare_ibv_rc.c:1292-1451), one workgroup
 * (one lane) per follower: walk its not-committed entries (log_entries_to_nc_buf, dare_log.h:339:
 * from its commit to its end), compare (idx, term) with what the new leader's log holds at the same
 * offset (log_find_remote_end_offset, :367) and set the follower's end to the first offset that
 * differs.  Then the follower's own next polling() pass: its end now lies BEHIND its old_end, so
 * persist_new_entries (dare_server.c:1793-1810) walks from old_end forward -- over the rest of its
 * old entries, the untouched ring behind them, around through 0 -- "storing" and ACKing every entry
 * shape it meets until it arrives at the new end (reproduced: it moves old_end and counts as store
 * upcalls; pinned on the reference, tests/traces.py:double_failover_truncate). */
__global__ void k_adjust(const EngDev E, uint32_t mask, uint32_t alive_mask)
{
    if (threadIdx.x != 0) return;
    int f = -1;
    for (int i = 0, k = 0; i < APUS_DEV_MAX_SERVERS; i++)
        if (mask & (1u << i)) { if (k == (int)blockIdx.x) { f = i; break; } k++; }
    if (f < 0 || !E.rep[f].ring) return;
    const RepDev &F = E.rep[f], &Ld = E.rep[E.leader];
    uint64_t *fh = F.hdr;
    const uint64_t L = E.log_len;
    const uint64_t f_end = fh[H_END], l_end = Ld.hdr[H_END];
    const uint64_t n_commit = fh[H_N_COMMIT], n_end = fh[H_N_END];
    if (f_end == L || n_end <= n_commit) return;                 /* empty not-committed buffer: nothing to adjust */
    uint64_t off = 0, s = n_commit;
    for (; s < n_end; s++) {
        off = F.dir_off[(uint32_t)s & E.dir_mask];
        if (l_end == L || off == l_end) break;                   /* log_get_entry: nothing there in the leader's log */
        if (ld8u(Ld.ring + off) != ld8u(F.ring + off) || ld8u(Ld.ring + off + 8) != ld8u(F.ring + off + 8)) break;
        const uint32_t type = Ld.ring[off + 26];
        const uint64_t elen = APUS_HDR + ((type == APUS_NOOP || type == APUS_CONFIG || type == APUS_HEAD) ? 0u : (uint32_t)(Ld.ring[off + 48] | (Ld.ring[off + 49] << 8)));
        if (L - off < elen) off = 0;
        off += elen;
    }
    const uint64_t rem_end = off;
    if (rem_end == f_end) return;                                /* every entry it has is the leader's too */
    fh[H_END] = rem_end; fh[H_N_END] = s; fh[H_N_PERSIST] = s; fh[H_TAIL] = L;
    /* the follower's pass */
    uint64_t old_end = fh[H_OLD_END], count = 0;
    for (uint64_t guard = 0; guard < 2 * (L / APUS_HDR) + 16 && apus_is_larger(rem_end, L, rem_end, old_end); guard++) {
        if (L - old_end < APUS_HDR) old_end = 0;                 /* log_get_entry */
        const uint32_t type = F.ring[old_end + 26];
        const uint64_t elen = APUS_HDR + ((type == APUS_NOOP || type == APUS_CONFIG || type == APUS_HEAD) ? 0u : (uint32_t)(F.ring[old_end + 48] | (F.ring[old_end + 49] << 8)));
        if (L - old_end < elen) { old_end = 0; continue; }
        count++;
        F.ring[old_end + 28 + f] = 1;                            /* rc_send_entries_reply: own copy ... */
        const uint32_t sender = F.ring[old_end + 27];
        if (sender < E.group_size && ((alive_mask >> sender) & 1u) && E.rep[sender].ring)
            E.rep[sender].ring[old_end + 28 + f] = 1;            /* ... and the sender's */
        old_end += elen;
    }
    fh[H_OLD_END] = old_end;
    fh[H_STORE_COUNT] += count;
}

/* poll_vote_count (dare_server.c:1327-1518) + the first steps of log_adjustment (dare_ibv_rc.c:1357-1368): every vote ACK
 * carries the voter's commit offset and the new leader's commit moves up to the largest of them -- what a voter knows to
 * be committed IS committed, and an elected leader holds it.  In the pinned schedules every server learns a commit in the
 * pass that made it, so the voters' commits equal the winner's; with a leader killed MID-FLIGHT the commit doorbells have
 * reached the survivors at different times. */
__global__ void k_inherit(const EngDev E, uint32_t voters)
{
    if (threadIdx.x || blockIdx.x) return;
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *lh = Ld.hdr;
    if (lh[H_END] == E.log_len) return;
    uint64_t nc = lh[H_N_COMMIT];
    for (uint32_t m = voters & ~(1u << E.leader); m; m &= m - 1) {
        const uint32_t f = (uint32_t)__builtin_ctz(m);
        if (f >= APUS_DEV_MAX_SERVERS || !E.rep[f].ring) continue;
        const uint64_t c = E.rep[f].hdr[H_N_COMMIT];
        if (c > nc) nc = c;
    }
    if (nc > lh[H_N_END]) nc = lh[H_N_END];
    if (nc > lh[H_N_COMMIT]) {
        lh[H_COMMIT] = nc == lh[H_N_END] ? lh[H_END] : Ld.dir_off[(uint32_t)nc & E.dir_mask];
        lh[H_N_COMMIT] = nc;
    }
}

/* INHERITED entries (deviation 4, DESIGN.md section 6: liveness where the reference has none).  A follower acknowledges
 * an entry to the server that SENT it (entry->sender, dare_server.c:1806) and the commit scan counts the reply bytes in
 * the leader's own copy (dare_ibv_rc.c:1725-1758).  An entry a new leader holds from its predecessor -- in its own log
 * since before the election, or shipped to a follower that lagged by the new leader's catch-up -- was and is acknowledged
 * to the DEAD sender: the new leader's copy never shows a majority, the scan stops in front of it for good, and nothing
 * behind it commits either.  (The median-of-end-offsets rule of DARE that would cover it is dead code in the reference:
 * its result is overwritten at :1725.)  The pinned schedules never get there -- every server has learnt every commit when
 * the leader dies -- a leader killed with rounds in flight does: the followers hold rounds whose commit doorbell they
 * have not seen.
 * Here: once the new term's own entry (the blank CONFIG the leader has just appended) is held and acknowledged by a
 * majority -- the condition under which a leader may count replicas of older terms' entries at all -- every follower the
 * leader pushes to counts as having acknowledged what it holds of the leader's log (its log IS a prefix of the leader's:
 * log adjustment cut what differed, the catch-up wrote the rest).  Only the derived ACK words are set, never the reply
 * bytes in a ring: the logs stay what the reference's would be had the old leader lived one pass longer.
 * flag[0] != 0: bits were set, a commit pass must follow. */
__global__ __launch_bounds__(256) void k_reack(const EngDev E, uint32_t fmask, uint32_t *flag)
{
    const RepDev &Ld = E.rep[E.leader];
    const uint64_t *lh = Ld.hdr;
    if (lh[H_END] == E.log_len) return;
    const uint64_t n_end = lh[H_N_END], nc = lh[H_N_COMMIT];
    if (nc >= n_end) return;
    const uint32_t size = E.group_size, size_mask = (1u << size) - 1;
    const uint32_t last = __hip_atomic_load(&Ld.ack[(uint32_t)(n_end - 1) & E.dir_mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)__popc((last | (1u << E.leader)) & size_mask) < size / 2 + 1) return;
    bool any = false;
    for (uint32_t m = fmask & ~(1u << E.leader); m; m &= m - 1) {
        const uint32_t f = (uint32_t)__builtin_ctz(m);
        if (f >= APUS_DEV_MAX_SERVERS || !E.rep[f].ring) continue;
        uint64_t upto = E.rep[f].hdr[H_N_PERSIST];
        if (upto > n_end) upto = n_end;
        for (uint64_t sl = nc + threadIdx.x; sl < upto; sl += blockDim.x) {
            const uint32_t di = (uint32_t)sl & E.dir_mask;
            if (!((__hip_atomic_load(&Ld.ack[di], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> f) & 1u)) { atomicOr(&Ld.ack[di], 1u << f); any = true; }
        }
    }
    if (any) atomicOr(flag, 1u);
}

extern "C" int apus_gpu_elect(apus_engine_t *e, uint32_t winner, uint32_t live_mask, uint32_t bitmask, uint64_t out[8])
{
    if (!e || !out || winner >= e->d.group_size) return APUS_E_ARG;
    if (e->batching) return APUS_E_STATE;
    if (e->d.leader < e->d.group_size) { int frc = flush_tick(e); if (frc) return frc; }
    hipLaunchKernelGGL(k_elect, dim3(1), dim3(64), 0, e->stream, e->d, winner, live_mask & e->local_mask, bitmask, e->d_elect);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, e->d_elect, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (out[0]) { e->no_access = (uint32_t)out[2]; e->adjust_mask = (uint32_t)out[1]; }
    return 0;
}

/* The election itself (who wins which term) is the trace's ELECT event; this is what the winner
 * does in poll_vote_count (dare_server.c:1389-1421) and in the pass that follows.  `removed` =
 * servers that are ON in `bitmask` but have reached PERMANENT_FAILURE during the election (both
 * vote requests to a dead server fail, dare_ibv_rc.c:2747): check_failure_count (:1189-1227)
 * opens the new leader's first pass and appends their removal as a second CONFIG entry BEFORE
 * persist/commit run, so both entries commit in ONE pass (pinned on the reference itself,
 * tests/test_oracle_vs_refloops.py). */
extern "C" int apus_gpu_become_leader_ex(apus_engine_t *e, uint32_t leader, uint64_t term, uint32_t bitmask,
                                         uint32_t removed)
{
    if (!e || leader >= e->d.group_size) return APUS_E_ARG;
    if (e->batching) return APUS_E_STATE;           /* close the batch first (apus_gpu_batch_end) */
    if (!((e->local_mask >> leader) & 1u)) return APUS_E_STATE;
    if (e->d.leader < e->d.group_size) { int frc = flush_tick(e); if (frc) return frc; }
    e->tick_pending = false;
    e->free_lb = 0;                                  /* another server's ring from now on */
    e->d.leader = leader;
    const uint64_t sid = (term << 9) | (1ull << 8) | leader;
    hipLaunchKernelGGL(k_set_roles, dim3(1), dim3(64), 0, e->stream, e->d, sid, bitmask, e->reachable);
    HIPCHK(hipGetLastError());
    /* the ACK byte maps of a new leader start empty (entries of these slots may have been acknowledged to it in an earlier term) */
    if (!((e->imported_mask >> leader) & 1u))
        HIPCHK(hipMemsetAsync(e->d.ackb[leader], 0, (size_t)e->cfg.group_size * e->dir_cap, e->stream));
    {
        /* the followers that voted: their logs are adjusted before anything is replicated to them */
        const uint32_t am = e->adjust_mask & e->local_mask & e->reachable & ~(1u << leader);
        /* ... and they took this candidate's configuration with their vote (dare_server.c:1697) */
        hipLaunchKernelGGL(k_cfg_note, dim3(1), dim3(64), 0, e->stream, e->d_cfgj, e->adjust_mask & ~(1u << leader), bitmask);
        /* ... and told it how far they know the log to be committed */
        if (am) hipLaunchKernelGGL(k_inherit, dim3(1), dim3(64), 0, e->stream, e->d, am);
        e->adjust_mask = 0;
        if (am) {
            hipLaunchKernelGGL(k_adjust, dim3(popc(am)), dim3(64), 0, e->stream, e->d, am, e->reachable);
            HIPCHK(hipGetLastError());
            e->lag_possible = true;
        }
    }
    /* blank CONFIG entry: dare_cid_t {epoch, size[2], state, pad, bitmask} */
    uint8_t cid[16] = {0};
    memcpy(cid, &e->cid_epoch, 8);
    cid[8] = (uint8_t)e->d.group_size;
    memcpy(cid + 12, &bitmask, 4);
    removed &= bitmask & ~(1u << leader);
    int rc;
    if (!removed) rc = apus_gpu_append_control(e, APUS_CONFIG, cid);
    else {
        uint64_t d0, d1;
        memcpy(&d0, cid, 8); memcpy(&d1, cid + 8, 8);
        rc = launch_control_round(e, 3, APUS_CONFIG, d0, d1);        /* append only */
        if (rc) return rc;
        const uint32_t left = bitmask & ~removed;
        memcpy(cid + 12, &left, 4);
        rc = apus_gpu_append_control(e, APUS_CONFIG, cid);            /* second entry + the pass */
    }
    if (rc) return rc;
    /* what this leader INHERITED and could not commit in that pass (k_reack above): one look, and only when there was
     * something -- never in a pinned schedule -- one more pass that commits and applies it */
    {
        uint32_t *flag = (uint32_t *)(e->d_elect + 24);
        HIPCHK(hipMemsetAsync(flag, 0, sizeof(uint32_t), e->stream));
        hipLaunchKernelGGL(k_reack, dim3(1), dim3(256), 0, e->stream, e->d, sync_mask(e), flag);
        HIPCHK(hipGetLastError());
        uint32_t h = 0;
        HIPCHK(hipMemcpyAsync(&h, flag, sizeof h, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (h) rc = launch_control_round(e, 2, 0, 0, 0);
    }
    return rc;
}

extern "C" int apus_gpu_become_leader(apus_engine_t *e, uint32_t leader, uint64_t term, uint32_t bitmask)
{
    return apus_gpu_become_leader_ex(e, leader, term, bitmask, 0);
}

extern "C" int apus_gpu_set_reachable(apus_engine_t *e, uint32_t mask)
{
    if (!e) return APUS_E_ARG;
    if (e->batching) return APUS_E_STATE;           /* close the batch first (apus_gpu_batch_end) */
    { int frc = flush_tick(e); if (frc) return frc; }
    if (mask & ~e->reachable) e->lag_possible = true;     /* somebody was released */
    e->reachable = mask;
    e->d.reachable = mask;
    return 0;
}


/* ---- a server joins ------------------------------------------------------------ */
/* The reference: handle_server_join_request (dare_ibv_ud.c:973-1068) on the leader, the CONFIG
 * branches of apply_committed_entries (dare_server.c:1858-1937: reply to the joiner; the 3-phase resize
 * EXTENDED -> TRANSIT -> STABLE when the group is full), and on the joiner handle_server_join_reply
 * (:1071-1088), rc_recover_sm (dare_ibv_rc.c:597-705: apply = the donor's last applied entry),
 * rc_recover_log (:726-866: the bytes between the leader's head and a server's end, read in ONE
 * transfer), server_to_follower + vote ACK.  Here the joiner's replica is a slot of this engine (local
 * HBM, or a peer's mapped over xGMI -- the kernels do not care): k_join_prepare works out what the
 * joiner will hold, k_join_copy is the bulk transfer (ring range + directory, every CU), k_join_finish
 * is the joiner's first polling() pass: persist_new_entries with old_end still at len (log_new,
 * dare_log.h:134) -- a walk from offset 0 through the zeroed part of its ring in 64-byte steps and on
 * through the recovered entries, ACKing whatever it meets -- then apply_committed_entries from the
 * snapshot's offset.  Pinned on the reference: oracle/apus_oracle.c:orc_join, tests/traces.py join_*. */
enum { J_HEAD = 0, J_COPY_FROM, J_COPY_BYTES, J_S_FIRST, J_HEAD_SLOT, J_N_END, J_EMPTY, J_WRAP, J_WORDS = 8 };

__global__ __launch_bounds__(256) void k_join_prepare(const EngDev E, uint32_t r, uint32_t src, uint32_t donor,
                                                      uint32_t bitmask, uint64_t epoch, uint64_t *jw)
{
    __shared__ unsigned long long s_max[3];
    const uint32_t tid = threadIdx.x;
    const RepDev &Ld = E.rep[E.leader], &Td = E.rep[src], &Dd = E.rep[donor], &Jd = E.rep[r];
    const uint64_t L = E.log_len;
    const uint64_t head = Ld.hdr[H_HEAD];
    uint64_t rend = Td.hdr[H_END], rcommit = Td.hdr[H_COMMIT];
    const uint64_t tn_end = Td.hdr[H_N_END], tn_commit = Td.hdr[H_N_COMMIT], t_last = Td.hdr[H_LAST_IDX];
    const uint64_t dn_apply = Dd.hdr[H_N_APPLY], d_sid = Dd.hdr[H_SID], d_end = Dd.hdr[H_END], dn_end = Dd.hdr[H_N_END];
    if (tid < 3) s_max[tid] = 0;
    __syncthreads();
    const bool empty = rend == L;
    const bool wrap = !empty && rend > 0 && rend < head;
    /* slots: the newest one that starts at `head` (+1), the newest one that still lies in the old lap
     * (offset >= head) when the log wraps (+1), the donor's newest applied client entry (+1) */
    const uint64_t span = tn_end < E.dir_mask + 1ull ? tn_end : E.dir_mask + 1ull;
    for (uint64_t k = tid; k < span && !empty; k += blockDim.x) {
        const uint64_t sl = tn_end - 1 - k;
        const uint64_t off = Td.dir_off[(uint32_t)sl & E.dir_mask];
        if (off == head) atomicMax(&s_max[0], (unsigned long long)(sl + 1));
        if (wrap && off >= head) atomicMax(&s_max[1], (unsigned long long)(sl + 1));
    }
    __syncthreads();
    const uint64_t head_slot = s_max[0] ? s_max[0] - 1 : (tn_end > span ? tn_end - span : 0);
    for (uint64_t sl = head_slot + tid; sl < dn_apply && !empty; sl += blockDim.x) {
        const uint32_t type = Dd.ring[Dd.dir_off[(uint32_t)sl & E.dir_mask] + 26];
        if (type != APUS_NOOP && type != APUS_CONFIG && type != APUS_HEAD) atomicMax(&s_max[2], (unsigned long long)(sl + 1));
    }
    __syncthreads();
    if (tid != 0) return;
    uint64_t *jh = Jd.hdr;
    for (int i = 0; i < 64; i++) jh[i] = 0;
    jh[H_LEN] = L; jh[H_END] = L; jh[H_TAIL] = L; jh[H_OLD_END] = L;
    jh[H_HEAD] = head;                                                   /* handle_server_join_reply :1084 */
    jh[H_SID] = d_sid;                                                   /* poll_sm_reply, dare_server.c:671 */
    jh[H_CID_BITMASK] = bitmask; jh[H_CID_EPOCH] = epoch;
    Ld.hdr[H_APPLY_OFFSETS + r] = head;                                  /* handle_server_join_request :1051 */
    jw[J_HEAD] = head; jw[J_EMPTY] = empty; jw[J_WRAP] = wrap; jw[J_HEAD_SLOT] = head_slot;
    jw[J_COPY_FROM] = head; jw[J_COPY_BYTES] = 0; jw[J_S_FIRST] = 0; jw[J_N_END] = 0;
    if (empty) return;
    /* rc_recover_sm: apply = offset behind the donor's last applied client entry */
    uint64_t s_first = s_max[2] ? s_max[2] : head_slot;
    uint64_t apply_off;
    if (s_max[2]) {
        const uint32_t di = (uint32_t)(s_first - 1) & E.dir_mask;
        apply_off = Dd.dir_off[di] + (Dd.dir_len[di] & 0xFFFFFFu);
    } else apply_off = (s_first == dn_end) ? d_end : Dd.dir_off[(uint32_t)s_first & E.dir_mask];
    /* rc_recover_log :806-818 */
    uint64_t n_end = tn_end, n_commit = tn_commit;
    if (wrap) { rend = 0; n_end = s_max[1]; }
    if (apus_is_larger(Td.hdr[H_END], L, rcommit, rend)) { rcommit = rend; n_commit = n_end; }
    if (n_commit > n_end) n_commit = n_end;
    jh[H_END] = rend; jh[H_COMMIT] = rcommit; jh[H_APPLY] = apply_off;
    jh[H_N_END] = n_end; jh[H_N_COMMIT] = n_commit; jh[H_N_APPLY] = s_first;
    if (wrap && apply_off != 0 && apply_off < head) {
        /* The snapshot ends in the new lap, but the joiner's end = commit = 0: commit is "larger" than apply
         * (log_is_offset_larger, dare_log.h:269), so apply_committed_entries runs from there through the
         * zeroed ring (64-byte "NOOP" steps, no upcall) up to head and RE-APPLIES [head, len) on top of the
         * snapshot, and [0, apply) too once the new lap has arrived -- what the reference does (pinned,
         * tests/traces.py:join_wrapped); here: the joiner's apply position goes back to the head slot. */
        jh[H_N_APPLY] = head_slot;
        if ((head - apply_off) & 63) set_status(E, 1u << 5);            /* the steps do not land on head: see k_join_finish */
    }
    jh[H_LAST_IDX] = t_last;
    jh[H_JOINED_AT] = tn_end;
    jh[H_N_PERSIST] = head_slot;                                         /* nothing persisted yet; k_join_finish walks */
    jw[J_COPY_BYTES] = wrap ? L - head : rend - head;                    /* log_offset_end_distance(head) */
    jw[J_S_FIRST] = s_first; jw[J_N_END] = n_end;
}

/* rc_recover_log's READ (:826-855) and the directory that goes with the bytes; any grid */
__global__ __launch_bounds__(256) void k_join_copy(const EngDev E, uint32_t r, uint32_t src, const uint64_t *jw)
{
    if (jw[J_EMPTY]) return;
    const RepDev &Td = E.rep[src], &Jd = E.rep[r];
    const uint64_t from = jw[J_COPY_FROM], bytes = jw[J_COPY_BYTES];
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t a = (from + 15) & ~15ull, b = (from + bytes) & ~15ull;
    if (b > a) for (uint64_t u = tid; u < (b - a) / 16; u += nth) st16u(Jd.ring + a + 16 * u, ld16u(Td.ring + a + 16 * u));
    for (uint64_t x = from + tid; x < from + bytes && x < a; x += nth) Jd.ring[x] = Td.ring[x];
    for (uint64_t x = (b > a ? b : a) + tid; x < from + bytes; x += nth) Jd.ring[x] = Td.ring[x];
    for (uint64_t i = tid; i <= E.dir_mask; i += nth) { Jd.dir_off[i] = Td.dir_off[i]; Jd.dir_len[i] = Td.dir_len[i]; }
}

/* the joiner's first pass: persist_new_entries from old_end = len (dare_server.c:1793-1810) */
__global__ __launch_bounds__(256) void k_join_finish(const EngDev E, uint32_t r, uint32_t alive_mask, const uint64_t *jw)
{
    if (jw[J_EMPTY]) return;
    const RepDev &Jd = E.rep[r];
    uint64_t *jh = Jd.hdr;
    const uint64_t L = E.log_len;
    const uint32_t tid = threadIdx.x;
    const uint64_t head = jw[J_HEAD], head_slot = jw[J_HEAD_SLOT], n_end = jw[J_N_END];
    const uint64_t end = jh[H_END];
    const bool zero_alive = E.rep[0].ring && ((alive_mask >> 0) & 1u) && E.group_size > 0;
    if (jw[J_WRAP] || end == 0) {
        /* end = 0 and old_end = len are the same place (log_is_offset_larger, dare_log.h:269): the pass has
         * nothing to do; the new lap arrives with the leader's log update and is persisted like any other */
        if (tid == 0) jh[H_N_PERSIST] = n_end;
        return;
    }
    if ((head & 63) == 0) {
        /* the 64-byte steps through the zeroed part [0, head) land exactly on head, from there the walk
         * follows the entries: head / 64 "NOOP entries" of sender 0, then slots [head_slot, n_end) */
        for (uint64_t k = tid; k < head / 64; k += blockDim.x) {
            Jd.ring[k * 64 + 28 + r] = 1;
            if (zero_alive && r != 0) E.rep[0].ring[k * 64 + 28 + r] = 1;
        }
        for (uint64_t sl = head_slot + tid; sl < n_end; sl += blockDim.x) {
            const uint64_t off = Jd.dir_off[(uint32_t)sl & E.dir_mask];
            Jd.ring[off + 28 + r] = 1;                                   /* rc_send_entries_reply: own copy ... */
            const uint32_t sender = Jd.ring[off + 27];
            if (sender != r && sender < E.group_size && ((alive_mask >> sender) & 1u) && E.rep[sender].ring)
                E.rep[sender].ring[off + 28 + r] = 1;                    /* ... and the sender's */
        }
        if (tid == 0) {
            jh[H_STORE_COUNT] = head / 64 + (n_end - head_slot);
            jh[H_OLD_END] = end; jh[H_N_PERSIST] = n_end;
        }
        return;
    }
    if (tid != 0) return;
    /* head is not a multiple of 64: the walk runs into the recovered entries misaligned and reads entry
     * shapes out of payload bytes.  The reference's joiner only leaves this loop if the walk happens to
     * land exactly on `end` (otherwise it spins in polling() forever, oracle/apus_oracle.c:orc_join -8);
     * followed step by step, bounded to four laps -- then the pass is taken as done (status bit) */
    uint64_t old_end = L, count = 0, guard = 0;
    const uint64_t limit = L / 16 + 1024;
    while (apus_is_larger(end, L, end, old_end) && ++guard <= limit) {
        if (L - old_end < APUS_HDR) old_end = 0;                         /* log_get_entry */
        const uint32_t type = Jd.ring[old_end + 26];
        const uint64_t elen = APUS_HDR + ((type == APUS_NOOP || type == APUS_CONFIG || type == APUS_HEAD) ? 0u : (uint32_t)(Jd.ring[old_end + 48] | (Jd.ring[old_end + 49] << 8)));
        if (L - old_end < elen) { old_end = 0; continue; }
        count++;
        Jd.ring[old_end + 28 + r] = 1;
        const uint32_t sender = Jd.ring[old_end + 27];
        if (sender != r && sender < E.group_size && ((alive_mask >> sender) & 1u) && E.rep[sender].ring)
            E.rep[sender].ring[old_end + 28 + r] = 1;
        old_end += elen;
    }
    if (guard > limit) { set_status(E, 1u << 5); old_end = end; count = n_end - head_slot; }
    jh[H_STORE_COUNT] = count; jh[H_OLD_END] = old_end; jh[H_N_PERSIST] = n_end;
}

/* JOIN(r): a new machine with LID `lid` joins the group this engine leads and gets slot r -- the lowest
 * slot that is OFF in `bitmask`, or the current group size when every slot is taken (the group is
 * extended; r must be below the capacity the engine was created with).  `reachable` = who answers (the
 * donor of the snapshot is the first follower in index order, the source of the log the first server).
 * out[0] = the new bitmask, out[1] = the new group size, out[2] = the new epoch.  One per-round record
 * for the whole join.  APUS_E_STATE: no leader, a batch is open, r is not the slot the leader would hand
 * out, or no follower to recover from. */
extern "C" int apus_gpu_join(apus_engine_t *e, uint32_t r, uint16_t lid, uint32_t bitmask, uint32_t reachable, uint64_t out[4])
{
    if (!e || !out || r >= e->cfg.group_size) return APUS_E_ARG;
    if (e->live_R) { int rc_ = flush_live(e); if (rc_) return rc_; }
    if (e->batching) return APUS_E_STATE;
    int rc = need_leader(e);
    if (rc) return rc;
    if ((rc = flush_tick(e))) return rc;
    const uint32_t size = e->d.group_size, leader = e->d.leader;
    uint32_t empty = size;
    for (int i = (int)size - 1; i >= 0; i--) if (!((bitmask >> i) & 1u)) empty = (uint32_t)i;   /* dare_ibv_ud.c:995-1021 */
    if (empty != r || !e->d.rep[r].ring) return APUS_E_STATE;
    /* a configured server that does not answer: the joiner needs RC connections to, and the replicated vote from, a
     * majority and retries until it has them (rc_get_replicated_vote dare_ibv_rc.c:874); the oracle's JOIN is one
     * event and covers joins into a fully reachable group (orc_join, -6): so does this one */
    for (uint32_t i = 0; i < size; i++)
        if (i != r && ((bitmask >> i) & 1u) && !((reachable >> i) & 1u)) return APUS_E_NOANSWER;
    const bool peer_owned = (e->imported_mask >> r) & 1u;      /* its process cleared it (apus_gpu_clear_replica) */
    int donor = -1, src = -1;
    for (uint32_t i = 0; i < size; i++) {
        if (i == r || !((bitmask >> i) & 1u) || !((reachable >> i) & 1u) || !e->d.rep[i].ring) continue;
        if (src < 0) src = (int)i;
        if (donor < 0 && i != leader) donor = (int)i;
    }
    if (donor < 0 || src < 0) return APUS_E_STATE;

    /* the journal of CONFIG entries so far (apus_members.h): what every member itself holds before this JOIN */
    uint64_t cfg_n0 = 0;
    HIPCHK(hipMemcpyAsync(&cfg_n0, &e->d_cfgj->n, sizeof cfg_n0, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));

    /* leader side: the CONFIG entries, each committed by the pass that follows it */
    uint8_t cid[16] = {0};
    uint64_t d0, d1;
    uint32_t nb = bitmask | (1u << r);
    auto put = [&](uint64_t epoch, uint32_t s0, uint32_t s1, uint32_t state) {
        memset(cid, 0, 16); memcpy(cid, &epoch, 8); cid[8] = (uint8_t)s0; cid[9] = (uint8_t)s1; cid[10] = (uint8_t)state;
        memcpy(cid + 12, &nb, 4); memcpy(&d0, cid, 8); memcpy(&d1, cid + 8, 8);
    };
    const uint64_t epoch0 = e->cid_epoch;
    if (r < size) {                                           /* Case 3: an empty place */
        put(e->cid_epoch, size, 0, 0);
        if ((rc = launch_control_round(e, 0 | 16, APUS_CONFIG, d0, d1, 1, lid))) return rc;
    } else {                                                  /* Case 4: [N,0,STABLE] -> [N,N+1,EXTENDED] -> [N,N+1,TRANSIT] -> [N+1,0,STABLE] */
        put(epoch0 + 1, size, size + 1, 2);
        if ((rc = launch_control_round(e, 0 | 16, APUS_CONFIG, d0, d1, 1, lid))) return rc;    /* old majority (EXTENDED) */
        e->cid_epoch = epoch0 + 1;
        e->d.group_size = size + 1;                           /* TRANSIT: the scan runs with cid.size[1] */
        put(e->cid_epoch, size, size + 1, 1);
        rc = launch_control_round(e, 0 | 16, APUS_CONFIG, d0, d1, 0, 0);
        put(e->cid_epoch, size + 1, 0, 0);
        if (!rc) rc = launch_control_round(e, 0 | 16, APUS_CONFIG, d0, d1, 0, 0);
        if (rc) return rc;                                    /* (a launch that cannot be issued: the stream is broken anyway) */
    }
    /* Who answers the joiner's RC_SYN (handle_rc_syn, dare_ibv_ud.c): a member whose OWN configuration has the
     * joiner's bit ON -- it took in one of the entries above: a server that itself joined ignores CONFIG entries whose
     * idx is not above the idx of the entry that admitted it (poll_config_entries dare_server.c:2152) -- and did not
     * still show the slot's former holder before.  The joiner needs more than half of the group it joins and retries
     * for ever otherwise (oracle/apus_oracle.c:orc_join, -6): refused here, with the CONFIG entries in the log as
     * they are in the reference's. */
    static_assert(sizeof(CfgJournal) < (1u << 16), "the journal is copied to the host in one piece");
    CfgJournal *H = (CfgJournal *)malloc(sizeof(CfgJournal));
    if (!H) return APUS_E_NOMEM;
    if (hipMemcpyAsync(H, e->d_cfgj, sizeof(CfgJournal), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) { free(H); return APUS_E_HIP; }
    const uint64_t cfg_n1 = H->n;
    {
        const uint32_t jsize = r < size ? size : size + 1;
        const uint32_t conn = join_answers(*H, e->mv, r, leader, nb, reachable, jsize, cfg_n0, cfg_n1);
        if (cfg_n1 == cfg_n0 || conn <= jsize / 2) {
            /* refused AFTER the CONFIG entries went into the log (as in the reference, whose joiner then retries for ever): the
             * configuration the device now holds goes back to the caller, who must adopt it -- its mirrors (bitmask, size,
             * epoch) would otherwise disagree with the device on every later call */
            free(H);
            if (cfg_n1 != cfg_n0) { out[0] = nb; out[1] = e->d.group_size; out[2] = e->cid_epoch; out[3] = 0; }
            return cfg_n1 == cfg_n0 ? APUS_E_FULL : APUS_E_NOANSWER;
        }
    }
    const uint64_t join_cid_idx = H->it[cfg_n0 % CFGJ_CAP].idx;
    free(H);

    /* joiner side */
    uint64_t *jw = e->d_elect + 8;                            /* scratch words behind k_elect's verdict */
    if (!peer_owned) {
        HIPCHK(hipMemsetAsync(e->d.rep[r].ring, 0, e->d.log_len + 4096, e->stream));   /* log_new() */
        HIPCHK(hipMemsetAsync(e->d.rep[r].ack, 0, sizeof(uint32_t) * e->dir_cap, e->stream));
    }
    hipLaunchKernelGGL(k_join_prepare, dim3(1), dim3(256), 0, e->stream, e->d, r, (uint32_t)src, (uint32_t)donor, nb, e->cid_epoch, jw);
    hipLaunchKernelGGL(k_join_copy, dim3(512), dim3(256), 0, e->stream, e->d, r, (uint32_t)src, jw);
    hipLaunchKernelGGL(k_join_finish, dim3(1), dim3(256), 0, e->stream, e->d, r, reachable | (1u << r), jw);
    HIPCHK(hipGetLastError());
    /* connected: vote ACK, log adjustment (nothing to cut: the joiner's log is a prefix of the leader's),
     * log update + lazy commit + the joiner's apply in the pass that closes the join */
    e->reachable |= 1u << r; e->d.reachable = e->reachable;
    e->lag_possible = true;
    {   /* the slot's former holder told this leader how far IT had got: a new machine starts over */
        RepBox *lb = e->d.box[leader];
        uint64_t *words[5] = { &lb->seqdone_by[r], &lb->persisted_by[r], &lb->applied_by[r], &lb->apply_off_by[r], &lb->sid_by[r] };
        for (auto w : words) HIPCHK(hipMemsetAsync(w, 0, sizeof(uint64_t), e->stream));
    }
    if ((rc = launch_control_round(e, 2 | 32, 0, 0, 0))) return rc;
    {   /* the new member's own configuration from now on: the one of the join reply, then whatever it polls from the
         * head it was given (apus_members.h) */
        uint64_t head_slot = 0;
        HIPCHK(hipMemcpyAsync(&head_slot, &jw[J_HEAD_SLOT], sizeof head_slot, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        e->mv[r] = MemberView{nb, cfg_n1, head_slot, join_cid_idx};
    }
    out[0] = nb; out[1] = e->d.group_size; out[2] = e->cid_epoch; out[3] = join_cid_idx;
    return 0;
}

/* ---- force_log_pruning (dare_server.c:2069-2122) ------------------------------------------------ */
/* Closes every leader pass in the reference: once the log is 75 % full the server whose sampled apply
 * offset holds the head back is REMOVED from the configuration (a CONFIG entry; the slip at :2113 --
 * apply_offsets[size] instead of [target] -- included), then log_pruning runs (head, <HEAD> entry, the
 * apply offsets are read again).  What it appends is committed by the NEXT pass.  One workgroup; the
 * decision on the device, from the leader's control block.  out: [0] 1 when the log was 75 % full,
 * [1] the server removed (0xFF none), [2] entries appended, [3] the bitmask afterwards. */
__global__ __launch_bounds__(128) void k_force_prune(const EngDev E, uint64_t epoch, uint32_t sample_mask, uint64_t *out)
{
    __shared__ uint64_t s_lh[64];
    __shared__ uint32_t s_go;
    const uint32_t tid = threadIdx.x;
    const RepDev &Ld = E.rep[E.leader];
    uint64_t *hdr = Ld.hdr;
    const uint64_t L = E.log_len;
    if (tid < 64) s_lh[tid] = hdr[tid];
    if (tid == 0) s_go = 0;
    __syncthreads();
    if (tid == 0) {
        out[0] = 0; out[1] = 0xFF; out[2] = 0; out[3] = s_lh[H_CID_BITMASK];
        const uint64_t end = s_lh[H_END];
        const uint64_t log_size = apus_end_distance(end, L, s_lh[H_HEAD]);
        if (!((double)log_size < 0.75 * (double)L)) {                           /* :2075 */
            out[0] = 1; s_go = 1;
            const uint32_t size = E.group_size;
            uint32_t target = E.leader, bitmask = (uint32_t)s_lh[H_CID_BITMASK];
            uint64_t min_off = s_lh[H_APPLY];
            for (uint32_t i = 0; i < size; i++)
                if (apus_is_larger(end, L, min_off, s_lh[H_APPLY_OFFSETS + i])) { min_off = s_lh[H_APPLY_OFFSETS + i]; target = i; }
            uint32_t appended = 0;
            if (target != E.leader && ((bitmask >> target) & 1u)) {
                bitmask &= ~(1u << target);                                     /* CID_SERVER_RM + dare_ib_disconnect_server */
                const uint64_t d1 = (uint64_t)size | ((uint64_t)bitmask << 32); /* size[0], size[1] = 0, state STABLE, pad, bitmask */
                appended += control_append<true>(E, 0, APUS_CONFIG, epoch, d1, 0, s_lh, *E.rec_count, 0, false).n;
                s_lh[H_CID_BITMASK] = bitmask;
                if (size < APUS_DEV_MAX_SERVERS) { s_lh[H_APPLY_OFFSETS + size] = s_lh[H_APPLY]; hdr[H_APPLY_OFFSETS + size] = s_lh[H_APPLY]; }
                out[1] = target;
            }
            appended += control_append<true>(E, 1, APUS_HEAD, 0, 0, 0, s_lh, *E.rec_count, 0, false).n;   /* log_pruning :1996-2067 */
            out[2] = appended; out[3] = bitmask;
            /* the reference takes this pass's end / commit record behind force_log_pruning */
            const uint64_t rc = *E.rec_count;
            if (appended && rc > 0 && rc - 1 < E.rec_cap) E.rec_end[rc - 1] = s_lh[H_END];
        }
    }
}

/* rc_get_remote_apply_offsets (dare_ibv_rc.c:1970-2034) behind it, with the configuration as it is now: a
 * launch of its own -- one lane per server, the leader's control block read afresh */
__global__ __launch_bounds__(64) void k_force_sample(const EngDev E, uint32_t sample_mask, const uint64_t *out)
{
    __shared__ uint64_t s_lh[64];
    if (!out[0]) return;
    s_lh[threadIdx.x] = E.rep[E.leader].hdr[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < APUS_DEV_MAX_SERVERS) sample_apply_offsets(E, s_lh, sample_mask, threadIdx.x, nullptr);
}

extern "C" int apus_gpu_force_prune(apus_engine_t *e, uint64_t out[4])
{
    if (!e || !out) return APUS_E_ARG;
    if (e->live_R) { int rc_ = flush_live(e); if (rc_) return rc_; }
    if (e->batching) return APUS_E_STATE;
    int rc = need_leader(e);
    if (rc) return rc;
    if ((rc = flush_tick(e))) return rc;
    hipLaunchKernelGGL(k_force_prune, dim3(1), dim3(128), 0, e->stream, e->d, e->cid_epoch, sync_mask(e), e->d_elect + 16);
    hipLaunchKernelGGL(k_cfg_journal, dim3(1), dim3(64), 0, e->stream, e->d, e->d_cfgj);       /* (it may have appended a removal) */
    hipLaunchKernelGGL(k_force_sample, dim3(1), dim3(64), 0, e->stream, e->d, sync_mask(e), e->d_elect + 16);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, e->d_elect + 16, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (out[2]) { e->free_lb = 0; e->lag_possible = true; }
    if (out[1] != 0xFF) { e->reachable &= ~(1u << out[1]); e->d.reachable = e->reachable; }
    return 0;
}

/* ---- graphs ----------------------------------------------------------------- */
extern "C" int apus_gpu_capture_begin(apus_engine_t *e)
{
    if (!e || e->capturing) return APUS_E_STATE;
    { int frc = flush_tick(e); if (frc) return frc; }
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    e->capturing = true;
    return 0;
}

extern "C" int apus_gpu_capture_end(apus_engine_t *e, int *graph_id)
{
    if (e && e->batching) return APUS_E_STATE;      /* close the batch first (apus_gpu_batch_end) */
    if (!e || !e->capturing || !graph_id) return APUS_E_STATE;
    { int frc = flush_tick(e); if (frc) return frc; }
    hipGraph_t g = nullptr;
    e->capturing = false;
    HIPCHK(hipStreamEndCapture(e->stream, &g));
    hipGraphExec_t ge = nullptr;
    HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipGraphDestroy(g);
    e->graphs.push_back(ge);
    *graph_id = (int)e->graphs.size() - 1;
    return 0;
}

extern "C" int apus_gpu_graph_launch(apus_engine_t *e, int graph_id)
{
    if (e && e->batching) return APUS_E_STATE;      /* close the batch first (apus_gpu_batch_end) */
    if (!e || graph_id < 0 || graph_id >= (int)e->graphs.size()) return APUS_E_ARG;
    HIPCHK(hipGraphLaunch(e->graphs[graph_id], e->stream));
    e->free_lb = 0;                                  /* (what the replay appends is not accounted: read back next time) */
    return 0;
}

/* ---- observation ------------------------------------------------------------ */
static int local_rep(apus_engine *e, uint32_t r)
{
    if (!e || r >= e->cfg.group_size || !e->d.rep[r].ring) return APUS_E_STATE;
    return 0;
}

extern "C" int apus_gpu_offsets(apus_engine_t *e, uint32_t replica, uint64_t out[8])
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    { int frc = flush_tick(e); if (frc) return frc; }
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, e->d.rep[replica].hdr, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int apus_gpu_counters(apus_engine_t *e, uint32_t replica, uint64_t out[8])
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    { int frc = flush_tick(e); if (frc) return frc; }
    uint64_t h[64];
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(h, e->d.rep[replica].hdr, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[H_N_END]; out[1] = h[H_N_PERSIST]; out[2] = h[H_N_COMMIT]; out[3] = h[H_N_APPLY];
    out[4] = h[H_LAST_IDX]; out[5] = h[H_SID]; out[6] = h[H_HIGHEST_REC]; out[7] = h[H_APPLY_HASH];
    return 0;
}

extern "C" int apus_gpu_hdr_words(apus_engine_t *e, uint32_t replica, uint64_t *out, uint32_t n_words)
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    { int frc = flush_tick(e); if (frc) return frc; }
    if (n_words > 64) n_words = 64;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, e->d.rep[replica].hdr, n_words * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int apus_gpu_read_ring(apus_engine_t *e, uint32_t replica, uint64_t off, uint64_t n, void *dst)
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    { int frc = flush_tick(e); if (frc) return frc; }
    if (off + n > e->d.log_len) return APUS_E_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(dst, e->d.rep[replica].ring + off, n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" uint64_t apus_gpu_round_count(apus_engine_t *e)
{
    if (!e) return 0;
    flush_tick(e);
    uint64_t n = 0;
    hipStreamSynchronize(e->stream);
    hipMemcpy(&n, e->d.rec_count, sizeof n, hipMemcpyDeviceToHost);
    return n;
}

extern "C" int apus_gpu_round_record(apus_engine_t *e, uint64_t first, uint64_t n, uint64_t *end_out, uint64_t *commit_out)
{
    if (!e || first + n > e->d.rec_cap) return APUS_E_ARG;
    { int frc = flush_tick(e); if (frc) return frc; }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (end_out) HIPCHK(hipMemcpy(end_out, e->d.rec_end + first, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (commit_out) HIPCHK(hipMemcpy(commit_out, e->d.rec_commit + first, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int apus_gpu_apply_records(apus_engine_t *e, uint32_t replica, uint64_t first, uint64_t n, apus_apply_t *out)
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    { int frc = flush_tick(e); if (frc) return frc; }
    if (n > e->dir_cap) return APUS_E_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    const apus_apply_rec *base = e->d.rep[replica].apply;
    for (uint64_t done = 0; done < n;) {
        const uint64_t i = (first + done) & e->d.dir_mask;
        const uint64_t chunk = (n - done < e->dir_cap - i) ? n - done : e->dir_cap - i;
        HIPCHK(hipMemcpy(out + done, base + i, chunk * sizeof(apus_apply_rec), hipMemcpyDeviceToHost));
        done += chunk;
    }
    return 0;
}


/* ---- durability side channel --------------------------------------------------- */
/* What persist_new_entries hands to proxy_store_cmd for every entry (dare_server.c:1802: &entry->clt_id)
 * and stablestorage_save_request (src/proxy/proxy.c:268-291) appends to BerkeleyDB: a CONNECT / CLOSE
 * record is the 4 bytes clt_id, type, sender; a SEND record is sizeof(proxy_send_msg) = 24 bytes from
 * clt_id on (sender, reply[13], the struct padding) plus "data.cmd.len" more -- which the overlay reads
 * at +8 of the record = reply[4] | reply[5] << 8, not the entry's cmd.len: no command
 * byte is stored unless servers 4 / 5 had acknowledged.  Back to back these records are dump_records'
 * output, i.e. the snapshot a joiner's donor ships (stablestorage_dump_records, proxy.c:300-304).
 * The callback sees the entry AS IT IS WHEN THIS SERVER PERSISTS IT: the leader right behind the append
 * (reply[] zero, `sender` not stamped yet); a follower after the bytes landed (the sender's stamp, the
 * replies the leader had collected when it sent them -- none in step, others' after a catch-up -- and
 * its own reply byte still zero).  A follower's ring keeps exactly those bytes (ACKs go to its own byte and
 * to the sender's copy), so the records are regenerated from the ring: own reply byte cleared; all
 * replies cleared where this server appended the entry itself.  One workgroup: block scan of the record
 * lengths, then each thread copies its record. */
__global__ __launch_bounds__(1024) void k_store_stream(const EngDev E, uint32_t r, uint64_t first, uint64_t n,
                                                       uint8_t *out, uint64_t cap, uint64_t *res)
{
    __shared__ uint64_t s_tot[16];
    __shared__ uint64_t s_base, s_recs;
    const RepDev &Rd = E.rep[r];
    const uint64_t joined_at = Rd.hdr[H_JOINED_AT];
    if (threadIdx.x == 0) { s_base = 0; s_recs = 0; }
    __syncthreads();
    for (uint64_t c0 = 0; c0 < n; c0 += blockDim.x) {
        const uint64_t k = c0 + threadIdx.x;
        uint64_t off = 0; uint32_t len = 0, type = 0, sender = 0;
        uint8_t rep[13];
        for (int i = 0; i < 13; i++) rep[i] = 0;
        if (k < n) {
            off = Rd.dir_off[(uint32_t)(first + k) & E.dir_mask];
            type = Rd.ring[off + 26]; sender = Rd.ring[off + 27];
            if (type == APUS_CONNECT || type == APUS_CLOSE) len = 4;
            else if (type == APUS_SEND) {
                const bool own = sender == r && first + k >= joined_at;      /* appended by this very machine */
                if (!own) { for (int i = 0; i < 13; i++) rep[i] = Rd.ring[off + 28 + i]; if (r < 13) rep[r] = 0; }
                len = 24u + (uint32_t)(rep[4] | (rep[5] << 8));
            }
        }
        uint64_t tot;
        const uint64_t incl = block_incl_scan(len, s_tot, &tot);
        const uint64_t at = s_base + incl - len;
        if (len && at + len <= cap) {
            uint8_t *o = out + at;
            o[0] = Rd.ring[off + 24]; o[1] = Rd.ring[off + 25]; o[2] = (uint8_t)type; o[3] = (uint8_t)sender;
            if (len > 4) {
                for (int i = 0; i < 13; i++) o[4 + i] = rep[i];
                for (uint32_t i = 17; i < len; i++) o[i] = Rd.ring[off + 24 + i];      /* padding 41..47, then what follows */
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_base += tot; }
        if (len) atomicAdd((unsigned long long *)&s_recs, 1ull);
        __syncthreads();
    }
    if (threadIdx.x == 0) { res[0] = s_base; res[1] = s_recs; }
}

/* records of entry slots [first, first + n) of `replica`, back to back, into dst (host memory, cap bytes);
 * *bytes = what they take (even when > cap: nothing beyond cap is written), *records = how many */
extern "C" int apus_gpu_store_stream(apus_engine_t *e, uint32_t replica, uint64_t first, uint64_t n,
                                     void *dst, uint64_t cap, uint64_t *bytes, uint64_t *records)
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    if (!bytes || n > e->dir_cap || (cap && !dst)) return APUS_E_ARG;
    if (e->batching) return APUS_E_STATE;
    if ((rc = flush_tick(e))) return rc;
    /* a scratch buffer kept between calls (two hipMalloc / hipFree per call synchronise the device), zeroed on the engine's
     * stream: a record that straddles `cap` is skipped as a whole by the kernel, the bytes behind the last record written are
     * zeros, never stale device memory */
    if (e->ss_cap < cap + 32) {
        if (e->ss_buf) { hipStreamSynchronize(e->stream); hipFree(e->ss_buf); e->ss_buf = nullptr; e->ss_cap = 0; }
        const uint64_t want = std::max<uint64_t>(cap + 32, 1u << 20);
        if (hipMalloc((void **)&e->ss_buf, want) != hipSuccess) return APUS_E_NOMEM;
        e->ss_cap = want;
    }
    uint8_t *d_out = e->ss_buf + 16; uint64_t *d_res = (uint64_t *)e->ss_buf;
    hipError_t he = hipMemsetAsync(e->ss_buf, 0, cap + 32, e->stream);
    if (he == hipSuccess) hipLaunchKernelGGL(k_store_stream, dim3(1), dim3(1024), 0, e->stream, e->d, replica, first, n, d_out, cap, d_res);
    uint64_t res[2] = {0, 0};
    if (he == hipSuccess) he = hipGetLastError();
    if (he == hipSuccess) he = hipMemcpyAsync(res, d_res, 16, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess && cap) he = hipMemcpy(dst, d_out, res[0] < cap ? res[0] : cap, hipMemcpyDeviceToHost);
    if (he != hipSuccess) return APUS_E_HIP;
    *bytes = res[0];
    if (records) *records = res[1];
    return 0;
}

extern "C" uint32_t apus_gpu_status(apus_engine_t *e)
{
    if (!e) return 0xFFFFFFFFu;
    flush_tick(e);
    uint32_t s[2] = {0, 0};
    hipStreamSynchronize(e->stream);
    hipMemcpy(s, e->d.status, sizeof s, hipMemcpyDeviceToHost);
    if ((s[0] & (1u << 4)) && getenv("APUS_DEBUG")) fprintf(stderr, "[apus] spin timeout first hit at apus_kernels.h:%u\n", s[1]);
    return s[0] | e->host_status;
}

/* diagnostics: the eight status words (bits, first spin-timeout site, fence word, five words a kernel left behind) */
extern "C" int apus_gpu_status_words(apus_engine_t *e, uint32_t out[8])
{
    if (!e || !out) return APUS_E_ARG;
    hipStreamSynchronize(e->stream);
    HIPCHK(hipMemcpy(out, e->d.status, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void apus_gpu_clear_status(apus_engine_t *e)
{
    if (!e) return;
    hipStreamSynchronize(e->stream);
    hipMemset(e->d.status, 0, 8 * sizeof(uint32_t));      /* bits, spin site, fence word, diagnostics */
    e->host_status = 0;
}

extern "C" void *apus_gpu_device_ptr(apus_engine_t *e, uint32_t replica, int which, uint64_t *bytes)
{
    if (local_rep(e, replica)) return nullptr;
    RepDev &r = e->d.rep[replica];
    uint64_t b = 0; void *p = nullptr;
    switch (which) {
    case 0: p = r.ring; b = e->d.log_len; break;
    case 1: p = r.hdr; b = 64 * sizeof(uint64_t); break;
    case 2: p = r.dir_off; b = sizeof(uint64_t) * e->dir_cap; break;
    case 3: p = r.dir_len; b = sizeof(uint32_t) * e->dir_cap; break;
    case 4: p = r.ack; b = sizeof(uint32_t) * e->dir_cap; break;
    case 5: p = r.apply; b = sizeof(apus_apply_rec) * (uint64_t)e->dir_cap; break;
    default: break;
    }
    if (bytes) *bytes = b;
    return p;
}

extern "C" int apus_gpu_set_timing(apus_engine_t *e, int on)
{
    if (!e) return APUS_E_ARG;
    e->timing = on != 0;
    e->timed_used = 0;
    return 0;
}

extern "C" int apus_gpu_kernel_time(apus_engine_t *e, int which, float *total_ms, uint64_t *launches)
{
    if (!e || which != 0) return APUS_E_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    float tot = 0.f;
    for (size_t i = 0; i < e->timed_used; i++) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->timed[i].a, e->timed[i].b));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = e->timed_used;
    e->timed_used = 0;
    return 0;
}

extern "C" void *apus_gpu_stream(apus_engine_t *e) { return e ? (void *)e->stream : nullptr; }

extern "C" int apus_gpu_bind_global(apus_engine_t *e) { g_engine = e; return 0; }
extern "C" apus_engine_t *apus_gpu_global(void) { return g_engine; }


/* ---- persistent consensus kernel ---------------------------------------------- */
__global__ void k_persist_init(const EngDev E, PersistDev *D)
{
    const int i = threadIdx.x;
    if (i >= APUS_DEV_MAX_SERVERS) return;
    if (i == 0) { D->quit = 0; D->lat_n = 0; }
    if (i < (int)E.group_size && E.rep[i].ring) {
        D->end_bell[i] = E.rep[i].hdr[H_N_PERSIST];
        D->commit_bell[i] = E.rep[i].hdr[H_N_APPLY];
        D->applied_bell[i] = E.rep[i].hdr[H_N_APPLY];
    }
}

static double mono_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

extern "C" int apus_gpu_persist_start(apus_engine_t *e, uint32_t idle_ms, uint32_t peer_ms)
{
    int rc = need_leader(e);
    if (rc) return rc;
    if (ref_quirks(e)) return APUS_E_STATE;      /* APUS_F_REF_QUIRKS covers the call-per-pass path only */
    { int frc = flush_tick(e); if (frc) return frc; }
    if (e->p_running) return APUS_E_STATE;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!e->ph) {
        HIPCHK(hipHostMalloc((void **)&e->ph, sizeof(PersistHost), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer((void **)&e->ph_dev, e->ph, 0));
        HIPCHK(hipMalloc((void **)&e->pd, sizeof(PersistDev)));
        HIPCHK(hipStreamCreateWithFlags(&e->pstream, hipStreamNonBlocking));
    }
    memset((void *)e->ph, 0, offsetof(PersistHost, ev));
    HIPCHK(hipMemset(e->pd, 0, sizeof(PersistDev)));
    e->p_ev_tail = e->p_req_tail = 0; e->p_arena_pos = 0;
    uint64_t h[64];
    HIPCHK(hipMemcpy(h, e->d.rep[e->d.leader].hdr, sizeof h, hipMemcpyDeviceToHost));
    e->ph->highest_rec = h[H_HIGHEST_REC];
    e->ph->commit_slot = h[H_N_COMMIT];
    hipLaunchKernelGGL(k_persist_init, dim3(1), dim3(64), 0, e->pstream, e->d, e->pd);
    /* one poll costs roughly 1 us (s_sleep + an uncached load) */
    const uint64_t idle_polls = (uint64_t)idle_ms * 1000ull;
    const uint64_t peer_polls = (uint64_t)peer_ms * 1000ull;
    const uint32_t fm = sync_mask(e);
    hipLaunchKernelGGL(k_consensus_persistent, dim3(popc(e->local_mask)), dim3(256), 0, e->pstream,
                       e->d, e->ph_dev, e->pd, e->local_mask, fm, idle_polls, peer_polls);
    HIPCHK(hipGetLastError());
    const double t0 = mono_s();
    while (e->ph->alive == 0) {
        if (mono_s() - t0 > 5.0) { fprintf(stderr, "[apus_gpu] persistent kernel did not start\n"); return APUS_E_HIP; }
    }
    e->p_running = true;
    return 0;
}

static int persist_push_event(apus_engine *e, uint32_t op, uint32_t n, uint64_t req_first, uint64_t req_end,
                              uint32_t arena_off = 0, uint32_t arena_bytes = 0)
{
    const double t0 = mono_s();
    while (e->p_ev_tail - e->ph->ev_head >= P_EV_CAP - 1) {
        if (e->ph->alive == 2 || mono_s() - t0 > 5.0) return APUS_E_STATE;
    }
    PEvent &ev = e->ph->ev[e->p_ev_tail % P_EV_CAP];
    ev.op = op; ev.n = n; ev.req_first = req_first; ev.arena_off = arena_off; ev.arena_bytes = arena_bytes;
    e->p_req_end[e->p_ev_tail % P_EV_CAP] = req_end;
    e->p_ev_tail++;
    __atomic_store_n((uint64_t *)&e->ph->ev_tail, e->p_ev_tail, __ATOMIC_RELEASE);
    return 0;
}

/* requests consumed by the kernel so far */
static uint64_t persist_req_consumed(apus_engine *e)
{
    const uint64_t head = e->ph->ev_head;
    return head ? e->p_req_end[(head - 1) % P_EV_CAP] : 0;
}

extern "C" int apus_gpu_persist_submit(apus_engine_t *e, const apus_req_t *reqs, uint32_t n,
                                       const uint8_t *arena, uint64_t arena_bytes)
{
    if (!e || !e->p_running || !reqs) return APUS_E_STATE;
    (void)arena_bytes;
    for (uint32_t g0 = 0; g0 < n; g0 += APUS_MAX_ROUND) {
        const uint32_t nr = (n - g0 < APUS_MAX_ROUND) ? n - g0 : APUS_MAX_ROUND;
        uint64_t need = 0;
        for (uint32_t k = 0; k < nr; k++) need += ((uint64_t)reqs[g0 + k].len + 15) & ~15ull;
        const double t0 = mono_s();
        /* payload ring: restart at 0 when the batch does not fit, once everything before was consumed */
        if (e->p_arena_pos + need + 32 > P_ARENA_CAP) {
            while (e->ph->ev_head < e->p_ev_tail) if (e->ph->alive == 2 || mono_s() - t0 > 5.0) return APUS_E_STATE;
            e->p_arena_pos = 0;
        }
        while (e->p_req_tail + nr - persist_req_consumed(e) > P_REQ_CAP)
            if (e->ph->alive == 2 || mono_s() - t0 > 5.0) return APUS_E_STATE;
        const uint64_t batch_pos = 16 + e->p_arena_pos;
        for (uint32_t k = 0; k < nr; k++) {
            const apus_req_t &q = reqs[g0 + k];
            if (q.type == APUS_NOOP || q.type == APUS_CONFIG || q.type == APUS_HEAD || q.type > 15) return APUS_E_ARG;
            const uint64_t slot = (e->p_req_tail + k) % P_REQ_CAP;
            const uint64_t pos = 16 + e->p_arena_pos;                 /* byte -2 of a payload must exist */
            if (q.len) memcpy((void *)(e->ph->arena + pos), arena + q.payload_off, q.len);
            ReqDev d;
            d.req_id = q.req_id;
            d.pay16_type = (uint32_t)(pos / 16) | ((uint32_t)q.type << 28);
            d.len = q.len; d.clt_id = q.clt_id;
            e->ph->req[slot] = d;
            e->ph->req_len[slot] = q.len;
            e->p_arena_pos += ((uint64_t)q.len + 15) & ~15ull;
        }
        const uint64_t first = e->p_req_tail % P_REQ_CAP;
        e->p_req_tail += nr;
        int rc = persist_push_event(e, P_OP_ROUND, nr, first, e->p_req_tail, (uint32_t)batch_pos,
                                    (uint32_t)(16 + e->p_arena_pos - batch_pos));
        if (rc) return rc;
    }
    return 0;
}

extern "C" int apus_gpu_persist_prune(apus_engine_t *e)
{
    if (!e || !e->p_running) return APUS_E_STATE;
    return persist_push_event(e, P_OP_PRUNE, 0, 0, e->p_req_tail);
}

/* block until the kernel consumed every published event (or timeout_ms) */
extern "C" int apus_gpu_persist_drain(apus_engine_t *e, uint32_t timeout_ms)
{
    if (!e || !e->p_running) return APUS_E_STATE;
    const double t0 = mono_s();
    while (e->ph->ev_head < e->p_ev_tail) {
        if (e->ph->alive == 2) return APUS_E_STATE;
        if ((mono_s() - t0) * 1e3 > timeout_ms) return -1;
    }
    return 0;
}

/* 1 once the persistent kernel refused a round because the log is full (the requests of that round
 * are dropped: their submitters are never released by highest_rec) */
extern "C" int apus_gpu_persist_full(apus_engine_t *e) { return (e && e->ph) ? (int)e->ph->full : 0; }
extern "C" uint64_t apus_gpu_persist_highest_rec(apus_engine_t *e) { return (e && e->ph) ? e->ph->highest_rec : 0; }
extern "C" const volatile uint64_t *apus_gpu_persist_highest_rec_ptr(apus_engine_t *e) { return (e && e->ph) ? &e->ph->highest_rec : nullptr; }

extern "C" int apus_gpu_persist_stop(apus_engine_t *e)
{
    if (!e || !e->p_running) return APUS_E_STATE;
    __atomic_store_n((uint64_t *)&e->ph->stop, 1ull, __ATOMIC_RELEASE);
    HIPCHK(hipStreamSynchronize(e->pstream));
    e->p_running = false;
    return (int)e->ph->exit_code;
}

/* append -> commit latency samples of the persistent kernel, in nanoseconds */
extern "C" int apus_gpu_persist_latency(apus_engine_t *e, uint32_t *out_ns, uint32_t cap, uint32_t *n_out)
{
    if (!e || !e->pd) return APUS_E_STATE;
    if (e->p_running) return APUS_E_STATE;
    uint32_t n = 0;
    HIPCHK(hipMemcpy(&n, &e->pd->lat_n, sizeof n, hipMemcpyDeviceToHost));
    if (n > cap) n = cap;
    if (n) HIPCHK(hipMemcpy(out_ns, e->pd->lat_ticks, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int khz = 100000;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->cfg.device);
    if (khz <= 0) khz = 100000;
    for (uint32_t i = 0; i < n; i++) out_ns[i] = (uint32_t)((uint64_t)out_ns[i] * 1000000ull / (uint64_t)khz);
    if (n_out) *n_out = n;
    return 0;
}

/* App-visible latency of the live path, measured in C: `iters` times, submit one round of
 * n <= 64 requests and spin on highest_rec (exactly what proxy.c:160 does); out_ns[i] =
 * submit -> released */
extern "C" int apus_gpu_persist_roundtrip(apus_engine_t *e, const apus_req_t *reqs, uint32_t n,
                                          const uint8_t *arena, uint64_t arena_bytes,
                                          uint32_t iters, uint32_t *out_ns)
{
    if (!e || !e->p_running || n == 0 || n > APUS_MAX_ROUND) return APUS_E_STATE;
    for (uint32_t i = 0; i < iters; i++) {
        const uint64_t target = e->ph->highest_rec + n;
        const double t0 = mono_s();
        int rc = apus_gpu_persist_submit(e, reqs, n, arena, arena_bytes);
        if (rc) return rc;
        while (e->ph->highest_rec < target) {
            if (e->ph->alive == 2 || mono_s() - t0 > 2.0) return -1;
        }
        out_ns[i] = (uint32_t)((mono_s() - t0) * 1e9);
    }
    return 0;
}

/* phase breakdown of the same samples: which = 1: event seen -> sequenced, 2: -> pushed + doorbell */
extern "C" int apus_gpu_persist_latency_phase(apus_engine_t *e, int which, uint32_t *out_ns, uint32_t cap, uint32_t *n_out)
{
    if (!e || !e->pd || e->p_running || (which != 1 && which != 2)) return APUS_E_STATE;
    uint32_t n = 0;
    HIPCHK(hipMemcpy(&n, &e->pd->lat_n, sizeof n, hipMemcpyDeviceToHost));
    if (n > cap) n = cap;
    const uint32_t *src = which == 1 ? e->pd->lat_seq : e->pd->lat_push;
    if (n) HIPCHK(hipMemcpy(out_ns, src, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int khz = 100000;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->cfg.device);
    if (khz <= 0) khz = 100000;
    for (uint32_t i = 0; i < n; i++) out_ns[i] = (uint32_t)((uint64_t)out_ns[i] * 1000000ull / (uint64_t)khz);
    if (n_out) *n_out = n;
    return 0;
}

extern "C" int apus_gpu_device_arch(int device, char *out, int cap)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return APUS_E_HIP;
    snprintf(out, cap, "%s", p.gcnArchName);
    return 0;
}


/* ---- multi-process groups: one replica per GPU / process ------------------------ */
/* follower process: adopt the leader's SID (heartbeat, dare_server.c:822-920) */
__global__ void k_follow(const EngDev E, uint32_t f, uint64_t sid, uint32_t bitmask)
{
    if (threadIdx.x || blockIdx.x) return;
    uint64_t *fh = E.rep[f].hdr;
    fh[H_SID] = sid; fh[H_TAIL] = E.log_len; fh[H_CID_BITMASK] = bitmask;
}

/* a server adopts a SID it heard of: its vote for a candidate (poll_vote_requests, dare_server.c:1690) or
 * the heartbeat of a leader of a newer term (hb_receive_cb :903-910), when that candidate / leader is
 * driven by ANOTHER engine; from then on launches of a leader of an older term are fenced off */
__global__ void k_adopt_sid(const EngDev E, uint32_t f, uint64_t sid)
{
    if (threadIdx.x || blockIdx.x) return;
    if (E.rep[f].hdr[H_SID] < sid) E.rep[f].hdr[H_SID] = sid;
}

extern "C" int apus_gpu_adopt_sid(apus_engine_t *e, uint32_t replica, uint64_t sid)
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    if (e->batching) return APUS_E_STATE;
    hipLaunchKernelGGL(k_adopt_sid, dim3(1), dim3(64), 0, e->stream, e->d, replica, sid);
    HIPCHK(hipGetLastError());
    return 0;
}

/* a process that hosts followers only learns who leads from its control plane (the election's messages in the
 * reference): nothing is launched, the replica kernels it starts afterwards look the leader's mailbox / log up by it */
extern "C" int apus_gpu_set_leader(apus_engine_t *e, uint32_t leader)
{
    if (!e || leader >= e->d.group_size) return APUS_E_ARG;
    if (e->r_running || e->p_running || e->batching) return APUS_E_STATE;
    e->d.leader = leader;
    return 0;
}

extern "C" int apus_gpu_follow(apus_engine_t *e, uint32_t replica, uint32_t leader, uint64_t term, uint32_t bitmask)
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    if (leader >= e->d.group_size || leader == replica) return APUS_E_ARG;
    e->d.leader = leader;
    hipLaunchKernelGGL(k_follow, dim3(1), dim3(64), 0, e->stream, e->d, replica,
                       (term << 9) | (1ull << 8) | leader, bitmask);
    HIPCHK(hipGetLastError());
    return 0;
}

/* leader: get_tailq_message + log_append_entry only (the tail follows after the ACKs came back) */
extern "C" int apus_gpu_append_rounds(apus_engine_t *e, uint64_t r0, uint64_t n_rounds)
{
    int rc = need_leader(e);
    if (rc) return rc;
    if (r0 + n_rounds > e->n_rounds_staged || n_rounds > e->max_rounds || n_rounds == 0) return APUS_E_ARG;
    return launch_append(e, e->d, r0, (uint32_t)n_rounds);
}

extern "C" int apus_gpu_commit_rounds(apus_engine_t *e, uint64_t r0, uint64_t n_rounds)
{
    int rc = need_leader(e);
    if (rc) return rc;
    if (r0 + n_rounds > e->n_rounds_staged || n_rounds == 0) return APUS_E_ARG;
    const uint64_t n = e->h_round_first[r0 + n_rounds] - e->h_round_first[r0];
    return launch_tail(e, r0, (uint32_t)n_rounds, 0, n);
}

/* out[8] = end offset before the append, end after, slot before, slot after,
 *          visible slot (== slot after unless the batch ended exactly on len), kstar >= 0,
 *          leader commit slot, leader commit offset   (synchronises) */
extern "C" int apus_gpu_ship_info(apus_engine_t *e, uint64_t out[8])
{
    int rc = need_leader(e);
    if (rc) return rc;
    SeqOut s;
    uint64_t h[64];
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(&s, e->d.seq, sizeof s, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h, e->d.rep[e->d.leader].hdr, sizeof h, hipMemcpyDeviceToHost));
    out[0] = s.e0; out[1] = h[H_END]; out[2] = s.n_end0; out[3] = h[H_N_END];
    out[4] = (h[H_END] == e->d.log_len) ? s.n_end0 : h[H_N_END];
    out[5] = s.kstar >= 0 ? 1 : 0;
    out[6] = h[H_N_COMMIT]; out[7] = h[H_COMMIT];
    return 0;
}

extern "C" int apus_gpu_ingest(apus_engine_t *e, uint32_t replica, uint64_t vis_slot, uint64_t n_hint)
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mp_ingest, dim3(cap_grid(n_hint ? n_hint : 1, 256, 1024)), dim3(256), 0, e->stream, e->d, replica, vis_slot);
    hipLaunchKernelGGL(k_mp_ingest_fin, dim3(1), dim3(64), 0, e->stream, e->d, replica, vis_slot);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int apus_gpu_ack_merge(apus_engine_t *e, uint32_t follower, uint64_t from_slot, uint64_t upto_slot)
{
    int rc = need_leader(e);
    if (rc) return rc;
    if (follower >= e->d.group_size || upto_slot < from_slot) return APUS_E_ARG;
    if (upto_slot == from_slot) return 0;
    hipLaunchKernelGGL(k_mp_ack_merge, dim3(cap_grid(upto_slot - from_slot, 256, 1024)), dim3(256), 0, e->stream,
                       e->d, follower, from_slot, upto_slot);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int apus_gpu_follower_commit(apus_engine_t *e, uint32_t replica, uint64_t commit_slot, uint64_t n_hint)
{
    int rc = local_rep(e, replica);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mp_apply, dim3(cap_grid(n_hint ? n_hint : 1, 256, 1024)), dim3(256), 0, e->stream, e->d, replica, commit_slot);
    hipLaunchKernelGGL(k_mp_apply_fin, dim3(1), dim3(64), 0, e->stream, e->d, replica, commit_slot);
    HIPCHK(hipGetLastError());
    return 0;
}


/* ---- replica kernels: every replica runs its own resident workgroups (apus_replica.h) -------------- */
/* Which followers are "in step" with the leader -- hold everything it has appended, an exact-fit round
 * they hold back included -- and can take the rounds of a run: the others need the leader's catch-up
 * (update_remote_logs step I, the control-plane pass) first. */
static int rep_in_step(apus_engine *e, uint32_t cand, uint32_t *out_mask, uint64_t qbase[APUS_MAX_SERVERS], uint64_t fruns[APUS_MAX_SERVERS])
{
    uint64_t lh[64];
    HIPCHK(hipMemcpy(lh, e->d.rep[e->d.leader].hdr, sizeof lh, hipMemcpyDeviceToHost));
    uint32_t in = 0;
    for (uint32_t m = cand; m; m &= m - 1) {
        const uint32_t f = (uint32_t)__builtin_ctz(m);
        uint64_t fh[64], notes[8];
        HIPCHK(hipMemcpy(fh, e->d.rep[f].hdr, sizeof fh, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(notes, &e->d.box[f]->f_seq_next, sizeof notes, hipMemcpyDeviceToHost));   /* f_seq_next, pend x 3, f_exit, f_runs */
        qbase[f] = notes[0]; fruns[f] = notes[5];
        uint64_t n_end = fh[H_N_END];
        if (notes[2] > notes[1] && notes[3] == fh[H_SID] && notes[1] == fh[H_N_END]) n_end = notes[2];
        if (n_end == lh[H_N_END] && (fh[H_SID] >> 9) == (lh[H_SID] >> 9)) in |= 1u << f;
        else if (n_end < lh[H_N_END]) {
            /* behind: whatever reply bytes rode with entries it never acknowledged (R_BELL_REPLY) do not stand */
            const uint64_t cap = (uint64_t)e->d.dir_mask + 1;
            const uint64_t s0 = lh[H_N_END] - n_end > cap ? lh[H_N_END] - cap : n_end;
            hipLaunchKernelGGL(k_rep_clear_reply, dim3((unsigned)std::min<uint64_t>(256, (lh[H_N_END] - s0 + 255) / 256)), dim3(256), 0, e->stream,
                               e->d, e->d.leader, f, s0, lh[H_N_END], lh[H_HEAD], lh[H_END], lh[H_LAST_IDX]);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    *out_mask = in;
    return 0;
}

extern "C" int apus_gpu_rep_start(apus_engine_t *e, uint32_t idle_ms, uint32_t peer_ms, uint32_t n_append, uint32_t n_fwork)
{
    if (!e || e->d.leader >= e->d.group_size) return APUS_E_STATE;
    if (e->r_running || e->p_running || e->batching || ref_quirks(e)) return APUS_E_STATE;
    if (e->live_R) { int rc_ = flush_live(e); if (rc_) return rc_; }
    const uint32_t leader = e->d.leader;
    const uint32_t hosted = e->local_mask & ~e->imported_mask;
    const bool lead_here = (hosted >> leader) & 1u;
    if (lead_here) { int frc = flush_tick(e); if (frc) return frc; }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!e->rstream) HIPCHK(hipStreamCreateWithFlags(&e->rstream, hipStreamNonBlocking));
    RepArgs A;
    memset(&A, 0, sizeof A);
    const uint32_t members = ((1u << e->d.group_size) - 1) & ~(1u << leader);
    A.follow_mask = hosted & members & e->reachable & ~e->r_test_skip;
    {
        /* every workgroup of the launch must be resident at once: the grid is cut to what the device holds.  The kernel is
         * chosen by what this process hosts (apus_replica.h, "the launch"): one role alone gets the kernel compiled for it. */
        const uint32_t nfh = (uint32_t)popc(A.follow_mask);
        const void *kern = lead_here && !nfh ? (const void *)k_replica_leader : !lead_here && nfh == 1 ? (const void *)k_replica_follower : (const void *)k_replica;
        int occ = 0, cus = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0));
        HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->cfg.device));
        const uint32_t room = (uint32_t)std::max(8, occ * cus - 8);
        { const char *ga = getenv("APUS_REP_DEFAULT_APPEND"), *gf = getenv("APUS_REP_DEFAULT_FWORK");      /* (sweeps) */
          if (!n_append && ga) n_append = (uint32_t)atoi(ga);
          if (!n_fwork && gf) n_fwork = (uint32_t)atoi(gf); }
        /* (measured, round 4: 192 append + 128 per follower at 3 replicas, 72 at 5, 48 at 7 for configs[1]'s 128-byte entries.
         *  Round 5: with entries of 512 bytes and more -- configs[2], configs[3] -- an append wavefront is bound by the rate at
         *  which it issues its own stores (170 per round of 32 x 1 KiB at five replicas, ~240 ns each), and the followers, who no
         *  longer read headers back, need few workgroups: 352 append + ~150 for all followers: +19 % / +6 %) */
        /*  Round 6, three workgroups per compute unit (760 slots): one replica 320 append workgroups (8.1 G entries/s; 256: 7.8, 448: 7.1 --
         *  the serial roles' passes get longer with every wavefront that shares their SIMDs); three replicas 384 + 96 per follower
         *  (4.6 G; 320 + 128: 4.6, 256 + 128: 4.2, 448 + 64: 3.9 -- too few follower workgroups and the doorbell rings run full); five
         *  384 + 64 (3.1 G; 320 + 96: 2.9); seven 384 + 48 (2.2 G; 320 + 64: 1.9).  profiles/r06_grid_sweeps.txt) */
        const bool big = e->n_rounds_staged && e->stage_max_T >= 512;
        if (!n_append) n_append = lead_here ? (big ? 352u : (nfh ? 384u : 320u)) : 0;
        if (!n_fwork) n_fwork = nfh ? (big ? std::min(128u, std::max(16u, 150u / nfh)) : std::min(96u, std::max(48u, 288u / nfh))) : 1;
        while ((lead_here ? 1 + n_append : 0) + nfh * n_fwork > room && (n_append > 8 || n_fwork > 2)) {
            if (n_append > 8) n_append -= n_append / 4;
            if (n_fwork > 2) n_fwork -= (n_fwork + 3) / 4;
        }
        if (!n_fwork) n_fwork = 1;
        if (getenv("APUS_DEBUG")) {
            for (uint32_t i = 0; i < e->d.group_size; i++) fprintf(stderr, "[apus_gpu]   replica %u: ring %p mailbox %p hdr %p%s\n", i, (void *)e->d.rep[i].ring, (void *)e->d.box[i], (void *)e->d.rep[i].hdr, ((e->imported_mask >> i) & 1u) ? " (mapped)" : "");
        }
        if (getenv("APUS_DEBUG")) fprintf(stderr, "[apus_gpu] replica launch: %s, %d workgroups per CU; grid %u append, %u per follower x %u\n",
                                          kern == (const void *)k_replica ? "k_replica" : lead_here ? "k_replica_leader" : "k_replica_follower", occ, n_append, n_fwork, nfh);
    }
    A.n_append = n_append; A.n_fwork = n_fwork;
    A.idle_polls = (uint64_t)idle_ms * 1000ull;
    A.peer_polls = (uint64_t)peer_ms * 1000ull;
    A.lead_here = lead_here ? 1u : 0u;
    { const char *dbg = getenv("APUS_REP_DBG"); A.dbg = dbg ? (uint32_t)atoi(dbg) : 0u; }
    if (lead_here) {
        if (!e->rh) {
            HIPCHK(hipHostMalloc((void **)&e->rh, sizeof(RepHost), hipHostMallocMapped | hipHostMallocCoherent));
            HIPCHK(hipHostGetDevicePointer((void **)&e->rh_dev, e->rh, 0));
            memset((void *)e->rh, 0, sizeof(RepHost));
            /* the rings the host fills: in device memory when the host can store into it (large BAR) */
            int large_bar = 0;
            const char *rr = getenv("APUS_REQ_RING");
            if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, e->cfg.device) != hipSuccess) { large_bar = 0; (void)hipGetLastError(); }
            if (large_bar && !(rr && !strcmp(rr, "host"))) {
                void *p = nullptr;
                if (hipExtMallocWithFlags(&p, sizeof(RepReq), hipDeviceMallocUncached) == hipSuccess) {
                    e->rq = e->rq_dev = (RepReq *)p; e->rq_bar = true;
                    HIPCHK(hipMemset(p, 0, offsetof(RepReq, slot)));
                } else (void)hipGetLastError();
            }
            if (!e->rq) {
                HIPCHK(hipHostMalloc((void **)&e->rq, sizeof(RepReq), hipHostMallocMapped | hipHostMallocCoherent));
                HIPCHK(hipHostGetDevicePointer((void **)&e->rq_dev, e->rq, 0));
                memset((void *)e->rq, 0, offsetof(RepReq, slot));
            }
            HIPCHK(hipMalloc((void **)&e->rl, sizeof(RepLead)));
            e->r_slot_aend = (uint64_t *)calloc(RQ_CAP, sizeof(uint64_t));
            e->r_win_cnt = (uint32_t *)calloc(RQ_CAP / WAVE, sizeof(uint32_t));
            e->r_win_len = (uint32_t *)calloc(RQ_CAP / WAVE, sizeof(uint32_t));
            if (!e->r_slot_aend || !e->r_win_cnt || !e->r_win_len) return APUS_E_NOMEM;
        }
        const uint32_t cand = sync_mask(e);
        uint32_t push = 0;
        int rc = rep_in_step(e, cand, &push, A.qbase, A.fruns);
        if (rc) return rc;
        A.push_mask = push; A.park_mask = cand;
        if (getenv("APUS_DEBUG")) { fprintf(stderr, "[apus_gpu] leader %u starts a run: push %#x of %#x;", leader, push, cand); for (uint32_t m = cand; m; m &= m - 1) fprintf(stderr, " q%u=%llu", __builtin_ctz(m), (unsigned long long)A.qbase[__builtin_ctz(m)]); fprintf(stderr, "\n"); }
        if (push != cand) e->lag_possible = true;
        HIPCHK(hipMemsetAsync(e->rl, 0, sizeof(RepLead), e->rstream));
        HIPCHK(hipMemsetAsync(&e->rl->seq_final, 0xFF, sizeof(uint64_t), e->rstream));
        HIPCHK(hipMemsetAsync(e->rl->t_drop, 0xFF, sizeof e->rl->t_drop, e->rstream));
        uint64_t h[64];
        HIPCHK(hipMemcpy(h, e->d.rep[leader].hdr, sizeof h, hipMemcpyDeviceToHost));
        /* (a run that ended abnormally may have left commands or slots behind: they are dropped) */
        e->rh->cmd_head = e->r_cmd_tail; e->rh->slots_done = e->r_slot_tail; e->r_done_seen = e->r_slot_tail;
        /* the windows' accounts start empty; the window the ring's tail stands in is charged with the slots in front of the tail
         * (used up by earlier runs) and never gets a word: its count still comes to 64 and is cleared for the next lap */
        for (uint32_t w = 0; w < RQ_CAP / WAVE; w++) { e->r_win_cnt[w] = 0; e->r_win_len[w] = R_WIN_UNSET; }
        if (e->r_slot_tail % WAVE) { e->r_win_cnt[(e->r_slot_tail / WAVE) % (RQ_CAP / WAVE)] = (uint32_t)(e->r_slot_tail % WAVE); e->r_win_len[(e->r_slot_tail / WAVE) % (RQ_CAP / WAVE)] = R_WIN_MIXED; }
        e->rh->settled = e->r_cmd_tail + e->r_slot_tail;
        e->rq->stop = 0; e->rh->alive = 0; e->rh->exit_code = 0; e->rh->full = 0; e->rh->rounds = 0;
        e->rh->highest_rec = h[H_HIGHEST_REC];
        e->rh->commit_slot = h[H_N_COMMIT];
        A.H = e->rh_dev; A.RQ = e->rq_dev; A.LS = e->rl;
    }
    for (uint32_t m = A.follow_mask; m; m &= m - 1) {
        const uint32_t f = (uint32_t)__builtin_ctz(m);
        if (!e->rfs[f]) HIPCHK(hipMalloc((void **)&e->rfs[f], sizeof(RepFollow)));
        HIPCHK(hipMemsetAsync(e->rfs[f], 0, sizeof(RepFollow), e->rstream));
        A.FS[f] = e->rfs[f];
        if (!e->rfh[f]) {
            HIPCHK(hipHostMalloc((void **)&e->rfh[f], sizeof(RepFHost), hipHostMallocMapped | hipHostMallocCoherent));
            HIPCHK(hipHostGetDevicePointer((void **)&e->rfh_dev[f], e->rfh[f], 0));
        }
        memset((void *)e->rfh[f], 0, sizeof(RepFHost));
        e->rfh[f]->consumer = e->r_consumer[f]; e->rfh[f]->replayed = e->r_replayed[f];
        A.FH[f] = e->rfh_dev[f];
    }
    const uint32_t nfh_l = (uint32_t)popc(A.follow_mask);
    const uint32_t grid = (lead_here ? 1 + n_append : 0) + nfh_l * n_fwork;
    if (!grid) return 0;                               /* nothing of this group runs here */
    if (!e->rev0) { HIPCHK(hipEventCreate(&e->rev0)); HIPCHK(hipEventCreate(&e->rev1)); }
    HIPCHK(hipEventRecord(e->rev0, e->rstream));
    if (lead_here && !nfh_l) hipLaunchKernelGGL(k_replica_leader, dim3(grid), dim3(256), 0, e->rstream, e->d, A);
    else if (!lead_here && nfh_l == 1) hipLaunchKernelGGL(k_replica_follower, dim3(grid), dim3(256), 0, e->rstream, e->d, A, (uint32_t)__builtin_ctz(A.follow_mask));
    else hipLaunchKernelGGL(k_replica, dim3(grid), dim3(256), 0, e->rstream, e->d, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->rev1, e->rstream));
    e->rev_valid = true;
    if (lead_here) {
        const double t0 = mono_s();
        while (e->rh->alive == 0)
            if (mono_s() - t0 > 10.0) { fprintf(stderr, "[apus_gpu] the leader's workgroups did not start\n"); return APUS_E_HIP; }
    }
    e->r_running = true; e->r_lead = lead_here; e->r_follow_mask = A.follow_mask;
    e->r_push_mask = A.push_mask; e->r_cand_mask = A.park_mask;
    for (uint32_t f = 0; f < APUS_MAX_SERVERS && f < APUS_DEV_MAX_SERVERS; f++) e->r_fruns[f] = A.fruns[f];
    return 0;
}

/* the last run the leader started here: out[0] = followers that get its rounds (they held everything when it began), out[1] = followers
 * it could reach -- a follower in out[1] and not in out[0] takes no part until the next control-plane pass has caught it up */
extern "C" int apus_gpu_rep_push_info(apus_engine_t *e, uint32_t out[2])
{
    if (!e || !out) return APUS_E_ARG;
    out[0] = e->r_push_mask; out[1] = e->r_cand_mask;
    return 0;
}

static int rep_push_cmd(apus_engine *e, uint32_t op, uint64_t a, uint64_t b)
{
    if (!e || !e->r_running || !e->r_lead) return APUS_E_STATE;
    const double t0 = mono_s();
    pthread_spin_lock(&e->r_lock);
    while (e->r_cmd_tail - e->rh->cmd_head >= RC_CAP - 1)
        if (e->rh->alive == 2 || mono_s() - t0 > 5.0) {
            fprintf(stderr, "[apus_gpu] command ring: no room for command %llu (op %u) after %.1f s: carried out %llu, alive %llu\n",
                    (unsigned long long)e->r_cmd_tail, op, mono_s() - t0, (unsigned long long)e->rh->cmd_head, (unsigned long long)e->rh->alive);
            pthread_spin_unlock(&e->r_lock);
            return APUS_E_STATE;
        }
    /* four self-tagged granules {command number + 1 : value}: the command is there once all four are */
    RepCmd &c = e->rq->cmd[e->r_cmd_tail % RC_CAP];
    const uint64_t tag = ((e->r_cmd_tail + 1) & 0xFFFFFFFFull) << 32;
    const uint64_t vals[4] = { op, __atomic_load_n(&e->r_slot_tail, __ATOMIC_ACQUIRE) & 0xFFFFFFFFull, a & 0xFFFFFFFFull, b & 0xFFFFFFFFull };
    for (int i = 3; i >= 0; i--) __atomic_store_n((uint64_t *)&c.g[i], tag | vals[i], __ATOMIC_RELEASE);
    if (e->rq_bar) __builtin_ia32_sfence();          /* (write-combined stores through the BAR: on their way now) */
    e->r_cmd_tail++;
    pthread_spin_unlock(&e->r_lock);
    return 0;
}

/* Admission, multi-producer (replaces the malloc'd TAILQ + tailq_lock of leader_handle_submit_req,
 * src/proxy/proxy.c:108-161): a producer RESERVES the next request slot -- and, for a payload that does not fit
 * into the slot itself, a range of the pinned payload arena -- in a short critical section (two counters),
 * copies its payload there itself, then PUBLISHES the slot.  The leader's sequencer takes published slots in
 * slot order, up to 64 per round.  *dst = where the len payload bytes go. */
/* Short payloads (<= R_INLINE bytes: they live in the slot itself) reserve WITHOUT a lock: one fetch-and-add hands out
 * `n` consecutive slots -- their place in the log order -- then the caller waits until the ring has room for them (slots
 * are consumed in order, so a producer only ever waits for producers in front of it).  What the slot remembers of the
 * payload arena is its tail as read BEFORE the fetch-and-add: an arena allocation takes its slot number first and its
 * bytes second (below), so no allocation of a LATER slot is included and the arena is never freed too early. */
/* Request slots that may be reserved and not yet consumed.  Smaller than the ring on purpose: a pass of full rounds from
 * the request ring is ONE pass record for >= 2 rounds (pinned bulk passes, apus_replica.h), and the ring of pass records
 * (PR_CAP = 1024) also holds the staged passes' (>= 64 tickets each) -- with at most 49152 slots whose bytes have not been read there
 * are at most 768 such rounds, hence at most 384 such passes, not yet appended: a record is never overwritten under an append
 * wavefront that still needs it, however far the followers fall behind.  Round 6: 16384 -> 49152 -- the producers ran into the
 * limit at 16384 / (sequenced -> read latency of ~50 us) = 300 M entries/s. */
/* Window words (RepReq.ready_win).  Slots [s0, s0 + n) have just been published (their payloads fenced, their own words stored):
 * every aligned window of 64 they touch is told so -- how many of its slots, and whether they all have ONE length so far --
 * and the caller that brings a window to 64 writes the window's word for the sequencer (none when the lengths differ).  Several
 * producers can share a window (a block does not start on a window boundary once anybody has submitted a number of requests
 * that is not a multiple of 64); the counts live in host memory.  lens == nullptr: n slots of length len1. */
static inline void rep_win_note(apus_engine *e, uint64_t s0, uint32_t n, const apus_req_t *reqs, uint32_t len1, bool all_same = false)
{
    const uint32_t NW = RQ_CAP / WAVE;
    for (uint64_t w = s0 / WAVE; w * WAVE < s0 + n; w++) {
        const uint64_t a = std::max<uint64_t>(s0, w * WAVE), b = std::min<uint64_t>(s0 + n, (w + 1) * WAVE);
        const uint32_t len = reqs ? reqs[a - s0].len : len1;
        bool same = true;
        if (reqs && !all_same) for (uint64_t i = a + 1; i < b && same; i++) same = reqs[i - s0].len == len;
        const uint32_t ix = (uint32_t)(w % NW);
        if (b - a == WAVE) {
            /* the whole window is this caller's: nobody else has an account to settle for it */
            if (same) __atomic_store_n((uint32_t *)&e->rq->ready_win[ix], (rep_slot_tag(w * WAVE) << 16) | len, __ATOMIC_RELEASE);
            continue;
        }
        if (!same) __atomic_store_n(&e->r_win_len[ix], R_WIN_MIXED, __ATOMIC_RELAXED);
        else {
            uint32_t seen = R_WIN_UNSET;
            if (!__atomic_compare_exchange_n(&e->r_win_len[ix], &seen, len, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED) && seen != len)
                __atomic_store_n(&e->r_win_len[ix], R_WIN_MIXED, __ATOMIC_RELAXED);
        }
        const uint32_t had = __atomic_fetch_add(&e->r_win_cnt[ix], (uint32_t)(b - a), __ATOMIC_ACQ_REL);
        if (had + (uint32_t)(b - a) == WAVE) {
            const uint32_t l = __atomic_load_n(&e->r_win_len[ix], __ATOMIC_RELAXED);
            __atomic_store_n(&e->r_win_len[ix], R_WIN_UNSET, __ATOMIC_RELAXED);
            __atomic_store_n(&e->r_win_cnt[ix], 0u, __ATOMIC_RELEASE);        /* (the window's next lap starts only when these slots are consumed) */
            if (l != R_WIN_MIXED && l != R_WIN_UNSET)
                __atomic_store_n((uint32_t *)&e->rq->ready_win[ix], (rep_slot_tag(w * WAVE) << 16) | l, __ATOMIC_RELEASE);
        }
    }
}

#define R_SLOTS_INFLIGHT 49152u
static_assert(R_SLOTS_INFLIGHT <= RQ_CAP && R_SLOTS_INFLIGHT / WAVE / 2 + RS_CAP / WAVE < PR_CAP, "pass records: pinned bulk passes + staged bulk passes in flight");
static inline int rep_reserve_inline(apus_engine *e, uint32_t n, uint64_t *first)
{
    const uint64_t atail = __atomic_load_n(&e->r_arena_tail, __ATOMIC_ACQUIRE);
    const uint64_t s0 = __atomic_fetch_add(&e->r_slot_tail, (uint64_t)n, __ATOMIC_ACQ_REL);
    /* room in the ring: by what some producer last READ of the device's count first -- slots_done is a line of pinned host memory
     * the device keeps writing, and every look at it is a miss that crosses to the device's side of the fabric (0.3 of a lone
     * producer's 1.2 us per block of 256 slots went into this look: profiles/r06_feed_profile.txt); the count only grows, an old
     * value errs on the side of waiting */
    if (s0 + n - __atomic_load_n(&e->r_done_seen, __ATOMIC_RELAXED) > R_SLOTS_INFLIGHT) {
        const double t0 = mono_s();
        for (;;) {
            const uint64_t done = e->rh->slots_done;
            __atomic_store_n(&e->r_done_seen, done, __ATOMIC_RELAXED);
            if (s0 + n - done <= R_SLOTS_INFLIGHT) break;
            if (e->rh->alive == 2) return APUS_E_STATE;
            if (mono_s() - t0 > 5.0) return -1;
        }
    }
    for (uint32_t i = 0; i < n; i++) e->r_slot_aend[(s0 + i) % RQ_CAP] = atail;
    *first = s0;
    return 0;
}

/* a payload that goes into the pinned arena: slot number first, then the bytes (under the lock) */
static inline int rep_reserve_arena(apus_engine *e, uint32_t len, uint64_t *slot, void **dst)
{
    const uint64_t need = ((uint64_t)len + 15) & ~15ull;
    const double t0 = mono_s();
    pthread_spin_lock(&e->r_lock);
    const uint64_t s0 = __atomic_fetch_add(&e->r_slot_tail, 1ull, __ATOMIC_ACQ_REL);
    for (;;) {
        const uint64_t done = e->rh->slots_done;
        uint64_t pos = e->r_arena_tail;
        uint64_t phys = pos % RA_CAP;
        if (phys + need + 16 > RA_CAP) { pos += RA_CAP - phys; phys = 0; }      /* the payload does not straddle the end */
        if (phys == 0) { pos += 16; phys = 16; }                                 /* bytes -2, -1 of a payload must exist */
        const uint64_t freed = done ? e->r_slot_aend[(done - 1) % RQ_CAP] : 0;
        if (s0 + 1 - done <= R_SLOTS_INFLIGHT && pos + need - freed <= RA_CAP) {
            __atomic_store_n(&e->r_arena_tail, pos + need, __ATOMIC_RELEASE);
            e->r_slot_aend[s0 % RQ_CAP] = pos + need;
            *slot = s0;
            *dst = (void *)(e->rq->arena + phys);
            pthread_spin_unlock(&e->r_lock);
            return 0;
        }
        if (e->rh->alive == 2 || mono_s() - t0 > 5.0) { pthread_spin_unlock(&e->r_lock); return e->rh->alive == 2 ? APUS_E_STATE : -1; }
    }
}

extern "C" int apus_gpu_rep_reserve(apus_engine_t *e, uint32_t len, uint64_t *slot, void **dst)
{
    if (!e || !e->r_running || !e->r_lead || !slot || !dst || len > 65535) return APUS_E_STATE;
    if (len > R_INLINE) return rep_reserve_arena(e, len, slot, dst);
    int rc = rep_reserve_inline(e, 1, slot);
    if (rc) return rc;
    *dst = (void *)e->rq->slot[*slot % RQ_CAP].pay;
    return 0;
}

extern "C" int apus_gpu_rep_publish(apus_engine_t *e, uint64_t slot, const void *dst, uint64_t req_id, uint16_t clt_id, uint8_t type, uint16_t len)
{
    if (!e || !e->rh || !e->rq) return APUS_E_STATE;
    if (type == APUS_NOOP || type == APUS_CONFIG || type == APUS_HEAD || type > 15) return APUS_E_ARG;
    RepSlot &sl = e->rq->slot[slot % RQ_CAP];
    const bool inl = (const uint8_t *)dst == sl.pay;
    const uint64_t phys = inl ? 0 : (uint64_t)((const uint8_t *)dst - e->rq->arena);
    ReqDev d;
    d.req_id = req_id; d.pay16_type = (inl ? R_PAY_INLINE : (uint32_t)(phys / 16)) | ((uint32_t)type << 28); d.len = len; d.clt_id = clt_id;
    sl.d = d;
    /* through the BAR the payload and the descriptor are write-combined stores: they leave in front of the publish word,
     * and the publish word leaves at once */
    if (e->rq_bar) __builtin_ia32_sfence();
    __atomic_store_n((uint32_t *)&e->rq->ready_len[slot % RQ_CAP], (rep_slot_tag(slot) << 16) | len, __ATOMIC_RELEASE);
    rep_win_note(e, slot, 1, nullptr, len);
    if (e->rq_bar) __builtin_ia32_sfence();
    return 0;
}

/* n requests: runs of short payloads take their slots with ONE fetch-and-add per block; payloads are copied and
 * slots published by the caller's thread */
extern "C" int apus_gpu_rep_submit(apus_engine_t *e, const apus_req_t *reqs, uint32_t n, const uint8_t *arena, uint64_t arena_bytes)
{
    if (!e || !reqs) return APUS_E_ARG;
    if (!e->r_running || !e->r_lead) return APUS_E_STATE;
    /* everything that can be refused is refused HERE, before any slot is reserved: a block of reserved slots that is never
     * published wedges the sequencer, which takes slots strictly in order */
    bool all_same = true;                          /* (one length all the way: the windows' words need no second look at the lengths) */
    for (uint32_t g = 0; g < n; g++) {
        if (reqs[g].payload_off + reqs[g].len > arena_bytes) return APUS_E_ARG;
        if (reqs[g].type == APUS_NOOP || reqs[g].type == APUS_CONFIG || reqs[g].type == APUS_HEAD || reqs[g].type > 15) return APUS_E_ARG;
        all_same = all_same && reqs[g].len == reqs[0].len;
    }
    /* A block = the slots one fetch-and-add hands out and ONE fence covers.  Through the BAR every fence waits for the
     * write-combining buffers to drain (~0.35 us): at 64 slots per block and two fences per block a producer thread spent
     * 0.7 of its 0.83 us per block in fences (round 4: 77 M entries/s with one producer).  Round 5: 256 slots per block,
     * and the fence behind a block's publish words is the NEXT block's payload fence (the words leave with it at the
     * latest; the last block of the call has its own) -- one fence per 256 entries.  A block's publish words never leave
     * before its payloads: the fence between them stays.  APUS_REP_SUBMIT_BLOCK: measurements. */
    static const uint32_t BLK = []() { const char *v = getenv("APUS_REP_SUBMIT_BLOCK"); const int x = v ? atoi(v) : 0; return (uint32_t)(x >= 1 && x <= 1024 ? x : 256); }();
    uint32_t g = 0;
    bool words_pending = false;
    if (e->feed_prof_on < 0) { const char *v = getenv("APUS_FEED_PROF"); e->feed_prof_on = v && atoi(v) ? 1 : 0; }
    const bool prof = e->feed_prof_on == 1;
    uint64_t pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = 0;
#define FEED_T(i) do { if (prof) { const uint64_t t_ = __builtin_ia32_rdtsc(); pf[i] += t_ - tp; tp = t_; } } while (0)
    while (g < n) {
        if (reqs[g].len > R_INLINE) {
            uint64_t slot; void *dst;
            int rc = rep_reserve_arena(e, reqs[g].len, &slot, &dst);
            if (rc) { if (words_pending && e->rq_bar) __builtin_ia32_sfence(); return rc; }
            memcpy(dst, arena + reqs[g].payload_off, reqs[g].len);
            if ((rc = apus_gpu_rep_publish(e, slot, dst, reqs[g].req_id, reqs[g].clt_id, reqs[g].type, reqs[g].len))) return rc;
            words_pending = false;                         /* (publish fences) */
            g++;
            continue;
        }
        uint32_t run = 1;
        while (run < BLK && g + run < n && reqs[g + run].len <= R_INLINE) run++;
        /* (a reserve that has to wait for the ring: the block before's publish words leave first -- the sequencer takes slots in
         *  order, words that sat in the write-combining buffers meanwhile would stall every other producer) */
        if (words_pending && e->rq_bar && __atomic_load_n(&e->r_slot_tail, __ATOMIC_RELAXED) + run - e->rh->slots_done > R_SLOTS_INFLIGHT) { __builtin_ia32_sfence(); words_pending = false; }
        uint64_t s0;
        if (prof) tp = __builtin_ia32_rdtsc();
        int rc = rep_reserve_inline(e, run, &s0);
        if (rc) { if (words_pending && e->rq_bar) __builtin_ia32_sfence(); return rc; }
        FEED_T(2);
        for (uint32_t i = 0; i < run; i++) {
            const apus_req_t &q = reqs[g + i];
            RepSlot &sl = e->rq->slot[(s0 + i) % RQ_CAP];
            if (q.len) memcpy((void *)sl.pay, arena + q.payload_off, q.len);
            ReqDev d;
            d.req_id = q.req_id; d.pay16_type = R_PAY_INLINE | ((uint32_t)q.type << 28); d.len = q.len; d.clt_id = q.clt_id;
            sl.d = d;
        }
        FEED_T(3);
        if (e->rq_bar) __builtin_ia32_sfence();            /* the payloads and descriptors (and the block before's publish words) */
        FEED_T(4);
        for (uint32_t i = 0; i < run; i++)
            __atomic_store_n((uint32_t *)&e->rq->ready_len[(s0 + i) % RQ_CAP], (rep_slot_tag(s0 + i) << 16) | reqs[g + i].len, __ATOMIC_RELEASE);
        FEED_T(5);
        rep_win_note(e, s0, run, reqs + g, 0, all_same);   /* the windows these slots lie in: whoever completes one writes its word */
        FEED_T(6);
        pf[0]++; pf[1] += run;
        words_pending = true;
        g += run;
    }
    if (prof) tp = __builtin_ia32_rdtsc();
    if (words_pending && e->rq_bar) __builtin_ia32_sfence();
    FEED_T(7);
    if (prof) for (int i = 0; i < 8; i++) __atomic_fetch_add(&e->feed_prof[i], pf[i], __ATOMIC_RELAXED);
#undef FEED_T
    return 0;
}
/* APUS_FEED_PROF=1: out[0..7] = the producers' phase counters so far (apus_engine.feed_prof: TSC ticks), out[8] = TSC ticks per
 * microsecond as measured here over 20 ms; the counters are cleared */
extern "C" int apus_gpu_rep_feed_profile(apus_engine_t *e, uint64_t out[9])
{
    if (!e || !out) return APUS_E_ARG;
    for (int i = 0; i < 8; i++) out[i] = __atomic_exchange_n(&e->feed_prof[i], 0ull, __ATOMIC_RELAXED);
    const double t0 = mono_s(); const uint64_t c0 = __builtin_ia32_rdtsc();
    while (mono_s() - t0 < 0.02) { }
    out[8] = (uint64_t)((double)(__builtin_ia32_rdtsc() - c0) / ((mono_s() - t0) * 1e6));
    return 0;
}

/* device-resident input: rounds [r0, r0 + n) of the staged requests (apus_gpu_stage) go through the leader's
 * workgroups -- the "single persistent kernel per replica" throughput path */
extern "C" int apus_gpu_rep_run(apus_engine_t *e, uint64_t r0, uint64_t n_rounds)
{
    if (!e || r0 + n_rounds > e->n_rounds_staged) return APUS_E_ARG;
    return rep_push_cmd(e, R_OP_RUN, r0, n_rounds);
}
extern "C" int apus_gpu_rep_prune(apus_engine_t *e) { return rep_push_cmd(e, R_OP_PRUNE, 0, 0); }
/* a whole list of commands in one call -- cmds[3 i] = 1 (a prune tick) or 2 (rounds [cmds[3 i + 1], + cmds[3 i + 2]) of the staged
 * input) -- `repeat` times over: what a host that feeds a step's commands from a loop of its own would issue one by one (a step
 * of configs[1] is 33 commands; from Python that is ~80 us of call overhead per step, next to 130 us of work at one replica) */
extern "C" int apus_gpu_rep_cmds(apus_engine_t *e, const uint64_t *cmds, uint32_t n, uint32_t repeat)
{
    if (!e || (n && !cmds)) return APUS_E_ARG;
    for (uint32_t i = 0; i < n; i++) {
        if (cmds[3 * i] != R_OP_PRUNE && cmds[3 * i] != R_OP_RUN) return APUS_E_ARG;
        if (cmds[3 * i] == R_OP_RUN && cmds[3 * i + 1] + cmds[3 * i + 2] > e->n_rounds_staged) return APUS_E_ARG;
    }
    for (uint32_t r = 0; r < repeat; r++)
        for (uint32_t i = 0; i < n; i++) {
            const int rc = rep_push_cmd(e, (uint32_t)cmds[3 * i], cmds[3 * i + 1], cmds[3 * i + 2]);
            if (rc) return rc;
        }
    return 0;
}

/* everything submitted so far is appended everywhere, and committed + applied as far as a majority allows */
extern "C" int apus_gpu_rep_drain(apus_engine_t *e, uint32_t timeout_ms)
{
    if (!e || !e->r_running || !e->r_lead) return APUS_E_STATE;
    const double t0 = mono_s();
    for (;;) {
        if (e->rh->settled >= e->r_cmd_tail + e->r_slot_tail) return 0;
        if (e->rh->alive == 2) return APUS_E_STATE;
        if ((mono_s() - t0) * 1e3 > timeout_ms) return -1;
    }
}

/* Stop the run: the leader's workgroups finish what they were given, wait (bounded) for the ACKs that can
 * still come, write the control words back and tell every follower to park; follower workgroups hosted here
 * leave once they have consumed their doorbells.  Returns the leader's exit code (0 stop, 1 idle, 2 timeout),
 * a follower-only process the worst exit code of its followers.  Afterwards the phased / control-plane
 * calls may be used again. */
extern "C" int apus_gpu_rep_park(apus_engine_t *e)
{
    if (!e || !e->r_running) return APUS_E_STATE;
    int code = 0;
    if (e->r_lead) {
        int rc = rep_push_cmd(e, R_OP_STOP, 0, 0);
        if (rc) { __atomic_store_n((uint64_t *)&e->rq->stop, 1ull, __ATOMIC_RELEASE); if (e->rq_bar) __builtin_ia32_sfence(); }
    }
    HIPCHK(hipStreamSynchronize(e->rstream));
    if (e->r_lead) code = (int)e->rh->exit_code;
    /* Followers in OTHER processes were told to park by the leader's workgroups; their kernels write their control words back
     * (end, commit, apply; then f_runs + 1, last) a moment later.  Whatever the leader does next reads those words -- the wide
     * catch-up of the control-plane launches first of all, which would send a follower whose words still show the run's first
     * slot everything it already holds, from the leader's copy (same entries; other servers' reply bytes with them: the soak's
     * one-in-fifteen "reply bytes differ from the oracle").  Bounded: a follower whose process died never writes them. */
    if (e->r_lead) {
        const double t0 = mono_s();
        for (uint32_t m = e->r_push_mask & e->imported_mask; m; m &= m - 1) {
            const uint32_t f = (uint32_t)__builtin_ctz(m);
            if (!e->d.box[f]) continue;
            for (;;) {
                uint64_t x = 0;
                if (hipMemcpy(&x, &e->d.box[f]->f_runs, sizeof x, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); break; }
                if (x > e->r_fruns[f]) break;
                if (mono_s() - t0 > 1.0) { if (getenv("APUS_DEBUG")) fprintf(stderr, "[apus_gpu] park: follower %u has not left the run after 1 s\n", f); break; }
                struct timespec ts = {0, 200000};
                nanosleep(&ts, nullptr);
            }
        }
    }
    for (uint32_t m = e->r_follow_mask; m; m &= m - 1) {
        uint64_t x = 0;
        HIPCHK(hipMemcpy(&x, &e->d.box[__builtin_ctz(m)]->f_exit, sizeof x, hipMemcpyDeviceToHost));
        if (x > 1 && (int)(x - 1) > code) code = (int)(x - 1);
    }
    e->r_running = false;
    e->free_lb = 0;
    e->lag_possible = true;
    return code;
}

/* A follower's process while its run is resident: out[0] = entry slots applied, [1] = persisted, [2] = 1 running /
 * 2 left / 0 not started, [3] = exit code.  (What its DARE thread replays into its own application.) */
extern "C" int apus_gpu_rep_follower_progress(apus_engine_t *e, uint32_t replica, uint64_t out[4])
{
    if (!e || replica >= APUS_MAX_SERVERS || !out) return APUS_E_ARG;
    RepFHost *h = e->rfh[replica];
    if (!h) { out[0] = out[1] = out[2] = out[3] = 0; return 0; }
    out[0] = h->n_apply; out[1] = h->n_end; out[2] = h->alive; out[3] = h->exit_code;
    return 0;
}
/* A host consumer of a hosted follower's apply stream says how far it has replayed (entry slots, the count
 * apus_gpu_rep_follower_progress reports as out[0]); from the first call on the follower's kernel tells the leader
 * min(device apply, host replay) as "applied".  Callable before and during a run. */
extern "C" int apus_gpu_rep_follower_replayed(apus_engine_t *e, uint32_t replica, uint64_t slots)
{
    if (!e || replica >= APUS_MAX_SERVERS) return APUS_E_ARG;
    e->r_consumer[replica] = 1; e->r_replayed[replica] = slots;
    RepFHost *h = e->rfh[replica];
    if (h) { __atomic_store_n((uint64_t *)&h->replayed, slots, __ATOMIC_RELEASE); __atomic_store_n((uint64_t *)&h->consumer, 1ull, __ATOMIC_RELEASE); }
    return 0;
}
/* ... and when its leader is gone (no park doorbell will come): ask the follower's workgroups to leave; then apus_gpu_rep_park */
extern "C" int apus_gpu_rep_follower_stop(apus_engine_t *e, uint32_t replica)
{
    if (!e || replica >= APUS_MAX_SERVERS || !e->rfh[replica]) return APUS_E_ARG;
    __atomic_store_n((uint64_t *)&e->rfh[replica]->stop, 1ull, __ATOMIC_RELEASE);
    return 0;
}

/* tests only: a hosted follower whose workgroups are not launched -- a dead follower process that the leader still pushes to */
extern "C" int apus_gpu_rep_test_skip_follower(apus_engine_t *e, uint32_t mask) { if (!e) return APUS_E_ARG; e->r_test_skip = mask; return 0; }
/* tests only: where the leader's own term begins in its log (H_TERM_SLOT0; set by become_leader) -- the gate of rep_commit_pass */
extern "C" int apus_gpu_rep_test_term_slot0(apus_engine_t *e, uint64_t slot)
{
    if (!e || e->d.leader >= e->d.group_size || e->r_running) return APUS_E_STATE;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(&e->d.rep[e->d.leader].hdr[H_TERM_SLOT0], &slot, sizeof slot, hipMemcpyHostToDevice));
    return 0;
}

extern "C" uint64_t apus_gpu_rep_highest_rec(apus_engine_t *e) { return (e && e->rh) ? e->rh->highest_rec : 0; }
extern "C" const volatile uint64_t *apus_gpu_rep_highest_rec_ptr(apus_engine_t *e) { return (e && e->rh) ? &e->rh->highest_rec : nullptr; }
/* where the command + request rings live: 1 device memory the host stores into through the BAR, 0 pinned host memory, -1 no run yet */
extern "C" int apus_gpu_rep_req_ring_kind(apus_engine_t *e) { return (e && e->rq) ? (e->rq_bar ? 1 : 0) : -1; }
extern "C" int apus_gpu_rep_full(apus_engine_t *e) { return (e && e->rh) ? (int)e->rh->full : 0; }
/* out[8] = rounds issued, request slots taken, commands carried out, committed slots, highest_rec, rounds refused,
 *          followers dropped from the push set (mask), alive */
extern "C" int apus_gpu_rep_stats(apus_engine_t *e, uint64_t out[8])
{
    if (!e || !e->rh || !out) return APUS_E_STATE;
    out[0] = e->rh->rounds; out[1] = e->rh->slots_done; out[2] = e->rh->cmd_head; out[3] = e->rh->commit_slot;
    out[4] = e->rh->highest_rec; out[5] = e->rh->full | (e->rh->exit_code << 32); out[6] = 0; out[7] = e->rh->alive;
    if (!e->r_running && e->rl) HIPCHK(hipMemcpy(&out[6], &e->rl->drop_mask, sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

/* duration of the last run's resident launch (k_replica* on the engine's replica stream), HIP events around the launch:
 * what rocprofv3 --kernel-trace reports for the same launch.  Only after the run was parked. */
/* diagnostics: the XCD (1..8; 0 = no such workgroup) every workgroup of the last resident launch ran on, by block index */
extern "C" int apus_gpu_rep_xcc_map(apus_engine_t *e, uint8_t out[1024])
{
    if (!e || !out || !e->rl) return APUS_E_STATE;
    if (e->r_running) return APUS_E_STATE;
    HIPCHK(hipMemcpy(out, e->rl->xcc, 1024, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int apus_gpu_rep_launch_ms(apus_engine_t *e, double *ms)
{
    if (!e || !ms || e->r_running || !e->rev_valid) return APUS_E_STATE;
    float f = 0;
    HIPCHK(hipEventSynchronize(e->rev1));
    HIPCHK(hipEventElapsedTime(&f, e->rev0, e->rev1));
    *ms = (double)f;
    return 0;
}

/* diagnostics: how the serial roles of the last run spent their passes.  out[role][8]: roles 0..2 = the leader's
 * sequencer, committer, applier; 3 + 2 i, 4 + 2 i = retire / apply wavefront of the i-th hosted follower (i < 6):
 * [0] passes [1] passes that moved something [2] rounds [3] wall-clock ticks (100 MHz); 15 = the append wavefronts' and
 * 16 = the first hosted follower's work wavefronts' phase timers (APUS_REP_DBG & 256) */
extern "C" int apus_gpu_rep_role_stats(apus_engine_t *e, uint64_t out[20][8])
{
    if (!e || e->r_running || !out) return APUS_E_STATE;
    memset(out, 0, sizeof(uint64_t) * 20 * 8);
    if (e->rl) HIPCHK(hipMemcpy(out, e->rl->stat, sizeof(uint64_t) * 3 * 8, hipMemcpyDeviceToHost));
    if (e->rl) HIPCHK(hipMemcpy(out[15], e->rl->stat[3], sizeof(uint64_t) * 8, hipMemcpyDeviceToHost));
    if (e->rl) HIPCHK(hipMemcpy(out[17], e->rl->stat[4], sizeof(uint64_t) * 8, hipMemcpyDeviceToHost));
    if (e->rl) HIPCHK(hipMemcpy(out[18], e->rl->stat[5], sizeof(uint64_t) * 8, hipMemcpyDeviceToHost));       /* the sequencer's pass / prune tick by phase (APUS_REP_DBG & 512) */
    int k = 0;
    for (uint32_t m = e->r_follow_mask; m && k < 6; m &= m - 1, k++) {
        RepFollow *fs = e->rfs[__builtin_ctz(m)];
        if (fs) HIPCHK(hipMemcpy(out[3 + 2 * k], fs->stat, sizeof(uint64_t) * 2 * 8, hipMemcpyDeviceToHost));
        if (fs && k == 0) HIPCHK(hipMemcpy(out[16], fs->stat[2], sizeof(uint64_t) * 8, hipMemcpyDeviceToHost));   /* its work wavefronts' phase timers */
    }
    return 0;
}

/* round latency samples of the last run, in nanoseconds: sequenced (the round's requests were seen by the
 * leader) -> committed by a majority and applied by the leader */
extern "C" int apus_gpu_rep_latency(apus_engine_t *e, uint32_t *out_ns, uint32_t cap, uint32_t *n_out)
{
    if (!e || !e->rl || e->r_running) return APUS_E_STATE;
    uint32_t n = 0;
    HIPCHK(hipMemcpy(&n, &e->rl->lat_n, sizeof n, hipMemcpyDeviceToHost));
    if (n > cap) n = cap;
    if (n) HIPCHK(hipMemcpy(out_ns, e->rl->lat_ticks, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int khz = 100000;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->cfg.device);
    if (khz <= 0) khz = 100000;
    for (uint32_t i = 0; i < n; i++) out_ns[i] = (uint32_t)((uint64_t)out_ns[i] * 1000000ull / (uint64_t)khz);
    if (n_out) *n_out = n;
    return 0;
}

/* ... and from "the round's bytes are in every pushed ring" (the end of the leader's append, SURVEY 8d's
 * definition of the consensus-round latency) to committed and applied */
extern "C" int apus_gpu_rep_latency_appended(apus_engine_t *e, uint32_t *out_ns, uint32_t cap, uint32_t *n_out)
{
    if (!e || !e->rl || e->r_running) return APUS_E_STATE;
    uint32_t n = 0;
    HIPCHK(hipMemcpy(&n, &e->rl->lat_n, sizeof n, hipMemcpyDeviceToHost));
    if (n > cap) n = cap;
    if (n) HIPCHK(hipMemcpy(out_ns, e->rl->lat_app, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int khz = 100000;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->cfg.device);
    if (khz <= 0) khz = 100000;
    for (uint32_t i = 0; i < n; i++) out_ns[i] = (uint32_t)((uint64_t)out_ns[i] * 1000000ull / (uint64_t)khz);
    if (n_out) *n_out = n;
    return 0;
}

/* submit one round of n <= 64 requests and spin on highest_rec, `iters` times (what proxy.c:160 does) */
extern "C" int apus_gpu_rep_roundtrip(apus_engine_t *e, const apus_req_t *reqs, uint32_t n, const uint8_t *arena, uint64_t arena_bytes,
                                      uint32_t iters, uint32_t *out_ns)
{
    if (!e || !e->r_running || !e->r_lead || n == 0 || n > APUS_MAX_ROUND) return APUS_E_STATE;
    for (uint32_t i = 0; i < iters; i++) {
        const uint64_t target = e->rh->highest_rec + n;
        const double t0 = mono_s();
        int rc = apus_gpu_rep_submit(e, reqs, n, arena, arena_bytes);
        if (rc) return rc;
        while (e->rh->highest_rec < target)
            if (e->rh->alive == 2 || mono_s() - t0 > 2.0) return -1;
        out_ns[i] = (uint32_t)((mono_s() - t0) * 1e9);
    }
    return 0;
}

/* ---- link calibration (the reference's rc_get_loggp_params, dare_ibv_rc.c:3323-3739, for peer stores) ---- */
/* role 0: starts `iters` 8-byte round trips with `peer` and returns the samples (ns); role 1: answers them.  Both
 * processes call at about the same time (the control plane pairs them); `base` makes the words of this exchange
 * distinct from an earlier one's.  The replicas must be idle (no run resident). */
extern "C" int apus_gpu_calib_pingpong(apus_engine_t *e, uint32_t me, uint32_t peer, uint32_t role, uint32_t iters, uint64_t base,
                                       uint32_t *out_ns, uint32_t timeout_ms)
{
    if (!e || me >= e->cfg.group_size || peer >= e->cfg.group_size || me == peer || !e->d.box[me] || !e->d.box[peer] || iters == 0 || iters > 65535)
        return APUS_E_ARG;
    if (e->r_running || e->p_running || e->batching) return APUS_E_STATE;
    uint32_t *d_t = nullptr;
    HIPCHK(hipMalloc((void **)&d_t, sizeof(uint32_t) * (iters + 1)));
    HIPCHK(hipMemsetAsync(d_t, 0, sizeof(uint32_t) * (iters + 1), e->stream));
    hipLaunchKernelGGL(k_calib_pingpong, dim3(1), dim3(64), 0, e->stream, e->d, me, peer, role, iters, base, d_t,
                       (uint64_t)timeout_ms * 2000ull);
    hipError_t er = hipStreamSynchronize(e->stream);
    std::vector<uint32_t> h(iters + 1);
    if (er == hipSuccess) er = hipMemcpy(h.data(), d_t, sizeof(uint32_t) * (iters + 1), hipMemcpyDeviceToHost);
    hipFree(d_t);
    if (er != hipSuccess) return APUS_E_HIP;
    if (h[0] != iters) return -1;                         /* the peer did not answer in time */
    if (role == 0 && out_ns) {
        int khz = 100000;
        hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->cfg.device);
        if (khz <= 0) khz = 100000;
        for (uint32_t i = 0; i < iters; i++) out_ns[i] = (uint32_t)((uint64_t)h[i + 1] * 1000000ull / (uint64_t)khz);
    }
    return 0;
}

/* First contact between two devices (apus_selftest.h): roles bit 0 = this process pushes `rounds` rounds of 8 KiB into
 * replica `owner`'s ring + doorbells into its mailbox, bit 1 = this process's RESIDENT kernel checks them (both: one launch on
 * one device).  The two processes call it at the same time.  out: [0] rounds checked, [1] 16-byte units that differed,
 * [2] first round that differed + 1 (0: none), [3] waits that timed out.  What the test touched (the first regions x 8 KiB of
 * the owner's ring, both mailboxes' doorbell lines) is cleared afterwards: run it before the group starts. */
extern "C" int apus_gpu_selftest(apus_engine_t *e, uint32_t pusher, uint32_t owner, uint32_t roles, uint64_t rounds, uint32_t regions,
                                 uint32_t timeout_ms, uint64_t out[4])
{
    const uint32_t mode = roles & ~3u;                     /* (experiments: 16 = a release behind the pusher's stores, 32 = an invalidate in front of the checker's loads) */
    roles &= 3u;
    if (!e || !out || pusher >= e->cfg.group_size || owner >= e->cfg.group_size || pusher == owner || !roles || !rounds) return APUS_E_ARG;
    if (!e->d.rep[owner].ring || !e->d.box[owner] || !e->d.box[pusher]) return APUS_E_ARG;
    if (regions < 64 || regions > RB_CAP || (uint64_t)regions * ST_ROUND_BYTES > e->d.log_len) return APUS_E_ARG;
    if (e->r_running || e->p_running || e->batching) return APUS_E_STATE;
    unsigned long long *d_res = nullptr;
    HIPCHK(hipMalloc((void **)&d_res, 16 * sizeof(unsigned long long)));
    const unsigned long long init[16] = { 0, 0, ~0ull, 0 };
    HIPCHK(hipMemcpyAsync(d_res, init, sizeof init, hipMemcpyHostToDevice, e->stream));
    const uint32_t wgs = 16;                              /* 64 wavefronts per role */
    hipLaunchKernelGGL(k_selftest, dim3(roles == 3 ? 2 * wgs : wgs), dim3(256), 0, e->stream, e->d, pusher, owner, roles | mode, rounds, regions,
                       0xA905000500000000ull ^ rounds, (uint64_t)timeout_ms * 1500ull, d_res);
    hipError_t er = hipStreamSynchronize(e->stream);
    unsigned long long h[16] = { 0 };
    if (er == hipSuccess) er = hipMemcpy(h, d_res, sizeof h, hipMemcpyDeviceToHost);
    hipFree(d_res);
    if (er != hipSuccess) return APUS_E_HIP;
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2] == ~0ull ? 0 : h[2]; out[3] = h[3];
    e->st_atomic_misses += h[10];
    if (h[1] && getenv("APUS_DEBUG"))
        fprintf(stderr, "[apus_gpu] selftest %u -> %u: first difference seen in round %llu unit %llu: there %016llx %016llx, pushed %016llx %016llx\n", pusher, owner,
                h[4] - 1, h[5], h[6], h[7], h[8], h[9]);
    /* leave no granule behind that a run's doorbell numbering could meet: whoever OWNS a buffer clears it */
    const bool own_owner = ((e->local_mask >> owner) & 1u) && !((e->imported_mask >> owner) & 1u);
    const bool own_pusher = ((e->local_mask >> pusher) & 1u) && !((e->imported_mask >> pusher) & 1u);
    if ((roles & 2u) && own_owner) {
        HIPCHK(hipMemsetAsync(e->d.rep[owner].ring, 0, (size_t)regions * ST_ROUND_BYTES, e->stream));
        HIPCHK(hipMemsetAsync(e->d.box[owner]->rnd, 0, sizeof e->d.box[owner]->rnd, e->stream));
    }
    if ((roles & 1u) && own_pusher) HIPCHK(hipMemsetAsync(e->d.box[pusher]->rnd, 0, sizeof e->d.box[pusher]->rnd, e->stream));
    if ((roles & 1u) && own_pusher) HIPCHK(hipMemsetAsync(&e->d.box[pusher]->persisted_fast_by[15], 0, sizeof(uint64_t), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
/* the pushing side's count, over this engine's self-tests so far, of regions it found freed before the checker's system-scope
 * atomic max (issued and drained in front of the freeing word) had reached its mailbox: 0 is what REP_FAST_ACK rests on */
extern "C" int apus_gpu_selftest_atomic_misses(apus_engine_t *e, uint64_t *misses)
{
    if (!e || !misses) return APUS_E_ARG;
    *misses = e->st_atomic_misses;
    return 0;
}
/* The write-through store ceiling of the data path's pattern (apus_selftest.h: k_calib_store_multi): 8 KiB chunks written at
 * the same offset into every hosted ring of `mask`, the whole ring, `passes` times, `wgs` workgroups of four wavefronts;
 * *gbps = bytes written / the launch's duration (HIP events).  Destroys the rings' contents. */
extern "C" int apus_gpu_calib_store_multi(apus_engine_t *e, uint32_t mask, uint32_t passes, uint32_t wgs, float *gbps)
{
    if (!e || !gbps || !passes || !wgs || wgs > 4096) return APUS_E_ARG;
    if (e->r_running || e->p_running || e->batching) return APUS_E_STATE;
    const uint32_t skew = (passes >> 16) & 0x70u;          /* (passes bits 16..22: the chunks' offset within a 128-byte line, a multiple of 16) */
    passes &= 0xFFFFu;
    CalibRings R; R.n = 0;
    for (uint32_t i = 0; i < e->cfg.group_size && i < APUS_DEV_MAX_SERVERS; i++)
        if (((mask >> i) & 1u) && e->d.rep[i].ring && !((e->imported_mask >> i) & 1u)) R.r[R.n++] = e->d.rep[i].ring;
    if (!R.n) return APUS_E_ARG;
    const uint64_t bytes = e->d.log_len & ~(uint64_t)(ST_ROUND_BYTES - 1);
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_calib_store_multi, dim3(wgs), dim3(256), 0, e->stream, R, bytes, 1u, 7u, skew);      /* (warm: page tables, clocks) */
    hipEventRecord(a, e->stream);
    hipLaunchKernelGGL(k_calib_store_multi, dim3(wgs), dim3(256), 0, e->stream, R, bytes, passes, 11u, skew);
    hipEventRecord(b, e->stream);
    int rc = hipEventSynchronize(b) == hipSuccess ? 0 : APUS_E_HIP;
    float ms = 0.f;
    if (!rc) hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    *gbps = ms > 0.f ? (float)((double)bytes * R.n * passes / (ms * 1e-3) / 1e9) : 0.f;
    return rc;
}
/* how this engine's log rings are allocated: 0 ordinary device memory, 1 fine-grained, 2 uncached (APUS_RING_ALLOC at create) */
extern "C" int apus_gpu_ring_alloc_kind(apus_engine_t *e) { return e ? e->ring_alloc : APUS_E_ARG; }

/* write-through 16-byte stores of `bytes` into replica `peer`'s ring, `iters` times, HIP events around each pass:
 * out_gbps[i] = GB/s of pass i.  The ring's contents are destroyed: calibrate before the group starts (or reset). */
extern "C" int apus_gpu_calib_store_bw(apus_engine_t *e, uint32_t peer, uint64_t bytes, uint32_t iters, float *out_gbps)
{
    if (!e || peer >= e->cfg.group_size || !e->d.rep[peer].ring || bytes < 16 || bytes > e->d.log_len || !out_gbps || iters == 0) return APUS_E_ARG;
    if (e->r_running || e->p_running || e->batching) return APUS_E_STATE;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, std::max<uint64_t>(1, bytes / 16 / 256));
    int rc = 0;
    for (uint32_t i = 0; i < iters && !rc; i++) {
        hipEventRecord(a, e->stream);
        hipLaunchKernelGGL(k_calib_store, dim3(grid), dim3(256), 0, e->stream, e->d.rep[peer].ring, bytes, i + 1);
        hipEventRecord(b, e->stream);
        if (hipEventSynchronize(b) != hipSuccess) { rc = APUS_E_HIP; break; }
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        out_gbps[i] = ms > 0.f ? (float)((double)bytes / (ms * 1e-3) / 1e9) : 0.f;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return rc;
}

/* Host-fed throughput of the multi-producer ring: n_threads application threads (what memcached's worker
 * threads are to leader_handle_submit_req, src/proxy/proxy.c:108-161) each submit the same block of requests
 * over and over for `seconds`; thread 0 also plays the prune timer (one tick per prune_bytes of log).  Then
 * everything is drained.  out[0] = requests submitted, out[1] = nanoseconds from the first submit to the
 * drain's end. */
/* Where the device hangs: the NUMA node of its PCIe root (sysfs), -1 when the platform does not say.  On a two-socket host a
 * producer on the OTHER socket writes the request ring through the inter-socket link as well: round 6 measured 130 against 166 M
 * entries/s for one producer, 280 against 346 for four (profiles/r06_numa.txt). */
extern "C" int apus_gpu_numa_node(int device)
{
    char bus[64] = {0}, path[160];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}
/* the CPUs of a NUMA node this process may run on, in the order of the node's cpulist (physical cores first, their sibling
 * threads behind them on the hosts seen here); returns how many */
static int node_cpus(int node, int *out, int cap)
{
    char path[96], buf[1024] = {0};
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!ok) return 0;
    cpu_set_t all;
    if (sched_getaffinity(0, sizeof all, &all)) return 0;
    int n = 0;
    for (char *p = buf; *p && *p != '\n';) {
        char *q; long a = strtol(p, &q, 10), b = a;
        if (q == p) break;
        if (*q == '-') { p = q + 1; b = strtol(p, &q, 10); }
        for (long c = a; c <= b && n < cap; c++) if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET((int)c, &all)) out[n++] = (int)c;
        p = *q == ',' ? q + 1 : q;
    }
    return n;
}
/* Binds the calling thread (who = 0) or the whole process as it stands (who = 1: every thread that exists now, and what they
 * start later) to the CPUs of the device's NUMA node.  0: done; 1: nothing to do (one node, or the platform does not say). */
extern "C" int apus_gpu_bind_near(apus_engine_t *e, int who)
{
    if (!e) return APUS_E_ARG;
    const int node = apus_gpu_numa_node(e->cfg.device);
    if (node < 0) return 1;
    static int cpus[CPU_SETSIZE];
    const int n = node_cpus(node, cpus, CPU_SETSIZE);
    cpu_set_t all;
    if (n <= 0 || sched_getaffinity(0, sizeof all, &all) || n >= CPU_COUNT(&all)) return 1;
    cpu_set_t set; CPU_ZERO(&set);
    for (int i = 0; i < n; i++) CPU_SET(cpus[i], &set);
    if (!who) return pthread_setaffinity_np(pthread_self(), sizeof set, &set) ? APUS_E_STATE : 0;
    /* every thread of the process: /proc/self/task */
    int rc = 0;
    if (DIR *d = opendir("/proc/self/task")) {
        while (struct dirent *de = readdir(d)) {
            const int tid = atoi(de->d_name);
            if (tid > 0 && sched_setaffinity(tid, sizeof set, &set)) rc = APUS_E_STATE;
        }
        closedir(d);
    } else rc = APUS_E_STATE;
    return rc;
}

struct RepFeedArg { apus_engine *e; const apus_req_t *reqs; uint32_t n; const uint8_t *arena; uint64_t arena_bytes;
                    double t_end; uint64_t prune_every; int tid; uint64_t done; int rc; uint64_t *total; };
static void *rep_feed_thread(void *p)
{
    RepFeedArg *a = (RepFeedArg *)p;
    {   /* Where the producers run.  Default: producer i on a CPU of its own on the DEVICE'S NUMA node (the node's third CPU onwards:
         * the first ones take the host's interrupts) -- on the two-socket hosts of the pool a producer on the other socket writes the
         * ring through the inter-socket link as well, and two producers on sibling threads share a core's write-combining buffers:
         * 131-162 / 201-231 / 284-291 / 391-402 M entries/s wherever the scheduler put 1 / 2 / 4 / 8 of them, 166 / 280 / 367 / 425
         * like this (profiles/r06_numa.txt).  APUS_FEED_PIN=0: the scheduler's choice; =<stride>: the (i x stride)-th CPU the
         * process may run on, whatever its node. */
        const char *pin = getenv("APUS_FEED_PIN");
        const int stride = pin && strcmp(pin, "node") ? atoi(pin) : -1;
        static int cpus[CPU_SETSIZE];
        cpu_set_t all;
        if (stride < 0) {
            const int node = apus_gpu_numa_node(a->e->cfg.device);
            const int n = node >= 0 ? node_cpus(node, cpus, CPU_SETSIZE) : 0;
            if (n > 0) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[(a->tid + (n > 8 ? 2 : 0)) % n], &one); pthread_setaffinity_np(pthread_self(), sizeof one, &one); }
        } else if (stride > 0 && sched_getaffinity(0, sizeof all, &all) == 0) {
            int want = a->tid * stride, seen = 0, n_all = CPU_COUNT(&all);
            if (n_all > 0) want %= n_all;
            for (int c = 0; c < CPU_SETSIZE; c++) {
                if (!CPU_ISSET(c, &all)) continue;
                if (seen++ == want) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(c, &one); pthread_setaffinity_np(pthread_self(), sizeof one, &one); break; }
            }
        }
    }
    while (mono_s() < a->t_end) {
        int rc = apus_gpu_rep_submit(a->e, a->reqs, a->n, a->arena, a->arena_bytes);
        if (rc) { a->rc = rc; break; }
        a->done += a->n;
        /* the prune tick belongs to the producers' TOTAL: whoever carries it over a multiple of prune_every issues it (round 5 let
         * producer 0 count its own share -- with the producers on CPUs of their own they no longer run at one rate, producer 0
         * fell behind once and the log ran full) */
        if (a->prune_every) {
            const uint64_t before = __atomic_fetch_add(a->total, (uint64_t)a->n, __ATOMIC_RELAXED);
            if (before / a->prune_every != (before + a->n) / a->prune_every) apus_gpu_rep_prune(a->e);
        }
    }
    return nullptr;
}
extern "C" int apus_gpu_rep_feed(apus_engine_t *e, const apus_req_t *reqs, uint32_t n, const uint8_t *arena, uint64_t arena_bytes,
                                 uint32_t n_threads, double seconds, uint64_t prune_every_reqs, uint64_t out[2])
{
    if (!e || !e->r_running || !e->r_lead || !reqs || !n || n_threads == 0 || n_threads > 64 || !out) return APUS_E_ARG;
    RepFeedArg args[64];
    pthread_t th[64];
    uint64_t shared_total = 0;
    const double t0 = mono_s();
    for (uint32_t i = 0; i < n_threads; i++) {
        args[i] = RepFeedArg{ e, reqs, n, arena, arena_bytes, t0 + seconds, prune_every_reqs, (int)i, 0, 0, &shared_total };
        if (pthread_create(&th[i], nullptr, rep_feed_thread, &args[i])) return APUS_E_STATE;
    }
    uint64_t total = 0; int rc = 0;
    for (uint32_t i = 0; i < n_threads; i++) { pthread_join(th[i], nullptr); total += args[i].done; if (args[i].rc) rc = args[i].rc; }
    if (!rc) rc = apus_gpu_rep_drain(e, 60000);
    out[0] = total; out[1] = (uint64_t)((mono_s() - t0) * 1e9);
    return rc;
}

#ifdef APUS_TRACE
/* diagnostics build only: copy out the in-kernel stamps ([16 kernels][64] u64) */
extern "C" int apus_gpu_trace(apus_engine_t *e, uint64_t *out, uint32_t n)
{
    if (!e || !out || n > 16 * 64) return APUS_E_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, e->d.trace, n * 8, hipMemcpyDeviceToHost));
    return 0;
}
#endif
