/*
 * apus_gpu.h -- C ABI of the MI355X consensus engine (libapus_gpu.so).
 *
 * Drop-in boundary for ONE path of hku-systems/apus: the dare_server consensus
 * loop (log append -> replicate -> ACK aggregation -> commit advance -> apply).
 * Plain pointers and sizes only; no torch / HIP types in any signature (a
 * hipStream_t travels as void*).  Citations are relative to /root/reference/.
 *
 * Three layers, outermost first:
 *
 *   B-outer  proxy_init / proxy_on_read / proxy_on_accept / proxy_on_close
 *            (src/include/rsm-interface.h:12-15) and
 *   B-inner  dare_server_init / is_leader / get_node_id
 *            (src/include/dare/dare_server.h:197-203)
 *            are implemented in apus_amd/host/ (plain C) on top of this header;
 *            INTEGRATION.md shows the two-line change that makes the
 *            reference's own interposer (src/spec_hooks.cpp) link against them.
 *
 *   B-transport  the hot subset of the dare_ib_* facade
 *            (src/include/dare/dare_ibv.h:140-201) is exported below under the
 *            reference's own names, so that dare_server.c's polling() loop can
 *            call them unchanged.
 *
 *   engine   apus_gpu_*: what those wrappers are made of, also used directly by
 *            the trace harness, the tests and bench.py.
 *
 * Error convention (same as the reference's rc_* functions,
 * src/dare/dare_ibv_rc.c:27-29): 0 = success, 1 = error, -1 = retry later.
 * apus_gpu_* additionally return negative APUS_E_* codes.
 */
#ifndef APUS_GPU_H
#define APUS_GPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APUS_MAX_SERVERS  13                    /* MAX_SERVER_COUNT, src/include/dare/dare.h:26      */
#define APUS_ENTRY_HDR    64                    /* sizeof(dare_log_entry_t), dare_log.h:33-47        */
#define APUS_LOG_SIZE     (16384ull * 4096ull)  /* LOG_SIZE, src/include/dare/dare_log.h:76          */
#define APUS_MAX_ROUND    64                    /* entries per device round = one wavefront          */

/* entry types: dare_log.h:22-25 and src/include/proxy/proxy.h:10-12 */
#define APUS_NOOP 0
#define APUS_CSM 1
#define APUS_CONFIG 2
#define APUS_HEAD 3
#define APUS_CONNECT 4
#define APUS_SEND 5
#define APUS_CLOSE 6

/* negative engine error codes */
#define APUS_E_ARG      (-2)
#define APUS_E_HIP      (-3)
#define APUS_E_NOMEM    (-4)
#define APUS_E_STATE    (-5)   /* no leader / not staged / bad replica */
#define APUS_E_FULL     (-6)   /* the batch does not fit into the free part of the log: nothing was appended (status bit APUS_ST_LOG_FULL) */
#define APUS_E_NOANSWER (-8)   /* apus_gpu_join: too few members would answer the joiner (their OWN configuration does not show it, or still shows the slot's former holder): the reference's joiner retries for ever */
#define APUS_E_DEVICE   (-6)   /* a device-side status bit is set: see apus_gpu_status */

/* device status bits (sticky until apus_gpu_clear_status) */
#define APUS_ST_SECOND_WRAP  (1u << 0)  /* one run_rounds batch wrapped the ring twice               */
#define APUS_ST_LOG_FULL     (1u << 1)  /* an append met end == head (dare_log.h:168) or ran over it  */
#define APUS_ST_TERM_FENCE   (1u << 2)  /* a follower saw an entry batch from a stale term            */
#define APUS_ST_DIR_OVERRUN  (1u << 3)  /* more live entries than directory slots                     */
#define APUS_ST_SPIN_TIMEOUT (1u << 4)  /* persistent kernel gave up waiting (bounded spin)           */
#define APUS_ST_JOIN_WALK    (1u << 5)  /* a joiner's first persist pass did not end (the reference spins there forever): taken as done */

/* one admitted client request = what a tailq_entry_t carries
 * (src/include/dare/message.h:11-18), produced by leader_handle_submit_req
 * (src/proxy/proxy.c:108-161).  24 bytes, little endian. */
typedef struct {
    uint64_t req_id;
    uint64_t payload_off;   /* byte offset into the payload arena, 16-byte aligned */
    uint16_t clt_id;        /* connection_id */
    uint16_t len;           /* payload bytes (cmd.len) */
    uint8_t  type;          /* APUS_CONNECT / APUS_SEND / APUS_CLOSE */
    uint8_t  pad[3];
} apus_req_t;

/* one record of the apply stream: the upcall apply_committed_entries would make
 * (src/dare/dare_server.c:1941-1955).  32 bytes. */
typedef struct {
    uint64_t slot;          /* position of the entry in the total order of the log */
    uint64_t off;           /* byte offset of the entry in the ring */
    uint64_t idx;           /* entry->idx */
    uint32_t len;           /* entry->data.cmd.len */
    uint16_t clt_id;
    uint8_t  type;
    uint8_t  kind;          /* 0 = none (control entry), 1 = proxy_update_state, 2 = proxy_do_action */
} apus_apply_t;

typedef struct {
    uint32_t group_size;                 /* N = cid.size[0]                                       */
    uint32_t n_local;                    /* logical replicas hosted by this process / device      */
    uint8_t  local_ids[APUS_MAX_SERVERS];/* their indices in the group                            */
    uint8_t  pad[3];
    uint64_t log_len;                    /* ring bytes; 0 = APUS_LOG_SIZE                         */
    int32_t  device;                     /* HIP device ordinal                                    */
    uint32_t flags;                      /* APUS_F_* bits, 0 by default                           */
    void    *stream;                     /* hipStream_t to launch on; NULL = engine-owned stream  */
} apus_cfg_t;

/* apus_cfg_t.flags */
/* (bit 1u, APUS_F_NO_FUSED_ACKS of rounds 1-5 -- the call-per-pass plane with every ACK forced through reply byte, ACK word and scan --
 * is retired: the replica kernels (apus_gpu_rep_*) ARE the data plane whose followers acknowledge for themselves.) */
/* The term fence in front of every launch that stores into followers (k_fence_check): a follower whose
 * control block shows a SID of a NEWER term than the leader's -- it voted for, or heard from, a leader
 * driven by another engine -- fences this leader off: APUS_ST_TERM_FENCE is raised and the launches behind
 * the check store nothing anywhere (the receiver-side QP reset of the reference, rc_revoke_log_access
 * src/dare/dare_ibv_rc.c:2156-2243, moved in front of the writer).  Always on in peer-mapped groups
 * (replicas imported with apus_gpu_import_replica); this flag turns it on for a single process.
 *  - The fence ends when this engine wins a term again that none of the followers it pushes to is ahead of
 *    (apus_gpu_elect / apus_gpu_become_leader: rc_restore_log_access, dare_ibv_rc.c:2245-2290).
 *  - The check runs IN FRONT of a launch, not inside it: a follower that adopts a newer SID after the check (another
 *    process's k_elect, apus_gpu_adopt_sid) is still written to by the launch behind it, and the kernels of JOIN, log
 *    adjustment, forced pruning and the persistent kernel carry no check of their own.  The control plane must
 *    therefore order an election behind the old leader's stream (apus_amd/peers.py: barrier + sync in elect() and
 *    join()); without that the fence is advisory.  The replica kernels (apus_gpu_rep_*) fence on the RECEIVER's side
 *    instead, like the reference: a follower's own workgroups do not acknowledge a round of a term older than their
 *    SID's -- no reply byte, no ACK: nothing of it can commit. */
#define APUS_F_TERM_FENCE 2u
/* (flags bit 4u is NOT part of this ABI.  It is the parity harness's diagnostic APUS_F_REF_QUIRKS, defined with its kernel in
 * apus_amd/csrc/apus_quirks.h: on the call-per-pass path it reproduces, bit for bit, the reference's state at a commit pointer
 * parked on a wrap position without a majority -- commit = 0 and, at a case-1 wrap, the entry at offset 0 APPLIED although it
 * is not committed (src/dare/dare_ibv_rc.c:1725-1758 + src/include/dare/dare_log.h:327-330): a client released by an entry only
 * the leader holds.  DECIDED, round 5: that is a safety violation of the reference, the deviation from it is permanent on every
 * data plane that carries traffic (the replica kernels, batches, the persistent kernel never apply an uncommitted entry and
 * return APUS_E_STATE under the bit); DESIGN.md section 6, Deviation 1, states the bound and the tests that hold it.) */

typedef struct apus_engine apus_engine_t;

/* ---- lifetime ------------------------------------------------------------- */
int  apus_gpu_create(const apus_cfg_t *cfg, apus_engine_t **out);
void apus_gpu_destroy(apus_engine_t *e);
int  apus_gpu_reset(apus_engine_t *e);     /* back to log_new() state (dare_log.h:120) on every replica; async */
int  apus_gpu_sync(apus_engine_t *e);      /* wait for everything queued on the engine stream */

/* ---- request admission (replaces the malloc'd TAILQ, message.h:20-22) ------ */
/* Copies n admitted requests, their payload arena and the round partition
 * (round_n[r] = number of requests the leader's r-th polling() pass drains,
 * each <= APUS_MAX_ROUND) into HBM. */
int  apus_gpu_stage(apus_engine_t *e, const apus_req_t *reqs, uint64_t n,
                    const uint8_t *arena, uint64_t arena_bytes,
                    const uint32_t *round_n, uint64_t n_rounds);

/* ---- one replica per GPU / process: peer-mapped logs ------------------------ */
/* The reference's followers are passive during replication: the leader's NIC writes their log,
 * `end` and `commit` in place (one-sided RDMA, src/dare/dare_ibv_rc.c:1465-1643) after the
 * RC_SYN / RC_SYNACK handshake has told it their MR address and rkey (dare_ibv_ud.c:1098-1380).
 * Here a process exports the HBM buffers of the replica it hosts as HIP IPC handles and imports
 * its peers': the leader's kernels then store into the peers' rings, directories, control
 * blocks and apply streams directly -- over xGMI when the peers sit on other GPUs -- with the
 * same fused push / ACK / commit path that serves logical replicas on one device, and a follower
 * process only reads its own memory.  Handles travel between processes by any means (the tests
 * and bench.py use torch.distributed all_gather). */
#define APUS_IPC_BUFFERS 8u              /* ring, control block, directory offsets / lengths, ACK words, apply stream,
                                          * mailbox + ACK byte maps of the replica kernels (apus_gpu_rep_*) */
#define APUS_FENCE_PAIRS 4u              /* the (log ring, mailbox) pairs a replica moves through at its fences (apus_gpu_fence_replica) */
typedef struct {
    uint8_t  handle[APUS_IPC_BUFFERS][64];   /* hipIpcMemHandle_t each; [0] and [6] = the pair the replica lives in now */
    uint64_t log_len;
    uint32_t dir_cap;
    uint32_t replica;                         /* group index of the replica */
    int32_t  device;                          /* HIP device ordinal in the exporting process */
    uint32_t fences;                          /* how often the replica's ring and mailbox have moved (apus_gpu_fence_replica):
                                               * it lives in pair[fences % APUS_FENCE_PAIRS] */
    uint8_t  pair[APUS_FENCE_PAIRS][2][64];   /* every (ring, mailbox) pair: a peer maps them ALL when it maps the replica */
} apus_ipc_replica_t;
/* replica must be hosted (allocated) by this engine */
int  apus_gpu_export_replica(apus_engine_t *e, uint32_t replica, apus_ipc_replica_t *out);
/* maps a peer process's replica; from then on this engine can lead a group that contains it.
 * The memory stays owned by the exporter (never reset or freed here). */
int  apus_gpu_import_replica(apus_engine_t *e, const apus_ipc_replica_t *in);
/* one peer's mapping alone: the members drop a dead process's replica before they map the machine that takes its slot */
int  apus_gpu_unmap_replica(apus_engine_t *e, uint32_t replica);
/* Orderly shutdown of a peer-mapped group, first half: close every imported mapping, free nothing.  (Every process
 * unmaps, a barrier, then every process destroys: an owner that frees a buffer a peer still has open cannot export
 * the memory it allocates next.)  APUS_E_STATE while a resident kernel or a batch is open. */
int  apus_gpu_unmap_peers(apus_engine_t *e);
/* The receiver's fence against a deposed leader (rc_revoke_log_access, src/dare/dare_ibv_rc.c:2156-2243: the voters reset
 * the old leader's QPs, its WRITEs bounce).  A mapped buffer cannot be taken back, but it can be left: the hosted replica
 * moves the two buffers peers store into during a run -- log ring and mailbox -- to the next of its APUS_FENCE_PAIRS pairs
 * (device copies); whoever still runs on the old pointers stores into memory nobody reads.  The pairs are allocated when the
 * replica is first exported and a peer maps them ALL with apus_gpu_import_replica: nothing is allocated, exported, opened or
 * closed at election time (the first cut of round 6 did, and the soak caught the runtime handing one address range to two
 * mappings after a few close/open cycles between runs).  out (may be NULL) = the handles, fences + 1.  A server calls it
 * when it adopts a newer term (apus_amd/peers.py: elect; apus_amd/host/apus_proxy.c: group_failover -- at EVERY election),
 * the members of the new term then call apus_gpu_remap_fenced with its handles: the pointers their next run is given are
 * switched to pair[fences % APUS_FENCE_PAIRS] (a no-op for a replica whose `fences` they already know).
 * BOUND: a pair is lived in again APUS_FENCE_PAIRS fences after it was left: a leader deposed that many terms ago whose
 * kernel is STILL storing is outside the failure model (its process steps down at the first newer announcement it sees and
 * parks its kernel; the reference's QP reset has no such bound).
 * APUS_E_STATE while a resident kernel or a batch is open, and for an engine that has captured graphs. */
int  apus_gpu_fence_replica(apus_engine_t *e, uint32_t replica, apus_ipc_replica_t *out);
int  apus_gpu_remap_fenced(apus_engine_t *e, const apus_ipc_replica_t *in);
/* tests: n bytes at off of the ring `replica` left `back` fences ago (1 = the last): where a deposed leader's stores went */
int  apus_gpu_read_retired_ring(apus_engine_t *e, uint32_t replica, uint32_t back, uint64_t off, uint64_t n, void *dst);

/* Host-side profile of the request ring's producers (APUS_FEED_PROF=1 in the environment, read at the first submit): TSC ticks
 * summed over the producers, by phase of apus_gpu_rep_submit -- out[0] blocks, [1] slots, [2] reserve (fetch-and-add + waiting for
 * room), [3] payload + descriptor stores (write-combined, through the BAR), [4] the fence behind them, [5] the slots' publish words,
 * [6] the windows' accounts and words, [7] the call's last fence; out[8] = TSC ticks per microsecond.  Clears the counters.
 * (profiles/r06_feed_profile.txt: what `perf stat` would have been asked about, from inside.) */
int  apus_gpu_rep_feed_profile(apus_engine_t *e, uint64_t out[9]);

/* ---- where the device hangs (two-socket hosts) ---------------------------- */
/* The NUMA node of the device's PCIe root (-1: the platform does not say).  A thread on the other socket reaches the request
 * ring and the device's answers through the inter-socket link as well: one producer 130 against 166 M entries/s, four 280
 * against 330-370 (round 6, profiles/r06_numa.txt).  apus_gpu_bind_near binds the calling thread (who = 0) or every thread of
 * the process as it stands (who = 1) to that node's CPUs: 0 done, 1 nothing to do (one node / unknown).  A lone request's
 * round trip does not change with it (17.5 us either way); nothing binds an application behind its back -- the producers of
 * apus_gpu_rep_feed (bench.py's host-fed leg) are placed on the device's node, one CPU each (APUS_FEED_PIN). */
int  apus_gpu_numa_node(int device);
int  apus_gpu_bind_near(apus_engine_t *e, int who);

/* ---- control plane (host-driven, ms-scale in the reference) ---------------- */
/* Role/term change: the caller (host election logic) decided that `leader` won
 * term `term`; appends the blank CONFIG entry a new leader always writes
 * (src/dare/dare_server.c:1411-1421).  bitmask = cid.bitmask. */
int  apus_gpu_become_leader(apus_engine_t *e, uint32_t leader, uint64_t term, uint32_t bitmask);
/* same, with the servers check_failure_count (dare_server.c:1189-1227) removes in the new
 * leader's first pass: blank CONFIG <bitmask> + CONFIG <bitmask & ~removed> commit in one pass */
/* ELECT(winner) decided on the device (start_election / poll_vote_requests / poll_vote_count,
 * src/dare/dare_server.c:1264-1743): live_mask = the servers that are up and not cut off (the winner
 * among them), bitmask = the configuration.  out[0] = 1 when the winner has size/2+1 votes, out[1] =
 * the servers that granted theirs (their logs are adjusted by apus_gpu_become_leader_ex: log_adjustment,
 * dare_ibv_rc.c:1292-1451, incl. the truncation of entries the winner does not have), out[2] = the
 * servers that refused (log longer than the winner's: they keep their log closed to it until the
 * next election), out[3] = the winner's SID as a candidate, out[4] = votes.  Synchronises. */
int  apus_gpu_elect(apus_engine_t *e, uint32_t winner, uint32_t live_mask, uint32_t bitmask, uint64_t out[8]);
int  apus_gpu_become_leader_ex(apus_engine_t *e, uint32_t leader, uint64_t term, uint32_t bitmask,
                               uint32_t removed);
/* the last entry of a replica's log, what a vote is decided on (dare_server.c:1661-1673): out[0] term, [1] idx (0, 0: the
 * log reads as empty), [2] entry slots held, [3] end offset.  Synchronises; not while a run of the replica kernels is resident. */
int  apus_gpu_last_entry(apus_engine_t *e, uint32_t replica, uint64_t out[4]);
/* cfg.group_size is the number of replicas that EXIST (the capacity); this sets the size of the
 * configuration the leader decides with (cid.size[0]: quorum = n/2+1, the servers a prune tick looks
 * at).  Default = the capacity; a group that is meant to grow starts smaller (apus_gpu_join extends it). */
int  apus_gpu_set_group_size(apus_engine_t *e, uint32_t n);
/* peer-mapped groups: the joiner's process zeroes the replica it hosts (log_new) before the leader's
 * apus_gpu_join; the other ranks take over size and epoch of the configuration the leader made */
int  apus_gpu_clear_replica(apus_engine_t *e, uint32_t replica);
int  apus_gpu_set_config(apus_engine_t *e, uint32_t group_size, uint64_t epoch);
/* JOIN: a new machine (LID `lid`) joins and is given slot r -- the lowest slot that is OFF in `bitmask`,
 * or the group size when all are taken: the group is then extended through the three CONFIG entries
 * EXTENDED -> TRANSIT -> STABLE (handle_server_join_request src/dare/dare_ibv_ud.c:973-1068,
 * apply_committed_entries src/dare/dare_server.c:1858-1937; TRANSIT commits with the new group's
 * majority, src/dare/dare_ibv_rc.c:1650-1758).  The joiner's replica is recovered on the device:
 * snapshot offset of the first follower (rc_recover_sm :597-705), the log between the leader's head and
 * the first server's end in one bulk transfer (rc_recover_log :726-866), its first persist and apply
 * passes, then it is a follower like the others.  reachable = who answers.  out[0] = new bitmask,
 * out[1] = new group size, out[2] = new epoch, out[3] = idx of the CONFIG entry that admitted the server (its
 * cid_idx).  One per-round record for the whole join.  Synchronises.
 * Refused with APUS_E_NOANSWER where the reference's joiner would retry for ever (oracle/apus_oracle.c:orc_join, -6):
 * a configured server is not reachable, or too few members would answer its RC_SYN -- a member answers only if its OWN
 * configuration shows the joiner and did not still show the slot's former holder, and a server that itself joined
 * ignores every CONFIG entry whose idx is not above the idx of the one that admitted it (all of them once the index
 * sequence has restarted at an exact-fit wrap); what each member holds is derived from the engine's journal of CONFIG
 * entries and votes (apus_amd/csrc/apus_members.h).  In the second case the CONFIG entries of the attempt are in the
 * log, as they are in the reference's, AND THE CONFIGURATION ON THE DEVICE HAS MOVED ON: out[0..2] carry it (out[1] != 0) and
 * the caller must adopt it for its later calls (elect, set_reachable, ...) although the call returned the refusal.  What the caller still decides (apus_amd/engine.py:Engine.join): no follower
 * is asked for its state machine twice without a committed <HEAD> entry in between (the reference answers from an
 * uninitialised pointer there, dare_server.c:604-651). */
int  apus_gpu_join(apus_engine_t *e, uint32_t r, uint16_t lid, uint32_t bitmask, uint32_t reachable, uint64_t out[4]);
/* replica adopts a SID it heard of from a candidate / leader that ANOTHER engine drives (its vote,
 * poll_vote_requests src/dare/dare_server.c:1690; a heartbeat of a newer term, hb_receive_cb :903-910);
 * never lowers the SID.  A leader of an older term is fenced off from then on (APUS_F_TERM_FENCE). */
int  apus_gpu_adopt_sid(apus_engine_t *e, uint32_t replica, uint64_t sid);
/* reachability of peers from the leader (KILL / HOLD / RELEASE of the trace;
 * fail_count >= PERMANENT_FAILURE or rc_connected == 0 in the reference) */
int  apus_gpu_set_reachable(apus_engine_t *e, uint32_t mask);
/* leader appends one CONFIG / HEAD / NOOP entry (data: 16-B cid, 8-B head or NULL) */
int  apus_gpu_append_control(apus_engine_t *e, uint8_t type, const void *data);

/* ---- the hot path ----------------------------------------------------------- */
/* Leader polling() passes r0 .. r0+n_rounds-1 over the staged requests:
 * get_tailq_message -> log_append_entry (dare_ibv_ud.c:780, dare_log.h:466),
 * persist_new_entries (dare_server.c:1792), update_remote_logs incl. the ACK
 * scan (dare_ibv_rc.c:1465-1826), follower ACKs (rc_send_entries_reply :1828),
 * apply_committed_entries (dare_server.c:1815).  Asynchronous on the engine
 * stream; capturable into a hipGraph. */
int  apus_gpu_run_rounds(apus_engine_t *e, uint64_t r0, uint64_t n_rounds);
/* log_pruning() timer tick (dare_server.c:1996-2067) incl. the R8 apply-offset gather */
int  apus_gpu_tick_prune(apus_engine_t *e);
/* force_log_pruning() (dare_server.c:2069-2122), the check that closes every leader pass of the
 * reference: at 75 % fill the server whose sampled apply offset holds the head back is removed from the
 * configuration (CONFIG entry), then the log is pruned.  Decided on the device.  out[0] = 1 when the log
 * was that full, out[1] = the server removed (0xFF: none), out[2] = entries appended (they commit with the
 * next pass), out[3] = the bitmask afterwards.  Synchronises; for per-pass operation (inside a batch the
 * engine refuses what does not fit instead). */
int  apus_gpu_force_prune(apus_engine_t *e, uint64_t out[4]);
/* Batching: between _begin and _end, apus_gpu_run_rounds calls (and the prune ticks that
 * fall between them, which the engine defers into the next call's sequencer) are recorded and
 * then issued as multi-segment launches -- the same polling() passes in the same order, without a
 * kernel boundary between consecutive calls.  While a batch is open only run_rounds and
 * tick_prune may be called.  No reference counterpart: polling() is a loop, a batch is a stretch
 * of its iterations submitted at once. */
int  apus_gpu_batch_begin(apus_engine_t *e);
int  apus_gpu_batch_end(apus_engine_t *e);
/* one more polling() pass everywhere: brings every reachable follower's end,
 * commit and apply up to the leader's (update_remote_logs re-send + lazy commit) */
int  apus_gpu_quiesce(apus_engine_t *e);

/* Live submission (what the DARE thread does per polling() pass; replaces the
 * malloc'd TAILQ drain of get_tailq_message, dare_ibv_ud.c:780-790):
 *   apus_gpu_append_live  n queued requests become log entries on the leader and are
 *                         pushed to the in-sync followers (payload_off relative to arena)
 *   apus_gpu_commit_live  follower ACKs + ACK scan + commit + apply for that batch;
 *                         wait_for_commit != 0 blocks until the device is done
 *   apus_gpu_submit       both, asynchronous */
int  apus_gpu_append_live(apus_engine_t *e, const apus_req_t *reqs, uint32_t n,
                          const uint8_t *arena, uint64_t arena_bytes);
int  apus_gpu_commit_live(apus_engine_t *e, int wait_for_commit);
int  apus_gpu_submit(apus_engine_t *e, const apus_req_t *reqs, uint32_t n,
                     const uint8_t *arena, uint64_t arena_bytes);

/* ---- persistent consensus kernel (the live / latency path) ------------------------
 * One resident kernel, one workgroup per local replica, runs the dare_server
 * polling() loop on the device; the host only publishes events into a pinned command
 * ring and spins on a host-visible highest_rec word (what proxy.c:160 spins on).
 * While it runs, the phased calls above must not be used.
 *   idle_ms : the kernel exits by itself after this long without an event
 *   peer_ms : bound of every device-side wait (ACKs, follower progress)            */
int  apus_gpu_persist_start(apus_engine_t *e, uint32_t idle_ms, uint32_t peer_ms);
int  apus_gpu_persist_submit(apus_engine_t *e, const apus_req_t *reqs, uint32_t n,
                             const uint8_t *arena, uint64_t arena_bytes);   /* rounds of <= 64 */
int  apus_gpu_persist_prune(apus_engine_t *e);                              /* log_pruning tick */
int  apus_gpu_persist_drain(apus_engine_t *e, uint32_t timeout_ms);         /* all events consumed */
int  apus_gpu_persist_full(apus_engine_t *e);     /* 1 once a round was refused: the log is full (the round's requests are dropped) */
uint64_t apus_gpu_persist_highest_rec(apus_engine_t *e);
const volatile uint64_t *apus_gpu_persist_highest_rec_ptr(apus_engine_t *e);
int  apus_gpu_persist_stop(apus_engine_t *e);            /* returns the kernel's exit code: 0 stop, 1 idle, 2 timeout */
int  apus_gpu_persist_latency(apus_engine_t *e, uint32_t *out_ns, uint32_t cap, uint32_t *n_out);
int  apus_gpu_persist_latency_phase(apus_engine_t *e, int which, uint32_t *out_ns, uint32_t cap, uint32_t *n_out);
/* submit one round of n <= 64 requests and spin on highest_rec, `iters` times (C-level timing) */
int  apus_gpu_persist_roundtrip(apus_engine_t *e, const apus_req_t *reqs, uint32_t n,
                                const uint8_t *arena, uint64_t arena_bytes, uint32_t iters, uint32_t *out_ns);
int  apus_gpu_device_arch(int device, char *out, int cap);

/* ---- replica kernels: every replica runs its OWN resident workgroups ---------------------
 * The one-server-per-machine structure of the reference on GPUs (apus_amd/csrc/apus_replica.h).  The
 * leader's workgroups run the leader's polling() loop pipelined (sequencer -> append wavefronts -> committer:
 * many rounds in flight) and push ONLY the log bytes + one doorbell line (128 bytes since round 6) per round to every follower;
 * each follower's workgroups -- on the follower's own device, in its own process when the replica is
 * peer-mapped -- poll their doorbells, build directory and apply records locally from the landed bytes,
 * persist, acknowledge (R3) and apply on the commit doorbell.  R3 since round 4: the reply bytes ride with the entries
 * (the leader's header stores carry reply[f] for follower f's copy and for its own; a follower that declines an entry
 * clears them), and what the leader commits on is each follower's CUMULATIVE, in-order count of the entry slots it
 * holds and has persisted -- an entry is acknowledged by f only when f holds everything in front of it; committed =
 * below the (quorum - 1)-th largest count (dare_ibv_rc.c:1738-1741), so a dead follower costs its ACK, not the round.
 * apus_gpu_rep_reserve hands out `payload_dst` in the request ring, which is DEVICE memory where the host can store into
 * it (large BAR; apus_gpu_rep_req_ring_kind) -- write-only for the caller: copy the payload there, never read it back.
 *
 * apus_gpu_rep_start launches the workgroups of every replica HOSTED by this engine (the leader's when it
 * is hosted here, and those of every hosted follower); in a peer-mapped group every process calls it.  While
 * a run is resident the phased calls must not be used; apus_gpu_rep_park ends the run (the leader's side
 * parks the followers through their mailboxes) and hands the replicas back to the control-plane calls
 * (election, JOIN, catch-up of a released follower run between runs).
 *   idle_ms : the leader's workgroups leave by themselves after this long without input
 *   peer_ms : bound of every device-side wait for a peer
 *   n_append, n_fwork : append workgroups of the leader / workgroups per follower (0 = defaults)   */
/* Link calibration -- what the reference's rc_get_loggp_params (src/dare/dare_ibv_rc.c:3323-3739) measures for RDMA, for
 * the path the replica kernels use: system-scope stores into a peer's HBM (xGMI between GPUs).  pingpong: `iters` 8-byte
 * doorbell round trips between this process's replica `me` and the mapped replica `peer` (role 0 starts and gets the
 * samples in ns, role 1 answers; both call at about the same time); store_bw: write-through 16-byte stores of `bytes`
 * into `peer`'s ring, GB/s per pass (destroys the ring's contents: before the group starts, or reset afterwards). */
int  apus_gpu_calib_pingpong(apus_engine_t *e, uint32_t me, uint32_t peer, uint32_t role, uint32_t iters, uint64_t base,
                             uint32_t *out_ns, uint32_t timeout_ms);
int  apus_gpu_calib_store_bw(apus_engine_t *e, uint32_t peer, uint64_t bytes, uint32_t iters, float *out_gbps);
/* First contact between two devices, before anything is measured (apus_amd/csrc/apus_selftest.h): a peer's kernel pushes
 * `rounds` rounds of 8 KiB (write-through stores + a doorbell, the data path's own instructions) into replica `owner`'s ring
 * and mailbox, the owner's RESIDENT kernel checks every byte and frees the region (regions: 64 .. 8192, reused every
 * `regions` rounds).  roles: 1 = this process pushes (it has `owner` mapped), 2 = this process checks (it hosts `owner`),
 * 3 = both in one launch (one device).  The two processes call it together.  out: [0] rounds checked, [1] 16-byte units that
 * differed, [2] first round that differed + 1, [3] waits that timed out.  Clears what it touched.
 * APUS_RING_ALLOC=finegrained | uncached (read by apus_gpu_create) puts the log rings into fine-grained / uncached device
 * memory: what bench.py --gpus N falls back to when the test finds a difference; apus_gpu_ring_alloc_kind: 0 / 1 / 2. */
int  apus_gpu_selftest(apus_engine_t *e, uint32_t pusher, uint32_t owner, uint32_t roles, uint64_t rounds, uint32_t regions,
                       uint32_t timeout_ms, uint64_t out[4]);
int  apus_gpu_ring_alloc_kind(apus_engine_t *e);
/* Round 6: the checker of apus_gpu_selftest also sends what a follower's work wavefront sends when it acknowledges a lone round
 * itself (apus_replica.h, REP_FAST_ACK): a system-scope atomic max into the PUSHER's mailbox, drained, in front of the word that
 * frees the region.  *misses = on the pushing side, over this engine's self-tests: regions found free before the atomic's word
 * had got that far.  Not 0: run the group with APUS_REP_DBG & 65536 (every ACK through the retire wavefronts' plain stores, as
 * in round 5) -- bench.py --gpus N does. */
int  apus_gpu_selftest_atomic_misses(apus_engine_t *e, uint64_t *misses);
/* The write-through store ceiling of the data path's own pattern: 8 KiB chunks (a round of 64 x 128 B) written at the same offset
 * into every hosted ring of `mask`, through the whole ring, `passes` times, by `wgs` workgroups of four wavefronts, nothing else
 * in the launch.  *gbps = bytes written / the launch's duration.  Destroys the rings' contents (calibration, before a group starts). */
int  apus_gpu_calib_store_multi(apus_engine_t *e, uint32_t mask, uint32_t passes, uint32_t wgs, float *gbps);
int  apus_gpu_set_leader(apus_engine_t *e, uint32_t leader);    /* a follower-only process: who leads (host mirror only, nothing is launched) */
int  apus_gpu_rep_start(apus_engine_t *e, uint32_t idle_ms, uint32_t peer_ms, uint32_t n_append, uint32_t n_fwork);
int  apus_gpu_rep_park(apus_engine_t *e);     /* exit code of the run: 0 stop, 1 idle, 2 a wait timed out, 3 a follower had a gap */
/* admission (leader side; replaces the TAILQ + tailq_lock, src/include/dare/message.h:20-22): thread safe */
int  apus_gpu_rep_reserve(apus_engine_t *e, uint32_t len, uint64_t *slot, void **payload_dst);
int  apus_gpu_rep_publish(apus_engine_t *e, uint64_t slot, const void *payload_dst, uint64_t req_id, uint16_t clt_id, uint8_t type, uint16_t len);
int  apus_gpu_rep_submit(apus_engine_t *e, const apus_req_t *reqs, uint32_t n, const uint8_t *arena, uint64_t arena_bytes);
int  apus_gpu_rep_run(apus_engine_t *e, uint64_t r0, uint64_t n_rounds);     /* staged (device-resident) rounds */
int  apus_gpu_rep_prune(apus_engine_t *e);                                   /* log_pruning tick */
/* a list of such commands in one call: cmds[3 i] = 1 (prune tick) | 2 (rounds [cmds[3 i + 1], + cmds[3 i + 2]) of the staged input),
 * the whole list `repeat` times over */
int  apus_gpu_rep_cmds(apus_engine_t *e, const uint64_t *cmds, uint32_t n, uint32_t repeat);
/* the run the leader started last: out[0] = the followers that get its rounds (they held everything the leader had when it began),
 * out[1] = the followers it could reach; one that is in out[1] only sits the run out until a control-plane pass has caught it up */
int  apus_gpu_rep_push_info(apus_engine_t *e, uint32_t out[2]);
int  apus_gpu_rep_drain(apus_engine_t *e, uint32_t timeout_ms);
int  apus_gpu_rep_full(apus_engine_t *e);                                    /* rounds refused because the log was full */
uint64_t apus_gpu_rep_highest_rec(apus_engine_t *e);
const volatile uint64_t *apus_gpu_rep_highest_rec_ptr(apus_engine_t *e);
int  apus_gpu_rep_stats(apus_engine_t *e, uint64_t out[8]);
int  apus_gpu_rep_latency(apus_engine_t *e, uint32_t *out_ns, uint32_t cap, uint32_t *n_out);
int  apus_gpu_rep_latency_appended(apus_engine_t *e, uint32_t *out_ns, uint32_t cap, uint32_t *n_out);   /* the round's bytes in every pushed ring -> committed + applied */
int  apus_gpu_rep_feed(apus_engine_t *e, const apus_req_t *reqs, uint32_t n, const uint8_t *arena, uint64_t arena_bytes,
                       uint32_t n_threads, double seconds, uint64_t prune_every_reqs, uint64_t out[2]);      /* n_threads producers on the pinned ring */
/* a process that hosts a follower, while its run is resident: progress of replica `replica` (out[0] entry slots applied,
 * [1] persisted, [2] 1 running / 2 left / 0 never started, [3] exit code) -- what its DARE thread replays into its own
 * application (proxy.c:341-439); and, when the leader is gone and nobody will ring the park doorbell: leave */
int  apus_gpu_rep_follower_progress(apus_engine_t *e, uint32_t replica, uint64_t out[4]);
int  apus_gpu_rep_follower_stop(apus_engine_t *e, uint32_t replica);
/* a host consumer replays the follower's apply stream: `slots` entry slots carried out so far.  From the first call on the
 * follower tells the leader min(device apply, host replay) as applied: pruning cannot outrun the application */
int  apus_gpu_rep_follower_replayed(apus_engine_t *e, uint32_t replica, uint64_t slots);
/* diagnostics / tests, between runs: out[0] the commit doorbell of `replica` as rung (R4, not clipped to what it holds), [1] ctrl,
 * [2] f_seq_next, [3] f_runs, [4] f_exit, [5..7] persisted_by / applied_by / seqdone_by[who] as follower `who` left them */
int  apus_gpu_rep_box_words(apus_engine_t *e, uint32_t replica, uint32_t who, uint64_t out[8]);
int  apus_gpu_rep_req_ring_kind(apus_engine_t *e);                            /* 1: the request ring is device memory behind the BAR, 0: pinned host memory, -1: no run yet */
int  apus_gpu_rep_launch_ms(apus_engine_t *e, double *ms);                    /* duration of the last (parked) run's resident launch, HIP events on its stream */
int  apus_gpu_rep_role_stats(apus_engine_t *e, uint64_t out[20][8]);          /* diagnostics: passes / rounds / time of the serial roles of the last run */
int  apus_gpu_rep_roundtrip(apus_engine_t *e, const apus_req_t *reqs, uint32_t n, const uint8_t *arena, uint64_t arena_bytes,
                            uint32_t iters, uint32_t *out_ns);

/* ---- multi-process groups: one replica per GPU / process --------------------------
 * The leader's new log range [end before, end after) and the matching directory
 * slots travel between processes by RCCL point-to-point (apus_amd/distributed.py);
 * these are the device halves on either side (R1/R2: ingest, R3: ack_merge,
 * R4: follower_commit). */
int  apus_gpu_follow(apus_engine_t *e, uint32_t replica, uint32_t leader, uint64_t term, uint32_t bitmask);
int  apus_gpu_append_rounds(apus_engine_t *e, uint64_t r0, uint64_t n_rounds);   /* leader: append only */
int  apus_gpu_commit_rounds(apus_engine_t *e, uint64_t r0, uint64_t n_rounds);   /* leader: ACK scan, commit, apply */
int  apus_gpu_ship_info(apus_engine_t *e, uint64_t out[8]);
int  apus_gpu_ingest(apus_engine_t *e, uint32_t replica, uint64_t vis_slot, uint64_t n_hint);
int  apus_gpu_ack_merge(apus_engine_t *e, uint32_t follower, uint64_t from_slot, uint64_t upto_slot);
int  apus_gpu_follower_commit(apus_engine_t *e, uint32_t replica, uint64_t commit_slot, uint64_t n_hint);

/* hipGraph capture of a sequence of the asynchronous calls above */
int  apus_gpu_capture_begin(apus_engine_t *e);
int  apus_gpu_capture_end(apus_engine_t *e, int *graph_id);
int  apus_gpu_graph_launch(apus_engine_t *e, int graph_id);

/* ---- observation (synchronising) ------------------------------------------ */
/* out[8] = head, apply, commit, end, tail, old_end, old_commit, len  (dare_log_t order) */
int  apus_gpu_offsets(apus_engine_t *e, uint32_t replica, uint64_t out[8]);
/* out[8] = n_end, n_persist, n_commit, n_apply (entry counts), last_idx, sid, highest_rec, apply_hash */
int  apus_gpu_counters(apus_engine_t *e, uint32_t replica, uint64_t out[8]);
int  apus_gpu_read_ring(apus_engine_t *e, uint32_t replica, uint64_t off, uint64_t n, void *dst);
/* per-round record of the leader: end and commit offsets after every round since the last reset */
int  apus_gpu_round_record(apus_engine_t *e, uint64_t first, uint64_t n, uint64_t *end_out, uint64_t *commit_out);
uint64_t apus_gpu_round_count(apus_engine_t *e);
/* apply stream records of slots [first, first+n) of one replica */
int  apus_gpu_apply_records(apus_engine_t *e, uint32_t replica, uint64_t first, uint64_t n, apus_apply_t *out);
/* Durability side channel: the records persist_new_entries -> proxy_store_cmd -> stablestorage_save_request
 * (src/dare/dare_server.c:1802, src/proxy/proxy.c:268-291) appends to BerkeleyDB for entry slots
 * [first, first + n) of `replica`, back to back = the bytes of dump_records / a joiner's snapshot
 * (proxy.c:300-304).  The reference's layout incl. its overlay: CONNECT / CLOSE 4 bytes,
 * SEND 24 + (reply[4] | reply[5] << 8) bytes from clt_id on.  *bytes = total size (nothing beyond cap is
 * written), *records = number of records.  Synchronises. */
int  apus_gpu_store_stream(apus_engine_t *e, uint32_t replica, uint64_t first, uint64_t n,
                           void *dst, uint64_t cap, uint64_t *bytes, uint64_t *records);
uint32_t apus_gpu_status(apus_engine_t *e);
void apus_gpu_clear_status(apus_engine_t *e);
int  apus_gpu_status_words(apus_engine_t *e, uint32_t out[8]);   /* diagnostics: status bits, first spin-timeout site, fence word, five words the flagging kernel left */
/* raw device pointers for zero-copy wrapping by the host transport (RCCL p2p):
 * which: 0 ring, 1 hdr, 2 dir_off, 3 dir_len, 4 ack, 5 apply ring */
void *apus_gpu_device_ptr(apus_engine_t *e, uint32_t replica, int which, uint64_t *bytes);
/* HIP-event timing of the dominant kernel (which = 0: k_append_push): enable, run
 * eagerly, then read the summed duration of its launches */
int  apus_gpu_set_timing(apus_engine_t *e, int on);
int  apus_gpu_kernel_time(apus_engine_t *e, int which, float *total_ms, uint64_t *launches);
/* first n_words (<= 64) of a replica's control block (apus_device.h H_* words) */
int  apus_gpu_hdr_words(apus_engine_t *e, uint32_t replica, uint64_t *out, uint32_t n_words);
void *apus_gpu_stream(apus_engine_t *e);          /* the hipStream_t the engine launches on */
apus_engine_t *apus_gpu_global(void);

/* ---- B-transport: the reference's own names ------------------------------- */
/* Bind the process-wide engine the dare_ib_* facade operates on (the reference
 * keeps a global singleton too: dare_ib_device, src/dare/dare_ibv.c:33). */
int  apus_gpu_bind_global(apus_engine_t *e);
void dare_ib_poll_tailq(void);                       /* dare_ibv.h:154 -> get_tailq_message, dare_ibv_ud.c:780 */
int  dare_ib_write_remote_logs(int wait_for_commit); /* dare_ibv.h:176 -> rc_write_remote_logs, dare_ibv_rc.c:1870 */
int  dare_ib_send_entries_reply(uint8_t idx);        /* dare_ibv.h:177 -> rc_send_entries_reply, dare_ibv_rc.c:1828 */
int  dare_ib_get_remote_apply_offsets(void);         /* dare_ibv.h:178 -> rc_get_remote_apply_offsets, dare_ibv_rc.c:1970 */

#ifdef __cplusplus
}
#endif
#endif /* APUS_GPU_H */
