/*
 * apus_oracle.h -- CPU restatement of the APUS/DARE consensus hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the reported CPU baseline.
 *
 * Parity pin: the reference has no tests, golden vectors or KATs for this
 * path, so the oracle is pinned on the reference ITSELF,
 * run here:
 *  - the log layer against the reference's own header, src/include/dare/dare_log.h,
 *    compiled unchanged into oracle/_ref/libapus_ref.so (ref_harness.c;
 *    tests/test_oracle_vs_ref.py, fixtures tests/golden/digests.json);
 *  - the consensus loops against the reference's own dare_server.c, dare_ibv.c,
 *    dare_ibv_rc.c, dare_ibv_ud.c, dare_ep_db.c, dare_kvs_sm.c, config-dare.c and
 *    rbtree.c, compiled UNMODIFIED into oracle/_ref/libapus_ref_loops.so behind
 *    in-process stand-ins for <infiniband/verbs.h>, <ev.h> and <libconfig.h>
 *    (oracle/refshim/, recipe oracle/Makefile `loops`): one private copy per
 *    server, a shared in-process fabric, driven one polling() pass at a time.
 *    tests/test_oracle_vs_refloops.py replays 22 named traces (steady state, hold /
 *    release, no quorum, fail-overs with divergence and truncation, JOIN into an empty
 *    slot / extending the group / while the log wraps, force_log_pruning evictions)
 *    + BASELINE configs[1] at full size on both in lock step (all offsets, every
 *    defined ring byte, SID, configuration, counters, apply upcalls, the bytes handed
 *    to the storage callback, the leader's end/commit after every pass); the same
 *    records are committed as tests/golden/cluster_ref.json (written from the
 *    reference's outputs by tests/golden/make_cluster_golden.py) and travel to
 *    the GPU box.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef APUS_ORACLE_H
#define APUS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SERVERS   13          /* MAX_SERVER_COUNT, src/include/dare/dare.h:26 */
#define ORC_HDR_BYTES     64          /* sizeof(dare_log_entry_t), dare_log.h:33-47   */
#define ORC_DEFAULT_LOG   (16384ull * 4096ull) /* LOG_SIZE, dare_log.h:76             */
#define ORC_MAX_NC        1024        /* MAX_NC_ENTRIES, dare_log.h:60                */

/* entry types: dare_log.h:22-25 and src/include/proxy/proxy.h:10-12 */
enum { ORC_NOOP = 0, ORC_CSM = 1, ORC_CONFIG = 2, ORC_HEAD = 3,
       ORC_CONNECT = 4, ORC_SEND = 5, ORC_CLOSE = 6 };

/* dare_cid_t, src/include/dare/dare_config.h:34-40 (16 bytes) */
typedef struct {
    uint64_t epoch;
    uint8_t  size[2];
    uint8_t  state;
    uint8_t  pad;
    uint32_t bitmask;
} orc_cid_t;

/* dare_log_entry_det_t, dare_log.h:51-56 */
typedef struct { uint64_t idx, term, offset; } orc_det_t;

/* dare_nc_buf_t, dare_log.h:61-65 */
typedef struct { uint64_t len; orc_det_t entries[ORC_MAX_NC]; } orc_ncbuf_t;

typedef struct orc_log orc_log_t;
typedef struct orc_cluster orc_cluster_t;

/* ------------------------------------------------------------------ */
/* log level: one function per dare_log.h function                     */
orc_log_t *orc_log_new(uint64_t len);                       /* log_new :120           */
void       orc_log_free(orc_log_t *log);
/* returns idx, 0 when the log is full (entry dropped)   log_append_entry :466-558     */
uint64_t   orc_log_append(orc_log_t *log, uint64_t term, uint64_t req_id,
                          uint16_t clt_id, uint8_t type,
                          const void *data, uint16_t data_len);
void       orc_log_offsets(const orc_log_t *log, uint64_t out[8]); /* head,apply,commit,end,tail,old_end,old_commit,len */
void       orc_log_set_offsets(orc_log_t *log, const uint64_t in[8]);
uint8_t   *orc_log_entries(orc_log_t *log);
int        orc_log_prev_head(const orc_log_t *log);          /* prev_log_entry_head    */
void       orc_log_set_prev_head(orc_log_t *log, int v);
uint64_t   orc_log_end_distance(const orc_log_t *log, uint64_t off);      /* :255      */
int        orc_log_is_larger(const orc_log_t *log, uint64_t l, uint64_t r); /* :269    */
/* log_get_entry :316 -- returns the (possibly redirected) offset or UINT64_MAX for NULL */
uint64_t   orc_log_get_entry(const orc_log_t *log, uint64_t off);
uint32_t   orc_log_entry_len_at(const orc_log_t *log, uint64_t off);       /* :228     */
uint64_t   orc_log_get_tail(const orc_log_t *log);                         /* :402     */
void       orc_log_to_ncbuf(const orc_log_t *log, orc_ncbuf_t *nc);        /* :339     */
uint64_t   orc_log_find_remote_end(const orc_log_t *log, const orc_ncbuf_t *nc); /* :367 */

/* ------------------------------------------------------------------ */
/* cluster level: N in-process replicas, "RDMA" = memcpy at same offset */

/* one admitted client request (what a tailq_entry_t carries, message.h:11-18) */
typedef struct {
    uint64_t req_id;
    uint64_t payload_off;     /* into the payload arena handed to orc_round */
    uint16_t clt_id;          /* connection_id */
    uint16_t len;
    uint8_t  type;            /* ORC_CONNECT / ORC_SEND / ORC_CLOSE */
    uint8_t  pad[3];
} orc_req_t;

/* one record of the apply stream (callbacks of apply_committed_entries) */
typedef struct {
    uint64_t slot;            /* number of entries this replica applied before this one */
    uint64_t off;             /* entry offset in the ring */
    uint64_t idx;
    uint32_t len;             /* cmd.len */
    uint16_t clt_id;
    uint8_t  type;
    uint8_t  kind;            /* 1 = proxy_update_state (leader), 2 = proxy_do_action (follower) */
} orc_apply_t;

orc_cluster_t *orc_cluster_new(int group_size, uint64_t log_len);
void           orc_cluster_free(orc_cluster_t *c);
void           orc_cluster_record_apply(orc_cluster_t *c, int on);
/* the durability side channel: the records proxy_store_cmd hands to BerkeleyDB, back to back (= what
 * dump_records / the snapshot of a joiner's donor holds), and db-interface.c's records_len */
void           orc_cluster_record_store(orc_cluster_t *c, int on);
const uint8_t *orc_replica_store_stream(const orc_cluster_t *c, int r, uint64_t *n);
uint32_t       orc_replica_records_len(const orc_cluster_t *c, int r);
/* by default orc_round fails (-3) when an append lands exactly on len (Q13) */
void           orc_cluster_allow_exact_fit(orc_cluster_t *c, int on);
/* 0 (default): a posted WR's completion is seen by the poll behind the post; 1: by the next loop pass */
void           orc_cluster_completion_delay(orc_cluster_t *c, int on);
/* how often force_log_pruning (dare_server.c:2069) found the log >= 75 % full and acted */
uint64_t       orc_force_prune_count(const orc_cluster_t *c);

/* ELECT(winner): start_election -> votes -> poll_vote_count -> blank CONFIG.
 * returns 0, or -1 if the winner cannot collect a majority. */
int      orc_elect(orc_cluster_t *c, int winner);
/* one leader polling() iteration that finds n requests queued */
int      orc_round(orc_cluster_t *c, const orc_req_t *reqs, int n, const uint8_t *arena);
int      orc_tick_prune(orc_cluster_t *c);                    /* prune_log_cb        */
int      orc_kill(orc_cluster_t *c, int r);
int      orc_hold(orc_cluster_t *c, int r);
int      orc_release(orc_cluster_t *c, int r);
int      orc_quiesce(orc_cluster_t *c);                       /* poll until fixpoint */
/* JOIN(r): a new server joins and must be given slot r -- an empty one, or group_size (the group is
 * extended: EXTENDED -> TRANSIT -> STABLE).  Return codes: see apus_oracle.c */
int      orc_join(orc_cluster_t *c, int r);

int        orc_leader(const orc_cluster_t *c);
int        orc_group_size(const orc_cluster_t *c);
orc_log_t *orc_replica_log(orc_cluster_t *c, int r);
uint64_t   orc_replica_sid(const orc_cluster_t *c, int r);
uint32_t   orc_replica_cid_bitmask(const orc_cluster_t *c, int r);   /* SID.cid.bitmask: the configured servers */
void       orc_replica_cid(const orc_cluster_t *c, int r, orc_cid_t *out);
int        orc_replica_alive(const orc_cluster_t *c, int r);
uint64_t   orc_replica_highest_rec(const orc_cluster_t *c, int r);
uint64_t   orc_replica_apply_count(const orc_cluster_t *c, int r);
uint64_t   orc_replica_apply_hash(const orc_cluster_t *c, int r);
uint64_t   orc_replica_store_count(const orc_cluster_t *c, int r);
const orc_apply_t *orc_replica_apply_log(const orc_cluster_t *c, int r, uint64_t *n);
/* per-round record kept by the leader: end and commit after each orc_round */
uint64_t   orc_round_count(const orc_cluster_t *c);
const uint64_t *orc_round_commit(const orc_cluster_t *c);
const uint64_t *orc_round_end(const orc_cluster_t *c);

/* whole-trace driver used by the cpu_baseline leg: rounds[r] = #requests in round r,
 * a prune tick is taken whenever >= prune_bytes were appended since the last one */
int      orc_run_rounds(orc_cluster_t *c, const orc_req_t *reqs, const uint32_t *round_n,
                        uint64_t n_rounds, const uint8_t *arena, uint64_t prune_bytes);

/* the same steady-state stream on one THREAD per server (pinned; the leader thread also does its NIC's
 * work: log WRITEs and doorbells), for bench.py's CPU baseline only.  No failures; ends
 * with every server in step, rings / offsets / upcall counts equal to orc_run_rounds'.  -10: a spin ran
 * past max_seconds. */
int      orc_run_rounds_mt(orc_cluster_t *c, const orc_req_t *reqs, const uint32_t *round_n,
                           uint64_t n_rounds, const uint8_t *arena, uint64_t prune_bytes, double max_seconds);

/* ------------------------------------------------------------------ */
/* helpers shared by tests: payload stream and canonical digest walk   */
uint64_t orc_splitmix64(uint64_t *state);
void     orc_fill_payload(uint64_t seed, uint8_t *dst, uint32_t len);

/* apply-stream hash = sum (mod 2^64) over all upcalls of orc_apply_mix(record);
 * position dependent through `slot`, so it can be accumulated in any order */
uint64_t orc_apply_mix(uint64_t slot, uint64_t off, uint64_t idx, uint32_t len,
                       uint16_t clt_id, uint8_t type, uint8_t kind);

/* Canonical serialisation of the entries in [from, to) of a ring whose end
 * offset is `end` (SURVEY.md section 8c digest): per entry
 *   u64 offset, u64 idx, u64 term, u64 req_id, u16 clt_id, u8 type, u8 0,
 *   u32 data_len, data bytes (payload | 16-B cid | 8-B head | none).
 * Returns the number of bytes produced (even when > cap; nothing beyond cap is
 * written) and stores the number of entries in *n_entries. */
uint64_t orc_canon(const uint8_t *ring, uint64_t len, uint64_t end,
                   uint64_t from, uint64_t to,
                   uint8_t *out, uint64_t cap, uint64_t *n_entries);
/* FNV-1a-64 of the canonical stream (no buffer needed) */
uint64_t orc_canon_hash(const uint8_t *ring, uint64_t len, uint64_t end,
                        uint64_t from, uint64_t to, uint64_t *n_entries);

/* mask of the bytes that entries in [from, to) define (see apus_oracle.c) */
uint64_t orc_defined_mask(const uint8_t *ring, uint64_t len, uint64_t end,
                          uint64_t from, uint64_t to, uint8_t *mask);

#ifdef __cplusplus
}
#endif
#endif
