/*
 * apus_oracle.c -- CPU restatement of the APUS/DARE consensus hot path.
 * TEST INFRASTRUCTURE ONLY (see apus_oracle.h for the rules and the parity pin).
 *
 * Part 1 restates src/include/dare/dare_log.h function by function.
 * Part 2 restates the loops of src/dare/dare_server.c and
 * src/dare/dare_ibv_rc.c over N in-process replicas; an RDMA WRITE is a
 * memcpy to the same offset of the peer's structure, an RDMA READ a load.
 *
 * Schedule (the reference is timing dependent, the oracle is not): every
 * posted WRITE/READ completes before the leader's next loop iteration, and a
 * follower runs one polling() pass right after each doorbell (end / commit)
 * lands in its memory.  Log bytes, offsets and the order of apply callbacks
 * at every quiescent point do not depend on that choice.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE            /* CPU_SET, pthread_setaffinity_np (Part 3) */
#endif
#include "apus_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ================================================================== */
/* Part 1: the log (dare_log.h)                                        */

/* dare_log_entry_t, dare_log.h:33-47; offsets probed in SURVEY.md section 10.
 * Entries lie at ANY byte offset of the ring (an entry is 64 B + its payload's length, nothing is rounded up), and the reference
 * reads them through a plain struct pointer -- fine on x86, undefined in C.  Here the type is packed (alignment 1, the padding in
 * front of `data` spelled out), so every access is an unaligned-safe one and the checker runs clean under UBSan
 * (tools/oracle_sanitize.sh); the asserts below hold the reference's offsets. */
typedef struct __attribute__((packed)) {
    uint64_t idx;                       /*  0 */
    uint64_t term;                      /*  8 */
    uint64_t req_id;                    /* 16 */
    uint16_t clt_id;                    /* 24 */
    uint8_t  type;                      /* 26 */
    uint8_t  sender;                    /* 27 */
    uint8_t  reply[ORC_MAX_SERVERS];    /* 28 */
    uint8_t  pad_[48 - 28 - ORC_MAX_SERVERS];          /* 41 (the reference's struct has this padding implicitly) */
    union __attribute__((packed)) {
        struct __attribute__((packed)) { uint16_t len; uint8_t cmd[]; } cmd;   /* 48 ; payload @50 */
        orc_cid_t cid;
        uint64_t  head;
    } data;
} orc_entry_t;

_Static_assert(sizeof(orc_entry_t) == ORC_HDR_BYTES, "entry header must be 64 bytes");
_Static_assert(offsetof(orc_entry_t, clt_id) == 24, "clt_id@24");
_Static_assert(offsetof(orc_entry_t, reply) == 28, "reply@28");
_Static_assert(offsetof(orc_entry_t, data) == 48, "data@48");
_Static_assert(sizeof(orc_cid_t) == 16, "cid is 16 bytes");

/* dare_log_t, dare_log.h:77-102 (same field order; entries kept out of line) */
struct orc_log {
    uint64_t head, apply, commit, end, tail, old_end, old_commit, len;
    orc_ncbuf_t nc_buf[ORC_MAX_SERVERS];
    int prev_head;          /* the global prev_log_entry_head, dare_server.c:71 */
    uint8_t *entries;
};

orc_log_t *orc_log_new(uint64_t len)           /* log_new, dare_log.h:120-136 */
{
    orc_log_t *log = calloc(1, sizeof *log);
    if (!log) return NULL;
    log->entries = calloc(1, len + 1024);       /* slack: a store record may read a little past the last entry */
    if (!log->entries) { free(log); return NULL; }
    log->len = len;
    log->end = len;
    log->tail = len;
    log->old_end = len;
    return log;
}

void orc_log_free(orc_log_t *log)
{
    if (!log) return;
    free(log->entries);
    free(log);
}

void orc_log_offsets(const orc_log_t *log, uint64_t out[8])
{
    out[0] = log->head; out[1] = log->apply; out[2] = log->commit; out[3] = log->end;
    out[4] = log->tail; out[5] = log->old_end; out[6] = log->old_commit; out[7] = log->len;
}

void orc_log_set_offsets(orc_log_t *log, const uint64_t in[8])
{
    log->head = in[0]; log->apply = in[1]; log->commit = in[2]; log->end = in[3];
    log->tail = in[4]; log->old_end = in[5]; log->old_commit = in[6];
    /* len is fixed at creation */
}

uint8_t *orc_log_entries(orc_log_t *log) { return log->entries; }
int orc_log_prev_head(const orc_log_t *log) { return log->prev_head; }
void orc_log_set_prev_head(orc_log_t *log, int v) { log->prev_head = v; }

static inline int log_empty(const orc_log_t *log) { return log->end == log->len; }   /* :158 */
static inline int log_full(const orc_log_t *log)  { return log->end == log->head; }  /* :168 */
static inline int fits_header(const orc_log_t *log, uint64_t off)                    /* :201 */
{ return log->len - off >= ORC_HDR_BYTES; }

static inline orc_entry_t *entry_at(const orc_log_t *log, uint64_t off)
{ return (orc_entry_t *)(log->entries + off); }

static inline uint32_t entry_len(const orc_entry_t *e)                               /* :228 */
{
    if (e->type == ORC_NOOP || e->type == ORC_CONFIG || e->type == ORC_HEAD)
        return ORC_HDR_BYTES;
    return ORC_HDR_BYTES + e->data.cmd.len;
}

static inline int fits_entry(const orc_log_t *log, uint64_t off, const orc_entry_t *e) /* :241 */
{ return log->len - off >= entry_len(e); }

uint64_t orc_log_end_distance(const orc_log_t *log, uint64_t off)                    /* :255 */
{
    uint64_t end = log->end;
    if (end == log->len) return 0;
    if (end >= off) return end - off;
    return log->len - (off - end);
}

int orc_log_is_larger(const orc_log_t *log, uint64_t l, uint64_t r)                  /* :269 */
{
    /* "larger" means closer to end */
    return orc_log_end_distance(log, l) < orc_log_end_distance(log, r);
}

/* log_get_entry :316-331; *off may be redirected to 0 */
static orc_entry_t *get_entry(const orc_log_t *log, uint64_t *off)
{
    if (log_empty(log)) return NULL;
    if (orc_log_end_distance(log, *off) == 0) return NULL;
    if (!fits_header(log, *off)) *off = 0;
    return entry_at(log, *off);
}

uint64_t orc_log_get_entry(const orc_log_t *log, uint64_t off)
{
    orc_entry_t *e = get_entry(log, &off);
    return e ? off : UINT64_MAX;
}

uint32_t orc_log_entry_len_at(const orc_log_t *log, uint64_t off)
{ return entry_len(entry_at(log, off)); }

/* the walk shared by log_entries_to_nc_buf, log_get_tail and all the readers:
 * step over one entry, honouring the "does not fit -> continues at 0" rule */
static inline void step_over(const orc_log_t *log, uint64_t *off, const orc_entry_t *e)
{
    if (!fits_entry(log, *off, e)) *off = 0;
    *off += entry_len(e);
}

void orc_log_to_ncbuf(const orc_log_t *log, orc_ncbuf_t *nc)                          /* :339-361 */
{
    uint64_t off = log->commit, n = 0;
    orc_entry_t *e;
    while ((e = get_entry(log, &off)) != NULL) {
        nc->entries[n].idx = e->idx;
        nc->entries[n].term = e->term;
        nc->entries[n].offset = off;
        n++;
        step_over(log, &off, e);
    }
    nc->len = n;
}

uint64_t orc_log_find_remote_end(const orc_log_t *log, const orc_ncbuf_t *nc)         /* :367-395 */
{
    uint64_t off = 0;
    for (uint64_t i = 0; i < nc->len; i++) {
        off = nc->entries[i].offset;
        orc_entry_t *e = get_entry(log, &off);
        if (!e) return off;
        if (e->idx != nc->entries[i].idx || e->term != nc->entries[i].term) return off;
        step_over(log, &off, e);
    }
    return off;
}

static uint64_t scan_for_tail(const orc_log_t *log, uint64_t from)
{
    uint64_t off = from, tail = log->len;
    orc_entry_t *e;
    while ((e = get_entry(log, &off)) != NULL) {
        tail = off;
        step_over(log, &off, e);
    }
    return tail;
}

uint64_t orc_log_get_tail(const orc_log_t *log)                                       /* :402-457 */
{
    if (log->tail != log->len) return log->tail;
    if (log_empty(log)) return log->len;
    uint64_t t = scan_for_tail(log, log->commit);
    if (t != log->len) return t;
    t = scan_for_tail(log, log->apply);
    if (t != log->len) return t;
    return scan_for_tail(log, log->head);
}

/* log_add_new_entry :213-221 */
static orc_entry_t *new_entry_slot(const orc_log_t *log)
{
    if (log_full(log)) return NULL;
    if (log_empty(log) || !fits_header(log, log->end)) return entry_at(log, 0);
    return entry_at(log, log->end);
}

static void fill_header(orc_entry_t *e, uint64_t idx, uint64_t term, uint64_t req_id,
                        uint16_t clt_id, uint8_t type)
{
    e->idx = idx; e->term = term; e->req_id = req_id;
    e->clt_id = clt_id; e->type = type;
    memset(e->reply, 0, ORC_MAX_SERVERS);
}

uint64_t orc_log_append(orc_log_t *log, uint64_t term, uint64_t req_id,
                        uint16_t clt_id, uint8_t type,
                        const void *data, uint16_t data_len)                          /* :466-558 */
{
    if (type != ORC_HEAD) log->prev_head = 0;                     /* :477-480 */

    if (log->tail == log->len) log->tail = orc_log_get_tail(log); /* :483-485 */
    uint64_t off = log->tail;
    orc_entry_t *last = get_entry(log, &off);
    uint64_t idx = last ? last->idx + 1 : 1;                       /* :486-488 */

    orc_entry_t *e = new_entry_slot(log);
    if (!e) return 0;                                              /* log full, :492-495 */
    fill_header(e, idx, term, req_id, clt_id, type);
    if (!fits_header(log, log->end)) log->end = 0;                 /* :502-504 */

    switch (type) {
    case ORC_CONFIG: memcpy(&e->data.cid, data, sizeof(orc_cid_t)); break;
    case ORC_HEAD:   memcpy(&e->data.head, data, sizeof(uint64_t)); break;
    case ORC_NOOP:   break;
    default:
        e->data.cmd.len = data_len;
        if (!fits_entry(log, log->end, e)) {                       /* :521-537 */
            /* the header just written stays behind as a stale header */
            log->end = 0;
            e = new_entry_slot(log);
            if (!e) return 0;
            fill_header(e, idx, term, req_id, clt_id, type);
            e->data.cmd.len = data_len;
        }
        if (data_len) memcpy(e->data.cmd.cmd, data, data_len);
        break;
    }
    log->tail = log->end;                                          /* :547 */
    log->end += entry_len(e);                                      /* :549 */
    return idx;
}

/* ================================================================== */
/* Part 2: the replicated state machine loops                          */

/* SID word, src/include/dare/dare_server.h:47-66 */
#define SID_IDX(s)    ((uint8_t)((s) & 0xFF))
#define SID_L(s)      ((s) & (1ull << 8))
#define SID_TERM(s)   ((s) >> 9)
#define SID_MAKE(t, l, i) (((uint64_t)(t) << 9) | ((l) ? (1ull << 8) : 0) | (uint64_t)(i))

/* log replication steps, dare_server.h:78-84 */
enum { LR_GET_WRITE = 1, LR_GET_NCE_LEN, LR_GET_NCE, LR_SET_END, LR_UPDATE_LOG, LR_UPDATE_END };

enum { PEND_NONE = 0, PEND_LOG, PEND_END, PEND_ADJ };

typedef struct {
    orc_log_t *log;
    uint64_t sid;                               /* ctrl_data->sid */
    orc_cid_t cid;                              /* config.cid     */
    uint64_t cid_offset, cid_idx;               /* server_config_t */
    uint8_t  idx;
    int alive, held;
    /* leader-side view of the peers (ctrl_data_t / server_t) */
    uint64_t rem_end[ORC_MAX_SERVERS];          /* log_offsets[i].end    */
    uint64_t rem_commit[ORC_MAX_SERVERS];       /* log_offsets[i].commit */
    uint64_t apply_offsets[ORC_MAX_SERVERS];
    uint64_t vote_ack[ORC_MAX_SERVERS];
    uint64_t cached_end[ORC_MAX_SERVERS];       /* server_t.cached_end_offset */
    uint8_t  lr_step[ORC_MAX_SERVERS];
    uint8_t  send_flag[ORC_MAX_SERVERS];
    uint8_t  pending[ORC_MAX_SERVERS];          /* what the outstanding WR was */
    /* upcall bookkeeping (what the proxy callbacks would observe) */
    uint64_t highest_rec;                       /* proxy.c:263 */
    uint64_t store_count;                       /* proxy_store_cmd calls */
    uint64_t apply_count, apply_hash, apply_rec_base;
    uint64_t apply_slot;                        /* entries walked by apply_committed_entries */
    orc_apply_t *apply_log; uint64_t apply_cap;
    orc_det_t last_applied;                     /* dare_server.c:73 */
    /* joining */
    uint16_t lid;                               /* the machine's LID: clt_id of the CONFIG entry that admits it */
    int      resync_armed;                      /* checker's numbering only: see orc_join */
    uint64_t resync_off, resync_slot;
    int      snapshot_on;                       /* dare_state & SNAPSHOT, dare_server.c:641 */
    uint64_t snapshot_last;                     /* snapshot->last_entry.offset, :637 */
    /* durability side channel: what proxy_store_cmd hands to BerkeleyDB */
    uint8_t *store_buf; uint64_t store_len, store_cap;
    uint32_t records_len;                       /* db-interface.c:17,81 */
} replica_t;

struct orc_cluster {
    int n;
    uint64_t log_len;
    int leader;                                  /* -1 when none */
    int record_apply;
    int allow_exact_fit;                         /* let SURVEY.md Q13 happen instead of failing */
    int committed_flag;                          /* `committed`, dare_ibv_rc.c:1461 */
    int completion_delay;                        /* see posted() */
    uint64_t force_prunes;                       /* times force_log_pruning found the log >= 75 % full */
    /* a JOIN in progress: what the leader's reply carries (reconf_rep_t, dare_ibv_ud.c:1451-1490) */
    int next_lid, join_slot, join_replied;
    int hung;                                    /* a persist walk that does not terminate (orc_join, -8) */
    int leader_quit;                             /* the leader shut itself down (check_failure_count, -9) */
    int record_store;
    uint64_t join_head, join_cid_idx;
    orc_cid_t join_cid;
    replica_t r[ORC_MAX_SERVERS];
    uint64_t *round_commit, *round_end; uint64_t n_rounds, rounds_cap;
};

static inline int is_leader_r(const replica_t *p)                /* IS_LEADER, dare_server.c:42-46 */
{ return SID_L(p->sid) && SID_IDX(p->sid) == p->idx; }

static inline int cid_on(const orc_cid_t *cid, int i) { return (cid->bitmask >> i) & 1; }

/* configuration states, src/include/dare/dare_config.h:18-24 */
enum { CID_STABLE = 0, CID_TRANSIT = 1, CID_EXTENDED = 2 };
static inline int ext_group_size(const orc_cid_t *cid)          /* get_extended_group_size, dare_config.h:78-86 */
{
    if (cid->state == CID_STABLE) return cid->size[0];
    return cid->size[0] < cid->size[1] ? cid->size[1] : cid->size[0];
}
static inline int group_size(const orc_cid_t *cid)              /* get_group_size, dare_config.h:89-97 */
{
    if (cid->state != CID_TRANSIT) return cid->size[0];
    return cid->size[0] < cid->size[1] ? cid->size[1] : cid->size[0];
}
/* the `size` the commit scan and the lazy commit loop of update_remote_logs run with: what the (dead)
 * offset-median loop above them leaves behind, dare_ibv_rc.c:1650-1723 -- cid.size[0] in a STABLE or
 * EXTENDED configuration ("only the old majority"), cid.size[1] in a TRANSIT one (the loop always
 * ends with j == 1 there), i.e. the NEW group's majority alone, not both */
static inline int scan_size(const orc_cid_t *cid)
{ return cid->state == CID_TRANSIT ? cid->size[1] : cid->size[0]; }

uint64_t orc_apply_mix(uint64_t slot, uint64_t off, uint64_t idx, uint32_t len,
                       uint16_t clt_id, uint8_t type, uint8_t kind)
{
    uint64_t x = slot * 0x9E3779B97F4A7C15ull ^ off * 0xC2B2AE3D27D4EB4Full
               ^ idx * 0x165667B19E3779F9ull
               ^ ((uint64_t)len << 32 | (uint64_t)clt_id << 16 | (uint64_t)type << 8 | kind);
    x ^= x >> 31;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 29;
    return x;
}

static void record_apply(orc_cluster_t *c, replica_t *p, uint64_t off, const orc_entry_t *e, uint8_t kind)
{
    p->apply_hash += orc_apply_mix(p->apply_slot, off, e->idx, e->data.cmd.len, e->clt_id, e->type, kind);
    if (c->record_apply) {
        const uint64_t k = p->apply_count - p->apply_rec_base;     /* (records are kept from where recording was last switched on) */
        if (k == p->apply_cap) {
            p->apply_cap = p->apply_cap ? p->apply_cap * 2 : 1024;
            p->apply_log = realloc(p->apply_log, p->apply_cap * sizeof(orc_apply_t));
        }
        orc_apply_t *a = &p->apply_log[k];
        a->slot = p->apply_slot; a->off = off; a->idx = e->idx; a->len = e->data.cmd.len;
        a->clt_id = e->clt_id; a->type = e->type; a->kind = kind;
    }
    p->apply_count++;
}

orc_cluster_t *orc_cluster_new(int group_size, uint64_t log_len)
{
    if (group_size < 1 || group_size > ORC_MAX_SERVERS) return NULL;
    orc_cluster_t *c = calloc(1, sizeof *c);
    if (!c) return NULL;
    c->n = group_size;
    c->log_len = log_len;
    c->leader = -1;
    for (int i = 0; i < group_size; i++) {
        replica_t *p = &c->r[i];
        p->log = orc_log_new(log_len);
        p->idx = (uint8_t)i;
        p->alive = 1;
        /* init_server_data, dare_server.c:283-296: stable cid of group_size, all ON */
        p->cid.epoch = 0; p->cid.size[0] = (uint8_t)group_size; p->cid.size[1] = 0;
        p->cid.state = 0; p->cid.bitmask = (1u << group_size) - 1;
        p->sid = SID_MAKE(0, 0, i);
        p->lid = (uint16_t)(i + 1);
        for (int j = 0; j < ORC_MAX_SERVERS; j++) {
            p->vote_ack[j] = log_len;
            p->lr_step[j] = LR_GET_WRITE;
            p->send_flag[j] = 1;
        }
    }
    c->next_lid = group_size;
    c->join_slot = -1;
    return c;
}

void orc_cluster_free(orc_cluster_t *c)
{
    if (!c) return;
    for (int i = 0; i < ORC_MAX_SERVERS; i++) { orc_log_free(c->r[i].log); free(c->r[i].apply_log); free(c->r[i].store_buf); }
    free(c->round_commit); free(c->round_end);
    free(c);
}

void orc_cluster_record_apply(orc_cluster_t *c, int on)
{
    /* switched on in mid-run (a long replay that keeps the upcalls of its last stretch only): the record starts here */
    if (on && !c->record_apply) for (int i = 0; i < ORC_MAX_SERVERS; i++) c->r[i].apply_rec_base = c->r[i].apply_count;
    c->record_apply = on;
}
void orc_cluster_record_store(orc_cluster_t *c, int on) { c->record_store = on; }
const uint8_t *orc_replica_store_stream(const orc_cluster_t *c, int r, uint64_t *n) { *n = c->r[r].store_len; return c->r[r].store_buf; }
uint32_t orc_replica_records_len(const orc_cluster_t *c, int r) { return c->r[r].records_len; }
void orc_cluster_allow_exact_fit(orc_cluster_t *c, int on) { c->allow_exact_fit = on; }
void orc_cluster_completion_delay(orc_cluster_t *c, int on) { c->completion_delay = on; }
uint64_t orc_force_prune_count(const orc_cluster_t *c) { return c->force_prunes; }
int orc_leader(const orc_cluster_t *c) { return c->leader; }
int orc_group_size(const orc_cluster_t *c) { return c->n; }
orc_log_t *orc_replica_log(orc_cluster_t *c, int r) { return c->r[r].log; }
uint64_t orc_replica_sid(const orc_cluster_t *c, int r) { return c->r[r].sid; }
uint32_t orc_replica_cid_bitmask(const orc_cluster_t *c, int r) { return c->r[r].cid.bitmask; }
void orc_replica_cid(const orc_cluster_t *c, int r, orc_cid_t *out) { *out = c->r[r].cid; }
int orc_replica_alive(const orc_cluster_t *c, int r) { return c->r[r].alive; }
uint64_t orc_replica_highest_rec(const orc_cluster_t *c, int r) { return c->r[r].highest_rec; }
uint64_t orc_replica_apply_count(const orc_cluster_t *c, int r) { return c->r[r].apply_count; }
uint64_t orc_replica_apply_hash(const orc_cluster_t *c, int r) { return c->r[r].apply_hash; }
uint64_t orc_replica_store_count(const orc_cluster_t *c, int r) { return c->r[r].store_count; }
const orc_apply_t *orc_replica_apply_log(const orc_cluster_t *c, int r, uint64_t *n)
{ *n = c->record_apply ? c->r[r].apply_count - c->r[r].apply_rec_base : 0; return c->r[r].apply_log; }
uint64_t orc_round_count(const orc_cluster_t *c) { return c->n_rounds; }
const uint64_t *orc_round_commit(const orc_cluster_t *c) { return c->round_commit; }
const uint64_t *orc_round_end(const orc_cluster_t *c) { return c->round_end; }

/* --- proxy_store_cmd = stablestorage_save_request, src/proxy/proxy.c:268-291 + store_record,
 *     src/db/db-interface.c:65-96.  `data` = &entry->clt_id (dare_server.c:1802) is read as a
 *     proxy_msg_header {u16 connection_id; u8 action} (proxy.h:57-60); a SEND record is
 *     PROXY_SEND_MSG_SIZE = sizeof(proxy_send_msg) + data.cmd.len long (proxy.h:83-91), where
 *     sizeof(proxy_send_msg) = 24 and the overlay puts data.cmd.len at +8 of the record = entry bytes
 *     32..33 = reply[4], reply[5] -- NOT the entry's cmd.len at 48: the record carries
 *     clt_id, type, sender, reply[] and the struct padding, no command byte, and grows by 1 / 256 / 257 bytes
 *     when server 4 / 5 had already acknowledged.  CONNECT / CLOSE records: 4 bytes.  Other types: nothing. */
#define STORE_CONNECT_BYTES 4u      /* sizeof(proxy_connect_msg), proxy.h:63-66 */
#define STORE_SEND_BYTES    24u     /* sizeof(proxy_send_msg), proxy.h:83-90 (SURVEY.md 9-Q1 [probed]) */
static void store_cmd(orc_cluster_t *c, replica_t *p, const orc_entry_t *e)
{
    const uint8_t *d = (const uint8_t *)&e->clt_id;
    uint32_t n = 0;
    switch (e->type) {
    case ORC_CONNECT: case ORC_CLOSE: n = STORE_CONNECT_BYTES; break;
    case ORC_SEND: n = STORE_SEND_BYTES + (uint32_t)(d[8] | (d[9] << 8)); break;
    default: return;
    }
    p->records_len += n;
    if (!c->record_store) return;
    if (p->store_len + n > p->store_cap) {
        p->store_cap = (p->store_len + n) * 2 + 4096;
        p->store_buf = realloc(p->store_buf, p->store_cap);
    }
    memcpy(p->store_buf + p->store_len, d, n);
    p->store_len += n;
}

/* --- persist_new_entries, dare_server.c:1792-1810 ------------------- */
/* the follower ACK inside it is rc_send_entries_reply, dare_ibv_rc.c:1828-1863:
 * reply[my_idx]=1 locally and, by a 1-byte WRITE, at the same offset of the
 * log of entry->sender */
static void persist_new_entries(orc_cluster_t *c, replica_t *p)
{
    orc_log_t *log = p->log;
    uint64_t guard = 0;
    while (orc_log_is_larger(log, log->end, log->old_end)) {
        /* four laps of the smallest entries and still not at `end`: the walk of a joiner that never gets
         * there (see orc_join, -8) -- the reference spins forever at this point */
        if (++guard > log->len / 16 + 1024) { c->hung = 1; return; }
        orc_entry_t *e = get_entry(log, &log->old_end);
        if (!fits_entry(log, log->old_end, e)) { log->old_end = 0; continue; }
        p->store_count++;                                   /* proxy_store_cmd(&entry->clt_id) */
        store_cmd(c, p, e);
        if (is_leader_r(p)) {
            e->sender = p->idx;
        } else {
            e->reply[p->idx] = 1;
            replica_t *dst = &c->r[e->sender];
            if (e->sender < c->n && dst->alive)              /* ep->rc_connected */
                entry_at(dst->log, log->old_end)->reply[p->idx] = 1;
        }
        log->old_end += entry_len(e);
    }
}

/* --- update_cid, dare_server.c:2192-2227 (connection side effects dropped) */
static int cid_equal(const orc_cid_t *a, const orc_cid_t *b)
{
    return a->epoch == b->epoch && a->state == b->state && a->size[0] == b->size[0] &&
           a->size[1] == b->size[1] && a->bitmask == b->bitmask;
}
static int update_cid(replica_t *p, const orc_cid_t *cid)
{
    if (cid_equal(&p->cid, cid)) return 1;
    p->cid = *cid;
    return 0;
}

/* --- poll_config_entries, dare_server.c:2133-2187 ------------------- */
static void poll_config_entries(replica_t *p)
{
    orc_log_t *log = p->log;
    uint64_t head_offset = log->head, off = p->cid_offset, commit = log->commit;
    while (orc_log_end_distance(log, off)) {
        orc_entry_t *e = get_entry(log, &off);
        if (!fits_entry(log, off, e)) { off = 0; continue; }
        if (e->type == ORC_CONFIG) {
            if (e->idx > p->cid_idx) { const orc_cid_t cid = e->data.cid; update_cid(p, &cid); }    /* (a copy: the entry is not aligned) */
        } else if (e->type == ORC_HEAD) {
            if (!orc_log_is_larger(log, off, commit)) {     /* committed HEAD entries only */
                head_offset = e->data.head;
                p->snapshot_on = 0;                         /* dare_state &= ~SNAPSHOT, :2171 */
            }
        }
        off += entry_len(e);
    }
    p->cid_offset = orc_log_is_larger(log, off, commit) ? commit : off;
    if (orc_log_is_larger(log, head_offset, log->head)) log->head = head_offset;
}

/* ud_send_clt_reply(lid, req_id, CONFIG), dare_ibv_ud.c:1451-1490: the answer to a JOIN request carries
 * the configuration as it is NOW, the index of the CONFIG entry that admitted the server and the
 * leader's head offset */
static void join_reply(orc_cluster_t *c, replica_t *L)
{
    if (c->join_slot < 0) return;               /* nobody is waiting (a new leader re-applying the entry) */
    c->join_head = L->log->head;
    c->join_cid = L->cid;
    c->join_replied = 1;
}

/* --- apply_committed_entries, dare_server.c:1815-1974 --------------- */
/* incl. the leader's CONFIG branches (:1858-1937): reply to the joiner, the 3-phase resize
 * EXTENDED -> TRANSIT -> STABLE, each phase a new CONFIG entry appended while the previous one is applied */
static void apply_committed_entries(orc_cluster_t *c, replica_t *p)
{
    orc_log_t *log = p->log;
    int leader = is_leader_r(p);
    while (orc_log_is_larger(log, log->commit, log->apply)) {
        orc_entry_t *e = get_entry(log, &log->apply);
        if (!fits_entry(log, log->apply, e)) { log->apply = 0; continue; }
        if (p->resync_armed && log->apply == p->resync_off) { p->apply_slot = p->resync_slot; p->resync_armed = 0; }
        if (leader && e->type == ORC_CONFIG) {
            uint64_t req_id = e->req_id;
            uint16_t clt_id = e->clt_id;
            if (e->data.cid.state == CID_STABLE) {                        /* :1859-1874 */
                if (req_id != 0) join_reply(c, p);
            } else if (p->cid.epoch > e->data.cid.epoch) {                 /* :1875-1879 */
                /* from a previous configuration: ignored */
            } else {
                if (e->data.cid.state == CID_EXTENDED) {                   /* :1886-1900 */
                    p->cid.state = CID_TRANSIT;
                    if (req_id != 0) { join_reply(c, p); req_id = 0; clt_id = 0; }
                } else if (e->data.cid.state == CID_TRANSIT) {             /* :1901-1927 */
                    p->cid.state = CID_STABLE;
                    for (int i = p->cid.size[1]; i < p->cid.size[0]; i++)  /* servers a down-size removes */
                        p->cid.bitmask &= ~(1u << i);
                    p->cid.size[0] = p->cid.size[1];
                    p->cid.size[1] = 0;
                }
                orc_log_append(log, SID_TERM(p->sid), req_id, clt_id, ORC_CONFIG, &p->cid, 0);   /* :1932 */
            }
            log->apply += entry_len(e);
            p->apply_slot++;
            continue;
        }
        int client = (e->type != ORC_CONFIG && e->type != ORC_NOOP && e->type != ORC_HEAD);
        if (client) {
            if (leader) { p->highest_rec++; record_apply(c, p, log->apply, e, 1); }
            else        { record_apply(c, p, log->apply, e, 2); }
            p->last_applied.idx = e->idx;
            p->last_applied.term = e->term;
            p->last_applied.offset = log->apply + entry_len(e);
        }
        log->apply += entry_len(e);
        p->apply_slot++;
    }
}

/* one follower polling() pass, dare_server.c:1012-1125 (non-leader branch) */
static void follower_poll(orc_cluster_t *c, replica_t *p)
{
    if (!p->alive) return;
    persist_new_entries(c, p);
    poll_config_entries(p);
    apply_committed_entries(c, p);
}

/* --- handle_lr_work_completion, dare_ibv_rc.c:3126-3196 (success path) */
static void complete_one(replica_t *L, int i)
{
    switch (L->pending[i]) {
    case PEND_LOG: L->lr_step[i] = LR_UPDATE_END; L->send_flag[i] = 1; break;
    case PEND_END: L->lr_step[i] = LR_UPDATE_LOG; L->send_flag[i] = 1; break;
    case PEND_ADJ: L->lr_step[i]++;               L->send_flag[i] = 1; break;
    default: break;
    }
    L->pending[i] = PEND_NONE;
}

static void complete_pending(orc_cluster_t *c, replica_t *L)
{
    for (int i = 0; i < c->n; i++) complete_one(L, i);
}

/* Completion timing.  post_send drains the LOG CQ right behind every post
 * (empty_completion_queue, dare_ibv_rc.c:2590).  Default schedule (the one the
 * reference-as-is harness oracle/refshim/fabric.c produces, and the one the loops are
 * pinned on): the completion of the WR just posted is already there, so the step machine
 * advances at once.  completion_delay = 1 gives the other legal schedule (the completion
 * is seen by the poll at the top of the next loop pass, :1890). */
static inline void posted(orc_cluster_t *c, replica_t *L, int i)
{
    if (!c->completion_delay) complete_one(L, i);
}

static inline int peer_reachable(const orc_cluster_t *c, const replica_t *L, int i)
{
    /* CID_IS_SERVER_ON && fail_count < PERMANENT_FAILURE && rc_connected;
     * a held peer models a link the leader cannot currently post to */
    return i != L->idx && cid_on(&L->cid, i) && c->r[i].alive && !c->r[i].held;
}

/* --- log_adjustment, dare_ibv_rc.c:1292-1451 ------------------------ */
static void follower_poll(orc_cluster_t *c, replica_t *p);
static void log_adjustment(orc_cluster_t *c, replica_t *L)
{
    orc_log_t *log = L->log;
    for (int i = 0; i < c->n; i++) {
        if (!peer_reachable(c, L, i) || !L->send_flag[i]) continue;
        uint64_t remote_commit = L->vote_ack[i];
        if (remote_commit == log->len) continue;             /* no vote ACK from this server */
        replica_t *F = &c->r[i];
        switch (L->lr_step[i]) {
        case LR_GET_WRITE:
            L->rem_commit[i] = remote_commit;
            L->lr_step[i] = LR_GET_NCE_LEN;
            __attribute__((fallthrough));    /* no break in the reference either, :1357 */
        case LR_GET_NCE_LEN:
            if (orc_log_is_larger(log, remote_commit, log->commit)) log->commit = remote_commit;
            log->nc_buf[i].len = F->log->nc_buf[i].len;       /* READ 8 bytes */
            break;
        case LR_GET_NCE:
            if (log->nc_buf[i].len == 0) {
                L->rem_end[i] = L->rem_commit[i];
                L->lr_step[i] = LR_UPDATE_LOG;
                continue;
            }
            memcpy(log->nc_buf[i].entries, F->log->nc_buf[i].entries,
                   log->nc_buf[i].len * sizeof(orc_det_t)); /* READ len*24 bytes */
            break;
        case LR_SET_END:
            L->rem_end[i] = orc_log_find_remote_end(log, &log->nc_buf[i]);
            F->log->end = L->rem_end[i];                      /* WRITE 8 bytes */
            /* the follower's next polling() pass sees an end that lies BEHIND its old_end:
             * persist_new_entries (dare_server.c:1793-1810) then walks from old_end forward -- over what
             * is left of its old entries, the untouched rest of the ring, and around through 0 -- until it
             * arrives at the new end, "storing" and ACKing everything on the way.  Harmless garbage in
             * the reference (the bytes are overwritten or lie outside the log), but it moves old_end to
             * the right place and counts as store upcalls: reproduced (pinned on the reference). */
            follower_poll(c, F);
            break;
        default:
            continue;
        }
        L->send_flag[i] = 0;
        L->pending[i] = PEND_ADJ;
        posted(c, L, i);
    }
}

static void ring_write(orc_log_t *dst, const orc_log_t *src, uint64_t from, uint64_t to)
{
    memcpy(dst->entries + from, src->entries + from, to - from);
}

/* --- update_remote_logs, dare_ibv_rc.c:1465-1826 -------------------- */
static void update_remote_logs(orc_cluster_t *c, replica_t *L)
{
    orc_log_t *log = L->log;
    int size = ext_group_size(&L->cid);                        /* :1489 */

    for (int i = 0; i < size; i++) {
        if (!peer_reachable(c, L, i) || !L->send_flag[i]) continue;
        replica_t *F = &c->r[i];
        if (L->lr_step[i] == LR_UPDATE_LOG) {                 /* :1507-1547 */
            uint64_t rend = L->rem_end[i];
            if (orc_log_end_distance(log, rend) == 0) continue;
            L->cached_end[i] = log->end;
            if (log->end > rend) {
                ring_write(F->log, log, rend, log->end);
            } else {                                          /* wrap: two WRs, :1538-1545 */
                ring_write(F->log, log, rend, log->len);
                ring_write(F->log, log, 0, log->end);
            }
            L->pending[i] = PEND_LOG;
        } else if (L->lr_step[i] == LR_UPDATE_END) {          /* :1549-1573 */
            L->rem_end[i] = L->cached_end[i];
            F->log->end = L->rem_end[i];
            L->pending[i] = PEND_END;
            follower_poll(c, F);                              /* doorbell seen: persist + ACK */
        } else {
            continue;
        }
        L->send_flag[i] = 0;
        posted(c, L, i);
    }

    /* commit scan over the ACK bytes, :1725-1758 (the offset-median code above
     * it is dead: its result is overwritten at :1725 -- but it leaves `size` behind) */
    size = scan_size(&L->cid);
    uint64_t min_offset = log->commit;
    while (orc_log_end_distance(log, min_offset)) {
        orc_entry_t *e = get_entry(log, &min_offset);
        if (!fits_entry(log, min_offset, e)) { min_offset = 0; continue; }
        int replies = 0;
        for (int i = 0; i < size; i++)
            if (i == L->idx || e->reply[i] == 1) replies++;
        if (replies < size / 2 + 1) break;
        min_offset += entry_len(e);
    }
    if (orc_log_is_larger(log, min_offset, log->commit)) {
        log->commit = min_offset;
        L->cid_offset = log->commit;
        c->committed_flag = 1;
    }

    /* lazy commit propagation, :1761-1819 */
    for (int i = 0; i < size; i++) {
        if (!peer_reachable(c, L, i) || L->lr_step[i] != LR_UPDATE_LOG) continue;
        uint64_t *rc = &L->rem_commit[i], *re = &L->rem_end[i];
        if (*rc == *re || *rc == log->commit) continue;
        *rc = log->commit;
        if (orc_log_is_larger(log, *rc, *re)) *rc = *re;
        replica_t *F = &c->r[i];
        F->log->commit = *rc;                                 /* WRITE 8 bytes */
        follower_poll(c, F);
    }
}

/* --- rc_write_remote_logs, dare_ibv_rc.c:1870-1948 ------------------ */
static void write_remote_logs(orc_cluster_t *c, replica_t *L, int wait_for_commit)
{
    int threshold = 0;
    if (wait_for_commit) c->committed_flag = 0;
    for (;;) {
        complete_pending(c, L);          /* empty_completion_queue(LOG_QP) */
        threshold++;
        log_adjustment(c, L);
        update_remote_logs(c, L);
        if (wait_for_commit && c->committed_flag) return;
        if (threshold == 1000) return;
        if (!wait_for_commit) return;
    }
}

/* --- commit_new_entries, dare_server.c:1751-1789 -------------------- */
static void commit_new_entries(orc_cluster_t *c, replica_t *L)
{
    orc_log_t *log = L->log;
    if (orc_log_end_distance(log, log->commit)) {
        write_remote_logs(c, L, 1);
    } else if (!log_empty(log)) {
        for (int i = 0; i < group_size(&L->cid); i++) {          /* :1766 */
            if (!peer_reachable(c, L, i)) continue;
            if (L->vote_ack[i] == log->len) continue;
            if (L->lr_step[i] != LR_UPDATE_LOG || L->rem_end[i] != log->end) {
                write_remote_logs(c, L, 0);
                break;
            }
        }
    }
}

static int  log_pruning(orc_cluster_t *c, replica_t *L);
static void force_log_pruning(orc_cluster_t *c, replica_t *L);

/* leader polling() pass after the tailq was drained, dare_server.c:1095-1124 */
static void leader_poll(orc_cluster_t *c, replica_t *L)
{
    /* check_failure_count opens the pass (dare_server.c:1189-1230): a server that counts no more than half of
     * the group as connected -- ON in its configuration and not at PERMANENT_FAILURE -- shuts itself down
     * ("Not enough connections... bye bye", :1212-1216).  Removals shrink the bitmask, never cid.size[0]: a
     * leader that has removed (or evicted, force_log_pruning) half of its group leaves; the group is gone. */
    {
        int size = group_size(&L->cid), on = 0;
        for (int i = 0; i < size; i++) on += cid_on(&L->cid, i);
        if (on <= size / 2) { c->leader_quit = 1; return; }
    }
    persist_new_entries(c, L);
    commit_new_entries(c, L);
    apply_committed_entries(c, L);
    force_log_pruning(c, L);                 /* :1121-1124 */
}

static void note_round(orc_cluster_t *c, replica_t *L)
{
    if (c->n_rounds == c->rounds_cap) {
        c->rounds_cap = c->rounds_cap ? c->rounds_cap * 2 : 1024;
        c->round_commit = realloc(c->round_commit, c->rounds_cap * sizeof(uint64_t));
        c->round_end = realloc(c->round_end, c->rounds_cap * sizeof(uint64_t));
    }
    c->round_commit[c->n_rounds] = L->log->commit;
    c->round_end[c->n_rounds] = L->log->end;
    c->n_rounds++;
}

int orc_round(orc_cluster_t *c, const orc_req_t *reqs, int n, const uint8_t *arena)
{
    if (c->leader < 0) return -1;
    replica_t *L = &c->r[c->leader];
    /* get_tailq_message, dare_ibv_ud.c:780-790 */
    for (int k = 0; k < n; k++) {
        uint64_t idx = orc_log_append(L->log, SID_TERM(L->sid), reqs[k].req_id, reqs[k].clt_id,
                                      reqs[k].type, arena ? arena + reqs[k].payload_off : NULL,
                                      reqs[k].len);
        if (idx == 0) return -2;        /* log full: the reference drops the request (Q6) */
        if (L->log->end == L->log->len && !c->allow_exact_fit) return -3;   /* exact-fit wrap, SURVEY.md Q13 */
    }
    leader_poll(c, L);
    if (c->leader_quit) { L->alive = 0; c->leader = -1; return -9; }
    note_round(c, L);
    return 0;
}

int orc_quiesce(orc_cluster_t *c)
{
    if (c->leader < 0) return -1;
    replica_t *L = &c->r[c->leader];
    /* poll until nothing moves any more */
    for (int it = 0; it < 64; it++) {
        uint64_t before[ORC_MAX_SERVERS][4];
        for (int i = 0; i < c->n; i++) {
            before[i][0] = c->r[i].log->end; before[i][1] = c->r[i].log->commit;
            before[i][2] = c->r[i].log->apply; before[i][3] = c->r[i].log->old_end;
        }
        uint8_t steps[ORC_MAX_SERVERS];
        memcpy(steps, L->lr_step, sizeof steps);
        leader_poll(c, L);
        for (int i = 0; i < c->n; i++) if (i != c->leader) follower_poll(c, &c->r[i]);
        int moved = memcmp(steps, L->lr_step, sizeof steps) != 0;
        for (int i = 0; i < c->n; i++) {
            moved |= before[i][0] != c->r[i].log->end || before[i][1] != c->r[i].log->commit ||
                     before[i][2] != c->r[i].log->apply || before[i][3] != c->r[i].log->old_end;
        }
        for (int i = 0; i < c->n; i++) moved |= L->pending[i] != PEND_NONE;
        if (c->leader_quit) { L->alive = 0; c->leader = -1; return -9; }
        if (!moved) return 0;
    }
    return 1;
}

/* --- log_pruning, dare_server.c:1996-2067 + rc_get_remote_apply_offsets,
 *     dare_ibv_rc.c:1970-2034.  Returns 1 when a <HEAD> entry was appended, -2 when the log is full */
static int log_pruning(orc_cluster_t *c, replica_t *L)
{
    orc_log_t *log = L->log;
    int size = ext_group_size(&L->cid);                        /* :2026 */
    uint64_t min_offset = log->apply;
    for (int i = 0; i < size; i++) {
        if (!cid_on(&L->cid, i)) L->apply_offsets[i] = log->apply;
        if (orc_log_is_larger(log, min_offset, L->apply_offsets[i])) min_offset = L->apply_offsets[i];
    }
    if (!orc_log_end_distance(log, min_offset)) min_offset = orc_log_get_tail(log);
    int appended = 0;
    if (orc_log_is_larger(log, min_offset, log->head) && !log->prev_head) {
        log->head = min_offset;
        uint64_t idx = orc_log_append(log, SID_TERM(L->sid), 0, 0, ORC_HEAD, &log->head, 0);
        if (idx == 0) return -2;
        log->prev_head = 1;
        appended = 1;
    }
    /* READ every reachable peer's apply offset for the next tick */
    for (int i = 0; i < size; i++) {
        if (i == L->idx || !cid_on(&L->cid, i)) { L->apply_offsets[i] = log->apply; continue; }
        if (!c->r[i].alive || c->r[i].held) continue;
        if (L->vote_ack[i] == log->len) continue;
        L->apply_offsets[i] = c->r[i].log->apply;
    }
    return appended;
}

/* --- force_log_pruning, dare_server.c:2069-2122: closes every leader pass.  At 75 % fill the
 * server whose sampled apply offset holds the head back is REMOVED from the configuration
 * (CONFIG entry), then the log is pruned.  Pinned on the reference, incl. its slip at :2113
 * (`apply_offsets[i]` with i == size after the loop, not `target`). */
static void force_log_pruning(orc_cluster_t *c, replica_t *L)
{
    orc_log_t *log = L->log;
    uint64_t log_size = orc_log_end_distance(log, log->head);
    if ((double)log_size < 0.75 * (double)log->len) return;
    c->force_prunes++;
    int size = ext_group_size(&L->cid), target = L->idx, i;    /* :2082 */
    uint64_t min_offset = log->apply;
    for (i = 0; i < size; i++)
        if (orc_log_is_larger(log, min_offset, L->apply_offsets[i])) { min_offset = L->apply_offsets[i]; target = i; }
    if (target != L->idx) {
        if (!cid_on(&L->cid, target)) { log_pruning(c, L); return; }
        L->cid.bitmask &= ~(1u << target);                  /* CID_SERVER_RM + dare_ib_disconnect_server */
        orc_log_append(log, SID_TERM(L->sid), 0, 0, ORC_CONFIG, &L->cid, 0);
        if (i < ORC_MAX_SERVERS) L->apply_offsets[i] = log->apply;      /* :2113, i == size */
    }
    log_pruning(c, L);
}

int orc_tick_prune(orc_cluster_t *c)
{
    if (c->leader < 0) return -1;
    replica_t *L = &c->r[c->leader];
    /* Trace semantics: the prune timer fires between polling() passes once every
     * follower has caught up (ms-scale timer vs us-scale rounds), so the apply
     * offsets sampled by R8 do not depend on the lazy commit lag. */
    if (orc_quiesce(c) == -9) return -9;
    int appended = log_pruning(c, L);
    if (appended < 0) return appended;
    if (appended) {
        leader_poll(c, L);
        if (c->leader_quit) { L->alive = 0; c->leader = -1; return -9; }
        note_round(c, L);
    }
    return appended;
}

int orc_kill(orc_cluster_t *c, int r)
{
    if (r < 0 || r >= c->n) return -1;
    c->r[r].alive = 0;
    if (c->leader == r) c->leader = -1;
    else if (c->leader >= 0) {
        /* check_failure_count, dare_server.c:1189-1230: the leader drops the peer
         * from the bitmask and logs a CONFIG entry */
        replica_t *L = &c->r[c->leader];
        if (cid_on(&L->cid, r)) {
            L->cid.bitmask &= ~(1u << r);
            uint64_t idx = orc_log_append(L->log, SID_TERM(L->sid), 0, 0, ORC_CONFIG, &L->cid, 0);
            if (idx == 0) return -2;
            leader_poll(c, L);
            if (c->leader_quit) { L->alive = 0; c->leader = -1; return -9; }
            note_round(c, L);
        }
    }
    return 0;
}

int orc_hold(orc_cluster_t *c, int r)    { if (r < 0 || r >= c->n) return -1; c->r[r].held = 1; return 0; }
int orc_release(orc_cluster_t *c, int r) { if (r < 0 || r >= c->n) return -1; c->r[r].held = 0; return 0; }

/* --- JOIN(r): a new server joins -----------------------------------------------
 * Leader: handle_server_join_request dare_ibv_ud.c:973-1068; joiner: handle_server_join_reply :1071-1088,
 * get_replicated_vote_cb dare_server.c:524, poll_sm_requests :599 (on the followers) / poll_sm_reply :658 /
 * rc_recover_sm dare_ibv_rc.c:597, rc_recover_log :726-866, recover_log_cb dare_server.c:710 ->
 * server_to_follower :2238 (vote ACK), then the leader's log_adjustment.  Schedule = the one
 * oracle/refshim/refcluster.c:refc_join drives the reference through: the joiner's timer fires once per
 * sweep, every sweep is four polling passes of everybody (leader, the others, the joiner).
 * The joiner is a NEW machine: a fresh log, fresh upcall counters, LID = number of machines so far.
 * Returns 0; -1 no leader / resize in progress (Case 1: the request is ignored, the joiner retries);
 * -4 the leader would hand out another slot than r; -5 a follower is asked for its state machine a second
 * time before a <HEAD> entry was committed (the reference answers from an uninitialised pointer there,
 * dare_server.c:604-651: `snapshot` is only set when SNAPSHOT is clear); -6 no follower to recover from;
 * -7 the CONFIG entry does not commit (no quorum); -8 the joiner's first persist_new_entries pass never
 * ends: old_end starts at len (log_new, dare_log.h:134; nothing in the recovery path sets it), so the
 * pass walks from offset 0 through the zeroed part of the joiner's ring in 64-byte "NOOP" steps and on
 * through the recovered entries, and only stops when it lands EXACTLY on `end` -- it does when head == 0
 * or when head and every entry length are multiples of 64, otherwise it runs into the entries
 * misaligned and cycles (found by running the reference; the joiner hangs in polling()).  Every entry
 * the pass meets is "stored" and ACKed into the log of the server its `sender` byte names. */
static void replica_fresh(orc_cluster_t *c, int r, uint16_t lid)
{
    replica_t *p = &c->r[r];
    orc_log_free(p->log); free(p->apply_log); free(p->store_buf);
    memset(p, 0, sizeof *p);
    p->log = orc_log_new(c->log_len);
    p->idx = (uint8_t)r;
    p->lid = lid;
    for (int j = 0; j < ORC_MAX_SERVERS; j++) {
        p->vote_ack[j] = c->log_len;
        p->lr_step[j] = LR_GET_WRITE;
        p->send_flag[j] = 1;
    }
}

static void join_pass(orc_cluster_t *c, replica_t *L, int times)
{
    for (int t = 0; t < times; t++) {
        leader_poll(c, L);
        for (int i = 0; i < c->n; i++)
            if (i != c->leader && c->r[i].log && !c->r[i].held) follower_poll(c, &c->r[i]);
    }
}

int orc_join(orc_cluster_t *c, int r)
{
    if (c->leader < 0) return -1;
    replica_t *L = &c->r[c->leader];
    orc_log_t *log = L->log;
    if (L->cid.state != CID_STABLE) return -1;                  /* Case 1, :978-982 */
    int size = L->cid.size[0], empty = size;
    for (int i = size - 1; i >= 0; i--) if (!cid_on(&L->cid, i)) empty = i;     /* :995-1021 */
    if (empty != r || r >= ORC_MAX_SERVERS) return -4;
    int donors = 0;
    for (int i = 0; i < size; i++) {
        if (i == c->leader || i == r || !cid_on(&L->cid, i)) continue;
        /* a configured server that cannot be reached: the joiner needs RC connections to and the
         * replicated vote from a majority (rc_get_replicated_vote dare_ibv_rc.c:874) and retries until it has
         * them -- this oracle's JOIN is one event, so it covers joins into a fully reachable group */
        if (!c->r[i].alive || c->r[i].held) return -6;
        if (c->r[i].snapshot_on) return -5;
        donors++;
    }
    if (!donors) return -6;

    /* a member whose configuration still shows the slot's FORMER holder (it never took in the removal, see
     * below) keeps that server's RC endpoint as connected and does not set up a new one for the joiner */
    uint32_t stale = 0;
    for (int i = 0; i < size; i++) if (i != r && c->r[i].log && cid_on(&c->r[i].cid, r)) stale |= 1u << i;
    /* Case 3 (an empty place) or Case 4 (the group is full: extend it), :1022-1041 */
    uint16_t lid = (uint16_t)++c->next_lid;
    uint64_t end0 = log->end;
    L->cid.bitmask |= 1u << empty;
    if (empty == size) {
        L->cid.state = CID_EXTENDED;
        L->cid.size[1] = (uint8_t)(size + 1);
        L->cid.epoch++;
    }
    L->lr_step[empty] = LR_GET_WRITE; L->send_flag[empty] = 1; L->pending[empty] = PEND_NONE;   /* :1043-1051 */
    L->vote_ack[empty] = log->len;
    L->apply_offsets[empty] = log->head;
    replica_fresh(c, r, lid);                                   /* not connected yet: alive == 0 */
    if (r >= c->n) c->n = r + 1;
    c->join_slot = r; c->join_replied = 0;
    /* the request id of a machine's first request is 1 (IBDEV->request_id, dare_ibv.c:130) */
    c->join_cid_idx = orc_log_append(log, SID_TERM(L->sid), 1, lid, ORC_CONFIG, &L->cid, 0);   /* :1057-1060 */
    if (c->join_cid_idx == 0) { c->join_slot = -1; return -2; }
    join_pass(c, L, 4);                                         /* sweep 0: commit, apply, reply, resize phases */
    if (!c->join_replied) { c->join_slot = -1; return -7; }

    replica_t *J = &c->r[r];
    J->cid = c->join_cid;                                       /* handle_server_join_reply */
    J->log->head = c->join_head;
    J->cid_idx = c->join_cid_idx;
    J->cid_offset = J->log->head;
    {
        /* sweep 1: RC_SYN / SYNACK.  A member answers the joiner's RC_SYN only if ITS OWN configuration has
         * the joiner's bit ON (handle_rc_syn, dare_ibv_ud.c: "Configuration inconsistency; it will be solved
         * later") -- and a server that itself joined ignores every CONFIG entry whose idx is not above the
         * idx of the entry that admitted it (poll_config_entries dare_server.c:2152, cid_idx from the join
         * reply), i.e. ALL of them once the index sequence has restarted at an exact-fit wrap.
         * The joiner then needs the replicated vote from more than half of the group it joins (the new size
         * while the configuration is not STABLE: wait_for_majority, dare_ibv_rc.c) and retries for ever if the
         * members that answer are too few (found by running the reference).  -6: not a schedule of this oracle. */
        int jsize = J->cid.state == CID_STABLE ? J->cid.size[0] : J->cid.size[1], conn = 0;
        for (int i = 0; i < jsize; i++)
            if (i != r && cid_on(&J->cid, i) && c->r[i].alive && !c->r[i].held && cid_on(&c->r[i].cid, r) &&
                !((stale >> i) & 1u) && i < ext_group_size(&J->cid)) conn++;
        if (conn <= jsize / 2) { c->join_slot = -1; return -6; }
    }
    join_pass(c, L, 4);
    J->sid = SID_MAKE(0, 1, r);                                 /* sweep 2: get_replicated_vote_cb :529-531 */
    join_pass(c, L, 4);
    /* sweep 3: the SM request reaches every connected server; the followers (the leader does not poll for
     * it, dare_server.c:1089-1091) dump their state machine and answer; the joiner takes the first answer
     * in index order */
    leader_poll(c, L);
    int target = -1;
    for (int i = 0; i < group_size(&J->cid); i++) {
        replica_t *F = &c->r[i];
        if (i == r || i == c->leader || !cid_on(&J->cid, i) || !F->alive || F->held) continue;
        follower_poll(c, F);
        F->snapshot_last = F->last_applied.offset;              /* :637 */
        F->snapshot_on = 1;
        if (target < 0) target = i;
    }
    if (target < 0) { c->join_slot = -1; return -6; }
    J->sid = c->r[target].sid;                                  /* poll_sm_reply :671 */
    J->log->apply = c->r[target].snapshot_last;                 /* rc_recover_sm dare_ibv_rc.c:691 */
    {
        /* `slot` of the apply stream = position of the entry in the total order of the log (not a field of
         * the reference: the checker's own numbering, orc_apply_t): the joiner continues the donor's count
         * from the entry its snapshot ends with */
        orc_log_t *dl = c->r[target].log;
        uint64_t k = 0, off = c->r[target].snapshot_last;
        while (off != dl->apply && k < (1ull << 32)) {
            orc_entry_t *e = get_entry(dl, &off);
            if (!e) break;
            if (!fits_entry(dl, off, e)) { off = 0; continue; }
            off += entry_len(e); k++;
        }
        J->apply_slot = c->r[target].apply_slot - k;
        /* A joiner that arrives while the log wraps gets apply = an offset of the NEW lap but end = commit = 0
         * (rc_recover_log fetches [head, len) only), so its apply loop (commit "larger" than apply) runs from
         * there through the zeroed ring up to head and RE-APPLIES [head, len) on top of the snapshot, later
         * [0, apply) as well (found by running the reference; pinned, tests/traces.py:join_wrapped).  The walk
         * through zeroes makes no upcalls; the numbering is put right where it reaches real entries again. */
        uint64_t kh = 0;
        off = J->log->head;
        while (off != dl->apply && kh < (1ull << 32)) {
            orc_entry_t *e = get_entry(dl, &off);
            if (!e) break;
            if (!fits_entry(dl, off, e)) { off = 0; continue; }
            off += entry_len(e); kh++;
        }
        J->resync_off = J->log->head; J->resync_slot = c->r[target].apply_slot - kh; J->resync_armed = 1;
    }
    join_pass(c, L, 3);
    /* sweep 4: rc_recover_log -- commit and end of the first connected server in index order, then the
     * bytes between the head the leader named and that end (up to the ring's end if they wrap) */
    {
        int t = -1;
        for (int i = 0; i < group_size(&J->cid) && t < 0; i++)
            if (i != r && cid_on(&J->cid, i) && c->r[i].alive && !c->r[i].held) t = i;
        orc_log_t *T = c->r[t].log, *jl = J->log;
        uint64_t rend = T->end, rcommit = T->commit;
        if (rend != T->len) {
            jl->end = rend;
            if (rend > 0 && rend < jl->head) rend = 0;
            if (orc_log_is_larger(jl, rcommit, rend)) rcommit = rend;
            jl->end = rend;
            jl->commit = rcommit;
            uint64_t n = orc_log_end_distance(jl, jl->head);
            memcpy(jl->entries + jl->head, T->entries + jl->head, n);
        }
    }
    J->alive = 1;                                               /* LOG_RECOVERED, RC connected */
    orc_log_to_ncbuf(J->log, &J->log->nc_buf[J->idx]);          /* server_to_follower :2260-2262 */
    L->vote_ack[r] = J->log->commit;                            /* rc_send_vote_ack */
    join_pass(c, L, 4);
    c->join_slot = -1;
    J->resync_armed = 0;                                        /* (the offset comes round again a lap later) */
    if (c->hung) return -8;
    if (log->end != end0) note_round(c, L);
    return 0;
}

/* --- election: start_election dare_server.c:1264-1322, poll_vote_requests
 *     :1526-1743, poll_vote_count :1327-1518, vote request / ack
 *     dare_ibv_rc.c:969-1045 / :1116-1180 -------------------------------- */
static void last_entry_of(const orc_log_t *log, uint64_t *idx, uint64_t *term)
{
    *idx = 0; *term = 0;
    if (log_empty(log)) return;
    uint64_t tail = orc_log_get_tail(log);
    if (tail == log->len) return;
    orc_entry_t *e = get_entry(log, &tail);
    if (e) { *idx = e->idx; *term = e->term; }
}

static void start_election(orc_cluster_t *c, replica_t *p)
{
    p->sid = SID_MAKE(SID_TERM(p->sid) + 1, 0, p->idx);
    for (int i = 0; i < c->n; i++) {
        p->vote_ack[i] = p->log->len;
        p->lr_step[i] = LR_GET_WRITE;
        p->send_flag[i] = 1;
        p->pending[i] = PEND_NONE;
    }
}

/* returns 1 when voter v grants its vote to candidate w */
static int vote_for(orc_cluster_t *c, replica_t *v, replica_t *w,
                    uint64_t req_sid, uint64_t req_idx, uint64_t req_term)
{
    if (SID_L(v->sid)) return 0;                          /* :1536-1541 */
    uint64_t old_sid = v->sid | (1ull << 8);              /* :1560 */
    if (old_sid >= req_sid) return 0;
    /* exclusive access to the local log; not-committed buffer, :1603-1626 */
    orc_ncbuf_t *nc = &v->log->nc_buf[v->idx];
    orc_log_to_ncbuf(v->log, nc);
    uint64_t my_idx, my_term;
    if (nc->len == 0) last_entry_of(v->log, &my_idx, &my_term);
    else { my_idx = nc->entries[nc->len - 1].idx; my_term = nc->entries[nc->len - 1].term; }
    if (my_term > req_term || (my_term == req_term && my_idx > req_idx)) {
        /* candidate's log is not good enough: raise own term, no vote, :1661-1673 */
        v->sid = SID_MAKE(SID_TERM(req_sid), 0, v->idx);
        return 0;
    }
    v->sid = req_sid;                                     /* :1690 */
    update_cid(v, &w->cid);                               /* :1697 */
    w->vote_ack[v->idx] = v->log->commit;                 /* rc_send_vote_ack */
    (void)c;
    return 1;
}

int orc_elect(orc_cluster_t *c, int winner)
{
    if (winner < 0 || winner >= c->n || !c->r[winner].alive) return -1;
    replica_t *w = &c->r[winner];
    if (c->leader >= 0 && c->leader != winner) {
        /* a live leader steps down only when it sees a higher term; the trace
         * must KILL it first */
        return -1;
    }
    /* Step 1: every live server misses the leader's heartbeat (or, at start-up,
     * reaches RC_ESTABLISHED, dare_server.c:1169) and becomes a candidate of
     * term t+1; same-term requests are mutually dropped (:1568). */
    /* (a held server is cut off and, in this schedule, does not time out on its own: it keeps its old
     * SID until it hears from the new leader -- the schedule tests/test_oracle_vs_refloops.py pins) */
    for (int i = 0; i < c->n; i++) if (c->r[i].alive && (!c->r[i].held || i == winner)) start_election(c, &c->r[i]);
    /* Step 2: the winner's election timeout fires first: term t+2 */
    start_election(c, w);
    uint64_t req_idx, req_term;
    last_entry_of(w->log, &req_idx, &req_term);            /* rc_send_vote_request */
    uint64_t req_sid = w->sid;
    int votes = 1;
    for (int i = 0; i < c->n; i++) {
        if (i == winner || !c->r[i].alive || c->r[i].held) continue;
        if (!cid_on(&w->cid, i)) continue;
        votes += vote_for(c, &c->r[i], w, req_sid, req_idx, req_term);
    }
    /* poll_vote_count, :1327-1518 */
    for (int i = 0; i < c->n; i++) {
        if (i == winner) continue;
        uint64_t rc = w->vote_ack[i];
        if (rc == w->log->len) continue;
        w->rem_commit[i] = rc;
        w->lr_step[i] = LR_GET_NCE_LEN;
        if (orc_log_is_larger(w->log, rc, w->log->commit)) w->log->commit = rc;
    }
    if (votes < w->cid.size[0] / 2 + 1) return -1;
    w->sid |= (1ull << 8);
    c->leader = winner;
    poll_config_entries(w);
    apply_committed_entries(c, w);
    /* blank CONFIG entry, :1411-1421 */
    if (orc_log_append(w->log, SID_TERM(w->sid), 0, 0, ORC_CONFIG, &w->cid, 0) == 0) return -2;
    for (int i = 0; i < w->cid.size[0]; i++) w->apply_offsets[i] = w->log->head;   /* :1504-1507 */
    /* heartbeats reach everybody: hb_receive_cb :822-920 resets tail and adopts
     * the leader's SID; servers that did not vote run server_to_follower :2238 */
    for (int i = 0; i < c->n; i++) {
        replica_t *p = &c->r[i];
        if (i == winner || !p->alive || p->held) continue;
        p->log->tail = p->log->len;
        /* hb_receive_cb :903-910: server_to_follower (restore the log access for the new leader,
         * send the vote ACK with the not-committed buffer) only when the HB carries a NEW term.  A
         * server that refused its vote raised its term to the candidate's (:1661-1673): same term,
         * so it adopts the SID and nothing else -- the leader never gets a vote ACK from it and
         * leaves its log alone until the next election (pinned on the reference). */
        if (SID_TERM(p->sid) != SID_TERM(w->sid)) {
            orc_log_to_ncbuf(p->log, &p->log->nc_buf[p->idx]);
            w->vote_ack[i] = p->log->commit;
        }
        p->sid = w->sid;
    }
    /* check_failure_count (dare_server.c:1189-1230) opens the new leader's first pass: a server
     * that is ON in the configuration but dead has already failed two CTRL writes (the vote
     * requests of the two start_election calls above, dare_ibv_rc.c:2747), i.e. it is at
     * PERMANENT_FAILURE, and is removed with a CONFIG entry BEFORE persist/commit run -- so
     * the blank entry and the removal commit in one pass (pinned on the reference itself,
     * tests/test_oracle_vs_refloops.py). */
    {
        uint32_t dead = 0;
        for (int i = 0; i < c->n; i++)
            if (i != winner && cid_on(&w->cid, i) && (!c->r[i].alive || c->r[i].held)) dead |= 1u << i;   /* dead, or cut off: both vote requests failed */
        if (dead) {
            w->cid.bitmask &= ~dead;
            if (orc_log_append(w->log, SID_TERM(w->sid), 0, 0, ORC_CONFIG, &w->cid, 0) == 0) return -2;
        }
    }
    leader_poll(c, w);
    note_round(c, w);
    return 0;
}

int orc_run_rounds(orc_cluster_t *c, const orc_req_t *reqs, const uint32_t *round_n,
                   uint64_t n_rounds, const uint8_t *arena, uint64_t prune_bytes)
{
    uint64_t since = 0, g = 0;
    for (uint64_t r = 0; r < n_rounds; r++) {
        uint32_t n = round_n[r];
        int rc = orc_round(c, reqs + g, (int)n, arena);
        if (rc) return rc;
        for (uint32_t k = 0; k < n; k++) {
            uint8_t t = reqs[g + k].type;
            since += ORC_HDR_BYTES + ((t == ORC_NOOP || t == ORC_CONFIG || t == ORC_HEAD) ? 0 : reqs[g + k].len);
        }
        g += n;
        if (prune_bytes && since >= prune_bytes) {
            rc = orc_tick_prune(c);
            if (rc < 0) return rc;
            since = 0;
        }
    }
    return 0;
}


/* ================================================================== */
/* Part 3: the same steady-state loops on N threads (CPU baseline only)  */
/*
 * "the CPU oracle compiled -O2, N replica threads pinned 1/core, memcpy RDMA".
 * One thread per server.  The leader thread does what its NIC would do as well (the log WRITE into every
 * follower's ring, the end and commit doorbells); a follower thread polls its own end / commit words,
 * persists + ACKs (the reply byte in its own ring and in the leader's), applies.  The leader waits for the
 * ACK majority of a round before it takes the next one (rc_write_remote_logs with wait_for_commit).
 * Steady state only -- no failures, the prune tick quiesces like orc_tick_prune.  At the end every ring,
 * offset and upcall count equals the single-threaded run's (tests/test_trace_oracle.py); the per-pass
 * record is not kept (passes are not a notion here).  Every spin checks a deadline: the function returns
 * -10 instead of hanging.
 */
#define _MT_RELAX() __asm__ __volatile__("pause" ::: "memory")
#include <pthread.h>
#include <sched.h>
#include <time.h>

typedef struct { orc_cluster_t *c; int idx; int stop; int failed; uint64_t deadline_ns; } mt_ctx_t;

static uint64_t mt_now(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

/* the entry that starts at *off in a log whose bytes up to `lim` are valid (reader rule of log_get_entry /
 * log_fit_entry, dare_log.h:316-331,241); returns NULL at lim */
static orc_entry_t *mt_next(orc_log_t *log, uint64_t *off, uint64_t lim)
{
    for (;;) {
        if (*off == lim) return NULL;
        if (log->len - *off < ORC_HDR_BYTES) { *off = 0; continue; }
        orc_entry_t *e = entry_at(log, *off);
        if (log->len - *off < entry_len(e)) { *off = 0; continue; }
        return e;
    }
}

/* Under ThreadSanitizer this port still reports races of two kinds, both the reference's own and on purpose: the circular comparison
 * of a HEAD entry's offset reads the live `end` word its leader's thread stores (dare_log.h:255-280 reads the word the NIC writes), and
 * a follower's thread stores its reply byte into the leader's copy of an entry (the RDMA WRITE of rc_send_entries_reply, dare_ibv_rc.c:1828-1863) while the leader's thread memcpy's a range
 * of its ring into another follower (its NIC reading host memory under update_remote_logs): which reply bytes of OTHER servers a
 * follower's copy holds depends on that timing in the reference too, and tests/test_trace_oracle.py compares everything but them. */
static void *mt_follower(void *arg)
{
    mt_ctx_t *x = arg;
    orc_cluster_t *c = x->c;
    replica_t *p = &c->r[x->idx];
    orc_log_t *log = p->log;
    cpu_set_t set; CPU_ZERO(&set); CPU_SET(x->idx, &set);
    pthread_setaffinity_np(pthread_self(), sizeof set, &set);          /* best effort */
    uint64_t old_end = log->old_end, apply = log->apply;
    for (;;) {
        const uint64_t end = __atomic_load_n(&log->end, __ATOMIC_ACQUIRE);
        const uint64_t commit = __atomic_load_n(&log->commit, __ATOMIC_ACQUIRE);
        int did = 0;
        if (end != log->len && old_end != end) {                       /* persist_new_entries + rc_send_entries_reply */
            if (old_end == log->len) old_end = 0;
            orc_entry_t *e;
            while ((e = mt_next(log, &old_end, end))) {
                p->store_count++;
                store_cmd(c, p, e);
                e->reply[p->idx] = 1;
                replica_t *dst = &c->r[e->sender];
                if (e->sender < c->n && dst->alive)
                    __atomic_store_n(&entry_at(dst->log, old_end)->reply[p->idx], 1, __ATOMIC_RELEASE);
                old_end += entry_len(e);
            }
            __atomic_store_n(&log->old_end, old_end, __ATOMIC_RELEASE);
            did = 1;
        }
        if (end != log->len && apply != commit) {                      /* poll_config_entries (HEAD) + apply_committed_entries */
            orc_entry_t *e;
            while ((e = mt_next(log, &apply, commit))) {
                if (e->type == ORC_HEAD) { if (orc_log_is_larger(log, e->data.head, log->head)) log->head = e->data.head; }
                else if (e->type != ORC_CONFIG && e->type != ORC_NOOP) {
                    __atomic_store_n(&log->apply, apply, __ATOMIC_RELAXED);      /* (the leader's prune tick reads it from its own thread) */
                    record_apply(c, p, apply, e, 2);
                    p->last_applied.idx = e->idx; p->last_applied.term = e->term; p->last_applied.offset = apply + entry_len(e);
                }
                apply += entry_len(e);
                p->apply_slot++;
            }
            __atomic_store_n(&log->apply, apply, __ATOMIC_RELEASE);
            p->cid_offset = commit;
            did = 1;
        }
        if (!did) {
            if (__atomic_load_n(&x->stop, __ATOMIC_ACQUIRE)) break;
            if (mt_now() > x->deadline_ns) { x->failed = 1; break; }
            _MT_RELAX();
        }
    }
    return NULL;
}

/* leader: replicate [from, log->end) of its ring into follower f (two pieces on a wrap), then the end doorbell */
static void mt_push(replica_t *L, replica_t *F, uint64_t from)
{
    orc_log_t *log = L->log;
    if (from == log->len) from = 0;
    if (log->end >= from) memcpy(F->log->entries + from, log->entries + from, log->end - from);
    else { memcpy(F->log->entries + from, log->entries + from, log->len - from); memcpy(F->log->entries, log->entries, log->end); }
    __atomic_store_n(&F->log->end, log->end, __ATOMIC_RELEASE);
}

/* leader: persist its own new entries, push, wait for the ACK majority up to its end, commit, doorbells, apply */
static int mt_leader_pass(orc_cluster_t *c, replica_t *L, uint64_t from, uint64_t deadline_ns)
{
    orc_log_t *log = L->log;
    persist_new_entries(c, L);
    for (int i = 0; i < c->n; i++) if (i != L->idx) mt_push(L, &c->r[i], from);
    /* a round that ends exactly on len: the log reads as empty (dare_log.h:158), nothing is committable
     * until the next append -- nobody persists or acknowledges it before then */
    if (log->end == log->len) return 0;
    const int size = L->cid.size[0];
    uint64_t off = log->commit;
    if (off == log->len) off = 0;
    orc_entry_t *e;
    while ((e = mt_next(log, &off, log->end))) {
        for (;;) {
            int replies = 0;
            for (int i = 0; i < size; i++)
                if (i == L->idx || __atomic_load_n(&e->reply[i], __ATOMIC_ACQUIRE) == 1) replies++;
            if (replies >= size / 2 + 1) break;
            if (mt_now() > deadline_ns) return -10;
            _MT_RELAX();
        }
        off += entry_len(e);
    }
    log->commit = off; L->cid_offset = off;
    for (int i = 0; i < c->n; i++) if (i != L->idx) __atomic_store_n(&c->r[i].log->commit, off, __ATOMIC_RELEASE);
    apply_committed_entries(c, L);
    return 0;
}

int orc_run_rounds_mt(orc_cluster_t *c, const orc_req_t *reqs, const uint32_t *round_n, uint64_t n_rounds,
                      const uint8_t *arena, uint64_t prune_bytes, double max_seconds)
{
    if (c->leader < 0 || c->n < 2) return -1;
    replica_t *L = &c->r[c->leader];
    for (int i = 0; i < c->n; i++) if (!c->r[i].alive || c->r[i].held || !cid_on(&L->cid, i)) return -1;
    if (orc_quiesce(c)) return -1;                                  /* every follower in step before the threads start */
    const uint64_t deadline = mt_now() + (uint64_t)(max_seconds * 1e9);
    mt_ctx_t ctx[ORC_MAX_SERVERS];
    pthread_t th[ORC_MAX_SERVERS];
    for (int i = 0; i < c->n; i++) {
        ctx[i] = (mt_ctx_t){ c, i, 0, 0, deadline };
        if (i != c->leader && pthread_create(&th[i], NULL, mt_follower, &ctx[i])) return -1;
    }
    { cpu_set_t set; CPU_ZERO(&set); CPU_SET(c->leader, &set); pthread_setaffinity_np(pthread_self(), sizeof set, &set); }
    int rc = 0;
    uint64_t since = 0, g = 0;
    for (uint64_t r = 0; r < n_rounds && !rc; r++) {
        const uint64_t from = L->log->end;
        for (uint32_t k = 0; k < round_n[r] && !rc; k++, g++) {
            if (!orc_log_append(L->log, SID_TERM(L->sid), reqs[g].req_id, reqs[g].clt_id, reqs[g].type,
                                arena ? arena + reqs[g].payload_off : NULL, reqs[g].len)) rc = -2;
            since += ORC_HDR_BYTES + reqs[g].len;
        }
        if (!rc) rc = mt_leader_pass(c, L, from, deadline);
        if (!rc && prune_bytes && since >= prune_bytes) {
            /* the prune tick: everybody caught up (trace semantics of orc_tick_prune), then log_pruning */
            since = 0;
            for (int i = 0; i < c->n && !rc; i++)
                while (i != c->leader && __atomic_load_n(&c->r[i].log->apply, __ATOMIC_ACQUIRE) != L->log->commit) {
                    if (mt_now() > deadline) { rc = -10; break; }
                    _MT_RELAX();
                }
            if (!rc) {
                const uint64_t f2 = L->log->end;
                const int appended = log_pruning(c, L);
                if (appended < 0) rc = appended;
                else if (appended) rc = mt_leader_pass(c, L, f2, deadline);
            }
        }
    }
    /* drain: every follower persisted and applied everything, then the threads leave */
    for (int i = 0; i < c->n && !rc; i++)
        while (i != c->leader && (__atomic_load_n(&c->r[i].log->apply, __ATOMIC_ACQUIRE) != L->log->commit ||
                                  __atomic_load_n(&c->r[i].log->old_end, __ATOMIC_ACQUIRE) != L->log->end)) {
            if (mt_now() > deadline) { rc = -10; break; }
            _MT_RELAX();
        }
    for (int i = 0; i < c->n; i++) if (i != c->leader) __atomic_store_n(&ctx[i].stop, 1, __ATOMIC_RELEASE);
    for (int i = 0; i < c->n; i++) if (i != c->leader) { pthread_join(th[i], NULL); if (ctx[i].failed && !rc) rc = -10; }
    { cpu_set_t set; CPU_ZERO(&set); for (int k = 0; k < CPU_SETSIZE; k++) CPU_SET(k, &set); pthread_setaffinity_np(pthread_self(), sizeof set, &set); }
    /* the leader-side step machine of the single-threaded loops, as it stands when everybody is in step */
    for (int i = 0; i < c->n; i++) if (i != c->leader) {
        L->rem_end[i] = L->log->end; L->rem_commit[i] = L->log->commit; L->cached_end[i] = L->log->end;
        L->lr_step[i] = LR_UPDATE_LOG; L->send_flag[i] = 1; L->pending[i] = PEND_NONE;
    }
    return rc;
}

/* ================================================================== */
/* helpers: payload stream, canonical digest                           */

uint64_t orc_splitmix64(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_fill_payload(uint64_t seed, uint8_t *dst, uint32_t len)
{
    uint64_t s = seed;
    uint32_t i = 0;
    while (i < len) {
        uint64_t v = orc_splitmix64(&s);
        for (int b = 0; b < 8 && i < len; b++, i++) dst[i] = (uint8_t)(v >> (8 * b));
    }
}

typedef struct { uint8_t *out; uint64_t cap, n, hash; int hashing; } sink_t;

static void sink_put(sink_t *s, const void *p, uint64_t n)
{
    const uint8_t *b = p;
    if (s->hashing) {
        uint64_t h = s->hash;
        for (uint64_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001B3ull; }
        s->hash = h;
    } else {
        for (uint64_t i = 0; i < n; i++) if (s->n + i < s->cap) s->out[s->n + i] = b[i];
    }
    s->n += n;
}

static uint64_t canon_walk(const uint8_t *ring, uint64_t len, uint64_t end,
                           uint64_t from, uint64_t to, sink_t *s)
{
    /* the same reader rule as log_get_entry / log_fit_entry, on a bare ring */
    uint64_t count = 0, off = from;
    if (end == len) return 0;
    while (off != to) {
        if (len - off < ORC_HDR_BYTES) { off = 0; if (off == to) break; }
        const orc_entry_t *e = (const orc_entry_t *)(ring + off);
        uint32_t elen = entry_len(e);
        if (len - off < elen) { off = 0; continue; }
        uint8_t zero = 0;
        uint32_t dlen;
        const void *data;
        switch (e->type) {
        case ORC_CONFIG: dlen = 16; data = &e->data.cid; break;
        case ORC_HEAD:   dlen = 8;  data = &e->data.head; break;
        case ORC_NOOP:   dlen = 0;  data = NULL; break;
        default:         dlen = e->data.cmd.len; data = e->data.cmd.cmd; break;
        }
        sink_put(s, &off, 8); sink_put(s, &e->idx, 8); sink_put(s, &e->term, 8);
        sink_put(s, &e->req_id, 8); sink_put(s, &e->clt_id, 2); sink_put(s, &e->type, 1);
        sink_put(s, &zero, 1); sink_put(s, &dlen, 4);
        if (dlen) sink_put(s, data, dlen);
        off += elen;
        count++;
        if (count > (1ull << 32)) break;        /* corrupt ring guard */
    }
    return count;
}

uint64_t orc_canon(const uint8_t *ring, uint64_t len, uint64_t end, uint64_t from, uint64_t to,
                   uint8_t *out, uint64_t cap, uint64_t *n_entries)
{
    sink_t s = { out, cap, 0, 0, 0 };
    uint64_t n = canon_walk(ring, len, end, from, to, &s);
    if (n_entries) *n_entries = n;
    return s.n;
}

uint64_t orc_canon_hash(const uint8_t *ring, uint64_t len, uint64_t end, uint64_t from, uint64_t to,
                        uint64_t *n_entries)
{
    sink_t s = { NULL, 0, 0, 0xCBF29CE484222325ull, 1 };
    uint64_t n = canon_walk(ring, len, end, from, to, &s);
    if (n_entries) *n_entries = n;
    return s.hash;
}

/* mask[i] = 1 for every byte of the ring that an entry in [from, to) DEFINES:
 * header bytes 0..40, cmd.len + payload of client entries, the 16-B cid, the
 * 8-B head, and the same fields of the stale header a case-2 wrap leaves behind.
 * Struct padding (41..47) and the 14 bytes behind a payload are never written
 * by log_append_entry (they keep whatever the previous lap left there) and are
 * therefore not part of any parity claim. */
uint64_t orc_defined_mask(const uint8_t *ring, uint64_t len, uint64_t end,
                          uint64_t from, uint64_t to, uint8_t *mask)
{
    uint64_t count = 0, off = from;
    if (end == len) return 0;
    while (off != to) {
        if (len - off < ORC_HDR_BYTES) { off = 0; if (off == to) break; }
        const orc_entry_t *e = (const orc_entry_t *)(ring + off);
        uint32_t elen = entry_len(e);
        if (len - off < elen) {
            /* stale header: log_append_entry never writes `sender` (only persist does,
             * and only on the real entry), so byte 27 keeps the previous lap's value */
            memset(mask + off, 1, 41);
            mask[off + 27] = 0;
            mask[off + 48] = mask[off + 49] = 1;
            off = 0;
            continue;
        }
        memset(mask + off, 1, 41);
        switch (e->type) {
        case ORC_CONFIG: memset(mask + off + 48, 1, 16); break;
        case ORC_HEAD:   memset(mask + off + 48, 1, 8); break;
        case ORC_NOOP:   break;
        default:         memset(mask + off + 48, 1, 2 + (size_t)e->data.cmd.len); break;
        }
        off += elen;
        if (++count > (1ull << 32)) break;
    }
    return count;
}
