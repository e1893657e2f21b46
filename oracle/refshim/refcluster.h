/*
 * refcluster.h -- C face of the reference-as-is cluster driver (refcluster.c).
 * TEST INFRASTRUCTURE ONLY.  Mirrors the cluster half of ../apus_oracle.h.
 */
#ifndef APUS_REFCLUSTER_H
#define APUS_REFCLUSTER_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct refc refc_t;

/* same layout as orc_req_t (apus_oracle.h) */
typedef struct {
    uint64_t req_id, payload_off;
    uint16_t clt_id, len;
    uint8_t  type, pad[3];
} refc_req_t;

/* one upcall of apply_committed_entries as the proxy callbacks saw it (glue.c) */
typedef struct { uint64_t off, idx; uint32_t len; uint16_t clt_id; uint8_t type, kind; } refc_apply_t;

refc_t *refc_new(int n, uint64_t log_len, const char *lib_path, const char *cfg_path, const char *log_dir);
void    refc_free(refc_t *c);
const char *refc_error(const refc_t *c);

int refc_elect(refc_t *c, int winner);
int refc_round(refc_t *c, const refc_req_t *reqs, int n, const uint8_t *arena);
int refc_tick_prune(refc_t *c);
int refc_kill(refc_t *c, int r);
int refc_hold(refc_t *c, int r);
int refc_release(refc_t *c, int r);
int refc_quiesce(refc_t *c);
int refc_join(refc_t *c, int r);                 /* a new server joins; the leader must hand out slot r */
uint64_t refc_state(refc_t *c, int r);
int refc_poll(refc_t *c, int r);
int refc_fire(refc_t *c, int r, int which);     /* 0 init/rc-info, 1 prune, 2 heartbeat, 3 timeout adjust */

int      refc_leader(const refc_t *c);
int      refc_group_size(const refc_t *c);
int      refc_alive(refc_t *c, int r);
int      refc_gone(refc_t *c, int r);            /* the server shut itself down and freed its state */
void     refc_offsets(refc_t *c, int r, uint64_t out[8]);   /* head apply commit end tail old_end old_commit len */
uint8_t *refc_entries(refc_t *c, int r);
uint64_t refc_sid(refc_t *c, int r);
int      refc_prev_head(refc_t *c, int r);
uint64_t refc_highest_rec(refc_t *c, int r);
uint64_t refc_store_count(refc_t *c, int r);
uint64_t refc_apply_count(refc_t *c, int r);
void     refc_record_apply(refc_t *c, int on);
const void *refc_apply_log(refc_t *c, int r, uint64_t *n);
/* the bytes handed to BerkeleyDB by persist_new_entries -> proxy_store_cmd, record after record (glue.c) */
void     refc_record_store(refc_t *c, int on);
const void *refc_store_stream(refc_t *c, int r, uint64_t *n);
uint32_t refc_records_len(refc_t *c, int r);
void     refc_cid(refc_t *c, int r, uint64_t out[4]);       /* epoch, size0|size1<<8|state<<16, bitmask, cid_offset */
void     refc_peer(refc_t *c, int r, int i, uint64_t out[6]);
uint64_t refc_round_count(const refc_t *c);
const uint64_t *refc_round_commit(const refc_t *c);
const uint64_t *refc_round_end(const refc_t *c);

#ifdef __cplusplus
}
#endif
#endif
