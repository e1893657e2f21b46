/*
 * ref_harness.c -- thin exported wrappers around the REFERENCE's own log
 * implementation, src/include/dare/dare_log.h, compiled unchanged from where
 * it lies under /root/reference (never copied into this repository).
 *
 * TEST INFRASTRUCTURE ONLY.  Output: oracle/_ref/libapus_ref.so (git-ignored,
 * built by oracle/Makefile only when /root/reference exists).  It pins the
 * restatement in apus_oracle.c (tests/test_oracle_vs_ref.py) and generates
 * the fixtures under tests/golden/ (tests/golden/make_golden.py).
 *
 * The reference header is all `static` functions over a malloc'd dare_log_t;
 * each wrapper below forwards to exactly one of them.
 */
#include <stdio.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

FILE *log_fp;                    /* debug.h:104 expects the including program to own it */
int prev_log_entry_head = 0;     /* dare_server.c:71 owns it in the reference */

#include "dare_log.h"            /* -I/root/reference/src/include/dare */

static void ensure_fp(void)
{
    if (!log_fp) log_fp = fopen("/dev/null", "w");
}

/* log_new() always allocates LOG_SIZE; a shorter ring is obtained by lowering
 * log->len afterwards (every other function reads log->len, not LOG_SIZE). */
void *ref_log_new(uint64_t len)
{
    ensure_fp();
    dare_log_t *log = log_new();
    if (!log) return NULL;
    if (len && len < log->len) {
        log->len = len;
        log->end = len;
        log->tail = len;
        log->old_end = len;
    }
    return log;
}

void ref_log_free(void *p) { log_free((dare_log_t *)p); }

uint64_t ref_log_append(void *p, uint64_t term, uint64_t req_id, uint16_t clt_id,
                        uint8_t type, const void *data, uint16_t data_len)
{
    dare_log_t *log = p;
    if (type == CONFIG || type == HEAD || type == NOOP)
        return log_append_entry(log, term, req_id, clt_id, type, (void *)data);
    /* client entry: the reference takes an sm_cmd_t {u16 len; u8 cmd[]} */
    uint8_t *buf = malloc(sizeof(sm_cmd_t) + (size_t)data_len + 1);
    sm_cmd_t *cmd = (sm_cmd_t *)buf;
    cmd->len = data_len;
    if (data_len) memcpy(cmd->cmd, data, data_len);
    uint64_t idx = log_append_entry(log, term, req_id, clt_id, type, cmd);
    free(buf);
    return idx;
}

void ref_log_offsets(void *p, uint64_t out[8])
{
    dare_log_t *log = p;
    out[0] = log->head; out[1] = log->apply; out[2] = log->commit; out[3] = log->end;
    out[4] = log->tail; out[5] = log->old_end; out[6] = log->old_commit; out[7] = log->len;
}

void ref_log_set_offsets(void *p, const uint64_t in[8])
{
    dare_log_t *log = p;
    log->head = in[0]; log->apply = in[1]; log->commit = in[2]; log->end = in[3];
    log->tail = in[4]; log->old_end = in[5]; log->old_commit = in[6];
}

uint8_t *ref_log_entries(void *p) { return ((dare_log_t *)p)->entries; }
int  ref_prev_head(void) { return prev_log_entry_head; }
void ref_set_prev_head(int v) { prev_log_entry_head = v; }

uint64_t ref_log_end_distance(void *p, uint64_t off) { return log_offset_end_distance(p, off); }
int ref_log_is_larger(void *p, uint64_t l, uint64_t r) { return log_is_offset_larger(p, l, r); }

uint64_t ref_log_get_entry(void *p, uint64_t off)
{
    dare_log_entry_t *e = log_get_entry(p, &off);
    return e ? off : UINT64_MAX;
}

uint32_t ref_log_entry_len_at(void *p, uint64_t off)
{
    dare_log_t *log = p;
    return log_entry_len((dare_log_entry_t *)(log->entries + off));
}

uint64_t ref_log_get_tail(void *p) { return log_get_tail(p); }

/* builds the NC-buffer of server `slot` inside the log and copies it out:
 * out[0] = len, then (idx, term, offset) triples */
uint64_t ref_log_to_ncbuf(void *p, int slot, uint64_t *out, uint64_t cap_entries)
{
    dare_log_t *log = p;
    log_entries_to_nc_buf(log, &log->nc_buf[slot]);
    uint64_t n = log->nc_buf[slot].len;
    for (uint64_t i = 0; i < n && i < cap_entries; i++) {
        out[3 * i + 0] = log->nc_buf[slot].entries[i].idx;
        out[3 * i + 1] = log->nc_buf[slot].entries[i].term;
        out[3 * i + 2] = log->nc_buf[slot].entries[i].offset;
    }
    return n;
}

/* loads `n` determinants into nc_buf[slot] and runs log_find_remote_end_offset */
uint64_t ref_log_find_remote_end(void *p, int slot, const uint64_t *dets, uint64_t n)
{
    dare_log_t *log = p;
    log->nc_buf[slot].len = n;
    for (uint64_t i = 0; i < n; i++) {
        log->nc_buf[slot].entries[i].idx = dets[3 * i + 0];
        log->nc_buf[slot].entries[i].term = dets[3 * i + 1];
        log->nc_buf[slot].entries[i].offset = dets[3 * i + 2];
    }
    return log_find_remote_end_offset(log, &log->nc_buf[slot]);
}

/* struct layout facts */
void ref_layout(uint64_t out[12])
{
    out[0] = sizeof(dare_log_entry_t);
    out[1] = offsetof(dare_log_entry_t, idx);
    out[2] = offsetof(dare_log_entry_t, term);
    out[3] = offsetof(dare_log_entry_t, req_id);
    out[4] = offsetof(dare_log_entry_t, clt_id);
    out[5] = offsetof(dare_log_entry_t, type);
    out[6] = offsetof(dare_log_entry_t, sender);
    out[7] = offsetof(dare_log_entry_t, reply);
    out[8] = offsetof(dare_log_entry_t, data);
    out[9] = sizeof(dare_cid_t);
    out[10] = offsetof(dare_log_t, entries);
    out[11] = LOG_SIZE;
}
