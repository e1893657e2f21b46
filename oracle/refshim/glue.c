/*
 * glue.c -- what one instance of the UNMODIFIED reference server needs around it when it
 * runs inside the oracle process.  TEST INFRASTRUCTURE ONLY; compiled together with
 * /root/reference/src/dare/{dare_server,dare_ibv,dare_ibv_rc,dare_ibv_ud,dare_ep_db,
 * dare_kvs_sm}.c, src/config-comp/config-dare.c and utils/rbtree/src/rbtree.c into
 * oracle/_ref/libapus_ref_loops.so (recipe: oracle/Makefile, target `loops`).  The driver
 * (refcluster.c) loads one private copy of that library per server, so each copy keeps
 * the reference's process-wide singletons (`data` dare_server.c:69, `dare_ib_device`
 * dare_ibv.c:33, `tailhead` message.h:20) to itself.
 *
 * Provided here, and nothing else:
 *   1. the libev names of refshim/ev.h (timers are fired by the trace driver);
 *   2. the libconfig names of refshim/libconfig.h;
 *   3. the proxy side of the B-inner boundary (dare_server_input_t callbacks,
 *      submission through tailhead/tailq_lock) with a recorder instead of redis;
 *   4. four libc names the reference calls that must not act on the whole process:
 *      signal(), pthread_exit(), gethostname() (get_unique_slid, dare_ibv_ud.c:1523),
 *      free() (dare_server_init frees `input` at :208 and keeps reading it at :410);
 *   5. accessors for the driver.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <setjmp.h>
#include <signal.h>
#include <pthread.h>

#include "dare_ibv.h"
#include "dare_server.h"
#include "dare_sm.h"
#include "message.h"
#include <libconfig.h>

/* the reference's own globals (dare_server.c:64-91) */
extern dare_server_data_t data;
extern uint64_t dare_state;
extern int prev_log_entry_head;
extern dare_log_entry_det_t last_applied_entry;
extern ev_idle poll_event;
extern ev_timer timer_event, hb_event, to_adjust_event, prune_event;

#ifndef GLUE_PROC          /* (one server per process, oracle/Makefile `procref`: the real libev, no harness) */
/* ------------------------------------------------------------------ 1. libev */
static struct ev_loop the_loop;
struct ev_loop *apus_ev_default_loop(void) { return &the_loop; }
void ev_timer_again(struct ev_loop *loop, ev_timer *w)
{
    if (w->repeat > 0.) { w->active = 1; w->at = loop->now + w->repeat; }
    else w->active = 0;
}
void ev_timer_stop(struct ev_loop *loop, ev_timer *w) { (void)loop; w->active = 0; }
void ev_idle_start(struct ev_loop *loop, ev_idle *w) { (void)loop; w->active = 1; }
ev_tstamp ev_now(struct ev_loop *loop) { return loop->now; }
int  ev_run(struct ev_loop *loop, int flags) { (void)loop; (void)flags; return 0; }
void ev_break(struct ev_loop *loop, int how) { (void)how; loop->broken = 1; }

#endif
/* ------------------------------------------------------------------ 2. libconfig */
void config_init(config_t *c) { memset(c, 0, sizeof *c); }
void config_destroy(config_t *c) { free(c->root.children); c->root.children = NULL; }

static char *skip_ws(char *p)
{
    for (;;) {
        while (*p && isspace((unsigned char)*p)) p++;
        if (*p == '#') { while (*p && *p != '\n') p++; continue; }
        return p;
    }
}

/* settings of the form  name = value;  and one level of  group = { name = value; ... };
 * children of every group are appended flat to root.children with the group name as prefix */
int config_read_file(config_t *c, const char *filename)
{
    FILE *f = fopen(filename, "r");
    if (!f) { c->error_text = "file I/O error"; c->error_file = filename; return CONFIG_FALSE; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *buf = malloc((size_t)n + 1);
    n = (long)fread(buf, 1, (size_t)n, f); buf[n] = 0; fclose(f);
    int cap = 64; c->root.children = calloc((size_t)cap, sizeof(config_setting_t)); c->root.n_children = 0;
    char group[64] = "";
    char *p = buf;
    for (;;) {
        p = skip_ws(p);
        if (!*p) break;
        if (*p == '}') { group[0] = 0; p++; p = skip_ws(p); if (*p == ';') p++; continue; }
        char name[64]; int k = 0;
        while (*p && (isalnum((unsigned char)*p) || *p == '_') && k < 63) name[k++] = *p++;
        name[k] = 0;
        p = skip_ws(p);
        if (!k || (*p != '=' && *p != ':')) { c->error_text = "syntax error"; c->error_file = filename; free(buf); return CONFIG_FALSE; }
        p = skip_ws(p + 1);
        if (*p == '{') { snprintf(group, sizeof group, "%s", name); p++;
            if (c->root.n_children < cap) { config_setting_t *s = &c->root.children[c->root.n_children++]; snprintf(s->name, sizeof s->name, "%s", name); s->is_group = 1; }
            continue; }
        char val[128]; k = 0;
        while (*p && *p != ';' && *p != '\n' && k < 127) val[k++] = *p++;
        while (k && isspace((unsigned char)val[k - 1])) k--;
        val[k] = 0;
        if (*p == ';') p++;
        if (c->root.n_children < cap) {
            config_setting_t *s = &c->root.children[c->root.n_children++];
            if (group[0]) snprintf(s->name, sizeof s->name, "%.30s.%.30s", group, name);
            else snprintf(s->name, sizeof s->name, "%s", name);
            snprintf(s->value, sizeof s->value, "%s", val);
        }
    }
    free(buf);
    return CONFIG_TRUE;
}

static config_t *lookup_owner;
config_setting_t *config_lookup(const config_t *c, const char *path)
{
    for (int i = 0; i < c->root.n_children; i++)
        if (!strcmp(c->root.children[i].name, path)) { lookup_owner = (config_t *)c; return &c->root.children[i]; }
    return NULL;
}
static const char *child_value(const config_setting_t *g, const char *name)
{
    if (!g || !g->is_group || !lookup_owner) return NULL;
    char full[64]; snprintf(full, sizeof full, "%.30s.%.30s", g->name, name);
    for (int i = 0; i < lookup_owner->root.n_children; i++)
        if (!strcmp(lookup_owner->root.children[i].name, full)) return lookup_owner->root.children[i].value;
    return NULL;
}
int config_setting_lookup_float(const config_setting_t *g, const char *name, double *v)
{
    const char *s = child_value(g, name);
    if (!s || !strpbrk(s, ".eE")) return CONFIG_FALSE;      /* libconfig is strict about the type */
    *v = strtod(s, NULL); return CONFIG_TRUE;
}
int config_setting_lookup_int64(const config_setting_t *g, const char *name, long long *v)
{
    const char *s = child_value(g, name);
    if (!s || strpbrk(s, ".eE")) return CONFIG_FALSE;
    *v = strtoll(s, NULL, 0); return CONFIG_TRUE;
}

#ifdef GLUE_PROC
/* the two top-level lookups src/config-comp/config-proxy.c makes (db_name, ip_address: strings; req_log, port: ints) */
static const char *top_value(const config_t *c, const char *name)
{
    for (int i = 0; i < c->root.n_children; i++)
        if (!c->root.children[i].is_group && !strcmp(c->root.children[i].name, name)) return c->root.children[i].value;
    return NULL;
}
int config_lookup_int(const config_t *c, const char *path, int *v)
{
    const char *s = top_value(c, path);
    if (!s || *s == '"' || strpbrk(s, ".eE")) return CONFIG_FALSE;
    *v = (int)strtol(s, NULL, 0); return CONFIG_TRUE;
}
int config_lookup_string(const config_t *c, const char *path, const char **v)
{
    config_t *cc = (config_t *)c;
    for (int i = 0; i < cc->root.n_children; i++) {
        config_setting_t *st = &cc->root.children[i];
        if (st->is_group || strcmp(st->name, path)) continue;
        char *s = st->value;
        if (*s != '"') return CONFIG_FALSE;
        size_t n = strlen(s);
        if (n >= 2 && s[n - 1] == '"') s[n - 1] = 0;           /* (in place: the quotes go) */
        *v = s + 1; return CONFIG_TRUE;
    }
    return CONFIG_FALSE;
}
/* get_unique_slid() = hostname[21] - '0' (dare_ibv_ud.c:1523-1530) must be this server's LID = server_idx + 1:
 * the reference assumes one server per host; here three share one */
int gethostname(char *name, size_t len)
{
    const char *idx = getenv("server_idx");
    if (len < 23) return -1;
    memset(name, 'a', 21); name[21] = (char)('0' + (idx ? atoi(idx) : 0) + 1); name[22] = 0;
    return 0;
}
#else
/* ------------------------------------------------------------------ 4. libc names */
static int my_index = -1;
static void *kept_input;
static jmp_buf exit_jmp;
static int exit_armed, exited;

__sighandler_t signal(int sig, __sighandler_t h) { (void)sig; (void)h; return SIG_DFL; }
void pthread_exit(void *ret)
{
    (void)ret; exited = 1;
    if (exit_armed) longjmp(exit_jmp, 1);
    fprintf(stderr, "ref instance %d: dare_server_shutdown outside a driver call\n", my_index);
    abort();
}
int gethostname(char *name, size_t len)
{
    /* get_unique_slid() = name[21] - '0' must equal this instance's port LID (index + 1) */
    if (len < 23) return -1;
    memset(name, 'a', 21); name[21] = (char)('0' + my_index + 1); name[22] = 0;
    return 0;
}
extern void __libc_free(void *);
void free(void *p) { if (p && p == kept_input) return; __libc_free(p); }

/* ------------------------------------------------------------------ 3. proxy side */
typedef struct { uint64_t off, idx; uint32_t len; uint16_t clt_id; uint8_t type, kind; } glue_apply_t;
static glue_apply_t *apply_log; static uint64_t apply_n, apply_cap; static int record_apply = 1;
static uint64_t highest_rec, store_count, follower_applied;

static void rec(uint8_t kind)
{
    if (!record_apply) return;
    /* both upcalls are made while log->apply still points at the entry (dare_server.c:1953-1966) */
    uint64_t off = data.log->apply;
    dare_log_entry_t *e = log_get_entry(data.log, &off);
    if (apply_n == apply_cap) { apply_cap = apply_cap ? apply_cap * 2 : 4096; apply_log = realloc(apply_log, apply_cap * sizeof *apply_log); }
    glue_apply_t *r = &apply_log[apply_n++];
    r->off = off; r->idx = e->idx; r->len = e->data.cmd.len; r->clt_id = e->clt_id; r->type = e->type; r->kind = kind;
}
/* what BerkeleyDB is handed: stablestorage_save_request (src/proxy/proxy.c:268-291) with the reference's own
 * structs -- the record starts at &entry->clt_id and its length comes out of the proxy_send_msg overlay
 * (SURVEY.md 9-Q1); store_record (src/db/db-interface.c:65-96) appends it and adds its size to records_len */
static uint8_t *store_buf; static uint64_t store_len, store_cap; static uint32_t records_len_; static int record_store;
static FILE *store_file;              /* record_store == 2: the records go to a file (32 KiB buffer = the reference's pagesize), not to memory */
static void store_record_(size_t n, const void *d)
{
    records_len_ += (uint32_t)n;
    if (!record_store) return;
    if (record_store == 2) {
        if (!store_file) { store_file = tmpfile(); if (store_file) setvbuf(store_file, NULL, _IOFBF, 32 * 1024); }
        if (store_file) fwrite(d, 1, n, store_file);
        return;
    }
    if (store_len + n > store_cap) { store_cap = (store_len + n) * 2 + 4096; store_buf = realloc(store_buf, store_cap); }
    memcpy(store_buf + store_len, d, n); store_len += n;
}
size_t glue_store_record_size(const void *d);      /* glue_store.c: the reference's own structs say how long */
static void cb_store_cmd(void *d, void *arg)
{
    (void)arg; store_count++;
    size_t n = glue_store_record_size(d);
    if (n) store_record_(n, d);
}
static void cb_do_action(uint16_t clt_id, uint8_t type, size_t n, void *d, void *arg)
{ (void)clt_id; (void)type; (void)n; (void)d; (void)arg; follower_applied++; rec(2); }
static void cb_update_state(void *arg) { (void)arg; highest_rec++; rec(1); }
static uint32_t cb_get_db_size(void *arg) { (void)arg; return 0; }
static void cb_create_snapshot(void *s, void *arg) { (void)s; (void)arg; }
static int cb_apply_snapshot(void *s, uint32_t n, void *arg) { (void)s; (void)n; (void)arg; return 0; }

/* ------------------------------------------------------------------ 5. driver API */
#define GUARD(stmt) do { exit_armed = 1; if (!setjmp(exit_jmp)) { stmt; } exit_armed = 0; } while (0)

int glue_start(int idx, int group_size, int join, const char *cfg, const char *logpath, uint64_t log_len)
{
    my_index = idx;
    dare_server_input_t *in = calloc(1, sizeof *in);
    in->log = fopen(logpath && logpath[0] ? logpath : "/dev/null", "w");
    in->name = ""; in->output = "dare_servers.out";
    in->srv_type = join ? SRV_TYPE_JOIN : SRV_TYPE_START;
    in->sm_type = CLT_KVS;                              /* proxy.c:29 */
    in->server_idx = (uint8_t)idx; in->group_size = (uint8_t)group_size;
    in->do_action = cb_do_action; in->store_cmd = cb_store_cmd; in->get_db_size = cb_get_db_size;
    in->create_db_snapshot = cb_create_snapshot; in->apply_db_snapshot = cb_apply_snapshot;
    in->update_state = cb_update_state;
    snprintf(in->config_path, sizeof in->config_path, "%s", cfg);
    kept_input = in;
    TAILQ_INIT(&tailhead);                              /* proxy.c:486-496 */
    pthread_spin_init(&tailq_lock, PTHREAD_PROCESS_PRIVATE);
    GUARD(dare_server_init(in));
    if (exited || !data.log) return -1;
    if (log_len && log_len < data.log->len) {
        /* every function reads log->len, not LOG_SIZE; same trick as ref_harness.c */
        data.log->len = log_len; data.log->end = log_len; data.log->tail = log_len; data.log->old_end = log_len;
    }
    return 0;
}

static ev_timer *timer_of(int which)
{
    switch (which) { case 0: return &timer_event; case 1: return &prune_event; case 2: return &hb_event; case 3: return &to_adjust_event; }
    return NULL;
}
int glue_timer_armed(int which) { ev_timer *w = timer_of(which); return w && w->active; }
double glue_timer_repeat(int which) { ev_timer *w = timer_of(which); return w ? w->repeat : 0.; }
int glue_fire(int which)
{
    ev_timer *w = timer_of(which);
    if (!w || !w->active || exited) return 0;
    /* libev: a repeating timer is re-armed before its callback runs */
    the_loop.now = w->at > the_loop.now ? w->at : the_loop.now;
    w->at = the_loop.now + w->repeat;
    GUARD(w->cb(&the_loop, w, 0x100));
    return 1;
}
int glue_poll(void)
{
    if (!poll_event.active || exited) return 0;
    GUARD(poll_event.cb(&the_loop, &poll_event, 0x2000));
    return 1;
}
void glue_advance(double dt) { the_loop.now += dt; }

/* leader_handle_submit_req's enqueue (proxy.c:145-158), ids already assigned by the trace */
void glue_submit(uint8_t type, uint16_t conn, uint64_t req_id, const void *buf, uint16_t len)
{
    tailq_entry_t *n = malloc(sizeof *n);
    n->type = type; n->connection_id = conn; n->req_id = req_id; n->cmd.len = len;
    if (len) memcpy(n->cmd.cmd, buf, len);
    pthread_spin_lock(&tailq_lock);
    TAILQ_INSERT_TAIL(&tailhead, n, entries);
    pthread_spin_unlock(&tailq_lock);
}

void    *glue_log(void) { return data.log; }
uint8_t *glue_entries(void) { return data.log->entries; }
void    *glue_ctrl(void) { return data.ctrl_data; }
int      glue_idx(void) { return data.config.idx; }
uint64_t glue_sid(void) { return data.ctrl_data ? data.ctrl_data->sid : 0; }
uint64_t glue_state(void) { return dare_state; }
int      glue_exited(void) { return exited; }
int      glue_is_leader(void) { return is_leader(); }
int      glue_prev_head(void) { return prev_log_entry_head; }
uint64_t glue_highest_rec(void) { return highest_rec; }
uint64_t glue_store_count(void) { return store_count; }
uint64_t glue_apply_count(void) { return highest_rec + follower_applied; }
void     glue_record_apply(int on) { record_apply = on; }
void     glue_record_store(int on) { record_store = on; }
const void *glue_store_stream(uint64_t *n) { *n = store_len; return store_buf; }
uint32_t glue_records_len(void) { return records_len_; }
const void *glue_apply_log(uint64_t *n) { *n = apply_n; return apply_log; }
void glue_cid(uint64_t out[4])
{
    out[0] = data.config.cid.epoch; out[1] = data.config.cid.size[0] | (uint64_t)data.config.cid.size[1] << 8 |
             (uint64_t)data.config.cid.state << 16; out[2] = data.config.cid.bitmask; out[3] = data.config.cid_offset;
}
/* per-peer replication state the leader keeps (server_t + ctrl_data.log_offsets) */
void glue_peer(int i, uint64_t out[6])
{
    server_t *s = &data.config.servers[i];
    dare_ib_ep_t *ep = (dare_ib_ep_t *)s->ep;
    out[0] = s->next_lr_step; out[1] = s->send_flag; out[2] = s->fail_count;
    out[3] = data.ctrl_data->log_offsets[i].end; out[4] = data.ctrl_data->log_offsets[i].commit;
    out[5] = (ep ? (uint64_t)ep->rc_connected : 0) | (ep ? (uint64_t)ep->log_access << 1 : 0) |
             (data.ctrl_data->vote_ack[i] != data.log->len ? 4u : 0);
}
int glue_all_connected(void)
{
    uint8_t size = get_extended_group_size(data.config);
    for (uint8_t i = 0; i < size; i++) {
        if (i == data.config.idx || !CID_IS_SERVER_ON(data.config.cid, i)) continue;
        dare_ib_ep_t *ep = (dare_ib_ep_t *)data.config.servers[i].ep;
        if (!ep || !ep->rc_connected) return 0;
    }
    return 1;
}
#endif
