/*
 * refcluster.c -- trace driver for N instances of the UNMODIFIED reference server.
 * TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libapus_fabric.so).
 *
 * It gives the reference itself the face the restated oracle has (apus_oracle.h
 * orc_elect / orc_round / orc_tick_prune / orc_kill / orc_hold / orc_release /
 * orc_quiesce), so the same event trace can be replayed on both and the oracle's
 * loops can be pinned to the reference's own outputs (tests/test_oracle_vs_refloops.py,
 * tests/golden/make_cluster_golden.py).
 *
 * One private copy of oracle/_ref/libapus_ref_loops.so is loaded per server (memfd +
 * dlopen: distinct inodes give distinct link maps, -Bsymbolic keeps every copy's globals
 * to itself).  All copies share the fabric in this library.  Everything runs on the
 * calling thread; "concurrency" is the write hook below: a server is polled at the moment
 * the leader's WRITE of its `end` or `commit` word lands -- the schedule the oracle fixes
 * (DESIGN.md section 3).  Timers fire only when the trace says so:
 *   ELECT(w)  every live non-candidate misses the heartbeat (hb_receive_cb,
 *             dare_server.c:822 -> start_election :1264), then w's election timeout
 *             fires first; votes; poll_vote_count :1327; first heartbeat.
 *   PRUNE     prune_log_cb :1977 on the leader.
 *   KILL(r)   the port dies; a live leader notices on its next two heartbeats
 *             (fail_count, dare_ibv_rc.c:2747) and removes r (check_failure_count :1189).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include "fabric.h"
#include "refcluster.h"

#define MAXN 13
enum { T_INIT = 0, T_PRUNE = 1, T_HB = 2, T_ADJ = 3 };
#define ST_LOG_RECOVERED 0x20

typedef struct {
    void *dl; int fd;
    int alive, busy;
    int      (*start)(int, int, int, const char *, const char *, uint64_t);
    int      (*fire)(int);
    int      (*timer_armed)(int);
    int      (*poll)(void);
    void     (*submit)(uint8_t, uint16_t, uint64_t, const void *, uint16_t);
    void    *(*log)(void);
    uint8_t *(*entries)(void);
    uint64_t (*sid)(void);
    uint64_t (*state)(void);
    int      (*exited)(void);
    int      (*is_leader)(void);
    int      (*prev_head)(void);
    uint64_t (*highest_rec)(void);
    uint64_t (*store_count)(void);
    uint64_t (*apply_count)(void);
    void     (*record_apply)(int);
    const void *(*apply_log)(uint64_t *);
    void     (*cid)(uint64_t *);
    void     (*peer)(int, uint64_t *);
    int      (*all_connected)(void);
} inst_t;

struct refc {
    int n;
    uint64_t log_len;
    inst_t in[MAXN];
    int leader;
    uint64_t n_rounds, rounds_cap, *round_commit, *round_end;
    char err[256];
};

static uint64_t *offs(inst_t *t) { return (uint64_t *)t->log(); }   /* head apply commit end tail old_end old_commit len */

static int call_poll(refc_t *c, int i)
{
    inst_t *t = &c->in[i];
    if (!t->alive || t->busy || t->exited()) return 0;
    t->busy = 1;
    int prev = fab_enter(i);
    int r = t->poll();
    fab_leave(prev);
    t->busy = 0;
    return r;
}
static int call_fire(refc_t *c, int i, int which)
{
    inst_t *t = &c->in[i];
    if (!t->alive || t->busy || t->exited()) return 0;
    t->busy = 1;
    int prev = fab_enter(i);
    int r = t->fire(which);
    fab_leave(prev);
    t->busy = 0;
    return r;
}

static void on_write(void *arg, int from, int to, uint64_t raddr, uint32_t len)
{
    (void)from;
    refc_t *c = arg;
    if (to < 0 || to >= c->n || len != 8) return;
    inst_t *t = &c->in[to];
    if (!t->alive || t->busy) return;
    uint64_t base = (uint64_t)(uintptr_t)t->log();
    /* offsetof(dare_log_t, commit) = 16, offsetof(dare_log_t, end) = 24 */
    if (raddr == base + 24 || raddr == base + 16) call_poll(c, to);
}

static int load_instance(refc_t *c, int i, const void *img, size_t img_len)
{
    inst_t *t = &c->in[i];
    char name[32]; snprintf(name, sizeof name, "apus_ref_loops_%d", i);
    t->fd = memfd_create(name, 0);
    if (t->fd < 0) return -1;
    if (write(t->fd, img, img_len) != (ssize_t)img_len) return -1;
    char path[64]; snprintf(path, sizeof path, "/proc/self/fd/%d", t->fd);
    t->dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!t->dl) { snprintf(c->err, sizeof c->err, "dlopen: %s", dlerror()); return -1; }
#define SYM(f) do { *(void **)&t->f = dlsym(t->dl, "glue_" #f); if (!t->f) { snprintf(c->err, sizeof c->err, "missing glue_%s", #f); return -1; } } while (0)
    SYM(start); SYM(fire); SYM(timer_armed); SYM(poll); SYM(submit); SYM(log); SYM(entries); SYM(sid); SYM(state);
    SYM(exited); SYM(is_leader); SYM(prev_head); SYM(highest_rec); SYM(store_count); SYM(apply_count);
    SYM(record_apply); SYM(apply_log); SYM(cid); SYM(peer); SYM(all_connected);
#undef SYM
    return 0;
}

const char *refc_error(const refc_t *c) { return c->err; }

refc_t *refc_new(int n, uint64_t log_len, const char *lib_path, const char *cfg_path, const char *log_dir)
{
    if (n < 1 || n > MAXN) return NULL;
    refc_t *c = calloc(1, sizeof *c);
    c->n = n; c->log_len = log_len; c->leader = -1;
    FILE *f = fopen(lib_path, "rb");
    if (!f) { free(c); return NULL; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    void *img = malloc((size_t)sz);
    if (fread(img, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(img); free(c); return NULL; }
    fclose(f);
    fab_reset();
    fab_set_write_hook(on_write, c);
    for (int i = 0; i < n; i++) {
        if (load_instance(c, i, img, (size_t)sz)) { fprintf(stderr, "refc_new: %s\n", c->err); free(img); return NULL; }
        char lp[256] = "";
        if (log_dir && log_dir[0]) snprintf(lp, sizeof lp, "%s/ref_srv%d.log", log_dir, i);
        int prev = fab_enter(i);
        int rc = c->in[i].start(i, n, 0, cfg_path, lp, log_len);
        fab_leave(prev);
        if (rc) { fprintf(stderr, "refc_new: dare_server_init failed on %d\n", i); free(img); return NULL; }
        c->in[i].alive = 1;
    }
    free(img);
    for (int i = 0; i < n; i++) call_fire(c, i, T_INIT);            /* init_network_cb :372 */
    if (n == 1) {
        /* "I'm the only one; I am the leader" (:416-424); LOG_RECOVERED is never set on this
         * path in the reference, so polling() stops at :1030 -- a 1-server group cannot be driven */
        c->leader = 0;
        return c;
    }
    for (int i = 0; i < n; i++) call_fire(c, i, T_INIT);            /* exchange_rc_info_cb :472 (mcast RC_SYN) */
    for (int sweep = 0; sweep < 64 * n; sweep++) {
        for (int i = 0; i < n; i++) call_poll(c, i);
        int done = 1;
        for (int i = 0; i < n; i++)
            if (fab_pending_ud(i) || !c->in[i].all_connected() || !(c->in[i].state() & ST_LOG_RECOVERED)) done = 0;
        if (done) break;
    }
    for (int i = 0; i < n; i++)
        if (!c->in[i].all_connected() || !(c->in[i].state() & ST_LOG_RECOVERED)) {
            snprintf(c->err, sizeof c->err, "server %d did not establish its connections", i);
            fprintf(stderr, "refc_new: %s\n", c->err);
            return NULL;
        }
    return c;
}

void refc_free(refc_t *c)
{
    if (!c) return;
    fab_set_write_hook(NULL, NULL);
    for (int i = 0; i < c->n; i++) {
        /* the instances are dropped without dare_server_shutdown (it would pthread_exit);
         * their 64 MiB logs are released here */
        if (c->in[i].dl) {
            void *lg = c->in[i].log ? c->in[i].log() : NULL;
            free(lg);
            dlclose(c->in[i].dl);
        }
        if (c->in[i].fd > 0) close(c->in[i].fd);
    }
    fab_reset();
    free(c->round_commit); free(c->round_end); free(c);
}

static void note_round(refc_t *c)
{
    if (c->leader < 0) return;
    if (c->n_rounds == c->rounds_cap) {
        c->rounds_cap = c->rounds_cap ? c->rounds_cap * 2 : 1024;
        c->round_commit = realloc(c->round_commit, c->rounds_cap * sizeof(uint64_t));
        c->round_end = realloc(c->round_end, c->rounds_cap * sizeof(uint64_t));
    }
    uint64_t *o = offs(&c->in[c->leader]);
    c->round_commit[c->n_rounds] = o[2];
    c->round_end[c->n_rounds] = o[3];
    c->n_rounds++;
}

#define SID_TERM(s) ((s) >> 9)
#define SID_L(s)    (((s) >> 8) & 1)
#define SID_IDX(s)  ((int)((s) & 0xFF))
static int is_candidate(refc_t *c, int i)
{
    uint64_t s = c->in[i].sid();
    return SID_IDX(s) == i && !SID_L(s) && SID_TERM(s) > 0;
}

int refc_leader(const refc_t *c) { return c->leader; }
int refc_group_size(const refc_t *c) { return c->n; }

int refc_elect(refc_t *c, int w)
{
    if (w < 0 || w >= c->n || !c->in[w].alive) return -1;
    if (c->leader >= 0 && c->leader != w) return -1;
    if (c->n == 1) return 0;
    /* step 1: every live server that still follows somebody misses the heartbeat */
    for (int i = 0; i < c->n; i++) {
        if (!c->in[i].alive || fab_port_held(i)) continue;
        for (int k = 0; k < 4 && !is_candidate(c, i); k++) call_fire(c, i, T_HB);
        if (!is_candidate(c, i)) { snprintf(c->err, sizeof c->err, "server %d did not become a candidate", i); return -1; }
    }
    /* step 2: w's election timeout fires first */
    uint64_t t0 = SID_TERM(c->in[w].sid());
    for (int k = 0; k < 4 && SID_TERM(c->in[w].sid()) == t0; k++) call_fire(c, w, T_HB);
    if (SID_TERM(c->in[w].sid()) != t0 + 1) { snprintf(c->err, sizeof c->err, "winner did not start an election"); return -1; }
    /* votes (poll_vote_requests :1526), then the count (poll_vote_count :1327) */
    for (int i = 0; i < c->n; i++) if (i != w) call_poll(c, i);
    call_poll(c, w);
    if (!c->in[w].is_leader()) { snprintf(c->err, sizeof c->err, "server %d did not win", w); return -1; }
    c->leader = w;
    /* the new leader's next pass commits the blank CONFIG entry (:1419) ... */
    call_poll(c, w);
    note_round(c);
    /* ... then its heartbeat timer fires (hb_send_cb :927).  For a peer that died this is the
     * second failed CTRL write after the vote request: PERMANENT_FAILURE, check_failure_count
     * removes it with a CONFIG entry in the following pass (:1189-1227). */
    call_fire(c, w, T_HB);
    for (int i = 0; i < c->n; i++) if (i != w) call_poll(c, i);          /* adopt the leader's SID (:1546) */
    for (int i = 0; i < c->n; i++) if (i != w && c->in[i].alive && !fab_port_held(i)) call_fire(c, i, T_HB);   /* hb_receive_cb */
    uint64_t end0 = offs(&c->in[w])[3];
    call_poll(c, w);
    if (offs(&c->in[w])[3] != end0) note_round(c);
    return 0;
}

int refc_round(refc_t *c, const refc_req_t *reqs, int n, const uint8_t *arena)
{
    if (c->leader < 0) return -1;
    inst_t *L = &c->in[c->leader];
    for (int k = 0; k < n; k++)
        L->submit(reqs[k].type, reqs[k].clt_id, reqs[k].req_id, arena ? arena + reqs[k].payload_off : NULL, reqs[k].len);
    call_poll(c, c->leader);
    note_round(c);
    return 0;
}

int refc_quiesce(refc_t *c)
{
    if (c->leader < 0) return -1;
    for (int it = 0; it < 64; it++) {
        uint64_t before[MAXN][4], pb[MAXN][2];
        for (int i = 0; i < c->n; i++) {
            uint64_t *o = offs(&c->in[i]);
            before[i][0] = o[3]; before[i][1] = o[2]; before[i][2] = o[1]; before[i][3] = o[5];
            uint64_t p[6]; c->in[c->leader].peer(i, p); pb[i][0] = p[0]; pb[i][1] = p[1];
        }
        call_poll(c, c->leader);
        for (int i = 0; i < c->n; i++) if (i != c->leader && !fab_port_held(i)) call_poll(c, i);
        int moved = 0;
        for (int i = 0; i < c->n; i++) {
            uint64_t *o = offs(&c->in[i]);
            moved |= before[i][0] != o[3] || before[i][1] != o[2] || before[i][2] != o[1] || before[i][3] != o[5];
            uint64_t p[6]; c->in[c->leader].peer(i, p);
            moved |= pb[i][0] != p[0] || pb[i][1] != p[1];
        }
        if (!moved) return 0;
    }
    return 1;
}

int refc_tick_prune(refc_t *c)
{
    if (c->leader < 0) return -1;
    refc_quiesce(c);                                 /* trace semantics: see orc_tick_prune */
    uint64_t end0 = offs(&c->in[c->leader])[3];
    if (!call_fire(c, c->leader, T_PRUNE)) return -1;       /* prune_log_cb :1977 */
    int appended = offs(&c->in[c->leader])[3] != end0;
    if (appended) { call_poll(c, c->leader); note_round(c); }
    return appended;
}

int refc_kill(refc_t *c, int r)
{
    if (r < 0 || r >= c->n) return -1;
    c->in[r].alive = 0;
    fab_kill_port(r);
    if (c->leader == r) { c->leader = -1; return 0; }
    if (c->leader >= 0) {
        uint64_t end0 = offs(&c->in[c->leader])[3];
        call_fire(c, c->leader, T_HB);
        call_fire(c, c->leader, T_HB);
        call_poll(c, c->leader);
        if (offs(&c->in[c->leader])[3] != end0) note_round(c);
    }
    return 0;
}

int refc_hold(refc_t *c, int r) { if (r < 0 || r >= c->n) return -1; fab_hold_port(r); return 0; }
int refc_release(refc_t *c, int r) { if (r < 0 || r >= c->n) return -1; fab_release_port(r); return 0; }

/* raw access for tests that want a schedule of their own */
int refc_poll(refc_t *c, int r) { return (r < 0 || r >= c->n) ? -1 : call_poll(c, r); }
int refc_fire(refc_t *c, int r, int which) { return (r < 0 || r >= c->n) ? -1 : call_fire(c, r, which); }

void refc_offsets(refc_t *c, int r, uint64_t out[8]) { memcpy(out, offs(&c->in[r]), 64); }
uint8_t *refc_entries(refc_t *c, int r) { return c->in[r].entries(); }
uint64_t refc_sid(refc_t *c, int r) { return c->in[r].sid(); }
int      refc_prev_head(refc_t *c, int r) { return c->in[r].prev_head(); }
uint64_t refc_highest_rec(refc_t *c, int r) { return c->in[r].highest_rec(); }
uint64_t refc_store_count(refc_t *c, int r) { return c->in[r].store_count(); }
uint64_t refc_apply_count(refc_t *c, int r) { return c->in[r].apply_count(); }
void     refc_record_apply(refc_t *c, int on) { for (int i = 0; i < c->n; i++) c->in[i].record_apply(on); }
const void *refc_apply_log(refc_t *c, int r, uint64_t *n) { return c->in[r].apply_log(n); }
void     refc_cid(refc_t *c, int r, uint64_t out[4]) { c->in[r].cid(out); }
void     refc_peer(refc_t *c, int r, int i, uint64_t out[6]) { c->in[r].peer(i, out); }
int      refc_alive(refc_t *c, int r) { return c->in[r].alive && !c->in[r].exited(); }
uint64_t refc_round_count(const refc_t *c) { return c->n_rounds; }
const uint64_t *refc_round_commit(const refc_t *c) { return c->round_commit; }
const uint64_t *refc_round_end(const refc_t *c) { return c->round_end; }
