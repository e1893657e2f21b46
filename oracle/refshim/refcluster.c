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
#include <signal.h>
#include <execinfo.h>
#include "fabric.h"
#include "refcluster.h"

/* APUS_REF_DEBUG=1: print a backtrace when the reference crashes (it has undefined behaviour of its
 * own on some schedules, e.g. poll_sm_requests dare_server.c:599-655 with SNAPSHOT already set) */
static void crash_bt(int sig)
{
    void *bt[48]; int n = backtrace(bt, 48);
    static const char msg[] = "refcluster: the reference crashed; backtrace:\n";
    if (write(2, msg, sizeof msg - 1) < 0) {}
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL); raise(sig);
}

#define MAXN 13
#define MAXI 16                    /* fabric ports: every server that ever ran, joiners included */
enum { T_INIT = 0, T_PRUNE = 1, T_HB = 2, T_ADJ = 3 };
#define ST_LOG_RECOVERED 0x20

typedef struct {
    void *dl; int fd;
    int alive, busy;
    int slot;                      /* index in the configuration (data.config.idx); -1: a joiner that has no answer yet */
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
    int      (*idx)(void);
    void     (*record_store)(int);
    const void *(*store_stream)(uint64_t *);
    uint32_t (*records_len)(void);
} inst_t;

/* A server instance owns fabric port k (LID k+1) for its whole life; the trace names servers by their
 * SLOT in the configuration.  At start-up slot == instance; a joiner is a new instance (a new machine)
 * that takes over an empty slot or the one the group is extended by. */
struct refc {
    int n;                         /* slots in use (the extended group size) */
    int n_inst;
    uint64_t log_len;
    inst_t in[MAXI];
    int slot_inst[MAXN];           /* slot -> instance that holds it now (-1: nobody ever did) */
    void *img; size_t img_len;     /* the library image joiners are loaded from */
    char cfg_path[256], log_dir[256];
    int leader;
    int record_store_on;
    uint64_t n_rounds, rounds_cap, *round_commit, *round_end;
    char err[256];
};

static uint64_t *offs(inst_t *t) { return (uint64_t *)t->log(); }   /* head apply commit end tail old_end old_commit len */

/* the instance that holds slot r (NULL: none) */
static inst_t *S(refc_t *c, int r)
{
    if (r < 0 || r >= c->n || c->slot_inst[r] < 0) return NULL;
    inst_t *t = &c->in[c->slot_inst[r]];
    /* a server that shut itself down ("Somebody removed me... bye bye", update_cid dare_server.c:2216) has
     * freed its log and control data (free_server_data :333): nothing of it can be looked at any more */
    if (t->exited && t->exited()) return NULL;
    return t;
}
static int port_of(refc_t *c, int r) { return (r < 0 || r >= c->n) ? -1 : c->slot_inst[r]; }

static int call_poll(refc_t *c, int k)
{
    inst_t *t = &c->in[k];
    if (!t->alive || t->busy || t->exited()) return 0;
    t->busy = 1;
    int prev = fab_enter(k);
    int r = t->poll();
    fab_leave(prev);
    t->busy = 0;
    return r;
}
static int call_fire(refc_t *c, int k, int which)
{
    inst_t *t = &c->in[k];
    if (!t->alive || t->busy || t->exited()) return 0;
    t->busy = 1;
    int prev = fab_enter(k);
    int r = t->fire(which);
    fab_leave(prev);
    t->busy = 0;
    return r;
}
/* the leader may shut itself down in any pass ("Not enough connections... bye bye", dare_server.c:1212) */
static int leader_gone(refc_t *c)
{
    if (c->leader >= 0 && !S(c, c->leader)) { snprintf(c->err, sizeof c->err, "the leader (server %d) shut itself down", c->leader); c->leader = -1; }
    return c->leader < 0;
}
static int poll_slot(refc_t *c, int r) { int k = port_of(c, r); return k < 0 ? 0 : call_poll(c, k); }
static int fire_slot(refc_t *c, int r, int which) { int k = port_of(c, r); return k < 0 ? 0 : call_fire(c, k, which); }
static int slot_up(refc_t *c, int r) { inst_t *t = S(c, r); return t && t->alive && !fab_port_held(port_of(c, r)); }

static void on_write(void *arg, int from, int to, uint64_t raddr, uint32_t len)
{
    (void)from;
    refc_t *c = arg;
    if (to < 0 || to >= c->n_inst || len != 8) return;
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
    SYM(record_apply); SYM(apply_log); SYM(cid); SYM(peer); SYM(all_connected); SYM(idx);
    SYM(record_store); SYM(store_stream); SYM(records_len);
#undef SYM
    return 0;
}

const char *refc_error(const refc_t *c) { return c->err; }

static int start_instance(refc_t *c, int k, int group_size, int join)
{
    char lp[300] = "";
    if (c->log_dir[0]) snprintf(lp, sizeof lp, "%s/ref_srv%d.log", c->log_dir, k);
    int prev = fab_enter(k);
    int rc = c->in[k].start(k, group_size, join, c->cfg_path, lp, c->log_len);
    fab_leave(prev);
    if (rc) return rc;
    c->in[k].alive = 1;
    return 0;
}

refc_t *refc_new(int n, uint64_t log_len, const char *lib_path, const char *cfg_path, const char *log_dir)
{
    if (n < 1 || n > MAXN) return NULL;
    refc_t *c = calloc(1, sizeof *c);
    c->n = n; c->n_inst = n; c->log_len = log_len; c->leader = -1;
    for (int i = 0; i < MAXN; i++) c->slot_inst[i] = -1;
    snprintf(c->cfg_path, sizeof c->cfg_path, "%s", cfg_path ? cfg_path : "");
    snprintf(c->log_dir, sizeof c->log_dir, "%s", log_dir ? log_dir : "");
    FILE *f = fopen(lib_path, "rb");
    if (!f) { free(c); return NULL; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    void *img = malloc((size_t)sz);
    if (fread(img, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(img); free(c); return NULL; }
    fclose(f);
    c->img = img; c->img_len = (size_t)sz;
    fab_reset();
    fab_set_write_hook(on_write, c);
    if (getenv("APUS_REF_DEBUG")) signal(SIGSEGV, crash_bt);
    for (int i = 0; i < n; i++) {
        if (load_instance(c, i, img, (size_t)sz)) { fprintf(stderr, "refc_new: %s\n", c->err); return NULL; }
        if (start_instance(c, i, n, 0)) { fprintf(stderr, "refc_new: dare_server_init failed on %d\n", i); return NULL; }
        c->in[i].slot = i; c->slot_inst[i] = i;
    }
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
    for (int i = 0; i < c->n_inst; i++) {
        /* the instances are dropped without dare_server_shutdown (it would pthread_exit);
         * their 64 MiB logs are released here */
        if (c->in[i].dl) {
            /* (a server that shut itself down has freed its own, free_server_data dare_server.c:333) */
            void *lg = (c->in[i].log && !(c->in[i].exited && c->in[i].exited())) ? c->in[i].log() : NULL;
            free(lg);
            dlclose(c->in[i].dl);
        }
        if (c->in[i].fd > 0) close(c->in[i].fd);
    }
    fab_reset();
    free(c->img);
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
    uint64_t *o = offs(S(c, c->leader));
    c->round_commit[c->n_rounds] = o[2];
    c->round_end[c->n_rounds] = o[3];
    c->n_rounds++;
}

#define SID_TERM(s) ((s) >> 9)
#define SID_L(s)    (((s) >> 8) & 1)
#define SID_IDX(s)  ((int)((s) & 0xFF))
static int is_candidate(refc_t *c, int r)
{
    uint64_t s = S(c, r)->sid();
    return SID_IDX(s) == r && !SID_L(s) && SID_TERM(s) > 0;
}

int refc_leader(const refc_t *c) { return c->leader; }
int refc_group_size(const refc_t *c) { return c->n; }

int refc_elect(refc_t *c, int w)
{
    if (w < 0 || w >= c->n || !S(c, w) || !S(c, w)->alive) return -1;
    if (c->leader >= 0 && c->leader != w) return -1;
    if (c->n == 1) return 0;
    /* step 1: every live server that still follows somebody misses the heartbeat */
    for (int i = 0; i < c->n; i++) {
        if (!slot_up(c, i)) continue;
        for (int k = 0; k < 4 && !is_candidate(c, i); k++) fire_slot(c, i, T_HB);
        if (!is_candidate(c, i)) { snprintf(c->err, sizeof c->err, "server %d did not become a candidate", i); return -1; }
    }
    /* step 2: w's election timeout fires first */
    uint64_t t0 = SID_TERM(S(c, w)->sid());
    for (int k = 0; k < 4 && SID_TERM(S(c, w)->sid()) == t0; k++) fire_slot(c, w, T_HB);
    if (SID_TERM(S(c, w)->sid()) != t0 + 1) { snprintf(c->err, sizeof c->err, "winner did not start an election"); return -1; }
    /* votes (poll_vote_requests :1526), then the count (poll_vote_count :1327) */
    for (int i = 0; i < c->n; i++) if (i != w) poll_slot(c, i);
    poll_slot(c, w);
    if (!S(c, w)->is_leader()) { snprintf(c->err, sizeof c->err, "server %d did not win", w); return -1; }
    c->leader = w;
    /* the new leader's next pass commits the blank CONFIG entry (:1419) ... */
    poll_slot(c, w);
    note_round(c);
    /* ... then its heartbeat timer fires (hb_send_cb :927).  For a peer that died this is the
     * second failed CTRL write after the vote request: PERMANENT_FAILURE, check_failure_count
     * removes it with a CONFIG entry in the following pass (:1189-1227). */
    fire_slot(c, w, T_HB);
    for (int i = 0; i < c->n; i++) if (i != w) poll_slot(c, i);          /* adopt the leader's SID (:1546) */
    for (int i = 0; i < c->n; i++) if (i != w && slot_up(c, i)) fire_slot(c, i, T_HB);   /* hb_receive_cb */
    uint64_t end0 = offs(S(c, w))[3];
    poll_slot(c, w);
    if (offs(S(c, w))[3] != end0) note_round(c);
    return 0;
}

int refc_round(refc_t *c, const refc_req_t *reqs, int n, const uint8_t *arena)
{
    if (c->leader < 0) return -1;
    inst_t *L = S(c, c->leader);
    for (int k = 0; k < n; k++)
        L->submit(reqs[k].type, reqs[k].clt_id, reqs[k].req_id, arena ? arena + reqs[k].payload_off : NULL, reqs[k].len);
    poll_slot(c, c->leader);
    if (leader_gone(c)) return -9;
    note_round(c);
    return 0;
}

int refc_quiesce(refc_t *c)
{
    if (leader_gone(c)) return -1;
    for (int it = 0; it < 64; it++) {
        uint64_t before[MAXN][4], pb[MAXN][2];
        for (int i = 0; i < c->n; i++) {
            if (!S(c, i)) continue;
            uint64_t *o = offs(S(c, i));
            before[i][0] = o[3]; before[i][1] = o[2]; before[i][2] = o[1]; before[i][3] = o[5];
            uint64_t p[6]; S(c, c->leader)->peer(i, p); pb[i][0] = p[0]; pb[i][1] = p[1];
        }
        poll_slot(c, c->leader);
        if (leader_gone(c)) return -9;
        for (int i = 0; i < c->n; i++) if (i != c->leader && S(c, i) && !fab_port_held(port_of(c, i))) poll_slot(c, i);
        int moved = 0;
        for (int i = 0; i < c->n; i++) {
            if (!S(c, i)) continue;
            uint64_t *o = offs(S(c, i));
            moved |= before[i][0] != o[3] || before[i][1] != o[2] || before[i][2] != o[1] || before[i][3] != o[5];
            uint64_t p[6]; S(c, c->leader)->peer(i, p);
            moved |= pb[i][0] != p[0] || pb[i][1] != p[1];
        }
        if (!moved) return 0;
    }
    return 1;
}

int refc_tick_prune(refc_t *c)
{
    if (leader_gone(c)) return -1;
    if (refc_quiesce(c) < 0) return -9;              /* trace semantics: see orc_tick_prune */
    uint64_t end0 = offs(S(c, c->leader))[3];
    if (!fire_slot(c, c->leader, T_PRUNE)) return -1;       /* prune_log_cb :1977 */
    int appended = offs(S(c, c->leader))[3] != end0;
    if (appended) { poll_slot(c, c->leader); if (leader_gone(c)) return -9; note_round(c); }
    return appended;
}

int refc_kill(refc_t *c, int r)
{
    if (!S(c, r)) return -1;
    S(c, r)->alive = 0;
    fab_kill_port(port_of(c, r));
    if (c->leader == r) { c->leader = -1; return 0; }
    if (c->leader >= 0) {
        uint64_t end0 = offs(S(c, c->leader))[3];
        fire_slot(c, c->leader, T_HB);
        fire_slot(c, c->leader, T_HB);
        poll_slot(c, c->leader);
        if (offs(S(c, c->leader))[3] != end0) note_round(c);
    }
    return 0;
}

int refc_hold(refc_t *c, int r) { if (!S(c, r)) return -1; fab_hold_port(port_of(c, r)); return 0; }
int refc_release(refc_t *c, int r) { if (!S(c, r)) return -1; fab_release_port(port_of(c, r)); return 0; }

/* JOIN(r): a new server (a new machine: its own port and LID) starts with SRV_TYPE_JOIN and walks
 * through the reference's own recovery: JOIN request over UD multicast (join_cluster_cb
 * dare_server.c:445 -> handle_server_join_request dare_ibv_ud.c:973: the leader turns the server's
 * bit on -- or extends the group when no slot is empty -- and logs a CONFIG entry), the reply once
 * that entry is applied (:1863 / :1895), RC_SYN / SYNACK, the replicated vote, the snapshot of a
 * follower (poll_sm_requests :599, rc_recover_sm dare_ibv_rc.c:597), the log between the leader's
 * head and a server's end (rc_recover_log :726), then server_to_follower + vote ACK and the leader's
 * log adjustment.  The joiner's timer is fired whenever a sweep over all servers left it armed and
 * moved nothing else -- i.e. every period is long against a polling pass.  `r` is the slot the trace
 * expects the leader to hand out; -1 on a mismatch. */
int refc_join(refc_t *c, int r)
{
    if (c->leader < 0 || c->n_inst >= MAXI || r < 0 || r >= MAXN) return -1;
    int k = c->n_inst;
    if (load_instance(c, k, c->img, c->img_len)) return -1;
    c->n_inst++;
    c->in[k].slot = -1;
    uint64_t cid[4]; S(c, c->leader)->cid(cid);
    int size0 = (int)(cid[1] & 0xFF);
    if (start_instance(c, k, size0, 1)) { snprintf(c->err, sizeof c->err, "joiner: dare_server_init failed"); return -1; }
    c->in[k].record_apply(1);
    c->in[k].record_store(c->record_store_on);
    call_fire(c, k, T_INIT);                                       /* init_network_cb -> join_cluster_cb armed */
    uint64_t end0 = offs(S(c, c->leader))[3];
    for (int sweep = 0; sweep < 400; sweep++) {
        if (c->in[k].state() & ST_LOG_RECOVERED) break;
        /* a server whose configuration was one pass behind when the joiner's RC_SYN came drops it
         * ("Configuration inconsistency; it will be solved later", dare_ibv_ud.c handle_rc_syn): solved by
         * the members' periodic RC-info timer (update_rc_info_cb, dare_server.c:501) -- fired here once the
         * joiner has been waiting for a while */
        if (sweep >= 8 && (sweep & 3) == 0)
            for (int i = 0; i < c->n_inst; i++) if (i != k && c->in[i].alive && !fab_port_held(i)) call_fire(c, i, T_INIT);
        call_fire(c, k, T_INIT);
        for (int pass = 0; pass < 4; pass++) {
            poll_slot(c, c->leader);
            for (int i = 0; i < c->n_inst; i++) if (i != k && i != port_of(c, c->leader) && !fab_port_held(i)) call_poll(c, i);
            call_poll(c, k);
        }
        if (c->in[k].exited()) { snprintf(c->err, sizeof c->err, "joiner shut down"); return -1; }
    }
    if (!(c->in[k].state() & ST_LOG_RECOVERED)) { snprintf(c->err, sizeof c->err, "joiner did not recover its log (state %llx)", (unsigned long long)c->in[k].state()); return -1; }
    int slot = c->in[k].idx();
    if (slot != r) { snprintf(c->err, sizeof c->err, "leader handed out slot %d, trace expected %d", slot, r); return -1; }
    if (slot >= c->n) c->n = slot + 1;
    c->in[k].slot = slot;
    c->slot_inst[slot] = k;
    if (offs(S(c, c->leader))[3] != end0) note_round(c);
    return 0;
}

/* raw access for tests that want a schedule of their own */
int refc_poll(refc_t *c, int r) { return S(c, r) ? poll_slot(c, r) : -1; }
int refc_fire(refc_t *c, int r, int which) { return S(c, r) ? fire_slot(c, r, which) : -1; }

void refc_offsets(refc_t *c, int r, uint64_t out[8]) { memcpy(out, offs(S(c, r)), 64); }
uint8_t *refc_entries(refc_t *c, int r) { return S(c, r)->entries(); }
uint64_t refc_sid(refc_t *c, int r) { return S(c, r)->sid(); }
int      refc_prev_head(refc_t *c, int r) { return S(c, r)->prev_head(); }
uint64_t refc_highest_rec(refc_t *c, int r) { return S(c, r)->highest_rec(); }
uint64_t refc_store_count(refc_t *c, int r) { return S(c, r)->store_count(); }
uint64_t refc_apply_count(refc_t *c, int r) { return S(c, r)->apply_count(); }
void     refc_record_apply(refc_t *c, int on) { for (int i = 0; i < c->n_inst; i++) c->in[i].record_apply(on); }
void     refc_record_store(refc_t *c, int on) { c->record_store_on = on; for (int i = 0; i < c->n_inst; i++) c->in[i].record_store(on); }
const void *refc_store_stream(refc_t *c, int r, uint64_t *n) { return S(c, r)->store_stream(n); }
uint32_t refc_records_len(refc_t *c, int r) { return S(c, r)->records_len(); }
const void *refc_apply_log(refc_t *c, int r, uint64_t *n) { return S(c, r)->apply_log(n); }
void     refc_cid(refc_t *c, int r, uint64_t out[4]) { S(c, r)->cid(out); }
void     refc_peer(refc_t *c, int r, int i, uint64_t out[6]) { S(c, r)->peer(i, out); }
int      refc_alive(refc_t *c, int r) { return S(c, r) && S(c, r)->alive; }
int      refc_gone(refc_t *c, int r) { return r >= 0 && r < c->n && c->slot_inst[r] >= 0 && !S(c, r); }
uint64_t refc_state(refc_t *c, int r) { return S(c, r) ? S(c, r)->state() : 0; }
uint64_t refc_round_count(const refc_t *c) { return c->n_rounds; }
const uint64_t *refc_round_commit(const refc_t *c) { return c->round_commit; }
const uint64_t *refc_round_end(const refc_t *c) { return c->round_end; }
