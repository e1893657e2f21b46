/*
 * apus_proxy.c -- host side of the drop-in: the reference's proxy and SMR-core
 * API surfaces (include/apus_smr.h) implemented as thin C over the GPU engine
 * (include/apus_gpu.h).  No consensus arithmetic happens here: requests are
 * admitted (ids assigned, bytes copied into the submission queue) and the
 * engine's apply stream is turned back into the reference's upcalls, in log order,
 * on the one DARE thread.
 *
 * Mirrors, with the same names and blocking behaviour:
 *   proxy_init / proxy_on_read / proxy_on_accept / proxy_on_close
 *                       /root/reference/src/proxy/proxy.c:441,230,241,252
 *   leader_handle_submit_req (admission + spin until applied)   proxy.c:108-161
 *   dare_server_init / polling()          src/dare/dare_server.c:173-250, 1012-1125
 *   is_leader / get_node_id               src/dare/dare_server.c:2299-2307
 *   dare_ib_poll_tailq / write_remote_logs / send_entries_reply / get_remote_apply_offsets
 *                                          src/include/dare/dare_ibv.h:154,176-178
 *
 * Configuration keeps the reference's environment variables (proxy.c:33-57):
 * server_idx, group_size, server_type, dare_log_file, config_path; the libconfig
 * file keys it needs (port, ip_address, req_log, db_name) are read with a tiny
 * key=value scanner (the file format of target/nodes.local.cfg).
 * Extra knobs: APUS_GPU_DEVICE, APUS_GPU_LOG_LEN, APUS_PRUNE_PERIOD_MS.
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <pthread.h>
#include <signal.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/queue.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#include "apus_gpu.h"
#include "apus_smr.h"

/* ------------------------------------------------------------------------- */
/* submission queue: fixed ring of admitted requests + a payload arena.       */
#define Q_CAP        4096u                 /* == the engine's live batch limit */
#define Q_ARENA      (8u << 20)

typedef struct {
    apus_req_t reqs[Q_CAP];
    uint8_t   *arena;                      /* Q_ARENA bytes */
    uint32_t   n;
    uint64_t   arena_used;
    pthread_spinlock_t lock;               /* tailq_lock, message.h:22 */
} subq_t;

static subq_t g_q;
static volatile int g_admission_closed;          /* the leader is between two runs (leader_pause): no request is admitted */
static pthread_once_t g_q_once = PTHREAD_ONCE_INIT;

extern pthread_spinlock_t tailq_lock;          /* message.h:22 -- defined below, with the reference's queue */
static void q_init(void)
{
    memset(&g_q, 0, sizeof g_q);
    g_q.arena = malloc(Q_ARENA);
    pthread_spin_init(&g_q.lock, PTHREAD_PROCESS_PRIVATE);
}
/* `tailq_lock` is usable from the moment the library is loaded (the reference's proxy_init initialises it again,
 * proxy.c:494, before any thread can hold it; it is never re-initialised later) */
__attribute__((constructor)) static void tailq_lock_init(void) { pthread_spin_init(&tailq_lock, PTHREAD_PROCESS_PRIVATE); }

/* caller holds the lock */
static int q_push_locked(uint8_t type, uint16_t connection_id, uint64_t req_id, const void *buf, uint16_t len)
{
    const uint64_t need = ((uint64_t)len + 15) & ~15ull;
    if (g_q.n == Q_CAP || g_q.arena_used + need > Q_ARENA) return -1;
    apus_req_t *q = &g_q.reqs[g_q.n++];
    memset(q, 0, sizeof *q);
    q->req_id = req_id;
    q->payload_off = g_q.arena_used;
    q->clt_id = connection_id;
    q->len = len;
    q->type = type;
    if (len) memcpy(g_q.arena + g_q.arena_used, buf, len);
    g_q.arena_used += need;
    return 0;
}

int apus_tailq_push(uint8_t type, uint16_t connection_id, uint64_t req_id, const void *buf, uint16_t len)
{
    pthread_once(&g_q_once, q_init);
    pthread_spin_lock(&g_q.lock);
    int rc = q_push_locked(type, connection_id, req_id, buf, len);
    pthread_spin_unlock(&g_q.lock);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* The reference's submission queue under its own names (src/include/dare/message.h:5-22): `tailhead` and
 * `tailq_lock` are the two globals its proxy.c fills (proxy.c:114-158) and its DARE thread drains
 * (get_tailq_message, dare_ibv_ud.c:780-790).  They are defined and exported here with the reference's layout, so
 * the reference's OWN proxy.c links against this library without a source change (INTEGRATION.md option B): whatever
 * it queues is taken over by this library's DARE thread (apus_tailq_drain) and freed like the reference frees it. */
struct tailq_cmd_t { uint16_t len; uint8_t cmd[87380]; };
typedef struct tailq_cmd_t tailq_cmd_t;
struct tailq_entry_t {
    uint8_t type;
    uint16_t connection_id;
    uint64_t req_id;
    tailq_cmd_t cmd;
    TAILQ_ENTRY(tailq_entry_t) entries;
};
typedef struct tailq_entry_t tailq_entry_t;
TAILQ_HEAD(apus_tailq_head_t, tailq_entry_t) tailhead = TAILQ_HEAD_INITIALIZER(tailhead);
pthread_spinlock_t tailq_lock;
_Static_assert(sizeof(tailq_entry_t) == 87416, "tailq_entry_t must keep the reference's layout (SURVEY.md section 10)");
static int submit_one(uint8_t type, uint16_t connection_id, uint64_t req_id, const void *buf, uint16_t len);

/* get_tailq_message (dare_ibv_ud.c:780-790): everything the reference's proxy queued, in order -> number taken */
int apus_tailq_drain(void)
{
    if (TAILQ_EMPTY(&tailhead)) return 0;
    int n = 0;
    pthread_spin_lock(&tailq_lock);
    while (!TAILQ_EMPTY(&tailhead)) {
        tailq_entry_t *n3 = TAILQ_FIRST(&tailhead);
        submit_one(n3->type, n3->connection_id, n3->req_id, n3->cmd.cmd, n3->cmd.len);
        TAILQ_REMOVE(&tailhead, n3, entries);
        free(n3);
        n++;
    }
    pthread_spin_unlock(&tailq_lock);
    return n;
}

/* ------------------------------------------------------------------------- */
/* SMR core state (the reference keeps it in the global `data`, dare_server.c:69) */
typedef struct { int64_t pid; uint64_t start; } g_stamp_t;      /* a process: pid + its start time (field 22 of /proc/<pid>/stat) */
typedef struct {
    apus_engine_t *eng;
    dare_server_input_t in;
    uint32_t group_size, idx, leader;
    uint64_t term;
    volatile int running, terminate, ready;
    volatile int failed;                       /* the log is full: admission is closed (fail-stop) */
    int live_persist;                          /* 1: the single-workgroup persistent consensus kernel is the live loop (APUS_LIVE_MODE=persist) */
    int live_replica;                          /* 1: the replica kernels are the live loop (default): application threads reserve and publish
                                                * their requests in the pinned multi-producer ring themselves, every replica runs its own workgroups */
    const volatile uint64_t *dev_hr;           /* proxy->highest_rec as the resident kernel publishes it (pinned host memory) */
    uint64_t upcalled;                         /* update_state upcalls delivered to a caller's own callback */
    /* one PROCESS per server (APUS_GROUP_DIR): this process hosts replica `idx` only, the others are peer-mapped */
    int group;                                 /* 1: group mode */
    char group_dir[256];
    uint32_t capacity;                         /* replicas that can exist (APUS_GROUP_CAPACITY >= group_size: room for machines that JOIN) */
    uint32_t mapped_mask;                      /* peers whose replica is mapped here */
    uint64_t seq;                              /* the newest announcement (cfg_<seq>) this server has acted on */
    uint64_t epoch;                            /* config.cid.epoch */
    uint32_t machines;                         /* LIDs handed out (a joiner is a new machine) */
    int pending_transfer;                      /* a joined machine: its application has not been given the recovered log yet (it may not be listening yet) */
    uint64_t xfer_head, xfer_apply;            /* ... the part of the log that stands for its snapshot: [head, apply) as recovered */
    volatile int serving;                      /* group mode: this server leads AND its run is resident -- the hooks admit requests (a client that
                                                * reaches a server between its election and the start of its workgroups is not replicated yet) */
    g_stamp_t peer_stamp[APUS_MAX_SERVERS];    /* every server's process of THIS run */
    uint32_t alive_mask, bitmask;              /* servers whose process answers; cid.bitmask as this process knows it */
    uint64_t replayed;                         /* follower: apply-stream slots handed to do_action so far */
    pthread_t thread;
    uint64_t applied_slot[APUS_MAX_SERVERS];   /* next apply-stream slot to hand to the upcalls */
    double prune_period_s;
    /* group mode, a leader that lives but does not answer (a partition, a process that hangs): the leader's process writes a
     * heartbeat word into the group directory from a thread of its own, a follower that sees none for hb_timeout_s starts the
     * election although the leader's process exists (hb_receive_cb / the election timeout, dare_server.c:822-920, 1237-1250).
     * Off (0) unless APUS_HB_TIMEOUT_MS is set: process liveness alone decides, as in rounds 3-5. */
    double hb_timeout_s, hb_seen_t;
    uint64_t hb_seen;
    pthread_t hb_thread; int hb_running;
    /* drained batch kept between poll_tailq and write_remote_logs */
    apus_req_t *batch; uint8_t *batch_arena; uint32_t batch_n; uint64_t batch_bytes;
    FILE *log;
} smr_t;

static smr_t g_smr;

/* one request of the reference's queue into this library's admission path */
static int submit_one(uint8_t type, uint16_t connection_id, uint64_t req_id, const void *buf, uint16_t len)
{
    smr_t *s = &g_smr;
    if (s->live_replica && s->eng && s->ready > 0) {
        uint64_t slot = 0; void *dst = NULL;
        if (apus_gpu_rep_reserve(s->eng, len, &slot, &dst)) { s->failed = 1; return -1; }
        if (len) memcpy(dst, buf, len);
        return apus_gpu_rep_publish(s->eng, slot, dst, req_id, connection_id, type, len);
    }
    return apus_tailq_push(type, connection_id, req_id, buf, len);
}

int is_leader(void) { return g_smr.ready > 0 && g_smr.leader == g_smr.idx && (!g_smr.group || g_smr.serving); }        /* dare_server.c:2299 */
uint8_t get_node_id(void) { return (uint8_t)g_smr.idx; }                         /* dare_server.c:2304 */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

/* dare_ib_poll_tailq -> get_tailq_message (dare_ibv_ud.c:780-790): everything that
 * queued up becomes log entries, under the same lock the submitters take */
void dare_ib_poll_tailq(void)
{
    smr_t *s = &g_smr;
    if (!s->eng || s->batch_n) return;
    pthread_once(&g_q_once, q_init);
    apus_tailq_drain();                            /* what the reference's own proxy.c queued (message.h:20-22) */
    pthread_spin_lock(&g_q.lock);
    if (g_q.n) {
        memcpy(s->batch, g_q.reqs, sizeof(apus_req_t) * g_q.n);
        memcpy(s->batch_arena, g_q.arena, g_q.arena_used);
        s->batch_n = g_q.n;
        s->batch_bytes = g_q.arena_used;
        g_q.n = 0;
        g_q.arena_used = 0;
    }
    pthread_spin_unlock(&g_q.lock);
    if (s->batch_n && s->live_persist) {
        /* straight into the pinned command ring of the persistent kernel: it appends, replicates,
         * aggregates the ACKs, commits, applies and bumps highest_rec -- no launch, no read-back */
        int rc = -1;
        for (int attempt = 0; attempt < 4 && rc == -1 && !s->terminate; attempt++)     /* (-1: the ring stayed full for seconds -- try again before giving up) */
            rc = apus_gpu_persist_submit(s->eng, s->batch, s->batch_n, s->batch_arena, s->batch_bytes);
        if (rc) { fprintf(stderr, "[apus] persist_submit failed rc=%d: admission closed, the hooks are inert from here on\n", rc); s->failed = 1; }
        s->batch_n = 0;
        return;
    }
    if (s->batch_n) {
        int rc = apus_gpu_append_live(s->eng, s->batch, s->batch_n, s->batch_arena, s->batch_bytes);
        if (rc == APUS_E_FULL) {
            /* log_append_entry refused the requests (dare_log.h:492-495): they are dropped; nobody
             * waiting for them can be released any more */
            fprintf(stderr, "[apus] the log is full: requests dropped, admission closed\n");
            s->failed = 1; s->batch_n = 0;
        } else if (rc) fprintf(stderr, "[apus] append_live failed rc=%d\n", rc);
    }
}

/* dare_ib_write_remote_logs -> rc_write_remote_logs (dare_ibv_rc.c:1870): replicate,
 * aggregate ACKs, advance commit; with wait_for_commit the reference loops until the
 * commit happened, here the stream is drained.  Returns 0 like the reference. */
int dare_ib_write_remote_logs(int wait_for_commit)
{
    smr_t *s = &g_smr;
    if (!s->eng) return 1;
    int rc = apus_gpu_commit_live(s->eng, wait_for_commit);
    s->batch_n = 0;
    return rc ? 1 : 0;
}

/* dare_ib_send_entries_reply -> rc_send_entries_reply (dare_ibv_rc.c:1828): the
 * follower-side ACK.  Followers hosted by this process are driven by the engine's
 * own persist/ACK kernel inside write_remote_logs, so nothing is left to post. */
int dare_ib_send_entries_reply(uint8_t idx)
{
    (void)idx;
    return g_smr.eng ? 0 : 1;
}

/* dare_ib_get_remote_apply_offsets -> rc_get_remote_apply_offsets (dare_ibv_rc.c:1970);
 * the gather is part of the prune tick kernel */
int dare_ib_get_remote_apply_offsets(void)
{
    return g_smr.eng ? 0 : 1;
}

/* apply_committed_entries' upcalls (dare_server.c:1941-1955), in log order */
static void deliver_upcalls(smr_t *s)
{
    uint64_t cnt[8];
    const uint32_t r = s->idx;
    if (apus_gpu_counters(s->eng, r, cnt)) return;
    const uint64_t n_apply = cnt[3];
    static apus_apply_t recs[1024];
    while (s->applied_slot[r] < n_apply) {
        uint64_t n = n_apply - s->applied_slot[r];
        if (n > 1024) n = 1024;
        if (apus_gpu_apply_records(s->eng, r, s->applied_slot[r], n, recs)) return;
        for (uint64_t i = 0; i < n; i++) {
            if (recs[i].kind == 1) {
                if (s->in.update_state) s->in.update_state(s->in.up_para);
            } else if (recs[i].kind == 2 && s->in.do_action) {
                static uint8_t payload[65536 + 64];
                if (recs[i].len)
                    apus_gpu_read_ring(s->eng, r, recs[i].off + 50, recs[i].len, payload);
                s->in.do_action(recs[i].clt_id, recs[i].type, recs[i].len, payload, s->in.up_para);
            }
        }
        s->applied_slot[r] += n;
    }
}

static void on_sigint(int sig) { (void)sig; g_smr.terminate = 1; }   /* int_handler, dare_server.c:2310 */

void dare_server_shutdown(void)
{
    g_smr.terminate = 1;
}

static void proxy_mirror_highest_rec(uint64_t v);   /* proxy->highest_rec follows the device's word */
static void update_highest_rec(void *arg);

/* apply_committed_entries on the leader = one update_state upcall per applied client entry (dare_server.c:1952-1955).
 * The proxy's own callback only counts (proxy.c:263-267): the word the device publishes is mirrored instead.  A
 * caller's own callback (INTEGRATION.md option B) is called once per applied entry, in order, on this thread. */
static void leader_upcalls(uint64_t hr)
{
    smr_t *s = &g_smr;
    proxy_mirror_highest_rec(hr);
    if (s->in.update_state && s->in.update_state != update_highest_rec)
        while (s->upcalled < hr) { s->in.update_state(s->in.up_para); s->upcalled++; }
}

/* APUS_PROXY_DUMP=<file>: what every replica holds when the server stops, for the end-to-end test of
 * an application under LD_PRELOAD (the process is the application's, nobody can ask it afterwards):
 * one text line per replica "replica i head apply commit end tail len highest_rec status", then the
 * ring bytes [0, len) of each, raw. */
static void dump_replicas(smr_t *s, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return;
    if (!s->group || s->leader == s->idx)          /* (only the process that leads may drive the group's control blocks) */
        apus_gpu_quiesce(s->eng);                  /* followers learn the newest commit (the lazy R4) */
    apus_gpu_sync(s->eng);
    const uint32_t st = apus_gpu_status(s->eng);
    uint64_t len = 0;
    const uint32_t nrep = s->group ? s->capacity : s->group_size;      /* (group mode: every place a machine can hold; a place nobody is mapped in reads as zeros) */
    for (uint32_t i = 0; i < nrep; i++) {
        uint64_t o[8] = {0}, c[8] = {0};
        apus_gpu_offsets(s->eng, i, o);
        apus_gpu_counters(s->eng, i, c);
        if (o[7]) len = o[7];
        fprintf(f, "replica %u head %llu apply %llu commit %llu end %llu tail %llu len %llu highest_rec %llu status %u\n", i,
                (unsigned long long)o[0], (unsigned long long)o[1], (unsigned long long)o[2], (unsigned long long)o[3],
                (unsigned long long)o[4], (unsigned long long)o[7], (unsigned long long)c[6], st);
    }
    fprintf(f, "rings\n");
    uint8_t *buf = malloc(len ? len : 1);
    for (uint32_t i = 0; buf && i < nrep; i++) {
        if (apus_gpu_read_ring(s->eng, i, 0, len, buf)) memset(buf, 0xEE, len);
        fwrite(buf, 1, len, f);
    }
    free(buf);
    fclose(f);
}

/* ------------------------------------------------------------------------- */
/* One process per server (benchmarks/run.sh starts one redis + interposer per node; here: per GPU).  The control plane
 * between the processes is a directory (APUS_GROUP_DIR): what the reference exchanges in UD messages -- MR addresses and
 * rkeys (RC_SYN / SYNACK, dare_ibv_ud.c:1098-1380), vote requests, "I lead term t with this configuration", JOIN
 * requests (dare_ibv_ud.c:973-1068) -- are small files written with rename(); any other transport would do
 * (INTEGRATION.md section 6).  The data plane is the replica kernels: every process runs the workgroups of the replica
 * it hosts.
 *
 *   replica_<i>.ipc    server i's hello: the HIP-IPC handles of its replica (RC_SYN)
 *   cfg_<seq>          an ANNOUNCEMENT by whoever leads: {seq, term, leader, bitmask, size, epoch} -- one per run of the
 *                      replica kernels (start-up, fail-over, removal of a dead follower, JOIN); cfg_latest names the newest
 *   ready_<seq>_<i>    follower i's workgroups for announcement seq are resident
 *   parked_<term>_<i>  election of `term`: server i has parked and this is its last entry (term, idx) -- its vote request
 *   join_<i>           a new machine asks for slot i (server_type=join)
 *
 * Election schedule: start-up = server 0 (the start-up ELECT of the pinned traces).  After the leader's PROCESS is gone:
 * the live server whose log is the most up to date -- last entry's (term, idx), ties to the lowest index -- is the one
 * whose timeout "fires first".  (The reference draws random timeouts, dare_server.c:1237-1250, and a candidate whose log
 * is behind a voter's is refused, :1661-1673, and tries again later; a trace's ELECT(w) names the winner.  Round 4 took
 * the lowest live index: with a leader killed MID-FLIGHT the followers' logs differ, the lowest index may be the shorter
 * one, its election fails on the device -- and the group stopped.  The rule here is the schedule in which the first
 * candidate is one that can win.) */
typedef struct { uint64_t seq, term; uint32_t leader, bitmask, size, kind; uint64_t epoch; } g_cfg_t;
enum { G_KIND_START = 0, G_KIND_FAILOVER, G_KIND_REMOVE, G_KIND_JOIN };

static void g_path(smr_t *s, char *out, size_t cap, const char *fmt, ...)
{
    va_list ap;
    int n = snprintf(out, cap, "%s/", s->group_dir);
    va_start(ap, fmt);
    vsnprintf(out + n, cap - (size_t)n, fmt, ap);
    va_end(ap);
}
/* Who wrote a file: every control file ends with its writer's stamp {pid, start time of that process (field 22 of
 * /proc/<pid>/stat)}.  A directory that still holds the files of an EARLIER run (INTEGRATION.md used a fixed /dev/shm path)
 * satisfies nobody: a hello file counts only while the process it names is alive with that start time, every other file only
 * when it carries the stamp its expected writer's hello gave (stale replica_N.ipc / ready_* / leader_* files made a
 * restarted server import dead IPC handles and follow a leader of the run before). */
static int proc_look(pid_t pid, char *state, uint64_t *start)
{
    char pth[64], buf[1024];
    snprintf(pth, sizeof pth, "/proc/%d/stat", (int)pid);
    FILE *f = fopen(pth, "r");
    if (!f) return -1;
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    char *p = strrchr(buf, ')');                         /* (the command name may hold anything, blanks and brackets included) */
    if (!p || !p[1] || !p[2]) return -1;
    *state = p[2];
    p += 2;                                              /* field 3 (state) */
    for (int field = 3; field < 22 && p; field++) { p = strchr(p, ' '); if (p) p++; }
    if (!p) return -1;
    *start = strtoull(p, NULL, 10);
    return 0;
}
static g_stamp_t g_my_stamp(void)
{
    g_stamp_t st = { (int64_t)getpid(), 0 };
    char c;
    proc_look(getpid(), &c, &st.start);
    return st;
}
/* a process that exists, is not a zombie (kill(pid, 0) still succeeds on an unreaped one: a launcher that never wait()s would
 * hide a dead leader for ever) and is the SAME process the stamp was taken from */
static int g_stamp_alive(const g_stamp_t *st)
{
    char state = 0; uint64_t start = 0;
    if (st->pid <= 0 || proc_look((pid_t)st->pid, &state, &start)) return 0;
    return state != 'Z' && state != 'X' && start == st->start;
}
static int g_write(smr_t *s, const void *buf, size_t len, const char *name)
{
    char tmp[480], dst[400];
    g_path(s, dst, sizeof dst, "%s", name);
    snprintf(tmp, sizeof tmp, "%s.tmp.%d", dst, (int)getpid());
    FILE *f = fopen(tmp, "wb");
    if (!f) return -1;
    const g_stamp_t me = g_my_stamp();
    const int ok = fwrite(buf, 1, len, f) == len && fwrite(&me, 1, sizeof me, f) == sizeof me;
    fclose(f);
    if (!ok || rename(tmp, dst)) { unlink(tmp); return -1; }
    return 0;
}
/* the file's `len` bytes and the stamp behind them */
static int g_read(smr_t *s, void *buf, size_t len, const char *name, g_stamp_t *who)
{
    char src[400];
    g_path(s, src, sizeof src, "%s", name);
    FILE *f = fopen(src, "rb");
    if (!f) return -1;
    g_stamp_t st;
    const int ok = fread(buf, 1, len, f) == len && fread(&st, 1, sizeof st, f) == sizeof st;
    fclose(f);
    if (ok && who) *who = st;
    return ok ? 0 : -1;
}
static int g_same(const g_stamp_t *a, const g_stamp_t *b) { return a->pid == b->pid && a->start == b->start; }
/* `name` as written by server `writer` of this run (its stamp from the hello exchange; writer >= capacity: any live
 * process -- the hello files themselves); 0 when it is there */
static int g_have_from(smr_t *s, const char *name, uint32_t writer, size_t len, void *out)
{
    uint8_t tmp[sizeof(apus_ipc_replica_t) + 64];        /* (the largest file: a vote request = 32 bytes + a hello) */
    g_stamp_t st;
    if (len > sizeof tmp || g_read(s, tmp, len, name, &st)) return -1;
    const int good = writer < s->capacity ? g_same(&st, &s->peer_stamp[writer]) : g_stamp_alive(&st);
    if (!good) return -1;
    if (out) memcpy(out, tmp, len);
    return 0;
}
/* waits for it; gives up when `seconds` are over, at shutdown, or -- writer < capacity -- when the writer's process is gone */
static int g_wait_from(smr_t *s, const char *name, uint32_t writer, size_t len, void *out, double seconds)
{
    const double t0 = now_s();
    for (unsigned i = 0;; i++) {
        if (!g_have_from(s, name, writer, len, out)) return 0;
        if (s->terminate || now_s() - t0 > seconds) return -1;
        if (writer < s->capacity && (i & 15) == 15 && !g_stamp_alive(&s->peer_stamp[writer])) return -1;
        struct timespec ts = {0, 2000000}; nanosleep(&ts, NULL);
    }
}
static int g_alive(smr_t *s, uint32_t i) { return g_stamp_alive(&s->peer_stamp[i]); }

typedef struct { apus_ipc_replica_t ipc; } g_hello_t;
/* a vote request: the server's last entry {term, idx, entry slots, end} and the handles of its replica as it stands behind the
 * election's fence (the harness reads the first 32 bytes: tests/_cluster.py) */
typedef struct { uint64_t last[4]; apus_ipc_replica_t ipc; } g_parked_t;

/* RC_SYN / SYNACK with server i: map the replica its LIVE process exports (dropping the mapping of a former holder of the
 * slot whose process is gone) */
static int group_map_peer(smr_t *s, uint32_t i, double seconds)
{
    char name[64];
    snprintf(name, sizeof name, "replica_%u.ipc", i);
    g_hello_t h;
    g_stamp_t st;
    /* (the hello of a LIVE process: a file an earlier run left behind names a process that is gone, or another one) */
    if (g_wait_from(s, name, s->capacity, sizeof h, &h, seconds)) { fprintf(stderr, "[apus] server %u: server %u never showed up in %s\n", s->idx, i, s->group_dir); return -1; }
    if (g_read(s, &h, sizeof h, name, &st) || h.ipc.replica != i || !g_stamp_alive(&st)) return -1;
    if ((s->mapped_mask >> i) & 1u) {
        if (g_same(&st, &s->peer_stamp[i])) return 0;                         /* mapped already: the same process */
        if (apus_gpu_unmap_replica(s->eng, i)) { fprintf(stderr, "[apus] server %u: cannot drop the mapping of server %u's former process\n", s->idx, i); return -1; }
        s->mapped_mask &= ~(1u << i);
    }
    if (apus_gpu_import_replica(s->eng, &h.ipc)) { fprintf(stderr, "[apus] server %u: cannot map server %u's replica\n", s->idx, i); return -1; }
    s->peer_stamp[i] = st;
    s->mapped_mask |= 1u << i;
    return 0;
}

/* export the replica this process hosts, map the members of `members` */
static int group_connect(smr_t *s, uint32_t members)
{
    g_hello_t me;
    memset(&me, 0, sizeof me);
    if (apus_gpu_export_replica(s->eng, s->idx, &me.ipc)) return -1;
    char name[64];
    snprintf(name, sizeof name, "replica_%u.ipc", s->idx);
    if (g_write(s, &me, sizeof me, name)) return -1;
    s->peer_stamp[s->idx] = g_my_stamp();
    for (uint32_t i = 0; i < s->capacity; i++)
        if (i != s->idx && ((members >> i) & 1u) && group_map_peer(s, i, 120.0)) return -1;
    return 0;
}

static uint32_t rep_grid_env(const char *name, uint32_t dflt) { const char *v = getenv(name); return v ? (uint32_t)atoi(v) : dflt; }

static int g_announce(smr_t *s, uint32_t kind)
{
    char name[64];
    g_cfg_t c = { ++s->seq, s->term, s->leader, s->bitmask, s->group_size, kind, s->epoch };
    snprintf(name, sizeof name, "cfg_%llu", (unsigned long long)c.seq);
    if (g_write(s, &c, sizeof c, name) || g_write(s, &c, sizeof c, "cfg_latest")) return -1;
    return 0;
}

static void proxy_align_cur_rec(uint64_t v);

/* the replica kernels of one announcement: followers first (their workgroups poll their mailboxes), then the leader */
static int group_start_run(smr_t *s)
{
    char name[64];
    const uint32_t na = rep_grid_env("APUS_REP_APPEND", 16), nf = rep_grid_env("APUS_REP_FWORK", 8);
    if (s->leader != s->idx) {
        /* what "applied" means on this server from the first round on: carried out by its application (the first
         * term used to publish the device's apply count alone) */
        apus_gpu_rep_follower_replayed(s->eng, s->idx, s->replayed);
        s->hb_seen_t = 0; s->hb_seen = 0;                    /* (another leader's heartbeat from here on) */
        {   uint64_t cnt[8] = {0};
            apus_gpu_counters(s->eng, s->idx, cnt);
            fprintf(s->log, "[T%lu] following server %u (announcement %llu): %llu entry slots, %llu committed, %llu applied by the device, %llu replayed here\n", (unsigned long)s->term, s->leader,
                    (unsigned long long)s->seq, (unsigned long long)cnt[0], (unsigned long long)cnt[2], (unsigned long long)cnt[3], (unsigned long long)s->replayed);
            fflush(s->log); }
        if (apus_gpu_set_leader(s->eng, s->leader) || apus_gpu_rep_start(s->eng, 24u * 3600u * 1000u, 2000, na, nf)) return -1;
        snprintf(name, sizeof name, "ready_%llu_%u", (unsigned long long)s->seq, s->idx);
        return g_write(s, "1", 1, name);
    }
    for (uint32_t i = 0; i < s->capacity; i++) {
        if (i == s->idx || !((s->alive_mask >> i) & 1u) || !((s->bitmask >> i) & 1u)) continue;
        snprintf(name, sizeof name, "ready_%llu_%u", (unsigned long long)s->seq, i);
        char one;
        if (g_wait_from(s, name, i, 1, &one, 30.0)) fprintf(stderr, "[apus] leader %u: follower %u did not start its workgroups for announcement %llu\n", s->idx, i, (unsigned long long)s->seq);
    }
    if (apus_gpu_rep_start(s->eng, 24u * 3600u * 1000u, 500, na, nf)) return -1;
    {   uint32_t pi[2] = {0, 0};
        apus_gpu_rep_push_info(s->eng, pi);
        if (pi[0] != pi[1]) { fprintf(s->log, "[T%lu] run of announcement %llu: followers %#x are reachable but NOT in step (pushed: %#x)\n", (unsigned long)s->term, (unsigned long long)s->seq, pi[1] & ~pi[0], pi[0]); fflush(s->log); } }
    s->dev_hr = apus_gpu_rep_highest_rec_ptr(s->eng);
    s->upcalled = *s->dev_hr;
    /* a request's place in the order of upcalls starts where the device's count stands: a new leader's applier has counted the
     * entries it found in its log (update_state, dare_server.c:1952-1955) -- with the reference's cur_rec = 0 the first that
     * many requests of the new term would return before they are committed (proxy.c:160 compares the two counters) */
    proxy_align_cur_rec(*s->dev_hr);
    __sync_synchronize();
    s->serving = 1;
    return 0;
}

/* apply_committed_entries' upcalls on a server that does not lead (dare_server.c:1941-1955 -> proxy_do_action,
 * proxy.c:341-439): the apply-stream records of entry slots [s->replayed, upto) replayed into the local application in
 * log order.  any_kind: records the device applied as the LEADER count too (below) */
static void replay_upto(smr_t *s, uint64_t upto, int any_kind)
{
    static apus_apply_t recs[512];
    static uint8_t *bytes;
    if (!bytes) bytes = malloc(65536u + 64u);             /* one record's payload at a time */
    while (s->replayed < upto && !s->failed) {
        uint64_t n = upto - s->replayed;
        if (n > 512) n = 512;
        if (apus_gpu_apply_records(s->eng, s->idx, s->replayed, n, recs)) { s->failed = 1; break; }
        uint64_t done = 0;
        for (uint64_t i = 0; i < n; i++, done++) {
            /* the record must be the one of the slot asked for: a replay that fell a lap of the apply ring behind would be fed
             * another entry's record (the kernel reports min(device apply, host replay) as applied, so the head cannot pass an
             * entry that is not replayed -- this is the check behind that) -- fail-stop, never a wrong upcall */
            if (recs[i].slot != s->replayed + i) {
                fprintf(stderr, "[apus] server %u: apply record of slot %llu reads as slot %llu: the replay fell behind its ring -- stopping\n",
                        s->idx, (unsigned long long)(s->replayed + i), (unsigned long long)recs[i].slot);
                s->failed = 1;
                break;
            }
            if ((recs[i].kind == 2 || (any_kind && recs[i].kind == 1)) && s->in.do_action) {
                if (recs[i].len && apus_gpu_read_ring(s->eng, s->idx, recs[i].off + 50, recs[i].len, bytes)) {
                    fprintf(stderr, "[apus] server %u: cannot read the payload of slot %llu -- stopping\n", s->idx, (unsigned long long)recs[i].slot);
                    s->failed = 1;
                    break;
                }
                s->in.do_action(recs[i].clt_id, recs[i].type, recs[i].len, bytes, s->in.up_para);
            }
        }
        s->replayed += done;
        apus_gpu_rep_follower_replayed(s->eng, s->idx, s->replayed);     /* what "applied" means on this server: carried out by its application */
    }
}
/* what the follower's own kernel has applied since the last look */
static void joiner_state_transfer(smr_t *s);
static int proxy_app_listening(void);
static void follower_upcalls(smr_t *s)
{
    uint64_t pr[4];
    if (s->pending_transfer) {                              /* a joined machine: first the part of the log that stands for its snapshot */
        if (!proxy_app_listening()) return;
        joiner_state_transfer(s);
        s->pending_transfer = 0;
    }
    if (apus_gpu_rep_follower_progress(s->eng, s->idx, pr)) return;
    replay_upto(s, pr[0], 0);
}

/* the leader's heartbeat: a word in the group directory, from a thread of its own (the DARE thread blocks for seconds in a JOIN) */
static void *hb_main(void *arg)
{
    smr_t *s = arg;
    uint64_t n = 0;
    char name[64];
    snprintf(name, sizeof name, "hb_%u", s->idx);
    while (!s->terminate && s->hb_running) {
        if (s->leader == s->idx && !s->failed) { n++; g_write(s, &n, sizeof n, name); }
        struct timespec ts = {0, 20000000}; nanosleep(&ts, NULL);
    }
    return NULL;
}
/* does the leader answer?  Its process exists -- and, when a heartbeat timeout is configured, its heartbeat word has moved within it */
static int leader_answers(smr_t *s)
{
    if (!g_alive(s, s->leader)) return 0;
    if (s->hb_timeout_s <= 0) return 1;
    char name[64];
    uint64_t n = 0;
    snprintf(name, sizeof name, "hb_%u", s->leader);
    const double t = now_s();
    if (s->hb_seen_t == 0) s->hb_seen_t = t;                 /* (the first look at this leader: the timeout starts here) */
    if (!g_have_from(s, name, s->leader, sizeof n, &n) && n != s->hb_seen) { s->hb_seen = n; s->hb_seen_t = t; }
    return t - s->hb_seen_t <= s->hb_timeout_s;
}

/* the leader's process is gone (or does not answer): park, LEAVE what it has mapped, tell the others where this log ends, the most
 * up-to-date survivor wins the next term */
static int group_failover(smr_t *s)
{
    char name[64];
    const uint32_t old_leader = s->leader;
    s->serving = 0;
    apus_gpu_rep_follower_stop(s->eng, s->idx);
    apus_gpu_rep_park(s->eng);
    follower_upcalls(s);
    s->alive_mask &= ~(1u << old_leader);
    for (uint32_t i = 0; i < s->capacity; i++) if (i != s->idx && ((s->alive_mask >> i) & 1u) && !g_alive(s, i)) s->alive_mask &= ~(1u << i);
    const uint64_t term = s->term + 2;
    /* the vote request: this server's last entry (start_election, dare_server.c:1264-1322) -- and, round 6, the RECEIVER'S FENCE
     * of every election, not only of one that follows a death (rc_revoke_log_access, dare_ibv_rc.c:2156-2243: a voter resets the
     * QPs of the leader it leaves, the old leader's WRITEs bounce): this server LEAVES the log ring and the mailbox the old
     * leader has mapped (apus_gpu_fence_replica: the next of its pre-mapped pairs, device copies) before it takes part in the
     * new term, and says where it lives now with its vote request; every member of the new term switches to that pair
     * (apus_gpu_remap_fenced: all pairs have been mapped since the hello, nothing is opened or closed at election time) before
     * anything is voted on, adjusted or replicated.  An old leader that is not dead -- stopped, partitioned, slow -- keeps the
     * old mappings: whatever its resident kernel still pushes lands in memory nobody reads.  (APUS_GROUP_NO_RING_FENCE: the
     * control experiment.) */
    static g_parked_t mine, theirs[APUS_MAX_SERVERS];
    memset(&mine, 0, sizeof mine);
    if (apus_gpu_last_entry(s->eng, s->idx, mine.last)) return -1;
    if (!getenv("APUS_GROUP_NO_RING_FENCE")) {
        if (apus_gpu_fence_replica(s->eng, s->idx, &mine.ipc)) { fprintf(stderr, "[apus] server %u: cannot leave the old leader's ring (fence)\n", s->idx); return -1; }
        fprintf(s->log, "[T%lu] election of term %llu: left the log ring and the mailbox server %u has mapped (fence %u); last entry (term %llu, idx %llu), %llu entry slots, replayed %llu\n",
                (unsigned long)s->term, (unsigned long long)term, old_leader, mine.ipc.fences, (unsigned long long)mine.last[0], (unsigned long long)mine.last[1], (unsigned long long)mine.last[2], (unsigned long long)s->replayed);
        fflush(s->log);
        {   /* (a machine that maps this server from now on -- a joiner -- must find the buffers it has moved to) */
            g_hello_t me; me.ipc = mine.ipc;
            snprintf(name, sizeof name, "replica_%u.ipc", s->idx);
            g_write(s, &me, sizeof me, name);
        }
    } else if (apus_gpu_export_replica(s->eng, s->idx, &mine.ipc)) return -1;
    snprintf(name, sizeof name, "parked_%llu_%u", (unsigned long long)term, s->idx);
    g_write(s, &mine, sizeof mine, name);
    memset(theirs, 0, sizeof theirs);
    theirs[s->idx] = mine;
    /* nothing of the old term may still be running on a survivor when the votes are cast through the mappings: every live
     * member's word is waited for (a process that dies meanwhile, or never answers, is cut off) */
    for (uint32_t i = 0; i < s->capacity; i++) {
        if (i == s->idx || !((s->alive_mask >> i) & 1u) || !((s->bitmask >> i) & 1u)) continue;
        snprintf(name, sizeof name, "parked_%llu_%u", (unsigned long long)term, i);
        if (g_wait_from(s, name, i, sizeof theirs[i], &theirs[i], 20.0)) { s->alive_mask &= ~(1u << i); continue; }
        if (((s->mapped_mask >> i) & 1u) && apus_gpu_remap_fenced(s->eng, &theirs[i].ipc)) {
            fprintf(stderr, "[apus] server %u: cannot map the ring server %u moved to\n", s->idx, i);
            s->alive_mask &= ~(1u << i);
        }
    }
    uint32_t winner = s->capacity;
    for (uint32_t i = 0; i < s->capacity; i++) {
        if (!((s->alive_mask >> i) & 1u) || !((s->bitmask >> i) & 1u)) continue;
        if (winner >= s->capacity || theirs[i].last[0] > theirs[winner].last[0] || (theirs[i].last[0] == theirs[winner].last[0] && theirs[i].last[1] > theirs[winner].last[1])) winner = i;
    }
    if (winner >= s->capacity) return -1;
    fprintf(s->log, "[T%lu] election of term %llu: server %u has the newest log (term %llu, idx %llu)\n", (unsigned long)s->term, (unsigned long long)term, winner,
            (unsigned long long)theirs[winner].last[0], (unsigned long long)theirs[winner].last[1]);
    if (winner == s->idx) {
        uint64_t out[8] = {0};
        const uint32_t live = s->alive_mask & s->bitmask;
        if (apus_gpu_set_reachable(s->eng, live) || apus_gpu_elect(s->eng, winner, live, s->bitmask, out) || !out[0]) {
            fprintf(stderr, "[apus] server %u: no majority for term %llu (%llu votes)\n", s->idx, (unsigned long long)term, (unsigned long long)out[4]);
            return -1;
        }
        const uint32_t dead = s->bitmask & ~live & ~(1u << winner);
        if (apus_gpu_become_leader_ex(s->eng, winner, term, s->bitmask, dead) || apus_gpu_sync(s->eng)) return -1;
        s->bitmask &= ~dead;
        s->term = term; s->leader = winner;
        /* What this server held but had not carried out when it stopped following: the new leader's first pass has committed
         * it (blank CONFIG + removal commit everything in front of them) and its applier has COUNTED it -- the reference's
         * leader applies a client entry with proxy_update_state alone (dare_server.c:1952-1955), so its application never
         * executes the commands it inherits and differs from its followers' from then on.  Here they are replayed into the
         * local application (proxy_do_action) before the first client of the new term is admitted: the log, the offsets and
         * the counters are the reference's, the application is a replica.  (Deviation 3) */
        uint64_t cnt[8];
        if (!apus_gpu_counters(s->eng, s->idx, cnt)) replay_upto(s, cnt[3], 1);
        if (g_announce(s, G_KIND_FAILOVER)) return -1;
    } else {
        g_cfg_t c;
        snprintf(name, sizeof name, "cfg_%llu", (unsigned long long)(s->seq + 1));
        if (g_wait_from(s, name, winner, sizeof c, &c, 60.0)) return -1;     /* (written by the server the same rule makes the winner HERE) */
        s->seq = c.seq; s->term = c.term; s->leader = c.leader; s->bitmask = c.bitmask;
        apus_gpu_set_reachable(s->eng, s->alive_mask & s->bitmask);
        if (!((s->bitmask >> s->idx) & 1u)) return -1;                   /* this server was removed */
    }
    const int rc = group_start_run(s);
    if (!rc && winner == s->idx) { fprintf(s->log, "[T%lu] LEADER\n", (unsigned long)s->term); fflush(s->log); }   /* dare_server.c:1396, grepped by run.sh */
    return rc;
}

/* ---- the leader changes the configuration between two runs: admission is CLOSED meanwhile -- a flag the application's threads
 *      look at under the admission lock and wait for outside it (round 5 held the spin lock itself across the drain, the
 *      mapping of a joiner and the start of the next run: seconds of every application thread spinning at 100 %).
 *      What was reserved before the flag went up is published by its own thread and drained with the run; a run that does
 *      not drain is a failure of this server (slots reserved and never published would leave their callers waiting). ---- */
static void leader_pause(smr_t *s)
{
    pthread_spin_lock(&g_q.lock);
    g_admission_closed = 1;
    pthread_spin_unlock(&g_q.lock);
    if (apus_gpu_rep_drain(s->eng, 5000)) {
        fprintf(stderr, "[apus] leader %u: the run did not drain before the reconfiguration -- the hooks are inert from here on\n", s->idx);
        s->failed = 1;
    }
    leader_upcalls(*s->dev_hr);
    apus_gpu_rep_park(s->eng);
}
static int leader_resume(smr_t *s, uint32_t kind)
{
    int rc = apus_gpu_sync(s->eng) || g_announce(s, kind) || group_start_run(s);
    __sync_synchronize();
    g_admission_closed = 0;
    return rc;
}

/* check_failure_count (dare_server.c:1190-1230): servers whose process is gone are removed from the configuration with a
 * CONFIG entry -- the rounds in between had their majority without them (a dead follower costs its ACK, not the round) */
static int leader_remove_dead(smr_t *s, uint32_t dead)
{
    fprintf(s->log, "[T%lu] removing the servers whose process is gone: mask %#x\n", (unsigned long)s->term, dead);
    leader_pause(s);
    s->alive_mask &= ~dead;
    s->bitmask &= ~dead;
    uint8_t cid[16] = {0};
    memcpy(cid, &s->epoch, 8);
    cid[8] = (uint8_t)s->group_size;
    memcpy(cid + 12, &s->bitmask, 4);
    int rc = apus_gpu_set_reachable(s->eng, s->alive_mask & s->bitmask) || apus_gpu_append_control(s->eng, APUS_CONFIG, cid);
    if (rc) fprintf(stderr, "[apus] leader %u: cannot append the CONFIG entry of the removal\n", s->idx);
    const int rr = leader_resume(s, G_KIND_REMOVE);
    return rc ? rc : rr;
}

/* handle_server_join_request (dare_ibv_ud.c:973-1068) for a machine that asked through join_<r>: map its replica, let the
 * device carry the JOIN out (CONFIG entries incl. the 3-phase extension, the joiner's recovery as a bulk transfer into ITS
 * memory, its first persist / apply passes: apus_gpu_join), announce the new configuration */
static int leader_join(smr_t *s, uint32_t r)
{
    fprintf(s->log, "[T%lu] JOIN request for slot %u\n", (unsigned long)s->term, r);
    leader_pause(s);
    int rc = group_map_peer(s, r, 10.0);
    uint64_t out[4] = {0};
    if (!rc) {
        rc = apus_gpu_join(s->eng, r, (uint16_t)(++s->machines), s->bitmask, s->alive_mask & s->bitmask, out);
        if (rc) fprintf(stderr, "[apus] leader %u: JOIN of slot %u refused (rc %d)\n", s->idx, r, rc);
        if (!rc || (rc == APUS_E_NOANSWER && out[1])) { s->bitmask = (uint32_t)out[0]; s->group_size = (uint32_t)out[1]; s->epoch = out[2]; }
        if (!rc) s->alive_mask |= 1u << r;
    }
    char name[64];
    snprintf(name, sizeof name, "join_%u", r);
    char pth[400];
    g_path(s, pth, sizeof pth, "%s", name);
    unlink(pth);                                           /* (answered, one way or the other) */
    const int rr = leader_resume(s, G_KIND_JOIN);
    return rc ? rc : rr;
}

/* a follower whose kernel has parked although its leader lives: the next announcement (a removal, a JOIN) */
static int follower_next_run(smr_t *s)
{
    char name[64];
    g_cfg_t c;
    snprintf(name, sizeof name, "cfg_%llu", (unsigned long long)(s->seq + 1));
    if (g_have_from(s, name, s->leader, sizeof c, &c)) return 1;            /* not yet */
    {   /* its leader's park word ends the run; a follower the leader no longer counted in (dropped from the push set and from
         * the park set) is asked to leave by its own process */
        uint64_t pr[4] = {0};
        const double t0 = now_s();
        while (!apus_gpu_rep_follower_progress(s->eng, s->idx, pr) && pr[2] == 1 && now_s() - t0 < 5.0) { struct timespec ts = {0, 200000}; nanosleep(&ts, NULL); }
        if (pr[2] == 1) apus_gpu_rep_follower_stop(s->eng, s->idx);
    }
    apus_gpu_rep_park(s->eng);
    follower_upcalls(s);
    s->seq = c.seq; s->term = c.term; s->leader = c.leader; s->bitmask = c.bitmask;
    if (!((s->bitmask >> s->idx) & 1u)) { fprintf(s->log, "[T%lu] this server was removed from the configuration\n", (unsigned long)s->term); return -1; }
    for (uint32_t i = 0; i < s->capacity; i++) {
        if (i == s->idx || !((c.bitmask >> i) & 1u)) continue;
        if (group_map_peer(s, i, 30.0)) return -1;                         /* a machine that joined (or took over a slot) */
        s->alive_mask |= 1u << i;
    }
    s->alive_mask &= c.bitmask | (1u << s->idx);
    if (c.size != s->group_size || c.epoch != s->epoch) {
        if (apus_gpu_set_config(s->eng, c.size, c.epoch)) return -1;
        s->group_size = c.size; s->epoch = c.epoch;
    }
    apus_gpu_set_reachable(s->eng, s->alive_mask & s->bitmask);
    return group_start_run(s);
}

/* What a joined machine's application is given.  The reference ships the donor's BerkeleyDB records (rc_recover_sm,
 * dare_ibv_rc.c:597-705 -> stablestorage_load_records, proxy.c:306-339) -- records that do not hold the commands (SURVEY Q1:
 * the length word read for a SEND record is reply[4..5], the stored bytes are the entry's header), so a joined redis
 * starts EMPTY whatever was written before.  Here the application is brought up to date from the log this server has just
 * recovered: every client entry in [head, apply) is carried out (proxy_do_action) in log order.  What was pruned before
 * the join (in front of head) is not available either way; that is said in the log. */
/* right after the JOIN, before this server's workgroups start: what its log holds as applied is the snapshot's part */
static void joiner_mark(smr_t *s)
{
    uint64_t o[8] = {0}, c[8] = {0};
    if (apus_gpu_offsets(s->eng, s->idx, o) || apus_gpu_counters(s->eng, s->idx, c)) return;
    s->xfer_head = o[0]; s->xfer_apply = o[1];
    s->replayed = c[3];
    s->pending_transfer = s->in.do_action && o[3] != o[7] && o[0] != o[1];
}
static void joiner_state_transfer(smr_t *s)
{
    uint64_t o[8] = {0};
    if (apus_gpu_offsets(s->eng, s->idx, o)) return;
    const uint64_t L = o[7], head = s->xfer_head, apply = s->xfer_apply;
    uint8_t *ring = malloc(L);
    if (!ring || apus_gpu_read_ring(s->eng, s->idx, 0, L, ring)) { free(ring); return; }
    uint64_t off = head, n = 0, first_idx = 0;
    for (uint64_t guard = 0; off != apply && guard < 2 * (L / 64) + 16; guard++) {
        if (L - off < 64) { off = 0; continue; }                                  /* log_get_entry, dare_log.h:300-337 */
        const uint8_t type = ring[off + 26];
        uint16_t len = 0;
        if (type != APUS_NOOP && type != APUS_CONFIG && type != APUS_HEAD) memcpy(&len, ring + off + 48, 2);
        if (L - off < 64u + len) { off = 0; continue; }                           /* log_fit_entry: the entry sits at 0 */
        if (!n) memcpy(&first_idx, ring + off, 8);
        if (type != APUS_NOOP && type != APUS_CONFIG && type != APUS_HEAD) {
            uint16_t clt; memcpy(&clt, ring + off + 24, 2);
            s->in.do_action(clt, type, len, ring + off + 50, s->in.up_para);
        }
        off += 64u + len;
        n++;
    }
    fprintf(s->log, "[T%lu] state transfer: %llu entries of the recovered log replayed into the application (the log starts at idx %llu%s)\n",
            (unsigned long)s->term, (unsigned long long)n, (unsigned long long)first_idx, first_idx > 1 ? ": entries pruned before the join are NOT part of it" : "");
    fflush(s->log);
    free(ring);
}

static void group_loop(smr_t *s)
{
    double last_prune = now_s(), last_look = now_s();
    while (!s->terminate) {
        if (s->leader == s->idx) {
            leader_upcalls(*s->dev_hr);
            apus_tailq_drain();
            if (!s->failed && apus_gpu_rep_full(s->eng)) {
                fprintf(stderr, "[apus] the log is full: requests were dropped, admission is closed, the hooks are inert from here on\n");
                s->failed = 1;
            }
            const double t = now_s();
            if (!s->failed && t - last_prune >= s->prune_period_s) { apus_gpu_rep_prune(s->eng); last_prune = t; }
            if (!s->failed && t - last_look > 0.01) {     /* the heartbeat timer: are the followers there?  does a machine want to join? */
                last_look = t;
                {   /* somebody else leads a newer term (this process was stopped or cut off, the others elected): step down.  The
                     * followers have LEFT what this server has mapped (the receiver's fence): whatever its kernel pushed meanwhile
                     * went nowhere; the hooks are inert from here on, like a removed server's (update_cid, dare_server.c:2216) */
                    char cname[64];
                    g_cfg_t c;
                    snprintf(cname, sizeof cname, "cfg_%llu", (unsigned long long)(s->seq + 1));
                    if (!g_have_from(s, cname, s->capacity, sizeof c, &c) && c.leader != s->idx && c.term > s->term) {
                        fprintf(s->log, "[T%lu] deposed: server %u leads term %llu -- stepping down\n", (unsigned long)s->term, c.leader, (unsigned long long)c.term);
                        fflush(s->log);
                        s->serving = 0;
                        apus_gpu_rep_park(s->eng);
                        s->failed = 1; s->leader = s->capacity;
                        continue;
                    }
                }
                uint32_t dead = 0;
                for (uint32_t i = 0; i < s->capacity; i++)
                    if (i != s->idx && ((s->alive_mask >> i) & 1u) && ((s->bitmask >> i) & 1u) && !g_alive(s, i)) dead |= 1u << i;
                if (dead && leader_remove_dead(s, dead)) { s->failed = 1; continue; }
                uint32_t r = s->group_size;                /* dare_ibv_ud.c:995-1021: the lowest empty place, else the group grows */
                for (int i = (int)s->group_size - 1; i >= 0; i--) if (!((s->bitmask >> i) & 1u)) r = (uint32_t)i;
                if (r < s->capacity) {
                    char name[64], one;
                    snprintf(name, sizeof name, "join_%u", r);
                    if (!g_have_from(s, name, s->capacity, 1, &one)) leader_join(s, r);
                }
            }
            struct timespec ts = {0, 100000}; nanosleep(&ts, NULL);
            continue;
        }
        follower_upcalls(s);
        const double t = now_s();
        if (t - last_look > 0.01) {                       /* the heartbeat timer (hb_period, nodes.local.cfg): is the leader there? */
            last_look = t;
            if (!leader_answers(s)) {
                fprintf(s->log, "[T%lu] the leader p%u is gone%s\n", (unsigned long)s->term, s->leader, g_alive(s, s->leader) ? " (its process exists: no heartbeat)" : "");
                fflush(s->log);
                if (group_failover(s)) { fprintf(stderr, "[apus] server %u: fail-over failed, the hooks are inert from here on\n", s->idx); s->failed = 1; s->leader = s->capacity; break; }
            } else {
                const int x = follower_next_run(s);
                if (x < 0) { fprintf(stderr, "[apus] server %u: cannot follow the new configuration, the hooks are inert from here on\n", s->idx); s->failed = 1; s->leader = s->capacity; break; }
            }
        }
        struct timespec ts = {0, 50000}; nanosleep(&ts, NULL);
    }
}

void *dare_server_init(void *arg)
{
    smr_t *s = &g_smr;
    dare_server_input_t *in = arg;
    pthread_once(&g_q_once, q_init);
    s->in = *in;
    free(in);                                      /* the reference frees it too, dare_server.c:208 */
    s->log = s->in.log ? s->in.log : stdout;
    s->group_size = s->in.group_size ? s->in.group_size : 3;
    s->idx = s->in.server_idx;
    s->batch = malloc(sizeof(apus_req_t) * Q_CAP);
    s->batch_arena = malloc(Q_ARENA);
    const char *pp = getenv("APUS_PRUNE_PERIOD_MS");
    s->prune_period_s = pp ? atof(pp) * 1e-3 : 0.05;   /* log_pruning_period, nodes.local.cfg:35 */

    /* Without APUS_GROUP_DIR this host layer runs ONE process: the group's replicas are logical replicas on this process's
     * GPU and this server leads them.  A process per server -- what benchmarks/run.sh starts on
     * three nodes -- is group mode (APUS_GROUP_DIR); a second, independent leader is refused. */
    const char *gdir = getenv("APUS_GROUP_DIR");
    s->group = gdir && *gdir;
    if (s->group) snprintf(s->group_dir, sizeof s->group_dir, "%s", gdir);
    if (!s->group && (s->in.srv_type == SRV_TYPE_JOIN || (s->idx != 0 && !getenv("APUS_ALLOW_ANY_SERVER_IDX")))) {
        fprintf(stderr, "[apus] server_idx=%u server_type=%s: without APUS_GROUP_DIR this build replicates inside ONE process (logical replicas on one "
                        "GPU, server_idx 0 leads); a process per server needs the group directory (INTEGRATION.md section 2)\n",
                s->idx, s->in.srv_type == SRV_TYPE_JOIN ? "join" : "start");
        s->ready = -1;
        return NULL;
    }
    s->capacity = s->group_size;
    const char *cap_env = getenv("APUS_GROUP_CAPACITY");
    if (s->group && cap_env && (uint32_t)atoi(cap_env) > s->capacity) s->capacity = (uint32_t)atoi(cap_env);
    if (s->capacity > APUS_MAX_SERVERS) s->capacity = APUS_MAX_SERVERS;
    g_cfg_t jc;
    memset(&jc, 0, sizeof jc);
    if (s->group && s->in.srv_type == SRV_TYPE_JOIN) {
        /* join_cluster_cb (dare_server.c:445-530): a new machine asks the group for a place.  The newest announcement says who
         * leads and which places are taken; the place is the leader's to give (handle_server_join_request,
         * dare_ibv_ud.c:995-1021: the lowest empty one, else the group grows) -- worked out here with the same rule because
         * the replica this process exports must sit in that slot before the request goes out. */
        s->peer_stamp[0].pid = 0;
        const double t0 = now_s();
        for (;;) {
            g_stamp_t st;
            if (!g_read(s, &jc, sizeof jc, "cfg_latest", &st) && g_stamp_alive(&st)) break;
            if (now_s() - t0 > 120.0) { fprintf(stderr, "[apus] join: no live group in %s\n", s->group_dir); s->ready = -1; return NULL; }
            struct timespec ts = {0, 5000000}; nanosleep(&ts, NULL);
        }
        uint32_t r = jc.size;
        for (int i = (int)jc.size - 1; i >= 0; i--) if (!((jc.bitmask >> i) & 1u)) r = (uint32_t)i;
        if (r >= s->capacity) { fprintf(stderr, "[apus] join: the group of %u is full and has no room to grow (APUS_GROUP_CAPACITY=%u)\n", jc.size, s->capacity); s->ready = -1; return NULL; }
        s->idx = r;
        s->group_size = jc.size;
    }
    apus_cfg_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.group_size = s->group ? s->capacity : s->group_size;
    if (s->group) { cfg.n_local = 1; cfg.local_ids[0] = (uint8_t)s->idx; }      /* one process per server: this one hosts replica idx */
    else {
        cfg.n_local = s->group_size;               /* logical replicas on one device */
        for (uint32_t i = 0; i < s->group_size; i++) cfg.local_ids[i] = (uint8_t)i;
    }
    const char *ll = getenv("APUS_GPU_LOG_LEN");
    cfg.log_len = ll ? strtoull(ll, NULL, 0) : 0;
    const char *dv = getenv("APUS_GPU_DEVICE");
    cfg.device = dv ? atoi(dv) : 0;
    if (apus_gpu_create(&cfg, &s->eng)) {
        fprintf(stderr, "[apus] cannot create the GPU consensus engine (no CPU path exists)\n");
        s->eng = NULL;
        s->ready = -1;
        return NULL;
    }
    apus_gpu_bind_global(s->eng);
    if (s->group) {
        /* ---- one process per server ---- */
        s->live_replica = 1; s->live_persist = 0;
        s->machines = s->group_size;
        if (s->capacity > s->group_size && apus_gpu_set_group_size(s->eng, s->group_size)) { s->ready = -1; return NULL; }
        if (s->in.srv_type == SRV_TYPE_JOIN) {
            if (apus_gpu_clear_replica(s->eng, s->idx) || group_connect(s, jc.bitmask)) { s->ready = -1; return NULL; }
            char name[64];
            snprintf(name, sizeof name, "join_%u", s->idx);
            if (g_write(s, "1", 1, name)) { s->ready = -1; return NULL; }
            /* the join reply (dare_ibv_ud.c:1071-1088): the first announcement whose configuration shows this server */
            const double t0 = now_s();
            for (;;) {
                g_stamp_t st;
                g_cfg_t c;
                if (!g_read(s, &c, sizeof c, "cfg_latest", &st) && g_stamp_alive(&st) && c.seq > jc.seq && ((c.bitmask >> s->idx) & 1u) &&
                    c.leader < s->capacity && g_same(&st, &s->peer_stamp[c.leader])) { jc = c; break; }
                if (s->terminate || now_s() - t0 > 120.0) { fprintf(stderr, "[apus] join: the group did not admit server %u\n", s->idx); s->ready = -1; return NULL; }
                struct timespec ts = {0, 2000000}; nanosleep(&ts, NULL);
            }
            s->seq = jc.seq; s->term = jc.term; s->leader = jc.leader; s->bitmask = jc.bitmask; s->group_size = jc.size; s->epoch = jc.epoch;
            s->alive_mask = jc.bitmask;
            for (uint32_t i = 0; i < s->capacity; i++)
                if (i != s->idx && ((jc.bitmask >> i) & 1u) && group_map_peer(s, i, 30.0)) { s->ready = -1; return NULL; }
            if (apus_gpu_set_config(s->eng, jc.size, jc.epoch) || apus_gpu_set_reachable(s->eng, jc.bitmask)) { s->ready = -1; return NULL; }
            joiner_mark(s);                  /* (the application is brought up to date by the DARE loop, once it listens) */
            if (group_start_run(s)) { fprintf(stderr, "[apus] server %u: cannot start the replica kernels\n", s->idx); s->ready = -1; return NULL; }
            fprintf(s->log, "[T%lu] joined as server %u\n", (unsigned long)s->term, s->idx); fflush(s->log);
        } else {
            const uint32_t all = (1u << s->group_size) - 1;
            if (group_connect(s, all)) { s->ready = -1; return NULL; }
            s->alive_mask = s->bitmask = all;
            s->term = 2; s->leader = 0;
            if (s->idx == 0) {
                /* start-up election (dare_server.c:1169, 1264-1518): server 0's timeout fires first */
                if (apus_gpu_become_leader(s->eng, 0, s->term, s->bitmask) || apus_gpu_sync(s->eng)) { fprintf(stderr, "[apus] election failed\n"); s->ready = -1; return NULL; }
                if (g_announce(s, G_KIND_START)) { s->ready = -1; return NULL; }
            } else {
                g_cfg_t c;
                if (g_wait_from(s, "cfg_1", 0, sizeof c, &c, 120.0)) { s->ready = -1; return NULL; }
                s->seq = c.seq;
            }
            if (group_start_run(s)) { fprintf(stderr, "[apus] server %u: cannot start the replica kernels\n", s->idx); s->ready = -1; return NULL; }
            if (s->idx == 0) { fprintf(s->log, "[T%lu] LEADER\n", (unsigned long)s->term); fflush(s->log); }
        }
        signal(SIGINT, on_sigint);
        s->running = 1;
        __sync_synchronize();
        s->ready = 1;
        {   const char *hb = getenv("APUS_HB_TIMEOUT_MS");
            s->hb_timeout_s = hb ? atof(hb) * 1e-3 : 0.0;
            s->hb_running = 1;
            if (pthread_create(&s->hb_thread, NULL, hb_main, s)) s->hb_running = 0; }
        group_loop(s);
        if (s->hb_running) { s->hb_running = 0; pthread_join(s->hb_thread, NULL); }
        /* shutdown: the leader drains and parks everybody; a follower waits for its workgroups (or asks them to leave) */
        if (s->leader == s->idx) { apus_gpu_rep_drain(s->eng, 5000); leader_upcalls(*s->dev_hr); s->dev_hr = NULL; apus_gpu_rep_park(s->eng); }
        else if (s->leader < s->capacity) { apus_gpu_rep_follower_stop(s->eng, s->idx); apus_gpu_rep_park(s->eng); follower_upcalls(s); }
        apus_gpu_sync(s->eng);
        const char *dump_g = getenv("APUS_PROXY_DUMP");
        if (dump_g && *dump_g) dump_replicas(s, dump_g);
        s->running = 0;
        return NULL;                                   /* (the engine stays: peers may still have this replica mapped) */
    }
    /* start-up election: every server becomes a candidate of term 1, this server's
     * timeout fires first (dare_server.c:1169, 1264-1518) */
    s->term = 2;
    s->leader = s->idx;
    if (apus_gpu_become_leader(s->eng, s->leader, s->term, (1u << s->group_size) - 1) || apus_gpu_sync(s->eng)) {
        fprintf(stderr, "[apus] election failed\n");
        s->ready = -1;
        return NULL;
    }
    /* APUS_LIVE_MODE: "replica" (default) the replica kernels, "persist" the single-workgroup persistent kernel,
     * "calls" one launch per drained batch */
    const char *lm = getenv("APUS_LIVE_MODE");
    s->live_persist = lm && !strcmp(lm, "persist");
    s->live_replica = !lm || !*lm || !strcmp(lm, "replica");
    if (s->live_replica) {
        const char *na = getenv("APUS_REP_APPEND"), *nf = getenv("APUS_REP_FWORK");
        if (apus_gpu_rep_start(s->eng, 24u * 3600u * 1000u, 500, na ? (uint32_t)atoi(na) : 16, nf ? (uint32_t)atoi(nf) : 8)) {
            fprintf(stderr, "[apus] cannot start the replica kernels\n");
            s->ready = -1;
            return NULL;
        }
        s->dev_hr = apus_gpu_rep_highest_rec_ptr(s->eng);
        s->upcalled = *s->dev_hr;
    }
    if (s->live_persist) {
        if (apus_gpu_persist_start(s->eng, 24u * 3600u * 1000u, 200)) {
            fprintf(stderr, "[apus] cannot start the persistent consensus kernel\n");
            s->ready = -1;
            return NULL;
        }
        s->dev_hr = apus_gpu_persist_highest_rec_ptr(s->eng);
        s->upcalled = *s->dev_hr;
    }
    fprintf(s->log, "[T%lu] LEADER\n", (unsigned long)s->term);     /* dare_server.c:1396, grepped by run.sh */
    fflush(s->log);
    signal(SIGINT, on_sigint);
    s->running = 1;
    __sync_synchronize();
    s->ready = 1;

    double last_prune = now_s();
    while (!s->terminate) {                        /* polling(), dare_server.c:1012-1125 */
        if (s->live_replica) {
            /* the consensus loop runs on the device and the application threads feed it themselves: this thread
             * keeps the prune timer, hands out the upcalls and watches the resident kernel */
            leader_upcalls(*s->dev_hr);
            static unsigned ref_queue_used;              /* the reference's own proxy.c feeds `tailhead`: busy-poll it like its DARE thread does */
            if (apus_tailq_drain() > 0) ref_queue_used = 20000;
            if (!s->failed && apus_gpu_rep_full(s->eng)) {
                fprintf(stderr, "[apus] the log is full: requests were dropped, admission is closed, the hooks are inert from here on\n");
                s->failed = 1;
            }
            uint64_t st[8];
            if (!s->failed && !apus_gpu_rep_stats(s->eng, st) && st[7] == 2) {
                /* the resident kernel left (idle limit, a bounded wait ran out): start it again */
                const int code = apus_gpu_rep_park(s->eng);
                fprintf(stderr, "[apus] the replica kernels left with code %d: starting them again\n", code);
                const char *na = getenv("APUS_REP_APPEND"), *nf = getenv("APUS_REP_FWORK");
                if (apus_gpu_quiesce(s->eng) || apus_gpu_sync(s->eng) ||
                    apus_gpu_rep_start(s->eng, 24u * 3600u * 1000u, 500, na ? (uint32_t)atoi(na) : 16, nf ? (uint32_t)atoi(nf) : 8)) {
                    fprintf(stderr, "[apus] restart failed: admission is closed, the hooks are inert from here on\n");
                    s->failed = 1;
                }
            }
            const double t = now_s();
            if (!s->failed && t - last_prune >= s->prune_period_s) {  /* prune_log_cb, dare_server.c:1977 */
                apus_gpu_rep_prune(s->eng);
                last_prune = t;
            }
            if (ref_queue_used) { ref_queue_used--; __builtin_ia32_pause(); }
            else { struct timespec ts = {0, 100000}; nanosleep(&ts, NULL); }
            continue;
        }
        if (s->live_persist) {
            /* the consensus loop itself runs on the device; this thread only moves what the
             * application threads queued into the command ring and keeps the prune timer */
            pthread_spin_lock(&g_q.lock);
            const uint32_t queued = g_q.n;
            pthread_spin_unlock(&g_q.lock);
            /* the reference's DARE thread busy-polls (polling() is libev's idle callback); so does this
             * one while requests keep coming -- nanosleep() costs 50+ us of timer slack per request
             * otherwise -- and backs off only after a millisecond of silence */
            static unsigned idle_spins;
            if (queued && !s->failed) { dare_ib_poll_tailq(); idle_spins = 0; }
            else if (++idle_spins < 20000) __builtin_ia32_pause();
            else { struct timespec ts = {0, 20000}; nanosleep(&ts, NULL); }
            leader_upcalls(*s->dev_hr);
            if (!s->failed && apus_gpu_persist_full(s->eng)) {
                fprintf(stderr, "[apus] the log is full: requests dropped, admission closed, the hooks are inert from here on\n");
                s->failed = 1;
            }
            const double t = now_s();
            if (t - last_prune >= s->prune_period_s) {  /* prune_log_cb, dare_server.c:1977 */
                apus_gpu_persist_prune(s->eng);
                last_prune = t;
            }
            continue;
        }
        dare_ib_poll_tailq();
        if (s->batch_n) {
            dare_ib_write_remote_logs(1);
            deliver_upcalls(s);
        } else {
            struct timespec ts = {0, 20000};
            nanosleep(&ts, NULL);
        }
        const double t = now_s();
        if (t - last_prune >= s->prune_period_s) {  /* prune_log_cb, dare_server.c:1977 */
            apus_gpu_tick_prune(s->eng);
            apus_gpu_sync(s->eng);
            dare_ib_get_remote_apply_offsets();
            last_prune = t;
        }
    }
    if (s->live_replica) {
        apus_gpu_rep_drain(s->eng, 5000);
        leader_upcalls(*s->dev_hr);
        s->dev_hr = NULL;
        apus_gpu_rep_park(s->eng);
    }
    if (s->live_persist) {
        apus_gpu_persist_drain(s->eng, 5000);
        leader_upcalls(*s->dev_hr);
        s->dev_hr = NULL;
        apus_gpu_persist_stop(s->eng);
    }
    apus_gpu_sync(s->eng);
    const uint32_t st = apus_gpu_status(s->eng);
    if (st) fprintf(stderr, "[apus] device status %#x at shutdown\n", st);
    const char *dump = getenv("APUS_PROXY_DUMP");
    if (dump && *dump) dump_replicas(s, dump);
    if (getenv("APUS_PROXY_KEEP_ENGINE")) {          /* tests: the caller inspects the engine (apus_gpu_global) and destroys it */
        s->running = 0;
        return NULL;
    }
    apus_gpu_destroy(s->eng);
    s->eng = NULL;
    s->running = 0;
    return NULL;
}

/* ------------------------------------------------------------------------- */
/* proxy (B-outer) */
#define MAX_FDS 65536

typedef struct { uint64_t req_id; uint16_t connection_id; uint8_t used; } lead_pair_t;
typedef struct { int sock; uint8_t used; } foll_pair_t;

struct proxy_node_t {
    struct sockaddr_in sys_addr;       /* where the local application listens (follower replay) */
    lead_pair_t *leader_map;           /* leader_hash_map keyed by fd, proxy.h:36-45 */
    foll_pair_t *follower_map;         /* follower_hash_map keyed by connection_id */
    volatile uint64_t highest_rec;     /* proxy.h:47 */
    volatile uint64_t cur_rec;
    uint8_t pair_count;                /* nc_t */
    int req_log;
    FILE *req_log_file;
    char db_name[128];
    pthread_t dare_thread;
};

static struct proxy_node_t *g_proxy;

static void proxy_mirror_highest_rec(uint64_t v) { if (g_proxy) g_proxy->highest_rec = v; }
/* does the local application accept connections yet?  (the interposer initialises from an ELF constructor: main() has not run) */
static int proxy_app_listening(void)
{
    if (!g_proxy) return 1;
    int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return 0;
    const int ok = connect(fd, (struct sockaddr *)&g_proxy->sys_addr, sizeof g_proxy->sys_addr) == 0;
    close(fd);
    return ok;
}
static void proxy_align_cur_rec(uint64_t v) { if (g_proxy && g_proxy->cur_rec < v) g_proxy->cur_rec = v; }

static void update_highest_rec(void *arg)                        /* proxy.c:263-267 */
{
    struct proxy_node_t *p = arg;
    p->highest_rec++;
}

static void do_action_to_server(uint16_t clt_id, uint8_t type, size_t data_size, void *data, void *arg)
{                                                                /* proxy.c:341-439 */
    struct proxy_node_t *p = arg;
    foll_pair_t *fp = &p->follower_map[clt_id];
    if (p->req_log && p->req_log_file)
        fprintf(p->req_log_file, type == PROXY_CONNECT ? "Operation: Connects.\n" :
                type == PROXY_SEND ? "Operation: Sends data.\n" : "Operation: Closes.\n");
    switch (type) {
    case PROXY_CONNECT:
        if (!fp->used) {
            int fd = socket(AF_INET, SOCK_STREAM, 0);
            if (fd < 0) { fprintf(stderr, "ERROR opening socket!\n"); return; }
            fp->sock = fd; fp->used = 1;
            if (connect(fd, (struct sockaddr *)&p->sys_addr, sizeof p->sys_addr) < 0) fprintf(stderr, "ERROR connecting!\n");
            int fl = fcntl(fd, F_GETFL); if (fl != -1) fcntl(fd, F_SETFL, fl | O_NONBLOCK);
            int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        }
        break;
    case PROXY_SEND:
        /* The reference writes once into its non-blocking socket (proxy.c:417-424): whatever the socket buffer does not
         * take -- a replay that runs ahead of the application, e.g. a joined machine's state transfer or a follower behind a
         * fast leader -- is LOST and the replica's application silently differs from then on.  Here the DARE thread waits
         * for the application instead (back-pressure: what it has not replayed is not reported as applied, so the leader's
         * head cannot pass it). */
        if (fp->used) {
            const uint8_t *b = data;
            size_t left = data_size;
            int stalls = 0;
            while (left) {
                const ssize_t w = write(fp->sock, b, left);
                if (w > 0) { b += w; left -= (size_t)w; stalls = 0; continue; }
                if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) && !g_smr.terminate && stalls < 30000) {
                    struct pollfd pf = { fp->sock, POLLOUT, 0 };
                    poll(&pf, 1, 10);
                    stalls++;
                    continue;
                }
                fprintf(stderr, "ERROR writing to socket!\n");
                break;
            }
        }
        break;
    case PROXY_CLOSE:
        if (fp->used) {
            /* The replies the application wrote on this connection were never read (proxy.c:341-439 does not read them
             * either): close() on a socket with unread data sends a RESET, and a reset makes the peer drop what it has not
             * read yet -- a replay that runs ahead of the application (a joined machine's state transfer) lost the last
             * commands of every connection that way.  So: half-close, let the application read to the end and close its
             * side, THEN close.  Bounded; the DARE thread waits for the application here as it does in PROXY_SEND. */
            shutdown(fp->sock, SHUT_WR);
            char sink[4096];
            for (int waits = 0; waits < 500 && !g_smr.terminate; ) {
                const ssize_t r = read(fp->sock, sink, sizeof sink);
                if (r == 0) break;                                         /* the application has closed its side */
                if (r > 0) continue;
                if (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) break;
                struct pollfd pf = { fp->sock, POLLIN, 0 };
                poll(&pf, 1, 10);
                waits++;
            }
            if (close(fp->sock)) fprintf(stderr, "ERROR closing socket!\n");
            fp->used = 0;
        }
        break;
    default: break;
    }
}

static int is_inner(pthread_t tid)                               /* proxy.c:91-99 */
{
    return g_proxy && pthread_equal(tid, g_proxy->dare_thread);
}

/* leader_handle_submit_req, proxy.c:108-161: same id assignment, same lock scope,
 * same blocking (the caller returns only after its entry was applied) */
static void leader_handle_submit_req(uint8_t type, ssize_t data_size, void *buf, int fd, struct proxy_node_t *p)
{
    if (fd < 0 || fd >= MAX_FDS || data_size < 0 || data_size > 65535) return;
    /* once admission is closed (log full, the engine is gone) the hooks are inert: the application goes on
     * unreplicated, loudly (the message is printed where `failed` is set) -- like the reference after a failed
     * proxy_init (proxy.c:502-506, the `proxy != NULL` guards of spec_hooks.cpp) */
    if (g_smr.failed || g_smr.terminate || g_smr.ready < 0) return;
    pthread_once(&g_q_once, q_init);
    lead_pair_t *pair = &p->leader_map[fd];
    uint64_t req_id = 0;
    uint16_t connection_id = 0;
    const int replica = g_smr.live_replica;
    for (;;) {
        pthread_spin_lock(&g_q.lock);
        if (!g_admission_closed && (replica || (g_q.n < Q_CAP && g_q.arena_used + (uint64_t)data_size + 16 <= Q_ARENA))) break;
        pthread_spin_unlock(&g_q.lock);              /* queue full, or the leader is between two runs (leader_pause): wait OUTSIDE the lock */
        if (g_smr.failed || g_smr.terminate || g_smr.ready < 0) return;
        sched_yield();
    }
    const uint64_t cur_rec = ++p->cur_rec;
    switch (type) {
    case PROXY_CONNECT:
        memset(pair, 0, sizeof *pair);
        pair->used = 1;
        pair->connection_id = (uint16_t)(((uint16_t)get_node_id() << 8) | p->pair_count++);   /* gen_key, proxy.c:101 */
        req_id = ++pair->req_id;
        connection_id = pair->connection_id;
        break;
    case PROXY_SEND:
        if (!pair->used) { p->cur_rec--; pthread_spin_unlock(&g_q.lock); return; }   /* the reference dereferences NULL here (Q8) */
        req_id = ++pair->req_id;
        connection_id = pair->connection_id;
        break;
    case PROXY_CLOSE:
        if (!pair->used) { p->cur_rec--; pthread_spin_unlock(&g_q.lock); return; }
        req_id = ++pair->req_id;
        connection_id = pair->connection_id;
        pair->used = 0;
        break;
    default:
        p->cur_rec--; pthread_spin_unlock(&g_q.lock); return;
    }
    if (replica) {
        /* the multi-producer ring: the slot (= the place in the log order) is reserved together with the ids, the
         * payload is copied and the slot published outside the lock, by this thread -- no queue, no second copy */
        uint64_t slot = 0; void *dst = NULL;
        const int rc = apus_gpu_rep_reserve(g_smr.eng, (uint32_t)data_size, &slot, &dst);
        pthread_spin_unlock(&g_q.lock);
        if (rc) {
            fprintf(stderr, "[apus] no request slot (rc=%d): admission is closed, the hooks are inert from here on\n", rc);
            g_smr.failed = 1;
            return;
        }
        if (data_size) memcpy(dst, buf, (size_t)data_size);
        apus_gpu_rep_publish(g_smr.eng, slot, dst, req_id, connection_id, type, (uint16_t)data_size);
    } else {
        q_push_locked(type, connection_id, req_id, buf, (uint16_t)data_size);
        pthread_spin_unlock(&g_q.lock);
    }
    /* proxy.c:160: the caller returns once its entry is applied.  With the persistent kernel the word
     * it spins on is the one the DEVICE bumps (pinned host memory): no host thread in between */
    for (;;) {
        const volatile uint64_t *hr = g_smr.dev_hr;
        if (cur_rec <= (hr ? *hr : p->highest_rec)) break;
        if (g_smr.terminate || g_smr.failed || g_smr.ready < 0) break;
        __builtin_ia32_pause();
    }
}

void proxy_on_read(struct proxy_node_t *p, void *buf, ssize_t bytes_read, int fd)
{
    if (!p || is_inner(pthread_self())) return;
    if (is_leader()) leader_handle_submit_req(PROXY_SEND, bytes_read, buf, fd, p);
}

void proxy_on_accept(struct proxy_node_t *p, int fd)
{
    if (!p || is_inner(pthread_self())) return;
    if (is_leader()) leader_handle_submit_req(PROXY_CONNECT, 0, NULL, fd, p);
}

void proxy_on_close(struct proxy_node_t *p, int fd)
{
    if (!p || is_inner(pthread_self())) return;
    if (is_leader()) leader_handle_submit_req(PROXY_CLOSE, 0, NULL, fd, p);
}

/* the four keys proxy_read_config takes from the libconfig file (config-proxy.c:6-60) */
static int read_cfg(struct proxy_node_t *p, const char *path)
{
    char ip[64] = "127.0.0.1";
    int port = 6379;
    if (path && *path) {
        FILE *f = fopen(path, "r");
        if (!f) return -1;
        char line[512];
        while (fgets(line, sizeof line, f)) {
            char key[64], val[256];
            char *h = strchr(line, '#'); if (h) *h = 0;
            if (sscanf(line, " %63[A-Za-z_] = %255[^;\n]", key, val) != 2) continue;
            char *v = val; while (*v == ' ' || *v == '"') v++;
            char *e = v + strlen(v); while (e > v && (e[-1] == ' ' || e[-1] == '"')) *--e = 0;
            if (!strcmp(key, "port")) port = atoi(v);
            else if (!strcmp(key, "ip_address")) snprintf(ip, sizeof ip, "%s", v);
            else if (!strcmp(key, "req_log")) p->req_log = atoi(v);
            else if (!strcmp(key, "db_name")) snprintf(p->db_name, sizeof p->db_name, "%s", v);
        }
        fclose(f);
    }
    memset(&p->sys_addr, 0, sizeof p->sys_addr);
    p->sys_addr.sin_family = AF_INET;
    p->sys_addr.sin_port = htons((uint16_t)port);
    inet_pton(AF_INET, ip, &p->sys_addr.sin_addr);
    return 0;
}

struct proxy_node_t *proxy_init(const char *config_path, const char *proxy_log_path)
{
    struct proxy_node_t *p = calloc(1, sizeof *p);
    if (!p) return NULL;
    if (read_cfg(p, config_path)) {                               /* proxy.c:452-455 */
        fprintf(stderr, "PROXY : Configuration File Reading Error.\n");
        free(p);
        return NULL;
    }
    p->leader_map = calloc(MAX_FDS, sizeof(lead_pair_t));
    p->follower_map = calloc(65536, sizeof(foll_pair_t));
    if (!p->leader_map || !p->follower_map) { free(p->leader_map); free(p->follower_map); free(p); return NULL; }
    if (p->req_log) {
        char path[512];
        snprintf(path, sizeof path, "%s/node-proxy-req.log", proxy_log_path ? proxy_log_path : ".");
        p->req_log_file = fopen(path, "w");
    }
    pthread_once(&g_q_once, q_init);

    /* dare_main, proxy.c:22-89: same environment variables, same defaults */
    dare_server_input_t *in = calloc(1, sizeof *in);
    if (!in) { free(p->leader_map); free(p->follower_map); free(p); return NULL; }      /* (the hooks stay inert: proxy != NULL guards) */
    in->log = stdout;
    in->name = "";
    in->output = "dare_servers.out";
    in->srv_type = SRV_TYPE_START;
    in->sm_type = 2;
    in->server_idx = 0xFF;
    const char *e;
    if ((e = getenv("server_idx"))) in->server_idx = (uint8_t)atoi(e);
    in->group_size = 3;
    if ((e = getenv("group_size"))) in->group_size = (uint8_t)atoi(e);
    if ((e = getenv("server_type")) && !strcmp(e, "join")) in->srv_type = SRV_TYPE_JOIN;
    if ((e = getenv("dare_log_file")) && *e) {
        in->log = fopen(e, "w+");
        if (!in->log) { printf("Cannot open log file\n"); exit(1); }
    }
    if (in->srv_type == SRV_TYPE_START && in->server_idx == 0xFF) {
        printf("A server cannot start without an index\n");
        exit(1);
    }
    in->do_action = do_action_to_server;
    in->update_state = update_highest_rec;
    if (config_path) snprintf(in->config_path, sizeof in->config_path, "%s", config_path);
    in->up_para = p;
    g_proxy = p;
    if (pthread_create(&p->dare_thread, NULL, dare_server_init, in)) {
        fprintf(stderr, "Cannot init dare_thread\n");
        free(p);
        g_proxy = NULL;
        return NULL;
    }
    /* unlike the reference, wait until the engine answered: the first hooked call
     * would otherwise race the election */
    while (!g_smr.ready) { struct timespec ts = {0, 1000000}; nanosleep(&ts, NULL); }
    if (g_smr.ready < 0) { g_proxy = NULL; return NULL; }
    return p;
}

/* test / shutdown helper: stop the DARE thread and wait for it */
void apus_proxy_shutdown(struct proxy_node_t *p)
{
    if (!p) return;
    dare_server_shutdown();
    pthread_join(p->dare_thread, NULL);
}

uint64_t apus_proxy_highest_rec(struct proxy_node_t *p)
{
    const volatile uint64_t *hr = g_smr.dev_hr;
    return hr ? *hr : (p ? p->highest_rec : 0);
}

/* 1 once requests were dropped because the log is full (admission is closed from then on) */
int apus_proxy_failed(void) { return g_smr.failed; }

/* stablestorage_load_records, src/proxy/proxy.c:306-339: a snapshot is the stored records back to back;
 * every record goes back into the store and is replayed into the application.  Record layout =
 * proxy_msg_header {u16 connection_id; u8 action; pad} (proxy.h:57-60); a SEND record is
 * sizeof(proxy_send_msg) = 24 bytes + data.cmd.len, data at +8 of the record (proxy.h:83-91). */
int apus_snapshot_replay(const void *buf, uint32_t size,
                         void (*store)(const void *rec, uint32_t n, void *arg),
                         proxy_do_action_cb_t do_action, void *arg)
{
    const uint8_t *b = (const uint8_t *)buf;
    uint32_t len = 0;
    int records = 0;
    while (len < size) {
        if (size - len < 4) return -1;
        const uint8_t *rec = b + len;
        uint16_t conn; memcpy(&conn, rec, 2);
        const uint8_t action = rec[2];
        uint32_t n;
        if (action == PROXY_SEND) {
            if (size - len < 24) return -1;
            uint16_t cmd_len; memcpy(&cmd_len, rec + 8, 2);
            n = 24u + cmd_len;
            if (size - len < n) return -1;
            if (store) store(rec, n, arg);
            if (do_action) do_action(conn, PROXY_SEND, cmd_len, (void *)(rec + 10), arg);
        } else if (action == PROXY_CONNECT || action == PROXY_CLOSE) {
            n = 4;
            if (store) store(rec, n, arg);
            if (do_action) do_action(conn, action, 0, NULL, arg);
        } else return -1;
        len += n;
        records++;
    }
    return records;
}
