/*
 * fabric.c -- the in-process "NIC" behind oracle/refshim/infiniband/verbs.h.
 * TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libapus_fabric.so).
 *
 * N instances of the UNMODIFIED reference server (each its own copy of
 * libapus_ref_loops.so, see refcluster.c) live in one process and one thread.  Every
 * instance owns one fabric "port" (LID = port index + 1).  Semantics kept from the verbs
 * API as the reference uses it:
 *   RC  ibv_post_send(RDMA_WRITE / RDMA_READ): executed synchronously as a memcpy against
 *       the responder's registered region, IF the responder QP exists, belongs to a live
 *       port, is in RTR/RTS and is connected back to the requester QP.  Otherwise the
 *       requester gets IBV_WC_RETRY_EXC_ERR and its QP moves to ERR (later posts are
 *       flushed with IBV_WC_WR_FLUSH_ERR) -- this is what makes QP reset a fence
 *       (rc_revoke_log_access, /root/reference/src/dare/dare_ibv_rc.c:2156).
 *       Errors always produce a completion; successes only when signaled.
 *   UD  SEND to (dlid, qpn) or to the multicast group (qpn 0xFFFFFF): copied behind a
 *       40-byte GRH into the next posted receive of every destination QP.
 *   A port can be HELD (link down for a while, the server itself keeps running): RC
 *   operations to or from it fail like those to a dead port.  That is what the reference's QP
 *   settings give on a real fabric: timeout = 1 (~8 us), retry_cnt = 0
 *   (dare_ibv_rc.c:2400-2402), so an unreachable peer means IBV_WC_RETRY_EXC_ERR at once.
 * The trace driver gets a callback after every remote write (fab_set_write_hook).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <infiniband/verbs.h>
#include "fabric.h"
#ifdef FAB_PROC
/* -DFAB_PROC: ONE SERVER PER PROCESS (BASELINE configs[0]: three redis-server processes under the reference's own
 * interposer on one host, "CPU loopback (no RDMA/GPU)").  The same fabric, with its state in a shared-memory arena
 * that every process maps at the same address (so the pointers in it mean the same everywhere), one process-shared
 * lock around every verb, and the data movement of RDMA WRITE / READ / UD SEND done with process_vm_writev / readv
 * against the process that owns the port.  The port of a process comes from the environment
 * (server_idx, the reference's own variable).  A process that is gone answers like a dead port: RETRY_EXC_ERR. */
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <sys/prctl.h>
#include <signal.h>
#include <sched.h>
#include <unistd.h>
#define FAB_BASE  ((void *)0x6f0000000000ull)
#define FAB_BYTES (1ull << 30)
#define FAB_STAGE 65536u
#endif

#define MAX_PORTS   16
#define MAX_QPS     4096
#define MAX_MRS     8192

typedef struct fcq {
    struct ibv_cq pub;
    struct ibv_wc *ring;
    uint64_t *visible_at;
    int cap, head, count;
    int port;
    uint64_t empty_polls;
} fcq_t;

typedef struct frecv { uint64_t wr_id, addr; uint32_t length; } frecv_t;

typedef struct fqp {
    struct ibv_qp pub;
    int port, live;
    struct ibv_qp_attr attr;
    struct ibv_qp_cap cap;
    fcq_t *scq, *rcq;
    frecv_t *rq; int rq_cap, rq_head, rq_count;
    int mcast; union ibv_gid mgid; uint16_t mlid;
} fqp_t;

typedef struct fmr { struct ibv_mr pub; int port, access, live; } fmr_t;

typedef struct fport {
    int used, alive, held;
    /* Completion timing.  A real HCA needs microseconds to complete a WRITE, so the poll that
     * the reference issues right behind every post (post_send -> empty_completion_queue,
     * dare_ibv_rc.c:2590) finds nothing and the completion is seen by the NEXT poll pass
     * (the one at the top of rc_write_remote_logs' loop, :1890).  Modelled as: a send
     * completion becomes visible when its port polls a CQ without having posted in between. */
    uint64_t epoch; int last_was_post;
    struct ibv_device dev;
} fport_t;

#ifndef FAB_PROC
static fport_t ports[MAX_PORTS];
static fqp_t *qps[MAX_QPS];
static fmr_t *mrs[MAX_MRS];
static uint32_t next_qpn = 16, next_key = 100;
static int current_port = -1;
static struct fab_stats stats;
#define FAB_LOCK()   ((void)0)
#define FAB_UNLOCK() ((void)0)
#define fab_malloc malloc
#define fab_calloc calloc
#define fab_free   free
static inline int fab_put(int port, uint64_t raddr, const void *local, uint32_t len) { (void)port; memcpy((void *)(uintptr_t)raddr, local, len); return 0; }
static inline int fab_get(int port, void *local, uint64_t raddr, uint32_t len) { (void)port; memcpy(local, (const void *)(uintptr_t)raddr, len); return 0; }
static inline int fab_zero(int port, uint64_t raddr, uint32_t len) { (void)port; memset((void *)(uintptr_t)raddr, 0, len); return 0; }
#else
typedef struct fab_shared {
    pthread_mutex_t mu;
    volatile int ready;
    size_t brk;                              /* bump allocator behind this struct */
    fport_t ports_[MAX_PORTS];
    int pid_of[MAX_PORTS];
    fqp_t *qps_[MAX_QPS];
    fmr_t *mrs_[MAX_MRS];
    uint32_t next_qpn_, next_key_;
    struct fab_stats stats_;
    /* Where one process may not write into another's memory (no CAP_SYS_PTRACE, Yama), the OWNER moves the bytes: every
     * process runs a "NIC" thread that serves one staged copy at a time (all verbs run under `mu`, so one slot per port). */
    volatile int use_nic;
    struct fab_nic { volatile uint32_t state; uint32_t op; uint64_t addr; uint32_t len; int32_t rc; uint8_t data[FAB_STAGE]; } nic[MAX_PORTS];
} fab_shared_t;
static fab_shared_t *G;
static int current_port = -1;
#define ports    (G->ports_)
#define qps      (G->qps_)
#define mrs      (G->mrs_)
#define next_qpn (G->next_qpn_)
#define next_key (G->next_key_)
#define stats    (G->stats_)
#define FAB_LOCK()   do { if (pthread_mutex_lock(&G->mu) == EOWNERDEAD) pthread_mutex_consistent(&G->mu); } while (0)
#define FAB_UNLOCK() pthread_mutex_unlock(&G->mu)
static void *fab_nic_thread(void *arg)
{
    (void)arg;
    struct fab_nic *n = &G->nic[current_port];
    for (unsigned idle = 0;;) {
        if (n->state == 1) {
            __sync_synchronize();
            if (n->op == 1) memcpy((void *)(uintptr_t)n->addr, (const void *)n->data, n->len);
            else memcpy((void *)n->data, (const void *)(uintptr_t)n->addr, n->len);
            n->rc = 0;
            __sync_synchronize();
            n->state = 2;
            idle = 0;
        } else if (++idle > 4000) { if (G->use_nic) sched_yield(); else usleep(2000); }
    }
    return NULL;
}
/* one staged copy carried out by the owner of `port`: op 1 = into its memory, 2 = out of it */
static int fab_nic_copy(int port, int op, uint64_t raddr, void *local, uint32_t len)
{
    struct fab_nic *n = &G->nic[port];
    for (uint32_t off = 0; off < len; off += FAB_STAGE) {
        const uint32_t k = len - off < FAB_STAGE ? len - off : FAB_STAGE;
        if (op == 1) memcpy((void *)n->data, (const uint8_t *)local + off, k);
        n->op = (uint32_t)op; n->addr = raddr + off; n->len = k;
        __sync_synchronize();
        n->state = 1;
        for (uint64_t spins = 0; n->state != 2; spins++) {
            if ((spins & 0xFFFFF) == 0xFFFFF && kill(G->pid_of[port], 0)) { n->state = 0; return -1; }   /* the owner is gone */
            if (spins > (1ull << 32)) { n->state = 0; return -1; }
        }
        if (op == 2) memcpy((uint8_t *)local + off, (const void *)n->data, k);
        n->state = 0;
    }
    return 0;
}
static void fab_attach(void)
{
    if (G) return;
    const char *name = getenv("APUS_FAB_SHM");
    if (!name || !*name) name = "/apus_fab";
    int creator = 1;
    int fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0) { creator = 0; fd = shm_open(name, O_RDWR, 0600); }
    if (fd < 0) { perror("fabric: shm_open"); abort(); }
    if (creator && ftruncate(fd, (off_t)FAB_BYTES)) { perror("fabric: ftruncate"); abort(); }
    if (!creator) { struct stat st; for (int i = 0; i < 2000; i++) { if (!fstat(fd, &st) && (unsigned long long)st.st_size >= FAB_BYTES) break; usleep(1000); } }
    void *m = mmap(FAB_BASE, FAB_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED_NOREPLACE, fd, 0);
    if (m != FAB_BASE) { perror("fabric: mmap at the agreed address"); abort(); }
    close(fd);
    G = (fab_shared_t *)m;
    if (creator) {
        pthread_mutexattr_t a; pthread_mutexattr_init(&a);
        pthread_mutexattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_mutexattr_setrobust(&a, PTHREAD_MUTEX_ROBUST);
        pthread_mutex_init(&G->mu, &a);
        G->brk = (sizeof(fab_shared_t) + 4095) & ~(size_t)4095;
        G->next_qpn_ = 16; G->next_key_ = 100;
        __sync_synchronize();
        G->ready = 1;
    } else while (!G->ready) usleep(1000);
    const char *idx = getenv("server_idx");
    current_port = idx ? atoi(idx) : 0;
    prctl(PR_SET_PTRACER, PR_SET_PTRACER_ANY, 0, 0, 0);          /* Yama scope 1: siblings may write into this process */
    if (getenv("APUS_FAB_NIC")) G->use_nic = 1;
    pthread_t th;
    pthread_create(&th, NULL, fab_nic_thread, NULL);
    pthread_detach(th);
}
static void *fab_malloc(size_t n)
{
    n = (n + 63) & ~(size_t)63;
    if (G->brk + n > FAB_BYTES) { fprintf(stderr, "fabric: shared arena exhausted\n"); abort(); }
    void *p = (char *)G + G->brk; G->brk += n;
    return p;
}
static void *fab_calloc(size_t a, size_t b) { void *p = fab_malloc(a * b); memset(p, 0, a * b); return p; }
static void fab_free(void *p) { (void)p; }
/* data movement against the process that owns `port` */
static int fab_put(int port, uint64_t raddr, const void *local, uint32_t len)
{
    if (port == current_port) { memcpy((void *)(uintptr_t)raddr, local, len); return 0; }
    if (!G->use_nic) {
        struct iovec l = { (void *)local, len }, r = { (void *)(uintptr_t)raddr, len };
        if (process_vm_writev(G->pid_of[port], &l, 1, &r, 1, 0) == (ssize_t)len) return 0;
        if (errno != EPERM) return -1;
        G->use_nic = 1;                                  /* not allowed here: from now on the owners move the bytes */
    }
    return fab_nic_copy(port, 1, raddr, (void *)local, len);
}
static int fab_get(int port, void *local, uint64_t raddr, uint32_t len)
{
    if (port == current_port) { memcpy(local, (const void *)(uintptr_t)raddr, len); return 0; }
    if (!G->use_nic) {
        struct iovec l = { local, len }, r = { (void *)(uintptr_t)raddr, len };
        if (process_vm_readv(G->pid_of[port], &l, 1, &r, 1, 0) == (ssize_t)len) return 0;
        if (errno != EPERM) return -1;
        G->use_nic = 1;
    }
    return fab_nic_copy(port, 2, raddr, local, len);
}
static int fab_zero(int port, uint64_t raddr, uint32_t len)
{
    static const char z[64];
    return fab_put(port, raddr, z, len > 64 ? 64 : len);
}
#endif
static fab_write_hook_t write_hook;
static void *write_hook_arg;
static int delay_completions;   /* 0: a completion is visible to the very next poll (default) */
void fab_set_completion_delay(int on) { delay_completions = on; }

/* ---- driver-facing control ---------------------------------------------------------- */
#ifdef FAB_PROC
void fab_reset(void) { }
#else
void fab_reset(void)
{
    memset(ports, 0, sizeof ports);
    /* QPs and MRs of a previous cluster: its instances are gone, nobody refers to them */
    for (int i = 0; i < MAX_QPS; i++) if (qps[i]) { free(qps[i]->rq); free(qps[i]); }
    for (int i = 0; i < MAX_MRS; i++) free(mrs[i]);
    memset(qps, 0, sizeof qps);
    memset(mrs, 0, sizeof mrs);
    next_qpn = 16; next_key = 100; current_port = -1;
    memset(&stats, 0, sizeof stats);
}
#endif
int  fab_enter(int port) { int prev = current_port; current_port = port; return prev; }
void fab_leave(int prev) { current_port = prev; }
int  fab_current(void) { return current_port; }
void fab_set_write_hook(fab_write_hook_t h, void *arg) { write_hook = h; write_hook_arg = arg; }
#ifndef FAB_PROC
void fab_kill_port(int port) { if (port >= 0 && port < MAX_PORTS) ports[port].alive = 0; }
int  fab_port_alive(int port) { return port >= 0 && port < MAX_PORTS && ports[port].alive; }
void fab_hold_port(int port) { if (port >= 0 && port < MAX_PORTS) ports[port].held = 1; }
int  fab_port_held(int port) { return port >= 0 && port < MAX_PORTS && ports[port].held; }
const struct fab_stats *fab_get_stats(void) { return &stats; }
#endif

static void cq_push(fcq_t *cq, const struct ibv_wc *wc, int delayed)
{
    if (cq->count == cq->cap) {            /* CQ overrun: grow (a real HCA would raise a fatal event) */
        int ncap = cq->cap * 2;
        struct ibv_wc *nr = fab_malloc(sizeof(*nr) * ncap);
        uint64_t *nv = fab_malloc(sizeof(*nv) * ncap);
        for (int i = 0; i < cq->count; i++) { nr[i] = cq->ring[(cq->head + i) % cq->cap]; nv[i] = cq->visible_at[(cq->head + i) % cq->cap]; }
        fab_free(cq->ring); fab_free(cq->visible_at); cq->ring = nr; cq->visible_at = nv; cq->cap = ncap; cq->head = 0;
    }
    cq->ring[(cq->head + cq->count) % cq->cap] = *wc;
    cq->visible_at[(cq->head + cq->count) % cq->cap] = ports[cq->port].epoch + (delayed ? 1 : 0);
    cq->count++;
    cq->empty_polls = 0;
}

static void complete(fqp_t *qp, uint64_t wr_id, enum ibv_wc_status st, int opcode)
{
    struct ibv_wc wc; memset(&wc, 0, sizeof wc);
    wc.wr_id = wr_id; wc.status = st; wc.qp_num = qp->pub.qp_num;
    wc.opcode = opcode == IBV_WR_RDMA_READ ? IBV_WC_RDMA_READ : opcode == IBV_WR_SEND ? IBV_WC_SEND : IBV_WC_RDMA_WRITE;
    cq_push(qp->scq, &wc, 1);
}

static fqp_t *find_qp(uint32_t qpn) { return qpn < MAX_QPS ? qps[qpn] : NULL; }

static fmr_t *find_mr(int port, uint32_t rkey, uint64_t addr, uint32_t len)
{
    if (rkey >= MAX_MRS) return NULL;
    fmr_t *m = mrs[rkey];
    if (!m || !m->live || m->port != port) return NULL;
    if (addr < (uint64_t)(uintptr_t)m->pub.addr || addr + len > (uint64_t)(uintptr_t)m->pub.addr + m->pub.length) return NULL;
    return m;
}

/* execute one RC operation against the responder; returns the completion status */
static enum ibv_wc_status rc_execute(fqp_t *src, int opcode, uint64_t local,
                                     uint64_t raddr, uint32_t rkey, uint32_t len, int *tport_out)
{
    int tport = (int)src->attr.ah_attr.dlid - 1;
    *tport_out = tport;
    if (tport < 0 || tport >= MAX_PORTS || !ports[tport].used || !ports[tport].alive) return IBV_WC_RETRY_EXC_ERR;
    if (ports[tport].held || ports[src->port].held) return IBV_WC_RETRY_EXC_ERR;
    fqp_t *dst = find_qp(src->attr.dest_qp_num);
    if (!dst || !dst->live || dst->port != tport || dst->pub.qp_type != IBV_QPT_RC) return IBV_WC_RETRY_EXC_ERR;
    if (dst->pub.state != IBV_QPS_RTR && dst->pub.state != IBV_QPS_RTS) return IBV_WC_RETRY_EXC_ERR;
    if (dst->attr.dest_qp_num != src->pub.qp_num) return IBV_WC_RETRY_EXC_ERR;
    /* packet sequence numbers must agree on both ends (the TERM_PSN build of the
     * reference fences by term through them, dare_ibv_rc.c:2325-2333) */
    if ((dst->attr.rq_psn & 0xFFFFFF) != (src->attr.sq_psn & 0xFFFFFF)) return IBV_WC_RETRY_EXC_ERR;
    fmr_t *m = find_mr(tport, rkey, raddr, len);
    if (!m) return IBV_WC_REM_ACCESS_ERR;
    if (opcode == IBV_WR_RDMA_WRITE) {
        if (!(m->access & IBV_ACCESS_REMOTE_WRITE)) return IBV_WC_REM_ACCESS_ERR;
        if (fab_put(tport, raddr, (const void *)(uintptr_t)local, len)) return IBV_WC_RETRY_EXC_ERR;
        stats.rc_writes++; stats.rc_write_bytes += len;
    } else {
        if (!(m->access & IBV_ACCESS_REMOTE_READ)) return IBV_WC_REM_ACCESS_ERR;
        if (fab_get(tport, (void *)(uintptr_t)local, raddr, len)) return IBV_WC_RETRY_EXC_ERR;
        stats.rc_reads++;
    }
    return IBV_WC_SUCCESS;
}

static void rc_finish(fqp_t *src, uint64_t wr_id, int opcode, int signaled, enum ibv_wc_status st,
                      int tport, uint64_t raddr, uint32_t len)
{
    if (st != IBV_WC_SUCCESS) {
        src->pub.state = IBV_QPS_ERR;
        complete(src, wr_id, st, opcode);
        stats.rc_failures++;
        return;
    }
    if (signaled) complete(src, wr_id, st, opcode);
    if (opcode == IBV_WR_RDMA_WRITE && write_hook) write_hook(write_hook_arg, src->port, tport, raddr, len);
}

#ifndef FAB_PROC
void fab_release_port(int port) { if (port >= 0 && port < MAX_PORTS) ports[port].held = 0; }
#endif

/* ---- device ---------------------------------------------------------------------------- */
struct ibv_device **ibv_get_device_list(int *num)
{
#ifdef FAB_PROC
    fab_attach();
    FAB_LOCK();
#endif
    int port = current_port;
    if (port < 0 || port >= MAX_PORTS) { if (num) *num = 0; FAB_UNLOCK(); return NULL; }
    fport_t *p = &ports[port];
    if (!p->used) {
        p->used = 1; p->alive = 1; p->held = 0;
        snprintf(p->dev.name, sizeof p->dev.name, "fab%d", port);
        p->dev.fab_port = port;
    }
#ifdef FAB_PROC
    G->pid_of[port] = (int)getpid();
#endif
    FAB_UNLOCK();
    struct ibv_device **l = calloc(2, sizeof *l);
    l[0] = &p->dev;
    if (num) *num = 1;
    return l;
}
void ibv_free_device_list(struct ibv_device **l) { free(l); }
const char *ibv_get_device_name(struct ibv_device *d) { return d->name; }
struct ibv_context *ibv_open_device(struct ibv_device *d)
{
    struct ibv_context *c = calloc(1, sizeof *c);
    c->device = d; c->fab_port = d->fab_port;
    return c;
}
int ibv_close_device(struct ibv_context *c) { free(c); return 0; }

int ibv_query_device(struct ibv_context *c, struct ibv_device_attr *a)
{
    (void)c; memset(a, 0, sizeof *a);
    a->max_qp = MAX_QPS; a->max_qp_wr = 16384; a->max_sge = 32; a->max_cq = 65536; a->max_cqe = 4194303;
    a->max_mr = MAX_MRS; a->max_pd = 32768; a->max_qp_rd_atom = 16; a->max_res_rd_atom = 16;
    a->max_qp_init_rd_atom = 16; a->atomic_cap = IBV_ATOMIC_HCA; a->max_mcast_grp = 8192;
    a->max_mcast_qp_attach = 248; a->max_ah = 65536; a->max_srq = 65472; a->max_srq_wr = 16383;
    a->max_pkeys = 128; a->phys_port_cnt = 1;
    return 0;
}
int ibv_query_port(struct ibv_context *c, uint8_t port_num, struct ibv_port_attr *a)
{
    (void)port_num; memset(a, 0, sizeof *a);
    a->state = IBV_PORT_ACTIVE; a->max_mtu = IBV_MTU_4096; a->active_mtu = IBV_MTU_4096;
    a->lid = (uint16_t)(c->fab_port + 1); a->link_layer = IBV_LINK_LAYER_INFINIBAND; a->pkey_tbl_len = 128;
    a->gid_tbl_len = 8;
    return 0;
}
int ibv_query_pkey(struct ibv_context *c, uint8_t port_num, int index, uint16_t *pkey)
{
    (void)c; (void)port_num;
    *pkey = index == 0 ? 0xFFFF : 0;           /* 0xFFFF is byte-order neutral (dare_ibv.c:216 ntohs) */
    return 0;
}
int ibv_query_gid(struct ibv_context *c, uint8_t port_num, int index, union ibv_gid *gid)
{
    (void)port_num; (void)index; memset(gid, 0, sizeof *gid);
    gid->raw[0] = 0xfe; gid->raw[1] = 0x80; gid->raw[15] = (uint8_t)(c->fab_port + 1);
    return 0;
}
struct ibv_pd *ibv_alloc_pd(struct ibv_context *c) { struct ibv_pd *p = calloc(1, sizeof *p); p->context = c; return p; }
int ibv_dealloc_pd(struct ibv_pd *p) { free(p); return 0; }

struct ibv_mr *ibv_reg_mr(struct ibv_pd *pd, void *addr, size_t length, int access)
{
    FAB_LOCK();
    if (next_key >= MAX_MRS) { FAB_UNLOCK(); errno = ENOMEM; return NULL; }
    fmr_t *m = fab_calloc(1, sizeof *m);
    m->pub.context = pd->context; m->pub.pd = pd; m->pub.addr = addr; m->pub.length = length;
    m->pub.lkey = m->pub.rkey = m->pub.handle = next_key;
    m->port = pd->context->fab_port; m->access = access; m->live = 1;
    mrs[next_key++] = m;
    FAB_UNLOCK();
    return &m->pub;
}
int ibv_dereg_mr(struct ibv_mr *mr)
{
    fmr_t *m = (fmr_t *)mr;
    FAB_LOCK();
    if (m->pub.rkey < MAX_MRS && mrs[m->pub.rkey] == m) mrs[m->pub.rkey] = NULL;
    m->live = 0; fab_free(m);
    FAB_UNLOCK();
    return 0;
}

struct ibv_cq *ibv_create_cq(struct ibv_context *c, int cqe, void *ctx, struct ibv_comp_channel *ch, int vec)
{
    (void)ch; (void)vec;
    FAB_LOCK();
    fcq_t *q = fab_calloc(1, sizeof *q);
    q->pub.context = c; q->pub.cq_context = ctx; q->pub.cqe = cqe; q->pub.fab = q;
    q->cap = cqe < 64 ? 64 : (cqe > 4096 ? 4096 : cqe);
    q->ring = fab_malloc(sizeof(struct ibv_wc) * q->cap);
    q->visible_at = fab_malloc(sizeof(uint64_t) * q->cap);
    q->port = c->fab_port;
    FAB_UNLOCK();
    return &q->pub;
}
int ibv_destroy_cq(struct ibv_cq *cq) { fcq_t *q = (fcq_t *)cq; FAB_LOCK(); fab_free(q->ring); fab_free(q->visible_at); fab_free(q); FAB_UNLOCK(); return 0; }

int ibv_poll_cq(struct ibv_cq *cq, int n, struct ibv_wc *wc)
{
    fcq_t *q = (fcq_t *)cq;
#ifdef FAB_PROC
    if (!q->count) return 0;                          /* (the common case of a polling loop: no lock) */
#endif
    FAB_LOCK();
    fport_t *P = &ports[q->port];
    if (!P->last_was_post || !delay_completions) P->epoch++;
    P->last_was_post = 0;
    int k = 0;
    while (k < n && q->count && q->visible_at[q->head] <= P->epoch) {
        wc[k++] = q->ring[q->head];
        q->head = (q->head + 1) % q->cap;
        q->count--;
    }
    FAB_UNLOCK();
#ifndef FAB_PROC
    if (!k && ++q->empty_polls > 200000000ull) {
        fprintf(stderr, "fabric: port %d spins on an empty CQ (a completion that a held/dead peer will never produce)\n", q->port);
        abort();
    }
#endif
    return k;
}

struct ibv_qp *ibv_create_qp(struct ibv_pd *pd, struct ibv_qp_init_attr *ia)
{
    /* find_max_inline (dare_ibv.c:672) probes downward from 1 MiB: accept <= 256 B like a ConnectX */
    if (ia->cap.max_inline_data > 256) { errno = EINVAL; return NULL; }
    FAB_LOCK();
    if (next_qpn >= MAX_QPS) { FAB_UNLOCK(); errno = ENOMEM; return NULL; }
    fqp_t *q = fab_calloc(1, sizeof *q);
    q->pub.context = pd->context; q->pub.pd = pd; q->pub.qp_context = ia->qp_context;
    q->pub.send_cq = ia->send_cq; q->pub.recv_cq = ia->recv_cq; q->pub.qp_type = ia->qp_type;
    q->pub.state = IBV_QPS_RESET; q->pub.qp_num = q->pub.handle = next_qpn; q->pub.fab = q;
    q->port = pd->context->fab_port; q->live = 1; q->cap = ia->cap;
    q->scq = (fcq_t *)ia->send_cq; q->rcq = (fcq_t *)ia->recv_cq;
    q->rq_cap = ia->cap.max_recv_wr ? (int)ia->cap.max_recv_wr : 1;
    q->rq = fab_calloc(q->rq_cap, sizeof *q->rq);
    qps[next_qpn++] = q;
    FAB_UNLOCK();
    return &q->pub;
}
int ibv_destroy_qp(struct ibv_qp *qp)
{
    fqp_t *q = (fqp_t *)qp;
    FAB_LOCK();
    if (q->pub.qp_num < MAX_QPS && qps[q->pub.qp_num] == q) qps[q->pub.qp_num] = NULL;
    q->live = 0; fab_free(q->rq); fab_free(q);
    FAB_UNLOCK();
    return 0;
}

int ibv_modify_qp(struct ibv_qp *qp, struct ibv_qp_attr *a, int mask)
{
    fqp_t *q = (fqp_t *)qp;
    FAB_LOCK();
    if (mask & IBV_QP_STATE) {
        if (a->qp_state == IBV_QPS_RESET) {
            /* RESET clears every attribute and drops queued receives */
            memset(&q->attr, 0, sizeof q->attr);
            q->rq_head = q->rq_count = 0;
        }
        q->pub.state = a->qp_state;
        q->attr.qp_state = a->qp_state;
    }
    if (mask & IBV_QP_PKEY_INDEX) q->attr.pkey_index = a->pkey_index;
    if (mask & IBV_QP_PORT) q->attr.port_num = a->port_num;
    if (mask & IBV_QP_QKEY) q->attr.qkey = a->qkey;
    if (mask & IBV_QP_ACCESS_FLAGS) q->attr.qp_access_flags = a->qp_access_flags;
    if (mask & IBV_QP_AV) q->attr.ah_attr = a->ah_attr;
    if (mask & IBV_QP_PATH_MTU) q->attr.path_mtu = a->path_mtu;
    if (mask & IBV_QP_DEST_QPN) q->attr.dest_qp_num = a->dest_qp_num;
    if (mask & IBV_QP_RQ_PSN) q->attr.rq_psn = a->rq_psn;
    if (mask & IBV_QP_SQ_PSN) q->attr.sq_psn = a->sq_psn;
    if (mask & IBV_QP_MAX_DEST_RD_ATOMIC) q->attr.max_dest_rd_atomic = a->max_dest_rd_atomic;
    if (mask & IBV_QP_MAX_QP_RD_ATOMIC) q->attr.max_rd_atomic = a->max_rd_atomic;
    if (mask & IBV_QP_MIN_RNR_TIMER) q->attr.min_rnr_timer = a->min_rnr_timer;
    if (mask & IBV_QP_TIMEOUT) q->attr.timeout = a->timeout;
    if (mask & IBV_QP_RETRY_CNT) q->attr.retry_cnt = a->retry_cnt;
    if (mask & IBV_QP_RNR_RETRY) q->attr.rnr_retry = a->rnr_retry;
    FAB_UNLOCK();
    return 0;
}
int ibv_query_qp(struct ibv_qp *qp, struct ibv_qp_attr *a, int mask, struct ibv_qp_init_attr *ia)
{
    (void)mask;
    fqp_t *q = (fqp_t *)qp;
    *a = q->attr; a->qp_state = a->cur_qp_state = q->pub.state; a->cap = q->cap;
    if (ia) { memset(ia, 0, sizeof *ia); ia->send_cq = qp->send_cq; ia->recv_cq = qp->recv_cq; ia->cap = q->cap; ia->qp_type = qp->qp_type; }
    return 0;
}

/* ---- UD ---------------------------------------------------------------------------------- */
static void ud_deliver(fqp_t *dst, fqp_t *src, const void *buf, uint32_t len)
{
    if (!dst->live || !ports[dst->port].alive || ports[dst->port].held || ports[src->port].held) return;
    if (dst->pub.state != IBV_QPS_RTR && dst->pub.state != IBV_QPS_RTS) return;
    if (!dst->rq_count) { stats.ud_dropped++; return; }       /* no receive posted: UD drops */
    frecv_t r = dst->rq[dst->rq_head];
    dst->rq_head = (dst->rq_head + 1) % dst->rq_cap; dst->rq_count--;
    if (r.length < len + 40) { stats.ud_dropped++; return; }
    if (fab_zero(dst->port, r.addr, 40) || fab_put(dst->port, r.addr + 40, buf, len)) { stats.ud_dropped++; return; }   /* GRH + payload */
    struct ibv_wc wc; memset(&wc, 0, sizeof wc);
    wc.wr_id = r.wr_id; wc.status = IBV_WC_SUCCESS; wc.opcode = IBV_WC_RECV; wc.byte_len = len + 40;
    wc.qp_num = dst->pub.qp_num; wc.src_qp = src->pub.qp_num; wc.slid = (uint16_t)(src->port + 1);
    wc.wc_flags = 1;                                           /* IBV_WC_GRH */
    cq_push(dst->rcq, &wc, 0);
    stats.ud_msgs++;
}

static int post_send_ud(fqp_t *q, struct ibv_send_wr *wr)
{
    if (q->pub.state != IBV_QPS_RTS) return EINVAL;
    if (wr->opcode != IBV_WR_SEND || wr->num_sge != 1) return EINVAL;
    const void *buf = (const void *)(uintptr_t)wr->sg_list[0].addr;
    uint32_t len = wr->sg_list[0].length;
    struct ibv_ah *ah = wr->wr.ud.ah;
    if (!ah) return EINVAL;
    if (wr->wr.ud.remote_qpn == 0xFFFFFF) {
        for (uint32_t i = 0; i < MAX_QPS; i++) {
            fqp_t *d = qps[i];
            if (d && d->pub.qp_type == IBV_QPT_UD && d->mcast && d->mlid == ah->attr.dlid &&
                !memcmp(d->mgid.raw, ah->attr.grh.dgid.raw, 16))
                ud_deliver(d, q, buf, len);
        }
    } else {
        fqp_t *d = find_qp(wr->wr.ud.remote_qpn);
        if (d && d->pub.qp_type == IBV_QPT_UD && d->port == (int)ah->attr.dlid - 1) ud_deliver(d, q, buf, len);
    }
    if (wr->send_flags & IBV_SEND_SIGNALED) complete(q, wr->wr_id, IBV_WC_SUCCESS, IBV_WR_SEND);
    return 0;
}

/* ---- RC ---------------------------------------------------------------------------------- */
static int post_send_rc(fqp_t *q, struct ibv_send_wr *wr)
{
    if (q->pub.state == IBV_QPS_ERR) {           /* flushed in error, always with a completion */
        complete(q, wr->wr_id, IBV_WC_WR_FLUSH_ERR, wr->opcode);
        return 0;
    }
    if (q->pub.state != IBV_QPS_RTS) return EINVAL;
    if (wr->num_sge != 1 || (wr->opcode != IBV_WR_RDMA_WRITE && wr->opcode != IBV_WR_RDMA_READ)) return EINVAL;
    uint64_t local = wr->sg_list[0].addr;
    uint32_t len = wr->sg_list[0].length;
    int signaled = (wr->send_flags & IBV_SEND_SIGNALED) != 0;
    int tport;
    enum ibv_wc_status st = rc_execute(q, wr->opcode, local, wr->wr.rdma.remote_addr, wr->wr.rdma.rkey, len, &tport);
    rc_finish(q, wr->wr_id, wr->opcode, signaled, st, tport, wr->wr.rdma.remote_addr, len);
    return 0;
}

int ibv_post_send(struct ibv_qp *qp, struct ibv_send_wr *wr, struct ibv_send_wr **bad)
{
    fqp_t *q = (fqp_t *)qp;
    FAB_LOCK();
    ports[q->port].last_was_post = 1;
    for (; wr; wr = wr->next) {
        int rc = q->pub.qp_type == IBV_QPT_UD ? post_send_ud(q, wr) : post_send_rc(q, wr);
        if (rc) { if (bad) *bad = wr; FAB_UNLOCK(); return rc; }
    }
    FAB_UNLOCK();
    return 0;
}

int ibv_post_recv(struct ibv_qp *qp, struct ibv_recv_wr *wr, struct ibv_recv_wr **bad)
{
    fqp_t *q = (fqp_t *)qp;
    FAB_LOCK();
    for (; wr; wr = wr->next) {
        if (q->rq_count == q->rq_cap) { if (bad) *bad = wr; FAB_UNLOCK(); return ENOMEM; }
        frecv_t *r = &q->rq[(q->rq_head + q->rq_count) % q->rq_cap];
        r->wr_id = wr->wr_id; r->addr = wr->sg_list[0].addr; r->length = wr->sg_list[0].length;
        q->rq_count++;
    }
    FAB_UNLOCK();
    return 0;
}

struct ibv_ah *ibv_create_ah(struct ibv_pd *pd, struct ibv_ah_attr *attr)
{
    struct ibv_ah *a = calloc(1, sizeof *a);
    a->context = pd->context; a->pd = pd; a->attr = *attr;
    return a;
}
int ibv_destroy_ah(struct ibv_ah *ah) { free(ah); return 0; }
int ibv_attach_mcast(struct ibv_qp *qp, const union ibv_gid *gid, uint16_t lid)
{
    fqp_t *q = (fqp_t *)qp; q->mcast = 1; q->mgid = *gid; q->mlid = lid; return 0;
}
int ibv_detach_mcast(struct ibv_qp *qp, const union ibv_gid *gid, uint16_t lid)
{
    (void)gid; (void)lid; ((fqp_t *)qp)->mcast = 0; return 0;
}

const char *ibv_wc_status_str(enum ibv_wc_status s)
{
    switch (s) {
    case IBV_WC_SUCCESS: return "success";
    case IBV_WC_WR_FLUSH_ERR: return "Work Request Flushed Error";
    case IBV_WC_RETRY_EXC_ERR: return "transport retry counter exceeded";
    case IBV_WC_REM_ACCESS_ERR: return "remote access error";
    default: return "error";
    }
}

#ifndef FAB_PROC
int fab_pending_ud(int port)
{
    int n = 0;
    for (uint32_t i = 0; i < MAX_QPS; i++) {
        fqp_t *q = qps[i];
        if (q && q->port == port && q->pub.qp_type == IBV_QPT_UD && q->rcq) n += q->rcq->count;
    }
    return n;
}
#endif
