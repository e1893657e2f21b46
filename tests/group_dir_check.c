/* The control files of the one-process-per-server layer (APUS_GROUP_DIR, apus_amd/host/apus_proxy.c): every file carries
 * its writer's stamp {pid, start time}; what an earlier run left in the directory satisfies nobody; a zombie is dead.
 * The functions under test are static: the harness includes the source (linked against libapus_gpu.so for the engine
 * symbols it references; none of them is called).  Prints "ok". */
#include "../apus_amd/host/apus_proxy.c"
#include <sys/wait.h>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    smr_t *s = &g_smr;
    memset(s, 0, sizeof *s);
    snprintf(s->group_dir, sizeof s->group_dir, "%s", argv[1]);
    s->group_size = 3; s->capacity = 3; s->idx = 0;

    /* (1) the stamp of this process: alive, and /proc agrees with itself */
    const g_stamp_t me = g_my_stamp();
    CHECK(me.pid == getpid() && me.start != 0 && g_stamp_alive(&me));
    g_stamp_t other = me; other.start += 1;                       /* the same pid, another process (pid reuse) */
    CHECK(!g_stamp_alive(&other));

    /* (2) a file written here reads back with this stamp */
    uint32_t cfg[2] = { 7, 0x15 }, got[2] = { 0, 0 };
    g_stamp_t who = { 0, 0 };
    CHECK(g_write(s, cfg, sizeof cfg, "leader_2") == 0);
    CHECK(g_read(s, got, sizeof got, "leader_2", &who) == 0 && got[0] == 7 && got[1] == 0x15 && who.pid == me.pid && who.start == me.start);

    /* (3) waiting for a file AS WRITTEN BY a given server: the right stamp is taken at once, a file with another stamp -- what an
     *     earlier run left behind -- is not (the wait times out) */
    s->peer_stamp[0] = me;
    g_stamp_t parent = { getppid(), 0 };                          /* another LIVE process: whoever started this test */
    { char st = 0; CHECK(proc_look(parent.pid, &st, &parent.start) == 0 && g_stamp_alive(&parent)); }
    s->peer_stamp[1] = parent;
    memset(got, 0, sizeof got);
    CHECK(g_wait_from(s, "leader_2", 0, sizeof got, got, 1.0) == 0 && got[0] == 7);
    double t0 = now_s();
    CHECK(g_wait_from(s, "leader_2", 1, sizeof got, got, 0.3) != 0 && now_s() - t0 >= 0.25);
    /* ... and the wait for a writer whose process is gone ends early (round 5: a dead follower must not cost a leader its timeouts) */
    s->peer_stamp[1] = other;
    t0 = now_s();
    CHECK(g_wait_from(s, "leader_2", 1, sizeof got, got, 5.0) != 0 && now_s() - t0 < 1.0);

    /* (4) a hello of a process that is gone: a child writes one and exits; reaped, its stamp is dead and the hello does not count */
    int pfd[2];
    CHECK(pipe(pfd) == 0);
    pid_t c = fork();
    if (c == 0) {
        g_hello_t h; memset(&h, 0, sizeof h); h.ipc.replica = 1;
        const int rc = g_write(s, &h, sizeof h, "replica_1.ipc");
        const g_stamp_t cs = g_my_stamp();
        if (write(pfd[1], &cs, sizeof cs) != (ssize_t)sizeof cs) _exit(3);
        _exit(rc ? 2 : 0);
    }
    g_stamp_t cs;
    CHECK(read(pfd[0], &cs, sizeof cs) == (ssize_t)sizeof cs && cs.pid == c);
    /* (5) ... and while it is a ZOMBIE (exited, not reaped) kill(pid, 0) still succeeds, the stamp says dead */
    for (int i = 0; i < 200; i++) { char st = 0; uint64_t x = 0; if (!proc_look(c, &st, &x) && st == 'Z') break; struct timespec ts = {0, 5000000}; nanosleep(&ts, NULL); }
    CHECK(kill(c, 0) == 0);
    CHECK(!g_stamp_alive(&cs));
    int status = 0;
    CHECK(waitpid(c, &status, 0) == c && WIFEXITED(status) && WEXITSTATUS(status) == 0);
    CHECK(!g_stamp_alive(&cs));
    g_hello_t h;
    CHECK(g_wait_from(s, "replica_1.ipc", s->group_size, sizeof h, &h, 0.3) != 0);     /* a stale hello: nobody home */
    /* (6) a hello of THIS (live) process counts */
    memset(&h, 0, sizeof h); h.ipc.replica = 0;
    CHECK(g_write(s, &h, sizeof h, "replica_0.ipc") == 0);
    CHECK(g_wait_from(s, "replica_0.ipc", s->group_size, sizeof h, &h, 1.0) == 0);
    /* (7) liveness by index */
    s->peer_stamp[2] = cs;
    CHECK(g_alive(s, 0) && !g_alive(s, 1) && !g_alive(s, 2));
    printf("ok\n");
    return 0;
}
