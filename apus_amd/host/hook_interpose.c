/*
 * hook_interpose.c -- the LD_PRELOAD side of the drop-in (libapus_interpose.so).
 *
 * Same four libc entry points, same conditions and the same proxy_* calls as the
 * reference's interposer (/root/reference/src/spec_hooks.cpp:102-178: accept,
 * accept4, close, read; every socket fd is reported to the proxy).  Two things
 * differ on purpose:
 *   - initialisation happens in an ELF constructor instead of a hooked
 *     __libc_start_main: on glibc >= 2.34 passing a non-NULL `init` selects the
 *     legacy start-up path and skips the application's own constructors
 *     (SURVEY.md section 9, Q9);
 *   - the engine's own threads (HIP runtime, DARE thread) close and read
 *     descriptors too; they are kept out with a thread-local guard around
 *     proxy_init and by the proxy's is_inner check.
 *
 * Usage (same variables as benchmarks/run.sh:23-41):
 *   server_idx=0 group_size=3 config_path=nodes.cfg \
 *   LD_PRELOAD=.../libapus_interpose.so redis-server --port 6379
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include "apus_smr.h"

static struct proxy_node_t *proxy;
static __thread int guard;
static int initialising;

/* APUS_DEBUG: a backtrace on SIGSEGV (the application may install its own handler later) */
static void on_segv(int sig)
{
    void *bt[64];
    int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void apus_hook_init(void)
{
    if (getenv("APUS_DEBUG")) signal(SIGSEGV, on_segv);
    const char *cfg = getenv("config_path");
    const char *st = getenv("server_type");
    if (!getenv("server_idx") && !(st && !strcmp(st, "join"))) return;      /* not an APUS-managed process (a joiner has no index yet) */
    initialising = 1;
    guard = 1;
    proxy = proxy_init(cfg ? cfg : "", NULL);       /* tern_init_func, spec_hooks.cpp:22-45 */
    {
        /* The ROCm libraries this one pulls in setenv() a few variables (GLOG_*) in their own
         * constructors, so `environ` is a glibc-owned heap array by the time main() runs.  redis 2.8.17's
         * setproctitle (spt_copyenv) then clearenv()s -- which frees that array -- and goes on reading
         * it.  Hand main() an array glibc does not own, as if nobody had touched the environment. */
        extern char **environ;
        int n = 0;
        while (environ && environ[n]) n++;
        char **copy = malloc(((size_t)n + 1) * sizeof *copy);
        if (copy) { for (int i = 0; i <= n; i++) copy[i] = environ[i]; environ = copy; }
    }
    guard = 0;
    initialising = 0;
    if (!proxy) fprintf(stderr, "[apus] proxy_init failed: hooks are inert\n");
}

/* the application exits (redis: exit() after SHUTDOWN): stop the DARE thread, which drains the
 * persistent kernel and, if asked to (APUS_PROXY_DUMP), leaves the replicas' state behind */
__attribute__((destructor)) static void apus_hook_fini(void)
{
    if (!proxy) return;
    guard = 1;
    struct proxy_node_t *p = proxy;
    proxy = NULL;
    apus_proxy_shutdown(p);
}

static int is_sock(int fd)
{
    struct stat sb;
    if (fstat(fd, &sb)) return 0;
    return (sb.st_mode & S_IFMT) == S_IFSOCK;
}

int accept(int socket, struct sockaddr *address, socklen_t *address_len)
{
    static int (*real)(int, struct sockaddr *, socklen_t *);
    if (!real) real = dlsym(RTLD_NEXT, "accept");
    int ret = real(socket, address, address_len);
    if (ret >= 0 && proxy && !guard && !initialising && is_sock(ret)) proxy_on_accept(proxy, ret);
    return ret;
}

int accept4(int sockfd, struct sockaddr *addr, socklen_t *addrlen, int flags)
{
    static int (*real)(int, struct sockaddr *, socklen_t *, int);
    if (!real) real = dlsym(RTLD_NEXT, "accept4");
    int ret = real(sockfd, addr, addrlen, flags);
    if (ret >= 0 && proxy && !guard && !initialising && is_sock(ret)) proxy_on_accept(proxy, ret);
    return ret;
}

int close(int fildes)
{
    static int (*real)(int);
    if (!real) real = dlsym(RTLD_NEXT, "close");
    if (proxy && !guard && !initialising && is_sock(fildes)) proxy_on_close(proxy, fildes);
    return real(fildes);
}

ssize_t read(int fd, void *buf, size_t count)
{
    static ssize_t (*real)(int, void *, size_t);
    if (!real) real = dlsym(RTLD_NEXT, "read");
    ssize_t n = real(fd, buf, count);
    if (n > 0 && proxy && !guard && !initialising && is_sock(fd)) proxy_on_read(proxy, buf, n, fd);
    return n;
}
