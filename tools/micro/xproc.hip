// micro-benchmark behind the replica kernels (apus_replica.h): what does a hand-off between two
// PROCESSES cost when each runs its own resident kernel and they talk through HIP-IPC-mapped
// uncached device memory (one device here: the test mode of a one-replica-per-GPU group; on a
// multi-GPU node the same words cross xGMI -- the analogue of rc_get_loggp_params,
// /root/reference/src/dare/dare_ibv_rc.c:3323-3739)?
//   1. 8-byte doorbell ping-pong between the two kernels (system-scope relaxed store / load):
//      round trip p50 / p99  -> the latency floor of one consensus round (R2 out, R3 back)
//   2. payload + doorbell: N x 16-B write-through stores into the peer's buffer, drained, then the
//      doorbell; the peer checks the last payload word after the doorbell (stale count)
//   3. store shapes a follower issues per entry, single process: scattered 1-byte stores (reply
//      bytes) and consecutive 1-byte stores (ACK byte map) over 1M "entries" of 128 B
// usage: xproc [device_parent] [device_child]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <algorithm>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));

__device__ static inline bool wait_eq(uint64_t *p, uint64_t want, uint64_t max_polls)
{
    for (uint64_t i = 0; i < max_polls; i++) {
        if (__hip_atomic_load(p, RLX_SYS) >= want) return true;
    }
    return false;
}

// side 0 initiates; mine = my box (peer writes it), theirs = the peer's box (I write it)
__global__ void pingpong(uint64_t *mine, uint64_t *theirs, uint8_t *their_buf, uint8_t *my_buf, int side, int iters,
                         int payload_units, uint32_t *lat, uint32_t *stale)
{
    const int tid = threadIdx.x;
    __shared__ int s_fail;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int i = 1; i <= iters; i++) {
        uint64_t t0 = 0;
        if (side == 1) {
            if (tid == 0 && !wait_eq(mine, (uint64_t)i, 1ull << 26)) s_fail = 1;
            __syncthreads();
            if (s_fail) return;
            if (payload_units && tid < payload_units) {
                // the payload must be there once the doorbell was seen
                const uint32_t w = __hip_atomic_load((uint32_t *)(my_buf + 16 * tid), RLX_SYS);
                if (w != (uint32_t)i) atomicAdd(stale, 1u);
            }
        }
        if (tid == 0) t0 = wall_clock64();
        if (payload_units) {
            if (tid < payload_units) {
                v4u_t d = {(unsigned)i, (unsigned)tid, 0u, 0u};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(their_buf + 16 * tid), "v"(d) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (tid == 0) __hip_atomic_store(theirs, (uint64_t)i, RLX_SYS);
        if (side == 0) {
            if (tid == 0 && !wait_eq(mine, (uint64_t)i, 1ull << 26)) s_fail = 1;
            __syncthreads();
            if (s_fail) return;
            if (payload_units && tid < payload_units) {
                const uint32_t w = __hip_atomic_load((uint32_t *)(my_buf + 16 * tid), RLX_SYS);
                if (w != (uint32_t)i) atomicAdd(stale, 1u);
            }
            if (tid == 0) lat[i - 1] = (uint32_t)(wall_clock64() - t0);
        }
        __syncthreads();
    }
}

__global__ void byte_stores(uint8_t *ring, uint8_t *ackb, int n, int mode)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) ring[(size_t)i * 128 + 29] = 1;                       // reply byte of entry i
    else if (mode == 1) ackb[i] = 1;                                       // ACK byte map
    else if (mode == 2) { ring[(size_t)i * 128 + 29] = 1; ackb[i] = 1; }
    else if (mode == 3) __hip_atomic_store(ring + (size_t)i * 128 + 29, (uint8_t)1, RLX_SYS);
    else if (mode == 4) atomicOr((uint32_t *)(ackb + 4 * (size_t)i), 2u);   // the ACK-word form
}

static void report(const char *what, std::vector<uint32_t> &l, int khz)
{
    std::sort(l.begin(), l.end());
    auto us = [&](size_t k) { return (double)l[k] * 1000.0 / khz; };
    printf("%-58s p50 %.2f us  p99 %.2f us  min %.2f us\n", what, us(l.size() / 2), us(l.size() * 99 / 100), us(0));
}

int main(int argc, char **argv)
{
    const int dev_p = argc > 1 ? atoi(argv[1]) : 0, dev_c = argc > 2 ? atoi(argv[2]) : 0;
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 2;
    const int ITERS = 2000;
    const pid_t pid = fork();                          // before any HIP call
    const int side = pid == 0 ? 1 : 0;
    CHK(hipSetDevice(side ? dev_c : dev_p));
    uint64_t *box; uint8_t *buf;
    CHK(hipExtMallocWithFlags((void **)&box, 4096, hipDeviceMallocUncached));
    CHK(hipExtMallocWithFlags((void **)&buf, 1 << 20, getenv("XPROC_COARSE") ? 0x1 /* hipDeviceMallocDefault */ : hipDeviceMallocFinegrained));
    CHK(hipMemset(box, 0, 4096)); CHK(hipMemset(buf, 0, 1 << 20));
    hipIpcMemHandle_t mine[2], theirs[2];
    CHK(hipIpcGetMemHandle(&mine[0], box)); CHK(hipIpcGetMemHandle(&mine[1], buf));
    const int wr = side ? c2p[1] : p2c[1], rd = side ? p2c[0] : c2p[0];
    if (write(wr, mine, sizeof mine) != (ssize_t)sizeof mine || read(rd, theirs, sizeof theirs) != (ssize_t)sizeof theirs) return 2;
    uint64_t *pbox; uint8_t *pbuf;
    CHK(hipIpcOpenMemHandle((void **)&pbox, theirs[0], hipIpcMemLazyEnablePeerAccess));
    CHK(hipIpcOpenMemHandle((void **)&pbuf, theirs[1], hipIpcMemLazyEnablePeerAccess));
    uint32_t *lat, *stale;
    CHK(hipMalloc(&lat, ITERS * 4)); CHK(hipMalloc(&stale, 64)); CHK(hipMemset(stale, 0, 64));
    int khz = 100000;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, side ? dev_c : dev_p);
    char token = 'x';
    const int shapes[4] = {0, 1, 64, 256};
    for (int s = 0; s < 4; s++) {
        // both sides reset their boxes, then a host barrier through the pipes, then both launch
        CHK(hipMemset(box, 0, 4096)); CHK(hipDeviceSynchronize());
        if (write(wr, &token, 1) != 1 || read(rd, &token, 1) != 1) return 2;
        hipLaunchKernelGGL(pingpong, 1, 256, 0, 0, box, pbox, pbuf, buf, side, ITERS, shapes[s], lat, stale);
        CHK(hipDeviceSynchronize());
        if (side == 0) {
            std::vector<uint32_t> l(ITERS);
            CHK(hipMemcpy(l.data(), lat, ITERS * 4, hipMemcpyDeviceToHost));
            uint32_t st[2]; CHK(hipMemcpy(st, stale, 8, hipMemcpyDeviceToHost));
            char what[128];
            snprintf(what, sizeof what, "2 processes, dev %d <-> %d, round trip, %4d B payload + doorbell", dev_p, dev_c, shapes[s] * 16);
            l.erase(l.begin(), l.begin() + 100);
            bool ok = true; for (auto v : l) if (!v) ok = false;
            if (!ok) printf("%s: TIMED OUT (kernels of two processes did not make progress together)\n", what);
            else report(what, l, khz);
            if (shapes[s]) printf("    stale payload words seen by the initiator: %u\n", st[0]);
        } else {
            uint32_t st[2]; CHK(hipMemcpy(st, stale, 8, hipMemcpyDeviceToHost));
            if (shapes[s] && st[0]) printf("    stale payload words seen by the responder: %u\n", st[0]);
        }
        if (write(wr, &token, 1) != 1 || read(rd, &token, 1) != 1) return 2;
    }
    if (side == 1) { hipIpcCloseMemHandle(pbox); hipIpcCloseMemHandle(pbuf); _exit(0); }
    int stt; waitpid(pid, &stt, 0);

    // 3. store shapes of a follower's per-entry work (single process)
    const int N = 1 << 20;
    uint8_t *ring, *ackb;
    for (int mt = 0; mt < 2; mt++) {
        if (mt == 0) { CHK(hipMalloc(&ring, (size_t)N * 128)); CHK(hipMalloc(&ackb, (size_t)N * 4)); }
        else { CHK(hipExtMallocWithFlags((void **)&ring, (size_t)N * 128, hipDeviceMallocFinegrained)); CHK(hipExtMallocWithFlags((void **)&ackb, (size_t)N * 4, hipDeviceMallocUncached)); }
        CHK(hipMemset(ring, 0, (size_t)N * 128)); CHK(hipMemset(ackb, 0, (size_t)N * 4));
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        const char *names[5] = {"scattered 1-B stores (reply bytes), one per 128-B entry", "consecutive 1-B stores (ACK byte map)", "both",
                                "scattered 1-B system-scope stores", "atomicOr per entry (ACK words)"};
        for (int mode = 0; mode < 5; mode++) {
            hipLaunchKernelGGL(byte_stores, N / 256, 256, 0, 0, ring, ackb, N, mode);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a, 0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(byte_stores, N / 256, 256, 0, 0, ring, ackb, N, mode);
            CHK(hipEventRecord(b, 0)); CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            printf("%-24s %-58s %.1f us per 1M entries (%.2f G entries/s)\n", mt ? "[fine-grained / uncached]" : "[hipMalloc]", names[mode], ms * 200.0, N / (ms / 5 * 1e-3) / 1e9);
        }
        hipFree(ring); hipFree(ackb);
    }
    return 0;
}
