// Can the HOST store into / load from device memory directly (large BAR)?  And what does a host store cost until a resident
// kernel sees it, compared with the kernel polling pinned host memory?  hipcc --offload-arch=gfx950 -O2 -o bar bar.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <signal.h>
#include <setjmp.h>
static sigjmp_buf jb;
static void on_segv(int s) { (void)s; siglongjmp(jb, 1); }
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
__global__ void echo(volatile uint64_t *in, volatile uint64_t *out, uint64_t n)
{
    for (uint64_t i = 1; i <= n; i++) {
        while (__hip_atomic_load((uint64_t *)in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != i) ;
        __hip_atomic_store((uint64_t *)out, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static int try_kind(const char *name, void *dev_in, volatile uint64_t *host_view_in, volatile uint64_t *pinned_out, void *pinned_out_dev)
{
    const uint64_t N = 20000;
    *pinned_out = 0;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, st, (volatile uint64_t *)dev_in, (volatile uint64_t *)pinned_out_dev, N);
    double t0 = now();
    for (uint64_t i = 1; i <= N; i++) {
        *host_view_in = i;
        __sync_synchronize();
        while (*pinned_out != i) if (now() - t0 > 20) { printf("%s: timeout at %llu\n", name, (unsigned long long)i); return 1; }
    }
    double dt = now() - t0;
    hipStreamSynchronize(st);
    printf("%s: host store -> kernel sees it -> kernel's store -> host sees it: %.2f us per round trip\n", name, dt / N * 1e6);
    return 0;
}
int main()
{
    uint64_t *pin_in, *pin_out, *pin_in_dev, *pin_out_dev;
    hipHostMalloc((void **)&pin_in, 4096, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostMalloc((void **)&pin_out, 4096, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostGetDevicePointer((void **)&pin_in_dev, pin_in, 0); hipHostGetDevicePointer((void **)&pin_out_dev, pin_out, 0);
    *pin_in = 0;
    try_kind("pinned host memory (kernel polls over PCIe)", pin_in_dev, pin_in, pin_out, pin_out_dev);
    const unsigned kinds[3] = { hipDeviceMallocFinegrained, hipDeviceMallocUncached, hipDeviceMallocDefault };
    const char *names[3] = { "device memory, fine-grained (host stores through the BAR)", "device memory, uncached", "device memory, default" };
    for (int k = 0; k < 3; k++) {
        uint64_t *d = nullptr;
        if (hipExtMallocWithFlags((void **)&d, 4096, kinds[k]) != hipSuccess) { printf("%s: allocation failed\n", names[k]); continue; }
        hipMemset(d, 0, 4096); hipDeviceSynchronize();
        signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1)) { printf("%s: the host cannot touch it (fault)\n", names[k]); continue; }
        volatile uint64_t probe = *(volatile uint64_t *)d;           /* faults here when there is no host mapping */
        (void)probe;
        try_kind(names[k], d, (volatile uint64_t *)d, pin_out, pin_out_dev);
    }
    return 0;
}
