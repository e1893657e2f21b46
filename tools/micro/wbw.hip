// micro-benchmark: how fast does HBM take the store pattern of the append blocks?  Every workgroup
// of 256 threads writes one "round" of 8 KiB as 16-byte units into each of three buffers at the
// same offset (1 KiB per wave-level store instruction and buffer), optionally reading the 4 KiB of
// payload first.  No sequencing, no dependencies: the ceiling for k_call / k_step's store phase.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void rounds(uint4* a, uint4* b, uint4* c, const uint4* src, size_t units_per_buf, int read_src, int nbuf) {
    const size_t base = (size_t)blockIdx.x * 512;            // 512 units = 8 KiB per round
    for (int k = 0; k < 2; k++) {
        const size_t u = (base + k * 256 + threadIdx.x) % units_per_buf;
        uint4 v = make_uint4(u, u >> 32, k, 7);
        if (read_src && (u & 7) >= 4) v = src[(base / 2 + k * 128 + (threadIdx.x >> 1)) % (units_per_buf / 2)];
        a[u] = v;
        if (nbuf > 1) b[u] = v;
        if (nbuf > 2) c[u] = v;
    }
}
int main() {
    const size_t BUF = 64u << 20, U = BUF / 16;
    uint4 *a, *b, *c, *s;
    hipMalloc(&a, BUF); hipMalloc(&b, BUF); hipMalloc(&c, BUF); hipMalloc(&s, BUF / 2);
    hipMemset(a, 0, BUF); hipMemset(b, 0, BUF); hipMemset(c, 0, BUF); hipMemset(s, 1, BUF / 2);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nbuf : {1, 3}) for (int rd : {0, 1}) for (int blocks : {1024, 7168}) {
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(rounds, blocks, 256, 0, st, a, b, c, s, U, rd, nbuf);
        hipEventRecord(e0, st);
        const int reps = 20;
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(rounds, blocks, 256, 0, st, a, b, c, s, U, rd, nbuf);
        hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000 / reps, wbytes = (double)blocks * 8192 * nbuf, rbytes = rd ? (double)blocks * 4096 : 0;
        printf("%d buffer(s), %s payload read, %5d rounds/launch (%6.1f MB written): %7.2f us/launch -> %5.2f TB/s written, %5.2f TB/s moved\n",
               nbuf, rd ? "with" : "no  ", blocks, wbytes / 1e6, us, wbytes / us / 1e6, (wbytes + rbytes) / us / 1e6);
    }
    return 0;
}
