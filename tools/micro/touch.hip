// micro-benchmark: cost of the FIRST access a kernel makes to a region, as a function of what
// the previous kernel did to it (nothing / read it / wrote it), and the one-CU gather patterns
// k_sequence uses.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/micro/touch tools/micro/touch.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")

__global__ void wr(uint64_t* buf, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = i;
}
__global__ void rd(const uint64_t* buf, size_t n, uint64_t* sink) {
    uint64_t a = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += buf[i];
    if (a == 0x1234567) *sink = a;
}
// every thread: one coalesced 8-B load, then a dependent 16-B load from a second region (dir -> ring)
__global__ void first_touch(const uint64_t* a, const uint64_t* b, uint32_t bmask, uint64_t* out, uint64_t* sink) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t t0 = wall_clock64();
    const uint64_t v = a[g];
    WAIT();
    const uint64_t t1 = wall_clock64();
    const uint64_t w = b[((uint32_t)v * 8u) & bmask];
    WAIT();
    const uint64_t t2 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t1; }
    if (w == 0x1234567) *sink = w;
}
// k_sequence phase A, pattern 1: thread t reads 8 consecutive 16-B pieces of "its" 128-B line
__global__ __launch_bounds__(1024) void gather_strided(const uint4* p, uint64_t* out, uint64_t* sink) {
    const uint64_t t0 = wall_clock64();
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) { const uint4 v = p[threadIdx.x * 8 + i]; acc += v.x + v.y + v.z + v.w; }
    WAIT();
    __syncthreads();
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0x1234567) *sink = acc;
}
// pattern 2: the same 128 KB read fully coalesced (lane-contiguous), 8 passes
__global__ __launch_bounds__(1024) void gather_coalesced(const uint4* p, uint64_t* out, uint64_t* sink) {
    const uint64_t t0 = wall_clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint4 v = p[i * 1024 + threadIdx.x]; acc += v.x + v.y + v.z + v.w; }
    WAIT();
    __syncthreads();
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0x1234567) *sink = acc;
}
// pattern 3: the same 128 KB spread over `gridDim.x` blocks of 64 threads
__global__ void gather_spread(const uint4* p, uint64_t* out, uint64_t* sink) {
    const uint64_t t0 = wall_clock64();
    uint32_t acc = 0;
    const uint32_t per = 8192 / (gridDim.x * blockDim.x);
    for (uint32_t i = 0; i < per; i++) { const uint4 v = p[(blockIdx.x * per + i) * blockDim.x + threadIdx.x]; acc += v.x + v.y; }
    WAIT();
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x1234567) *sink = acc;
}

static void report(const char* tag, uint64_t* o, int blocks) {
    std::vector<uint64_t> h(blocks * 2); hipMemcpy(h.data(), o, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> a, b; for (int i = 0; i < blocks; i++) { a.push_back(h[2 * i] / 100.0); b.push_back(h[2 * i + 1] / 100.0); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-44s first load  min %.2f p50 %.2f max %.2f us | dependent load  min %.2f p50 %.2f max %.2f us\n", tag,
           a[0], a[blocks / 2], a[blocks - 1], b[0], b[blocks / 2], b[blocks - 1]);
}

int main() {
    const size_t N = (64u << 20) / 8;
    uint64_t *A, *B, *C, *o, *sink;
    hipMalloc(&A, N * 8); hipMalloc(&B, N * 8); hipMalloc(&C, N * 8); hipMalloc(&o, 1 << 16); hipMalloc(&sink, 64);
    hipMemset(A, 0, N * 8); hipMemset(B, 0, N * 8); hipMemset(C, 0, N * 8);
    hipLaunchKernelGGL(wr, 1024, 256, 0, 0, A, N); hipLaunchKernelGGL(wr, 1024, 256, 0, 0, B, N);
    hipDeviceSynchronize();
    const int blocks = 256, th = 256; const uint32_t bmask = (uint32_t)(N - 1);
    const size_t touched = (size_t)blocks * th;         // words of A the probe reads
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(first_touch, blocks, th, 0, 0, A, B, bmask, o, sink); hipDeviceSynchronize();
        report(rep ? "idle, second launch" : "idle, first launch", o, blocks);
    }
    // previous kernel (back to back, same stream) read the same words
    hipLaunchKernelGGL(rd, 1024, 256, 0, 0, A, touched, sink);
    hipLaunchKernelGGL(first_touch, blocks, th, 0, 0, A, B, bmask, o, sink); hipDeviceSynchronize();
    report("after a kernel that READ the words", o, blocks);
    // previous kernel wrote the same words (dirty in some XCD's L2)
    hipLaunchKernelGGL(wr, 1024, 256, 0, 0, A, touched);
    hipLaunchKernelGGL(first_touch, blocks, th, 0, 0, A, B, bmask, o, sink); hipDeviceSynchronize();
    report("after a kernel that WROTE the words", o, blocks);
    // previous kernel wrote 32 MB elsewhere (a big copy just finished)
    hipLaunchKernelGGL(wr, 1024, 256, 0, 0, C, N / 2);
    hipLaunchKernelGGL(first_touch, blocks, th, 0, 0, A, B, bmask, o, sink); hipDeviceSynchronize();
    report("after a kernel that wrote 32 MB elsewhere", o, blocks);
    // previous kernel wrote the words AND 32 MB elsewhere
    hipLaunchKernelGGL(wr, 1024, 256, 0, 0, C, N / 2);
    hipLaunchKernelGGL(wr, 1024, 256, 0, 0, A, touched);
    hipLaunchKernelGGL(first_touch, blocks, th, 0, 0, A, B, bmask, o, sink); hipDeviceSynchronize();
    report("after 32 MB elsewhere + WROTE the words", o, blocks);
    for (int b2 : {16, 64, 1024}) {
        hipLaunchKernelGGL(first_touch, b2, th, 0, 0, A, B, bmask, o, sink); hipDeviceSynchronize();
        char tag[64]; snprintf(tag, sizeof tag, "idle, %d blocks", b2); report(tag, o, b2);
    }
    uint64_t r[64];
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(gather_strided, 1, 1024, 0, 0, (const uint4*)C, o, sink); hipDeviceSynchronize();
        hipMemcpy(r, o, 8, hipMemcpyDeviceToHost); printf("gather 128 KB, 1 block, thread-strided 8x16 B: %.2f us\n", r[0] / 100.0);
        hipLaunchKernelGGL(gather_coalesced, 1, 1024, 0, 0, (const uint4*)C + (1 << 20), o, sink); hipDeviceSynchronize();
        hipMemcpy(r, o, 8, hipMemcpyDeviceToHost); printf("gather 128 KB, 1 block, coalesced:             %.2f us\n", r[0] / 100.0);
        for (int g : {16, 64}) {
            hipLaunchKernelGGL(gather_spread, g, 64, 0, 0, (const uint4*)C + (2 << 20) + g * 65536, o, sink); hipDeviceSynchronize();
            hipMemcpy(r, o, 8 * g, hipMemcpyDeviceToHost); uint64_t mx = 0; for (int i = 0; i < g; i++) mx = std::max(mx, r[i]);
            printf("gather 128 KB, %d blocks x 64 threads:          %.2f us (slowest block)\n", g, mx / 100.0);
        }
    }
    return 0;
}
