// micro-benchmark: many blocks reading a ~1 KB by-value kernel argument (dynamic index, like
// E.rep[E.leader].dir_off) vs the same struct in device memory; per-block in-kernel time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
struct Big { uint64_t* p[120]; uint32_t idx; uint32_t pad; };
#define WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
__global__ void by_value(const Big B, uint64_t* out, uint32_t late) {
    const uint64_t t0 = wall_clock64();
    const uint32_t i = B.idx;
    const uint64_t v = B.p[i][blockIdx.x * blockDim.x + threadIdx.x];       // first kernarg lines
    WAIT();
    const uint64_t t1 = wall_clock64();
    const uint64_t w = B.p[(i + late) % 120][blockIdx.x * blockDim.x + threadIdx.x];   // a kernarg line not touched yet
    WAIT();
    const uint64_t t2 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t1; }
    if (v + w == 0x1234567) out[0] = 0;
}
__global__ void by_pointer(const Big* __restrict__ Bp, uint64_t* out, uint32_t late) {
    const uint64_t t0 = wall_clock64();
    const uint32_t i = Bp->idx;
    const uint64_t v = Bp->p[i][blockIdx.x * blockDim.x + threadIdx.x];
    WAIT();
    const uint64_t t1 = wall_clock64();
    const uint64_t w = Bp->p[(i + late) % 120][blockIdx.x * blockDim.x + threadIdx.x];
    WAIT();
    const uint64_t t2 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t1; }
    if (v + w == 0x1234567) out[0] = 0;
}
static void report(const char* tag, uint64_t* o, int blocks) {
    std::vector<uint64_t> h(blocks * 2); hipMemcpy(h.data(), o, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> a, b; for (int i = 0; i < blocks; i++) { a.push_back(h[2 * i] / 100.0); b.push_back(h[2 * i + 1] / 100.0); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-34s first  min %.2f p50 %.2f max %.2f us | late line  min %.2f p50 %.2f max %.2f us\n", tag,
           a[0], a[blocks / 2], a[blocks - 1], b[0], b[blocks / 2], b[blocks - 1]);
}
int main() {
    Big h; uint64_t* buf; uint64_t* o; Big* dB;
    hipMalloc(&buf, 120ull << 21); hipMalloc(&o, 1 << 16); hipMalloc(&dB, sizeof(Big));
    hipMemset(buf, 0, 120ull << 21);
    for (int i = 0; i < 120; i++) h.p[i] = buf + (size_t)i * (1 << 18); h.idx = 3; h.pad = 0;
    hipMemcpy(dB, &h, sizeof(Big), hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    for (int blocks : {1, 64, 256, 1024}) {
        for (int rep = 0; rep < 2; rep++) {
            char tag[64];
            hipLaunchKernelGGL(by_value, blocks, 256, 0, st, h, o, 77u); hipStreamSynchronize(st);
            snprintf(tag, sizeof tag, "by value,   %4d blocks", blocks); report(tag, o, blocks);
            hipLaunchKernelGGL(by_pointer, blocks, 256, 0, st, dB, o, 77u); hipStreamSynchronize(st);
            snprintf(tag, sizeof tag, "by pointer, %4d blocks", blocks); report(tag, o, blocks);
        }
    }
    // graph replay: 100 launches each, 256 blocks
    for (int mode = 0; mode < 2; mode++) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < 100; k++) {
            if (mode == 0) hipLaunchKernelGGL(by_value, 256, 256, 0, st, h, o, 77u);
            else hipLaunchKernelGGL(by_pointer, 256, 256, 0, st, dB, o, 77u);
        }
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(a, st); hipGraphLaunch(ge, st); hipEventRecord(b, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("graph of 100 %s kernels (256 blocks): %.2f us per kernel\n", mode == 0 ? "by-value" : "by-pointer", ms * 1000 / 100);
        report(mode == 0 ? "  in graph, by value" : "  in graph, by pointer", o, 256);
    }
    return 0;
}
