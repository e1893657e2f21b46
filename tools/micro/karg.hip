// micro-benchmark: cost of reading a large by-value kernel argument vs a device-resident copy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
struct Big { uint64_t* p[120]; uint32_t idx; uint32_t pad; };   // ~1 KB like EngDev
__global__ void by_value(const Big B, uint64_t* out) {
    uint64_t t0 = wall_clock64();
    uint64_t acc = 0;
    const uint32_t i = B.idx;                       // dependent index like E.rep[E.leader]
    acc += (uint64_t)B.p[i] + (uint64_t)B.p[(i + 17) % 120] + (uint64_t)B.p[(i + 40) % 120] + (uint64_t)B.p[(i + 77) % 120] + (uint64_t)B.p[119];
    uint64_t v = *B.p[i];                           // then a dependent HBM load
    uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc + v; }
}
__global__ void by_pointer(const Big* __restrict__ Bp, uint64_t* out) {
    uint64_t t0 = wall_clock64();
    const Big& B = *Bp;
    uint64_t acc = 0;
    const uint32_t i = B.idx;
    acc += (uint64_t)B.p[i] + (uint64_t)B.p[(i + 17) % 120] + (uint64_t)B.p[(i + 40) % 120] + (uint64_t)B.p[(i + 77) % 120] + (uint64_t)B.p[119];
    uint64_t v = *B.p[i];
    uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc + v; }
}
__global__ void empty_k(uint64_t* out) { if (threadIdx.x == 999999) out[3] = 1; }
int main() {
    Big h; uint64_t* buf; uint64_t* o; Big* dB;
    hipMalloc(&buf, 1 << 20); hipMalloc(&o, 64); hipMalloc(&dB, sizeof(Big));
    for (int i = 0; i < 120; i++) h.p[i] = buf + i * 512; h.idx = 3; h.pad = 0;
    hipMemcpy(dB, &h, sizeof(Big), hipMemcpyHostToDevice);
    uint64_t r[4];
    hipStream_t st; hipStreamCreate(&st);
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL(by_value, 1, 1024, 0, st, h, o); hipStreamSynchronize(st);
        hipMemcpy(r, o, 32, hipMemcpyDeviceToHost); printf("by value  : %.2f us in-kernel\n", r[0] / 100.0);
        hipLaunchKernelGGL(by_pointer, 1, 1024, 0, st, dB, o); hipStreamSynchronize(st);
        hipMemcpy(r, o, 32, hipMemcpyDeviceToHost); printf("by pointer: %.2f us in-kernel\n", r[0] / 100.0);
    }
    // graph replay of both, timed by events over 200 launches each
    for (int mode = 0; mode < 3; mode++) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < 200; k++) {
            if (mode == 0) hipLaunchKernelGGL(by_value, 1, 1024, 0, st, h, o);
            else if (mode == 1) hipLaunchKernelGGL(by_pointer, 1, 1024, 0, st, dB, o);
            else hipLaunchKernelGGL(empty_k, 1, 1024, 0, st, o);
        }
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(a, st); hipGraphLaunch(ge, st); hipEventRecord(b, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("graph of 200 %s kernels: %.2f us per kernel\n", mode == 0 ? "by-value" : mode == 1 ? "by-pointer" : "empty", ms * 1000 / 200);
    }
    return 0;
}
