// micro-benchmark: does a kernel pay for fetching its code on every launch?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int N> __device__ __forceinline__ uint32_t chain(uint32_t x) {
    // N straight-line dependent integer ops (no loop): ~N*8 bytes of code
#pragma unroll
    for (int i = 0; i < N; i++) x = x * 1664525u + 1013904223u + (x >> 7);
    return x;
}
template <int N> __global__ void big(uint32_t* out, uint32_t seed) {
    uint64_t t0 = wall_clock64();
    uint32_t x = chain<N>(seed + threadIdx.x);
    uint64_t t1 = wall_clock64();
    if (x == 0x12345) out[1] = x;
    if (threadIdx.x == 0) out[0] = (uint32_t)(t1 - t0);
}
template <int N> void run(const char* name, hipStream_t st, uint32_t* o) {
    uint32_t r[2];
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(big<N>, 1, 64, 0, st, o, 7u + rep); hipStreamSynchronize(st);
        hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);
        printf("%s: %d straight-line ops, in-kernel %.2f us (launch %d)\n", name, N * 3, r[0] / 100.0, rep);
    }
}
int main() {
    uint32_t* o; hipMalloc(&o, 64); hipStream_t st; hipStreamCreate(&st);
    run<100>("small", st, o);
    run<1000>("big  ", st, o);
    run<100>("small", st, o);
    run<1000>("big  ", st, o);
    // back-to-back in a graph: same big kernel 100 times
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < 100; k++) hipLaunchKernelGGL(big<1000>, 1, 64, 0, st, o, 9u);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(a, st); hipGraphLaunch(ge, st); hipEventRecord(b, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b);
    uint32_t r[2]; hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);
    printf("graph of 100 big kernels: %.2f us per kernel, last in-kernel %.2f us\n", ms * 10, r[0] / 100.0);
    return 0;
}
