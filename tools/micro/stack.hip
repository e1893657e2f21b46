// micro-benchmark: how long a wave waits for its stores to be acknowledged (s_waitcnt vmcnt(0)),
// for the store shapes the engine uses; and kernel time per launch when the region moves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
// mode 0: one 16-B store per thread, lanes contiguous
// mode 1: 32-B record per thread as two 16-B stores (each instruction covers every other 16 B)
// mode 2: four records per thread (ILP 4), like k_apply
// mode 3: four 1-byte stores per thread, 128 B apart (reply bytes)
// mode 4: 64-B per thread as four 16-B stores (lane-contiguous 64 B)
__global__ void st(uint8_t* base, int mode, uint64_t* out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 v = make_uint4(g, g + 1, g + 2, g + 3);
    const uint64_t t0 = wall_clock64();
    if (mode == 0) ((uint4*)base)[g] = v;
    else if (mode == 1) { uint4* p = (uint4*)(base + (size_t)g * 32); p[0] = v; p[1] = v; }
    else if (mode == 2) { for (int k = 0; k < 4; k++) { uint4* p = (uint4*)(base + ((size_t)k * gridDim.x * blockDim.x + g) * 32); p[0] = v; p[1] = v; } }
    else if (mode == 3) { for (int k = 0; k < 4; k++) base[((size_t)g * 4 + k) * 128 + 29] = 1; }
    else { uint4* p = (uint4*)(base + (size_t)g * 64); p[0] = v; p[1] = v; p[2] = v; p[3] = v; }
    WAIT();
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    uint8_t* buf; uint64_t* o;
    const size_t SZ = 1ull << 30;
    hipMalloc(&buf, SZ); hipMalloc(&o, 1 << 16); hipMemset(buf, 0, SZ);
    hipStream_t s; hipStreamCreate(&s);
    const char* names[5] = {"16 B/thread contiguous", "32-B record, 2 stores", "4 x 32-B records (ILP 4)", "4 x 1-byte, 128 B apart", "64 B/thread, 4 stores"};
    for (int blocks : {1, 64, 256, 768}) for (int mode = 0; mode < 5; mode++) {
        std::vector<double> p50s, maxs; float ktime = 0;
        // 20 launches back to back over moving (cold) regions, per-block ack wait of the last one
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
        for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL(st, blocks, 256, 0, s, buf + (size_t)rep * (48u << 20), mode, o);
        hipEventRecord(b, s); hipStreamSynchronize(s); hipEventElapsedTime(&ktime, a, b);
        std::vector<uint64_t> h(blocks); hipMemcpy(h.data(), o, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%4d blocks  %-26s ack wait  min %.2f p50 %.2f max %.2f us   (%.2f us per launch, eager)\n", blocks, names[mode],
               h[0] / 100.0, h[blocks / 2] / 100.0, h[blocks - 1] / 100.0, ktime * 1000 / 20);
    }
    return 0;
}
