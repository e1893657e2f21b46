// micro-benchmark: dependent global-load latency and a few kernel-shape costs on MI355X
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#include <random>
__global__ void chase(const uint32_t* p, uint32_t start, int n, uint64_t* out) {
    uint32_t i = start; uint64_t t0 = wall_clock64();
    for (int k = 0; k < n; k++) i = p[i];
    uint64_t t1 = wall_clock64();
    out[0] = t1 - t0; out[1] = i;
}
__global__ void chase_blocks(const uint32_t* p, int n, uint64_t* out) {   // every block/thread chases concurrently
    uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * 977u % (1u << 24); uint64_t t0 = wall_clock64();
    for (int k = 0; k < n; k++) i = p[i];
    uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; } if (i == 0xFFFFFFFF) out[2] = i;
}
__global__ void store_then_load(uint32_t* q, uint64_t* out) {   // same-block RAW through global memory
    uint64_t t0 = wall_clock64();
    q[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t v = q[(threadIdx.x + 64) % blockDim.x];
    __syncthreads();
    uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
int main() {
    const size_t N = 1u << 24;  // 64 MB of uint32
    std::vector<uint32_t> h(N); for (size_t i = 0; i < N; i++) h[i] = (uint32_t)i;
    std::mt19937 g(1); std::shuffle(h.begin(), h.end(), g);
    // make it one cycle
    std::vector<uint32_t> perm(N); for (size_t i = 0; i < N; i++) perm[h[i]] = h[(i + 1) % N];
    uint32_t* d; uint64_t* o; uint32_t* q;
    hipMalloc(&d, N * 4); hipMalloc(&o, 64); hipMalloc(&q, 4096 * 4);
    hipMemcpy(d, perm.data(), N * 4, hipMemcpyHostToDevice);
    uint64_t r[4];
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(chase, 1, 1, 0, 0, d, 123u + rep, 64, o); hipDeviceSynchronize();
        hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
        printf("single-thread chase: %.1f ns per dependent load (random 64 MB)\n", r[0] * 10.0 / 64);
    }
    for (int blocks : {256, 1024}) {
        hipLaunchKernelGGL(chase_blocks, blocks, 256, 0, 0, d, 16, o); hipDeviceSynchronize();
        hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
        printf("%d blocks x 256 threads chasing: %.1f ns per dependent load\n", blocks, r[0] * 10.0 / 16);
    }
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(store_then_load, 1, 1024, 0, 0, q, o); hipDeviceSynchronize();
        hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
        printf("store -> barrier -> load -> barrier (1024 threads): %.2f us\n", r[0] / 100.0);
    }
    // small-buffer chase (L2 resident)
    hipLaunchKernelGGL(chase, 1, 1, 0, 0, q, 0u, 1, o);
    return 0;
}
