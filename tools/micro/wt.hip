// micro-benchmark: what does a WRITE-THROUGH store cost the wavefront that issues it?  Every wavefront issues `per` rounds
// of K 16-byte-per-lane stores (1 KiB per instruction, contiguous) of one kind, drains (s_waitcnt vmcnt(0)), and
// notes the time to ISSUE the K stores and the time to DRAIN them.  Kinds: plain, nt (streaming), sc1 (agent scope,
// write-through), sc0 sc1 (system scope).  hipcc --offload-arch=gfx950 -O2 -o wt wt.hip ; ./wt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
template <int KIND> __device__ inline void st(uint8_t *p, v4u_t d)
{
    if (KIND == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(d) : "memory");
    if (KIND == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(d) : "memory");
    if (KIND == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(d) : "memory");
    if (KIND == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(d) : "memory");
}
template <int KIND, int K>
__global__ __launch_bounds__(256) void wt(uint8_t *buf, size_t bytes, int per, uint64_t *out)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * wpb + wave, nw = (uint64_t)gridDim.x * wpb;
    uint64_t t_issue = 0, t_drain = 0;
    v4u_t d = {lane, wave, blockIdx.x, 7};
    for (int r = 0; r < per; r++) {
        uint8_t *p = buf + ((gw + (uint64_t)r * nw) * (uint64_t)K * 1024) % (bytes - (uint64_t)K * 1024) + lane * 16;
        const uint64_t t0 = wall_clock64();
#pragma unroll
        for (int k = 0; k < K; k++) st<KIND>(p + k * 1024, d);
        const uint64_t t1 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t2 = wall_clock64();
        t_issue += t1 - t0; t_drain += t2 - t1;
    }
    if (lane == 0) { atomicAdd((unsigned long long *)&out[0], (unsigned long long)t_issue); atomicAdd((unsigned long long *)&out[1], (unsigned long long)t_drain); }
}
template <int KIND, int K> static void run(const char *name, uint8_t *buf, size_t bytes, uint64_t *out, int blocks, int threads)
{
    const int per = 200;
    hipMemset(out, 0, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((wt<KIND, K>), blocks, threads, 0, 0, buf, bytes, 20, out);
    hipDeviceSynchronize(); hipMemset(out, 0, 16);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((wt<KIND, K>), blocks, threads, 0, 0, buf, bytes, per, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double waves = (double)blocks * threads / 64, n = waves * per;
    printf("%-8s K=%2d blocks=%4d waves/blk=%d : issue %.0f ns per store, drain %.0f ns per round, %.0f GB/s aggregate\n", name, K, blocks, threads / 64,
           h[0] * 10.0 / n / K, h[1] * 10.0 / n, waves * per * K * 1024.0 / (ms * 1e-3) / 1e9);
}
int main()
{
    const size_t BUF = 256u << 20;
    uint8_t *buf; uint64_t *out;
    hipMalloc(&buf, BUF); hipMalloc(&out, 64); hipMemset(buf, 0, BUF);
    for (int blocks : {1, 96, 192, 768}) for (int threads : {64, 256}) {
        run<0, 8>("plain", buf, BUF, out, blocks, threads);
        run<1, 8>("nt", buf, BUF, out, blocks, threads);
        run<2, 8>("sc1", buf, BUF, out, blocks, threads);
        run<3, 8>("sc0sc1", buf, BUF, out, blocks, threads);
        run<3, 24>("sc0sc1", buf, BUF, out, blocks, threads);
    }
    return 0;
}
