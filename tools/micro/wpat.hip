// micro-benchmark: which part of the append blocks' store pattern costs the bandwidth?
// One wavefront per "round" of 64 entries x 128 B (8 KiB): lane l stores 16-byte units l, l+64, ...
// into NBUF rings at the same offset.  Optional extras, switched on one at a time:
//   +shift   rings start 64 B off a 128-B boundary (the real log after an odd number of 64-B entries)
//   +dir     per entry and ring: 8-B offset + 4-B length (directory), + one 4-B ACK word
//   +apply   per entry and ring: one 32-B apply record
//   +read    payload: 4 of every 8 units are loaded from an arena first
// hipcc --offload-arch=gfx950 -O2 tools/micro/wpat.hip -o tools/micro/wpat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
struct Bufs { uint4 *ring[3]; uint64_t *doff[3]; uint32_t *dlen[3]; uint32_t *ack; uint4 *apply[3]; const uint4 *arena; };
__global__ __launch_bounds__(256) void rounds(Bufs B, size_t units_per_ring, size_t slots, int shift, int dir, int apply, int rd, int nbuf)
{
    extern __shared__ uint32_t occupancy_limiter[];           // dynamic LDS only limits the workgroups per CU
    if (slots == 1) occupancy_limiter[threadIdx.x] = 0;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t r = (size_t)blockIdx.x * 4 + wv;            // round
    const size_t base = (r * 512 + (shift ? 4 : 0)) % units_per_ring;
    uint4 pv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t u = lane + 64 * k;
        pv[k] = make_uint4(u, (uint32_t)r, k, 7);
        if (rd && (u & 7) >= 4) pv[k] = B.arena[(r * 256 + (u >> 3) * 4 + (u & 3)) % (units_per_ring / 2)];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const size_t u = (base + lane + 64 * k) % units_per_ring;
        for (int b = 0; b < nbuf; b++) B.ring[b][u] = pv[k];
    }
    const size_t slot = (r * 64 + lane + 17) % slots;
    if (dir) {
        for (int b = 0; b < nbuf; b++) { B.doff[b][slot] = base * 16 + lane * 128; B.dlen[b][slot] = 128; }
        B.ack[slot] = 6;
    }
    if (apply) {
        for (int b = 0; b < nbuf; b++) { B.apply[b][2 * slot] = make_uint4(slot, 0, base, 0); B.apply[b][2 * slot + 1] = make_uint4(lane, 0, 64, 5); }
    }
}
int main()
{
    const size_t RING = 64u << 20, U = RING / 16, SLOTS = 1u << 20;
    Bufs B;
    for (int b = 0; b < 3; b++) {
        hipMalloc(&B.ring[b], RING + 4096); hipMemset(B.ring[b], 0, RING);
        hipMalloc(&B.doff[b], SLOTS * 8); hipMalloc(&B.dlen[b], SLOTS * 4); hipMalloc(&B.apply[b], SLOTS * 32);
    }
    hipMalloc(&B.ack, SLOTS * 4);
    uint4 *ar; hipMalloc(&ar, RING / 2); hipMemset(ar, 1, RING / 2); B.arena = ar;
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { const char *name; int shift, dir, apply, rd; } cfgs[] = {
        {"rings only", 0, 0, 0, 0}, {"+shift", 1, 0, 0, 0}, {"+dir", 0, 1, 0, 0}, {"+apply", 0, 0, 1, 0}, {"+read", 0, 0, 0, 1},
        {"+dir+apply", 0, 1, 1, 0}, {"all", 1, 1, 1, 1}, {"all but shift", 0, 1, 1, 1}};
    hipFuncSetAttribute((const void *)rounds, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10);
    for (int per_cu : {8, 5, 3, 2, 1}) for (int rounds_per_launch : {5464, 1024}) for (auto &c : cfgs) {
        if (per_cu != 8 && (c.shift + c.dir + c.apply + c.rd) != 4 && (c.shift + c.dir + c.apply + c.rd) != 0) continue;
        const int blocks = rounds_per_launch / 4;
        const size_t lds = per_cu >= 8 ? 0 : (size_t)(160 << 10) / per_cu - 1024;     // workgroups per CU by LDS
        if (c.shift + c.dir + c.apply + c.rd == 0 || per_cu == 8) printf("-- %d workgroup(s) per CU\n", per_cu);
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(rounds, blocks, 256, lds, st, B, U, SLOTS, c.shift, c.dir, c.apply, c.rd, 3);
        hipEventRecord(e0, st);
        const int reps = 20;
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(rounds, blocks, 256, lds, st, B, U, SLOTS, c.shift, c.dir, c.apply, c.rd, 3);
        hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000 / reps;
        const double wb = (double)rounds_per_launch * (8192.0 * 3 + (c.dir ? 64 * (12.0 * 3 + 4) : 0) + (c.apply ? 64 * 32.0 * 3 : 0));
        const double rb = c.rd ? (double)rounds_per_launch * 4096 : 0;
        printf("%5d rounds/launch %-14s %7.2f us/launch  %5.2f TB/s written  %5.2f TB/s moved  (%.1f MB)\n",
               rounds_per_launch, c.name, us, wb / us / 1e6, (wb + rb) / us / 1e6, (wb + rb) / 1e6);
    }
    return 0;
}
