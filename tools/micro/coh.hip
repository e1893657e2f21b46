// micro-benchmark: are plain loads/stores to fine-grained / uncached device memory coherent across
// workgroups (other CUs, other XCDs) INSIDE one kernel?  Block 0 waits, stores 64 words plainly, waits
// for the stores (s_waitcnt), then raises a flag with an agent-scope atomic.  Every other block first
// reads the words (pulls stale zeros into its caches), polls the flag, then reads the words again
// with plain loads and counts the ones that are still stale.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void coh(uint64_t* words, uint32_t* flag, uint32_t* stale, uint64_t* lat, int acq) {
    __shared__ uint64_t first[64];
    if (blockIdx.x == 0) {
        for (int i = 0; i < 400; i++) __builtin_amdgcn_s_sleep(32);          // ~6 us
        if (threadIdx.x < 64) words[threadIdx.x] = threadIdx.x + 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x < 64) first[threadIdx.x] = words[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1ull << 22)) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    if (acq == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");               // reader side: drop L1 / non-coherent L2 lines
    if (threadIdx.x < 64) {
        const uint64_t t0 = wall_clock64();
        const uint64_t v = (acq == 2) ? __hip_atomic_load((unsigned long long*)&words[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : words[threadIdx.x];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t1 = wall_clock64();
        if (v != threadIdx.x + 1) atomicAdd(stale, 1u);
        if (threadIdx.x == 0) lat[blockIdx.x] = t1 - t0;
        if (first[threadIdx.x] != 0) atomicAdd(stale + 1, 1u);                 // sanity: should stay 0
    }
}
int main() {
    const char* names[3] = {"hipMalloc (coarse-grained)", "hipExtMallocWithFlags(Finegrained)", "hipExtMallocWithFlags(Uncached)"};
    unsigned flags[3] = {0, hipDeviceMallocFinegrained, hipDeviceMallocUncached};
    for (int acq = 0; acq < 3; acq++) for (int m = 0; m < 3; m++) for (int rep = 0; rep < 2; rep++) {
        uint64_t* w; uint32_t *flag, *stale; uint64_t* lat;
        if (flags[m]) { if (hipExtMallocWithFlags((void**)&w, 4096, flags[m]) != hipSuccess) { printf("%s: alloc failed\n", names[m]); break; } }
        else hipMalloc(&w, 4096);
        hipMalloc(&flag, 64); hipMalloc(&stale, 64); hipMalloc(&lat, 8 * 1024);
        hipMemset(w, 0, 4096); hipMemset(flag, 0, 64); hipMemset(stale, 0, 64);
        hipLaunchKernelGGL(coh, 1024, 256, 0, 0, w, flag, stale, lat, acq); hipDeviceSynchronize();
        uint32_t s[2]; uint64_t l[1024]; hipMemcpy(s, stale, 8, hipMemcpyDeviceToHost); hipMemcpy(l, lat, 8 * 1024, hipMemcpyDeviceToHost);
        double a = 0; for (int i = 1; i < 1024; i++) a += l[i];
        printf("%s %-36s run %d: stale words after the flag: %u of %d (pre-read nonzero: %u), plain re-read latency avg %.2f us\n",
               acq == 2 ? "[atomic re-read] " : acq ? "[reader acquires]" : "[plain re-read]  ", names[m], rep, s[0], 1023 * 64, s[1], a / 1023 / 100.0);
        hipFree(w); hipFree(flag); hipFree(stale); hipFree(lat);
    }
    return 0;
}
