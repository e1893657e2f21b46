// micro-benchmark: cost of taken branches to far (cold) code lines, first pass vs second pass in the same kernel,
// and across kernel launches.  16 jumps, each over 4 KB of never-executed padding.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define PAD ".fill 1024, 4, 0xBF800000\n"
#define HOP(n) "s_branch L" #n "_%=\n" PAD "L" #n "_%=:\n"
__global__ void jumps(uint64_t* out) {
    uint64_t t0, t1, t2;
    for (int pass = 0; pass < 2; pass++) {
        asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)\n"
                     HOP(1) HOP(2) HOP(3) HOP(4) HOP(5) HOP(6) HOP(7) HOP(8) HOP(9) HOP(10) HOP(11) HOP(12) HOP(13) HOP(14) HOP(15) HOP(16)
                     "s_memrealtime %1\n s_waitcnt lgkmcnt(0)\n" : "=s"(t0), "=s"(t1) :: "memory");
        if (threadIdx.x == 0) out[blockIdx.x * 2 + pass] = t1 - t0;
    }
}
__global__ void straight(uint64_t* out) {       // same instruction count, no padding: baseline
    uint64_t t0, t1;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)\n"
                 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
                 "s_memrealtime %1\n s_waitcnt lgkmcnt(0)\n" : "=s"(t0), "=s"(t1) :: "memory");
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
}
int main() {
    uint64_t* o; hipMalloc(&o, 1 << 16); uint64_t r[2048];
    for (int blocks : {1, 256}) for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(jumps, blocks, 64, 0, 0, o); hipDeviceSynchronize();
        hipMemcpy(r, o, blocks * 16, hipMemcpyDeviceToHost);
        double a = 0, b = 0, am = 0; for (int i = 0; i < blocks; i++) { a += r[2 * i]; b += r[2 * i + 1]; if (r[2 * i] > am) am = r[2 * i]; }
        printf("%3d blocks, launch %d: 16 far jumps  first pass avg %.2f us (max %.2f), second pass avg %.2f us\n", blocks, rep, a / blocks / 100, am / 100, b / blocks / 100);
    }
    hipLaunchKernelGGL(straight, 1, 64, 0, 0, o); hipDeviceSynchronize(); hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    printf("straight-line 16 s_nop: %.2f us\n", r[0] / 100.0);
    return 0;
}
