// VALU issue-rate microbenchmark (development aid): cycles per VALU instruction for one wave, dependent vs independent,
// with different numbers of active lanes, and with a second wave running on another SIMD of the same CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void issue(uint32_t* out, long long* cyc, int iters, int lanes)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane >= lanes) return;
    uint32_t a = lane, b = lane * 3, c = lane * 5, d = lane * 7, e = lane + 11, f = lane + 13, g = lane + 17, h = lane + 19;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {      // 8 dependent adds
#pragma unroll
            for (int k = 0; k < 8; ++k) __asm__ volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if (MODE == 1) {               // 8 independent adds
            __asm__ volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(lane));
        } else if (MODE == 2) {               // 8 independent 3-operand ops (VOP3)
            __asm__ volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n"
                             "v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(lane));
        } else {                              // dependent pairs: 4 chains of 2
            __asm__ volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n"
                             "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(lane));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
    if (lane == 0) cyc[wave] = t1 - t0;
}
int main()
{
    uint32_t* o; long long* c;
    hipMalloc(&o, 4096); hipMalloc(&c, 64);
    const int iters = 20000;
    for (int mode = 0; mode < 4; ++mode)
        for (int waves : {1, 2, 4, 8})
            for (int lanes : {16, 32, 33, 64}) {
                long long cy[8] = {0};
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(issue<0>, dim3(1), dim3(64 * waves), 0, 0, o, c, iters, lanes);
                    if (mode == 1) hipLaunchKernelGGL(issue<1>, dim3(1), dim3(64 * waves), 0, 0, o, c, iters, lanes);
                    if (mode == 2) hipLaunchKernelGGL(issue<2>, dim3(1), dim3(64 * waves), 0, 0, o, c, iters, lanes);
                    if (mode == 3) hipLaunchKernelGGL(issue<3>, dim3(1), dim3(64 * waves), 0, 0, o, c, iters, lanes);
                    hipDeviceSynchronize();
                    hipMemcpy(cy, c, 64, hipMemcpyDeviceToHost);
                }
                printf("mode %d waves %d lanes %2d : %.2f cycles/instr (wave 0, s_memtime units)\n", mode, waves, lanes, (double)cy[0] / iters / 8);
            }
    return 0;
}
