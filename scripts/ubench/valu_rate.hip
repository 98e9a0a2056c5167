// VALU throughput per instruction kind (development aid): 16 waves of one workgroup (4 per SIMD) each issue a stream of 8 independent
// instructions of one kind; cycles per wave-instruction per SIMD = (cycles of the run) / (instructions per wave * waves per SIMD).
// 4 = full rate, 8 = half, 16 = quarter.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(S) S S S S S S S S
template <int MODE>
__global__ __launch_bounds__(1024) void rate(uint32_t* out, long long* cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    uint32_t a = lane, b = lane * 3 + 1, c = lane * 5 + 2, d = lane * 7 + 3;
    uint64_t q = ((uint64_t)a << 32) | b, r = ((uint64_t)c << 32) | d;
    uint32_t sh = lane & 31;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP8(__asm__ volatile("v_lshlrev_b32 %0, %2, %1" : "=v"(a) : "v"(b), "v"(sh));) }
        else if (MODE == 1) { REP8(__asm__ volatile("v_lshlrev_b64 %0, %2, %1" : "=v"(q) : "v"(r), "v"(sh));) }
        else if (MODE == 2) { REP8(__asm__ volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(c), "v"(sh));) }
        else if (MODE == 3) { REP8(__asm__ volatile("v_bfe_u32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(sh), "v"(c));) }
        else if (MODE == 4) { REP8(__asm__ volatile("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(sh), "v"(c));) }
        else if (MODE == 5) { REP8(__asm__ volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(a) : "v"(b), "v"(c));) }
        else if (MODE == 6) { REP8(__asm__ volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c));) }
        else if (MODE == 7) { REP8(__asm__ volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(sh), "v"(c));) }
        else if (MODE == 8) { REP8(__asm__ volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(sh), "v"(c));) }
        else if (MODE == 9) { REP8(__asm__ volatile("v_lshrrev_b64 %0, %2, %1" : "=v"(q) : "v"(r), "v"(sh));) }
        else if (MODE == 10) { REP8(__asm__ volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(c), "v"(d));) }
        else if (MODE == 11) { REP8(__asm__ volatile("v_and_b32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a) : "v"(b), "v"(c));) }
        else if (MODE == 12) { REP8(__asm__ volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));) }
        else if (MODE == 13) { REP8(__asm__ volatile("v_add_lshl_u32 %0, %1, %2, 1" : "=v"(a) : "v"(b), "v"(c));) }
        else if (MODE == 14) { REP8(__asm__ volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a) : "v"(b), "v"(c) : "s20", "s21");) }
        else if (MODE == 15) { REP8(__asm__ volatile("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c) : "vcc");) }
        else if (MODE == 16) { REP8(__asm__ volatile("v_cmp_lt_u32 vcc, %1, %2" : : "v"(a), "v"(b), "v"(c) : "vcc");) }
        else if (MODE == 17) { REP8(__asm__ volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(c), "v"(d));) }
        else if (MODE == 18) { REP8(__asm__ volatile("v_max_u32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));) }
        else if (MODE == 19) { REP8(__asm__ volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c));) }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a ^ (uint32_t)q ^ (uint32_t)(q >> 32);
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    uint32_t* o; long long* c;
    hipMalloc(&o, 8192); hipMalloc(&c, 64);
    const int iters = 20000;
    const char* names[] = { "v_lshlrev_b32", "v_lshlrev_b64", "v_alignbit_b32", "v_bfe_u32", "v_lshl_or_b32", "v_add_u32_sdwa", "v_cndmask_b32", "v_and_or_b32", "v_add3_u32",
                            "v_lshrrev_b64", "v_perm_b32", "v_and_b32_dpp", "v_mov_b32", "v_add_lshl_u32", "v_cndmask_e64 sgpr", "v_cmp+v_cndmask (x2)", "v_cmp_lt_u32", "v_bfi_b32", "v_max_u32", "v_cndmask vcc" };
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    for (int mode = 0; mode < 20; ++mode) {
        long long cy = 0;
        for (int rep = 0; rep < 2; ++rep) {
#define L(M) if (mode == M) hipLaunchKernelGGL(rate<M>, dim3(1), dim3(1024), 0, 0, o, c, iters);
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19)
            hipDeviceSynchronize();
            hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        }
                const double cycles = (double)cy;
        printf("%-16s : %.2f s_memtime ticks per wave-instruction per SIMD (4 waves per SIMD; a full-rate instruction = the v_mov row)\n", names[mode], cycles / ((double)iters * 8 * 4));
    }
    return 0;
}
