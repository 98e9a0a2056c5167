// LDS feature probes (development aid): does ds_read_u16_d16 keep the high half of its destination on this part
// (sramecc), do unaligned ds_read_b64 work and what do they cost, v_bfe_u32 with width 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
__global__ void probe(const uint8_t* init, uint32_t* out, long long* cyc, int iters)
{
    extern __shared__ uint8_t lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = init[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    // (1) d16 load into a register whose high half is preset
    uint32_t reg = 0xABCD0000u | lane;
    const uint32_t addr = 2 * lane + 64;
    __asm__ volatile("ds_read_u16_d16 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(reg) : "v"(addr) : "memory");
    out[lane] = reg;
    // (2) unaligned 64-bit reads
    uint64_t v;
    const uint32_t a2 = 3 * lane + 1;
    __asm__ volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a2) : "memory");
    out[64 + 2 * lane] = (uint32_t)v; out[65 + 2 * lane] = (uint32_t)(v >> 32);
    // (3) bfe width 0 / offset+width beyond 32
    out[192 + lane] = __builtin_amdgcn_ubfe(0xFFFFFFFFu, lane & 31, 0u) | (__builtin_amdgcn_ubfe(0xFFFFFFFFu, 28u, 32u + 8u) << 8);
    // (4) latency of dependent unaligned b64 reads vs aligned
    for (int mode = 0; mode < 2; ++mode) {
        uint32_t x = (lane * 24 + (mode ? 3 : 0)) & 4095;
        long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            uint64_t w;
            __asm__ volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(x) : "memory");
            x = (((uint32_t)w & 0xFF8u) | (mode ? 3u : 0u)) & 4095u;
        }
        long long t1 = __builtin_readcyclecounter();
        if (lane == 0) cyc[mode] = t1 - t0;
        out[256 + lane] ^= x;
    }
}
int main()
{
    uint8_t h[8192];
    for (int i = 0; i < 8192; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t* d; uint32_t* o; long long* c;
    hipMalloc(&d, 8192); hipMalloc(&o, 4096); hipMalloc(&c, 64);
    hipMemcpy(d, h, 8192, hipMemcpyHostToDevice);
    hipMemset(o, 0, 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, d, o, c, 20000);
    hipDeviceSynchronize();
    uint32_t ho[1024]; long long hc[8];
    hipMemcpy(ho, o, 4096, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 64, hipMemcpyDeviceToHost);
    int okd16 = 1, okun = 1;
    for (int l = 0; l < 64; ++l) {
        uint16_t e; memcpy(&e, h + 2 * l + 64, 2);
        if (ho[l] != (0xABCD0000u | e)) okd16 = 0;
        uint64_t e64; memcpy(&e64, h + 3 * l + 1, 8);
        if (ho[64 + 2 * l] != (uint32_t)e64 || ho[65 + 2 * l] != (uint32_t)(e64 >> 32)) okun = 0;
    }
    printf("d16 keeps high half: %s (lane 5 -> %08x)\n", okd16 ? "yes" : "NO", ho[5]);
    printf("unaligned ds_read_b64: %s\n", okun ? "ok" : "WRONG");
    printf("bfe width0 -> %x, bfe(.,28,40) -> %x\n", ho[192 + 7] & 0xFF, ho[192 + 7] >> 8);
    printf("dependent ds_read_b64: aligned %.1f, unaligned %.1f ticks\n", hc[0] / 20000.0, hc[1] / 20000.0);
    return 0;
}
