// dependent-chain microbenchmark of one FSE decode link (development aid): LDS u16 lookup + the VALU ops between lookups
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
template <int MODE>
__global__ void chain(const uint32_t* init, uint32_t* out, long long* cyc, int iters, int lanes)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = init[i & 8191];
    __syncthreads();
    if ((int)threadIdx.x >= lanes) return;
    const uint32_t tab = threadIdx.x * 4096u;
    uint32_t x = tab, t = 0x9E3779B9u * (threadIdx.x + 1);
    typedef const __attribute__((address_space(3))) uint16_t* lp;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        const uint32_t c = *(lp)(uintptr_t)(base + (x & 0xFFFEu));
        if (MODE == 0) { x = (c & 0xFFEu) | tab; }                                                      // bare lookup chain
        else if (MODE == 1) { const uint32_t nb = (c >> 12) | 1u; const uint32_t bits = __builtin_amdgcn_ubfe(t, 32u - nb, nb);
                              x = (bits << 1) + ((c & 0xFFEu) | tab); t = (t << nb) | (t >> (32u - nb)); }  // FSE link
        else { const uint32_t nb = (c >> 12) | 1u; const uint32_t bits = __builtin_amdgcn_ubfe(t, 32u - nb, nb);
               x = (bits << 1) + ((c & 0xFFEu) | tab); t = t * 2654435761u + nb; x ^= (t >> 31) << 1; }   // + extra dependent ops
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + t;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// one full decode-like iteration: two chains (4 links, pairwise parallel), a 2-dword ring read, an 8-byte ring write
template <int EXTRA>
__global__ void iter4(const uint32_t* init, uint32_t* out, long long* cyc, int iters, int lanes)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = init[i & 8191];
    __syncthreads();
    if ((int)threadIdx.x >= lanes) return;
    const uint32_t tab = threadIdx.x * 4096u;
    typedef const __attribute__((address_space(3))) uint16_t* lp;
    typedef const __attribute__((address_space(3))) uint32_t* lp32;
    typedef __attribute__((address_space(3))) uint2* lpw;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    uint32_t s1 = tab, s2 = tab + 64, thi = 0x9E3779B9u * (threadIdx.x + 1), tlo = thi * 7u, q = 1000;
#define CELL(a) (*(lp)(uintptr_t)(base + ((a) & 0xFFFEu)))
#define LINK(c, st, t, nb) { nb = ((c) >> 12) | 1u; const uint32_t bits = __builtin_amdgcn_ubfe(t, 32u - nb, nb); st = (bits << 1) + (((c) & 0xFFEu) | tab); }
    uint2 pf = make_uint2(1, 2);
    uint32_t c1 = CELL(s1), c2 = CELL(s2);
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        uint32_t n0 = 0, n1 = 0;
        if (EXTRA == 1 || EXTRA == 3) { const lp32 np = (lp32)(uintptr_t)(base + 61440u + (q & 508u)); n0 = np[0]; n1 = np[1]; }
        if (EXTRA == 2) { n0 = pf.x; n1 = pf.y; __builtin_memcpy(&pf, (const char*)init + threadIdx.x * 512 + (q & 255u), 8); }
        uint32_t nb1, nb2, nb3, nb4;
        const uint32_t rx = s1 | (s2 << 16);
        LINK(c1, s1, thi, nb1) const uint32_t c3 = CELL(s1);
        LINK(c2, s2, thi << nb1, nb2) const uint32_t c4 = CELL(s2);
        const uint32_t t3 = __builtin_amdgcn_alignbit(thi, tlo, 32u - (nb1 + nb2));
        const uint32_t ry = s1 | (s2 << 16);
        LINK(c3, s1, t3, nb3) c1 = CELL(s1);
        LINK(c4, s2, t3 << nb3, nb4) c2 = CELL(s2);
        const uint32_t cons = nb1 + nb2 + nb3 + nb4;
        thi = __builtin_amdgcn_alignbit(thi, tlo, cons & 31u) ^ n0; tlo = (tlo << 7) + n1 + cons; q += cons;
        if (EXTRA == 1 || EXTRA == 2 || EXTRA == 4) { typedef __attribute__((address_space(3))) uint32_t* lpw32; lpw32 wp2 = (lpw32)(uintptr_t)(base + 63488u + ((i & 63) << 3)); wp2[0] = rx; wp2[1] = ry; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = s1 + s2 + thi;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    static uint32_t h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
    uint32_t *d, *o; long long* c;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4096); hipMalloc(&c, 8);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int lanes : {1, 15, 64}) {
            long long cy = 0; float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 65536, 0, d, o, c, iters, lanes);
                if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 65536, 0, d, o, c, iters, lanes);
                if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(1), dim3(64), 65536, 0, d, o, c, iters, lanes);
                hipEventRecord(e1); hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            }
            printf("mode %d lanes %2d : %.1f s_memtime units / link, %.1f ns / link\n", mode, lanes, (double)cy / iters, ms * 1e6 / iters);
        }
    for (int extra = 0; extra < 5; ++extra)
        for (int lanes : {15}) {
            long long cy = 0; float ms = 0;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (extra == 4) hipLaunchKernelGGL(iter4<4>, dim3(1), dim3(64), 65536, 0, d, o, c, iters, lanes);
                else if (extra == 3) hipLaunchKernelGGL(iter4<3>, dim3(1), dim3(64), 65536, 0, d, o, c, iters, lanes);
                else if (extra == 2) hipLaunchKernelGGL(iter4<2>, dim3(1), dim3(64), 65536, 0, d, o, c, iters, lanes);
                else if (extra) hipLaunchKernelGGL(iter4<1>, dim3(1), dim3(64), 65536, 0, d, o, c, iters, lanes);
                else hipLaunchKernelGGL(iter4<0>, dim3(1), dim3(64), 65536, 0, d, o, c, iters, lanes);
                hipEventRecord(e1); hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            }
            printf("iteration (4 symbols) extra-LDS %d lanes %2d : %.1f units, %.1f ns\n", extra, lanes, (double)cy / iters, ms * 1e6 / iters);
        }
    return 0;
}
