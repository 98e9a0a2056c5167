// LDS dependent-load latency microbenchmark (development aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
template <int MODE>
__global__ void chase(const uint32_t* init, uint32_t* out, long long* cyc, int iters, int lanes, int stride)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = init[i];
    __syncthreads();
    if ((int)threadIdx.x >= lanes) return;
    uint32_t x = threadIdx.x * stride;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) x = lds[x & 8191];                                   // ds_read_b32 chain
        else if (MODE == 1) x = ((const uint16_t*)lds)[x & 16383];           // ds_read_u16 chain
        else { uint32_t a = lds[x & 8191]; uint32_t b = lds[(x + 977) & 8191]; x = a ^ (b & 1); }  // two independent reads
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    uint32_t h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = (rand() & 8191) | ((rand() & 8191) << 16);
    uint32_t *d, *o; long long* c;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4096); hipMalloc(&c, 8);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode)
        for (int lanes : {1, 4, 9, 15, 17, 32, 64}) {
            long long cy = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(1), dim3(64), 32768, 0, d, o, c, iters, lanes, 131);
                if (mode == 1) hipLaunchKernelGGL(chase<1>, dim3(1), dim3(64), 32768, 0, d, o, c, iters, lanes, 131);
                if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(1), dim3(64), 32768, 0, d, o, c, iters, lanes, 131);
                hipDeviceSynchronize();
                hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            }
            printf("mode %d lanes %2d : %.1f cycles/iter (s_memtime units)\n", mode, lanes, (double)cy / iters);
        }
    return 0;
}
