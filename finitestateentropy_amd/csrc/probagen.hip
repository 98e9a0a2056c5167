// probagen.hip -- the reference benchmark's synthetic workload on the device
// (reference: programs/probaGenerator.c:70-74 LCG, :95-126 table lookup).
// Block b = table[(seed_i >> 11) & 4095] for seed_{i+1} = seed_i * 2654435761 + 2246822519 (mod 2^32),
// seed_0 = firstSeed + b * seedStep.  The LCG is affine, so thread t jumps straight to its bytes:
// x -> A_k x + C_k with (A_k, C_k) obtained by square-and-multiply.
#include "internal.h"

#define PG_A 2654435761u
#define PG_C 2246822519u

DEV void affine_pow(u32 k, u32& A, u32& C)
{
    u32 ba = PG_A, bc = PG_C;
    A = 1; C = 0;
    while (k) {
        if (k & 1) { C = ba * C + bc; A = ba * A; }
        bc = ba * bc + bc; ba = ba * ba;
        k >>= 1;
    }
}

__global__ __launch_bounds__(256) void k_probagen(u8* dst, size_t dstStride, size_t blockSize, size_t nBlocks, const u8* table, u32 firstSeed, u32 seedStep)
{
    __shared__ u8 tab[4096];
    for (u32 i = threadIdx.x; i < 1024; i += 256) ((u32*)tab)[i] = ((const u32*)table)[i];
    __syncthreads();
    const size_t b = blockIdx.x;
    if (b >= nBlocks) return;
    u8* const out = dst + b * dstStride;
    const u32 t = threadIdx.x;
    u32 A, C, Aj, Cj;
    affine_pow(4 * t, A, C);
    affine_pow(1020, Aj, Cj);
    u32 x = A * (firstSeed + (u32)b * seedStep) + C;
    const bool al4 = ((uintptr_t)out & 3u) == 0;
    for (size_t i = 4 * (size_t)t; i < blockSize; i += 1024) {
        u32 w = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { x = x * PG_A + PG_C; w |= (u32)tab[(x >> 11) & 4095u] << (8 * k); }
        if (al4 && i + 4 <= blockSize) *(u32*)(out + i) = w;
        else for (int k = 0; k < 4 && i + k < blockSize; ++k) out[i + k] = (u8)(w >> (8 * k));
        x = Aj * x + Cj;
    }
}

hipError_t launch_probagen(u8* dst, size_t dstStride, size_t blockSize, size_t nBlocks, const u8* d_table, u32 firstSeed, u32 seedStep, hipStream_t s)
{
    if (nBlocks == 0 || blockSize == 0) return hipSuccess;
    hipLaunchKernelGGL(k_probagen, dim3((unsigned)nBlocks), dim3(256), 0, s, dst, dstStride, blockSize, nBlocks, d_table, firstSeed, seedStep);
    return hipGetLastError();
}
