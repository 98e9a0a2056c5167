// compact.hip -- the packed (variable-length) form of a batch of compressed blocks.
//
// The batched codecs write fixed-stride slots (programs/bench.c:514-516 sizes them FSE_compressBound(blockSize) apart, of which a
// Proba14 block uses half and a Proba80 block a ninth).  What the reference's container stores (programs/fileio.c:343-400), and what is
// worth shipping between GPUs or to the host ("variable-length compressed shards"), is every block at its real size:
//   record of block b = its compressed bytes            when results[b] > 1,
//                     = the block itself (srcSize bytes) when results[b] == 0   (programs/bench.c:393-396: "not compressed block; just memcpy() it"),
//                     = its first byte                   when results[b] == 1   (:397-400: "single value byte; just memset() it"),
//                     = nothing                          when results[b] is an error code,
// records back to back, offsets[b] = where record b starts, offsets[nBlocks] = the packed size.  A record as long as the block is the
// block, a record of one byte the repeated byte: no compressor returns sizes >= srcSize - 1 (lib/fse_compress.c:674, lib/huf_compress.c:625),
// so the decoders of a packed batch (FSEHIP_*_decompress_packed_batch) tell the three apart by size, like HUF_decompress itself does
// (lib/huf_decompress.c:1063-1066).
//   k_compact_sums    : record lengths of 1024 blocks per workgroup -> their sum
//   k_compact_offsets : exclusive scan of the sums (one workgroup), then of the lengths inside every group -> offsets[]
//   k_compact_copy    : one workgroup per block, 16-byte pieces on aligned destination addresses
//   k_rawrle_expand   : the inverse for the two record kinds that are not compressed blocks (the compressed ones are decoded where they lie:
//                       BlockView::offsets)
// HBM-bound copies: bytes moved = 2 x the packed size.
#include "internal.h"

#define CP_THREADS 256
#define CP_PER_WG 1024

// (a result beyond the slot cannot have come from the call that filled these slots: no record, rather than a read behind the slot)
DEV u64 cp_len(size_t r, size_t n, size_t slotStride) { return is_err(r) ? 0 : r == 0 ? (u64)n : r == 1 ? (n ? 1u : 0u) : r > slotStride ? 0 : (u64)r; }

// inclusive scan over the 256 threads of a workgroup (u64), through LDS
DEV u64 cp_wg_scan(u64 v, u64* sh, u32 tid)
{
    const u32 lane = tid & 63u, w = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u64 o = __shfl_up(v, off, WAVE); if ((int)lane >= off) v += o; }
    if (lane == 63) sh[w] = v;
    __syncthreads();
    u64 base = 0;
    for (u32 k = 0; k < w; ++k) base += sh[k];
    __syncthreads();
    return v + base;
}

__global__ __launch_bounds__(CP_THREADS) void k_compact_sums(const size_t* results, BlockView src, size_t nBlocks, u64* partials, size_t slotStride)
{
    __shared__ u64 sh[4];
    const u32 tid = threadIdx.x;
    const size_t b0 = (size_t)blockIdx.x * CP_PER_WG;
    u64 sum = 0;
    for (u32 k = 0; k < CP_PER_WG / CP_THREADS; ++k) {
        const size_t b = b0 + tid + (size_t)k * CP_THREADS;
        if (b < nBlocks) sum += cp_len(results[b], view_size(src, b), slotStride);
    }
    const u64 incl = cp_wg_scan(sum, sh, tid);
    if (tid == CP_THREADS - 1) partials[blockIdx.x] = incl;
}

// one workgroup: exclusive scan of the group sums in place (nGroups <= a few thousand), total behind them
__global__ __launch_bounds__(CP_THREADS) void k_compact_scan_groups(u64* partials, u32 nGroups)
{
    __shared__ u64 sh[4];
    __shared__ u64 carry;
    const u32 tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (u32 g0 = 0; g0 < nGroups; g0 += CP_THREADS) {
        const u32 g = g0 + tid;
        const u64 v = g < nGroups ? partials[g] : 0;
        const u64 incl = cp_wg_scan(v, sh, tid);
        const u64 c = carry;
        if (g < nGroups) partials[g] = c + incl - v;
        __syncthreads();
        if (tid == CP_THREADS - 1) carry = c + incl;
        __syncthreads();
    }
    if (tid == 0) partials[nGroups] = carry;
}

__global__ __launch_bounds__(CP_THREADS) void k_compact_offsets(const size_t* results, BlockView src, size_t nBlocks, const u64* partials, u64* offsets, u32 nGroups, size_t slotStride)
{
    __shared__ u64 sh[4];
    __shared__ u64 carry;
    const u32 tid = threadIdx.x;
    const size_t b0 = (size_t)blockIdx.x * CP_PER_WG;
    if (tid == 0) carry = partials[blockIdx.x];
    __syncthreads();
    for (u32 k = 0; k < CP_PER_WG / CP_THREADS; ++k) {
        const size_t b = b0 + tid + (size_t)k * CP_THREADS;
        const u64 v = b < nBlocks ? cp_len(results[b], view_size(src, b), slotStride) : 0;
        const u64 incl = cp_wg_scan(v, sh, tid);
        const u64 c = carry;
        if (b < nBlocks) offsets[b] = c + incl - v;
        __syncthreads();
        if (tid == CP_THREADS - 1) carry = c + incl;
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid == 0) offsets[nBlocks] = partials[nGroups];
}

// copy `len` bytes from s to d (any alignments): bytes up to the next 16-byte boundary of d, 16-byte pieces (unaligned loads, aligned
// stores), the rest bytewise
DEV void cp_copy(u8* d, const u8* s, size_t len, u32 tid)
{
    size_t head = (size_t)((0 - (uintptr_t)d) & 15u);
    if (head > len) head = len;
    if (tid < head) d[tid] = s[tid];
    const size_t body = (len - head) & ~(size_t)15;
    for (size_t off = head + 16 * (size_t)tid; off < head + body; off += 16 * CP_THREADS) {
        uint4 v; __builtin_memcpy(&v, s + off, 16);
        *(uint4*)(d + off) = v;
    }
    const size_t done = head + body;
    if (done + tid < len) d[done + tid] = s[done + tid];
}

__global__ __launch_bounds__(CP_THREADS) void k_compact_copy(u8* packed, u64 packedCapacity, const u64* offsets, const u8* slots, size_t slotStride,
                                                             const size_t* results, BlockView src, size_t nBlocks)
{
    const size_t b = blockIdx.x;
    const size_t r = results[b];
    const u64 o = offsets[b], len = offsets[b + 1] - o;
    if (len == 0 || o + len > packedCapacity) return;                     // (a packed buffer too small: the caller sees offsets[nBlocks] > capacity)
    const u8* const from = r > 1 ? slots + b * slotStride : view_ptr(src, b);     // compressed bytes / the raw block / its first byte
    cp_copy(packed + o, from, (size_t)len, threadIdx.x);
}

hipError_t launch_compact(u8* packed, size_t packedCapacity, u64* offsets, const u8* slots, size_t slotStride, const size_t* results, const BlockView& src,
                          size_t nBlocks, u64* partials, hipStream_t s)
{
    const u32 nGroups = (u32)((nBlocks + CP_PER_WG - 1) / CP_PER_WG);
    if (nBlocks == 0) { hipLaunchKernelGGL(k_compact_scan_groups, dim3(1), dim3(CP_THREADS), 0, s, offsets, 0u); return hipGetLastError(); }   // offsets[0] = 0
    hipLaunchKernelGGL(k_compact_sums, dim3(nGroups), dim3(CP_THREADS), 0, s, results, src, nBlocks, partials, slotStride);
    hipLaunchKernelGGL(k_compact_scan_groups, dim3(1), dim3(CP_THREADS), 0, s, partials, nGroups);
    hipLaunchKernelGGL(k_compact_offsets, dim3(nGroups), dim3(CP_THREADS), 0, s, results, src, nBlocks, (const u64*)partials, offsets, nGroups, slotStride);
    hipLaunchKernelGGL(k_compact_copy, dim3((unsigned)nBlocks), dim3(CP_THREADS), 0, s, packed, (u64)packedCapacity, (const u64*)offsets, slots, slotStride, results, src, nBlocks);
    return hipGetLastError();
}

// the records of a packed batch that are not compressed blocks, regenerated: as long as the block -> the block; one byte -> that byte
// repeated (programs/bench.c:393-400).  One workgroup per block; everything else returns at once and is decoded by the codec's kernels.
__global__ __launch_bounds__(CP_THREADS) void k_rawrle_expand(u8* dst, size_t dstStride, size_t dstCapacity, size_t* results, BlockView csrc,
                                                              const size_t* origSizes, size_t uniformOrig, size_t nBlocks)
{
    const size_t b = blockIdx.x;
    const size_t len = view_size(csrc, b), orig = origSizes ? origSizes[b] : uniformOrig;
    if (len != orig && len != 1) return;                                  // uniform
    const u32 tid = threadIdx.x;
    if (orig > dstCapacity) { if (tid == 0) results[b] = FERR(dstSize_tooSmall); return; }
    u8* const d = dst + b * dstStride;
    const u8* const in = view_ptr(csrc, b);
    if (len == orig) cp_copy(d, in, orig, tid);
    else {
        const u32 fill = (u32)in[0] * 0x01010101u;
        size_t head = (size_t)((0 - (uintptr_t)d) & 15u);
        if (head > orig) head = orig;
        if (tid < head) d[tid] = (u8)fill;
        const size_t body = (orig - head) & ~(size_t)15;
        for (size_t off = head + 16 * (size_t)tid; off < head + body; off += 16 * CP_THREADS) *(uint4*)(d + off) = make_uint4(fill, fill, fill, fill);
        const size_t done = head + body;
        if (done + tid < orig) d[done + tid] = (u8)fill;
    }
    if (tid == 0) results[b] = orig;
}
hipError_t launch_rawrle_expand(u8* dst, size_t dstStride, size_t dstCapacity, size_t* results, const BlockView& csrc, const size_t* origSizes, size_t uniformOrig,
                                size_t nBlocks, hipStream_t s)
{
    if (nBlocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rawrle_expand, dim3((unsigned)nBlocks), dim3(CP_THREADS), 0, s, dst, dstStride, dstCapacity, results, csrc, origSizes, uniformOrig, nBlocks);
    return hipGetLastError();
}
