// huf_encode.hip -- a4: HUF_compress1X_usingCTable / HUF_compress4X_usingCTable over a batch
// (reference: lib/huf_compress.c:457-608; format SURVEY A.6).
//
// A prefix code has no loop-carried state, so the encoder is fully parallel inside a block:
//   one workgroup per block, one 64-lane wavefront per stream (4 waves for the 4-stream layout).
//   pass 1: every lane sums the code lengths of a strided share of its stream  -> exact stream sizes,
//           hence the 6-byte jump table, every stream's start offset and the return value (0 when any
//           stream fails BIT_closeCStream's capacity rule, lib/bitstream.h:254-260) before a bit is written;
//   pass 2: the stream is walked in emission order (last symbol first) in rows of 256 symbols; lane L owns
//           4 consecutive symbols, packs their codes LSB-first into one <=48-bit chunk, a wave prefix sum of
//           the chunk lengths gives its bit position and the chunk is OR-ed into an LDS image of the block's
//           output (ds_or_b32); source bytes are read with coalesced loads.
//   The image is then copied to global memory with coalesced stores.  Outputs too large for the LDS image
//   (only possible for blocks > 32 KB) take the same path with global atomics on a pre-zeroed destination.
#include "internal.h"

#define HUF_ENC_THREADS 256

DEV u32 wave_incl_scan_u32(u32 v, u32 lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)v, off, WAVE); if ((int)lane >= off) v += o; }
    return v;
}
DEV u32 wave_sum_u32(u32 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (u32)__shfl_xor((int)v, off, WAVE);
    return v;
}

// OR `nb` (<= 48) bits of `bits` into the bit image at absolute bit position P
template <bool GLOBAL>
DEV void or_bits(u32* img, u64 P, u64 bits, u32 nb)
{
    if (nb == 0) return;
    const size_t w = (size_t)(P >> 5);
    const u32 sh = (u32)P & 31u;
    const u64 lo = bits << sh;                           // bits < 2^48, sh < 32 : up to 79 bits in all
    const u32 hi = sh ? (u32)(bits >> (64 - sh)) : 0u;
    const u32 a0 = (u32)lo, a1 = (u32)(lo >> 32);
    if (a0) atomicOr(&img[w], a0);
    if (a1) atomicOr(&img[w + 1], a1);
    if (hi) atomicOr(&img[w + 2], hi);
}

template <bool GLOBAL>
DEV void emit_stream(u32* img, u64 bitBase, const u8* seg, u32 len, const u32* ct, u32 lane)
{
    // emission order: j = 0 is the LAST symbol of the segment (huf_compress.c:474-499 net effect)
    u64 rowBase = bitBase;
    for (u32 j0 = 0; j0 < len; j0 += 256) {
        const u32 j = j0 + 4 * lane;
        u64 bits = 0; u32 nb = 0;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            if (j + k < len) {
                const u32 e = ct[seg[len - 1 - (j + k)]];
                bits |= (u64)(e & 0xFFFFu) << nb;
                nb += (e >> 16) & 0xFFu;
            }
        }
        const u32 incl = wave_incl_scan_u32(nb, lane);
        or_bits<GLOBAL>(img, rowBase + (incl - nb), bits, nb);
        rowBase += (u32)__shfl((int)incl, 63, WAVE);
    }
}

__global__ __launch_bounds__(HUF_ENC_THREADS) void k_huf_encode(HufEncArgs a, u32 imgBytes)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    u32* const ct = lds;                 // 256 HUF_CElt
    u32* const sh = lds + 256;           // [0..3] stream bit counts
    u32* const img = lds + 256 + 8;      // output image (imgBytes)
    const size_t b = blockIdx.x;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int streams = a.streams;

    u32 hdr = 0;
    if (a.meta) { if (a.meta[b].state == 0) return; hdr = a.meta[b].hdrSize; }
    const u8* const src = view_ptr(a.src, b);
    const size_t n64 = view_size(a.src, b);
    u8* const dst = a.dst + b * a.dstStride + hdr;
    const size_t cap = a.dstCapacity - hdr;

    // early-outs (huf_compress.c:564-565 for 4X; :470-472 for 1X)
    bool fail = false;
    if (streams == 4) { if (cap < 6 + 1 + 1 + 1 + 8) fail = true; if (n64 < 12) fail = true; }
    if (n64 >= ((size_t)1 << 31)) fail = true;           // batch path: blocks < 2 GiB
    const u32 n = (u32)n64;
    if (fail) { if (tid == 0) a.results[b] = 0; return; }

    const u32* gct = a.ctables + b * a.ctStrideU32;
    for (u32 i = tid; i < 256; i += blockDim.x) ct[i] = gct[i];
    __syncthreads();

    const u32 segSize = streams == 4 ? (n + 3) / 4 : n;
    const u32 myStart = wave * segSize;
    const u32 myLen = (int)wave < streams ? (streams == 4 && wave == 3 ? n - 3 * segSize : segSize) : 0;

    // ---- pass 1: code bits of my stream
    {   u32 bits = 0;
        const u8* seg = src + myStart;
        for (u32 i = lane; i < myLen; i += 64) bits += (ct[seg[i]] >> 16) & 0xFFu;
        bits = wave_sum_u32(bits);
        if (lane == 0 && (int)wave < streams) sh[wave] = bits;
    }
    __syncthreads();

    // ---- layout + verdict (uniform)
    size_t op = streams == 4 ? 6 : 0;
    size_t start[4] = { 0, 0, 0, 0 }, ssize[4] = { 0, 0, 0, 0 };
    for (int k = 0; k < streams; ++k) {
        const size_t capk = cap - op;                    // oend - op
        if (capk <= 8) { fail = true; break; }           // dstSize < 8 / BIT_initCStream (bitstream.h:191)
        const size_t tot = (size_t)sh[k] + 1;            // + end mark
        if ((tot >> 3) >= capk - 8) { fail = true; break; }   // BIT_closeCStream overflow rule
        start[k] = op; ssize[k] = (tot + 7) >> 3;
        op += ssize[k];
    }
    const size_t total = op;
    size_t result = fail ? 0 : total;
    if (a.meta && !fail) result = ((size_t)hdr + total < n64 - 1) ? (size_t)hdr + total : 0;   // huf_compress.c:625
    if (result == 0) { if (tid == 0) a.results[b] = 0; return; }

    // ---- pass 2: emit.  The image is addressed from the 4-byte aligned word holding dst[0].
    const u32 lead = (u32)((uintptr_t)dst & 3u);
    u8* const dstAl = dst - lead;
    const size_t imgWords = (lead + total + 3) >> 2;
    const bool inLds = (imgWords * 4 <= imgBytes);
    if (inLds) {
        for (size_t i = tid; i < imgWords; i += blockDim.x) img[i] = 0;
    } else {
        for (size_t i = tid; i < total; i += blockDim.x) dst[i] = 0;
    }
    __syncthreads();
    if ((int)wave < streams) {
        const u64 base = 8 * ((u64)lead + start[wave]);
        if (inLds) {                                     // separate call sites keep the LDS / global address spaces visible
            emit_stream<false>(img, base, src + myStart, myLen, ct, lane);
            if (lane == 0) {                             // end mark, then the jump table entry of this stream
                or_bits<false>(img, base + sh[wave], 1, 1);
                if (streams == 4 && wave < 3) or_bits<false>(img, 8 * ((u64)lead + 2 * wave), ssize[wave], 16);
            }
        } else {
            u32* const g = (u32*)dstAl;
            emit_stream<true>(g, base, src + myStart, myLen, ct, lane);
            if (lane == 0) {
                or_bits<true>(g, base + sh[wave], 1, 1);
                if (streams == 4 && wave < 3) or_bits<true>(g, 8 * ((u64)lead + 2 * wave), ssize[wave], 16);
            }
        }
    }
    __syncthreads();
    if (inLds) {                                         // coalesced copy-out: whole aligned words, edge words bytewise
        const u8* ib = (const u8*)img;
        const size_t end = lead + total;
        for (size_t w = tid; w < imgWords; w += blockDim.x) {
            const size_t lo = 4 * w, hi = 4 * w + 4;
            if (lo >= lead && hi <= end) ((u32*)dstAl)[w] = img[w];
            else for (size_t i = (lo > lead ? lo : lead); i < (hi < end ? hi : end); ++i) dstAl[i] = ib[i];
        }
    }
    if (tid == 0) a.results[b] = result;
}

hipError_t launch_huf_encode(const HufEncArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    static bool attrSet = false;
    const size_t maxLds = 64 * 1024;
    if (!attrSet) {
        hipError_t e = hipFuncSetAttribute((const void*)k_huf_encode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLds);
        if (e != hipSuccess) return e;
        attrSet = true;
    }
    size_t img = a.dstCapacity + 16;
    if (img > maxLds - 1100) img = maxLds - 1100;
    img = (img + 15) & ~(size_t)15;
    const size_t ldsBytes = (256 + 8) * 4 + img;
    const unsigned threads = a.streams == 4 ? HUF_ENC_THREADS : 64;
    probe_before(PK_HUF_ENCODE, s);
    hipLaunchKernelGGL(k_huf_encode, dim3((unsigned)a.nBlocks), dim3(threads), ldsBytes, s, a, (u32)img);
    probe_after(PK_HUF_ENCODE, s);
    return hipGetLastError();
}
