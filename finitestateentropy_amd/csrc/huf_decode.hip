// huf_decode.hip -- a5: HUF_decompress4X1_usingDTable over a batch
// (reference: lib/huf_decompress.c:194-354, dispatcher :980-997; lib/bitstream.h:272-448).
//
// The 4-stream layout gives four independent serial chains per block, so the mapping is "one lane per
// stream": a 128-lane workgroup stages G X1 tables (reference layout: 4-byte DTableDesc + 2-byte
// {byte, nbBits} cells) into LDS with coalesced copies, then lane 4g+k decodes stream k of block g.
//
// Per stream the kernel keeps the reference's own reader state (64-bit window at byte offset `at`, consumed
// bits `used`), so the verdict -- every stream must end with BIT_endOfDStream, lib/huf_decompress.c:348-349 --
// is the reference's by construction, also on corrupt input:
//   * bulk loop = the 4-symbols-per-reload iterations (HUF_decodeStreamX1 :219-224 / the lock-step loop
//     :310-331; both reload variants coincide while the window is >= 8 bytes above the stream start),
//     with the next window funnel-shifted out of two prefetched 8-byte words instead of being loaded;
//   * the last symbols of a stream run through the literal BitReader (bitreader.h).
// A stream's decoded symbols and final reader state depend only on that stream, so decoding the four
// streams independently (instead of in lock-step) yields the same result as the reference.
#include "internal.h"
#include "bitreader.h"

#define HUF_DEC_THREADS 128

DEV u32 hufx1_step(BitReader& r, const u16* cells, u32 dtLog)                // HUF_decodeSymbolX1, :194-201
{
    const u32 v = (u32)((r.win << (r.used & 63u)) >> ((64u - dtLog) & 63u)); // BIT_lookBitsFast
    const u32 c = cells[v];
    r.used += c >> 8;
    return c & 0xFFu;
}

#define HUF_BULK_STEP(SEL)                                                                 \
    {   const u32 c = cells[(u32)(t >> 32) >> shIdx];                                      \
        const u32 nb = c >> 8;                                                             \
        t <<= nb; used += nb;                                                              \
        word = __builtin_amdgcn_perm(c, word, SEL);                                        \
    }

__global__ __launch_bounds__(HUF_DEC_THREADS) void k_huf_decode(HufDecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 tid = threadIdx.x;
    const size_t first = (size_t)blockIdx.x * a.G;

    for (int g = 0; g < a.G; ++g) {                  // stage X1 tables: uniform control flow, coalesced
        const size_t b = first + g;
        if (b >= a.nBlocks) break;
        if (a.meta && a.meta[b].state == 0) continue;
        const u32* t = a.dtables + b * a.dtStrideU32;
        const u32 desc = t[0];
        const u32 tl = (desc >> 16) & 0xFFu;
        if (tl > a.maxTableLog || ((desc >> 8) & 0xFFu) != 0) continue;
        const u32 words = 1 + (tl ? (1u << (tl - 1)) : 1u);
        u32* s = lds + (size_t)g * a.slotU32;
        for (u32 i = tid; i < words; i += blockDim.x) s[i] = t[i];
    }
    __syncthreads();
    const u32 g = tid >> 2, k = tid & 3u;            // block slot, stream
    const size_t b = first + g;
    const bool live = (int)g < a.G && b < a.nBlocks && !(a.meta && a.meta[b].state == 0);

    size_t ierr = 0;                                 // BIT_initDStream error of my stream (0 = none)
    int endBad = 0;                                  // my stream did not end exactly
    size_t blockErr = 0;                             // errors detected before any stream is touched
    size_t dstSize = 0;
    if (live) {
        u32 hdr = a.meta ? a.meta[b].hdrSize : 0;
        const u32 desc = a.dtables[b * a.dtStrideU32];
        const u32 dtLog = (desc >> 16) & 0xFFu;
        const u8* const in = view_ptr(a.csrc, b) + hdr;
        const size_t cSize = view_size(a.csrc, b) - hdr;
        dstSize = view_size(a.dstSizes, b);
        u8* const out = a.dst + b * a.dstStride;
        if (((desc >> 8) & 0xFFu) != 0) blockErr = FERR(GENERIC);                       // X2 table: huf_decompress.c:411-412
        else if (dtLog > a.maxTableLog) blockErr = FERR(tableLog_tooLarge);
        else if (cSize < 10) blockErr = FERR(corruption_detected);                      // :269
        if (!blockErr) {
            const size_t l1 = ld16(in), l2 = ld16(in + 2), l3 = ld16(in + 4);
            const size_t l4 = cSize - (l1 + l2 + l3 + 6);
            if (l4 > cSize) blockErr = FERR(corruption_detected);                       // :303
            else {
                const size_t seg = (dstSize + 3) / 4;
                const size_t sStart = 6 + (k > 0 ? l1 : 0) + (k > 1 ? l2 : 0) + (k > 2 ? l3 : 0);
                const size_t sLen = k == 0 ? l1 : k == 1 ? l2 : k == 2 ? l3 : l4;
                const u8* const sp = in + sStart;
                const size_t oStart = (size_t)k * seg;
                const size_t oEndRaw = k < 3 ? oStart + seg : dstSize;                  // pEnd of this stream (:292-299)
                long cnt = oEndRaw > oStart ? (long)(oEndRaw - oStart) : 0;             // symbols to regenerate
                const u16* const cells = (const u16*)(lds + (size_t)g * a.slotU32 + 1);
                BitReader r;
                const size_t e = r.init(sp, sLen);                                      // :304-307
                if (is_err(e)) ierr = e;
                else {
                    long p = 0;
                    // bulk: reloads are the fast ones while at >= 24; each iteration = reload + 4 symbols
                    if (r.at >= 24 && cnt >= 4 && dtLog >= 1) {
                        u64 at = r.at; u32 used = r.used; u64 win = r.win;
                        u64 lo1 = ldg64u(sp + at - 8), lo2 = ldg64u(sp + at - 16);
                        const u32 shIdx = 32u - dtLog;
                        long groups = cnt >> 2;
                        do {
                            const u32 k8 = used & ~7u;
                            at -= used >> 3; used &= 7;
                            win = (win << k8) | ((lo1 >> 1) >> (63 - k8));
                            lo1 = (lo1 << k8) | ((lo2 >> 1) >> (63 - k8));
                            lo2 = ldg64u(sp + at - 16);
                            u64 t = win << used;
                            u32 word = 0;
                            HUF_BULK_STEP(0x03020104u)                                  // byte k <- c.byte0 (perm index 4)
                            HUF_BULK_STEP(0x03020400u)
                            HUF_BULK_STEP(0x03040100u)
                            HUF_BULK_STEP(0x04020100u)
                            const size_t o = oStart + (size_t)p;
                            if (o + 4 <= dstSize) __builtin_memcpy(out + o, &word, 4);
                            else for (u32 q = 0; q < 4; ++q) if (o + q < dstSize) out[o + q] = (u8)(word >> (8 * q));
                            p += 4; --groups;
                        } while (at >= 24 && groups > 0);
                        r.at = (size_t)at; r.used = used; r.win = win;
                    }
                    // literal: HUF_decodeStreamX1 (:214-237)
                    while ((r.reload() == BR_UNFINISHED) & (p < cnt - 3)) {
                        for (u32 q = 0; q < 4; ++q) { const u32 sym = hufx1_step(r, cells, dtLog); if (oStart + p < dstSize) out[oStart + p] = (u8)sym; ++p; }
                    }
                    while (p < cnt) { const u32 sym = hufx1_step(r, cells, dtLog); if (oStart + p < dstSize) out[oStart + p] = (u8)sym; ++p; }
                    if (!(r.at == 0 && r.used == 64)) endBad = 1;                       // BIT_endOfDStream, :348-349
                }
            }
        }
    }
    // combine the four streams of a block: all four inits come first and the first failing one is returned
    // (:304-307); otherwise every stream must have ended exactly (:348-349).  All lanes take part in the shuffles.
    const u32 lane = tid & 63u, base4 = lane & ~3u;
    size_t res = 0;
    int anyEnd = 0;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        const unsigned long long iq = __shfl((unsigned long long)ierr, (int)(base4 + q), WAVE);
        const int eq = __shfl(endBad, (int)(base4 + q), WAVE);
        if (res == 0 && iq != 0) res = (size_t)iq;
        anyEnd |= eq;
    }
    if (res == 0 && anyEnd) res = FERR(corruption_detected);
    if (live && k == 0) {
        size_t result;
        if (blockErr) result = blockErr;
        else if (res) result = res;
        else result = dstSize;
        a.results[b] = result;
    }
}

hipError_t launch_huf_decode(HufDecArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    static bool attrSet = false;
    const size_t ldsBytes = 80 * 1024;
    if (!attrSet) {
        hipError_t e = hipFuncSetAttribute((const void*)k_huf_decode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        attrSet = true;
    }
    a.slotU32 = (1 + (1u << (a.maxTableLog - 1))) | 1u;
    a.G = (int)(ldsBytes / (a.slotU32 * 4));
    if (a.G > HUF_DEC_THREADS / 4) a.G = HUF_DEC_THREADS / 4;
    const size_t groups = (a.nBlocks + a.G - 1) / a.G;
    probe_before(PK_HUF_DECODE, s);
    hipLaunchKernelGGL(k_huf_decode, dim3((unsigned)groups), dim3(HUF_DEC_THREADS), ldsBytes, s, a);
    probe_after(PK_HUF_DECODE, s);
    return hipGetLastError();
}
