// fse_decode.hip -- a3: FSE_decompress_usingDTable over a batch
// (reference: lib/fse_decompress.c:178-252, lib/fse.h:577-622, lib/bitstream.h:272-448).
//
// tANS decoding is one loop-carried chain per block (two interleaved states sharing one bit cursor), so
// the mapping is "one lane per block, many blocks per CU": a 64-lane workgroup stages G DTables into LDS
// (verbatim reference layout, coalesced copy) and lane g walks block g's stream.  Throughput comes from
// the number of blocks resident per CU (bounded by LDS: 160 KiB / 8 KiB tables at tableLog 11).
//
// Exactness: the reference's result -- also on truncated / corrupt input -- is defined through its
// 64-bit window mechanics (BIT_reloadDStream status codes).  The kernel therefore has two parts:
//   * `BitReader`: a literal device restatement of that window (used for init and for the last few
//     symbols of every stream), and
//   * a bulk loop that fast-forwards whole 4-symbol iterations of lib/fse_decompress.c:201-218 while it is
//     provable that the reference's loop condition holds (>= 128 unread bits, >= 4 output bytes left);
//     inside that region the bits read are a pure function of the absolute bit position.
#include "internal.h"

enum { BR_UNFINISHED = 0, BR_END_OF_BUFFER = 1, BR_COMPLETED = 2, BR_OVERFLOW = 3 };   // bitstream.h:99-102

struct BitReader {                                   // bitstream.h:91-97
    const u8* base; size_t size; size_t at; u64 win; u32 used;
    DEV size_t init(const u8* src, size_t n)         // BIT_initDStream, bitstream.h:272-318
    {
        base = src; size = n; at = 0; win = 0; used = 0;
        if (n < 1) return FERR(srcSize_wrong);
        const u32 last = src[n - 1];
        if (n >= 8) {
            at = n - 8; win = ld64(src + at);
            if (last == 0) return FERR(GENERIC);
            used = 8 - hibit32(last);
        } else {
            for (size_t k = 0; k < n; ++k) win |= (u64)src[k] << (8 * k);
            if (last == 0) return FERR(corruption_detected);
            used = 8 - hibit32(last) + (u32)(8 - n) * 8;
        }
        return n;
    }
    DEV u32 read(u32 nb)                             // BIT_readBits (lookBits :345 + skipBits)
    {
        const u32 v = (u32)((win >> ((64u - used - nb) & 63u)) & (((u64)1 << nb) - 1));
        used += nb; return v;
    }
    DEV u32 read_fast(u32 nb)                        // BIT_readBitsFast (:361)
    {
        const u32 v = (u32)((win << (used & 63u)) >> ((64u - nb) & 63u));
        used += nb; return v;
    }
    DEV int reload()                                 // BIT_reloadDStream, :400-439
    {
        if (used > 64) return BR_OVERFLOW;
        if (at >= 8) { at -= used >> 3; used &= 7; win = ld64(base + at); return BR_UNFINISHED; }
        if (at == 0) return used < 64 ? BR_END_OF_BUFFER : BR_COMPLETED;
        u32 nbytes = used >> 3; int res = BR_UNFINISHED;
        if (at < nbytes) { nbytes = (u32)at; res = BR_END_OF_BUFFER; }
        at -= nbytes; used -= nbytes * 8; win = ld64(base + at);
        return res;
    }
};

DEV u32 fse_step(u32& state, BitReader& r, const u32* cells, bool fast)   // FSE_decodeSymbol(Fast), fse.h:600-622
{
    const u32 c = cells[state];
    const u32 nb = c >> 24;
    const u32 low = fast ? r.read_fast(nb) : r.read(nb);
    state = (c & 0xFFFFu) + low;
    return (c >> 16) & 0xFFu;
}

__global__ __launch_bounds__(64) void k_fse_decode(FseDecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int lane = threadIdx.x;
    const size_t first = (size_t)blockIdx.x * a.G;

    for (int g = 0; g < a.G; ++g) {                  // stage DTables: wave-uniform control flow, coalesced
        const size_t b = first + g;
        if (b >= a.nBlocks) break;
        if (a.meta && a.meta[b].state == 0) continue;
        const u32* t = a.dtables + b * a.dtStrideU32;
        const u32 tl = t[0] & 0xFFFFu;
        if (tl > a.maxTableLog) continue;
        const u32 words = 1 + (1u << tl);
        u32* s = lds + (size_t)g * a.slotU32;
        for (u32 i = lane; i < words; i += 64) s[i] = t[i];
    }
    __syncthreads();
    if (lane >= a.G) return;
    const size_t b = first + lane;
    if (b >= a.nBlocks) return;
    u32 hdr = 0;
    if (a.meta) { if (a.meta[b].state == 0) return; hdr = a.meta[b].hdrSize; }
    const u32 h0 = a.dtables[b * a.dtStrideU32];
    const u32 tl = h0 & 0xFFFFu;
    if (tl > a.maxTableLog) { a.results[b] = FERR(tableLog_tooLarge); return; }
    const bool fast = (h0 >> 16) != 0;
    const u32* const cells = lds + (size_t)lane * a.slotU32 + 1;

    const u8* const in = view_ptr(a.csrc, b) + hdr;
    const size_t S = view_size(a.csrc, b) - hdr;     // hdr <= cSrcSize (FSE_readNCount never returns more)
    u8* const out = a.dst + b * a.dstStride;
    const long omax = (long)a.dstCapacity;
    long op = 0;

    BitReader r;
    {   const size_t e = r.init(in, S);
        if (is_err(e)) { a.results[b] = e; return; }
    }
    u32 s1 = r.read(tl); r.reload();                 // FSE_initDState x2, fse.h:577-584
    u32 s2 = r.read(tl); r.reload();

    // ------------------------------------------------------------------------------------------------
    // bulk fast-forward.  C = bits consumed from the end of the stream (pad + end mark included),
    // R = 8*S - C unread bits.  Start of an iteration with R >= 128 implies the reference's window
    // pointer stays >= start+8 (reload "unfinished"), and op < omax-3 is its second loop condition.
    // ------------------------------------------------------------------------------------------------
    if (S >= 16) {
        long R = (long)(8 * S) - (long)(8 * (S - 8 - r.at) + r.used);
        if (R >= 128 && op < omax - 3) {
            const u64 topbit = 8 * (u64)(uintptr_t)in + (u64)R;         // absolute bit address one above the next unread bit
            const u32* const wmin = (const u32*)((uintptr_t)in & ~(uintptr_t)3);
            const u32* w = (const u32*)(uintptr_t)(((topbit - 1) >> 5) << 2);   // aligned word holding the next unread bit
            const u32 sh = (u32)(32 * ((u64)((uintptr_t)w >> 2) + 1) - topbit);  // bits of *w above the cursor (0..31)
            u64 win = (((u64)w[0] << 32) | (u64)w[-1]) << sh;             // next unread bit sits at bit 63
            u32 avail = 64 - sh;
            w -= 2;
            u32 nxt = *(w >= wmin ? w : wmin);
            const bool al4 = (((uintptr_t)out) & 3u) == 0;
            do {
                u32 c, nb, word;
                // state1
                c = cells[s1]; nb = c >> 24;
                s1 = (c & 0xFFFFu) + (u32)((win >> 1) >> (63 - nb)); win <<= nb; avail -= nb; R -= nb;
                word = (c >> 16) & 0xFFu;
                // state2
                c = cells[s2]; nb = c >> 24;
                s2 = (c & 0xFFFFu) + (u32)((win >> 1) >> (63 - nb)); win <<= nb; avail -= nb; R -= nb;
                word |= ((c >> 16) & 0xFFu) << 8;
                if (avail <= 32) { win |= (u64)nxt << (32 - avail); avail += 32; --w; nxt = *(w >= wmin ? w : wmin); }
                // state1
                c = cells[s1]; nb = c >> 24;
                s1 = (c & 0xFFFFu) + (u32)((win >> 1) >> (63 - nb)); win <<= nb; avail -= nb; R -= nb;
                word |= ((c >> 16) & 0xFFu) << 16;
                // state2
                c = cells[s2]; nb = c >> 24;
                s2 = (c & 0xFFFFu) + (u32)((win >> 1) >> (63 - nb)); win <<= nb; avail -= nb; R -= nb;
                word |= (c >> 16) << 24;        // nbBits is shifted out of the top
                if (avail <= 32) { win |= (u64)nxt << (32 - avail); avail += 32; --w; nxt = *(w >= wmin ? w : wmin); }
                if (al4) *(u32*)(out + op) = word;
                else { out[op] = (u8)word; out[op + 1] = (u8)(word >> 8); out[op + 2] = (u8)(word >> 16); out[op + 3] = (u8)(word >> 24); }
                op += 4;
            } while (R >= 128 && op < omax - 3);
            // hand the exact (normalised) window back to the literal reader
            const u64 C = 8 * (u64)S - (u64)R;
            r.at = S - 8 - (size_t)(C >> 3);
            r.used = (u32)(C & 7);
            r.win = ld64(in + r.at);
        }
    }

    // ---- literal tail: remaining iterations of :201-218, then :222-235
    for (;;) {
        const int st = r.reload();
        if (!((st == BR_UNFINISHED) & (op < omax - 3))) break;
        out[op + 0] = (u8)fse_step(s1, r, cells, fast);
        out[op + 1] = (u8)fse_step(s2, r, cells, fast);
        out[op + 2] = (u8)fse_step(s1, r, cells, fast);
        out[op + 3] = (u8)fse_step(s2, r, cells, fast);
        op += 4;
    }
    size_t result;
    for (;;) {
        if (op > omax - 2) { result = FERR(dstSize_tooSmall); break; }
        out[op++] = (u8)fse_step(s1, r, cells, fast);
        if (r.reload() == BR_OVERFLOW) { out[op++] = (u8)fse_step(s2, r, cells, fast); result = (size_t)op; break; }
        if (op > omax - 2) { result = FERR(dstSize_tooSmall); break; }
        out[op++] = (u8)fse_step(s2, r, cells, fast);
        if (r.reload() == BR_OVERFLOW) { out[op++] = (u8)fse_step(s1, r, cells, fast); result = (size_t)op; break; }
    }
    a.results[b] = result;
}

hipError_t launch_fse_decode(FseDecArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    static bool attrSet = false;
    const size_t ldsBytes = 80 * 1024;
    if (!attrSet) {
        hipError_t e = hipFuncSetAttribute((const void*)k_fse_decode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        attrSet = true;
    }
    a.slotU32 = 1 + (1u << a.maxTableLog);           // odd word stride: slots start on rotating banks
    a.G = (int)(ldsBytes / (a.slotU32 * 4));
    if (a.G > 64) a.G = 64;
    if (a.G < 1) return hipErrorInvalidValue;
    const size_t groups = (a.nBlocks + a.G - 1) / a.G;
    hipLaunchKernelGGL(k_fse_decode, dim3((unsigned)groups), dim3(64), ldsBytes, s, a);
    return hipGetLastError();
}
