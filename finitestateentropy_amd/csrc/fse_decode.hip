// fse_decode.hip -- a3: FSE_decompress_usingDTable over a batch
// (reference: lib/fse_decompress.c:178-252, lib/fse.h:577-622, lib/bitstream.h:272-448).
//
// tANS decoding is one loop-carried chain per block (two interleaved states sharing one bit cursor), so
// the mapping is "one lane per block, many blocks per CU": a 64-lane workgroup stages G DTables into LDS
// (verbatim reference layout, coalesced copy) and lane g walks block g's stream.  Throughput comes from
// the number of blocks resident per CU (bounded by LDS: 160 KiB / 8 KiB tables at tableLog 11) and from
// the instruction count of the per-symbol step, because a wavefront issues one instruction every few
// cycles no matter how many of its lanes are active.
//
// The kernel keeps the reference's own decoder state -- a 64-bit little-endian window at byte offset
// `at` plus a consumed-bit count `used` (lib/bitstream.h:91-97) -- so results are identical by
// construction, including on truncated / corrupt input:
//   * `BitReader` is a literal device restatement of BIT_initDStream / BIT_readBits / BIT_reloadDStream;
//   * the bulk loop is lib/fse_decompress.c:201-218 (4 symbols per reload) for as long as the window is at
//     least 24 bytes above the stream start.  There the reload is always the "fast" one (:378-388), and
//     the new window is not loaded but funnel-shifted out of two speculatively prefetched 8-byte words
//     below it, so no memory latency sits on the dependent chain (gfx950 global memory accepts the
//     unaligned 8-byte accesses this needs).
#include "internal.h"

#include "bitreader.h"

// one bulk step: cell lookup, take nb bits from the top of `t` (the not-yet-consumed window, MSB aligned),
// next state = newState + bits, symbol byte inserted into `word` with one v_perm.
#define FSE_BULK_STEP(state, SEL)                                                          \
    {   const u32 c = cells[state];                                                        \
        const u32 nb = c >> 24;                                                            \
        const u32 bits = __builtin_amdgcn_ubfe((u32)(t >> 32), 32u - nb, nb);              \
        t <<= nb; used += nb;                                                              \
        state = (c & 0xFFFFu) + bits;                                                      \
        word = __builtin_amdgcn_perm(c, word, SEL);                                        \
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

    // ---- bulk: iterations of fse_decompress.c:201-218 whose loop-head reload is provably the fast one
    if (r.at >= 24 && op < omax - 3 && r.used <= 64) {
        u64 at = r.at;
        u32 used = r.used;
        u64 win = r.win;
        u64 lo1 = ldg64u(in + at - 8);               // bytes [at-8, at)
        u64 lo2 = ldg64u(in + at - 16);              // bytes [at-16, at-8)
        long groups = (omax - 3 - op + 3) >> 2;      // iterations allowed by "op < olimit"
        do {
            // BIT_reloadDStreamFast: at -= used>>3; used &= 7; window = 8 bytes at `at`
            const u32 k8 = used & ~7u;               // whole consumed bytes, in bits (0..48)
            at -= used >> 3; used &= 7;
            win = (win << k8) | ((lo1 >> 1) >> (63 - k8));
            lo1 = (lo1 << k8) | ((lo2 >> 1) >> (63 - k8));
            lo2 = ldg64u(in + at - 16);              // needed two reloads from now
            u64 t = win << used;
            u32 word = 0;
            FSE_BULK_STEP(s1, 0x03020106u)           // v_perm: {c = bytes 4..7, word = bytes 0..3}; byte k <- c.byte2 (index 6)
            FSE_BULK_STEP(s2, 0x03020600u)
            FSE_BULK_STEP(s1, 0x03060100u)
            FSE_BULK_STEP(s2, 0x06020100u)
            __builtin_memcpy(out + op, &word, 4);
            op += 4;
            --groups;
        } while (at >= 24 && groups > 0);
        r.at = (size_t)at; r.used = used; r.win = win;
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
    probe_before(PK_FSE_DECODE, s);
    hipLaunchKernelGGL(k_fse_decode, dim3((unsigned)groups), dim3(64), ldsBytes, s, a);
    probe_after(PK_FSE_DECODE, s);
    return hipGetLastError();
}
