// huf_decode.hip -- a5: HUF_decompress4X1_usingDTable over a batch
// (reference: lib/huf_decompress.c:194-354, dispatcher :980-997; lib/bitstream.h:272-448).
//
// The 4-stream layout gives four independent serial chains per block (one table lookup per symbol on the chain), so
// the mapping is "one lane per stream, as many blocks per CU as LDS holds", with the same division of labour as
// fse_decode.hip:
//   * a workgroup = 1 decoder wave + HD_SRV_WAVES service waves over G blocks (lane 4g+k of the decoder wave walks
//     stream k of block g); 2 workgroups per CU;
//   * the decoder lane touches registers and LDS only: the X1 table (2-byte cells, staged bit-reversed as {nbBits, byte}), a
//     256-byte ring of compressed input and a ring of 4-symbol output words per stream;
//   * the service waves own all global-memory traffic of the bulk loop, coalesced: 128-byte input refills, 64..128-byte
//     output rows; the two sides talk through per-stream control words in LDS (acquire/release, workgroup scope).
//
// Per stream the kernel reproduces the reference's reader state (64-bit window at byte offset `at`, consumed bits
// `used`), so the verdict -- every stream must end with BIT_endOfDStream, lib/huf_decompress.c:348-349 -- is the
// reference's by construction, also on corrupt input:
//   * bulk loop = the 4-symbols-per-reload iterations (HUF_decodeStreamX1 :219-224 / the lock-step loop :310-331) for
//     as long as the window stays at least 24 bytes above the stream start, where every reload is the "fast" one
//     (bitstream.h:378-388) and (ptr, bitsConsumed) are a function of the absolute bit position alone;
//   * initialisation and the last symbols of a stream run through the literal BitReader (bitreader.h).
// A stream's decoded symbols and final reader state depend only on that stream, so decoding the four streams
// independently (instead of in lock-step) yields the same result as the reference.
#include "internal.h"
#include "bitreader.h"

#define HD_PHASE 8               // bulk iterations per phase (4 symbols, <= 6 bytes each)
#define HD_OUT_RING 32           // per-stream ring of output words (4 symbols each); flushed from 16 words on
#define HD_IN_RING 256           // per-stream ring of compressed input, direct-mapped by offset mod 256
#define HD_IN_CHUNK 128          // refill granule: 32 lanes x 4 bytes
#define HD_IN_MIRROR 8           // the first bytes are mirrored behind the ring so reads of 3 dwords never wrap
#define HD_MAXG 16               // blocks per workgroup (64 streams = the lanes of the decoder wave)
#define HD_SRV_WAVES 4
#define HD_SRV_S (4 * HD_MAXG / HD_SRV_WAVES)       // streams per service wave
#define HD_THREADS (64 * (1 + HD_SRV_WAVES))
#define HD_SLOT_LOG 11u          // LDS table slots hold 1 << 11 cells (4 KiB) for the common class, 1 << 12 for tableLog-12 blocks

#ifdef HD_TIMING             // development aid: decoder-wave cycle accounting (phases run / polls waited)
__device__ unsigned long long g_hdTiming[4096 * 4];
extern "C" __attribute__((visibility("default"))) int FSEHIP_debug_hdTiming(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hdTiming), sizeof(g_hdTiming)); }
#define HDT(x) x
#else
#define HDT(x)
#endif
struct HdCtl {             // per stream, in LDS
    u32 pubIters;          // decoder -> service: bulk iterations completed (output words produced)
    u32 pubPofs;           // decoder -> service: byte offset of the topmost dword still read; bit 31 = bulk finished
    u32 srvFlushed;        // service -> decoder: output words already written to global memory
    int srvValidLo;        // service -> decoder: the input ring holds stream bytes [validLo, validLo + 256); INT_MAX = not yet
    int initValidLo;       // set-up constants for the service wave
    int S32;
    u32 inLo, inHi, outLo, outHi;
};                         // 40 bytes: with 392 bytes of rings per stream 14 blocks (56 streams) fit a workgroup's 80 KiB
#define HD_STREAM_AUX (HD_IN_RING + HD_IN_MIRROR + HD_OUT_RING * 4)     // rings of one stream, bytes

DEV u32 hd_load(const u32* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV int hd_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void hd_store(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void hd_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

DEV u32 hufx1_step(BitReader& r, const u16* cells, u32 dtLog)                // HUF_decodeSymbolX1, :194-201 (reference cells)
{
    const u32 v = (u32)((r.win << (r.used & 63u)) >> ((64u - dtLog) & 63u)); // BIT_lookBitsFast
    const u32 c = cells[v];
    r.used += c >> 8;
    return c & 0xFFu;
}

// Bulk loop, laid out like fse_decode.hip's bit-reversed loop: a lone wave is bound by its dependent chain and by the
// number of instructions it issues, so both are kept minimal -- per symbol one v_and_or (table address), the LDS read and
// one v_alignbit on the chain, a shift and a byte permute beside it.
//   * The service waves write the input ring in CONSUMPTION order (ring dword m = bit-reversed stream dword
//     Stop/4 - 1 - m), so bits are taken from the low end of the window and the cell's nbBits (its low 5 bits) is used
//     as shift amount as it comes.  The cursor is Q = (consumed bits) - 1, so that the window's bit 1 is the next
//     unread bit and "2 * index" is a mask of the window: cell address = tableBase | (window & (tableMask << 1)).
//   * Low-end-first bits are the code bit-reversed, so the table is staged bit-reversed: cell rev(i) = nbBits | byte << 8.
//   * Window = three ring dwords {w2:w1:w0} in registers (Q's dword and the two after it); it slides by selects and the
//     two dwords behind it are prefetched at the top of every iteration, all off the dependent chain: the first lookup
//     of an iteration takes its index from the window register carried over from the previous one (its low 20+ bits
//     are valid), the fresh {hi:lo} pair is formed while that lookup is in flight.
// One iteration = 4 symbols = at most 44 bits.
typedef const __attribute__((address_space(3))) u16* hd_lds_u16;
typedef const __attribute__((address_space(3))) u32* hd_lds_u32;
DEV u32 hd_cell(u32 win, u32 mask2, u32 tabOff) { return *(hd_lds_u16)(uintptr_t)((win & mask2) | tabOff); }
DEV void hd_bulk_phase(u32& Qref, u32 tabOff, u32 mask2, u32 myIn, u32* ring)
{
    u32 q4 = (Qref >> 5) << 2, bq = Qref & 31u;                  // byte offset of Q's dword in the consumption-order stream, Q's bit in it
    u32 w0, w1, w2;
    {   const hd_lds_u32 wp = (hd_lds_u32)(uintptr_t)(myIn + (q4 & (HD_IN_RING - 4)));
        w0 = wp[0]; w1 = wp[1]; w2 = wp[2]; }
    u32 lo = __builtin_amdgcn_alignbit(w1, w0, bq);
#pragma unroll 8
    for (int it = 0; it < HD_PHASE; ++it) {
        const hd_lds_u32 np = (hd_lds_u32)(uintptr_t)(myIn + ((q4 + 12u) & (HD_IN_RING - 4)));
        const u32 n0 = np[0], n1 = np[1];                        // the two dwords behind the window
        const u32 c1 = hd_cell(lo, mask2, tabOff);
        u32 l = __builtin_amdgcn_alignbit(w1, w0, bq), h = __builtin_amdgcn_alignbit(w2, w1, bq);   // 64 bits from Q
        l = __builtin_amdgcn_alignbit(h, l, c1); h >>= (c1 & 31u);
        const u32 c2 = hd_cell(l, mask2, tabOff);
        l = __builtin_amdgcn_alignbit(h, l, c2); h >>= (c2 & 31u);
        const u32 c3 = hd_cell(l, mask2, tabOff);
        l = __builtin_amdgcn_alignbit(h, l, c3); h >>= (c3 & 31u);
        const u32 c4 = hd_cell(l, mask2, tabOff);
        lo = __builtin_amdgcn_alignbit(h, l, c4);                // at least 20 valid bits: the next iteration's first index
        u32 word = __builtin_amdgcn_perm(c1, 0u, 0x03020105u);   // byte 0 <- symbol of c1 (its byte 1)
        word = __builtin_amdgcn_perm(c2, word, 0x03020500u);
        word = __builtin_amdgcn_perm(c3, word, 0x03050100u);
        word = __builtin_amdgcn_perm(c4, word, 0x05020100u);
        const u32 bqn = bq + ((c1 + c2 + c3 + c4) & 0xFFu);      // low bytes = nbBits (<= 11 each): no carry into the symbols
        const bool k1 = bqn >= 32u, k2 = bqn >= 64u;             // the window slides up by one / two dwords
        w0 = k2 ? w2 : (k1 ? w1 : w0);
        w1 = k2 ? n0 : (k1 ? w2 : w1);
        w2 = k2 ? n1 : (k1 ? n0 : w2);
        q4 += (bqn >> 5) << 2;
        bq = bqn & 31u;
        ring[it] = word;
    }
    Qref = (q4 << 3) + bq;
}

// stream dword at byte offset `off` (S = stream size) -> ring, bit-reversed, at the consumption-order position
DEV void hd_ring_put(u32* rg, int S, int off, u32 w)
{
    const u32 j = (u32)(((S + 3) & ~3) - 4 - off) & (HD_IN_RING - 1);
    w = __brev(w);
    rg[j >> 2] = w;
    if (j < HD_IN_MIRROR) rg[(HD_IN_RING + j) >> 2] = w;
}

// ---- service wave: looks after HD_SRV_S streams (lane l keeps the books of stream s0 + l)
DEV void hd_service(int nStreams, u8* aux, HdCtl* ctlAll, int lane, int s0)
{
    const int myS = s0 + (lane < HD_SRV_S ? lane : 0);
    HdCtl* const ctl = ctlAll + myS;
    const unsigned long long inBits = ((unsigned long long)ctl->inHi << 32) | ctl->inLo;
    const unsigned long long outBits = ((unsigned long long)ctl->outHi << 32) | ctl->outLo;
    const int S32 = ctl->S32;
    int validLo = ctl->initValidLo;
    u32 flushed = 0;
    bool live = lane < HD_SRV_S && myS < nStreams && !(ctl->pubPofs >> 31);
    // (global-address-space pointers: a flat_* access would count on lgkmcnt too and every LDS wait of this wave would wait for the
    //  global loads in flight -- fse_decode.hip)
    typedef const __attribute__((address_space(1))) u8* hd_g_u8; typedef u32 __attribute__((aligned(1))) hd_u32_u;
    const int half = lane >> 5, l32 = lane & 31;             // input refills: 32 lanes per stream, two streams per instruction

    // initial fill: both chunks of every live stream (the topmost dword may straddle the end of the stream), then publish
    {   const unsigned long long am = __ballot(live);
#pragma unroll
        for (int l = 0; l < HD_SRV_S; ++l) {
            if (!((am >> l) & 1ull)) continue;               // uniform
            const int vlo = __shfl(validLo, l, WAVE), Sg = __shfl(S32, l, WAVE);
            const hd_g_u8 ig = (hd_g_u8)(uintptr_t)__shfl(inBits, l, WAVE);
            u32* const rg = (u32*)(aux + (size_t)(s0 + l) * HD_STREAM_AUX);
            const int off = vlo + 4 * lane;                  // 64 lanes x 4 bytes = the whole ring
            if (off >= 0 && off + 4 <= Sg) { const u32 w = *(const __attribute__((address_space(1))) hd_u32_u*)(ig + off); hd_ring_put(rg, Sg, off, w); }
            else if (off >= 0 && off < Sg) {
                u32 w = 0;
                for (int i = 0; i < 3; ++i) if (off + i < Sg) w |= (u32)ig[off + i] << (8 * i);
                hd_ring_put(rg, Sg, off, w);
            }
        }
        if (live) hd_store(&ctl->srvValidLo, validLo);
    }

    u32 pend[HD_SRV_S / 2];
    u32 outw[HD_SRV_S / 2];
#pragma unroll
    for (int l = 0; l < HD_SRV_S / 2; ++l) { pend[l] = 0; outw[l] = 0; }
    for (;;) {
        u32 pp = 0x80000000u, it = flushed;
        if (live) { pp = hd_load(&ctl->pubPofs); it = hd_load(&ctl->pubIters); }   // finished flag before the count it guards
        const bool fin = (pp >> 31) != 0;
        const int P = (int)(pp & 0x7FFFFFFFu);
        const u32 avail = it - flushed;
        const bool wantFlush = live && (avail >= HD_OUT_RING / 2 || (fin && avail > 0));
        // the chunk [validLo-128, validLo) lands on the ring bytes of [validLo+128, validLo+256): the decoder must be below
        const bool wantFill = live && !fin && validLo > 0 && P + 4 <= validLo + HD_IN_CHUNK;
        const unsigned long long fm = __ballot(wantFlush), rm = __ballot(wantFill);
        if (live && fin && avail == 0) live = false;
        if (!(fm | rm)) {
            if (!__any(live)) break;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        // (1) request input chunks: streams 2p and 2p+1 of this wave share one load instruction (32 lanes each)
#pragma unroll
        for (int p = 0; p < HD_SRV_S / 2; ++p) {
            if (!((rm >> (2 * p)) & 3ull)) continue;         // uniform
            const int l = 2 * p + half;
            const bool on = (rm >> l) & 1ull;
            const int off = __shfl(validLo, l, WAVE) - HD_IN_CHUNK + 4 * l32;
            const int Sg = __shfl(S32, l, WAVE);
            const hd_g_u8 ig = (hd_g_u8)(uintptr_t)__shfl(inBits, l, WAVE);
            u32 w = 0;
            if (on && off >= 0 && off + 4 <= Sg) w = *(const __attribute__((address_space(1))) hd_u32_u*)(ig + off);
            pend[p] = w;
        }
        // (2) read the output words of every stream with enough of them: at most 32 words each, so streams 2p and 2p+1 of
        //     this wave share one instruction (32 lanes each)
#pragma unroll
        for (int p = 0; p < HD_SRV_S / 2; ++p) {
            if (!((fm >> (2 * p)) & 3ull)) continue;         // uniform
            const int l = 2 * p + half;
            const bool on = (fm >> l) & 1ull;
            const u32 cnt = (u32)__shfl((int)avail, l, WAVE), fl = (u32)__shfl((int)flushed, l, WAVE);
            const u32* const og = (const u32*)(aux + (size_t)(s0 + l) * HD_STREAM_AUX + HD_IN_RING + HD_IN_MIRROR);
            if (on && (u32)l32 < cnt) outw[p] = og[(fl + l32) & (HD_OUT_RING - 1)];
        }
        // (3) install the input chunks and publish them
#pragma unroll
        for (int p = 0; p < HD_SRV_S / 2; ++p) {
            if (!((rm >> (2 * p)) & 3ull)) continue;         // uniform
            const int l = 2 * p + half;
            const bool on = (rm >> l) & 1ull;
            const int nlo = __shfl(validLo, l, WAVE) - HD_IN_CHUNK;
            const int Sg = __shfl(S32, l, WAVE);
            if (on) hd_ring_put((u32*)(aux + (size_t)(s0 + l) * HD_STREAM_AUX), Sg, nlo + 4 * l32, pend[p]);
        }
        if (wantFill) { validLo -= HD_IN_CHUNK; hd_store(&ctl->srvValidLo, validLo); }
        // (4) the output words are in registers: hand the slots back, then store them (rows of 16..32 words)
        if (wantFlush) hd_store(&ctl->srvFlushed, it);
#pragma unroll
        for (int p = 0; p < HD_SRV_S / 2; ++p) {
            if (!((fm >> (2 * p)) & 3ull)) continue;         // uniform
            const int l = 2 * p + half;
            const bool on = (fm >> l) & 1ull;
            const u32 cnt = (u32)__shfl((int)avail, l, WAVE), fl = (u32)__shfl((int)flushed, l, WAVE);
            __attribute__((address_space(1))) u8* const og = (__attribute__((address_space(1))) u8*)(uintptr_t)(__shfl(outBits, l, WAVE) + 4ull * fl);
            if (on && (u32)l32 < cnt) *(__attribute__((address_space(1))) hd_u32_u*)(og + 4u * l32) = outw[p];
        }
        if (wantFlush) flushed = it;
    }
}

// LDS: G tables (2-byte cells, 1 << ldsLog of them) | HdCtl[4 * HD_MAXG] | per stream: input ring (256 + 8 B), output ring (32 x 4 B)
__global__ __launch_bounds__(HD_THREADS) void k_huf_decode(HufDecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // slot g of this workgroup = entry first + g of the launch's block list (or simply block first + g)
    const size_t first = (size_t)blockIdx.x * a.G;
    const size_t nTot = a.count ? (size_t)*a.count : a.nBlocks;
    if (first >= nTot) return;                                   // uniform: the grid is sized for the worst case
    u8* const lds8 = (u8*)lds;
    const u32 tabStride = 2u << a.ldsLog;
    HdCtl* const ctlAll = (HdCtl*)(lds8 + (size_t)a.G * tabStride);
    u8* const aux = (u8*)ctlAll + (((size_t)4 * a.G * sizeof(HdCtl) + 15) & ~(size_t)15);
    const int nStreams = 4 * a.G;

    // ---- stage the X1 tables: reference cells {byte, nbBits} -> bit-reversed order, {nbBits, byte} (uniform control flow, all waves).
    //      Tables that do not fit the slot (tableLog 12) stay in global memory and are decoded by the literal path.
    for (int g = 0; g < a.G; ++g) {
        if (first + g >= nTot) break;
        const size_t b = a.list ? (size_t)a.list[first + g] : first + g;
        if (a.meta && a.meta[b].state == 0) continue;
        if (a.onlyDeclined && a.results[b] != HUF_DECLINED) continue;
        const u32* t = a.dtables + b * a.dtStrideU32;
        const u32 desc = t[0];
        const u32 tl = (desc >> 16) & 0xFFu;
        if (tl > a.maxTableLog || tl > a.ldsLog || tl < 1 || ((desc >> 8) & 0xFFu) != 0) continue;
        const u32 words = 1u << (tl - 1);                        // two cells per word
        u16* s = (u16*)(lds8 + (size_t)g * tabStride);
        for (u32 i = tid; i < words; i += HD_THREADS) {
            const u32 w = t[1 + i];                              // cells 2i, 2i+1: byte | nbBits << 8 each
            const u32 r0 = __brev(2u * i) >> (32u - tl);         // cell 2i+1 goes to r0 | tableSize/2
            s[r0] = (u16)(((w >> 8) & 0xFFu) | ((w & 0xFFu) << 8));
            s[r0 | (1u << (tl - 1))] = (u16)(((w >> 24) & 0xFFu) | (((w >> 16) & 0xFFu) << 8));
        }
    }
    __syncthreads();

    // ---- per-stream set-up by the decoder wave: lane 4g+k = stream k of block first+g
    const u32 g = (u32)lane >> 2, k = (u32)lane & 3u;
    const bool inRange = (int)g < a.G && first + g < nTot;
    const size_t b = inRange ? (a.list ? (size_t)a.list[first + g] : first + g) : 0;
    const bool live = wave == 0 && inRange && !(a.meta && a.meta[b].state == 0) && !(a.onlyDeclined && a.results[b] != HUF_DECLINED);

    size_t ierr = 0;                                 // BIT_initDStream error of my stream (0 = none)
    size_t blockErr = 0;                             // errors detected before any stream is touched
    size_t dstSize = 0;
    u32 dtLog = 0;
    const u8* sp = nullptr; u8* out = nullptr;
    size_t oStart = 0; long cnt = 0;
    BitReader r; r.base = nullptr; r.size = 0; r.at = 0; r.win = 0; r.used = 0;
    bool streamOk = false;                           // the stream was initialised and has to be decoded
    const u16* gcells = nullptr;
    if (live) {
        const u32 hdr = a.meta ? a.meta[b].hdrSize : 0;
        const u32* const gt = a.dtables + b * a.dtStrideU32;
        const u32 desc = gt[0];
        dtLog = (desc >> 16) & 0xFFu;
        gcells = (const u16*)(gt + 1);
        const u8* const in = view_ptr(a.csrc, b) + hdr;
        const size_t cSize = view_size(a.csrc, b) - hdr;
        dstSize = view_size(a.dstSizes, b);
        out = a.dst + b * a.dstStride;
        if (((desc >> 8) & 0xFFu) != 0) blockErr = FERR(GENERIC);                       // X2 table: huf_decompress.c:411-412 (4X1 entry point)
        else if (dtLog > a.maxTableLog) blockErr = FERR(tableLog_tooLarge);
        else if (dtLog < 1) blockErr = FERR(corruption_detected);                       // no table has tableLog 0 (a zero-filled DTable): the reference's look-up
                                                                                        // would shift by 64 and index with the whole container -- undefined there, refused here
        else if (cSize < 10) blockErr = FERR(corruption_detected);                      // :269
        if (!blockErr) {
            const size_t l1 = ld16(in), l2 = ld16(in + 2), l3 = ld16(in + 4);
            const size_t l4 = cSize - (l1 + l2 + l3 + 6);
            if (l4 > cSize) blockErr = FERR(corruption_detected);                       // :303
            else {
                const size_t seg = (dstSize + 3) / 4;
                const size_t sStart = 6 + (k > 0 ? l1 : 0) + (k > 1 ? l2 : 0) + (k > 2 ? l3 : 0);
                const size_t sLen = k == 0 ? l1 : k == 1 ? l2 : k == 2 ? l3 : l4;
                sp = in + sStart;
                oStart = (size_t)k * seg;
                const size_t oEndRaw = k < 3 ? oStart + seg : dstSize;                  // pEnd of this stream (:292-299)
                cnt = oEndRaw > oStart ? (long)(oEndRaw - oStart) : 0;                  // symbols to regenerate
                const size_t e = r.init(sp, sLen);                                      // :304-307
                if (is_err(e)) ierr = e; else streamOk = true;
            }
        }
    }
    // bulk: iterations whose reload is certainly the fast one; 4 symbols each, all inside the stream's output range
    const bool inLds = dtLog >= 1 && dtLog <= a.ldsLog;
    bool can = streamOk && inLds && r.at >= 24 + 6 * HD_PHASE + 8 && cnt / 4 >= HD_PHASE && r.size < (1ull << 31)
               && oStart + (size_t)cnt <= dstSize;      // (degenerate tiny blocks whose segments overhang go bytewise)
    struct { u32 q, bq; } bs; bs.q = 0; bs.bq = 0;      // the reference-order view of the cursor: q = 4*(unread bits >> 5) - 8, bq = unread bits & 31
    const u32 R8 = 8u * (((u32)r.size + 3u) & ~3u);            // consumption-order cursor Q = R8 - unread bits - 1
    u32 Q = 0;
    long groups = 0;
    u32 iters = 0;
    int validLo = 0;
    if (can) {
        const u32 B = 8u * ((u32)r.at + 8u) - r.used;            // unread bits = bits [0, B) of the stream
        bs.q = 4u * (B >> 5) - 8u; bs.bq = B & 31u; Q = R8 - B - 1u;
        groups = cnt >> 2;
        validLo = ((int)bs.q + 8 - 124) & ~127;                  // P - validLo in [124, 252): ring reaches up to P + 4 and down to P - 16 - 6*HD_PHASE
    }
    HdCtl* const ctl = ctlAll + (lane < nStreams ? lane : 0);        // (the array holds 4 * G entries)
    if (wave == 0 && lane < nStreams) {
        ctl->pubIters = 0; ctl->pubPofs = can ? bs.q + 8u : 0x80000000u;
        ctl->srvFlushed = 0; ctl->srvValidLo = 0x7FFFFFFF;
        ctl->initValidLo = validLo; ctl->S32 = (int)(r.size < (1ull << 31) ? r.size : 0);
        const unsigned long long ib = (unsigned long long)(uintptr_t)sp, ob = (unsigned long long)(uintptr_t)(out + oStart);
        ctl->inLo = (u32)ib; ctl->inHi = (u32)(ib >> 32); ctl->outLo = (u32)ob; ctl->outHi = (u32)(ob >> 32);
    }
    __syncthreads();
    if (wave >= 1) { hd_service(nStreams, aux, ctlAll, lane, (wave - 1) * HD_SRV_S); return; }

    __builtin_amdgcn_s_setprio(3);                   // the decoder wave is the critical path of the workgroup
    const u32 ldsBase = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)lds8;
    if (ldsBase & (tabStride - 1)) __builtin_trap();             // the table address is formed with an OR (dynamic LDS starts at 0: no static LDS here)
    const u32 tabOff = ldsBase + (u32)((int)g < a.G ? g : 0) * tabStride;
    const u32 myIn = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)(aux + (size_t)lane * HD_STREAM_AUX);
    u32* const myOut = (u32*)(aux + (size_t)lane * HD_STREAM_AUX + HD_IN_RING + HD_IN_MIRROR);
    const u32 mask2 = ((1u << dtLog) - 1u) << 1;
    HDT(unsigned long long tRun = 0; unsigned long long tWait = 0; unsigned long long nRun = 0; unsigned long long nWait = 0; unsigned long long tA = __builtin_readcyclecounter();)
    while (__any(can)) {
        bool ready = false;
        if (can) {
            const u32 fl = hd_load(&ctl->srvFlushed);
            const int vlo = hd_load(&ctl->srvValidLo);
            ready = (iters + HD_PHASE - fl <= HD_OUT_RING) && ((int)bs.q - 6 * HD_PHASE - 8 >= vlo);
        }
        if (ready) {
            hd_bulk_phase(Q, tabOff, mask2, myIn, myOut + (iters & (HD_OUT_RING - 1)));
            {   const u32 B = R8 - Q - 1u; bs.q = 4u * (B >> 5) - 8u; bs.bq = B & 31u; }
            iters += HD_PHASE; groups -= HD_PHASE;
            can = bs.q >= 24u + 6u * HD_PHASE && groups >= HD_PHASE;
            hd_store(&ctl->pubIters, iters);
            hd_store(&ctl->pubPofs, can ? bs.q + 8u : ((bs.q + 8u) | 0x80000000u));
        }
        HDT({ const unsigned long long tB = __builtin_readcyclecounter(); if (__any(ready)) { tRun += tB - tA; ++nRun; } else { tWait += tB - tA; ++nWait; } tA = tB; })
        if (!__any(ready)) __builtin_amdgcn_s_sleep(2);
    }
    HDT(if (lane == 0 && blockIdx.x < 4096) { unsigned long long* t = g_hdTiming + 4 * blockIdx.x; t[0] = tRun; t[1] = tWait; t[2] = nRun; t[3] = nWait; })
    int endBad = 0;                                  // my stream did not end exactly
    if (streamOk) {
        long p = 4 * (long)iters;
        if (iters) {                                 // back to the reference's (ptr, bitsConsumed, container) after a reload
            const u32 B = 8u * (bs.q + 8u) + bs.bq;
            r.at = (size_t)((B + 7u) >> 3) - 8; r.used = 8u * ((u32)r.at + 8u) - B; r.win = ldg64u(sp + r.at);
        }
        u8* const o = out + oStart;
        // literal: HUF_decodeStreamX1 (:214-237) on the reference cells
        while ((r.reload() == BR_UNFINISHED) & (p < cnt - 3)) {
            for (u32 q4 = 0; q4 < 4; ++q4) { const u32 sym = hufx1_step(r, gcells, dtLog); if (oStart + p < dstSize) o[p] = (u8)sym; ++p; }
        }
        while (p < cnt) { const u32 sym = hufx1_step(r, gcells, dtLog); if (oStart + p < dstSize) o[p] = (u8)sym; ++p; }
        if (!(r.at == 0 && r.used == 64)) endBad = 1;                                   // BIT_endOfDStream, :348-349
    }
    // combine the four streams of a block: all four inits come first and the first failing one is returned
    // (:304-307); otherwise every stream must have ended exactly (:348-349).  All lanes take part in the shuffles.
    const u32 base4 = (u32)lane & ~3u;
    size_t res = 0;
    int anyEnd = 0;
#pragma unroll
    for (u32 q4 = 0; q4 < 4; ++q4) {
        const unsigned long long iq = __shfl((unsigned long long)ierr, (int)(base4 + q4), WAVE);
        const int eq = __shfl(endBad, (int)(base4 + q4), WAVE);
        if (res == 0 && iq != 0) res = (size_t)iq;
        anyEnd |= eq;
    }
    if (res == 0 && anyEnd) res = FERR(corruption_detected);
    if (live && k == 0 && !(a.acceptX2 && ((a.dtables[b * a.dtStrideU32] >> 8) & 0xFFu) == 1u)) {   // (double-symbol tables: k_huf_decode_x2)
        size_t result;
        if (blockErr) result = blockErr;
        else if (res) result = res;
        else result = dstSize;
        a.results[b] = result;
    }
}

// ---- double-symbol (X2) tables: HUF_decompress4X_usingDTable's other branch (lib/huf_decompress.c:749-862, 980-997) --------------
// A caller that built its table with the reference's HUF_readDTableX2 hands over cells {u16 sequence; u8 nbBits; u8 length}: a
// look-up yields one or two symbols.  The reference walks the four streams in lock step, stores two bytes per look-up and advances
// by `length`, so what lands in the overlap between neighbouring segments -- and the verdict on corrupt input -- depends on that
// lock step.  This is an acceptance path, not a throughput path: one lane walks one block with the reference's reader state
// (BitReader) and its loop structure, the table stays in global memory.
struct X2Stream { BitReader r; u8* op; };
// (a cell's length field is 1 or 2 in any table HUF_readDTableX2 builds; the low two bits keep the advance of a damaged caller table
//  within what the two bytes just stored cover, so the cursors cannot run away from the segment checks of :850-853)
DEV void x2_cell(X2Stream& s, const u32* cells, u32 dtLog)                               // HUF_decodeSymbolX2, :663-670
{
    const u32 v = (u32)((s.r.win << (s.r.used & 63u)) >> ((64u - dtLog) & 63u));
    const u32 c = cells[v];
    s.op[0] = (u8)c; s.op[1] = (u8)(c >> 8);
    s.r.used += (c >> 16) & 0xFFu;
    const u32 len = c >> 24;
    s.op += len > 2u ? 2u : len;
}
DEV int x2_reload_fast(BitReader& r)                                                     // BIT_reloadDStreamFast, bitstream.h:400-409
{
    if (r.at < 8) return BR_OVERFLOW;
    r.at -= r.used >> 3; r.used &= 7; r.win = ldg64u(r.base + r.at);
    return BR_UNFINISHED;
}
DEV void x2_finish(X2Stream& s, u8* pEnd, const u32* cells, u32 dtLog)                   // HUF_decodeStreamX2, :692-722
{
    while ((s.r.reload() == BR_UNFINISHED) & (s.op < pEnd - 7)) { x2_cell(s, cells, dtLog); x2_cell(s, cells, dtLog); x2_cell(s, cells, dtLog); x2_cell(s, cells, dtLog); }
    while ((s.r.reload() == BR_UNFINISHED) & (s.op <= pEnd - 2)) x2_cell(s, cells, dtLog);
    while (s.op <= pEnd - 2) x2_cell(s, cells, dtLog);
    if (s.op < pEnd) {                                                                   // HUF_decodeLastSymbolX2, :672-688
        const u32 v = (u32)((s.r.win << (s.r.used & 63u)) >> ((64u - dtLog) & 63u));
        const u32 c = cells[v];
        s.op[0] = (u8)c;
        if ((c >> 24) == 1) s.r.used += (c >> 16) & 0xFFu;
        else if (s.r.used < 64) { s.r.used += (c >> 16) & 0xFFu; if (s.r.used > 64) s.r.used = 64; }
        s.op += 1;
    }
}
__global__ __launch_bounds__(64) void k_huf_decode_x2(HufDecArgs a)
{
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= a.nBlocks) return;
    const u32* const gt = a.dtables + b * a.dtStrideU32;
    const u32 desc = gt[0];
    if (((desc >> 8) & 0xFFu) != 1u) return;                                              // single-symbol table: k_huf_decode's block
    if (a.onlyDeclined && a.results[b] != HUF_DECLINED) return;                           // decoded by the stream-parallel decoder
    const u32 dtLog = (desc >> 16) & 0xFFu;
    const u32* const cells = gt + 1;
    const u8* const in = view_ptr(a.csrc, b);
    const size_t cSize = view_size(a.csrc, b), dstSize = view_size(a.dstSizes, b);
    u8* const ostart = a.dst + b * a.dstStride;
    size_t result;
    do {
        if (dtLog > a.maxTableLog) { result = FERR(tableLog_tooLarge); break; }
        if (dtLog < 1) { result = FERR(corruption_detected); break; }                    // no table has tableLog 0 (the look-up would shift by 64)
        if (cSize < 10) { result = FERR(corruption_detected); break; }                   // :758
        const size_t l1 = ld16(in), l2 = ld16(in + 2), l3 = ld16(in + 4), l4 = cSize - (l1 + l2 + l3 + 6);
        if (l4 > cSize) { result = FERR(corruption_detected); break; }                   // :795
        const size_t seg = (dstSize + 3) / 4;
        u8* const oend = ostart + dstSize;
        u8* const start2 = ostart + seg; u8* const start3 = start2 + seg; u8* const start4 = start3 + seg;
        X2Stream s1, s2, s3, s4;
        s1.op = ostart; s2.op = start2; s3.op = start3; s4.op = start4;
        size_t e;
        e = s1.r.init(in + 6, l1); if (is_err(e)) { result = e; break; }                  // :796-799
        e = s2.r.init(in + 6 + l1, l2); if (is_err(e)) { result = e; break; }
        e = s3.r.init(in + 6 + l1 + l2, l3); if (is_err(e)) { result = e; break; }
        e = s4.r.init(in + 6 + l1 + l2 + l3, l4); if (is_err(e)) { result = e; break; }
        bool go = true;
        while (go && ((long)(s4.op - ostart) < (long)dstSize - 7)) {                       // :802-847 (the order of the non-x86-clang build)
            x2_cell(s1, cells, dtLog); x2_cell(s2, cells, dtLog); x2_cell(s3, cells, dtLog); x2_cell(s4, cells, dtLog);
            x2_cell(s1, cells, dtLog); x2_cell(s2, cells, dtLog); x2_cell(s3, cells, dtLog); x2_cell(s4, cells, dtLog);
            x2_cell(s1, cells, dtLog); x2_cell(s2, cells, dtLog); x2_cell(s3, cells, dtLog); x2_cell(s4, cells, dtLog);
            x2_cell(s1, cells, dtLog); x2_cell(s2, cells, dtLog); x2_cell(s3, cells, dtLog); x2_cell(s4, cells, dtLog);
            const int r1 = x2_reload_fast(s1.r), r2 = x2_reload_fast(s2.r), r3 = x2_reload_fast(s3.r), r4 = x2_reload_fast(s4.r);   // all four, always
            go = r1 == BR_UNFINISHED && r2 == BR_UNFINISHED && r3 == BR_UNFINISHED && r4 == BR_UNFINISHED;
        }
        if (s1.op > start2 || s2.op > start3 || s3.op > start4) { result = FERR(corruption_detected); break; }   // :850-853
        x2_finish(s1, start2, cells, dtLog); x2_finish(s2, start3, cells, dtLog); x2_finish(s3, start4, cells, dtLog); x2_finish(s4, oend, cells, dtLog);
        const bool ended = (s1.r.at == 0 && s1.r.used == 64) & (s2.r.at == 0 && s2.r.used == 64) & (s3.r.at == 0 && s3.r.used == 64) & (s4.r.at == 0 && s4.r.used == 64);
        result = ended ? dstSize : FERR(corruption_detected);                             // :861-866
    } while (0);
    a.results[b] = result;
}

// ---- HUF_decompress1X1_usingDTable / HUF_decompress1X_usingDTable, the literal path (lib/huf_decompress.c:239-260, 724-747, 955-975):
// the block is ONE stream.  The stream-parallel decoder takes it first (huf_decode_par.hip, a.streams == 1: the stream in pieces); this
// kernel -- one lane per block, the reference's reader state and loops, tables in global memory -- is for what it declines: tiny,
// irregular and corrupt blocks, and whatever the reference makes of a damaged table.
__global__ __launch_bounds__(64) void k_huf_decode_1x(HufDecArgs a)
{
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= a.nBlocks) return;
    if (a.onlyDeclined && a.results[b] != HUF_DECLINED) return;                           // decoded by the stream-parallel decoder
    const u32* const gt = a.dtables + b * a.dtStrideU32;
    const u32 desc = gt[0];
    const u32 tableType = (desc >> 8) & 0xFFu, dtLog = (desc >> 16) & 0xFFu;
    const u8* const in = view_ptr(a.csrc, b);
    const size_t cSize = view_size(a.csrc, b), dstSize = view_size(a.dstSizes, b);
    u8* const ostart = a.dst + b * a.dstStride;
    size_t result;
    do {
        if (tableType != 0 && !(a.acceptX2 && tableType == 1)) { result = FERR(GENERIC); break; }   // :367-369 (the 1X1 entry point)
        if (dtLog > a.maxTableLog) { result = FERR(tableLog_tooLarge); break; }
        if (dtLog < 1) { result = FERR(corruption_detected); break; }                    // no table has tableLog 0 (the look-up would shift by 64)
        if (tableType == 0) {
            const u16* const cells = (const u16*)(gt + 1);
            BitReader r;
            const size_t e = r.init(in, cSize); if (is_err(e)) { result = e; break; }     // :252
            size_t p = 0;                                                                 // HUF_decodeStreamX1, :214-237
            while ((r.reload() == BR_UNFINISHED) & ((long)p < (long)dstSize - 3)) {
                for (u32 q4 = 0; q4 < 4; ++q4) ostart[p++] = (u8)hufx1_step(r, cells, dtLog);
            }
            while (p < dstSize) ostart[p++] = (u8)hufx1_step(r, cells, dtLog);
            result = (r.at == 0 && r.used == 64) ? dstSize : FERR(corruption_detected);   // :256-258
        } else {
            X2Stream st; st.op = ostart;
            const size_t e = st.r.init(in, cSize); if (is_err(e)) { result = e; break; }  // :735
            x2_finish(st, ostart + dstSize, gt + 1, dtLog);                               // HUF_decodeStreamX2, :692-722
            result = (st.r.at == 0 && st.r.used == 64) ? dstSize : FERR(corruption_detected);
        }
    } while (0);
    a.results[b] = result;
}

static int huf_decode_G(size_t ldsBytes, unsigned ldsLog)
{
    const size_t perBlock = (2u << ldsLog) + 4 * (HD_STREAM_AUX + sizeof(HdCtl));
    int g = (int)((ldsBytes - 16) / perBlock);
    return g > HD_MAXG ? HD_MAXG : g;
}

static hipError_t huf_decode_launch(HufDecArgs a, hipStream_t s)
{
    const size_t ldsBytes = 80 * 1024;               // two workgroups per CU
    {   const hipError_t e = ensure_dyn_lds((const void*)k_huf_decode, (int)ldsBytes); if (e != hipSuccess) return e; }
    a.G = huf_decode_G(ldsBytes, a.ldsLog);
    a.slotU32 = 0;
    const size_t groups = (a.nBlocks + a.G - 1) / a.G;
    hipLaunchKernelGGL(k_huf_decode, dim3((unsigned)groups), dim3(HD_THREADS), ldsBytes, s, a);
    return hipGetLastError();
}

// caller-built tables (HUF_decompress4X1_usingDTable over a batch): one launch, slots sized by the caller's maxTableLog
// Four-stream blocks go to the stream-parallel decoder first (single-symbol tables as they are, double-symbol tables through the
// single-symbol cells it derives from them, huf_decode_par.hip); what it declines -- tiny / irregular / corrupt blocks, tables it
// cannot vouch for -- is marked HUF_DECLINED and taken by the serial kernel (X1) and the literal lock-step kernel (X2) below.
hipError_t launch_huf_decode(HufDecArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    a.list = nullptr; a.count = nullptr;
    a.ldsLog = a.maxTableLog > HD_SLOT_LOG ? FSEHIP_HUF_TABLELOG_MAX : HD_SLOT_LOG;
    a.onlyDeclined = 0;
    probe_before(PK_HUF_DECODE, s);
    hipError_t e = hipSuccess;
    if (!a.meta) {
        a.classLo = 0;
        if (HPAR_USE_TINY) { e = launch_huf_decode_par(a, HPAR_DATA_TINY, nullptr, nullptr, s); a.classLo = HPAR_DATA_TINY; }
        if (e == hipSuccess) e = launch_huf_decode_par(a, HPAR_DATA_SMALL, nullptr, nullptr, s);
        a.classLo = HPAR_DATA_SMALL;
        if (e == hipSuccess && !HPAR_ALL_SMALL) e = launch_huf_decode_par(a, HPAR_DATA_LARGE, nullptr, nullptr, s);
        a.classLo = 0;
        if (e == hipSuccess && a.acceptX2) e = launch_huf_decode_par_x2(a, s);
        a.onlyDeclined = 1;
    }
    if (a.streams == 1) {                               // HUF_decompress1X[1]_usingDTable: one stream per block
        if (e == hipSuccess) { hipLaunchKernelGGL(k_huf_decode_1x, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a); e = hipGetLastError(); }
        probe_after(PK_HUF_DECODE, s);
        return e;
    }
    if (e == hipSuccess) e = huf_decode_launch(a, s);
    if (e == hipSuccess && a.acceptX2) {                 // HUF_decompress4X_usingDTable: blocks whose table is a double-symbol one
        hipLaunchKernelGGL(k_huf_decode_x2, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a);
        e = hipGetLastError();
    }
    probe_after(PK_HUF_DECODE, s);
    return e;
}

// one-shot path: k_huf_dprep has sorted the blocks into class lists (internal.h: by tableLog -- 4 KiB or 8 KiB LDS table slots --
// and by decoder); one launch per list, the serial decoder's last because the parallel one appends what it declines to them
hipError_t launch_huf_decode_classes(HufDecArgs a, u32* lists, u32* counts, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    probe_before(PK_HUF_DECODE, s);
    for (int c = 0; c < HUF_DCLS_COUNT && e == hipSuccess; ++c) {
        const int kind = c >> 1, ser = 2 * HUF_DKIND_SERIAL + (c & 1);
        a.list = lists + (size_t)c * a.nBlocks; a.count = counts + c;
        a.ldsLog = (c & 1) ? FSEHIP_HUF_TABLELOG_MAX : HD_SLOT_LOG;
        if (kind == HUF_DKIND_PAR_TINY && !HPAR_USE_TINY) continue;           // (nothing is filed there)
        if (kind == HUF_DKIND_PAR_LARGE && HPAR_ALL_SMALL) continue;
        if (kind == HUF_DKIND_SERIAL) e = huf_decode_launch(a, s);
        else e = launch_huf_decode_par(a, kind == HUF_DKIND_PAR_TINY ? HPAR_DATA_TINY : kind == HUF_DKIND_PAR_SMALL ? HPAR_DATA_SMALL : HPAR_DATA_LARGE, lists + (size_t)ser * a.nBlocks, counts + ser, s);
    }
    probe_after(PK_HUF_DECODE, s);
    return e;
}
