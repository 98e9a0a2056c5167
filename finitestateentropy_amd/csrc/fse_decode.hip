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
#include <stdlib.h>

#include "bitreader.h"

// one bulk step against the compact LDS table: A[x] = newState (12 bits) | nbBits << 12.
// The unread window is {thi:tlo}, MSB aligned, handled with full-rate 32-bit ops (64-bit shifts are quarter rate):
//   bits = top nb bits of thi;  {thi:tlo} <<= nb via one v_alignbit;  next state = newState | bits
// (FSE_buildDTable makes the low nbBits of newState zero, lib/fse_decompress.c:121-122; tables that violate this are
// decoded by the literal path).  NB0 = some cell of some table in this wave has nbBits == 0 (fastMode 0):
// v_alignbit with a shift of 32 would return tlo, so the window update is made conditional.
template <bool NB0>
DEV void fse_bulk_consume(u32 c, u32& state, u32& thi, u32& tlo, u32& u)
{
    const u32 nb = c >> 12;
    const u32 m = 32u - nb;
    const u32 bits = __builtin_amdgcn_ubfe(thi, m, nb);
    const u32 nhi = __builtin_amdgcn_alignbit(thi, tlo, m);
    thi = NB0 ? (nb ? nhi : thi) : nhi;
    tlo <<= nb;
    u += nb;
    state = (c & 0xFFFu) | bits;
}

// 8 bytes at (byte offset - k) of the 16-byte little-endian value {a3:a2:a1:a0} whose upper half {a3:a2} sits at the
// current offset, k = 0..7: two v_perm with a computed selector after a word-level select.
DEV void funnel_bytes(u32 a3, u32 a2, u32 a1, u32 a0, u32 k, u32& hi, u32& lo)
{
    const bool far = k > 4u;
    const u32 b2 = far ? a2 : a3, b1 = far ? a1 : a2, b0 = far ? a0 : a1;
    const u32 r = 4u - (far ? k - 4u : k);              // byte offset inside the selected pair, 0..4
    const u32 sel = 0x03020100u + r * 0x01010101u;
    hi = __builtin_amdgcn_perm(b2, b1, sel);
    lo = __builtin_amdgcn_perm(b1, b0, sel);
}

#define FSE_DEC_RING 64          // per-block LDS state ring: one entry (4 states, 8 bytes) per bulk iteration
#define FSE_IN_RING 512          // per-block LDS input ring (bytes of compressed stream, direct-mapped by offset mod 512)
#define FSE_IN_CHUNK 256         // refill granule: one coalesced 4-byte load per lane
#define FSE_IN_MIRROR 16         // the first bytes are mirrored behind the ring so 12-byte reads never wrap
#define FSE_CHECK_EVERY 16       // bulk iterations per phase (<= 6 bytes consumed per iteration)
#define FSE_MAXG 16              // blocks per workgroup (one pending refill register per block)

struct BulkState { u32 s1, s2, pofs, u, whi, wlo, l1hi, l1lo, l2hi, l2lo; };

// One phase = FSE_CHECK_EVERY iterations of lib/fse_decompress.c:201-218 for one lane, registers + LDS only.
// The table lookups of the two interleaved states are issued together so each pair costs one LDS round trip.
template <bool NB0>
DEV void fse_bulk_phase(BulkState& b, const u16* A, const u8* myIn, uint2* myRing, u32 iters)
{
    u32 s1 = b.s1, s2 = b.s2, pofs = b.pofs, u = b.u, whi = b.whi, wlo = b.wlo, l1hi = b.l1hi, l1lo = b.l1lo, l2hi = b.l2hi, l2lo = b.l2lo;
#pragma unroll 2
    for (int it = 0; it < FSE_CHECK_EVERY; ++it) {
        const u32 c1 = A[s1], c2 = A[s2];            // first pair of lookups in flight during the reload arithmetic
        // BIT_reloadDStreamFast (at -= used>>3; used &= 7) in the shifted convention: k = used>>3 = 0..6 bytes
        const u32 k = (u >> 3) - 1u;
        pofs -= k; u = (u & 7u) + 8u;
        u32 nwhi, nwlo, n1hi, n1lo;
        funnel_bytes(whi, wlo, l1hi, l1lo, k, nwhi, nwlo);
        funnel_bytes(l1hi, l1lo, l2hi, l2lo, k, n1hi, n1lo);
        whi = nwhi; wlo = nwlo; l1hi = n1hi; l1lo = n1lo;
        {   // bytes [p-16, p-8) from the input ring (needed at the next reload)
            const u32 j = (pofs - 16u) & (FSE_IN_RING - 1);
            const u32* rp = (const u32*)(myIn + (j & ~3u));
            const u32 w0 = rp[0], w1 = rp[1], w2 = rp[2];
            const u32 sh = (j & 3u) * 8u;
            l2lo = __builtin_amdgcn_alignbit(w1, w0, sh);
            l2hi = __builtin_amdgcn_alignbit(w2, w1, sh);
        }
        u32 thi = __builtin_amdgcn_alignbit(whi, wlo, 32u - u), tlo = wlo << u;   // window << u, u in [8,15]
        uint2 rec;
        rec.x = s1 | (s2 << 16);
        fse_bulk_consume<NB0>(c1, s1, thi, tlo, u);
        fse_bulk_consume<NB0>(c2, s2, thi, tlo, u);
        const u32 c3 = A[s1], c4 = A[s2];            // second pair
        rec.y = s1 | (s2 << 16);
        fse_bulk_consume<NB0>(c3, s1, thi, tlo, u);
        fse_bulk_consume<NB0>(c4, s2, thi, tlo, u);
        myRing[(iters + it) & (FSE_DEC_RING - 1)] = rec;
    }
    b.s1 = s1; b.s2 = s2; b.pofs = pofs; b.u = u; b.whi = whi; b.wlo = wlo; b.l1hi = l1hi; b.l1lo = l1lo; b.l2hi = l2hi; b.l2lo = l2lo;
}

// Per-block control words in LDS: the decoder wave and the service wave of a workgroup talk through these only.
struct DecCtl {
    u32 pubIters;      // decoder -> service: bulk iterations completed (state-ring records produced)
    u32 pubPofs;       // decoder -> service: window position p = at+1; bit 31 = bulk finished (pubIters is final)
    u32 srvFlushed;    // service -> decoder: state-ring records already turned into output bytes
    int srvValidLo;    // service -> decoder: the input ring holds stream bytes [validLo, validLo + 512); INT_MAX = not yet
    int initValidLo;   // set-up constants for the service wave
    int S32;
    u32 inLo, inHi, outLo, outHi, symLo, symHi;
    u32 pad[4];
};
#define FSE_DEC_THREADS 128
#define FSE_CTL_BYTES (FSE_MAXG * (u32)sizeof(DecCtl))

DEV u32 ctl_load(const u32* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV int ctl_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void ctl_store(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void ctl_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- service wave: keeps the input rings filled and turns state-ring records into output bytes, so that the decoder
//      wave spends its cycles on the dependent chains only.  Everything here is wave-cooperative and coalesced:
//        * input : 256-byte chunks, one 4-byte load per lane, written below the bytes the decoder is reading;
//        * output: record i of a block = the 4 states iteration i decoded FROM; symbol = cell[state].symbol, gathered
//                  from the L2-resident table in global memory, packed, stored as one 256-byte row per 64 records.
DEV void fse_decode_service(const FseDecArgs& a, u8* ldsb, DecCtl* ctlAll, u32 slotBytes, u32 ringOff, u32 inOff, int lane)
{
    DecCtl* const ctl = ctlAll + (lane < FSE_MAXG ? lane : 0);
    const bool mineValid = lane < a.G;
    // per-block constants live in the registers of lane g of this wave
    const unsigned long long inBits = ((unsigned long long)ctl->inHi << 32) | ctl->inLo;
    const unsigned long long outBits = ((unsigned long long)ctl->outHi << 32) | ctl->outLo;
    const unsigned long long tabBits = ((unsigned long long)ctl->symHi << 32) | ctl->symLo;
    const int S32 = ctl->S32;
    int validLo = ctl->initValidLo;
    u32 flushed = 0;
    bool live = mineValid && !(ctl->pubPofs >> 31);          // blocks that never enter the bulk loop need no service

    // initial fill: two chunks per live block, then publish
    {   const unsigned long long am = __ballot(live);
#pragma unroll
        for (int g = 0; g < FSE_MAXG; ++g) {
            if (!((am >> g) & 1ull)) continue;               // uniform
            const int vlo = __shfl(validLo, g, WAVE), Sg = __shfl(S32, g, WAVE);
            const u8* const ig = (const u8*)(uintptr_t)__shfl(inBits, g, WAVE);
            u32* const rg = (u32*)(ldsb + (size_t)g * slotBytes + inOff);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int off = vlo + FSE_IN_CHUNK * c + 4 * lane;
                if (off >= 0 && off + 4 <= Sg) {
                    u32 w; __builtin_memcpy(&w, ig + off, 4);
                    const u32 j = (u32)off & (FSE_IN_RING - 1);
                    rg[j >> 2] = w;
                    if (j < FSE_IN_MIRROR) rg[(FSE_IN_RING + j) >> 2] = w;
                }
            }
        }
        if (live) ctl_store(&ctl->srvValidLo, validLo);
    }

    u32 pend[FSE_MAXG];
    u32 yq[FSE_MAXG][4];
#pragma unroll
    for (int g = 0; g < FSE_MAXG; ++g) { pend[g] = 0; yq[g][0] = yq[g][1] = yq[g][2] = yq[g][3] = 0; }
    for (;;) {
        // snapshot of the decoder's progress (the finished flag is read before the iteration count it guards)
        u32 pp = 0x80000000u, it = flushed;
        if (live) { pp = ctl_load(&ctl->pubPofs); it = ctl_load(&ctl->pubIters); }
        const bool fin = (pp >> 31) != 0;
        const int pofs = (int)(pp & 0x7FFFFFFFu);
        const u32 avail = it - flushed;
        const bool wantFlush = live && (avail >= 32u || (fin && avail > 0));
        const bool wantFill = live && !fin && validLo > 0 && pofs <= validLo + FSE_IN_CHUNK + 8;
        const unsigned long long fm = __ballot(wantFlush), rm = __ballot(wantFill);
        if (live && fin && avail == 0) live = false;
        if (!(fm | rm)) {
            if (!__any(live)) break;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        // (1) request the next input chunk of every block that is about to need it
#pragma unroll
        for (int g = 0; g < FSE_MAXG; ++g) {
            if (!((rm >> g) & 1ull)) continue;               // uniform
            const int off = __shfl(validLo, g, WAVE) - FSE_IN_CHUNK + 4 * lane;
            const int Sg = __shfl(S32, g, WAVE);
            const u8* const ig = (const u8*)(uintptr_t)__shfl(inBits, g, WAVE);
            u32 w = 0;
            if (off >= 0 && off + 4 <= Sg) __builtin_memcpy(&w, ig + off, 4);
            pend[g] = w;
        }
        // (2) issue the symbol gathers of every block with enough records
#pragma unroll
        for (int g = 0; g < FSE_MAXG; ++g) {
            if (!((fm >> g) & 1ull)) continue;               // uniform
            const u32 cnt = (u32)__shfl((int)avail, g, WAVE), fl_g = (u32)__shfl((int)flushed, g, WAVE);
            const u8* const tg = (const u8*)(uintptr_t)__shfl(tabBits, g, WAVE);
            if ((u32)lane < cnt) {
                const uint2 rec = ((const uint2*)(ldsb + (size_t)g * slotBytes + ringOff))[(fl_g + lane) & (FSE_DEC_RING - 1)];
                yq[g][0] = tg[4u * (rec.x & 0xFFFFu) + 2]; yq[g][1] = tg[4u * (rec.x >> 16) + 2];
                yq[g][2] = tg[4u * (rec.y & 0xFFFFu) + 2]; yq[g][3] = tg[4u * (rec.y >> 16) + 2];
            }
        }
        // (3) install the input chunks and publish them
#pragma unroll
        for (int g = 0; g < FSE_MAXG; ++g) {
            if (!((rm >> g) & 1ull)) continue;               // uniform
            const int nlo = __shfl(validLo, g, WAVE) - FSE_IN_CHUNK;
            const u32 j = (u32)(nlo + 4 * lane) & (FSE_IN_RING - 1);
            u32* const rg = (u32*)(ldsb + (size_t)g * slotBytes + inOff);
            rg[j >> 2] = pend[g];
            if (j < FSE_IN_MIRROR) rg[(FSE_IN_RING + j) >> 2] = pend[g];
        }
        if (wantFill) { validLo -= FSE_IN_CHUNK; ctl_store(&ctl->srvValidLo, validLo); }
        // (4) the ring records are in registers now: hand the slots back, then pack and store the symbols
        if (wantFlush) { ctl_store(&ctl->srvFlushed, it); }
#pragma unroll
        for (int g = 0; g < FSE_MAXG; ++g) {
            if (!((fm >> g) & 1ull)) continue;               // uniform
            const u32 cnt = (u32)__shfl((int)avail, g, WAVE), fl_g = (u32)__shfl((int)flushed, g, WAVE);
            u8* const og = (u8*)(uintptr_t)__shfl(outBits, g, WAVE) + 4ull * fl_g;
            if ((u32)lane < cnt) {
                const u32 w = yq[g][0] | (yq[g][1] << 8) | (yq[g][2] << 16) | (yq[g][3] << 24);
                __builtin_memcpy(og + 4u * lane, &w, 4);
            }
        }
        if (wantFlush) flushed = it;
    }
}

// LDS: DecCtl[FSE_MAXG] | per block: A[2^maxTableLog] (u16) | state ring (64 x 8 B) | input ring (512 + 16 B)
__global__ __launch_bounds__(FSE_DEC_THREADS) void k_fse_decode(FseDecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t first = (size_t)blockIdx.x * a.G;
    const u32 slotBytes = a.slotU32 * 4u;                        // multiple of 8
    const u32 ringOff = 2u << a.maxTableLog;                     // state ring offset inside a slot
    const u32 inOff = ringOff + FSE_DEC_RING * 8;                // input ring offset inside a slot
    DecCtl* const ctlAll = (DecCtl*)lds;
    u8* const ldsb = (u8*)lds + FSE_CTL_BYTES;

    // ---- stage: reference cells {u16 newState; u8 symbol; u8 nbBits} -> compact u16 (uniform control flow, both waves).
    //      A table whose fields do not fit 12+4 bits (cannot come from FSE_buildDTable) is flagged and decoded
    //      by the literal path only.
    u32* const flagsSh = ctlAll[0].pad;                          // [0] bad-table mask, [1] any nbBits == 0
    if (tid < 2) flagsSh[tid] = 0;
    __syncthreads();
    {   u32 badBits = 0; bool anyNb0 = false;
        for (int g = 0; g < a.G; ++g) {
            const size_t b = first + g;
            if (b >= a.nBlocks) break;
            if (a.meta && a.meta[b].state == 0) continue;
            const u32* t = a.dtables + b * a.dtStrideU32;
            const u32 tl = t[0] & 0xFFFFu;
            if (tl > a.maxTableLog) continue;
            const u32 ts = 1u << tl;
            u16* A = (u16*)(ldsb + (size_t)g * slotBytes);
            bool bad = false;
            for (u32 i = tid; i < ts; i += FSE_DEC_THREADS) {
                const u32 c = t[1 + i];
                const u32 ns = c & 0xFFFFu, nb = c >> 24;
                bad |= (ns > 0xFFFu) | (nb > 15u) | ((ns & ((1u << (nb & 15u)) - 1u)) != 0);
                anyNb0 |= (nb == 0);
                A[i] = (u16)((ns & 0xFFFu) | (nb << 12));
            }
            if (bad) badBits |= 1u << g;
        }
        if (badBits) atomicOr(&flagsSh[0], badBits);
        if (anyNb0) atomicOr(&flagsSh[1], 1u);
    }
    __syncthreads();
    const u32 badMask = flagsSh[0];
    const bool nb0 = flagsSh[1] != 0;

    // ---- per-block set-up by the decoder wave (lane g owns block first+g)
    const size_t b = first + (size_t)lane;
    bool owner = wave == 0 && lane < a.G && b < a.nBlocks;
    u32 hdr = 0;
    if (owner && a.meta) { if (a.meta[b].state == 0) owner = false; else hdr = a.meta[b].hdrSize; }
    u32 tl = 0; bool fast = false;
    const u32* const gtab = a.dtables + (owner ? b : 0) * a.dtStrideU32;   // reference-layout table in global memory
    if (owner) {
        const u32 h0 = gtab[0];
        tl = h0 & 0xFFFFu; fast = (h0 >> 16) != 0;
        if (tl > a.maxTableLog) { a.results[b] = FERR(tableLog_tooLarge); owner = false; }
    }
    const u32* const cells = gtab + 1;                          // literal path reads the reference cells
    const u16* const A = (const u16*)(ldsb + (size_t)(lane < a.G ? lane : 0) * slotBytes);
    const u8* in = nullptr; size_t S = 0; u8* out = nullptr;
    const long omax = (long)a.dstCapacity;
    long op = 0;
    BitReader r; r.base = nullptr; r.size = 0; r.at = 0; r.win = 0; r.used = 0;
    u32 s1 = 0, s2 = 0;
    if (owner) {
        in = view_ptr(a.csrc, b) + hdr;
        S = view_size(a.csrc, b) - hdr;              // hdr <= cSrcSize (FSE_readNCount never returns more)
        out = a.dst + b * a.dstStride;
        const size_t e = r.init(in, S);
        if (is_err(e)) { a.results[b] = e; owner = false; }
        else {
            s1 = r.read(tl); r.reload();             // FSE_initDState x2, fse.h:577-584
            s2 = r.read(tl); r.reload();
        }
    }

    // ---- bulk: iterations of fse_decompress.c:201-218 whose loop-head reload is provably the fast one.
    //      The decoder lane touches only registers and LDS:
    //        * input: the 8 bytes below the window come from this block's LDS input ring;
    //        * output: each iteration appends the 4 states it decoded FROM to the block's state ring.
    //      The service wave (above) owns all global-memory traffic of the bulk loop.
    // Phase structure: a lane runs a phase only if FSE_CHECK_EVERY more iterations are certainly valid for it
    // (>= 16 output groups left and the window stays >= 24 bytes above the stream start: at >= 24 + 16*6), so the
    // 16 iterations of a phase run without any per-iteration bookkeeping; whatever is left goes to the literal tail.
    // Bulk state uses p = at+1, u = used+8 (u in [8,16) after a reload), so no shift amount is ever 0 or 32.
    bool can = owner && r.at >= 24 + 6 * FSE_CHECK_EVERY && (omax - 3 - op + 3) / 4 >= FSE_CHECK_EVERY && S < (1ull << 31) && !((badMask >> lane) & 1u);
    BulkState bs; bs.s1 = s1; bs.s2 = s2; bs.pofs = (u32)r.at + 1u; bs.u = r.used + 8u;
    bs.whi = bs.wlo = bs.l1hi = bs.l1lo = bs.l2hi = bs.l2lo = 0;
    long groups = 0;
    u32 iters = 0;
    int validLo = 0;                                 // ring holds stream bytes [validLo, validLo + 512)
    if (can) {
        const u64 w = ldg64u(in + bs.pofs), a1 = ldg64u(in + bs.pofs - 8), a2 = ldg64u(in + bs.pofs - 16);
        bs.whi = (u32)(w >> 32); bs.wlo = (u32)w; bs.l1hi = (u32)(a1 >> 32); bs.l1lo = (u32)a1; bs.l2hi = (u32)(a2 >> 32); bs.l2lo = (u32)a2;
        groups = (omax - 3 - op + 3) >> 2;
        validLo = ((int)bs.pofs - 16 - 208) & ~255;  // slack below the lowest byte read next
    }
    DecCtl* const ctl = ctlAll + (lane < FSE_MAXG ? lane : 0);
    if (wave == 0 && lane < FSE_MAXG) {
        ctl->pubIters = 0; ctl->pubPofs = can ? bs.pofs : 0x80000000u;
        ctl->srvFlushed = 0; ctl->srvValidLo = 0x7FFFFFFF;
        ctl->initValidLo = validLo; ctl->S32 = (int)(S < (1ull << 31) ? S : 0);
        const unsigned long long ib = (unsigned long long)(uintptr_t)in, ob = (unsigned long long)(uintptr_t)out, tb = (unsigned long long)(uintptr_t)cells;
        ctl->inLo = (u32)ib; ctl->inHi = (u32)(ib >> 32); ctl->outLo = (u32)ob; ctl->outHi = (u32)(ob >> 32); ctl->symLo = (u32)tb; ctl->symHi = (u32)(tb >> 32);
    }
    __syncthreads();
    if (wave == 1) { fse_decode_service(a, ldsb, ctlAll, slotBytes, ringOff, inOff, lane); return; }

    uint2* const myRing = (uint2*)(ldsb + (size_t)(lane < a.G ? lane : 0) * slotBytes + ringOff);
    const u8* const myIn = ldsb + (size_t)(lane < a.G ? lane : 0) * slotBytes + inOff;
    while (__any(can)) {
        bool ready = false;
        if (can) {
            const u32 fl = ctl_load(&ctl->srvFlushed);
            const int vlo = ctl_load(&ctl->srvValidLo);
            // room for 16 more records, and the lowest byte this phase can read (p - 6*16 - 16) is in the ring
            ready = (iters + FSE_CHECK_EVERY - fl <= FSE_DEC_RING) && ((int)bs.pofs - 6 * FSE_CHECK_EVERY - 16 >= vlo);
        }
        if (ready) {
            if (nb0) fse_bulk_phase<true>(bs, A, myIn, myRing, iters);
            else     fse_bulk_phase<false>(bs, A, myIn, myRing, iters);
            iters += FSE_CHECK_EVERY; groups -= FSE_CHECK_EVERY;
            can = bs.pofs - 1u >= 24u + 6u * FSE_CHECK_EVERY && groups >= FSE_CHECK_EVERY;
            ctl_store(&ctl->pubIters, iters);
            ctl_store(&ctl->pubPofs, can ? bs.pofs : (bs.pofs | 0x80000000u));
        }
        if (!__any(ready)) __builtin_amdgcn_s_sleep(2);
    }
    if (!owner) return;
    op = 4 * (long)iters;
    if (iters) { r.at = (size_t)bs.pofs - 1; r.used = bs.u - 8u; r.win = ldg64u(in + r.at); s1 = bs.s1; s2 = bs.s2; }   // back to the reference's (at, used, window)

    // ---- literal tail: remaining iterations of :201-218, then :222-235 (cells read from the reference table)
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

static void fse_decode_geometry(unsigned maxTableLog, size_t ldsBytes, unsigned* slotU32, int* G)
{
    *slotU32 = ((2u << maxTableLog) + FSE_DEC_RING * 8 + FSE_IN_RING + FSE_IN_MIRROR + 8) / 4;   // table + rings (+8: rotating banks)
    int g = (int)((ldsBytes - FSE_CTL_BYTES) / (*slotU32 * 4));
    if (g > FSE_MAXG) g = FSE_MAXG;
    if (const char* dbg = getenv("FSEHIP_DEBUG_G")) { int v = atoi(dbg); if (v >= 1 && v < g) g = v; }   // tuning aid
    *G = g;
}
#define FSE_DEC_LDS (80 * 1024)   // two workgroups per CU (measured: 2 x 80 KiB are co-resident on gfx950)
size_t fse_decode_blocks_per_round(unsigned maxTableLog)
{
    unsigned slot; int G;
    fse_decode_geometry(maxTableLog, FSE_DEC_LDS, &slot, &G);
    const int cus = dev_props().ok ? dev_props().cus : 256;
    return (size_t)G * 2 * cus;
}

hipError_t launch_fse_decode(FseDecArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    static bool attrSet = false;
    size_t ldsBytes = FSE_DEC_LDS;
    if (const char* dbg = getenv("FSEHIP_DEBUG_LDS")) ldsBytes = (size_t)atoi(dbg);   // tuning aid
    if (!attrSet) {
        hipError_t e = hipFuncSetAttribute((const void*)k_fse_decode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        attrSet = true;
    }
    fse_decode_geometry(a.maxTableLog, ldsBytes, &a.slotU32, &a.G);
    if (a.G < 1) return hipErrorInvalidValue;
    const size_t groups = (a.nBlocks + a.G - 1) / a.G;
    probe_before(PK_FSE_DECODE, s);
    hipLaunchKernelGGL(k_fse_decode, dim3((unsigned)groups), dim3(FSE_DEC_THREADS), ldsBytes, s, a);
    probe_after(PK_FSE_DECODE, s);
    return hipGetLastError();
}
