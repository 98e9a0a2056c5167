// fse_encode_wave.hip -- a2: FSE_compress_usingCTable, 32 lanes per block (two blocks per wave)
// (reference: lib/fse_compress.c:554-623, lib/fse.h:503-527, lib/bitstream.h:183-260; format SURVEY A.1/A.3).
//
// tANS encoding is a loop-carried chain (two interleaved chains per block), but the encoder state is only
// tableLog bits wide and every step replaces part of it by a function of the symbol alone, so a chain "forgets"
// where it came from.  That makes the block splittable without changing a single output bit:
//
//   WV_LANES (32) lanes per block, two blocks per 64-lane wave; only the packed CTable lives in LDS, so many blocks are
//   resident per CU.  In emission order (last source byte first) the block is cut into 32 contiguous ranges of symbols.
//   Pass 1 (counting, ONE CHAIN PER LANE: the two chains never interact): lane 2k + c walks chain c over the super-range
//   k = ranges 2k and 2k+1.  It warms the chain up over `warm` symbols in front of the super-range starting from an
//   arbitrary state, remembers the state it arrives with (its speculated start), then runs the super-range counting
//   bits, noting state and count where the two ranges meet, and remembers the state it ends with.  Verification: a
//   lane's speculated start must equal the end of the same chain's previous super-range; super-range 0 starts from the
//   exact FSE_initCState2 state, so if every link matches, every start is exact by induction.  A lane whose link does not
//   match re-runs from its predecessor's end -- only as far as the first checkpoint at which it has merged with the
//   trajectory its previous run recorded -- and the check is repeated until no link changes (worst case this
//   degenerates into the serial algorithm).  Nothing is assumed: the output is bit-exact by construction.
//   (Half as many links per chain as ranges, each twice as long; a link fails when one chain has not merged, not either
//   of two; a repair re-runs one chain: measured against two-chain lanes over single ranges, P80 4.5 repair rounds
//   instead of 9.4 and 8.2 ms instead of 11.0 per 100k blocks, P14 1.0 instead of 1.4 rounds.)
//   Pass 2: every range now knows both start states and its bit count; a wave prefix sum of the counts gives every lane
//   its bit offset (and the exact compressed size / the BIT_closeCStream verdict before a single bit is written); lane t
//   runs range t, both chains interleaved, from the verified states and emits its bits through a small per-lane LDS
//   ring, written out in aligned 32-byte pieces, the last whole bytes at the end.  The byte shared by two neighbouring
//   ranges is stored by the upper lane with the lower lane's bits OR-ed in afterwards (one global atomic per lane).
//   CState2, CState1 and the end mark are appended by the lane that owns the last range (lib/fse_compress.c:608-610).
//   Source bytes are streamed per lane in 64-byte aligned segments (four 16-byte loads), one segment ahead.
#include "internal.h"

// Warm-up symbols (half per chain) in front of every range.  Two states fed the same symbols merge with probability
// ~ present/tableSize per step (sum_s p_s / norm_s), so the warm-up is sized as a multiple of tableSize/present.
#ifndef FSE_WV_WARM_FACTOR
#define FSE_WV_WARM_FACTOR 2u
#endif
#define FSE_WV_WARM_MIN 64u
#define FSE_WV_WARM_MAX 4096u

// LDS is addressed by absolute byte addresses (u32) through address_space(3) pointers, so that address arithmetic is
// plain 32-bit VALU work.  The kernel is bound by the CU's LDS pipe (two table reads per symbol for every lane), so the
// per-symbol entry is packed into 4 bytes whenever the state fits 12 bits (maxTableLog <= 11):
//   TT4 entry = d16 | fs16 << 16, with deltaNbBits = (maxBitsOut << 16) - minStatePlus (fse_compress.c:131-154):
//     d16  = (maxBitsOut << 12) - minStatePlus   ->  nbBits = (state + d16) >> 12     (state, minStatePlus <= 4096)
//     fs16 = deltaFindState + (absolute LDS address of stateTable[0]) / 2, signed     ->  next = lds16[((state >> nbBits) + fs16) << 1]
//   TT8 entry (maxTableLog 12) = {2*deltaFindState + address of the stateTable, deltaNbBits}: nbBits = (state + deltaNbBits) >> 16.
typedef const __attribute__((address_space(3))) u16* wv_lds_u16;
typedef const __attribute__((address_space(3))) u32* wv_lds_u32;
typedef u32 wv_u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) wv_u32x2* wv_lds_u64;
DEV u32 wv_lds_addr(const void* p) { return (u32)(uintptr_t)(const __attribute__((address_space(3))) u8*)p; }
template <bool TT4>
DEV void wv_step(u32& st, u32 ttb, u32 sym, u32& nb)                    // FSE_encodeSymbol (fse.h:514-521) without the bit output
{
    if (TT4) {
        const u32 e = *(wv_lds_u32)(uintptr_t)(ttb + 4u * sym);
        nb = (st + (e & 0xFFFFu)) >> 12;
        st = *(wv_lds_u16)(uintptr_t)(((st >> nb) + (u32)((int)e >> 16)) << 1);
    } else {
        const wv_u32x2 e = *(wv_lds_u64)(uintptr_t)(ttb + 8u * sym);
        nb = (st + e.y) >> 16;
        st = *(wv_lds_u16)(uintptr_t)(((st >> nb) << 1) + e.x);
    }
}
template <bool TT4>
DEV void wv_step_bits(u32& st, u32 ttb, u32 sym, u32& nb, u32& bits)
{
    const u32 before = st;
    wv_step<TT4>(st, ttb, sym, nb);
    bits = __builtin_amdgcn_ubfe(before, 0u, nb);
}
#define WV_STEP(ST, sym, nb) wv_step<TT4>(ST, ttb, sym, nb);
#define WV_STEP_BITS(ST, sym, nb, bits) wv_step_bits<TT4>(ST, ttb, sym, nb, bits);

DEV u32 wv_byte0(u32 w) { u32 r; __asm__("v_and_b32 %0, 0xff, %1" : "=v"(r) : "v"(w)); return r; }   // (opaque: or the shift of the address is pulled in front of the mask, one instruction more)
DEV uint4 wv_load16(const u8* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }

// Bit sink of one lane.  Bits are appended LSB-first; completed 32-bit words go to a small per-lane LDS ring that is
// indexed by the low bits of the word's global address, and every time a WV_LINE-byte aligned piece of the output is
// complete it is written to global memory with 16-byte stores (whole 32-byte sectors; measured, 64-byte lines still leave
// L2 in two halves often enough that 31 KB are written for 17 KB of payload).  The first bytes of a lane (up to the first line boundary) and its last ones go out bytewise / wordwise.
#ifndef WV_LINE
#define WV_LINE 32u
#endif
#define WV_RING (2u * WV_LINE)
struct WvSink {
    u8* dstAl;            // destination rounded down to WV_RING bytes
    u32* ring;            // this lane's LDS ring
    u32 ringA;            // its absolute LDS byte address
    u32 woff;             // offset from dstAl of the next 32-bit word (multiple of 4)
    u32 done;             // offset from dstAl up to which this lane's bytes are in global memory
    u32 acc, nacc;        // the bits not yet in a complete word (nacc < 32 between calls)
    // append nb (<= 32) bits.  Branch-free: the word in progress is stored to the ring every time (it is overwritten until
    // complete); a conditional flush costs more instructions than the store, and its branch is taken by some lane of the
    // wave at nearly every call anyway (measured on P14: 618 GB/s against 604 with the store under `if (full)`, 584 with the
    // branchy 64-bit accumulator)
    DEV void push(u32 v, u32 nb)
    {
        const u64 sh = (u64)v << nacc;
        acc |= (u32)sh;
        nacc += nb;
        const bool full = nacc >= 32u;
        *(__attribute__((address_space(3))) u32*)(uintptr_t)(ringA | (woff & (WV_RING - 4u))) = acc;   // (the rings are WV_RING-aligned)
        woff += full ? 4u : 0u;
        acc = full ? (u32)(sh >> 32) : acc;
        nacc &= 31u;
    }
    DEV void copy_out(u32 upTo)                                         // bytes [done, upTo), any alignment
    {
        const u8* const rb = (const u8*)ring;
        u32 o = done;
        while (o < upTo) {
            if (((o & 3u) == 0) && o + 4 <= upTo) { const u32 w = ring[(o & (WV_RING - 1)) >> 2]; __builtin_memcpy(dstAl + o, &w, 4); o += 4; }
            else { dstAl[o] = rb[o & (WV_RING - 1)]; ++o; }
        }
        done = upTo;
    }
    DEV void line()                                                     // once per 16 symbols (<= 24 bytes): at most one line completes
    {
        const u32 L = done & ~(WV_LINE - 1);
        if (woff >= L + WV_LINE) {
            if (done == L) {
                const uint4* const r4 = (const uint4*)(ring + ((L & (WV_RING - 1)) >> 2));
#pragma unroll
                for (u32 q = 0; q < WV_LINE / 16; ++q) { const uint4 v = r4[q]; __builtin_memcpy(dstAl + L + 16 * q, &v, 16); }
                done = L + WV_LINE;
            } else copy_out(L + WV_LINE);
        }
    }
};

// EMIT symbols j in [ja, jb) (distance from the block end; even j -> chain A, odd j -> chain B; ja is even) from the verified
// states (xa, xb) into the sink.  The source is streamed downwards: 64 bytes (four 16-byte loads of one 64-byte segment, so
// the segment is fetched from memory once) one segment ahead of its use.
template <bool TT4>
DEV void wv_emit(u32 ttb, const u8* src, u32 n, u32 ja, u32 jb, u32& xa, u32& xb, WvSink& k)
{
    u32 j = ja;
    u32 na, nbb, ba, bb;
#define WV_PAIR(w, hiA, hiB)                                                                         \
    {   const u32 sa = __builtin_amdgcn_ubfe(w, hiA, 8u), sb = hiB ? __builtin_amdgcn_ubfe(w, hiB, 8u) : wv_byte0(w); \
        WV_STEP_BITS(xa, sa, na, ba) WV_STEP_BITS(xb, sb, nbb, bb) k.push(ba | (bb << na), na + nbb); }
#define WV_QUAD(v)                                                                                   \
    WV_PAIR(v.w, 24u, 16u) WV_PAIR(v.w, 8u, 0u) WV_PAIR(v.z, 24u, 16u) WV_PAIR(v.z, 8u, 0u)           \
    WV_PAIR(v.y, 24u, 16u) WV_PAIR(v.y, 8u, 0u) WV_PAIR(v.x, 24u, 16u) WV_PAIR(v.x, 8u, 0u)          \
    k.line();
    if (j + 64 <= jb) {
        const u8* p = src + (n - 64 - j);
        uint4 c0 = wv_load16(p), c1 = wv_load16(p + 16), c2 = wv_load16(p + 32), c3 = wv_load16(p + 48);
        while (j + 64 <= jb) {
            const u32 nj = j + 64;
            const u8* const q = src + (n - 64 - (nj + 64 <= jb ? nj : j));
            const uint4 n0 = wv_load16(q), n1 = wv_load16(q + 16), n2 = wv_load16(q + 32), n3 = wv_load16(q + 48);
            __asm__ volatile("" ::: "memory");                          // keep the prefetch up here: one segment ahead of its use
            WV_QUAD(c3) WV_QUAD(c2) WV_QUAD(c1) WV_QUAD(c0)
            c0 = n0; c1 = n1; c2 = n2; c3 = n3; j = nj;
        }
    }
    while (j + 16 <= jb) {
        const uint4 c = wv_load16(src + (n - 16 - j));
        WV_QUAD(c)
        j += 16;
    }
#undef WV_QUAD
#undef WV_PAIR
    for (; j < jb; ++j) {
        const u32 sym = src[n - 1 - j];
        u32 nb, b1;
        if (j & 1u) { WV_STEP_BITS(xb, sym, nb, b1) } else { WV_STEP_BITS(xa, sym, nb, b1) }
        k.push(b1, nb);
    }
}

// COUNT one chain over [ja, jb) (ja even; c = 0: the even j, 1: the odd j): its bits are added to `bits`, which -- like the
// checkpoint bookkeeping in ck -- runs on over the caller's pieces.  Same source streaming as wv_emit; a lane uses every
// other byte.  Checkpoints: after every ck->every 64-symbol groups the pass records its state and bit count
// (WV_CK_RECORD), or -- when the chain is re-run from a corrected start state (WV_CK_MERGE) -- compares them with what
// the previous run recorded there: once the states agree the rest repeats the previous run (same state, same symbols),
// so every later count is the recorded one shifted by the difference, and the pass stops.
enum { WV_CK_NONE = 0, WV_CK_RECORD = 1, WV_CK_MERGE = 2 };
#define WV_CK_MAX 8u
struct WvCk { uint2* slot; u32 every, left, idx, total; u32 d; bool merged; };
template <bool TT4, int CK>
DEV void wv_chain(u32 ttb, const u8* src, u32 n, u32 ja, u32 jb, u32 c, u32& x, u32& bits, WvCk* ck = nullptr)
{
    u32 j = ja;
    const u32 shHi = 24u - 8u * c, shLo = 8u - 8u * c;                      // my byte of a word's upper / lower half (even j = the higher address)
#define WV_CSTEP(w, sh) { const u32 sy = __builtin_amdgcn_ubfe(w, sh, 8u); u32 nb; WV_STEP(x, sy, nb) bits += nb; }
#define WV_CQUAD(v) WV_CSTEP(v.w, shHi) WV_CSTEP(v.w, shLo) WV_CSTEP(v.z, shHi) WV_CSTEP(v.z, shLo) WV_CSTEP(v.y, shHi) WV_CSTEP(v.y, shLo) WV_CSTEP(v.x, shHi) WV_CSTEP(v.x, shLo)
    if (j + 64 <= jb) {
        const u8* p = src + (n - 64 - j);
        uint4 c0 = wv_load16(p), c1 = wv_load16(p + 16), c2 = wv_load16(p + 32), c3 = wv_load16(p + 48);
        while (j + 64 <= jb) {
            const u32 nj = j + 64;
            const u8* const q = src + (n - 64 - (nj + 64 <= jb ? nj : j));
            const uint4 n0 = wv_load16(q), n1 = wv_load16(q + 16), n2 = wv_load16(q + 32), n3 = wv_load16(q + 48);
            __asm__ volatile("" ::: "memory");                          // keep the prefetch up here: one segment ahead of its use
            WV_CQUAD(c3) WV_CQUAD(c2) WV_CQUAD(c1) WV_CQUAD(c0)
            c0 = n0; c1 = n1; c2 = n2; c3 = n3; j = nj;
            if (CK != WV_CK_NONE && --ck->left == 0) {
                ck->left = ck->every;
                if (CK == WV_CK_MERGE && ck->idx < WV_CK_MAX) {
                    const uint2 o = ck->slot[ck->idx];
                    if (o.x == x) {
                        // merged with the recorded trajectory: everything behind this point repeats it, its bit counts shifted by d
                        const u32 d = bits - o.y;
                        for (u32 i = ck->idx; i < ck->total && i < WV_CK_MAX; ++i) ck->slot[i].y += d;
                        ck->d = d; ck->merged = true;
                        return;
                    }
                }
                if (ck->idx < WV_CK_MAX) ck->slot[ck->idx] = make_uint2(x, bits);
                ++ck->idx;
            }
        }
    }
    while (j + 16 <= jb) {
        const uint4 v = wv_load16(src + (n - 16 - j));
        WV_CQUAD(v)
        j += 16;
    }
#undef WV_CQUAD
#undef WV_CSTEP
    for (j += c; j < jb; j += 2) { u32 nb; WV_STEP(x, (u32)src[n - 1 - j], nb) bits += nb; }
}

template <bool TT4>
DEV u32 wv_init_state(u32 ttb, u32 sym)                  // FSE_initCState2, lib/fse.h:503-512
{
    if (TT4) {                                           // nbBitsOut = maxBitsOut, value = minStatePlus (see the entry format)
        const u32 e = *(wv_lds_u32)(uintptr_t)(ttb + 4u * sym);
        const u32 d = e & 0xFFFFu, nb = (d >> 12) + 1u, msp = (nb << 12) - d;
        return *(wv_lds_u16)(uintptr_t)(((msp >> nb) + (u32)((int)e >> 16)) << 1);
    }
    const wv_u32x2 e = *(wv_lds_u64)(uintptr_t)(ttb + 8u * sym);
    const u32 nb = (e.y + (1u << 15)) >> 16;
    return *(wv_lds_u16)(uintptr_t)(((((nb << 16) - e.y) >> nb) << 1) + e.x);
}

// the whole block by one lane, byte by byte (lib/fse_compress.c:554-623 as written): used when a lane's share of the
// output is shorter than one byte, which the word-wise writer above does not handle
template <bool TT4>
DEV size_t wv_serial(u32 ttb, const u8* src, u32 n, u8* dst, size_t cap, u32 tl)
{
    const u32 lim = (u32)(cap - 8);
    u64 acc = 0; u32 nacc = 0, pos = 0;
    u32 xa = wv_init_state<TT4>(ttb, src[n - 1]), xb = wv_init_state<TT4>(ttb, src[n - 2]);
    for (u32 j = 2; j < n; ++j) {
        const u32 sym = src[n - 1 - j];
        u32 nb, bits;
        if (j & 1u) { WV_STEP_BITS(xb, sym, nb, bits) } else { WV_STEP_BITS(xa, sym, nb, bits) }
        acc |= (u64)bits << nacc; nacc += nb;
        if (nacc >= 32u || j + 1 == n) {                                   // BIT_flushBits, bitstream.h:239-249
            __builtin_memcpy(dst + pos, &acc, 8);
            const u32 nby = nacc >> 3; pos += nby; pos = pos > lim ? lim : pos; acc >>= (nby << 3); nacc &= 7u;
        }
    }
    const u32 c2 = (n & 1u) ? xb : xa, c1 = (n & 1u) ? xa : xb, mask = (1u << tl) - 1u;
#define WV_FLUSH() { __builtin_memcpy(dst + pos, &acc, 8); const u32 nby = nacc >> 3; pos += nby; pos = pos > lim ? lim : pos; acc >>= (nby << 3); nacc &= 7u; }
    acc |= (u64)(c2 & mask) << nacc; nacc += tl; WV_FLUSH()
    acc |= (u64)(c1 & mask) << nacc; nacc += tl; WV_FLUSH()
    acc |= (u64)1 << nacc; nacc += 1; WV_FLUSH()
#undef WV_FLUSH
    return (pos >= lim) ? 0 : (size_t)pos + (nacc > 0);
}

#ifdef FSE_ENC_TIMING       // development aid: per-block cycle accounting
__device__ unsigned long long g_encTiming[4096 * 8];
extern "C" __attribute__((visibility("default"))) int FSEHIP_debug_encTiming(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_encTiming), sizeof(g_encTiming)); }
#define ETIMING(x) x
#else
#define ETIMING(x)
#endif

#define FSE_WV_WAVES 1               // waves per workgroup; the waves never synchronise with each other
// Lanes per block.  A range must be long compared with the time two encoders need to merge (P14: ~90 symbols per chain,
// so with 64 ranges of 512 symbols over half of the speculated starts fail and the repair rounds cost as much as the
// counting pass); 32 lanes per block = two blocks per wave, ranges twice as long, half as many links to verify.
// (Measured again in round 4 with the per-chain counting pass, -DWV_LANES=64u: twice the waves per CU, but 4.07 instead of 3.89 ms per
// 100k P14 blocks.)
#ifndef WV_LANES
#define WV_LANES 32u
#endif
#define WV_BPW (64u / WV_LANES)      // blocks per wave
template <bool TT4>
__global__ __launch_bounds__(64 * FSE_WV_WAVES) void k_fse_encode_wave(FseEncArgs a, u32 slotWords, u32 tableWords)
{
    ETIMING(unsigned long long T0 = __builtin_readcyclecounter(); unsigned long long T1 = 0; unsigned long long T2 = 0; unsigned long long T3 = 0; unsigned long long T4 = 0; u32 rounds = 0; u32 nBad0 = 0; u32 firstBad = 99;)
    extern __shared__ __attribute__((aligned(16))) u32 ldsAll[];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32 part = lane / WV_LANES, hl = lane % WV_LANES;                 // my block of this wave, my lane within it
    const u32 partBase = lane - hl;
    const u32 slot = wv * WV_BPW + part;
    u32* const lds = ldsAll + slot * slotWords;
    const u32 ldsOff = wv_lds_addr(ldsAll) + slot * slotWords * 4u;         // absolute LDS byte address of this block's slot
    const size_t pos = ((size_t)blockIdx.x * FSE_WV_WAVES + wv) * WV_BPW + part;

    // ---- per-block set-up; `on` = this block is (still) being encoded by its lanes.  Everything below is uniform per block,
    //      the blocks of a wave diverge freely; wave-wide shuffles are only read inside a block's own lanes.
    //      One-shot path: entry `pos` of the pace-bin lists taken one after the other (internal.h), so that the two blocks of a wave are
    //      of a kind and the wave does not wait for a slowly mixing block next to a fast one.
    size_t b = pos;
    bool on = pos < a.nBlocks;
    if (a.list) {
        size_t q = pos; int i = 0;
        for (; i < FSE_EBINS; ++i) { const u32 c = a.count[i]; if (q < c) break; q -= c; }
        on = i < FSE_EBINS;
        b = on ? (size_t)a.list[(size_t)i * a.nBlocks + q] : 0;
    }
    u32 hdr = 0;
    if (on && a.meta) { if (fse_enc_skip(a.meta[b].state, a.onlyState)) on = false; else hdr = a.meta[b].hdrSize; }
    const u32* const gct = a.ctables + (on ? b : 0) * a.ctStrideU32;
    u32 tl = 0, msv = 0;
    if (on) {
        const u32 h0 = gct[0];
        tl = h0 & 0xFFFFu; msv = (h0 >> 16) > 255u ? 255u : (h0 >> 16);      // symbols are bytes: larger entries are unreachable (raw tables, fse_compress.c:498-528)
        if (tl > a.maxTableLog) { if (hl == 0) a.results[b] = FERR(tableLog_tooLarge); on = false; }
    }
    const u8* const src = on ? view_ptr(a.src, b) : nullptr;
    const size_t n64 = on ? view_size(a.src, b) : 0;
    u8* const dst = a.dst + (on ? b : 0) * a.dstStride + hdr;
    const size_t cap = a.dstCapacity - hdr;
    if (on && a.sizeSplit && n64 < FSE_ENC_WAVE_MIN) on = false;              // the lane-per-block kernel's block (internal.h)
    if (on && n64 >= ((size_t)1 << 31)) { if (hl == 0) a.results[b] = FERR(srcSize_wrong); on = false; }
    const u32 n = (u32)n64;
    if (on && (n <= 2 || cap <= 8)) { if (hl == 0) a.results[b] = 0; on = false; }    // fse_compress.c:566-568
    // tableLog 0 = the fake table of FSE_buildCTable_rle (fse_compress.c:531-551): its one symbol costs no bits and the state
    // stays 0, so the stream is the end mark alone (two 0-bit states, then BIT_closeCStream: one byte).  The packed entry
    // format below cannot express minStatePlus == 0, so the block is finished here.
    if (on && tl == 0) {
        if (hl == 0) { dst[0] = 1; a.results[b] = a.meta ? (((size_t)hdr + 1 < n64 - 1) ? (size_t)hdr + 1 : 0) : 1; }
        on = false;
    }

    // ---- stage the CTable (coalesced): word 0 header, stateTable at byte 4, symbolTT rebased to LDS byte addresses
    const u32 ttStart = 1 + (tl ? (1u << (tl - 1)) : 1u);
    const u32 ttAl = (ttStart + 1u) & ~1u;                                  // 8-byte aligned symbolTT copy
    u32 presentLane = 0;
    if (on) {
        for (u32 i = hl; i < ttStart; i += WV_LANES) lds[i] = gct[i];
        for (u32 sy = hl; sy <= msv; sy += WV_LANES) {
            const u32 dfs = gct[ttStart + 2 * sy], dnb = gct[ttStart + 2 * sy + 1];
            presentLane += dnb != ((tl + 1) << 16) - (1u << tl);             // deltaNbBits of a symbol that does not occur (fse_compress.c:143)
            if (TT4) {
                const u32 mbo = (dnb >> 16) + 1u, msp = (mbo << 16) - dnb;    // 1 <= minStatePlus <= 2 * tableSize
                lds[ttAl + sy] = (((mbo << 12) - msp) & 0xFFFFu) | ((dfs + 2u + (ldsOff >> 1)) << 16);
            } else {
                lds[ttAl + 2 * sy] = 2u * dfs + 4u + ldsOff;
                lds[ttAl + 2 * sy + 1] = dnb;
            }
        }
    }
    u32 present = presentLane;
#pragma unroll
    for (int off = (int)WV_LANES / 2; off > 0; off >>= 1) present += (u32)__shfl_xor((int)present, off, WAVE);
    u32 warm = (FSE_WV_WARM_FACTOR << tl) / (present ? present : 1u);
    warm = (warm + 63u) & ~63u;
    warm = warm < FSE_WV_WARM_MIN ? FSE_WV_WARM_MIN : (warm > FSE_WV_WARM_MAX ? FSE_WV_WARM_MAX : warm);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");               // the slots are private to this wave: LDS is in order per wave
    u32 ttb = ldsOff + 4u * ttAl;                                          // absolute LDS address of the symbolTT copy
    __asm__("" : "+v"(ttb));                                               // (opaque: entry address = symbol * 4 + ttb in one v_lshl_add, not (symbol + ttb / 4) * 4 in two)
    u32* const ringBase = lds + tableWords;                                // WV_LANES output rings behind the table
    ETIMING(T1 = __builtin_readcyclecounter();)

    // ---- ranges in emission order: symbols j = 2 .. n-1 (j = 0, 1 only initialise the chains).  Lane t of the block owns
    //      [bound(t), bound(t+1)); the boundaries are even (every range starts on chain A) and, for blocks of >= 4 KiB,
    //      shifted so that the ranges above lane 0's end on 64-byte aligned source addresses (whole segments per load group)
    const u32 m = on ? n - 2 : 0u;
    u32 C = (m + WV_LANES - 1u) / WV_LANES;
    C = m >= 4096u ? (C + 63u) & ~63u : (C + 1u) & ~1u;
    C = C ? C : 2u;
    const u32 delta = m >= 4096u ? (u32)((0 - ((uintptr_t)src + n - 2u)) & 62u) : 0u;   // < 64 <= C, even
    const u32 lo0 = hl ? 2 + hl * C - delta : 2u, hi0 = 2 + (hl + 1) * C - delta;
    const u32 j0 = lo0 < n ? lo0 : n;
    const u32 j1 = (hl == WV_LANES - 1u || hi0 > n) ? n : hi0;
    const bool mine = on && j0 < n;                                         // non-empty range
    const u32 lastLane = m ? ((m + delta - 1) / C < WV_LANES - 1u ? (m + delta - 1) / C : WV_LANES - 1u) : 0u;   // owner of the final states

    // ---- pass 1: COUNTING, one chain per lane.  The two chains of a block never interact, so the block's lanes are paired up:
    //      lane 2k + c walks chain c alone over the super-range k = ranges 2k and 2k+1 (bound(2k) .. bound(2k+2)): half as many
    //      links per chain as ranges, each twice as long as a range -- a link fails when ONE chain has not merged (not either of
    //      two), a repair re-runs one chain (not both), and a wrong end state needs two ranges' worth of symbols to survive into
    //      the next link: what decides the repair rounds of slowly mixing tables.  The state and bit count at bound(2k+1) are
    //      recorded on the way, so afterwards every range knows both start states and its bit count.
    //      The output rings are idle until pass 2: they hold the checkpoints.
    const u32 cc = hl & 1u, kk = hl >> 1;
    const u32 sLo0 = kk ? 2 + 2u * kk * C - delta : 2u, sMid0 = 2 + (2u * kk + 1u) * C - delta, sHi0 = 2 + (2u * kk + 2u) * C - delta;
    const u32 sLo = sLo0 < n ? sLo0 : n, sMid = sMid0 < n ? sMid0 : n;
    const u32 sHi = (2u * kk + 2u >= WV_LANES || sHi0 > n) ? n : sHi0;
    const bool mineC = on && sLo < n;
    WvCk ck; ck.slot = (uint2*)(ringBase + hl * (WV_RING / 4)); ck.merged = false; ck.d = 0; ck.idx = 0;
    ck.every = (2u * C / 64u + WV_CK_MAX - 1u) / WV_CK_MAX; ck.every = ck.every ? ck.every : 1u;   // at most WV_CK_MAX checkpoints per super-range
    if (C & 63u) ck.every = 0x7FFFFFFFu;                                     // (small blocks: the ranges are not whole segments and too short to bother)
    ck.left = ck.every; ck.total = (2u * C / 64u) / ck.every;
    u32 cstart = 0, cmid = 0, cend = 0, bitsLo = 0, bitsTot = 0;
    if (mineC) {
        u32 x, scratch = 0;
        if (sLo <= 2 + warm) {                                              // the warm-up would reach the block end: be exact
            x = wv_init_state<TT4>(ttb, src[n - 1 - cc]);
            wv_chain<TT4, WV_CK_NONE>(ttb, src, n, 2, sLo, cc, x, scratch);
        } else {
            x = 1u << tl;                                                   // any state will do (measured: the choice does not matter): it is verified below
            wv_chain<TT4, WV_CK_NONE>(ttb, src, n, sLo - warm, sLo, cc, x, scratch);
        }
        cstart = x;
        u32 lo = sLo, hi = sMid, nbits = 0;
#pragma nounroll
        for (int piece = 0; piece < 2; ++piece) {
            wv_chain<TT4, WV_CK_RECORD>(ttb, src, n, lo, hi, cc, x, nbits, &ck);
            if (piece == 0) { cmid = x; bitsLo = nbits; }
            lo = sMid; hi = sHi;
        }
        cend = x; bitsTot = nbits;
    }
    ETIMING(T2 = __builtin_readcyclecounter();)
    // ---- verification / repair per chain: start[k] must equal end[k-1]; super-range 0 (and every one that ran from the block end) is exact.
    //      A lane keeps the sample it had before its last re-run (start -> end, mid, counts: WV_SAMPLE_CACHE).  For a fixed run of symbols the
    //      map start state -> (bits, end state) is a monotone step function with a handful of steps (a composition of degree-1 circle maps;
    //      2 .. 8 distinct values per 1024-symbol super-range on Proba80, never one: scripts/sim_repair_policies.py), so while the links
    //      above a lane are still settling its predecessor's end flips between the same few states -- A, B, A again -- and the lane has
    //      already walked from A: it takes the sample back instead of re-running (free rounds of look-ups between two rounds of runs;
    //      simulated on P80: 3.9 -> 3.1 rounds per wave; P14 has one round and does not get here).
#ifndef WV_SAMPLE_CACHE
#define WV_SAMPLE_CACHE 1
#endif
    u32 oStart = 0xFFFFFFFFu, oEnd = 0, oMid = 0, oLo = 0, oTot = 0;        // the older sample (no state is 0xFFFFFFFF)
    u32 rEnd = cend, rMid = cmid, rLo = bitsLo, rTot = bitsTot;             // the run the checkpoints describe (= the lane's last run)
    for (;;) {
        u32 prevEnd; bool bad;
        for (;;) {
            prevEnd = (u32)__shfl_up((int)cend, 2, WAVE);
            bad = mineC && kk > 0 && cstart != prevEnd;
            const bool hit = WV_SAMPLE_CACHE && bad && oStart == prevEnd;
            if (!__any(hit)) break;
            if (hit) {
                u32 t;
                t = cstart; cstart = oStart; oStart = t;  t = cend; cend = oEnd; oEnd = t;  t = cmid; cmid = oMid; oMid = t;
                t = bitsLo; bitsLo = oLo; oLo = t;  t = bitsTot; bitsTot = oTot; oTot = t;
            }
        }
        if (!__any(bad)) break;
        ETIMING(if (rounds == 0) { const unsigned long long bm = __ballot(bad); nBad0 = (u32)__builtin_popcountll(bm); firstBad = (u32)__builtin_ctzll(bm); })
        if (bad) {
            oStart = cstart; oEnd = cend; oMid = cmid; oLo = bitsLo; oTot = bitsTot;
            cstart = prevEnd; cend = rEnd; cmid = rMid; bitsLo = rLo; bitsTot = rTot;   // a merge continues the run the checkpoints describe
            u32 x = cstart, lo = sLo, hi = sMid, nbits = 0;
            ck.merged = false; ck.idx = 0; ck.left = ck.every;
            bool midDone = false;
#pragma nounroll
            for (int piece = 0; piece < 2 && !ck.merged; ++piece) {
                wv_chain<TT4, WV_CK_MERGE>(ttb, src, n, lo, hi, cc, x, nbits, &ck);
                if (piece == 0 && !ck.merged) { cmid = x; bitsLo = nbits; midDone = true; }
                lo = sMid; hi = sHi;
            }
            if (ck.merged) {                                                // the rest of the super-range, and its end state, repeat the recorded run
                if (!midDone) bitsLo += ck.d;
                bitsTot += ck.d;
            } else { cend = x; bitsTot = nbits; }
            rEnd = cend; rMid = cmid; rLo = bitsLo; rTot = bitsTot;
        }
        ETIMING(++rounds;)
    }
    ETIMING(T3 = __builtin_readcyclecounter();)
    // ---- hand the results to the ranges: lane t (range t) takes both chains' states at bound(t) and its bit count
    u32 start, bits;
    {   const int la = (int)(partBase + (hl & ~1u)), lb = la + 1;
        const u32 sA = (u32)__shfl((int)cstart, la, WAVE), sB = (u32)__shfl((int)cstart, lb, WAVE);
        const u32 mA = (u32)__shfl((int)cmid, la, WAVE), mB = (u32)__shfl((int)cmid, lb, WAVE);
        const u32 lA = (u32)__shfl((int)bitsLo, la, WAVE), lB = (u32)__shfl((int)bitsLo, lb, WAVE);
        const u32 tA = (u32)__shfl((int)bitsTot, la, WAVE), tB = (u32)__shfl((int)bitsTot, lb, WAVE);
        start = cc ? (mA | (mB << 16)) : (sA | (sB << 16));
        bits = cc ? (tA - lA) + (tB - lB) : lA + lB;
    }
    u32 xa = 0, xb = 0;

    // ---- prefix sum of the bit counts over the block's lanes
    u32 incl = bits;
#pragma unroll
    for (int off = 1; off < (int)WV_LANES; off <<= 1) { const u32 o = (u32)__shfl_up((int)incl, off, WAVE); if ((int)hl >= off) incl += o; }
    const u32 excl = incl - bits;
    const u64 bodyBits = (u32)__shfl((int)incl, (int)(partBase + WV_LANES - 1u), WAVE);

    // ---- verdict (BIT_closeCStream, bitstream.h:254-260): total bits incl. the two states and the end mark
    const u64 totalBits = bodyBits + 2u * tl + 1u;
    const size_t whole = (size_t)(totalBits >> 3);
    size_t csize = (whole >= cap - 8) ? 0 : (size_t)((totalBits + 7) >> 3);
    size_t result = csize;
    if (a.meta) result = (csize != 0 && (size_t)hdr + csize < n64 - 1) ? (size_t)hdr + csize : 0;   // fse_compress.c:668-676
    if (on && result == 0) { if (hl == 0) a.results[b] = 0; on = false; }

    // ---- pass 2.  The word-wise writer needs every range (but the last) to span at least one byte of output
    {   const unsigned long long thin = __ballot(on && mine && hl < lastLane && bits < 8u);
        const unsigned long long partMask = (WV_LANES == 64u ? ~0ull : ((1ull << (WV_LANES & 63u)) - 1ull)) << partBase;
        if (on && (thin & partMask)) {
            if (a.meta && a.onlyState == FSE_ENC_PAR) {                     // the lane-per-block kernel runs right after this one
                if (hl == 0) const_cast<FseMeta*>(a.meta)[b].state = FSE_ENC_LANE;
            } else if (hl == 0) {
                const size_t cs = wv_serial<TT4>(ttb, src, n, dst, cap, tl);
                a.results[b] = a.meta ? ((cs != 0 && (size_t)hdr + cs < n64 - 1) ? (size_t)hdr + cs : 0) : cs;
            }
            on = false;
        }
    }
    u32 tail = 0;
    if (on && mine) {
        WvSink k;
        const u32 lead = (u32)((uintptr_t)dst & (WV_RING - 1));
        const u32 off0 = lead + (excl >> 3);                                // my first byte, as an offset from dstAl
        k.dstAl = dst - lead; k.ring = ringBase + hl * (WV_RING / 4); k.ringA = wv_lds_addr(k.ring);
        k.woff = off0 & ~3u; k.done = off0; k.acc = 0; k.nacc = 8u * (off0 & 3u) + (excl & 7u);
        xa = start & 0xFFFFu; xb = start >> 16;
        wv_emit<TT4>(ttb, src, n, j0, j1, xa, xb, k);
        if (hl == lastLane) {
            // fse_compress.c:608-609 : CState2 then CState1.  n even -> CState2 is the even-distance chain (:577-580), n odd -> CState1 (:572-576)
            const u32 c2 = (n & 1u) ? xb : xa, c1 = (n & 1u) ? xa : xb;      // (the owner of the last range has just arrived at the final states)
            const u32 mask = (1u << tl) - 1u;
            k.push(c2 & mask, tl);
            k.push(c1 & mask, tl);
            k.push(1u, 1u);
            k.nacc = (k.nacc + 7u) & ~7u;                                   // the last partial byte is this lane's
        }
        k.ring[(k.woff & (WV_RING - 1)) >> 2] = k.acc;                      // < 32 bits left
        k.copy_out(k.woff + (k.nacc >> 3));                                 // every complete byte
        tail = k.acc >> (k.nacc & ~7u);                                   // < 8 bits, belong to the next lane's first byte
    }
    const u32 prevTail = (u32)__shfl_up((int)tail, 1, WAVE);
    if (on && mine && hl > 0 && (excl & 7u)) {
        u8* const p = dst + (excl >> 3);
        const uintptr_t ad = (uintptr_t)p;
        atomicOr((u32*)(ad & ~(uintptr_t)3), prevTail << (8u * (u32)(ad & 3u)));
    }
    ETIMING(T4 = __builtin_readcyclecounter(); if (hl == 0 && b < 4096) { unsigned long long* t = g_encTiming + 8 * b; t[0] = T1 - T0; t[1] = T2 - T1; t[2] = T3 - T2; t[3] = T4 - T3; t[4] = rounds; t[5] = nBad0; t[6] = firstBad; })
    if (on && hl == 0) a.results[b] = result;
}

// the FSE_ENC_PAR blocks by pace bin: 64 blocks per wave, one atomic per bin and wave
__global__ __launch_bounds__(64) void k_fse_enc_lists(FseEncArgs a)
{
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    const u32 lane = threadIdx.x;
    int bin = -1;
    if (b < a.nBlocks && a.meta[b].state == FSE_ENC_PAR) { const u32 p = a.meta[b].pace; bin = (int)(p < FSE_EBINS ? p : FSE_EBINS - 1); }
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int c = 0; c < FSE_EBINS; ++c) {
        const unsigned long long mask = __ballot(bin == c);
        if (!mask) continue;                                               // uniform
        const int leader = __builtin_ctzll(mask);
        u32 base = 0;
        if ((int)lane == leader) base = atomicAdd(&a.count[c], (u32)__builtin_popcountll(mask));
        base = (u32)__shfl((int)base, leader, WAVE);
        if (bin == c) a.list[(size_t)c * a.nBlocks + base + (u32)__builtin_popcountll(mask & below)] = (u32)b;
    }
}
hipError_t launch_fse_enc_lists(const FseEncArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    const hipError_t e = launch_zero_u32(a.count, FSE_EBINS, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fse_enc_lists, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fse_encode_wave(FseEncArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    const bool tt4 = a.maxTableLog <= 11;                                   // states fit 12 bits: 4-byte symbolTT entries
    const u32 tableWords = (2 + (1u << (a.maxTableLog - 1)) + (tt4 ? 256 : 512) + 2 + 31) & ~31u;    // rings start 128-byte aligned
#ifndef WV_EXTRA_LDS_WORDS
#define WV_EXTRA_LDS_WORDS 0u       // A/B aid: unused LDS per block, i.e. fewer waves per CU (how much does the kernel live on residency?)
#endif
    const u32 slotWords = tableWords + WV_LANES * (WV_RING / 4) + WV_EXTRA_LDS_WORDS;
    const size_t ldsBytes = 4 * (size_t)slotWords * WV_BPW * FSE_WV_WAVES;
    const size_t perGroup = (size_t)WV_BPW * FSE_WV_WAVES;
    probe_before(PK_FSE_ENCODE_WAVE, s);
    const dim3 grid((unsigned)((a.nBlocks + perGroup - 1) / perGroup));
    if (tt4) hipLaunchKernelGGL(k_fse_encode_wave<true>, grid, dim3(64 * FSE_WV_WAVES), ldsBytes, s, a, slotWords, tableWords);
    else     hipLaunchKernelGGL(k_fse_encode_wave<false>, grid, dim3(64 * FSE_WV_WAVES), ldsBytes, s, a, slotWords, tableWords);
    probe_after(PK_FSE_ENCODE_WAVE, s);
    return hipGetLastError();
}
