// fse_u16_decode.hip -- FSE_decompressU16_usingDTable (lib/fseU16.c:262-301) with the chain cells in LDS.
//
// The 16-bit-symbol stream carries ONE tANS state per block, and a tANS decoder cannot be split (it does not resynchronise), so the
// parallelism is blocks: as many chains per CU as LDS holds tables.  What the chain needs per cell is nbBits and newState -- 16 bits
// for table logs up to 12 -- so a table is 8 KiB of LDS and one 160 KB workgroup per CU carries 18 blocks; the 9-bit symbols are not
// on the chain: like k_fse_decode, the decoder lane only appends the states it decoded FROM to a ring in LDS and service waves turn
// them into symbols (gathered from the block's packed 9-bit symbol table in global memory, L2-resident) and store them as coalesced rows.
//   workgroup = 1 decoder wave (lane g walks block g: registers + LDS only) + 5 service waves (4 blocks each: input-ring refills of
//   64 bytes, state-ring records -> 8 output bytes each), talking through per-block control words in LDS.
// The bulk loop takes iterations of 4 symbols; the reference reloads its reader in front of EVERY symbol (fseU16.c:289), and a reload
// that does not clamp leaves (ptr, bitsConsumed) a function of the bits unread alone (see fse_decode.hip): the bulk runs while at least
// 65 + 48 * iterations bits are unread and four symbols fit the destination, then the reference's loop and end game are run literally
// by BitReader (:288-298), which also defines the result for truncated / corrupt input.  Table log 13 (17 bits per chain cell; the
// reference's compressor never writes one, fse_u16.hip) and whatever else k_u16_dprep marks stay with the lane-per-block kernel.
#include "internal.h"
#include "bitreader.h"

#define U16D_LOG 12u                 // largest table log of this kernel: cells = newState (12 bits) | nbBits << 12; the kernel is instantiated for
                                     // table slots of 4 KiB (table logs up to 11: 33 blocks per CU) and of 8 KiB (table log 12: 18 blocks per CU)
                                     // (measured: cells carrying 15 - nbBits, so that the bits are (t >> 17) >> field without the subtraction on the
                                     //  chain, run 2.4 % SLOWER -- 145.6 vs 149.0 GB/s -- the 64-bit shifts feeding it cost more than the subtraction)
#define U16D_TAB(LOG) (2u << (LOG))   // LDS bytes per table slot
#define U16D_RING 64u                // state ring: one entry (4 cell indices, 8 bytes) per iteration
#define U16D_IN_RING 256u
#define U16D_IN_MIRROR 16u
#define U16D_IN_CHUNK 64u
#define U16D_PHASE 16                // iterations per phase (<= 6 bytes each)
#define U16D_SRV_G 4
#define U16D_SRV_WAVES(LOG) ((LOG) >= 12u ? 5 : 9)
#define U16D_MAXG(LOG) (U16D_SRV_G * U16D_SRV_WAVES(LOG))
#define U16D_THREADS(LOG) (64 * (1 + U16D_SRV_WAVES(LOG)))
#define U16D_LDS (160 * 1024)
#define U16D_FLUSH_MIN 32u
#ifndef U16D_UNROLL
#define U16D_UNROLL 16
#endif

struct U16Ctl {
    u32 pubIters, pubPofs;           // decoder -> service (one 8-byte store): iterations done; top dword offset, bit 31 = bulk finished
    u32 srvFlushed; int srvValidLo;  // service -> decoder (read as one 8-byte load)
    int initValidLo, S32;
    u32 inLo, inHi, outLo, outHi, symLo, symHi;
};
typedef const __attribute__((address_space(3))) u16* u16d_lds_u16;
typedef const __attribute__((address_space(3))) u32* u16d_lds_u32;
DEV u32 u16d_cell(u32 addr) { return *(u16d_lds_u16)(uintptr_t)addr; }
DEV u64 u16d_peek2(const u32* p) { const u64 v = __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __asm__ volatile("" ::: "memory"); return v; }
DEV u64 u16d_load2(const u32* p) { return __hip_atomic_load((const u64*)p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void u16d_store2(u32* p, u32 lo, u32 hi) { __hip_atomic_store((u64*)p, (u64)lo | ((u64)hi << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void u16d_store(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void u16d_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// One phase: U16D_PHASE iterations of four symbols, registers + LDS only.  Window {w2:w1:w0} = payload dwords at q+8, q+4, q (ring
// coordinates), bq = unread bits of the top dword; T = the next 64 unread bits, shifted up symbol by symbol; the window slides
// down by selects and the two dwords below it are read at the top of the iteration, off the chain.
template <int NITER>
DEV void u16d_phase(u32& sRef, u32& qRef, u32& bqRef, u32 tabOff, u32 myIn, uint2* ring)
{
    u32 s = sRef, q = qRef, bq = bqRef;
    u32 w0, w1, w2;
    {   const u16d_lds_u32 wp = (u16d_lds_u32)(uintptr_t)(myIn + (q & (U16D_IN_RING - 4)));
        w0 = wp[0]; w1 = wp[1]; w2 = wp[2]; }
#pragma unroll U16D_UNROLL
    for (int it = 0; it < NITER; ++it) {
        const u16d_lds_u32 np = (u16d_lds_u32)(uintptr_t)(myIn + ((q - 8u) & (U16D_IN_RING - 4)));
        const u32 n0 = np[0], n1 = np[1];
        u64 T = ((u64)__builtin_amdgcn_alignbit(w2, w1, bq) << 32) | __builtin_amdgcn_alignbit(w1, w0, bq);
        u32 used = 0, st[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 c = u16d_cell(s);
            const u32 nb = c >> 12;
            const u32 bits = __builtin_amdgcn_ubfe((u32)(T >> 32), 32u - nb, nb);
            st[k] = s;
            s = (((c & 0xFFFu) | bits) << 1) + tabOff;
            T <<= nb; used += nb;
        }
        ring[it] = make_uint2((st[0] & 0xFFFFu) | (st[1] << 16), (st[2] & 0xFFFFu) | (st[3] << 16));
        const int left = (int)bq - (int)used;                   // unread bits of the top dword after this iteration (>= -48)
        const bool k1 = left < 0, k2 = left < -32;
        w2 = k2 ? w0 : (k1 ? w1 : w2);
        w1 = k2 ? n1 : (k1 ? w0 : w1);
        w0 = k2 ? n0 : (k1 ? n1 : w0);
        q += (u32)((left >> 5) << 2);
        bq = (u32)left & 31u;
    }
    sRef = s; qRef = q; bqRef = bq;
}

// symbol of a cell: 9 bits at bit 9 * state of the packed table k_u16_dprep leaves behind the chain cells (an unaligned 2-byte load)
typedef u16 __attribute__((aligned(1))) u16d_u16u;
DEV u32 u16d_sym_g(const __attribute__((address_space(1))) u8* base, u32 state)
{
    const u32 bit = state * 9u;
    return ((u32)*(const __attribute__((address_space(1))) u16d_u16u*)(base + (bit >> 3)) >> (bit & 7u)) & 0x1FFu;
}
DEV u32 u16d_rl(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
DEV unsigned long long u16d_rl64(unsigned long long v, int l) { return (unsigned long long)u16d_rl((u32)v, l) | ((unsigned long long)u16d_rl((u32)(v >> 32), l) << 32); }
DEV void u16d_ring_put(u32* rg, int off, u32 w)
{
    const u32 j = (u32)off & (U16D_IN_RING - 1);
    rg[j >> 2] = w;
    if (j < U16D_IN_MIRROR) rg[(U16D_IN_RING + j) >> 2] = w;
}

// service wave: lane l (< 4) keeps the books of block g0 + l; the 16-lane group k moves the input chunk of block g0 + k
template <u32 LOG>
DEV void u16d_service(u8* ldsb, U16Ctl* ctlAll, u32 slotBytes, int G, int lane, int g0)
{
    const int myG = g0 + (lane < U16D_SRV_G ? lane : 0);
    U16Ctl* const ctl = ctlAll + (myG < G ? myG : 0);
    const unsigned long long inBits = ((unsigned long long)ctl->inHi << 32) | ctl->inLo;
    const unsigned long long outBits = ((unsigned long long)ctl->outHi << 32) | ctl->outLo;
    const unsigned long long symBits = ((unsigned long long)ctl->symHi << 32) | ctl->symLo;
    const int S32 = ctl->S32;
    int validLo = ctl->initValidLo;
    u32 flushed = 0, fpos = 0;
    bool live = lane < U16D_SRV_G && myG < G && !(ctl->pubPofs >> 31);
    const int grp = lane / 16, sub = lane % 16;
    const int SgK = __shfl(S32, grp, WAVE);
    // (global-address-space pointers: a flat_* access would count on lgkmcnt too and every LDS wait of this wave would wait for the
    //  symbol gathers in flight -- fse_decode.hip)
    typedef const __attribute__((address_space(1))) u8* g_u8;
    typedef u32 __attribute__((aligned(1))) u32_u; typedef unsigned long long __attribute__((aligned(1))) u64_u;
    const g_u8 igK = (g_u8)(uintptr_t)__shfl(inBits, grp, WAVE);
    u32* const rgK = (u32*)(ldsb + (size_t)(g0 + grp) * slotBytes + U16D_RING * 8);
    {   const bool liveK = (__ballot(live) >> grp) & 1ull;
        const int vlo = __shfl(validLo, grp, WAVE);
#pragma unroll
        for (int c = 0; c < (int)(U16D_IN_RING / U16D_IN_CHUNK); ++c) {
            const int off = liveK ? vlo + (int)U16D_IN_CHUNK * c + 4 * sub : -1;
            if (off >= 0 && off + 4 <= SgK) { const u32 w = *(const __attribute__((address_space(1))) u32_u*)(igK + off); u16d_ring_put(rgK, off, w); }
            else if (off >= 0 && off < SgK) { u32 w = 0; for (int i = 0; i < 3; ++i) if (off + i < SgK) w |= (u32)igK[off + i] << (8 * i); u16d_ring_put(rgK, off, w); }
        }
        if (live) u16d_store(&ctl->srvValidLo, validLo);
    }
    u32 pend = 0;
    u32 yq[U16D_SRV_G][4];
#pragma unroll
    for (int l = 0; l < U16D_SRV_G; ++l) { yq[l][0] = yq[l][1] = yq[l][2] = yq[l][3] = 0; }
    for (;;) {
        u32 pp = 0x80000000u, it = flushed;
        if (live) { const u64 pub = u16d_load2(&ctl->pubIters); it = (u32)pub; pp = (u32)(pub >> 32); }
        const bool fin = (pp >> 31) != 0;
        const int P = (int)(pp & 0x7FFFFFFFu);
        const u32 avail = it - flushed;
        const bool wantFlush = live && (avail >= U16D_FLUSH_MIN || (fin && avail > 0));
        const bool wantFill = live && !fin && validLo > 0 && P + 4 <= validLo + (int)(U16D_IN_RING - U16D_IN_CHUNK);
        const unsigned long long fm = __ballot(wantFlush), rm = __ballot(wantFill);
        if (live && fin && avail == 0) live = false;
        if (!(fm | rm)) {
            if (!__any(live)) break;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        const bool fillK = (rm >> grp) & 1ull;
        int fillOff = -1;
        if (rm) {
            fillOff = __shfl(validLo, grp, WAVE) - (int)U16D_IN_CHUNK + 4 * sub;
            pend = 0;
            if (fillK && fillOff >= 0 && fillOff + 4 <= SgK) pend = *(const __attribute__((address_space(1))) u32_u*)(igK + fillOff);
        }
#pragma unroll
        for (int l = 0; l < U16D_SRV_G; ++l) {
            if (!((fm >> l) & 1ull)) continue;
            // (v_readlane instead of __shfl = ds_bpermute: the lane index is a constant of the unrolled loop, and the LDS pipe is what the
            //  decoder wave's chain waits on -- fse_decode.hip, FSE_SRV_READLANE)
            const u32 cnt = u16d_rl(avail, l), fp_g = u16d_rl(fpos, l);
            const g_u8 tg = (g_u8)(uintptr_t)u16d_rl64(symBits, l);
            if ((u32)lane < cnt) {
                const u32 ri = (fp_g + (u32)lane) & (U16D_RING - 1);
                const uint2 rec = *(const uint2*)(ldsb + (size_t)(g0 + l) * slotBytes + 8u * ri);
                // a record holds the low 16 bits of 4 cell addresses; the tables are table-size aligned: state = address bits [1, 1 + LOG)
                yq[l][0] = u16d_sym_g(tg, __builtin_amdgcn_ubfe(rec.x, 1u, LOG)); yq[l][1] = u16d_sym_g(tg, __builtin_amdgcn_ubfe(rec.x, 17u, LOG));
                yq[l][2] = u16d_sym_g(tg, __builtin_amdgcn_ubfe(rec.y, 1u, LOG)); yq[l][3] = u16d_sym_g(tg, __builtin_amdgcn_ubfe(rec.y, 17u, LOG));
            }
        }
        if (fillK) u16d_ring_put(rgK, fillOff, pend);
        if (wantFill) { validLo -= (int)U16D_IN_CHUNK; u16d_store(&ctl->srvValidLo, validLo); }
        if (wantFlush) u16d_store(&ctl->srvFlushed, it);
#pragma unroll
        for (int l = 0; l < U16D_SRV_G; ++l) {
            if (!((fm >> l) & 1ull)) continue;
            const u32 cnt = u16d_rl(avail, l), fl_g = u16d_rl(flushed, l);
            __attribute__((address_space(1))) u8* const og = (__attribute__((address_space(1))) u8*)(uintptr_t)(u16d_rl64(outBits, l) + 8ull * fl_g);
            if ((u32)lane < cnt) {
                const uint2 w = make_uint2(yq[l][0] | (yq[l][1] << 16), yq[l][2] | (yq[l][3] << 16));
                *(__attribute__((address_space(1))) u64_u*)(og + 8u * lane) = (unsigned long long)w.x | ((unsigned long long)w.y << 32);
            }
        }
        if (wantFlush) { fpos = (fpos + (it - flushed)) & (U16D_RING - 1); flushed = it; }
    }
}

// the blocks of an instance: marked by k_u16_dprep (state 1) and of its table-log class
template <u32 LOG> DEV bool u16d_mine(const U16Meta& m) { return m.state == 1 && (LOG >= 12u ? m.tableLog >= 12u : m.tableLog < 12u); }

// LDS: G table slots (4 or 8 KiB) on table-size aligned addresses | U16Ctl[G] | per block: state ring (64 x 8 B), input ring (256 + 16 B).
// A workgroup takes G consecutive blocks of the batch and decodes those of its class; the other instance's launch takes the rest (a batch
// of one class -- blocks of one size and kind, as a caller of FSE_decompressU16 cuts them -- fills every slot of one of the two).
template <u32 LOG>
__global__ __launch_bounds__(U16D_THREADS(LOG)) void k_u16_decode_lds(U16DArgs a, int G, u32 slotBytes)
{
    constexpr u32 TAB = U16D_TAB(LOG), THREADS = U16D_THREADS(LOG);
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t first = (size_t)blockIdx.x * G;
    u8* const lds8 = (u8*)lds;
    U16Ctl* const ctlAll = (U16Ctl*)(lds8 + (size_t)G * TAB);
    u8* const ldsb = (u8*)ctlAll + (size_t)G * sizeof(U16Ctl);
    const size_t nTab = a.nBlocks - first < (size_t)G ? a.nBlocks - first : (size_t)G;

    // ---- stage the chain cells of every block this kernel decodes (LDS-DMA: 1 KiB per wave instruction)
    for (u32 p = (u32)wave; p < (u32)nTab * (TAB >> 10); p += THREADS / 64) {
        const u32 g = p / (TAB >> 10), k = p % (TAB >> 10);
        if (!u16d_mine<LOG>(a.meta[first + g])) continue;                 // uniform per wave
        const u8* const src = (const u8*)(a.cells + ((first + g) << FSEHIP_FSEU16_MAX_TABLELOG)) + 1024u * k + 16u * (u32)lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds8 + (size_t)g * TAB + 1024u * k), 16, 0, 0);
    }

    // ---- per-block set-up by the decoder wave: lane g walks block first + g
    const int gsl = lane;
    const bool inRange = wave == 0 && gsl < G && first + (size_t)gsl < a.nBlocks;
    const size_t b = inRange ? first + (size_t)gsl : 0;
    bool owner = inRange && u16d_mine<LOG>(a.meta[b]);
    const u32 tl = owner ? a.meta[b].tableLog : 0u;
    const u32 ldsBase = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)lds8;
    if (ldsBase & (TAB - 1)) __builtin_trap();                           // the records carry 16 address bits: slots are table-size aligned
    const u32 tabOff = ldsBase + (u32)(gsl < G ? gsl : 0) * TAB;
    const u16* const A = (const u16*)(lds8 + (tabOff - ldsBase));
    const u8* const syms = (const u8*)((const u16*)(a.cells + (b << FSEHIP_FSEU16_MAX_TABLELOG)) + ((size_t)1 << FSEHIP_FSEU16_MAX_TABLELOG));   // 9-bit symbols, packed
    const u8* in = nullptr; size_t S = 0; u16* out = nullptr;
    const size_t cap = a.dstCapacity;
    BitReader r; r.base = nullptr; r.size = 0; r.at = 0; r.win = 0; r.used = 0;
    u32 state = 0;
    bool initOk = false;
    if (owner) {
        const u32 hdr = a.meta[b].hdrSize;
        in = a.csrc + b * a.cStride + hdr;
        S = (a.cSizes ? a.cSizes[b] : a.uniformCSize) - hdr;
        out = (u16*)((u8*)a.dst + b * a.dstStrideBytes);
        initOk = !is_err(r.init(in, S));                                 // (the verdict of BIT_initDStream is not looked at, fseU16.c:284)
        state = r.read(tl); (void)r.reload();                            // FSE_initDState
    }
    // bulk: B = bits unread (negative once a short stream has been read beyond its end); every reload of a phase is one that does not
    // clamp and does not start from the stream start while B >= 65 (fse_decode.hip), 48 bits per iteration at most
    const int B0 = (owner && initOk && S < (1ull << 28)) ? (int)(8u * ((u32)r.at + 8u)) - (int)r.used : 0;
    int grp = cap / 4 > (size_t)(1 << 30) ? (1 << 30) : (int)(cap / 4);   // iterations of four symbols that fit the destination
    bool can = B0 >= 65 + 48 * U16D_PHASE && grp >= U16D_PHASE;
    bool can1 = B0 >= 65 + 48 && grp >= 1;                               // finishing phases of one iteration, down to 113 unread bits
    const bool everBulk = can || can1;
    const u32 inA = (u32)((uintptr_t)in & 3u);
    u32 s = tabOff + 2u * state, q = 0, bq = 0, iters = 0;
    int validLo = 0;
    if (everBulk) {
        const u32 B = (u32)B0 + 8u * inA;                                // bits counted from the aligned base below the payload
        q = 4u * (B >> 5) - 8u; bq = B & 31u;
        const int c0 = (int)((0 - ((uintptr_t)in - inA)) & (U16D_IN_CHUNK - 1));
        validLo = (((int)q + 8 - 112 - c0) & ~(int)(U16D_IN_CHUNK - 1)) + c0;
    }
    U16Ctl* const ctl = ctlAll + (gsl < G ? gsl : 0);
    if (wave == 0 && gsl < G) {
        ctl->pubIters = 0; ctl->pubPofs = everBulk ? q + 8u : 0x80000000u;
        ctl->srvFlushed = 0; ctl->srvValidLo = 0x7FFFFFFF;
        ctl->initValidLo = validLo; ctl->S32 = (int)(S + inA);
        const unsigned long long ib = (unsigned long long)(uintptr_t)(in - inA), ob = (unsigned long long)(uintptr_t)out, sb = (unsigned long long)(uintptr_t)syms;
        ctl->inLo = (u32)ib; ctl->inHi = (u32)(ib >> 32); ctl->outLo = (u32)ob; ctl->outHi = (u32)(ob >> 32); ctl->symLo = (u32)sb; ctl->symHi = (u32)(sb >> 32);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);                                  // vmcnt(0): the LDS-DMA pieces have landed
    __syncthreads();
    if (wave >= 1) { u16d_service<LOG>(ldsb, ctlAll, slotBytes, G, lane, (wave - 1) * U16D_SRV_G); return; }

    __builtin_amdgcn_s_setprio(3);
    uint2* const myRing = (uint2*)(ldsb + (size_t)(gsl < G ? gsl : 0) * slotBytes);
    const u32 myIn = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)(ldsb + (size_t)(gsl < G ? gsl : 0) * slotBytes + U16D_RING * 8);
    u32 rpos = 0;
    u64 srvNext = u16d_peek2(&ctl->srvFlushed);
    while (__any(can)) {
        const u32 fl = (u32)srvNext;
        const int vlo = (int)(u32)(srvNext >> 32);
        srvNext = u16d_peek2(&ctl->srvFlushed);
        const int lowest = (int)q - (6 * U16D_PHASE + 8);
        const bool ready = can & (iters + U16D_PHASE - fl <= U16D_RING) & ((lowest > 0 ? lowest : 0) >= vlo);
        if (ready) {
            u16d_phase<U16D_PHASE>(s, q, bq, tabOff, myIn, myRing + rpos);
            rpos = (rpos + U16D_PHASE) & (U16D_RING - 1);
            iters += U16D_PHASE; grp -= U16D_PHASE;
            const u32 Bp = 8u * (q + 8u) + bq - 8u * inA;
            can = (Bp >= 65u + 48u * U16D_PHASE) & (grp >= U16D_PHASE);
            can1 = (Bp >= 65u + 48u) & (grp >= 1);
            u16d_store2(&ctl->pubIters, iters, (can | can1) ? q + 8u : ((q + 8u) | 0x80000000u));
        }
    }
    while (__any(can1)) {                                                // every lane is through with the long phases
        const u32 fl = (u32)srvNext;
        const int vlo = (int)(u32)(srvNext >> 32);
        srvNext = u16d_peek2(&ctl->srvFlushed);
        const int lowest = (int)q - 16;
        const bool ready = can1 & (iters + 1 - fl <= U16D_RING) & ((lowest > 0 ? lowest : 0) >= vlo);
        if (ready) {
            u16d_phase<1>(s, q, bq, tabOff, myIn, myRing + rpos);
            rpos = (rpos + 1) & (U16D_RING - 1);
            iters += 1; grp -= 1;
            const u32 Bp = 8u * (q + 8u) + bq - 8u * inA;
            can1 = (Bp >= 65u + 48u) & (grp >= 1);
            u16d_store2(&ctl->pubIters, iters, can1 ? q + 8u : ((q + 8u) | 0x80000000u));
        }
    }
    if (!owner) return;
    size_t op = 4 * (size_t)iters;
    if (iters) {                                                         // back to the reference's reader: the state after a reload at this position
        const u32 B = 8u * (q + 8u) + bq - 8u * inA;                     // (>= 65: the reload in front of the next symbol is the same as the reference's)
        r.at = (size_t)((B + 7u) >> 3) - 8; r.used = 8u * ((u32)r.at + 8u) - B; r.win = ldg64u(in + r.at);
        state = (s - tabOff) >> 1;
    }
    // ---- literal: fseU16.c:288-298 on the LDS cells and the symbol table in global memory
    auto step = [&]() {
        const u32 c = A[state];
        const u16 sym = (u16)u16d_sym_g((const __attribute__((address_space(1))) u8*)(uintptr_t)syms, state);
        const u32 low = r.read(c >> 12);
        state = (c & 0xFFFu) + low;
        return sym;
    };
    while (r.reload() < BR_COMPLETED && op < cap) out[op++] = step();
    size_t result;
    if (!(r.at == 0 && r.used == 64)) result = FERR(corruption_detected);   // BIT_endOfDStream
    else {
        while (state && op < cap) out[op++] = step();
        result = state ? FERR(corruption_detected) : op;
    }
    a.results[b] = result;
}

template <u32 LOG>
static hipError_t u16d_launch(const U16DArgs& a, hipStream_t s)
{
    const hipError_t e = ensure_dyn_lds((const void*)k_u16_decode_lds<LOG>, U16D_LDS);
    if (e != hipSuccess) return e;
    const u32 slotBytes = U16D_RING * 8 + U16D_IN_RING + U16D_IN_MIRROR;
    int G = (int)((U16D_LDS - 16) / (U16D_TAB(LOG) + slotBytes + sizeof(U16Ctl)));
    if (G > U16D_MAXG(LOG)) G = U16D_MAXG(LOG);
    hipLaunchKernelGGL(k_u16_decode_lds<LOG>, dim3((unsigned)((a.nBlocks + G - 1) / G)), dim3(U16D_THREADS(LOG)), U16D_LDS, s, a, G, slotBytes);
    return hipGetLastError();
}
hipError_t launch_u16_decode_lds(const U16DArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    const hipError_t e = u16d_launch<11>(a, s);                           // table logs up to 11: 4 KiB slots
    if (e != hipSuccess) return e;
    return u16d_launch<12>(a, s);                                         // table log 12: 8 KiB slots
}
