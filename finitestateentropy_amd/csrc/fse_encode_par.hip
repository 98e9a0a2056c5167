// fse_encode_par.hip -- a2: FSE_compress_usingCTable, block-parallel variant
// (reference: lib/fse_compress.c:554-623, lib/fse.h:503-527; format SURVEY A.1/A.3).
//
// tANS encoding is a loop-carried chain (two interleaved chains per block), but the encoder state is only
// tableLog bits wide and every step replaces part of it by a function of the symbol alone, so the state
// "forgets" where it came from after roughly tableLog emitted bits.  That makes the chains splittable:
//
//   one 256-thread workgroup per block.  In emission order (last source byte first) thread t owns a contiguous
//   range of symbols.  Pass 1: it warms both chains up over the FSE_WARM symbols in front of its range starting
//   from an arbitrary state, remembers the states it arrives with (its speculated start), then runs its range
//   counting bits and remembers the states it ends with.  Verification: thread t's speculated start must equal
//   thread t-1's end; thread 0 starts from the exact FSE_initCState2 states, so if every link matches, every
//   start is exact by induction.  A thread whose link does not match re-runs its range from its predecessor's
//   end, and the check is repeated until no link changes (worst case this degenerates into the serial
//   algorithm; on probagen data a repair round is rare).  Nothing is ever assumed: the output is bit-exact
//   by construction.
//   Pass 2: a prefix sum of the per-thread bit counts gives every thread its bit offset (and the exact
//   compressed size / the BIT_closeCStream verdict before a single bit is written); each thread re-runs its
//   range from its verified start and ORs its bits into an LDS image of the output, which the workgroup then
//   copies out with coalesced stores.  CState2, CState1 and the end mark follow (lib/fse_compress.c:608-610).
#include "internal.h"

#define FSE_PAR_THREADS 256
#define FSE_WARM 256            // warm-up symbols (128 per chain) in front of every range

// LDS-resident CTable copy: tt[2*sym] = 2*deltaFindState + byte offset of stateTable, tt[2*sym+1] = deltaNbBits
#define PAR_STEP_NB(x, sym, nb)                                                                      \
    {   const u32 f2 = tt[2 * (sym)], dn = tt[2 * (sym) + 1];                                        \
        nb = ((x) + dn) >> 16;                                                                       \
        (x) = *(const u16*)(ldsb + ((((x) >> nb) << 1) + f2));                                       \
    }

DEV u32 par_init_state(const u8* ldsb, const u32* tt, u32 sym)                  // FSE_initCState2, lib/fse.h:503-512
{
    const u32 f2 = tt[2 * sym], dn = tt[2 * sym + 1];
    const u32 nb = (dn + (1u << 15)) >> 16;
    return *(const u16*)(ldsb + (((((nb << 16) - dn) >> nb) << 1) + f2));
}

// The block is staged in LDS with a 4-byte skew per 128-byte row, so that the threads of a wave -- whose ranges are
// a whole number of rows apart -- read different banks.  SRC(i) = byte i of the block.
#define SRC(i) srcl[(i) + (((i) >> 7) << 2)]

// one chain step with the symbolTT entry already in registers
#define PAR_STEP_TT(x, f2, dn, nb)                                                                   \
    {   nb = ((x) + (dn)) >> 16;                                                                     \
        (x) = *(const u16*)(ldsb + ((((x) >> nb) << 1) + (f2)));                                     \
    }

// run symbols j in [j0, j1) (distance from the block end; even j -> chain A, odd j -> chain B); returns emitted bits.
// Only the stateTable lookup is on the dependent chain: symbol bytes are fetched two pairs ahead and their
// symbolTT entries one pair ahead.
DEV u32 par_run(const u8* ldsb, const u32* tt, const u8* srcl, u32 n, u32 j0, u32 j1, u32& xa, u32& xb)
{
    u32 bits = 0;
    u32 j = j0;
    if ((j & 1u) && j < j1) { u32 nb; PAR_STEP_NB(xb, SRC(n - 1 - j), nb) bits += nb; ++j; }
    const u32 pairs = (j1 - j) >> 1;
    if (pairs) {
        // indices clamp at 0: reads past the range are harmless (inside the block copy) and unused
        #define SYM_AT(jj) SRC((n - 1 - (jj)) < n ? (n - 1 - (jj)) : 0u)
        u32 sa1 = SYM_AT(j + 2), sb1 = SYM_AT(j + 3);                           // pair +1 symbols
        u32 fa, da, fb, db;
        {   const u32 sa0 = SRC(n - 1 - j), sb0 = SRC(n - 2 - j);
            fa = tt[2 * sa0]; da = tt[2 * sa0 + 1]; fb = tt[2 * sb0]; db = tt[2 * sb0 + 1]; }
        for (u32 p = 0; p < pairs; ++p, j += 2) {
            const u32 sa2 = SYM_AT(j + 4), sb2 = SYM_AT(j + 5);                 // pair +2 symbols
            const u32 nfa = tt[2 * sa1], nda = tt[2 * sa1 + 1], nfb = tt[2 * sb1], ndb = tt[2 * sb1 + 1];   // pair +1 entries
            u32 na, nbb;
            PAR_STEP_TT(xa, fa, da, na)
            PAR_STEP_TT(xb, fb, db, nbb)
            bits += na + nbb;
            fa = nfa; da = nda; fb = nfb; db = ndb; sa1 = sa2; sb1 = sb2;
        }
        #undef SYM_AT
    }
    if (j < j1) { u32 nb; PAR_STEP_NB(xa, SRC(n - 1 - j), nb) bits += nb; }
    return bits;
}

DEV void par_or_bits(u32* img, u64 P, u64 bits, u32 nb)      // nb <= 32
{
    if (nb == 0) return;
    const size_t w = (size_t)(P >> 5);
    const u32 sh = (u32)P & 31u;
    const u64 v = bits << sh;
    const u32 a0 = (u32)v, a1 = (u32)(v >> 32);
    if (a0) atomicOr(&img[w], a0);
    if (a1) atomicOr(&img[w + 1], a1);
}

template <bool GLOBAL>
DEV void par_emit(u32* img, u64 P, const u8* ldsb, const u32* tt, const u8* srcl, u32 n, u32 j0, u32 j1, u32 xa, u32 xb)
{
    // the chunk's bits are accumulated in a 64-bit register; completed 32-bit words are OR-ed into the image
    size_t w = (size_t)(P >> 5);
    u32 nacc = (u32)P & 31u;
    u64 acc = 0;
    u32 nsym = j0 < j1 ? SRC(n - 1 - j0) : 0u;
    for (u32 j = j0; j < j1; ++j) {
        u32& x = (j & 1u) ? xb : xa;
        const u32 sym = nsym;
        if (j + 1 < j1) nsym = SRC(n - 2 - j);
        const u32 f2 = tt[2 * sym], dn = tt[2 * sym + 1];
        const u32 nb = (x + dn) >> 16;
        acc |= (u64)(x & ((1u << nb) - 1u)) << nacc;
        nacc += nb;
        x = *(const u16*)(ldsb + (((x >> nb) << 1) + f2));
        if (nacc >= 32) { atomicOr(&img[w], (u32)acc); acc >>= 32; nacc -= 32; ++w; }
    }
    if (nacc && (u32)acc) atomicOr(&img[w], (u32)acc);
}

__global__ __launch_bounds__(FSE_PAR_THREADS) void k_fse_encode_par(FseEncArgs a, u32 tableWords, u32 srcBytes, u32 imgBytes)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u8* const ldsb = (const u8*)lds;
    u32* const startArr = lds + tableWords;                  // [256] speculated / verified start states  (A | B << 16)
    u32* const endArr = startArr + FSE_PAR_THREADS;          // [256] end states
    u32* const bitsArr = endArr + FSE_PAR_THREADS;           // [256] bit counts, later exclusive prefix
    u32* const misc = bitsArr + FSE_PAR_THREADS;             // [8]
    u8* const srcl = (u8*)(misc + 8);                        // skewed copy of the source block (srcBytes)
    u32* const img = (u32*)(srcl + srcBytes);
    const size_t b = blockIdx.x;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    u32 hdr = 0;
    if (a.meta) { if (fse_enc_skip(a.meta[b].state, a.onlyState)) return; hdr = a.meta[b].hdrSize; }
    const u32* const gct = a.ctables + b * a.ctStrideU32;
    const u32 h0 = gct[0];
    const u32 tl = h0 & 0xFFFFu, msv = h0 >> 16;
    if (tl > a.maxTableLog || msv > 255u) { if (tid == 0) a.results[b] = FERR(tableLog_tooLarge); return; }
    const u8* const src = view_ptr(a.src, b);
    const size_t n64 = view_size(a.src, b);
    u8* const dst = a.dst + b * a.dstStride + hdr;
    const size_t cap = a.dstCapacity - hdr;
    if (n64 >= ((size_t)1 << 31)) { if (tid == 0) a.results[b] = FERR(srcSize_wrong); return; }
    const u32 n = (u32)n64;
    if (n <= 2 || cap <= 8) { if (tid == 0) a.results[b] = 0; return; }      // fse_compress.c:566-568

    // ---- stage the CTable (coalesced), rebasing deltaFindState to LDS byte addresses of stateTable (offset 4)
    const u32 ttStart = 1 + (tl ? (1u << (tl - 1)) : 1u);
    const u32 words = ttStart + 2 * (msv + 1);
    for (u32 i = tid; i < words; i += FSE_PAR_THREADS) {
        u32 v = gct[i];
        if (i >= ttStart && (((i - ttStart) & 1u) == 0)) v = 2u * v + 4u;
        lds[i] = v;
    }
    // ---- stage the source block (coalesced global reads, skewed LDS layout)
    {   const u32 head = (u32)((0 - (uintptr_t)src) & 3u) < n ? (u32)((0 - (uintptr_t)src) & 3u) : n;
        if (tid < head) SRC(tid) = src[tid];
        const u32 nw = (n - head) >> 2;
        const u32* s32 = (const u32*)(src + head);
        for (u32 i = tid; i < nw; i += FSE_PAR_THREADS) {
            const u32 w = s32[i];
            const u32 p = head + 4 * i;
            SRC(p) = (u8)w; SRC(p + 1) = (u8)(w >> 8); SRC(p + 2) = (u8)(w >> 16); SRC(p + 3) = (u8)(w >> 24);
        }
        const u32 done = head + 4 * nw;
        if (tid < n - done) SRC(done + tid) = src[done + tid];
    }
    __syncthreads();
    const u32* const tt = lds + ttStart;

    // ---- ranges in emission order: symbols j = 2 .. n-1 (j = 0, 1 only initialise the chains)
    const u32 m = n - 2;
    u32 C = (m + FSE_PAR_THREADS - 1) / FSE_PAR_THREADS;
    C = (C + 1u) & ~1u;                                                     // even: every range starts on chain A
    if (C < 2) C = 2;
    const u32 j0 = 2 + tid * C;
    const u32 j1 = j0 + C < n ? j0 + C : n;
    const bool mine = j0 < n;                                               // non-empty range

    // ---- pass 1: speculated start, bit count, end states
    u32 xa = 0, xb = 0, start = 0, end = 0, bits = 0;
    if (mine) {
        if (j0 <= 2 + FSE_WARM) {                                           // the warm-up would reach the block end: be exact
            xa = par_init_state(ldsb, tt, SRC(n - 1));
            xb = par_init_state(ldsb, tt, SRC(n - 2));
            par_run(ldsb, tt, srcl, n, 2, j0, xa, xb);
        } else {
            xa = xb = 1u << tl;                                             // any state will do: it is verified below
            par_run(ldsb, tt, srcl, n, j0 - FSE_WARM, j0, xa, xb);
        }
        start = xa | (xb << 16);
        bits = par_run(ldsb, tt, srcl, n, j0, j1, xa, xb);
        end = xa | (xb << 16);
    }
    startArr[tid] = start; endArr[tid] = end; bitsArr[tid] = bits;

    // ---- verification / repair: start[t] must equal end[t-1]; thread 0 (and every thread that ran from the block end) is exact
    for (;;) {
        __syncthreads();
        const u32 prevEnd = tid ? endArr[tid - 1] : 0u;
        const bool bad = mine && tid > 0 && start != prevEnd;
        __syncthreads();                                                    // all links read before anybody rewrites
        if (bad) {
            start = prevEnd;
            xa = start & 0xFFFFu; xb = start >> 16;
            bits = par_run(ldsb, tt, srcl, n, j0, j1, xa, xb);
            end = xa | (xb << 16);
            startArr[tid] = start; endArr[tid] = end; bitsArr[tid] = bits;
        }
        if (!__syncthreads_or(bad ? 1 : 0)) break;
    }

    // ---- prefix sum of the bit counts (4 waves)
    u32 incl = bits;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)incl, off, WAVE); if ((int)lane >= off) incl += o; }
    if (lane == 63) misc[wave] = incl;
    __syncthreads();
    u32 waveBase = 0;
    for (u32 w = 0; w < wave; ++w) waveBase += misc[w];
    const u32 excl = waveBase + incl - bits;
    const u64 bodyBits = (u64)misc[0] + misc[1] + misc[2] + misc[3];
    const u32 lastThread = (m + C - 1) / C - 1;                              // owner of the final states
    const u32 fin = endArr[lastThread];

    // ---- verdict (BIT_closeCStream, bitstream.h:254-260): total bits incl. the two states and the end mark
    const u64 totalBits = bodyBits + 2u * tl + 1u;
    const size_t whole = (size_t)(totalBits >> 3);
    size_t csize = (whole >= cap - 8) ? 0 : (size_t)((totalBits + 7) >> 3);
    size_t result = csize;
    if (a.meta) result = (csize != 0 && (size_t)hdr + csize < n64 - 1) ? (size_t)hdr + csize : 0;   // fse_compress.c:668-676
    if (result == 0) { if (tid == 0) a.results[b] = 0; return; }

    // ---- pass 2: emit into an image addressed from the 4-byte aligned word holding dst[0]
    const u32 lead = (u32)((uintptr_t)dst & 3u);
    u8* const dstAl = dst - lead;
    const size_t imgWords = (lead + csize + 3) >> 2;
    const bool inLds = (imgWords * 4 + 8 <= imgBytes);
    if (inLds) { for (size_t i = tid; i < imgWords + 1; i += FSE_PAR_THREADS) img[i] = 0; }
    else { for (size_t i = tid; i < csize; i += FSE_PAR_THREADS) dst[i] = 0; }
    __syncthreads();
    const u64 base = 8ull * lead;
    if (mine) {
        xa = start & 0xFFFFu; xb = start >> 16;
        if (inLds) par_emit<false>(img, base + excl, ldsb, tt, srcl, n, j0, j1, xa, xb);
        else par_emit<true>((u32*)dstAl, base + excl, ldsb, tt, srcl, n, j0, j1, xa, xb);
    }
    if (tid == 0) {
        // fse_compress.c:608-609 : CState2 then CState1.  n even -> CState2 is the even-distance chain (:577-580), n odd -> CState1 (:572-576)
        const u32 fa = fin & 0xFFFFu, fb = fin >> 16;
        const u32 c2 = (n & 1u) ? fb : fa, c1 = (n & 1u) ? fa : fb;
        const u32 mask = (1u << tl) - 1u;
        u32* const target = inLds ? img : (u32*)dstAl;
        par_or_bits(target, base + bodyBits, c2 & mask, tl);
        par_or_bits(target, base + bodyBits + tl, c1 & mask, tl);
        par_or_bits(target, base + bodyBits + 2u * tl, 1, 1);
    }
    __syncthreads();
    if (inLds) {                                         // coalesced copy-out: whole aligned words, edge words bytewise
        const u8* ib = (const u8*)img;
        const size_t endb = lead + csize;
        for (size_t w = tid; w < imgWords; w += FSE_PAR_THREADS) {
            const size_t lo = 4 * w, hi = 4 * w + 4;
            if (lo >= lead && hi <= endb) ((u32*)dstAl)[w] = img[w];
            else for (size_t i = (lo > lead ? lo : lead); i < (hi < endb ? hi : endb); ++i) dstAl[i] = ib[i];
        }
    }
    if (tid == 0) a.results[b] = result;
}

hipError_t launch_fse_encode_par(FseEncArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    static bool attrSet = false;
    const size_t maxLds = 80 * 1024;
    if (!attrSet) {
        hipError_t e = hipFuncSetAttribute((const void*)k_fse_encode_par, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLds);
        if (e != hipSuccess) return e;
        attrSet = true;
    }
    const u32 tableWords = (1 + (1u << (a.maxTableLog - 1)) + 512 + 3) & ~3u;
    const size_t n = a.src.uniform;
    const size_t srcBytes = (n + ((n >> 7) << 2) + 16 + 15) & ~(size_t)15;      // skewed copy of the block
    const size_t fixed = (size_t)tableWords * 4 + (3 * FSE_PAR_THREADS + 8) * 4 + srcBytes;
    if (fixed + 64 > maxLds) return launch_fse_encode(a, s);                    // block too large for the LDS copy: lane-per-block kernel
    size_t img = a.dstCapacity + 32;
    if (fixed + img > maxLds) img = maxLds - fixed;
    img &= ~(size_t)15;
    probe_before(PK_FSE_ENCODE_PAR, s);
    hipLaunchKernelGGL(k_fse_encode_par, dim3((unsigned)a.nBlocks), dim3(FSE_PAR_THREADS), fixed + img, s, a, tableWords, (u32)srcBytes, (u32)img);
    probe_after(PK_FSE_ENCODE_PAR, s);
    return hipGetLastError();
}
