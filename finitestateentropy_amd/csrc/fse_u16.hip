// fse_u16.hip -- SURVEY 8(f) rank 4: the 16-bit-symbol variant of FSE (lib/fseU16.c: FSE_countU16 :121-146, FSE_compressU16 :203-256
// with FSE_compressU16_usingCTable :150-200, FSE_decompressU16 :306-329 with FSE_decompressU16_usingDTable :273-301).
//
// Same table construction as the byte coder (lib/fse_compress.c:66-169, lib/fse_decompress.c:71-126 compiled with
// FSE_FUNCTION_TYPE U16) over a wider alphabet -- up to FSEU16_MAX_SYMBOL_VALUE = 286 symbols, table logs up to 13, default 12
// (fseU16.c:43-48) -- but a different stream: ONE tANS state (fseU16.c:159-199), decoded until the bit stream is used up and the
// state is back at 0 (:288-298).  Not a throughput configuration of the reference (no BASELINE entry), so the split is:
//   * everything around the chain is wave-parallel, one wave per block: histogram (LDS atomics), table log, normalisation and
//     NCount header with the per-symbol-lane code of wave_glue.h (five symbols per lane), and a table builder restated for any
//     alphabet: the spread as "the k-th kept visit of the walk m -> m*step gets the symbol whose cumulative range holds k" (every
//     lane takes a run of visits and walks the cumulative counts alongside), the ranks -- the order of a symbol's cells -- by
//     walking the cells 64 at a time: lanes holding the same symbol find each other with nine ballots (one per symbol bit);
//   * the encoder's chain is split across the 64 lanes of a wave by speculation and verification (k_u16_encode_wave: the scheme of
//     fse_encode_wave.hip on one chain); blocks under 2048 symbols take the reference's loop on one lane (k_u16_encode), with its flush
//     cadence and clamping;
//   * a tANS decoder cannot be split (it does not resynchronise): the decoder is one chain per block with the 16-bit chain cells in LDS
//     (18 blocks per CU) and the symbols gathered off the chain by service waves (k_u16_decode_lds, fse_u16_decode.hip); table log 13,
//     which the reference's compressor never writes, stays with the reference's loop one lane per block on 32-bit cells in global
//     memory (k_u16_decode).
// Where the reference's behaviour is undefined the device path refuses instead: FSE_compressU16 with 8 bytes or less behind the
// header (BIT_initCStream's error is ignored at fseU16.c:164 and the flushes then write in front of the buffer) stores no payload
// and returns what the reference would (the header size alone); FSE_decompressU16 with nothing behind the header (the reference
// dereferences a null stream pointer) returns srcSize_wrong.
#include "internal.h"
#include "wave_glue.h"
#include "ncount_reader.h"
#include "bitreader.h"

#define U16_MAXSV   FSEHIP_FSEU16_MAX_SYMBOL_VALUE      // 286
#define U16_MAXTL   FSEHIP_FSEU16_MAX_TABLELOG          // 13
#define U16_DEFTL   FSEHIP_FSEU16_DEFAULT_TABLELOG      // 12
#define U16_SPL     5                                    // symbols per lane: 64 x 5 = 320 >= 287
#define U16_SYMS    (64 * U16_SPL)

// ---- LDS of the builders (one wave per block) ------------------------------------------------------------------------------
// TLMAX = the largest table log the instance builds: the cell -> symbol map is most of the footprint, and the footprint is what bounds
// the builders (latency-bound, one wave per block: waves per CU).  The compress side never goes beyond 12 (k_u16_cprep), a stream may
// carry 13: k_u16_dprep<12> leaves such blocks (still marked U16_PARSED) to a second launch of the 13-bit instance -- 12.5 KB -> 12 waves per CU
// instead of 7 for everything the reference's compressor writes.
template <u32 TLMAX>
struct U16Lds {
    u32 cnt[U16_SYMS];          // histogram / scratch
    s16 nrm[U16_SYMS];          // normalised counters
    u16 cum[U16_SYMS + 1];      // cells in front of every symbol (|counter|, symbol order)
    u16 pcum[U16_SYMS + 1];     // kept visits in front of every symbol (positive counters only)
    u16 seen[U16_SYMS];         // cells of the symbol met so far (rank pass)
    u32 img[160];               // NCount header image
    u32 scal[8];
    u16 symTab[1u << TLMAX];       // symbol of every cell
};

DEV u32 u16_scan_excl(u32 v, u32 lane, u32* total)
{
    u32 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)incl, off, WAVE); if ((int)lane >= off) incl += o; }
    *total = (u32)__shfl((int)incl, 63, WAVE);
    return incl - v;
}

// Spread + rank for the counters in L.nrm (zero beyond maxSV).  Fills L.symTab and L.cum, then calls emit(u, symbol, rank) once per
// cell, ranks ascending with u inside a symbol.  All 64 lanes, uniform arguments, the workgroup is this wave.
template <u32 TLMAX, class Emit>
DEV void u16_spread_rank(U16Lds<TLMAX>& L, u32 maxSV, u32 tl, u32 lane, Emit&& emit)
{
    const u32 ts = 1u << tl, mask = ts - 1u, step = (ts >> 1) + (ts >> 3) + 3u;
    // ---- cumulative counts (lane l: symbols 5l .. 5l+4)
    int n[U16_SPL]; u32 mineAbs = 0, minePos = 0, mineLow = 0;
#pragma unroll
    for (int i = 0; i < U16_SPL; ++i) {
        const u32 s = U16_SPL * lane + i;
        n[i] = s <= maxSV ? (int)L.nrm[s] : 0;
        mineAbs += (u32)(n[i] < 0 ? 1 : n[i]); minePos += (u32)(n[i] > 0 ? n[i] : 0); mineLow += n[i] == -1;
    }
    u32 tAbs, tPos, nLow;
    u32 a = u16_scan_excl(mineAbs, lane, &tAbs), p = u16_scan_excl(minePos, lane, &tPos), lw = u16_scan_excl(mineLow, lane, &nLow);
#pragma unroll
    for (int i = 0; i < U16_SPL; ++i) {
        const u32 s = U16_SPL * lane + i;
        L.cum[s] = (u16)a; L.pcum[s] = (u16)p; L.seen[s] = 0;
        if (n[i] == -1) { L.symTab[ts - 1u - lw] = (u16)s; ++lw; }         // low-probability symbols take the top cells, in symbol order downwards
        a += (u32)(n[i] < 0 ? 1 : n[i]); p += (u32)(n[i] > 0 ? n[i] : 0);
    }
    if (lane == 63) { L.cum[U16_SYMS] = (u16)a; L.pcum[U16_SYMS] = (u16)p; }
    __syncthreads();
    const u32 high = ts - 1u - nLow;                                         // highThreshold (ts - 1 - nLow >= 0: a table of only such symbols has nLow == ts)
    // ---- spread: my run of visits m, the kept ones numbered by a wave scan
    const u32 C = ts >= 64u ? ts >> 6 : 1u;
    const bool act = lane * C < ts;
    u32 kept = 0;
    if (act && nLow < ts) for (u32 i = 0; i < C; ++i) kept += (((lane * C + i) * step) & mask) <= high;
    u32 tk;
    u32 k = u16_scan_excl(kept, lane, &tk);
    if (act && kept) {
        // symbol of visit k: the last s with pcum[s] <= k among the symbols with a positive counter (binary search, then walk along)
        u32 lo = 0, hi = U16_SYMS;                                           // invariant: pcum[lo] <= k < pcum[hi]
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if ((u32)L.pcum[mid] <= k) lo = mid; else hi = mid; }
        u32 s = lo, next = L.pcum[s + 1];
        for (u32 i = 0; i < C; ++i) {
            const u32 u = ((lane * C + i) * step) & mask;
            if (u > high) continue;
            while (k >= next) { ++s; next = L.pcum[s + 1]; }
            L.symTab[u] = (u16)s;
            ++k;
        }
    }
    __syncthreads();
    // ---- ranks: the cells in ascending order, 64 at a time
    for (u32 u0 = 0; u0 < ts; u0 += 64) {
        const u32 u = u0 + lane;
        const bool in = u < ts;
        const u32 s = in ? (u32)L.symTab[u] : 0xFFFFu;
        unsigned long long same = __ballot(in);
#pragma unroll
        for (int b = 0; b < 9; ++b) {                                        // lanes with my symbol: agree on every bit of it
            const unsigned long long bal = __ballot((s >> b) & 1u);
            same &= ((s >> b) & 1u) ? bal : ~bal;
        }
        if (in) {
            const u32 before = (u32)__builtin_popcountll(same & ((1ull << lane) - 1ull));
            const u32 base = L.seen[s];
            emit(u, s, base + before);
            if ((same >> lane) == 1ull) L.seen[s] = (u16)(base + (u32)__builtin_popcountll(same));   // the group's top lane
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
    }
}

// ---- compress side: FSE_compressU16 up to the table (fseU16.c:203-249) -------------------------------------------------------
// Two instances, launched one after the other: <11, false> for every block -- 8.6 KB of LDS, 18 waves per CU (the table build is latency-bound;
// blocks of up to 16,384 symbols get table log 11 or less from FSE_optimalTableLog) -- which leaves a block whose table log comes out as 12
// marked U16_WIDE, and <12, true>, which takes only those (12.7 KB, 12 waves per CU).
#define U16_WIDE 6u
#define U16_SKEWED 7u            // compress side: a table for the lane-per-block encoder (one symbol holds more than 63/64 of the cells)
template <u32 TLMAX, bool SECOND>
__global__ __launch_bounds__(64) void k_u16_cprep(U16CArgs a)
{
    __shared__ __attribute__((aligned(16))) U16Lds<TLMAX> L;                // (the compressor's table log is clamped to the byte coder's limit, see below)
    const u32 lane = threadIdx.x;
    const size_t b = blockIdx.x;
    if (SECOND && a.meta[b].state != U16_WIDE) return;                       // uniform
    const u16* const src = (const u16*)((const u8*)a.src + b * a.srcStrideBytes);
    const size_t n = a.srcSizes ? a.srcSizes[b] : a.uniformSrcSize;
    const bool countOnly = a.countsOut != nullptr;
    U16Meta m; m.state = 0; m.hdrSize = 0; m.tableLog = 0; m.maxSV = 0;
    size_t result = 0; bool done = false;
    u32 maxSV = a.maxSVReq, tlReq = a.tableLogReq;
    if (!countOnly) {                                                        // fseU16.c:217-222
        if (n <= 1) { result = n; done = true; }
        else {
            if (!maxSV) maxSV = U16_MAXSV;
            if (!tlReq) tlReq = U16_DEFTL;
            if (maxSV > U16_MAXSV) { result = FERR(maxSymbolValue_tooLarge); done = true; }
            else if (tlReq > U16_MAXTL) { result = FERR(tableLog_tooLarge); done = true; }
        }
    } else if (maxSV > U16_MAXSV) { result = FERR(maxSymbolValue_tooLarge); done = true; }   // (our count array holds 287 entries)
    if (done) { if (lane == 0) { a.results[b] = result; if (a.meta) a.meta[b] = m; } return; }   // uniform

    // ---- FSE_countU16 (:121-146)
    // Four columns per symbol (col4[symbol][lane & 3], over the head of L, all of which is idle until the counts are known): the lanes of one LDS
    // pass that hold the same symbol spread over four words, and the source arrives in 16-byte loads, one 4 KiB group ahead of the updates
    // (hist.hip's scheme).  Round 4 counted with one 2-byte load per lane and step into a single column: 1.12 ms per 25k blocks of 32 KB,
    // more than the wave encoder behind it.
    static_assert(sizeof(L) >= U16_SYMS * 4 * sizeof(u32), "the count columns live in the builder's LDS");
    u32* const col4 = (u32*)&L;
    for (u32 i = lane; i < U16_SYMS; i += 64) ((uint4*)col4)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    u32 over = 0;
    {   const u32 col = lane & 3u;
        auto add = [&](u32 sy) { over |= sy > maxSV ? 1u : 0u; atomicAdd(&col4[((sy < U16_SYMS ? sy : U16_SYMS - 1u) << 2) | col], 1u); };   // (a symbol beyond the limit is an error: what it counts into is never read)
        auto add8 = [&](const uint4& v) { add(v.x & 0xFFFFu); add(v.x >> 16); add(v.y & 0xFFFFu); add(v.y >> 16); add(v.z & 0xFFFFu); add(v.z >> 16); add(v.w & 0xFFFFu); add(v.w >> 16); };
        size_t head = (size_t)(((0 - (uintptr_t)src) & 15u) >> 1);           // symbols in front of the first 16-byte boundary
        if (head > n) head = n;
        if (lane < head) add(src[lane]);
        const uint4* const v = (const uint4*)(src + head);
        const size_t nvec = (n - head) >> 3;
        size_t i = lane;
        if (i + 192 < nvec) {
            uint4 x0 = v[i], x1 = v[i + 64], x2 = v[i + 128], x3 = v[i + 192];
            for (;;) {
                const size_t j = i + 256;
                const bool more = j + 192 < nvec;
                uint4 y0 = x0, y1 = x1, y2 = x2, y3 = x3;
                if (more) { y0 = v[j]; y1 = v[j + 64]; y2 = v[j + 128]; y3 = v[j + 192]; }
                __asm__ volatile("" ::: "memory");                          // keep the next group's loads above this group's LDS updates
                add8(x0); add8(x1); add8(x2); add8(x3);
                i = j;
                if (!more) break;
                x0 = y0; x1 = y1; x2 = y2; x3 = y3;
            }
        }
        for (; i < nvec; i += 64) { const uint4 x = v[i]; add8(x); }
        const size_t done = head + (nvec << 3);
        if (lane < n - done) add(src[done + lane]);
    }
    __syncthreads();
    if (__any(over != 0)) { if (lane == 0) { a.results[b] = FERR(maxSymbolValue_tooSmall); if (a.meta) a.meta[b] = m; } return; }
    u32 c[U16_SPL], top = 0, big = 0;
#pragma unroll
    for (int i = 0; i < U16_SPL; ++i) {
        const u32 s = U16_SPL * lane + i;
        const uint4 q = ((const uint4*)col4)[s];
        c[i] = q.x + q.y + q.z + q.w;
        if (c[i]) top = s; big = c[i] > big ? c[i] : big;
    }
    __syncthreads();                                                         // (col4 is read: L becomes the builder's)
#pragma unroll
    for (int i = 0; i < U16_SPL; ++i) L.cnt[U16_SPL * lane + i] = c[i];
    for (u32 s = lane; s < U16_SYMS; s += 64) L.nrm[s] = 0;
    if (lane < 8) L.scal[lane] = 0;
    __syncthreads();
    top = wg_max<64>(top); big = wg_max<64>(big);
    if (n == 0) top = 0;
    if (countOnly) {
        for (u32 s = lane; s <= a.maxSVReq; s += 64) a.countsOut[b * (U16_MAXSV + 1) + s] = L.cnt[s];
        if (lane == 0) { a.maxSVOut[b] = top; a.results[b] = big; }
        return;
    }
    maxSV = top;
    if (big == n) { if (lane == 0) { a.results[b] = 1; a.meta[b] = m; } return; }    // one symbol only: the caller should use RLE (:228)

    // ---- table log, normalisation, header (:231-240)
    // FSE_optimalTableLog, FSE_normalizeCount and FSE_writeNCount are NOT re-instantiated for 16-bit symbols (fseU16.c:88 includes the
    // templates only): the calls at :231-240 reach the byte coder's objects, whose table-log limit is 12 -- a request of 13 passes
    // the check at :222 and is then clamped to 12.  (The decoder side does accept 13: FSE_buildDTableU16 is a template instance.)
    const u32 tl = wg_optimal_tablelog(tlReq, n, maxSV, 2, U16_DEFTL, FSE_MAX_TL);
    if (tl > TLMAX) { m.state = U16_WIDE; if (lane == 0) a.meta[b] = m; return; }   // uniform: the second launch's block
    int nn[U16_SPL];
    size_t e = wg_normalize<64, U16_SPL, FSE_MAX_TL>(nn, c, (u64)n, maxSV, tl, lane);
    if (is_err(e)) { if (lane == 0) { a.results[b] = e; a.meta[b] = m; } return; }
    for (u32 i = lane; i < 160; i += 64) L.img[i] = 0;
    __syncthreads();
    const size_t hdr = wg_write_ncount<64, U16_SPL>(L.img, a.dstCapacity, nn, maxSV, tl, lane);
    if (is_err(hdr)) { if (lane == 0) { a.results[b] = hdr; a.meta[b] = m; } return; }
    __syncthreads();
    u8* const dst = a.dst + b * a.dstStride;
    for (u32 i = lane; i < (u32)hdr; i += 64) dst[i] = ((const u8*)L.img)[i];
#pragma unroll
    for (int i = 0; i < U16_SPL; ++i) L.nrm[U16_SPL * lane + i] = (s16)nn[i];
    __syncthreads();

    // ---- FSE_buildCTable (lib/fse_compress.c:66-169 as instantiated at fseU16.c:103-111)
    u16* const st = a.stateTables + (b << U16_MAXTL);
    u32* const tt = a.symTT + b * 2 * (U16_MAXSV + 1);
    const u32 ts = 1u << tl;
    u16_spread_rank(L, maxSV, tl, lane, [&](u32 u, u32 s, u32 r) { st[(u32)L.cum[s] + r] = (u16)(ts + u); });
#pragma unroll
    for (int i = 0; i < U16_SPL; ++i) {                                     // the symbols' transforms (:131-154)
        const u32 s = U16_SPL * lane + i;
        if (s > maxSV) continue;
        const int v = nn[i];
        const u32 total = L.cum[s];
        u32 dnb, dfs = 0;
        if (v == 0) dnb = ((tl + 1u) << 16) - ts;
        else if (v == -1 || v == 1) { dnb = (tl << 16) - ts; dfs = total - 1u; }
        else { const u32 mbo = tl - hibit32((u32)(v - 1)); dnb = (mbo << 16) - ((u32)v << mbo); dfs = total - (u32)v; }
        tt[2 * s] = dfs; tt[2 * s + 1] = dnb;
    }
    // Encoder choice, as k_fse_cprep makes it for the byte coder: when one symbol holds more than 63/64 of the cells a lane's share of the output does not
    // span a byte, the wave encoder would find that out only after its warm-up, counting and link-repair work (k_u16_encode_wave, "tables that skewed")
    // and leave the block to the lane-per-block kernel -- such blocks go there at once.  (The wave kernel still re-checks exactly.)
    int top1 = 0;
#pragma unroll
    for (int i = 0; i < U16_SPL; ++i) top1 = nn[i] > top1 ? nn[i] : top1;
    top1 = wave_max_i32(top1);
    m.state = (u32)top1 * 64u > (63u << tl) ? U16_SKEWED : 1u; m.hdrSize = (u32)hdr; m.tableLog = tl; m.maxSV = maxSV;
    if (lane == 0) a.meta[b] = m;
}

// ---- FSE_compressU16_usingCTable (:150-200) split across the 64 lanes of a wave -------------------------------------------------
// The single chain "forgets" like the byte coder's two (fse_encode_wave.hip): lane t owns the t-th 64th of the symbols in emission
// order (last symbol first); pass 1 warms up in front of its range from an arbitrary state, takes the state it arrives with as its
// speculated start, runs its range counting bits and keeps its end state; a lane whose start differs from its predecessor's end
// re-runs from there until no link changes (lane 0 starts from FSE_initCState's exact state, so verified links make every lane exact);
// a prefix sum places every lane's bits, the verdicts of BIT_closeCStream (bitstream.h:254-260) and FSE_compressU16 (:252-253) are
// known before a bit is written, and pass 2 emits through per-lane LDS rings (round 5; the scheme of fse_encode_wave.hip's WvSink): a lane's
// completed words go to its 64-byte ring, every complete 32-byte aligned piece of its byte range leaves with two 16-byte stores (whole
// sectors: no read-modify-write at the memory side), the bytes in front of its first piece boundary and behind its last one go out
// bytewise / wordwise, and the byte it shares with its predecessor receives the predecessor's last bits with one atomic OR issued AFTER
// the lane's own store of that byte.  Nothing is zeroed and nothing is OR-ed into zeros any more: the zero-fill pass, 40 scattered
// 4-byte stores and two atomics per lane moved 301 KB per block at the memory side for 43 KB of input + output (round 4's counters).
// Blocks it finishes are marked U16_DONE; shorter ones are left to the lane-per-block kernel behind it.
#define U16_DONE 3u
#define U16_WAVE_MIN 2048u          // symbols: below this the launch of 64 lanes per block does not pay
#define U16_LINE 32u                // bytes per piece written to global memory
#define U16_RING (2u * U16_LINE)    // per-lane LDS ring
struct U16Sink {                    // bit sink of one lane (see above)
    u8* dstAl;                      // the payload address rounded down to U16_RING bytes: ring offsets = offsets from here mod U16_RING
    u32* ring;                      // this lane's LDS ring (U16_RING-aligned)
    u32 woff;                       // offset from dstAl of the word in progress (multiple of 4)
    u32 done;                       // offset from dstAl up to which this lane's bytes are in global memory
    u32 acc, nacc;                  // the bits not yet in a complete word (nacc < 32 between calls)
    DEV void open(u8* payload, u32 bit0, u32* myRing)
    {
        const u32 lead = (u32)((uintptr_t)payload & (U16_RING - 1));
        const u32 off0 = lead + (bit0 >> 3);                                 // my first byte
        dstAl = payload - lead; ring = myRing;
        woff = off0 & ~3u; done = off0; acc = 0; nacc = 8u * (off0 & 3u) + (bit0 & 7u);
    }
    DEV void put(u32 v, u32 nb)                                              // nb <= 32.  Branch-free: the word in progress is stored every time
    {
        const u64 sh = (u64)v << nacc;
        acc |= (u32)sh;
        nacc += nb;
        const bool full = nacc >= 32u;
        ring[(woff & (U16_RING - 1)) >> 2] = acc;
        woff += full ? 4u : 0u;
        acc = full ? (u32)(sh >> 32) : acc;
        nacc &= 31u;
    }
    DEV void copy_out(u32 upTo)                                              // bytes [done, upTo), any alignment
    {
        const u8* const rb = (const u8*)ring;
        u32 o = done;
        while (o < upTo) {
            if (((o & 3u) == 0) && o + 4 <= upTo) { const u32 w = ring[(o & (U16_RING - 1)) >> 2]; __builtin_memcpy(dstAl + o, &w, 4); o += 4; }
            else { dstAl[o] = rb[o & (U16_RING - 1)]; ++o; }
        }
        done = upTo;
    }
    DEV void line()                                                          // at least once per 32 bytes put: at most one piece completes between two calls
    {
        const u32 L = done & ~(U16_LINE - 1);
        if (woff >= L + U16_LINE) {
            if (done == L) {
                const uint4* const r4 = (const uint4*)(ring + ((L & (U16_RING - 1)) >> 2));
#pragma unroll
                for (u32 q = 0; q < U16_LINE / 16; ++q) { const uint4 v = r4[q]; __builtin_memcpy(dstAl + L + 16 * q, &v, 16); }
                done = L + U16_LINE;
            } else copy_out(L + U16_LINE);
        }
    }
    // the rest: every complete byte goes out; returns the < 8 bits that belong to the next lane's first byte (or, lastByte: pads them into a byte of mine)
    DEV u32 close(bool lastByte)
    {
        if (lastByte) nacc = (nacc + 7u) & ~7u;
        ring[(woff & (U16_RING - 1)) >> 2] = acc;
        copy_out(woff + (nacc >> 3));
        return (nacc & 7u) ? acc >> (nacc & ~7u) : 0u;
    }
};
__global__ __launch_bounds__(64) void k_u16_encode_wave(U16CArgs a)
{
    __shared__ u16 st[1u << FSEHIP_FSE_MAX_TABLELOG];                        // the encoder never picks a table log above 12 (see k_u16_cprep)
    __shared__ uint2 tt[U16_SYMS];
    __shared__ __attribute__((aligned(U16_RING))) u32 rings[64 * (U16_RING / 4)];   // one output ring per lane (pass 2)
    const u32 lane = threadIdx.x;
    const size_t b = blockIdx.x;
    const U16Meta m = a.meta[b];
    const size_t n64 = a.srcSizes ? a.srcSizes[b] : a.uniformSrcSize;
    if (m.state != 1 || n64 < U16_WAVE_MIN || n64 >= ((size_t)1 << 27) || m.tableLog > FSEHIP_FSE_MAX_TABLELOG) return;   // uniform
    {   const uintptr_t blockDst = (uintptr_t)(a.dst + b * a.dstStride);     // the aligned word holding the first payload byte must belong to this block
        if (((blockDst + m.hdrSize) & ~(uintptr_t)3) < blockDst) return; }
    const u32 n = (u32)n64, tl = m.tableLog, ts = 1u << tl;
    const u16* const src = (const u16*)((const u8*)a.src + b * a.srcStrideBytes);
    u32 presentLane = 0;
    {   const u32* const g = (const u32*)(a.stateTables + (b << U16_MAXTL));
        for (u32 i = lane; i < ts / 2; i += 64) ((u32*)st)[i] = g[i];
        const uint2* const gt = (const uint2*)(a.symTT + b * 2 * (U16_MAXSV + 1));
        for (u32 sy = lane; sy <= m.maxSV; sy += 64) { const uint2 e = gt[sy]; tt[sy] = e; presentLane += e.y != ((tl + 1u) << 16) - ts; }
    }
    __syncthreads();
    const u32 present = wg_sum<64>(presentLane);
    u32 warm = (4u << tl) / (present ? present : 1u);                        // two states fed the same symbols merge with probability ~ present / tableSize per step
    warm = warm < 64u ? 64u : (warm > 2048u ? 2048u : warm);
    auto step = [&](u32& x, u32 sym, u32& nb) { const uint2 e = tt[sym]; nb = (x + e.y) >> 16; const u32 low = x & ((1u << nb) - 1u); x = st[(x >> nb) + e.x]; return low; };
    // symbols j in [ja, jb) in emission order (j = 0 is the LAST symbol of the source): 32 per step = four 16-byte loads of one 64-byte
    // stretch issued together, one stretch ahead of its use (a load behind every symbol would cost a memory round trip per step; 16-byte
    // loads spread over time let the line leave the L2 between them: 206 KB were fetched per 32 KB block).  every8() runs after each
    // eight symbols.
    auto walk = [&](u32 ja, u32 jb, auto&& f, auto&& every8) {
        u32 j = ja;
#define U16_OCT(v) f(v.w >> 16); f(v.w & 0xFFFFu); f(v.z >> 16); f(v.z & 0xFFFFu); f(v.y >> 16); f(v.y & 0xFFFFu); f(v.x >> 16); f(v.x & 0xFFFFu); every8();
        if (j + 32u <= jb) {
            uint4 c0, c1, c2, c3;
            {   const u16* const q = src + (n - 32u - j);
                __builtin_memcpy(&c3, q + 24, 16); __builtin_memcpy(&c2, q + 16, 16); __builtin_memcpy(&c1, q + 8, 16); __builtin_memcpy(&c0, q, 16); }
            for (;;) {
                const u32 jn = j + 32u;
                const bool more = jn + 32u <= jb;
                uint4 n0 = c0, n1 = c1, n2 = c2, n3 = c3;
                if (more) {
                    const u16* const q = src + (n - 32u - jn);
                    __builtin_memcpy(&n3, q + 24, 16); __builtin_memcpy(&n2, q + 16, 16); __builtin_memcpy(&n1, q + 8, 16); __builtin_memcpy(&n0, q, 16);
                }
                __asm__ volatile("" ::: "memory");
                U16_OCT(c3) U16_OCT(c2) U16_OCT(c1) U16_OCT(c0)
                j = jn;
                if (!more) break;
                c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            }
        }
        while (j + 8u <= jb) { uint4 v; __builtin_memcpy(&v, src + (n - 8u - j), 16); U16_OCT(v) j += 8u; }
#undef U16_OCT
        if (j < jb) { for (; j < jb; ++j) f((u32)src[n - 1u - j]); every8(); }
    };
    auto count = [&](u32& x, u32 ja, u32 jb) { u32 bits = 0; walk(ja, jb, [&](u32 sym) { u32 nb; (void)step(x, sym, nb); bits += nb; }, [] {}); return bits; };
    const u32 C = (n + 63u) / 64u;
    const u32 j0 = lane * C < n ? lane * C : n, j1 = (lane + 1u) * C < n ? (lane + 1u) * C : n;
    const bool mine = j0 < j1;
    const u32 lastLane = (n - 1u) / C;
    // ---- pass 1
    u32 x = ts, start = ts, end = ts, bits = 0;
    if (mine) {
        if (j0 > warm) (void)count(x, j0 - warm, j0); else if (j0) (void)count(x, 0, j0);      // (close to the end of the source: exact from FSE_initCState)
        start = x; bits = count(x, j0, j1); end = x;
    }
    for (;;) {                                                               // ---- verification / repair
        const u32 prevEnd = (u32)__shfl_up((int)end, 1, WAVE);
        const bool bad = mine && lane > 0 && start != prevEnd;
        if (!__any(bad)) break;
        if (bad) { start = prevEnd; x = start; bits = count(x, j0, j1); end = x; }
    }
    // a lane without a whole byte of its own cannot take over the byte it shares with its predecessor (pass 2): tables that skewed are left
    // to the lane-per-block kernel behind this one (uniform; nothing has been written)
    if (__any(mine && bits < 8u)) return;
    if (lane == lastLane) bits += tl + 1u;                                    // FSE_flushCState and the end mark
    u32 incl = bits;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)incl, off, WAVE); if ((int)lane >= off) incl += o; }
    const u32 total = (u32)__shfl((int)incl, 63, WAVE), excl = incl - bits;
    // ---- verdicts
    u8* const payload = a.dst + b * a.dstStride + m.hdrSize;
    const size_t cap = a.dstCapacity - m.hdrSize;
    const size_t cs = (cap > 8 && (size_t)(total >> 3) < cap - 8) ? (size_t)((total + 7u) >> 3) : 0;
    const size_t sum = (size_t)m.hdrSize + cs, result = sum >= (n64 - 1) * 2 ? 0 : sum;
    if (cs) {
        // ---- pass 2: every lane emits its bits through its LDS ring (U16Sink); lanes without symbols have no bits and no bytes
        u32 tail = 0;
        if (mine) {
            U16Sink k; k.open(payload, excl, rings + lane * (U16_RING / 4));
            x = start;
            walk(j0, j1, [&](u32 sym) { u32 nb; const u32 low = step(x, sym, nb); k.put(low, nb); }, [&] { k.line(); });   // eight symbols: <= 13 bytes
            if (lane == lastLane) { k.put(x & (ts - 1u), tl); k.put(1u, 1u); }
            tail = k.close(lane == lastLane);
        }
        // the byte my range starts in (when it starts inside a byte) also holds the last bits of the lane in front of me: OR them in,
        // after my own store of that byte (program order of this lane)
        const u32 prevTail = (u32)__shfl_up((int)tail, 1, WAVE);
        if (mine && lane > 0 && (excl & 7u)) {
            const uintptr_t ad = (uintptr_t)(payload + (excl >> 3));
            atomicOr((u32*)(ad & ~(uintptr_t)3), prevTail << (8u * (u32)(ad & 3u)));
        }
    }
    if (lane == 0) { a.results[b] = result; a.meta[b].state = U16_DONE; }
}

// FSE_compressU16_usingCTable (:150-200) + the verdicts of FSE_compressU16 (:245-255) as the reference loops, one lane per block: short
// blocks (and whatever the wave kernel above leaves)
__global__ void k_u16_encode(U16CArgs a)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nBlocks) return;
    const U16Meta m = a.meta[b];
    if (m.state != 1 && m.state != U16_SKEWED) return;
    const u16* const src = (const u16*)((const u8*)a.src + b * a.srcStrideBytes);
    const size_t n = a.srcSizes ? a.srcSizes[b] : a.uniformSrcSize;
    const u16* const st = a.stateTables + (b << U16_MAXTL);
    const uint2* const tt = (const uint2*)(a.symTT + b * 2 * (U16_MAXSV + 1));
    u8* const dst = a.dst + b * a.dstStride + m.hdrSize;
    const size_t cap = a.dstCapacity - m.hdrSize;
    const u32 tl = m.tableLog;
    size_t cs = 0;
    if (cap > 8) {
        const size_t lim = cap - 8;                                          // endPtr (bitstream.h:190)
        u64 acc = 0; u32 nacc = 0; size_t pos = 0;
        u32 x = 1u << tl;                                                    // FSE_initCState
        auto enc = [&](u32 sym) {                                            // FSE_encodeSymbol (fse.h:514-521)
            const uint2 e = tt[sym];
            const u32 nb = (x + e.y) >> 16;
            acc |= (u64)(x & ((1u << nb) - 1u)) << nacc; nacc += nb;
            x = st[(x >> nb) + e.x];
        };
        auto flush = [&]() {                                                 // BIT_flushBits (bitstream.h:239-249)
            __builtin_memcpy(dst + pos, &acc, 8);
            const u32 nby = nacc >> 3;
            pos += nby; pos = pos > lim ? lim : pos;
            acc = nby >= 8 ? 0 : acc >> (8 * nby); nacc &= 7u;
        };
        size_t ip = n;
        if (n & 1) { enc(src[--ip]); flush(); }
        if (n & 2) { enc(src[ip - 1]); enc(src[ip - 2]); ip -= 2; flush(); }
        while (ip > 0) {                                                     // four symbols (<= 52 bits) per flush
            u64 w; __builtin_memcpy(&w, src + ip - 4, 8);
            enc((u32)(w >> 48)); enc((u32)(w >> 32) & 0xFFFFu); enc((u32)(w >> 16) & 0xFFFFu); enc((u32)w & 0xFFFFu);
            ip -= 4; flush();
        }
        acc |= (u64)(x & ((1u << tl) - 1u)) << nacc; nacc += tl; flush();    // FSE_flushCState
        acc |= (u64)1 << nacc; nacc += 1; flush();                           // BIT_closeCStream
        cs = pos >= lim ? 0 : pos + (nacc > 0);
    }
    const size_t total = (size_t)m.hdrSize + cs;
    a.results[b] = total >= (n - 1) * 2 ? 0 : total;                         // "no compression" (:252-253)
}

// ---- decompress side ------------------------------------------------------------------------------------------------------------
// The NCount header is a serial variable-length code (ncount_reader.h): one LANE per block reads it -- 64 headers per wave -- into a row of
// LDS, and the rows leave as coalesced 640-byte pieces into the head of the block's table slot, where the builder picks them up.  (Until round
// 5 lane 0 of the builder's wave read the header while 63 lanes waited: 0.22 of the 0.49 ms k_u16_dprep took per 25k blocks.)
#define U16_PARSED 5u
#define U16_NRM_ROW (U16_SYMS + 2)                  // s16 per LDS row: 161 dwords, odd, so the 64 lanes' rows start on different banks
__global__ __launch_bounds__(64) void k_u16_dparse(U16DArgs a)
{
    __shared__ s16 rows[64 * U16_NRM_ROW];
    const u32 lane = threadIdx.x;
    const size_t b0 = (size_t)blockIdx.x * 64, b = b0 + lane;
    for (u32 i = lane; i < 64 * U16_NRM_ROW / 2; i += 64) ((u32*)rows)[i] = 0;
    __syncthreads();
    if (b < a.nBlocks) {                                                     // fseU16.c:316-325
        const u8* const in = a.csrc + b * a.cStride;
        const size_t cSize = a.cSizes ? a.cSizes[b] : a.uniformCSize;
        U16Meta m; m.state = 0; m.hdrSize = 0; m.tableLog = 0; m.maxSV = 0;
        size_t r = 0; u32 maxSV = U16_MAXSV, tl = 0;
        if (cSize < 2) r = FERR(srcSize_wrong);
        else {
            r = ncount_read<1>(rows + lane * U16_NRM_ROW, &maxSV, &tl, in, cSize);
            if (!is_err(r) && tl > U16_MAXTL) r = FERR(tableLog_tooLarge);   // FSE_buildDTable's check (fse_decompress.c:83)
            if (!is_err(r) && r >= cSize) r = FERR(srcSize_wrong);           // nothing behind the header (undefined in the reference)
        }
        if (is_err(r)) a.results[b] = r;
        else { m.state = U16_PARSED; m.hdrSize = (u32)r; m.tableLog = tl; m.maxSV = maxSV; }
        a.meta[b] = m;
    }
    __syncthreads();
    const size_t nRows = a.nBlocks - b0 < 64 ? a.nBlocks - b0 : 64;
    for (size_t k = 0; k < nRows; ++k) {                                     // (rows of headers that failed are written too: nobody reads them)
        u32* const out = a.cells + ((b0 + k) << U16_MAXTL);
        const u32* const row = (const u32*)(rows + k * U16_NRM_ROW);
        for (u32 i = lane; i < U16_SYMS / 2; i += 64) out[i] = row[i];
    }
}

template <u32 TLMAX>
__global__ __launch_bounds__(64) void k_u16_dprep(U16DArgs a)
{
    __shared__ U16Lds<TLMAX> L;
    const u32 lane = threadIdx.x;
    const size_t b = blockIdx.x;
    U16Meta m = a.meta[b];
    if (m.state != U16_PARSED || m.tableLog > TLMAX) return;                 // uniform: nothing to build, or the wide instance's block (second launch)
    {   const u32* const nrmIn = a.cells + (b << U16_MAXTL);                 // the counters k_u16_dparse left (zero beyond the header's last symbol)
        for (u32 i = lane; i < U16_SYMS / 2; i += 64) ((u32*)L.nrm)[i] = nrmIn[i];
    }
    __syncthreads();                                                         // (every lane has read the slot's head: the tables may overwrite it)
    const size_t r = m.hdrSize;
    const u32 maxSV = m.maxSV, tl = m.tableLog, ts = 1u << tl;
    // Two table formats in the block's 32 KiB slot.  Table logs up to 12 (all the reference's compressor writes): the CHAIN cells
    // newState | nbBits << 12 as 16-bit words (the image k_u16_decode_lds keeps in LDS) followed, 16 KiB further on, by the 9-bit symbols
    // of the cells, packed (gathered by its service waves) -- state 1.  Table log 13 needs 17 bits per chain cell: one 32-bit word per cell
    // for the lane-per-block kernel -- state 3.
    u32* const cells = a.cells + (b << U16_MAXTL);
    u16* const cells16 = (u16*)cells; u32* const syms9 = (u32*)(cells16 + ((size_t)1 << U16_MAXTL));
    const bool wide = tl > 12;
    u16_spread_rank(L, maxSV, tl, lane, [&](u32 u, u32 s, u32 rk) {       // FSE_buildDTable (fse_decompress.c:116-123)
        const int v = L.nrm[s];
        const u32 next = (v == -1 ? 1u : (u32)v) + rk;
        const u32 nb = tl - hibit32(next);
        const u32 ns = ((next << nb) - ts) & 0xFFFFu;
        if (wide) cells[u] = ns | (nb << 16) | (s << 20);
        else cells16[u] = (u16)(ns | (nb << 12));                              // (fse_u16_decode.hip: U16D_LOG)
    });
    // The symbols of the cells, for the decoder's service waves to gather: 9 bits each, packed (cell u at bit 9u: 2.3 KB at table log 11).  As
    // 16-bit words (4 KB) the tables of the 33 blocks a CU decodes at once did not fit the 4 MB of L2 an XCD's 32 CUs share: 294 KB per block
    // were fetched from beyond it (round 5 counters), 40 KB now.  A lane packs 32 cells into nine words.
    if (!wide) {
        __syncthreads();
        for (u32 j = lane; j < (ts >> 5); j += 64) {                         // (table log 12: two rounds)
            const u16* const sy = L.symTab + 32u * j;
            u32* const out = syms9 + 9u * j;
            unsigned long long acc = 0; u32 nacc = 0, w = 0;
#pragma unroll
            for (u32 i = 0; i < 32; ++i) {
                acc |= (unsigned long long)sy[i] << nacc; nacc += 9;
                if (nacc >= 32) { out[w++] = (u32)acc; acc >>= 32; nacc -= 32; }
            }
        }
    }
    m.state = wide ? 3u : 1u; m.hdrSize = (u32)r; m.tableLog = tl; m.maxSV = maxSV;
    if (lane == 0) a.meta[b] = m;
}

// FSE_decompressU16_usingDTable (:273-301): one lane per block, the reference's loop with its bit reader -- the blocks with a table
// log of 13 (32-bit cells in global memory; everything else is k_u16_decode_lds's, fse_u16_decode.hip)
__global__ void k_u16_decode(U16DArgs a)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nBlocks) return;
    const U16Meta m = a.meta[b];
    if (m.state != 3) return;
    const u8* const in = a.csrc + b * a.cStride + m.hdrSize;
    const size_t size = (a.cSizes ? a.cSizes[b] : a.uniformCSize) - m.hdrSize;
    const u32* const cells = a.cells + (b << U16_MAXTL);
    u16* const out = (u16*)((u8*)a.dst + b * a.dstStrideBytes);
    const size_t cap = a.dstCapacity;
    BitReader r;
    (void)r.init(in, size);                                                  // (the verdict of BIT_initDStream is not looked at, :284)
    u32 state = r.read(m.tableLog); (void)r.reload();                        // FSE_initDState
    size_t op = 0;
    auto step = [&]() {                                                      // FSE_decodeSymbolU16 (:262-271)
        const u32 c = cells[state];
        const u32 low = r.read((c >> 16) & 15u);
        state = (c & 0xFFFFu) + low;
        return (u16)(c >> 20);
    };
    // (Measured: walking the bulk with a lazily refilled window instead of a reload per symbol is SLOWER, 15.1 vs 12.1 ms per 25k blocks --
    // the kernel is bound by the random 4-byte table reads, 26 GB of 64-byte sectors per 25k blocks, not by the chain's length.)
    while (r.reload() < BR_COMPLETED && op < cap) out[op++] = step();
    size_t result;
    if (!(r.at == 0 && r.used == 64)) result = FERR(corruption_detected);   // BIT_endOfDStream
    else {
        while (state && op < cap) out[op++] = step();
        result = state ? FERR(corruption_detected) : op;
    }
    a.results[b] = result;
}

// ---- launchers --------------------------------------------------------------------------------------------------------------------
hipError_t launch_u16_compress(const U16CArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    hipLaunchKernelGGL((k_u16_cprep<11, false>), dim3((unsigned)a.nBlocks), dim3(64), 0, s, a);
    if (!a.countsOut) {
        hipLaunchKernelGGL((k_u16_cprep<FSE_MAX_TL, true>), dim3((unsigned)a.nBlocks), dim3(64), 0, s, a);   // table log 12 (returns at once otherwise)
        hipLaunchKernelGGL(k_u16_encode_wave, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_u16_encode, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_u16_decompress(const U16DArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_u16_dparse, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_u16_dprep<11>, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a);         // 8.6 KB of LDS: 18 waves per CU (the builders are latency-bound)
    hipLaunchKernelGGL(k_u16_dprep<12>, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a);         // 12.7 KB: 12 waves; returns at once for what is built
    hipLaunchKernelGGL(k_u16_dprep<U16_MAXTL>, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a);   // table log 13 (returns at once otherwise)
    const hipError_t e = launch_u16_decode_lds(a, s);                     // table logs up to 12: chain cells in LDS
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_u16_decode, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a);   // table log 13
    return hipGetLastError();
}
