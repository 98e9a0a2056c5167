// huf_prep.hip -- everything around the Huff0 hot loops, on the device (SURVEY 8(a') rows g5-g7):
//   compress side   : early outs of HUF_compress_internal, table log, code lengths (sort, tree, height limit), canonical codes,
//                     weights header incl. its small FSE coder      (behaviour of lib/huf_compress.c:637-724, :202-410, :63-147)
//   decompress side : raw / RLE decisions of HUF_decompress, weights header reader, X1 decoding table
//                     (behaviour of lib/huf_decompress.c:1056-1066, :118-185; lib/entropy_common.c:154-215)
//
// Mapping.  The work per block is a mix of wide steps (sorting up to 256 counts, ranking symbols, filling a table) and
// strictly serial ones (the two-queue Huffman merge, the tANS coding of the weights, bit parsers).  A lone wave issues
// one instruction every ~7 cycles however many of its lanes are active, so a serial step costs a whole wave-instruction
// per block unless several blocks share the wave.  Hence: one 64-lane wave looks after G = 2..4 blocks; the wide steps run
// over the blocks one after another with all 64 lanes (lane l: symbols 4l .. 4l+3, or cell / visit l), the serial steps
// run for the G blocks at once, lane g on block g.  Everything lives in registers and in the block's LDS slot
// (about 3.7 KiB: sorted keys, internal-node counts, parent links, lengths, the small FSE tables, the header image);
// global memory sees the counts once (coalesced) and the CElt table / header / X1 table once (coalesced).
//
// Restated pieces (see scripts/sim/huf_glue_sim.py for the CPU model checked against the compiled reference):
//   * order: keys count << 9 | 1 << 8 | 255 - symbol sorted descending by a bitonic network in registers = the reference's
//     order (count descending, ties in symbol order);
//   * tree: the two-queue merge on the sorted leaves (ties prefer the internal queue), recording parent links only; depths by
//     chasing the links from the leaves, 64 leaves at a time;
//   * height limit: lengths are monotone along the sorted leaves, so classes of equal length are contiguous runs and the
//     repair is a walk over the run boundaries (`last[k]` = last leaf of length limit - k): overlong leaves are cut to the
//     limit, the Kraft debt is repaid by moving boundary leaves one class down, an over-payment by moving the first leaves of
//     the limit class one class up;
//   * codes: start value per length by the usual recurrence over <= 12 lengths; value of a symbol = start + its rank among
//     the symbols of the same length, by wave ballots;
//   * header: weight histogram by ballots, the wave-level normalisation / NCount writer of wave_glue.h, a <= 64-state tANS
//     table built one spread visit per lane, the weights coded serially (lane g).
#include "internal.h"
#include "wave_glue.h"
#include "ncount_reader.h"
#include "bitreader.h"

#ifdef HP_TIMING            // development aid: per-phase cycle accounting of the first workgroups (scripts/hptiming.py)
__device__ unsigned long long g_hpTiming[2][2048 * 12];
extern "C" __attribute__((visibility("default"))) int FSEHIP_debug_hpTiming(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hpTiming), sizeof(g_hpTiming)); }
#define HPT_DECL unsigned long long hpT[12]; int hpN = 0; hpT[hpN++] = __builtin_readcyclecounter();
#define HPT_MARK hpT[hpN++] = __builtin_readcyclecounter();
#define HPT_DUMP(which) if (threadIdx.x == 0 && blockIdx.x < 2048) for (int q = 0; q + 1 < hpN; ++q) g_hpTiming[which][12 * blockIdx.x + q] = hpT[q + 1] - hpT[q];
#else
#define HPT_DECL
#define HPT_MARK
#define HPT_DUMP(which)
#endif
#define HUF_MAX_TL FSEHIP_HUF_TABLELOG_MAX
#define HUF_DEF_TL FSEHIP_HUF_TABLELOG_DEFAULT
// blocks per wave (template parameter G_ of the kernels): measured on MI355X, per 100k blocks P14 / P02 --
//   compress side  G = 1: 0.96 / 3.56 ms   2: 0.58 / 1.53   4: 0.76 / 1.54   8: 1.22 / 1.95   16: 2.64 / 3.70
//   decompress     G = 1: 1.17 / 1.37 ms   2: 0.61 / 0.74   4: 0.42 / 0.83   8: 0.64 / 1.14   16: 1.48 / 2.29
// (few blocks per wave = small LDS footprint = many waves per CU to hide the LDS / cross-lane latencies of the wide steps;
//  one block per wave leaves the serial steps with a single busy lane)
#ifndef HP_G_COMPRESS
#define HP_G_COMPRESS 2
#endif
#ifndef HP_G_DECOMPRESS
#define HP_G_DECOMPRESS 4
#endif

// ---- LDS slot of one block (bytes) ------------------------------------------------------------------------------------
#define HP_KEYS   0                     // u32[256] sorted keys, rank order (count = key >> 9)
#define HP_ICNT   1024                  // u32[256] counts of the internal nodes, creation order; later: per-symbol coder entries of the weights
#define HP_PAR    2048                  // u8[512]  parent (internal-node index) of leaf rank r [r] and of internal node k [256 + k]
#define HP_NBRANK 2560                  // u8[256]  code length by rank
#define HP_NBSYM  2816                  // u8[256]  code length by symbol; decode side: weights by symbol
#define HP_ST     3072                  // u16[64]  small tANS table: next-state table (encode) / cells (decode: u32[64] spans ST and TT)
#define HP_TT     3200                  // u32[32]  small tANS table: per-symbol transforms
#define HP_CELL   3328                  // u8[64]   symbol of every cell of the small table
#define HP_NRM    3392                  // s16[16]  normalised counters of the weights (encode side)
#define HP_LAST   3424                  // u16[16]  height limit: last rank of every length class
#define HP_HDR    3456                  // u8[288]  header image (u32 aligned)
#define HP_SCAL   3744                  // u32[8]   per-block scalars
#define HP_SLOT   3776
// decode side reuses the slot: HP_KEYS region = s16[256] counters of the weights' NCount header + u8[256] symbols sorted by weight
enum { SC_STATE = 0, SC_MAXSV, SC_LOG, SC_LEAVES, SC_RESULT_LO, SC_RESULT_HI, SC_HDR, SC_AUX };

DEV unsigned long long hp_below_mask(u32 lane) { return (1ull << lane) - 1ull; }

// ---- bitonic sort, descending, of 64 * R keys held R per lane (element index = lane * R + r) ----------------------------
template <int R>
DEV void hp_sort_desc(u32 (&k)[R], u32 lane)
{
    constexpr u32 N = 64u * R;
#pragma unroll
    for (u32 size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (u32 j = size >> 1; j > 0; j >>= 1) {
            if (j >= (u32)R) {                                            // partner in another lane, same register
                const u32 lj = j / R;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const u32 i = lane * R + r;
                    const u32 o = lane_xor_any(k[r], lj, lane);                          // (lj is a constant of the unrolled loops: DPP / permlane, dev_common.h)
                    const bool keepMax = ((i & size) == 0) == ((i & j) == 0);
                    const u32 hi = k[r] > o ? k[r] : o, lo = k[r] > o ? o : k[r];
                    k[r] = keepMax ? hi : lo;
                }
            } else {                                                     // partner in this lane
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r & j) continue;
                    const u32 i = lane * R + r;
                    const bool desc = (i & size) == 0;
                    const u32 a = k[r], b = k[r | j];
                    const u32 hi = a > b ? a : b, lo = a > b ? b : a;
                    k[r] = desc ? hi : lo; k[r | j] = desc ? lo : hi;
                }
            }
        }
    }
}

// ---- small tANS tables (table log <= 6): one spread visit per lane ---------------------------------------------------------
// nrm(s) = counter of symbol s (uniform access).  Cell u of the table belongs to symbol cellSym[u]; `rank` = number of lower
// cells of the same symbol.  The spread visits cells (m * step) mod size for m = 0, 1, ...; the low-probability symbols
// (counter -1) take the top cells; the k-th visit that is not a top cell belongs to the symbol whose cumulative range holds k.
// Calls emit(lane's cell u, symbol, rank, slot = first slot of the symbol counting -1 as 1) for every cell, lane u = cell u.
template <class Nrm, class Emit>
DEV void hp_small_table(u8* cellSym, u32 maxSV, u32 tl, u32 lane, Nrm&& nrm, Emit&& emit)
{
    const u32 ts = 1u << tl, mask = ts - 1, step = (ts >> 1) + (ts >> 3) + 3;
    u32 nLow = 0;
    for (u32 s = 0; s <= maxSV; ++s) nLow += nrm(s) == -1;
    const int high = (int)ts - 1 - (int)nLow;
    const u32 u0 = (lane * step) & mask;
    const bool kept = lane < ts && (int)u0 <= high;
    const u32 k = (u32)__builtin_popcountll(__ballot(kept) & hp_below_mask(lane));
    u32 mine = 0, cum = 0, low = 0;
    for (u32 s = 0; s <= maxSV; ++s) {                                    // uniform walk over the alphabet
        const int n = nrm(s);
        if (n == -1) { if (lane == 0) cellSym[ts - 1 - low] = (u8)s; ++low; }
        else if (n > 0) { if (k >= cum) mine = s; cum += (u32)n; }
    }
    if (kept) cellSym[u0] = (u8)mine;
    __syncthreads();
    const bool cellOn = lane < ts;
    const u32 sy = cellOn ? cellSym[lane] : 0xFFFFu;
    u32 first = 0;
    for (u32 s = 0; s <= maxSV; ++s) {
        const int n = nrm(s);
        if (n == 0) continue;                                            // uniform
        const unsigned long long m = __ballot(cellOn && sy == s);
        if (cellOn && sy == s) emit(lane, s, (u32)__builtin_popcountll(m & hp_below_mask(lane)), first, n);
        first += n == -1 ? 1u : (u32)n;
    }
}

// ---- class counting with packed counters ---------------------------------------------------------------------------------
// Symbols fall into <= 14 classes (code lengths, or weights); what the table builders need is, per class, how many symbols it
// has and the rank of every symbol among its class mates in symbol order.  Fourteen 9-bit counters fit two 64-bit words, so one
// 64-bit pair scan over the lanes answers both for all classes at once (instead of a ballot per class).
struct Pk { u64 a, b; };                                    // classes 0..6 in a, 7..13 in b, 9 bits each (counts <= 256)
DEV void pk_add(Pk& p, u32 cls) { if (cls < 7) p.a += (u64)1 << (9 * cls); else p.b += (u64)1 << (9 * (cls - 7)); }
DEV u32 pk_get(const Pk& p, u32 cls) { return cls < 7 ? (u32)(p.a >> (9 * cls)) & 511u : (u32)(p.b >> (9 * (cls - 7))) & 511u; }
// cls[i] = class of symbol 4*lane + i (>= 14: not counted).  rank[i] = class mates in front of it; returns the class totals.
DEV Pk pk_rank(const u32 cls[4], u32 rank[4], u32 lane)
{
    Pk mine = { 0, 0 };
#pragma unroll
    for (int i = 0; i < 4; ++i) if (cls[i] < 14) pk_add(mine, cls[i]);
    Pk incl = mine;
#if FSEHIP_DPP_SCANS
    incl.a = group_scan_incl_add64<64>(incl.a, lane); incl.b = group_scan_incl_add64<64>(incl.b, lane);      // (dev_common.h: DPP instead of ds_bpermute)
#else
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 oa = (u64)__shfl_up((unsigned long long)incl.a, off, WAVE), ob = (u64)__shfl_up((unsigned long long)incl.b, off, WAVE);
        if ((int)lane >= off) { incl.a += oa; incl.b += ob; }
    }
#endif
    Pk run = { incl.a - mine.a, incl.b - mine.b };
#pragma unroll
    for (int i = 0; i < 4; ++i) { rank[i] = 0; if (cls[i] < 14) { rank[i] = pk_get(run, cls[i]); pk_add(run, cls[i]); } }
    Pk tot;
#if FSEHIP_DPP_SCANS
    tot.a = group_last64<64>(incl.a, lane); tot.b = group_last64<64>(incl.b, lane);
#else
    tot.a = (u64)__shfl((unsigned long long)incl.a, 63, WAVE); tot.b = (u64)__shfl((unsigned long long)incl.b, 63, WAVE);
#endif
    return tot;
}

// =====================================================================================================================
//  compress side
// =====================================================================================================================
// serial tANS coder of the weights (lane g): FSE_compress_usingCTable on the small table (lib/fse_compress.c:554-611);
// returns the payload size (0 = does not fit / not worth it) by BIT_closeCStream's rule (lib/bitstream.h:254-260).
// ent[s] = coder entry of symbol s's weight, prepared by the wide phase: low half (maxBitsOut << 8) - minStatePlus (so that
// nbBits = (state + low) >> 8: states are below 128), high half deltaFindState (signed).  The two chains are independent:
// one step of each per iteration, so that their table look-ups overlap.
DEV u32 hp_encode_weights(u8* out, long cap, const u32* ent, u32 n, const u16* st, u32 tl)
{
    if (n <= 2 || cap <= 8) return 0;
    u32 ch[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {                                        // FSE_initCState2 (fse.h:503-512): starts at minStatePlus, no bits
        const u32 e = ent[n - 1 - j];
        const int lo = (int)(s16)(e & 0xFFFFu), dfs = (int)e >> 16;
        const u32 mbo = (u32)(lo + 255) >> 8, msp = (mbo << 8) - (u32)lo;
        ch[j] = st[(int)(msp >> mbo) + dfs];
    }
    u64 acc = 0; u32 nacc = 0, pos = 0, total = 0;
#define HP_STEP(c, e)                                                                              \
    {   const u32 x = ch[c];                                                                       \
        const u32 nb = (u32)((int)x + (int)(s16)((e) & 0xFFFFu)) >> 8;                               \
        acc |= (u64)(x & ((1u << nb) - 1u)) << nacc; nacc += nb; total += nb;                      \
        ch[c] = st[(int)(x >> nb) + ((int)(e) >> 16)]; }
    u32 j = 2;
    for (; j + 1 < n; j += 2) {                                          // j even: chain 0 then chain 1
        const u32 e0 = ent[n - 1 - j], e1 = ent[n - 2 - j];
        HP_STEP(0, e0) HP_STEP(1, e1)
        while (nacc >= 8) { if ((long)pos < cap) out[pos] = (u8)acc; ++pos; acc >>= 8; nacc -= 8; }
    }
    if (j < n) { const u32 e0 = ent[n - 1 - j]; HP_STEP(0, e0) }
#undef HP_STEP
    const u32 c2 = (n & 1u) ? ch[1] : ch[0], c1 = (n & 1u) ? ch[0] : ch[1];
    acc |= (u64)(c2 & ((1u << tl) - 1u)) << nacc; nacc += tl;
    while (nacc >= 8) { if ((long)pos < cap) out[pos] = (u8)acc; ++pos; acc >>= 8; nacc -= 8; }
    acc |= (u64)(c1 & ((1u << tl) - 1u)) << nacc; nacc += tl;
    acc |= (u64)1 << nacc; nacc += 1;
    total += 2 * tl + 1;
    if ((long)(total >> 3) >= cap - 8) return 0;
    while (nacc > 0) { if ((long)pos < cap) out[pos] = (u8)acc; ++pos; acc >>= 8; nacc = nacc > 8 ? nacc - 8 : 0; }
    return (total + 7) >> 3;
}

// two-queue merge (lane g): leaves keys[0 .. L-1] descending; records parents only.  Branch-free picks (the lanes of the wave
// work on different blocks), both queue heads re-read after every pick (one LDS round trip).
DEV void hp_merge(const u32* keys, u32* icnt, u8* par, u32 L)
{
    const u32 NO_LEAF = 1u << 31, NO_NODE = 1u << 30;                     // exhausted / not yet created: never the smaller one
    int li = (int)L - 1;
    u32 ii = 0;
    u32 leaf = keys[li] >> 9, node = NO_NODE;
    for (u32 ni = 0; ni + 1 < L; ++ni) {
        u32 sum = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool takeLeaf = leaf < node;                             // ties go to the internal queue
            sum += takeLeaf ? leaf : node;
            par[takeLeaf ? (u32)li : 256u + ii] = (u8)ni;
            li -= takeLeaf ? 1 : 0; ii += takeLeaf ? 0u : 1u;
            const u32 lk = keys[li > 0 ? li : 0], nk = icnt[ii & 255u];
            leaf = li >= 0 ? lk >> 9 : NO_LEAF;
            node = ii < ni ? nk : NO_NODE;
        }
        icnt[ni] = sum;
        node = ii == ni ? sum : node;
    }
}

// height limit, repair part (lane g).  The wide phase has cut the overlong leaves to the limit M, computed the Kraft debt (in
// units of 2^-M) and the class boundaries: last[k] = last rank of length M - k (NONE if empty), n = last rank shorter than M.
DEV void hp_repair_lengths(u8* nb, const u32* keys, u16* last, int n, int debt, u32 M)
{
    const u32 NONE = 0xFFFFu;
    while (debt > 0) {
        u32 k = hibit32((u32)debt) + 1;                                   // a leaf leaving class k pays 2^(k-1)
        for (; k > 1; --k) {                                              // a cheaper class if its last leaf is rare enough
            const u32 hi = last[k], lo = last[k - 1];
            if (hi == NONE) continue;
            if (lo == NONE) break;
            if ((keys[hi] >> 9) <= 2 * (keys[lo] >> 9)) break;
        }
        while (k <= HUF_MAX_TL && last[k] == NONE) ++k;
        debt -= 1 << (k - 1);
        const u32 r = last[k];
        if (last[k - 1] == NONE) last[k - 1] = (u16)r;
        nb[r] = (u8)(nb[r] + 1);
        if (r == 0) last[k] = (u16)NONE;
        else last[k] = (u16)(nb[r - 1] == M - k ? r - 1 : NONE);
    }
    while (debt < 0) {                                                    // overpaid: the first leaves of the limit class move up
        if (last[1] == NONE) {
            while (nb[n] == M) --n;
            nb[n + 1] = (u8)(nb[n + 1] - 1); last[1] = (u16)(n + 1);
        } else {
            const u32 r = (u32)last[1] + 1;
            nb[r] = (u8)(nb[r] - 1); last[1] = (u16)r;
        }
        ++debt;
    }
}

enum { SC_DEBT = SC_RESULT_LO, SC_NLAST = SC_RESULT_HI };                 // (the result words are free until phase E)

// MODE 0: the one-shot path (counts from k_hist, everything between the histogram and the hot loop).
// MODE 1: HUF_buildCTable on the CALLER's counters (lib/huf_compress.c:334-409): phases A-E1 without the early outs of HUF_compress_internal and with
//         the caller's length limit as it is (no HUF_optimalTableLog); results[b] = the table log, as the reference returns it.
// MODE 2: HUF_writeCTable on the CALLER's table (lib/huf_compress.c:113-148): the code lengths come from a.ctables, phases E1 (weights) - G;
//         results[b] = the header size.
enum { HPM_ONESHOT = 0, HPM_BUILD = 1, HPM_WRITE = 2 };
template <int G_, int MODE = HPM_ONESHOT>
__global__ __launch_bounds__(64) void k_huf_cprep(HufCPrepArgs a)
{
    constexpr u32 HP_G = G_, HP_GL = 64 / G_;      // blocks per wave; lanes per block when all blocks are worked on at once
    extern __shared__ __attribute__((aligned(16))) u8 hpLds[];
    const u32 lane = threadIdx.x;
    const size_t b0 = (size_t)blockIdx.x * HP_G;
    HPT_DECL

    // ---- phase A (wide, block after block): early outs, table log, sorted keys.  All global loads are issued up front.
    uint4 cvAll[HP_G]; size_t topAll[HP_G]; u32 msvAll[HP_G];
#pragma unroll
    for (u32 g = 0; g < HP_G; ++g) {
        const size_t b = b0 + g < a.nBlocks ? b0 + g : a.nBlocks - 1;
        if (MODE == HPM_WRITE) cvAll[g] = ((const uint4*)(a.ctables + b * a.ctStrideU32))[lane];     // the caller's HUF_CElt entries of symbols 4*lane .. 4*lane+3
        else cvAll[g] = ((const uint4*)(a.counts + b * 256))[lane];
        topAll[g] = MODE == HPM_ONESHOT ? a.histResults[b] : 0; msvAll[g] = a.maxSVs[b];
    }
#pragma unroll
    for (u32 g = 0; g < HP_G; ++g) {
        u8* const slot = hpLds + g * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        const size_t b = b0 + g;
        if (b >= a.nBlocks) { if (lane == 0) sc[SC_STATE] = 0; continue; }  // uniform
        const size_t n = MODE == HPM_ONESHOT ? view_size(a.src, b) : 0;
        const u32 maxSV = msvAll[g];
        if (MODE == HPM_WRITE) {
            // HUF_writeCTable: the lengths of the caller's table instead of a tree; a length above huffLog indexes beyond the reference's
            // bitsToWeight[] (lib/huf_compress.c:127-130, symbols below maxSymbolValue only) and a huffLog above 12 overruns it: refused
            const uint4 cv = cvAll[g];
            const u32 e[4] = { cv.x, cv.y, cv.z, cv.w };
            const u32 huffLog = a.huffLogReq;
            size_t result = 0;
            u32 w4 = 0; bool over = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const u32 sy = 4 * lane + i, nbv = sy <= maxSV ? (e[i] >> 16) & 0xFFu : 0u; over |= sy < maxSV && nbv > huffLog; w4 |= nbv << (8 * i); }
            if (maxSV > 255u) result = FERR(maxSymbolValue_tooLarge);      // :123
            else if (huffLog > HUF_MAX_TL || __any(over)) result = FERR(GENERIC);
            if (result) { if (lane == 0) { sc[SC_STATE] = 0; a.results[b] = result; } continue; }      // uniform
            if (maxSV < 255u && lane == (maxSV >> 2)) w4 &= (0x100u << (8 * (maxSV & 3u))) - 1u;    // (nothing beyond maxSV; the last symbol's length is only used to be left out)
            ((u32*)(slot + HP_NBSYM))[lane] = w4;
            if (lane == 0) { sc[SC_STATE] = 1; sc[SC_MAXSV] = maxSV; sc[SC_LOG] = huffLog; sc[SC_LEAVES] = 0; }
            continue;
        }
        size_t result = 0; bool go = false;
        if (MODE == HPM_ONESHOT) {
            const size_t top = topAll[g];
            if (!n || !a.dstCapacity) result = 0;                              // huf_compress.c:656-657
            else if (n > FSEHIP_HUF_BLOCKSIZE_MAX) result = FERR(srcSize_wrong);
            else if (is_err(top)) result = top;
            else if (top == n) { if (lane == 0) a.dst[b * a.dstStride] = view_ptr(a.src, b)[0]; result = 1; }   // rle (:673)
            else if (top <= (n >> 7) + 4) result = 0;                          // not compressible enough (:674)
            else go = true;
        } else go = true;
        const uint4 cv = cvAll[g];
        u32 k[4] = { cv.x, cv.y, cv.z, cv.w };
        u32 huffLog;
        if (MODE == HPM_BUILD) {
            // HUF_buildCTable_wksp: the caller's limit as it is (0 = default, lib/huf_compress.c:349); the sort keys hold 23 bits of count
            bool big = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) big |= 4 * lane + i <= maxSV && k[i] >= (1u << 23);
            huffLog = a.huffLogReq ? a.huffLogReq : HUF_DEF_TL;
            if (maxSV > 255u) { result = FERR(maxSymbolValue_tooLarge); go = false; }   // :350
            else if (__any(big)) { result = FERR(GENERIC); go = false; }
        } else huffLog = go ? wg_optimal_tablelog(a.huffLogReq ? a.huffLogReq : HUF_DEF_TL, n, maxSV, 1) : 0;   // :691
        u32 present = 0;
        if (go) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const u32 s = 4 * lane + i; k[i] = (s <= maxSV && k[i]) ? (k[i] << 9) | (1u << 8) | (255u - s) : 0u; }
#pragma unroll
            for (int i = 0; i < 4; ++i) present += k[i] != 0;
            present = wg_sum<64>(present);
            // no symbol at all: the reference walks off the front of its node table (:357); more symbols than codes of the limit's length: its repair never ends
            if (MODE == HPM_BUILD && (present == 0 || (huffLog <= HUF_MAX_TL && present > (1u << huffLog)))) { result = FERR(GENERIC); go = false; }
        }
        if (!go) {
            if (lane == 0) {
                sc[SC_STATE] = 0; a.results[b] = result;
                if (MODE == HPM_ONESHOT) { HufMeta m; m.state = 0; m.hdrSize = 0; m.tableLog = 0; m.maxSV = 0; a.meta[b] = m; }
            }
            continue;
        }
        if (maxSV < 64) {                                                  // uniform: one key per lane is enough
            // symbols 0..63 sit four per lane in lanes 0..15: bring symbol `lane` to lane `lane`
            u32 one[1];
            const u32 src = lane >> 2;
            const u32 v0 = (u32)__shfl((int)k[0], (int)src, WAVE), v1 = (u32)__shfl((int)k[1], (int)src, WAVE);
            const u32 v2 = (u32)__shfl((int)k[2], (int)src, WAVE), v3 = (u32)__shfl((int)k[3], (int)src, WAVE);
            one[0] = (lane & 2u) ? ((lane & 1u) ? v3 : v2) : ((lane & 1u) ? v1 : v0);
            hp_sort_desc<1>(one, lane);
            ((u32*)(slot + HP_KEYS))[lane] = one[0];
        } else {
            hp_sort_desc<4>(k, lane);
            ((uint4*)(slot + HP_KEYS))[lane] = make_uint4(k[0], k[1], k[2], k[3]);
        }
        if (lane == 0) { sc[SC_STATE] = 1; sc[SC_MAXSV] = maxSV; sc[SC_LOG] = huffLog; sc[SC_LEAVES] = present; }
    }
    __syncthreads();
    HPT_MARK

    // ---- phase B (serial, lane g on block g): the tree
    if (MODE != HPM_WRITE && lane < HP_G) {
        u8* const slot = hpLds + lane * HP_SLOT;
        const u32* const sc = (const u32*)(slot + HP_SCAL);
        if (sc[SC_STATE] && (MODE == HPM_ONESHOT || sc[SC_LEAVES] >= 2)) hp_merge((const u32*)(slot + HP_KEYS), (u32*)(slot + HP_ICNT), slot + HP_PAR, sc[SC_LEAVES]);
    }
    __syncthreads();
    HPT_MARK

    // ---- phase C (wide): leaf depths by chasing the parent links; when the tree is too high, the cut to the limit, its debt
    //      and the class boundaries for the repair
    if (MODE != HPM_WRITE)
    for (u32 g = 0; g < HP_G; ++g) {
        u8* const slot = hpLds + g * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        if (!sc[SC_STATE]) continue;                                       // uniform
        const u32 L = sc[SC_LEAVES], root = L - 2, M = sc[SC_LOG];
        if (MODE == HPM_BUILD && L < 2) {                                  // uniform: one symbol in use -- the reference's walk gives it one bit (lib/huf_compress.c:357-378 with nonNullRank 0)
            if (lane == 0) { slot[HP_NBRANK] = 1; sc[SC_LOG] = 1; }
            continue;
        }
        const u8* const par = slot + HP_PAR;
        u32 p[4], d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const u32 r = lane + 64u * i; d[i] = 1; p[i] = r < L ? par[r] : root; }
        for (;;) {
            bool more = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (p[i] != root) { p[i] = par[256 + p[i]]; ++d[i]; more = true; }
            if (!__any(more)) break;
        }
        u32 deepest = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const u32 r = lane + 64u * i; if (r < L) deepest = d[i] > deepest ? d[i] : deepest; }
        const u32 largest = wg_max<64>(deepest);                           // = depth of the last leaf
        int debt = 0;
        if (MODE == HPM_BUILD && M > HUF_MAX_TL && largest > HUF_MAX_TL) { // uniform: HUF_setMaxHeight returns min(largest, limit), and what exceeds HUF_TABLELOG_MAX is refused (:385)
            if (lane == 0) { sc[SC_STATE] = 0; a.results[b0 + g] = FERR(GENERIC); }
            continue;
        }
        if (largest > M) {                                                 // uniform
            // classes of equal length are contiguous runs of ranks in ascending length: class counts give the boundaries
            Pk cnt = { 0, 0 };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 r = lane + 64u * i;
                if (r >= L) continue;
                if (d[i] > M) { debt += (int)((1u << (largest - M)) - (1u << (largest - d[i]))); d[i] = M; }
                pk_add(cnt, d[i]);
            }
            debt = (int)wg_sum<64>((u32)debt) >> (largest - M);
            cnt.a = wg_sum64<64>(cnt.a); cnt.b = wg_sum64<64>(cnt.b);
            u16* const last = (u16*)(slot + HP_LAST);
            u32 upTo = 0, shorter = 0;                                     // ranks with length <= len / < M
            for (u32 len = 1; len <= M; ++len) {                           // uniform
                const u32 c = pk_get(cnt, len);
                upTo += c;
                if (len < M) { shorter = upTo; if (lane == 0) last[M - len] = (u16)(c ? upTo - 1 : 0xFFFFu); }
            }
            if (lane == 0) { last[0] = 0xFFFFu; for (u32 k = M; k < HUF_MAX_TL + 2; ++k) last[k] = 0xFFFFu; sc[SC_NLAST] = shorter - 1u; sc[SC_DEBT] = (u32)debt; sc[SC_LOG] = M | 0x100u; }
        } else if (lane == 0) sc[SC_LOG] = largest;                        // the tree fits: its height is the table log
#pragma unroll
        for (int i = 0; i < 4; ++i) { const u32 r = lane + 64u * i; if (r < L) slot[HP_NBRANK + r] = (u8)d[i]; }
    }
    __syncthreads();
    HPT_MARK

    // ---- phase D (serial): repair of the cut lengths
    if (MODE != HPM_WRITE && lane < HP_G) {
        u8* const slot = hpLds + lane * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        if (sc[SC_STATE] && (sc[SC_LOG] & 0x100u)) {
            const u32 M = sc[SC_LOG] & 0xFFu;
            hp_repair_lengths(slot + HP_NBRANK, (const u32*)(slot + HP_KEYS), (u16*)(slot + HP_LAST), (int)sc[SC_NLAST], (int)sc[SC_DEBT], M);
            sc[SC_LOG] = M;
        }
    }
    __syncthreads();
    HPT_MARK

    // ---- phase E1 (wide): lengths by symbol, canonical values, CElt table; weight statistics into the lanes of group g
    u32 cw[4] = { 0, 0, 0, 0 };                                            // lanes 8g .. 8g+7: counts of weights 4*sub .. 4*sub+3 of block g
    u32 wTop = 0, wMax = 0;
    for (u32 g = 0; g < HP_G; ++g) {
        u8* const slot = hpLds + g * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        if (!sc[SC_STATE]) continue;                                       // uniform
        const size_t b = b0 + g;
        const u32 L = sc[SC_LEAVES], maxSV = sc[SC_MAXSV], huffLog = sc[SC_LOG];
        u8* const nbSym = slot + HP_NBSYM;
        if (MODE != HPM_WRITE) ((u32*)nbSym)[lane] = 0;                    // (HUF_writeCTable: the caller's lengths are there since phase A)
        {   u32* const img = (u32*)(slot + HP_HDR); for (u32 i = lane; i < 72; i += 64) img[i] = 0; }
        __syncthreads();
        if (MODE != HPM_WRITE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 r = lane + 64u * i;
            if (r < L) nbSym[255u - (((const u32*)(slot + HP_KEYS))[r] & 255u)] = slot[HP_NBRANK + r];
        }
        }
        __syncthreads();
        const u32 w4 = ((const u32*)nbSym)[lane];                          // lengths of symbols 4*lane .. 4*lane+3
        u32 nb[4], cls[4], rank[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { nb[i] = (w4 >> (8 * i)) & 0xFFu; cls[i] = 4 * lane + i <= maxSV ? nb[i] : 15u; }
        // per length: how many symbols and my rank among them in symbol order (absent symbols, length 0, are numbered too, as
        // the reference does); start values by the usual recurrence (huf_compress.c:394-400)
        const Pk tot = pk_rank(cls, rank, lane);
        u32 val[4] = { rank[0], rank[1], rank[2], rank[3] };
        {   u32 carry = 0;
            for (u32 len = huffLog; len >= 1; --len) {                     // uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) if (nb[i] == len) val[i] += carry;
                carry = (carry + pk_get(tot, len)) >> 1;
            }
        }
        if (MODE != HPM_WRITE) {
            u32 e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = 4 * lane + i <= maxSV ? ((val[i] & 0xFFFFu) | (nb[i] << 16)) : 0u;   // zeroed beyond maxSV (:697-699)
            ((uint4*)(a.ctables + b * a.ctStrideU32))[lane] = make_uint4(e[0], e[1], e[2], e[3]);
        }
        if (MODE == HPM_BUILD) { if (lane == 0) a.results[b] = huffLog; continue; }          // HUF_buildCTable is done: it returns the table log (:408)
        // weights (symbols 0 .. maxSV-1; the last one is implied): weight v has the symbols of length huffLog + 1 - v, less the
        // last symbol; statistics for the small FSE coder (huf_compress.c:63-103)
        const u32 lastNb = (u32)nbSym[maxSV];
        u32 top = 0, mw = 0;
        for (u32 v = 0; v <= huffLog; ++v) {                               // uniform
            const u32 len = v ? huffLog + 1 - v : 0;
            const u32 c = pk_get(tot, len) - (len == lastNb ? 1u : 0u);
            if (c) mw = v;
            top = c > top ? c : top;
        }
        if ((lane / HP_GL) == g) {
            const u32 sub = lane % HP_GL;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 v = 4 * sub + i;
                const u32 len = v ? huffLog + 1 - v : 0;
                cw[i] = v <= huffLog ? pk_get(tot, len) - (len == lastNb ? 1u : 0u) : 0u;
            }
            wTop = top; wMax = mw;
        }
    }
    __syncthreads();
    HPT_MARK
    if (MODE == HPM_BUILD) return;

    // ---- phase E2 (all blocks of the wave at once, 64 / G lanes per block): table log, counters and NCount header of the weights
    {   const u32 g = lane / HP_GL, sub = lane % HP_GL;
        u8* const slot = hpLds + g * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        if (sc[SC_STATE]) {                                                // uniform per group
            const u32 wtSize = sc[SC_MAXSV];
            u32 hs = 0, wtl = 0, ncBytes = 0; bool tableReady = false; size_t err = 0;
            if (wtSize > 1) {
                if (wTop == wtSize) hs = 1;                                // one weight only: not worth coding
                else if (wTop == 1) hs = 0;
                else {
                    wtl = wg_optimal_tablelog(6, wtSize, wMax, 2);
                    int nn[4];
                    const long wcap = (long)a.dstCapacity - 1;             // HUF_compressWeights(op + 1, maxDstSize - 1, ...)
                    size_t e = wg_normalize<HP_GL>(nn, cw, (u64)wtSize, wMax, wtl, lane);
                    if (!is_err(e)) e = wg_write_ncount<HP_GL>((u32*)(slot + HP_HDR) + 1, (size_t)(wcap < 0 ? 0 : wcap), nn, wMax, wtl, lane);   // image byte 4.. = header byte 1..
                    if (is_err(e)) err = e;
                    else {
                        ncBytes = (u32)e; tableReady = true;
                        if (sub < 4) *(uint2*)(slot + HP_NRM + 8 * sub) = make_uint2(((u32)nn[0] & 0xFFFFu) | ((u32)nn[1] << 16), ((u32)nn[2] & 0xFFFFu) | ((u32)nn[3] << 16));
                    }
                }
            }
            if (sub == 0) { sc[SC_HDR] = hs; sc[SC_AUX] = (tableReady ? 1u : 0u) | (wtl << 8) | (ncBytes << 16) | (wMax << 24); sc[SC_RESULT_LO] = (u32)err; sc[SC_RESULT_HI] = (u32)(err >> 32); }
        }
    }
    __syncthreads();
    HPT_MARK

    // ---- phase E3 (wide): the small tANS table of the weights and the per-symbol coder entries
    for (u32 g = 0; g < HP_G; ++g) {
        u8* const slot = hpLds + g * HP_SLOT;
        const u32* const sc = (const u32*)(slot + HP_SCAL);
        if (!sc[SC_STATE] || !(sc[SC_AUX] & 1u)) continue;                 // uniform
        const u32 wtl = (sc[SC_AUX] >> 8) & 0xFFu, maxW = (sc[SC_AUX] >> 24) & 0xFFu, ts = 1u << wtl;
        const u32 maxSV = sc[SC_MAXSV], huffLog = sc[SC_LOG];
        const s16* const nrm = (const s16*)(slot + HP_NRM);
        u16* const st = (u16*)(slot + HP_ST);
        u32* const tt = (u32*)(slot + HP_TT);
        hp_small_table(slot + HP_CELL, maxW, wtl, lane, [&](u32 s) { return (int)nrm[s]; },
                       [&](u32 u, u32 s, u32 r, u32 first, int n) { (void)s; (void)n; st[first + r] = (u16)(ts + u); });
        // per-weight transforms (lib/fse_compress.c:136-166) packed for the serial coder, lane w <= maxW
        if (lane <= maxW) {
            int total = 0;
            for (u32 s = 0; s < lane; ++s) { const int n = nrm[s]; total += n == -1 ? 1 : n; }
            const int n = nrm[lane];
            int dfs = 0; u32 mbo = wtl + 1, msp = ts;                      // absent weight: never used
            if (n == -1 || n == 1) { dfs = total - 1; mbo = wtl; msp = ts; }
            else if (n > 1) { mbo = wtl - hibit32((u32)n - 1); dfs = total - n; msp = (u32)n << mbo; }
            tt[lane] = (((mbo << 8) - msp) & 0xFFFFu) | ((u32)dfs << 16);
        }
        __syncthreads();
        {   const u32 w4 = ((const u32*)(slot + HP_NBSYM))[lane];
            u32 e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const u32 nbv = (w4 >> (8 * i)) & 0xFFu; e[i] = tt[nbv ? huffLog + 1 - nbv : 0]; }
            if (4 * lane < maxSV) ((uint4*)(slot + HP_ICNT))[lane] = make_uint4(e[0], e[1], e[2], e[3]);
        }
    }
    __syncthreads();
    HPT_MARK

    // ---- phase F (serial): the weights through the small coder
    if (lane < HP_G) {
        u8* const slot = hpLds + lane * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        if (sc[SC_STATE] && (sc[SC_AUX] & 1u)) {
            const u32 wtl = (sc[SC_AUX] >> 8) & 0xFFu, ncBytes = (sc[SC_AUX] >> 16) & 0xFFu;
            const long wcap = (long)a.dstCapacity - 1 - (long)ncBytes;
            const u32 cs = hp_encode_weights(slot + HP_HDR + 4 + ncBytes, wcap, (const u32*)(slot + HP_ICNT), sc[SC_MAXSV], (const u16*)(slot + HP_ST), wtl);
            sc[SC_HDR] = cs ? ncBytes + cs : 0;
        }
    }
    __syncthreads();
    HPT_MARK

    // ---- phase G (wide): header out (FSE-coded weights, or 4 bits per weight), final checks, meta
    for (u32 g = 0; g < HP_G; ++g) {
        u8* const slot = hpLds + g * HP_SLOT;
        const u32* const sc = (const u32*)(slot + HP_SCAL);
        if (!sc[SC_STATE]) continue;                                       // uniform
        const size_t b = b0 + g;
        const size_t n = MODE == HPM_ONESHOT ? view_size(a.src, b) : 0;
        const u32 maxSV = sc[SC_MAXSV], huffLog = sc[SC_LOG], hs = sc[SC_HDR];
        const size_t err = (size_t)sc[SC_RESULT_LO] | ((size_t)sc[SC_RESULT_HI] << 32);
        u8* const dst = a.dst + b * a.dstStride;
        HufMeta m; m.state = 0; m.hdrSize = 0; m.tableLog = 0; m.maxSV = 0;
        size_t result = 0, h = 0;
        if (err) result = err;
        else if (hs > 1 && hs < maxSV / 2) {                               // FSE-coded weights (huf_compress.c:131-135)
            const u8* const img = slot + HP_HDR + 4;
            for (u32 i = lane; i < hs; i += 64) dst[1 + i] = img[i];
            if (lane == 0) dst[0] = (u8)hs;
            h = hs + 1;
        } else if (maxSV > 128) result = FERR(GENERIC);                    // :138
        else if ((size_t)((maxSV + 1) / 2) + 1 > a.dstCapacity) result = FERR(dstSize_tooSmall);
        else {
            const u8* const nbSym = slot + HP_NBSYM;
            for (u32 i = lane; 2 * i < maxSV; i += 64) {
                const u32 n0 = nbSym[2 * i], n1 = 2 * i + 1 < maxSV ? nbSym[2 * i + 1] : 0;
                const u32 w0 = n0 ? huffLog + 1 - n0 : 0, w1 = n1 ? huffLog + 1 - n1 : 0;
                dst[1 + i] = (u8)((w0 << 4) + w1);
            }
            if (lane == 0) dst[0] = (u8)(128 + (maxSV - 1));
            h = (maxSV + 1) / 2 + 1;
        }
        if (MODE == HPM_WRITE) { if (lane == 0) a.results[b] = h ? h : result; continue; }   // HUF_writeCTable returns the header size
        if (h) {
            if (h + 12ul >= n) result = 0;                                 // :715
            else { m.state = 1; m.hdrSize = (u32)h; m.tableLog = huffLog; m.maxSV = maxSV; }
        }
        if (lane == 0) { a.meta[b] = m; if (m.state == 0) a.results[b] = result; }
    }
    HPT_MARK
    HPT_DUMP(0)
}

// =====================================================================================================================
//  decompress side
// =====================================================================================================================
// serial decode of the FSE-coded weights (lane g): FSE_decompress_usingDTable on the small table (lib/fse_decompress.c:178-238)
// with the literal bit reader; cells = {u16 newState; u8 symbol; u8 nbBits}
DEV size_t hp_decode_weights(u8* w, long omax, const u8* in, size_t n, const u32* cells, u32 tl, bool fast)
{
    BitReader r;
    {   const size_t e = r.init(in, n); if (is_err(e)) return e; }
    u32 s1 = r.read(tl); r.reload();
    u32 s2 = r.read(tl); r.reload();
    long op = 0;
    for (;;) {
        const int st = r.reload();
        if (!((st == BR_UNFINISHED) & (op < omax - 3))) break;
        w[op + 0] = (u8)fse_step(s1, r, cells, fast); w[op + 1] = (u8)fse_step(s2, r, cells, fast);
        w[op + 2] = (u8)fse_step(s1, r, cells, fast); w[op + 3] = (u8)fse_step(s2, r, cells, fast);
        op += 4;
    }
    for (;;) {
        if (op > omax - 2) return FERR(dstSize_tooSmall);
        w[op++] = (u8)fse_step(s1, r, cells, fast);
        if (r.reload() == BR_OVERFLOW) { w[op++] = (u8)fse_step(s2, r, cells, fast); break; }
        if (op > omax - 2) return FERR(dstSize_tooSmall);
        w[op++] = (u8)fse_step(s2, r, cells, fast);
        if (r.reload() == BR_OVERFLOW) { w[op++] = (u8)fse_step(s1, r, cells, fast); break; }
    }
    return (size_t)op;
}

// decode-side slot: counters of the weights' NCount header (s16[256]) at HP_KEYS, symbols sorted by (weight, symbol) (u8[256]) at
// HP_KEYS + 512, weights by symbol (u8[256 + 4]) at HP_ICNT, small DTable cells u32[64] at HP_ST (spans ST, TT), the FSE-coded
// weights themselves (<= 127 bytes, copied from the block so that the serial decoder reads LDS) at HP_HDR, first cell / first
// sorted position of every weight class (u16[16] each) at HP_LAST / HP_NRM
#define HD_NORM   HP_KEYS
#define HD_SORTED (HP_KEYS + 512)
#define HD_WGT    HP_ICNT
#define HD_CELLS  HP_ST
#define HD_BYTES  HP_HDR
#define HD_CSTART HP_LAST
#define HD_CBASE  HP_NRM
enum { DS_STATE = 0, DS_HDR, DS_NSYM, DS_FSE, DS_RESULT_LO, DS_RESULT_HI, DS_AUX, DS_CLS };   // DS_STATE: 0 done, 1 table pending, 2 raw copy, 3 rle fill

template <int G_>
__global__ __launch_bounds__(64) void k_huf_dprep(HufDPrepArgs a)
{
    constexpr u32 HP_G = G_;
    extern __shared__ __attribute__((aligned(16))) u8 hpLds[];
    const u32 lane = threadIdx.x;
    const size_t b0 = (size_t)blockIdx.x * HP_G;
    HPT_DECL

    // ---- phase A (serial, lane g): HUF_decompress's raw / RLE decisions, the weights' header (4-bit weights are unpacked here,
    //      FSE-coded ones are copied into LDS and get their NCount header parsed)
    if (lane < HP_G) {
        u8* const slot = hpLds + lane * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        const size_t b = b0 + lane;
        u32 state = 0; size_t result = 0;
        sc[DS_FSE] = 0; sc[DS_CLS] = 0xFFFFFFFFu;
        if (b < a.nBlocks) {
            const u8* const in = view_ptr(a.csrc, b);
            const size_t cSize = view_size(a.csrc, b), dstSize = view_size(a.dstSizes, b);
            u8* const w = slot + HD_WGT;
            if (a.tableOnly && cSize == 0) result = FERR(srcSize_wrong);  // HUF_readStats, entropy_common.c:165
            else if (!a.tableOnly && dstSize == 0) result = FERR(dstSize_tooSmall);            // huf_decompress.c:1063-1066
            else if (!a.tableOnly && cSize > dstSize) result = FERR(corruption_detected);
            else if (!a.tableOnly && cSize == dstSize) { state = 2; result = dstSize; }   // not compressed: copied below, coalesced
            else if (!a.tableOnly && cSize == 1) { state = 3; result = dstSize; }         // one byte repeated
            else {                                                         // weights header (entropy_common.c:154-182); cSize >= 2 here (table only: >= 1)
                const u32 first = in[0];
                if (first >= 128) {                                        // 4 bits per weight
                    const u32 nW = first - 127, bytes = (nW + 1) / 2;
                    if ((size_t)bytes + 1 > cSize) result = FERR(srcSize_wrong);
                    else {
                        for (u32 i = 0; i < bytes; ++i) { const u32 v = in[1 + i]; w[2 * i] = (u8)(v >> 4); w[2 * i + 1] = (u8)(v & 15u); }
                        state = 1; sc[DS_NSYM] = nW; sc[DS_HDR] = bytes + 1;
                    }
                } else if ((size_t)first + 1 > cSize) result = FERR(srcSize_wrong);
                else {                                                     // FSE-coded: FSE_decompress_wksp(weights, 255, in + 1, first, .., 6)
                    u8* const copy = slot + HD_BYTES;
                    u32 i = 0;
                    for (; i + 4 <= first; i += 4) { u32 v; __builtin_memcpy(&v, in + 1 + i, 4); *(u32*)(copy + i) = v; }
                    for (; i < first; ++i) copy[i] = in[1 + i];
                    u32 tl = 0, maxSV = 255;
                    const size_t h = ncount_read<1>((s16*)(slot + HD_NORM), &maxSV, &tl, copy, first);
                    if (is_err(h)) result = h;
                    else if (tl > 6) result = FERR(tableLog_tooLarge);
                    else { state = 1; sc[DS_FSE] = 1u | (tl << 8) | (maxSV << 16); sc[DS_HDR] = first + 1; sc[DS_AUX] = (u32)h; }
                }
            }
        }
        if (state == 1) {                                                  // which decoder?  (until phase D: the kind, kept in DS_CLS)
            const u8* const in = view_ptr(a.csrc, b);
            const size_t cSize = view_size(a.csrc, b), hdr = sc[DS_HDR];
            u32 kind = HUF_DKIND_SERIAL;
            if (!a.tableOnly && cSize >= hdr + 10 && view_size(a.dstSizes, b) >= 64) {     // the jump table (huf_decompress.c:277-287)
                const u8* const jt = in + hdr;
                const size_t l0 = jt[0] | ((u32)jt[1] << 8), l1 = jt[2] | ((u32)jt[3] << 8), l2 = jt[4] | ((u32)jt[5] << 8), used = 6 + l0 + l1 + l2;
                if (used < cSize - hdr) {
                    const size_t l3 = cSize - hdr - used;
                    const size_t mn = min(min(l0, l1), min(l2, l3)), mx = max(max(l0, l1), max(l2, l3));
                    // (a stream beyond the largest budget -- blocks of 64 / 128 KB -- is decoded in pieces by the class with the largest one)
                    if (8 * mn >= HPAR_MIN_BITS + 8) kind = HPAR_USE_TINY && mx + 96 <= HPAR_DATA_TINY ? HUF_DKIND_PAR_TINY : (mx + 96 <= HPAR_DATA_SMALL || HPAR_ALL_SMALL) ? HUF_DKIND_PAR_SMALL : HUF_DKIND_PAR_LARGE;
                }
            }
            sc[DS_CLS] = kind;
        }
        sc[DS_STATE] = state; sc[DS_RESULT_LO] = (u32)result; sc[DS_RESULT_HI] = (u32)(result >> 32);
    }
    __syncthreads();
    HPT_MARK

    // ---- phase B (wide): small decoding table of the FSE-coded weights (FSE_buildDTable, lib/fse_decompress.c:71-126)
    for (u32 g = 0; g < HP_G; ++g) {
        u8* const slot = hpLds + g * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        if (sc[DS_STATE] != 1 || !(sc[DS_FSE] & 1u)) continue;             // uniform
        const u32 tl = (sc[DS_FSE] >> 8) & 0xFFu, maxSV = sc[DS_FSE] >> 16, ts = 1u << tl;
        const s16* const nrm = (const s16*)(slot + HD_NORM);
        u32* const cells = (u32*)(slot + HD_CELLS);
        bool big = false;
        for (u32 s = 0; s <= maxSV; ++s) big |= nrm[s] >= (int)(ts >> 1);
        hp_small_table(slot + HP_CELL, maxSV, tl, lane, [&](u32 s) { return (int)nrm[s]; },
                       [&](u32 u, u32 s, u32 r, u32 first, int n) {
                           (void)first;
                           const u32 next = (n > 0 ? (u32)n : 1u) + r;    // symbolNext[s]++ (fse_decompress.c:117-122)
                           const u32 nb = tl - hibit32(next);
                           cells[u] = (((next << nb) - ts) & 0xFFFFu) | (s << 16) | (nb << 24);
                       });
        // (the counters of a parsed header sum to the table size, so the spread always closes: fse_decompress.c:113 cannot fire)
        if (lane == 0) sc[DS_FSE] |= big ? 0u : 2u;                         // fastMode: no counter takes half the table (:95)
        __syncthreads();
    }
    __syncthreads();
    HPT_MARK

    // ---- phase C (serial): the weights themselves
    if (lane < HP_G) {
        u8* const slot = hpLds + lane * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        if (sc[DS_STATE] == 1 && (sc[DS_FSE] & 1u)) {
            const u32 first = sc[DS_HDR] - 1, h = sc[DS_AUX];
            const size_t nW = hp_decode_weights(slot + HD_WGT, 255, slot + HD_BYTES + h, first - h, (const u32*)(slot + HD_CELLS), (sc[DS_FSE] >> 8) & 0xFFu, (sc[DS_FSE] & 2u) != 0);
            if (is_err(nW)) { sc[DS_STATE] = 0; sc[DS_RESULT_LO] = (u32)nW; sc[DS_RESULT_HI] = (u32)(nW >> 32); }
            else sc[DS_NSYM] = (u32)nW;
        }
    }
    __syncthreads();
    HPT_MARK

    // ---- phase D (wide): weight statistics, implied last weight, the X1 table; raw / RLE blocks are copied here
    for (u32 g = 0; g < HP_G; ++g) {
        u8* const slot = hpLds + g * HP_SLOT;
        u32* const sc = (u32*)(slot + HP_SCAL);
        const size_t b = b0 + g;
        if (b >= a.nBlocks) continue;                                      // uniform
        const u32 state = sc[DS_STATE];
        size_t result = (size_t)sc[DS_RESULT_LO] | ((size_t)sc[DS_RESULT_HI] << 32);
        HufMeta m; m.state = 0; m.hdrSize = 0; m.tableLog = 0; m.maxSV = 0;
        if (state == 2 || state == 3) {                                    // coalesced copy / fill of the whole block
            const u8* const in = view_ptr(a.csrc, b);
            u8* const dst = a.dst + b * a.dstStride;
            const size_t dstSize = view_size(a.dstSizes, b);
            const u32 fill = (u32)in[0] * 0x01010101u;
            const size_t head = dstSize < 4 ? dstSize : (size_t)((0 - (uintptr_t)dst) & 3u);
            if (lane < head) dst[lane] = state == 2 ? in[lane] : (u8)fill;
            for (size_t i = head + 4 * (size_t)lane; i + 4 <= dstSize; i += 256) {
                u32 v = fill;
                if (state == 2) __builtin_memcpy(&v, in + i, 4);
                *(u32*)(dst + i) = v;
            }
            const size_t done = head + ((dstSize - head) & ~(size_t)3);
            if (done + lane < dstSize) dst[done + lane] = state == 2 ? in[done + lane] : (u8)fill;
        } else if (state == 1) {
            const u32 nW = sc[DS_NSYM];                                    // weights read; the last symbol's is implied
            u32 wt[4];
            {   const u32 w4 = ((const u32*)(slot + HD_WGT))[lane];
#pragma unroll
                for (int i = 0; i < 4; ++i) wt[i] = 4 * lane + i < nW ? (w4 >> (8 * i)) & 0xFFu : 0xFFu; }
            bool bad = false; u32 mass = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (wt[i] != 0xFFu) { bad |= wt[i] >= HUF_MAX_TL; mass += (1u << (wt[i] & 15u)) >> 1; }
            mass = wg_sum<64>(mass);
            size_t err = 0;
            u32 tl = 0, lastW = 0;
            if (__any(bad) || mass == 0) err = FERR(corruption_detected);   // entropy_common.c:188-192
            else {
                tl = hibit32(mass) + 1;
                const u32 rest = (1u << tl) - mass;
                if (tl > HUF_MAX_TL || (rest & (rest - 1)) != 0) err = FERR(corruption_detected);   // the rest must be a clean power of 2
                else lastW = hibit32(rest) + 1;
            }
            if (!err) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (4 * lane + i == nW) wt[i] = lastW;
                // class = weight: symbols per weight and my rank among the symbols of the same weight (symbol order)
                u32 cls[4], rank[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) cls[i] = wt[i] == 0xFFu ? 15u : wt[i];
                const Pk tot = pk_rank(cls, rank, lane);
                const u32 cnt1 = pk_get(tot, 1);
                if (cnt1 < 2 || (cnt1 & 1u)) err = FERR(corruption_detected);   // entropy_common.c:208
                else if (tl > a.dtMaxLog + 1) err = FERR(tableLog_tooLarge);   // huf_decompress.c:137 (one-shot path: the DTable of HUF_CREATE_STATIC_DTABLEX1(.., HUF_TABLELOG_MAX))
                else {
                    // X1 cells {byte, nbBits} (huf_decompress.c:158-183): symbol n owns (1 << w) >> 1 consecutive cells, the symbols of
                    // a weight follow each other in symbol order, the weights in ascending order.  Restated by cell: the symbols are
                    // ranked by (weight, symbol) into a list; a cell finds its weight class from the class starts and its symbol
                    // by position -- every lane fills its own cells, two per 4-byte store.
                    u16* const cstart = (u16*)(slot + HD_CSTART);           // first cell of every weight class (tl + 2 entries)
                    u16* const cbase = (u16*)(slot + HD_CBASE);             // first list position of every weight class
                    u8* const sorted = slot + HD_SORTED;
                    u32 myStart[4] = { 0, 0, 0, 0 };
                    {   u32 cell = 0, pos = 0;
                        for (u32 v = 1; v <= tl + 1; ++v) {                // uniform
                            if (lane == 0) { cstart[v] = (u16)cell; cbase[v] = (u16)pos; }
#pragma unroll
                            for (int i = 0; i < 4; ++i) if (cls[i] == v) myStart[i] = pos;
                            const u32 c = v <= tl ? pk_get(tot, v) : 0;
                            cell += c << (v - 1); pos += c;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (cls[i] >= 1 && cls[i] < 14) sorted[myStart[i] + rank[i]] = (u8)(4 * lane + i);
                    __syncthreads();
                    u32* const dt = a.dtables + b * a.dtStrideU32;
                    u32* const out = dt + 1;
                    const u32 pairs = 1u << (tl - 1);
                    u32 csr[HUF_MAX_TL + 1];                                // class starts in registers (uniform)
#pragma unroll
                    for (u32 t = 2; t <= HUF_MAX_TL; ++t) csr[t] = t <= tl ? (u32)cstart[t] : 0xFFFFu;
                    for (u32 q = lane; q < pairs; q += 64) {
                        const u32 u = 2 * q;
                        u32 v = 1;
#pragma unroll
                        for (u32 t = 2; t <= HUF_MAX_TL; ++t) v = u >= csr[t] ? t : v;   // classes ascend with the cell index
                        const u32 cs = cstart[v], cb = cbase[v], nbits = (tl + 1 - v) << 8;
                        u32 lo, hi;
                        if (v == 1) { lo = sorted[cb + (u - cs)]; hi = sorted[cb + (u + 1 - cs)]; }   // one cell per symbol (their number is even)
                        else lo = hi = sorted[cb + ((u - cs) >> (v - 1))];
                        out[q] = (lo | nbits) | ((hi | nbits) << 16);
                    }
                    if (lane == 0) dt[0] = ((a.dtMaxLog & 0xFFu) * 0x01000001u) | (tl << 16);   // {maxTableLog, tableType 0, tableLog, reserved as HUF_CREATE_STATIC_DTABLEX1 leaves it}
                }
            }
            const u32 hdr = sc[DS_HDR];
            if (!err && !a.tableOnly && (size_t)hdr >= view_size(a.csrc, b)) err = FERR(srcSize_wrong);   // nothing behind the header (huf_decompress.c:432)
            if (err) result = err;
            else if (a.tableOnly) result = hdr;                           // (state stays 0: nothing to decode, the result is final)
            else { m.state = 1; m.hdrSize = hdr; m.tableLog = tl; }
        }
        if (lane == 0) { a.meta[b] = m; if (m.state == 0) a.results[b] = result; sc[DS_CLS] = m.state ? 2u * sc[DS_CLS] + (m.tableLog > 11u ? 1u : 0u) : 0xFFFFFFFFu; }
        __syncthreads();
    }
    __syncthreads();
    HPT_MARK
    HPT_DUMP(1)

    // ---- append the pending blocks to their decoder-class lists (one atomic per class and wave)
    {   int cls = -1;
        if (lane < HP_G && b0 + lane < a.nBlocks) cls = (int)((const u32*)(hpLds + lane * HP_SLOT + HP_SCAL))[DS_CLS];
#pragma unroll
        for (int c = 0; c < HUF_DCLS_COUNT; ++c) {
            const unsigned long long mask = __ballot(cls == c);
            if (!mask) continue;                                           // uniform
            const int leader = __builtin_ctzll(mask);
            u32 base = 0;
            if ((int)lane == leader) base = atomicAdd(&a.counts[c], (u32)__builtin_popcountll(mask));
            base = (u32)__shfl((int)base, leader, WAVE);
            if (cls == c) a.lists[(size_t)c * a.nBlocks + base + (u32)__builtin_popcountll(mask & hp_below_mask(lane))] = (u32)(b0 + lane);
        }
    }
}

hipError_t launch_huf_cprep(const HufCPrepArgs& a, hipStream_t s, void* /*unused*/)
{
    if (a.nBlocks == 0) return hipSuccess;
    probe_before(PK_HUF_CPREP, s);
    hipLaunchKernelGGL(k_huf_cprep<HP_G_COMPRESS>, dim3((unsigned)((a.nBlocks + HP_G_COMPRESS - 1) / HP_G_COMPRESS)), dim3(64), HP_G_COMPRESS * HP_SLOT, s, a);
    probe_after(PK_HUF_CPREP, s);
    return hipGetLastError();
}
// HUF_buildCTable on the caller's counters (mode 1: a.counts, a.maxSVs, a.huffLogReq in; a.ctables, a.results out) / HUF_writeCTable on the caller's
// tables (mode 2: a.ctables, a.maxSVs, a.huffLogReq in; a.dst, a.results out)
hipError_t launch_huf_cprep_glue(const HufCPrepArgs& a, int mode, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    const dim3 grid((unsigned)((a.nBlocks + HP_G_COMPRESS - 1) / HP_G_COMPRESS));
    if (mode == HPM_BUILD) hipLaunchKernelGGL((k_huf_cprep<HP_G_COMPRESS, HPM_BUILD>), grid, dim3(64), HP_G_COMPRESS * HP_SLOT, s, a);
    else hipLaunchKernelGGL((k_huf_cprep<HP_G_COMPRESS, HPM_WRITE>), grid, dim3(64), HP_G_COMPRESS * HP_SLOT, s, a);
    return hipGetLastError();
}
hipError_t launch_huf_dprep(const HufDPrepArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    {   const hipError_t e = launch_zero_u32(a.counts, HUF_DCLS_COUNT, s); if (e != hipSuccess) return e; }
    probe_before(PK_HUF_DPREP, s);
    hipLaunchKernelGGL(k_huf_dprep<HP_G_DECOMPRESS>, dim3((unsigned)((a.nBlocks + HP_G_DECOMPRESS - 1) / HP_G_DECOMPRESS)), dim3(64), HP_G_DECOMPRESS * HP_SLOT, s, a);
    probe_after(PK_HUF_DPREP, s);
    return hipGetLastError();
}
