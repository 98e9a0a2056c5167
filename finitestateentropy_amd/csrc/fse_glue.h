// fse_glue.h -- device-side FSE glue shared by the FSE and Huff0 prepare kernels (SURVEY 8(a') rows g1-g3):
//   FSE_optimalTableLog, FSE_normalizeCount, FSE_writeNCount, FSE_readNCount, FSE_buildCTable, FSE_buildDTable
//   (reference: lib/fse_compress.c:316-342, :348-494, :186-298, :66-169; lib/entropy_common.c:41-144;
//    lib/fse_decompress.c:71-126).  Plain serial device functions: one lane per table.
#pragma once
#include "dev_common.h"

#define FSE_MIN_TL FSEHIP_FSE_MIN_TABLELOG
#define FSE_MAX_TL FSEHIP_FSE_MAX_TABLELOG
#define FSE_DEF_TL FSEHIP_FSE_DEFAULT_TABLELOG
#define FSE_ABS_MAX_TL 15
#define FSE_TSTEP(ts) (((ts) >> 1) + ((ts) >> 3) + 3)     // lib/fse.h:683

// ---------------------------------------------------------------------------------------------------
//  table log selection (lib/fse_compress.c:316-342)
// ---------------------------------------------------------------------------------------------------
DEV u32 fse_min_tablelog(size_t srcSize, u32 maxSV)
{
    const u32 a = hibit32((u32)srcSize) + 1, b = hibit32(maxSV) + 2;
    return a < b ? a : b;
}
__device__ inline u32 fse_optimal_tablelog(u32 maxTableLog, size_t srcSize, u32 maxSV, u32 minus)
{
    const u32 bySrc = hibit32((u32)(srcSize - 1)) - minus;
    const u32 floorBits = fse_min_tablelog(srcSize, maxSV);
    u32 tl = maxTableLog ? maxTableLog : FSE_DEF_TL;
    if (bySrc < tl) tl = bySrc;
    if (floorBits > tl) tl = floorBits;
    if (tl < FSE_MIN_TL) tl = FSE_MIN_TL;
    if (tl > FSE_MAX_TL) tl = FSE_MAX_TL;
    return tl;
}

// ---------------------------------------------------------------------------------------------------
//  FSE_normalizeCount (lib/fse_compress.c:435-494) and its fallback (:348-432)
// ---------------------------------------------------------------------------------------------------
__device__ inline size_t fse_normalize_fallback(s16* norm, u32 tl, const unsigned* count, size_t total, u32 maxSV)
{
    const s16 PENDING = -2;
    u32 given = 0, left;
    const u32 tiny = (u32)(total >> tl);
    u32 one = (u32)((total * 3) >> (tl + 1));
    for (u32 s = 0; s <= maxSV; s++) {
        const u32 c = count[s];
        if (c == 0) { norm[s] = 0; continue; }
        if (c <= tiny) { norm[s] = -1; given++; total -= c; continue; }
        if (c <= one) { norm[s] = 1; given++; total -= c; continue; }
        norm[s] = PENDING;
    }
    left = (1u << tl) - given;
    if (left == 0) return 0;
    if ((total / left) > one) {
        one = (u32)((total * 3) / (left * 2));
        for (u32 s = 0; s <= maxSV; s++)
            if (norm[s] == PENDING && count[s] <= one) { norm[s] = 1; given++; total -= count[s]; }
        left = (1u << tl) - given;
    }
    if (given == maxSV + 1) {
        u32 arg = 0, best = 0;
        for (u32 s = 0; s <= maxSV; s++) if (count[s] > best) { arg = s; best = count[s]; }
        norm[arg] = (s16)(norm[arg] + (s16)left);
        return 0;
    }
    if (total == 0) {
        for (u32 s = 0; left > 0; s = (s + 1) % (maxSV + 1)) if (norm[s] > 0) { left--; norm[s]++; }
        return 0;
    }
    {   const u64 vlog = 62 - tl;
        const u64 mid = ((u64)1 << (vlog - 1)) - 1;
        const u64 rstep = ((((u64)1 << vlog) * left) + mid) / total;
        u64 run = mid;
        for (u32 s = 0; s <= maxSV; s++) {
            if (norm[s] == PENDING) {
                const u64 end = run + (u64)count[s] * rstep;
                const u32 w = (u32)(end >> vlog) - (u32)(run >> vlog);
                if (w < 1) return FERR(GENERIC);
                norm[s] = (s16)w;
                run = end;
            }
        }
    }
    return 0;
}

__device__ inline size_t fse_normalize_count(s16* norm, u32 tl, const unsigned* count, size_t total, u32 maxSV)
{
    const u32 rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };   // :445
    if (tl == 0) tl = FSE_DEF_TL;
    if (tl < FSE_MIN_TL) return FERR(GENERIC);
    if (tl > FSE_MAX_TL) return FERR(tableLog_tooLarge);
    if (tl < fse_min_tablelog(total, maxSV)) return FERR(GENERIC);
    const u64 scale = 62 - tl;
    const u64 step = ((u64)1 << 62) / total;
    const u64 vstep = (u64)1 << (scale - 20);
    int still = 1 << tl;
    u32 argmax = 0;
    s16 pmax = 0;
    const u32 tiny = (u32)(total >> tl);
    for (u32 s = 0; s <= maxSV; s++) {
        const u32 c = count[s];
        if (c == total) return 0;
        if (c == 0) { norm[s] = 0; continue; }
        if (c <= tiny) { norm[s] = -1; still--; continue; }
        s16 p = (s16)(((u64)c * step) >> scale);
        if (p < 8) {
            const u64 beat = vstep * rtb[p];
            p = (s16)(p + ((((u64)c * step) - ((u64)p << scale)) > beat));
        }
        if (p > pmax) { pmax = p; argmax = s; }
        norm[s] = p;
        still -= p;
    }
    if (-still >= (norm[argmax] >> 1)) {
        const size_t e = fse_normalize_fallback(norm, tl, count, total, maxSV);
        if (is_err(e)) return e;
    } else norm[argmax] = (s16)(norm[argmax] + (s16)still);
    return tl;
}

// ---------------------------------------------------------------------------------------------------
//  FSE_writeNCount (lib/fse_compress.c:186-298)
// ---------------------------------------------------------------------------------------------------
__device__ inline size_t fse_write_ncount(u8* out, size_t cap, const s16* norm, u32 maxSV, u32 tl)
{
    long o = 0;
    const long lim = (long)cap - 2;
    const size_t bound = maxSV ? (size_t)((((maxSV + 1) * tl) >> 3) + 3) : (size_t)FSEHIP_FSE_NCOUNTBOUND;
    const bool safe = cap >= bound;
    const u32 alphabet = maxSV + 1;
    const int tsize = 1 << tl;
    int remaining = tsize + 1, threshold = tsize, nbBits = (int)tl + 1;
    u32 acc; int nacc;
    u32 sym = 0; bool prevZero = false;
    if (tl > FSE_MAX_TL) return FERR(tableLog_tooLarge);
    if (tl < FSE_MIN_TL) return FERR(GENERIC);
    acc = tl - FSE_MIN_TL; nacc = 4;
#define NC_SPILL() do { if (!safe && o > lim) return FERR(dstSize_tooSmall); \
                        out[o] = (u8)acc; out[o + 1] = (u8)(acc >> 8); o += 2; acc >>= 16; } while (0)
    while (sym < alphabet && remaining > 1) {
        if (prevZero) {
            u32 from = sym;
            while (sym < alphabet && !norm[sym]) sym++;
            if (sym == alphabet) break;
            while (sym >= from + 24) { from += 24; acc += 0xFFFFu << nacc; NC_SPILL(); }
            while (sym >= from + 3) { from += 3; acc += 3u << nacc; nacc += 2; }
            acc += (sym - from) << nacc; nacc += 2;
            if (nacc > 16) { NC_SPILL(); nacc -= 16; }
        }
        {   int c = norm[sym++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= c < 0 ? -c : c;
            c++;
            if (c >= threshold) c += max;
            acc += (u32)c << nacc;
            nacc += nbBits;
            nacc -= (c < max);
            prevZero = (c == 1);
            if (remaining < 1) return FERR(GENERIC);
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (nacc > 16) { NC_SPILL(); nacc -= 16; }
    }
    if (remaining != 1) return FERR(GENERIC);
    if (!safe && o > lim) return FERR(dstSize_tooSmall);
    out[o] = (u8)acc; out[o + 1] = (u8)(acc >> 8);
    o += (nacc + 7) / 8;
#undef NC_SPILL
    return (size_t)o;
}

// ---------------------------------------------------------------------------------------------------
//  FSE_readNCount (lib/entropy_common.c:41-144)
// ---------------------------------------------------------------------------------------------------
__device__ inline size_t fse_read_ncount_ge4(s16* norm, u32* maxSVPtr, u32* tlPtr, const u8* in, size_t hbSize)
{
    long ip = 0;
    const long iend = (long)hbSize;
    int nbBits, remaining, threshold, bitCount;
    u32 bits, charnum = 0;
    bool prevZero = false;
    for (u32 s = 0; s <= *maxSVPtr; s++) norm[s] = 0;
    bits = ld32(in);
    nbBits = (int)(bits & 0xF) + FSE_MIN_TL;
    if (nbBits > FSE_ABS_MAX_TL) return FERR(tableLog_tooLarge);
    bits >>= 4; bitCount = 4;
    *tlPtr = (u32)nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    while ((remaining > 1) & (charnum <= *maxSVPtr)) {
        if (prevZero) {
            u32 n0 = charnum;
            while ((bits & 0xFFFF) == 0xFFFF) {
                n0 += 24;
                if (ip < iend - 5) { ip += 2; bits = ld32(in + ip) >> bitCount; }
                else { bits >>= 16; bitCount += 16; }
            }
            while ((bits & 3) == 3) { n0 += 3; bits >>= 2; bitCount += 2; }
            n0 += bits & 3; bitCount += 2;
            if (n0 > *maxSVPtr) return FERR(maxSymbolValue_tooSmall);
            while (charnum < n0) norm[charnum++] = 0;
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) {
                ip += bitCount >> 3; bitCount &= 7; bits = ld32(in + ip) >> bitCount;
            } else bits >>= 2;
        }
        {   const int max = (2 * threshold - 1) - remaining;
            int c;
            if ((bits & (u32)(threshold - 1)) < (u32)max) { c = (int)(bits & (u32)(threshold - 1)); bitCount += nbBits - 1; }
            else { c = (int)(bits & (u32)(2 * threshold - 1)); if (c >= threshold) c -= max; bitCount += nbBits; }
            c--;
            remaining -= c < 0 ? -c : c;
            norm[charnum++] = (s16)c;
            prevZero = !c;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= (int)(8 * (iend - 4 - ip)); ip = iend - 4; }
            bits = ld32(in + ip) >> (bitCount & 31);
        }
    }
    if (remaining != 1) return FERR(corruption_detected);
    if (bitCount > 32) return FERR(corruption_detected);
    *maxSVPtr = charnum - 1;
    ip += (bitCount + 7) >> 3;
    return (size_t)ip;
}

__device__ inline size_t fse_read_ncount(s16* norm, u32* maxSVPtr, u32* tlPtr, const u8* in, size_t hbSize)
{
    if (hbSize < 4) {                                     // :55-64 : zero-extended private copy
        u8 tmp[4] = { 0, 0, 0, 0 };
        for (size_t i = 0; i < hbSize; i++) tmp[i] = in[i];
        const size_t r = fse_read_ncount_ge4(norm, maxSVPtr, tlPtr, tmp, 4);
        if (is_err(r)) return r;
        if (r > hbSize) return FERR(corruption_detected);
        return r;
    }
    return fse_read_ncount_ge4(norm, maxSVPtr, tlPtr, in, hbSize);
}

// ---------------------------------------------------------------------------------------------------
//  symbol spread shared by both table builders (lib/fse_compress.c:96-122, lib/fse_decompress.c:86-114)
//  cellSym[u] = symbol owning state u.  Returns false when the walk does not close (bad norm).
// ---------------------------------------------------------------------------------------------------
template <int STRIDE>
__device__ inline bool fse_spread(u8* cellSym, const s16* norm, u32 maxSV, u32 tl)
{
    const u32 ts = 1u << tl, mask = ts - 1, step = FSE_TSTEP(ts);
    u32 high = ts - 1, pos = 0;
    for (u32 s = 0; s <= maxSV; s++) if (norm[s] == -1) cellSym[(high--) * STRIDE] = (u8)s;
    for (u32 s = 0; s <= maxSV; s++) {
        const int n = norm[s];
        for (int k = 0; k < n; k++) {
            cellSym[pos * STRIDE] = (u8)s;
            do pos = (pos + step) & mask; while (pos > high);
        }
    }
    return pos == 0;
}

// FSE_buildCTable_wksp (lib/fse_compress.c:66-169); ct in the reference layout
__device__ inline void fse_build_ctable(u32* ct, u8* cellSym, const s16* norm, u32 maxSV, u32 tl)
{
    const u32 ts = 1u << tl;
    u16* const head = (u16*)ct;
    u16* const stateTable = head + 2;
    u32* const tt = ct + 1 + (tl ? ts >> 1 : 1);
    u16 first[257];
    head[0] = (u16)tl; head[1] = (u16)maxSV;
    first[0] = 0;
    for (u32 s = 0; s <= maxSV; s++) first[s + 1] = (u16)(first[s] + (norm[s] == -1 ? 1 : norm[s]));
    fse_spread<1>(cellSym, norm, maxSV, tl);
    for (u32 u = 0; u < ts; u++) { const u32 s = cellSym[u]; stateTable[first[s]++] = (u16)(ts + u); }
    int total = 0;
    for (u32 s = 0; s <= maxSV; s++) {
        const int n = norm[s];
        if (n == 0) { tt[2 * s] = 0; tt[2 * s + 1] = ((tl + 1) << 16) - (1u << tl); continue; }
        if (n == -1 || n == 1) {
            tt[2 * s + 1] = (tl << 16) - (1u << tl);
            tt[2 * s] = (u32)(total - 1);
            total++;
        } else {
            const u32 maxBitsOut = tl - hibit32((u32)n - 1);
            tt[2 * s + 1] = (maxBitsOut << 16) - ((u32)n << maxBitsOut);
            tt[2 * s] = (u32)(total - n);
            total += n;
        }
    }
}

// FSE_buildDTable (lib/fse_decompress.c:71-126); dt in the reference layout: header {u16 tableLog; u16 fastMode}
// then cells {u16 newState; u8 symbol; u8 nbBits}.
__device__ inline size_t fse_build_dtable(u32* dt, const s16* norm, u32 maxSV, u32 tl)
{
    const u32 ts = 1u << tl;
    u16 next[256];
    u32 fast = 1;
    if (maxSV > 255) return FERR(maxSymbolValue_tooLarge);
    if (tl > FSE_MAX_TL) return FERR(tableLog_tooLarge);
    for (u32 s = 0; s <= maxSV; s++) {
        if (norm[s] == -1) next[s] = 1;
        else { if (norm[s] >= (s16)(1 << (tl - 1))) fast = 0; next[s] = (u16)norm[s]; }
    }
    dt[0] = tl | (fast << 16);
    u8* const cells = (u8*)(dt + 1);
    if (!fse_spread<4>(cells + 2, norm, maxSV, tl)) return FERR(GENERIC);   // symbol byte of every cell
    for (u32 u = 0; u < ts; u++) {
        const u32 s = cells[4 * u + 2];
        const u32 ns = next[s]++;
        const u32 nb = tl - hibit32(ns);
        dt[1 + u] = (((ns << nb) - ts) & 0xFFFFu) | (s << 16) | (nb << 24);
    }
    return 0;
}

