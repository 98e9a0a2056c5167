/* oracle/fse_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU restatement of the reference's FSE / Huff0 block path, written from the
 * reference's behaviour (every function cites the reference file:line it follows).  It is the
 * checker for the HIP product path and nothing else: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or call it.  The product library
 * (finitestateentropy_amd/csrc) never links, loads or falls back to anything in oracle/.
 *
 * Parity pinning: this restatement is itself checked (tests/test_oracle_vs_ref.py) against the
 * unmodified reference compiled by oracle/Makefile into oracle/_ref/libfse_ref.so, against the
 * known-answer table of SURVEY.md Appendix B, and against the fixtures in tests/golden/ that were
 * produced by that compiled reference (tests/golden/make_golden.py).
 *
 * Formulation notes (deliberately different from the reference's register mechanics):
 *  - the LIFO bit writer is a byte-granular appender; the "does not fit" rule of
 *    BIT_closeCStream (lib/bitstream.h:254-260) is the closed form floor(totalBits/8) >= cap-8;
 *  - the FSE encoder walks j = distance-from-the-end and picks the chain by the parity of j;
 *  - the bit reader keeps the reference's (window, used-bits, byte-offset) triple because the
 *    decoder's termination is defined through it (lib/bitstream.h:400-448).
 */
#include "fse_oracle.h"
#include <string.h>
#include <stdlib.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define ERR(name) ((size_t)0 - (size_t)ORC_E_##name)
unsigned orc_is_error(size_t code) { return code > ERR(maxCode); }   /* lib/error_private.h:79 */

#define FSE_MIN_TL 5            /* lib/fse.h:677 */
#define FSE_MAX_TL 12           /* lib/fse.h:673 (FSE_MAX_MEMORY_USAGE 14 - 2) */
#define FSE_DEFAULT_TL 11       /* lib/fse.h:676 */
#define FSE_ABS_MAX_TL 15       /* lib/fse.h:679 */
#define FSE_MAX_SV 255          /* lib/fse.h:655 */
#define HUF_MAX_TL 12           /* lib/huf.h:117 */
#define HUF_DEFAULT_TL 11       /* lib/huf.h:118 */
#define HUF_MAX_SV 255          /* lib/huf.h:119 */
#define HUF_ABS_MAX_TL 15       /* lib/huf.h:121 */
#define HUF_BLOCK_MAX (128 * 1024) /* lib/huf.h:72 */

static unsigned hibit(u32 v) { return 31u - (unsigned)__builtin_clz(v); }   /* lib/bitstream.h:139 */
static u32 le32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
static u64 le64(const u8* p) { return (u64)le32(p) | ((u64)le32(p + 4) << 32); }
static u32 le16(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8); }
static u64 lowmask(unsigned nb) { return nb >= 64 ? ~(u64)0 : (((u64)1 << nb) - 1); }

/* ------------------------------------------------------------------------------------------
 *  probagen  (programs/probaGenerator.c:70-74 LCG, :95-126 table + fill)
 * ---------------------------------------------------------------------------------------- */
void orc_probagen_table(u8 table[4096], double p)
{
    int remaining = 4096;
    unsigned pos = 0, s = 0;
    if (p == 0.0) p = 0.005;                       /* :105 */
    while (remaining) {                            /* :109-118 */
        unsigned n = (unsigned)(remaining * p);
        if (!n) n = 1;
        memset(table + pos, (int)(u8)s, n);
        pos += n; s++; remaining -= (int)n;
    }
}

void orc_probagen_block(u8* dst, size_t n, const u8 table[4096], u32 seed)
{
    size_t i;
    for (i = 0; i < n; i++) {
        seed = seed * 2654435761u + 2246822519u;   /* :70-74 */
        dst[i] = table[(seed >> 11) & 4095];
    }
}

void orc_probagen_batch(u8* dst, size_t stride, size_t n, size_t nBlocks, const u8 table[4096], u32 firstSeed)
{
    long b;
#pragma omp parallel for schedule(static)
    for (b = 0; b < (long)nBlocks; b++) orc_probagen_block(dst + (size_t)b * stride, n, table, firstSeed + (u32)b);
}

/* ------------------------------------------------------------------------------------------
 *  HIST_count  (lib/hist.c:175 -> :163 -> :141/:66 ; semantics SURVEY A.7)
 * ---------------------------------------------------------------------------------------- */
size_t orc_hist_count(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize)
{
    const u8* ip = (const u8*)src;
    unsigned full[256];
    unsigned const limitIn = *maxSymbolValuePtr;
    int const checked = limitIn < 255;                 /* hist.c:169 : caller-imposed alphabet limit */
    unsigned const nOut = checked ? limitIn + 1 : 256; /* entries written: hist.c:74,130 / :37 */
    unsigned top = 255, best = 0, s;
    size_t i;

    memset(full, 0, sizeof(full));
    if (srcSize == 0) {                                /* hist.c:83-87 / :38 */
        memset(count, 0, nOut * sizeof(unsigned));
        *maxSymbolValuePtr = 0;
        return 0;
    }
    for (i = 0; i < srcSize; i++) full[ip[i]]++;
    while (!full[top]) top--;
    if (checked && top > limitIn) return ERR(maxSymbolValue_tooSmall);   /* hist.c:128 */
    for (s = 0; s <= top; s++) if (full[s] > best) best = full[s];
    memcpy(count, full, nOut * sizeof(unsigned));
    *maxSymbolValuePtr = top;
    return best;
}

/* ------------------------------------------------------------------------------------------
 *  LIFO bit writer (lib/bitstream.h:183-260, closed form of SURVEY A.1)
 * ---------------------------------------------------------------------------------------- */
typedef struct { u8* out; size_t cap; size_t nbytes; u64 acc; unsigned fill; } bitw_t;

static void bitw_init(bitw_t* w, void* dst, size_t cap) { w->out = (u8*)dst; w->cap = cap; w->nbytes = 0; w->acc = 0; w->fill = 0; }

static void bitw_put(bitw_t* w, u64 value, unsigned nb)      /* nb <= 31 */
{
    w->acc |= (value & lowmask(nb)) << w->fill;
    w->fill += nb;
    while (w->fill >= 8) {
        if (w->nbytes < w->cap) w->out[w->nbytes] = (u8)w->acc;
        w->nbytes++; w->acc >>= 8; w->fill -= 8;
    }
}

/* append the end mark; size in bytes, or 0 when floor(totalBits/8) >= cap-8 (bitstream.h:190,258) */
static size_t bitw_close(bitw_t* w)
{
    bitw_put(w, 1, 1);
    if (w->cap <= 8 || w->nbytes >= w->cap - 8) return 0;
    if (w->fill) { w->out[w->nbytes] = (u8)w->acc; return w->nbytes + 1; }
    return w->nbytes;
}

/* ------------------------------------------------------------------------------------------
 *  LIFO bit reader (lib/bitstream.h:272-448)
 * ---------------------------------------------------------------------------------------- */
enum { BR_UNFINISHED = 0, BR_END_OF_BUFFER = 1, BR_COMPLETED = 2, BR_OVERFLOW = 3 };   /* bitstream.h:99-102 */
typedef struct { const u8* base; size_t size; size_t at; u64 win; unsigned used; } bitr_t;

static size_t bitr_init(bitr_t* r, const void* src, size_t size)
{
    const u8* b = (const u8*)src;
    memset(r, 0, sizeof(*r));
    if (size < 1) return ERR(srcSize_wrong);           /* :274 */
    r->base = b; r->size = size;
    if (size >= 8) {                                   /* :279-284 */
        r->at = size - 8;
        r->win = le64(b + r->at);
        if (b[size - 1] == 0) return ERR(GENERIC);
        r->used = 8 - hibit(b[size - 1]);
    } else {                                           /* :286-315 : short stream, zero-extended window */
        size_t k;
        r->at = 0;
        for (k = 0; k < size; k++) r->win |= (u64)b[k] << (8 * k);
        if (b[size - 1] == 0) return ERR(corruption_detected);
        r->used = 8 - hibit(b[size - 1]) + (unsigned)(8 - size) * 8;
    }
    return size;
}
static u64 bitr_peek(const bitr_t* r, unsigned nb) { return (r->win >> ((64u - r->used - nb) & 63u)) & lowmask(nb); }   /* :345-351, :333 */
static u64 bitr_peek_fast(const bitr_t* r, unsigned nb) { return (r->win << (r->used & 63u)) >> ((64u - nb) & 63u); }   /* :361-366 */
static u64 bitr_read(bitr_t* r, unsigned nb) { u64 v = bitr_peek(r, nb); r->used += nb; return v; }
static u64 bitr_read_fast(bitr_t* r, unsigned nb) { u64 v = bitr_peek_fast(r, nb); r->used += nb; return v; }

static int bitr_reload_fast(bitr_t* r)                 /* :378-388 */
{
    if (r->at < 8) return BR_OVERFLOW;
    r->at -= r->used >> 3; r->used &= 7; r->win = le64(r->base + r->at);
    return BR_UNFINISHED;
}
static int bitr_reload(bitr_t* r)                      /* :400-439 */
{
    if (r->used > 64) return BR_OVERFLOW;
    if (r->at >= 8) return bitr_reload_fast(r);
    if (r->at == 0) return r->used < 64 ? BR_END_OF_BUFFER : BR_COMPLETED;
    {   unsigned nbytes = r->used >> 3;
        int res = BR_UNFINISHED;
        if (r->at < nbytes) { nbytes = (unsigned)r->at; res = BR_END_OF_BUFFER; }
        r->at -= nbytes; r->used -= nbytes * 8; r->win = le64(r->base + r->at);
        return res;
    }
}
static int bitr_at_end(const bitr_t* r) { return r->at == 0 && r->used == 64; }   /* :445-448 */

/* ------------------------------------------------------------------------------------------
 *  FSE table log, normalisation  (lib/fse_compress.c:316-342, 348-494)
 * ---------------------------------------------------------------------------------------- */
static unsigned fse_min_tablelog(size_t srcSize, unsigned maxSV)       /* :316-323 */
{
    unsigned a = hibit((u32)srcSize) + 1, b = hibit(maxSV) + 2;
    return a < b ? a : b;
}

unsigned orc_fse_optimal_tablelog(unsigned maxTableLog, size_t srcSize, unsigned maxSV, unsigned minus)   /* :325-337 */
{
    unsigned const bySrc = hibit((u32)(srcSize - 1)) - minus;
    unsigned const floorBits = fse_min_tablelog(srcSize, maxSV);
    unsigned tl = maxTableLog ? maxTableLog : FSE_DEFAULT_TL;
    if (bySrc < tl) tl = bySrc;
    if (floorBits > tl) tl = floorBits;
    if (tl < FSE_MIN_TL) tl = FSE_MIN_TL;
    if (tl > FSE_MAX_TL) tl = FSE_MAX_TL;
    return tl;
}

static size_t fse_normalize_fallback(short* norm, u32 tl, const unsigned* count, size_t total, u32 maxSV)   /* :348-432 */
{
    short const PENDING = -2;
    u32 s, given = 0, left;
    u32 const tiny = (u32)(total >> tl);
    u32 one = (u32)((total * 3) >> (tl + 1));

    for (s = 0; s <= maxSV; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= tiny) { norm[s] = -1; given++; total -= count[s]; continue; }
        if (count[s] <= one) { norm[s] = 1; given++; total -= count[s]; continue; }
        norm[s] = PENDING;
    }
    left = (1u << tl) - given;
    if (left == 0) return 0;

    if ((total / left) > one) {                        /* :384-395 */
        one = (u32)((total * 3) / (left * 2));
        for (s = 0; s <= maxSV; s++)
            if (norm[s] == PENDING && count[s] <= one) { norm[s] = 1; given++; total -= count[s]; }
        left = (1u << tl) - given;
    }
    if (given == maxSV + 1) {                          /* :397-406 : everything is poor, dump on the max */
        u32 arg = 0, best = 0;
        for (s = 0; s <= maxSV; s++) if (count[s] > best) { arg = s; best = count[s]; }
        norm[arg] += (short)left;
        return 0;
    }
    if (total == 0) {                                  /* :408-413 */
        for (s = 0; left > 0; s = (s + 1) % (maxSV + 1)) if (norm[s] > 0) { left--; norm[s]++; }
        return 0;
    }
    {   u64 const vlog = 62 - tl;                      /* :415-429 */
        u64 const mid = ((u64)1 << (vlog - 1)) - 1;
        u64 const rstep = ((((u64)1 << vlog) * left) + mid) / total;
        u64 run = mid;
        for (s = 0; s <= maxSV; s++) {
            if (norm[s] == PENDING) {
                u64 const end = run + (u64)count[s] * rstep;
                u32 const w = (u32)(end >> vlog) - (u32)(run >> vlog);
                if (w < 1) return ERR(GENERIC);
                norm[s] = (short)w;
                run = end;
            }
        }
    }
    return 0;
}

size_t orc_fse_normalize_count(short* norm, unsigned tl, const unsigned* count, size_t total, unsigned maxSV)   /* :435-494 */
{
    static const u32 rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };   /* :445 */
    if (tl == 0) tl = FSE_DEFAULT_TL;
    if (tl < FSE_MIN_TL) return ERR(GENERIC);
    if (tl > FSE_MAX_TL) return ERR(tableLog_tooLarge);
    if (tl < fse_min_tablelog(total, maxSV)) return ERR(GENERIC);
    {   u64 const scale = 62 - tl;
        u64 const step = ((u64)1 << 62) / total;
        u64 const vstep = (u64)1 << (scale - 20);
        int still = 1 << tl;
        unsigned s, argmax = 0;
        short pmax = 0;
        u32 const tiny = (u32)(total >> tl);
        for (s = 0; s <= maxSV; s++) {
            if (count[s] == total) return 0;            /* rle */
            if (count[s] == 0) { norm[s] = 0; continue; }
            if (count[s] <= tiny) { norm[s] = -1; still--; continue; }
            {   short p = (short)((count[s] * step) >> scale);
                if (p < 8) {
                    u64 const beat = vstep * rtb[p];
                    p += (count[s] * step) - ((u64)p << scale) > beat;
                }
                if (p > pmax) { pmax = p; argmax = s; }
                norm[s] = p;
                still -= p;
            }
        }
        if (-still >= (norm[argmax] >> 1)) {
            size_t const e = fse_normalize_fallback(norm, tl, count, total, maxSV);
            if (orc_is_error(e)) return e;
        } else norm[argmax] += (short)still;
    }
    return tl;
}

/* ------------------------------------------------------------------------------------------
 *  NCount header  (write: lib/fse_compress.c:186-298 ; read: lib/entropy_common.c:41-144)
 * ---------------------------------------------------------------------------------------- */
size_t orc_fse_ncount_write_bound(unsigned maxSV, unsigned tl)           /* :186-190 */
{
    return maxSV ? (((maxSV + 1) * tl) >> 3) + 3 : 512;
}

size_t orc_fse_write_ncount(void* dst, size_t cap, const short* norm, unsigned maxSV, unsigned tl)
{
    u8* const out = (u8*)dst;
    long o = 0;
    long const lim = (long)cap - 2;                    /* "out > oend-2" guards, only when !safe */
    int const safe = cap >= orc_fse_ncount_write_bound(maxSV, tl);   /* :294-297 */
    unsigned const alphabet = maxSV + 1;
    int const tsize = 1 << tl;
    int remaining = tsize + 1, threshold = tsize, nbBits = (int)tl + 1;
    u32 acc; int nacc;
    unsigned sym = 0; int prevZero = 0;

    if (tl > FSE_MAX_TL) return ERR(tableLog_tooLarge);   /* :291-292 */
    if (tl < FSE_MIN_TL) return ERR(GENERIC);

    acc = tl - FSE_MIN_TL; nacc = 4;                   /* :211-212 */
#define NC_SPILL() do { if (!safe && o > lim) return ERR(dstSize_tooSmall); \
                        out[o] = (u8)acc; out[o + 1] = (u8)(acc >> 8); o += 2; acc >>= 16; } while (0)
    while (sym < alphabet && remaining > 1) {          /* :219 */
        if (prevZero) {                                /* :220-249 : run of zero counts */
            unsigned from = sym;
            while (sym < alphabet && !norm[sym]) sym++;
            if (sym == alphabet) break;
            while (sym >= from + 24) { from += 24; acc += 0xFFFFu << nacc; NC_SPILL(); }
            while (sym >= from + 3) { from += 3; acc += 3u << nacc; nacc += 2; }
            acc += (sym - from) << nacc; nacc += 2;
            if (nacc > 16) { NC_SPILL(); nacc -= 16; }
        }
        {   int c = norm[sym++];                        /* :250-262 */
            int const max = (2 * threshold - 1) - remaining;
            remaining -= c < 0 ? -c : c;
            c++;
            if (c >= threshold) c += max;
            acc += (u32)c << nacc;
            nacc += nbBits;
            nacc -= (c < max);
            prevZero = (c == 1);
            if (remaining < 1) return ERR(GENERIC);
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (nacc > 16) { NC_SPILL(); nacc -= 16; }
    }
    if (remaining != 1) return ERR(GENERIC);           /* :273 */
    if (!safe && o > lim) return ERR(dstSize_tooSmall);
    out[o] = (u8)acc; out[o + 1] = (u8)(acc >> 8);     /* :280-282 */
    o += (nacc + 7) / 8;
#undef NC_SPILL
    return (size_t)o;
}

size_t orc_fse_read_ncount(short* norm, unsigned* maxSVPtr, unsigned* tlPtr, const void* src, size_t hbSize)
{
    const u8* const in = (const u8*)src;
    long ip = 0;
    long const iend = (long)hbSize;
    int nbBits, remaining, threshold, bitCount;
    u32 bits;
    unsigned charnum = 0;
    int prevZero = 0;

    if (hbSize < 4) {                                  /* :55-64 */
        u8 tmp[4] = { 0, 0, 0, 0 };
        size_t r;
        memcpy(tmp, in, hbSize);
        r = orc_fse_read_ncount(norm, maxSVPtr, tlPtr, tmp, 4);
        if (orc_is_error(r)) return r;
        if (r > hbSize) return ERR(corruption_detected);
        return r;
    }
    memset(norm, 0, (*maxSVPtr + 1) * sizeof(norm[0]));   /* :68 */
    bits = le32(in);
    nbBits = (int)(bits & 0xF) + FSE_MIN_TL;
    if (nbBits > FSE_ABS_MAX_TL) return ERR(tableLog_tooLarge);
    bits >>= 4; bitCount = 4;
    *tlPtr = (unsigned)nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;

    while ((remaining > 1) & (charnum <= *maxSVPtr)) {  /* :79 */
        if (prevZero) {                                /* :80-107 */
            unsigned n0 = charnum;
            while ((bits & 0xFFFF) == 0xFFFF) {
                n0 += 24;
                if (ip < iend - 5) { ip += 2; bits = le32(in + ip) >> bitCount; }
                else { bits >>= 16; bitCount += 16; }
            }
            while ((bits & 3) == 3) { n0 += 3; bits >>= 2; bitCount += 2; }
            n0 += bits & 3; bitCount += 2;
            if (n0 > *maxSVPtr) return ERR(maxSymbolValue_tooSmall);
            while (charnum < n0) norm[charnum++] = 0;
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) {
                ip += bitCount >> 3; bitCount &= 7; bits = le32(in + ip) >> bitCount;
            } else bits >>= 2;
        }
        {   int const max = (2 * threshold - 1) - remaining;   /* :108-135 */
            int c;
            if ((bits & (u32)(threshold - 1)) < (u32)max) { c = (int)(bits & (u32)(threshold - 1)); bitCount += nbBits - 1; }
            else { c = (int)(bits & (u32)(2 * threshold - 1)); if (c >= threshold) c -= max; bitCount += nbBits; }
            c--;
            remaining -= c < 0 ? -c : c;
            norm[charnum++] = (short)c;
            prevZero = !c;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= (int)(8 * (iend - 4 - ip)); ip = iend - 4; }
            bits = le32(in + ip) >> (bitCount & 31);
        }
    }
    if (remaining != 1) return ERR(corruption_detected);
    if (bitCount > 32) return ERR(corruption_detected);
    *maxSVPtr = charnum - 1;
    ip += (bitCount + 7) >> 3;
    return (size_t)ip;
}

/* ------------------------------------------------------------------------------------------
 *  FSE tables  (CTable: lib/fse_compress.c:66-169 ; DTable: lib/fse_decompress.c:71-126)
 *  in-memory layouts: SURVEY A.2
 * ---------------------------------------------------------------------------------------- */
#define FSE_STEP(ts) (((ts) >> 1) + ((ts) >> 3) + 3)    /* lib/fse.h:683 */

/* symbol of every table cell after the spread; returns 0 or GENERIC when the walk does not close */
static size_t fse_spread(u8* cellSym, const short* norm, unsigned maxSV, unsigned tl)
{
    u32 const ts = 1u << tl, mask = ts - 1, step = FSE_STEP(ts);
    u32 high = ts - 1, pos = 0, s;
    for (s = 0; s <= maxSV; s++) if (norm[s] == -1) cellSym[high--] = (u8)s;   /* low-proba symbols at the top */
    for (s = 0; s <= maxSV; s++) {
        int k;
        for (k = 0; k < norm[s]; k++) {
            cellSym[pos] = (u8)s;
            do pos = (pos + step) & mask; while (pos > high);
        }
    }
    return pos == 0 ? 0 : ERR(GENERIC);
}

size_t orc_fse_build_ctable(u32* ct, const short* norm, unsigned maxSV, unsigned tl)
{
    u32 const ts = 1u << tl;
    u16* const head = (u16*)ct;
    u16* const stateTable = head + 2;                                  /* :73 */
    u32* const tt = ct + 1 + (tl ? ts >> 1 : 1);                       /* :74 : pairs {deltaFindState, deltaNbBits} */
    u32 first[FSE_MAX_SV + 2];
    u8* cellSym = (u8*)malloc(ts);
    u32 u, s;
    int total = 0;

    head[0] = (u16)tl; head[1] = (u16)maxSV;                           /* :84-85 */
    first[0] = 0;
    for (s = 0; s <= maxSV; s++) first[s + 1] = first[s] + (u32)(norm[s] == -1 ? 1 : norm[s]);   /* :96-106 */
    fse_spread(cellSym, norm, maxSV, tl);                              /* :108-122 */
    for (u = 0; u < ts; u++) stateTable[first[cellSym[u]]++] = (u16)(ts + u);   /* :125-128 */
    for (s = 0; s <= maxSV; s++) {                                     /* :131-154 */
        int const n = norm[s];
        if (n == 0) { tt[2 * s + 1] = ((tl + 1) << 16) - (1u << tl); continue; }   /* deltaFindState left untouched */
        if (n == -1 || n == 1) {
            tt[2 * s + 1] = (tl << 16) - (1u << tl);
            tt[2 * s] = (u32)(total - 1);
            total++;
        } else {
            u32 const maxBitsOut = tl - hibit((u32)n - 1);
            tt[2 * s + 1] = (maxBitsOut << 16) - ((u32)n << maxBitsOut);
            tt[2 * s] = (u32)(total - n);
            total += n;
        }
    }
    free(cellSym);
    return 0;
}

size_t orc_fse_build_ctable_raw(u32* ct, unsigned nbBits)              /* lib/fse_compress.c:498-528 */
{
    u32 const ts = 1u << nbBits;
    u16* const head = (u16*)ct;
    u32* const tt = ct + 1 + (ts >> 1);
    u32 s;
    if (nbBits < 1) return ERR(GENERIC);
    head[0] = (u16)nbBits; head[1] = (u16)(ts - 1);
    for (s = 0; s < ts; s++) head[2 + s] = (u16)(ts + s);
    for (s = 0; s < ts; s++) { tt[2 * s + 1] = (nbBits << 16) - ts; tt[2 * s] = s - 1; }
    return 0;
}

size_t orc_fse_build_dtable(u32* dt, const short* norm, unsigned maxSV, unsigned tl)
{
    u32 const ts = 1u << tl;
    u16* const head = (u16*)dt;
    u8* const cells = (u8*)(dt + 1);                   /* {u16 newState; u8 symbol; u8 nbBits} : lib/fse.h:570-575 */
    u16 next[FSE_MAX_SV + 1];
    u8* cellSym;
    u32 u, s;
    u16 fast = 1;

    if (maxSV > FSE_MAX_SV) return ERR(maxSymbolValue_tooLarge);      /* :82-83 */
    if (tl > FSE_MAX_TL) return ERR(tableLog_tooLarge);
    for (s = 0; s <= maxSV; s++) {                     /* :86-99 */
        if (norm[s] == -1) next[s] = 1;
        else { if (norm[s] >= (short)(1 << (tl - 1))) fast = 0; next[s] = (u16)norm[s]; }
    }
    head[0] = (u16)tl; head[1] = fast;
    cellSym = (u8*)malloc(ts);
    if (fse_spread(cellSym, norm, maxSV, tl)) { free(cellSym); return ERR(GENERIC); }   /* :101-113 */
    for (u = 0; u < ts; u++) {                         /* :116-123 */
        u32 const ns = next[cellSym[u]]++;
        u32 const nb = tl - hibit(ns);
        u16 const newState = (u16)((ns << nb) - ts);
        cells[4 * u + 0] = (u8)newState; cells[4 * u + 1] = (u8)(newState >> 8);
        cells[4 * u + 2] = cellSym[u];
        cells[4 * u + 3] = (u8)nb;
    }
    free(cellSym);
    return 0;
}

size_t orc_fse_build_dtable_raw(u32* dt, unsigned nbBits)              /* lib/fse_decompress.c:152-176 */
{
    u16* const head = (u16*)dt;
    u8* const cells = (u8*)(dt + 1);
    u32 s;
    if (nbBits < 1) return ERR(GENERIC);
    head[0] = (u16)nbBits; head[1] = 1;
    for (s = 0; s < (1u << nbBits); s++) { cells[4 * s] = 0; cells[4 * s + 1] = 0; cells[4 * s + 2] = (u8)s; cells[4 * s + 3] = (u8)nbBits; }
    return 0;
}

size_t orc_fse_build_dtable_rle(u32* dt, u8 symbol)                    /* lib/fse_decompress.c:134-149 */
{
    u8* const cells = (u8*)(dt + 1);
    dt[0] = 0;
    cells[0] = 0; cells[1] = 0; cells[2] = symbol; cells[3] = 0;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 *  a2: FSE_compress_usingCTable  (lib/fse_compress.c:554-623 ; lib/fse.h:503-527 )
 * ---------------------------------------------------------------------------------------- */
size_t orc_fse_compress_using_ctable(void* dst, size_t cap, const void* src, size_t n, const u32* ct)
{
    const u8* const s = (const u8*)src;
    const u16* const head = (const u16*)ct;
    u32 const tl = head[0];
    const u16* const stateTable = head + 2;
    const u32* const tt = ct + 1 + (tl ? (1u << (tl - 1)) : 1);
    u32 chain[2];          /* chain[0] = symbols at even distance j from the end, chain[1] = odd j */
    bitw_t w;
    size_t j;

    if (n <= 2) return 0;                              /* :566 */
    if (cap <= 8) return 0;                            /* :567-568 via bitstream.h:191 */
    bitw_init(&w, dst, cap);
    for (j = 0; j < 2; j++) {                          /* FSE_initCState2, fse.h:503-512 : no bits emitted */
        u32 const dfs = tt[2 * s[n - 1 - j]], dnb = tt[2 * s[n - 1 - j] + 1];
        u32 const nb = (dnb + (1u << 15)) >> 16;
        u32 const v = (nb << 16) - dnb;
        chain[j] = stateTable[(v >> nb) + dfs];
    }
    for (j = 2; j < n; j++) {                          /* FSE_encodeSymbol, fse.h:514-521 */
        u32 const sym = s[n - 1 - j];
        u32 const dfs = tt[2 * sym], dnb = tt[2 * sym + 1];
        u32 const x = chain[j & 1];
        u32 const nb = (x + dnb) >> 16;
        bitw_put(&w, x, nb);
        chain[j & 1] = stateTable[(x >> nb) + dfs];
    }
    /* :608-609 : CState2 first, then CState1.  n even: CState2 is the even-j chain (:577-580);
     * n odd: CState1 is (:572-576). */
    bitw_put(&w, chain[(n & 1) ? 1 : 0], tl);
    bitw_put(&w, chain[(n & 1) ? 0 : 1], tl);
    return bitw_close(&w);
}

/* ------------------------------------------------------------------------------------------
 *  a3: FSE_decompress_usingDTable  (lib/fse_decompress.c:178-252 ; lib/fse.h:577-622)
 * ---------------------------------------------------------------------------------------- */
typedef struct { u32 state; } dstate_t;

static u8 fse_step(u32* state, bitr_t* r, const u8* cells, int fast)   /* fse.h:600-622 */
{
    const u8* c = cells + 4 * (size_t)*state;
    u32 const nb = c[3];
    u64 const low = fast ? bitr_read_fast(r, nb) : bitr_read(r, nb);
    *state = (u32)(((u32)c[0] | ((u32)c[1] << 8)) + low);
    return c[2];
}

size_t orc_fse_decompress_using_dtable(void* dst, size_t cap, const void* cSrc, size_t cSize, const u32* dt)
{
    u8* const out = (u8*)dst;
    const u16* const head = (const u16*)dt;
    u32 const tl = head[0];
    int const fast = head[1] != 0;
    const u8* const cells = (const u8*)(dt + 1);
    long op = 0;
    long const omax = (long)cap;
    bitr_t r;
    u32 s1, s2;

    {   size_t const e = bitr_init(&r, cSrc, cSize); if (orc_is_error(e)) return e; }
    s1 = (u32)bitr_read(&r, tl); bitr_reload(&r);      /* fse.h:577-584 */
    s2 = (u32)bitr_read(&r, tl); bitr_reload(&r);

    for (;;) {                                         /* :201-218 : 4 symbols per refill on 64-bit */
        int const st = bitr_reload(&r);
        if (!((st == BR_UNFINISHED) & (op < omax - 3))) break;
        out[op + 0] = fse_step(&s1, &r, cells, fast);
        out[op + 1] = fse_step(&s2, &r, cells, fast);
        out[op + 2] = fse_step(&s1, &r, cells, fast);
        out[op + 3] = fse_step(&s2, &r, cells, fast);
        op += 4;
    }
    for (;;) {                                         /* :222-235 */
        if (op > omax - 2) return ERR(dstSize_tooSmall);
        out[op++] = fse_step(&s1, &r, cells, fast);
        if (bitr_reload(&r) == BR_OVERFLOW) { out[op++] = fse_step(&s2, &r, cells, fast); break; }
        if (op > omax - 2) return ERR(dstSize_tooSmall);
        out[op++] = fse_step(&s2, &r, cells, fast);
        if (bitr_reload(&r) == BR_OVERFLOW) { out[op++] = fse_step(&s1, &r, cells, fast); break; }
    }
    return (size_t)op;
}

/* ------------------------------------------------------------------------------------------
 *  g4: one-shot FSE block  (lib/fse_compress.c:632-698 ; lib/fse_decompress.c:255-283)
 * ---------------------------------------------------------------------------------------- */
size_t orc_fse_compress2(void* dst, size_t cap, const void* src, size_t n, unsigned maxSV, unsigned tl)
{
    u8* const out = (u8*)dst;
    unsigned count[FSE_MAX_SV + 1];
    short norm[FSE_MAX_SV + 1];
    u32 ct[1 + (1 << (FSE_MAX_TL - 1)) + (FSE_MAX_SV + 1) * 2];
    size_t h, c;

    if (tl > FSE_MAX_TL) return ERR(tableLog_tooLarge);   /* :691 */
    if (n <= 1) return 0;                              /* :647 */
    if (!maxSV) maxSV = FSE_MAX_SV;
    if (!tl) tl = FSE_DEFAULT_TL;
    {   size_t const top = orc_hist_count(count, &maxSV, src, n);   /* :652-655 */
        if (orc_is_error(top)) return top;
        if (top == n) return 1;
        if (top == 1) return 0;
        if (top < (n >> 7)) return 0;
    }
    tl = orc_fse_optimal_tablelog(tl, n, maxSV, 2);    /* :658 */
    {   size_t const e = orc_fse_normalize_count(norm, tl, count, n, maxSV); if (orc_is_error(e)) return e; }
    h = orc_fse_write_ncount(out, cap, norm, maxSV, tl);   /* :662 */
    if (orc_is_error(h)) return h;
    orc_fse_build_ctable(ct, norm, maxSV, tl);         /* :667 */
    c = orc_fse_compress_using_ctable(out + h, cap - h, src, n, ct);
    if (c == 0) return 0;
    if (h + c >= n - 1) return 0;                      /* :674 */
    return h + c;
}

size_t orc_fse_decompress(void* dst, size_t cap, const void* cSrc, size_t cSize)
{
    short norm[FSE_MAX_SV + 1];
    u32 dt[1 + (1 << FSE_MAX_TL)];
    unsigned tl, maxSV = FSE_MAX_SV;
    size_t const h = orc_fse_read_ncount(norm, &maxSV, &tl, cSrc, cSize);   /* :264 */
    if (orc_is_error(h)) return h;
    if (tl > FSE_MAX_TL) return ERR(tableLog_tooLarge);
    {   size_t const e = orc_fse_build_dtable(dt, norm, maxSV, tl); if (orc_is_error(e)) return e; }
    return orc_fse_decompress_using_dtable(dst, cap, (const u8*)cSrc + h, cSize - h, dt);
}

/* ------------------------------------------------------------------------------------------
 *  g5: Huffman code construction  (lib/huf_compress.c:202-410)
 *  celt[s] = val | nbBits << 16   (struct HUF_CElt_s {U16 val; BYTE nbBits;}, :106-109)
 * ---------------------------------------------------------------------------------------- */
typedef struct { u32 count; u16 parent; u8 byte; u8 nbBits; } hnode_t;

static u32 huf_limit_height(hnode_t* node, u32 lastNonNull, u32 maxNbBits)   /* :215-291 */
{
    u32 const largest = node[lastNonNull].nbBits;
    if (largest <= maxNbBits) return largest;
    {   int debt = 0;
        u32 const unit = 1u << (largest - maxNbBits);
        int n = (int)lastNonNull;
        while (node[n].nbBits > maxNbBits) {
            debt += (int)(unit - (1u << (largest - node[n].nbBits)));
            node[n].nbBits = (u8)maxNbBits;
            n--;
        }
        while (node[n].nbBits == maxNbBits) n--;
        debt >>= (largest - maxNbBits);
        {   u32 const NONE = 0xF0F0F0F0u;
            u32 rankLast[HUF_MAX_TL + 2];
            memset(rankLast, 0xF0, sizeof(rankLast));
            {   u32 cur = maxNbBits; int pos;
                for (pos = n; pos >= 0; pos--) {
                    if (node[pos].nbBits >= cur) continue;
                    cur = node[pos].nbBits;
                    rankLast[maxNbBits - cur] = (u32)pos;
                }
            }
            while (debt > 0) {
                u32 dec = hibit((u32)debt) + 1;
                for (; dec > 1; dec--) {
                    u32 const hi = rankLast[dec], lo = rankLast[dec - 1];
                    if (hi == NONE) continue;
                    if (lo == NONE) break;
                    if (node[hi].count <= 2 * node[lo].count) break;
                }
                while (dec <= HUF_MAX_TL && rankLast[dec] == NONE) dec++;
                debt -= 1 << (dec - 1);
                if (rankLast[dec - 1] == NONE) rankLast[dec - 1] = rankLast[dec];
                node[rankLast[dec]].nbBits++;
                if (rankLast[dec] == 0) rankLast[dec] = NONE;
                else {
                    rankLast[dec]--;
                    if (node[rankLast[dec]].nbBits != maxNbBits - dec) rankLast[dec] = NONE;
                }
            }
            while (debt < 0) {
                if (rankLast[1] == NONE) {
                    while (node[n].nbBits == maxNbBits) n--;
                    node[n + 1].nbBits--;
                    rankLast[1] = (u32)(n + 1);
                    debt++;
                    continue;
                }
                node[rankLast[1] + 1].nbBits--;
                rankLast[1]++;
                debt++;
            }
        }
    }
    return maxNbBits;
}

static void huf_sort_nodes(hnode_t* node, const unsigned* count, u32 maxSV)   /* :307-329 */
{
    struct { u32 base, cur; } rank[32];
    u32 n;
    memset(rank, 0, sizeof(rank));
    for (n = 0; n <= maxSV; n++) rank[hibit(count[n] + 1)].base++;
    for (n = 30; n > 0; n--) rank[n - 1].base += rank[n].base;
    for (n = 0; n < 32; n++) rank[n].cur = rank[n].base;
    for (n = 0; n <= maxSV; n++) {
        u32 const c = count[n];
        u32 const r = hibit(c + 1) + 1;
        u32 pos = rank[r].cur++;
        while (pos > rank[r].base && c > node[pos - 1].count) { node[pos] = node[pos - 1]; pos--; }
        node[pos].count = c;
        node[pos].byte = (u8)n;
    }
}

size_t orc_huf_build_ctable(u32* celt, const unsigned* count, unsigned maxSV, unsigned maxNbBits)   /* :338-410 */
{
    enum { START = HUF_MAX_SV + 1 };
    hnode_t node0[2 * HUF_MAX_SV + 2];
    hnode_t* const node = node0 + 1;
    int last, lowS, lowN, nodeNb = START, root, n;

    if (maxNbBits == 0) maxNbBits = HUF_DEFAULT_TL;
    if (maxSV > HUF_MAX_SV) return ERR(maxSymbolValue_tooLarge);
    memset(node0, 0, sizeof(node0));
    huf_sort_nodes(node, count, maxSV);

    last = (int)maxSV;
    while (node[last].count == 0) last--;
    lowS = last; root = nodeNb + lowS - 1; lowN = nodeNb;
    node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
    node[lowS].parent = node[lowS - 1].parent = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= root; n++) node[n].count = 1u << 30;
    node0[0].count = 1u << 31;
    while (nodeNb <= root) {
        int const a = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        int const b = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        node[nodeNb].count = node[a].count + node[b].count;
        node[a].parent = node[b].parent = (u16)nodeNb;
        nodeNb++;
    }
    node[root].nbBits = 0;
    for (n = root - 1; n >= START; n--) node[n].nbBits = (u8)(node[node[n].parent].nbBits + 1);
    for (n = 0; n <= last; n++) node[n].nbBits = (u8)(node[node[n].parent].nbBits + 1);

    maxNbBits = huf_limit_height(node, (u32)last, maxNbBits);

    {   u16 perRank[HUF_MAX_TL + 1] = { 0 }, valRank[HUF_MAX_TL + 1] = { 0 };
        u8 bitsOf[HUF_MAX_SV + 1];
        int const alphabet = (int)maxSV + 1;
        if (maxNbBits > HUF_MAX_TL) return ERR(GENERIC);
        for (n = 0; n <= last; n++) perRank[node[n].nbBits]++;
        {   u16 min = 0;
            for (n = (int)maxNbBits; n > 0; n--) { valRank[n] = min; min = (u16)(min + perRank[n]); min >>= 1; }
        }
        for (n = 0; n < alphabet; n++) bitsOf[node[n].byte] = node[n].nbBits;
        for (n = 0; n < alphabet; n++) celt[n] = (u32)(valRank[bitsOf[n]]++) | ((u32)bitsOf[n] << 16);
    }
    return maxNbBits;
}

/* ------------------------------------------------------------------------------------------
 *  g6: Huffman table header  (lib/huf_compress.c:63-147 ; lib/entropy_common.c:154-215 ;
 *      lib/huf_decompress.c:118-185)
 * ---------------------------------------------------------------------------------------- */
static size_t huf_compress_weights(void* dst, size_t cap, const u8* weights, size_t n)   /* :63-103 */
{
    u8* const out = (u8*)dst;
    unsigned maxSV = HUF_MAX_TL, tl = 6, s;
    unsigned count[HUF_MAX_TL + 1];
    short norm[HUF_MAX_TL + 1];
    u32 ct[1 + (1 << 5) + (HUF_MAX_TL + 1) * 2];
    size_t h, c, i, best = 0;

    if (n <= 1) return 0;
    memset(count, 0, sizeof(count));                   /* HIST_count_simple, hist.c:29-54 */
    for (i = 0; i < n; i++) count[weights[i]]++;
    while (!count[maxSV]) maxSV--;
    for (s = 0; s <= maxSV; s++) if (count[s] > best) best = count[s];
    if (best == n) return 1;
    if (best == 1) return 0;

    tl = orc_fse_optimal_tablelog(tl, n, maxSV, 2);
    {   size_t const e = orc_fse_normalize_count(norm, tl, count, n, maxSV); if (orc_is_error(e)) return e; }
    h = orc_fse_write_ncount(out, cap, norm, maxSV, tl);
    if (orc_is_error(h)) return h;
    orc_fse_build_ctable(ct, norm, maxSV, tl);
    c = orc_fse_compress_using_ctable(out + h, cap - h, weights, n, ct);
    if (c == 0) return 0;
    return h + c;
}

size_t orc_huf_write_ctable(void* dst, size_t cap, const u32* celt, unsigned maxSV, unsigned huffLog)   /* :114-147 */
{
    u8* const out = (u8*)dst;
    u8 w[HUF_MAX_SV + 1];
    u32 n;
    if (maxSV > HUF_MAX_SV) return ERR(maxSymbolValue_tooLarge);
    for (n = 0; n < maxSV; n++) { u32 const nb = (celt[n] >> 16) & 0xFF; w[n] = nb ? (u8)(huffLog + 1 - nb) : 0; }
    {   size_t const hs = huf_compress_weights(out + 1, cap - 1, w, maxSV);
        if (orc_is_error(hs)) return hs;
        if ((hs > 1) & (hs < maxSV / 2)) { out[0] = (u8)hs; return hs + 1; }
    }
    if (maxSV > 128) return ERR(GENERIC);
    if (((maxSV + 1) / 2) + 1 > cap) return ERR(dstSize_tooSmall);
    out[0] = (u8)(128 + (maxSV - 1));
    w[maxSV] = 0;
    for (n = 0; n < maxSV; n += 2) out[(n / 2) + 1] = (u8)((w[n] << 4) + w[n + 1]);
    return ((maxSV + 1) / 2) + 1;
}

size_t orc_huf_read_stats(u8* w, size_t hwSize, u32* rankStats, u32* nbSymbolsPtr, u32* tlPtr, const void* src, size_t srcSize)
{
    const u8* ip = (const u8*)src;
    size_t iSize, oSize, n;
    u32 total = 0;
    if (!srcSize) return ERR(srcSize_wrong);
    iSize = ip[0];
    if (iSize >= 128) {                                /* raw 4-bit weights, entropy_common.c:168-178 */
        oSize = iSize - 127;
        iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return ERR(srcSize_wrong);
        if (oSize >= hwSize) return ERR(corruption_detected);
        for (n = 0; n < oSize; n += 2) { w[n] = ip[1 + n / 2] >> 4; w[n + 1] = ip[1 + n / 2] & 15; }
    } else {                                           /* FSE-compressed weights, :179-184 */
        short norm[FSE_MAX_SV + 1];
        u32 dt[1 + (1 << 6)];
        unsigned tl, maxSV = FSE_MAX_SV;
        size_t h;
        if (iSize + 1 > srcSize) return ERR(srcSize_wrong);
        h = orc_fse_read_ncount(norm, &maxSV, &tl, ip + 1, iSize);    /* FSE_decompress_wksp(..., maxLog 6) */
        if (orc_is_error(h)) return h;
        if (tl > 6) return ERR(tableLog_tooLarge);
        {   size_t const e = orc_fse_build_dtable(dt, norm, maxSV, tl); if (orc_is_error(e)) return e; }
        oSize = orc_fse_decompress_using_dtable(w, hwSize - 1, ip + 1 + h, iSize - h, dt);
        if (orc_is_error(oSize)) return oSize;
    }
    memset(rankStats, 0, (HUF_MAX_TL + 1) * sizeof(u32));
    for (n = 0; n < oSize; n++) {
        if (w[n] >= HUF_MAX_TL) return ERR(corruption_detected);
        rankStats[w[n]]++;
        total += (1u << w[n]) >> 1;
    }
    if (total == 0) return ERR(corruption_detected);
    {   u32 const tl = hibit(total) + 1;
        if (tl > HUF_MAX_TL) return ERR(corruption_detected);
        *tlPtr = tl;
        {   u32 const rest = (1u << tl) - total;
            u32 const lastW = hibit(rest) + 1;
            if ((1u << hibit(rest)) != rest) return ERR(corruption_detected);
            w[oSize] = (u8)lastW;
            rankStats[lastW]++;
        }
    }
    if ((rankStats[1] < 2) || (rankStats[1] & 1)) return ERR(corruption_detected);
    *nbSymbolsPtr = (u32)(oSize + 1);
    return iSize + 1;
}

size_t orc_huf_read_dtable_x1(u32* dtable, const void* src, size_t srcSize)   /* huf_decompress.c:118-185 */
{
    u8* const desc = (u8*)dtable;                      /* {maxTableLog, tableType, tableLog, reserved} :101 */
    u8* const cells = (u8*)(dtable + 1);               /* {byte, nbBits} :116 */
    u8 w[HUF_MAX_SV + 1];
    u32 rankVal[HUF_ABS_MAX_TL + 1];
    u32 tl = 0, nbSym = 0, n;
    size_t const iSize = orc_huf_read_stats(w, HUF_MAX_SV + 1, rankVal, &nbSym, &tl, src, srcSize);
    if (orc_is_error(iSize)) return iSize;
    if (tl > (u32)desc[0] + 1) return ERR(tableLog_tooLarge);
    desc[1] = 0; desc[2] = (u8)tl;
    {   u32 next = 0;
        for (n = 1; n < tl + 1; n++) { u32 const cur = next; next += rankVal[n] << (n - 1); rankVal[n] = cur; }
    }
    for (n = 0; n < nbSym; n++) {
        u32 const wt = w[n];
        u32 const len = (1u << wt) >> 1;
        u32 u;
        for (u = rankVal[wt]; u < rankVal[wt] + len; u++) { cells[2 * u] = (u8)n; cells[2 * u + 1] = (u8)(tl + 1 - wt); }
        rankVal[wt] += len;
    }
    return iSize;
}

/* ------------------------------------------------------------------------------------------
 *  a4: Huffman encoders  (lib/huf_compress.c:457-608 )
 * ---------------------------------------------------------------------------------------- */
size_t orc_huf_compress1x_using_ctable(void* dst, size_t cap, const void* src, size_t n, const u32* celt)
{
    const u8* const s = (const u8*)src;
    bitw_t w;
    size_t i;
    if (cap < 8) return 0;                             /* :470 */
    if (cap <= 8) return 0;                            /* :471-472 */
    bitw_init(&w, dst, cap);
    for (i = n; i > 0; i--) {                          /* :474-499 : net effect = symbols last -> first */
        u32 const e = celt[s[i - 1]];
        bitw_put(&w, e & 0xFFFF, (e >> 16) & 0xFF);
    }
    return bitw_close(&w);
}

size_t orc_huf_compress4x_using_ctable(void* dst, size_t cap, const void* src, size_t n, const u32* celt)   /* :552-603 */
{
    u8* const out = (u8*)dst;
    const u8* ip = (const u8*)src;
    size_t const seg = (n + 3) / 4;
    size_t op = 6;
    int k;
    if (cap < 6 + 1 + 1 + 1 + 8) return 0;
    if (n < 12) return 0;
    for (k = 0; k < 4; k++) {
        size_t const len = k < 3 ? seg : n - 3 * seg;
        size_t const c = orc_huf_compress1x_using_ctable(out + op, cap - op, ip, len, celt);
        if (c == 0) return 0;
        if (k < 3) { out[2 * k] = (u8)c; out[2 * k + 1] = (u8)(c >> 8); }
        op += c; ip += len;
    }
    return op;
}

/* ------------------------------------------------------------------------------------------
 *  a5: Huffman X1 decoders  (lib/huf_decompress.c:194-354)
 * ---------------------------------------------------------------------------------------- */
static u8 hufx1_step(bitr_t* r, const u8* cells, u32 dtLog)            /* :194-201 */
{
    size_t const v = (size_t)bitr_peek_fast(r, dtLog);
    r->used += cells[2 * v + 1];
    return cells[2 * v];
}

static void hufx1_stream(u8* p, u8* const pEnd, bitr_t* r, const u8* cells, u32 dtLog)   /* :214-237 */
{
    while ((bitr_reload(r) == BR_UNFINISHED) & (p < pEnd - 3)) {
        p[0] = hufx1_step(r, cells, dtLog); p[1] = hufx1_step(r, cells, dtLog);
        p[2] = hufx1_step(r, cells, dtLog); p[3] = hufx1_step(r, cells, dtLog);
        p += 4;
    }
    while (p < pEnd) *p++ = hufx1_step(r, cells, dtLog);
}

size_t orc_huf_decompress1x1_using_dtable(void* dst, size_t dstSize, const void* cSrc, size_t cSize, const u32* dtable)   /* :239-260 */
{
    const u8* const desc = (const u8*)dtable;
    bitr_t r;
    size_t const e = bitr_init(&r, cSrc, cSize);
    if (orc_is_error(e)) return e;
    hufx1_stream((u8*)dst, (u8*)dst + dstSize, &r, (const u8*)(dtable + 1), desc[2]);
    if (!bitr_at_end(&r)) return ERR(corruption_detected);
    return dstSize;
}

size_t orc_huf_decompress4x1_using_dtable(void* dst, size_t dstSize, const void* cSrc, size_t cSize, const u32* dtable)   /* :262-354 */
{
    const u8* const in = (const u8*)cSrc;
    const u8* const desc = (const u8*)dtable;
    const u8* const cells = (const u8*)(dtable + 1);
    u32 const dtLog = desc[2];
    u8* const out = (u8*)dst;
    if (cSize < 10) return ERR(corruption_detected);
    {   size_t const l1 = le16(in), l2 = le16(in + 2), l3 = le16(in + 4);
        size_t const l4 = cSize - (l1 + l2 + l3 + 6);
        size_t const seg = (dstSize + 3) / 4;
        u8* const o2 = out + seg; u8* const o3 = o2 + seg; u8* const o4 = o3 + seg; u8* const oend = out + dstSize;
        u8 *p1 = out, *p2 = o2, *p3 = o3, *p4 = o4;
        bitr_t r1, r2, r3, r4;
        u32 go = 1;
        size_t e;
        if (l4 > cSize) return ERR(corruption_detected);
        e = bitr_init(&r1, in + 6, l1); if (orc_is_error(e)) return e;
        e = bitr_init(&r2, in + 6 + l1, l2); if (orc_is_error(e)) return e;
        e = bitr_init(&r3, in + 6 + l1 + l2, l3); if (orc_is_error(e)) return e;
        e = bitr_init(&r4, in + 6 + l1 + l2 + l3, l4); if (orc_is_error(e)) return e;
        for (; go & (p4 < oend - 3);) {                /* :310-331 : lock-step, 4 symbols x 4 streams */
            int k;
            for (k = 0; k < 4; k++) {
                *p1++ = hufx1_step(&r1, cells, dtLog); *p2++ = hufx1_step(&r2, cells, dtLog);
                *p3++ = hufx1_step(&r3, cells, dtLog); *p4++ = hufx1_step(&r4, cells, dtLog);
            }
            go &= bitr_reload_fast(&r1) == BR_UNFINISHED;
            go &= bitr_reload_fast(&r2) == BR_UNFINISHED;
            go &= bitr_reload_fast(&r3) == BR_UNFINISHED;
            go &= bitr_reload_fast(&r4) == BR_UNFINISHED;
        }
        if (p1 > o2) return ERR(corruption_detected);
        if (p2 > o3) return ERR(corruption_detected);
        if (p3 > o4) return ERR(corruption_detected);
        hufx1_stream(p1, o2, &r1, cells, dtLog);
        hufx1_stream(p2, o3, &r2, cells, dtLog);
        hufx1_stream(p3, o4, &r3, cells, dtLog);
        hufx1_stream(p4, oend, &r4, cells, dtLog);
        if (!(bitr_at_end(&r1) & bitr_at_end(&r2) & bitr_at_end(&r3) & bitr_at_end(&r4))) return ERR(corruption_detected);
    }
    return dstSize;
}

/* ------------------------------------------------------------------------------------------
 *  g7: one-shot Huff0 block  (lib/huf_compress.c:637-798 without table reuse ;
 *      lib/huf_decompress.c:1056-1081 taking the 4X1 branch, :431-450)
 * ---------------------------------------------------------------------------------------- */
size_t orc_huf_compress2(void* dst, size_t cap, const void* src, size_t n, unsigned maxSV, unsigned huffLog)
{
    u8* const out = (u8*)dst;
    unsigned count[HUF_MAX_SV + 1];
    u32 celt[HUF_MAX_SV + 1];
    size_t h, c;

    if (!n) return 0;
    if (!cap) return 0;
    if (n > HUF_BLOCK_MAX) return ERR(srcSize_wrong);
    if (huffLog > HUF_MAX_TL) return ERR(tableLog_tooLarge);
    if (maxSV > HUF_MAX_SV) return ERR(maxSymbolValue_tooLarge);
    if (!maxSV) maxSV = HUF_MAX_SV;
    if (!huffLog) huffLog = HUF_DEFAULT_TL;
    {   size_t const top = orc_hist_count(count, &maxSV, src, n);   /* :672-674 */
        if (orc_is_error(top)) return top;
        if (top == n) { out[0] = ((const u8*)src)[0]; return 1; }
        if (top <= (n >> 7) + 4) return 0;
    }
    huffLog = orc_fse_optimal_tablelog(huffLog, n, maxSV, 1);      /* HUF_optimalTableLog, :48-51 */
    {   size_t const mb = orc_huf_build_ctable(celt, count, maxSV, huffLog);
        if (orc_is_error(mb)) return mb;
        huffLog = (unsigned)mb;
    }
    h = orc_huf_write_ctable(out, cap, celt, maxSV, huffLog);      /* :703 */
    if (orc_is_error(h)) return h;
    if (h + 12ul >= n) return 0;                       /* :715 */
    c = orc_huf_compress4x_using_ctable(out + h, cap - h, src, n, celt);   /* :612-627 */
    if (c == 0) return 0;
    if (h + c >= n - 1) return 0;
    return h + c;
}

size_t orc_huf_decompress(void* dst, size_t dstSize, const void* cSrc, size_t cSize)
{
    u32 dtable[1 + (1 << HUF_MAX_TL)];
    size_t h;
    if (dstSize == 0) return ERR(dstSize_tooSmall);    /* :1063-1066 */
    if (cSize > dstSize) return ERR(corruption_detected);
    if (cSize == dstSize) { memcpy(dst, cSrc, dstSize); return dstSize; }
    if (cSize == 1) { memset(dst, *(const u8*)cSrc, dstSize); return dstSize; }
    memset(dtable, 0, sizeof(dtable));
    dtable[0] = (u32)(HUF_MAX_TL - 1) * 0x01000001u;   /* HUF_CREATE_STATIC_DTABLEX1(DTable, HUF_TABLELOG_MAX), lib/huf.h:146-147 */
    h = orc_huf_read_dtable_x1(dtable, cSrc, cSize);   /* :421-426 */
    if (orc_is_error(h)) return h;
    if (h >= cSize) return ERR(srcSize_wrong);
    return orc_huf_decompress4x1_using_dtable(dst, dstSize, (const u8*)cSrc + h, cSize - h, dtable);
}

/* ------------------------------------------------------------------------------------------
 *  batch drivers
 * ---------------------------------------------------------------------------------------- */
static double wall_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static void use_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

double orc_compress_batch(int codec, const u8* src, size_t srcStride, size_t srcSize, u8* dst, size_t dstStride,
                          size_t dstCapacity, u64* results, size_t nBlocks, unsigned maxSV, unsigned tl, int nthreads)
{
    double t0; long b;
    use_threads(nthreads);
    t0 = wall_s();
#pragma omp parallel for schedule(static)
    for (b = 0; b < (long)nBlocks; b++)
        results[b] = codec == 0 ? orc_fse_compress2(dst + (size_t)b * dstStride, dstCapacity, src + (size_t)b * srcStride, srcSize, maxSV, tl)
                                : orc_huf_compress2(dst + (size_t)b * dstStride, dstCapacity, src + (size_t)b * srcStride, srcSize, maxSV, tl);
    return wall_s() - t0;
}

double orc_decompress_batch(int codec, const u8* cSrc, size_t cStride, const u64* cSizes, u8* dst, size_t dstStride,
                            size_t dstSize, u64* results, size_t nBlocks, int nthreads)
{
    double t0; long b;
    use_threads(nthreads);
    t0 = wall_s();
#pragma omp parallel for schedule(static)
    for (b = 0; b < (long)nBlocks; b++)
        results[b] = codec == 0 ? orc_fse_decompress(dst + (size_t)b * dstStride, dstSize, cSrc + (size_t)b * cStride, (size_t)cSizes[b])
                                : orc_huf_decompress(dst + (size_t)b * dstStride, dstSize, cSrc + (size_t)b * cStride, (size_t)cSizes[b]);
    return wall_s() - t0;
}

/* multi-core CPU baseline driver (bench.py cpu_baseline, kind "port") */
#define now_s wall_s
#define CPUB_NAME orc_bench_roundtrip
#define CPUB_STREAM_NAME orc_stream_bandwidth
#define CPUB_COMPRESS(codec, d, cap, s, n, msv, tl) ((codec) == 0 ? orc_fse_compress2(d, cap, s, n, msv, tl) : orc_huf_compress2(d, cap, s, n, msv, tl))
#define CPUB_DECOMPRESS(codec, d, n, s, cs) ((codec) == 0 ? orc_fse_decompress(d, n, s, cs) : orc_huf_decompress(d, n, s, cs))
#include "cpu_bench.h"
#undef now_s

/* ------------------------------------------------------------------------------------------
 *  XXH64 (public spec) -- only for the Appendix-B known-answer table
 * ---------------------------------------------------------------------------------------- */
#define XP1 11400714785074694791ULL
#define XP2 14029467366897019727ULL
#define XP3 1609587929392839161ULL
#define XP4 9650029242287828579ULL
#define XP5 2870177450012600261ULL
static u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
static u64 xround(u64 acc, u64 in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
static u64 xmerge(u64 acc, u64 v) { acc ^= xround(0, v); return acc * XP1 + XP4; }

u64 orc_xxh64(const void* data, size_t len, u64 seed)
{
    const u8* p = (const u8*)data;
    const u8* const end = p + len;
    u64 h;
    if (len >= 32) {
        u64 v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        do { v1 = xround(v1, le64(p)); v2 = xround(v2, le64(p + 8)); v3 = xround(v3, le64(p + 16)); v4 = xround(v4, le64(p + 24)); p += 32; } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else h = seed + XP5;
    h += (u64)len;
    while (p + 8 <= end) { h ^= xround(0, le64(p)); h = rotl64(h, 27) * XP1 + XP4; p += 8; }
    if (p + 4 <= end) { h ^= (u64)le32(p) * XP1; h = rotl64(h, 23) * XP2 + XP3; p += 4; }
    while (p < end) { h ^= (*p) * XP5; h = rotl64(h, 11) * XP1; p++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}


/* =====================================================================================================
 *  .fse frame: programs/fileio.c:266-432 (writer), :462-626 (reader), restated on memory buffers.
 *  XXH32: public algorithm (xxhash by Y. Collet), one-shot form.
 * ===================================================================================================== */
static uint32_t rd32le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t orc_xxh32(const void* data, size_t len, uint32_t seed)
{
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* const end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* const limit = end - 16;
        do {
            v1 = rotl32(v1 + rd32le(p) * P2, 13) * P1; p += 4;
            v2 = rotl32(v2 + rd32le(p) * P2, 13) * P1; p += 4;
            v3 = rotl32(v3 + rd32le(p) * P2, 13) * P1; p += 4;
            v4 = rotl32(v4 + rd32le(p) * P2, 13) * P1; p += 4;
        } while (p <= limit);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else h = seed + P5;
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + rd32le(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl32(h + (*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

#define FRAME_MAGIC_FSE 0x183E2309u      /* fileio.c:121 */
#define FRAME_MAGIC_HUF 0x183E3309u      /* fileio.c:122 */
#define FRAME_MAX_BSID 6                 /* 64 KB blocks */
enum { FBT_COMPRESSED = 0, FBT_RAW = 1, FBT_RLE = 2, FBT_CRC = 3 };   /* fileio.c:137 */

static size_t orc_fse_compress_bound(size_t n) { return 512 + n + (n >> 7) + 4 + sizeof(size_t); }   /* FSE_COMPRESSBOUND, fse.h:290-292 */
static size_t frame_block_size(unsigned id) { return (size_t)1024 << id; }   /* fileio.c:219 */

size_t orc_frame_compress_bound(size_t srcSize, unsigned blockSizeId)
{
    const size_t bs = frame_block_size(blockSizeId);
    const size_t nb = (srcSize + bs - 1) / bs;
    return 5 + srcSize + 5 * nb + 3;      /* header, worst case raw blocks with 3..5-byte headers, end mark */
}

size_t orc_frame_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned blockSizeId, int codec)
{
    uint8_t* const out = (uint8_t*)dst;
    const uint8_t* const in = (const uint8_t*)src;
    if (blockSizeId > FRAME_MAX_BSID) return ERR(GENERIC);
    if (dstCapacity < orc_frame_compress_bound(srcSize, blockSizeId)) return ERR(dstSize_tooSmall);
    const size_t bs = frame_block_size(blockSizeId);
    const uint32_t magic = codec == 1 ? FRAME_MAGIC_HUF : FRAME_MAGIC_FSE;
    size_t o = 0, pos = 0;
    uint8_t* tmp = (uint8_t*)malloc(orc_fse_compress_bound(bs) + 8);
    if (!tmp) return ERR(GENERIC);
    out[0] = (uint8_t)magic; out[1] = (uint8_t)(magic >> 8); out[2] = (uint8_t)(magic >> 16); out[3] = (uint8_t)(magic >> 24);
    out[4] = (uint8_t)blockSizeId; o = 5;                                        /* fileio.c:324-325 */
    while (pos < srcSize) {
        const size_t inSize = srcSize - pos < bs ? srcSize - pos : bs;
        const size_t cap = orc_fse_compress_bound(bs);                            /* FSE_compressBound(inputBlockSize), :340 */
        const size_t cSize = codec == 1 ? orc_huf_compress2(tmp, cap, in + pos, inSize, 255, 11)     /* HUF_compress */
                                        : orc_fse_compress2(tmp, cap, in + pos, inSize, 255, 11);    /* FSE_compress */
        const int full = inSize == bs;
        if (orc_is_error(cSize)) { free(tmp); return cSize; }
        if (cSize == 0 || cSize == 1) {                                           /* raw / rle, :347-379 */
            const unsigned bt = cSize == 0 ? FBT_RAW : FBT_RLE;
            if (full) out[o++] = (uint8_t)((bt << 6) + 0x20);
            else { out[o++] = (uint8_t)(bt << 6); out[o++] = (uint8_t)(inSize >> 8); out[o++] = (uint8_t)inSize; }
            if (cSize == 0) { memcpy(out + o, in + pos, inSize); o += inSize; }
            else out[o++] = in[pos];
        } else {                                                                  /* compressed, :381-401 */
            if (full) out[o++] = (uint8_t)((FBT_COMPRESSED << 6) + 0x20);
            else { out[o++] = (uint8_t)(FBT_COMPRESSED << 6); out[o++] = (uint8_t)(inSize >> 8); out[o++] = (uint8_t)inSize; }
            out[o++] = (uint8_t)(cSize >> 8); out[o++] = (uint8_t)cSize;
            memcpy(out + o, tmp, cSize); o += cSize;
        }
        pos += inSize;
    }
    {   const uint32_t checksum = (orc_xxh32(in, srcSize, 0) >> 5) & ((1u << 22) - 1);   /* :408-416 */
        out[o++] = (uint8_t)((checksum >> 16) + (FBT_CRC << 6));
        out[o++] = (uint8_t)(checksum >> 8);
        out[o++] = (uint8_t)checksum;
    }
    free(tmp);
    return o;
}

size_t orc_frame_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    uint8_t* const out = (uint8_t*)dst;
    const uint8_t* const in = (const uint8_t*)src;
    size_t ip = 0, o = 0;
    int codec;
    if (srcSize < 5 + 3) return ERR(srcSize_wrong);
    {   const uint32_t magic = rd32le(in);
        if (magic == FRAME_MAGIC_FSE) codec = 0; else if (magic == FRAME_MAGIC_HUF) codec = 1; else return ERR(GENERIC);   /* :484-499 */
    }
    if (in[4] > FRAME_MAX_BSID) return ERR(GENERIC);                               /* :502-504 */
    const size_t bs = frame_block_size(in[4]);
    ip = 5;
    for (;;) {
        if (ip >= srcSize) return ERR(srcSize_wrong);
        const unsigned b0 = in[ip++];
        const unsigned bt = b0 >> 6;
        size_t rSize = bs, cSize;
        if (bt == FBT_CRC) {                                                       /* :600-606 */
            if (ip + 2 > srcSize) return ERR(srcSize_wrong);
            const uint32_t saved = in[ip + 1] + ((uint32_t)in[ip] << 8) + ((uint32_t)(b0 & 0x3F) << 16);
            const uint32_t calc = (orc_xxh32(out, o, 0) >> 5) & ((1u << 22) - 1);
            if (saved != calc) return ERR(corruption_detected);
            return o;
        }
        if (!(b0 & 0x20)) { if (ip + 2 > srcSize) return ERR(srcSize_wrong); rSize = ((size_t)in[ip] << 8) + in[ip + 1]; ip += 2; }   /* :527-532 */
        if (bt == FBT_COMPRESSED) { if (ip + 2 > srcSize) return ERR(srcSize_wrong); cSize = ((size_t)in[ip] << 8) + in[ip + 1]; ip += 2; }
        else if (bt == FBT_RAW) cSize = rSize; else cSize = 1;
        if (ip + cSize > srcSize) return ERR(srcSize_wrong);
        /* the reference tool decodes into malloc(blockSize) buffers (:509-510): an announced size above the frame's block size is
           outside its contract (it would overrun them); both this restatement and the device path reject such a frame here */
        if (rSize > bs) return ERR(corruption_detected);
        if (bt == FBT_COMPRESSED) {
            if (o + rSize > dstCapacity) return ERR(dstSize_tooSmall);
            const size_t r = codec == 1 ? orc_huf_decompress(out + o, rSize, in + ip, cSize) : orc_fse_decompress(out + o, rSize, in + ip, cSize);   /* :570-573 */
            if (orc_is_error(r)) return r;
            o += r;
        } else {
            if (o + rSize > dstCapacity) return ERR(dstSize_tooSmall);
            if (bt == FBT_RAW) memcpy(out + o, in + ip, rSize); else memset(out + o, in[ip], rSize);
            o += rSize;
        }
        ip += cSize;
    }
}
