// huf_tables.hip -- glue around the Huff0 hot loops, on the device (SURVEY 8(a') rows g5-g7):
//   compress side   : early-outs of HUF_compress_internal, HUF_optimalTableLog, HUF_buildCTable (sort, tree,
//                     HUF_setMaxHeight, canonical codes), HUF_writeCTable (+ HUF_compressWeights through FSE)
//                     (reference: lib/huf_compress.c:637-724, :48-51, :202-410, :63-147)
//   decompress side : raw / RLE decisions of HUF_decompress, HUF_readStats, HUF_readDTableX1
//                     (reference: lib/huf_decompress.c:1056-1066, :118-185; lib/entropy_common.c:154-215)
// v1 mapping: one lane per block; tables are produced in the reference's in-memory layout in global
// scratch, the hot-loop kernels stage them into LDS.
#include "internal.h"
#include "fse_glue.h"
#include "bitreader.h"

#define HUF_MAX_TL FSEHIP_HUF_TABLELOG_MAX
#define HUF_DEF_TL FSEHIP_HUF_TABLELOG_DEFAULT
#define HUF_MAX_SV 255

#define HUF_BIG_ALPHABET 96u     // alphabets from this size on are sorted by the LDS network of k_huf_presort
struct hnode_t { u32 count; u16 parent; u8 byte; u8 nbBits; };        // lib/huf_compress.c:201-206

// Node array of one lane inside the workgroup's scratch: element i of lane l sits at (i * 64 + l), so that the 64 lanes
// of a wave, which walk their trees in lock step, touch one contiguous 512-byte row per access.
struct NodeArr {
    hnode_t* p;
    DEV hnode_t& operator[](int i) const { return p[(ptrdiff_t)i * 64]; }
    DEV hnode_t& operator[](u32 i) const { return p[(size_t)i * 64]; }
    DEV NodeArr operator+(int k) const { NodeArr r; r.p = p + (ptrdiff_t)k * 64; return r; }
};

// ---------------------------------------------------------------------------------------------------
//  HUF_setMaxHeight (lib/huf_compress.c:215-291)
// ---------------------------------------------------------------------------------------------------
__device__ u32 huf_limit_height(NodeArr node, u32 lastNonNull, u32 maxNbBits)
{
    const u32 largest = node[lastNonNull].nbBits;
    if (largest <= maxNbBits) return largest;
    int debt = 0;
    const u32 unit = 1u << (largest - maxNbBits);
    int n = (int)lastNonNull;
    while (node[n].nbBits > maxNbBits) {
        debt += (int)(unit - (1u << (largest - node[n].nbBits)));
        node[n].nbBits = (u8)maxNbBits;
        n--;
    }
    while (node[n].nbBits == maxNbBits) n--;
    debt >>= (largest - maxNbBits);
    const u32 NONE = 0xF0F0F0F0u;
    u32 rankLast[HUF_MAX_TL + 2];
    for (u32 i = 0; i < HUF_MAX_TL + 2; i++) rankLast[i] = NONE;
    {   u32 cur = maxNbBits;
        for (int pos = n; pos >= 0; pos--) {
            if (node[pos].nbBits >= cur) continue;
            cur = node[pos].nbBits;
            rankLast[maxNbBits - cur] = (u32)pos;
        }
    }
    while (debt > 0) {
        u32 dec = hibit32((u32)debt) + 1;
        for (; dec > 1; dec--) {
            const u32 hi = rankLast[dec], lo = rankLast[dec - 1];
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
    return maxNbBits;
}

// HUF_sort (lib/huf_compress.c:307-329)
__device__ void huf_sort_nodes(NodeArr node, const unsigned* count, u32 maxSV)
{
    u32 base[32], cur[32];
    for (u32 n = 0; n < 32; n++) base[n] = 0;
    for (u32 n = 0; n <= maxSV; n++) base[hibit32(count[n] + 1)]++;
    for (u32 n = 30; n > 0; n--) base[n - 1] += base[n];
    for (u32 n = 0; n < 32; n++) cur[n] = base[n];
    for (u32 n = 0; n <= maxSV; n++) {
        const u32 c = count[n];
        const u32 r = hibit32(c + 1) + 1;
        u32 pos = cur[r]++;
        while (pos > base[r] && c > node[pos - 1].count) { node[pos] = node[pos - 1]; pos--; }
        node[pos].count = c;
        node[pos].byte = (u8)n;
    }
}

// The same order -- count descending, ties in symbol order (the insertion above only moves an element in front of strictly
// smaller counts, and the buckets run from the largest counts down) -- by a stable bottom-up merge sort: n log n steps
// instead of a quadratic number on flat histograms.  Ping-pongs between node0[0..255] and node0[256..511]; leaves the
// result in node0[1 .. n] with everything else zero.
__device__ void huf_sort_nodes_merge(NodeArr node0, const unsigned* count, u32 maxSV)
{
    const u32 n = maxSV + 1;
    NodeArr src = node0, dst = node0 + 256;
    for (u32 i = 0; i < n; i++) { hnode_t h; h.count = count[i]; h.parent = 0; h.byte = (u8)i; h.nbBits = 0; src[i] = h; }
    for (u32 width = 1; width < n; width <<= 1) {
        for (u32 lo = 0; lo < n; lo += 2 * width) {
            const u32 mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
            u32 i = lo, j = mid;
            hnode_t a = src[i < n ? i : 0], bb = src[j < n ? j : 0];
            for (u32 k = lo; k < hi; k++) {
                const bool left = i < mid && (j >= hi || a.count >= bb.count);   // ties: the earlier (smaller) symbol first
                dst[k] = left ? a : bb;
                if (left) { ++i; if (i < mid) a = src[i]; } else { ++j; if (j < hi) bb = src[j]; }
            }
        }
        const NodeArr t = src; src = dst; dst = t;
    }
    hnode_t z; z.count = 0; z.parent = 0; z.byte = 0; z.nbBits = 0;
    if (src.p == node0.p) { for (u32 i = n; i > 0; i--) node0[i] = node0[i - 1]; }        // shift up by one
    else { for (u32 i = 0; i < n; i++) node0[1 + i] = src[i]; }                           // ascending: never overtakes its source
    node0[0] = z;
    for (u32 i = n + 1; i < 512; i++) node0[i] = z;
}

// The same order once more, for large alphabets by a kernel of its own (k_huf_presort): every lane sorts its
// own column keys[i * 64 + lane] of 256 keys with a bitonic network.  key = count << 9 | 1 << 8 | (255 - symbol) is unique,
// so "descending key" = count descending, ties in symbol order; the unused tail of the column holds zeros.  The network
// has no data-dependent control flow and its compare-exchanges within a stage are independent, so the LDS round trips
// overlap (the merge sort above is a chain of dependent global-memory accesses: 1.5 ms of the 4 ms this kernel took on
// 256-symbol alphabets).
__device__ void huf_sort_nodes_lds(NodeArr node0, const unsigned* count, u32 maxSV, u32* keys)
{
    const u32 n = maxSV + 1;
    for (u32 i = 0; i < 256; i++) keys[i * 64] = i < n ? (count[i] << 9) | (1u << 8) | (255u - i) : 0u;
    for (u32 k = 2; k <= 256; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
#pragma unroll 8
            for (u32 t = 0; t < 128; t++) {
                const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1));     // the t-th index with bit j clear
                const u32 l = i | j;
                const u32 a = keys[i * 64], b = keys[l * 64];
                const bool up = (i & k) != 0;                            // descending overall: ascending runs where bit k is set
                const bool swap = up ? a > b : a < b;
                keys[i * 64] = swap ? b : a;
                keys[l * 64] = swap ? a : b;
            }
        }
    }
    hnode_t z; z.count = 0; z.parent = 0; z.byte = 0; z.nbBits = 0;
    node0[0] = z;
    for (u32 i = 0; i < n; i++) { const u32 key = keys[i * 64]; hnode_t h; h.count = key >> 9; h.parent = 0; h.byte = (u8)(255u - (key & 0xFFu)); h.nbBits = 0; node0[1 + i] = h; }
    for (u32 i = n + 1; i < 512; i++) node0[i] = z;
}

// HUF_buildCTable_wksp (lib/huf_compress.c:338-410).  celt[s] = val | nbBits << 16 (struct HUF_CElt_s, :106-109).
// node0: scratch of 2*256 entries (global memory, interleaved across the lanes of the wave).
__device__ size_t huf_build_ctable(u32* celt, const unsigned* count, u32 maxSV, u32 maxNbBits, NodeArr node0, bool presorted)
{
    const int START = HUF_MAX_SV + 1;
    const NodeArr node = node0 + 1;
    int last, lowS, lowN, nodeNb = START, root, n;
    if (maxNbBits == 0) maxNbBits = HUF_DEF_TL;
    if (maxSV > HUF_MAX_SV) return FERR(maxSymbolValue_tooLarge);
    if (maxSV < HUF_BIG_ALPHABET) {                        // small alphabets: the reference's bucket + insertion sort is cheapest
        hnode_t z; z.count = 0; z.parent = 0; z.byte = 0; z.nbBits = 0; for (u32 i = 0; i < 512; i++) node0[i] = z;
        huf_sort_nodes(node, count, maxSV);
    } else if (!presorted) huf_sort_nodes_merge(node0, count, maxSV);      // (k_huf_presort has left the sorted nodes in node0)
    last = (int)maxSV;
    while (node[last].count == 0) last--;
    lowS = last; root = nodeNb + lowS - 1; lowN = nodeNb;
    node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
    node[lowS].parent = node[lowS - 1].parent = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= root; n++) node[n].count = 1u << 30;
    node0[0].count = 1u << 31;
    while (nodeNb <= root) {
        const int a = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        const int b = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        node[nodeNb].count = node[a].count + node[b].count;
        node[a].parent = node[b].parent = (u16)nodeNb;
        nodeNb++;
    }
    node[root].nbBits = 0;
    for (n = root - 1; n >= START; n--) node[n].nbBits = (u8)(node[node[n].parent].nbBits + 1);
    for (n = 0; n <= last; n++) node[n].nbBits = (u8)(node[node[n].parent].nbBits + 1);
    maxNbBits = huf_limit_height(node, (u32)last, maxNbBits);
    {   u16 perRank[HUF_MAX_TL + 1], valRank[HUF_MAX_TL + 1];
        for (n = 0; n <= HUF_MAX_TL; n++) { perRank[n] = 0; valRank[n] = 0; }
        const int alphabet = (int)maxSV + 1;
        if (maxNbBits > HUF_MAX_TL) return FERR(GENERIC);
        for (n = 0; n <= last; n++) perRank[node[n].nbBits]++;
        {   u16 min = 0;
            for (n = (int)maxNbBits; n > 0; n--) { valRank[n] = min; min = (u16)(min + perRank[n]); min >>= 1; }
        }
        for (n = 0; n < alphabet; n++) celt[node[n].byte] = (u32)node[n].nbBits << 16;        // nbBits per symbol
        for (n = 0; n < alphabet; n++) { const u32 nb = celt[n] >> 16; celt[n] = (u32)(valRank[nb]++) | (nb << 16); }
    }
    return maxNbBits;
}

// ---------------------------------------------------------------------------------------------------
//  serial FSE_compress_usingCTable for the (<= 255) Huffman weights (lib/fse_compress.c:554-611)
// ---------------------------------------------------------------------------------------------------
struct ByteBits {      // byte-granular LIFO writer; closed-form close rule of BIT_closeCStream (bitstream.h:254-260)
    u8* out; size_t cap; size_t nbytes; u64 acc; u32 fill;
    DEV void init(u8* o, size_t c) { out = o; cap = c; nbytes = 0; acc = 0; fill = 0; }
    DEV void put(u32 v, u32 nb)
    {
        acc |= (u64)(v & ((1u << nb) - 1u)) << fill; fill += nb;
        while (fill >= 8) { if (nbytes < cap) out[nbytes] = (u8)acc; nbytes++; acc >>= 8; fill -= 8; }
    }
    DEV size_t close()
    {
        put(1, 1);
        if (cap <= 8 || nbytes >= cap - 8) return 0;
        if (fill) { out[nbytes] = (u8)acc; return nbytes + 1; }
        return nbytes;
    }
};

__device__ size_t fse_encode_serial(u8* dst, size_t cap, const u8* src, size_t n, const u32* ct)
{
    const u16* head = (const u16*)ct;
    const u32 tl = head[0];
    const u16* stateTable = head + 2;
    const u32* tt = ct + 1 + (tl ? (1u << (tl - 1)) : 1u);
    u32 chain[2];
    if (n <= 2) return 0;
    if (cap <= 8) return 0;
    ByteBits w; w.init(dst, cap);
    for (u32 j = 0; j < 2; j++) {
        const u32 dfs = tt[2 * src[n - 1 - j]], dnb = tt[2 * src[n - 1 - j] + 1];
        const u32 nb = (dnb + (1u << 15)) >> 16;
        chain[j] = stateTable[(((nb << 16) - dnb) >> nb) + dfs];
    }
    for (size_t j = 2; j < n; j++) {
        const u32 sym = src[n - 1 - j];
        const u32 dfs = tt[2 * sym], dnb = tt[2 * sym + 1];
        const u32 x = chain[j & 1];
        const u32 nb = (x + dnb) >> 16;
        w.put(x, nb);
        chain[j & 1] = stateTable[(x >> nb) + dfs];
    }
    w.put(chain[(n & 1) ? 1 : 0], tl);
    w.put(chain[(n & 1) ? 0 : 1], tl);
    return w.close();
}

// HUF_compressWeights (lib/huf_compress.c:63-103)
__device__ size_t huf_compress_weights(u8* out, size_t cap, const u8* weights, size_t n)
{
    u32 maxSV = HUF_MAX_TL, tl = 6;
    unsigned count[HUF_MAX_TL + 1];
    s16 norm[HUF_MAX_TL + 1];
    u32 ct[1 + (1 << 5) + (HUF_MAX_TL + 1) * 2];
    u8 cellSym[64];
    u32 best = 0;
    if (n <= 1) return 0;
    for (u32 s = 0; s <= HUF_MAX_TL; s++) count[s] = 0;
    for (size_t i = 0; i < n; i++) count[weights[i]]++;              // HIST_count_simple (hist.c:29-54)
    while (!count[maxSV]) maxSV--;
    for (u32 s = 0; s <= maxSV; s++) if (count[s] > best) best = count[s];
    if (best == n) return 1;
    if (best == 1) return 0;
    tl = fse_optimal_tablelog(tl, n, maxSV, 2);
    {   const size_t e = fse_normalize_count(norm, tl, count, n, maxSV); if (is_err(e)) return e; }
    const size_t h = fse_write_ncount(out, cap, norm, maxSV, tl);
    if (is_err(h)) return h;
    fse_build_ctable(ct, cellSym, norm, maxSV, tl);
    const size_t c = fse_encode_serial(out + h, cap - h, weights, n, ct);
    if (c == 0) return 0;
    return h + c;
}

// HUF_writeCTable (lib/huf_compress.c:114-147)
__device__ size_t huf_write_ctable(u8* out, size_t cap, const u32* celt, u32 maxSV, u32 huffLog)
{
    u8 w[HUF_MAX_SV + 1];
    if (maxSV > HUF_MAX_SV) return FERR(maxSymbolValue_tooLarge);
    for (u32 n = 0; n < maxSV; n++) { const u32 nb = (celt[n] >> 16) & 0xFF; w[n] = nb ? (u8)(huffLog + 1 - nb) : 0; }
    {   const size_t hs = huf_compress_weights(out + 1, cap - 1, w, maxSV);
        if (is_err(hs)) return hs;
        if ((hs > 1) & (hs < maxSV / 2)) { out[0] = (u8)hs; return hs + 1; }
    }
    if (maxSV > 128) return FERR(GENERIC);
    if (((maxSV + 1) / 2) + 1 > cap) return FERR(dstSize_tooSmall);
    out[0] = (u8)(128 + (maxSV - 1));
    w[maxSV] = 0;
    for (u32 n = 0; n < maxSV; n += 2) out[(n / 2) + 1] = (u8)((w[n] << 4) + w[n + 1]);
    return ((maxSV + 1) / 2) + 1;
}

// ---------------------------------------------------------------------------------------------------
//  HUF_readStats (lib/entropy_common.c:154-215) and HUF_readDTableX1 (lib/huf_decompress.c:118-185)
// ---------------------------------------------------------------------------------------------------
__device__ size_t huf_read_stats(u8* w, size_t hwSize, u32* rankStats, u32* nbSymbolsPtr, u32* tlPtr, const u8* ip, size_t srcSize)
{
    size_t iSize, oSize;
    u32 total = 0;
    if (!srcSize) return FERR(srcSize_wrong);
    iSize = ip[0];
    if (iSize >= 128) {
        oSize = iSize - 127;
        iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return FERR(srcSize_wrong);
        if (oSize >= hwSize) return FERR(corruption_detected);
        for (size_t n = 0; n < oSize; n += 2) { w[n] = ip[1 + n / 2] >> 4; w[n + 1] = ip[1 + n / 2] & 15; }
    } else {                                           // FSE_decompress_wksp(huffWeight, hwSize-1, ip+1, iSize, wksp, 6)
        s16 norm[256];
        u32 dt[1 + (1 << 6)];
        u32 tl = 0, maxSV = 255;
        if (iSize + 1 > srcSize) return FERR(srcSize_wrong);
        const size_t h = fse_read_ncount(norm, &maxSV, &tl, ip + 1, iSize);
        if (is_err(h)) return h;
        if (tl > 6) return FERR(tableLog_tooLarge);
        {   const size_t e = fse_build_dtable(dt, norm, maxSV, tl); if (is_err(e)) return e; }
        // FSE_decompress_usingDTable (lib/fse_decompress.c:178-238), literal
        const bool fast = (dt[0] >> 16) != 0;
        const u32* cells = dt + 1;
        const long omax = (long)(hwSize - 1);
        long op = 0;
        BitReader r;
        {   const size_t e = r.init(ip + 1 + h, iSize - h); if (is_err(e)) return e; }
        u32 s1 = r.read(tl); r.reload();
        u32 s2 = r.read(tl); r.reload();
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
        oSize = (size_t)op;
    }
    for (u32 n = 0; n <= HUF_MAX_TL; n++) rankStats[n] = 0;
    for (size_t n = 0; n < oSize; n++) {
        if (w[n] >= HUF_MAX_TL) return FERR(corruption_detected);
        rankStats[w[n]]++;
        total += (1u << w[n]) >> 1;
    }
    if (total == 0) return FERR(corruption_detected);
    {   const u32 tl = hibit32(total) + 1;
        if (tl > HUF_MAX_TL) return FERR(corruption_detected);
        *tlPtr = tl;
        const u32 rest = (1u << tl) - total;
        const u32 lastW = hibit32(rest) + 1;
        if ((1u << hibit32(rest)) != rest) return FERR(corruption_detected);
        w[oSize] = (u8)lastW;
        rankStats[lastW]++;
    }
    if ((rankStats[1] < 2) || (rankStats[1] & 1)) return FERR(corruption_detected);
    *nbSymbolsPtr = (u32)(oSize + 1);
    return iSize + 1;
}

// dtable[0] = DTableDesc {maxTableLog, tableType, tableLog, reserved} (huf_decompress.c:101), cells {byte, nbBits} (:116)
__device__ size_t huf_read_dtable_x1(u32* dtable, u32 maxTableLogField, const u8* src, size_t srcSize)
{
    u8 w[HUF_MAX_SV + 1];
    u32 rankVal[16];
    u32 tl = 0, nbSym = 0;
    const size_t iSize = huf_read_stats(w, HUF_MAX_SV + 1, rankVal, &nbSym, &tl, src, srcSize);
    if (is_err(iSize)) return iSize;
    if (tl > maxTableLogField + 1) return FERR(tableLog_tooLarge);
    dtable[0] = (maxTableLogField & 0xFF) | (tl << 16);
    {   u32 next = 0;
        for (u32 n = 1; n < tl + 1; n++) { const u32 cur = next; next += rankVal[n] << (n - 1); rankVal[n] = cur; }
    }
    u16* const cells = (u16*)(dtable + 1);
    for (u32 n = 0; n < nbSym; n++) {
        const u32 wt = w[n];
        const u32 len = (1u << wt) >> 1;
        const u16 cell = (u16)(n | ((tl + 1 - wt) << 8));
        for (u32 u = rankVal[wt]; u < rankVal[wt] + len; u++) cells[u] = cell;
        rankVal[wt] += len;
    }
    return iSize;
}

// ---------------------------------------------------------------------------------------------------
//  prepare kernels (one lane per block)
// ---------------------------------------------------------------------------------------------------
// Large alphabets are sorted by a kernel of their own: it needs 64 KiB of LDS per wave for the lanes' sort columns, which
// would cost the rest of the (latency-bound, lane-per-block) prepare work two thirds of its occupancy.  It leaves the
// sorted nodes in the node scratch of the same (workgroup, lane) that k_huf_cprep then uses.
__global__ __launch_bounds__(64) void k_huf_presort(HufCPrepArgs a, hnode_t* nodeScratch)
{
    extern __shared__ __attribute__((aligned(16))) u32 sortLds[];
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= a.nBlocks) return;
    if (is_err(a.histResults[b])) return;
    const u32 maxSV = a.maxSVs[b];
    if (maxSV < HUF_BIG_ALPHABET || maxSV > HUF_MAX_SV) return;
    huf_sort_nodes_lds(NodeArr{nodeScratch + (size_t)blockIdx.x * 64 * 512 + threadIdx.x}, a.counts + b * 256, maxSV, sortLds + threadIdx.x);
}

__global__ __launch_bounds__(64) void k_huf_cprep(HufCPrepArgs a, hnode_t* nodeScratch)
{
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= a.nBlocks) return;
    HufMeta m; m.state = 0; m.hdrSize = 0; m.tableLog = 0; m.maxSV = 0;
    const size_t n = view_size(a.src, b);
    u8* const dst = a.dst + b * a.dstStride;
    size_t result = 0;
    do {                                                                    // huf_compress.c:654-674
        if (!n) { result = 0; break; }
        if (!a.dstCapacity) { result = 0; break; }
        if (n > FSEHIP_HUF_BLOCKSIZE_MAX) { result = FERR(srcSize_wrong); break; }
        const size_t top = a.histResults[b];
        if (is_err(top)) { result = top; break; }
        if (top == n) { dst[0] = view_ptr(a.src, b)[0]; result = 1; break; }
        if (top <= (n >> 7) + 4) { result = 0; break; }
        const u32 maxSV = a.maxSVs[b];
        u32 huffLog = fse_optimal_tablelog(a.huffLogReq ? a.huffLogReq : HUF_DEF_TL, n, maxSV, 1);   // :691, :48-51
        u32* const celt = a.ctables + b * a.ctStrideU32;
        {   const size_t mb = huf_build_ctable(celt, a.counts + b * 256, maxSV, huffLog, NodeArr{nodeScratch + (size_t)blockIdx.x * 64 * 512 + threadIdx.x}, true);
            if (is_err(mb)) { result = mb; break; }
            huffLog = (u32)mb;
        }
        for (u32 s = maxSV + 1; s < 256; s++) celt[s] = 0;                  // :697-699
        const size_t h = huf_write_ctable(dst, a.dstCapacity, celt, maxSV, huffLog);   // :703
        if (is_err(h)) { result = h; break; }
        if (h + 12ul >= n) { result = 0; break; }                           // :715
        m.state = 1; m.hdrSize = (u32)h; m.tableLog = huffLog; m.maxSV = maxSV;
    } while (0);
    a.meta[b] = m;
    if (m.state == 0) a.results[b] = result;
}

__global__ __launch_bounds__(64) void k_huf_dprep(HufDPrepArgs a)
{
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    const u32 lane = threadIdx.x;
    int cls = -1;                                                           // decoder class of my block (-1: none / finished here)
    if (b < a.nBlocks) {
    HufMeta m; m.state = 0; m.hdrSize = 0; m.tableLog = 0; m.maxSV = 0;
    const u8* const in = view_ptr(a.csrc, b);
    const size_t cSize = view_size(a.csrc, b);
    const size_t dstSize = view_size(a.dstSizes, b);
    u8* const dst = a.dst + b * a.dstStride;
    size_t result = 0;
    do {                                                                    // huf_decompress.c:1063-1066
        if (dstSize == 0) { result = FERR(dstSize_tooSmall); break; }
        if (cSize > dstSize) { result = FERR(corruption_detected); break; }
        if (cSize == dstSize) { for (size_t i = 0; i < dstSize; i++) dst[i] = in[i]; result = dstSize; break; }   // not compressed
        if (cSize == 1) { const u8 v = in[0]; for (size_t i = 0; i < dstSize; i++) dst[i] = v; result = dstSize; break; }   // RLE
        u32* const dt = a.dtables + b * a.dtStrideU32;
        const size_t h = huf_read_dtable_x1(dt, HUF_MAX_TL - 1, in, cSize);  // HUF_CREATE_STATIC_DTABLEX1(.., HUF_TABLELOG_MAX), :421-426,445-449
        if (is_err(h)) { result = h; break; }
        if (h >= cSize) { result = FERR(srcSize_wrong); break; }
        m.state = 1; m.hdrSize = (u32)h; m.tableLog = (dt[0] >> 16) & 0xFF;
        cls = m.tableLog > 11u ? 1 : 0;
    } while (0);
    a.meta[b] = m;
    if (m.state == 0) a.results[b] = result;
    }
    // append my block to its class list: one atomic per class and wave
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int c = 0; c < HUF_DCLS_COUNT; ++c) {
        const unsigned long long mask = __ballot(cls == c);
        if (!mask) continue;                                                // uniform
        const int leader = __builtin_ctzll(mask);
        u32 base = 0;
        if ((int)lane == leader) base = atomicAdd(&a.counts[c], (u32)__builtin_popcountll(mask));
        base = (u32)__shfl((int)base, leader, WAVE);
        if (cls == c) a.lists[(size_t)c * a.nBlocks + base + (u32)__builtin_popcountll(mask & below)] = (u32)b;
    }
}

hipError_t launch_huf_cprep(const HufCPrepArgs& a, hipStream_t s, void* nodeScratch)
{
    if (a.nBlocks == 0) return hipSuccess;
    probe_before(PK_HUF_CPREP, s);
    {   const hipError_t e = ensure_dyn_lds((const void*)k_huf_presort, 64 * 1024); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(k_huf_presort, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 64 * 1024, s, a, (hnode_t*)nodeScratch);
    hipLaunchKernelGGL(k_huf_cprep, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a, (hnode_t*)nodeScratch);
    probe_after(PK_HUF_CPREP, s);
    return hipGetLastError();
}
hipError_t launch_huf_dprep(const HufDPrepArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    {   const hipError_t e = hipMemsetAsync(a.counts, 0, HUF_DCLS_COUNT * sizeof(u32), s); if (e != hipSuccess) return e; }
    probe_before(PK_HUF_DPREP, s);
    hipLaunchKernelGGL(k_huf_dprep, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a);
    probe_after(PK_HUF_DPREP, s);
    return hipGetLastError();
}
