// wave_glue.h -- the FSE "glue" between histogram and hot loop (SURVEY 8(a') rows g1-g2), one 64-lane wave per table:
//   table-log selection, count normalisation (with its fallback) and the NCount header writer
//   (behaviour of lib/fse_compress.c:316-342, :348-494, :186-298 -- restated, not transliterated).
//
// The reference walks the alphabet with loop-carried state (points still to distribute, best symbol so far, bit cursor of
// the header).  Here lane l holds symbols 4l .. 4l+3 in registers and nothing is carried from symbol to symbol:
//   * normalisation: every symbol is scaled independently; "points left" is a wave sum, "the first most probable symbol" a
//     wave arg-max over (probability, -symbol) keys; the fallback classifies every symbol independently against two thresholds,
//     gets its totals from wave sums and hands out the remaining points by prefix sums (the reference's running 62-bit
//     accumulator is an exclusive scan of count * step; its round-robin top-up has the closed form quotient + (rank < rest));
//   * header: the cost of a symbol depends only on the points still unassigned in front of it (an exclusive scan of |counter|):
//     threshold = 2^floor(log2(remaining)), so each lane forms its symbols' (value, nbBits) chunks directly; a run of zero
//     counters is coded by the lane holding its first zero (its length comes from a suffix-min scan of the non-zero
//     positions); bit offsets are one more scan and the chunks are OR-ed into an LDS image of the header.
// All functions are templated on the group width W (a power of two, 4 .. 64): the wave is split into 64 / W groups of W lanes,
// each working on its own table (lane `sub` of a group holds symbols 4*sub .. 4*sub+3, so a group covers alphabets of up to
// 4*W symbols); scalar arguments and results are uniform per group.  W = 64 is one table per wave (the FSE block tables),
// W = 8 is eight tables at once (the 13-symbol weight alphabets of eight Huffman headers, huf_prep.hip).
#pragma once
#include "dev_common.h"

#define FSE_MIN_TL FSEHIP_FSE_MIN_TABLELOG
#define FSE_MAX_TL FSEHIP_FSE_MAX_TABLELOG
#define FSE_DEF_TL FSEHIP_FSE_DEFAULT_TABLELOG

// ---- group primitives -----------------------------------------------------------------------------------------------------
template <int W> DEV u32 wg_sub(u32 lane) { return lane & (u32)(W - 1); }
// (lane-invariant callers pass no lane: these primitives are called by all lanes of the wave in uniform control flow, and a group's lanes are
//  lane & ~(W - 1) .. | (W - 1); the DPP forms need the lane id only for groups smaller than the wave)
DEV u32 wg_lane() { return (u32)threadIdx.x & 63u; }
template <int W> DEV u32 wg_sum(u32 v)
{
#if FSEHIP_DPP_SCANS
    return group_reduce<W, ScanAdd>(v, wg_lane());
#else
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += (u32)__shfl_xor((int)v, off, WAVE);
    return v;
#endif
}
template <int W> DEV u64 wg_sum64(u64 v)
{
#if FSEHIP_DPP_SCANS
    const u32 lane = wg_lane();
    return group_last64<W>(group_scan_incl_add64<W>(v, lane & (u32)(W - 1)), lane);
#else
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += (u64)__shfl_xor((unsigned long long)v, off, WAVE);
    return v;
#endif
}
template <int W> DEV u32 wg_max(u32 v)
{
#if FSEHIP_DPP_SCANS
    return group_reduce<W, ScanMax>(v, wg_lane());
#else
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) { const u32 o = (u32)__shfl_xor((int)v, off, WAVE); v = o > v ? o : v; }
    return v;
#endif
}
template <int W> DEV u32 wg_min(u32 v)
{
#if FSEHIP_DPP_SCANS
    return group_reduce<W, ScanMin>(v, wg_lane());
#else
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) { const u32 o = (u32)__shfl_xor((int)v, off, WAVE); v = o < v ? o : v; }
    return v;
#endif
}
template <int W> DEV bool wg_any(bool p, u32 lane)          // any lane of my group
{
    const unsigned long long m = __ballot(p);
    if (W == 64) return m != 0;
    return ((m >> (lane & ~(u32)(W - 1))) & ((1ull << (W & 63)) - 1ull)) != 0;
}
template <int W> DEV u32 wg_scan_excl(u32 v, u32 lane)     // exclusive prefix sum over the lanes of my group
{
    const u32 sub = wg_sub<W>(lane);
#if FSEHIP_DPP_SCANS
    return group_scan_incl<W, ScanAdd>(v, sub) - v;
#else
    u32 incl = v;
#pragma unroll
    for (int off = 1; off < W; off <<= 1) { const u32 o = (u32)__shfl_up((int)incl, off, WAVE); if ((int)sub >= off) incl += o; }
    return incl - v;
#endif
}
template <int W> DEV u64 wg_scan_excl64(u64 v, u32 lane)
{
    const u32 sub = wg_sub<W>(lane);
#if FSEHIP_DPP_SCANS
    return group_scan_incl_add64<W>(v, sub) - v;
#else
    u64 incl = v;
#pragma unroll
    for (int off = 1; off < W; off <<= 1) { const u64 o = (u64)__shfl_up((unsigned long long)incl, off, WAVE); if ((int)sub >= off) incl += o; }
    return incl - v;
#endif
}
template <int W> DEV u32 wg_suffix_min_excl(u32 v, u32 lane)   // min over the lanes of my group above this one (0xFFFFFFFF for the last)
{
    const u32 sub = wg_sub<W>(lane);
#if FSEHIP_DPP_SCANS
    if (W == 64) {
        // inside the rows of 16: inclusive suffix minimum by row_shl:1/2/4/8 (a lane the shift does not reach keeps its value); across the rows: every
        // row's minimum now sits in its first lane -- three v_readlane, the rows above folded on the scalar unit; then one wave_shl:1 makes it exclusive
        u32 m = v;
        { const u32 t = dpp_mov<0x101, 0xF, false>(0xFFFFFFFFu, m); m = t < m ? t : m; }
        { const u32 t = dpp_mov<0x102, 0xF, false>(0xFFFFFFFFu, m); m = t < m ? t : m; }
        { const u32 t = dpp_mov<0x104, 0xF, false>(0xFFFFFFFFu, m); m = t < m ? t : m; }
        { const u32 t = dpp_mov<0x108, 0xF, false>(0xFFFFFFFFu, m); m = t < m ? t : m; }
        const u32 r1 = (u32)__builtin_amdgcn_readlane((int)m, 16), r2 = (u32)__builtin_amdgcn_readlane((int)m, 32), r3 = (u32)__builtin_amdgcn_readlane((int)m, 48);
        const u32 a2 = r3, a1 = r2 < r3 ? r2 : r3, a0 = r1 < a1 ? r1 : a1;               // minimum of the rows above row 2 / 1 / 0
        const u32 above = lane < 16 ? a0 : lane < 32 ? a1 : lane < 48 ? a2 : 0xFFFFFFFFu;
        m = above < m ? above : m;
        return dpp_mov<0x130, 0xF, false>(0xFFFFFFFFu, m);                                 // wave_shl:1: lane i takes lane i + 1's, the last lane keeps all ones
    }
#endif
    u32 m = v;
#pragma unroll
    for (int off = 1; off < W; off <<= 1) { const u32 o = (u32)__shfl_down((int)m, off, WAVE); if ((int)sub + off < W) m = o < m ? o : m; }
    const u32 up = (u32)__shfl_down((int)m, 1, WAVE);
    return sub == (u32)(W - 1) ? 0xFFFFFFFFu : up;
}

// ---- table log (lib/fse_compress.c:316-342): the largest log the source size supports, not above the request, not below what
//      the alphabet needs
DEV u32 wg_min_tablelog(size_t srcSize, u32 maxSV)
{
    const u32 bySize = hibit32((u32)srcSize) + 1, byAlphabet = hibit32(maxSV) + 2;
    return bySize < byAlphabet ? bySize : byAlphabet;
}
DEV u32 wg_optimal_tablelog(u32 request, size_t srcSize, u32 maxSV, u32 minus, u32 defTl = FSE_DEF_TL, u32 maxTl = FSE_MAX_TL)
{
    u32 tl = request ? request : defTl;
    const u32 bySrc = hibit32((u32)(srcSize - 1)) - minus;
    tl = bySrc < tl ? bySrc : tl;
    const u32 need = wg_min_tablelog(srcSize, maxSV);
    tl = need > tl ? need : tl;
    tl = tl < FSE_MIN_TL ? FSE_MIN_TL : tl;
    return tl > maxTl ? maxTl : tl;
}

// ---- normalisation ---------------------------------------------------------------------------------------------------
// c[i] = count of symbol SPL*lane + i (SPL symbols per lane, 4 unless the alphabet is wider than 4*W; zero beyond maxSV), total = their sum (>= 2, no symbol owns it all).  Leaves the
// normalised counters in n[i] (-1 = "less than one point") and returns 0, or an error code.
#define WG_PENDING (-2)
template <int W, int SPL = 4>
DEV size_t wg_normalize_fallback(int n[SPL], const u32 c[SPL], u64 total, u32 maxSV, u32 tl, u32 lane)
{
    const u32 sub = wg_sub<W>(lane);
    const u32 ts = 1u << tl;
    const u32 tiny = (u32)(total >> tl);
    u32 one = (u32)((total * 3) >> (tl + 1));
    // every symbol on its own: absent / below one point / about one point / still pending
    u32 given = 0; u64 taken = 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const bool in = SPL * sub + i <= maxSV;
        if (!in || c[i] == 0) n[i] = 0;
        else if (c[i] <= tiny) { n[i] = -1; ++given; taken += c[i]; }
        else if (c[i] <= one) { n[i] = 1; ++given; taken += c[i]; }
        else n[i] = WG_PENDING;
    }
    given = wg_sum<W>(given); total -= wg_sum64<W>(taken);
    u32 left = ts - given;
    if (left == 0) return FERR(GENERIC);                      // (cannot happen once the table log covers the alphabet)
    if (total / left > one) {                                  // the pending ones would round to zero: widen "about one point"
        one = (u32)((total * 3) / ((u64)left * 2));
        u32 g2 = 0; u64 t2 = 0;
#pragma unroll
        for (int i = 0; i < SPL; ++i) if (n[i] == WG_PENDING && c[i] <= one) { n[i] = 1; ++g2; t2 += c[i]; }
        given += wg_sum<W>(g2); total -= wg_sum64<W>(t2);
        left = ts - given;
    }
    if (given == maxSV + 1) {                                  // nothing pending: the first most frequent symbol takes the rest
        u32 best = 0;
#pragma unroll
        for (int i = 0; i < SPL; ++i) best = c[i] > best ? c[i] : best;
        best = wg_max<W>(best);
        u32 who = 0xFFFFFFFFu;
#pragma unroll
        for (int i = SPL - 1; i >= 0; --i) if (c[i] == best && SPL * sub + i <= maxSV) who = SPL * sub + i;
        who = wg_min<W>(who);
#pragma unroll
        for (int i = 0; i < SPL; ++i) if (SPL * sub + i == who) n[i] += (int)left;
        return 0;
    }
    if (total == 0) {                                          // the rest goes round robin over the one-point symbols:
        u32 mine = 0;                                          // quotient each, one more for the first (rest) of them
#pragma unroll
        for (int i = 0; i < SPL; ++i) mine += n[i] > 0;
        const u32 P = wg_sum<W>(mine);
        u32 rank = wg_scan_excl<W>(mine, lane);
        const u32 q = left / P, r = left % P;
#pragma unroll
        for (int i = 0; i < SPL; ++i) if (n[i] > 0) { n[i] += (int)(q + (rank < r)); ++rank; }
        return 0;
    }
    // the pending symbols share `left` points in proportion to their counts: cumulative positions on a 2^(62-tl) grid
    const u32 vlog = 62 - tl;
    const u64 mid = ((u64)1 << (vlog - 1)) - 1;
    const u64 rstep = ((((u64)1 << vlog) * left) + mid) / total;
    u64 span = 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) if (n[i] == WG_PENDING) span += (u64)c[i] * rstep;
    u64 run = mid + wg_scan_excl64<W>(span, lane);
    bool starved = false;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        if (n[i] != WG_PENDING) continue;
        const u64 end = run + (u64)c[i] * rstep;
        const u32 w = (u32)(end >> vlog) - (u32)(run >> vlog);
        starved |= w < 1;
        n[i] = (int)(s16)w;
        run = end;
    }
    return wg_any<W>(starved, lane) ? FERR(GENERIC) : 0;
}

template <int W, int SPL = 4, int MAXTL = FSE_MAX_TL>
DEV size_t wg_normalize(int n[SPL], const u32 c[SPL], u64 total, u32 maxSV, u32 tl, u32 lane)
{
    const u32 sub = wg_sub<W>(lane);
    if (tl < FSE_MIN_TL) return FERR(GENERIC);
    if (tl > (u32)MAXTL) return FERR(tableLog_tooLarge);
    if (tl < wg_min_tablelog((size_t)total, maxSV)) return FERR(GENERIC);
    const u32 scale = 62 - tl;
    const u64 step = ((u64)1 << 62) / total;
    const u64 vstep = (u64)1 << (scale - 20);
    const u32 tiny = (u32)(total >> tl);
    // rounding thresholds of the small probabilities (lib/fse_compress.c:445), in units of vstep
    const u32 beat[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    u32 used = 0, key = 0;                                     // key = probability << 12 | 4095 - symbol: the arg-max prefers the lower symbol
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const u32 s = SPL * sub + i;
        if (s > maxSV || c[i] == 0) { n[i] = 0; continue; }
        if (c[i] <= tiny) { n[i] = -1; ++used; continue; }
        const u64 prod = (u64)c[i] * step;
        u32 p = (u32)(prod >> scale) & 0xFFFFu;
        if (p < 8) p += (prod - ((u64)p << scale)) > vstep * beat[p];
        n[i] = (int)(s16)p;
        used += p;
        const u32 k = (p << 12) | (4095u - s);
        key = k > key ? k : key;
    }
    const int still = (int)(1u << tl) - (int)wg_sum<W>(used);
    key = wg_max<W>(key);
    const u32 largest = (key >> 12) ? 4095u - (key & 4095u) : 0u;
    int nl = 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) if (SPL * sub + i == largest) nl = n[i];
    nl = __shfl(nl, (int)((lane - sub) + largest / (u32)SPL), WAVE);
    if (-still >= (nl >> 1)) return wg_normalize_fallback<W, SPL>(n, c, total, maxSV, tl, lane);
#pragma unroll
    for (int i = 0; i < SPL; ++i) if (SPL * sub + i == largest) n[i] += still;
    return 0;
}

// ---- NCount header -----------------------------------------------------------------------------------------------------
// OR `nb` (<= 16) bits into the little-endian bit image at bit position pos (LDS atomics: lanes share words)
DEV void wg_or_bits(u32* img, u32 pos, u32 v, u32 nb)
{
    if (nb == 0) return;
    const u32 w = pos >> 5, sh = pos & 31u;
    atomicOr(&img[w], v << sh);
    if (sh + nb > 32u) atomicOr(&img[w + 1], v >> (32u - sh));
}
// n[i]: counters of symbols 4*lane + i, summing (in absolute value) to 1 << tl with a non-zero counter at maxSV.  img: zeroed
// LDS words (>= 132).  Returns the header size in bytes (the image then holds the header) or an error code: dstSize_tooSmall by the
// reference's rule -- only checked when the destination is below the worst-case header size, at every 16-bit flush
// (lib/fse_compress.c:186-190, :228-272), i.e. against the position of the last flush.
template <int W, int SPL = 4>
DEV size_t wg_write_ncount(u32* img, size_t cap, const int n[SPL], u32 maxSV, u32 tl, u32 lane)
{
    const u32 sub = wg_sub<W>(lane);
    const u32 ts = 1u << tl;
    u32 a = 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) a += (u32)(n[i] < 0 ? -n[i] : n[i]);
    u32 before = wg_scan_excl<W>(a, lane);                         // points assigned in front of my first symbol
    // position of the first non-zero counter above each of my symbols
    u32 nzLane = 0xFFFFFFFFu;
#pragma unroll
    for (int i = SPL - 1; i >= 0; --i) if (n[i] != 0 && SPL * sub + i <= maxSV) nzLane = SPL * sub + i;
    const u32 nzAbove = wg_suffix_min_excl<W>(nzLane, lane);
    int prevN = __shfl_up(n[SPL - 1], 1, WAVE);                      // counter of the symbol in front of my first one
    if (sub == 0) prevN = 1;
    // chunks of my symbols: value + zero-run code; sizes first
    u32 val[SPL], vnb[SPL], run[SPL], bits = 0;
    bool broken = false;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const u32 s = SPL * sub + i;
        const int remaining = (int)(ts + 1) - (int)before;
        const int pn = i ? n[i - 1] : prevN;
        const bool coded = s <= maxSV && remaining > 1 && !(n[i] == 0 && pn == 0);
        val[i] = 0; vnb[i] = 0; run[i] = 0xFFFFFFFFu;
        if (coded) {
            const u32 hb = hibit32((u32)remaining);
            const int threshold = 1 << hb, slack = 2 * threshold - 1 - remaining;
            int v = n[i] + 1;
            if (v >= threshold) v += slack;
            val[i] = (u32)v; vnb[i] = hb + 1 - (v < slack);
            bits += vnb[i];
            if (n[i] == 0) {                                   // first zero of a run: the rest of the run is counted here
                u32 e = 0xFFFFFFFFu;
#pragma unroll
                for (int j = SPL - 1; j > 0; --j) if (j > i && n[j] != 0 && SPL * sub + j <= maxSV) e = SPL * sub + j;
                if (e == 0xFFFFFFFFu) e = nzAbove;
                if (e == 0xFFFFFFFFu) broken = true;           // zeros up to the end of the alphabet: not a distribution
                else { const u32 R = e - (s + 1); run[i] = R; bits += 16u * (R / 24u) + 2u * ((R % 24u) / 3u) + 2u; }
            }
        }
        before += (u32)(n[i] < 0 ? -n[i] : n[i]);
    }
    const u32 total = wg_sum<W>(a);
    if (wg_any<W>(broken, lane) || total != ts) return FERR(GENERIC);
    const u32 mine = bits + (sub == 0 ? 4u : 0u);
    u32 pos = wg_scan_excl<W>(mine, lane);
    const u32 totalBits = wg_sum<W>(mine);
    const size_t bound = maxSV ? (size_t)((((maxSV + 1) * tl) >> 3) + 3) : (size_t)FSEHIP_FSE_NCOUNTBOUND;
    if (cap < bound) {
        const long lastFlush = 2 * (long)((totalBits - 1) >> 4);
        if (lastFlush > (long)cap - 2) return FERR(dstSize_tooSmall);
    }
    if (sub == 0) { wg_or_bits(img, 0, tl - FSE_MIN_TL, 4); pos += 4; }
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        wg_or_bits(img, pos, val[i], vnb[i]); pos += vnb[i];
        if (run[i] != 0xFFFFFFFFu) {
            u32 R = run[i];
            for (; R >= 24; R -= 24) { wg_or_bits(img, pos, 0xFFFFu, 16); pos += 16; }
            for (; R >= 3; R -= 3) { wg_or_bits(img, pos, 3u, 2); pos += 2; }
            wg_or_bits(img, pos, R, 2); pos += 2;
        }
    }
    return (size_t)((totalBits + 7) >> 3);
}
