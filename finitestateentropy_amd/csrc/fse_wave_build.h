// fse_wave_build.h -- wave-cooperative core of FSE_buildCTable / FSE_buildDTable (one 64-lane wave per table).
//
// Both reference builders (lib/fse_compress.c:96-133, lib/fse_decompress.c:86-124) do the same two serial walks:
//   1. "spread": visit table cells u = 0, step, 2*step, ... (mod tableSize), skipping cells above highThreshold, and hand
//      them out to the symbols in symbol order, normalizedCounter[s] cells each; the low-probability symbols (-1) take the
//      cells above highThreshold;
//   2. "rank": walk the cells in ascending u and number the cells of every symbol 0, 1, 2, ... in that order
//      (CTable: stateTable[cumul[s] + rank] = tableSize + u;  DTable: nextState = symbolNext[s] + rank).
// Restated without the loop-carried dependencies:
//   1. step is odd, so m -> (m * step) mod tableSize is a bijection and the serial walk visits m = 0, 1, 2, ... in
//      order; the k-th *kept* visit (u <= highThreshold) gets the symbol whose cumulative count range contains k.  Each lane
//      takes a contiguous range of m, a wave prefix sum of the kept visits gives its first k, and it walks the
//      (compacted) cumulative-count list from there.
//   2. rank(u) = number of cells u' < u with the same symbol.  Lane l owns the contiguous cells [l*C, (l+1)*C);
//      a byte matrix cnt[symbol][lane] counts the symbols per lane range (LDS atomic add with return = the rank inside
//      the range), a per-symbol running sum over groups of 4 lanes gives the ranks of everything before the range.
// The result is handed to `emit(u, symbol, rank)` once per cell, lane l emitting its own range in ascending u.
#pragma once
#include "dev_common.h"

#define WB_MAXSYM 256
#define WB_TSTEP(ts) (((ts) >> 1) + ((ts) >> 3) + 3)     // lib/fse.h:683

struct WaveBuildLds {        // LDS scratch of one wave, tableSize = 1 << tl <= capTs
    s16* nrm;                // [256] normalized counters, zero beyond maxSV (input)
    u16* cumP;               // [257] cumulative positive counts of the symbols listed in symP
    u8*  symP;               // [256] symbols with a positive count, ascending
    u8*  symTab;             // [capTs] symbol of every cell (output of the spread)
    u16* cell;               // [capTs] rank inside the lane range; the emitter may overwrite cell[u] with its result
    u32* cnt;                // [256 * 16] byte matrix cnt[symbol][lane]
    u16* coarse;             // [256 * 16] per symbol: cells before lane group j (4 lanes per group)
};
DEV size_t wave_build_lds_bytes(u32 capTs) { return 512 + 520 + 256 + (size_t)capTs + 2 * (size_t)capTs + 16384 + 8192; }
DEV WaveBuildLds wave_build_carve(u8* base, u32 capTs)
{
    WaveBuildLds w;
    w.cnt = (u32*)base; base += 16384;                   // 16-byte aligned parts first
    w.coarse = (u16*)base; base += 8192;
    w.cell = (u16*)base; base += 2 * (size_t)capTs;
    w.symTab = base; base += capTs;
    w.nrm = (s16*)base; base += 512;
    w.cumP = (u16*)base; base += 520;
    w.symP = base;
    return w;
}

DEV u32 wb_bytesum(u32 v) { return __builtin_amdgcn_sad_u8(v, 0u, 0u); }
DEV u32 wb_scan_excl(u32 v, u32 lane, u32* total)         // exclusive prefix sum over the 64 lanes
{
    u32 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)incl, off, WAVE); if ((int)lane >= off) incl += o; }
    *total = (u32)__shfl((int)incl, 63, WAVE);
    return incl - v;
}

// All 64 lanes of one wave call this with uniform arguments; w.nrm holds the counters.  Uses __syncthreads(), so the
// workgroup must be exactly this wave.  Returns the fastMode flag of FSE_buildDTable (no counter >= tableSize/2).
template <class Emit>
DEV bool wave_spread_rank(const WaveBuildLds& w, u32 maxSV, u32 tl, u32 lane, Emit&& emit)
{
    const u32 ts = 1u << tl, mask = ts - 1, step = WB_TSTEP(ts);
    // ---- per symbol: lane l looks after symbols 4l .. 4l+3
    int n[4];
    {   const uint2 raw = *(const uint2*)(w.nrm + 4 * lane);
        n[0] = (s16)(raw.x & 0xFFFFu); n[1] = (s16)(raw.x >> 16); n[2] = (s16)(raw.y & 0xFFFFu); n[3] = (s16)(raw.y >> 16);
    }
    u32 lanePos = 0, laneLow = 0, lanePres = 0; bool big = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (4 * lane + i > maxSV) n[i] = 0;
        lanePos += n[i] > 0 ? (u32)n[i] : 0u; laneLow += n[i] == -1; lanePres += n[i] > 0;
        big |= n[i] >= (int)(ts >> 1);
    }
    u32 sumPos, nLow, nPres;
    u32 posBase = wb_scan_excl(lanePos, lane, &sumPos);
    u32 lowBase = wb_scan_excl(laneLow, lane, &nLow);
    u32 presBase = wb_scan_excl(lanePres, lane, &nPres);
    const int high = (int)ts - 1 - (int)nLow;                             // highThreshold (-1: every cell is a low-probability one)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (n[i] > 0) { w.symP[presBase] = (u8)(4 * lane + i); w.cumP[presBase] = (u16)posBase; ++presBase; posBase += (u32)n[i]; }
        else if (n[i] == -1) { w.symTab[ts - 1 - lowBase] = (u8)(4 * lane + i); ++lowBase; }
    }
    if (lane == 0) w.cumP[nPres] = (u16)sumPos;
    // ---- clear the count matrix rows in use
    {   const u32 rows16 = (maxSV + 1) * 16;                              // dwords
        for (u32 i = 4 * lane; i < rows16; i += 256) *(uint4*)(w.cnt + i) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // ---- spread: lane l visits m in [l*C, (l+1)*C)
    const u32 C = ts >= 64 ? ts >> 6 : 1;
    const bool act = lane * C < ts;
    const u32 m0 = lane * C;
    u32 nv = 0;
    if (act) { u32 u = (m0 * step) & mask; for (u32 i = 0; i < C; ++i) { nv += (int)u <= high; u = (u + step) & mask; } }
    u32 totalKept;
    u32 k = wb_scan_excl(nv, lane, &totalKept);
    if (act && nv) {
        u32 lo = 0, hi = nPres;                                           // cumP[lo] <= k < cumP[hi]
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (w.cumP[mid] <= k) lo = mid; else hi = mid; }
        u32 j = lo, nextCum = w.cumP[j + 1], cur = w.symP[j];
        u32 u = (m0 * step) & mask;
        for (u32 i = 0; i < C; ++i) {
            if ((int)u <= high) {
                while (k >= nextCum) { ++j; nextCum = w.cumP[j + 1]; cur = w.symP[j]; }
                w.symTab[u] = (u8)cur; ++k;
            }
            u = (u + step) & mask;
        }
    }
    __syncthreads();

    // ---- rank inside the lane range: LDS atomics return the previous count, in program order
    const u32 sh8 = 8 * (lane & 3u), grp = lane >> 2;
    if (act) {
        for (u32 i = 0; i < C; ++i) {
            const u32 u = m0 + i;
            const u32 s = w.symTab[u];
            const u32 old = atomicAdd(&w.cnt[s * 16 + grp], 1u << sh8);
            w.cell[u] = (u16)((old >> sh8) & 0xFFu);
        }
    }
    __syncthreads();
    // ---- per symbol: running sum over the 16 lane groups
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (n[i] == 0) continue;
        const u32 s = 4 * lane + i;
        u32 run = 0;
        for (u32 g4 = 0; g4 < 16; g4 += 4) {
            const uint4 c = *(const uint4*)(w.cnt + s * 16 + g4);
            const u32 r0 = run, r1 = r0 + wb_bytesum(c.x), r2 = r1 + wb_bytesum(c.y), r3 = r2 + wb_bytesum(c.z);
            run = r3 + wb_bytesum(c.w);
            *(uint2*)(w.coarse + s * 16 + g4) = make_uint2(r0 | (r1 << 16), r2 | (r3 << 16));
        }
    }
    __syncthreads();
    // ---- emit
    if (act) {
        const u32 belowMask = (1u << sh8) - 1u;
        for (u32 i = 0; i < C; ++i) {
            const u32 u = m0 + i;
            const u32 s = w.symTab[u];
            const u32 r = (u32)w.cell[u] + (u32)w.coarse[s * 16 + grp] + wb_bytesum(w.cnt[s * 16 + grp] & belowMask);
            emit(u, s, r);
        }
    }
    __syncthreads();
    return !__any(big);
}
