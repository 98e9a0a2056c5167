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
//      order; the k-th *kept* visit (u <= highThreshold) gets the symbol whose cumulative count range contains k.  Every
//      symbol marks the first k of its range with symbol+1; each lane takes a contiguous range of m, a wave prefix sum of
//      the kept visits gives its first k, and the symbol of a visit is the running maximum of the marks up to its k
//      (a wave prefix maximum carries it across lanes).
//   2. rank(u) = number of cells u' < u with the same symbol.  Lane l owns the contiguous cells [l*C, (l+1)*C);
//      a byte matrix cnt[symbol][lane] counts the symbols per lane range (LDS atomic add with return = the rank inside
//      the range), a per-symbol running sum over groups of 4 lanes gives the ranks of everything before the range.
//      The matrix holds WB_WIN symbols; larger alphabets take one rank / sum / emit pass per window of symbols.
// The result is handed to `emit(u, symbol, rank, payload(symbol))` once per cell, lane l emitting its own range in
// ascending u; `payload` is a per-symbol LDS lookup of the caller's that is gathered together with the core's own.
#pragma once
#include "dev_common.h"

#define WB_MAXSYM 256
// The rank matrix covers a window of WB_WIN symbols at a time (alphabets up to 64 symbols: one pass; measured per 100k blocks,
// compress + decompress build: window 128 -> 64: Proba14 1.88 -> 1.64 ms, tableLog 12 4.48 -> 3.55 ms, Proba02 3.24 -> 3.72 ms; the LDS footprint
// decides how many table builds a CU runs at once, and these builds are latency-bound).
#ifndef WB_WIN
#define WB_WIN 64u
#endif
#define WB_TSTEP(ts) (((ts) >> 1) + ((ts) >> 3) + 3)     // lib/fse.h:683

#ifdef FSE_WB_TIMING         // development aid: phase cycle accounting of the last table build of each workgroup
__device__ unsigned long long g_wbTiming[4096 * 8];
#define WBT(k) TT[k] = __builtin_readcyclecounter();
#else
#define WBT(k)
#endif
struct WaveBuildLds {        // LDS scratch of one wave, tableSize = 1 << tl <= capTs
    s16* nrm;                // [256] normalized counters, zero beyond maxSV (input)
    u16* cumP;               // [257] (unused by the core; callers may use it)
    u8*  symP;               // [256] symbols in use (counter != 0), ascending
    u8*  symTab;             // [wb_si(capTs)] symbol of every cell (output of the spread), index through wb_si()
    u16* cell;               // [wb_ci(capTs)] rank inside the lane range, index through wb_ci(); the emitter may overwrite cell[wb_ci(u)] with its result
    u32* cnt;                // [WB_WIN * 16] byte matrix cnt[symbol - window base][lane]
    u16* coarse;             // [WB_WIN * 16] per symbol of the window: cells before lane group j (4 lanes per group)
};
// cell[] / symTab[] are indexed through wb_ci() / wb_si(): every row of 32 cells is followed by 4 bytes of padding.  Lane l
// works on cells [l*C, (l+1)*C) (C = 32 at tableLog 11), so without the padding the 64 lanes of one LDS instruction
// would sit 64 (or 32) bytes apart -- on 2 (4) of the 32 banks; with it they are 17 (9) dwords apart: conflict-free.
__host__ __device__ inline u32 wb_ci(u32 u) { return u + ((u >> 5) << 1); }     // u16 arrays (cell, marks)
__host__ __device__ inline u32 wb_si(u32 u) { return u + ((u >> 5) << 2); }     // u8 array (symTab)
__host__ __device__ inline size_t wave_build_lds_bytes(u32 capTs) { return WB_WIN * 64 + WB_WIN * 32 + 2 * (size_t)wb_ci(capTs) + wb_si(capTs) + 512 + 520 + 256; }
DEV WaveBuildLds wave_build_carve(u8* base, u32 capTs)
{
    WaveBuildLds w;
    w.cnt = (u32*)base; base += WB_WIN * 64;             // 16-byte aligned parts first
    w.coarse = (u16*)base; base += WB_WIN * 32;
    w.cell = (u16*)base; base += 2 * (size_t)wb_ci(capTs);
    w.symTab = base; base += wb_si(capTs);
    w.nrm = (s16*)base; base += 512;
    w.cumP = (u16*)base; base += 520;
    w.symP = base;
    return w;
}

DEV u32 wb_bytesum(u32 v) { return __builtin_amdgcn_sad_u8(v, 0u, 0u); }
DEV u32 wb_scan_excl(u32 v, u32 lane, u32* total)         // exclusive prefix sum over the 64 lanes
{
#if FSEHIP_DPP_SCANS
    const u32 incl = group_scan_incl<64, ScanAdd>(v, lane);
    *total = group_last<64>(incl, lane);
#else
    u32 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)incl, off, WAVE); if ((int)lane >= off) incl += o; }
    *total = (u32)__shfl((int)incl, 63, WAVE);
#endif
    return incl - v;
}

// All 64 lanes of one wave call this with uniform arguments; w.nrm holds the counters.  Uses __syncthreads(), so the
// workgroup must be exactly this wave.  Returns the fastMode flag of FSE_buildDTable (no counter >= tableSize/2).
template <class Payload, class Emit>
DEV bool wave_spread_rank(const WaveBuildLds& w, u32 maxSV, u32 tl, u32 lane, Payload&& payload, Emit&& emit)
{
#ifdef FSE_WB_TIMING
    unsigned long long TT[8];
#endif
    WBT(0)
    const u32 ts = 1u << tl, mask = ts - 1, step = WB_TSTEP(ts);
    const u32 C = ts >= 64 ? ts >> 6 : 1;                                 // cells (and visits) per lane
    const bool act = lane * C < ts;
    const u32 m0 = lane * C;
    u16* const marks = w.cell;                                            // [ts] scratch of the spread (cell[] is not in use yet)
    // ---- per symbol: lane l looks after symbols 4l .. 4l+3
    int n[4];
    {   const uint2 raw = *(const uint2*)(w.nrm + 4 * lane);
        n[0] = (s16)(raw.x & 0xFFFFu); n[1] = (s16)(raw.x >> 16); n[2] = (s16)(raw.y & 0xFFFFu); n[3] = (s16)(raw.y >> 16);
    }
    u32 lanePos = 0, laneLow = 0, laneAny = 0; bool big = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (4 * lane + i > maxSV) n[i] = 0;
        lanePos += n[i] > 0 ? (u32)n[i] : 0u; laneLow += n[i] == -1; laneAny += n[i] != 0;
        big |= n[i] >= (int)(ts >> 1);
    }
    // one packed scan: positive counts (<= 4096: 13 bits) | low-probability symbols (9 bits) | symbols in use (9 bits)
    u32 totals;
    const u32 base3 = wb_scan_excl(lanePos | (laneLow << 13) | (laneAny << 22), lane, &totals);
    u32 posBase = base3 & 0x1FFFu, lowBase = (base3 >> 13) & 0x1FFu, anyBase = base3 >> 22;
    const u32 nLow = (totals >> 13) & 0x1FFu, nAny = totals >> 22;
    const int high = (int)ts - 1 - (int)nLow;                             // highThreshold (-1: every cell is a low-probability one)
    // clear the spread marks and the count matrix rows in use
    if (act) { if (C >= 2) for (u32 i = 0; i < C; i += 2) *(u32*)(marks + wb_ci(m0 + i)) = 0; else marks[wb_ci(m0)] = 0; }
    __syncthreads();
    WBT(1)
    // symbols in use (for the per-symbol pass below); low-probability symbols take the top cells; every symbol with a
    // positive count marks the first of its kept visits with symbol+1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (n[i] != 0) w.symP[anyBase++] = (u8)(4 * lane + i);
        if (n[i] > 0) { marks[wb_ci(posBase)] = (u16)(4 * lane + i + 1); posBase += (u32)n[i]; }
        else if (n[i] == -1) { w.symTab[wb_si(ts - 1 - lowBase)] = (u8)(4 * lane + i); ++lowBase; }
    }
    // ---- spread: lane l visits m in [l*C, (l+1)*C); the k-th kept visit belongs to the symbol of the last mark at or
    //      before k, i.e. a running maximum over the marks (symbols ascend with k)
    u32 nv = 0;
    if (act) { u32 u = (m0 * step) & mask; for (u32 i = 0; i < C; ++i) { nv += (int)u <= high; u = (u + step) & mask; } }
    u32 totalKept;
    const u32 k0 = wb_scan_excl(nv, lane, &totalKept);
    __syncthreads();
    WBT(2)
    // (eight marks at a time so that the LDS round trips overlap; indices are clamped, unused values are ignored)
    u32 localMax = 0;
    for (u32 i = 0; i < nv; i += 8) {
        u32 v[8];
#pragma unroll
        for (u32 j = 0; j < 8; ++j) { const u32 idx = k0 + i + j; v[j] = marks[wb_ci(idx < ts ? idx : ts - 1)]; }
#pragma unroll
        for (u32 j = 0; j < 8; ++j) if (i + j < nv) localMax = v[j] > localMax ? v[j] : localMax;
    }
    u32 run = localMax;                                                   // inclusive prefix maximum over the lanes ...
#if FSEHIP_DPP_SCANS
    run = group_scan_incl<64, ScanMax>(run, lane);
    run = dpp_mov<0x138, 0xF, true>(0u, run);                            // ... made exclusive: wave_shr:1, lane 0 gets 0
#else
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)run, off, WAVE); if ((int)lane >= off) run = o > run ? o : run; }
    run = (u32)__shfl_up((int)run, 1, WAVE);                             // ... made exclusive
    if (lane == 0) run = 0;
#endif
    if (act && nv) {
        u32 u = (m0 * step) & mask, k = k0;
        u32 i = 0;
        for (; i + 8 <= C; i += 8) {
            u32 uu[8], v[8]; bool keep[8];
#pragma unroll
            for (u32 j = 0; j < 8; ++j) {
                uu[j] = u; keep[j] = (int)u <= high;
                v[j] = marks[wb_ci(k < ts ? k : ts - 1)];
                k += keep[j]; u = (u + step) & mask;
            }
#pragma unroll
            for (u32 j = 0; j < 8; ++j) if (keep[j]) { run = v[j] > run ? v[j] : run; w.symTab[wb_si(uu[j])] = (u8)(run - 1u); }
        }
        for (; i < C; ++i) {
            if ((int)u <= high) { const u32 v = marks[wb_ci(k)]; ++k; run = v > run ? v : run; w.symTab[wb_si(u)] = (u8)(run - 1u); }
            u = (u + step) & mask;
        }
    }
    __syncthreads();
    WBT(3)

    // ---- per window of WB_WIN symbols: rank inside the lane range, running sums over the lane groups, emit.
    //      (The emitter may overwrite cell[wb_ci(u)] of the cells it is called for; the other cells still hold their
    //      local rank, which later windows read.)
    const u32 sh8 = 8 * (lane & 3u), grp = lane >> 2;
    const u32 belowMask = (1u << sh8) - 1u;
    const bool single = maxSV < WB_WIN;                                   // uniform: one window, nothing to predicate
    for (u32 base = 0; base <= maxSV; base += WB_WIN) {
        {   const u32 rows = (maxSV + 1 - base) < WB_WIN ? (maxSV + 1 - base) : WB_WIN;
            for (u32 i = 4 * lane; i < rows * 16; i += 256) *(uint4*)(w.cnt + i) = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        // rank inside the lane range: LDS atomics return the previous count, in program order.  Eight cells at a time so
        // that the LDS round trips overlap (the symbols of a lane's cells are contiguous bytes).
        if (act) {
            u32 i = 0;
            for (; i + 8 <= C; i += 8) {
                const u32* const syp = (const u32*)(w.symTab + wb_si(m0 + i));    // 8 cells of one row: two aligned dwords
                const uint2 sy = make_uint2(syp[0], syp[1]);
                u32* const cp = (u32*)(w.cell + wb_ci(m0 + i));
                uint4 prev = make_uint4(0, 0, 0, 0);
                if (!single) prev = make_uint4(cp[0], cp[1], cp[2], cp[3]);        // local ranks written by earlier windows
                u32 old[8]; bool in[8];
#pragma unroll
                for (u32 j = 0; j < 8; ++j) {
                    const u32 s = (((j < 4 ? sy.x : sy.y) >> (8 * (j & 3))) & 0xFFu) - base;
                    in[j] = single || s < WB_WIN;
                    old[j] = 0;
                    if (in[j]) old[j] = atomicAdd(&w.cnt[s * 16 + grp], 1u << sh8);
                }
#pragma unroll
                for (u32 j = 0; j < 8; ++j) {
                    const u32 pw = j < 2 ? prev.x : j < 4 ? prev.y : j < 6 ? prev.z : prev.w;
                    old[j] = in[j] ? (old[j] >> sh8) & 0xFFu : (pw >> (16 * (j & 1))) & 0xFFFFu;
                }
                cp[0] = old[0] | (old[1] << 16); cp[1] = old[2] | (old[3] << 16); cp[2] = old[4] | (old[5] << 16); cp[3] = old[6] | (old[7] << 16);
            }
            for (; i < C; ++i) {
                const u32 u = m0 + i;
                const u32 s = (u32)w.symTab[wb_si(u)] - base;
                if (single || s < WB_WIN) {
                    const u32 old = atomicAdd(&w.cnt[s * 16 + grp], 1u << sh8);
                    w.cell[wb_ci(u)] = (u16)((old >> sh8) & 0xFFu);
                }
            }
        }
        __syncthreads();
        WBT(4)
        // per symbol in use (lane l: the l-th, l+64-th, ... of them): running sum over the 16 lane groups
        for (u32 j = lane; j < nAny; j += 64) {
            const u32 s = (u32)w.symP[j] - base;
            if (!single && s >= WB_WIN) continue;
            const uint4* const row = (const uint4*)(w.cnt + s * 16);
            const uint4 c0 = row[0], c1 = row[1], c2 = row[2], c3 = row[3];
            u32 r[16];
            u32 acc = 0;
#define WB_ACC(k, v) r[k] = acc; acc += wb_bytesum(v);
            WB_ACC(0, c0.x) WB_ACC(1, c0.y) WB_ACC(2, c0.z) WB_ACC(3, c0.w) WB_ACC(4, c1.x) WB_ACC(5, c1.y) WB_ACC(6, c1.z) WB_ACC(7, c1.w)
            WB_ACC(8, c2.x) WB_ACC(9, c2.y) WB_ACC(10, c2.z) WB_ACC(11, c2.w) WB_ACC(12, c3.x) WB_ACC(13, c3.y) WB_ACC(14, c3.z) WB_ACC(15, c3.w)
#undef WB_ACC
            uint4* const out = (uint4*)(w.coarse + s * 16);
            out[0] = make_uint4(r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16));
            out[1] = make_uint4(r[8] | (r[9] << 16), r[10] | (r[11] << 16), r[12] | (r[13] << 16), r[14] | (r[15] << 16));
        }
        __syncthreads();
        WBT(5)
        // emit: everything a cell needs is gathered for eight cells before the first one is emitted
        if (act) {
            u32 i = 0;
            for (; i + 8 <= C; i += 8) {
                const u32* const syp = (const u32*)(w.symTab + wb_si(m0 + i));
                const uint2 sy = make_uint2(syp[0], syp[1]);
                const u32* const cp = (const u32*)(w.cell + wb_ci(m0 + i));
                const uint4 lr = make_uint4(cp[0], cp[1], cp[2], cp[3]);
                u32 sf[8], co[8], cn[8], pl[8]; bool in[8];
#pragma unroll
                for (u32 j = 0; j < 8; ++j) {
                    sf[j] = ((j < 4 ? sy.x : sy.y) >> (8 * (j & 3))) & 0xFFu;
                    const u32 sw = sf[j] - base;
                    in[j] = single || sw < WB_WIN;
                    const u32 row = in[j] ? sw : 0u;
                    co[j] = w.coarse[row * 16 + grp]; cn[j] = w.cnt[row * 16 + grp]; pl[j] = payload(sf[j]);
                }
#pragma unroll
                for (u32 j = 0; j < 8; ++j) {
                    const u32 lrw = j < 2 ? lr.x : j < 4 ? lr.y : j < 6 ? lr.z : lr.w;
                    const u32 local = (lrw >> (16 * (j & 1))) & 0xFFFFu;
                    if (in[j]) emit(m0 + i + j, sf[j], local + co[j] + wb_bytesum(cn[j] & belowMask), pl[j]);
                }
            }
            for (; i < C; ++i) {
                const u32 u = m0 + i;
                const u32 sfull = w.symTab[wb_si(u)];
                const u32 sw = sfull - base;
                if (single || sw < WB_WIN) {
                    const u32 r = (u32)w.cell[wb_ci(u)] + (u32)w.coarse[sw * 16 + grp] + wb_bytesum(w.cnt[sw * 16 + grp] & belowMask);
                    emit(u, sfull, r, payload(sfull));
                }
            }
        }
        __syncthreads();
    }
    WBT(6)
#ifdef FSE_WB_TIMING
    if (lane == 0 && blockIdx.x < 4096) for (int q = 0; q < 6; ++q) g_wbTiming[8 * blockIdx.x + q] = TT[q + 1] - TT[q];
#endif
    return !__any(big);
}
