// fse_tables.hip -- glue around the FSE hot loops, on the device (SURVEY 8(a') rows g1-g4):
//   compress side   : early-outs, FSE_optimalTableLog, FSE_normalizeCount, FSE_writeNCount, FSE_buildCTable
//                     (reference: lib/fse_compress.c:632-677, :325-342, :348-494, :192-298, :66-169)
//   decompress side : FSE_readNCount, FSE_buildDTable
//                     (reference: lib/entropy_common.c:41-144, lib/fse_decompress.c:71-126,255-274)
// Mapping: one wave per block wherever the work is wide (k_fse_cprep, k_fse_dbuild: per-symbol lanes, scans, the wave-cooperative
// spread / rank); one lane per block only for the NCount header parser, whose fields depend on all fields before them
// (k_fse_dparse).  The CTable is produced in the reference's in-memory layout, the decoding table in the decoder's own cell formats,
// both in global scratch; the hot-loop kernels stage them into LDS.
#include "internal.h"

#include "wave_glue.h"
#include "ncount_reader.h"
#include "fse_wave_build.h"

// ---------------------------------------------------------------------------------------------------
//  prepare kernels
// ---------------------------------------------------------------------------------------------------
// FSE_buildCTable_wksp (lib/fse_compress.c:66-169) from the normalised counters a wave holds in registers (lane l: symbols 4l .. 4l+3, zero beyond
// maxSV): the table is written straight to global memory, symbolTT entries coalesced, stateTable entries as scattered 2-byte stores inside the
// block's 4 KiB (the L2 merges them) -- an LDS image would cost 6 KiB per build, i.e. occupancy.  Shared by k_fse_cprep (counters it has just
// normalised) and k_fse_ctable_from_norm (the caller's counters: FSE_buildCTable as a call of its own).  Uses __syncthreads(): the workgroup is the wave.
DEV void fse_wave_build_ctable(const WaveBuildLds& w, const int nn[4], u32 maxSV, u32 tl, u32* img, u32 lane)
{
    u16* const cumAll = w.cumP;                                            // [256] first stateTable slot of every symbol (the core leaves cumP to its caller)
    *(uint2*)(w.nrm + 4 * lane) = make_uint2(((u32)nn[0] & 0xFFFFu) | ((u32)nn[1] << 16), ((u32)nn[2] & 0xFFFFu) | ((u32)nn[3] << 16));
    __syncthreads();
    const u32 ts = 1u << tl;
    u16* const stateTable = (u16*)(img + 1);
    u32* const tt = img + 1 + (ts >> 1);                                   // tl >= 1 here (lib/fse_compress.c:75: one word of state table when tableLog is 0)
    {   // per symbol (lane l: symbols 4l..4l+3): cumulative slot, symbolTT (fse_compress.c:136-166)
        u32 eff[4], laneSum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { eff[i] = nn[i] == -1 ? 1u : (u32)nn[i]; laneSum += eff[i]; }
        u32 total;
        u32 run = wb_scan_excl(laneSum, lane, &total);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 s = 4 * lane + i;
            cumAll[s] = (u16)run;
            if (s <= maxSV) {
                if (nn[i] == 0) { tt[2 * s] = 0; tt[2 * s + 1] = ((tl + 1) << 16) - ts; }
                else if (nn[i] == -1 || nn[i] == 1) { tt[2 * s] = run - 1u; tt[2 * s + 1] = (tl << 16) - ts; }
                else {
                    const u32 maxBitsOut = tl - hibit32((u32)nn[i] - 1);
                    tt[2 * s] = run - (u32)nn[i];
                    tt[2 * s + 1] = (maxBitsOut << 16) - ((u32)nn[i] << maxBitsOut);
                }
            }
            run += eff[i];
        }
        if (lane == 0) img[0] = tl | (maxSV << 16);
    }
    __syncthreads();
    wave_spread_rank(w, maxSV, tl, lane, [&](u32 s) { return (u32)cumAll[s]; },
                     [&](u32 u, u32 s, u32 r, u32 first) { (void)s; stateTable[first + r] = (u16)(ts + u); });   // :125-133
}

// Compress side: k_fse_cprep, one wave per block -- everything between the histogram and the hot loop
// (lib/fse_compress.c:632-665): the early outs of FSE_compress_wksp, the table log, the normalised counters and the NCount
// header (wave_glue.h: every symbol handled independently, totals by wave reductions, bit offsets by scans), then
// FSE_buildCTable_wksp (lib/fse_compress.c:66-169) with the wave-cooperative spread / rank of fse_wave_build.h.  The counters
// never leave the wave's registers / LDS; the header is assembled in LDS and copied out once.
__global__ __launch_bounds__(64) void k_fse_cprep(FseCPrepArgs a, u32 capTs)
{
    extern __shared__ __attribute__((aligned(16))) u8 wbLds[];
    const size_t b = blockIdx.x;
    const u32 lane = threadIdx.x;
    const size_t n = view_size(a.src, b);
    FseMeta m; m.state = 0; m.hdrSize = 0; m.tableLog = 0; m.maxSV = 0; m.pace = 0;
    // early outs (fse_compress.c:647-655), uniform
    const size_t top = a.histResults[b];
    size_t result = 0; bool go = false;
    if (n <= 1) result = 0;                                                // not compressible
    else if (is_err(top)) result = top;
    else if (top == n) result = 1;                                         // one symbol only: rle
    else if (top == 1 || top < (n >> 7)) result = 0;                       // every symbol once / too flat to pay off
    else go = true;
    const WaveBuildLds w = wave_build_carve(wbLds, capTs);
    u32* const hdrImg = w.cnt;                                             // header image; the rank matrix is idle until the build
    u32 maxSV = 0, tl = 0;
    int nn[4] = { 0, 0, 0, 0 };
    if (go) {
        maxSV = a.maxSVs[b];
        tl = wg_optimal_tablelog(a.tableLogReq, n, maxSV, 2);             // :649,658
        const uint4 cv = ((const uint4*)(a.counts + b * 256))[lane];       // symbols 4*lane .. 4*lane+3
        const u32 c[4] = { cv.x, cv.y, cv.z, cv.w };
        for (u32 i = lane; i < 136; i += 64) hdrImg[i] = 0;
        const size_t e = wg_normalize<64>(nn, c, (u64)n, maxSV, tl, lane);     // :659
        if (is_err(e)) { result = e; go = false; }
    }
    __syncthreads();
    if (go) {
        const size_t h = wg_write_ncount<64>(hdrImg, a.dstCapacity, nn, maxSV, tl, lane);   // :662
        if (is_err(h)) { result = h; go = false; }
        else m.hdrSize = (u32)h;
    }
    __syncthreads();
    if (!go) { if (lane == 0) { a.meta[b] = m; a.results[b] = result; } return; }          // uniform
    {   u8* const dst = a.dst + b * a.dstStride;
        const u8* const hb = (const u8*)hdrImg;
        for (u32 i = lane; i < m.hdrSize; i += 64) dst[i] = hb[i];
    }
    // Encoder choice.  The wave-per-block kernel (fse_encode_wave.hip) is the default; it needs every lane's share of the
    // output to span at least one byte, which fails when one symbol takes (almost) the whole table: those blocks go to the
    // lane-per-block kernel.  (The wave kernel re-checks exactly and hands further blocks over at run time.)
    int top1 = nn[0] > nn[1] ? nn[0] : nn[1]; top1 = nn[2] > top1 ? nn[2] : top1; top1 = nn[3] > top1 ? nn[3] : top1;
    top1 = wave_max_i32(top1);
    m.state = ((u32)top1 * 64u > (63u << tl) || n < FSE_ENC_WAVE_MIN) ? FSE_ENC_LANE : FSE_ENC_PAR;     // (short blocks: 64 lanes do not pay, internal.h)
    {   // pace bin for the encoder's block order (internal.h): tableSize / symbols in use = the merging time of two encoder states
        u32 present = (nn[0] != 0) + (nn[1] != 0) + (nn[2] != 0) + (nn[3] != 0);
        present = wg_sum<64>(present);
        const u32 mix = (1u << tl) / (present ? present : 1u);
        m.pace = mix < 64u ? 0u : mix < 128u ? 1u : mix < 256u ? 2u : 3u;
    }
    m.tableLog = tl; m.maxSV = maxSV;
    if (lane == 0) a.meta[b] = m;
    fse_wave_build_ctable(w, nn, maxSV, tl, a.ctables + b * a.ctStrideU32, lane);
}

// Decompress side, two kernels:
//   k_fse_dparse : one lane per block (the header is a serial variable-length code: ncount_reader.h) plus the checks of
//                  FSE_decompress_wksp (lib/fse_decompress.c:264-269); sorts the blocks into the decoder classes of
//                  internal.h and leaves the counters in scratch;
//   k_fse_dbuild : one wave per block -- FSE_buildDTable (lib/fse_decompress.c:71-126) with the wave-cooperative
//                  spread / rank of fse_wave_build.h, emitting the decoder's compact cell formats (u16 cell, symbol in
//                  a separate byte table; bit-reversed layout when maxLog <= 11, see fse_decode.hip) with coalesced stores.
__global__ __launch_bounds__(64) void k_fse_dparse(FseDPrepArgs a)
{
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    const u32 lane = threadIdx.x;
    int cls = -1;                                                          // decoder class of my block (-1: none / finished here)
    if (b < a.nBlocks) {
        FseMeta m; m.state = 0; m.hdrSize = 0; m.tableLog = 0; m.maxSV = 0; m.pace = 0; m.pace = 0;
        const u8* const in = view_ptr(a.csrc, b);
        const size_t cSize = view_size(a.csrc, b);
        size_t result = 0;
        bool done = false;                                                 // a raw / RLE record of a packed batch: regenerated by k_rawrle_expand, result written there
        if (a.rawRle) { const size_t orig = a.origSizes ? a.origSizes[b] : a.uniformOrig; done = cSize == orig || cSize == 1; }
        if (!done) do {
            u32 tl = 0, maxSV = 255;
            s16* const norm = a.norms + b * 256;
            const size_t h = ncount_read<1>(norm, &maxSV, &tl, in, cSize);     // fse_decompress.c:264
            if (is_err(h)) { result = h; break; }
            if (tl > a.maxLog) { result = FERR(tableLog_tooLarge); break; }    // :266
            // class by the block's own tableLog (internal.h); a counter above half the table makes cells with nbBits == 0
            cls = FSE_DCLS_REV11;
            if (tl > FSE_DEC_FAST_MAXLOG) {
                int top = 0;
                for (u32 s = 0; s <= maxSV; ++s) top = norm[s] > top ? norm[s] : top;
                cls = 2 * top > (1 << tl) ? FSE_DCLS_PLAIN : FSE_DCLS_REV12;
            }
            m.state = 1u | ((u32)cls << 2); m.hdrSize = (u32)h; m.tableLog = tl; m.maxSV = maxSV;
        } while (0);
        a.meta[b] = m;
        if (m.state == 0 && !done) a.results[b] = result;
    }
    // append my block to the list of its class and size bin: one atomic per list and wave
    if (cls >= 0) { const size_t bin = view_size(a.csrc, b) >> FSE_DBIN_LOG; cls = cls * FSE_DBINS + (int)(bin < FSE_DBINS - 1 ? bin : FSE_DBINS - 1); }
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long todo = __ballot(cls >= 0);
    while (todo) {                                                         // uniform: one round per list some block of this wave goes to
        const int leader = __builtin_ctzll(todo);
        const int c = __shfl(cls, leader, WAVE);
        const unsigned long long mask = __ballot(cls == c);
        u32 base = 0;
        if ((int)lane == leader) {
            base = atomicAdd(&a.counts[c], (u32)__builtin_popcountll(mask));
            atomicAdd(&a.counts[FSE_DCLS_COUNT + c / FSE_DBINS], (u32)__builtin_popcountll(mask));     // the class's total (what its launches' surplus workgroups look at)
        }
        base = (u32)__shfl((int)base, leader, WAVE);
        if (cls == c) a.lists[(size_t)c * a.nBlocks + base + (u32)__builtin_popcountll(mask & below)] = (u32)b;
        todo &= ~mask;
    }
}

// Workgroup w builds the w-th block of the class lists [firstList, firstList + nLists) taken one after the other (the lists sit
// `nBlocks` entries apart; their lengths are device-side, so the grid is sized for the worst case and the surplus exits).
__global__ __launch_bounds__(64) void k_fse_dbuild(FseDPrepArgs a, u32 capTs, int firstList, int nLists, u32 ldsCapTs)
{
    extern __shared__ __attribute__((aligned(16))) u8 wbLds[];
    // the grid is sized for the worst case: a surplus workgroup (all of them in a launch for a class nobody belongs to) leaves after one look
    // at the class totals
    {   u32 total = 0;
        for (int k = firstList / (int)FSE_DBINS; k < (firstList + nLists) / (int)FSE_DBINS; ++k) total += a.counts[FSE_DCLS_COUNT + k];
        if (blockIdx.x >= total) return;                                   // uniform
    }
    // the list lengths are loaded together (16-byte loads, one memory latency), then scanned in registers
    constexpr int MAXL = 2 * FSE_DBINS;                                    // a launch walks the lists of one or two classes
    static_assert(FSE_DBINS % 4 == 0, "the list lengths are read as 16-byte vectors");
    uint4 cv[MAXL / 4];
#pragma unroll
    for (int i = 0; i < MAXL / 4; ++i) cv[i] = 4 * i < nLists ? ((const uint4*)(a.counts + firstList))[i] : make_uint4(0, 0, 0, 0);
    u32 idx = blockIdx.x;
    int c = firstList + nLists;
    {   u32 before = 0; bool found = false;
#pragma unroll
        for (int i = 0; i < MAXL; ++i) {
            const uint4 v = cv[i >> 2];
            const u32 n = (i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w;
            const bool here = !found & (i < nLists) & (idx < before + n);      // (selects, not branches: the branchy form of the same scan in k_fse_decode lost an assignment)
            c = here ? firstList + i : c; idx = here ? idx - before : idx;
            found |= here;
            before += n;
        }
    }
    if (c == firstList + nLists) return;                                   // uniform: beyond the last list
    const size_t b = a.lists[(size_t)c * a.nBlocks + idx];
    const u32 lane = threadIdx.x;
    const FseMeta m = a.meta[b];
    if (m.state == 0) return;                                              // uniform
    const WaveBuildLds w = wave_build_carve(wbLds, ldsCapTs);
    *(uint2*)(w.nrm + 4 * lane) = *(const uint2*)(a.norms + b * 256 + 4 * lane);
    __syncthreads();
    const u32 tl = m.tableLog, ts = 1u << tl;
    const bool rev = ((m.state >> 2) & 3u) != FSE_DCLS_PLAIN;              // uniform
    const bool fast = wave_spread_rank(w, m.maxSV, tl, lane, [&](u32 s) { return (u32)(int)w.nrm[s]; }, [&](u32 u, u32 s, u32 r, u32 nrm) {
        (void)s;
        const int n = (int)nrm;
        const u32 next = (n > 0 ? (u32)n : 1u) + r;                        // symbolNext[s]++, fse_decompress.c:117-122
        const u32 nb = tl - hibit32(next);
        const u32 ns = (next << nb) - ts;
        // bit-reversed format (see fse_decode.hip): nbBits | rev_tl(newState) << 5 (rev_tl(newState) < 2048: tableLog <= 11, or
        // tableLog 12 with nbBits >= 1, i.e. newState even); the cell of state u goes to position rev_tl(u), which the copy-out
        // below takes care of
        if (rev) w.cell[wb_ci(u)] = (u16)(nb | ((__brev(ns) >> (32u - tl)) << 5));
        else     w.cell[wb_ci(u)] = (u16)((ns & 0xFFFu) | (nb << 12));
    });
    u32* const A32 = (u32*)(a.atab + b * capTs);
    u32* const S32 = (u32*)(a.symtab + b * capTs);
    if (ts < 4u) {                                                         // uniform: a table of two cells (FSE_buildDTable on a caller's counters; no header says tableLog 1)
        if (lane < ts) { const u32 x = rev ? __brev(lane) >> (32u - tl) : lane; (a.atab + b * capTs)[lane] = w.cell[wb_ci(x)]; (a.symtab + b * capTs)[lane] = w.symTab[wb_si(x)]; }
    } else if (rev) {
        const u32 rs = 32u - tl;                                           // tl >= FSE_MIN_TABLELOG = 5
        for (u32 i = lane; i < ts / 2; i += 64)
            A32[i] = (u32)w.cell[wb_ci(__brev(2u * i) >> rs)] | ((u32)w.cell[wb_ci(__brev(2u * i + 1u) >> rs)] << 16);
        for (u32 i = lane; i < ts / 4; i += 64) {
            u32 y = 0;
            for (u32 k = 0; k < 4; ++k) y |= (u32)w.symTab[wb_si(__brev(4u * i + k) >> rs)] << (8u * k);
            S32[i] = y;
        }
    } else {
        for (u32 i = lane; i < ts / 2; i += 64) A32[i] = *(const u32*)(w.cell + wb_ci(2u * i));
        for (u32 i = lane; i < ts / 4; i += 64) S32[i] = *(const u32*)(w.symTab + wb_si(4u * i));
    }
    if (lane == 0) a.meta[b].state = m.state | (fast ? 2u : 0u);
}

// FSE_buildDTable over a batch: what k_fse_dparse / k_fse_dbuild left in the decoder's own formats, written out in the reference's
// layout (lib/fse.h:565-575: {U16 tableLog; U16 fastMode} then {U16 newState; BYTE symbol; BYTE nbBits} per state), coalesced
__global__ __launch_bounds__(256) void k_fse_export_dtable(FseDPrepArgs a, u32 capTs, u32* dtables, size_t dtStrideU32)
{
    const size_t b = blockIdx.x;
    const FseMeta m = a.meta[b];
    if (m.state == 0) return;                                              // (its error is in results[b])
    const u32 tl = m.tableLog, ts = 1u << tl;
    const bool rev = ((m.state >> 2) & 3u) != FSE_DCLS_PLAIN;
    const u16* const A = a.atab + b * capTs;
    const u8* const S = a.symtab + b * capTs;
    u32* const dt = dtables + b * dtStrideU32;
    if (threadIdx.x == 0) dt[0] = tl | ((m.state & 2u) ? 1u << 16 : 0u);
    for (u32 x = threadIdx.x; x < ts; x += blockDim.x) {
        const u32 i = rev ? __brev(x) >> (32u - tl) : x;
        const u32 c = A[i];
        const u32 nb = rev ? (c & 31u) : (c >> 12);
        const u32 ns = rev ? __brev(c >> 5) >> (32u - tl) : (c & 0xFFFu);
        dt[1 + x] = ns | ((u32)S[i] << 16) | (nb << 24);
    }
}
hipError_t launch_fse_export_dtables(const FseDPrepArgs& a, u32* dtables, size_t dtStrideU32, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fse_export_dtable, dim3((unsigned)a.nBlocks), dim3(256), 0, s, a, 1u << a.maxLog, dtables, dtStrideU32);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
//  FSE_buildCTable / FSE_buildDTable on normalised counters the CALLER supplies (lib/fse.h:162-163 / :240-241): the table builders as calls of
//  their own, beside the forms that start from the blocks (FSE_buildCTable_batch) or from the headers (FSE_buildDTable_batch).  One wave per
//  table.  What the reference leaves undefined is refused: counters that do not describe a table of 1 << tableLog cells (it asserts in the
//  CTable builder, lib/fse_compress.c:127, and returns GENERIC from the DTable builder, lib/fse_decompress.c:107), counters below -1, tableLog 0,
//  and the table logs 1 and 3, whose FSE_TABLESTEP (lib/fse.h:683) is even: the reference's spread then writes cell 0 over and over and leaves the
//  other cells as it found them (every table log from FSE_MIN_TABLELOG = 5 up, and 2 and 4, has an odd step).
// ---------------------------------------------------------------------------------------------------
DEV size_t fse_norm_load_checked(int nn[4], const s16* nb, u32 maxSV, u32 tl, u32 maxTl, u32 lane)
{
    if (maxSV > 255u) return FERR(maxSymbolValue_tooLarge);                // lib/fse_decompress.c:83 (FSE_MAX_SYMBOL_VALUE)
    if (tl > maxTl) return FERR(tableLog_tooLarge);                        // :84; the CTable builder's workspace check, lib/fse_compress.c:86
    if (tl == 0 || tl == 1 || tl == 3) return FERR(GENERIC);               // FSE_TABLESTEP(2) = 4, FSE_TABLESTEP(8) = 8: the reference's spread never leaves cell 0 there
    u32 cells = 0; bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        nn[i] = 4 * lane + i <= maxSV ? (int)nb[4 * lane + i] : 0;
        bad |= nn[i] < -1;
        cells += nn[i] == -1 ? 1u : nn[i] > 0 ? (u32)nn[i] : 0u;
    }
    cells = wg_sum<64>(cells);
    return (wg_any<64>(bad, lane) || cells != (1u << tl)) ? FERR(GENERIC) : 0;
}
__global__ __launch_bounds__(64) void k_fse_ctable_from_norm(const s16* norms, size_t normStride, const u32* maxSVs, u32 tl, u32* ctables, size_t ctStrideU32,
                                                            size_t* results, u32 capTs)
{
    extern __shared__ __attribute__((aligned(16))) u8 wbLds[];
    const size_t b = blockIdx.x;
    const u32 lane = threadIdx.x;
    const u32 maxSV = maxSVs[b];
    int nn[4] = { 0, 0, 0, 0 };
    const size_t r = fse_norm_load_checked(nn, norms + b * normStride, maxSV, tl, FSE_MAX_TL, lane);
    if (lane == 0) results[b] = r;
    if (is_err(r)) return;                                                 // uniform
    fse_wave_build_ctable(wave_build_carve(wbLds, capTs), nn, maxSV, tl, ctables + b * ctStrideU32, lane);
}
hipError_t launch_fse_ctable_from_norm(const s16* norms, size_t normStride, const u32* maxSVs, u32 tl, u32* ctables, size_t ctStrideU32, size_t* results,
                                       size_t nBlocks, hipStream_t s)
{
    if (nBlocks == 0) return hipSuccess;
    const u32 capTs = 1u << (tl >= 1 && tl <= FSE_MAX_TL ? tl : FSE_MAX_TL);
    hipLaunchKernelGGL(k_fse_ctable_from_norm, dim3((unsigned)nBlocks), dim3(64), wave_build_lds_bytes(capTs), s, norms, normStride, maxSVs, tl, ctables, ctStrideU32,
                       results, capTs);
    return hipGetLastError();
}

// the counters -> what k_fse_dparse leaves behind for k_fse_dbuild (meta, counters in scratch, the block filed under its decoder class)
__global__ __launch_bounds__(64) void k_fse_dmeta_from_norm(FseDPrepArgs a, const s16* norms, size_t normStride, const u32* maxSVs, u32 tl)
{
    const size_t b = blockIdx.x;
    const u32 lane = threadIdx.x;
    const u32 maxSV = maxSVs[b];
    int nn[4] = { 0, 0, 0, 0 };
    const size_t r = fse_norm_load_checked(nn, norms + b * normStride, maxSV, tl, a.maxLog, lane);
    FseMeta m; m.state = 0; m.hdrSize = 0; m.tableLog = tl; m.maxSV = maxSV; m.pace = 0;
    if (!is_err(r)) {
        *(uint2*)(a.norms + b * 256 + 4 * lane) = make_uint2(((u32)nn[0] & 0xFFFFu) | ((u32)nn[1] << 16), ((u32)nn[2] & 0xFFFFu) | ((u32)nn[3] << 16));
        int top = nn[0] > nn[1] ? nn[0] : nn[1]; top = nn[2] > top ? nn[2] : top; top = nn[3] > top ? nn[3] : top;
        top = wave_max_i32(top);
        const u32 cls = tl <= FSE_DEC_FAST_MAXLOG ? (u32)FSE_DCLS_REV11 : 2 * top > (1 << tl) ? (u32)FSE_DCLS_PLAIN : (u32)FSE_DCLS_REV12;   // as k_fse_dparse
        m.state = 1u | (cls << 2);
        if (lane == 0) {
            const u32 at = atomicAdd(&a.counts[cls * FSE_DBINS], 1u);      // (size bin 0: there is no stream)
            atomicAdd(&a.counts[FSE_DCLS_COUNT + cls], 1u);
            a.lists[(size_t)cls * FSE_DBINS * a.nBlocks + at] = (u32)b;
        }
    }
    if (lane == 0) { a.meta[b] = m; if (is_err(r)) a.results[b] = r; }
}
hipError_t launch_fse_dprep_from_norm(const FseDPrepArgs& a, const s16* norms, size_t normStride, const u32* maxSVs, u32 tl, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    const u32 capTs = 1u << a.maxLog;
    hipError_t e = launch_zero_u32(a.counts, FSE_DCLS_COUNT + FSE_DCLS_KINDS, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fse_dmeta_from_norm, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a, norms, normStride, maxSVs, tl);
    const u32 capA = a.maxLog < FSE_DEC_FAST_MAXLOG ? capTs : (1u << FSE_DEC_FAST_MAXLOG);
    hipLaunchKernelGGL(k_fse_dbuild, dim3((unsigned)a.nBlocks), dim3(64), wave_build_lds_bytes(capA), s, a, capTs, (int)FSE_DCLS_REV11 * FSE_DBINS, (int)FSE_DBINS, capA);
    if (a.maxLog > FSE_DEC_FAST_MAXLOG)
        hipLaunchKernelGGL(k_fse_dbuild, dim3((unsigned)a.nBlocks), dim3(64), wave_build_lds_bytes(capTs), s, a, capTs, (int)FSE_DCLS_REV12 * FSE_DBINS, 2 * (int)FSE_DBINS, capTs);
    return hipGetLastError();
}

#ifdef FSE_WB_TIMING
extern "C" __attribute__((visibility("default"))) int FSEHIP_debug_wbTiming(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wbTiming), sizeof(g_wbTiming)); }
#endif
// ---------------------------------------------------------------------------------------------------
//  The glue steps as calls of their own (fsehip.h "Table glue, step by step"): the SAME wave routines k_fse_cprep / k_fse_dparse run
//  (wg_normalize<64>, wg_write_ncount<64>, ncount_read), on caller-supplied counters / headers, so that the reference's own unit vectors for
//  FSE_normalizeCount / FSE_writeNCount / FSE_readNCount (programs/fuzzer.c:325-417) can be put to the device exactly as written.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_fse_glue_normalize(s16* norms, size_t normStride, u32 tlReq, const u32* counts, size_t countStride,
                                                          const size_t* totals, const u32* maxSVs, size_t* results)
{
    const size_t b = blockIdx.x;
    const u32 lane = threadIdx.x;
    const u32 maxSV = maxSVs[b] > 255u ? 255u : maxSVs[b];
    const u32* const cb = counts + b * countStride;
    u32 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = 4 * lane + i <= maxSV ? cb[4 * lane + i] : 0u;
    int nn[4] = { 0, 0, 0, 0 };
    const u32 tl = tlReq ? tlReq : FSE_DEF_TL;                                 // lib/fse_compress.c:434
    const u64 total = (u64)totals[b];
    size_t r = total ? wg_normalize<64>(nn, c, total, maxSV, tl, lane) : FERR(GENERIC);
    if (!is_err(r)) r = tl;
    s16* const nb = norms + b * normStride;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (4 * lane + i <= maxSV && !is_err(r)) nb[4 * lane + i] = (s16)nn[i];
    if (lane == 0) results[b] = r;
}

__global__ __launch_bounds__(64) void k_fse_glue_write_ncount(u8* headers, size_t headerStride, size_t headerCapacity, const s16* norms, size_t normStride,
                                                             const u32* maxSVs, u32 tl, size_t* results)
{
    __shared__ u32 img[136];
    const size_t b = blockIdx.x;
    const u32 lane = threadIdx.x;
    const u32 maxSV = maxSVs[b];
    size_t r;
    if (tl > FSE_MAX_TL) r = FERR(tableLog_tooLarge);                            // lib/fse_compress.c:281-282
    else if (tl < FSE_MIN_TL || maxSV > 255u) r = FERR(GENERIC);
    else {
        const s16* const nb = norms + b * normStride;
        int nn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) nn[i] = 4 * lane + i <= maxSV ? (int)nb[4 * lane + i] : 0;
        for (u32 i = lane; i < 136; i += 64) img[i] = 0;
        __syncthreads();
        r = wg_write_ncount<64>(img, headerCapacity, nn, maxSV, tl, lane);
        __syncthreads();
        if (!is_err(r)) {
            u8* const dst = headers + b * headerStride;
            const u8* const hb = (const u8*)img;
            for (u32 i = lane; i < (u32)r; i += 64) dst[i] = hb[i];
        }
    }
    if (lane == 0) results[b] = r;
}

__global__ __launch_bounds__(64) void k_fse_glue_read_ncount(s16* norms, size_t normStride, u32* maxSVs, u32* tableLogs, BlockView headers, size_t nBlocks, size_t* results)
{
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= nBlocks) return;
    u32 maxSV = maxSVs[b], tl = 0;
    size_t r;
    if ((size_t)maxSV >= normStride) r = FERR(maxSymbolValue_tooLarge);       // (the reference trusts its caller's array; a batch call can check)
    else r = ncount_read<1>(norms + b * normStride, &maxSV, &tl, view_ptr(headers, b), view_size(headers, b));
    if (!is_err(r)) { maxSVs[b] = maxSV; tableLogs[b] = tl; }
    results[b] = r;
}

hipError_t launch_fse_glue_normalize(s16* norms, size_t normStride, u32 tl, const u32* counts, size_t countStride, const size_t* totals, const u32* maxSVs,
                                     size_t* results, size_t nBlocks, hipStream_t s)
{
    if (nBlocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fse_glue_normalize, dim3((unsigned)nBlocks), dim3(64), 0, s, norms, normStride, tl, counts, countStride, totals, maxSVs, results);
    return hipGetLastError();
}
hipError_t launch_fse_glue_write_ncount(u8* headers, size_t headerStride, size_t headerCapacity, const s16* norms, size_t normStride, const u32* maxSVs, u32 tl,
                                        size_t* results, size_t nBlocks, hipStream_t s)
{
    if (nBlocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fse_glue_write_ncount, dim3((unsigned)nBlocks), dim3(64), 0, s, headers, headerStride, headerCapacity, norms, normStride, maxSVs, tl, results);
    return hipGetLastError();
}
hipError_t launch_fse_glue_read_ncount(s16* norms, size_t normStride, u32* maxSVs, u32* tableLogs, const BlockView& headers, size_t* results, size_t nBlocks, hipStream_t s)
{
    if (nBlocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fse_glue_read_ncount, dim3((unsigned)((nBlocks + 63) / 64)), dim3(64), 0, s, norms, normStride, maxSVs, tableLogs, headers, nBlocks, results);
    return hipGetLastError();
}

hipError_t launch_fse_cprep(const FseCPrepArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    const u32 capTs = 1u << a.maxTl;
    const size_t ldsBytes = wave_build_lds_bytes(capTs);
    probe_before(PK_FSE_CPREP, s);
    hipLaunchKernelGGL(k_fse_cprep, dim3((unsigned)a.nBlocks), dim3(64), ldsBytes, s, a, capTs);
    probe_after(PK_FSE_CPREP, s);
    return hipGetLastError();
}
hipError_t launch_fse_dprep(const FseDPrepArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    const u32 capTs = 1u << a.maxLog;
    hipError_t e = launch_zero_u32(a.counts, FSE_DCLS_COUNT + FSE_DCLS_KINDS, s);      // the lists' lengths and the classes' totals
    if (e != hipSuccess) return e;
    probe_before(PK_FSE_DPREP, s);
    hipLaunchKernelGGL(k_fse_dparse, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a);
    // the builds are latency-bound single-wave workgroups: their LDS footprint (sized by the largest table of the launch)
    // decides how many run per CU, so the tableLog <= 11 class gets a launch of its own
    const u32 capA = a.maxLog < FSE_DEC_FAST_MAXLOG ? capTs : (1u << FSE_DEC_FAST_MAXLOG);
    hipLaunchKernelGGL(k_fse_dbuild, dim3((unsigned)a.nBlocks), dim3(64), wave_build_lds_bytes(capA), s, a, capTs, (int)FSE_DCLS_REV11 * FSE_DBINS, (int)FSE_DBINS, capA);
    if (a.maxLog > FSE_DEC_FAST_MAXLOG)
        hipLaunchKernelGGL(k_fse_dbuild, dim3((unsigned)a.nBlocks), dim3(64), wave_build_lds_bytes(capTs), s, a, capTs, (int)FSE_DCLS_REV12 * FSE_DBINS, 2 * (int)FSE_DBINS, capTs);
    probe_after(PK_FSE_DPREP, s);
    return hipGetLastError();
}
