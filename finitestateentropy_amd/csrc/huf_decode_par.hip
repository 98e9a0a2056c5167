// huf_decode_par.hip -- a5: HUF_decompress4X1 with every stream split across the 64 lanes of a wave (one wave per block)
// (reference: lib/huf_decompress.c:194-354; bit reader lib/bitstream.h:272-448; format SURVEY A.6).
//
// A prefix code is self-synchronising: a decoder dropped at an arbitrary bit of the stream falls into step with the true
// codeword boundaries after a few symbols.  That makes a Huffman stream splittable without changing a byte of the result --
// the same speculate / verify / repair idea as the tANS encoder (fse_encode_wave.hip), here on bit positions:
//
//   The four streams of a block are taken one after the other; lane j walks the j-th 64th (by bit position) of the stream.
//   Positions are counted as C = bits consumed from the top of the stream (the reference reads it from its last byte down).
//   Pass 1: lane 0 starts at the true first bit; lane j > 0 warms up from 128 bits in front of its range and takes the first
//     codeword boundary inside the range as its start S_j (its guess at where lane j-1 will end); it then decodes to the first
//     boundary at or beyond the end of its range, E_j, counting its symbols n_j.
//   Verify / repair: S_j must equal E_(j-1).  Lane 0 is exact, so if every link holds all lanes are exact by induction.  A lane
//     whose link fails (its warm-up had not fallen into step yet) re-runs its range from E_(j-1); the check repeats until no
//     link changes (worst case: 64 rounds = the serial walk).
//   The stream is accepted only if it regenerates exactly its segment and ends exactly on its first bit (BIT_endOfDStream,
//   lib/huf_decompress.c:348-349).  Otherwise -- corrupt input, tiny or irregular blocks, streams beyond the LDS budget -- the
//   whole block is handed to the serial decoder (k_huf_decode), which reproduces the reference's verdict literally.
//   Pass 2: a 64-lane prefix sum of n_j gives every lane its place in the output; it decodes its symbols again and stores them
//     sixteen at a time.
//
// The kernel is latency-bound (a dependent LDS look-up per symbol), so what counts is waves per CU = LDS per wave: the table slot
// plus ONE staged stream.  k_huf_dprep reads the jump table and files every block under the smallest of three per-stream budgets
// that holds its longest stream (internal.h: 2.3 / 4.6 / 8.4 KB -> 25 / 18 / 12 waves per CU with 4 KiB tables), one launch each;
// measured per 100k 32 KB blocks: P80 3.46 -> 2.7 ms, P14 3.4 -> 3.0 ms, P02 (incompressible, 7 KB streams) 3.65 unchanged.
//   (Dead end: a workgroup of four waves, one per stream, sharing the table -- 28 waves per CU with the small budget -- P80 2.6 ms
//   but P14 3.8 and P02 5.0: the workgroup lives as long as its slowest stream and the LDS pipe is already half busy.)
//
// 64 busy lanes per block instead of 4 and no service waves.  In LDS: the table (bit-reversed {nbBits, byte} cells as in
// huf_decode.hip) and the current stream, staged with coalesced loads IN CONSUMPTION ORDER (dword m = bit-reversed stream dword
// top - m), so that the decoding loop is the serial kernel's: a cursor Q = consumed bits - 1, bits taken from the low end of a
// three-dword register window, four symbols per iteration with one table look-up each on the dependent chain (hd_bulk_phase).
// Near the ends of a range the lane steps symbol by symbol so that "first boundary at or beyond" is exact.
//   (Measured dead ends: every lane refilling its own window from global memory, 20 ms per 100k blocks -- a dependent,
//   uncoalesced load behind nearly every symbol; all four streams staged at once with 16 lanes each, 10 ms -- 24 KB of LDS per
//   block leaves six waves per CU; the serial kernel: 8.0 ms.)
#include "internal.h"

#ifndef HPAR_WARM
#define HPAR_WARM 128u                // warm-up bits in front of a range
#endif
#define HPAR_MAX_REPAIR 8u            // repair rounds per stream before the block is handed to the serial decoder
#define HPAR_NEAR 48u                 // four symbols consume at most 48 bits: closer than this to a limit the lane steps by symbols

typedef const __attribute__((address_space(3))) u16* hpar_lds_u16;
typedef const __attribute__((address_space(3))) u32* hpar_lds_u32;
// (A/B, round 4: the cell through an aligned 8-byte read -- 64 LDS banks instead of the 32 that 2- and 4-byte reads use -- and a 64-bit
//  shift: 3.42 instead of 3.20 ms per 100k P14 blocks in the same run, 5.29 instead of 4.87 on P02: the two extra dependent operations
//  per symbol cost more than the conflicts they avoid.  Not kept.)
DEV u32 hpar_cell(u32 win, u32 mask2, u32 tabOff) { return *(hpar_lds_u16)(uintptr_t)((win & mask2) | tabOff); }

// one symbol at cursor C (consumed bits); returns the cell (nbBits | byte << 8)
DEV u32 hpar_single(u32 arr, u32& C, u32 tabOff, u32 mask2)
{
    const u32 Q = C - 1u;
    const hpar_lds_u32 wp = (hpar_lds_u32)(uintptr_t)(arr + ((Q >> 5) << 2));
    const u32 lo = __builtin_amdgcn_alignbit(wp[1], wp[0], Q & 31u);      // bit 1 = next unread bit
    const u32 c = hpar_cell(lo, mask2, tabOff);
    C += c & 0xFFu;
    return c;
}

// Four symbols per iteration on the flat consumption-order array.  The kernel is bound by instruction issue (EXPERIMENTS.md section 3, round 6), so the
// loop is laid out for instruction count: the cursor is ONE number, QA = (consumed bits - 1) + 8 * (LDS byte address of the array) -- the
// bit address of the bit below the next unread one -- and every iteration reads its three window dwords afresh at (QA >> 3) & ~3 (two
// address instructions, two LDS reads, two v_alignbit) instead of sliding a register window by selects (round 5: fourteen instructions).
// Per symbol: v_and (cell address), ds_read_u16, v_alignbit + v_lshrrev (drop the code's bits), v_perm (collect the byte).
struct HparBulk {
    u32 QA, bias;
    DEV void open(u32 arr, u32 C) { bias = 8u * arr - 1u; QA = C + bias; }
    DEV u32 cursor() const { return QA - bias; }
    DEV u32 q_of(u32 C) const { return C + bias; }
    DEV u32 iter(u32 tabOff, u32 mask2)                                   // returns the four symbols, first in the low byte
    {
        const hpar_lds_u32 wp = (hpar_lds_u32)(uintptr_t)((QA >> 3) & ~3u);
        const u32 w0 = wp[0], w1 = wp[1], w2 = wp[2];
        u32 l = __builtin_amdgcn_alignbit(w1, w0, QA), h = __builtin_amdgcn_alignbit(w2, w1, QA);   // (the shift is QA & 31: 8 * arr is a multiple of 32)
        const u32 c1 = hpar_cell(l, mask2, tabOff);
        l = __builtin_amdgcn_alignbit(h, l, c1); h >>= (c1 & 31u);
        const u32 c2 = hpar_cell(l, mask2, tabOff);
        l = __builtin_amdgcn_alignbit(h, l, c2); h >>= (c2 & 31u);
        const u32 c3 = hpar_cell(l, mask2, tabOff);
        l = __builtin_amdgcn_alignbit(h, l, c3);
        const u32 c4 = hpar_cell(l, mask2, tabOff);
        u32 word = __builtin_amdgcn_perm(c1, 0u, 0x03020105u);
        word = __builtin_amdgcn_perm(c2, word, 0x03020500u);
        word = __builtin_amdgcn_perm(c3, word, 0x03050100u);
        word = __builtin_amdgcn_perm(c4, word, 0x05020100u);
        QA += (c1 + c2 + c3 + c4) & 0xFFu;
        return word;
    }
};

// decode from C up to the first boundary at or beyond `limit`, counting symbols: four at a time until an iteration crosses the limit; that
// iteration is taken back and redone symbol by symbol
DEV u32 hpar_run(u32 arr, u32& C, u32 limit, u32 tabOff, u32 mask2)
{
    u32 n = 0;
    if (C < limit) {
        HparBulk bk; bk.open(arr, C);
        const u32 limQ = bk.q_of(limit);
        u32 Qh;
        do { Qh = bk.QA; (void)bk.iter(tabOff, mask2); n += 4; } while (bk.QA < limQ);
        n -= 4; C = Qh - bk.bias;
        while (C < limit) { (void)hpar_single(arr, C, tabOff, mask2); ++n; }   // one to four
    }
    return n;
}

// The same walk KEEPING its symbols: iteration k's four bytes stay in register sym[k] (the loop is fully unrolled so that every index is
// static; all lanes step together, a lane whose range is done sits out).  The iteration that crosses the limit is kept whole -- its first
// symbols are the range's last -- and redone symbol by symbol for the exact boundary and count.  A lane that needs more than HPAR_KEEP
// iterations (a stretch of very short codes) counts on without keeping and reports `spill`: the piece then takes the re-decoding pass 2.
#ifndef HPAR_KEEP
#define HPAR_KEEP 40                  // iterations kept = 160 symbols per lane and piece (a lane's share of a stream averages 128; 48 registers cost the
#endif                                //  4.5 KiB class two of its 18 waves per CU: 2.91 -> 2.74 ms per 100k P14 blocks with 40 and five waves per SIMD)
DEV u32 hpar_run_keep(u32 arr, u32& C, u32 limit, u32 tabOff, u32 mask2, u32 (&sym)[HPAR_KEEP], bool& spill)
{
    u32 n = 0;
    spill = false;
    bool active = C < limit;
    if (__any(active)) {
        HparBulk bk; bk.open(arr, C);
        const u32 limQ = bk.q_of(limit);
        u32 Qh = bk.QA;
#pragma unroll
        for (int k = 0; k < HPAR_KEEP; ++k) {
            if (active) {
                Qh = bk.QA;
                sym[k] = bk.iter(tabOff, mask2);
                n = 4u * (u32)k;                                           // symbols in front of the iteration taken last
                active = bk.QA < limQ;
            }
            if (!__any(active)) break;                                     // uniform
        }
        C = bk.cursor();
        if (active) { spill = true; n += 4u + hpar_run(arr, C, limit, tabOff, mask2); }   // (rare)
        else if (C >= limit && bk.QA != Qh) {                              // (took at least one iteration)
            C = Qh - bk.bias;
            while (C < limit) { (void)hpar_single(arr, C, tabOff, mask2); ++n; }     // one to four
        }
    }
    return n;
}

// pass 2 from the registers: lane's n symbols to p -- single bytes up to a 4-byte boundary, then the register stream shifted by that many
// bytes (v_alignbyte) in 16-byte stores, the last 1..15 bytes as dwords and bytes.  Static register indices throughout (unrolled).
__device__ __attribute__((always_inline)) void hpar_store_kept(u8* p, u32 n, const u32 (&sym)[HPAR_KEEP])
{
    u32 a = (0u - (u32)(uintptr_t)p) & 3u;
    a = a < n ? a : n;
    {   const u32 w = sym[0];
        if (a > 0) p[0] = (u8)w;
        if (a > 1) p[1] = (u8)(w >> 8);
        if (a > 2) p[2] = (u8)(w >> 16); }
    p += a;
    u32 left = n - a;
#pragma unroll
    for (int g = 0; g < HPAR_KEEP / 4; ++g) {
        if (left == 0) continue;
        u32 w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = 4 * g + t;
            const u32 hi = k + 1 < HPAR_KEEP ? sym[k + 1 < HPAR_KEEP ? k + 1 : k] : 0u;
            w[t] = __builtin_amdgcn_alignbyte(hi, sym[k], a);
        }
        if (left >= 16) { __builtin_memcpy(p, w, 16); p += 16; left -= 16; }
        else {
            if (left >= 4) __builtin_memcpy(p, &w[0], 4);
            if (left >= 8) __builtin_memcpy(p + 4, &w[1], 4);
            if (left >= 12) __builtin_memcpy(p + 8, &w[2], 4);
            const u32 r = left >= 12 ? w[3] : left >= 8 ? w[2] : left >= 4 ? w[1] : w[0];
            u8* const t = p + (left & ~3u);
            const u32 m = left & 3u;
            if (m > 0) t[0] = (u8)r;
            if (m > 1) t[1] = (u8)(r >> 8);
            if (m > 2) t[2] = (u8)(r >> 16);
            left = 0;
        }
    }
}

// pass 2 through an LDS line buffer (the stream's staging area, idle once its links are verified): the piece's output is taken in chunks of
// CH bytes of the 16-byte grid of the DESTINATION (logical byte x of the piece <-> global address gBase + x, gBase 16-byte aligned, the
// piece's first byte at x = h).  A lane's run is [s, s + n): `a` single bytes up to the dword grid, then its register stream shifted by `a`
// bytes (v_alignbyte) as dwords D0 + k, D0 = (s + a) / 4.  Per chunk, every lane whose first dword lies in the chunk or at most HPAR_KEEP
// dwords in front of it drops ALL its words into the buffer, last word first, without asking which of them are real: a word beyond the
// lane's last falls into the territory of a LATER lane at one of that lane's EARLIER words (runs follow one another in lane order), and
// going downwards that lane writes there afterwards -- so the real word always lands last.  The buffer has HPAR_KEEP dwords of slack on
// either side for what sticks out of the chunk.  Two instructions per word (v_alignbyte, ds_write_b32 at a static offset), no predicate.
// Then the 1..3 bytes in front of each lane's first dword (behind all words: a neighbour's last word carries garbage there), and the wave
// copies the chunk out with one aligned 16-byte store per lane, 1 KiB per instruction: every line of the destination is written once and
// whole (the first and last 16-byte unit of a piece may be partial: those go byte by byte).
// The workgroup is ONE wave: its LDS operations execute in order, so "everything before is visible to every lane" needs no barrier and, above
// all, no wait for the wave's global stores (which __syncthreads' workgroup-scope fence brings along).  A compiler-level fence keeps the order.
DEV void hpar_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
typedef __attribute__((address_space(3))) u32* hpar_lds_w32;
typedef __attribute__((address_space(3))) u8* hpar_lds_w8;
#define HPAR_FLUSH_SLACK (4u * HPAR_KEEP)
template <u32 CH>
DEV void hpar_flush_kept(u32 bufOff, u8* gBase, u32 h, u32 total, u32 s, u32 n, const u32 (&sym)[HPAR_KEEP], u32 lane)
{
    const u32 end = h + total;                                            // logical end of the piece's output
    const u32 a0 = (0u - s) & 3u, a = a0 < n ? a0 : n;                    // bytes in front of my first whole dword
    const u32 D0 = (s + a) >> 2;
    const u32 nW = (n - a + 3u) >> 2;                                     // my dwords (the last may carry up to 3 bytes of garbage)
    const u32 nWmax = __builtin_amdgcn_readfirstlane(wave_max_u32(nW));
    const u32 chunk = bufOff + HPAR_FLUSH_SLACK;                          // LDS address of the chunk's first byte
    for (u32 c0 = 0; c0 < end; c0 += CH) {                                // uniform
        hpar_wave_sync();                                                 // (the buffer's previous content is done with)
        const int rel = (int)D0 - (int)(c0 >> 2);                         // my first dword relative to the chunk
        if (nW != 0 && rel >= -(int)HPAR_KEEP && rel < (int)(CH >> 2)) {     // (a lane without a whole dword has no territory: its D0 lies inside a neighbour's)
            const u32 wordAddr = chunk + 4u * (u32)rel;                   // (>= bufOff)
#pragma unroll
            for (int k = HPAR_KEEP - 1; k >= 0; --k) {
                if ((u32)k < nWmax) {                                     // uniform
                    const u32 hi = sym[k + 1 < HPAR_KEEP ? k + 1 : k];
                    *(hpar_lds_w32)(uintptr_t)(wordAddr + 4u * (u32)k) = __builtin_amdgcn_alignbyte(hi, sym[k], a);
                }
            }
        }
        hpar_wave_sync();
        {   const u32 w = sym[0];
#pragma unroll
            for (u32 i = 0; i < 3; ++i) {
                const u32 x = s + i;
                if (i < a && x - c0 < CH) *(hpar_lds_w8)(uintptr_t)(chunk + (x - c0)) = (u8)(w >> (8u * i));
            }
        }
        hpar_wave_sync();
        // copy out: unit = 16 bytes at logical c0 + 16 u
#pragma unroll
        for (u32 r = 0; r < CH / 1024u; ++r) {
            const u32 x = c0 + 1024u * r + 16u * lane;
            if (x + 16u <= h || x >= end) continue;
            typedef u32 hpar_v4 __attribute__((ext_vector_type(4)));
            const hpar_v4 v = *(const __attribute__((address_space(3))) hpar_v4*)(uintptr_t)(chunk + (x - c0));
#ifdef HPAR_ABL_NOGSTORE   // measurement aid: the line buffer is filled and read, the global stores are left out (results wrong)
            if (v.x == 0x12345u && v.y == 0x777u) gBase[x] = 0;
            continue;
#endif
            if (x >= h && x + 16u <= end) *(hpar_v4*)(gBase + x) = v;
            else {
                const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                for (u32 i = 0; i < 16; ++i) if (x + i >= h && x + i < end) gBase[x + i] = (u8)(w[i >> 2] >> (8u * (i & 3u)));
            }
        }
    }
}

// ---- staging of (a piece of) a stream: consumption order -- array dword m = bit-reversed stream dword dTop - m, zeros below the stream's first
// bit -- in units of four dwords.  Two steps so that a stream's loads can be issued EARLY (before the stores of the stream in front of it:
// loads and stores share one in-order counter, a load issued behind a burst of stores is not seen before the stores are acknowledged):
//   hpar_stage_load   : one 16-byte load per unit into registers, every load of a lane issued before anything waits; the units that touch either
//                       end of the stream (dwords beyond its last byte, the partial top dword: one or two lanes) are put together afterwards;
//   hpar_stage_commit : bit reversal and the LDS stores.
template <u32 MAXU>
DEV void hpar_stage_load(uint4 (&buf)[MAXU], const u8* sp, u32 L, u32 Sd, u32 mTop, u32 nd, u32 lane)
{
    const int dTop = (int)Sd - 1 - (int)mTop;
    const u32 units = (nd + 3) / 4 + 4;                                   // + 16 dwords of slack (the stream goes on there, or zeros)
    u32 edge = 0;
#pragma unroll
    for (u32 t = 0; t < MAXU; ++t) {
        const u32 u = lane + 64 * t;
        buf[t] = make_uint4(0, 0, 0, 0);
        if (u >= units) continue;
        const int dLo = dTop - 3 - 4 * (int)u;                            // lowest stream dword of the unit
        if (dLo >= 0 && 4 * ((u32)dLo + 4) <= L) __builtin_memcpy(&buf[t], sp + 4 * dLo, 16);
        else edge |= 1u << t;
    }
    if (__any(edge != 0)) {
#pragma unroll
        for (u32 t = 0; t < MAXU; ++t) {
            if (!((edge >> t) & 1u)) continue;
            const int dLo = dTop - 3 - 4 * (int)(lane + 64 * t);
            u32 w[4] = { 0, 0, 0, 0 };
            for (int i = 0; i < 4; ++i) {
                const int d = dLo + i;
                if (d < 0 || d >= (int)Sd) continue;
                if (4 * (u32)d + 4 <= L) __builtin_memcpy(&w[i], sp + 4 * d, 4);
                else for (u32 b8 = 4 * (u32)d; b8 < L; ++b8) w[i] |= (u32)sp[b8] << (8 * (b8 & 3u));
            }
            buf[t] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}
template <u32 MAXU>
DEV void hpar_stage_commit(const uint4 (&buf)[MAXU], u32* data, u32 nd, u32 lane)
{
    const u32 units = (nd + 3) / 4 + 4;
#pragma unroll
    for (u32 t = 0; t < MAXU; ++t) {
        const u32 u = lane + 64 * t;
        if (u < units) ((uint4*)data)[u] = make_uint4(__brev(buf[t].w), __brev(buf[t].z), __brev(buf[t].y), __brev(buf[t].x));
    }
}

#ifdef HPAR_STATS            // development aid: repair rounds / bad links / cycles of the first blocks (scripts/hparstats.py)
__device__ unsigned long long g_hparStats[4096 * 8];
extern "C" __attribute__((visibility("default"))) int FSEHIP_debug_hparStats(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hparStats), sizeof(g_hparStats)); }
#define HST(...) __VA_ARGS__
#else
#define HST(...)
#endif
// X2CAP: the instantiation that can also take double-symbol tables (it holds a whole table in registers while deriving the single-symbol
// cells: 200 VGPRs).  The one-shot path and single-symbol caller tables use the lean instantiation.
// (register budget: the kept symbols must not cost residency -- five waves per SIMD (96 registers: 18 waves per CU by LDS) for the 4.5 KiB budget,
//  three for the large one (12 by LDS); measured with the large budget for every block, i.e. 12 waves per CU and one output pass: P14 3.32 ms
//  against 2.91 -- the kernel lives on waves per CU)
template <u32 DATA, bool X2CAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(X2CAP ? 2 : DATA >= 8192u ? 3 : 5))) void k_huf_decode_par(HufDecArgs a, u32* fbList, u32* fbCount)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 lane = threadIdx.x;
    const size_t slot = blockIdx.x;
    const size_t nTot = a.count ? (size_t)*a.count : a.nBlocks;
    if (slot >= nTot) return;                                            // uniform
    const size_t b = a.list ? (size_t)__builtin_amdgcn_readfirstlane(a.list[slot]) : slot;
    if (a.meta && a.meta[b].state == 0) return;                          // (finished by the prepare kernel)
    if (a.onlyDeclined && a.results[b] != HUF_DECLINED) return;          // (second launch of the caller-table path: what the lean one left)
    const u32 hdr = __builtin_amdgcn_readfirstlane(a.meta ? a.meta[b].hdrSize : 0u);                      // (caller-built tables: the payload starts the block)
    const u32* const gt = a.dtables + b * a.dtStrideU32;
    const u32 desc = __builtin_amdgcn_readfirstlane(gt[0]);
    const u32 dtLog = (desc >> 16) & 0xFFu;
    const u8* const in = view_ptr(a.csrc, b) + hdr;
    const size_t cSize = view_size(a.csrc, b) - hdr;
    const size_t dstSize = view_size(a.dstSizes, b);
    u8* const out = a.dst + b * a.dstStride;

    // ---- can this block take the parallel path?  (uniform)  Anything unusual is the serial decoder's business.
    const u32 tableType = (desc >> 8) & 0xFFu;                           // 0: single-symbol cells (X1); 1: double-symbol cells (X2), caller-built only
    if (X2CAP && tableType != 1) return;                                 // (single-symbol tables were the lean launch's)
    bool ok = (tableType == 0 || (X2CAP && tableType == 1 && a.acceptX2 && !a.meta)) && dtLog >= 1 && dtLog <= a.ldsLog && dtLog <= a.maxTableLog
              && cSize >= 10 && cSize < (1u << 28) && dstSize >= 64 && dstSize < (1u << 28);
    u32 len[4] = { 0, 0, 0, 0 }, T0[4] = { 0, 0, 0, 0 };
    const int nStreams = a.streams == 1 ? 1 : 4;                         // HUF_decompress1X1_usingDTable: the block is ONE stream, no jump table
    const u32 seg = nStreams == 4 ? (u32)((dstSize + 3) / 4) : (u32)dstSize;
    const u32 jump = nStreams == 4 ? 6u : 0u;
    if (ok && nStreams == 4) {
        len[0] = __builtin_amdgcn_readfirstlane(ld16(in)); len[1] = __builtin_amdgcn_readfirstlane(ld16(in + 2)); len[2] = __builtin_amdgcn_readfirstlane(ld16(in + 4));      // (uniform values: scalar registers)
        const size_t used = (size_t)len[0] + len[1] + len[2] + 6;
        if (used > cSize) ok = false; else len[3] = (u32)(cSize - used);
        if (3 * (size_t)seg >= dstSize) ok = false;
    }
    if (ok && nStreams == 1) len[0] = (u32)cSize;
    if (ok) {
        const u8* sp = in + jump;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= nStreams) break;
            const u32 last = __builtin_amdgcn_readfirstlane(len[q] ? (u32)sp[len[q] - 1] : 0u);
            if (len[q] < 8 || last == 0) ok = false;                      // (a stream beyond the LDS budget is staged and decoded in pieces, below)
            else T0[q] = 8u * (len[q] - 1) + hibit32(last);              // unread bits under the end mark (bitstream.h:285-290)
            if (T0[q] < HPAR_MIN_BITS) ok = false;
            sp += len[q];
        }
    }
    // caller tables: one launch per LDS budget over all blocks, every block taken by the smallest budget that holds its longest stream
    // (the waves per CU follow the budget: 25 / 18 / 12); a block of another class leaves at once -- its launch has marked or will take it
    if (ok && !a.meta && !a.list) {
        const u32 mx = max(max(len[0], len[1]), max(len[2], len[3])) + 96u;
        if (mx <= a.classLo || (!HPAR_ALL_SMALL && mx > DATA && DATA != HPAR_DATA_LARGE)) return;      // uniform
    }
    // declined: the serial decoder's block -- on its list (one-shot path) or marked in results[] (caller tables: no workspace)
    auto decline = [&]() { if (lane == 0) { if (fbList) fbList[atomicAdd(fbCount, 1u)] = (u32)b; else a.results[b] = HUF_DECLINED; } };
    if (!ok) { decline(); return; }                                      // uniform (rare: k_huf_dprep has looked at the jump table)

    u32* const data = lds + ((size_t)1 << (a.ldsLog - 1));               // the staged stream, behind the table slot
    // ---- stage the table: reference cells {byte, nbBits} -> bit-reversed order, {nbBits, byte}
    if (tableType == 0) {
        const u32 words = 1u << (dtLog - 1);
        u16* const s = (u16*)lds;
        // The loops below advance by the cells' nbBits alone: a cell of a caller-built (zero-filled, half-built, damaged) table with
        // nbBits 0 would pin the cursor for ever, one beyond the table log breaks the 48-bits-per-iteration bound of hpar_run.  The
        // reference's loop is bounded by its output pointer whatever the table says (lib/huf_decompress.c:214-237), and so is the
        // serial kernel: such a table is declined to it.
        bool badCell = false;
        for (u32 i = lane; i < words; i += 64) {
            const u32 w = gt[1 + i];
            const u32 nb0 = (w >> 8) & 0xFFu, nb1 = (w >> 24) & 0xFFu;
            badCell |= (nb0 - 1u >= dtLog) | (nb1 - 1u >= dtLog);
            const u32 r0 = __brev(2u * i) >> (32u - dtLog);
            s[r0] = (u16)(nb0 | ((w & 0xFFu) << 8));
            s[r0 | (1u << (dtLog - 1))] = (u16)(nb1 | (((w >> 16) & 0xFFu) << 8));
        }
        if (__any(badCell)) { decline(); return; }                       // uniform
    } else if constexpr (X2CAP) {
        // ---- a double-symbol table from the reference's HUF_readDTableX2 (lib/huf_decompress.c:460-640): cell v = {u16 sequence; u8 nbBits;
        //      u8 length} -- the one or two symbols the next tableLog bits v start with and what they consume together.  A prefix code
        //      decodes the same symbols one at a time, so the single-symbol cell of v is {first symbol, ITS length}.  That length is not in
        //      the cell, but the cells beginning with symbol s form one aligned run of 2^(tableLog - len(s)) indices: a histogram of the
        //      first symbols gives every length.  Everything the equivalence rests on is checked -- runs aligned and contiguous, one-symbol
        //      cells carrying len(first), two-symbol cells carrying len(first) + len(second) with `second` the first symbol of what
        //      follows the first code -- and a table that fails any of it goes to the literal lock-step decoder (k_huf_decode_x2).
        // The table is read ONCE, 64 consecutive cells per step (cell 64k + lane in register k of the lane); the passes below work on the
        // registers.  Runs are found at their boundaries (a neighbour-lane comparison per cell, an LDS store per run), the second-symbol
        // check reads the single-symbol cells just staged in LDS instead of gathering from the table in global memory.
        u32* const cnt = data; u32* const lo = data + 256; u32* const hi = data + 512; u32* const nbs = data + 768;   // (the stream area is idle)
        const u32 ts = 1u << dtLog;
        constexpr u32 XK = (1u << FSEHIP_HUF_TABLELOG_MAX) / 64u;
        u32 cel[XK];
#pragma unroll
        for (u32 k = 0; k < XK; ++k) { const u32 i = 64u * k + lane; cel[k] = i < ts ? gt[1 + i] : 0u; }
        for (u32 i = lane; i < 256; i += 64) { cnt[i] = 0; lo[i] = 0; hi[i] = 0; }
        __syncthreads();
        u32 carry = 0;                                                   // first symbol of the cell in front of this step's cells
#pragma unroll
        for (u32 k = 0; k < XK; ++k) {
            const u32 i = 64u * k + lane;
            const u32 s1 = cel[k] & 0xFFu;
            const u32 up = (u32)__shfl_up((int)s1, 1, WAVE);
            const u32 prev = lane ? up : carry;
            if (i < ts && (i == 0 || s1 != prev)) {                      // a run starts here (and the one of `prev` ended in front of it)
                atomicAdd(&cnt[s1], 1u); lo[s1] = i;
                if (i) hi[prev] = i - 1;
            }
            if (i == ts - 1) hi[s1] = i;
            carry = (u32)__shfl((int)s1, 63, WAVE);
        }
        __syncthreads();
        bool bad = false;
        for (u32 sy = lane; sy < 256; sy += 64) {
            const u32 runs = cnt[sy];
            u32 nb = 0xFFu;                                              // no cell starts with this symbol
            if (runs) {
                const u32 n = hi[sy] - lo[sy] + 1;
                if (runs == 1 && (n & (n - 1)) == 0 && (lo[sy] & (n - 1)) == 0 && n < ts) nb = dtLog - hibit32(n);
                else bad = true;
            }
            nbs[sy] = nb;
        }
        __syncthreads();
        u16* const s = (u16*)lds;
#pragma unroll
        for (u32 k = 0; k < XK; ++k) {
            const u32 i = 64u * k + lane;
            if (i >= ts) continue;
            const u32 c = cel[k];
            const u32 s1 = c & 0xFFu, s2 = (c >> 8) & 0xFFu, nbTot = (c >> 16) & 0xFFu, len = c >> 24;
            const u32 n1 = nbs[s1];
            if (len == 1) bad |= nbTot != n1;
            else if (len == 2) bad |= (nbs[s2] == 0xFFu) | (nbTot != n1 + nbs[s2]) | (nbTot > dtLog);
            else bad = true;
            s[__brev(i) >> (32u - dtLog)] = (u16)(n1 | (s1 << 8));
        }
        __syncthreads();
        if (!__any(bad)) {                                               // (uniform; the lengths are sane: the shifts below stay inside the table)
#pragma unroll
            for (u32 k = 0; k < XK; ++k) {
                const u32 i = 64u * k + lane;
                const u32 c = cel[k];
                if (i >= ts || (c >> 24) != 2u) continue;
                const u32 n1 = nbs[c & 0xFFu];
                const u32 j = (i << n1) & (ts - 1);                      // what follows the first code, zero-extended: inside the second symbol's run
                bad |= (u32)(s[__brev(j) >> (32u - dtLog)] >> 8) != ((c >> 8) & 0xFFu);
            }
        }
        bad = __any(bad);
        __syncthreads();
        if (bad) { decline(); return; }                                  // uniform
    }
    const u32 tabOff = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)lds;
    if (tabOff & ((2u << a.ldsLog) - 1u)) __builtin_trap();               // the cell address is formed with an OR (dynamic LDS starts at 0)
    const u32 mask2 = ((1u << dtLog) - 1u) << 1;
    const u32 arr = tabOff + (2u << a.ldsLog);

    const u8* sp = in + jump;
    bool good = true;
    HST(unsigned long long stRounds = 0, stBad = 0, tStage = 0, tP1 = 0, tRep = 0, tP2 = 0, tA = __builtin_readcyclecounter();)
    // A stream longer than the LDS budget is taken in PIECES of PDW dwords: lane 63 of a piece ends on an exact boundary, which is where
    // lane 0 of the next piece starts (no speculation across pieces: they run one after the other), and the piece's 16 dwords of slack
    // hold the stream's next dwords instead of zeros.  One piece is the common case (the classes of k_huf_dprep fit their streams).
    constexpr u32 PDW = DATA / 4u - 24u;                                 // dwords of a piece: DATA bytes hold them, 16 dwords of slack and rounding to units
    constexpr u32 MAXU = (DATA / 16 + 63) / 64;
#ifndef HPAR_EARLY_LOADS
#define HPAR_EARLY_LOADS 0
#endif
    constexpr bool EARLY = HPAR_EARLY_LOADS && DATA < 8192u;                                  // the next stream's loads in front of this stream's stores (registers permitting)
    uint4 buf[MAXU];
    bool staged = false;                                                 // buf holds the next piece to stage already
    __syncthreads();                                                     // (the table is staged)
#pragma unroll 1
    for (int q = 0; q < nStreams; ++q) {                                 // uniform: stream after stream
        const u32 L = len[q], Sd = (L + 3) / 4;
        const u32 want = nStreams == 1 ? (u32)dstSize : (q < 3 ? seg : (u32)dstSize - 3 * seg);
        const u32 Cend = 32u * Sd;                                       // global cursor (bits consumed from the top of dword Sd-1) at the stream's first bit
        u32 Cstart = Cend - T0[q];                                       // ... and at the first code bit (above it: padding, end mark); later: where the next piece starts
        u32 outBase = 0;                                                 // symbols regenerated by the pieces so far
        // warm-up distance: a decoder dropped at an arbitrary bit falls into step after a number of CODES, so the distance follows the stream's bits
        // per symbol -- 24 codes' worth, at least 48 bits (round 6, per 100k blocks: P80 at 1.3 bits per symbol 2.78 -> 2.55 ms with 64 bits and
        // still no repair round; P14 at 4.3: 96 bits no repairs, 64 bits a repair round in 8 % of the streams); long codes resynchronise later
        // (8-bit data: 192 bits, where 128 cost 4 % and 96 a repair round per block)
        u32 warm = T0[q] > 6u * want ? HPAR_WARM + HPAR_WARM / 2 : (24u * T0[q]) / (want ? want : 1u);
        warm = warm < 48u ? 48u : (warm > HPAR_WARM + HPAR_WARM / 2 ? HPAR_WARM + HPAR_WARM / 2 : warm);
#pragma unroll 1
        while (good && Cstart < Cend) {                                  // uniform: piece after piece
            const u32 mTop = (Cstart - 1u) >> 5;                         // array dword 0 = global dword mTop = stream dword Sd - 1 - mTop; the local cursor stays >= 1
                                                                         // (the loops address the bit BELOW the cursor: Q = C - 1; Cstart >= 1: the end mark)
            const u32 rest = Sd - mTop, nPc = (rest + PDW - 1u) / PDW;   // what is left of the stream goes in pieces of equal size
            const u32 nd = nPc <= 1u ? rest : (rest + nPc - 1u) / nPc;   // dwords of this piece (<= PDW)
            const bool lastPiece = mTop + nd == Sd;
            if (!staged) hpar_stage_load<MAXU>(buf, sp, L, Sd, mTop, nd, lane);
            staged = false;
            hpar_wave_sync();                                            // (table staged / previous piece done: one wave, its LDS operations run in order)
            hpar_stage_commit<MAXU>(buf, data, nd, lane);
            hpar_wave_sync();
            HST({ const unsigned long long tB = __builtin_readcyclecounter(); tStage += tB - tA; tA = tB; })
#if defined(HPAR_ABL) && HPAR_ABL == 1     // measurement aid: staging only (results wrong)
            if (data[lane] == 0x12345u) out[lane] = 0;
            break;
#endif
            // cursors below are local to the piece's array: local = global - 32 * mTop
            const u32 C0 = Cstart - 32u * mTop;                          // exact: the stream's first code bit, or where the piece before ended
            const u32 CendL = (lastPiece ? Cend : 32u * (mTop + nd)) - 32u * mTop;
            const u32 Tp = CendL - C0;                                   // bits of this piece
            const u32 stepA = (Tp + 63u) / 64u;
            const u32 aLo = lane * stepA < Tp ? lane * stepA : Tp;
            const u32 aHi = (lane + 1) * stepA < Tp ? (lane + 1) * stepA : Tp;
            const u32 cLo = C0 + aLo, cHi = C0 + aHi;
            // ---- pass 1: warm up to my start, then my range
            u32 S = C0;
            if (lane > 0) { u32 C = aLo > warm ? cLo - warm : C0; (void)hpar_run(arr, C, cLo, tabOff, mask2); S = C; }
            // ---- my range, KEEPING the symbols in registers; verify / repair in the same loop (a repaired lane walks its range again).
            u32 sym[HPAR_KEEP];
#pragma unroll
            for (int k = 0; k < HPAR_KEEP; ++k) sym[k] = 0;
            u32 E = S, n = 0;
            bool spill = false, todo = true;
            // ---- verify / repair.  A code that never falls into step (say 256 equal weights: every code 8 bits, a lane dropped off the
            //      byte grid stays off it) gains one exact lane per round -- the serial walk at the price of a wave.  Real data needs
            //      0..2 rounds; beyond HPAR_MAX_REPAIR the block is the serial decoder's (crafted input cannot pin a wave to a block).
            for (u32 round = 0;; ++round) {
                if (todo) { E = S; n = hpar_run_keep(arr, E, cHi, tabOff, mask2, sym, spill); }
                HST(if (round == 0) { const unsigned long long tB = __builtin_readcyclecounter(); tP1 += tB - tA; tA = tB; })
                const u32 prevE = dpp_mov<0x138, 0xF, true>(0u, E);                   // wave_shr:1 (the value of lane - 1: DPP, not ds_bpermute -- the LDS pipe is this kernel's bound)
                const bool bad = lane > 0 && S != prevE;
                if (!__any(bad)) break;
                if (round == HPAR_MAX_REPAIR) { good = false; break; }              // uniform
                HST(++stRounds; stBad += __builtin_popcountll(__ballot(bad));)
                todo = bad;
                if (bad) S = prevE;
            }
            if (!good) break;
            HST({ const unsigned long long tB = __builtin_readcyclecounter(); tRep += tB - tA; tA = tB; })
            // ---- verdict for this piece
            const u32 incl = group_scan_incl<64, ScanAdd>(n, lane);                  // (dev_common.h: six DPP adds instead of six ds_bpermute)
            const u32 total = group_last<64>(incl, lane), endC = (u32)__builtin_amdgcn_readlane((int)E, 63);
            // the last piece must regenerate exactly what is left of the segment and end exactly on the stream's first bit; a piece before
            // it must leave symbols to regenerate
            if (lastPiece ? (outBase + total != want || endC != CendL) : (outBase + total >= want)) { good = false; break; }   // uniform: not a stream the reference accepts as is
            // ---- the next stream's loads go out now, in front of this one's stores
            if (EARLY && lastPiece && q + 1 < nStreams) {
                const u32 L2 = len[q + 1], Sd2 = (L2 + 3) / 4, Cs2 = 32u * Sd2 - T0[q + 1];
                const u32 mTop2 = (Cs2 - 1u) >> 5, rest2 = Sd2 - mTop2, nPc2 = (rest2 + PDW - 1u) / PDW, nd2 = nPc2 <= 1u ? rest2 : (rest2 + nPc2 - 1u) / nPc2;
                hpar_stage_load<MAXU>(buf, sp + L, L2, Sd2, mTop2, nd2, lane);
                staged = true;
            }
            // ---- pass 2: my symbols out of the registers; only a piece with a spilled lane decodes them again
#if defined(HPAR_ABL) && HPAR_ABL == 2     // measurement aid: no stores (results wrong)
            if (!__any(spill)) { u32 x = 0;
#pragma unroll
                for (int k = 0; k < HPAR_KEEP; ++k) x ^= sym[k];
                if (x == 0x12345u) out[lane] = 0; }
#else
#if defined(HPAR_DIRECT_STORE)              // A/B aid (EXPERIMENTS.md): every lane stores its own run straight from the registers
            if (!__any(spill)) hpar_store_kept(out + (size_t)q * seg + outBase + (incl - n), n, sym);
#else
            if (!__any(spill)) {
                u8* const g0 = out + (size_t)q * seg + outBase;
                const u32 h = (u32)(uintptr_t)g0 & 15u;
                constexpr u32 CH = DATA >= 8192u + 2u * HPAR_FLUSH_SLACK ? 8192u : DATA >= 4096u + 2u * HPAR_FLUSH_SLACK ? 4096u : 1024u;
                static_assert(CH + 2u * HPAR_FLUSH_SLACK <= DATA, "the line buffer must fit the stream's staging area");
                hpar_flush_kept<CH>(arr, g0 - h, h, total, h + (incl - n), n, sym, lane);
            }
#endif
#endif
            else {
                u8* p = out + (size_t)q * seg + outBase + (incl - n);
                u32 left = n, C = S;
#ifdef HPAR_NO_STORE
                u32 dummyAcc = 0;
#endif
                while (left && ((uintptr_t)p & 3u)) { const u32 c = hpar_single(arr, C, tabOff, mask2); *p++ = (u8)(c >> 8); --left; }
                if (left >= 16) {
                    HparBulk bk; bk.open(arr, C);
                    do {
                        u32 w[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) w[t] = bk.iter(tabOff, mask2);
#ifdef HPAR_NO_STORE      // measurement aid (EXPERIMENTS.md): pass 2 without its 16-byte stores -- is the scattered store pattern what bounds the kernel?
                        dummyAcc ^= w[0] ^ w[1] ^ w[2] ^ w[3];
#else
                        __builtin_memcpy(p, w, 16);
#endif
                        p += 16; left -= 16;
                    } while (left >= 16);
                    while (left >= 4) { const u32 w = bk.iter(tabOff, mask2); __builtin_memcpy(p, &w, 4); p += 4; left -= 4; }
                    C = bk.cursor();
                }
                while (left) { const u32 c = hpar_single(arr, C, tabOff, mask2); *p++ = (u8)(c >> 8); --left; }
#ifdef HPAR_NO_STORE
                if (dummyAcc == 0x12345u) *p = 0;
#endif
            }
            outBase += total;
            Cstart = endC + 32u * mTop;
            HST({ const unsigned long long tB = __builtin_readcyclecounter(); tP2 += tB - tA; tA = tB; })
        }
        if (!good) break;
        sp += L;
    }
    HST(if (lane == 0 && slot < 4096) { unsigned long long* t = g_hparStats + 8 * slot; t[0] = stRounds; t[1] = stBad; t[2] = tStage; t[3] = tP1; t[4] = tRep; t[5] = tP2; })
    if (!good) { decline(); return; }
    if (lane == 0) a.results[b] = dstSize;
}

// (Measured: a grid of resident waves striding over the list instead of one wave per block -- cheaper empty launches for the
// classes without blocks -- is 5 to 35 % slower on the class that has them.)
template <u32 DATA>
static hipError_t hpar_launch(const HufDecArgs& a, u32* serialList, u32* serialCount, hipStream_t s)
{
    hipLaunchKernelGGL((k_huf_decode_par<DATA, false>), dim3((unsigned)a.nBlocks), dim3(64), ((size_t)2 << a.ldsLog) + DATA, s, a, serialList, serialCount);
    return hipGetLastError();
}
// caller-built tables, second launch: the blocks with double-symbol tables the lean launch marked HUF_DECLINED
hipError_t launch_huf_decode_par_x2(HufDecArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    a.onlyDeclined = 1;
    hipLaunchKernelGGL((k_huf_decode_par<HPAR_DATA_LARGE, true>), dim3((unsigned)a.nBlocks), dim3(64), ((size_t)2 << a.ldsLog) + HPAR_DATA_LARGE, s, a, (u32*)nullptr, (u32*)nullptr);
    return hipGetLastError();
}

// one-shot path: the parallel decoder over a class list, `dataBytes` of LDS per staged stream (with 4 KiB tables HPAR_DATA_TINY:
// 25 waves per CU, HPAR_DATA_SMALL: 18, HPAR_DATA_LARGE: 12); what it declines is appended to the serial decoder's list
hipError_t launch_huf_decode_par(HufDecArgs a, unsigned dataBytes, u32* serialList, u32* serialCount, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    if (dataBytes == HPAR_DATA_TINY) return hpar_launch<HPAR_DATA_TINY>(a, serialList, serialCount, s);
    if (dataBytes == HPAR_DATA_SMALL) return hpar_launch<HPAR_DATA_SMALL>(a, serialList, serialCount, s);
    return hpar_launch<HPAR_DATA_LARGE>(a, serialList, serialCount, s);
}
