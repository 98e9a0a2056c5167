// fse_decode.hip -- a3: FSE_decompress_usingDTable over a batch
// (reference: lib/fse_decompress.c:178-252, lib/fse.h:577-622, lib/bitstream.h:272-448).
//
// tANS decoding is one loop-carried chain per block (two interleaved states sharing one bit cursor; a simulation of
// speculative starts shows that this two-state decoder does not re-synchronise within thousands of symbols, so unlike
// the encoder it cannot be split), hence "a lane pair per block, as many blocks per CU as LDS holds":
//   * ONE workgroup per CU over all 160 KB of LDS: FSE_DEC_WAVES = 2 decoder waves + FSE_SRV_WAVES = 9 service waves (704 threads)
//     over G = 33 blocks with 4 KiB tables (tableLog <= 11; 18 blocks with the 8 KiB tables of tableLog 12) -- 4944 bytes per block:
//     the u16 table, a 64-entry ring of decoded states, a 256-byte ring of compressed input, 48 bytes of control words;
//   * a block is walked by a PAIR of decoder lanes (lane A owns state 1, lane B state 2; wave 0 holds 17 pairs, wave 1 holds 16) that
//     touch registers and LDS only: one ds_read fetches both states' cells, what the other lane needs moves through DPP;
//   * the service waves (four blocks each) own all global-memory traffic of the bulk loop, wave-cooperative and coalesced: they
//     refill the input rings (64-byte chunks requested one ahead) and turn state-ring records into output bytes (symbol gathers
//     from an L2-resident byte table, 256-byte rows);
//   * the two sides talk through per-block control words in LDS (acquire/release, workgroup scope); the rounds between two phases
//     of 16 iterations are wave-uniform (a chain that cannot take a phase decodes on as a "zombie" inside its own LDS slot).
// Throughput = blocks resident per CU (LDS capacity: 33 at tableLog 11) / latency of one iteration of the chain (two dependent
// LDS look-ups and the VALU ops between them; a lone wave issues an instruction per ~7-8 cycles whatever its lane count, so the
// loop is laid out for instruction count: ~33 instructions, ~230 cycles per 4 symbols).  DESIGN.md 4.3 / 4.3a.
//
// The kernel reproduces the reference's decoder state -- a 64-bit little-endian window at byte offset `at` plus a
// consumed-bit count `used` (lib/bitstream.h:91-97) -- exactly, including on truncated / corrupt input:
//   * `BitReader` is a literal device restatement of BIT_initDStream / BIT_readBits / BIT_reloadDStream and runs the
//     initialisation and the last symbols of every stream;
//   * the bulk loop is lib/fse_decompress.c:201-218 (4 symbols per reload) for as long as the window is at least 24
//     bytes above the stream start, where every reload is the "fast" one (:378-388) and (ptr, bitsConsumed) are a
//     function of the absolute bit position alone.
#include "internal.h"
#include <atomic>

#include "bitreader.h"

// Two bulk loops share the kernel:
//   * fse_bulk_phase_rev (k_fse_decode<true>): tables from k_fse_dbuild in the bit-reversed layout (tableLog <= 11, or 12 when no
//     cell has nbBits == 0: the classes of internal.h, each launched over its own list of blocks) --
//     the one the one-shot decompressor uses; described at the function;
//   * fse_bulk_phase (k_fse_decode<false>): caller-built reference-layout DTables (staged as A[x] = newState (12 bits) |
//     nbBits << 12) and maxTableLog 12.  FSE_buildDTable makes the low nbBits of newState zero
//     (lib/fse_decompress.c:121-122), so "newState + bits" is an OR; tables that violate this take the literal path.
//
// Window of fse_bulk_phase: bq = number of still unread bits of stream dword dp (its low bits; 0..31); q = 4*dp - 8 is the
// payload byte offset of the lowest of the three dwords {d2:d1:d0} = dwords dp, dp-1, dp-2 (dword-aligned with respect to
// the payload start, which is how the LDS input ring is laid out).  One iteration of lib/fse_decompress.c:201-218 = 4
// symbols = at most 48 bits:
//   {thi:tlo} = the next 64 unread bits (two v_alignbit);
//   symbol 1 reads the top of thi, symbol 2 the top of thi << nb1 (>= 20 valid bits), symbols 3/4 the same from
//   t3 = the 32 bits that follow the first two symbols (one more v_alignbit) -- no per-symbol window shifting;
//   then bq -= consumed, carrying into q; the window registers slide by selects and the two dwords below them are
//   prefetched at the top of the iteration.
// The reference's (ptr, bitsConsumed) pair is a function of the absolute bit position alone while its reloads are
// the fast ones (bitstream.h:378-388), so it is reconstructed from the cursor when the bulk loop ends.
// NB0 = some cell of some table staged by this workgroup has nbBits == 0: a v_alignbit by 32 would return the low
// word, so that one select is made explicit.
#ifndef FSE_SYM_L1
#define FSE_SYM_L1 1             // caller tables: the service waves' symbol gathers go through the CU's vector cache (invalidated when the slot is claimed)
#endif
#ifndef FSE_DEC_RING
#define FSE_DEC_RING 64          // per-block LDS state ring: one entry (4 states, 8 bytes) per bulk iteration = four phases.  (Any multiple of
#endif                           // FSE_CHECK_EVERY works.  Measured with 48 entries, which makes room for a 17th block per workgroup: the lanes of
                                 // the decoder wave fall out of step waiting for their flushes -- 541 instead of 481 phase rounds per 32 KB block,
                                 // service waves 96 % busy -- 14.6 ms per 100k blocks instead of 12.9.)
#define FSE_IN_RING 256          // per-block LDS input ring (bytes of compressed stream, direct-mapped by offset mod 256)
#define FSE_IN_RING_LOG 8
#define FSE_IN_CHUNK 64          // refill granule: one 4-byte load per lane of a 16-lane group (one group per block of a service wave)
#define FSE_IN_LANES (FSE_IN_CHUNK / 4)
#define FSE_IN_MIRROR 16         // the first bytes are mirrored behind the ring so reads of 2 dwords never wrap
#ifndef FSE_CHECK_EVERY
#define FSE_CHECK_EVERY 16       // bulk iterations per phase (<= 6 bytes consumed per iteration)
#endif
#ifndef FSE_PHASE_UNROLL
#define FSE_PHASE_UNROLL 16      // the phase is straight-line code: with an inner loop the compiler shuffles every loop-carried register of the
#endif                           // round through copies at each loop header (measured in instructions per round, see DESIGN)
#define FSE_FINISH_EVERY 2       // ... and per finishing phase of the bit-reversed loop (a ring slot pair holds two iterations)
// Workgroup geometry.  A CU's LDS (160 KB) is what bounds the blocks in flight; ONE workgroup per CU over all of it holds 33 blocks
// with 4 KiB tables (4944 bytes each) where two workgroups of 80 KB hold 2 x 16.  FSE_DEC_WAVES decoder waves share the lane pairs
// (17 + 16), FSE_SRV_WAVES service waves look after four blocks each.  (FSE_DEC_LDS_KB 80 / FSE_DEC_WAVES 1 / FSE_SRV_WAVES 4 is the
// earlier two-workgroups-per-CU layout, kept buildable for A/B runs: make B=variants/wg80 EXTRA="-DFSE_DEC_LDS_KB=80 ...".)
#ifndef FSE_DEC_LDS_KB
#define FSE_DEC_LDS_KB 160
#endif
#ifndef FSE_DEC_WAVES
#define FSE_DEC_WAVES 2
#endif
#ifndef FSE_SRV_WAVES
#define FSE_SRV_WAVES 9
#endif
#define FSE_MAXG (4 * FSE_SRV_WAVES)              // blocks per workgroup at most (FSE_SRV_WAVES x FSE_SRV_G)
#define FSE_WGS_PER_CU (160 / FSE_DEC_LDS_KB)

// Cycle accounting of the decoder and service waves (bench.py's `roofline.secondary`): the TIMED instantiation of the kernel brackets
// every round of its phase loop with s_memtime and adds its totals to g_decTiming when the workgroup ends.  Off by default: the
// launcher picks the TIMED kernel only between FSEHIP_debug_decodeTiming(1, ..) and (0, ..).
//   [0] cycles of decoder-wave rounds in which some lane pair ran a phase, [1] cycles of rounds in which none could, [2] / [3] their
//   numbers, [4] workgroups, [5] / [6] busy / idle cycles of the first service wave of every workgroup, [7] service rounds that did work,
//   [11] lifetime of the workgroups' decoder waves in cycles and [12] in ticks of the constant 100 MHz clock (their ratio is the engine
//   clock the kernel really ran at), [13] cycles from kernel entry to the first phase (table staging, reader set-up, first ring fill),
//   [14] cycles from the last phase to the end (literal tail), [15] / [10] cycles / number of rounds that ran finishing phases only
__device__ unsigned long long g_decTiming[16];
static std::atomic<bool> g_decTimingOn{false};   // benchmark-only switch (FSEHIP_debug_decodeTiming): process-wide, meant for one thread that owns the device while it is on
#define TIMING(...) do { if constexpr (TIMED) { __VA_ARGS__ } } while (0)
struct BulkState { u32 s, q, bq; };     // this lane's state (cell address) and the pair's bit cursor

// The decoder lanes keep their states as absolute LDS byte addresses of the table cells.
// The service waves' traffic with global memory goes through global-address-space pointers: a pointer rebuilt from shuffled integers
// is generic, its accesses become flat_* instructions, and those count on lgkmcnt as well as vmcnt -- every wait of the wave's LDS
// traffic (control words, ring records) then waits for the symbol gathers in flight too.  Measured: 10.73 -> 10.54 ms per 100k P14
// blocks, P02 13.23 -> 13.00.
typedef const __attribute__((address_space(1))) u8* gbl_u8_ptr;
typedef __attribute__((address_space(1))) u8* gbl_u8_w_ptr;
typedef u32 __attribute__((aligned(1))) u32_unaligned;
DEV u32 gbl_load_u32(gbl_u8_ptr p) { return *(const __attribute__((address_space(1))) u32_unaligned*)p; }      // (any alignment: one global_load_dword)
DEV void gbl_store_u32(gbl_u8_w_ptr p, u32 w) { *(__attribute__((address_space(1))) u32_unaligned*)p = w; }
typedef const __attribute__((address_space(3))) u16* lds_u16_ptr;
typedef const __attribute__((address_space(3))) u32* lds_u32_ptr;
DEV u32 lds_cell(u32 addr) { return *(lds_u16_ptr)(uintptr_t)addr; }   // absolute LDS byte address -> table cell

DEV u32 dpp_swap(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }   // value of the neighbour lane (quad_perm [1,0,3,2])

// One phase = FSE_CHECK_EVERY iterations, registers + LDS only.
// A lone wave pays for every LDS *instruction* (they hardly overlap within a wave), so a block is walked by a pair of
// lanes: lane A (even) owns state 1, lane B (odd) owns state 2.  One ds_read then fetches the cells of both states, one
// ds_read2 the four window dwords (two per lane), one ds_write the states both lanes decoded from; what the other lane
// needs (the neighbour's cell for its bit count, the window dwords) moves through DPP, which costs a VALU slot.
// Both lanes keep identical copies of the bit cursor (q, bq).  maskB = all ones in lane B: its symbol's bits come
// after lane A's.
template <bool NB0>
DEV void fse_bulk_phase(u32& sMine, u32& qRef, u32& bqRef, u32 tabOff, u32 myIn, u32 half, u32 maskB, uint2* ringMine)
{
    u32 s = sMine, q = qRef, bq = bqRef;
    u32 prev = 0;
    (void)half;
    // window {w2:w1:w0} = payload dwords at q+8, q+4, q in registers (both lanes identical); the two dwords below it are
    // read at the top of every iteration so that the read is off the dependent chain, and the window slides by selects
    u32 w0, w1, w2;
    {   const lds_u32_ptr wp = (lds_u32_ptr)(uintptr_t)(myIn + (q & (FSE_IN_RING - 4)));
        w0 = wp[0]; w1 = wp[1]; w2 = wp[2]; }
#pragma unroll 8
    for (int it = 0; it < FSE_CHECK_EVERY; ++it) {
        const u32 c = lds_cell(s);                               // lane A: state 1's cell, lane B: state 2's
        const lds_u32_ptr np = (lds_u32_ptr)(uintptr_t)(myIn + ((q - 8u) & (FSE_IN_RING - 4)));
        const u32 n0 = np[0], n1 = np[1];                        // payload dwords at q-8, q-4
        const u32 thi = __builtin_amdgcn_alignbit(w2, w1, bq), tlo = __builtin_amdgcn_alignbit(w1, w0, bq);
        const u32 cO = dpp_swap(c);
        const u32 nbM = c >> 12, nbO = cO >> 12;
        const u32 sStart = s;
        {   const u32 t = thi << (nbO & maskB);
            const u32 bits = __builtin_amdgcn_ubfe(t, 32u - nbM, nbM);
            s = (((c & 0xFFFu) | bits) << 1) + tabOff; }
        const u32 c2 = lds_cell(s);
        const u32 s12 = nbM + nbO;
        u32 t3 = __builtin_amdgcn_alignbit(thi, tlo, 32u - s12);
        if (NB0) t3 = s12 ? t3 : thi;
        const u32 rec = __builtin_amdgcn_perm(s, sStart, 0x05040100u);   // low 16 bits of the two cell addresses this lane decoded from
        const u32 cO2 = dpp_swap(c2);
        const u32 nbM2 = c2 >> 12, nbO2 = cO2 >> 12;
        {   const u32 t = t3 << (nbO2 & maskB);
            const u32 bits = __builtin_amdgcn_ubfe(t, 32u - nbM2, nbM2);
            s = (((c2 & 0xFFFu) | bits) << 1) + tabOff; }
        const int left = (int)bq - (int)(s12 + nbM2 + nbO2);     // unread bits of dword dp after this iteration (>= -48)
        const bool k1 = left < 0, k2 = left < -32;               // the window slides down by one / two dwords
        w2 = k2 ? w0 : (k1 ? w1 : w2);
        w1 = k2 ? n1 : (k1 ? w0 : w1);
        w0 = k2 ? n0 : (k1 ? n1 : w0);
        q += (u32)((left >> 5) << 2);                            // arithmetic shift: 0, -1 or -2 dwords
        bq = (u32)left & 31u;
        // two iterations per ring slot pair: this lane's half of slot pair (it >> 1) holds its states of both iterations
        if (it & 1) ringMine[it & ~1] = make_uint2(prev, rec); else prev = rec;
    }
    sMine = s; qRef = q; bqRef = bq;
}

// ---- bit-reversed bulk loop (tables from k_fse_dbuild) -------------------------------------------------------------
// A block's chain is two LDS round trips per iteration plus whatever stands between a cell's arrival and the next request, so this
// variant is laid out to need as little as possible there: 6 VALU per symbol, the bit cursor and the window off the chain (the
// issue order is written out in fse_bulk_phase_rev below).
//   * The input ring holds the stream in CONSUMPTION order: ring dword m = bit-reversed payload dword (Stop/4 - 1 - m)
//     (Stop = payload size rounded up to 4), written that way by the service waves.  The cursor is one number,
//     P = 8*Stop - (unread bits): the next bit to read is bit P & 31 of ring dword P >> 5, and bits are taken from the
//     low end -- v_bfe_u32(window, offset, nbBits) with the offset and width operands used as they come (the hardware
//     reads their low 5 bits), no per-iteration select of window registers, no special case for nbBits == 0.
//   * Taking bits low-end-first yields them bit-reversed, so the table is stored bit-reversed too: the cell of state x
//     sits at index rev(x), and holds nbBits (low 5 bits) | rev(newState) << 5 (11 bits: rev(newState) < 2048 because tableLog <= 11,
//     or tableLog is 12 and newState is even -- every nbBits >= 1 -- which is what the 8 KiB class requires).  newState is a
//     multiple of 1 << nbBits (lib/fse_decompress.c:121-122), hence rev(newState + bits) = rev(newState) | rev_nb(bits) << (tableLog - nbBits):
//     next address = tableBase | cell >> 4 | bits << (tableLog + 1 - nbBits)   (v_lshrrev, v_or, v_sub, v_lshl_or; bit 0 of cell >> 4
//     is bit 4 of nbBits = 0).
//   * Lane B's bits follow lane A's: its offset is lane A's nbBits, fetched and masked by one v_and_b32_dpp
//     (maskB = 31 in lane B, 0 in lane A); the pair's bit count is one v_add_u32_dpp.
// One iteration = 2 symbols per lane = at most 44 bits out of the 64+ bits {d2:d1:d0} >> (P & 31) read at its top.
DEV u32 dpp_swap_and(u32 v, u32 m) { return dpp_swap(v) & m; }      // (the compiler folds these into v_and_b32_dpp / v_add_u32_dpp)
DEV u32 dpp_swap_add(u32 v, u32 w) { return dpp_swap(v) + w; }
// instruction selection helpers: keep the shapes the instruction count above relies on
DEV u32 lshl_or(u32 v, u32 sh, u32 o) { u32 r; __asm__("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(sh), "v"(o)); return r; }   // (v << sh[4:0]) | o
DEV u32 ring_dword(u32 P) { u32 r; __asm__("v_bfe_u32 %0, %1, 5, %2" : "=v"(r) : "v"(P), "n"(FSE_IN_RING_LOG - 2)); return r; }              // (P >> 5) mod ring dwords
// PheadRef: the cursor at the head of the phase's LAST iteration (the reference's reader state between two loop-head reloads is
// fixed by the bits unread at the last reload and the bits unread now: k_fse_decode rebuilds it from the two).
#ifndef FSE_SCHED
#define FSE_SCHED 1
#endif
#if FSE_SCHED
// The phase with its issue order written out (sched_barrier between the groups) -- the compiler's own schedule (FSE_SCHED 0, below)
// requested a cell only after the window reads and waited for the window behind a cursor update that itself waited for the cell.
// Per iteration there are two LDS round trips nothing can hide (cell of symbol pair 1 -> cell of pair 2 -> next iteration's cell); the
// rest is arranged around them:
//   * a cell is requested the moment its address exists, and only the six instructions that need it (v_and_b32_dpp, v_bfe, v_sub,
//     v_lshrrev, v_or, v_lshl_or) stand between its arrival and the next request;
//   * the window is off the chain: the four dwords the NEXT iteration may need are requested as soon as the first symbol pair's bit
//     count is known (position P1; the second pair moves the cursor by < 32 bits more, so the next window starts in the dword of P1
//     or in the one after it) and the iteration picks its three by selects -- five more VALU instructions, all issued while a cell
//     is on its way, and no LDS round trip between the cursor and the window.
// Per 100k Proba14 blocks (decode call = dparse + dbuild + this kernel): compiler's schedule 11.59 ms; cell requested early, cell-only
// work before the window wait 11.27; with the window requested one iteration ahead 10.72.
#define SB __builtin_amdgcn_sched_barrier(0)
// Measurement aid (EXPERIMENTS.md, "what a smaller table cell may cost"): FSE_EXTRA_CHAIN_OPS dependent no-op VALU instructions behind every
// next-state computation -- the slope ms per instruction on the chain, against which any cell format that saves LDS but adds work
// between a cell's arrival and the next request has to be weighed.  0 in the product.
#ifndef FSE_EXTRA_CHAIN_OPS
#define FSE_EXTRA_CHAIN_OPS 0
#endif
DEV u32 fse_chain_pad(u32 s)
{
#pragma unroll
    for (int i = 0; i < FSE_EXTRA_CHAIN_OPS / 2; ++i) __asm__ volatile("v_xor_b32 %0, 1, %0\n\tv_xor_b32 %0, 1, %0" : "+v"(s));
    return s;
}
// Measurement aid (EXPERIMENTS.md, "the state ring in global memory"): FSE_PROBE_GSTORE 1 = the decoder lanes ALSO store every record pair to
// global memory (one global_store_dwordx2 per lane and two iterations, SGPR base + VGPR offset), 2 = INSTEAD of the LDS ring (results wrong:
// timing only) -- what would a state ring outside LDS (36 instead of 33 blocks per CU) cost the decoder wave's instruction stream?
#ifndef FSE_PROBE_GSTORE
#define FSE_PROBE_GSTORE 0
#endif
#if FSE_PROBE_GSTORE
__device__ u8 g_probeRing[512u * 64u * 2u * 4096u];                 // 512 workgroup slots x 64 lanes x 2 decoder waves x 4 KiB
DEV void fse_probe_gstore(u32 off, u32 a, u32 b)
{
    typedef u32 v2 __attribute__((ext_vector_type(2)));
    __attribute__((address_space(1))) u8* const base = (__attribute__((address_space(1))) u8*)g_probeRing;
    *(__attribute__((address_space(1))) v2*)(base + off) = (v2){ a, b };
}
#endif
// Measurement aid (EXPERIMENTS.md section 1, round 6): FSE_PEEK_IN_PHASE 1 requests the service's progress words {srvFlushed, srvValidLo} the NEXT round
// looks at HERE, a few iterations before the phase ends, instead of in the round's poll loop.  Reason to try: that loop is a do-while, and the
// compiler waits for the words requested in it at the bottom of the loop body on the exit path too (the loop-carried copy needs the data: s_waitcnt
// lgkmcnt(0) + v_mov_b64 in front of the loop's branch), so every round pays part of an LDS round trip that the request "one round ahead" was meant
// to hide.  Measured: 3700 -> 3671 cycles per productive round, and the decode call 0.3 - 0.8 % SLOWER on every distribution.  0 in the product.
#ifndef FSE_PEEK_IN_PHASE
#define FSE_PEEK_IN_PHASE 0
#endif
#ifndef FSE_PEEK_AT
#define FSE_PEEK_AT 4            // iterations before the end of the phase
#endif
template <int NITER>
DEV void fse_bulk_phase_rev(u32& sMine, u32& Pref, u32& PheadRef, u32 K, u32 cellShift, u32 tabOff, u32 myIn, u32 maskB, uint2* ringMine,
                            const u32* peekAt = nullptr, u64* peekOut = nullptr)
{
    u32 s = sMine, P = Pref;
    u32 prev = 0;
    __asm__ volatile("" : "+v"(myIn));
#if FSE_PROBE_GSTORE
    // (a 4 KiB stretch per lane, the position inside it following the LDS ring's: 512 bytes of it are ever touched)
    const u32 probeOff = (((blockIdx.x & 511u) * 128u + (threadIdx.x & 127u)) << 12) + ((u32)(uintptr_t)ringMine & 0x1F8u);
#endif
    u32 c = lds_cell(s);
    u32 Pw = P;
    lds_u32_ptr wp = (lds_u32_ptr)(uintptr_t)(myIn + (ring_dword(P) << 2));
    u32 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
#pragma unroll FSE_PHASE_UNROLL
    for (int it = 0; it < NITER; ++it) {
        if (it == NITER - 1) PheadRef = P;
        // (while the cell is on its way) this iteration's window out of the dwords requested at Pw
        const bool k = ((P ^ Pw) & 32u) != 0;
        const u32 e0 = k ? w1 : w0, e1 = k ? w2 : w1, e2 = k ? w3 : w2;
        u32 lo = __builtin_amdgcn_alignbit(e1, e0, P);
        u32 hi = __builtin_amdgcn_alignbit(e2, e1, P);
        __asm__ volatile("" : "+v"(lo), "+v"(hi));
        SB;
        const u32 sStart = s;
        s = fse_chain_pad(lshl_or(__builtin_amdgcn_ubfe(lo, dpp_swap_and(c, maskB), c), K - c, (c >> cellShift) | tabOff));
        const u32 c2 = lds_cell(s);
        SB;
        const u32 n1 = dpp_swap_add(c, c);
        const u32 P1 = P + (n1 & 31u);
        if (it + 1 < NITER) {
            Pw = P1;
            wp = (lds_u32_ptr)(uintptr_t)(myIn + (ring_dword(P1) << 2));
            w0 = wp[0]; w1 = wp[1]; w2 = wp[2]; w3 = wp[3];
        }
        u32 lo2 = __builtin_amdgcn_alignbit(hi, lo, n1);
        u32 rec = __builtin_amdgcn_perm(s, sStart, 0x05040100u);
        __asm__ volatile("" : "+v"(lo2), "+v"(rec));
#if FSE_PROBE_GSTORE == 2
        if (it & 1) fse_probe_gstore(probeOff + 8u * (u32)(it & ~1), prev, rec); else prev = rec;
#elif FSE_PROBE_GSTORE == 1
        if (it & 1) { ringMine[it & ~1] = make_uint2(prev, rec); fse_probe_gstore(probeOff + 8u * (u32)(it & ~1), prev, rec); } else prev = rec;
#else
        if (it & 1) ringMine[it & ~1] = make_uint2(prev, rec); else prev = rec;
#endif
#if FSE_PEEK_IN_PHASE
        if (peekOut && it == (NITER > FSE_PEEK_AT ? NITER - FSE_PEEK_AT : 0)) *peekOut = __hip_atomic_load((const u64*)peekAt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        SB;
        s = fse_chain_pad(lshl_or(__builtin_amdgcn_ubfe(lo2, dpp_swap_and(c2, maskB), c2), K - c2, (c2 >> cellShift) | tabOff));
        if (it + 1 < NITER) c = lds_cell(s);
        SB;
        const u32 n2 = dpp_swap_add(c2, c2);
        P = P1 + (n2 & 31u);
    }
    sMine = s; Pref = P;
}
#undef SB
#else
template <int NITER>
DEV void fse_bulk_phase_rev(u32& sMine, u32& Pref, u32& PheadRef, u32 K, u32 cellShift, u32 tabOff, u32 myIn, u32 maskB, uint2* ringMine,
                            const u32* peekAt = nullptr, u64* peekOut = nullptr)
{
    u32 s = sMine, P = Pref;
    u32 prev = 0;
    if (peekOut) *peekOut = __hip_atomic_load((const u64*)peekAt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __asm__ volatile("" : "+v"(myIn));                           // one register: the three window reads then differ by their immediate offsets
#pragma unroll FSE_PHASE_UNROLL
    for (int it = 0; it < NITER; ++it) {
        if (it == NITER - 1) PheadRef = P;
        const u32 c = lds_cell(s);                               // lane A: state 1's cell, lane B: state 2's
        const lds_u32_ptr wp = (lds_u32_ptr)(uintptr_t)(myIn + (ring_dword(P) << 2));
        const u32 d0 = wp[0], d1 = wp[1], d2 = wp[2];
        const u32 lo = __builtin_amdgcn_alignbit(d1, d0, P), hi = __builtin_amdgcn_alignbit(d2, d1, P);
        const u32 sStart = s;
        {   const u32 bits = __builtin_amdgcn_ubfe(lo, dpp_swap_and(c, maskB), c);
            s = lshl_or(bits, K - c, (c >> cellShift) | tabOff); }
        const u32 c2 = lds_cell(s);
        const u32 n1 = dpp_swap_add(c, c);                       // low 5 bits: bits of this symbol pair
        const u32 lo2 = __builtin_amdgcn_alignbit(hi, lo, n1);
        const u32 rec = __builtin_amdgcn_perm(s, sStart, 0x05040100u);   // low 16 bits of the two cell addresses this lane decoded from
        {   const u32 bits = __builtin_amdgcn_ubfe(lo2, dpp_swap_and(c2, maskB), c2);
            s = lshl_or(bits, K - c2, (c2 >> cellShift) | tabOff); }
        const u32 n2 = dpp_swap_add(c2, c2);
        P += (n1 & 31u) + (n2 & 31u);
        // two iterations per ring slot pair: this lane's half of slot pair (it >> 1) holds its states of both iterations
        if (it & 1) ringMine[it & ~1] = make_uint2(prev, rec); else prev = rec;
    }
    sMine = s; Pref = P;
}

#endif

// Per-block control words in LDS: the decoder wave and the service wave of a workgroup talk through these only.
struct DecCtl {
    u32 pubIters;      // decoder -> service: bulk iterations completed (state-ring records produced)
    u32 pubPofs;       // decoder -> service: window position p = at+1; bit 31 = bulk finished (pubIters is final)
    u32 srvFlushed;    // service -> decoder: state-ring records already turned into output bytes
    int srvValidLo;    // service -> decoder: the input ring holds stream bytes [validLo, validLo + FSE_IN_RING); INT_MAX = not yet
    int initValidLo;   // set-up constants for the service wave
    int S32;
    u32 inLo, inHi, outLo, outHi, symLo, symHi;
};
#define FSE_DEC_THREADS (64 * (FSE_DEC_WAVES + FSE_SRV_WAVES))     // waves 0 .. FSE_DEC_WAVES-1 decode, the others serve

DEV u32 ctl_load(const u32* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV int ctl_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV u32 ctl_peek(const u32* p) { const u32 v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __asm__ volatile("" ::: "memory"); return v; }
DEV int ctl_peek(const int* p) { const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __asm__ volatile("" ::: "memory"); return v; }
DEV void ctl_store(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// two neighbouring control words as one LDS access (8-byte aligned pairs of DecCtl): the decoder publishes {pubIters, pubPofs} with ONE
// release store and reads {srvFlushed, srvValidLo} with one load -- every LDS instruction and every s_waitcnt between two phases is paid
// by all the chains of the decoder wave
DEV u64 ctl_load2(const u32* p) { return __hip_atomic_load((const u64*)p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV u64 ctl_peek2(const u32* p) { const u64 v = __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __asm__ volatile("" ::: "memory"); return v; }
DEV void ctl_store2(u32* p, u32 lo, u32 hi) { __hip_atomic_store((u64*)p, (u64)lo | ((u64)hi << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void ctl_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- service waves: keep the input rings filled and turn state-ring records into output bytes, so that the decoder
//      wave spends its cycles on the dependent chains only.  Each of the FSE_SRV_WAVES service waves looks after
//      FSE_SRV_G of the workgroup's blocks; everything here is wave-cooperative and coalesced:
//        * input : 256-byte chunks, one 4-byte load per lane, written below the bytes the decoder is reading;
//        * output: record i of a block = the 4 states (as cell addresses) iteration i decoded FROM; symbol = symbolOf[state], gathered
//                  from the L2-resident table in global memory (byte table from k_fse_dbuild, or the symbol bytes of the
//                  reference-layout cells: stride 1 << symShift), packed, stored as one 256-byte row per 64 records.
#define FSE_SRV_G 4
#ifndef FSE_FLUSH_MIN
#define FSE_FLUSH_MIN 32u        // records (of 4 symbols) a block must have before its row is flushed
#endif
static_assert(FSE_SRV_WAVES * FSE_SRV_G == FSE_MAXG, "every block of a workgroup has a service wave");
static_assert((FSE_DEC_RING & (FSE_DEC_RING - 1)) == 0 && FSE_DEC_RING % FSE_CHECK_EVERY == 0, "ring positions wrap by masking; a phase never wraps");
DEV void fse_ring_put(u32* rg, int off, u32 w)
{
    const u32 j = (u32)off & (FSE_IN_RING - 1);
    rg[j >> 2] = w;
    if (j < FSE_IN_MIRROR) rg[(FSE_IN_RING + j) >> 2] = w;
}
// rev: the ring is kept in consumption order for the bit-reversed bulk loop -- payload dword at offset o goes, bit-reversed,
// to ring offset (Stop - 4 - o) mod FSE_IN_RING (Stop = payload size rounded up to 4).  Either way ring byte x <-> payload
// byte is a bijection on windows of FSE_IN_RING aligned-dword bytes, so the validLo protocol is the same.
DEV void fse_ring_put_rev(u32* rg, int Sg, int off, u32 w) { fse_ring_put(rg, ((Sg + 3) & ~3) - 4 - off, __brev(w)); }
// caller tables: the workgroup's slot of the symbol scratch goes back when the LAST of its waves is through with it -- the service waves
// gather from it during the bulk, the decoder waves' literal tails read it after that (FseCellsRev / FseCellsCompact take their symbols
// there), so every wave of the workgroup reports here once, after its last access (flagsSh[7] counts them), and the last one releases.
DEV void fse_scratch_slot_done(const FseDecArgs& a, u32* flagsSh, int lane)
{
    if (lane == 0 && atomicAdd(&flagsSh[7], 1u) == FSE_SRV_WAVES + FSE_DEC_WAVES - 1) {
        const u32 slot = flagsSh[6];
        __hip_atomic_fetch_and(a.slotBitmap + (slot >> 5), ~(1u << (slot & 31u)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#ifndef FSE_SRV_READLANE
#define FSE_SRV_READLANE 1
#endif
DEV u32 srv_rl(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
DEV unsigned long long srv_rl64(unsigned long long v, int l) { return (unsigned long long)srv_rl((u32)v, l) | ((unsigned long long)srv_rl((u32)(v >> 32), l) << 32); }
template <bool TIMED, bool CALLER>
DEV void fse_decode_service(const FseDecArgs& a, u8* ldsb, DecCtl* ctlAll, u32 slotBytes, u32 ringOff, u32 inOff, int lane, int g0, bool rev, u32* flagsSh)
{
    const int myG = g0 + (lane < FSE_SRV_G ? lane : 0);      // lane l of this wave keeps the books of block g0 + l
    DecCtl* const ctl = ctlAll + (myG < a.G ? myG : 0);        // (the control area holds G entries)
    // per-block constants live in the registers of lane l of this wave
    const unsigned long long inBits = ((unsigned long long)ctl->inHi << 32) | ctl->inLo;
    const unsigned long long outBits = ((unsigned long long)ctl->outHi << 32) | ctl->outLo;
    const unsigned long long tabBits = ((unsigned long long)ctl->symHi << 32) | ctl->symLo;
    const int S32 = ctl->S32;
    int validLo = ctl->initValidLo;
    u32 flushed = 0, fpos = 0;                               // records flushed so far, and that count modulo the ring size
    bool live = lane < FSE_SRV_G && myG < a.G && !(ctl->pubPofs >> 31);   // blocks that never enter the bulk loop need no service

    // Input refills: the 16-lane group k of this wave serves block g0 + k, so one load instruction and one LDS store
    // refill all blocks of the wave that ask for it in the same round.
    static_assert(64 / FSE_IN_LANES == FSE_SRV_G, "one 16-lane group per block of a service wave");
    const int grp = lane / FSE_IN_LANES, sub = lane % FSE_IN_LANES;
    const int SgK = __shfl(S32, grp, WAVE);
    const gbl_u8_ptr igK = (gbl_u8_ptr)(uintptr_t)__shfl(inBits, grp, WAVE);
    u32* const rgK = (u32*)(ldsb + (size_t)(g0 + grp) * slotBytes + inOff);
    // initial fill: the whole ring of every live block (the topmost dword may straddle the end of the payload), then publish
    {   const bool liveK = (__ballot(live) >> grp) & 1ull;
        const int vlo = __shfl(validLo, grp, WAVE);
#pragma unroll
        for (int c = 0; c < FSE_IN_RING / FSE_IN_CHUNK; ++c) {
            const int off = liveK ? vlo + FSE_IN_CHUNK * c + 4 * sub : -1;
            if (off >= 0 && off + 4 <= SgK) { const u32 w = gbl_load_u32(igK + off); if (rev) fse_ring_put_rev(rgK, SgK, off, w); else fse_ring_put(rgK, off, w); }
            else if (off >= 0 && off < SgK) {
                u32 w = 0;
                for (int i = 0; i < 3; ++i) if (off + i < SgK) w |= (u32)igK[off + i] << (8 * i);
                if (rev) fse_ring_put_rev(rgK, SgK, off, w); else fse_ring_put(rgK, off, w);
            }
        }
        if (live) ctl_store(&ctl->srvValidLo, validLo);
    }

    // The chunk a block will ask for next is requested AHEAD (the input is read-only, its position is known): when the decoder has moved
    // far enough the refill is an LDS store of data that arrived long ago, and a round that only refills does not wait for memory
    // (decoder-wave waiting 4.2 -> 0.9 % of its time; decode call per 100k blocks P14 10.52 -> 10.40 ms, P02 12.95 -> 12.30).
    int nextOff = __shfl(validLo, grp, WAVE) - FSE_IN_CHUNK + 4 * sub;     // my dword of my block's next chunk [validLo - CHUNK, validLo)
    u32 pend = 0;
    if (((__ballot(live) >> grp) & 1ull) && nextOff >= 0 && nextOff + 4 <= SgK) pend = gbl_load_u32(igK + nextOff);
    u32 yq[FSE_SRV_G][4];
#pragma unroll
    for (int l = 0; l < FSE_SRV_G; ++l) { yq[l][0] = yq[l][1] = yq[l][2] = yq[l][3] = 0; }
    unsigned long long sBusy = 0, sIdle = 0, nBusy = 0, sA = 0;
    (void)sBusy; (void)sIdle; (void)nBusy; (void)sA;
    TIMING(sA = __builtin_readcyclecounter(););
    for (;;) {
        // snapshot of the decoder's progress (the finished flag is read before the iteration count it guards)
        u32 pp = 0x80000000u, it = flushed;
        if (live) { const u64 pub = ctl_load2(&ctl->pubIters); it = (u32)pub; pp = (u32)(pub >> 32); }     // (published together: the count is final when the flag is set)
        const bool fin = (pp >> 31) != 0;
        const int P = (int)(pp & 0x7FFFFFFFu);               // byte offset of the topmost dword the decoder still reads
        const u32 avail = it - flushed;
        const bool wantFlush = live && (avail >= FSE_FLUSH_MIN || (fin && avail > 0));
        // the chunk [validLo-CHUNK, validLo) lands on the ring bytes of [validLo+RING-CHUNK, validLo+RING): the decoder must be below
        const bool wantFill = live && !fin && validLo > 0 && P + 4 <= validLo + (FSE_IN_RING - FSE_IN_CHUNK);
        const unsigned long long fm = __ballot(wantFlush), rm = __ballot(wantFill);
        if (live && fin && avail == 0) live = false;
        if (!(fm | rm)) {
            if (!__any(live)) break;
            __builtin_amdgcn_s_sleep(4);
            TIMING(const unsigned long long sB = __builtin_readcyclecounter(); sIdle += sB - sA; sA = sB;);
            continue;
        }
        // (1) install the input chunks that are asked for (requested ahead: see above), publish them, request the ones below them
        const bool fillK = (rm >> grp) & 1ull;
        if (rm) {                                            // uniform
            if (fillK) { if (rev) fse_ring_put_rev(rgK, SgK, nextOff, pend); else fse_ring_put(rgK, nextOff, pend); }
            if (wantFill) { validLo -= FSE_IN_CHUNK; ctl_store(&ctl->srvValidLo, validLo); }
            if (fillK) {
                nextOff -= FSE_IN_CHUNK;
                pend = 0;
                if (nextOff >= 0 && nextOff + 4 <= SgK) pend = gbl_load_u32(igK + nextOff);
            }
        }
        // (2) issue the symbol gathers of every block with enough records
        // (FSE_SRV_READLANE: the per-block values of lane l reach the wave through v_readlane -- the lane index is a constant of the unrolled loop --
        //  instead of __shfl = ds_bpermute: eight LDS-pipe instructions per flushed block were about half of all LDS instructions of the CU)
#pragma unroll
        for (int l = 0; l < FSE_SRV_G; ++l) {
            if (!((fm >> l) & 1ull)) continue;               // uniform
#if FSE_SRV_READLANE
            const u32 cnt = srv_rl(avail, l), fp_g = srv_rl(fpos, l);
            const gbl_u8_ptr tg = (gbl_u8_ptr)(uintptr_t)srv_rl64(tabBits, l);
#else
            const u32 cnt = (u32)__shfl((int)avail, l, WAVE), fp_g = (u32)__shfl((int)fpos, l, WAVE);
            const gbl_u8_ptr tg = (gbl_u8_ptr)(uintptr_t)__shfl(tabBits, l, WAVE);
#endif
            if ((u32)lane < cnt) {
                // iteration i lives in slot pair (i >> 1): 16 bytes = lane A's {iteration 2p, 2p+1} words, then lane B's
                u32 ri = fp_g + (u32)lane;
                ri = ri >= FSE_DEC_RING ? ri - FSE_DEC_RING : ri;
                const u32* const rw = (const u32*)(ldsb + (size_t)(g0 + l) * slotBytes + ringOff) + 4u * (ri >> 1) + (ri & 1u);
                uint2 rec; rec.x = rw[0]; rec.y = rw[2];      // x: state 1 before symbols 0 / 2, y: state 2 before symbols 1 / 3
                // a record holds the low 16 bits of 4 cell addresses; the tables are table-size aligned: state = address bits [1, 1+ldsLog)
                const u32 x0 = __builtin_amdgcn_ubfe(rec.x, 1u, a.ldsLog), x1 = __builtin_amdgcn_ubfe(rec.x, 17u, a.ldsLog);
                const u32 x2 = __builtin_amdgcn_ubfe(rec.y, 1u, a.ldsLog), x3 = __builtin_amdgcn_ubfe(rec.y, 17u, a.ldsLog);
#if FSE_SYM_L1
                yq[l][0] = tg[x0]; yq[l][2] = tg[x1]; yq[l][1] = tg[x2]; yq[l][3] = tg[x3];
#else
                if (CALLER) {   // read at the L2
                    yq[l][0] = __hip_atomic_load(tg + x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); yq[l][2] = __hip_atomic_load(tg + x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    yq[l][1] = __hip_atomic_load(tg + x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); yq[l][3] = __hip_atomic_load(tg + x3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else { yq[l][0] = tg[x0]; yq[l][2] = tg[x1]; yq[l][1] = tg[x2]; yq[l][3] = tg[x3]; }
#endif
            }
        }
        // (4) the ring records are in registers now: hand the slots back, then pack and store the symbols
        if (wantFlush) { ctl_store(&ctl->srvFlushed, it); }
#pragma unroll
        for (int l = 0; l < FSE_SRV_G; ++l) {
            if (!((fm >> l) & 1ull)) continue;               // uniform
#if FSE_SRV_READLANE
            const u32 cnt = srv_rl(avail, l), fl_g = srv_rl(flushed, l);
            gbl_u8_w_ptr const og = (gbl_u8_w_ptr)(uintptr_t)(srv_rl64(outBits, l) + 4ull * fl_g);
#else
            const u32 cnt = (u32)__shfl((int)avail, l, WAVE), fl_g = (u32)__shfl((int)flushed, l, WAVE);
            gbl_u8_w_ptr const og = (gbl_u8_w_ptr)(uintptr_t)(__shfl(outBits, l, WAVE) + 4ull * fl_g);
#endif
            if ((u32)lane < cnt) {
                const u32 w = yq[l][0] | (yq[l][1] << 8) | (yq[l][2] << 16) | (yq[l][3] << 24);
                gbl_store_u32(og + 4u * lane, w);
            }
        }
        if (wantFlush) { fpos += it - flushed; fpos = fpos >= FSE_DEC_RING ? fpos - FSE_DEC_RING : fpos; flushed = it; }
        TIMING(const unsigned long long sB = __builtin_readcyclecounter(); sBusy += sB - sA; sA = sB; ++nBusy;);
    }
    TIMING(if (lane == 0 && g0 == 0) { atomicAdd(&g_decTiming[5], sBusy); atomicAdd(&g_decTiming[6], sIdle); });
    if (CALLER) fse_scratch_slot_done(a, flagsSh, lane);
}

// cell access of the literal path: reference-layout cells in global memory, or LDS cells + global symbol bytes
struct FseCellsRef { const u32* cells;
    DEV void get(u32 st, u32& ns, u32& nb, u32& sym) const { const u32 c = cells[st]; ns = c & 0xFFFFu; sym = (c >> 16) & 0xFFu; nb = c >> 24; } };
struct FseCellsCompact { const u16* A; const u8* syms;
    DEV void get(u32 st, u32& ns, u32& nb, u32& sym) const { const u32 c = A[st]; ns = c & 0xFFFu; nb = c >> 12; sym = syms[st]; } };
struct FseCellsRev { const u16* A; const u8* syms; u32 tl, cellShift;       // bit-reversed tables (see fse_bulk_phase_rev): cell shift = 5
    DEV void get(u32 st, u32& ns, u32& nb, u32& sym) const { const u32 i = __brev(st) >> (32u - tl); const u32 c = A[i]; nb = c & 31u; ns = __brev(c >> cellShift) >> (32u - tl); sym = syms[i]; } };
template <class Cells>
DEV u32 fse_tail_step(const Cells& t, u32& state, BitReader& r, bool fast)          // FSE_decodeSymbol(Fast), fse.h:600-622
{
    u32 ns, nb, sym;
    t.get(state, ns, nb, sym);
    const u32 low = fast ? r.read_fast(nb) : r.read(nb);
    state = ns + low;
    return sym;
}
template <class Cells>
DEV size_t fse_tail(const Cells& t, u32 s1, u32 s2, BitReader& r, u8* out, long op, long omax, bool fast)
{
    for (;;) {                                                   // remaining iterations of fse_decompress.c:201-218
        const int st = r.reload();
        if (!((st == BR_UNFINISHED) & (op < omax - 3))) break;
        u32 w = fse_tail_step(t, s1, r, fast);                     // four symbols, one store
        w |= fse_tail_step(t, s2, r, fast) << 8;
        w |= fse_tail_step(t, s1, r, fast) << 16;
        w |= fse_tail_step(t, s2, r, fast) << 24;
        __builtin_memcpy(out + op, &w, 4);
        op += 4;
    }
    for (;;) {                                                   // :222-235
        if (op > omax - 2) return FERR(dstSize_tooSmall);
        out[op++] = (u8)fse_tail_step(t, s1, r, fast);
        if (r.reload() == BR_OVERFLOW) { out[op++] = (u8)fse_tail_step(t, s2, r, fast); return (size_t)op; }
        if (op > omax - 2) return FERR(dstSize_tooSmall);
        out[op++] = (u8)fse_tail_step(t, s2, r, fast);
        if (r.reload() == BR_OVERFLOW) { out[op++] = (u8)fse_tail_step(t, s1, r, fast); return (size_t)op; }
    }
}

// Entry `pos` of a launch's block list.  The one-shot path hands over a class's FSE_DBINS size-bin lists (internal.h), walked one
// after the other as if they were one list sorted by compressed size: a workgroup lasts as long as its slowest block, and the pace
// of a block follows its input rate.
// LDS: G tables A[2^ldsLog] (u16) on table-size aligned addresses | DecCtl[G] | per block: state ring
// (FSE_DEC_RING x 8 B), input ring (256 + 16 B) | two flag words
template <bool FAST, bool TIMED, bool CALLER>
__global__ __launch_bounds__(FSE_DEC_THREADS) void k_fse_decode(FseDecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long tBorn = 0, wBorn = 0;
    (void)tBorn; (void)wBorn;
    TIMING(tBorn = __builtin_readcyclecounter(); wBorn = wall_clock64(););
    // slot g of this workgroup = entry first + g of the launch's block list (or simply block first + g)
    const size_t first = (size_t)blockIdx.x * a.G;
    // The one-shot path hands over a class's FSE_DBINS size-bin lists (internal.h), walked one after the other as if they were one
    // list sorted by compressed size.  Their lengths come in one 16-byte load; where the workgroup's first slot falls is uniform, and
    // its other slots are nearly always in the same bin.
    // (uniform: where the workgroup's first slot falls -- bin0, entry q0 of it, cnt0 entries in that bin -- is found once, from the bin lengths
    //  loaded together (16-byte loads, one memory latency); the workgroup's other slots are nearly always in the same bin)
    static_assert(FSE_DBINS % 4 == 0, "the bin lengths are read as 16-byte vectors");
    size_t nTot = a.nBlocks, q0 = first;
    u32 bin0 = 0, cnt0 = 0xFFFFFFFFu;
    if (a.count) {
        uint4 cv[FSE_DBINS / 4];
#pragma unroll
        for (u32 i = 0; i < FSE_DBINS / 4; ++i) cv[i] = ((const uint4*)a.count)[i];
        u64 before = 0;
        bool found = false;
#pragma unroll
        for (u32 i = 0; i < FSE_DBINS; ++i) {
            const uint4 v = cv[i >> 2];
            const u32 c = (i & 3u) == 0 ? v.x : (i & 3u) == 1 ? v.y : (i & 3u) == 2 ? v.z : v.w;
            const bool here = !found & (((u64)first < before + c) | (i == FSE_DBINS - 1));
            bin0 = here ? i : bin0; q0 = here ? (size_t)((u64)first - before) : q0; cnt0 = here ? c : cnt0;
            found |= here;
            before += c;
        }
        nTot = (size_t)before;
    }
    if (first >= nTot) return;                                   // uniform: the grid is sized for the worst case
    auto slotBlock = [&](size_t g) -> size_t {                   // block of slot g of this workgroup (first + g < nTot)
        if (!a.list) return first + g;
        size_t q = q0 + g; u32 i = bin0;
        if (q >= cnt0) {                                         // (rare: the workgroup straddles bins)
            while (i < FSE_DBINS - 1) { const u32 c = a.count[i]; if (q < c) break; q -= c; ++i; }
        }
        return a.list[(size_t)i * a.nBlocks + q];
    };
    u8* const lds8 = (u8*)lds;
    const u32 tabStride = 2u << a.ldsLog;                        // bytes per LDS table slot
    DecCtl* const ctlAll = (DecCtl*)(lds8 + (size_t)a.G * tabStride);
    u8* const ldsb = (u8*)ctlAll + (size_t)a.G * sizeof(DecCtl); // ring area: one slot of slotBytes per block
    const u32 slotBytes = a.slotU32 * 4u;                        // multiple of 8
    const u32 ringOff = 0;                                       // state ring offset inside a slot
    const u32 inOff = FSE_DEC_RING * 8;                          // input ring offset inside a slot

    // ---- stage: reference cells {u16 newState; u8 symbol; u8 nbBits} -> compact u16 (uniform control flow, both waves).
    //      A table whose fields do not fit 12+4 bits (cannot come from FSE_buildDTable) is flagged and decoded
    //      by the literal path only.
    u32* const flagsSh = (u32*)(ldsb + (size_t)a.G * slotBytes);  // behind the slots: [0] any nbBits == 0, [1] some table is this launch's (caller tables), [2..3] bad-table mask, [4..5] decline mask,
                                                                  // [6] slot of the symbol scratch (caller tables), [7] service waves finished
    if (tid < 6 || tid == 7) flagsSh[tid] = 0;
    const bool decWave = wave < FSE_DEC_WAVES;
    u8* const tlSh = (u8*)(flagsSh + 8);                                  // caller tables: table log of every slot's table, 0xFF = not this launch's
    if (CALLER) {
        // Caller tables come without a workspace, and the symbols must not be gathered from the reference cells where they lie: 8 KiB per
        // block, 33 blocks per CU, 32 CUs per L2 -- 8.6 MB of tables in flight on a 4 MB L2 (measured: 20.5 ms per 100k blocks against
        // 12.2 with the one-shot path's byte tables).  The staging pass below therefore writes the symbols of the workgroup's tables
        // into one slot of a small scratch the LIBRARY keeps per device (FseDecArgs::symScratch: 2 x CUs slots of 72 KB, allocated at the
        // first such call); a slot is claimed here and handed back by the last service wave.  At most one workgroup of this kernel is
        // resident per CU (160 KB of LDS), so with 2 x CUs slots the search below finds a free one at once.
        // Meanwhile wave 1 looks at the headers of the workgroup's tables: which of them are this launch's (FseDecArgs: tlMin, ldsLog,
        // onlyDeclined)?  A workgroup without any returns at once -- the launches for the classes a batch does not contain cost next to nothing.
        __syncthreads();                                                 // (the flag words are zero)
        bool mine = false;
        if (tid >= 64 && tid < 64 + a.G) {
            const size_t g = (size_t)tid - 64;
            u32 tl = 0xFFu;
            if (first + g < nTot) {
                const size_t bi = slotBlock(g);
                if (!(a.onlyDeclined && a.results[bi] != FSE_DECLINED)) {
                    const u32 t0 = a.dtables[bi * a.dtStrideU32] & 0xFFFFu;
                    if (t0 <= a.ldsLog && t0 >= a.tlMin) tl = t0;
                    else if (t0 > a.maxTableLog && a.tlMin == 0 && !a.onlyDeclined) a.results[bi] = FERR(tableLog_tooLarge);   // (reported by the first launch)
                }
            }
            tlSh[g] = (u8)tl;
            mine = tl != 0xFFu;
        }
        if (mine) flagsSh[1] = 1u;                                       // (plain stores of the same value; no __syncthreads_or: it brings static LDS)
        u32 slot = 0;
        if (tid == 0) {
            slot = (u32)blockIdx.x % a.nSlots;
            for (;;) {
                const u32 bit = 1u << (slot & 31u);
                const u32 old = __hip_atomic_fetch_or(a.slotBitmap + (slot >> 5), bit, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (!(old & bit)) break;
                slot = slot + 1 == a.nSlots ? 0 : slot + 1;
            }
            flagsSh[6] = slot;
#if FSE_SYM_L1
            // the slot may have been used from this CU before (by a workgroup long gone): whatever the CU's vector cache still holds of it
            // is dropped before this workgroup writes and reads it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        }
        __syncthreads();
        if (!flagsSh[1]) {                                               // uniform: nothing of this launch's here
            if (tid == 0) __hip_atomic_fetch_and(a.slotBitmap + (slot >> 5), ~(1u << (slot & 31u)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    else __syncthreads();
    u8* const symSlot = CALLER ? a.symScratch + (size_t)flagsSh[6] * a.scratchSlotBytes : nullptr;   // table g's symbols: symSlot + (g << ldsLog), in the order of its LDS cells
    {   u64 badBits = 0, declBits = 0; bool anyNb0 = false;
        if (!CALLER) {
            // k_fse_dbuild output: already in the LDS format; the first tabStride bytes of every block's global table slot are
            // its LDS image.  Every load is issued before the first store (a lone copy loop would pay the memory latency once
            // per table).
            const size_t nTab = nTot - first < (size_t)a.G ? nTot - first : (size_t)a.G;
            if (tabStride >= 1024u) {
                // LDS-DMA: a wave instruction moves 64 x 16 bytes from per-lane global addresses to 1 KiB of consecutive LDS -- no
                // registers, no LDS store pass; the pieces of all tables are dealt round the waves of the workgroup and are in
                // flight together (drained by the s_waitcnt vmcnt(0) in front of the barrier below)
                const u32 ppt = tabStride >> 10;                                 // 1 KiB pieces per table
                const u32 nPieces = (u32)nTab * ppt;
                for (u32 p = (u32)wave; p < nPieces; p += FSE_DEC_THREADS / 64) {
                    const u32 g = p / ppt, k = p - g * ppt;
                    const size_t bi = slotBlock(g);
                    const u8* const src = (const u8*)(a.atab + (bi << a.maxTableLog)) + 1024u * k + 16u * (u32)lane;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(lds8 + 1024u * p), 16, 0, 0);
                }
            }
            else {
            uint4* const dstv = (uint4*)lds8;
            const u32 vptLog = a.ldsLog - 3u;                            // 16-byte vectors per table: tabStride / 16
            const u32 nvec = (u32)nTab << vptLog;
            constexpr u32 MAXV = (FSE_MAXG * 512u / 16u + FSE_DEC_THREADS - 1) / FSE_DEC_THREADS;      // (tables of 512 bytes at most here)
            uint4 buf[MAXV];
#pragma unroll
            for (u32 k = 0; k < MAXV; ++k) {
                const u32 idx = tid + k * FSE_DEC_THREADS;
                if (idx < nvec) {
                    const size_t sl = first + (idx >> vptLog);
                    const size_t bi = slotBlock(sl - first);
                    buf[k] = ((const uint4*)(a.atab + (bi << a.maxTableLog)))[idx & ((1u << vptLog) - 1u)];
                }
            }
#pragma unroll
            for (u32 k = 0; k < MAXV; ++k) { const u32 idx = tid + k * FSE_DEC_THREADS; if (idx < nvec) dstv[idx] = buf[k]; }
            }
            if (!FAST) for (size_t g = 0; g < nTab; ++g) {                 // a cell with nbBits == 0 needs a counter > tableSize/2
                const u32 st = a.meta[slotBlock(g)].state; anyNb0 |= st != 0 && !(st & 2u); }
        }
        else {
            // Caller-built reference-layout tables {u16 newState; u8 symbol; u8 nbBits} (lib/fse.h:570-575), taken straight from global
            // memory: FAST stages them as the bit-reversed cells of fse_bulk_phase_rev -- cell rev(x) = nbBits | rev(newState) << 5 --
            // the plain loop as newState | nbBits << 12.  Every table is checked while it passes through: what FSE_buildDTable
            // guarantees (newState below the table size and a multiple of 1 << nbBits, nbBits <= tableLog: lib/fse_decompress.c:
            // 113-124) is what the address arithmetic of the bulk loops rests on; a table that fails it is decoded by the literal
            // path alone (bad mask), which does with it whatever the reference does.  A table of log 12 with a cell of nbBits 0
            // (rev(newState) then needs 12 bits) is the plain loop's: marked in the decline mask when this launch may hand it on.
            // Every service wave takes whole tables (wave w: tables w, w + 9, ...): lane l loads the cell quads l, l + 64, ... of the table
            // (16-byte loads; the table starts one word into its slot, which dwordx4 loads do not mind), all before it converts the first.
            // The symbols go to the scratch slot in the order of the LDS cells, so that the service loop gathers them by the record's
            // cell index as it does on the one-shot path (index arithmetic there is paid by every flush: measured +5 % rounds).  For the
            // bit-reversed cells that order is scattered -- as scattered BYTE stores to global memory it cost 100k cycles per workgroup
            // (67k write requests at the L2's request rate) -- so the wave first lays the symbol bytes out in the table's own LDS slot,
            // copies the image to the scratch slot with coalesced 16-byte stores, and only then writes the cells over it (they wait
            // in registers; LDS operations of one wave execute in order).  No workgroup barrier inside: the decoder waves are at their
            // readers' set-up below meanwhile.
            const size_t nTab = nTot - first < (size_t)a.G ? nTot - first : (size_t)a.G;
            constexpr u32 QL = 2048u / 4u / 64u;                             // cell quads per lane and pass: a table of 2048 cells is one pass
            if (!decWave) for (u32 g = (u32)wave - FSE_DEC_WAVES; g < (u32)nTab; g += FSE_SRV_WAVES) {
                const u32 tl = tlSh[g];
                if (tl == 0xFFu) continue;                                // uniform per wave: not this launch's
                const u32 ts = 1u << tl;
                const u32* const t = a.dtables + slotBlock(g) * a.dtStrideU32;
                u8* const slotB = lds8 + (size_t)g * tabStride;
                u16* const A = (u16*)slotB;
                u8* const symOut = symSlot + ((size_t)g << a.ldsLog);
                bool bad = false, nbz = false;
                if (FAST && tl >= 4u) {
                    // the common shape, written for instruction count (a lone wave issues an instruction per ~8 cycles) and for registers
                    // (eight quads per lane in flight; a table of 4096 cells takes two passes and reads its cells a second time -- from the
                    // L2 -- when it writes them): the four cells of a quad sit at rev(i0) | rev2(q) << (tl - 2); the checks are OR-ed into
                    // one word and looked at once
                    const u32 sR = 32u - tl;
                    const u32 qo[4] = { 0u, 1u << (tl - 1u), 1u << (tl - 2u), 3u << (tl - 2u) };
                    const u32 passes = ts > 2048u ? ts / 2048u : 1u;
                    u32 acc = 0, minNb = 15u;
                    uint4 cv[QL];
                    for (u32 h = 0; h < passes; ++h) {
#pragma unroll
                        for (u32 j = 0; j < QL; ++j) {
                            const u32 i0 = 2048u * h + 4u * ((u32)lane + 64u * j);
                            cv[j] = make_uint4(0, 0, 0, 0);
                            if (i0 < ts) __builtin_memcpy(&cv[j], t + 1 + i0, 16);
                        }
#pragma unroll
                        for (u32 j = 0; j < QL; ++j) {
                            const u32 i0 = 2048u * h + 4u * ((u32)lane + 64u * j);
                            if (i0 >= ts) continue;
                            const u32 r0 = __brev(i0) >> sR;
                            const u32 cw[4] = { cv[j].x, cv[j].y, cv[j].z, cv[j].w };
#pragma unroll
                            for (u32 q = 0; q < 4; ++q) {
                                const u32 c = cw[q], ns = c & 0xFFFFu, nb = c >> 24;
                                acc |= (ns >> tl) | ((tl - nb) >> 5) | (ns & ((1u << (nb & 31u)) - 1u));     // newState beyond the table / nbBits beyond tableLog / newState not a multiple of 1 << nbBits
                                minNb = nb < minNb ? nb : minNb;
                                slotB[r0 | qo[q]] = (u8)(c >> 16);
                            }
                        }
                    }
                    { for (u32 off = 16u * (u32)lane; off < ts; off += 1024u) { const uint4 v = *(const uint4*)(slotB + off); *(uint4*)(symOut + off) = v; } }
                    for (u32 h = 0; h < passes; ++h) {
                        if (passes > 1u) {
#pragma unroll
                            for (u32 j = 0; j < QL; ++j) { const u32 i0 = 2048u * h + 4u * ((u32)lane + 64u * j); __builtin_memcpy(&cv[j], t + 1 + i0, 16); }
                        }
#pragma unroll
                        for (u32 j = 0; j < QL; ++j) {
                            const u32 i0 = 2048u * h + 4u * ((u32)lane + 64u * j);
                            if (i0 >= ts) continue;
                            const u32 r0 = __brev(i0) >> sR;
                            const u32 cw[4] = { cv[j].x, cv[j].y, cv[j].z, cv[j].w };
#pragma unroll
                            for (u32 q = 0; q < 4; ++q) {
                                const u32 c = cw[q];
                                A[r0 | qo[q]] = (u16)(((__brev(c << 16) >> (16u - tl)) << 5) | ((c >> 24) & 31u));   // (rev over 32 bits of newState << 16 = its 16-bit reversal; down to tl bits)
                            }
                        }
                    }
                    bad = acc != 0; nbz = minNb == 0;
                } else {
                    // tables of fewer than 16 cells, and the plain cells of the register-window loop (a tableLog-12 table with a cell of
                    // nbBits 0): cell by cell; the symbols of the plain cells sit in state order anyway
#pragma unroll 2
                    for (u32 i = (u32)lane; i < ts; i += 64u) {
                        const u32 c = t[1 + i];
                        const u32 ns = c & 0xFFFFu, nb = c >> 24;
                        bad |= (ns >> tl) != 0 || nb > tl || nb > 15u || (ns & ((1u << (nb & 15u)) - 1u)) != 0;
                        nbz |= nb == 0;
                        const u32 pos = FAST ? __brev(i) >> (32u - tl) : i;
                        symOut[pos] = (u8)(c >> 16);
                        if (FAST) A[pos] = (u16)((nb & 31u) | ((__brev(ns & ((1u << tl) - 1u)) >> (32u - tl)) << 5));
                        else A[pos] = (u16)((ns & 0xFFFu) | (nb << 12));
                    }
                }
                if (__any(bad)) badBits |= 1ull << g;
                if (FAST && tl == 12u && __any(nbz)) declBits |= 1ull << g;
                anyNb0 |= __any(nbz) != 0;
            }
        }
        if (badBits) { atomicOr(&flagsSh[2], (u32)badBits); atomicOr(&flagsSh[3], (u32)(badBits >> 32)); }
        if (declBits) { atomicOr(&flagsSh[4], (u32)declBits); atomicOr(&flagsSh[5], (u32)(declBits >> 32)); }
        if (anyNb0) atomicOr(&flagsSh[0], 1u);
    }
    // ---- per-block set-up by the decoder waves, BEFORE the staged tables are waited for (the readers' first loads overlap the staging): lanes 2g and 2g+1 walk block first+g together and both run this set-up
    //      (identical values in both; only the even lane publishes, finishes the block and writes its result)
    //      (decoder wave w takes the slots [w * ppw, (w+1) * ppw))
    const int ppw = (a.G + FSE_DEC_WAVES - 1) / FSE_DEC_WAVES;
    const int gsl = (decWave ? wave * ppw : 0) + (lane >> 1);
    const u32 half = (u32)lane & 1u, maskB = half ? ~0u : 0u;
    const bool inRange = (lane >> 1) < ppw && gsl < a.G && first + (size_t)gsl < nTot;
    const size_t b = inRange ? slotBlock((size_t)gsl) : 0;
    bool owner = decWave && inRange;
    u32 hdr = 0;
    if (owner && a.meta) { if (a.meta[b].state == 0) owner = false; else hdr = a.meta[b].hdrSize; }
    u32 tl = 0; bool fast = false;
    constexpr bool compact = !CALLER;
    const u32* const gtab = compact ? nullptr : a.dtables + (owner ? b : 0) * a.dtStrideU32;   // reference-layout table in global memory
    if (owner) {
        if (compact) { tl = a.meta[b].tableLog; fast = (a.meta[b].state & 2u) != 0; if (tl > a.ldsLog) { a.results[b] = FERR(tableLog_tooLarge); owner = false; } }
        else {
            // caller tables: one launch per class over all blocks (FseDecArgs)
            const u32 h0 = gtab[0]; tl = h0 & 0xFFFFu; fast = (h0 >> 16) != 0;
            if (tlSh[gsl] == 0xFFu) owner = false;                       // not this launch's (see the head of the kernel)
        }
    }
    const u32* const cells = compact ? nullptr : gtab + 1;      // literal path: reference cells, or the LDS cells + symbol table
    const u8* const syms = compact ? a.symtab + ((owner ? b : 0) << a.maxTableLog) : symSlot + ((size_t)(gsl < a.G ? gsl : 0) << a.ldsLog);
    // absolute LDS byte address of my table: the dynamic LDS segment starts at 0 (no static LDS in this kernel), which the
    // table-size alignment of the FAST address arithmetic relies on
    const u32 ldsBase = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)lds8;
    if (ldsBase & (tabStride - 1)) __builtin_trap();
    const u32 tabOff = ldsBase + (u32)(gsl < a.G ? gsl : 0) * tabStride;
    const u16* const A = (const u16*)(lds8 + (tabOff - ldsBase));
    const u8* in = nullptr; size_t S = 0; u8* out = nullptr;
    const long omax = (long)a.dstCapacity;
    long op = 0;
    BitReader r; r.base = nullptr; r.size = 0; r.at = 0; r.win = 0; r.used = 0;
    u32 s1 = 0, s2 = 0;
    if (owner) {
        in = view_ptr(a.csrc, b) + hdr;
        S = view_size(a.csrc, b) - hdr;              // hdr <= cSrcSize (FSE_readNCount never returns more)
        out = a.dst + b * a.dstStride;
        const size_t e = r.init(in, S);
        if (is_err(e)) { a.results[b] = e; owner = false; }
        else {
            s1 = r.read(tl); r.reload();             // FSE_initDState x2, fse.h:577-584
            s2 = r.read(tl); r.reload();
        }
    }

    __builtin_amdgcn_s_waitcnt(0x0f70);             // vmcnt(0): the LDS-DMA pieces have landed (lgkmcnt / expcnt fields left at their maxima)
    __syncthreads();
    const u64 badMask = (u64)flagsSh[2] | ((u64)flagsSh[3] << 32);
    const u64 declMask = (u64)flagsSh[4] | ((u64)flagsSh[5] << 32);
    const bool nb0 = flagsSh[0] != 0;
    // a launch that may hand blocks on marks the tables it leaves to the plain-cell launch (tableLog 12 with a cell of nbBits 0)
    if (CALLER && owner && a.declineNb0 && ((declMask >> gsl) & 1ull)) { if (half == 0) a.results[b] = FSE_DECLINED; owner = false; }

    // ---- bulk: iterations of fse_decompress.c:201-218 whose loop-head reload is provably the fast one.
    //      The decoder lane touches only registers and LDS:
    //        * input: the 8 bytes below the window come from this block's LDS input ring;
    //        * output: each iteration appends the 4 states it decoded FROM to the block's state ring.
    //      The service wave (above) owns all global-memory traffic of the bulk loop.
    // Phase structure: a lane runs a phase only if FSE_CHECK_EVERY more iterations are certainly valid for it
    // (>= 16 output groups left and the window stays >= 24 bytes above the stream start: at >= 24 + 16*6), so the
    // 16 iterations of a phase run without any per-iteration bookkeeping; whatever is left goes to the literal tail.
    // Bulk state uses p = at+1, u = used+8 (u in [8,16) after a reload), so no shift amount is ever 0 or 32.
    // Which iterations may the bulk loop take?  A loop-head reload of the reference (lib/bitstream.h:400-439) that does not clamp leaves
    // ptr = ceil(B/8) - 8 and bitsConsumed = 8*(ptr+8) - B, B = the bits unread at that moment, on the fast path (:405-409) and on
    // the slow one (:428-438) alike.  With B' the bits unread at the PREVIOUS reload, it returns "unfinished" -- the loop of
    // lib/fse_decompress.c:201 goes on -- iff the ptr it starts from is not the stream start (ceil(B'/8) - 8 >= 1: B' >= 65) and
    // it need not clamp (ceil(B/8) >= 8: B >= 57); B >= 65 at every loop head is sufficient.  One iteration takes at most
    // 4 * tableLog <= 48 bits, so a phase of N iterations is the reference's loop, iteration for iteration, if B >= 65 + 48*(N-1)
    // at its start (and N groups of four bytes fit the output).  The bit-reversed loop runs phases of FSE_CHECK_EVERY iterations
    // while it can and then FINISHING phases of two (FSE_FINISH_EVERY) down to B < 113: the literal path is left with the last
    // two or three iterations and the reference's end game.  Between two reloads the reader state is (ptr of the last reload,
    // bitsConsumed counted from it): rebuilt below from the cursor at the head of the last iteration and the cursor now.
    // (The plain-cell loop keeps its coarser rule: window at least 24 bytes above the stream start.)
    // unread bits of the payload after the two state reads (a stream of a few bytes has been read beyond its end by now: negative)
    const bool bulkOk = owner && S < (1ull << 28) && !((badMask >> gsl) & 1ull) && !(FAST && !compact && ((declMask >> gsl) & 1ull));
    const int Bstart = bulkOk ? (int)(8u * ((u32)r.at + 8u)) - (int)r.used : 0;
    const long groups0 = (omax - 3 - op + 3) / 4;
    bool can = bulkOk && (FAST ? Bstart >= 65 + 48 * (FSE_CHECK_EVERY - 1) : r.at >= 24 + 6 * FSE_CHECK_EVERY + 8) && groups0 >= FSE_CHECK_EVERY;
    bool can2 = FAST && bulkOk && Bstart >= 65 + 48 * (FSE_FINISH_EVERY - 1) && groups0 >= FSE_FINISH_EVERY;
    const bool everBulk = can || can2;
    BulkState bs; bs.q = 0; bs.bq = 0;
    {   const u32 st = half ? s2 : s1;                                           // my state as a cell address
        bs.s = tabOff + 2u * (FAST ? __brev(st) >> (32u - (tl ? tl : 1u)) : st); }
    // Ring coordinates count bytes from the 4-byte aligned address at or below the payload (inA bytes lower), so that the
    // service's refills are aligned dwords and its 64-byte chunks whole 64-byte sectors of memory: every byte of the stream
    // is fetched exactly once.  "Unread bits" B below therefore includes those 8*inA bits.
    const u32 inA = (u32)((uintptr_t)in & 3u);
    const u32 R8 = 8u * (((u32)S + inA + 3u) & ~3u);                             // bit-reversed loop: cursor P = R8 - unread bits
    u32 P = 0;
    long groups = 0;
    u32 iters = 0;
    int validLo = 0;                                 // ring holds stream bytes [validLo, validLo + FSE_IN_RING)
    if (everBulk) {
        const u32 B = 8u * ((u32)r.at + 8u + inA) - r.used;      // unread bits = bits [0, B) counted from the aligned base
        bs.q = 4u * (B >> 5) - 8u; bs.bq = B & 31u; P = R8 - B;
        groups = (omax - 3 - op + 3) >> 2;
        // P = q + 8 = byte offset of dword dp.  The ring must reach up to P + 4 and down to the lowest byte a phase can
        // read, P - 8 - 6*16 = P - 104.  Chunk boundaries sit on 64-byte aligned addresses (offset c0 in ring coordinates).
        const int c0 = (int)((0 - ((uintptr_t)in - inA)) & (FSE_IN_CHUNK - 1));
        validLo = (((int)bs.q + 8 - 112 - c0) & ~(FSE_IN_CHUNK - 1)) + c0;   // the ring then reaches from below q - 104 up to above q + 12
    }
    DecCtl* const ctl = ctlAll + (gsl < a.G ? gsl : 0);
    if (decWave && (lane >> 1) < ppw && gsl < a.G && half == 0) {
        ctl->pubIters = 0; ctl->pubPofs = everBulk ? bs.q + 8u : 0x80000000u;
        ctl->srvFlushed = 0; ctl->srvValidLo = 0x7FFFFFFF;
        ctl->initValidLo = validLo; ctl->S32 = (int)(S < (1ull << 31) ? S + inA : 0);
        const unsigned long long ib = (unsigned long long)(uintptr_t)(in - inA), ob = (unsigned long long)(uintptr_t)out, tb = (unsigned long long)(uintptr_t)syms;
        ctl->inLo = (u32)ib; ctl->inHi = (u32)(ib >> 32); ctl->outLo = (u32)ob; ctl->outHi = (u32)(ob >> 32); ctl->symLo = (u32)tb; ctl->symHi = (u32)(tb >> 32);
    }
    __syncthreads();
    if (!decWave) { fse_decode_service<TIMED, CALLER>(a, ldsb, ctlAll, slotBytes, ringOff, inOff, lane, (wave - FSE_DEC_WAVES) * FSE_SRV_G, FAST, flagsSh); return; }

    __builtin_amdgcn_s_setprio(3);                   // the decoder wave is the critical path of the workgroup
    uint2* const myRing = (uint2*)(ldsb + (size_t)(gsl < a.G ? gsl : 0) * slotBytes + ringOff) + half;   // my half of every slot pair
    const u32 myIn = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)(ldsb + (size_t)(gsl < a.G ? gsl : 0) * slotBytes + inOff);   // absolute LDS address of my input ring
    unsigned long long tRun = 0, tWait = 0, nRun = 0, nWait = 0, tA = 0, tFin = 0, nFin = 0, tInner = 0;
    (void)tRun; (void)tWait; (void)nRun; (void)nWait; (void)tA; (void)tFin; (void)nFin; (void)tInner;
    TIMING(tA = __builtin_readcyclecounter(););
    const unsigned long long tBulk0 = tA;
    (void)tBulk0;
    // The service's progress words are read one round ahead: the loads issued here are consumed at the top of the next
    // round, so their LDS round trip hides under this round's phase.  Stale values are conservative (srvFlushed only grows,
    // srvValidLo only falls); LDS operations of one wave execute in order, so the ring reads of a phase cannot overtake
    // the progress loads they depend on (the compiler is held back by the barrier in ctl_peek).
    u32 rpos = 0;                                    // iters modulo the ring size
    u64 srvNext = ctl_peek2(&ctl->srvFlushed);       // {srvFlushed, srvValidLo}
    const u32 inA8 = 8u * inA;
    u32 Phead = P;                                   // bit-reversed loop: cursor at the head of the last iteration taken
    // Everything between two phases is paid by all chains of the wave (a wave issues an instruction per ~8 cycles whatever it
    // does), so the round is kept short: one LDS load, one LDS store, 32-bit counters, one wave vote.
    int grp = groups > (1 << 30) ? (1 << 30) : (int)groups;                 // groups of four output bytes that fit (beyond 2^30: the literal path goes on)
    if constexpr (FAST) {
        // ---- the bit-reversed loop's rounds, WAVE-UNIFORM control flow.  With a divergent `if (ready) { phase }` the compiler merges every
        //      loop-carried register of the round through copies on both sides of the branch and juggles exec masks around it: 75
        //      instructions between two phases, 19 of them v_mov (620 of a round's 4190 cycles; a lone wave issues an instruction per ~8).
        //      Here ALL lanes run every phase: a lane whose block cannot take one (finished, or no block at all) decodes on as a "zombie"
        //      -- its table and ring addresses stay inside its own LDS slot whatever bits it reads (the address arithmetic of
        //      fse_bulk_phase_rev), its records go to a dummy ring -- and its real state is kept by selects.  A lane that has to wait for
        //      its service wave (ring full / input not there yet) makes the wave poll instead of sitting a round out: the lagging chain
        //      sets the pace either way, but it no longer costs the others a whole extra round at the end.
        uint2* const dummyRing = (uint2*)(flagsSh + 32) + half;          // 128 bytes behind the flag words: a phase's records of the zombies
        // (the lanes beyond the wave's lane pairs never have a block: they stay out altogether -- one branch for the whole bulk, so that the
        //  LDS instructions of a phase serve 34 lanes, not 64)
        // Likewise the pairs that never run a phase (no block, an early error, a table the staging pass did not vouch for): a zombie is only
        // ever a chain that WAS decoding -- its table is sound, so every address it forms stays inside its slot.
        // (Measured and not kept, round 4: cells with nbBits in the top four bits and rev(newState) << 1 below -- every use of nbBits clean
        //  without masks, the table part of the address one v_and_or: two instructions per iteration fewer, but the shift that extracts
        //  nbBits sits on the dependent chain cell -> offset -> bits -> address: 11.08 instead of 10.76 ms per 100k P14 blocks.)
        if ((lane >> 1) < ppw && everBulk) {
        for (;;) {
            // room for 16 more records, and the lowest byte this phase can read is in the ring: its last iteration starts at most
            // 15 * 48 bits further down (23 dwords) and reads the three dwords from there -> 92 bytes below q.  Nothing below the
            // stream start is ever consumed: once the ring reaches down to it (validLo <= 0) the phase may run.
            const int lowest = (int)bs.q - 4 * ((48 * (FSE_CHECK_EVERY - 1) + 31) / 32);
            const int need = lowest > 0 ? lowest : 0;
#if FSE_PEEK_IN_PHASE
            // srvNext was requested inside the previous phase (fse_bulk_phase_rev) and arrived long ago: the common round looks at it and goes on
            // without an LDS round trip of its own; only a round that has to poll asks again and waits.
            {   bool rdy = !can | ((iters + FSE_CHECK_EVERY - (u32)srvNext <= FSE_DEC_RING) & (need >= (int)(u32)(srvNext >> 32)));
                while (!__all(rdy)) {                                            // uniform: poll until every chain of the wave may run
                    TIMING(const unsigned long long tB = __builtin_readcyclecounter(); tWait += tB - tA; ++nWait; tA = tB;);
                    srvNext = ctl_peek2(&ctl->srvFlushed);
                    rdy = !can | ((iters + FSE_CHECK_EVERY - (u32)srvNext <= FSE_DEC_RING) & (need >= (int)(u32)(srvNext >> 32)));
                }
            }
#else
            bool rdy;
            do {                                                                 // uniform: poll until every chain of the wave may run
                const u32 fl = (u32)srvNext;
                const int vlo = (int)(u32)(srvNext >> 32);
                srvNext = ctl_peek2(&ctl->srvFlushed);
                rdy = !can | ((iters + FSE_CHECK_EVERY - fl <= FSE_DEC_RING) & (need >= vlo));
                TIMING(if (!__all(rdy)) { const unsigned long long tB = __builtin_readcyclecounter(); tWait += tB - tA; ++nWait; tA = tB; });
            } while (!__all(rdy));
#endif
            if (!__any(can)) break;                                              // uniform
            uint2* const ring = can ? myRing + rpos : dummyRing;
            u32 sN = bs.s, Pn = P, PhN = Phead;
            unsigned long long tI = 0; (void)tI;
            TIMING(tI = __builtin_readcyclecounter(););
#if FSE_PEEK_IN_PHASE
            fse_bulk_phase_rev<FSE_CHECK_EVERY>(sN, Pn, PhN, tl + 1u, 4u, tabOff, myIn, maskB & 31u, ring, &ctl->srvFlushed, &srvNext);
#else
            fse_bulk_phase_rev<FSE_CHECK_EVERY>(sN, Pn, PhN, tl + 1u, 4u, tabOff, myIn, maskB & 31u, ring);
#endif
            TIMING(tInner += __builtin_readcyclecounter() - tI;);
            bs.s = can ? sN : bs.s; P = can ? Pn : P; Phead = can ? PhN : Phead;
            const u32 adv = can ? (u32)FSE_CHECK_EVERY : 0u;
            rpos = (rpos + adv) & (FSE_DEC_RING - 1);
            iters += adv; grp -= (int)adv;
            const u32 B = R8 - P;
            bs.q = 4u * (B >> 5) - 8u;
            const u32 Bp = B - inA8;                                          // unread bits of the payload proper
            const bool canN = can & (Bp >= 65u + 48u * (FSE_CHECK_EVERY - 1)) & (grp >= FSE_CHECK_EVERY);
            const bool more = canN | ((Bp >= 65u + 48u * (FSE_FINISH_EVERY - 1)) & (grp >= FSE_FINISH_EVERY));     // finishing phases to come
            if (can & (half == 0)) ctl_store2(&ctl->pubIters, iters, more ? bs.q + 8u : ((bs.q + 8u) | 0x80000000u));
            can = canN;
            TIMING(const unsigned long long tB = __builtin_readcyclecounter(); tRun += tB - tA; ++nRun; tA = tB;);
        }
        {   const u32 B = R8 - P; bs.bq = B & 31u;
            can2 = bulkOk & (B - inA8 >= 65u + 48u * (FSE_FINISH_EVERY - 1)) & (grp >= FSE_FINISH_EVERY); }
        can2 = can2 & bulkOk;
        for (;;) {                                   // ---- finishing phases of FSE_FINISH_EVERY iterations (every lane is through with the long ones)
            const u32 fl = (u32)srvNext;
            const int vlo = (int)(u32)(srvNext >> 32);
#if !FSE_PEEK_IN_PHASE
            srvNext = ctl_peek2(&ctl->srvFlushed);
#endif
            const int lowest = (int)bs.q - 8;                                    // the second iteration's window starts at most two dwords further down
            const bool rdy = !can2 | ((iters + FSE_FINISH_EVERY - fl <= FSE_DEC_RING) & ((lowest > 0 ? lowest : 0) >= vlo));
            if (!__all(rdy)) {
                TIMING(const unsigned long long tB = __builtin_readcyclecounter(); tWait += tB - tA; ++nWait; tA = tB;);
#if FSE_PEEK_IN_PHASE
                srvNext = ctl_peek2(&ctl->srvFlushed);
#endif
                continue;
            }
            if (!__any(can2)) break;
            uint2* const ring = can2 ? myRing + rpos : dummyRing;
            u32 sN = bs.s, Pn = P, PhN = Phead;
#if FSE_PEEK_IN_PHASE
            fse_bulk_phase_rev<FSE_FINISH_EVERY>(sN, Pn, PhN, tl + 1u, 4u, tabOff, myIn, maskB & 31u, ring, &ctl->srvFlushed, &srvNext);
#else
            fse_bulk_phase_rev<FSE_FINISH_EVERY>(sN, Pn, PhN, tl + 1u, 4u, tabOff, myIn, maskB & 31u, ring);
#endif
            bs.s = can2 ? sN : bs.s; P = can2 ? Pn : P; Phead = can2 ? PhN : Phead;
            rpos = can2 ? (rpos + FSE_FINISH_EVERY) & (FSE_DEC_RING - 1) : rpos;
            iters += can2 ? FSE_FINISH_EVERY : 0u; grp -= can2 ? FSE_FINISH_EVERY : 0;
            const u32 B = R8 - P;
            bs.q = 4u * (B >> 5) - 8u; bs.bq = B & 31u;
            const bool can2N = can2 & (B - inA8 >= 65u + 48u * (FSE_FINISH_EVERY - 1)) & (grp >= FSE_FINISH_EVERY);
            if (can2 & (half == 0)) ctl_store2(&ctl->pubIters, iters, can2N ? bs.q + 8u : ((bs.q + 8u) | 0x80000000u));
            can2 = can2N;
            TIMING(const unsigned long long tB = __builtin_readcyclecounter(); tFin += tB - tA; ++nFin; tA = tB;);
        }
        }
    } else {
    while (__any(can)) {                             // ---- phases of FSE_CHECK_EVERY iterations
        const u32 fl = (u32)srvNext;
        const int vlo = (int)(u32)(srvNext >> 32);
        srvNext = ctl_peek2(&ctl->srvFlushed);
        // room for 16 more records, and the lowest byte this phase can read is in the ring: its last iteration starts at most
        // 15 * 48 bits further down (23 dwords) and reads the three dwords from there -> 92 bytes below q (plain loop: 6 bytes per
        // iteration and a window of 8).  Nothing below the stream start is ever consumed: once the ring reaches down to it
        // (validLo <= 0) the phase may run.
        const int lowest = (int)bs.q - (FAST ? 4 * ((48 * (FSE_CHECK_EVERY - 1) + 31) / 32) : 6 * FSE_CHECK_EVERY + 8);
        // (bitwise &: ONE divergent branch per round -- the compiler turns && chains into nested branches with a copy of every
        //  loop-carried register at each level)
        const bool ready = can & (iters + FSE_CHECK_EVERY - fl <= FSE_DEC_RING) & ((lowest > 0 ? lowest : 0) >= vlo);
        if (ready) {
            uint2* const ring = myRing + rpos;                               // 16 consecutive slots: a phase never wraps
            rpos = (rpos + FSE_CHECK_EVERY) & (FSE_DEC_RING - 1);
            if (FAST) {
                unsigned long long tI = 0; (void)tI;
                TIMING(tI = __builtin_readcyclecounter(););
                fse_bulk_phase_rev<FSE_CHECK_EVERY>(bs.s, P, Phead, tl + 1u, 4u, tabOff, myIn, maskB & 31u, ring);
                TIMING(tInner += __builtin_readcyclecounter() - tI;);
                const u32 B = R8 - P;
                bs.q = 4u * (B >> 5) - 8u; bs.bq = B & 31u;
            }
            else if (nb0) fse_bulk_phase<true>(bs.s, bs.q, bs.bq, tabOff, myIn, half, maskB, ring);
            else          fse_bulk_phase<false>(bs.s, bs.q, bs.bq, tabOff, myIn, half, maskB, ring);
            iters += FSE_CHECK_EVERY; grp -= FSE_CHECK_EVERY;
            if (FAST) {
                const u32 Bp = R8 - P - inA8;                                 // unread bits of the payload proper
                can = (Bp >= 65u + 48u * (FSE_CHECK_EVERY - 1)) & (grp >= FSE_CHECK_EVERY);
                can2 = (Bp >= 65u + 48u * (FSE_FINISH_EVERY - 1)) & (grp >= FSE_FINISH_EVERY);
            }
            // plain loop: the reference's ptr offset after its next reload is >= 4*dp - 8 = q: keep 16 more fast reloads certain
            else can = (bs.q >= 24u + 6u * FSE_CHECK_EVERY + 4u) & (grp >= FSE_CHECK_EVERY);     // (+4: q counts from the aligned base)
            if (half == 0) ctl_store2(&ctl->pubIters, iters, (can | can2) ? bs.q + 8u : ((bs.q + 8u) | 0x80000000u));
        }
        TIMING(const unsigned long long tB = __builtin_readcyclecounter(); if (__any(ready)) { tRun += tB - tA; ++nRun; } else { tWait += tB - tA; ++nWait; } tA = tB;);
    }
    if (false) while (__any(can2)) {                  // ---- finishing phases of FSE_FINISH_EVERY iterations (every lane is through with the long ones)
        const u32 fl = (u32)srvNext;
        const int vlo = (int)(u32)(srvNext >> 32);
        srvNext = ctl_peek2(&ctl->srvFlushed);
        const int lowest = (int)bs.q - 8;                                    // the second iteration's window starts at most two dwords further down
        const bool ready = can2 & (iters + FSE_FINISH_EVERY - fl <= FSE_DEC_RING) & ((lowest > 0 ? lowest : 0) >= vlo);
        if (ready) {
            uint2* const ring = myRing + rpos;
            rpos = (rpos + FSE_FINISH_EVERY) & (FSE_DEC_RING - 1);
            fse_bulk_phase_rev<FSE_FINISH_EVERY>(bs.s, P, Phead, tl + 1u, 4u, tabOff, myIn, maskB & 31u, ring);
            const u32 B = R8 - P;
            bs.q = 4u * (B >> 5) - 8u; bs.bq = B & 31u;
            iters += FSE_FINISH_EVERY; grp -= FSE_FINISH_EVERY;
            can2 = (B - inA8 >= 65u + 48u * (FSE_FINISH_EVERY - 1)) & (grp >= FSE_FINISH_EVERY);
            if (half == 0) ctl_store2(&ctl->pubIters, iters, can2 ? bs.q + 8u : ((bs.q + 8u) | 0x80000000u));
        }
        TIMING(const unsigned long long tB = __builtin_readcyclecounter(); if (__any(ready)) { tFin += tB - tA; ++nFin; } else { tWait += tB - tA; ++nWait; } tA = tB;);
    }
    }
    TIMING(if (lane == 0) { atomicAdd(&g_decTiming[0], tRun); atomicAdd(&g_decTiming[1], tWait); atomicAdd(&g_decTiming[2], nRun); atomicAdd(&g_decTiming[3], nWait); atomicAdd(&g_decTiming[4], 1ull);
                            atomicAdd(&g_decTiming[13], tBulk0 - tBorn); atomicAdd(&g_decTiming[15], tFin); atomicAdd(&g_decTiming[10], nFin); atomicAdd(&g_decTiming[7], tInner); });
    const u32 sOther = dpp_swap(bs.s);               // (all lanes of the wave are still here)
    const unsigned long long tBulk1 = tA;
    (void)tBulk1;
    // (the odd lanes and the pairs without a block have nothing left to do; the wave lives until its even lanes are through their tails.
    //  No early return: with caller tables the wave reports below, as a whole, when its last tail has read the symbol scratch.)
    if (owner && !half) {
    op = 4 * (long)iters;
    if (iters) {                                     // back to the reference's (ptr, bitsConsumed, container)
        const u32 B = 8u * (bs.q + 8u) + bs.bq - 8u * inA;      // back to bits of the payload proper
        // bit-reversed loop: ptr is where the reload at the head of the last iteration put it (B there >= 65), bitsConsumed counts from it
        // (up to 7 + 48); plain loop: the state after a reload (it stops where every reload is still the fast one, and a reload of a
        // reloaded reader changes nothing)
        const u32 Bh = FAST ? R8 - Phead - 8u * inA : B;
        r.at = (size_t)((Bh + 7u) >> 3) - 8; r.used = 8u * ((u32)r.at + 8u) - B; r.win = ldg64u(in + r.at); s1 = (bs.s - tabOff) >> 1; s2 = (sOther - tabOff) >> 1;
        if (FAST) { s1 = __brev(s1) >> (32u - tl); s2 = __brev(s2) >> (32u - tl); }
    }

    // ---- literal tail: remaining iterations of :201-218, then :222-235
    size_t result;
    if (!compact) {   // caller tables: the staged cells and the scratch symbols; a table the staging pass did not vouch for: the reference's own cells
        if ((badMask >> gsl) & 1ull) result = fse_tail(FseCellsRef{cells}, s1, s2, r, out, op, omax, fast);
        else if (FAST)               result = fse_tail(FseCellsRev{A, syms, tl, 5u}, s1, s2, r, out, op, omax, fast);
        else                         result = fse_tail(FseCellsCompact{A, syms}, s1, s2, r, out, op, omax, fast);
    }
    else if (FAST)    result = fse_tail(FseCellsRev{A, syms, tl, 5u}, s1, s2, r, out, op, omax, fast);
    else              result = fse_tail(FseCellsCompact{A, syms}, s1, s2, r, out, op, omax, fast);
    a.results[b] = result;
    TIMING(if (lane == 0) { const unsigned long long tE = __builtin_readcyclecounter(); atomicAdd(&g_decTiming[11], tE - tBorn); atomicAdd(&g_decTiming[12], wall_clock64() - wBorn);
                            atomicAdd(&g_decTiming[14], tE - tBulk1); });
    }
    if (CALLER) fse_scratch_slot_done(a, flagsSh, lane);        // (all lanes of the wave are here again: its tails are done)
}

static void fse_decode_geometry(unsigned ldsLog, size_t ldsBytes, unsigned* slotU32, int* G)
{
    *slotU32 = (FSE_DEC_RING * 8 + FSE_IN_RING + FSE_IN_MIRROR + 16) / 4;  // rings (16-byte multiples: records are written in pairs)
    int g = (int)((ldsBytes - 256) / ((2u << ldsLog) + *slotU32 * 4 + sizeof(DecCtl)));   // (256 bytes: the flag words, the caller tables' logs, the zombies' dummy ring)
    if (g > FSE_MAXG) g = FSE_MAXG;
    *G = g;
}
#define FSE_DEC_LDS (FSE_DEC_LDS_KB * 1024)
size_t fse_decode_blocks_per_round(unsigned maxTableLog)
{
    unsigned slot; int G;
    fse_decode_geometry(maxTableLog < FSE_DEC_FAST_MAXLOG ? maxTableLog : FSE_DEC_FAST_MAXLOG, FSE_DEC_LDS, &slot, &G);   // the common class
    const int cus = dev_props().ok ? dev_props().cus : 256;
    return (size_t)G * FSE_WGS_PER_CU * cus;
}

// The library's symbol scratch for caller-built tables (see the slot claim in k_fse_decode): per device, allocated at the first
// FSE_decompress_usingDTable batch call on it and kept for the life of the process -- 2 x CUs slots of FSE_SYM_SLOT_BYTES plus the claim
// bitmap.  (The reference's call takes no workspace, lib/fse.h:247; everything else the batched calls need comes from the caller.)
#define FSE_SYM_SLOT_BYTES (FSE_MAXG * 2048u)            // 33 tables of 2 KiB (table logs up to 11) or 18 of 4 KiB
#include <mutex>
namespace {
std::mutex g_symMutex;
struct SymScratch { u8* base = nullptr; u32* bitmap = nullptr; u32 nSlots = 0; };
SymScratch g_symScratch[64];
}
static hipError_t fse_sym_scratch(FseDecArgs& a, hipStream_t s)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_symMutex);
    SymScratch& sc = g_symScratch[dev];
    if (!sc.base && s) {
        // the first call on a device allocates (hipMalloc + a synchronous hipMemset): not something a stream capture survives.  A caller who
        // captures runs one ordinary call first, as for every batched call (include/fsehip.h, "Streams and graphs"), or FSEHIP_prepareDevice().
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return hipErrorStreamCaptureUnsupported;
        (void)hipGetLastError();
    }
    if (!sc.base) {
        const u32 nSlots = 2u * (u32)(dev_props().ok ? dev_props().cus : 256);
        const size_t bitmapBytes = ((nSlots + 31u) / 32u * 4u + 255u) & ~(size_t)255;
        u8* p = nullptr;
        e = hipMalloc((void**)&p, bitmapBytes + (size_t)nSlots * FSE_SYM_SLOT_BYTES);
        if (e != hipSuccess) return e;
        e = hipMemset(p, 0, bitmapBytes);                          // (first call on this device only: synchronous, before any launch uses it)
        if (e != hipSuccess) { (void)hipFree(p); return e; }
        sc.bitmap = (u32*)p; sc.base = p + bitmapBytes; sc.nSlots = nSlots;
    }
    a.symScratch = sc.base; a.slotBitmap = sc.bitmap; a.nSlots = sc.nSlots; a.scratchSlotBytes = FSE_SYM_SLOT_BYTES;
    return hipSuccess;
}

static hipError_t fse_decode_launch(FseDecArgs a, bool rev, hipStream_t s)
{
#ifdef FSE_DEC_LDS12_KB             // A/B aid: another workgroup size for the classes with 8 KiB tables
    const size_t ldsBytes = a.ldsLog >= 12 ? FSE_DEC_LDS12_KB * 1024 : FSE_DEC_LDS;
#else
    const size_t ldsBytes = FSE_DEC_LDS;
#endif
    const bool caller = a.atab == nullptr;
    {   hipError_t e = ensure_dyn_lds((const void*)k_fse_decode<true, false, false>, FSE_DEC_LDS);
        if (e == hipSuccess) e = ensure_dyn_lds((const void*)k_fse_decode<false, false, false>, FSE_DEC_LDS);
        if (e == hipSuccess) e = ensure_dyn_lds((const void*)k_fse_decode<true, true, false>, FSE_DEC_LDS);
        if (e == hipSuccess && caller) e = ensure_dyn_lds((const void*)k_fse_decode<true, false, true>, FSE_DEC_LDS);
        if (e == hipSuccess && caller) e = ensure_dyn_lds((const void*)k_fse_decode<false, false, true>, FSE_DEC_LDS);
        if (e == hipSuccess && caller) e = fse_sym_scratch(a, s);
        if (e != hipSuccess) return e;
    }
    fse_decode_geometry(a.ldsLog, ldsBytes, &a.slotU32, &a.G);
    if (a.G < 1) return hipErrorInvalidValue;
    if (caller && ((size_t)a.G << a.ldsLog) > FSE_SYM_SLOT_BYTES) return hipErrorInvalidValue;
    const size_t groups = (a.nBlocks + a.G - 1) / a.G;
    if (caller) {
        if (rev && g_decTimingOn.load(std::memory_order_relaxed)) {
            hipError_t e = ensure_dyn_lds((const void*)k_fse_decode<true, true, true>, FSE_DEC_LDS);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_fse_decode<true, true, true>), dim3((unsigned)groups), dim3(FSE_DEC_THREADS), ldsBytes, s, a);
        }
        else if (rev) hipLaunchKernelGGL((k_fse_decode<true, false, true>), dim3((unsigned)groups), dim3(FSE_DEC_THREADS), ldsBytes, s, a);
        else     hipLaunchKernelGGL((k_fse_decode<false, false, true>), dim3((unsigned)groups), dim3(FSE_DEC_THREADS), ldsBytes, s, a);
    }
    else if (rev && g_decTimingOn.load(std::memory_order_relaxed)) hipLaunchKernelGGL((k_fse_decode<true, true, false>), dim3((unsigned)groups), dim3(FSE_DEC_THREADS), ldsBytes, s, a);
    else if (rev) hipLaunchKernelGGL((k_fse_decode<true, false, false>), dim3((unsigned)groups), dim3(FSE_DEC_THREADS), ldsBytes, s, a);
    else          hipLaunchKernelGGL((k_fse_decode<false, false, false>), dim3((unsigned)groups), dim3(FSE_DEC_THREADS), ldsBytes, s, a);
    return hipGetLastError();
}

// What the batched calls otherwise do at their first use on a device, done now: the library's one allocation on these paths (the symbol
// scratch of FSE_decompress_usingDTable over a batch, 2 x CUs slots of 72 KB) -- for callers who want their first such call inside a
// stream capture, or no allocation inside a timed region.  Idempotent; the scratch lives until the process ends.
extern "C" __attribute__((visibility("default"))) int FSEHIP_prepareDevice(void)
{
    FseDecArgs a = FseDecArgs();
    return (int)fse_sym_scratch(a, nullptr);
}

// enable = 1: zero the counters and send the bit-reversed classes through the TIMED kernel; enable = 0: back to the plain kernel and, if
// out16 is given, the counters (see g_decTiming) plus [8] the device's engine clock in kHz, [9] blocks per workgroup and [10]
// workgroups per CU of the 4 KiB class.  Synchronises the device; a measurement aid for bench.py, not part of the codec API.
extern "C" __attribute__((visibility("default"))) int FSEHIP_debug_decodeTiming(int enable, unsigned long long* out16)
{
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    if (enable) {
        unsigned long long zero[16] = { 0 };
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_decTiming), zero, sizeof(zero));
        g_decTimingOn.store(e == hipSuccess);
        return (int)e;
    }
    g_decTimingOn.store(false);
    if (!out16) return 0;
    e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_decTiming), sizeof(g_decTiming));
    if (e != hipSuccess) return (int)e;
    int dev = 0, khz = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
    unsigned slot; int G;
    fse_decode_geometry(FSE_DEC_FAST_MAXLOG, FSE_DEC_LDS, &slot, &G);
    out16[8] = (unsigned long long)khz; out16[9] = (unsigned long long)G | ((unsigned long long)FSE_WGS_PER_CU << 32) | ((unsigned long long)FSE_DEC_WAVES << 40) | ((unsigned long long)FSE_CHECK_EVERY << 48);
    return 0;
}

// Caller-built reference-layout DTables (FSE_decompress_usingDTable over a batch).  The call has no workspace, so there are no class
// lists: every launch walks all blocks and takes the ones of its class, judged from the table's own header (a uniform batch -- the
// usual case -- fills every workgroup; in a mixed one the other classes' slots stay empty for the launch):
//   1. tableLog <= 11 (or the caller's smaller maxTableLog): bit-reversed cells converted while staging, 4 KiB LDS slots -- the fast loop;
//   2. tableLog 12 (only if maxTableLog allows it): the same with 8 KiB slots; a table with a cell of nbBits 0 is marked FSE_DECLINED
//   3. ... and taken by the plain-cell loop.
hipError_t launch_fse_decode(FseDecArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    a.list = nullptr; a.count = nullptr;
    probe_before(PK_FSE_DECODE, s);
    a.ldsLog = a.maxTableLog < FSE_DEC_FAST_MAXLOG ? a.maxTableLog : FSE_DEC_FAST_MAXLOG;
    a.tlMin = 0; a.declineNb0 = 0; a.onlyDeclined = 0;
    hipError_t e = fse_decode_launch(a, true, s);
    if (e == hipSuccess && a.maxTableLog > FSE_DEC_FAST_MAXLOG) {
        a.ldsLog = a.maxTableLog; a.tlMin = FSE_DEC_FAST_MAXLOG + 1; a.declineNb0 = 1;
        e = fse_decode_launch(a, true, s);
        if (e == hipSuccess) { a.tlMin = 0; a.declineNb0 = 0; a.onlyDeclined = 1; e = fse_decode_launch(a, false, s); }
    }
    probe_after(PK_FSE_DECODE, s);
    return e;
}

// one-shot path: the tables come from k_fse_dbuild in the decoder's own formats, one launch per class (internal.h); a class
// nobody belongs to costs a launch of workgroups that return at once
hipError_t launch_fse_decode_classes(FseDecArgs a, const u32* lists, const u32* counts, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    probe_before(PK_FSE_DECODE, s);
    hipError_t e = hipSuccess;
    for (int c = 0; c < FSE_DCLS_KINDS && e == hipSuccess; ++c) {              // the class's FSE_DBINS lists = one launch (slotBlock in k_fse_decode)
        if (c != FSE_DCLS_REV11 && a.maxTableLog <= FSE_DEC_FAST_MAXLOG) break;      // those classes need tableLog 12
        a.list = lists + (size_t)c * FSE_DBINS * a.nBlocks; a.count = counts + c * FSE_DBINS;
        a.ldsLog = c == FSE_DCLS_REV11 ? (a.maxTableLog < FSE_DEC_FAST_MAXLOG ? a.maxTableLog : FSE_DEC_FAST_MAXLOG) : a.maxTableLog;
        e = fse_decode_launch(a, c != FSE_DCLS_PLAIN, s);
    }
    probe_after(PK_FSE_DECODE, s);
    return e;
}
