// fse_encode.hip -- a2: FSE_compress_usingCTable over a batch
// (reference: lib/fse_compress.c:554-623, lib/fse.h:503-527, lib/bitstream.h:183-260; format SURVEY A.1/A.3).
//
// v1 mapping ("one lane per block"): a 64-lane workgroup stages G CTables into LDS (coalesced copy of the
// reference layout; deltaFindState is rebased on the fly so that one shift-add yields the LDS byte address
// of the next state) and lane g runs both tANS chains of block g.  The two chains are independent, so each
// lane keeps two dependent LDS lookups in flight; the instruction count per symbol is what bounds a
// partially filled wavefront, so the loop is kept lean:
//   * the source is consumed backwards 16 bytes (one aligned, prefetched 16-byte load) at a time with
//     static byte extraction; odd head/tail symbols go through a short generic loop;
//   * bits are packed LSB-first into a 64-bit register and flushed every 4 symbols exactly like
//     BIT_flushBits (lib/bitstream.h:239-249): one unaligned 8-byte store, advance by whole bytes, clamp
//     at capacity-8.  gfx950 global memory accepts the unaligned store; the clamp keeps every store inside
//     [dst, dst+capacity) and makes the return value literally BIT_closeCStream's (:254-260).
#include "internal.h"

// FSE_encodeSymbol (lib/fse.h:514-521) against the LDS-resident table.
//   tt[2*sym]   = 2*deltaFindState + byte offset of this slot's stateTable inside LDS
//   tt[2*sym+1] = deltaNbBits
#define FSE_ENC_STEP(x, sym, bits, nb)                                                              \
    {   const u32 f2 = tt[2 * (sym)], dn = tt[2 * (sym) + 1];                                        \
        nb = ((x) + dn) >> 16;                                                                       \
        bits = __builtin_amdgcn_ubfe((x), 0u, nb);                                                   \
        (x) = *(const u16*)(ldsb + ((((x) >> nb) << 1) + f2));                                       \
    }

__global__ __launch_bounds__(64) void k_fse_encode(FseEncArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u8* const ldsb = (const u8*)lds;
    const u32 lane = threadIdx.x;
    const size_t first = (size_t)blockIdx.x * a.G;

    // A group none of whose blocks is this kernel's (the usual case behind the wave kernel: FSE_ENC_LANE blocks are rare) leaves after ONE
    // look at its blocks' states, lane g at block g -- the staging loop below would read them one after the other (G dependent loads per
    // workgroup: the empty pass over 100k blocks took 59 us, now the time of its launches).  The workgroup is one wave.
    if (a.meta) {
        const size_t bb = first + lane;
        const bool mine = lane < (u32)a.G && bb < a.nBlocks && !fse_enc_skip(a.meta[bb].state, a.onlyState);
        if (!__any(mine)) return;
    }

    // ---- stage the CTables of this group (wave-uniform control flow, coalesced copies)
    for (int g = 0; g < a.G; ++g) {
        const size_t b = first + g;
        if (b >= a.nBlocks) break;
        if (a.meta && fse_enc_skip(a.meta[b].state, a.onlyState)) continue;
        if (a.sizeSplit && view_size(a.src, b) >= FSE_ENC_WAVE_MIN) continue;    // the wave kernel's block
        const u32* t = a.ctables + b * a.ctStrideU32;
        const u32 h = t[0];
        // symbols are bytes: entries above 255 of a table with a larger maxSymbolValue (FSE_buildCTable_raw with nbBits > 8,
        // lib/fse_compress.c:498-528) can never be addressed and are not staged
        const u32 tl = h & 0xFFFFu, msv = (h >> 16) > 255u ? 255u : (h >> 16);
        if (tl > a.maxTableLog) continue;
        const u32 ttStart = 1 + (tl ? (1u << (tl - 1)) : 1u);
        const u32 words = ttStart + 2 * (msv + 1);
        u32* s = lds + (size_t)g * a.slotU32;
        const u32 stOff = (u32)(g * a.slotU32 + 1) * 4u;             // LDS byte offset of this slot's stateTable
        for (u32 i = lane; i < words; i += 64) {
            u32 v = t[i];
            if (i >= ttStart && (((i - ttStart) & 1u) == 0)) v = 2u * v + stOff;
            s[i] = v;
        }
    }
    __syncthreads();
    if (lane >= (u32)a.G) return;
    const size_t b = first + lane;
    if (b >= a.nBlocks) return;
    u32 hdr = 0;
    if (a.meta) { if (fse_enc_skip(a.meta[b].state, a.onlyState)) return; hdr = a.meta[b].hdrSize; }
    if (a.sizeSplit && view_size(a.src, b) >= FSE_ENC_WAVE_MIN) return;

    const u32 h0 = a.ctables[b * a.ctStrideU32];
    const u32 tl = h0 & 0xFFFFu;
    if (tl > a.maxTableLog) {                                        // table does not fit the slot the caller configured
        a.results[b] = FERR(tableLog_tooLarge);
        return;
    }
    const u32* const tt = lds + (size_t)lane * a.slotU32 + 1 + (tl ? (1u << (tl - 1)) : 1u);

    // uniform base + 32-bit lane offset addressing (launcher guarantees G*stride + size < 4 GiB)
    const u8* const sbase = a.src.base + first * a.src.stride;
    const u32 soff = lane * (u32)a.src.stride;
    u8* const dbase = a.dst + first * a.dstStride;
    const u32 doff = lane * (u32)a.dstStride + hdr;
    const size_t n64 = view_size(a.src, b);
    const size_t cap64 = a.dstCapacity - hdr;                        // hdr <= dstCapacity (written by the prepare step)

    size_t csize = 0;
    if (n64 >= ((size_t)1 << 31)) { a.results[b] = FERR(srcSize_wrong); return; }   // batch path: blocks < 2 GiB
    const u32 n = (u32)n64;
    if (n > 2 && cap64 > 8) {                                        // fse_compress.c:566-568
        const u32 lim = (u32)(cap64 - 8);                            // endPtr = start + cap - 8 (bitstream.h:190)
        u64 acc = 0; u32 nacc = 0, pos = 0;
#define FSE_FLUSH()                                                                                  \
        {   __builtin_memcpy(dbase + (doff + pos), &acc, 8);                                          \
            const u32 nby = nacc >> 3;                                                                \
            pos += nby; pos = pos > lim ? lim : pos;                                                  \
            acc >>= (nby << 3); nacc &= 7u; }
        u32 xa, xb;                                                  // xa: chain of the NEXT symbol, xb: the other one
        {   const u32 s0 = sbase[soff + n - 1], s1 = sbase[soff + n - 2];   // FSE_initCState2 (fse.h:503-512): no bits
            const u32 f0 = tt[2 * s0], d0 = tt[2 * s0 + 1], f1 = tt[2 * s1], d1 = tt[2 * s1 + 1];
            const u32 n0 = (d0 + (1u << 15)) >> 16, n1 = (d1 + (1u << 15)) >> 16;
            xa = *(const u16*)(ldsb + (((((n0 << 16) - d0) >> n0) << 1) + f0));
            xb = *(const u16*)(ldsb + (((((n1 << 16) - d1) >> n1) << 1) + f1));
        }
        u32 idx = n - 2;                                             // symbols left: src[0 .. idx-1], read downwards
        bool swapped = false;
        // head: single symbols until the read cursor is 16-byte aligned
        while (idx > 0 && (((uintptr_t)(sbase + soff) + idx) & 15u)) {
            const u32 sym = sbase[soff + idx - 1];
            u32 bits, nb;
            FSE_ENC_STEP(xa, sym, bits, nb)
            acc |= (u64)bits << nacc; nacc += nb;
            FSE_FLUSH()
            const u32 tsw = xa; xa = xb; xb = tsw; swapped = !swapped;
            --idx;
        }
        // body: 16 symbols per aligned 16-byte load, next load already in flight
        if (idx >= 16) {
            uint4 cur = *(const uint4*)(sbase + (soff + idx - 16));
            while (idx >= 16) {
                const u32 nidx = idx - 16;
                const uint4 nxt = *(const uint4*)(sbase + (soff + (nidx >= 16 ? nidx - 16 : nidx)));
                u32 ba, na, bb, nbb;
#define FSE_PAIR(w, hiA, hiB)                                                                        \
                {   const u32 sa = __builtin_amdgcn_ubfe(w, hiA, 8u), sb = __builtin_amdgcn_ubfe(w, hiB, 8u); \
                    FSE_ENC_STEP(xa, sa, ba, na)                                                      \
                    FSE_ENC_STEP(xb, sb, bb, nbb)                                                     \
                    acc |= (u64)(ba | (bb << na)) << nacc; nacc += na + nbb; }
                FSE_PAIR(cur.w, 24u, 16u) FSE_PAIR(cur.w, 8u, 0u) FSE_FLUSH()
                FSE_PAIR(cur.z, 24u, 16u) FSE_PAIR(cur.z, 8u, 0u) FSE_FLUSH()
                FSE_PAIR(cur.y, 24u, 16u) FSE_PAIR(cur.y, 8u, 0u) FSE_FLUSH()
                FSE_PAIR(cur.x, 24u, 16u) FSE_PAIR(cur.x, 8u, 0u) FSE_FLUSH()
                cur = nxt;
                idx = nidx;
            }
        }
        // tail
        while (idx > 0) {
            const u32 sym = sbase[soff + idx - 1];
            u32 bits, nb;
            FSE_ENC_STEP(xa, sym, bits, nb)
            acc |= (u64)bits << nacc; nacc += nb;
            FSE_FLUSH()
            const u32 tsw = xa; xa = xb; xb = tsw; swapped = !swapped;
            --idx;
        }
        if (swapped) { const u32 tsw = xa; xa = xb; xb = tsw; }     // xa = even-distance chain, xb = odd-distance chain
        // fse_compress.c:608-609 : CState2 then CState1.  n even -> CState2 is the even chain (:577-580), n odd -> CState1 (:572-576)
        const u32 c2 = (n & 1) ? xb : xa, c1 = (n & 1) ? xa : xb;
        acc |= (u64)(c2 & ((1u << tl) - 1u)) << nacc; nacc += tl; FSE_FLUSH()
        acc |= (u64)(c1 & ((1u << tl) - 1u)) << nacc; nacc += tl; FSE_FLUSH()
        acc |= (u64)1 << nacc; nacc += 1; FSE_FLUSH()                // BIT_closeCStream: end mark, flush
        csize = (pos >= lim) ? 0 : (size_t)pos + (nacc > 0);
#undef FSE_FLUSH
#undef FSE_PAIR
    }
    if (a.meta) {                                                    // one-shot wrap-up, fse_compress.c:668-676
        size_t r = 0;
        if (csize != 0 && (size_t)hdr + csize < n64 - 1) r = (size_t)hdr + csize;
        a.results[b] = r;
    } else a.results[b] = csize;
}

#define FSE_ENC_LDS (80 * 1024)   // two workgroups per CU (measured: 2 x 80 KiB are co-resident on gfx950)
static void fse_encode_geometry(unsigned maxTableLog, unsigned* slotU32, int* G)
{
    *slotU32 = (1 + (1u << (maxTableLog - 1)) + 512) | 1u;           // odd word stride: slots start on rotating banks
    int g = (int)(FSE_ENC_LDS / (*slotU32 * 4));
    if (g > 64) g = 64;
    *G = g;
}
size_t fse_encode_blocks_per_round(unsigned maxTableLog)
{
    unsigned slot; int G;
    fse_encode_geometry(maxTableLog, &slot, &G);
    const int cus = dev_props().ok ? dev_props().cus : 256;
    return (size_t)G * 2 * cus;
}

hipError_t launch_fse_encode(FseEncArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    const size_t ldsBytes = FSE_ENC_LDS;
    {   const hipError_t e = ensure_dyn_lds((const void*)k_fse_encode, (int)ldsBytes); if (e != hipSuccess) return e; }
    fse_encode_geometry(a.maxTableLog, &a.slotU32, &a.G);
    // 32-bit lane offsets inside a group: blocks of 64 MB and more get a group each
    if (a.nBlocks > 1 && ((a.src.stride > 0x3FFFFFFu) || (a.dstStride > 0x3FFFFFFu))) a.G = 1;
    if (a.dstCapacity > 0x7FFFFFF0u) a.dstCapacity = 0x7FFFFFF0u;   // blocks are < 2 GiB on this path (see kernel)
    const size_t groups = (a.nBlocks + a.G - 1) / a.G;
    probe_before(PK_FSE_ENCODE, s);
    hipLaunchKernelGGL(k_fse_encode, dim3((unsigned)groups), dim3(64), ldsBytes, s, a);
    probe_after(PK_FSE_ENCODE, s);
    return hipGetLastError();
}
