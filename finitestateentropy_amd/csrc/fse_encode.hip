// fse_encode.hip -- a2: FSE_compress_usingCTable over a batch
// (reference: lib/fse_compress.c:554-623, lib/fse.h:503-527, lib/bitstream.h:183-260; format SURVEY A.1/A.3).
//
// v1 mapping ("one lane per block"): a 64-lane workgroup stages G CTables (reference layout, copied
// verbatim with coalesced loads) into LDS; lane g then runs both tANS chains of block g against its
// LDS-resident stateTable / symbolTT.  The two chains are independent, so each lane keeps two
// dependent LDS lookups in flight; many blocks per CU hide the rest of the latency.  Bits are packed
// LSB-first into a 64-bit register and leave as 32-bit words.
// The byte stream is the plain concatenation of the (state & mask, nbBits) chunks in decreasing source
// order, then CState2, CState1 (tableLog bits each) and the end-mark bit; the return value follows the
// closed form of BIT_closeCStream: 0 when floor(totalBits/8) >= capacity-8, else ceil(totalBits/8).
#include "internal.h"

// Backwards byte reader over global memory using aligned 32-bit loads with one word of prefetch.
struct RevBytes {
    const u32* w;      // next (lower) aligned word to fetch
    const u32* wmin;   // lowest word that may be touched
    u32 cur;           // bytes not yet delivered sit in the top of `cur`
    u32 have;          // how many
    u32 nxt;
    DEV void init(const u8* base, size_t n)   // will deliver base[n-1], base[n-2], ...
    {
        const uintptr_t end = (uintptr_t)base + n;              // one past the last byte
        wmin = (const u32*)((uintptr_t)base & ~(uintptr_t)3);
        const u32* top = (const u32*)((end - 1) & ~(uintptr_t)3);
        const u32 valid = (u32)(end - (uintptr_t)top);           // 1..4 bytes of the top word belong to the block
        cur = *top << (8 * (4 - valid));
        have = valid;
        w = top - 1;
        nxt = *(w >= wmin ? w : wmin);
    }
    DEV u32 get()
    {
        if (have == 0) {
            cur = nxt; have = 4;
            --w;
            nxt = *(w >= wmin ? w : wmin);
        }
        const u32 b = cur >> 24;
        cur <<= 8; --have;
        return b;
    }
};

// Forward bit sink: LSB-first, 32-bit word stores once aligned, never writes at or beyond `cap`.
struct BitSink {
    u8* out; size_t cap;
    size_t pos;        // bytes already stored
    u64 acc; u32 nacc;
    u32 lead;          // bytes still to emit one by one until out+pos is 4-byte aligned
    bool dead;         // floor(totalBits/8) has reached cap-8: the result will be 0 (bitstream.h:258)
    DEV void init(u8* o, size_t c) { out = o; cap = c; pos = 0; acc = 0; nacc = 0; dead = false; lead = (u32)((0 - (uintptr_t)o) & 3u); }
    DEV void put(u32 v, u32 nb) { acc |= (u64)(v & ((1u << nb) - 1u)) << nacc; nacc += nb; }   // nb <= 16
    DEV void drain()                                         // keeps nacc < 32
    {
        if (dead) { if (nacc >= 32) { acc >>= 32; nacc -= 32; pos += 4; } return; }
        while (lead && nacc >= 8) {
            if (pos + 8 >= cap) { dead = true; return; }
            out[pos++] = (u8)acc; acc >>= 8; nacc -= 8; --lead;
        }
        if (nacc >= 32 && !lead) {
            if (pos + 12 > cap) { dead = true; return; }     // pos+4 > cap-8  =>  final floor(bits/8) >= cap-8
            *(u32*)(out + pos) = (u32)acc;
            acc >>= 32; nacc -= 32; pos += 4;
        }
    }
    DEV size_t close()                                       // end mark + size rule
    {
        acc |= (u64)1 << nacc; nacc += 1;
        const size_t whole = pos + (nacc >> 3);
        if (dead || cap <= 8 || whole >= cap - 8) return 0;
        const u32 nbytes = (nacc + 7) >> 3;
        for (u32 k = 0; k < nbytes; ++k) { out[pos + k] = (u8)acc; acc >>= 8; }
        return pos + nbytes;
    }
};

__global__ __launch_bounds__(64) void k_fse_encode(FseEncArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int lane = threadIdx.x;
    const size_t first = (size_t)blockIdx.x * a.G;

    // ---- stage the CTables of this group (wave-uniform control flow, coalesced copies)
    for (int g = 0; g < a.G; ++g) {
        const size_t b = first + g;
        if (b >= a.nBlocks) break;
        if (a.meta && a.meta[b].state == 0) continue;
        const u32* t = a.ctables + b * a.ctStrideU32;
        const u32 h = t[0];
        const u32 tl = h & 0xFFFFu, msv = h >> 16;
        if (tl > a.maxTableLog || msv > 255u) continue;
        const u32 words = 1 + (tl ? (1u << (tl - 1)) : 1u) + 2 * (msv + 1);
        u32* s = lds + (size_t)g * a.slotU32;
        for (u32 i = lane; i < words; i += 64) s[i] = t[i];
    }
    __syncthreads();
    if (lane >= a.G) return;
    const size_t b = first + lane;
    if (b >= a.nBlocks) return;
    u32 hdr = 0;
    if (a.meta) { if (a.meta[b].state == 0) return; hdr = a.meta[b].hdrSize; }

    const u32* const T = lds + (size_t)lane * a.slotU32;
    const u32 h0 = a.ctables[b * a.ctStrideU32];                     // header word as staged (or not) above
    const u32 tl = h0 & 0xFFFFu;
    if (tl > a.maxTableLog || (h0 >> 16) > 255u) {                   // table does not fit the slot the caller configured
        a.results[b] = FERR(tableLog_tooLarge);
        return;
    }
    const u16* const stateTable = (const u16*)(T + 1);
    const u32* const tt = T + 1 + (tl ? (1u << (tl - 1)) : 1u);     // {deltaFindState, deltaNbBits} pairs

    const u8* const src = view_ptr(a.src, b);
    const size_t n = view_size(a.src, b);
    u8* const dst = a.dst + b * a.dstStride + hdr;
    const size_t cap = a.dstCapacity - hdr;                          // hdr <= dstCapacity (written by the prepare step)

    size_t csize = 0;
    if (n > 2 && cap > 8) {                                          // fse_compress.c:566-568
        RevBytes in; in.init(src, n);
        BitSink bs; bs.init(dst, cap);
        u32 xa, xb;                                                  // chain of even / odd distance-from-the-end
        {   const u32 s0 = in.get(), s1 = in.get();                  // FSE_initCState2 (fse.h:503-512): no bits
            const u32 f0 = tt[2 * s0], d0 = tt[2 * s0 + 1], f1 = tt[2 * s1], d1 = tt[2 * s1 + 1];
            const u32 n0 = (d0 + (1u << 15)) >> 16, n1 = (d1 + (1u << 15)) >> 16;
            xa = stateTable[(((n0 << 16) - d0) >> n0) + f0];
            xb = stateTable[(((n1 << 16) - d1) >> n1) + f1];
        }
        size_t left = n - 2;
        for (; left >= 2; left -= 2) {                               // FSE_encodeSymbol (fse.h:514-521), one step per chain
            const u32 sa = in.get(), sb = in.get();
            const u32 fa = tt[2 * sa], da = tt[2 * sa + 1], fb = tt[2 * sb], db = tt[2 * sb + 1];   // 4-byte aligned pairs
            const u32 na = (xa + da) >> 16, nb = (xb + db) >> 16;
            bs.put(xa, na);
            bs.put(xb, nb);
            xa = stateTable[(xa >> na) + fa];
            xb = stateTable[(xb >> nb) + fb];
            bs.drain();
        }
        if (left) {
            const u32 sa = in.get();
            const u32 fa = tt[2 * sa], da = tt[2 * sa + 1];
            const u32 na = (xa + da) >> 16;
            bs.put(xa, na);
            xa = stateTable[(xa >> na) + fa];
            bs.drain();
        }
        // fse_compress.c:608-609 : CState2 then CState1.  n even -> CState2 is the even chain (:577-580), n odd -> CState1 (:572-576)
        const u32 c2 = (n & 1) ? xb : xa, c1 = (n & 1) ? xa : xb;
        bs.put(c2, tl); bs.drain();
        bs.put(c1, tl); bs.drain();
        csize = bs.close();
    }
    if (a.meta) {                                                    // one-shot wrap-up, fse_compress.c:668-676
        size_t r = 0;
        if (csize != 0 && (size_t)hdr + csize < n - 1) r = (size_t)hdr + csize;
        a.results[b] = r;
    } else a.results[b] = csize;
}

hipError_t launch_fse_encode(FseEncArgs a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    static bool attrSet = false;
    const size_t ldsBytes = 80 * 1024;
    if (!attrSet) {
        hipError_t e = hipFuncSetAttribute((const void*)k_fse_encode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        attrSet = true;
    }
    const u32 tl = a.maxTableLog;
    a.slotU32 = (1 + (1u << (tl - 1)) + 512) | 1u;                   // odd word stride: slots start on rotating banks
    a.G = (int)(ldsBytes / (a.slotU32 * 4));
    if (a.G > 64) a.G = 64;
    const size_t groups = (a.nBlocks + a.G - 1) / a.G;
    hipLaunchKernelGGL(k_fse_encode, dim3((unsigned)groups), dim3(64), ldsBytes, s, a);
    return hipGetLastError();
}
