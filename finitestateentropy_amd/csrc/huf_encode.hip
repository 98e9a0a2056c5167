// huf_encode.hip -- a4: HUF_compress1X_usingCTable / HUF_compress4X_usingCTable over a batch
// (reference: lib/huf_compress.c:457-608; format SURVEY A.6).
//
// A prefix code has no loop-carried state, so the encoder is fully parallel inside a block:
//   one workgroup per block, one 64-lane wavefront per stream (4 waves for the 4-stream layout).
//   the stream is read in emission order (last symbol first) with 16-byte loads, a whole 8 KiB stream at once, and
//   kept in registers for both passes;
//   pass 1: every lane sums the code lengths of its symbols  -> exact stream sizes, hence the 6-byte jump table,
//           every stream's start offset and the return value (0 when any stream fails BIT_closeCStream's capacity
//           rule, lib/bitstream.h:254-260) before a bit is written;
//   pass 2: rows of 1024 symbols; lane L owns 16 consecutive symbols, packs their codes LSB-first into four
//           <=48-bit chunks, a wave prefix sum of the lane totals gives its bit position and the chunks are OR-ed
//           into an LDS image of the block's output (ds_or_b32).
//   The image is then copied to global memory with coalesced stores.  Outputs too large for the LDS image
//   (only possible for blocks > 32 KB) take the same path with global atomics on a pre-zeroed destination.
#include "internal.h"

#define HUF_ENC_THREADS 256

// (on the DPP path, dev_common.h: the row scan was seven ds_bpermute per row of 1024 symbols beside the row's ~24 table reads and ORs -- a fifth of
//  the kernel's LDS-pipe instructions)
DEV u32 wave_incl_scan_u32(u32 v, u32 lane)
{
#if FSEHIP_DPP_SCANS
    return group_scan_incl<64, ScanAdd>(v, lane);
#else
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 o = (u32)__shfl_up((int)v, off, WAVE); if ((int)lane >= off) v += o; }
    return v;
#endif
}
DEV u32 wave_sum_u32(u32 v)
{
#if FSEHIP_DPP_SCANS
    return group_reduce<64, ScanAdd>(v, 0);
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (u32)__shfl_xor((int)v, off, WAVE);
    return v;
#endif
}

// OR `nb` (<= 48) bits of `bits` into the bit image at absolute bit position P
template <bool GLOBAL>
DEV void or_bits(u32* img, u64 P, u64 bits, u32 nb)
{
    if (nb == 0) return;
    const size_t w = (size_t)(P >> 5);
    const u32 sh = (u32)P & 31u;
    const u64 lo = bits << sh;                           // bits < 2^48, sh < 32 : up to 79 bits in all
    const u32 hi = sh ? (u32)(bits >> (64 - sh)) : 0u;
    const u32 a0 = (u32)lo, a1 = (u32)(lo >> 32);
    if (a0) atomicOr(&img[w], a0);
    if (a1) atomicOr(&img[w + 1], a1);
    if (hi) atomicOr(&img[w + 2], hi);
}

// A stream is walked in emission order (j = 0 is the LAST symbol of the segment, huf_compress.c:474-499 net effect) in
// tiles of HE_TILE symbols: a tile is HE_ROWS rows of 1024 symbols, lane L of the wave owning the 16 consecutive symbols
// j = row*1024 + 16*L .. +15 of every row = one (unaligned) 16-byte load whose bytes are taken from the top down.
// All loads of a tile are issued before the first one is used, and a stream of up to HE_TILE symbols (the 8 KiB streams
// of a 32 KiB block) is loaded once for both passes.
#define HE_ROWS 4u
#define HE_TILE (HE_ROWS * 1024u)
struct HeTile { uint4 v[HE_ROWS]; };
DEV uint4 he_load16(const u8* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
DEV void he_load_tile(HeTile& t, const u8* seg, u32 len, u32 j0, u32 lane)
{
#pragma unroll
    for (u32 r = 0; r < HE_ROWS; ++r) {
        const u32 j = j0 + r * 1024u + 16u * lane;          // my first symbol of this row
        t.v[r] = make_uint4(0, 0, 0, 0);
        if (j + 16 <= len) t.v[r] = he_load16(seg + (len - j - 16));
        else if (j < len) {                                  // the row that reaches the start of the segment: bytewise
            u32 w[4] = { 0, 0, 0, 0 };
            for (u32 k = 0; k < len - j; ++k) { const u32 pos = 15u - k; w[pos >> 2] |= (u32)seg[len - 1 - (j + k)] << (8u * (pos & 3u)); }
            t.v[r] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}
// symbol k (0..15, emission order) of a row register: byte 15 - k
DEV u32 he_sym(const uint4& v, u32 k) { const u32 w = k < 4 ? v.w : k < 8 ? v.z : k < 12 ? v.y : v.x; return (w >> (8u * (3u - (k & 3u)))) & 0xFFu; }

// pass 1 over one tile: code bits of my symbols
DEV u32 he_count_tile(const HeTile& t, u32 len, u32 j0, const u32* ct, u32 lane)
{
    u32 bits = 0;
#pragma unroll
    for (u32 r = 0; r < HE_ROWS; ++r) {
        const u32 j = j0 + r * 1024u + 16u * lane;
#pragma unroll
        for (u32 k = 0; k < 16; ++k) if (j + k < len) bits += (ct[he_sym(t.v[r], k)] >> 16) & 0xFFu;
    }
    return bits;
}
// pass 2 over one tile: every lane packs its 16 symbols of a row into four <= 48-bit chunks; one wave prefix sum per row
// gives the lane's bit position; the chunks are OR-ed into the image.  Returns the bit position behind the tile.
// `limit`: bits the image may hold; a row that would pass it is not written and HE_OVERFLOW is returned (uniform).
#define HE_OVERFLOW (~(u64)0)
template <bool GLOBAL>
DEV u64 he_emit_tile(u32* img, u64 rowBase, const HeTile& t, u32 len, u32 j0, const u32* ct, u32 lane, u64 limit = HE_OVERFLOW)
{
#pragma unroll
    for (u32 r = 0; r < HE_ROWS; ++r) {
        const u32 j = j0 + r * 1024u + 16u * lane;
        if (j0 + r * 1024u >= len) break;                    // uniform
        u64 cb[4]; u32 cn[4]; u32 tot = 0;
        if (j0 + (r + 1) * 1024u <= len) {                   // uniform: a whole row -- no per-symbol bounds, codes joined in pairs (<= 22 bits) first
#pragma unroll
            for (u32 c = 0; c < 4; ++c) {
                const u32 e0 = ct[he_sym(t.v[r], 4 * c)], e1 = ct[he_sym(t.v[r], 4 * c + 1)], e2 = ct[he_sym(t.v[r], 4 * c + 2)], e3 = ct[he_sym(t.v[r], 4 * c + 3)];
                const u32 n0 = (e0 >> 16) & 0xFFu, n1 = (e1 >> 16) & 0xFFu, n2 = (e2 >> 16) & 0xFFu, n3 = (e3 >> 16) & 0xFFu;
                const u32 p01 = ((e1 & 0xFFFFu) << n0) | (e0 & 0xFFFFu), p23 = ((e3 & 0xFFFFu) << n2) | (e2 & 0xFFFFu);
                const u32 n01 = n0 + n1;
                cb[c] = (u64)p01 | ((u64)p23 << n01); cn[c] = n01 + n2 + n3; tot += cn[c];
            }
        } else {
#pragma unroll
            for (u32 c = 0; c < 4; ++c) {
                u64 bits = 0; u32 nb = 0;
#pragma unroll
                for (u32 k = 0; k < 4; ++k) {
                    if (j + 4 * c + k < len) {
                        const u32 e = ct[he_sym(t.v[r], 4 * c + k)];
                        bits |= (u64)(e & 0xFFFFu) << nb;
                        nb += (e >> 16) & 0xFFu;
                    }
                }
                cb[c] = bits; cn[c] = nb; tot += nb;
            }
        }
        const u32 incl = wave_incl_scan_u32(tot, lane);
        const u32 rowBits = group_last<64>(incl, lane);
        if (rowBase + rowBits > limit) return HE_OVERFLOW;   // uniform
        u64 pos = rowBase + (incl - tot);
#pragma unroll
        for (u32 c = 0; c < 4; ++c) { or_bits<GLOBAL>(img, pos, cb[c], cn[c]); pos += cn[c]; }
        rowBase += rowBits;
    }
    return rowBase;
}

#ifndef HUF_ENC_WAVES_PER_EU
#define HUF_ENC_WAVES_PER_EU 5          // 96 registers per lane (see launch_huf_encode)
#endif
#ifndef HUF_ENC_IMG_MAX
#define HUF_ENC_IMG_MAX 30720u
#endif
__global__ __launch_bounds__(HUF_ENC_THREADS) __attribute__((amdgpu_waves_per_eu(HUF_ENC_WAVES_PER_EU))) void k_huf_encode(HufEncArgs a, u32 imgBytes)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    u32* const ct = lds;                 // 256 HUF_CElt
    u32* const sh = lds + 256;           // [0..3] stream bit counts
    u32* const img = lds + 256 + 8;      // output image (imgBytes)
    const size_t b = blockIdx.x;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int streams = a.streams;

    u32 hdr = 0;
    if (a.meta) { if (a.meta[b].state == 0) return; hdr = a.meta[b].hdrSize; }
    const u8* const src = view_ptr(a.src, b);
    const size_t n64 = view_size(a.src, b);
    u8* const dst = a.dst + b * a.dstStride + hdr;
    const size_t cap = a.dstCapacity - hdr;

    // early-outs (huf_compress.c:564-565 for 4X; :470-472 for 1X)
    bool fail = false;
    if (streams == 4) { if (cap < 6 + 1 + 1 + 1 + 8) fail = true; if (n64 < 12) fail = true; }
    if (n64 >= ((size_t)1 << 31)) fail = true;           // batch path: blocks < 2 GiB
    const u32 n = (u32)n64;
    if (fail) { if (tid == 0) a.results[b] = 0; return; }

    const u32* gct = a.ctables + b * a.ctStrideU32;
    for (u32 i = tid; i < 256; i += blockDim.x) ct[i] = gct[i];
    __syncthreads();

    // 1X over a batch: the single stream is the plain concatenation of the codes in emission order (last symbol first), so the four
    // waves of the workgroup take a quarter of the symbols each -- wave w the emission range [w*q, (w+1)*q) -- and the bit counts of
    // pass 1 place their pieces (the single-block host call keeps one wave)
    const bool split1 = streams == 1 && blockDim.x == HUF_ENC_THREADS;
    const u32 segSize = streams == 4 ? (n + 3) / 4 : n;
    u32 myStart = wave * segSize;
    u32 myLen = (int)wave < streams ? (streams == 4 && wave == 3 ? n - 3 * segSize : segSize) : 0;
    if (split1) {
        const u32 q = (((n + 3) / 4) + 15u) & ~15u;
        const u32 e0 = wave * q;
        myLen = e0 < n ? (n - e0 < q ? n - e0 : q) : 0;
        myStart = n - e0 - myLen;
    }

    const u8* const seg = src + myStart;
    HeTile tile0;
    he_load_tile(tile0, seg, myLen, 0, lane);

    // ---- single pass (4 streams): every wave emits its stream from bit 0 of its own quarter of the image; the sizes are then
    //      known and the streams are moved to their places while being copied out (byte-granular: every stream starts on a
    //      byte).  A stream that does not fit its quarter (> 8 KB from an 8 KB segment: the block is all but incompressible)
    //      sends the block to the two-pass path below.
    const u32 regionBytes = (imgBytes / 4u) & ~15u;
    bool single = streams == 4 && regionBytes >= 1024u;
    if (single) {
        {   uint4* const z = (uint4*)img;
            for (u32 i = tid; i < regionBytes / 4u; i += blockDim.x) z[i] = make_uint4(0, 0, 0, 0); }
        if (tid < 8) sh[tid] = 0;
        __syncthreads();
        u32* const region = img + wave * (regionBytes / 4u);
        const u64 limit = 8ull * regionBytes - 64u;
        u64 pos = 0;
        HeTile cur = tile0;
        for (u32 j0 = 0; j0 < myLen && pos != HE_OVERFLOW; j0 += HE_TILE) {
            HeTile nxt = cur;
            if (j0 + HE_TILE < myLen) he_load_tile(nxt, seg, myLen, j0 + HE_TILE, lane);
            __asm__ volatile("" ::: "memory");
            pos = he_emit_tile<false>(region, pos, cur, myLen, j0, ct, lane, limit);
            cur = nxt;
        }
        if (lane == 0) {
            if (pos == HE_OVERFLOW) sh[4] = 1;
            else { or_bits<false>(region, pos, 1, 1); sh[wave] = (u32)pos; }
        }
        __syncthreads();
        if (sh[4]) single = false;                               // uniform
        __syncthreads();
    }

    // ---- pass 1 of the two-pass path: code bits of my stream (the first tile stays in registers for pass 2)
    if (!single) {   // every tile is loaded while the one before it is being processed
        u32 bits = 0;
        HeTile cur = tile0;
        for (u32 j0 = 0; j0 < myLen; j0 += HE_TILE) {
            HeTile nxt = cur;
            if (j0 + HE_TILE < myLen) he_load_tile(nxt, seg, myLen, j0 + HE_TILE, lane);
            __asm__ volatile("" ::: "memory");               // keep the loads above the LDS lookups of the current tile
            bits += he_count_tile(cur, myLen, j0, ct, lane);
            cur = nxt;
        }
        bits = wave_sum_u32(bits);
        if (lane == 0 && ((int)wave < streams || split1)) sh[wave] = bits;
    }
    __syncthreads();
    const u32 bitsBefore = split1 ? (wave > 0 ? sh[0] : 0) + (wave > 1 ? sh[1] : 0) + (wave > 2 ? sh[2] : 0) : 0;   // bits of the waves in front of mine
    const u32 streamBits = split1 ? sh[0] + sh[1] + sh[2] + sh[3] : 0;

    // ---- layout + verdict (uniform)
    size_t op = streams == 4 ? 6 : 0;
    size_t start[4] = { 0, 0, 0, 0 }, ssize[4] = { 0, 0, 0, 0 };
    for (int k = 0; k < streams; ++k) {
        const size_t capk = cap - op;                    // oend - op
        if (capk <= 8) { fail = true; break; }           // dstSize < 8 / BIT_initCStream (bitstream.h:191)
        const size_t tot = (size_t)(split1 ? streamBits : sh[k]) + 1;            // + end mark
        if ((tot >> 3) >= capk - 8) { fail = true; break; }   // BIT_closeCStream overflow rule
        start[k] = op; ssize[k] = (tot + 7) >> 3;
        op += ssize[k];
    }
    const size_t total = op;
    size_t result = fail ? 0 : total;
    if (a.meta && !fail) result = ((size_t)hdr + total < n64 - 1) ? (size_t)hdr + total : 0;   // huf_compress.c:625
    if (result == 0) { if (tid == 0) a.results[b] = 0; return; }

    if (single) {
        // ---- copy-out of the single-pass path: wave k moves stream k; output dwords on aligned addresses, each from two image
        //      dwords (v_alignbyte), the edges bytewise; the jump table goes out bytewise too
        const u32* const region = img + wave * (regionBytes / 4u);
        u8* const D = dst + start[wave];
        const u32 size = (u32)ssize[wave];
        u32 h = (u32)((0 - (uintptr_t)D) & 3u);
        h = h < size ? h : size;
        const u8* const rb = (const u8*)region;
        if (lane < h) D[lane] = rb[lane];
        const u32 nW = (size - h) / 4u;
        for (u32 i = lane; i < nW; i += 64) {
            const u32 w = __builtin_amdgcn_alignbyte(region[i + 1], region[i], h);
            *(u32*)(D + h + 4u * i) = w;
        }
        const u32 done = h + 4u * nW;
        if (lane < size - done) D[done + lane] = rb[done + lane];
        if (lane == 0 && wave < 3) { dst[2 * wave] = (u8)ssize[wave]; dst[2 * wave + 1] = (u8)(ssize[wave] >> 8); }
        if (tid == 0) a.results[b] = result;
        return;
    }

    // ---- pass 2: emit.  The image is addressed from the 4-byte aligned word holding dst[0].
    const u32 lead = (u32)((uintptr_t)dst & 3u);
    u8* const dstAl = dst - lead;
    const size_t imgWords = (lead + total + 3) >> 2;
    const bool inLds = (imgWords * 4 <= imgBytes);
    // The four streams of a block that compresses by less than 6 % do not fit the image together: they are emitted in two HALVES -- the jump
    // table and streams 0, 1, then streams 2, 3 (every stream starts on a byte) -- each through the image like a block of its own.
    if (!inLds && streams == 4 && start[2] + 8 <= imgBytes && (total - start[2]) + 8 <= imgBytes) {
        for (int hf = 0; hf < 2; ++hf) {                                  // uniform
            const size_t off = hf ? start[2] : 0, len = hf ? total - start[2] : start[2];
            u8* const dH = dst + off;
            const u32 leadH = (u32)((uintptr_t)dH & 3u);
            u8* const dHAl = dH - leadH;
            const size_t wordsH = (leadH + len + 3) >> 2;
            for (size_t i = tid; i < wordsH; i += blockDim.x) img[i] = 0;
            __syncthreads();
            if ((int)(wave >> 1) == hf) {
                const u64 base = 8 * ((u64)leadH + start[wave] - off);
                u64 pos = base;
                HeTile cur = tile0;
                for (u32 j0 = 0; j0 < myLen; j0 += HE_TILE) {
                    HeTile nxt = cur;
                    if (j0 + HE_TILE < myLen) he_load_tile(nxt, seg, myLen, j0 + HE_TILE, lane);
                    __asm__ volatile("" ::: "memory");
                    pos = he_emit_tile<false>(img, pos, cur, myLen, j0, ct, lane);
                    cur = nxt;
                }
                if (lane == 0) or_bits<false>(img, base + sh[wave], 1, 1);                       // the stream's end mark
            }
            if (hf == 0 && tid < 3) or_bits<false>(img, 8 * ((u64)leadH + 2 * tid), ssize[tid], 16);     // the jump table
            __syncthreads();
            {   const u8* ib = (const u8*)img;
                const size_t end = leadH + len;
                for (size_t w = tid; w < wordsH; w += blockDim.x) {
                    const size_t lo = 4 * w, hi = 4 * w + 4;
                    if (lo >= leadH && hi <= end) ((u32*)dHAl)[w] = img[w];
                    else for (size_t i = (lo > leadH ? lo : leadH); i < (hi < end ? hi : end); ++i) dHAl[i] = ib[i];
                }
            }
            __syncthreads();
        }
        if (tid == 0) a.results[b] = result;
        return;
    }
    if (inLds) {
        for (size_t i = tid; i < imgWords; i += blockDim.x) img[i] = 0;
    } else {
        for (size_t i = tid; i < total; i += blockDim.x) dst[i] = 0;
    }
    __syncthreads();
    if ((int)wave < streams || split1) {
        const u64 base = 8 * ((u64)lead + start[split1 ? 0 : wave]) + bitsBefore;
        const u32 markAt = split1 ? streamBits - bitsBefore : sh[wave];          // the end mark follows the stream's last bit (written by its first wave)
        if (inLds) {                                     // separate call sites keep the LDS / global address spaces visible
            u64 pos = base;
            HeTile cur = tile0;
            for (u32 j0 = 0; j0 < myLen; j0 += HE_TILE) {
                HeTile nxt = cur;
                if (j0 + HE_TILE < myLen) he_load_tile(nxt, seg, myLen, j0 + HE_TILE, lane);
                __asm__ volatile("" ::: "memory");
                pos = he_emit_tile<false>(img, pos, cur, myLen, j0, ct, lane);
                cur = nxt;
            }
            if (lane == 0 && (!split1 || wave == 0)) {   // end mark, then the jump table entry of this stream
                or_bits<false>(img, base + markAt, 1, 1);
                if (streams == 4 && wave < 3) or_bits<false>(img, 8 * ((u64)lead + 2 * wave), ssize[wave], 16);
            }
        } else {
            u32* const g = (u32*)dstAl;
            u64 pos = base;
            HeTile cur = tile0;
            for (u32 j0 = 0; j0 < myLen; j0 += HE_TILE) {
                HeTile nxt = cur;
                if (j0 + HE_TILE < myLen) he_load_tile(nxt, seg, myLen, j0 + HE_TILE, lane);
                __asm__ volatile("" ::: "memory");
                pos = he_emit_tile<true>(g, pos, cur, myLen, j0, ct, lane);
                cur = nxt;
            }
            if (lane == 0 && (!split1 || wave == 0)) {
                or_bits<true>(g, base + markAt, 1, 1);
                if (streams == 4 && wave < 3) or_bits<true>(g, 8 * ((u64)lead + 2 * wave), ssize[wave], 16);
            }
        }
    }
    __syncthreads();
    if (inLds) {                                         // coalesced copy-out: whole aligned words, edge words bytewise
        const u8* ib = (const u8*)img;
        const size_t end = lead + total;
        for (size_t w = tid; w < imgWords; w += blockDim.x) {
            const size_t lo = 4 * w, hi = 4 * w + 4;
            if (lo >= lead && hi <= end) ((u32*)dstAl)[w] = img[w];
            else for (size_t i = (lo > lead ? lo : lead); i < (hi < end ? hi : end); ++i) dstAl[i] = ib[i];
        }
    }
    if (tid == 0) a.results[b] = result;
}

// streams == 1 with a.nBlocks > 1 or a.split1X: four waves per block (see split1 in the kernel)
hipError_t launch_huf_encode(const HufEncArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    const size_t maxLds = 64 * 1024;
    {   const hipError_t e = ensure_dyn_lds((const void*)k_huf_encode, (int)maxLds); if (e != hipSuccess) return e; }
    size_t img = a.dstCapacity + 16;
    if (img > maxLds - 1100) img = maxLds - 1100;
    // Round 6: the image is capped at HUF_ENC_IMG_MAX = 30 KiB so that FIVE workgroups (20 waves, with 96 registers per lane) share a CU's 160 KB
    // instead of four: encode call per 100k blocks P80 2.23 -> 2.07 ms, P14 2.65 -> 2.49, P02 3.80 -> 3.61.  (32 KiB - the code table, 5 x 32 KiB
    // = all 160 KB, was measured too and is not granted five times; 28 KiB regions no longer hold P02's 7.1 KB streams.)  A block whose four
    // streams total more than the image (it compresses by less than 6 %) is emitted in two halves, below.
    if (img > HUF_ENC_IMG_MAX) img = HUF_ENC_IMG_MAX;
    img = (img + 15) & ~(size_t)15;
    const size_t ldsBytes = (256 + 8) * 4 + img;
    const unsigned threads = (a.streams == 4 || a.split1X) ? HUF_ENC_THREADS : 64;
    probe_before(PK_HUF_ENCODE, s);
    hipLaunchKernelGGL(k_huf_encode, dim3((unsigned)a.nBlocks), dim3(threads), ldsBytes, s, a, (u32)img);
    probe_after(PK_HUF_ENCODE, s);
    return hipGetLastError();
}
