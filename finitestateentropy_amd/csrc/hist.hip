// hist.hip -- a1: HIST_count over a batch of blocks (reference: lib/hist.c:66-133,163-180).
//
// One 64-lane wavefront per block.  The wave streams the block with coalesced 16-byte loads and
// counts into a wave-private LDS histogram laid out so that no two lanes of one LDS pass ever
// touch the same word:   cnt[row = byte & 127][col = lane & 31]   (16 KiB),
// low half-word counts bytes 0..127, high half-word bytes 128..255.  A ds_add_u32 services lanes
// 0-31 and 32-63 in separate passes, and inside a pass every lane has its own column (= its own
// bank), so the skew of the data (Proba80: 80 % of the bytes hit one bin) costs nothing -- this is
// the "wavefront-privatised LDS histogram" of the north star without same-address serialisation.
// The 128x32 partial table is then summed with a bank-rotated read and reduced to
// (count[], max count, largest present symbol) exactly as HIST_count reports them.
#include "internal.h"

#define HIST_SEG (1u << 20)   // bytes per pass: a column (2 lanes) sees <= 2*(HIST_SEG/64+32) < 65536 hits

DEV void hist_add(u32* cnt, u32 col, u32 byte)
{
    // non-returning LDS atomic (ds_add_u32); row = byte&127, half selected by bit 7
    atomicAdd(&cnt[((byte & 127u) << 5) | col], 1u << ((byte >> 3) & 16u));
}
DEV void hist_add4(u32* cnt, u32 col, u32 w)
{
    hist_add(cnt, col, w & 0xFFu);
    hist_add(cnt, col, (w >> 8) & 0xFFu);
    hist_add(cnt, col, (w >> 16) & 0xFFu);
    hist_add(cnt, col, w >> 24);
}
DEV void hist_add16(u32* cnt, u32 col, const uint4& v)
{
    hist_add4(cnt, col, v.x); hist_add4(cnt, col, v.y); hist_add4(cnt, col, v.z); hist_add4(cnt, col, v.w);
}

__global__ __launch_bounds__(64) void k_hist(HistArgs a)
{
    __shared__ u32 cnt[128 * 32];
    const u32 lane = threadIdx.x;
    const u32 col = lane & 31u;
    const size_t b = blockIdx.x;
    const u8* const p = view_ptr(a.src, b);
    const size_t n = view_size(a.src, b);
    const unsigned limitIn = (a.maxSVs && !a.useUniformIn) ? a.maxSVs[b] : a.uniformMaxSV;
    const bool checked = limitIn < 255u && !a.trustInput;   // lib/hist.c:169 (HIST_countFast: trustInput, :141-159)
    const unsigned nOut = limitIn < 255u ? limitIn + 1u : 256u;    // entries the reference writes (hist.c:74,130)

    // totals of the four symbols this lane owns: lane, lane+64, lane+128, lane+192
    u32 t0 = 0, t1 = 0, t2 = 0, t3 = 0;

    for (size_t seg = 0; seg < n; seg += HIST_SEG) {
        for (u32 i = lane; i < 128u * 32u; i += 64u) cnt[i] = 0;
        __syncthreads();
        const u8* q = p + seg;
        const size_t len = (n - seg) < (size_t)HIST_SEG ? (n - seg) : (size_t)HIST_SEG;
        size_t head = (size_t)((0 - (uintptr_t)q) & 15u);
        if (head > len) head = len;
        if (lane < head) hist_add(cnt, col, q[lane]);
        const uint4* v = (const uint4*)(q + head);
        const size_t nvec = (len - head) >> 4;
        size_t i = lane;
        if (i + 192 < nvec) {                               // 4 x 1 KiB coalesced loads per group, one group ahead of its use
            uint4 x0 = v[i], x1 = v[i + 64], x2 = v[i + 128], x3 = v[i + 192];
            for (;;) {
                const size_t j = i + 256;
                const bool more = j + 192 < nvec;           // (per lane; lanes that stop early finish in the loop below)
                uint4 y0 = x0, y1 = x1, y2 = x2, y3 = x3;
                if (more) { y0 = v[j]; y1 = v[j + 64]; y2 = v[j + 128]; y3 = v[j + 192]; }
                __asm__ volatile("" ::: "memory");          // keep the next group's loads above this group's LDS updates
                hist_add16(cnt, col, x0); hist_add16(cnt, col, x1); hist_add16(cnt, col, x2); hist_add16(cnt, col, x3);
                i = j;
                if (!more) break;
                x0 = y0; x1 = y1; x2 = y2; x3 = y3;
            }
        }
        for (; i < nvec; i += 64) { const uint4 x = v[i]; hist_add16(cnt, col, x); }
        const size_t done = head + (nvec << 4);
        if (lane < len - done) hist_add(cnt, col, q[done + lane]);
        __syncthreads();
        // bank-rotated column sums: row `lane` -> symbols lane / lane+128, row lane+64 -> lane+64 / lane+192
        u32 lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
#pragma unroll 8
        for (u32 c = 0; c < 32; ++c) {
            const u32 cc = (c + lane) & 31u;
            const u32 x = cnt[(lane << 5) | cc];
            const u32 y = cnt[((lane + 64u) << 5) | cc];
            lo0 += x & 0xFFFFu; hi0 += x >> 16;
            lo1 += y & 0xFFFFu; hi1 += y >> 16;
        }
        t0 += lo0; t2 += hi0; t1 += lo1; t3 += hi1;
        __syncthreads();
    }

    // largest count and largest present symbol (hist.c:120-129)
    u32 best = t0 > t1 ? t0 : t1; best = t2 > best ? t2 : best; best = t3 > best ? t3 : best;
    int top = -1;
    if (t0) top = (int)lane;
    if (t1) top = (int)lane + 64;
    if (t2) top = (int)lane + 128;
    if (t3) top = (int)lane + 192;
    best = wave_max_u32(best);
    top = wave_max_i32(top);

    if (n == 0) {                                           // hist.c:83-87 / :38
        for (u32 s = lane; s < nOut; s += 64) a.counts[b * 256 + s] = 0;
        if (lane == 0) { if (a.maxSVs) a.maxSVs[b] = 0; a.results[b] = 0; }
        return;
    }
    if (checked && (unsigned)top > limitIn) {               // hist.c:128 : nothing else is written
        if (lane == 0) a.results[b] = FERR(maxSymbolValue_tooSmall);
        return;
    }
    unsigned* out = a.counts + b * 256;
    if (lane < nOut) out[lane] = t0;
    if (lane + 64 < nOut) out[lane + 64] = t1;
    if (lane + 128 < nOut) out[lane + 128] = t2;
    if (lane + 192 < nOut) out[lane + 192] = t3;
    if (lane == 0) { if (a.maxSVs) a.maxSVs[b] = (unsigned)top; a.results[b] = (size_t)best; }
}

__global__ __launch_bounds__(64) void k_zero_u32(u32* p, u32 n)
{
    for (u32 i = blockIdx.x * 64u + threadIdx.x; i < n; i += gridDim.x * 64u) p[i] = 0;
}
hipError_t launch_zero_u32(u32* p, u32 n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const u32 groups = (n + 63u) / 64u;
    hipLaunchKernelGGL(k_zero_u32, dim3(groups < 1024u ? groups : 1024u), dim3(64), 0, s, p, n);
    return hipGetLastError();
}

hipError_t launch_hist(const HistArgs& a, hipStream_t s)
{
    if (a.nBlocks == 0) return hipSuccess;
    probe_before(PK_HIST, s);
    hipLaunchKernelGGL(k_hist, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a);
    probe_after(PK_HIST, s);
    return hipGetLastError();
}
