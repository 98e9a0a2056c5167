// hist.hip -- a1: HIST_count over a batch of blocks (reference: lib/hist.c:66-133,163-180).
//
// One 64-lane wavefront per block.  The wave streams the block with coalesced 16-byte loads (4 KiB in flight ahead of the 4 KiB being
// counted) and counts into a wave-private LDS histogram   cnt[row = byte][col = lane & 15]   (256 x 16 words, 16 KiB: ten waves per CU).
// The skew of the data (Proba80: 80 % of the bytes hit one bin) spreads over the sixteen columns of the hot row -- sixteen banks -- and the
// four lanes that share a column are serviced by the LDS atomic unit; per byte the wave spends two VALU instructions (shift, v_and_or) and
// one ds_add_u32 with a constant increment.  That is what the layout is chosen for: the earlier 128 x 32 half-word table (row = byte & 127,
// count in the low or high half-word) was conflict-free but needed five VALU per byte, and the kernel was as much VALU- as memory-bound
// (round 5: Proba14 0.77 -> 0.68 ms per 100,000 blocks, Proba02 0.75 -> 0.61; 8 columns lose to conflicts, 32 columns to occupancy --
// EXPERIMENTS.md).  The 256 x 16 partial table is then summed with a bank-rotated read and reduced to (count[], max count, largest present
// symbol) exactly as HIST_count reports them.
//
// One input larger than a few blocks (the one-shot HIST_count of a whole buffer) is cut into 64 KiB pieces that are counted as a batch and
// folded by k_hist_fold -- a single wave would stream it at 2 GB/s.
#include "internal.h"

#define HIST_SEG (1u << 28)   // bytes per pass of the LDS table (a column counts at most a sixteenth of them: no word can wrap)

#ifndef HIST_COLS_LOG
#define HIST_COLS_LOG 4           // columns of the 256-row layout: 16 (16 KiB: 10 waves per CU; lanes l and l + 16 of one LDS pass share a column) or 32 (32 KiB: conflict-free, 5 waves)
#endif
#define HIST_COLS (1u << HIST_COLS_LOG)
#ifndef HIST_GROUP
#define HIST_GROUP 4              // KiB per load group (one group ahead of its use)
#endif
DEV void hist_add(u32* cnt, u32 col, u32 byte)
{
    // non-returning LDS atomic (ds_add_u32) on cnt[byte][lane & 15]: the address is a shift and a v_and_or of the source word, the increment a
    // constant -- two VALU instructions per byte
    atomicAdd(&cnt[(byte << HIST_COLS_LOG) | col], 1u);
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
    __shared__ __attribute__((aligned(16))) u32 cnt[256 * HIST_COLS];
    const u32 lane = threadIdx.x;
    const u32 col = lane & (HIST_COLS - 1u);
    const size_t b = blockIdx.x;
    const u8* const p = view_ptr(a.src, b);
    const size_t n = view_size(a.src, b);
    const unsigned limitIn = (a.maxSVs && !a.useUniformIn) ? a.maxSVs[b] : a.uniformMaxSV;
    const bool checked = limitIn < 255u && !a.trustInput;   // lib/hist.c:169 (HIST_countFast: trustInput, :141-159)
    const unsigned nOut = limitIn < 255u ? limitIn + 1u : 256u;    // entries the reference writes (hist.c:74,130)

    // totals of the four symbols this lane owns: lane, lane+64, lane+128, lane+192
    u32 t0 = 0, t1 = 0, t2 = 0, t3 = 0;

    for (size_t seg = 0; seg < n; seg += HIST_SEG) {
        for (u32 i = lane; i < sizeof(cnt) / 16u; i += 64u) ((uint4*)cnt)[i] = make_uint4(0, 0, 0, 0);      // (16-byte stores: a quarter of the LDS instructions)
        __syncthreads();
        const u8* q = p + seg;
        const size_t len = (n - seg) < (size_t)HIST_SEG ? (n - seg) : (size_t)HIST_SEG;
        size_t head = (size_t)((0 - (uintptr_t)q) & 15u);
        if (head > len) head = len;
        if (lane < head) hist_add(cnt, col, q[lane]);
        const uint4* v = (const uint4*)(q + head);
        const size_t nvec = (len - head) >> 4;
        size_t i = lane;
        constexpr size_t HG = HIST_GROUP;                   // 1 KiB coalesced loads per group; one group is in flight while the one before it is counted
        if (i + 64 * (HG - 1) < nvec) {
            uint4 x[HG];
#pragma unroll
            for (size_t g = 0; g < HG; ++g) x[g] = v[i + 64 * g];
            for (;;) {
                const size_t j = i + 64 * HG;
                const bool more = j + 64 * (HG - 1) < nvec;  // (per lane; lanes that stop early finish in the loop below)
                uint4 y[HG];
#pragma unroll
                for (size_t g = 0; g < HG; ++g) y[g] = x[g];
                if (more) {
#pragma unroll
                    for (size_t g = 0; g < HG; ++g) y[g] = v[j + 64 * g];
                }
                __asm__ volatile("" ::: "memory");          // keep the next group's loads above this group's LDS updates
#pragma unroll
                for (size_t g = 0; g < HG; ++g) hist_add16(cnt, col, x[g]);
                i = j;
                if (!more) break;
#pragma unroll
                for (size_t g = 0; g < HG; ++g) x[g] = y[g];
            }
        }
        for (; i < nvec; i += 64) { const uint4 x = v[i]; hist_add16(cnt, col, x); }
        const size_t done = head + (nvec << 4);
        if (lane < len - done) hist_add(cnt, col, q[done + lane]);
        __syncthreads();
        // bank-rotated column sums: rows lane, lane + 64, lane + 128, lane + 192 (16 words each)
        u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll 8
        for (u32 c = 0; c < HIST_COLS; ++c) {
            const u32 cc = (c + lane) & (HIST_COLS - 1u);
            s0 += cnt[(lane << HIST_COLS_LOG) | cc]; s1 += cnt[((lane + 64u) << HIST_COLS_LOG) | cc];
            s2 += cnt[((lane + 128u) << HIST_COLS_LOG) | cc]; s3 += cnt[((lane + 192u) << HIST_COLS_LOG) | cc];
        }
        t0 += s0; t1 += s1; t2 += s2; t3 += s3;
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

// ---- one large input: pieces counted as a batch, folded here (hist.c:120-131 over the sum) ----------------------------------------
__global__ __launch_bounds__(1024) void k_hist_fold(const unsigned* __restrict__ part, size_t nPart, unsigned limitIn, int checked, size_t n,
                                                   unsigned* count, unsigned* maxSV, size_t* result)
{
    __shared__ u32 sums[4][256];
    __shared__ u32 wbest[4];
    __shared__ int wtop[4];
    const u32 t = threadIdx.x, sym = t & 255u, q = t >> 8;
    u32 acc = 0;
    for (size_t p = q; p < nPart; p += 4) acc += part[p * 256 + sym];
    sums[q][sym] = acc;
    __syncthreads();
    if (t >= 256) return;
    const u32 c = sums[0][t] + sums[1][t] + sums[2][t] + sums[3][t];
    const u32 best = wave_max_u32(c);
    const int top = wave_max_i32(c ? (int)t : -1);
    if ((t & 63u) == 0) { wbest[t >> 6] = best; wtop[t >> 6] = top; }
    __syncthreads();
    u32 b = wbest[0]; int tp = wtop[0];
    for (int w = 1; w < 4; ++w) { b = wbest[w] > b ? wbest[w] : b; tp = wtop[w] > tp ? wtop[w] : tp; }
    const unsigned nOut = limitIn < 255u ? limitIn + 1u : 256u;
    if (checked && (unsigned)tp > limitIn) { if (t == 0) *result = FERR(maxSymbolValue_tooSmall); return; }
    (void)n;
    if (t < nOut) count[t] = c;
    if (t == 0) { *maxSV = (unsigned)tp; *result = (size_t)b; }
}

hipError_t launch_hist_large(const u8* d_src, size_t n, unsigned limitIn, int trustInput, unsigned* d_part, unsigned* d_count, unsigned* d_maxSV,
                             size_t* d_result, size_t* d_scratchResults, hipStream_t s)
{
    const size_t nFull = n / HIST_PIECE, tail = n % HIST_PIECE, nPart = nFull + (tail ? 1 : 0);
    HistArgs a;
    a.counts = d_part; a.maxSVs = nullptr; a.uniformMaxSV = 255; a.useUniformIn = 1; a.trustInput = 1; a.results = d_scratchResults;
    a.src = BlockView{d_src, HIST_PIECE, nullptr, HIST_PIECE, nullptr}; a.nBlocks = nFull;
    hipError_t e = launch_hist(a, s);
    if (e != hipSuccess) return e;
    if (tail) {
        a.counts = d_part + nFull * 256; a.results = d_scratchResults + nFull;
        a.src = BlockView{d_src + nFull * HIST_PIECE, tail, nullptr, tail, nullptr}; a.nBlocks = 1;
        e = launch_hist(a, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_hist_fold, dim3(1), dim3(1024), 0, s, (const unsigned*)d_part, nPart, limitIn, (int)(limitIn < 255u && !trustInput), n,
                       d_count, d_maxSV, d_result);
    return hipGetLastError();
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
