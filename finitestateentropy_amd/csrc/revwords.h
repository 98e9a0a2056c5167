// revwords.h -- per-lane backwards stream reader over global memory with a deep register prefetch.
//
// Each lane of the "one lane per block" kernels walks its own block from high to low addresses.  A lane's
// loads cannot be coalesced with its neighbours' (they belong to other blocks), so what matters is that
// no load is ever on the critical path: the reader keeps the current 32 bytes in registers and the NEXT
// 32 bytes already in flight (two aligned 16-byte loads issued one full period -- 256 stream bits --
// before their first use), which covers an HBM miss (~900 cycles) at every symbol rate the codecs reach.
#pragma once
#include "dev_common.h"

struct RevWords8 {
    const uint4* p;       // next (lower) 32-byte group to fetch: chunks p[1] (high), p[0] (low)
    const uint4* pmin;    // lowest 16-byte chunk that may be touched
    u32 c[8];             // current words, c[7] is delivered next
    u32 n[8];             // prefetched group
    u32 left;             // words still undelivered in c[]

    DEV void fetch()      // load the group ending just below the last one into n[]
    {
        const uint4* hi = p + 1;
        const uint4* lo = p;
        if (hi < pmin) hi = pmin;
        if (lo < pmin) lo = pmin;
        const uint4 a = *hi, b = *lo;
        n[7] = a.w; n[6] = a.z; n[5] = a.y; n[4] = a.x; n[3] = b.w; n[2] = b.z; n[1] = b.y; n[0] = b.x;
        p -= 2;
    }
    // first delivered word = *topWord (4-byte aligned); nothing below the 16-byte chunk holding `lowest` is touched
    DEV void init(const u32* topWord, const void* lowest)
    {
        pmin = (const uint4*)((uintptr_t)lowest & ~(uintptr_t)15);
        const uint4* ch = (const uint4*)((uintptr_t)topWord & ~(uintptr_t)15);
        const u32 idx = (u32)(((uintptr_t)topWord >> 2) & 3u);     // word slot inside its 16-byte chunk
        const uint4 a = *ch;
        const uint4* lo = ch - 1; if (lo < pmin) lo = pmin;
        const uint4 b = *lo;
        // lay the 8 words out high -> low, then drop the (3-idx) words above topWord
        c[0] = b.x; c[1] = b.y; c[2] = b.z; c[3] = b.w; c[4] = a.x; c[5] = a.y; c[6] = a.z; c[7] = a.w;
        const u32 drop = 3u - idx;
#pragma unroll
        for (u32 r = 0; r < 3; ++r) {          // predicated shift-by-one, all indices static -> stays in registers
            const bool go = r < drop;
#pragma unroll
            for (int k = 7; k > 0; --k) c[k] = go ? c[k - 1] : c[k];
        }
        left = 8u - drop;
        p = ch - 3;
        fetch();
    }
    DEV u32 next()
    {
        if (left == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = n[k];
            fetch();
            left = 8;
        }
        const u32 w = c[7];
#pragma unroll
        for (int k = 7; k > 0; --k) c[k] = c[k - 1];
        --left;
        return w;
    }
};
