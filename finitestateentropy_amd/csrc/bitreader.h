// bitreader.h -- literal device restatement of the reference's LIFO bit reader (lib/bitstream.h:91-102,272-448).
#pragma once
#include "dev_common.h"

enum { BR_UNFINISHED = 0, BR_END_OF_BUFFER = 1, BR_COMPLETED = 2, BR_OVERFLOW = 3 };   // bitstream.h:99-102

DEV u64 ldg64u(const u8* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }           // unaligned global load

struct BitReader {                                   // bitstream.h:91-97
    const u8* base; size_t size; size_t at; u64 win; u32 used;
    DEV size_t init(const u8* src, size_t n)         // BIT_initDStream, bitstream.h:272-318
    {
        base = src; size = n; at = 0; win = 0; used = 0;
        if (n < 1) return FERR(srcSize_wrong);
        const u32 last = src[n - 1];
        if (n >= 8) {
            at = n - 8; win = ldg64u(src + at);
            if (last == 0) return FERR(GENERIC);
            used = 8 - hibit32(last);
        } else {
            for (size_t k = 0; k < n; ++k) win |= (u64)src[k] << (8 * k);
            if (last == 0) return FERR(corruption_detected);
            used = 8 - hibit32(last) + (u32)(8 - n) * 8;
        }
        return n;
    }
    DEV u32 read(u32 nb)                             // BIT_readBits (lookBits :345 + skipBits)
    {
        const u32 v = (u32)((win >> ((64u - used - nb) & 63u)) & (((u64)1 << nb) - 1));
        used += nb; return v;
    }
    DEV u32 read_fast(u32 nb)                        // BIT_readBitsFast (:361)
    {
        const u32 v = (u32)((win << (used & 63u)) >> ((64u - nb) & 63u));
        used += nb; return v;
    }
    DEV int reload()                                 // BIT_reloadDStream, :400-439
    {
        if (used > 64) return BR_OVERFLOW;
        if (at >= 8) { at -= used >> 3; used &= 7; win = ldg64u(base + at); return BR_UNFINISHED; }
        if (at == 0) return used < 64 ? BR_END_OF_BUFFER : BR_COMPLETED;
        u32 nbytes = used >> 3; int res = BR_UNFINISHED;
        if (at < nbytes) { nbytes = (u32)at; res = BR_END_OF_BUFFER; }
        at -= nbytes; used -= nbytes * 8; win = ldg64u(base + at);
        return res;
    }
};

DEV u32 fse_step(u32& state, BitReader& r, const u32* cells, bool fast)   // FSE_decodeSymbol(Fast), fse.h:600-622
{
    const u32 c = cells[state];
    const u32 nb = c >> 24;
    const u32 low = fast ? r.read_fast(nb) : r.read(nb);
    state = (c & 0xFFFFu) + low;
    return (c >> 16) & 0xFFu;
}

