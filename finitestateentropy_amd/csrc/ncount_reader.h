// ncount_reader.h -- parser of the NCount header (the table description in front of every FSE payload; format SURVEY A.4,
// behaviour of lib/entropy_common.c:41-144 including what it does on truncated and garbage input).
//
// The format is a serial variable-length code: the width of every field depends on the points still unassigned, i.e. on all
// fields before it, so one lane parses one header (the callers map 64 headers to the 64 lanes of a wave).  What has to be
// reproduced exactly is the reference's window discipline, because it defines the result on corrupt input:
//   * fields are taken from the low end of a 32-bit little-endian window at byte `at`, of which `skip` bits are spent;
//   * the window slides forward by whole bytes only while that keeps it inside the buffer (at <= end - 4); near the end the
//     byte position freezes and `skip` grows instead -- bits shifted in from above the window are zeros -- and a header
//     whose fields end more than 32 bits into the last window is corrupt;
//   * inside a zero-run the 2-bit codes are taken from the register without touching the window, and a 16-bit repeat code
//     moves the window by two bytes or, within the last five bytes, only shifts the register.
// `HeaderWindow` holds that state; the three ways of advancing are its three methods.  The field loop below is ours.
#pragma once
#include "dev_common.h"

DEV u32 hw_ld32(const u8* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }    // unaligned little-endian load

struct HeaderWindow {
    const u8* base; long at, end; u32 bits; int skip;
    DEV void open(const u8* p, long n) { base = p; at = 0; end = n; skip = 0; bits = hw_ld32(p); }
    DEV bool can_slide() const { return at <= end - 7 || at + (skip >> 3) <= end - 4; }
    // after a counter field of `k` bits: slide by the whole bytes spent, or freeze on the last window
    DEV void spend_field(int k)
    {
        skip += k;
        if (can_slide()) { at += skip >> 3; skip &= 7; }
        else { skip -= (int)(8 * (end - 4 - at)); at = end - 4; }
        bits = hw_ld32(base + at) >> (skip & 31);
    }
    // 2-bit run codes are consumed from the register alone ...
    DEV void spend_in_register(int k) { bits >>= k; skip += k; }
    // ... and the run ends with a slide when there is room, else with the register shifted past its last code
    DEV void end_of_run()
    {
        skip += 2;
        if (can_slide()) { at += skip >> 3; skip &= 7; bits = hw_ld32(base + at) >> skip; }
        else bits >>= 2;
    }
    // 16-bit repeat code: two bytes forward, or (last five bytes) a register shift
    DEV void spend_repeat16()
    {
        if (at < end - 5) { at += 2; bits = hw_ld32(base + at) >> skip; }
        else { bits >>= 16; skip += 16; }
    }
};

// Parses the header at `in` (n >= 4 bytes readable).  norm[0 .. maxSV] receives the counters (-1 = "less than one point"),
// *maxSVPtr (in: alphabet limit) the last symbol described, *tlPtr the table log.  Returns the header size or an error code.
template <int STRIDE>
DEV size_t ncount_parse(s16* norm, u32* maxSVPtr, u32* tlPtr, const u8* in, long n)
{
    const u32 limit = *maxSVPtr;
    for (u32 s = 0; s <= limit; ++s) norm[s * STRIDE] = 0;       // symbols the header does not mention have no points
    HeaderWindow w; w.open(in, n);
    const u32 tl = (w.bits & 15u) + FSEHIP_FSE_MIN_TABLELOG;
    if (tl > 15u) return FERR(tableLog_tooLarge);               // FSE_TABLELOG_ABSOLUTE_MAX
    *tlPtr = tl;
    w.spend_in_register(4);
    int left = (1 << tl) + 1;                                    // points still unassigned, plus one
    u32 sym = 0;
    bool afterZero = false;
    while (left > 1 && sym <= limit) {
        if (afterZero) {                                         // run of further zeros: 24 per 0xFFFF, 3 per '11', then 0..2
            u32 to = sym;
            while ((w.bits & 0xFFFFu) == 0xFFFFu) { to += 24; w.spend_repeat16(); }
            while ((w.bits & 3u) == 3u) { to += 3; w.spend_in_register(2); }
            to += w.bits & 3u;
            if (to > limit) return FERR(maxSymbolValue_tooSmall);
            sym = to;                                            // (their counters are zero already)
            w.end_of_run();
        }
        // one counter: values below `spare` take one bit less (threshold = the power of two at or below `left`)
        const int threshold = 1 << hibit32((u32)left), spare = 2 * threshold - 1 - left;
        const int width = (int)hibit32((u32)left) + 1;
        int v = (int)(w.bits & (u32)(threshold - 1)), used = width - 1;
        if (v >= spare) {
            v = (int)(w.bits & (u32)(2 * threshold - 1));
            if (v >= threshold) v -= spare;
            used = width;
        }
        --v;                                                     // stored value is counter + 1
        left -= v < 0 ? -v : v;
        norm[sym * STRIDE] = (s16)v;
        ++sym;
        afterZero = v == 0;
        w.spend_field(used);
    }
    if (left != 1 || w.skip > 32) return FERR(corruption_detected);
    *maxSVPtr = sym - 1;
    return (size_t)(w.at + ((w.skip + 7) >> 3));
}

// any length: headers shorter than 4 bytes are parsed from a zero-extended copy and must not claim more than they have
template <int STRIDE>
DEV size_t ncount_read(s16* norm, u32* maxSVPtr, u32* tlPtr, const u8* in, size_t n)
{
    if (n >= 4) return ncount_parse<STRIDE>(norm, maxSVPtr, tlPtr, in, (long)n);
    u8 tmp[4] = { 0, 0, 0, 0 };
    for (size_t i = 0; i < n; ++i) tmp[i] = in[i];
    const size_t r = ncount_parse<STRIDE>(norm, maxSVPtr, tlPtr, tmp, 4);
    if (!is_err(r) && r > n) return FERR(corruption_detected);
    return r;
}
