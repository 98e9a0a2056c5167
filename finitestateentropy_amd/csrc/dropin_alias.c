/* dropin_alias.c -- libfse_dropin.so: the reference's OWN symbol names (lib/hist.h:30,46,54, lib/fse.h:76,90,104,119-163,174,222-247,315,335,341,
 * lib/huf.h:66,82,95,164,190,275-277,289-290, lib/error_public.h / fse.h:124-128) as real exported functions that forward to libfsehip.so.
 * include/fsehip.h renames at COMPILE time (FSEHIP_DROPIN_NAMES); this is the LINK-time form: an object file or application already
 * compiled against the reference's headers is relinked with `-lfse_dropin -lfsehip` instead of the reference's lib/ *.o and runs on
 * the MI355X unchanged.  It must not be linked beside the reference's own objects (same names), which is why it is a separate
 * library and libfsehip.so keeps its prefix. */
#define FSEHIP_INTERNAL
#include "../../include/fsehip.h"

#define DROPIN __attribute__((visibility("default")))

DROPIN unsigned FSE_isError(size_t c) { return FSEHIP_isError(c); }
DROPIN unsigned HUF_isError(size_t c) { return FSEHIP_isError(c); }
DROPIN unsigned HIST_isError(size_t c) { return FSEHIP_isError(c); }
DROPIN const char* FSE_getErrorName(size_t c) { return FSEHIP_getErrorName(c); }
DROPIN const char* HUF_getErrorName(size_t c) { return FSEHIP_getErrorName(c); }
DROPIN size_t FSE_compressBound(size_t n) { return FSEHIP_FSE_COMPRESSBOUND(n); }
DROPIN size_t HUF_compressBound(size_t n) { return FSEHIP_HUF_COMPRESSBOUND(n); }

DROPIN size_t HIST_count(unsigned* count, unsigned* msv, const void* src, size_t n) { return FSEHIP_HIST_count(count, msv, src, n); }
DROPIN size_t HIST_count_wksp(unsigned* count, unsigned* msv, const void* src, size_t n, void* ws, size_t wsn) { return FSEHIP_HIST_count_wksp(count, msv, src, n, ws, wsn); }
DROPIN size_t HIST_countFast(unsigned* count, unsigned* msv, const void* src, size_t n) { return FSEHIP_HIST_countFast(count, msv, src, n); }
DROPIN size_t HIST_countFast_wksp(unsigned* count, unsigned* msv, const void* src, size_t n, void* ws, size_t wsn) { return FSEHIP_HIST_countFast_wksp(count, msv, src, n, ws, wsn); }
DROPIN unsigned HIST_count_simple(unsigned* count, unsigned* msv, const void* src, size_t n) { return FSEHIP_HIST_count_simple(count, msv, src, n); }

DROPIN size_t FSE_compress(void* dst, size_t cap, const void* src, size_t n) { return FSEHIP_FSE_compress(dst, cap, src, n); }
DROPIN size_t FSE_compress2(void* dst, size_t cap, const void* src, size_t n, unsigned msv, unsigned tl) { return FSEHIP_FSE_compress2(dst, cap, src, n, msv, tl); }
DROPIN size_t FSE_compress_wksp(void* dst, size_t cap, const void* src, size_t n, unsigned msv, unsigned tl, void* ws, size_t wsn) { return FSEHIP_FSE_compress_wksp(dst, cap, src, n, msv, tl, ws, wsn); }
DROPIN size_t FSE_decompress(void* dst, size_t cap, const void* src, size_t n) { return FSEHIP_FSE_decompress(dst, cap, src, n); }
DROPIN size_t FSE_decompress_wksp(void* dst, size_t cap, const void* src, size_t n, FSEHIP_FSE_DTable* ws, unsigned maxLog) { return FSEHIP_FSE_decompress_wksp(dst, cap, src, n, ws, maxLog); }
DROPIN size_t FSE_compress_usingCTable(void* dst, size_t cap, const void* src, size_t n, const FSEHIP_FSE_CTable* ct) { return FSEHIP_FSE_compress_usingCTable(dst, cap, src, n, ct); }
DROPIN size_t FSE_decompress_usingDTable(void* dst, size_t cap, const void* src, size_t n, const FSEHIP_FSE_DTable* dt) { return FSEHIP_FSE_decompress_usingDTable(dst, cap, src, n, dt); }

/* the table glue of the advanced flow, lib/fse.h:119-163, :222-241, :341 */
DROPIN unsigned FSE_optimalTableLog(unsigned maxTl, size_t n, unsigned msv) { return FSEHIP_FSE_optimalTableLog(maxTl, n, msv); }
DROPIN size_t FSE_normalizeCount(short* norm, unsigned tl, const unsigned* count, size_t n, unsigned msv) { return FSEHIP_FSE_normalizeCount(norm, tl, count, n, msv); }
DROPIN size_t FSE_NCountWriteBound(unsigned msv, unsigned tl) { return FSEHIP_FSE_NCountWriteBound(msv, tl); }
DROPIN size_t FSE_writeNCount(void* buf, size_t cap, const short* norm, unsigned msv, unsigned tl) { return FSEHIP_FSE_writeNCount(buf, cap, norm, msv, tl); }
DROPIN size_t FSE_readNCount(short* norm, unsigned* msv, unsigned* tl, const void* src, size_t n) { return FSEHIP_FSE_readNCount(norm, msv, tl, src, n); }
DROPIN size_t FSE_buildCTable(FSEHIP_FSE_CTable* ct, const short* norm, unsigned msv, unsigned tl) { return FSEHIP_FSE_buildCTable(ct, norm, msv, tl); }
DROPIN size_t FSE_buildCTable_wksp(FSEHIP_FSE_CTable* ct, const short* norm, unsigned msv, unsigned tl, void* ws, size_t wsn) { return FSEHIP_FSE_buildCTable_wksp(ct, norm, msv, tl, ws, wsn); }
DROPIN size_t FSE_buildDTable(FSEHIP_FSE_DTable* dt, const short* norm, unsigned msv, unsigned tl) { return FSEHIP_FSE_buildDTable(dt, norm, msv, tl); }

DROPIN size_t HUF_compress(void* dst, size_t cap, const void* src, size_t n) { return FSEHIP_HUF_compress(dst, cap, src, n); }
DROPIN size_t HUF_compress2(void* dst, size_t cap, const void* src, size_t n, unsigned msv, unsigned tl) { return FSEHIP_HUF_compress2(dst, cap, src, n, msv, tl); }
DROPIN size_t HUF_compress4X_wksp(void* dst, size_t cap, const void* src, size_t n, unsigned msv, unsigned tl, void* ws, size_t wsn) { return FSEHIP_HUF_compress4X_wksp(dst, cap, src, n, msv, tl, ws, wsn); }
DROPIN size_t HUF_compress1X_wksp(void* dst, size_t cap, const void* src, size_t n, unsigned msv, unsigned tl, void* ws, size_t wsn) { return FSEHIP_HUF_compress1X_wksp(dst, cap, src, n, msv, tl, ws, wsn); }
DROPIN size_t HUF_compress1X(void* dst, size_t cap, const void* src, size_t n, unsigned msv, unsigned tl) { return FSEHIP_HUF_compress1X(dst, cap, src, n, msv, tl); }
DROPIN size_t HUF_decompress(void* dst, size_t orig, const void* src, size_t n) { return FSEHIP_HUF_decompress(dst, orig, src, n); }
DROPIN size_t HUF_decompress4X1_DCtx_wksp(FSEHIP_HUF_DTable* dctx, void* dst, size_t dn, const void* src, size_t n, void* ws, size_t wsn) { return FSEHIP_HUF_decompress4X1_DCtx_wksp(dctx, dst, dn, src, n, ws, wsn); }
/* the Huff0 table calls on the caller's statistics, lib/huf.h:204-218 */
DROPIN size_t HUF_buildCTable(FSEHIP_HUF_CElt* t, const unsigned* count, unsigned msv, unsigned maxNb) { return FSEHIP_HUF_buildCTable(t, count, msv, maxNb); }
DROPIN size_t HUF_buildCTable_wksp(FSEHIP_HUF_CElt* t, const unsigned* count, unsigned msv, unsigned maxNb, void* ws, size_t wsn) { return FSEHIP_HUF_buildCTable_wksp(t, count, msv, maxNb, ws, wsn); }
DROPIN size_t HUF_writeCTable(void* dst, size_t cap, const FSEHIP_HUF_CElt* t, unsigned msv, unsigned huffLog) { return FSEHIP_HUF_writeCTable(dst, cap, t, msv, huffLog); }
/* the header-reading single-symbol decoders, lib/huf.h:141-143,161-163,209-211,299-304 */
DROPIN size_t HUF_readDTableX1(FSEHIP_HUF_DTable* dt, const void* src, size_t n) { return FSEHIP_HUF_readDTableX1(dt, src, n); }
DROPIN size_t HUF_readDTableX1_wksp(FSEHIP_HUF_DTable* dt, const void* src, size_t n, void* ws, size_t wsn) { return FSEHIP_HUF_readDTableX1_wksp(dt, src, n, ws, wsn); }
DROPIN size_t HUF_decompress4X1(void* dst, size_t dn, const void* src, size_t n) { return FSEHIP_HUF_decompress4X1(dst, dn, src, n); }
DROPIN size_t HUF_decompress4X1_DCtx(FSEHIP_HUF_DTable* dctx, void* dst, size_t dn, const void* src, size_t n) { return FSEHIP_HUF_decompress4X1_DCtx(dctx, dst, dn, src, n); }
DROPIN size_t HUF_decompress1X1(void* dst, size_t dn, const void* src, size_t n) { return FSEHIP_HUF_decompress1X1(dst, dn, src, n); }
DROPIN size_t HUF_decompress1X1_DCtx(FSEHIP_HUF_DTable* dctx, void* dst, size_t dn, const void* src, size_t n) { return FSEHIP_HUF_decompress1X1_DCtx(dctx, dst, dn, src, n); }
DROPIN size_t HUF_decompress1X1_DCtx_wksp(FSEHIP_HUF_DTable* dctx, void* dst, size_t dn, const void* src, size_t n, void* ws, size_t wsn) { return FSEHIP_HUF_decompress1X1_DCtx_wksp(dctx, dst, dn, src, n, ws, wsn); }
DROPIN size_t HUF_compress1X_usingCTable(void* dst, size_t cap, const void* src, size_t n, const FSEHIP_HUF_CElt* ct) { return FSEHIP_HUF_compress1X_usingCTable(dst, cap, src, n, ct); }
DROPIN size_t HUF_compress4X_usingCTable(void* dst, size_t cap, const void* src, size_t n, const FSEHIP_HUF_CElt* ct) { return FSEHIP_HUF_compress4X_usingCTable(dst, cap, src, n, ct); }
DROPIN size_t HUF_decompress4X_usingDTable(void* dst, size_t cap, const void* src, size_t n, const FSEHIP_HUF_DTable* dt) { return FSEHIP_HUF_decompress4X_usingDTable(dst, cap, src, n, dt); }
DROPIN size_t HUF_decompress4X1_usingDTable(void* dst, size_t cap, const void* src, size_t n, const FSEHIP_HUF_DTable* dt) { return FSEHIP_HUF_decompress4X1_usingDTable(dst, cap, src, n, dt); }
DROPIN size_t HUF_decompress1X_usingDTable(void* dst, size_t cap, const void* src, size_t n, const FSEHIP_HUF_DTable* dt) { return FSEHIP_HUF_decompress1X_usingDTable(dst, cap, src, n, dt); }
DROPIN size_t HUF_decompress1X1_usingDTable(void* dst, size_t cap, const void* src, size_t n, const FSEHIP_HUF_DTable* dt) { return FSEHIP_HUF_decompress1X1_usingDTable(dst, cap, src, n, dt); }

/* lib/fseU16.h:44-60 */
DROPIN size_t FSE_countU16(unsigned* count, unsigned* msv, const unsigned short* src, size_t n) { return FSEHIP_FSE_countU16(count, msv, src, n); }
DROPIN size_t FSE_compressU16(void* dst, size_t cap, const unsigned short* src, size_t n, unsigned msv, unsigned tl) { return FSEHIP_FSE_compressU16(dst, cap, src, n, msv, tl); }
DROPIN size_t FSE_decompressU16(unsigned short* dst, size_t cap, const void* src, size_t n) { return FSEHIP_FSE_decompressU16(dst, cap, src, n); }
