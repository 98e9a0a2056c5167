/* oracle/fse_oracle.h -- TEST INFRASTRUCTURE ONLY (see fse_oracle.c header). */
#ifndef FSE_ORACLE_H
#define FSE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error convention of the reference: (size_t)-code, lib/error_private.h:77, lib/error_public.h:45-56 */
enum {
    ORC_E_GENERIC = 1, ORC_E_dstSize_tooSmall = 2, ORC_E_srcSize_wrong = 3, ORC_E_corruption_detected = 4,
    ORC_E_tableLog_tooLarge = 5, ORC_E_maxSymbolValue_tooLarge = 6, ORC_E_maxSymbolValue_tooSmall = 7,
    ORC_E_workSpace_tooSmall = 8, ORC_E_maxCode = 9
};
unsigned orc_is_error(size_t code);

/* workload definition: programs/probaGenerator.c:70-74,95-126 */
void orc_probagen_table(uint8_t table[4096], double p);
void orc_probagen_block(uint8_t* dst, size_t n, const uint8_t table[4096], uint32_t seed);
void orc_probagen_batch(uint8_t* dst, size_t stride, size_t n, size_t nBlocks, const uint8_t table[4096], uint32_t firstSeed);

/* a1: lib/hist.c:175 */
size_t orc_hist_count(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize);

/* g1..g3: lib/fse_compress.c:325-494,192-298,66-169; lib/entropy_common.c:41-144; lib/fse_decompress.c:71-126 */
unsigned orc_fse_optimal_tablelog(unsigned maxTableLog, size_t srcSize, unsigned maxSymbolValue, unsigned minus);
size_t orc_fse_normalize_count(short* norm, unsigned tableLog, const unsigned* count, size_t total, unsigned maxSymbolValue);
size_t orc_fse_ncount_write_bound(unsigned maxSymbolValue, unsigned tableLog);
size_t orc_fse_write_ncount(void* dst, size_t dstCapacity, const short* norm, unsigned maxSymbolValue, unsigned tableLog);
size_t orc_fse_read_ncount(short* norm, unsigned* maxSVPtr, unsigned* tableLogPtr, const void* src, size_t srcSize);
size_t orc_fse_build_ctable(uint32_t* ct, const short* norm, unsigned maxSymbolValue, unsigned tableLog);
size_t orc_fse_build_ctable_raw(uint32_t* ct, unsigned nbBits);
size_t orc_fse_build_dtable(uint32_t* dt, const short* norm, unsigned maxSymbolValue, unsigned tableLog);
size_t orc_fse_build_dtable_raw(uint32_t* dt, unsigned nbBits);
size_t orc_fse_build_dtable_rle(uint32_t* dt, uint8_t symbol);

/* a2, a3: lib/fse_compress.c:613, lib/fse_decompress.c:241 */
size_t orc_fse_compress_using_ctable(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const uint32_t* ct);
size_t orc_fse_decompress_using_dtable(void* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize, const uint32_t* dt);

/* g4: lib/fse_compress.c:632-698, lib/fse_decompress.c:255-283 */
size_t orc_fse_compress2(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog);
size_t orc_fse_decompress(void* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize);

/* g5, g6: lib/huf_compress.c:338-410,114-147; lib/entropy_common.c:154-215; lib/huf_decompress.c:118-185 */
size_t orc_huf_build_ctable(uint32_t* celt /* 256 x {u16 val; u8 nbBits; u8 pad} */, const unsigned* count, unsigned maxSymbolValue, unsigned maxNbBits);
size_t orc_huf_write_ctable(void* dst, size_t dstCapacity, const uint32_t* celt, unsigned maxSymbolValue, unsigned huffLog);
size_t orc_huf_read_stats(uint8_t* huffWeight, size_t hwSize, uint32_t* rankStats, uint32_t* nbSymbolsPtr, uint32_t* tableLogPtr, const void* src, size_t srcSize);
size_t orc_huf_read_dtable_x1(uint32_t* dtable /* [0] = desc, maxTableLog preset in byte 0 */, const void* src, size_t srcSize);

/* a4, a5: lib/huf_compress.c:546,605; lib/huf_decompress.c:406,980 */
size_t orc_huf_compress1x_using_ctable(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const uint32_t* celt);
size_t orc_huf_compress4x_using_ctable(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const uint32_t* celt);
size_t orc_huf_decompress1x1_using_dtable(void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize, const uint32_t* dtable);
size_t orc_huf_decompress4x1_using_dtable(void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize, const uint32_t* dtable);

/* g7: lib/huf_compress.c:637-798 (4 streams, no table reuse), lib/huf_decompress.c:1056 (X1 path) */
size_t orc_huf_compress2(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned huffLog);
size_t orc_huf_decompress(void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize);

/* batch helpers (OpenMP over blocks) for the tests and the "port" CPU baseline; codec 0 = FSE, 1 = Huff0.
 * return wall seconds. */
double orc_compress_batch(int codec, const uint8_t* src, size_t srcStride, size_t srcSize, uint8_t* dst, size_t dstStride,
                          size_t dstCapacity, uint64_t* results, size_t nBlocks, unsigned maxSymbolValue, unsigned tableLog, int nthreads);
double orc_decompress_batch(int codec, const uint8_t* cSrc, size_t cStride, const uint64_t* cSizes, uint8_t* dst,
                            size_t dstStride, size_t dstSize, uint64_t* results, size_t nBlocks, int nthreads);

/* ---- .fse frame (the container written by the reference's command-line tool, programs/fileio.c:266-285):
 *      magic (LE32) | block-size id | { block header | block }* | 3-byte end mark with a 22-bit XXH32 of the content */
size_t orc_frame_compress_bound(size_t srcSize, unsigned blockSizeId);
size_t orc_frame_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned blockSizeId, int codec /* 0 fse, 1 huf */);
size_t orc_frame_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);
uint32_t orc_xxh32(const void* data, size_t len, uint32_t seed);

/* XXH64 (public algorithm by Y. Collet; used only to check SURVEY Appendix B known-answer vectors) */
uint64_t orc_xxh64(const void* data, size_t len, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
