/* oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin batch driver around the UNMODIFIED reference library.  It is compiled together with the
 * reference's own lib/*.c (taken where they lie under /root/reference, see oracle/Makefile) into
 * oracle/_ref/libfse_ref.so.  Nothing from the reference is copied into this repository; this file
 * only *calls* the reference's public API (lib/fse.h, lib/huf.h, lib/hist.h).
 *
 * Uses: (1) pin our own restatement (oracle/fse_oracle.c) against the real thing,
 *       (2) generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py),
 *       (3) serve as bench.py's `cpu_baseline` with kind "reference".
 * It is never linked into, loaded by, or called from the product library.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FSE_STATIC_LINKING_ONLY
#define HUF_STATIC_LINKING_ONLY
#include "fse.h"
#include "huf.h"
#include "hist.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void set_threads(int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}

int ref_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* codec: 0 = FSE (FSE_compress2 / FSE_decompress), 1 = Huff0 (HUF_compress2 / HUF_decompress).
 * Blocks live at src + b*srcStride (srcSize bytes each); outputs at dst + b*dstStride.
 * Returns the wall time in seconds of the loop (programs/bench.c:353-364 is the serial analogue). */
double ref_compress_batch(int codec, const uint8_t* src, size_t srcStride, size_t srcSize,
                          uint8_t* dst, size_t dstStride, size_t dstCapacity,
                          uint64_t* results, size_t nBlocks,
                          unsigned maxSymbolValue, unsigned tableLog, int nthreads)
{
    double t0;
    long b;
    set_threads(nthreads);
    t0 = now_s();
#pragma omp parallel for schedule(static)
    for (b = 0; b < (long)nBlocks; b++) {
        const uint8_t* s = src + (size_t)b * srcStride;
        uint8_t* d = dst + (size_t)b * dstStride;
        size_t r = codec == 0 ? FSE_compress2(d, dstCapacity, s, srcSize, maxSymbolValue, tableLog)
                              : HUF_compress2(d, dstCapacity, s, srcSize, maxSymbolValue, tableLog);
        results[b] = (uint64_t)r;
    }
    return now_s() - t0;
}

/* cSizes[b] = compressed size of block b (as returned by the compressor, must be > 1). */
double ref_decompress_batch(int codec, const uint8_t* cSrc, size_t cStride, const uint64_t* cSizes,
                            uint8_t* dst, size_t dstStride, size_t dstSize,
                            uint64_t* results, size_t nBlocks, int nthreads)
{
    double t0;
    long b;
    set_threads(nthreads);
    t0 = now_s();
#pragma omp parallel for schedule(static)
    for (b = 0; b < (long)nBlocks; b++) {
        const uint8_t* s = cSrc + (size_t)b * cStride;
        uint8_t* d = dst + (size_t)b * dstStride;
        size_t r = codec == 0 ? FSE_decompress(d, dstSize, s, (size_t)cSizes[b])
                              : HUF_decompress(d, dstSize, s, (size_t)cSizes[b]);
        results[b] = (uint64_t)r;
    }
    return now_s() - t0;
}

double ref_hist_batch(const uint8_t* src, size_t srcStride, size_t srcSize,
                      unsigned* counts /* nBlocks x 256 */, unsigned* maxSymbolValues,
                      uint64_t* results, size_t nBlocks, int nthreads)
{
    double t0;
    long b;
    set_threads(nthreads);
    t0 = now_s();
#pragma omp parallel for schedule(static)
    for (b = 0; b < (long)nBlocks; b++) {
        unsigned msv = maxSymbolValues[b];
        results[b] = (uint64_t)HIST_count(counts + (size_t)b * 256, &msv, src + (size_t)b * srcStride, srcSize);
        maxSymbolValues[b] = msv;
    }
    return now_s() - t0;
}

/* sizeof(HUF_CElt) is private to huf_compress.c (struct {U16 val; BYTE nbBits;} -> 4 bytes);
 * expose what the tests need to size their buffers. */
size_t ref_sizeof_FSE_CTable_U32(unsigned tableLog, unsigned maxSymbolValue) { return FSE_CTABLE_SIZE_U32(tableLog, maxSymbolValue); }
size_t ref_sizeof_FSE_DTable_U32(unsigned tableLog) { return FSE_DTABLE_SIZE_U32(tableLog); }
size_t ref_FSE_compressBound(size_t n) { return FSE_COMPRESSBOUND(n); }
size_t ref_FSE_blockBound(size_t n) { return FSE_BLOCKBOUND(n); }
size_t ref_HUF_compressBound(size_t n) { return HUF_COMPRESSBOUND(n); }

/* multi-core CPU baseline driver (bench.py cpu_baseline, kind "reference") */
#define CPUB_NAME ref_bench_roundtrip
#define CPUB_STREAM_NAME ref_stream_bandwidth
#define CPUB_COMPRESS(codec, d, cap, s, n, msv, tl) ((codec) == 0 ? FSE_compress2(d, cap, s, n, msv, tl) : HUF_compress2(d, cap, s, n, msv, tl))
#define CPUB_DECOMPRESS(codec, d, n, s, cs) ((codec) == 0 ? FSE_decompress(d, n, s, cs) : HUF_decompress(d, n, s, cs))
#include "cpu_bench.h"
