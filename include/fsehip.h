/* fsehip.h -- C ABI of libfsehip.so: the MI355X (gfx950) block-entropy-coding hot path.
 *
 * Drop-in boundary for the reference library's block API (Cyan4973/FiniteStateEntropy):
 *   lib/hist.h:30-31      HIST_count
 *   lib/fse.h:67-105      FSE_compress / FSE_decompress / FSE_compress2 (one-shot block API)
 *   lib/fse.h:142-247     FSE_compress_usingCTable (:174), FSE_decompress_usingDTable (:247)
 *   lib/huf.h:54-98       HUF_compress / HUF_decompress / HUF_compress2
 *   lib/huf.h:190,290     HUF_compress4X_usingCTable, HUF_compress1X_usingCTable
 *   lib/huf.h:275-277     HUF_decompress4X_usingDTable, HUF_decompress4X1_usingDTable
 *
 * Every function keeps the reference's signature, argument meaning, in-memory table layouts
 * (FSE_CTable / FSE_DTable = unsigned[], HUF_CElt = {U16 val; BYTE nbBits;} stride 4,
 * HUF_DTable = U32[] with a 4-byte DTableDesc), bitstream/header format and the size_t error
 * convention (lib/error_private.h:77-79: (size_t)-code, code in FSEHIP_ErrorCode).  Encoders
 * return 0 ("not compressible / does not fit") and, for the one-shot forms, 1 ("single repeated
 * byte") exactly where the reference does (lib/fse.h:62-65, lib/huf.h:50-52).
 *
 * All computation happens in hand-written HIP kernels; there is NO CPU fallback: if no gfx950
 * device is usable the single-block calls return FSEHIP_ERROR(GENERIC) and the batch calls return
 * the failing hipError_t.
 *
 * Two layers:
 *   1. single-block calls on HOST pointers (same signatures as the reference, prefix FSEHIP_;
 *      define FSEHIP_DROPIN_NAMES before including this header to also get the original names
 *      as macros so reference callers such as programs/fuzzer.c or programs/fullbench.c link
 *      against this library unchanged);
 *   2. batched calls on DEVICE pointers (many independent blocks per launch -- the data-parallel
 *      axis of programs/bench.c:353-364,389-424), which is what reaches throughput.
 */
#ifndef FSEHIP_H
#define FSEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSEHIP_API __attribute__((visibility("default")))

/* ---- error convention (lib/error_public.h:45-56, lib/error_private.h:77-79) ------------------ */
typedef enum {
    FSEHIP_error_no_error = 0,
    FSEHIP_error_GENERIC = 1,
    FSEHIP_error_dstSize_tooSmall = 2,
    FSEHIP_error_srcSize_wrong = 3,
    FSEHIP_error_corruption_detected = 4,
    FSEHIP_error_tableLog_tooLarge = 5,
    FSEHIP_error_maxSymbolValue_tooLarge = 6,
    FSEHIP_error_maxSymbolValue_tooSmall = 7,
    FSEHIP_error_workSpace_tooSmall = 8,
    FSEHIP_error_maxCode = 9
} FSEHIP_ErrorCode;
#define FSEHIP_ERROR(name) ((size_t)0 - (size_t)FSEHIP_error_##name)
FSEHIP_API unsigned FSEHIP_isError(size_t code);            /* FSE_isError / HUF_isError / HIST_isError */
FSEHIP_API const char* FSEHIP_getErrorName(size_t code);    /* FSE_getErrorName */

/* ---- sizes (lib/fse.h:290-296, lib/huf.h:131-145) --------------------------------------------- */
#define FSEHIP_FSE_MAX_TABLELOG 12
#define FSEHIP_FSE_DEFAULT_TABLELOG 11
#define FSEHIP_FSE_MIN_TABLELOG 5
#define FSEHIP_FSE_NCOUNTBOUND 512
#define FSEHIP_FSE_BLOCKBOUND(size) ((size) + ((size) >> 7) + 4 + sizeof(size_t))
#define FSEHIP_FSE_COMPRESSBOUND(size) (FSEHIP_FSE_NCOUNTBOUND + FSEHIP_FSE_BLOCKBOUND(size))
#define FSEHIP_FSE_CTABLE_SIZE_U32(maxTableLog, maxSymbolValue) (1 + (1 << ((maxTableLog)-1)) + (((maxSymbolValue) + 1) * 2))
#define FSEHIP_FSE_DTABLE_SIZE_U32(maxTableLog) (1 + (1 << (maxTableLog)))
#define FSEHIP_FSE_WKSP_SIZE_U32(maxTableLog, maxSymbolValue) (FSEHIP_FSE_CTABLE_SIZE_U32(maxTableLog, maxSymbolValue) + (((maxTableLog) > 12) ? (1 << ((maxTableLog)-2)) : 1024))   /* lib/fse.h:314 */
#define FSEHIP_HIST_WKSP_SIZE_U32 1024                                             /* lib/hist.h:38-39 */
#define FSEHIP_HIST_WKSP_SIZE (FSEHIP_HIST_WKSP_SIZE_U32 * sizeof(unsigned))
#define FSEHIP_HUF_WORKSPACE_SIZE ((6 << 10) + 256)                                /* lib/huf.h:93-94 */
#define FSEHIP_HUF_WORKSPACE_SIZE_U32 (FSEHIP_HUF_WORKSPACE_SIZE / sizeof(uint32_t))
#define FSEHIP_HUF_DECOMPRESS_WORKSPACE_SIZE (2 << 10)                             /* lib/huf.h:263 */
#define FSEHIP_HUF_TABLELOG_MAX 12
#define FSEHIP_HUF_TABLELOG_DEFAULT 11
#define FSEHIP_HUF_BLOCKSIZE_MAX (128 * 1024)
#define FSEHIP_HUF_COMPRESSBOUND(size) (129 + ((size) + ((size) >> 8) + 8))
#define FSEHIP_HUF_CTABLE_SIZE_U32(maxSymbolValue) ((maxSymbolValue) + 1)
#define FSEHIP_HUF_DTABLE_SIZE_U32(maxTableLog) (1 + (1 << (maxTableLog)))

typedef unsigned FSEHIP_FSE_CTable;   /* lib/fse.h:160 */
typedef unsigned FSEHIP_FSE_DTable;   /* lib/fse.h:233 */
typedef uint32_t FSEHIP_HUF_CElt;     /* lib/huf.h:187 + lib/huf_compress.c:106-109 : {U16 val; BYTE nbBits; pad} */
typedef uint32_t FSEHIP_HUF_DTable;   /* lib/huf.h:144 */

/* =================================================================================================
 *  Layer 1 -- single-block, HOST pointers, reference signatures
 * ================================================================================================= */
/* lib/hist.h:30  (semantics lib/hist.c:163-180) */
FSEHIP_API size_t FSEHIP_HIST_count(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize);

/* lib/hist.h:46, :54.  The _wksp forms below are what the reference's own callers go through (SURVEY 8(b) "what calls it": fse_compress.c:652,
 * huf_compress.c:672, huf_decompress.c:428).  Their workspaces are validated exactly as the reference validates them -- same checks, same order, same
 * error codes -- and then left alone: every table and counter lives in device memory.  HIST_countFast is the unchecked variant: a limit below
 * 255 bounds the entries written, larger symbols are counted into the result and *maxSymbolValuePtr but are no error (lib/hist.c:120-131). */
FSEHIP_API size_t FSEHIP_HIST_count_wksp(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize, void* workSpace, size_t workSpaceSize);
FSEHIP_API size_t FSEHIP_HIST_countFast(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize);
/* lib/hist.h:62 (below 1500 bytes the workspace is not looked at, lib/hist.c:141-150) and :74 (returns the largest count as `unsigned`) */
FSEHIP_API size_t FSEHIP_HIST_countFast_wksp(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize, void* workSpace, size_t workSpaceSize);
FSEHIP_API unsigned FSEHIP_HIST_count_simple(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize);

/* lib/fse.h:174 */
FSEHIP_API size_t FSEHIP_FSE_compress_usingCTable(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const FSEHIP_FSE_CTable* ct);
/* lib/fse.h:247 */
FSEHIP_API size_t FSEHIP_FSE_decompress_usingDTable(void* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize, const FSEHIP_FSE_DTable* dt);
/* lib/fse.h:76, :104, :90 */
FSEHIP_API size_t FSEHIP_FSE_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);
FSEHIP_API size_t FSEHIP_FSE_compress2(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog);
FSEHIP_API size_t FSEHIP_FSE_decompress(void* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize);
/* lib/fse.h:315 (wkspSize in bytes against FSE_WKSP_SIZE_U32, as lib/fse_compress.c:646 compares them; a table log above 12 codes like 12,
 * :340) and lib/fse.h:335 (maxLog honoured, limits above FSE_MAX_TABLELOG count as 12; `workSpace`, when given, receives the DTable the
 * reference would have built there) */
FSEHIP_API size_t FSEHIP_FSE_compress_wksp(void* dst, size_t dstSize, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog, void* workSpace, size_t wkspSize);
FSEHIP_API size_t FSEHIP_FSE_decompress_wksp(void* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize, FSEHIP_FSE_DTable* workSpace, unsigned maxLog);
/* The steps between the histogram and the hot loops under their own names -- the reference's "advanced" flow (lib/fse.h:107-163, :218-241: count,
 * FSE_optimalTableLog, FSE_normalizeCount, FSE_writeNCount, FSE_buildCTable, FSE_compress_usingCTable / FSE_readNCount, FSE_buildDTable,
 * FSE_decompress_usingDTable), each a batch of one on the device routines the one-shot calls run (csrc/wave_glue.h, ncount_reader.h,
 * fse_wave_build.h); FSE_optimalTableLog and FSE_NCountWriteBound are arithmetic on their arguments (lib/fse_compress.c:186-190, :325-347).
 * lib/fse.h:119, :137, :144, :150, :162 (+ :341 the _wksp form: workspace checked as lib/fse_compress.c:86 checks it, then left alone), :222, :240.
 * Where the reference is undefined the call refuses: normalised counters that do not add up to 1 << tableLog cells or go below -1 (the CTable
 * builder asserts, lib/fse_compress.c:127; the DTable builder returns GENERIC, lib/fse_decompress.c:107) -> GENERIC from both builders;
 * tableLog 0, 1 or 3 -> GENERIC (FSE_TABLESTEP(2) and (8) are even, lib/fse.h:683: the reference's spread never leaves cell 0 and the rest of its table is
 * whatever the memory held; 2, 4 and everything from FSE_MIN_TABLELOG up are exact); maxSymbolValue > 255 -> maxSymbolValue_tooLarge (byte alphabets; FSE_buildDTable says so itself, :83).  A failed
 * FSE_writeNCount / FSE_readNCount leaves its output untouched (the reference may have written part of it). */
FSEHIP_API unsigned FSEHIP_FSE_optimalTableLog(unsigned maxTableLog, size_t srcSize, unsigned maxSymbolValue);
FSEHIP_API size_t FSEHIP_FSE_normalizeCount(short* normalizedCounter, unsigned tableLog, const unsigned* count, size_t srcSize, unsigned maxSymbolValue);
FSEHIP_API size_t FSEHIP_FSE_NCountWriteBound(unsigned maxSymbolValue, unsigned tableLog);
FSEHIP_API size_t FSEHIP_FSE_writeNCount(void* buffer, size_t bufferSize, const short* normalizedCounter, unsigned maxSymbolValue, unsigned tableLog);
FSEHIP_API size_t FSEHIP_FSE_readNCount(short* normalizedCounter, unsigned* maxSymbolValuePtr, unsigned* tableLogPtr, const void* rBuffer, size_t rBuffSize);
FSEHIP_API size_t FSEHIP_FSE_buildCTable(FSEHIP_FSE_CTable* ct, const short* normalizedCounter, unsigned maxSymbolValue, unsigned tableLog);
FSEHIP_API size_t FSEHIP_FSE_buildCTable_wksp(FSEHIP_FSE_CTable* ct, const short* normalizedCounter, unsigned maxSymbolValue, unsigned tableLog, void* workSpace, size_t wkspSize);
FSEHIP_API size_t FSEHIP_FSE_buildDTable(FSEHIP_FSE_DTable* dt, const short* normalizedCounter, unsigned maxSymbolValue, unsigned tableLog);

/* lib/huf.h:290, :190 */
FSEHIP_API size_t FSEHIP_HUF_compress1X_usingCTable(void* dst, size_t dstSize, const void* src, size_t srcSize, const FSEHIP_HUF_CElt* CTable);
FSEHIP_API size_t FSEHIP_HUF_compress4X_usingCTable(void* dst, size_t dstSize, const void* src, size_t srcSize, const FSEHIP_HUF_CElt* CTable);
/* lib/huf.h:275-277.  HUF_decompress4X_usingDTable dispatches on the table type like lib/huf_decompress.c:980-997: tableType 0 = X1
 * (single-symbol) cells, tableType 1 = X2 (double-symbol) cells from the reference's HUF_readDTableX2 -- both accepted.
 * HUF_decompress4X1_usingDTable rejects an X2 table with GENERIC exactly as the reference does (lib/huf_decompress.c:411-412). */
FSEHIP_API size_t FSEHIP_HUF_decompress4X_usingDTable(void* dst, size_t maxDstSize, const void* cSrc, size_t cSrcSize, const FSEHIP_HUF_DTable* DTable);
FSEHIP_API size_t FSEHIP_HUF_decompress4X1_usingDTable(void* dst, size_t maxDstSize, const void* cSrc, size_t cSrcSize, const FSEHIP_HUF_DTable* DTable);
/* lib/huf.h:318-320: single-stream blocks, what HUF_compress1X_usingCTable writes (lib/huf_decompress.c:239-260, 961-975).  The 1X1 form
 * rejects a double-symbol table with GENERIC (:367-369), the 1X form takes both table types like its 4X sibling. */
FSEHIP_API size_t FSEHIP_HUF_decompress1X_usingDTable(void* dst, size_t maxDstSize, const void* cSrc, size_t cSrcSize, const FSEHIP_HUF_DTable* DTable);
FSEHIP_API size_t FSEHIP_HUF_decompress1X1_usingDTable(void* dst, size_t maxDstSize, const void* cSrc, size_t cSrcSize, const FSEHIP_HUF_DTable* DTable);
/* lib/huf.h:66, :95, :82 */
FSEHIP_API size_t FSEHIP_HUF_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);
FSEHIP_API size_t FSEHIP_HUF_compress2(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog);
FSEHIP_API size_t FSEHIP_HUF_decompress(void* dst, size_t originalSize, const void* cSrc, size_t cSrcSize);
/* lib/huf.h:95, :289 (workSpace: 4-byte aligned, at least HUF_WORKSPACE_SIZE bytes -- GENERIC / workSpace_tooSmall otherwise, lib/huf_compress.c:654-655;
 * the 1X form writes a single stream without jump table) and lib/huf.h:164 (dctx: a DTable whose descriptor holds the table-log limit, as
 * HUF_CREATE_STATIC_DTABLEX1 leaves it; it receives the table read from the block's header, lib/huf_decompress.c:417-431) */
FSEHIP_API size_t FSEHIP_HUF_compress4X_wksp(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog, void* workSpace, size_t wkspSize);
FSEHIP_API size_t FSEHIP_HUF_compress1X(void* dst, size_t dstSize, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog);   /* lib/huf.h:288: HUF_compress1X_wksp with a workspace of its own */
FSEHIP_API size_t FSEHIP_HUF_compress1X_wksp(void* dst, size_t dstSize, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog, void* workSpace, size_t wkspSize);
FSEHIP_API size_t FSEHIP_HUF_decompress4X1_DCtx_wksp(FSEHIP_HUF_DTable* dctx, void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize, void* workSpace, size_t wkspSize);
/* the rest of the single-symbol family (lib/huf.h:141-143 HUF_decompress4X1, :161-163 its DCtx form, :209-211 HUF_readDTableX1[_wksp], :299-304 the 1X1
 * forms; lib/huf_decompress.c:118-192, :377-404, :439-452): the header's table into the caller's DTable / DCtx, then the four streams (4X1) or the single
 * stream (1X1) behind it.  Workspaces are checked as the reference checks them ((16 + 64) words, tableLog_tooLarge) and then left alone. */
/* lib/huf.h:204-218 (lib/huf_compress.c:113-148, :334-421): the compress-side table calls on the caller's own statistics -- HUF_buildCTable[_wksp] from
 * count[0 .. maxSymbolValue] (returns the table log; tree[0 .. maxSymbolValue] receives the codes), HUF_writeCTable from such a table (returns the header
 * size).  Refused where the reference is undefined: no symbol in use, a count of 2^23 or more (blocks are at most HUF_BLOCKSIZE_MAX = 128 KB), more
 * symbols in use than codes of maxNbBits bits, a code length above huffLog or a huffLog above 12 in HUF_writeCTable -> GENERIC. */
FSEHIP_API size_t FSEHIP_HUF_buildCTable(FSEHIP_HUF_CElt* tree, const unsigned* count, unsigned maxSymbolValue, unsigned maxNbBits);
FSEHIP_API size_t FSEHIP_HUF_buildCTable_wksp(FSEHIP_HUF_CElt* tree, const unsigned* count, unsigned maxSymbolValue, unsigned maxNbBits, void* workSpace, size_t wkspSize);
FSEHIP_API size_t FSEHIP_HUF_writeCTable(void* dst, size_t maxDstSize, const FSEHIP_HUF_CElt* CTable, unsigned maxSymbolValue, unsigned huffLog);
FSEHIP_API size_t FSEHIP_HUF_readDTableX1(FSEHIP_HUF_DTable* DTable, const void* src, size_t srcSize);
FSEHIP_API size_t FSEHIP_HUF_readDTableX1_wksp(FSEHIP_HUF_DTable* DTable, const void* src, size_t srcSize, void* workSpace, size_t wkspSize);
FSEHIP_API size_t FSEHIP_HUF_decompress4X1(void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize);
FSEHIP_API size_t FSEHIP_HUF_decompress4X1_DCtx(FSEHIP_HUF_DTable* dctx, void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize);
FSEHIP_API size_t FSEHIP_HUF_decompress1X1(void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize);
FSEHIP_API size_t FSEHIP_HUF_decompress1X1_DCtx(FSEHIP_HUF_DTable* dctx, void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize);
FSEHIP_API size_t FSEHIP_HUF_decompress1X1_DCtx_wksp(FSEHIP_HUF_DTable* dctx, void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize, void* workSpace, size_t wkspSize);

/* =================================================================================================
 *  Layer 2 -- batched, DEVICE pointers.  Block b lives at base + b*stride.  `d_sizes` may be
 *  NULL, in which case every block has `uniformSize` bytes.  `d_results[b]` receives exactly what
 *  the corresponding single-block reference call would return for block b.  `stream` is a
 *  hipStream_t (NULL = default stream).  Return value: 0 (hipSuccess) or a hipError_t.
 *  Launches are asynchronous on `stream`; the library never allocates on these paths (one exception, made once per device: FSEHIP_prepareDevice below).
 *  Streams and graphs: a call is kernel launches on `stream` and nothing else -- no host synchronisation, no
 *  read-back (what one stage decides for the next travels in device-side lists inside the workspace), no
 *  memset or copy nodes -- so after one ordinary call (which sets the kernels' function attributes) the calls
 *  can be captured into a HIP graph (hipStreamBeginCapture on `stream`) and replayed on new contents of the
 *  same buffers (tests/test_gpu_graph.py).
 * ================================================================================================= */
/* HIST_count over a batch.  d_counts: nBlocks x 256 unsigned (entries 0..min(maxSV_in,255) written).
 * d_maxSymbolValues: nBlocks in/out values (NULL = 255 in, not reported). */
FSEHIP_API int FSEHIP_HIST_count_batch(unsigned* d_counts, unsigned* d_maxSymbolValues, size_t* d_results,
                                       const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                       size_t nBlocks, void* stream);

/* FSE_compress_usingCTable over a batch.  Table of block b: d_ctables + b*ctableStrideU32 (reference
 * layout; pass ctableStrideU32 = 0 to share one table).  maxTableLog bounds the tableLog found in the
 * tables (<= 12; smaller values raise occupancy); a table exceeding it yields ERROR(tableLog_tooLarge). */
FSEHIP_API int FSEHIP_FSE_compress_usingCTable_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                     const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                                     const FSEHIP_FSE_CTable* d_ctables, size_t ctableStrideU32, unsigned maxTableLog,
                                                     size_t nBlocks, void* stream);
/* FSE_decompress_usingDTable over a batch (d_cSizes: exact compressed size per block). */
FSEHIP_API int FSEHIP_FSE_decompress_usingDTable_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                       const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                       const FSEHIP_FSE_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                       size_t nBlocks, void* stream);

/* One-shot block API over a batch: FSE_compress2 (histogram, normalisation, NCount header, CTable,
 * payload) and FSE_decompress_wksp(maxLog) entirely on the device.  d_workspace/workspaceBytes: scratch,
 * at least FSEHIP_*_workspaceSize(1,...) bytes; larger workspaces process more blocks per pass (the
 * *_workspaceSize(nBlocks, ...) value never needs to be exceeded).  Every d_workspace of this header must be
 * 256-byte aligned (hipMalloc'ed memory is); a misaligned one is refused with hipErrorInvalidValue. */
FSEHIP_API size_t FSEHIP_FSE_compress_batch_workspaceSize(size_t nBlocks, unsigned tableLog);
FSEHIP_API int FSEHIP_FSE_compress_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                         const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                         unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks,
                                         void* d_workspace, size_t workspaceBytes, void* stream);
FSEHIP_API size_t FSEHIP_FSE_decompress_batch_workspaceSize(size_t nBlocks, unsigned maxLog);
FSEHIP_API int FSEHIP_FSE_decompress_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                           const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                           unsigned maxLog, size_t nBlocks,
                                           void* d_workspace, size_t workspaceBytes, void* stream);

/* Huff0: HUF_compress4X_usingCTable / HUF_decompress4X1_usingDTable over a batch, and the one-shot
 * HUF_compress2 / HUF_decompress (4X1 decoder) over a batch.  For the one-shot decoder d_dstSizes (or
 * uniformDstSize) is the exact regenerated size of each block, as HUF_decompress requires.
 * Memory contract of the batched calls (tests/test_gpu_edges.py, guard mode of tests/conftest.py): nothing is read behind
 * src + srcSize / cSrc + cSrcSize, nothing is written behind dst + dstCapacity (decoders: dst + dstSize); FSE tables are read
 * at their exact sizes (FSE_CTABLE_SIZE_U32(tableLog, maxSymbolValue) / FSE_DTABLE_SIZE_U32(tableLog) of the table's own header),
 * Huff0 DTables at 1 + (1 << tableLog) cells of their type.  A HUF_CElt table carries no header, so every table of a batch must
 * be readable at HUF_CTABLE_SIZE_U32(255) = 256 entries (what the reference's own callers declare, lib/huf_compress.c:628-632);
 * entries of symbols that do not occur are not interpreted.  The single-block calls on host pointers read only the entries of
 * symbols present in src, like the reference. */
FSEHIP_API int FSEHIP_HUF_compress4X_usingCTable_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                       const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                                       const FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32,
                                                       size_t nBlocks, void* stream);
/* HUF_compress1X_usingCTable over a batch (lib/huf.h:290; lib/huf_compress.c:457-502): one stream per block, no jump table */
FSEHIP_API int FSEHIP_HUF_compress1X_usingCTable_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                       const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                                       const FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32,
                                                       size_t nBlocks, void* stream);
FSEHIP_API int FSEHIP_HUF_decompress4X1_usingDTable_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                                          size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                          const FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                          size_t nBlocks, void* stream);
/* HUF_decompress4X_usingDTable over a batch (lib/huf.h:275, lib/huf_decompress.c:980-997): per block, tables of tableType 0
 * (single-symbol cells, from HUF_readDTableX1) take the fast decoder, tables of tableType 1 (double-symbol cells, from the
 * reference's HUF_readDTableX2: 4 bytes per cell, 1 + (1 << tableLog) words) an acceptance path with the reference's lock-step
 * semantics.  The 4X1 call above rejects tableType 1 with GENERIC exactly like HUF_decompress4X1_usingDTable. */
FSEHIP_API int FSEHIP_HUF_decompress4X_usingDTable_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                                         size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                         const FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                         size_t nBlocks, void* stream);
/* HUF_decompress1X1_usingDTable / HUF_decompress1X_usingDTable over a batch: one stream per block (the inverse of FSEHIP_HUF_compress1X_usingCTable_batch).
 * The stream is split across the 64 lanes of a wave like the streams of the 4X layout, in pieces when it exceeds the LDS budget. */
FSEHIP_API int FSEHIP_HUF_decompress1X1_usingDTable_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                                          size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                          const FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                          size_t nBlocks, void* stream);
FSEHIP_API int FSEHIP_HUF_decompress1X_usingDTable_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                                         size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                         const FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                         size_t nBlocks, void* stream);
FSEHIP_API size_t FSEHIP_HUF_compress_batch_workspaceSize(size_t nBlocks);
FSEHIP_API int FSEHIP_HUF_compress_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                         const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                         unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks,
                                         void* d_workspace, size_t workspaceBytes, void* stream);
FSEHIP_API size_t FSEHIP_HUF_decompress_batch_workspaceSize(size_t nBlocks);
FSEHIP_API int FSEHIP_HUF_decompress_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                           size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                           size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream);

/* ---- Tables for the *_usingCTable / *_usingDTable batch calls, built on the device from the blocks themselves: the steps the
 * reference runs between HIST_count and the hot loops (SURVEY 8(a') g1-g3, g5-g6), as calls of their own so that a caller of the
 * using-table forms never has to build tables on the host and upload them.
 *   FSE_buildCTable_batch  = HIST_count + the early-outs of FSE_compress_wksp + FSE_optimalTableLog + FSE_normalizeCount +
 *                            FSE_writeNCount + FSE_buildCTable_wksp per block (lib/fse_compress.c:646-665).  d_ctables + b*ctableStrideU32
 *                            receives the CTable in the reference's layout (ctableStrideU32 >= FSE_CTABLE_SIZE_U32(max(tableLog, 9), 255):
 *                            FSE_optimalTableLog may raise a small request), d_headers + b*headerStride the NCount header (at most
 *                            headerCapacity bytes), d_results[b] the header size (> 1: table and header valid) or what FSE_compress2 returns
 *                            when it stops before coding: 0 (not compressible), 1 (one repeated byte) or an error code.
 *   FSE_buildDTable_batch  = FSE_readNCount + the maxLog check + FSE_buildDTable per block (lib/fse_decompress.c:264-271): d_headers
 *                            points at the NCount headers (or at whole compressed blocks); d_dtables + b*dtableStrideU32 receives the DTable
 *                            in the reference's layout (dtableStrideU32 >= FSE_DTABLE_SIZE_U32(maxLog)); d_results[b] = bytes the header
 *                            takes (where the payload starts) or an error code.
 *   HUF_buildCTable_batch  = HIST_count + early-outs + HUF_buildCTable_wksp + HUF_writeCTable per block (lib/huf_compress.c:654-703): 256
 *                            HUF_CElt per table (ctableStrideU32 >= 256), the weights header in d_headers, d_results[b] = header size or
 *                            0 / 1 (the repeated byte goes to d_headers[b][0]) / error as HUF_compress2.
 *   HUF_readDTableX1_batch = HUF_readDTableX1 per block (lib/huf_decompress.c:118-185): single-symbol DTable with DTableDesc.maxTableLog =
 *                            maxTableLog (dtableStrideU32 >= HUF_DTABLE_SIZE(maxTableLog) = 1 + (1 << maxTableLog)); d_results[b] = header size or error. */
FSEHIP_API size_t FSEHIP_FSE_buildCTable_batch_workspaceSize(size_t nBlocks);
FSEHIP_API int FSEHIP_FSE_buildCTable_batch(FSEHIP_FSE_CTable* d_ctables, size_t ctableStrideU32, void* d_headers, size_t headerStride, size_t headerCapacity,
                                            size_t* d_results, const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                            unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream);
FSEHIP_API size_t FSEHIP_FSE_buildDTable_batch_workspaceSize(size_t nBlocks, unsigned maxLog);
FSEHIP_API int FSEHIP_FSE_buildDTable_batch(FSEHIP_FSE_DTable* d_dtables, size_t dtableStrideU32, size_t* d_results,
                                            const void* d_headers, size_t headerStride, const size_t* d_headerSizes, size_t uniformHeaderSize,
                                            unsigned maxLog, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream);
FSEHIP_API size_t FSEHIP_HUF_buildCTable_batch_workspaceSize(size_t nBlocks);
FSEHIP_API int FSEHIP_HUF_buildCTable_batch(FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32, void* d_headers, size_t headerStride, size_t headerCapacity,
                                            size_t* d_results, const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                            unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream);
/*   HUF_buildCTable_fromCount / HUF_writeCTable : the two halves of HUF_buildCTable_batch on the CALLER's statistics -- d_counts + b*256 holds
 *                            count[0 .. d_maxSymbolValues[b]] (countStride must be 256), d_ctables + b*ctableStrideU32 the 256 HUF_CElt of table b
 *                            (ctableStrideU32 >= 256, a multiple of 4); d_results[b] = the table log / the header size, or an error. */
FSEHIP_API int FSEHIP_HUF_buildCTable_fromCount_batch(FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32, const unsigned* d_counts, size_t countStride,
                                                      const unsigned* d_maxSymbolValues, unsigned maxNbBits, size_t nBlocks, size_t* d_results, void* stream);
FSEHIP_API int FSEHIP_HUF_writeCTable_batch(void* d_headers, size_t headerStride, size_t headerCapacity, const FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32,
                                            const unsigned* d_maxSymbolValues, unsigned huffLog, size_t nBlocks, size_t* d_results, void* stream);
FSEHIP_API size_t FSEHIP_HUF_readDTableX1_batch_workspaceSize(size_t nBlocks);
FSEHIP_API int FSEHIP_HUF_readDTableX1_batch(FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog, size_t* d_results,
                                             const void* d_src, size_t srcStride, const size_t* d_srcSizes, size_t uniformSrcSize,
                                             size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream);

/* ---- Table glue, step by step: the reference's public FSE_normalizeCount (lib/fse.h:137-144, lib/fse_compress.c:431-494), FSE_writeNCount
 * (lib/fse.h:150-156, lib/fse_compress.c:275-298) and FSE_readNCount (lib/fse.h:222-229, lib/entropy_common.c:41-144) as batch calls on counters /
 * headers the CALLER supplies -- the same device routines FSE_buildCTable_batch / FSE_buildDTable_batch run between the histogram and the tables,
 * reachable one step at a time (a caller with its own statistics; the reference's unit vectors, programs/fuzzer.c:325-417).
 *   normalizeCount : d_counts + b*countStride holds count[0 .. maxSymbolValues[b]] (countStride, normStride >= 256 elements), d_totals[b] their sum;
 *                    d_norms + b*normStride receives normalizedCounter[0 .. maxSymbolValues[b]]; d_results[b] = tableLog used (0 = default 11) or error.
 *   writeNCount    : d_headers + b*headerStride receives at most headerCapacity bytes; d_results[b] = header size, or dstSize_tooSmall / GENERIC as
 *                    the reference (no byte is written for a failed block).
 *   readNCount     : d_maxSymbolValues[b] in = the alphabet limit (< normStride), out = last symbol described; d_tableLogs[b] out;
 *                    d_results[b] = bytes read or error (maxSymbolValue_tooSmall, tableLog_tooLarge, corruption_detected). */
FSEHIP_API int FSEHIP_FSE_normalizeCount_batch(short* d_norms, size_t normStride, unsigned tableLog, const unsigned* d_counts, size_t countStride,
                                               const size_t* d_totals, const unsigned* d_maxSymbolValues, size_t nBlocks, size_t* d_results, void* stream);
FSEHIP_API int FSEHIP_FSE_writeNCount_batch(void* d_headers, size_t headerStride, size_t headerCapacity, const short* d_norms, size_t normStride,
                                            const unsigned* d_maxSymbolValues, unsigned tableLog, size_t nBlocks, size_t* d_results, void* stream);
FSEHIP_API int FSEHIP_FSE_readNCount_batch(short* d_norms, size_t normStride, unsigned* d_maxSymbolValues, unsigned* d_tableLogs,
                                           const void* d_headers, size_t headerStride, const size_t* d_headerSizes, size_t uniformHeaderSize,
                                           size_t nBlocks, size_t* d_results, void* stream);
/*   buildCTable_fromNorm / buildDTable_fromNorm : FSE_buildCTable (lib/fse.h:162, lib/fse_compress.c:66-177) and FSE_buildDTable (lib/fse.h:240,
 *                    lib/fse_decompress.c:71-126) on normalised counters the CALLER supplies -- d_norms + b*normStride holds
 *                    normalizedCounter[0 .. d_maxSymbolValues[b]] (normStride >= 256), one tableLog (2, 4 .. 12) per call; tables in the reference's
 *                    layouts (ctableStrideU32 >= FSE_CTABLE_SIZE_U32(tableLog, 255), dtableStrideU32 >= FSE_DTABLE_SIZE_U32(tableLog));
 *                    d_results[b] = 0 or an error (see the single calls above for what is refused).  The DTable form needs the workspace of
 *                    FSEHIP_FSE_buildDTable_fromNorm_batch_workspaceSize(nBlocks, tableLog). */
FSEHIP_API int FSEHIP_FSE_buildCTable_fromNorm_batch(FSEHIP_FSE_CTable* d_ctables, size_t ctableStrideU32, const short* d_norms, size_t normStride,
                                                     const unsigned* d_maxSymbolValues, unsigned tableLog, size_t nBlocks, size_t* d_results, void* stream);
FSEHIP_API size_t FSEHIP_FSE_buildDTable_fromNorm_batch_workspaceSize(size_t nBlocks, unsigned tableLog);
FSEHIP_API int FSEHIP_FSE_buildDTable_fromNorm_batch(FSEHIP_FSE_DTable* d_dtables, size_t dtableStrideU32, const short* d_norms, size_t normStride,
                                                     const unsigned* d_maxSymbolValues, unsigned tableLog, size_t nBlocks, size_t* d_results,
                                                     void* d_workspace, size_t workspaceBytes, void* stream);

/* ---- Packed (variable-length) batches.  The batched compressors write fixed-stride slots like the reference bench's buffers
 * (programs/bench.c:514-516); what the reference's container stores (programs/fileio.c:343-400) and what is worth moving between GPUs
 * or to the host is every block at its real size.  FSEHIP_compact_batch turns slots + results into records back to back:
 *   record b = the compressed bytes (d_results[b] > 1) | the block itself, srcSize bytes (d_results[b] == 0: programs/bench.c:393-396) |
 *              its first byte (d_results[b] == 1: :397-400) | nothing (d_results[b] an error code),
 * d_offsets[b] = where record b starts, d_offsets[nBlocks] = the packed size (nBlocks + 1 entries).  A device exclusive scan and one
 * coalesced copy; if the packed size exceeds packedCapacity the records that do not fit are not written and d_offsets[nBlocks] tells.
 * d_results must be what the call that filled d_slots returned: a value above slotStride (it cannot have come from that call) yields no record.
 * FSEHIP_compact_batch_bound(nBlocks, blockSize) = nBlocks * blockSize always suffices for blocks of at most blockSize bytes.
 * The decoders of a packed batch read the records where they lie and tell the three kinds apart by size, as HUF_decompress itself does
 * (lib/huf_decompress.c:1063-1066): a record as long as the block is the block, a record of one byte that byte repeated, anything else
 * goes through FSE_decompress / HUF_decompress.  d_origSizes / d_dstSizes: the regenerated size of every block. */
FSEHIP_API size_t FSEHIP_compact_batch_workspaceSize(size_t nBlocks);
FSEHIP_API size_t FSEHIP_compact_batch_bound(size_t nBlocks, size_t blockSize);
FSEHIP_API int FSEHIP_compact_batch(void* d_packed, size_t packedCapacity, uint64_t* d_offsets, const void* d_slots, size_t slotStride, const size_t* d_results,
                                    const void* d_src, size_t srcStride, const size_t* d_srcSizes, size_t uniformSrcSize, size_t nBlocks,
                                    void* d_workspace, size_t workspaceBytes, void* stream);
FSEHIP_API int FSEHIP_FSE_decompress_packed_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                  const void* d_packed, const uint64_t* d_offsets, const size_t* d_origSizes, size_t uniformOrigSize,
                                                  unsigned maxLog, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream);
FSEHIP_API int FSEHIP_HUF_decompress_packed_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize, size_t* d_results,
                                                  const void* d_packed, const uint64_t* d_offsets, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream);

/* Workload generator of the reference's benchmark (programs/probaGenerator.c:70-74,95-126), on the
 * device: block b = generate(blockSize bytes, table, seed = firstSeed + b).  h_table4096 is the
 * HOST 4096-entry symbol table built by FSEHIP_probagen_table(); it is consumed before the call returns (the
 * call is asynchronous on `stream` like the others). */
FSEHIP_API void FSEHIP_probagen_table(uint8_t table4096[4096], double p);
FSEHIP_API int FSEHIP_probagen_batch(void* d_dst, size_t dstStride, size_t blockSize, size_t nBlocks,
                                     const uint8_t h_table4096[4096], uint32_t firstSeed, void* stream);
/* same with seed = firstSeed + b * seedStep: lets a caller interleave several distributions in one corpus (BASELINE config 5:
 * block g of the corpus = distribution g mod 3, seed g + 1 -> three calls with dstStride = 3 blocks and seedStep = 3) */
FSEHIP_API int FSEHIP_probagen_batch_ex(void* d_dst, size_t dstStride, size_t blockSize, size_t nBlocks,
                                        const uint8_t h_table4096[4096], uint32_t firstSeed, uint32_t seedStep, void* stream);

/* ---- .fse frames on HOST buffers: the container written / read by the reference's command-line tool
 * (programs/fileio.c:266-285 format, FIO_compressFilename :286-432, FIO_decompressFilename :462-626) around blocks of
 * 1 KB << blockSizeId (id 0..6, the tool's default is 5 = 32 KB).  codec 0 = FSE_compress blocks (magic 0x183E2309),
 * 1 = HUF_compress blocks (magic 0x183E3309); blocks the codec declines are stored raw, single-byte blocks as RLE, and
 * the frame ends with 22 bits of XXH32 over the content, exactly as the tool writes them.  All blocks of a frame are
 * coded by one batched device call.  Returns the frame size / the regenerated size, or an error code (bad magic or
 * block-size id: GENERIC; truncated frame: srcSize_wrong; checksum mismatch: corruption_detected; a block decoder's
 * own code otherwise).  The zlibh mode of the tool is not supported. */
FSEHIP_API size_t FSEHIP_frame_compressBound(size_t srcSize, unsigned blockSizeId);
FSEHIP_API size_t FSEHIP_frame_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned blockSizeId, int codec);
FSEHIP_API size_t FSEHIP_frame_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);
/* Many frames per call (no counterpart in the tool, which walks its files one by one: programs/fileio.c:286-432 per file).  Frame i
 * is coded from srcs[i] / srcSizes[i] into dsts[i] / dstCapacities[i] exactly as the single-frame call would code it -- same bytes,
 * same result, in results[i] -- by a pool of nThreads host threads (0: half the host's hardware threads, at most 4; never more than
 * nFrames), each with a device stream and a scratch arena of its own: one frame is bound by one host thread's XXH32 and its copies,
 * many frames are not.  The helper threads are persistent: created at the first such call, they keep their streams and arenas between calls
 * (no hipMalloc / hipFree per call) and sit idle on a condition variable otherwise; FSEHIP_releaseScratch() makes the idle ones give their arenas
 * back.  One batch call uses them at a time -- a second caller arriving meanwhile runs on threads of its own.  All pointers are HOST pointers.  Returns 0, or an error code when the call itself cannot run (null arrays,
 * no device); a frame's own failure is in its results entry only. */
FSEHIP_API size_t FSEHIP_frame_compress_batch(void* const* dsts, const size_t* dstCapacities, const void* const* srcs, const size_t* srcSizes,
                                              size_t* results, size_t nFrames, unsigned blockSizeId, int codec, unsigned nThreads);
FSEHIP_API size_t FSEHIP_frame_decompress_batch(void* const* dsts, const size_t* dstCapacities, const void* const* srcs, const size_t* srcSizes,
                                                size_t* results, size_t nFrames, unsigned nThreads);

/* ---- FSE for 16-bit symbols (lib/fseU16.h:62-80, lib/fseU16.c) -- SURVEY 8(f) rank 4.  Alphabets of up to
 * FSEHIP_FSEU16_MAX_SYMBOL_VALUE + 1 symbols, table logs up to 13 (default 12), ONE tANS state per stream: a different format from
 * the byte coder's.  Sizes of the uncompressed side are in SYMBOLS (as in the reference), strides in bytes.
 *   FSE_countU16      lib/fseU16.c:121-146   count[0..*maxSymbolValuePtr], returns the largest count; a symbol above the limit:
 *                                            maxSymbolValue_tooSmall (limits above FSEHIP_FSEU16_MAX_SYMBOL_VALUE: maxSymbolValue_tooLarge)
 *   FSE_compressU16   lib/fseU16.c:203-256   returns the compressed size, 0 (not compressible), 1 (one symbol only), or an error
 *   FSE_decompressU16 lib/fseU16.c:306-329   returns the number of symbols regenerated or an error
 * Undefined behaviour of the reference is refused instead of reproduced: 8 bytes or less of room behind the header store no payload
 * (the reference writes in front of its buffer, fseU16.c:164 ignoring bitstream.h:191) and a stream without payload is
 * srcSize_wrong (the reference reads through a null pointer). */
#define FSEHIP_FSEU16_MAX_SYMBOL_VALUE 286
#define FSEHIP_FSEU16_MAX_TABLELOG 13
#define FSEHIP_FSEU16_DEFAULT_TABLELOG 12
FSEHIP_API size_t FSEHIP_FSE_countU16(unsigned* count, unsigned* maxSymbolValuePtr, const unsigned short* src, size_t srcSize);
FSEHIP_API size_t FSEHIP_FSE_compressU16(void* dst, size_t dstCapacity, const unsigned short* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog);
FSEHIP_API size_t FSEHIP_FSE_decompressU16(unsigned short* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize);
/* batched, device-resident: block b reads d_src + b * srcStrideBytes (d_srcSizes[b] or uniformSrcSize symbols) and writes
 * d_dst + b * dstStride; d_counts holds (FSEHIP_FSEU16_MAX_SYMBOL_VALUE + 1) entries per block */
FSEHIP_API size_t FSEHIP_FSE_compressU16_batch_workspaceSize(size_t nBlocks);
FSEHIP_API size_t FSEHIP_FSE_decompressU16_batch_workspaceSize(size_t nBlocks);
FSEHIP_API int FSEHIP_FSE_countU16_batch(unsigned* d_counts, unsigned* d_maxSymbolValues, size_t* d_results, const unsigned short* d_src, size_t srcStrideBytes,
                                         const size_t* d_srcSizes, size_t uniformSrcSize, unsigned maxSymbolValue, size_t nBlocks, void* stream);
FSEHIP_API int FSEHIP_FSE_compressU16_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results, const unsigned short* d_src, size_t srcStrideBytes,
                                            const size_t* d_srcSizes, size_t uniformSrcSize, unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks,
                                            void* d_workspace, size_t workspaceBytes, void* stream);
FSEHIP_API int FSEHIP_FSE_decompressU16_batch(unsigned short* d_dst, size_t dstStrideBytes, size_t dstCapacity, size_t* d_results, const void* d_cSrc, size_t cStride,
                                              const size_t* d_cSizes, size_t uniformCSize, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream);

/* ---- UNSTABLE: measurement aids, not part of the codec API.  Process-wide switches meant for ONE thread that owns the device while they are on
 * (bench.py's roofline figures); their numbering and meaning change with the kernels.  Declared only for a translation unit that defines
 * FSEHIP_INTERNAL before including this header; a product caller has no business with them.
 * Kernel timing probe for benchmarks: between probe_begin and probe_collect every kernel launch of the library is
 * bracketed by HIP events on its own stream.  probe_collect synchronises and returns, per kernel id
 * (0 hist, 1 fse_cprep, 2 fse_encode [lane per block], 3 fse_dprep, 4 fse_decode, 5 huf_cprep, 6 huf_encode, 7 huf_dprep,
 * 8 huf_decode, 9 fse_encode_wave [wave per block]), the summed duration in milliseconds and the number of launches.  Arrays hold 16 entries. */
#ifdef FSEHIP_INTERNAL
FSEHIP_API int FSEHIP_probe_begin(void);
FSEHIP_API int FSEHIP_probe_collect(double* totalMs16, unsigned* launches16);
/* Cycle accounting inside the FSE decoder (the bound that actually holds for it is chain latency x LDS-resident blocks, not HBM):
 * between (1, NULL) and (0, out16) the one-shot FSE decompressor runs an instrumented instantiation of its hot-loop kernel.
 * out16: [0] cycles of decoder-wave rounds that ran a phase of 16 iterations (4 symbols each), [1] cycles of rounds that waited,
 * [2] / [3] their numbers, [4] workgroups, [5] / [6] busy / idle cycles of one service wave per workgroup, [7] cycles inside the phases proper (without the
 * bookkeeping between them),
 * [8] engine clock in kHz, [9] blocks per workgroup | workgroups per CU << 32 | decoder waves per workgroup << 40 | iterations per phase << 48, [10] rounds of finishing phases (two iterations) and
 * [15] their cycles, [11] / [12] lifetime of the decoder waves in cycles / in ticks of the constant 100 MHz clock, [13] cycles before
 * the first phase (set-up), [14] cycles after the last (literal tail).  Synchronises the device; for benchmarks only. */
FSEHIP_API int FSEHIP_debug_decodeTiming(int enable, unsigned long long* out16);
#endif /* FSEHIP_INTERNAL */

/* Sharding a batch over the GPUs of a node (one process per GPU; the reference's chunk loop, programs/bench.c:353-364,389-424, has no
 * carried dependence, so contiguous ranges of blocks need no collective on the data path): rank `rank` of `world` codes blocks
 * [*first, *first + *count), ranges differ by at most one block.  The RCCL calls a C host puts around it when the corpus lives on
 * one rank -- one ncclGroup per direction -- are in INTEGRATION.md section 2c. */
FSEHIP_API void FSEHIP_shardRange(size_t nBlocks, int rank, int world, size_t* first, size_t* count);

/* The calls on HOST pointers (layer 1, the frames) take their device scratch from an arena the calling thread keeps between calls
 * (grow-only, at most 1 GiB; larger buffers are allocated and freed per call): no hipMalloc / hipFree on the repeated-call path.
 * FSEHIP_releaseScratch() gives the calling thread's arena back (and those of the batched frame calls' idle helper threads); a thread that
 * exits without calling it leaves its arena to the process teardown.  Returns 0 when everything was given back, the hipError_t of the first
 * hipFree that failed, hipErrorInvalidValue (1) when the calling thread is inside a call, or FSEHIP_SCRATCH_BUSY when a batched frame call
 * is running on the helper pool right now: those helpers keep their arenas (up to 64 x 1 GiB) -- call again when it has returned.
 *
 * Threads.  The batched frame calls keep a pool of helper threads inside the library (created at the first such call; each holds a HIP stream
 * and a scratch arena and sleeps on a condition variable between calls).  Consequences for the host process:
 *   - do not dlclose() the library while the pool exists: FSEHIP_shutdown() ends and joins the helpers (and frees their arenas and the calling
 *     thread's) first; it returns 0, or FSEHIP_SCRATCH_BUSY if a batch call is still running on them.  The pool restarts on demand.
 *   - do not fork() and go on using the library in the child: neither the helper threads nor the HIP runtime's own state exist there. */
#define FSEHIP_SCRATCH_BUSY (-2)
FSEHIP_API int FSEHIP_releaseScratch(void);
FSEHIP_API int FSEHIP_shutdown(void);

/* The one allocation the batched calls make themselves: FSEHIP_FSE_decompress_usingDTable_batch (whose reference signature, lib/fse.h:247, has
 * no workspace) keeps the symbol bytes of its tables in a per-device scratch of 2 x CUs slots of 72 KB that the library allocates at the first such
 * call on a device -- a hipMalloc and a synchronous hipMemset, once, kept until the process ends.  FSEHIP_prepareDevice() does that now, on the
 * current device (idempotent; returns 0 or a hipError_t): call it before a timed region or before capturing such a call into a graph -- a first
 * call that finds itself inside a stream capture without the scratch returns hipErrorStreamCaptureUnsupported instead of breaking the capture. */
FSEHIP_API int FSEHIP_prepareDevice(void);

/* build / device info: returns 0 and fills the fields when a gfx950 device is current */
typedef struct { int deviceOrdinal; int computeUnits; int ldsBytesPerCU; int wavefrontSize; char archName[64]; } FSEHIP_DeviceInfo;
FSEHIP_API int FSEHIP_deviceInfo(FSEHIP_DeviceInfo* info);
FSEHIP_API const char* FSEHIP_versionString(void);

#ifdef FSEHIP_DROPIN_NAMES
#define HIST_count FSEHIP_HIST_count
#define FSE_compress FSEHIP_FSE_compress
#define FSE_compress2 FSEHIP_FSE_compress2
#define FSE_decompress FSEHIP_FSE_decompress
#define FSE_compress_usingCTable FSEHIP_FSE_compress_usingCTable
#define FSE_decompress_usingDTable FSEHIP_FSE_decompress_usingDTable
#define HUF_compress FSEHIP_HUF_compress
#define HUF_compress2 FSEHIP_HUF_compress2
#define HUF_decompress FSEHIP_HUF_decompress
#define HUF_compress1X_usingCTable FSEHIP_HUF_compress1X_usingCTable
#define HUF_compress4X_usingCTable FSEHIP_HUF_compress4X_usingCTable
#define HUF_decompress4X_usingDTable FSEHIP_HUF_decompress4X_usingDTable
#define HUF_decompress4X1_usingDTable FSEHIP_HUF_decompress4X1_usingDTable
#define HUF_decompress1X_usingDTable FSEHIP_HUF_decompress1X_usingDTable
#define HUF_decompress1X1_usingDTable FSEHIP_HUF_decompress1X1_usingDTable
#define HIST_count_wksp FSEHIP_HIST_count_wksp
#define HIST_countFast FSEHIP_HIST_countFast
#define HIST_countFast_wksp FSEHIP_HIST_countFast_wksp
#define HIST_count_simple FSEHIP_HIST_count_simple
#define FSE_compress_wksp FSEHIP_FSE_compress_wksp
#define FSE_decompress_wksp FSEHIP_FSE_decompress_wksp
#define HUF_compress4X_wksp FSEHIP_HUF_compress4X_wksp
#define HUF_compress1X_wksp FSEHIP_HUF_compress1X_wksp
#define HUF_compress1X FSEHIP_HUF_compress1X
#define HUF_decompress4X1_DCtx_wksp FSEHIP_HUF_decompress4X1_DCtx_wksp
#endif
#ifdef FSEHIP_DROPIN_GLUE_NAMES      /* separate switch: the table glue and the header-reading single-symbol decoders (a program that wants the reference's own builders beside the device's hot loops leaves it off) */
#define FSE_optimalTableLog FSEHIP_FSE_optimalTableLog
#define FSE_normalizeCount FSEHIP_FSE_normalizeCount
#define FSE_NCountWriteBound FSEHIP_FSE_NCountWriteBound
#define FSE_writeNCount FSEHIP_FSE_writeNCount
#define FSE_readNCount FSEHIP_FSE_readNCount
#define FSE_buildCTable FSEHIP_FSE_buildCTable
#define FSE_buildCTable_wksp FSEHIP_FSE_buildCTable_wksp
#define FSE_buildDTable FSEHIP_FSE_buildDTable
#define HUF_buildCTable FSEHIP_HUF_buildCTable
#define HUF_buildCTable_wksp FSEHIP_HUF_buildCTable_wksp
#define HUF_writeCTable FSEHIP_HUF_writeCTable
#define HUF_readDTableX1 FSEHIP_HUF_readDTableX1
#define HUF_readDTableX1_wksp FSEHIP_HUF_readDTableX1_wksp
#define HUF_decompress4X1 FSEHIP_HUF_decompress4X1
#define HUF_decompress4X1_DCtx FSEHIP_HUF_decompress4X1_DCtx
#define HUF_decompress1X1 FSEHIP_HUF_decompress1X1
#define HUF_decompress1X1_DCtx FSEHIP_HUF_decompress1X1_DCtx
#define HUF_decompress1X1_DCtx_wksp FSEHIP_HUF_decompress1X1_DCtx_wksp
#endif
#ifdef FSEHIP_DROPIN_U16_NAMES       /* separate switch: programs/fuzzer.c declares FSE_countU16 with another (stale) prototype */
#define FSE_countU16 FSEHIP_FSE_countU16
#define FSE_compressU16 FSEHIP_FSE_compressU16
#define FSE_decompressU16 FSEHIP_FSE_decompressU16
#endif

#ifdef __cplusplus
}
#endif
#endif /* FSEHIP_H */
