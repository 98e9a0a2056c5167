/* oracle/fse_on_mi355x.h -- TEST INFRASTRUCTURE: the 3-line shim of INTEGRATION.md section 1.
 * Force-included (gcc -include) when compiling the reference's own programs/fuzzer.c and programs/fuzzerHuff0.c, AFTER
 * which every hot-path call written in those programs (HIST_count, FSE_compress[2], FSE_decompress,
 * FSE_compress_usingCTable, FSE_decompress_usingDTable, HUF_compress[2], HUF_decompress, HUF_compress1X/4X_usingCTable,
 * HUF_decompress4X[1]_usingDTable; in programs/fuzzerU16.c FSE_countU16, FSE_compressU16, FSE_decompressU16) resolves to libfsehip.so -- and, in programs/fuzzer.c
 * and programs/fullbench.c, the table glue they call as well (FSE_optimalTableLog, FSE_normalizeCount, FSE_NCountWriteBound, FSE_writeNCount, FSE_readNCount,
 * FSE_buildCTable, FSE_buildDTable: the unit vectors of fuzzer.c:325-417 then run on the device's wave routines).  The reference's lib/ objects are compiled
 * WITHOUT it and still provide what has no device counterpart (FSE_buildCTable_raw, FSE_buildDTable_raw, the Huff0 table calls of fullbench.c, ...). */
#ifndef FSE_ON_MI355X_H
#define FSE_ON_MI355X_H
#define FSE_STATIC_LINKING_ONLY
#define HUF_STATIC_LINKING_ONLY
#include "fse.h"
#include "huf.h"
#include "hist.h"
#ifdef FSE_ON_MI355X_fuzzerU16
#include "fseU16.h"
#define FSEHIP_DROPIN_U16_NAMES
#endif
#define FSEHIP_DROPIN_NAMES
#if defined(FSE_ON_MI355X_fuzzer) || defined(FSE_ON_MI355X_fullbench)
#define FSEHIP_DROPIN_GLUE_NAMES     /* FSE_optimalTableLog, FSE_normalizeCount, FSE_NCountWriteBound, FSE_writeNCount, FSE_readNCount, FSE_buildCTable, FSE_buildDTable */
#endif
#include "fsehip.h"
#endif
