// internal.h -- kernel argument blocks and host-side launchers (one per kernel family).
#pragma once
#include "dev_common.h"

// ---- a1: HIST_count ------------------------------------------------------------------------------
struct HistArgs {
    unsigned* counts;            // nBlocks x 256
    unsigned* maxSVs;            // in/out per block, or nullptr (255 in)
    unsigned uniformMaxSV;       // used when maxSVs == nullptr or useUniformIn
    int useUniformIn;            // 1: limit = uniformMaxSV for every block, maxSVs (if any) is output only
    int trustInput = 0;          // 1: HIST_countFast (lib/hist.c:141-159): a limit below 255 only bounds the entries written, symbols above it are not an error
    size_t* results;
    BlockView src;
    size_t nBlocks;
};
hipError_t launch_hist(const HistArgs& a, hipStream_t s);
// one input of n >= HIST_PIECE bytes cut into HIST_PIECE-byte pieces: d_part holds ceil(n / HIST_PIECE) x 256 words, d_scratchResults as many size_t
#define HIST_PIECE ((size_t)65536)
#define HIST_LARGE_MIN ((size_t)262144)
hipError_t launch_hist_large(const u8* d_src, size_t n, unsigned limitIn, int trustInput, unsigned* d_part, unsigned* d_count, unsigned* d_maxSV,
                             size_t* d_result, size_t* d_scratchResults, hipStream_t s);
// n 32-bit words at p set to zero by a kernel of ours (p 4-byte aligned).  Used instead of hipMemsetAsync for the few counter words
// the pipelines clear per call: captured into a HIP graph, the runtime's small-memset node faulted on later replays (ROCm 7.2, after
// other work had been synchronised on the stream in between) -- a plain kernel node does not.
hipError_t launch_zero_u32(u32* p, u32 n, hipStream_t s);

// ---- FSE ------------------------------------------------------------------------------------------
// per-block record shared by the prepare and the hot-loop kernels
struct FseMeta {
    u32 state;      // 0 = final result already in results[b]; 1 = run the hot loop
    u32 hdrSize;    // bytes of NCount header in front of the payload
    u32 tableLog;
    u32 maxSV;
    u32 pace;       // compress side: how slowly the encoder states mix with this table (bin 0 .. FSE_EBINS-1 of tableSize / symbols in use)
};

struct FseCPrepArgs {            // glue g1-g4 (compress side): lib/fse_compress.c:632-677 minus the hot loops
    const unsigned* counts;      // from k_hist, nBlocks x 256
    const unsigned* maxSVs;      // from k_hist
    const size_t* histResults;
    BlockView src;
    u8* dst; size_t dstStride; size_t dstCapacity;
    unsigned maxSVReq, tableLogReq;
    u32* ctables; size_t ctStrideU32;
    unsigned maxTl;              // largest tableLog FSE_optimalTableLog can pick for this request (sizes the tables)
    FseMeta* meta;
    size_t* results;
    size_t nBlocks;
};
hipError_t launch_fse_cprep(const FseCPrepArgs& a, hipStream_t s);
// the glue steps as calls of their own (fse_tables.hip): FSE_normalizeCount / FSE_writeNCount / FSE_readNCount on caller-supplied counters and headers
hipError_t launch_fse_glue_normalize(s16* norms, size_t normStride, u32 tl, const u32* counts, size_t countStride, const size_t* totals, const u32* maxSVs,
                                     size_t* results, size_t nBlocks, hipStream_t s);
hipError_t launch_fse_glue_write_ncount(u8* headers, size_t headerStride, size_t headerCapacity, const s16* norms, size_t normStride, const u32* maxSVs, u32 tl,
                                        size_t* results, size_t nBlocks, hipStream_t s);
hipError_t launch_fse_glue_read_ncount(s16* norms, size_t normStride, u32* maxSVs, u32* tableLogs, const BlockView& headers, size_t* results, size_t nBlocks, hipStream_t s);
// FSE_buildCTable on the caller's normalised counters (one table log per call, lib/fse.h:162), FSE_buildDTable likewise (declared below FseDPrepArgs)
hipError_t launch_fse_ctable_from_norm(const s16* norms, size_t normStride, const u32* maxSVs, u32 tl, u32* ctables, size_t ctStrideU32, size_t* results,
                                       size_t nBlocks, hipStream_t s);

struct FseEncArgs {              // a2: FSE_compress_usingCTable, one lane per block
    u8* dst; size_t dstStride; size_t dstCapacity;
    size_t* results;
    BlockView src;
    const u32* ctables; size_t ctStrideU32;
    const FseMeta* meta;         // nullptr for the plain usingCTable batch
    unsigned maxTableLog;
    int G;                       // blocks per workgroup
    unsigned slotU32;            // LDS words per table slot
    size_t nBlocks;
    unsigned onlyState;          // with meta: 0 = every prepared block, else only blocks whose meta.state equals it
    unsigned sizeSplit;          // without meta, ragged batch: 1 = the wave kernel takes the blocks of FSE_ENC_WAVE_MIN bytes and more, the lane kernel the others
    u32* list; u32* count;       // wave kernel, one-shot path: FSE_EBINS lists of the FSE_ENC_PAR blocks by pace (nBlocks entries apart) and their
};                               // lengths (device memory, filled by launch_fse_encode_auto), or nullptr: blocks in order
// A wave of the encoder carries two blocks and lasts as long as the slower one; how long a block takes follows its table (few symbols:
// the speculated states merge slowly, more repair rounds).  The blocks are therefore walked in order of their pace bin.
enum { FSE_EBINS = 4 };
// meta.state on the compress side: 0 = finished by the prepare kernel, FSE_ENC_PAR / FSE_ENC_LANE = table ready, preferred kernel
enum { FSE_ENC_PAR = 1, FSE_ENC_LANE = 2 };
#define FSE_ENC_WAVE_MIN 2048u   // bytes: below this a block's 32 lanes of the wave kernel do not pay; k_fse_cprep marks such blocks FSE_ENC_LANE
__host__ __device__ inline bool fse_enc_skip(unsigned state, unsigned onlyState) { return state == 0 || (onlyState != 0 && state != onlyState); }
hipError_t launch_fse_encode(FseEncArgs a, hipStream_t s);
size_t fse_encode_blocks_per_round(unsigned maxTableLog);
hipError_t launch_fse_encode_wave(FseEncArgs a, hipStream_t s);
hipError_t launch_fse_enc_lists(const FseEncArgs& a, hipStream_t s);     // fills a.list / a.count from meta (state == FSE_ENC_PAR, pace)  // block-parallel variant (one wave per block, streaming I/O)
// The wave-per-block kernel takes the blocks of FSE_ENC_WAVE_MIN bytes and more, the lane-per-block kernel the others (both exact).  A batch
// of uniformly sized blocks needs one launch; otherwise the choice is per block and both kernels are launched, each skipping the other's
// blocks: with prepare-kernel metadata by meta.state (k_fse_cprep marks short blocks and extremely skewed tables FSE_ENC_LANE, the wave
// kernel hands over blocks it cannot write word-wise), without it (caller tables, ragged batch) by the block's size.  (Until round 5 a
// ragged batch went to the lane-per-block kernel whole: 75 instead of 600 GB/s for blocks of 20-64 KB.)
inline hipError_t launch_fse_encode_auto(FseEncArgs a, hipStream_t s)
{
    a.onlyState = 0; a.sizeSplit = 0;
    const bool ragged = a.src.sizes || a.src.offsets;
    if (a.dstCapacity > 0x7FFFFFF0u || (!ragged && a.src.uniform < FSE_ENC_WAVE_MIN)) return launch_fse_encode(a, s);
    if (!a.meta) {
        a.list = nullptr; a.count = nullptr;
        if (!ragged) return launch_fse_encode_wave(a, s);
        a.sizeSplit = 1;
        const hipError_t e = launch_fse_encode_wave(a, s);
        return e != hipSuccess ? e : launch_fse_encode(a, s);
    }
    a.onlyState = FSE_ENC_PAR;
    hipError_t e = a.list ? launch_fse_enc_lists(a, s) : hipSuccess;
    if (e == hipSuccess) e = launch_fse_encode_wave(a, s);
    if (e != hipSuccess) return e;
    a.onlyState = FSE_ENC_LANE;
    return launch_fse_encode(a, s);
}

// Decoder classes, chosen per block by k_fse_dparse from the block's own tableLog (the caller's maxLog only bounds it):
//   FSE_DCLS_REV11 : tableLog <= 11            -> bit-reversed cells, 4 KiB of LDS per table   (the fast loop, 16 blocks / workgroup)
//   FSE_DCLS_REV12 : tableLog 12, nbBits >= 1  -> bit-reversed cells, 8 KiB of LDS per table   (the fast loop,  9 blocks / workgroup)
//   FSE_DCLS_PLAIN : tableLog 12 with a symbol owning more than half the table (some nbBits == 0: rev(newState) then needs 12 bits)
//                    -> cells newState | nbBits << 12, the register-window loop
// Each class has its own list of block indices (dense workgroups) and its own launch; see fse_decode.hip.
// Inside a class the blocks are binned by compressed size (FSE_DBINS bins of FSE_DBIN_BYTES): a workgroup lasts as long as its slowest
// block, and how fast a block decodes follows its input rate -- a mixed batch (BASELINE config 5: P02 / P14 / P80 interleaved) otherwise
// runs every workgroup at the pace of its P02 blocks.  List index = class * FSE_DBINS + bin.
// (round 6: 16 bins of 2 KiB instead of 4 of 8 KiB -- a ragged batch of 12..32 KB blocks had blocks of 8 and 16 KB of payload in one workgroup,
//  which lasts as long as the longer; per uncompressed byte it decoded at 0.83 of the uniform batch's rate)
#ifndef FSE_DBINS_N
#define FSE_DBINS_N 16
#define FSE_DBIN_LOG 11          // 2 KiB of compressed bytes per bin (the last bin is open-ended)
#endif
enum { FSE_DCLS_REV11 = 0, FSE_DCLS_REV12 = 1, FSE_DCLS_PLAIN = 2, FSE_DCLS_KINDS = 3, FSE_DBINS = FSE_DBINS_N, FSE_DCLS_COUNT = FSE_DCLS_KINDS * FSE_DBINS };
struct FseDPrepArgs {            // glue g2,g3,g4 (decompress side): FSE_readNCount + FSE_buildDTable
    BlockView csrc;
    unsigned maxLog;
    // outputs, `capTs` = 1 << maxLog cells per block: the decoder's own table format (fse_decode.hip)
    u16* atab;                   // bit-reversed classes: cell rev(x) = nbBits | rev(newState) << 5; plain class: cell x = newState | nbBits << 12
    u8* symtab;                  // symbol of the cell at the same index
    s16* norms;                  // scratch between the two prepare kernels: 256 counters per block
    FseMeta* meta;               // state: 0 = result final, else 1 | fastMode << 1 | class << 2
    u32* lists;                  // FSE_DCLS_COUNT lists (class x size bin) of block indices, `nBlocks` entries apart
    u32* counts;                 // their lengths, then FSE_DCLS_KINDS class totals (zeroed by the launcher)
    size_t* results;
    size_t nBlocks;
    // packed batches with the bench loop's semantics (programs/bench.c:393-406): a record as long as the block (origSizes / uniformOrig)
    // or of one byte is not a compressed block -- k_rawrle_expand has regenerated it; the parser leaves it alone
    int rawRle; const size_t* origSizes; size_t uniformOrig;
};
hipError_t launch_fse_dprep(const FseDPrepArgs& a, hipStream_t s);
hipError_t launch_fse_dprep_from_norm(const FseDPrepArgs& a, const s16* norms, size_t normStride, const u32* maxSVs, u32 tl, hipStream_t s);

struct FseDecArgs {              // a3: FSE_decompress_usingDTable, one lane per block
    u8* dst; size_t dstStride; size_t dstCapacity;
    size_t* results;
    BlockView csrc;
    const u32* dtables; size_t dtStrideU32;   // reference-layout tables (usingDTable batch), or nullptr:
    const u16* atab; const u8* symtab;        // tables in the decoder's own format from k_fse_dbuild, 1 << maxTableLog cells per block
    const FseMeta* meta;         // nullptr for the plain usingDTable batch
    unsigned maxTableLog;        // global table slots hold 1 << maxTableLog cells
    unsigned ldsLog;             // LDS table slots hold 1 << ldsLog cells (set by the launcher from the class)
    const u32* list;             // one-shot path: the class's FSE_DBINS lists of block indices (nBlocks entries apart) and their
    const u32* count;            // FSE_DBINS lengths, in device memory (slotBlock, fse_decode.hip); nullptr: all blocks in order
    int G;
    unsigned slotU32;
    size_t nBlocks;
    // caller-built tables (usingDTable batch: no workspace, hence no class lists): every launch walks all blocks and takes those of its
    // class, judged from the table's own header -- tableLog in [tlMin, ldsLog]; `declineNb0`: a table with a cell of nbBits 0 is left to
    // the plain-cell launch (marked FSE_DECLINED in results[]); `onlyDeclined`: take exactly the marked blocks
    unsigned tlMin; int declineNb0; int onlyDeclined;
    // ... and their symbols go through a scratch the library keeps per device (set by the launcher, fse_decode.hip)
    u8* symScratch; u32* slotBitmap; u32 nSlots; u32 scratchSlotBytes;
};
// marker in results[] between the launches of the caller-table batch (no size_t a decoder returns, cf. HUF_DECLINED)
#define FSE_DECLINED ((size_t)0 - (size_t)0x7001)
#define FSE_DEC_FAST_MAXLOG 11u   // largest tableLog of the 4 KiB class
hipError_t launch_fse_decode(FseDecArgs a, hipStream_t s);
// one-shot path: one launch per decoder class over the lists written by k_fse_dparse
hipError_t launch_fse_decode_classes(FseDecArgs a, const u32* lists, const u32* counts, hipStream_t s);
size_t fse_decode_blocks_per_round(unsigned maxTableLog);   // blocks that fill the device once (for chunk sizing)

// ---- Huff0 ----------------------------------------------------------------------------------------
struct HufMeta {
    u32 state;      // 0 = final, 1 = run the hot loop
    u32 hdrSize;
    u32 tableLog;
    u32 maxSV;
};

struct HufCPrepArgs {            // glue g5-g7: lib/huf_compress.c:637-724 minus the hot loop
    const unsigned* counts; const unsigned* maxSVs; const size_t* histResults;
    BlockView src;
    u8* dst; size_t dstStride; size_t dstCapacity;
    unsigned maxSVReq, huffLogReq;
    u32* ctables; size_t ctStrideU32;    // 256 HUF_CElt per block
    HufMeta* meta;
    size_t* results;
    size_t nBlocks;
};
hipError_t launch_huf_cprep(const HufCPrepArgs& a, hipStream_t s, void* nodeScratch /* 4 KiB per block */);
hipError_t launch_huf_cprep_glue(const HufCPrepArgs& a, int mode /* 1: HUF_buildCTable on a.counts, 2: HUF_writeCTable on a.ctables */, hipStream_t s);

struct HufEncArgs {              // a4: HUF_compress4X_usingCTable (streams = 4) / 1X (streams = 1), one workgroup per block
    u8* dst; size_t dstStride; size_t dstCapacity;
    size_t* results;
    BlockView src;
    const u32* ctables; size_t ctStrideU32;
    const HufMeta* meta;
    int streams;
    int split1X;                 // streams == 1: four waves per block, a quarter of the symbols each (the batched 1X call)
    size_t nBlocks;
};
hipError_t launch_huf_encode(const HufEncArgs& a, hipStream_t s);

// decoder classes: class = 2 * kind + (tableLog == 12).  tableLog up to 11: 4 KiB LDS table slots, 12: 8 KiB.  kind: the
// stream-parallel decoder (huf_decode_par.hip) with one of its three LDS budgets per staged stream, or the serial decoder (tiny and
// irregular blocks, chosen by k_huf_dprep from the jump table; the parallel decoder appends what it declines to the serial lists)
enum { HUF_DKIND_PAR_TINY = 0, HUF_DKIND_PAR_SMALL = 1, HUF_DKIND_PAR_LARGE = 2, HUF_DKIND_SERIAL = 3, HUF_DCLS_COUNT = 8 };
#ifndef HPAR_ALL_SMALL
#define HPAR_ALL_SMALL 1                // round 6: streams beyond the 4.5 KiB budget are taken in equal PIECES by the same class (18 waves per CU) instead of whole by the
#endif                                  // 8.4 KiB class (12 waves): P02 (7 KB streams) 3.90 -> 3.73 ms per 100k blocks; the large budget stays for double-symbol caller tables
#ifndef HPAR_USE_TINY
#define HPAR_USE_TINY 0                 // round 6: the kept symbols (48 registers) bound the residency at 16 waves per CU; the smallest budget's 25 are out of reach and its
#endif                                  // 2 KiB line buffer would take four output passes per stream -- its blocks go with the 4.5 KiB budget (18 by LDS)
#define HPAR_DATA_TINY  2304u           // LDS budgets for one staged stream (+ 96 bytes of zero padding behind its end)
#ifndef HPAR_DATA_SMALL
#define HPAR_DATA_SMALL 4608u
#endif
#define HPAR_DATA_LARGE (8192u + 384u)  // (also the line buffer of the output pass: 8 KiB + two slacks of 192 bytes)
#define HPAR_MIN_BITS 4096u             // streams shorter than this go to the serial decoder (ranges must dwarf warm-up and codes)
struct HufDPrepArgs {            // glue g6: HUF_readStats + HUF_readDTableX1 (+ raw / RLE decisions of HUF_decompress)
    BlockView csrc;
    BlockView dstSizes;          // only sizes/uniform used
    u8* dst; size_t dstStride;
    u32* dtables; size_t dtStrideU32;
    HufMeta* meta;
    u32* lists;                  // HUF_DCLS_COUNT lists of block indices, `nBlocks` entries apart, and their lengths (zeroed by the launcher)
    u32* counts;
    size_t* results;
    size_t nBlocks;
    int tableOnly;               // HUF_readDTableX1 over a batch (lib/huf_decompress.c:118-185): no raw / RLE decisions, nothing of the payload
                                 // looked at; results[b] = header size (HUF_readStats' return value) or its error
    unsigned dtMaxLog;           // DTableDesc.maxTableLog of the tables (the one-shot path: HUF_TABLELOG_MAX - 1, lib/huf_decompress.c:1030)
};
hipError_t launch_huf_dprep(const HufDPrepArgs& a, hipStream_t s);
// results[b] = meta[b].hdrSize for every block a prepare kernel left pending (state != 0): the table-building batch calls
hipError_t launch_hdr_results(const void* meta, size_t metaStride, size_t* results, size_t nBlocks, hipStream_t s);
// FSE_buildDTable over a batch: the decoder-format tables of k_fse_dbuild written out in the reference's layout (lib/fse.h:565-575)
hipError_t launch_fse_export_dtables(const FseDPrepArgs& a, u32* dtables, size_t dtStrideU32, hipStream_t s);

struct HufDecArgs {              // a5: HUF_decompress4X1_usingDTable, 4 lanes per block (one per stream)
    u8* dst; size_t dstStride;
    BlockView dstSizes;
    size_t* results;
    BlockView csrc;
    const u32* dtables; size_t dtStrideU32;
    const HufMeta* meta;
    unsigned maxTableLog;
    unsigned ldsLog;             // LDS table slots hold 1 << ldsLog cells (set by the launcher)
    const u32* list;             // block indices of this launch and their number (device memory), or nullptr: all blocks
    const u32* count;
    int G; unsigned slotU32;
    int streams;                 // 4 (4X1) or 1 (1X1)
    int acceptX2;                // usingDTable batch only: blocks with a double-symbol table (tableType 1) are decoded instead of failing
    int onlyDeclined;            // usingDTable batch only: decode just the blocks whose result is HUF_DECLINED (left by the stream-parallel decoder)
    unsigned classLo;            // usingDTable batch only (no workspace, hence no class lists): a launch of the stream-parallel decoder takes the blocks whose
                                 // longest stream needs more than classLo bytes of LDS (and fits its own budget, or it is the largest class)
    size_t nBlocks;
};
// The caller-table batch has no workspace for a list of declined blocks: the stream-parallel decoder marks a block it declines with
// this value in results[b] (no size_t a decoder returns: sizes are below 2^28 there, error codes above (size_t)-9), and the
// serial / literal kernels that run afterwards take exactly the marked blocks.
#define HUF_DECLINED ((size_t)0 - (size_t)0x7000)
hipError_t launch_huf_decode(HufDecArgs a, hipStream_t s);
// one-shot path: one launch per class list, the stream-parallel decoder's first; what it declines -- corrupt or irregular blocks --
// it appends to the serial list of the same tableLog, decoded by the serial kernel afterwards
hipError_t launch_huf_decode_classes(HufDecArgs a, u32* lists, u32* counts, hipStream_t s);
hipError_t launch_huf_decode_par(HufDecArgs a, unsigned dataBytes, u32* serialList, u32* serialCount, hipStream_t s);
hipError_t launch_huf_decode_par_x2(HufDecArgs a, hipStream_t s);     // caller-built double-symbol tables marked HUF_DECLINED by the launch above

// ---- FSE for 16-bit symbols (fse_u16.hip) --------------------------------------------------------------------
struct U16Meta { u32 state, hdrSize, tableLog, maxSV; };      // state 0: result final; 1: run the chain
struct U16CArgs {
    const u16* src; size_t srcStrideBytes; const size_t* srcSizes; size_t uniformSrcSize;     // sizes in symbols
    u8* dst; size_t dstStride; size_t dstCapacity;
    u32 maxSVReq, tableLogReq;
    u16* stateTables; u32* symTT; U16Meta* meta;       // per block: 1 << 13 u16, 2 * 287 u32
    unsigned* countsOut; unsigned* maxSVOut;           // FSE_countU16 only (then nothing else is done)
    size_t* results; size_t nBlocks;
};
struct U16DArgs {
    u16* dst; size_t dstStrideBytes; size_t dstCapacity;                     // capacity in symbols
    const u8* csrc; size_t cStride; const size_t* cSizes; size_t uniformCSize;
    u32* cells; U16Meta* meta;                                               // per block a 32 KiB slot: first the normalised counters k_u16_dparse read from the header
                                                                             // (640 bytes), then, over them, the table k_u16_dprep builds -- 16-bit chain cells + the packed
                                                                             // 9-bit symbols 16 KiB further on (table logs up to 12), or 32-bit cells (table log 13)
    size_t* results; size_t nBlocks;
};
hipError_t launch_u16_compress(const U16CArgs& a, hipStream_t s);
hipError_t launch_u16_decompress(const U16DArgs& a, hipStream_t s);
hipError_t launch_u16_decode_lds(const U16DArgs& a, hipStream_t s);   // fse_u16_decode.hip: blocks k_u16_dprep marked state 1 (table log <= 12), one launch per slot class (<= 11 / 12)

// ---- workload generator -----------------------------------------------------------------------------
hipError_t launch_probagen(u8* dst, size_t dstStride, size_t blockSize, size_t nBlocks, const u8* d_table, u32 firstSeed, u32 seedStep, hipStream_t s);

// ---- kernel timing probe (HIP events on the launch stream; used by bench.py for the live roofline figure) ----
enum { PK_HIST = 0, PK_FSE_CPREP, PK_FSE_ENCODE, PK_FSE_DPREP, PK_FSE_DECODE, PK_HUF_CPREP, PK_HUF_ENCODE, PK_HUF_DPREP, PK_HUF_DECODE, PK_FSE_ENCODE_WAVE, PK_COUNT };
void probe_before(int kernelId, hipStream_t s);
void probe_after(int kernelId, hipStream_t s);

// Scratch for the calls on HOST pointers (single-block calls, .fse frames): a per-thread, per-device arena of device memory that only
// grows -- a call carves its buffers from it stack-wise instead of paying hipMalloc / hipFree (tens of microseconds each, and a
// device-wide synchronisation in hipFree) four or five times.  A buffer that does not fit is allocated the old way and the arena grows
// to the call's peak before the next call (up to FSEHIP_SCRATCH_MAX); the batched calls on device pointers never allocate at all.
struct HostCallBuf {
    void* p = nullptr;
    hipError_t alloc(size_t n);
    ~HostCallBuf();
    HostCallBuf() = default;
    HostCallBuf(const HostCallBuf&) = delete; HostCallBuf& operator=(const HostCallBuf&) = delete;
private:
    size_t carved = 0; bool owned = false;
};
#define FSEHIP_SCRATCH_MAX ((size_t)1 << 30)

// per-device caches (capi.hip): properties of the current device; "this kernel may use `bytes` of dynamic LDS" is set once
// per (device, kernel)
struct DevProps { int cus; int ldsPerCU; bool ok; };
const DevProps& dev_props();
hipError_t ensure_dyn_lds(const void* kernel, int bytes);
int release_thread_scratch(void);          // capi.hip: gives the calling thread's host-call arena back
int frame_pool_release_scratch(void);       // frame.hip: the same for the idle helper threads of the batched frame calls (0, a hipError_t, or FSEHIP_SCRATCH_BUSY)
int frame_pool_shutdown(void);              // frame.hip: ends and joins those threads
