// capi.hip -- the C ABI of libfsehip.so (include/fsehip.h): host-side mirror of the reference's block API
// (lib/fse.h, lib/huf.h, lib/hist.h) on top of the HIP kernels.  No CPU compute path exists here: every
// result is produced by a kernel; without a usable device the calls fail.
#include "internal.h"
#include <string.h>
#include <stdlib.h>
#include <new>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return (int)e__; } while (0)
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" unsigned FSEHIP_isError(size_t code) { return code > FSEHIP_ERROR(maxCode); }

extern "C" const char* FSEHIP_getErrorName(size_t code)   // lib/error_private.h:88-104
{
    if (!FSEHIP_isError(code)) return "No error detected";
    switch ((int)(0 - code)) {
        case FSEHIP_error_GENERIC: return "Error (generic)";
        case FSEHIP_error_dstSize_tooSmall: return "Destination buffer is too small";
        case FSEHIP_error_srcSize_wrong: return "Src size is incorrect";
        case FSEHIP_error_corruption_detected: return "Corrupted block detected";
        case FSEHIP_error_tableLog_tooLarge: return "tableLog requires too much memory : unsupported";
        case FSEHIP_error_maxSymbolValue_tooLarge: return "Unsupported max Symbol Value : too large";
        case FSEHIP_error_maxSymbolValue_tooSmall: return "Specified maxSymbolValue is too small";
        case FSEHIP_error_workSpace_tooSmall: return "workspace buffer is too small";
        default: return "Unspecified error code";
    }
}

extern "C" const char* FSEHIP_versionString(void) { return "fsehip 0.3 (gfx950)"; }

extern "C" void FSEHIP_shardRange(size_t nBlocks, int rank, int world, size_t* first, size_t* count)   // = shard.shard_range
{
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    const size_t base = nBlocks / (size_t)world, rem = nBlocks % (size_t)world, r = (size_t)rank;
    const size_t lo = r * base + (r < rem ? r : rem);
    if (first) *first = lo;
    if (count) *count = r < (size_t)world ? base + (r < rem ? 1 : 0) : 0;
}

// Per-device caches (a process may drive several devices: the attribute and the CU count belong to the current one),
// guarded for concurrent host threads.
#include <mutex>
#include <vector>
#include <utility>
namespace {
std::mutex g_devMutex;
const int MAX_DEVS = 64;
DevProps g_props[MAX_DEVS];
std::vector<std::pair<int, const void*>> g_ldsAttrDone;      // (device, kernel) pairs already configured
}
const DevProps& dev_props()
{
    static const DevProps none = { 0, 0, false };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVS) return none;
    std::lock_guard<std::mutex> lock(g_devMutex);
    DevProps& p = g_props[dev];
    if (!p.ok) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            p.cus = prop.multiProcessorCount;
            p.ldsPerCU = (int)prop.maxSharedMemoryPerMultiProcessor;
            p.ok = true;
        }
    }
    return p;
}
hipError_t ensure_dyn_lds(const void* kernel, int bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_devMutex);
    for (auto& d : g_ldsAttrDone) if (d.first == dev && d.second == kernel) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) g_ldsAttrDone.emplace_back(dev, kernel);
    return e;
}

extern "C" int FSEHIP_deviceInfo(FSEHIP_DeviceInfo* info)
{
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDevice(&dev));
    CK(hipGetDeviceProperties(&prop, dev));
    info->deviceOrdinal = dev;
    info->computeUnits = prop.multiProcessorCount;
    info->ldsBytesPerCU = (int)prop.maxSharedMemoryPerMultiProcessor;
    info->wavefrontSize = prop.warpSize;
    strncpy(info->archName, prop.gcnArchName, sizeof(info->archName) - 1);
    info->archName[sizeof(info->archName) - 1] = 0;
    return 0;
}

// ---- kernel timing probe ------------------------------------------------------------------------------
namespace {
std::mutex g_probeMutex;
struct Probe {
    bool on = false;
    std::vector<hipEvent_t> pool;            // reusable events
    size_t used = 0;
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<Rec> recs;
    hipEvent_t pending[PK_COUNT] = {};
    hipEvent_t get()
    {
        if (used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
        return pool[used++];
    }
} g_probe;
}
void probe_before(int id, hipStream_t s)
{
    if (!g_probe.on) return;
    std::lock_guard<std::mutex> lock(g_probeMutex);
    hipEvent_t e = g_probe.get();
    g_probe.pending[id] = e;
    if (e) (void)hipEventRecord(e, s);
}
void probe_after(int id, hipStream_t s)
{
    if (!g_probe.on) return;
    std::lock_guard<std::mutex> lock(g_probeMutex);
    hipEvent_t e = g_probe.get();
    if (e && g_probe.pending[id]) { (void)hipEventRecord(e, s); g_probe.recs.push_back({ id, g_probe.pending[id], e }); }
}
extern "C" int FSEHIP_probe_begin(void)
{
    std::lock_guard<std::mutex> lock(g_probeMutex);
    g_probe.on = true; g_probe.used = 0; g_probe.recs.clear();
    return 0;
}
extern "C" int FSEHIP_probe_collect(double* totalMs, unsigned* launches)
{
    for (int i = 0; i < 16; ++i) { totalMs[i] = 0; launches[i] = 0; }
    std::lock_guard<std::mutex> lock(g_probeMutex);
    g_probe.on = false;
    for (auto& r : g_probe.recs) {
        CK(hipEventSynchronize(r.b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, r.a, r.b));
        totalMs[r.id] += ms; launches[r.id] += 1;
    }
    g_probe.recs.clear(); g_probe.used = 0;
    return 0;
}

static inline BlockView mkview(const void* base, size_t stride, const size_t* sizes, size_t uniform)
{
    BlockView v; v.base = (const u8*)base; v.stride = stride; v.sizes = sizes; v.uniform = uniform; v.offsets = nullptr; return v;
}

// =====================================================================================================
//  workload generator
// =====================================================================================================
extern "C" void FSEHIP_probagen_table(uint8_t table[4096], double p)   // programs/probaGenerator.c:95-118
{
    int remaining = 4096;
    unsigned pos = 0, s = 0;
    if (p == 0.0) p = 0.005;
    while (remaining) {
        unsigned n = (unsigned)(remaining * p);
        if (!n) n = 1;
        memset(table + pos, (int)(uint8_t)s, n);
        pos += n; s++; remaining -= (int)n;
    }
}

extern "C" int FSEHIP_probagen_batch(void* d_dst, size_t dstStride, size_t blockSize, size_t nBlocks,
                                     const uint8_t h_table[4096], uint32_t firstSeed, void* stream)
{
    return FSEHIP_probagen_batch_ex(d_dst, dstStride, blockSize, nBlocks, h_table, firstSeed, 1, stream);
}
extern "C" int FSEHIP_probagen_batch_ex(void* d_dst, size_t dstStride, size_t blockSize, size_t nBlocks,
                                        const uint8_t h_table[4096], uint32_t firstSeed, uint32_t seedStep, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    // The 4 KiB table goes through a small per-thread, per-device ring of slots (device memory + pinned staging, allocated once): no
    // hipMalloc / hipFree and no host synchronisation per call.  h_table is consumed before the call returns (host copy into
    // the pinned slot); a slot is reused 16 calls later, after the event recorded behind the kernel that read it.
    enum { SLOTS = 16 };
    struct Ring { u8* dev = nullptr; u8* pin = nullptr; int device = -1; unsigned next = 0; hipEvent_t ev[SLOTS] = {}; bool used[SLOTS] = {}; int nEv = 0;
                  void drop() { for (int i = 0; i < nEv; ++i) (void)hipEventDestroy(ev[i]); nEv = 0;
                                if (pin) (void)hipHostFree(pin); if (dev) (void)hipFree(dev); pin = nullptr; dev = nullptr; device = -1; (void)hipGetLastError(); } };
    struct Rings { std::vector<Ring> v; ~Rings() { for (auto& r : v) r.drop(); } };
    static thread_local Rings rings;                              // one ring per device the thread has generated on
    int dev = 0;
    CK(hipGetDevice(&dev));
    Ring* rp = nullptr;
    for (auto& r : rings.v) if (r.device == dev) rp = &r;
    if (!rp) {
        Ring r;
        hipError_t e = hipMalloc((void**)&r.dev, SLOTS * 4096);
        if (e == hipSuccess) e = hipHostMalloc((void**)&r.pin, SLOTS * 4096, hipHostMallocDefault);
        for (int i = 0; i < SLOTS && e == hipSuccess; ++i) { e = hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming); if (e == hipSuccess) r.nEv = i + 1; }
        if (e != hipSuccess) { r.drop(); return (int)e; }         // nothing half-built is kept
        r.device = dev;
        rings.v.push_back(r);
        rp = &rings.v.back();
    }
    Ring& ring = *rp;
    const unsigned k = ring.next++ % SLOTS;
    if (ring.used[k]) CK(hipEventSynchronize(ring.ev[k]));
    memcpy(ring.pin + 4096u * k, h_table, 4096);
    u8* const d_table = ring.dev + 4096u * k;
    hipError_t e = hipMemcpyAsync(d_table, ring.pin + 4096u * k, 4096, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_probagen((u8*)d_dst, dstStride, blockSize, nBlocks, d_table, firstSeed, seedStep, s);
    if (e == hipSuccess) { e = hipEventRecord(ring.ev[k], s); ring.used[k] = e == hipSuccess; }
    return (int)e;
}

// Argument errors of the one-shot calls are per-block results, like everything else the reference call would return for
// block b (include/fsehip.h): the batch call itself still succeeds.
//   mode 0 (FSE_compress2, lib/fse_compress.c:691): every block gets `code`;
//   mode 1 (HUF_compress_internal, lib/huf_compress.c:654-660): srcSize 0 or dstCapacity 0 -> 0, srcSize > HUF_BLOCKSIZE_MAX ->
//          srcSize_wrong come first, then `code`;
//   mode 2 (FSE_compress_wksp, lib/fse_compress.c:646-650): srcSize <= 1 -> 0 comes first, then `code`.
__global__ void k_batch_arg_error(size_t* results, const size_t* sizes, size_t uniform, size_t dstCapacity, size_t nBlocks, size_t code, int mode)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    size_t r = code;
    if (mode == 1) {
        const size_t n = sizes ? sizes[b] : uniform;
        if (n == 0 || dstCapacity == 0) r = 0;
        else if (n > FSEHIP_HUF_BLOCKSIZE_MAX) r = FERR(srcSize_wrong);
    }
    if (mode == 2) { const size_t n = sizes ? sizes[b] : uniform; if (n <= 1) r = 0; }
    results[b] = r;
}
static int batch_arg_error(size_t* d_results, const size_t* d_sizes, size_t uniform, size_t dstCapacity, size_t nBlocks, size_t code, int mode, hipStream_t s)
{
    hipLaunchKernelGGL(k_batch_arg_error, dim3((unsigned)((nBlocks + 255) / 256)), dim3(256), 0, s, d_results, d_sizes, uniform, dstCapacity, nBlocks, code, mode);
    return (int)hipGetLastError();
}

// =====================================================================================================
//  a1: HIST_count
// =====================================================================================================
extern "C" int FSEHIP_HIST_count_batch(unsigned* d_counts, unsigned* d_maxSymbolValues, size_t* d_results,
                                       const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                       size_t nBlocks, void* stream)
{
    HistArgs a;
    a.counts = d_counts; a.maxSVs = d_maxSymbolValues; a.uniformMaxSV = 255; a.useUniformIn = 0;
    a.results = d_results; a.src = mkview(d_src, srcStride, d_sizes, uniformSize); a.nBlocks = nBlocks;
    return (int)launch_hist(a, (hipStream_t)stream);
}

// =====================================================================================================
//  a2 / a3: FSE hot loops over a batch
// =====================================================================================================
extern "C" int FSEHIP_FSE_compress_usingCTable_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                     const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                                     const FSEHIP_FSE_CTable* d_ctables, size_t ctableStrideU32, unsigned maxTableLog,
                                                     size_t nBlocks, void* stream)
{
    if (maxTableLog == 0 || maxTableLog > FSEHIP_FSE_MAX_TABLELOG) maxTableLog = FSEHIP_FSE_MAX_TABLELOG;
    FseEncArgs a;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.dstCapacity = dstCapacity; a.results = d_results;
    a.src = mkview(d_src, srcStride, d_sizes, uniformSize);
    a.ctables = d_ctables; a.ctStrideU32 = ctableStrideU32; a.meta = nullptr;
    a.maxTableLog = maxTableLog; a.G = 0; a.slotU32 = 0; a.nBlocks = nBlocks; a.list = nullptr; a.count = nullptr;
    return (int)launch_fse_encode_auto(a, (hipStream_t)stream);
}

extern "C" int FSEHIP_FSE_decompress_usingDTable_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                       const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                       const FSEHIP_FSE_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                       size_t nBlocks, void* stream)
{
    if (maxTableLog == 0 || maxTableLog > FSEHIP_FSE_MAX_TABLELOG) maxTableLog = FSEHIP_FSE_MAX_TABLELOG;
    FseDecArgs a;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.dstCapacity = dstCapacity; a.results = d_results;
    a.csrc = mkview(d_cSrc, cStride, d_cSizes, uniformCSize);
    a.dtables = d_dtables; a.dtStrideU32 = dtableStrideU32; a.atab = nullptr; a.symtab = nullptr; a.meta = nullptr;
    a.maxTableLog = maxTableLog; a.G = 0; a.slotU32 = 0; a.nBlocks = nBlocks; a.tlMin = 0; a.declineNb0 = 0; a.onlyDeclined = 0;
    a.symScratch = nullptr; a.slotBitmap = nullptr; a.nSlots = 0; a.scratchSlotBytes = 0;
    return (int)launch_fse_decode(a, (hipStream_t)stream);
}

// =====================================================================================================
//  one-shot FSE block API over a batch
// =====================================================================================================
struct FseCWs { size_t perBlock; size_t ctU32; size_t ts; unsigned maxTl; };
static FseCWs fse_cws(unsigned tableLog)
{
    FseCWs w;
    unsigned tl = tableLog ? tableLog : FSEHIP_FSE_DEFAULT_TABLELOG;
    if (tl < 9) tl = 9;                  // FSE_optimalTableLog may raise a small request up to highbit(255)+2 (fse_compress.c:316-333)
    if (tl > FSEHIP_FSE_MAX_TABLELOG) tl = FSEHIP_FSE_MAX_TABLELOG;
    w.maxTl = tl;
    w.ctU32 = FSEHIP_FSE_CTABLE_SIZE_U32(tl, 255);
    w.ts = (size_t)1 << tl;
    w.perBlock = 1024 + 4 + 8 + sizeof(FseMeta) + 4 * w.ctU32 + FSE_EBINS * sizeof(u32);
    return w;
}
#define WS_SLACK 2048
#define WS_MAX_CHUNK 131072          // blocks per pass over the workspace (tables of 131072 blocks: 0.8 GiB)
// largest chunk <= limit that is a whole number of device-filling rounds of the hot-loop kernel (no ragged last wave of workgroups)
static size_t round_chunk(size_t limit, size_t perRound)
{
    if (perRound == 0 || limit < perRound) return limit;
    return limit / perRound * perRound;
}

extern "C" size_t FSEHIP_FSE_compress_batch_workspaceSize(size_t nBlocks, unsigned tableLog)
{
    const FseCWs w = fse_cws(tableLog);
    size_t c = nBlocks < WS_MAX_CHUNK ? nBlocks : WS_MAX_CHUNK;
    if (c == 0) c = 1;
    return c * w.perBlock + WS_SLACK;
}

extern "C" int FSEHIP_FSE_compress_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                         const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                         unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks,
                                         void* d_workspace, size_t workspaceBytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;          // include/fsehip.h: workspaces are 256-byte aligned; checked before anything else
    if (nBlocks == 0) return 0;
    if (tableLog > FSEHIP_FSE_MAX_TABLELOG)                                       // FSE_compress2 -> tableLog_tooLarge for every block (fse_compress.c:691)
        return batch_arg_error(d_results, nullptr, 0, dstCapacity, nBlocks, FSEHIP_ERROR(tableLog_tooLarge), 0, s);
    if (maxSymbolValue > 255 && tableLog != 0) {
        // FSE_compress2 carves its histogram scratch out of a fixed workspace behind a CTable sized from the REQUESTED maxSymbolValue
        // (lib/fse_compress.c:640-642,680-686): a request above 255 at tableLog 12 leaves the histogram less than HIST_WKSP_SIZE and
        // HIST_count_wksp refuses (lib/hist.c:168) -- after the srcSize <= 1 early-out.  Where the table still fits, the histogram
        // clamps the limit to 255 (lib/hist.c:169-172) and the call behaves as with 255; beyond the workspace the reference is undefined.
        const size_t wksp = 4 * (size_t)FSEHIP_FSE_CTABLE_SIZE_U32(FSEHIP_FSE_MAX_TABLELOG, 255) + ((size_t)1 << FSEHIP_FSE_MAX_TABLELOG);
        const size_t ctBytes = 4 * (1 + ((size_t)1 << (tableLog - 1)) + 2 * ((size_t)maxSymbolValue + 1));
        if (ctBytes <= wksp && wksp - ctBytes < 4096)
            return batch_arg_error(d_results, d_sizes, uniformSize, dstCapacity, nBlocks, FSEHIP_ERROR(workSpace_tooSmall), 2, s);
    }
    const FseCWs w = fse_cws(tableLog);
    if (workspaceBytes < w.perBlock + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / w.perBlock;
    if (chunk >= nBlocks) chunk = nBlocks;
    else chunk = round_chunk(chunk, fse_encode_blocks_per_round(w.maxTl));
    // carve the workspace
    u8* p = (u8*)d_workspace;
    auto carve = [&](size_t bytes) { u8* r = p; p += align_up(bytes, 256); return r; };
    unsigned* counts = (unsigned*)carve(chunk * 1024);
    unsigned* maxSVs = (unsigned*)carve(chunk * 4);
    size_t* hres = (size_t*)carve(chunk * 8);
    FseMeta* meta = (FseMeta*)carve(chunk * sizeof(FseMeta));
    u32* ctables = (u32*)carve(chunk * 4 * w.ctU32);
    u32* encLists = (u32*)carve(chunk * FSE_EBINS * sizeof(u32));
    u32* encCounts = (u32*)carve(FSE_EBINS * sizeof(u32));
    if ((size_t)(p - (u8*)d_workspace) > workspaceBytes) {
        // alignment slack exhausted (WS_SLACK covers the 256-byte padding of the seven regions)
        return (int)hipErrorInvalidValue;
    }
    unsigned msv = maxSymbolValue ? maxSymbolValue : 255;        // fse_compress.c:648
    if (msv > 255) msv = 255;
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        const BlockView src = mkview((const u8*)d_src + b0 * srcStride, srcStride, d_sizes ? d_sizes + b0 : nullptr, uniformSize);
        HistArgs h;
        h.counts = counts; h.maxSVs = maxSVs; h.uniformMaxSV = msv; h.useUniformIn = 1; h.results = hres; h.src = src; h.nBlocks = nb;
        CK(launch_hist(h, s));
        FseCPrepArgs c;
        c.counts = counts; c.maxSVs = maxSVs; c.histResults = hres; c.src = src;
        c.dst = (u8*)d_dst + b0 * dstStride; c.dstStride = dstStride; c.dstCapacity = dstCapacity;
        c.maxSVReq = msv; c.tableLogReq = tableLog;
        c.ctables = ctables; c.ctStrideU32 = w.ctU32; c.maxTl = w.maxTl;
        c.meta = meta; c.results = d_results + b0; c.nBlocks = nb;
        CK(launch_fse_cprep(c, s));
        FseEncArgs e;
        e.dst = (u8*)d_dst + b0 * dstStride; e.dstStride = dstStride; e.dstCapacity = dstCapacity; e.results = d_results + b0;
        e.src = src; e.ctables = ctables; e.ctStrideU32 = w.ctU32; e.meta = meta;
        e.maxTableLog = w.maxTl; e.G = 0; e.slotU32 = 0; e.nBlocks = nb; e.list = encLists; e.count = encCounts;
        CK(launch_fse_encode_auto(e, s));
    }
    return 0;
}

// per block: meta, 256 counters, the decoder-format table (2 + 1 bytes per cell) and one entry in each decoder-class list
static size_t fse_dws_per_block(unsigned maxLog) { return sizeof(FseMeta) + 512 + 3 * ((size_t)1 << maxLog) + FSE_DCLS_COUNT * sizeof(u32); }
static unsigned clamp_maxlog(unsigned maxLog) { return (maxLog == 0 || maxLog > FSEHIP_FSE_MAX_TABLELOG) ? FSEHIP_FSE_MAX_TABLELOG : maxLog; }

extern "C" size_t FSEHIP_FSE_decompress_batch_workspaceSize(size_t nBlocks, unsigned maxLog)
{
    size_t c = nBlocks < WS_MAX_CHUNK ? nBlocks : WS_MAX_CHUNK;
    if (c == 0) c = 1;
    return c * fse_dws_per_block(clamp_maxlog(maxLog)) + WS_SLACK;
}

static inline BlockView subview(const BlockView& v, size_t b0)      // blocks b0.. of a view
{
    BlockView r = v;
    if (v.offsets) r.offsets = v.offsets + b0;
    else { r.base = v.base + b0 * v.stride; r.sizes = v.sizes ? v.sizes + b0 : nullptr; }
    return r;
}
// rawRle: the bench loop's treatment of blocks the compressor declined (programs/bench.c:393-406) -- a record as long as the block is the
// block itself, a record of one byte is that byte repeated -- applied by k_rawrle_expand (compact.hip); k_fse_dparse then leaves those alone
hipError_t launch_rawrle_expand(u8* dst, size_t dstStride, size_t dstCapacity, size_t* results, const BlockView& csrc, const size_t* origSizes, size_t uniformOrig,
                                size_t nBlocks, hipStream_t s);
static int fse_decompress_impl(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results, const BlockView& csAll, unsigned maxLog, size_t nBlocks,
                               void* d_workspace, size_t workspaceBytes, hipStream_t s, const size_t* d_origSizes, size_t uniformOrig, int rawRle)
{
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;          // include/fsehip.h: workspaces are 256-byte aligned; checked before anything else
    if (nBlocks == 0) return 0;
    maxLog = clamp_maxlog(maxLog);
    const size_t per = fse_dws_per_block(maxLog);
    if (workspaceBytes < per + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / per;
    if (chunk >= nBlocks) chunk = nBlocks;
    else chunk = round_chunk(chunk, fse_decode_blocks_per_round(maxLog));
    u8* p = (u8*)d_workspace;
    FseMeta* meta = (FseMeta*)p; p += align_up(chunk * sizeof(FseMeta), 256);
    s16* norms = (s16*)p; p += align_up(chunk * 512, 256);
    u16* atab = (u16*)p; p += align_up((chunk * 2) << maxLog, 256);
    u8* symtab = p; p += align_up(chunk << maxLog, 256);
    u32* lists = (u32*)p; p += align_up(chunk * FSE_DCLS_COUNT * sizeof(u32), 256);
    u32* counts = (u32*)p;                                      // FSE_DCLS_COUNT words (WS_SLACK covers the padding and this)
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        const BlockView cs = subview(csAll, b0);
        if (rawRle) CK(launch_rawrle_expand((u8*)d_dst + b0 * dstStride, dstStride, dstCapacity, d_results + b0, cs, d_origSizes ? d_origSizes + b0 : nullptr, uniformOrig, nb, s));
        FseDPrepArgs d;
        d.csrc = cs; d.maxLog = maxLog; d.atab = atab; d.symtab = symtab; d.norms = norms; d.meta = meta; d.lists = lists; d.counts = counts;
        d.results = d_results + b0; d.nBlocks = nb;
        d.rawRle = rawRle; d.origSizes = d_origSizes ? d_origSizes + b0 : nullptr; d.uniformOrig = uniformOrig;
        CK(launch_fse_dprep(d, s));
        FseDecArgs e;
        e.dst = (u8*)d_dst + b0 * dstStride; e.dstStride = dstStride; e.dstCapacity = dstCapacity; e.results = d_results + b0;
        e.csrc = cs; e.dtables = nullptr; e.dtStrideU32 = 0; e.atab = atab; e.symtab = symtab; e.meta = meta;
        e.maxTableLog = maxLog; e.G = 0; e.slotU32 = 0; e.nBlocks = nb; e.tlMin = 0; e.declineNb0 = 0; e.onlyDeclined = 0;
        e.symScratch = nullptr; e.slotBitmap = nullptr; e.nSlots = 0; e.scratchSlotBytes = 0;
        CK(launch_fse_decode_classes(e, lists, counts, s));
    }
    return 0;
}
extern "C" int FSEHIP_FSE_decompress_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                           const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                           unsigned maxLog, size_t nBlocks,
                                           void* d_workspace, size_t workspaceBytes, void* stream)
{
    return fse_decompress_impl(d_dst, dstStride, dstCapacity, d_results, mkview(d_cSrc, cStride, d_cSizes, uniformCSize), maxLog, nBlocks,
                               d_workspace, workspaceBytes, (hipStream_t)stream, nullptr, 0, 0);
}
// FSE_decompress over a PACKED batch (FSEHIP_compact_batch), with the bench loop's treatment of declined blocks
extern "C" int FSEHIP_FSE_decompress_packed_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                  const void* d_packed, const uint64_t* d_offsets, const size_t* d_origSizes, size_t uniformOrigSize,
                                                  unsigned maxLog, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream)
{
    BlockView v = mkview(d_packed, 0, nullptr, 0);
    v.offsets = (const u64*)d_offsets;
    return fse_decompress_impl(d_dst, dstStride, dstCapacity, d_results, v, maxLog, nBlocks, d_workspace, workspaceBytes, (hipStream_t)stream,
                               d_origSizes, uniformOrigSize, 1);
}

// =====================================================================================================
//  Tables for the *_usingCTable / *_usingDTable batch calls, built on the device (SURVEY 8(a') g1-g3, g5-g6 as calls of their own)
// =====================================================================================================
__global__ void k_hdr_results(const u8* meta, size_t metaStride, size_t* results, size_t nBlocks)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    const u32* const m = (const u32*)(meta + b * metaStride);            // {state, hdrSize, ...} in FseMeta and HufMeta alike
    if (m[0] != 0) results[b] = m[1];
}
hipError_t launch_hdr_results(const void* meta, size_t metaStride, size_t* results, size_t nBlocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_hdr_results, dim3((unsigned)((nBlocks + 255) / 256)), dim3(256), 0, s, (const u8*)meta, metaStride, results, nBlocks);
    return hipGetLastError();
}

static const size_t FSE_BCT_PER_BLOCK = 1024 + 4 + 8 + sizeof(FseMeta);
extern "C" size_t FSEHIP_FSE_buildCTable_batch_workspaceSize(size_t nBlocks)
{
    size_t c = nBlocks < WS_MAX_CHUNK ? nBlocks : WS_MAX_CHUNK;
    return (c ? c : 1) * FSE_BCT_PER_BLOCK + WS_SLACK;
}
extern "C" int FSEHIP_FSE_buildCTable_batch(FSEHIP_FSE_CTable* d_ctables, size_t ctableStrideU32, void* d_headers, size_t headerStride, size_t headerCapacity,
                                            size_t* d_results, const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                            unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;
    if (nBlocks == 0) return 0;
    if (tableLog > FSEHIP_FSE_MAX_TABLELOG) return batch_arg_error(d_results, nullptr, 0, headerCapacity, nBlocks, FSEHIP_ERROR(tableLog_tooLarge), 0, s);
    const FseCWs w = fse_cws(tableLog);
    if (ctableStrideU32 < w.ctU32) return (int)hipErrorInvalidValue;              // room for FSE_CTABLE_SIZE_U32(largest table log the request can lead to, 255)
    if (workspaceBytes < FSE_BCT_PER_BLOCK + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / FSE_BCT_PER_BLOCK;
    if (chunk > nBlocks) chunk = nBlocks;
    u8* p = (u8*)d_workspace;
    auto carve = [&](size_t bytes) { u8* r = p; p += align_up(bytes, 256); return r; };
    unsigned* counts = (unsigned*)carve(chunk * 1024);
    unsigned* maxSVs = (unsigned*)carve(chunk * 4);
    size_t* hres = (size_t*)carve(chunk * 8);
    FseMeta* meta = (FseMeta*)carve(chunk * sizeof(FseMeta));
    if ((size_t)(p - (u8*)d_workspace) > workspaceBytes) return (int)hipErrorInvalidValue;
    unsigned msv = maxSymbolValue ? maxSymbolValue : 255;
    if (msv > 255) msv = 255;
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        const BlockView src = mkview((const u8*)d_src + b0 * srcStride, srcStride, d_sizes ? d_sizes + b0 : nullptr, uniformSize);
        HistArgs h;
        h.counts = counts; h.maxSVs = maxSVs; h.uniformMaxSV = msv; h.useUniformIn = 1; h.results = hres; h.src = src; h.nBlocks = nb;
        CK(launch_hist(h, s));
        FseCPrepArgs c;
        c.counts = counts; c.maxSVs = maxSVs; c.histResults = hres; c.src = src;
        c.dst = (u8*)d_headers + b0 * headerStride; c.dstStride = headerStride; c.dstCapacity = headerCapacity;
        c.maxSVReq = msv; c.tableLogReq = tableLog;
        c.ctables = d_ctables + b0 * ctableStrideU32; c.ctStrideU32 = ctableStrideU32; c.maxTl = w.maxTl;
        c.meta = meta; c.results = d_results + b0; c.nBlocks = nb;
        CK(launch_fse_cprep(c, s));
        CK(launch_hdr_results(meta, sizeof(FseMeta), d_results + b0, nb, s));
    }
    return 0;
}

// ---- the glue steps as calls of their own (fsehip.h "Table glue, step by step")
extern "C" int FSEHIP_FSE_normalizeCount_batch(short* d_norms, size_t normStride, unsigned tableLog, const unsigned* d_counts, size_t countStride,
                                               const size_t* d_totals, const unsigned* d_maxSymbolValues, size_t nBlocks, size_t* d_results, void* stream)
{
    if (nBlocks == 0) return 0;
    if (!d_norms || !d_counts || !d_totals || !d_maxSymbolValues || !d_results || normStride < 256 || countStride < 256) return (int)hipErrorInvalidValue;
    return (int)launch_fse_glue_normalize((s16*)d_norms, normStride, tableLog, d_counts, countStride, d_totals, d_maxSymbolValues, d_results, nBlocks, (hipStream_t)stream);
}
extern "C" int FSEHIP_FSE_writeNCount_batch(void* d_headers, size_t headerStride, size_t headerCapacity, const short* d_norms, size_t normStride,
                                            const unsigned* d_maxSymbolValues, unsigned tableLog, size_t nBlocks, size_t* d_results, void* stream)
{
    if (nBlocks == 0) return 0;
    if (!d_headers || !d_norms || !d_maxSymbolValues || !d_results || normStride < 256 || headerCapacity > headerStride) return (int)hipErrorInvalidValue;
    return (int)launch_fse_glue_write_ncount((u8*)d_headers, headerStride, headerCapacity, (const s16*)d_norms, normStride, d_maxSymbolValues, tableLog, d_results, nBlocks,
                                             (hipStream_t)stream);
}
extern "C" int FSEHIP_FSE_readNCount_batch(short* d_norms, size_t normStride, unsigned* d_maxSymbolValues, unsigned* d_tableLogs,
                                           const void* d_headers, size_t headerStride, const size_t* d_headerSizes, size_t uniformHeaderSize,
                                           size_t nBlocks, size_t* d_results, void* stream)
{
    if (nBlocks == 0) return 0;
    if (!d_norms || !d_maxSymbolValues || !d_tableLogs || !d_headers || !d_results) return (int)hipErrorInvalidValue;
    return (int)launch_fse_glue_read_ncount((s16*)d_norms, normStride, d_maxSymbolValues, d_tableLogs, mkview(d_headers, headerStride, d_headerSizes, uniformHeaderSize),
                                            d_results, nBlocks, (hipStream_t)stream);
}

extern "C" size_t FSEHIP_FSE_buildDTable_batch_workspaceSize(size_t nBlocks, unsigned maxLog) { return FSEHIP_FSE_decompress_batch_workspaceSize(nBlocks, maxLog); }
extern "C" int FSEHIP_FSE_buildDTable_batch(FSEHIP_FSE_DTable* d_dtables, size_t dtableStrideU32, size_t* d_results,
                                            const void* d_headers, size_t headerStride, const size_t* d_headerSizes, size_t uniformHeaderSize,
                                            unsigned maxLog, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;
    if (nBlocks == 0) return 0;
    maxLog = clamp_maxlog(maxLog);
    if (dtableStrideU32 < FSEHIP_FSE_DTABLE_SIZE_U32(maxLog)) return (int)hipErrorInvalidValue;
    const size_t per = fse_dws_per_block(maxLog);
    if (workspaceBytes < per + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / per;
    if (chunk > nBlocks) chunk = nBlocks;
    u8* p = (u8*)d_workspace;
    FseMeta* meta = (FseMeta*)p; p += align_up(chunk * sizeof(FseMeta), 256);
    s16* norms = (s16*)p; p += align_up(chunk * 512, 256);
    u16* atab = (u16*)p; p += align_up((chunk * 2) << maxLog, 256);
    u8* symtab = p; p += align_up(chunk << maxLog, 256);
    u32* lists = (u32*)p; p += align_up(chunk * FSE_DCLS_COUNT * sizeof(u32), 256);
    u32* counts = (u32*)p;
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        FseDPrepArgs d;
        d.csrc = mkview((const u8*)d_headers + b0 * headerStride, headerStride, d_headerSizes ? d_headerSizes + b0 : nullptr, uniformHeaderSize);
        d.maxLog = maxLog; d.atab = atab; d.symtab = symtab; d.norms = norms; d.meta = meta; d.lists = lists; d.counts = counts;
        d.results = d_results + b0; d.nBlocks = nb; d.rawRle = 0; d.origSizes = nullptr; d.uniformOrig = 0;
        CK(launch_fse_dprep(d, s));
        CK(launch_fse_export_dtables(d, d_dtables + b0 * dtableStrideU32, dtableStrideU32, s));
        CK(launch_hdr_results(meta, sizeof(FseMeta), d_results + b0, nb, s));
    }
    return 0;
}

// ---- the table builders on counters the caller supplies (fsehip.h "Table glue, step by step")
extern "C" int FSEHIP_FSE_buildCTable_fromNorm_batch(FSEHIP_FSE_CTable* d_ctables, size_t ctableStrideU32, const short* d_norms, size_t normStride,
                                                     const unsigned* d_maxSymbolValues, unsigned tableLog, size_t nBlocks, size_t* d_results, void* stream)
{
    if (nBlocks == 0) return 0;
    if (!d_ctables || !d_norms || !d_maxSymbolValues || !d_results || normStride < 256) return (int)hipErrorInvalidValue;
    if (tableLog >= 1 && tableLog <= FSEHIP_FSE_MAX_TABLELOG && ctableStrideU32 < FSEHIP_FSE_CTABLE_SIZE_U32(tableLog, 255)) return (int)hipErrorInvalidValue;
    return (int)launch_fse_ctable_from_norm((const s16*)d_norms, normStride, d_maxSymbolValues, tableLog, d_ctables, ctableStrideU32, d_results, nBlocks, (hipStream_t)stream);
}
extern "C" size_t FSEHIP_FSE_buildDTable_fromNorm_batch_workspaceSize(size_t nBlocks, unsigned tableLog) { return FSEHIP_FSE_decompress_batch_workspaceSize(nBlocks, tableLog); }
extern "C" int FSEHIP_FSE_buildDTable_fromNorm_batch(FSEHIP_FSE_DTable* d_dtables, size_t dtableStrideU32, const short* d_norms, size_t normStride,
                                                     const unsigned* d_maxSymbolValues, unsigned tableLog, size_t nBlocks, size_t* d_results,
                                                     void* d_workspace, size_t workspaceBytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;
    if (nBlocks == 0) return 0;
    if (!d_dtables || !d_norms || !d_maxSymbolValues || !d_results || normStride < 256) return (int)hipErrorInvalidValue;
    if (tableLog > FSEHIP_FSE_MAX_TABLELOG) return batch_arg_error(d_results, nullptr, 0, 0, nBlocks, FSEHIP_ERROR(tableLog_tooLarge), 0, s);   // lib/fse_decompress.c:84
    const unsigned maxLog = tableLog ? tableLog : 1;                     // (tableLog 0: refused per block, GENERIC)
    if (dtableStrideU32 < FSEHIP_FSE_DTABLE_SIZE_U32(maxLog)) return (int)hipErrorInvalidValue;
    const size_t per = fse_dws_per_block(maxLog);
    if (workspaceBytes < per + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / per;
    if (chunk > nBlocks) chunk = nBlocks;
    u8* p = (u8*)d_workspace;
    FseMeta* meta = (FseMeta*)p; p += align_up(chunk * sizeof(FseMeta), 256);
    s16* norms = (s16*)p; p += align_up(chunk * 512, 256);
    u16* atab = (u16*)p; p += align_up((chunk * 2) << maxLog, 256);
    u8* symtab = p; p += align_up(chunk << maxLog, 256);
    u32* lists = (u32*)p; p += align_up(chunk * FSE_DCLS_COUNT * sizeof(u32), 256);
    u32* counts = (u32*)p;
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        FseDPrepArgs d;
        d.csrc = mkview(nullptr, 0, nullptr, 0);
        d.maxLog = maxLog; d.atab = atab; d.symtab = symtab; d.norms = norms; d.meta = meta; d.lists = lists; d.counts = counts;
        d.results = d_results + b0; d.nBlocks = nb; d.rawRle = 0; d.origSizes = nullptr; d.uniformOrig = 0;
        CK(launch_fse_dprep_from_norm(d, (const s16*)d_norms + b0 * normStride, normStride, d_maxSymbolValues + b0, tableLog, s));
        CK(launch_fse_export_dtables(d, d_dtables + b0 * dtableStrideU32, dtableStrideU32, s));
        CK(launch_hdr_results(meta, sizeof(FseMeta), d_results + b0, nb, s));   // (hdrSize 0: FSE_buildDTable returns 0)
    }
    return 0;
}

static const size_t HUF_BCT_PER_BLOCK = 1024 + 4 + 8 + sizeof(HufMeta);
extern "C" size_t FSEHIP_HUF_buildCTable_batch_workspaceSize(size_t nBlocks)
{
    size_t c = nBlocks < WS_MAX_CHUNK ? nBlocks : WS_MAX_CHUNK;
    return (c ? c : 1) * HUF_BCT_PER_BLOCK + WS_SLACK;
}
extern "C" int FSEHIP_HUF_buildCTable_batch(FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32, void* d_headers, size_t headerStride, size_t headerCapacity,
                                            size_t* d_results, const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                            unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;
    if (nBlocks == 0) return 0;
    if (tableLog > FSEHIP_HUF_TABLELOG_MAX || maxSymbolValue > 255)
        return batch_arg_error(d_results, d_sizes, uniformSize, headerCapacity, nBlocks,
                               tableLog > FSEHIP_HUF_TABLELOG_MAX ? FSEHIP_ERROR(tableLog_tooLarge) : FSEHIP_ERROR(maxSymbolValue_tooLarge), 1, s);
    if (ctableStrideU32 < 256) return (int)hipErrorInvalidValue;
    if (workspaceBytes < HUF_BCT_PER_BLOCK + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / HUF_BCT_PER_BLOCK;
    if (chunk > nBlocks) chunk = nBlocks;
    u8* p = (u8*)d_workspace;
    auto carve = [&](size_t bytes) { u8* r = p; p += align_up(bytes, 256); return r; };
    unsigned* counts = (unsigned*)carve(chunk * 1024);
    unsigned* maxSVs = (unsigned*)carve(chunk * 4);
    size_t* hres = (size_t*)carve(chunk * 8);
    HufMeta* meta = (HufMeta*)carve(chunk * sizeof(HufMeta));
    if ((size_t)(p - (u8*)d_workspace) > workspaceBytes) return (int)hipErrorInvalidValue;
    const unsigned msv = maxSymbolValue ? maxSymbolValue : 255;
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        const BlockView src = mkview((const u8*)d_src + b0 * srcStride, srcStride, d_sizes ? d_sizes + b0 : nullptr, uniformSize);
        HistArgs h;
        h.counts = counts; h.maxSVs = maxSVs; h.uniformMaxSV = msv; h.useUniformIn = 1; h.results = hres; h.src = src; h.nBlocks = nb;
        CK(launch_hist(h, s));
        HufCPrepArgs c;
        c.counts = counts; c.maxSVs = maxSVs; c.histResults = hres; c.src = src;
        c.dst = (u8*)d_headers + b0 * headerStride; c.dstStride = headerStride; c.dstCapacity = headerCapacity;
        c.maxSVReq = msv; c.huffLogReq = tableLog; c.ctables = d_ctables + b0 * ctableStrideU32; c.ctStrideU32 = ctableStrideU32;
        c.meta = meta; c.results = d_results + b0; c.nBlocks = nb;
        CK(launch_huf_cprep(c, s, nullptr));
        CK(launch_hdr_results(meta, sizeof(HufMeta), d_results + b0, nb, s));
    }
    return 0;
}

// ---- the Huff0 table glue on counters / tables the caller supplies (fsehip.h "Table glue, step by step"): HUF_buildCTable and HUF_writeCTable
extern "C" int FSEHIP_HUF_buildCTable_fromCount_batch(FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32, const unsigned* d_counts, size_t countStride,
                                                      const unsigned* d_maxSymbolValues, unsigned maxNbBits, size_t nBlocks, size_t* d_results, void* stream)
{
    if (nBlocks == 0) return 0;
    if (!d_ctables || !d_counts || !d_maxSymbolValues || !d_results || ctableStrideU32 < 256 || countStride != 256) return (int)hipErrorInvalidValue;
    HufCPrepArgs c;
    c.counts = d_counts; c.maxSVs = d_maxSymbolValues; c.histResults = nullptr; c.src = mkview(nullptr, 0, nullptr, 0);
    c.dst = nullptr; c.dstStride = 0; c.dstCapacity = 0; c.maxSVReq = 255; c.huffLogReq = maxNbBits;
    c.ctables = d_ctables; c.ctStrideU32 = ctableStrideU32; c.meta = nullptr; c.results = d_results; c.nBlocks = nBlocks;
    return (int)launch_huf_cprep_glue(c, 1, (hipStream_t)stream);
}
extern "C" int FSEHIP_HUF_writeCTable_batch(void* d_headers, size_t headerStride, size_t headerCapacity, const FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32,
                                            const unsigned* d_maxSymbolValues, unsigned huffLog, size_t nBlocks, size_t* d_results, void* stream)
{
    if (nBlocks == 0) return 0;
    if (!d_headers || !d_ctables || !d_maxSymbolValues || !d_results || ctableStrideU32 < 256 || (ctableStrideU32 & 3) || headerCapacity > headerStride) return (int)hipErrorInvalidValue;
    HufCPrepArgs c;
    c.counts = nullptr; c.maxSVs = d_maxSymbolValues; c.histResults = nullptr; c.src = mkview(nullptr, 0, nullptr, 0);
    c.dst = (u8*)d_headers; c.dstStride = headerStride; c.dstCapacity = headerCapacity; c.maxSVReq = 255; c.huffLogReq = huffLog;
    c.ctables = (u32*)d_ctables; c.ctStrideU32 = ctableStrideU32; c.meta = nullptr; c.results = d_results; c.nBlocks = nBlocks;
    return (int)launch_huf_cprep_glue(c, 2, (hipStream_t)stream);
}

static const size_t HUF_RDT_PER_BLOCK = sizeof(HufMeta) + HUF_DCLS_COUNT * sizeof(u32);
extern "C" size_t FSEHIP_HUF_readDTableX1_batch_workspaceSize(size_t nBlocks)
{
    size_t c = nBlocks < WS_MAX_CHUNK ? nBlocks : WS_MAX_CHUNK;
    return (c ? c : 1) * HUF_RDT_PER_BLOCK + WS_SLACK;
}
extern "C" int FSEHIP_HUF_readDTableX1_batch(FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog, size_t* d_results,
                                             const void* d_src, size_t srcStride, const size_t* d_srcSizes, size_t uniformSrcSize,
                                             size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;
    if (nBlocks == 0) return 0;
    if (maxTableLog == 0 || maxTableLog > FSEHIP_HUF_TABLELOG_MAX) maxTableLog = FSEHIP_HUF_TABLELOG_MAX;
    if (dtableStrideU32 < 1 + ((size_t)1 << maxTableLog)) return (int)hipErrorInvalidValue;       // HUF_DTABLE_SIZE(maxTableLog)
    if (workspaceBytes < HUF_RDT_PER_BLOCK + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / HUF_RDT_PER_BLOCK;
    if (chunk > nBlocks) chunk = nBlocks;
    u8* p = (u8*)d_workspace;
    HufMeta* meta = (HufMeta*)p; p += align_up(chunk * sizeof(HufMeta), 256);
    u32* lists = (u32*)p; p += align_up(chunk * HUF_DCLS_COUNT * sizeof(u32), 256);
    u32* counts = (u32*)p;
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        HufDPrepArgs d;
        d.csrc = mkview((const u8*)d_src + b0 * srcStride, srcStride, d_srcSizes ? d_srcSizes + b0 : nullptr, uniformSrcSize);
        d.dstSizes = mkview(nullptr, 0, nullptr, 0); d.dst = nullptr; d.dstStride = 0;
        d.dtables = d_dtables + b0 * dtableStrideU32; d.dtStrideU32 = dtableStrideU32; d.meta = meta; d.lists = lists; d.counts = counts;
        d.results = d_results + b0; d.nBlocks = nb; d.tableOnly = 1; d.dtMaxLog = maxTableLog;
        CK(launch_huf_dprep(d, s));
    }
    return 0;
}

// =====================================================================================================
//  Layer 1: single-block calls on host pointers = batch of one (H2D, kernels, D2H)
// =====================================================================================================
namespace {
// hipFree takes a pointer of any device, so an arena is given back wherever it was allocated: when the thread moves to another device,
// when it grows, on FSEHIP_releaseScratch and when the thread ends (thread_local destructors run at thread exit and, for the main
// thread, before the destructors of static objects -- the runtime is still there; an error from a runtime already shut down is ignored).
struct Arena { void* base = nullptr; size_t cap = 0, used = 0, live = 0, peak = 0; int dev = -1;
               void drop() { if (base) { (void)hipFree(base); (void)hipGetLastError(); } base = nullptr; cap = 0; used = 0; dev = -1; }
               ~Arena() { if (live == 0) drop(); } };
thread_local Arena t_arena;
}
hipError_t HostCallBuf::alloc(size_t n)
{
    Arena& A = t_arena;
    const size_t need = align_up(n ? n : 1, 256);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (A.live == 0) {                                            // between calls: follow the current device, grow to the last call's peak
        size_t want = A.peak > need ? A.peak : need;
        if (want > FSEHIP_SCRATCH_MAX) want = FSEHIP_SCRATCH_MAX;
        if (A.dev != dev || A.cap < want) {
            A.drop();                                             // (also an arena left on the device the thread used before)
            A.dev = dev;
            if (want < ((size_t)1 << 20)) want = (size_t)1 << 20;
            if (hipMalloc(&A.base, want) == hipSuccess) A.cap = want; else { A.base = nullptr; (void)hipGetLastError(); }
        }
        A.used = 0; A.peak = 0;
    }
    ++A.live;
    A.peak += need;
    if (A.base && A.dev == dev && A.used + need <= A.cap) { p = (u8*)A.base + A.used; A.used += need; carved = need; owned = false; return hipSuccess; }
    owned = true; carved = 0;
    e = hipMalloc(&p, need);
    if (e != hipSuccess) { p = nullptr; --A.live; }
    return e;
}
HostCallBuf::~HostCallBuf()
{
    if (!p) return;
    Arena& A = t_arena;
    if (owned) (void)hipFree(p); else A.used -= carved;           // (stack order: destructors run in reverse order of the allocations)
    --A.live;
}
typedef HostCallBuf DevBuf;
// gives the calling thread's scratch arena back (between calls); the next call on host pointers allocates a new one
int release_thread_scratch(void)                                  // the calling thread's arena
{
    Arena& A = t_arena;
    if (A.live) return (int)hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    if (A.base) e = hipFree(A.base);                              // whatever device the thread is on now
    A.base = nullptr; A.cap = 0; A.used = 0; A.peak = 0; A.dev = -1;
    return (int)e;
}
extern "C" int FSEHIP_releaseScratch(void)
{
    const int r = release_thread_scratch();
    const int rp = frame_pool_release_scratch();                  // ... and those of the frame calls' idle helper threads (frame.hip)
    return r ? r : rp;
}
extern "C" int FSEHIP_shutdown(void)
{
    const int r = release_thread_scratch();
    const int rp = frame_pool_shutdown();
    return r ? r : rp;
}

// result transport for one block: returns GENERIC when the device path itself fails
#define HK(x) do { if ((x) != hipSuccess) return FSEHIP_ERROR(GENERIC); } while (0)

static size_t hist_count_host(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize, int trustInput)
{
    DevBuf dsrc, dcnt, dmsv, dres;
    HK(dsrc.alloc(srcSize)); HK(dcnt.alloc(1024)); HK(dmsv.alloc(4)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, src, srcSize, hipMemcpyHostToDevice));
    HK(hipMemcpy(dmsv.p, maxSymbolValuePtr, 4, hipMemcpyHostToDevice));
    if (srcSize >= HIST_LARGE_MIN) {                        // a whole buffer: pieces counted as a batch and folded (hist.hip)
        const size_t nPart = (srcSize + HIST_PIECE - 1) / HIST_PIECE;
        DevBuf dpart, dpr;
        HK(dpart.alloc(nPart * 1024)); HK(dpr.alloc(nPart * 8));
        HK(launch_hist_large((const u8*)dsrc.p, srcSize, *maxSymbolValuePtr, trustInput, (unsigned*)dpart.p, (unsigned*)dcnt.p, (unsigned*)dmsv.p,
                             (size_t*)dres.p, (size_t*)dpr.p, nullptr));
        HK(hipDeviceSynchronize());
    } else {
        HistArgs a;
        a.counts = (unsigned*)dcnt.p; a.maxSVs = (unsigned*)dmsv.p; a.uniformMaxSV = 255; a.useUniformIn = 0; a.trustInput = trustInput;
        a.results = (size_t*)dres.p; a.src = mkview(dsrc.p, srcSize, nullptr, srcSize); a.nBlocks = 1;
        HK(launch_hist(a, nullptr));
    }
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (FSEHIP_isError(r)) return r;
    const unsigned in = *maxSymbolValuePtr;
    const unsigned nOut = in < 255 ? in + 1 : 256;
    HK(hipMemcpy(count, dcnt.p, nOut * 4, hipMemcpyDeviceToHost));
    HK(hipMemcpy(maxSymbolValuePtr, dmsv.p, 4, hipMemcpyDeviceToHost));
    return r;
}
extern "C" size_t FSEHIP_HIST_count(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize)
{
    return hist_count_host(count, maxSymbolValuePtr, src, srcSize, 0);
}
// lib/hist.h:46 (lib/hist.c:163-173): the workspace is validated exactly as the reference validates it and then left alone -- the counting
// happens in the kernel's LDS
extern "C" size_t FSEHIP_HIST_count_wksp(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize, void* workSpace, size_t workSpaceSize)
{
    if ((size_t)workSpace & 3) return FSEHIP_ERROR(GENERIC);
    if (workSpaceSize < FSEHIP_HIST_WKSP_SIZE) return FSEHIP_ERROR(workSpace_tooSmall);
    return hist_count_host(count, maxSymbolValuePtr, src, srcSize, 0);
}
// lib/hist.h:54 (lib/hist.c:141-159): the unchecked variant.  A limit below 255 bounds the entries written to count[] but a larger symbol in
// src is not an error: the result and *maxSymbolValuePtr are taken over all 256 symbols (HIST_count_parallel_wksp with trustInput, :120-131).
// Below 1500 bytes the reference runs HIST_count_simple, which writes beyond count[] for such input; the defined behaviour is kept at every size.
extern "C" size_t FSEHIP_HIST_countFast(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize)
{
    return hist_count_host(count, maxSymbolValuePtr, src, srcSize, 1);
}
// lib/hist.h:62 (lib/hist.c:141-150): below 1500 bytes the reference takes HIST_count_simple and never looks at the workspace; from there on
// the workspace is checked like HIST_count_wksp's
extern "C" size_t FSEHIP_HIST_countFast_wksp(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize, void* workSpace, size_t workSpaceSize)
{
    if (srcSize >= 1500) {
        if ((size_t)workSpace & 3) return FSEHIP_ERROR(GENERIC);
        if (workSpaceSize < FSEHIP_HIST_WKSP_SIZE) return FSEHIP_ERROR(workSpace_tooSmall);
    }
    return hist_count_host(count, maxSymbolValuePtr, src, srcSize, 1);
}
// lib/hist.h:74 (lib/hist.c:29-54): the unchecked loop; returns the largest count as `unsigned`.  A symbol above the limit makes the reference write
// beyond count[]; here it is counted into the result and *maxSymbolValuePtr like HIST_countFast does (defined behaviour at every size).  A device
// failure reads as 0 (the function has no error channel).
extern "C" unsigned FSEHIP_HIST_count_simple(unsigned* count, unsigned* maxSymbolValuePtr, const void* src, size_t srcSize)
{
    const size_t r = hist_count_host(count, maxSymbolValuePtr, src, srcSize, 1);
    return FSEHIP_isError(r) ? 0u : (unsigned)r;
}

extern "C" size_t FSEHIP_FSE_compress_usingCTable(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const FSEHIP_FSE_CTable* ct)
{
    const u16* h = (const u16*)ct;
    const unsigned tl = h[0], msv = h[1];
    if (tl > FSEHIP_FSE_MAX_TABLELOG) return FSEHIP_ERROR(tableLog_tooLarge);
    const size_t words = 1 + (tl ? ((size_t)1 << (tl - 1)) : 1) + 2 * ((size_t)(msv > 255 ? 255 : msv) + 1);   // byte symbols: larger entries are unreachable
    DevBuf dsrc, ddst, dct, dres;
    HK(dsrc.alloc(srcSize)); HK(ddst.alloc(dstCapacity)); HK(dct.alloc(words * 4)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, src, srcSize, hipMemcpyHostToDevice));
    HK(hipMemcpy(dct.p, ct, words * 4, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_compress_usingCTable_batch(ddst.p, dstCapacity, dstCapacity, (size_t*)dres.p, dsrc.p, srcSize, nullptr, srcSize,
                                                         (const unsigned*)dct.p, 0, FSEHIP_FSE_MAX_TABLELOG, 1, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, ddst.p, r, hipMemcpyDeviceToHost));
    return r;
}

extern "C" size_t FSEHIP_FSE_decompress_usingDTable(void* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize, const FSEHIP_FSE_DTable* dt)
{
    const u16* h = (const u16*)dt;
    const unsigned tl = h[0];
    if (tl > FSEHIP_FSE_MAX_TABLELOG) return FSEHIP_ERROR(tableLog_tooLarge);
    const size_t words = 1 + ((size_t)1 << tl);
    DevBuf dsrc, ddst, ddt, dres;
    HK(dsrc.alloc(cSrcSize)); HK(ddst.alloc(dstCapacity)); HK(ddt.alloc(words * 4)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, cSrc, cSrcSize, hipMemcpyHostToDevice));
    HK(hipMemcpy(ddt.p, dt, words * 4, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_decompress_usingDTable_batch(ddst.p, dstCapacity, dstCapacity, (size_t*)dres.p, dsrc.p, cSrcSize, nullptr, cSrcSize,
                                                           (const unsigned*)ddt.p, 0, tl ? tl : 1, 1, nullptr));   // (the table's own log: one launch, its class)
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, ddst.p, r <= dstCapacity ? r : dstCapacity, hipMemcpyDeviceToHost));
    return r;
}

extern "C" size_t FSEHIP_FSE_compress2(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog)
{
    if (tableLog > FSEHIP_FSE_MAX_TABLELOG) return FSEHIP_ERROR(tableLog_tooLarge);   // fse_compress.c:691
    const size_t wsBytes = FSEHIP_FSE_compress_batch_workspaceSize(1, tableLog);
    DevBuf dsrc, ddst, dws, dres;
    HK(dsrc.alloc(srcSize)); HK(ddst.alloc(dstCapacity)); HK(dws.alloc(wsBytes)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, src, srcSize, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_compress_batch(ddst.p, dstCapacity, dstCapacity, (size_t*)dres.p, dsrc.p, srcSize, nullptr, srcSize,
                                             maxSymbolValue, tableLog, 1, dws.p, wsBytes, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 1) HK(hipMemcpy(dst, ddst.p, r, hipMemcpyDeviceToHost));
    return r;
}

extern "C" size_t FSEHIP_FSE_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)   // fse_compress.c:695-698
{
    return FSEHIP_FSE_compress2(dst, dstCapacity, src, srcSize, 255, FSEHIP_FSE_DEFAULT_TABLELOG);
}

extern "C" size_t FSEHIP_FSE_decompress(void* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize)   // fse_decompress.c:279-283
{
    const size_t wsBytes = FSEHIP_FSE_decompress_batch_workspaceSize(1, FSEHIP_FSE_MAX_TABLELOG);
    DevBuf dsrc, ddst, dws, dres;
    HK(dsrc.alloc(cSrcSize)); HK(ddst.alloc(dstCapacity)); HK(dws.alloc(wsBytes)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, cSrc, cSrcSize, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_decompress_batch(ddst.p, dstCapacity, dstCapacity, (size_t*)dres.p, dsrc.p, cSrcSize, nullptr, cSrcSize,
                                               FSEHIP_FSE_MAX_TABLELOG, 1, dws.p, wsBytes, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, ddst.p, r <= dstCapacity ? r : dstCapacity, hipMemcpyDeviceToHost));
    return r;
}

// lib/fse.h:315 (lib/fse_compress.c:632-677).  The workspace is checked as the reference checks it -- its size in BYTES against
// FSE_WKSP_SIZE_U32(tableLog, maxSymbolValue), the comparison of :646 as written, on the arguments as passed (before 0 -> 255 / default) --
// and then left alone: tables and counters live in device memory.  A table log above FSE_MAX_TABLELOG is not refused here (FSE_compress2
// refuses it, :691): FSE_optimalTableLog clamps it to 12 (:340), so it codes like 12.
extern "C" size_t FSEHIP_FSE_compress_wksp(void* dst, size_t dstSize, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog,
                                           void* workSpace, size_t wkspSize)
{
    (void)workSpace;
    // 1 << (tableLog - 1) with tableLog 0 is undefined in the reference's macro (x86: 1 << 31); such a call cannot pass the check
    if (tableLog == 0 || tableLog > 31) return FSEHIP_ERROR(tableLog_tooLarge);
    const unsigned long long need = 1ull + (1ull << (tableLog - 1)) + 2ull * ((unsigned long long)maxSymbolValue + 1) + (tableLog > 12 ? (1ull << (tableLog - 2)) : 1024ull);
    if (wkspSize < need) return FSEHIP_ERROR(tableLog_tooLarge);
    return FSEHIP_FSE_compress2(dst, dstSize, src, srcSize, maxSymbolValue, tableLog > FSEHIP_FSE_MAX_TABLELOG ? FSEHIP_FSE_MAX_TABLELOG : tableLog);
}

// lib/fse.h:335 (lib/fse_decompress.c:255-274): FSE_decompress with the caller's table-log limit.  The reference builds its DTable in
// `workSpace` (FSE_DTABLE_SIZE_U32(maxLog) words); when one is given it receives the same table here (built on the device in the reference's
// layout), so a caller that looks at it afterwards finds what it expects.  Limits above FSE_MAX_TABLELOG count as 12, the library's
// build-time limit (a stream with a larger table log: tableLog_tooLarge).
extern "C" size_t FSEHIP_FSE_decompress_wksp(void* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize, FSEHIP_FSE_DTable* workSpace, unsigned maxLog)
{
    const unsigned ml = maxLog > FSEHIP_FSE_MAX_TABLELOG ? FSEHIP_FSE_MAX_TABLELOG : maxLog;
    if (ml == 0) {      // no table log fits: FSE_readNCount's own errors first, then tableLog_tooLarge (every valid header has tableLog >= 5)
        const size_t wsB = FSEHIP_FSE_buildDTable_batch_workspaceSize(1, FSEHIP_FSE_MAX_TABLELOG);
        DevBuf dsrc, ddt, dws, dres;
        HK(dsrc.alloc(cSrcSize)); HK(ddt.alloc(4 * (size_t)FSEHIP_FSE_DTABLE_SIZE_U32(FSEHIP_FSE_MAX_TABLELOG))); HK(dws.alloc(wsB)); HK(dres.alloc(8));
        HK(hipMemcpy(dsrc.p, cSrc, cSrcSize, hipMemcpyHostToDevice));
        HK((hipError_t)FSEHIP_FSE_buildDTable_batch((unsigned*)ddt.p, FSEHIP_FSE_DTABLE_SIZE_U32(FSEHIP_FSE_MAX_TABLELOG), (size_t*)dres.p, dsrc.p, cSrcSize, nullptr, cSrcSize,
                                                    FSEHIP_FSE_MAX_TABLELOG, 1, dws.p, wsB, nullptr));
        size_t r = 0;
        HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
        return FSEHIP_isError(r) ? r : FSEHIP_ERROR(tableLog_tooLarge);
    }
    const size_t wsBytes = FSEHIP_FSE_decompress_batch_workspaceSize(1, ml);
    DevBuf dsrc, ddst, dws, dres;
    HK(dsrc.alloc(cSrcSize)); HK(ddst.alloc(dstCapacity)); HK(dws.alloc(wsBytes)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, cSrc, cSrcSize, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_decompress_batch(ddst.p, dstCapacity, dstCapacity, (size_t*)dres.p, dsrc.p, cSrcSize, nullptr, cSrcSize, ml, 1, dws.p, wsBytes, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, ddst.p, r <= dstCapacity ? r : dstCapacity, hipMemcpyDeviceToHost));
    if (workSpace) {    // the table the reference leaves in the workspace (whenever the header parsed and its table log fits)
        const size_t wsB = FSEHIP_FSE_buildDTable_batch_workspaceSize(1, ml);
        const size_t dtU32 = FSEHIP_FSE_DTABLE_SIZE_U32(ml);
        DevBuf ddt, dws2, dres2;
        HK(ddt.alloc(4 * dtU32)); HK(dws2.alloc(wsB)); HK(dres2.alloc(8));
        HK((hipError_t)FSEHIP_FSE_buildDTable_batch((unsigned*)ddt.p, dtU32, (size_t*)dres2.p, dsrc.p, cSrcSize, nullptr, cSrcSize, ml, 1, dws2.p, wsB, nullptr));
        size_t hr = 0;
        HK(hipMemcpy(&hr, dres2.p, 8, hipMemcpyDeviceToHost));
        if (!FSEHIP_isError(hr)) {
            u32 h0 = 0;
            HK(hipMemcpy(&h0, ddt.p, 4, hipMemcpyDeviceToHost));
            const unsigned tl = h0 & 0xFFFFu;
            if (tl <= ml) HK(hipMemcpy(workSpace, ddt.p, 4 * ((size_t)1 + ((size_t)1 << tl)), hipMemcpyDeviceToHost));
        }
    }
    return r;
}

// ---- the table glue on host pointers, reference signatures (lib/fse.h:111-163, :222-241): what a caller of the "advanced" flow -- count, normalise,
//      write the header, build the table, code with it -- finds under the reference's names in libfse_dropin.so.  FSE_optimalTableLog and
//      FSE_NCountWriteBound are arithmetic on the arguments (lib/fse_compress.c:186-190, :325-347); the others are batches of one.
extern "C" unsigned FSEHIP_FSE_optimalTableLog(unsigned maxTableLog, size_t srcSize, unsigned maxSymbolValue)
{
    auto hb = [](u32 v) { return 31u - (u32)__builtin_clz(v); };       // lib/bitstream.h:139 (v != 0: srcSize > 1 and maxSymbolValue >= 1 are the reference's preconditions too)
    const u32 maxBitsSrc = hb((u32)(srcSize - 1)) - 2;
    const u32 minBitsSrc = hb((u32)srcSize) + 1, minBitsSymbols = hb(maxSymbolValue) + 2;
    const u32 minBits = minBitsSrc < minBitsSymbols ? minBitsSrc : minBitsSymbols;
    u32 tl = maxTableLog ? maxTableLog : FSEHIP_FSE_DEFAULT_TABLELOG;
    if (maxBitsSrc < tl) tl = maxBitsSrc;
    if (minBits > tl) tl = minBits;
    if (tl < FSEHIP_FSE_MIN_TABLELOG) tl = FSEHIP_FSE_MIN_TABLELOG;
    if (tl > FSEHIP_FSE_MAX_TABLELOG) tl = FSEHIP_FSE_MAX_TABLELOG;
    return tl;
}
extern "C" size_t FSEHIP_FSE_NCountWriteBound(unsigned maxSymbolValue, unsigned tableLog)
{
    return maxSymbolValue ? (size_t)((((maxSymbolValue + 1) * tableLog) >> 3) + 3) : (size_t)FSEHIP_FSE_NCOUNTBOUND;
}
extern "C" size_t FSEHIP_FSE_normalizeCount(short* normalizedCounter, unsigned tableLog, const unsigned* count, size_t total, unsigned maxSymbolValue)
{
    if (maxSymbolValue > 255) return FSEHIP_ERROR(maxSymbolValue_tooLarge);      // (byte alphabets; the reference would index beyond its FSE_MAX_SYMBOL_VALUE-sized users' arrays)
    DevBuf dn, dc, dt, dm, dr;
    HK(dn.alloc(512)); HK(dc.alloc(1024)); HK(dt.alloc(8)); HK(dm.alloc(4)); HK(dr.alloc(8));
    HK(hipMemcpy(dc.p, count, 4 * ((size_t)maxSymbolValue + 1), hipMemcpyHostToDevice));
    HK(hipMemcpy(dt.p, &total, 8, hipMemcpyHostToDevice));
    HK(hipMemcpy(dm.p, &maxSymbolValue, 4, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_normalizeCount_batch((short*)dn.p, 256, tableLog, (const unsigned*)dc.p, 256, (const size_t*)dt.p, (const unsigned*)dm.p, 1, (size_t*)dr.p, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dr.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r)) HK(hipMemcpy(normalizedCounter, dn.p, 2 * ((size_t)maxSymbolValue + 1), hipMemcpyDeviceToHost));
    return r;
}
extern "C" size_t FSEHIP_FSE_writeNCount(void* buffer, size_t bufferSize, const short* normalizedCounter, unsigned maxSymbolValue, unsigned tableLog)
{
    if (tableLog > FSEHIP_FSE_MAX_TABLELOG) return FSEHIP_ERROR(tableLog_tooLarge);   // lib/fse_compress.c:281-282
    if (tableLog < FSEHIP_FSE_MIN_TABLELOG || maxSymbolValue > 255) return FSEHIP_ERROR(GENERIC);
    const size_t cap = bufferSize < 512 ? bufferSize : 512;                      // (no header is longer than FSE_NCOUNTBOUND = 512 bytes)
    DevBuf dh, dn, dm, dr;
    HK(dh.alloc(512)); HK(dn.alloc(512)); HK(dm.alloc(4)); HK(dr.alloc(8));
    HK(hipMemset(dn.p, 0, 512));
    HK(hipMemcpy(dn.p, normalizedCounter, 2 * ((size_t)maxSymbolValue + 1), hipMemcpyHostToDevice));
    HK(hipMemcpy(dm.p, &maxSymbolValue, 4, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_writeNCount_batch(dh.p, 512, cap, (const short*)dn.p, 256, (const unsigned*)dm.p, tableLog, 1, (size_t*)dr.p, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dr.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(buffer, dh.p, r, hipMemcpyDeviceToHost));
    return r;
}
extern "C" size_t FSEHIP_FSE_readNCount(short* normalizedCounter, unsigned* maxSVPtr, unsigned* tableLogPtr, const void* rBuffer, size_t rBuffSize)
{
    const unsigned limit = *maxSVPtr;
    if (limit > 255) return FSEHIP_ERROR(maxSymbolValue_tooLarge);
    const size_t n = rBuffSize < 1024 ? rBuffSize : 1024;                          // (a header describes at most 256 symbols: it ends long before)
    DevBuf dh, dn, dm, dl, dr;
    HK(dh.alloc(n ? n : 1)); HK(dn.alloc(512)); HK(dm.alloc(4)); HK(dl.alloc(4)); HK(dr.alloc(8));
    if (n) HK(hipMemcpy(dh.p, rBuffer, n, hipMemcpyHostToDevice));
    HK(hipMemcpy(dm.p, &limit, 4, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_readNCount_batch((short*)dn.p, 256, (unsigned*)dm.p, (unsigned*)dl.p, dh.p, n, nullptr, n, 1, (size_t*)dr.p, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dr.p, 8, hipMemcpyDeviceToHost));
    if (FSEHIP_isError(r)) return r;
    HK(hipMemcpy(normalizedCounter, dn.p, 2 * ((size_t)limit + 1), hipMemcpyDeviceToHost));   // (the reference clears [0, limit] first: lib/entropy_common.c:68)
    HK(hipMemcpy(maxSVPtr, dm.p, 4, hipMemcpyDeviceToHost));
    HK(hipMemcpy(tableLogPtr, dl.p, 4, hipMemcpyDeviceToHost));
    return r;
}
static size_t fse_norm_to_device(DevBuf& dn, DevBuf& dm, const short* normalizedCounter, unsigned maxSymbolValue)
{
    HK(dn.alloc(512)); HK(dm.alloc(4));
    HK(hipMemset(dn.p, 0, 512));
    HK(hipMemcpy(dn.p, normalizedCounter, 2 * ((size_t)maxSymbolValue + 1), hipMemcpyHostToDevice));
    HK(hipMemcpy(dm.p, &maxSymbolValue, 4, hipMemcpyHostToDevice));
    return 0;
}
extern "C" size_t FSEHIP_FSE_buildCTable(FSEHIP_FSE_CTable* ct, const short* normalizedCounter, unsigned maxSymbolValue, unsigned tableLog)
{
    if (maxSymbolValue > 255) return FSEHIP_ERROR(maxSymbolValue_tooLarge);
    if (tableLog > FSEHIP_FSE_MAX_TABLELOG) return FSEHIP_ERROR(tableLog_tooLarge);   // lib/fse_compress.c:86 with the 4096-byte workspace of :172-176
    if (tableLog == 0 || tableLog == 1 || tableLog == 3) return FSEHIP_ERROR(GENERIC);   // (no table, or an even FSE_TABLESTEP: fsehip.h)
    const size_t words = FSEHIP_FSE_CTABLE_SIZE_U32(tableLog, 255);
    DevBuf dn, dm, dct, dr;
    { const size_t e = fse_norm_to_device(dn, dm, normalizedCounter, maxSymbolValue); if (e) return e; }
    HK(dct.alloc(4 * words)); HK(dr.alloc(8));
    HK((hipError_t)FSEHIP_FSE_buildCTable_fromNorm_batch((unsigned*)dct.p, words, (const short*)dn.p, 256, (const unsigned*)dm.p, tableLog, 1, (size_t*)dr.p, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dr.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r)) HK(hipMemcpy(ct, dct.p, 4 * (size_t)FSEHIP_FSE_CTABLE_SIZE_U32(tableLog, maxSymbolValue), hipMemcpyDeviceToHost));
    return r;
}
// lib/fse.h:341 (lib/fse_compress.c:70-87): the workspace is checked as the reference checks it and then left alone
extern "C" size_t FSEHIP_FSE_buildCTable_wksp(FSEHIP_FSE_CTable* ct, const short* normalizedCounter, unsigned maxSymbolValue, unsigned tableLog, void* workSpace, size_t wkspSize)
{
    (void)workSpace;
    if (tableLog > 31 || ((size_t)1 << tableLog) > wkspSize) return FSEHIP_ERROR(tableLog_tooLarge);
    return FSEHIP_FSE_buildCTable(ct, normalizedCounter, maxSymbolValue, tableLog);
}
extern "C" size_t FSEHIP_FSE_buildDTable(FSEHIP_FSE_DTable* dt, const short* normalizedCounter, unsigned maxSymbolValue, unsigned tableLog)
{
    if (maxSymbolValue > 255) return FSEHIP_ERROR(maxSymbolValue_tooLarge);       // lib/fse_decompress.c:83-84
    if (tableLog > FSEHIP_FSE_MAX_TABLELOG) return FSEHIP_ERROR(tableLog_tooLarge);
    if (tableLog == 0 || tableLog == 1 || tableLog == 3) return FSEHIP_ERROR(GENERIC);   // (no table, or an even FSE_TABLESTEP: fsehip.h)
    const size_t words = FSEHIP_FSE_DTABLE_SIZE_U32(tableLog);
    const size_t wsB = FSEHIP_FSE_buildDTable_fromNorm_batch_workspaceSize(1, tableLog);
    DevBuf dn, dm, ddt, dws, dr;
    { const size_t e = fse_norm_to_device(dn, dm, normalizedCounter, maxSymbolValue); if (e) return e; }
    HK(ddt.alloc(4 * words)); HK(dws.alloc(wsB)); HK(dr.alloc(8));
    HK((hipError_t)FSEHIP_FSE_buildDTable_fromNorm_batch((unsigned*)ddt.p, words, (const short*)dn.p, 256, (const unsigned*)dm.p, tableLog, 1, (size_t*)dr.p, dws.p, wsB, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dr.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r)) HK(hipMemcpy(dt, ddt.p, 4 * words, hipMemcpyDeviceToHost));
    return r;
}

// =====================================================================================================
//  a4 / a5: Huff0 hot loops over a batch
// =====================================================================================================
extern "C" int FSEHIP_HUF_compress4X_usingCTable_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                       const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                                       const FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32,
                                                       size_t nBlocks, void* stream)
{
    HufEncArgs a;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.dstCapacity = dstCapacity; a.results = d_results;
    a.src = mkview(d_src, srcStride, d_sizes, uniformSize);
    a.ctables = d_ctables; a.ctStrideU32 = ctableStrideU32; a.meta = nullptr; a.streams = 4; a.split1X = 0; a.nBlocks = nBlocks;
    return (int)launch_huf_encode(a, (hipStream_t)stream);
}

// HUF_compress1X_usingCTable over a batch (lib/huf.h:290, body lib/huf_compress.c:457-502): one stream per block
extern "C" int FSEHIP_HUF_compress1X_usingCTable_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                                       const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                                       const FSEHIP_HUF_CElt* d_ctables, size_t ctableStrideU32,
                                                       size_t nBlocks, void* stream)
{
    HufEncArgs a;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.dstCapacity = dstCapacity; a.results = d_results;
    a.src = mkview(d_src, srcStride, d_sizes, uniformSize);
    a.ctables = d_ctables; a.ctStrideU32 = ctableStrideU32; a.meta = nullptr; a.streams = 1; a.split1X = 1; a.nBlocks = nBlocks;
    return (int)launch_huf_encode(a, (hipStream_t)stream);
}

extern "C" int FSEHIP_HUF_decompress4X1_usingDTable_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                                          size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                          const FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                          size_t nBlocks, void* stream)
{
    if (maxTableLog == 0 || maxTableLog > FSEHIP_HUF_TABLELOG_MAX) maxTableLog = FSEHIP_HUF_TABLELOG_MAX;
    HufDecArgs a;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.dstSizes = mkview(nullptr, 0, d_dstSizes, uniformDstSize);
    a.results = d_results; a.csrc = mkview(d_cSrc, cStride, d_cSizes, uniformCSize);
    a.dtables = d_dtables; a.dtStrideU32 = dtableStrideU32; a.meta = nullptr;
    a.maxTableLog = maxTableLog; a.G = 0; a.slotU32 = 0; a.streams = 4; a.acceptX2 = 0; a.onlyDeclined = 0; a.classLo = 0; a.nBlocks = nBlocks;
    return (int)launch_huf_decode(a, (hipStream_t)stream);
}

// HUF_decompress4X_usingDTable over a batch (lib/huf_decompress.c:980-997): dispatches per block on the table's type
extern "C" int FSEHIP_HUF_decompress4X_usingDTable_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                                         size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                         const FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                         size_t nBlocks, void* stream)
{
    if (maxTableLog == 0 || maxTableLog > FSEHIP_HUF_TABLELOG_MAX) maxTableLog = FSEHIP_HUF_TABLELOG_MAX;
    HufDecArgs a;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.dstSizes = mkview(nullptr, 0, d_dstSizes, uniformDstSize);
    a.results = d_results; a.csrc = mkview(d_cSrc, cStride, d_cSizes, uniformCSize);
    a.dtables = d_dtables; a.dtStrideU32 = dtableStrideU32; a.meta = nullptr;
    a.maxTableLog = maxTableLog; a.G = 0; a.slotU32 = 0; a.streams = 4; a.acceptX2 = 1; a.onlyDeclined = 0; a.classLo = 0; a.nBlocks = nBlocks;
    return (int)launch_huf_decode(a, (hipStream_t)stream);
}

// HUF_decompress1X1_usingDTable / HUF_decompress1X_usingDTable over a batch (lib/huf.h:318-320; lib/huf_decompress.c:239-260,367-375,961-975):
// one stream per block -- what HUF_compress1X_usingCTable writes
static int huf_1x_dtable_batch(int acceptX2, void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                               size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                               const FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog, size_t nBlocks, void* stream)
{
    if (maxTableLog == 0 || maxTableLog > FSEHIP_HUF_TABLELOG_MAX) maxTableLog = FSEHIP_HUF_TABLELOG_MAX;
    HufDecArgs a;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.dstSizes = mkview(nullptr, 0, d_dstSizes, uniformDstSize);
    a.results = d_results; a.csrc = mkview(d_cSrc, cStride, d_cSizes, uniformCSize);
    a.dtables = d_dtables; a.dtStrideU32 = dtableStrideU32; a.meta = nullptr;
    a.maxTableLog = maxTableLog; a.G = 0; a.slotU32 = 0; a.streams = 1; a.acceptX2 = acceptX2; a.onlyDeclined = 0; a.classLo = 0; a.nBlocks = nBlocks;
    return (int)launch_huf_decode(a, (hipStream_t)stream);
}
extern "C" int FSEHIP_HUF_decompress1X1_usingDTable_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                                          size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                          const FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                          size_t nBlocks, void* stream)
{
    return huf_1x_dtable_batch(0, d_dst, dstStride, d_dstSizes, uniformDstSize, d_results, d_cSrc, cStride, d_cSizes, uniformCSize, d_dtables, dtableStrideU32, maxTableLog, nBlocks, stream);
}
extern "C" int FSEHIP_HUF_decompress1X_usingDTable_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                                         size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                                         const FSEHIP_HUF_DTable* d_dtables, size_t dtableStrideU32, unsigned maxTableLog,
                                                         size_t nBlocks, void* stream)
{
    return huf_1x_dtable_batch(1, d_dst, dstStride, d_dstSizes, uniformDstSize, d_results, d_cSrc, cStride, d_cSizes, uniformCSize, d_dtables, dtableStrideU32, maxTableLog, nBlocks, stream);
}

// =====================================================================================================
//  one-shot Huff0 block API over a batch
// =====================================================================================================
static const size_t HUF_CWS_PER_BLOCK = 1024 + 4 + 8 + sizeof(HufMeta) + 1024 + 4096;
static const size_t HUF_CWS_NODE_PAD = 64 * 4096;   // node scratch is interleaved per workgroup of 64 blocks: the last workgroup needs a whole slab
extern "C" size_t FSEHIP_HUF_compress_batch_workspaceSize(size_t nBlocks)
{
    size_t c = nBlocks < WS_MAX_CHUNK ? nBlocks : WS_MAX_CHUNK;
    if (c == 0) c = 1;
    return c * HUF_CWS_PER_BLOCK + HUF_CWS_NODE_PAD + WS_SLACK;
}

static int huf_compress_impl(int streams, void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                             const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                             unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks,
                             void* d_workspace, size_t workspaceBytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;          // include/fsehip.h: workspaces are 256-byte aligned; checked before anything else
    if (nBlocks == 0) return 0;
    if (tableLog > FSEHIP_HUF_TABLELOG_MAX || maxSymbolValue > 255)               // huf_compress.c:656-660, in the reference's order, per block
        return batch_arg_error(d_results, d_sizes, uniformSize, dstCapacity, nBlocks,
                               tableLog > FSEHIP_HUF_TABLELOG_MAX ? FSEHIP_ERROR(tableLog_tooLarge) : FSEHIP_ERROR(maxSymbolValue_tooLarge), 1, s);
    if (workspaceBytes < HUF_CWS_PER_BLOCK + HUF_CWS_NODE_PAD + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK - HUF_CWS_NODE_PAD) / HUF_CWS_PER_BLOCK;
    if (chunk >= nBlocks) chunk = nBlocks;
    u8* p = (u8*)d_workspace;
    auto carve = [&](size_t bytes) { u8* r = p; p += align_up(bytes, 256); return r; };
    unsigned* counts = (unsigned*)carve(chunk * 1024);
    unsigned* maxSVs = (unsigned*)carve(chunk * 4);
    size_t* hres = (size_t*)carve(chunk * 8);
    HufMeta* meta = (HufMeta*)carve(chunk * sizeof(HufMeta));
    u32* ctables = (u32*)carve(chunk * 1024);
    void* nodes = (void*)p;
    const unsigned msv = maxSymbolValue ? maxSymbolValue : 255;   // huf_compress.c:661
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        const BlockView src = mkview((const u8*)d_src + b0 * srcStride, srcStride, d_sizes ? d_sizes + b0 : nullptr, uniformSize);
        HistArgs h;
        h.counts = counts; h.maxSVs = maxSVs; h.uniformMaxSV = msv; h.useUniformIn = 1; h.results = hres; h.src = src; h.nBlocks = nb;
        CK(launch_hist(h, s));
        HufCPrepArgs c;
        c.counts = counts; c.maxSVs = maxSVs; c.histResults = hres; c.src = src;
        c.dst = (u8*)d_dst + b0 * dstStride; c.dstStride = dstStride; c.dstCapacity = dstCapacity;
        c.maxSVReq = msv; c.huffLogReq = tableLog; c.ctables = ctables; c.ctStrideU32 = 256;
        c.meta = meta; c.results = d_results + b0; c.nBlocks = nb;
        CK(launch_huf_cprep(c, s, nodes));
        HufEncArgs e;
        e.dst = (u8*)d_dst + b0 * dstStride; e.dstStride = dstStride; e.dstCapacity = dstCapacity; e.results = d_results + b0;
        e.src = src; e.ctables = ctables; e.ctStrideU32 = 256; e.meta = meta; e.streams = streams; e.split1X = 0; e.nBlocks = nb;
        CK(launch_huf_encode(e, s));
    }
    return 0;
}

extern "C" int FSEHIP_HUF_compress_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results,
                                         const void* d_src, size_t srcStride, const size_t* d_sizes, size_t uniformSize,
                                         unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks,
                                         void* d_workspace, size_t workspaceBytes, void* stream)
{
    return huf_compress_impl(4, d_dst, dstStride, dstCapacity, d_results, d_src, srcStride, d_sizes, uniformSize, maxSymbolValue, tableLog, nBlocks,
                             d_workspace, workspaceBytes, stream);
}

static const size_t HUF_DWS_PER_BLOCK = sizeof(HufMeta) + 4 * (size_t)FSEHIP_HUF_DTABLE_SIZE_U32(FSEHIP_HUF_TABLELOG_MAX - 1) + HUF_DCLS_COUNT * sizeof(u32);
extern "C" size_t FSEHIP_HUF_decompress_batch_workspaceSize(size_t nBlocks)
{
    size_t c = nBlocks < WS_MAX_CHUNK ? nBlocks : WS_MAX_CHUNK;
    if (c == 0) c = 1;
    return c * HUF_DWS_PER_BLOCK + WS_SLACK;
}

static int huf_decompress_impl(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize, size_t* d_results, const BlockView& csAll,
                               size_t nBlocks, void* d_workspace, size_t workspaceBytes, hipStream_t s)
{
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;          // include/fsehip.h: workspaces are 256-byte aligned; checked before anything else
    if (nBlocks == 0) return 0;
    if (workspaceBytes < HUF_DWS_PER_BLOCK + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / HUF_DWS_PER_BLOCK;
    if (chunk >= nBlocks) chunk = nBlocks;
    u8* p = (u8*)d_workspace;
    const size_t dtU32 = FSEHIP_HUF_DTABLE_SIZE_U32(FSEHIP_HUF_TABLELOG_MAX - 1);      // 2-byte cells: 2^tableLog cells = 2^(tableLog-1) words
    HufMeta* meta = (HufMeta*)p; p += align_up(chunk * sizeof(HufMeta), 256);
    u32* dtables = (u32*)p; p += align_up(chunk * dtU32 * 4, 256);
    u32* lists = (u32*)p; p += align_up(chunk * HUF_DCLS_COUNT * sizeof(u32), 256);
    u32* counts = (u32*)p;
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        const BlockView cs = subview(csAll, b0);
        const BlockView ds = mkview(nullptr, 0, d_dstSizes ? d_dstSizes + b0 : nullptr, uniformDstSize);
        HufDPrepArgs d;
        d.csrc = cs; d.dstSizes = ds; d.dst = (u8*)d_dst + b0 * dstStride; d.dstStride = dstStride;
        d.dtables = dtables; d.dtStrideU32 = dtU32; d.meta = meta; d.lists = lists; d.counts = counts; d.results = d_results + b0; d.nBlocks = nb;
        d.tableOnly = 0; d.dtMaxLog = FSEHIP_HUF_TABLELOG_MAX - 1;
        CK(launch_huf_dprep(d, s));
        HufDecArgs e;
        e.dst = (u8*)d_dst + b0 * dstStride; e.dstStride = dstStride; e.dstSizes = ds; e.results = d_results + b0;
        e.csrc = cs; e.dtables = dtables; e.dtStrideU32 = dtU32; e.meta = meta;
        e.maxTableLog = FSEHIP_HUF_TABLELOG_MAX; e.G = 0; e.slotU32 = 0; e.streams = 4; e.acceptX2 = 0; e.onlyDeclined = 0; e.classLo = 0; e.nBlocks = nb;
        CK(launch_huf_decode_classes(e, lists, counts, s));
    }
    return 0;
}
extern "C" int FSEHIP_HUF_decompress_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize,
                                           size_t* d_results, const void* d_cSrc, size_t cStride, const size_t* d_cSizes, size_t uniformCSize,
                                           size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream)
{
    return huf_decompress_impl(d_dst, dstStride, d_dstSizes, uniformDstSize, d_results, mkview(d_cSrc, cStride, d_cSizes, uniformCSize), nBlocks,
                               d_workspace, workspaceBytes, (hipStream_t)stream);
}
// HUF_decompress over a PACKED batch (FSEHIP_compact_batch): HUF_decompress itself takes a record as long as the block for the block and
// a record of one byte for that byte repeated (lib/huf_decompress.c:1063-1066), which is how the compaction stores what HUF_compress declined
extern "C" int FSEHIP_HUF_decompress_packed_batch(void* d_dst, size_t dstStride, const size_t* d_dstSizes, size_t uniformDstSize, size_t* d_results,
                                                  const void* d_packed, const uint64_t* d_offsets, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream)
{
    BlockView v = mkview(d_packed, 0, nullptr, 0);
    v.offsets = (const u64*)d_offsets;
    return huf_decompress_impl(d_dst, dstStride, d_dstSizes, uniformDstSize, d_results, v, nBlocks, d_workspace, workspaceBytes, (hipStream_t)stream);
}

// =====================================================================================================
//  Packed (variable-length) form of a batch of compressed blocks -- compact.hip
// =====================================================================================================
hipError_t launch_compact(u8* packed, size_t packedCapacity, u64* offsets, const u8* slots, size_t slotStride, const size_t* results, const BlockView& src,
                          size_t nBlocks, u64* partials, hipStream_t s);
extern "C" size_t FSEHIP_compact_batch_workspaceSize(size_t nBlocks) { return ((nBlocks + 1023) / 1024 + 2) * sizeof(u64) + 256; }
extern "C" size_t FSEHIP_compact_batch_bound(size_t nBlocks, size_t blockSize) { return nBlocks * blockSize; }
extern "C" int FSEHIP_compact_batch(void* d_packed, size_t packedCapacity, uint64_t* d_offsets, const void* d_slots, size_t slotStride, const size_t* d_results,
                                    const void* d_src, size_t srcStride, const size_t* d_srcSizes, size_t uniformSrcSize, size_t nBlocks,
                                    void* d_workspace, size_t workspaceBytes, void* stream)
{
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;
    if (workspaceBytes < FSEHIP_compact_batch_workspaceSize(nBlocks)) return (int)hipErrorInvalidValue;
    return (int)launch_compact((u8*)d_packed, packedCapacity, (u64*)d_offsets, (const u8*)d_slots, slotStride, d_results,
                               mkview(d_src, srcStride, d_srcSizes, uniformSrcSize), nBlocks, (u64*)d_workspace, (hipStream_t)stream);
}

// ---- Layer 1, Huff0 ---------------------------------------------------------------------------------
static size_t huf_using_ctable_host(int streams, void* dst, size_t dstSize, const void* src, size_t srcSize, const FSEHIP_HUF_CElt* CTable)
{
    // the opaque HUF_CElt table holds maxSymbolValue+1 entries; only entries of symbols present in src are read
    unsigned maxByte = 0;
    for (size_t i = 0; i < srcSize; i++) { const unsigned v = ((const u8*)src)[i]; if (v > maxByte) maxByte = v; }
    u32 table[256];
    memset(table, 0, sizeof(table));
    memcpy(table, CTable, ((size_t)maxByte + 1) * 4);
    DevBuf dsrc, ddst, dct, dres;
    HK(dsrc.alloc(srcSize)); HK(ddst.alloc(dstSize)); HK(dct.alloc(1024)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, src, srcSize, hipMemcpyHostToDevice));
    HK(hipMemcpy(dct.p, table, 1024, hipMemcpyHostToDevice));
    HufEncArgs a;
    a.dst = (u8*)ddst.p; a.dstStride = dstSize; a.dstCapacity = dstSize; a.results = (size_t*)dres.p;
    a.src = mkview(dsrc.p, srcSize, nullptr, srcSize);
    a.ctables = (const u32*)dct.p; a.ctStrideU32 = 0; a.meta = nullptr; a.streams = streams; a.split1X = 0; a.nBlocks = 1;
    HK(launch_huf_encode(a, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, ddst.p, r, hipMemcpyDeviceToHost));
    return r;
}
extern "C" size_t FSEHIP_HUF_compress1X_usingCTable(void* dst, size_t dstSize, const void* src, size_t srcSize, const FSEHIP_HUF_CElt* CTable)
{
    return huf_using_ctable_host(1, dst, dstSize, src, srcSize, CTable);
}
extern "C" size_t FSEHIP_HUF_compress4X_usingCTable(void* dst, size_t dstSize, const void* src, size_t srcSize, const FSEHIP_HUF_CElt* CTable)
{
    return huf_using_ctable_host(4, dst, dstSize, src, srcSize, CTable);
}

static size_t huf_using_dtable_host(bool acceptX2, void* dst, size_t maxDstSize, const void* cSrc, size_t cSrcSize, const FSEHIP_HUF_DTable* DTable, int streams = 4)
{
    const u32 desc = DTable[0];
    const unsigned type = (desc >> 8) & 0xFF;
    if (type != 0 && !(acceptX2 && type == 1)) return FSEHIP_ERROR(GENERIC);   // huf_decompress.c:411-412
    const unsigned tl = (desc >> 16) & 0xFF;
    if (tl > FSEHIP_HUF_TABLELOG_MAX) return FSEHIP_ERROR(tableLog_tooLarge);
    // single-symbol cells are 2 bytes, double-symbol cells 4 (lib/huf_decompress.c:116, :480)
    const size_t words = 1 + (type ? ((size_t)1 << tl) : (tl ? ((size_t)1 << (tl - 1)) : 1));
    DevBuf dsrc, ddst, ddt, dres;
    HK(dsrc.alloc(cSrcSize)); HK(ddst.alloc(maxDstSize)); HK(ddt.alloc(words * 4)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, cSrc, cSrcSize, hipMemcpyHostToDevice));
    HK(hipMemcpy(ddt.p, DTable, words * 4, hipMemcpyHostToDevice));
    HK((hipError_t)(streams == 1 ? (acceptX2 ? FSEHIP_HUF_decompress1X_usingDTable_batch : FSEHIP_HUF_decompress1X1_usingDTable_batch)
                                 : (acceptX2 ? FSEHIP_HUF_decompress4X_usingDTable_batch : FSEHIP_HUF_decompress4X1_usingDTable_batch))(
        ddst.p, maxDstSize, nullptr, maxDstSize, (size_t*)dres.p, dsrc.p, cSrcSize, nullptr, cSrcSize, (const u32*)ddt.p, 0, FSEHIP_HUF_TABLELOG_MAX, 1, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, ddst.p, r <= maxDstSize ? r : maxDstSize, hipMemcpyDeviceToHost));
    return r;
}
extern "C" size_t FSEHIP_HUF_decompress4X1_usingDTable(void* dst, size_t maxDstSize, const void* cSrc, size_t cSrcSize, const FSEHIP_HUF_DTable* DTable)
{
    return huf_using_dtable_host(false, dst, maxDstSize, cSrc, cSrcSize, DTable);
}
extern "C" size_t FSEHIP_HUF_decompress4X_usingDTable(void* dst, size_t maxDstSize, const void* cSrc, size_t cSrcSize, const FSEHIP_HUF_DTable* DTable)
{
    // lib/huf_decompress.c:980-997 dispatches on tableType: single-symbol (X1) cells -> k_huf_decode, double-symbol (X2) cells -> k_huf_decode_x2
    return huf_using_dtable_host(true, dst, maxDstSize, cSrc, cSrcSize, DTable);
}

extern "C" size_t FSEHIP_HUF_decompress1X1_usingDTable(void* dst, size_t maxDstSize, const void* cSrc, size_t cSrcSize, const FSEHIP_HUF_DTable* DTable)
{
    return huf_using_dtable_host(false, dst, maxDstSize, cSrc, cSrcSize, DTable, 1);
}
extern "C" size_t FSEHIP_HUF_decompress1X_usingDTable(void* dst, size_t maxDstSize, const void* cSrc, size_t cSrcSize, const FSEHIP_HUF_DTable* DTable)
{
    return huf_using_dtable_host(true, dst, maxDstSize, cSrc, cSrcSize, DTable, 1);
}

static size_t huf_compress_host(int streams, void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog)
{
    // argument checks in the reference's order (huf_compress.c:654-660)
    if (!srcSize) return 0;
    if (!dstCapacity) return 0;
    if (srcSize > FSEHIP_HUF_BLOCKSIZE_MAX) return FSEHIP_ERROR(srcSize_wrong);
    if (tableLog > FSEHIP_HUF_TABLELOG_MAX) return FSEHIP_ERROR(tableLog_tooLarge);
    if (maxSymbolValue > 255) return FSEHIP_ERROR(maxSymbolValue_tooLarge);
    const size_t wsBytes = FSEHIP_HUF_compress_batch_workspaceSize(1);
    DevBuf dsrc, ddst, dws, dres;
    HK(dsrc.alloc(srcSize)); HK(ddst.alloc(dstCapacity)); HK(dws.alloc(wsBytes)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, src, srcSize, hipMemcpyHostToDevice));
    HK((hipError_t)huf_compress_impl(streams, ddst.p, dstCapacity, dstCapacity, (size_t*)dres.p, dsrc.p, srcSize, nullptr, srcSize,
                                     maxSymbolValue, tableLog, 1, dws.p, wsBytes, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, ddst.p, r, hipMemcpyDeviceToHost));   // r == 1: the RLE byte sits in dst[0] (:673)
    return r;
}
extern "C" size_t FSEHIP_HUF_compress2(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog)
{
    return huf_compress_host(4, dst, dstCapacity, src, srcSize, maxSymbolValue, tableLog);
}
extern "C" size_t FSEHIP_HUF_compress1X(void* dst, size_t dstSize, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog)   // lib/huf.h:288 (huf_compress.c:750-756)
{
    return huf_compress_host(1, dst, dstSize, src, srcSize, maxSymbolValue, tableLog);
}
// lib/huf.h:95, :289 (lib/huf_compress.c:727-768 -> HUF_compress_internal :637-724): the workspace is validated as :654-655 validate it
// (alignment first, then size) and then left alone; the 1X form writes one stream without a jump table (HUF_singleStream, :615-617)
extern "C" size_t FSEHIP_HUF_compress4X_wksp(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog,
                                             void* workSpace, size_t wkspSize)
{
    if (((size_t)workSpace & 3) != 0) return FSEHIP_ERROR(GENERIC);
    if (wkspSize < FSEHIP_HUF_WORKSPACE_SIZE) return FSEHIP_ERROR(workSpace_tooSmall);
    return huf_compress_host(4, dst, dstCapacity, src, srcSize, maxSymbolValue, tableLog);
}
extern "C" size_t FSEHIP_HUF_compress1X_wksp(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog,
                                             void* workSpace, size_t wkspSize)
{
    if (((size_t)workSpace & 3) != 0) return FSEHIP_ERROR(GENERIC);
    if (wkspSize < FSEHIP_HUF_WORKSPACE_SIZE) return FSEHIP_ERROR(workSpace_tooSmall);
    return huf_compress_host(1, dst, dstCapacity, src, srcSize, maxSymbolValue, tableLog);
}
// lib/huf.h:164 (lib/huf_decompress.c:417-438): HUF_readDTableX1_wksp into the caller's DTable -- whose descriptor carries the table-log limit
// (HUF_CREATE_STATIC_DTABLEX1) and receives {tableType 0, tableLog}, the cells behind it -- then the four streams behind the header.  The
// workspace is checked as :137 checks it ((16 + 64) words) and then left alone.
// HUF_readDTableX1_wksp (lib/huf_decompress.c:118-185) on a block that is in device memory already: dctx (host) receives descriptor and cells as the
// reference leaves them, ddt (device) the same table for a decoder call behind it.  Returns the header size or an error code.
static size_t huf_read_x1_host(FSEHIP_HUF_DTable* dctx, const void* d_src, size_t cSrcSize, DevBuf& ddt)
{
    const u32 desc = dctx[0];
    unsigned mtl = desc & 0xFFu;                                   // DTableDesc.maxTableLog: tables up to mtl + 1 fit (:149)
    if (mtl > FSEHIP_HUF_TABLELOG_MAX - 1) mtl = FSEHIP_HUF_TABLELOG_MAX - 1;      // (HUF_readStats refuses table logs above 12 anyway)
    const unsigned mtlDev = mtl ? mtl : 1;                          // the batch call reads 0 as "default"; a limit of 0 is enforced below
    const size_t dtU32 = 1 + ((size_t)1 << mtlDev);
    const size_t wsB = FSEHIP_HUF_readDTableX1_batch_workspaceSize(1);
    DevBuf dws, dres;
    HK(ddt.alloc(4 * dtU32)); HK(dws.alloc(wsB)); HK(dres.alloc(8));
    HK((hipError_t)FSEHIP_HUF_readDTableX1_batch((u32*)ddt.p, dtU32, mtlDev, (size_t*)dres.p, d_src, cSrcSize, nullptr, cSrcSize, 1, dws.p, wsB, nullptr));
    size_t hSize = 0;
    HK(hipMemcpy(&hSize, dres.p, 8, hipMemcpyDeviceToHost));
    if (FSEHIP_isError(hSize)) return hSize;
    u32 d0 = 0;
    HK(hipMemcpy(&d0, ddt.p, 4, hipMemcpyDeviceToHost));
    const unsigned tl = (d0 >> 16) & 0xFFu;
    if (tl > (desc & 0xFFu) + 1) return FSEHIP_ERROR(tableLog_tooLarge);
    HK(hipMemcpy(dctx + 1, (const u32*)ddt.p + 1, tl ? ((size_t)2 << tl) : 2, hipMemcpyDeviceToHost));
    dctx[0] = (desc & 0xFF0000FFu) | (tl << 16);                    // maxTableLog and the reserved byte stay the caller's (:150-152)
    const u32 dNew = dctx[0];
    HK(hipMemcpy(ddt.p, &dNew, 4, hipMemcpyHostToDevice));
    return hSize;
}
// HUF_decompress4X1_DCtx_wksp / HUF_decompress1X1_DCtx_wksp (lib/huf_decompress.c:377-389, :416-436): the table from the block's header into dctx, then the
// four streams (or the one stream) behind it
static size_t huf_x1_dctx_host(int streams, FSEHIP_HUF_DTable* dctx, void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize)
{
    DevBuf dsrc, ddst, dres, ddt;                                   // (in the order they are carved: ddt inside huf_read_x1_host)
    HK(dsrc.alloc(cSrcSize)); HK(ddst.alloc(dstSize)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, cSrc, cSrcSize, hipMemcpyHostToDevice));
    const size_t hSize = huf_read_x1_host(dctx, dsrc.p, cSrcSize, ddt);
    if (FSEHIP_isError(hSize)) return hSize;
    if (hSize >= cSrcSize) return FSEHIP_ERROR(srcSize_wrong);
    if (streams == 4)
        HK((hipError_t)FSEHIP_HUF_decompress4X1_usingDTable_batch(ddst.p, dstSize, nullptr, dstSize, (size_t*)dres.p, (const u8*)dsrc.p + hSize, cSrcSize - hSize, nullptr, cSrcSize - hSize,
                                                                  (const u32*)ddt.p, 0, FSEHIP_HUF_TABLELOG_MAX, 1, nullptr));
    else
        HK((hipError_t)FSEHIP_HUF_decompress1X1_usingDTable_batch(ddst.p, dstSize, nullptr, dstSize, (size_t*)dres.p, (const u8*)dsrc.p + hSize, cSrcSize - hSize, nullptr, cSrcSize - hSize,
                                                                  (const u32*)ddt.p, 0, FSEHIP_HUF_TABLELOG_MAX, 1, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, ddst.p, r <= dstSize ? r : dstSize, hipMemcpyDeviceToHost));
    return r;
}
extern "C" size_t FSEHIP_HUF_decompress4X1_DCtx_wksp(FSEHIP_HUF_DTable* dctx, void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize,
                                                     void* workSpace, size_t wkspSize)
{
    (void)workSpace;
    if (wkspSize < 4 * (16 + 64)) return FSEHIP_ERROR(tableLog_tooLarge);
    return huf_x1_dctx_host(4, dctx, dst, dstSize, cSrc, cSrcSize);
}
// the rest of the single-symbol family, lib/huf.h:141-143,161-167,209-211,299-304 (lib/huf_decompress.c:118-192, :377-404, :439-452): the forms without a
// workspace are the reference's wrappers around the forms with one; a DTable on the stack where the reference has one (HUF_CREATE_STATIC_DTABLEX1 with
// HUF_TABLELOG_MAX - 1: descriptor 0x0100000B)
extern "C" size_t FSEHIP_HUF_decompress4X1_DCtx(FSEHIP_HUF_DTable* dctx, void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize)
{
    return huf_x1_dctx_host(4, dctx, dst, dstSize, cSrc, cSrcSize);
}
extern "C" size_t FSEHIP_HUF_decompress4X1(void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize)
{
    std::vector<u32> dt(FSEHIP_HUF_DTABLE_SIZE_U32(FSEHIP_HUF_TABLELOG_MAX - 1), 0);
    dt[0] = (u32)(FSEHIP_HUF_TABLELOG_MAX - 1) * 0x01000001u;
    return huf_x1_dctx_host(4, dt.data(), dst, dstSize, cSrc, cSrcSize);
}
extern "C" size_t FSEHIP_HUF_decompress1X1_DCtx_wksp(FSEHIP_HUF_DTable* dctx, void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize,
                                                     void* workSpace, size_t wkspSize)
{
    (void)workSpace;
    if (wkspSize < 4 * (16 + 64)) return FSEHIP_ERROR(tableLog_tooLarge);
    return huf_x1_dctx_host(1, dctx, dst, dstSize, cSrc, cSrcSize);
}
extern "C" size_t FSEHIP_HUF_decompress1X1_DCtx(FSEHIP_HUF_DTable* dctx, void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize)
{
    return huf_x1_dctx_host(1, dctx, dst, dstSize, cSrc, cSrcSize);
}
extern "C" size_t FSEHIP_HUF_decompress1X1(void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize)
{
    std::vector<u32> dt(FSEHIP_HUF_DTABLE_SIZE_U32(FSEHIP_HUF_TABLELOG_MAX - 1), 0);
    dt[0] = (u32)(FSEHIP_HUF_TABLELOG_MAX - 1) * 0x01000001u;
    return huf_x1_dctx_host(1, dt.data(), dst, dstSize, cSrc, cSrcSize);
}
extern "C" size_t FSEHIP_HUF_readDTableX1_wksp(FSEHIP_HUF_DTable* DTable, const void* src, size_t srcSize, void* workSpace, size_t wkspSize)
{
    (void)workSpace;
    if (wkspSize < 4 * (16 + 64)) return FSEHIP_ERROR(tableLog_tooLarge);
    DevBuf dsrc, ddt;
    HK(dsrc.alloc(srcSize));
    HK(hipMemcpy(dsrc.p, src, srcSize, hipMemcpyHostToDevice));
    return huf_read_x1_host(DTable, dsrc.p, srcSize, ddt);
}
extern "C" size_t FSEHIP_HUF_readDTableX1(FSEHIP_HUF_DTable* DTable, const void* src, size_t srcSize)
{
    u32 ws[FSEHIP_HUF_DECOMPRESS_WORKSPACE_SIZE / 4];
    return FSEHIP_HUF_readDTableX1_wksp(DTable, src, srcSize, ws, sizeof(ws));
}
// lib/huf.h:204-218 (lib/huf_compress.c:334-421): HUF_buildCTable[_wksp] on the caller's counters, HUF_writeCTable (lib/huf.h:205, lib/huf_compress.c:113-148) on the
// caller's table -- batches of one on the phases of k_huf_cprep (huf_prep.hip).  The workspace is checked as the reference checks it (:345-348) and left alone.
extern "C" size_t FSEHIP_HUF_buildCTable(FSEHIP_HUF_CElt* tree, const unsigned* count, unsigned maxSymbolValue, unsigned maxNbBits)
{
    if (maxSymbolValue > 255) return FSEHIP_ERROR(maxSymbolValue_tooLarge);       // :350
    DevBuf dc, dm, dct, dr;
    HK(dc.alloc(1024)); HK(dm.alloc(4)); HK(dct.alloc(1024)); HK(dr.alloc(8));
    HK(hipMemset(dc.p, 0, 1024));
    HK(hipMemcpy(dc.p, count, 4 * ((size_t)maxSymbolValue + 1), hipMemcpyHostToDevice));
    HK(hipMemcpy(dm.p, &maxSymbolValue, 4, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_HUF_buildCTable_fromCount_batch((u32*)dct.p, 256, (const unsigned*)dc.p, 256, (const unsigned*)dm.p, maxNbBits, 1, (size_t*)dr.p, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dr.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r)) HK(hipMemcpy(tree, dct.p, 4 * ((size_t)maxSymbolValue + 1), hipMemcpyDeviceToHost));
    return r;
}
extern "C" size_t FSEHIP_HUF_buildCTable_wksp(FSEHIP_HUF_CElt* tree, const unsigned* count, unsigned maxSymbolValue, unsigned maxNbBits, void* workSpace, size_t wkspSize)
{
    if ((size_t)workSpace & 3) return FSEHIP_ERROR(GENERIC);
    if (wkspSize < 4352) return FSEHIP_ERROR(workSpace_tooSmall);                 // sizeof(HUF_buildCTable_wksp_tables): 512 nodes of 8 bytes + 32 rank positions of 8
    return FSEHIP_HUF_buildCTable(tree, count, maxSymbolValue, maxNbBits);
}
extern "C" size_t FSEHIP_HUF_writeCTable(void* dst, size_t maxDstSize, const FSEHIP_HUF_CElt* CTable, unsigned maxSymbolValue, unsigned huffLog)
{
    if (maxSymbolValue > 255) return FSEHIP_ERROR(maxSymbolValue_tooLarge);       // :123
    const size_t cap = maxDstSize < 512 ? maxDstSize : 512;                       // (no header is longer than 1 + 255 bytes)
    DevBuf dh, dct, dm, dr;
    HK(dh.alloc(512)); HK(dct.alloc(1024)); HK(dm.alloc(4)); HK(dr.alloc(8));
    HK(hipMemset(dct.p, 0, 1024));
    HK(hipMemcpy(dct.p, CTable, 4 * ((size_t)maxSymbolValue + 1), hipMemcpyHostToDevice));
    HK(hipMemcpy(dm.p, &maxSymbolValue, 4, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_HUF_writeCTable_batch(dh.p, 512, cap, (const u32*)dct.p, 256, (const unsigned*)dm.p, huffLog, 1, (size_t*)dr.p, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dr.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, dh.p, r, hipMemcpyDeviceToHost));
    return r;
}
extern "C" size_t FSEHIP_HUF_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)   // huf_compress.c:795-798
{
    return FSEHIP_HUF_compress2(dst, dstCapacity, src, srcSize, 255, FSEHIP_HUF_TABLELOG_DEFAULT);
}
extern "C" size_t FSEHIP_HUF_decompress(void* dst, size_t dstSize, const void* cSrc, size_t cSrcSize)   // huf_decompress.c:1056-1081 (4X1 branch)
{
    if (dstSize == 0) return FSEHIP_ERROR(dstSize_tooSmall);
    const size_t wsBytes = FSEHIP_HUF_decompress_batch_workspaceSize(1);
    DevBuf dsrc, ddst, dws, dres;
    HK(dsrc.alloc(cSrcSize)); HK(ddst.alloc(dstSize)); HK(dws.alloc(wsBytes)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, cSrc, cSrcSize, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_HUF_decompress_batch(ddst.p, dstSize, nullptr, dstSize, (size_t*)dres.p, dsrc.p, cSrcSize, nullptr, cSrcSize,
                                               1, dws.p, wsBytes, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 0) HK(hipMemcpy(dst, ddst.p, r <= dstSize ? r : dstSize, hipMemcpyDeviceToHost));
    return r;
}

// =====================================================================================================
//  FSE for 16-bit symbols (lib/fseU16.c)
// =====================================================================================================
static const size_t U16_CWS_PER_BLOCK = ((size_t)2 << FSEHIP_FSEU16_MAX_TABLELOG) + 8 * (FSEHIP_FSEU16_MAX_SYMBOL_VALUE + 1) + sizeof(U16Meta);
static const size_t U16_DWS_PER_BLOCK = ((size_t)4 << FSEHIP_FSEU16_MAX_TABLELOG) + sizeof(U16Meta);
extern "C" size_t FSEHIP_FSE_compressU16_batch_workspaceSize(size_t nBlocks)
{
    size_t c = nBlocks < WS_MAX_CHUNK ? nBlocks : WS_MAX_CHUNK;
    return (c ? c : 1) * U16_CWS_PER_BLOCK + WS_SLACK;
}
extern "C" size_t FSEHIP_FSE_decompressU16_batch_workspaceSize(size_t nBlocks)
{
    size_t c = nBlocks < WS_MAX_CHUNK ? nBlocks : WS_MAX_CHUNK;
    return (c ? c : 1) * U16_DWS_PER_BLOCK + WS_SLACK;
}

extern "C" int FSEHIP_FSE_countU16_batch(unsigned* d_counts, unsigned* d_maxSymbolValues, size_t* d_results, const unsigned short* d_src, size_t srcStrideBytes,
                                         const size_t* d_srcSizes, size_t uniformSrcSize, unsigned maxSymbolValue, size_t nBlocks, void* stream)
{
    if (nBlocks == 0) return 0;
    U16CArgs a;
    a.src = d_src; a.srcStrideBytes = srcStrideBytes; a.srcSizes = d_srcSizes; a.uniformSrcSize = uniformSrcSize;
    a.dst = nullptr; a.dstStride = 0; a.dstCapacity = 0; a.maxSVReq = maxSymbolValue; a.tableLogReq = 0;
    a.stateTables = nullptr; a.symTT = nullptr; a.meta = nullptr; a.countsOut = d_counts; a.maxSVOut = d_maxSymbolValues;
    a.results = d_results; a.nBlocks = nBlocks;
    return (int)launch_u16_compress(a, (hipStream_t)stream);
}

extern "C" int FSEHIP_FSE_compressU16_batch(void* d_dst, size_t dstStride, size_t dstCapacity, size_t* d_results, const unsigned short* d_src, size_t srcStrideBytes,
                                            const size_t* d_srcSizes, size_t uniformSrcSize, unsigned maxSymbolValue, unsigned tableLog, size_t nBlocks,
                                            void* d_workspace, size_t workspaceBytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;          // include/fsehip.h: workspaces are 256-byte aligned; checked before anything else
    if (nBlocks == 0) return 0;
    if (workspaceBytes < U16_CWS_PER_BLOCK + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / U16_CWS_PER_BLOCK;
    if (chunk >= nBlocks) chunk = nBlocks;
    u8* p = (u8*)d_workspace;
    u16* stateTables = (u16*)p; p += align_up(chunk * ((size_t)2 << FSEHIP_FSEU16_MAX_TABLELOG), 256);
    u32* symTT = (u32*)p; p += align_up(chunk * 8 * (FSEHIP_FSEU16_MAX_SYMBOL_VALUE + 1), 256);
    U16Meta* meta = (U16Meta*)p;
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        U16CArgs a;
        a.src = (const u16*)((const u8*)d_src + b0 * srcStrideBytes); a.srcStrideBytes = srcStrideBytes;
        a.srcSizes = d_srcSizes ? d_srcSizes + b0 : nullptr; a.uniformSrcSize = uniformSrcSize;
        a.dst = (u8*)d_dst + b0 * dstStride; a.dstStride = dstStride; a.dstCapacity = dstCapacity;
        a.maxSVReq = maxSymbolValue; a.tableLogReq = tableLog;
        a.stateTables = stateTables; a.symTT = symTT; a.meta = meta; a.countsOut = nullptr; a.maxSVOut = nullptr;
        a.results = d_results + b0; a.nBlocks = nb;
        CK(launch_u16_compress(a, s));
    }
    return 0;
}

extern "C" int FSEHIP_FSE_decompressU16_batch(unsigned short* d_dst, size_t dstStrideBytes, size_t dstCapacity, size_t* d_results, const void* d_cSrc, size_t cStride,
                                              const size_t* d_cSizes, size_t uniformCSize, size_t nBlocks, void* d_workspace, size_t workspaceBytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if ((uintptr_t)d_workspace & 255u) return (int)hipErrorInvalidValue;          // include/fsehip.h: workspaces are 256-byte aligned; checked before anything else
    if (nBlocks == 0) return 0;
    if (workspaceBytes < U16_DWS_PER_BLOCK + WS_SLACK) return (int)hipErrorInvalidValue;
    size_t chunk = (workspaceBytes - WS_SLACK) / U16_DWS_PER_BLOCK;
    if (chunk >= nBlocks) chunk = nBlocks;
    u8* p = (u8*)d_workspace;
    u32* cells = (u32*)p; p += align_up(chunk * ((size_t)4 << FSEHIP_FSEU16_MAX_TABLELOG), 256);
    U16Meta* meta = (U16Meta*)p;
    for (size_t b0 = 0; b0 < nBlocks; b0 += chunk) {
        const size_t nb = (nBlocks - b0) < chunk ? (nBlocks - b0) : chunk;
        U16DArgs a;
        a.dst = (u16*)((u8*)d_dst + b0 * dstStrideBytes); a.dstStrideBytes = dstStrideBytes; a.dstCapacity = dstCapacity;
        a.csrc = (const u8*)d_cSrc + b0 * cStride; a.cStride = cStride; a.cSizes = d_cSizes ? d_cSizes + b0 : nullptr; a.uniformCSize = uniformCSize;
        a.cells = cells; a.meta = meta; a.results = d_results + b0; a.nBlocks = nb;
        CK(launch_u16_decompress(a, s));
    }
    return 0;
}

extern "C" size_t FSEHIP_FSE_countU16(unsigned* count, unsigned* maxSymbolValuePtr, const unsigned short* src, size_t srcSize)
{
    const unsigned in = *maxSymbolValuePtr;
    if (in > FSEHIP_FSEU16_MAX_SYMBOL_VALUE) return FSEHIP_ERROR(maxSymbolValue_tooLarge);
    DevBuf dsrc, dcnt, dmsv, dres;
    HK(dsrc.alloc(srcSize * 2)); HK(dcnt.alloc(4 * (FSEHIP_FSEU16_MAX_SYMBOL_VALUE + 1))); HK(dmsv.alloc(4)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, src, srcSize * 2, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_countU16_batch((unsigned*)dcnt.p, (unsigned*)dmsv.p, (size_t*)dres.p, (const unsigned short*)dsrc.p, srcSize * 2, nullptr, srcSize, in, 1, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (FSEHIP_isError(r)) return r;
    HK(hipMemcpy(count, dcnt.p, 4 * ((size_t)in + 1), hipMemcpyDeviceToHost));
    HK(hipMemcpy(maxSymbolValuePtr, dmsv.p, 4, hipMemcpyDeviceToHost));
    return r;
}

extern "C" size_t FSEHIP_FSE_compressU16(void* dst, size_t dstCapacity, const unsigned short* src, size_t srcSize, unsigned maxSymbolValue, unsigned tableLog)
{
    const size_t wsBytes = FSEHIP_FSE_compressU16_batch_workspaceSize(1);
    DevBuf dsrc, ddst, dws, dres;
    HK(dsrc.alloc(srcSize * 2)); HK(ddst.alloc(dstCapacity)); HK(dws.alloc(wsBytes)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, src, srcSize * 2, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_compressU16_batch(ddst.p, dstCapacity, dstCapacity, (size_t*)dres.p, (const unsigned short*)dsrc.p, srcSize * 2, nullptr, srcSize,
                                                maxSymbolValue, tableLog, 1, dws.p, wsBytes, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (!FSEHIP_isError(r) && r > 1) HK(hipMemcpy(dst, ddst.p, r <= dstCapacity ? r : dstCapacity, hipMemcpyDeviceToHost));
    return r;
}

extern "C" size_t FSEHIP_FSE_decompressU16(unsigned short* dst, size_t dstCapacity, const void* cSrc, size_t cSrcSize)
{
    const size_t wsBytes = FSEHIP_FSE_decompressU16_batch_workspaceSize(1);
    DevBuf dsrc, ddst, dws, dres;
    HK(dsrc.alloc(cSrcSize)); HK(ddst.alloc(dstCapacity * 2)); HK(dws.alloc(wsBytes)); HK(dres.alloc(8));
    HK(hipMemcpy(dsrc.p, cSrc, cSrcSize, hipMemcpyHostToDevice));
    // the device buffer starts as a copy of the caller's: what the decoder does not write stays what it was, as with the reference
    if (dstCapacity) HK(hipMemcpy(ddst.p, dst, dstCapacity * 2, hipMemcpyHostToDevice));
    HK((hipError_t)FSEHIP_FSE_decompressU16_batch((unsigned short*)ddst.p, dstCapacity * 2, dstCapacity, (size_t*)dres.p, dsrc.p, cSrcSize, nullptr, cSrcSize,
                                                  1, dws.p, wsBytes, nullptr));
    size_t r = 0;
    HK(hipMemcpy(&r, dres.p, 8, hipMemcpyDeviceToHost));
    if (dstCapacity) HK(hipMemcpy(dst, ddst.p, dstCapacity * 2, hipMemcpyDeviceToHost));   // (the reference writes what it decoded before it notices corruption)
    return r;
}
