// frame.hip -- the .fse frame around the block codecs, on host buffers
// (reference: programs/fileio.c:266-285 format, :286-432 writer, :462-626 reader).
//
//   frame = magic (LE32: 0x183E2309 FSE / 0x183E3309 Huff0) | block-size id (1 KB << id, id <= 6) |
//           { block header | block }* | end mark: 3 bytes = type 3 + 22 bits of XXH32(content, seed 0) >> 5
//   block header: byte0 = type << 6 (0 compressed, 1 raw, 2 RLE) | 0x20 if the block regenerates exactly the
//           block size, else 2 bytes regenerated size (big endian) follow; compressed blocks add 2 bytes compressed size.
//
// The blocks are independent, so the writer codes all of them with one batched device call (FSE_compress /
// HUF_compress semantics per block: result 0 -> stored raw, 1 -> RLE, as fileio.c:347-401 does) and assembles the frame on
// the host; the reader parses the headers on the host, decodes every compressed block with one batched device call and
// checks the content checksum.  Host side of the boundary: PCIe and host memory bandwidth bound, not a throughput path.
#include "internal.h"
#include <string.h>
#include <stdlib.h>
#include <thread>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <atomic>

namespace {
typedef HostCallBuf DevMem;            // device scratch from the per-thread arena (internal.h): no hipMalloc / hipFree per frame
#define FK(x) do { if ((x) != hipSuccess) return FSEHIP_ERROR(GENERIC); } while (0)
// Copies between the caller's host buffers and the device.  A lone frame call (s = null stream) uses the plain blocking copies; a worker of
// the batched frame calls has a stream of its own, so that its copies and kernels overlap the other workers' (the host side has to
// see the bytes right after every one of these copies, hence the synchronise).
inline hipError_t cp(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s)
{
    if (!s) return hipMemcpy(dst, src, n, kind);
    const hipError_t e = hipMemcpyAsync(dst, src, n, kind, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
inline hipError_t cp2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s)
{
    if (!s) return hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind);
    const hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
struct HostMem {           // uninitialised host staging (a std::vector would zero-fill hundreds of megabytes)
    u8* p = nullptr;
    bool alloc(size_t n) { p = (u8*)malloc(n ? n : 1); return p != nullptr; }
    ~HostMem() { free(p); }
};
// XXH32 of the content runs on its own host thread while the device codes the blocks
struct Checksum {
    u32 value = 0; std::thread th;
    void start(const u8* p, size_t n);
    u32 get() { if (th.joinable()) th.join(); return value; }
    ~Checksum() { if (th.joinable()) th.join(); }
};

const u32 MAGIC_FSE = 0x183E2309u, MAGIC_HUF = 0x183E3309u;      // fileio.c:121-122
const unsigned MAX_BSID = 6;
enum { BT_COMPRESSED = 0, BT_RAW = 1, BT_RLE = 2, BT_CRC = 3 };  // fileio.c:137

inline u32 rd32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
inline u32 rotl(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
// XXH32, one-shot (public algorithm; the reference streams the same function over the content, fileio.c:303,339,408)
u32 xxh32(const u8* p, size_t len, u32 seed)
{
    const u32 P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const u8* const end = p + len;
    u32 h;
    if (len >= 16) {
        u32 v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const u8* const limit = end - 16;
        do {
            v1 = rotl(v1 + rd32(p) * P2, 13) * P1; v2 = rotl(v2 + rd32(p + 4) * P2, 13) * P1;
            v3 = rotl(v3 + rd32(p + 8) * P2, 13) * P1; v4 = rotl(v4 + rd32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (p <= limit);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else h = seed + P5;
    h += (u32)len;
    while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl(h + (*p) * P5, 11) * P1; ++p; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}
// the same hash fed piece by piece (the decompressor hashes the regenerated content while later blocks are still being decoded)
struct Xxh32Stream {
    u32 v1, v2, v3, v4, seed; unsigned long long total = 0; u8 buf[16]; u32 bufn = 0;
    explicit Xxh32Stream(u32 sd = 0) : seed(sd) { const u32 P1 = 2654435761u, P2 = 2246822519u; v1 = sd + P1 + P2; v2 = sd + P2; v3 = sd; v4 = sd - P1; }
    void stripe(const u8* p) { const u32 P1 = 2654435761u, P2 = 2246822519u;
        v1 = rotl(v1 + rd32(p) * P2, 13) * P1; v2 = rotl(v2 + rd32(p + 4) * P2, 13) * P1; v3 = rotl(v3 + rd32(p + 8) * P2, 13) * P1; v4 = rotl(v4 + rd32(p + 12) * P2, 13) * P1; }
    void update(const u8* p, size_t n)
    {
        total += n;
        if (bufn) { while (n && bufn < 16) { buf[bufn++] = *p++; --n; } if (bufn < 16) return; stripe(buf); bufn = 0; }
        while (n >= 16) { stripe(p); p += 16; n -= 16; }
        while (n) { buf[bufn++] = *p++; --n; }
    }
    u32 digest() const
    {
        const u32 P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
        u32 h = total >= 16 ? rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18) : seed + P5;
        h += (u32)total;
        const u8* p = buf; const u8* const end = buf + bufn;
        while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
        while (p < end) { h = rotl(h + (*p) * P5, 11) * P1; ++p; }
        h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
        return h;
    }
};
void Checksum::start(const u8* p, size_t n)
{
    if (n < ((size_t)1 << 20)) { value = (xxh32(p, n, 0) >> 5) & ((1u << 22) - 1); return; }     // small inputs: not worth a thread
    th = std::thread([this, p, n] { value = (xxh32(p, n, 0) >> 5) & ((1u << 22) - 1); });
}
inline size_t block_size(unsigned id) { return (size_t)1024 << id; }   // fileio.c:219
inline size_t cbound(size_t n) { return FSEHIP_FSE_COMPRESSBOUND(n); }

// one-shot block coder over `n` uniform blocks already on the device
int code_blocks(int codec, void* d_dst, size_t stride, size_t* d_res, const void* d_src, size_t blockBytes, size_t n, DevMem& ws, size_t wsBytes, hipStream_t s)
{
    if (n == 0) return 0;
    if (codec == 1) return FSEHIP_HUF_compress_batch(d_dst, stride, stride, d_res, d_src, blockBytes, nullptr, blockBytes, 255, FSEHIP_HUF_TABLELOG_DEFAULT, n, ws.p, wsBytes, s);
    return FSEHIP_FSE_compress_batch(d_dst, stride, stride, d_res, d_src, blockBytes, nullptr, blockBytes, 255, FSEHIP_FSE_DEFAULT_TABLELOG, n, ws.p, wsBytes, s);
}
}   // namespace

extern "C" size_t FSEHIP_frame_compressBound(size_t srcSize, unsigned blockSizeId)
{
    if (blockSizeId > MAX_BSID) return FSEHIP_ERROR(GENERIC);
    const size_t bs = block_size(blockSizeId);
    return 5 + srcSize + 5 * ((srcSize + bs - 1) / bs) + 3;
}

static size_t frame_compress_impl(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned blockSizeId, int codec, hipStream_t s)
{
    if (blockSizeId > MAX_BSID || (codec != 0 && codec != 1)) return FSEHIP_ERROR(GENERIC);
    if (dstCapacity < FSEHIP_frame_compressBound(srcSize, blockSizeId)) return FSEHIP_ERROR(dstSize_tooSmall);
    u8* const out = (u8*)dst;
    const u8* const in = (const u8*)src;
    const size_t bs = block_size(blockSizeId);
    const size_t nFull = srcSize / bs, tail = srcSize % bs, nBlocks = nFull + (tail ? 1 : 0);
    const size_t stride = (cbound(bs) + 15) & ~(size_t)15;               // FSE_compressBound(inputBlockSize), fileio.c:340
    std::vector<size_t> res(nBlocks);
    HostMem comp;
    size_t pitch = 0;                                                     // bytes kept per block on the host = the largest result
    Checksum crc; crc.start(in, srcSize);
    if (nBlocks) {
        DevMem dsrc, ddst, dres, dws;
        const size_t wsBytes = codec == 1 ? FSEHIP_HUF_compress_batch_workspaceSize(nFull ? nFull : 1)
                                          : FSEHIP_FSE_compress_batch_workspaceSize(nFull ? nFull : 1, FSEHIP_FSE_DEFAULT_TABLELOG);
        FK(dsrc.alloc(srcSize)); FK(ddst.alloc(nBlocks * stride)); FK(dres.alloc(nBlocks * sizeof(size_t))); FK(dws.alloc(wsBytes));
        FK(cp(dsrc.p, in, srcSize, hipMemcpyHostToDevice, s));
        if (code_blocks(codec, ddst.p, stride, (size_t*)dres.p, dsrc.p, bs, nFull, dws, wsBytes, s)) return FSEHIP_ERROR(GENERIC);
        if (tail && code_blocks(codec, (u8*)ddst.p + nFull * stride, stride, (size_t*)dres.p + nFull, (const u8*)dsrc.p + nFull * bs, tail, 1, dws, wsBytes, s))
            return FSEHIP_ERROR(GENERIC);
        FK(cp(res.data(), dres.p, nBlocks * sizeof(size_t), hipMemcpyDeviceToHost, s));
        for (size_t b = 0; b < nBlocks; ++b) if (!FSEHIP_isError(res[b]) && res[b] > pitch) pitch = res[b];
        pitch = (pitch + 15) & ~(size_t)15;
        if (pitch) {
            if (!comp.alloc(nBlocks * pitch)) return FSEHIP_ERROR(GENERIC);
            FK(cp2d(comp.p, pitch, ddst.p, stride, pitch, nBlocks, hipMemcpyDeviceToHost, s));
        }
    }
    size_t o = 0;
    const u32 magic = codec == 1 ? MAGIC_HUF : MAGIC_FSE;
    out[0] = (u8)magic; out[1] = (u8)(magic >> 8); out[2] = (u8)(magic >> 16); out[3] = (u8)(magic >> 24); out[4] = (u8)blockSizeId; o = 5;
    for (size_t b = 0; b < nBlocks; ++b) {
        const size_t inSize = b < nFull ? bs : tail;
        const size_t cSize = res[b];
        const bool full = inSize == bs;
        if (FSEHIP_isError(cSize)) return cSize;                                   // fileio.c:341
        const unsigned bt = cSize == 0 ? BT_RAW : cSize == 1 ? BT_RLE : BT_COMPRESSED;
        if (full) out[o++] = (u8)((bt << 6) + 0x20);
        else { out[o++] = (u8)(bt << 6); out[o++] = (u8)(inSize >> 8); out[o++] = (u8)inSize; }
        if (bt == BT_RAW) { memcpy(out + o, in + b * bs, inSize); o += inSize; }
        else if (bt == BT_RLE) out[o++] = in[b * bs];
        else { out[o++] = (u8)(cSize >> 8); out[o++] = (u8)cSize; memcpy(out + o, comp.p + b * pitch, cSize); o += cSize; }
    }
    const u32 checksum = crc.get();                                               // fileio.c:408-416
    out[o++] = (u8)((checksum >> 16) + (BT_CRC << 6)); out[o++] = (u8)(checksum >> 8); out[o++] = (u8)checksum;
    return o;
}

static size_t frame_decompress_impl(void* dst, size_t dstCapacity, const void* src, size_t srcSize, hipStream_t s)
{
    u8* const out = (u8*)dst;
    const u8* const in = (const u8*)src;
    if (srcSize < 5 + 3) return FSEHIP_ERROR(srcSize_wrong);
    int codec;
    {   const u32 magic = rd32(in);
        if (magic == MAGIC_FSE) codec = 0; else if (magic == MAGIC_HUF) codec = 1; else return FSEHIP_ERROR(GENERIC);   // fileio.c:484-499
    }
    if (in[4] > MAX_BSID) return FSEHIP_ERROR(GENERIC);                           // :502-504
    const size_t bs = block_size(in[4]);

    // ---- pass 1 (host): parse the block headers.  The first structural problem ends the walk with its error; the
    //      blocks before it are still decoded, because a decoding error in an earlier block has precedence (the
    //      reference reads and decodes sequentially).
    struct Blk { unsigned bt; size_t rSize, cSize, at; };
    std::vector<Blk> blocks;
    size_t ip = 5, frameErr = 0;
    u32 savedCrc = 0;
    for (;;) {
        if (ip >= srcSize) { frameErr = FSEHIP_ERROR(srcSize_wrong); break; }
        const unsigned b0 = in[ip++];
        Blk k; k.bt = b0 >> 6; k.rSize = bs; k.cSize = 0; k.at = 0;
        if (k.bt == BT_CRC) {
            if (ip + 2 > srcSize) { frameErr = FSEHIP_ERROR(srcSize_wrong); break; }
            savedCrc = in[ip + 1] + ((u32)in[ip] << 8) + ((u32)(b0 & 0x3F) << 16);
            break;
        }
        if (!(b0 & 0x20)) { if (ip + 2 > srcSize) { frameErr = FSEHIP_ERROR(srcSize_wrong); break; } k.rSize = ((size_t)in[ip] << 8) + in[ip + 1]; ip += 2; }
        if (k.bt == BT_COMPRESSED) { if (ip + 2 > srcSize) { frameErr = FSEHIP_ERROR(srcSize_wrong); break; } k.cSize = ((size_t)in[ip] << 8) + in[ip + 1]; ip += 2; }
        else k.cSize = k.bt == BT_RAW ? k.rSize : 1;
        if (ip + k.cSize > srcSize) { frameErr = FSEHIP_ERROR(srcSize_wrong); break; }
        // the reference tool's buffers hold blockSize bytes (fileio.c:509-510): a larger announced size is outside its contract and
        // would overrun the bs-byte device slots below -- rejected here (the oracle does the same)
        if (k.rSize > bs) { frameErr = FSEHIP_ERROR(corruption_detected); break; }
        k.at = ip; ip += k.cSize;
        blocks.push_back(k);
    }

    // ---- pass 2 (device): every compressed block in one batch per distinct capacity (full blocks together; the
    //      reference passes the announced regenerated size as the capacity, fileio.c:570)
    const size_t nB = blocks.size();
    std::vector<size_t> result(nB, 0);                       // regenerated size or error, per block
    std::vector<size_t> slot(nB, (size_t)-1);                // position in the device batch
    std::vector<size_t> order;                               // compressed blocks, full ones first
    for (size_t b = 0; b < nB; ++b) if (blocks[b].bt == BT_COMPRESSED && blocks[b].rSize == bs) order.push_back(b);
    const size_t nFullC = order.size();
    for (size_t b = 0; b < nB; ++b) if (blocks[b].bt == BT_COMPRESSED && blocks[b].rSize != bs) order.push_back(b);
    const size_t nC = order.size();
    HostMem regen;
    size_t cStride = 16;
    for (size_t i = 0; i < nC; ++i) if (blocks[order[i]].cSize > cStride) cStride = blocks[order[i]].cSize;
    cStride = (cStride + 15) & ~(size_t)15;
    const size_t oStride = bs;
    // common case: every block is a full compressed one -> the regenerated blocks are contiguous and land in dst directly
    const bool direct = nC == nB && nFullC == nB && nB * bs <= dstCapacity;
    // Large frames of nothing but full compressed blocks (what the tool writes for compressible files) are decoded in pieces: while the
    // device decodes and delivers piece k, a helper thread streams XXH32 over the pieces already in `dst` -- the hash is serial by
    // construction (about 6 GB/s on one host core) and would otherwise start when everything else has finished.  Anything unusual
    // (a block that does not regenerate its full size, a structural error) drops back to the one-shot path below, which owns the
    // error semantics.
    if (direct && !frameErr && nB >= 2048) {
        const size_t pieces = nB / 1024 < 16 ? nB / 1024 : 16, per = (nB + pieces - 1) / pieces;
        HostMem stage;
        if (!stage.alloc(per * cStride)) return FSEHIP_ERROR(GENERIC);
        std::vector<size_t> cs(per), rr(per);
        DevMem dc, dcs, dout, dres, dws;
        const size_t wsBytes = codec == 1 ? FSEHIP_HUF_decompress_batch_workspaceSize(per) : FSEHIP_FSE_decompress_batch_workspaceSize(per, FSEHIP_FSE_MAX_TABLELOG);
        FK(dc.alloc(per * cStride)); FK(dcs.alloc(per * 8)); FK(dout.alloc(per * bs)); FK(dres.alloc(per * 8)); FK(dws.alloc(wsBytes));
        std::mutex mu; std::condition_variable cv;
        size_t readyBytes = 0; bool finished = false;
        Xxh32Stream xs(0);
        std::thread hasher([&] {
            size_t hashed = 0;
            for (;;) {
                size_t upTo; bool fin;
                {   std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return readyBytes > hashed || finished; }); upTo = readyBytes; fin = finished; }
                if (upTo > hashed) { xs.update(out + hashed, upTo - hashed); hashed = upTo; }
                else if (fin) return;
            }
        });
        struct Joiner { std::thread& t; std::mutex& m; std::condition_variable& c; bool& f; ~Joiner() { { std::lock_guard<std::mutex> lk(m); f = true; } c.notify_all(); if (t.joinable()) t.join(); } } joiner{ hasher, mu, cv, finished };
        bool regular = true;
        for (size_t b0 = 0; b0 < nB && regular; b0 += per) {
            const size_t n = nB - b0 < per ? nB - b0 : per;
            for (size_t i = 0; i < n; ++i) { const Blk& k = blocks[b0 + i]; cs[i] = k.cSize; memcpy(stage.p + i * cStride, in + k.at, k.cSize); }
            FK(cp(dc.p, stage.p, n * cStride, hipMemcpyHostToDevice, s));
            FK(cp(dcs.p, cs.data(), n * 8, hipMemcpyHostToDevice, s));
            if (codec == 1 ? FSEHIP_HUF_decompress_batch(dout.p, bs, nullptr, bs, (size_t*)dres.p, dc.p, cStride, (const size_t*)dcs.p, 0, n, dws.p, wsBytes, s)
                           : FSEHIP_FSE_decompress_batch(dout.p, bs, bs, (size_t*)dres.p, dc.p, cStride, (const size_t*)dcs.p, 0, FSEHIP_FSE_MAX_TABLELOG, n, dws.p, wsBytes, s))
                return FSEHIP_ERROR(GENERIC);
            FK(cp(rr.data(), dres.p, n * 8, hipMemcpyDeviceToHost, s));
            for (size_t i = 0; i < n && regular; ++i) regular = rr[i] == bs;
            if (!regular) break;
            FK(cp(out + b0 * bs, dout.p, n * bs, hipMemcpyDeviceToHost, s));
            { std::lock_guard<std::mutex> lk(mu); readyBytes = (b0 + n) * bs; }
            cv.notify_all();
        }
        if (regular) {
            { std::lock_guard<std::mutex> lk(mu); finished = true; }
            cv.notify_all();
            hasher.join();
            const u32 calc = (xs.digest() >> 5) & ((1u << 22) - 1);
            if (calc != savedCrc) return FSEHIP_ERROR(corruption_detected);
            return nB * bs;
        }
        // (irregular: the joiner stops the hasher; fall through to the one-shot path)
    }
    if (nC) {
        HostMem stage;
        if (!stage.alloc(nC * cStride)) return FSEHIP_ERROR(GENERIC);
        std::vector<size_t> cs(nC), rs(nC), rr(nC);
        for (size_t i = 0; i < nC; ++i) {
            const Blk& k = blocks[order[i]];
            slot[order[i]] = i; cs[i] = k.cSize; rs[i] = k.rSize;
            memcpy(stage.p + i * cStride, in + k.at, k.cSize);
        }
        DevMem dc, dcs, drs, dout, dres, dws;
        const size_t wsBytes = codec == 1 ? FSEHIP_HUF_decompress_batch_workspaceSize(nC) : FSEHIP_FSE_decompress_batch_workspaceSize(nC, FSEHIP_FSE_MAX_TABLELOG);
        FK(dc.alloc(nC * cStride)); FK(dcs.alloc(nC * 8)); FK(drs.alloc(nC * 8)); FK(dout.alloc(nC * oStride)); FK(dres.alloc(nC * 8)); FK(dws.alloc(wsBytes));
        FK(cp(dc.p, stage.p, nC * cStride, hipMemcpyHostToDevice, s));
        FK(cp(dcs.p, cs.data(), nC * 8, hipMemcpyHostToDevice, s));
        FK(cp(drs.p, rs.data(), nC * 8, hipMemcpyHostToDevice, s));
        if (codec == 1) {
            if (FSEHIP_HUF_decompress_batch(dout.p, oStride, (const size_t*)drs.p, 0, (size_t*)dres.p, dc.p, cStride, (const size_t*)dcs.p, 0, nC, dws.p, wsBytes, s))
                return FSEHIP_ERROR(GENERIC);
        } else {
            if (nFullC && FSEHIP_FSE_decompress_batch(dout.p, oStride, bs, (size_t*)dres.p, dc.p, cStride, (const size_t*)dcs.p, 0, FSEHIP_FSE_MAX_TABLELOG, nFullC, dws.p, wsBytes, s))
                return FSEHIP_ERROR(GENERIC);
            for (size_t i = nFullC; i < nC; ++i)             // blocks with their own announced size (normally only the last one)
                if (FSEHIP_FSE_decompress_batch((u8*)dout.p + i * oStride, oStride, rs[i], (size_t*)dres.p + i, (const u8*)dc.p + i * cStride, cStride,
                                                (const size_t*)dcs.p + i, 0, FSEHIP_FSE_MAX_TABLELOG, 1, dws.p, wsBytes, s))
                    return FSEHIP_ERROR(GENERIC);
        }
        FK(cp(rr.data(), dres.p, nC * 8, hipMemcpyDeviceToHost, s));
        bool allFull = direct;
        for (size_t i = 0; i < nC && allFull; ++i) allFull = rr[i] == bs;
        if (allFull) FK(cp(out, dout.p, nC * oStride, hipMemcpyDeviceToHost, s));
        else {
            if (!regen.alloc(nC * oStride)) return FSEHIP_ERROR(GENERIC);
            FK(cp(regen.p, dout.p, nC * oStride, hipMemcpyDeviceToHost, s));
        }
        for (size_t i = 0; i < nC; ++i) result[order[i]] = rr[i];
    }

    // ---- pass 3 (host): lay the blocks out in order, first error wins
    size_t o = 0;
    for (size_t b = 0; b < nB; ++b) {
        const Blk& k = blocks[b];
        if (o + k.rSize > dstCapacity) return FSEHIP_ERROR(dstSize_tooSmall);
        if (k.bt == BT_COMPRESSED) {
            const size_t r = result[b];
            if (FSEHIP_isError(r)) return r;                                       // fileio.c:571-572
            if (regen.p) memcpy(out + o, regen.p + slot[b] * oStride, r);      // (otherwise already in place)
            o += r;
        } else if (k.bt == BT_RAW) { memcpy(out + o, in + k.at, k.rSize); o += k.rSize; }
        else { memset(out + o, in[k.at], k.rSize); o += k.rSize; }
    }
    if (frameErr) return frameErr;
    const u32 calc = (xxh32(out, o, 0) >> 5) & ((1u << 22) - 1);                  // :604-607
    if (calc != savedCrc) return FSEHIP_ERROR(corruption_detected);
    return o;
}

// The C ABI never lets a C++ exception through (std::bad_alloc from the block vectors, whose sizes come from an untrusted
// frame, or std::system_error from the checksum thread): it becomes the generic error code.
extern "C" size_t FSEHIP_frame_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, unsigned blockSizeId, int codec)
{
    try { return frame_compress_impl(dst, dstCapacity, src, srcSize, blockSizeId, codec, nullptr); }
    catch (...) { return FSEHIP_ERROR(GENERIC); }
}
extern "C" size_t FSEHIP_frame_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    try { return frame_decompress_impl(dst, dstCapacity, src, srcSize, nullptr); }
    catch (...) { return FSEHIP_ERROR(GENERIC); }
}

// ---- many frames per call.  One frame is bound by one host thread's XXH32 (about 6 GB/s) and its pageable copies whatever the device
// does, so frames are handed to a pool of host threads: each worker takes the next frame, runs the single-frame code above on a stream
// of its own (its copies and kernels overlap the other workers') with its own scratch arena, and writes that frame's result.  A
// frame's bytes and result are those of the single-frame call by construction.
namespace {

struct FrameJob {
    bool releaseOnly = false;                                 // no frames: every helper gives its scratch arena back (FSEHIP_releaseScratch)
    bool compress;
    void* const* dsts; const size_t* caps; const void* const* srcs; const size_t* sizes; size_t* results; size_t n;
    unsigned bsid; int codec; int dev;
    std::atomic<size_t> next{ 0 };
    int ranOn = -1;                                           // helpers the pool gave the job to (0: it has none)
    std::atomic<int> releaseErr{ 0 };                         // releaseOnly: the first error a helper met giving its arena back
};
// `s`: the worker's stream (null: the null stream -- the lone caller, exactly the single-frame call)
void frame_worker(FrameJob* j, hipStream_t s)
{
    if (j->releaseOnly) { const int e = release_thread_scratch(); int none = 0; if (e) (void)j->releaseErr.compare_exchange_strong(none, e); return; }
    const bool ok = hipSetDevice(j->dev) == hipSuccess;
    for (;;) {
        const size_t i = j->next.fetch_add(1);
        if (i >= j->n) break;
        size_t r;
        if (!ok || !j->dsts[i] || (!j->srcs[i] && j->sizes[i])) r = FSEHIP_ERROR(GENERIC);
        else {
            try {
                r = j->compress ? frame_compress_impl(j->dsts[i], j->caps[i], j->srcs[i], j->sizes[i], j->bsid, j->codec, s)
                                : frame_decompress_impl(j->dsts[i], j->caps[i], j->srcs[i], j->sizes[i], s);
            } catch (...) { r = FSEHIP_ERROR(GENERIC); }
            if (s) (void)hipStreamSynchronize(s);           // (an early error return may have left kernels behind: the arena is reused next)
        }
        j->results[i] = r;
    }
}

// The helpers of the batched frame calls are PERSISTENT (round 5): a worker keeps its thread, its stream and -- the point -- its
// thread_local scratch arena (capi.hip) from call to call.  Spawned per call, every worker paid a hipMalloc for its arena at its first
// frame and a hipFree at thread exit, and hipFree synchronises the device under the other workers' streams.
// One batch call uses the pool at a time; a second caller arriving meanwhile runs on threads of its own (the former behaviour).  The pool
// object is created on first use and never destroyed: its idle threads sit on a condition variable until the process ends -- no joins in
// static destructors, nothing to hang on at exit.
struct FramePool {
    std::mutex call;                                          // one batch call at a time
    std::mutex m; std::condition_variable work, done;
    std::vector<std::thread> threads;
    FrameJob* job = nullptr; unsigned wanted = 0, active = 0; unsigned long long gen = 0;
    bool stop = false;                                        // FSEHIP_shutdown: the helpers leave their loops (their arenas and streams go with them)
    void loop(unsigned id)
    {
        unsigned long long seen = 0;
        hipStream_t s = nullptr; int sDev = -1;
        for (;;) {
            FrameJob* j;
            {   std::unique_lock<std::mutex> lk(m);
                work.wait(lk, [&] { return stop || gen != seen; });
                if (stop) break;
                seen = gen;
                if (id >= wanted) continue;                      // this call asked for fewer helpers
                j = job;
            }
            if (!j->releaseOnly && sDev != j->dev) {             // first job, or the caller moved to another device: a stream there
                if (s) { (void)hipStreamDestroy(s); s = nullptr; }
                if (hipSetDevice(j->dev) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { s = nullptr; (void)hipGetLastError(); }
                sDev = j->dev;
            }
            frame_worker(j, s);
            {   std::lock_guard<std::mutex> lk(m); if (--active == 0) done.notify_all(); }
        }
        (void)release_thread_scratch();
        if (s) (void)hipStreamDestroy(s);
    }
    // FSEHIP_shutdown: ends and joins the helper threads (false: a batch call is running on them -- nothing was touched).  The pool
    // starts new threads at the next batched frame call.
    bool shutdown()
    {
        std::unique_lock<std::mutex> c(call, std::try_to_lock);
        if (!c.owns_lock()) return false;
        {   std::lock_guard<std::mutex> lk(m); stop = true; }
        work.notify_all();
        for (std::thread& t : threads) if (t.joinable()) t.join();
        threads.clear();
        {   std::lock_guard<std::mutex> lk(m); stop = false; }
        return true;
    }
    // runs `j` on `helpers` pool threads beside the caller; false: the pool is busy (or cannot grow), nothing was started
    bool run(FrameJob& j, unsigned helpers)
    {
        std::unique_lock<std::mutex> c(call, std::try_to_lock);
        if (!c.owns_lock()) return false;
        {   std::lock_guard<std::mutex> lk(m);
            try { while (!j.releaseOnly && threads.size() < helpers) { const unsigned id = (unsigned)threads.size(); threads.emplace_back([this, id] { loop(id); }); } }
            catch (...) { /* fewer helpers than asked for */ }
            if (threads.empty()) { j.ranOn = 0; return !j.releaseOnly ? false : true; }   // (nothing to release: not "busy")
            helpers = helpers < threads.size() ? helpers : (unsigned)threads.size();
            job = &j; wanted = helpers; active = helpers; ++gen;
        }
        work.notify_all();
        if (!j.releaseOnly) {
            hipStream_t mine = nullptr;                         // the caller works too, on a stream of its own beside the helpers'
            if (hipStreamCreateWithFlags(&mine, hipStreamNonBlocking) != hipSuccess) { mine = nullptr; (void)hipGetLastError(); }
            frame_worker(&j, mine);
            if (mine) (void)hipStreamDestroy(mine);
        }
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return active == 0; });
        job = nullptr;
        return true;
    }
};
FramePool& frame_pool() { static FramePool* const p = new FramePool; return *p; }      // (never destroyed: see above)

size_t run_frames(FrameJob& j, unsigned nThreads)
{
    if (!j.n) return 0;
    if (!j.dsts || !j.caps || !j.srcs || !j.sizes || !j.results) return FSEHIP_ERROR(GENERIC);
    if (hipGetDevice(&j.dev) != hipSuccess) return FSEHIP_ERROR(GENERIC);
    unsigned want = nThreads;
    // default pool: 4 workers (each may start a hashing helper of its own).  Measured on 8 frames of 32 MB (64-core host, pageable
    // buffers): 1 / 4 / 8 workers compress at 13.6 / 19.7 / 9.6 GB/s -- beyond four the workers contend for the runtime's pageable-copy
    // staging, not for the device
    if (!want) { want = std::thread::hardware_concurrency() / 2; if (want > 4) want = 4; }
    if (want < 1) want = 1;
    if ((size_t)want > j.n) want = (unsigned)j.n;
    if (want > 64) want = 64;
    if (want == 1) { frame_worker(&j, nullptr); return 0; }    // alone: on the null stream, exactly the single-frame call
    if (frame_pool().run(j, want - 1)) return 0;
    // the pool is serving another caller: threads of this call's own
    std::vector<std::thread> pool;
    auto helper = [&j] {
        hipStream_t s = nullptr;
        if (hipSetDevice(j.dev) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { s = nullptr; (void)hipGetLastError(); }
        frame_worker(&j, s);
        if (s) (void)hipStreamDestroy(s);
    };
    try { for (unsigned t = 1; t < want; ++t) pool.emplace_back(helper); } catch (...) { /* fewer workers than asked for */ }
    helper();
    for (auto& t : pool) t.join();
    return 0;
}
}   // namespace

// FSEHIP_releaseScratch: the idle helpers of the pool give their arenas back too.  Returns 0, the first error a helper met in hipFree, or
// FSEHIP_SCRATCH_BUSY when a batched frame call is running on the pool right now (its helpers' arenas -- up to 64 x 1 GiB -- stay held).
int frame_pool_release_scratch(void)
{
    FramePool& p = frame_pool();
    FrameJob j; j.releaseOnly = true; j.compress = false; j.dsts = nullptr; j.caps = nullptr; j.srcs = nullptr; j.sizes = nullptr; j.results = nullptr;
    j.n = 0; j.bsid = 0; j.codec = 0; j.dev = 0;
    if (!p.run(j, 64)) return FSEHIP_SCRATCH_BUSY;
    return j.releaseErr.load();
}
// FSEHIP_shutdown: see fsehip.h
int frame_pool_shutdown(void) { return frame_pool().shutdown() ? 0 : FSEHIP_SCRATCH_BUSY; }

extern "C" size_t FSEHIP_frame_compress_batch(void* const* dsts, const size_t* dstCapacities, const void* const* srcs, const size_t* srcSizes,
                                              size_t* results, size_t nFrames, unsigned blockSizeId, int codec, unsigned nThreads)
{
    FrameJob j; j.compress = true; j.dsts = dsts; j.caps = dstCapacities; j.srcs = srcs; j.sizes = srcSizes; j.results = results; j.n = nFrames;
    j.bsid = blockSizeId; j.codec = codec; j.dev = 0;
    try { return run_frames(j, nThreads); } catch (...) { return FSEHIP_ERROR(GENERIC); }
}
extern "C" size_t FSEHIP_frame_decompress_batch(void* const* dsts, const size_t* dstCapacities, const void* const* srcs, const size_t* srcSizes,
                                                size_t* results, size_t nFrames, unsigned nThreads)
{
    FrameJob j; j.compress = false; j.dsts = dsts; j.caps = dstCapacities; j.srcs = srcs; j.sizes = srcSizes; j.results = results; j.n = nFrames;
    j.bsid = 0; j.codec = 0; j.dev = 0;
    try { return run_frames(j, nThreads); } catch (...) { return FSEHIP_ERROR(GENERIC); }
}
