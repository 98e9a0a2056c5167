/* shard_rccl.c -- a C host around libfsehip.so for BASELINE config 5 with the corpus on rank 0:
 * one process per GPU; rank 0 scatters the raw blocks, every rank codes its contiguous range with the batched one-shot call and packs
 * the results (FSEHIP_compact_batch), the ranks exchange their packed sizes (ncclAllGather of one uint64), rank 0 gathers the
 * variable-length records and their offsets and decodes the whole packed stream as the check.  Each direction is ONE RCCL group
 * (ncclGroupStart ... ncclGroupEnd), so the root's transfers to / from its peers are in flight together -- one per xGMI link, a star,
 * not a ring.  No collective touches the coding itself: the reference's chunk loop (programs/bench.c:353-364,389-424) has no carried
 * dependence.
 *
 *   make -C examples                      builds examples/shard_rccl (needs libfsehip.so; RCCL and the HIP runtime from /opt/rocm)
 *   examples/shard_rccl [nBlocks]         one rank (the 1-GPU box): the root's shard travels root -> root through RCCL (a send and
 *                                         a receive to oneself inside one group are legal), so every RCCL call of the multi-rank
 *                                         path executes
 *   examples/shard_rccl nBlocks rank world idfile     one of `world` processes; rank 0 writes its ncclUniqueIds to `idfile` / `idfile.2`, the
 *                                         others wait for them (a file on a path all ranks see stands in for MPI_Bcast)
 *   ... [pieces]                          a last argument K > 1 (after nBlocks, or after idfile) also runs the PIPELINED protocol of
 *                                         every shard in K pieces, the scatter of piece k + 1, the coding of piece k
 *                                         and the gather of piece k - 1 in flight together -- three streams, one communicator per
 *                                         direction (RCCL runs the operations of one communicator in order), the compute stream waiting
 *                                         only for "piece k has landed", the host only for "piece k - 1 is packed"
 * Exit code 0 and a line "shard_rccl OK ..." on success.  tests/test_gpu_rccl.py runs the one-rank form on the GPU box. */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "fsehip.h"

#define BLOCK 32768
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 10; } } while (0)
#define NCCLCK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, ncclGetErrorString(r_)); return 11; } } while (0)
#define FSECK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s:%d: libfsehip call failed (%d)\n", __FILE__, __LINE__, r_); return 12; } } while (0)

typedef struct {
    unsigned char* mine;      /* every rank: its shard of raw blocks */
    unsigned char* slots;     /* every rank: fixed-stride compressed slots of its shard (programs/bench.c:514-516) */
    size_t* sizes;            /* every rank: what FSE_compress2 returned per block */
    unsigned char* packed;    /* every rank: its records back to back */
    uint64_t* offsets;        /* every rank: n + 1 record offsets */
    uint64_t* totals;         /* every rank: world packed sizes (device; the all-gather lands here) */
    void* ws; size_t wsBytes; /* workspace of the batched calls */
} RankBufs;

/* scatter -> FSE_compress2 of every block -> pack -> size exchange -> variable-length gather.  On rank 0: d_corpus holds nBlocks x BLOCK
 * bytes, d_allPacked receives every rank's records (rank after rank), d_allOffsets nBlocks + 1 offsets into it (rebased on the host:
 * h_offsets is scratch of nBlocks + 1 entries).  selfTransfers: the root's own shard goes through RCCL too (world 1). */
static int sharded_fse_compress(ncclComm_t comm, hipStream_t stream, int rank, int world, size_t nBlocks, int selfTransfers,
                                const unsigned char* d_corpus, RankBufs* b, unsigned char* d_allPacked, uint64_t* d_allOffsets, uint64_t* h_offsets,
                                uint64_t* h_totals)
{
    const size_t bound = FSEHIP_FSE_COMPRESSBOUND(BLOCK);
    size_t lo, n;
    FSEHIP_shardRange(nBlocks, rank, world, &lo, &n);

    /* ---- scatter: one group, one send per peer */
    NCCLCK(ncclGroupStart());
    if (rank == 0) {
        for (int r = selfTransfers ? 0 : 1; r < world; ++r) {
            size_t rlo, rn;
            FSEHIP_shardRange(nBlocks, r, world, &rlo, &rn);
            if (rn) NCCLCK(ncclSend(d_corpus + rlo * BLOCK, rn * BLOCK, ncclUint8, r, comm, stream));
        }
    }
    if ((rank != 0 || selfTransfers) && n) NCCLCK(ncclRecv(b->mine, n * BLOCK, ncclUint8, 0, comm, stream));
    NCCLCK(ncclGroupEnd());
    const unsigned char* src = (rank == 0 && !selfTransfers) ? d_corpus + lo * BLOCK : b->mine;

    /* ---- the hot path: FSE_compress2 of every block of the shard (no collective), then the records back to back */
    FSECK(FSEHIP_FSE_compress_batch(b->slots, bound, bound, b->sizes, src, BLOCK, NULL, BLOCK, 255, FSEHIP_FSE_DEFAULT_TABLELOG, n, b->ws, b->wsBytes, stream));
    FSECK(FSEHIP_compact_batch(b->packed, FSEHIP_compact_batch_bound(n, BLOCK), b->offsets, b->slots, bound, b->sizes, src, BLOCK, NULL, BLOCK, n,
                               b->ws, b->wsBytes, stream));

    /* ---- size exchange: every rank's packed size on every rank ("ncclAllGather of one uint64") */
    NCCLCK(ncclAllGather(b->offsets + n, b->totals, 1, ncclUint64, comm, stream));
    HIPCK(hipMemcpyAsync(h_totals, b->totals, world * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIPCK(hipStreamSynchronize(stream));          /* the receive sizes have to be known on the host before the gather can be posted */

    /* ---- gather: one group, two receives per peer (records, then offsets -- the same order as the peer's sends) */
    NCCLCK(ncclGroupStart());
    if (rank == 0) {
        uint64_t pos = 0;
        for (int r = 0; r < world; ++r) {
            size_t rlo, rn;
            FSEHIP_shardRange(nBlocks, r, world, &rlo, &rn);
            if (r != 0 || selfTransfers) {
                if (h_totals[r]) NCCLCK(ncclRecv(d_allPacked + pos, h_totals[r], ncclUint8, r, comm, stream));
                if (rn) NCCLCK(ncclRecv(d_allOffsets + rlo, rn, ncclUint64, r, comm, stream));
            }
            pos += h_totals[r];
        }
    }
    if (rank != 0 || selfTransfers) {
        if (h_totals[rank]) NCCLCK(ncclSend(b->packed, h_totals[rank], ncclUint8, 0, comm, stream));
        if (n) NCCLCK(ncclSend(b->offsets, n, ncclUint64, 0, comm, stream));
    }
    NCCLCK(ncclGroupEnd());
    if (rank == 0) {
        if (!selfTransfers) {   /* the root's own shard stays on the device */
            HIPCK(hipMemcpyAsync(d_allPacked, b->packed, h_totals[0], hipMemcpyDeviceToDevice, stream));
            HIPCK(hipMemcpyAsync(d_allOffsets, b->offsets, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
        }
        /* every rank's offsets count from its own first record: rebase them onto the gathered stream (nBlocks + 1 integers: on the host) */
        HIPCK(hipMemcpyAsync(h_offsets, d_allOffsets, nBlocks * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        HIPCK(hipStreamSynchronize(stream));
        uint64_t pos = 0;
        for (int r = 0; r < world; ++r) {
            size_t rlo, rn;
            FSEHIP_shardRange(nBlocks, r, world, &rlo, &rn);
            for (size_t i = 0; i < rn; ++i) h_offsets[rlo + i] += pos;
            pos += h_totals[r];
        }
        h_offsets[nBlocks] = pos;
        HIPCK(hipMemcpyAsync(d_allOffsets, h_offsets, (nBlocks + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    }
    HIPCK(hipStreamSynchronize(stream));
    return 0;
}

/* ---- the same job, pipelined.  Streams: sC = compute (codec + compaction of piece k), sIn = scatter lane, sOut = gather
 * lane; commIn / commOut: a communicator per direction.  Events: landed[k] (sIn: piece k is here), packedEv[k] (sC: piece k is compacted and
 * its packed size is in pinned host memory).  Who waits for what:
 *   sC    waits for landed[k]                       -- its input, nothing else
 *   sOut  waits for packedEv[k - 1]                 -- not for piece k's kernels
 *   host  waits for packedEv[k - 1] after piece k's kernels are queued, then for the gather lane's size exchange (ncclAllGather on sOut)
 * The root's packed stream arrives piece after piece, rank after rank; rowBlock[i] = the corpus block of packed row i (for the check).
 * Piece k of rank r = blocks [pieceLo(r, k), pieceLo(r, k + 1)) of its shard. */
static size_t piece_lo(size_t first, size_t count, int k, int K) { return first + (count * (size_t)k) / (size_t)K; }

static int sharded_fse_compress_pipelined(ncclComm_t commIn, ncclComm_t commOut, hipStream_t sC, hipStream_t sIn, hipStream_t sOut, int rank, int world,
                                          size_t nBlocks, int K, int selfTransfers, const unsigned char* d_corpus, RankBufs* b,
                                          unsigned char* d_allPacked, uint64_t* d_allOffsets, uint64_t* h_offsets, uint64_t* h_pieceTotals /* pinned, K */,
                                          uint64_t* d_totals /* world */, uint64_t* h_totals /* pinned, world */, size_t* rowBlock, uint64_t* packedBytes)
{
    const size_t bound = FSEHIP_FSE_COMPRESSBOUND(BLOCK);
    size_t lo, n;
    FSEHIP_shardRange(nBlocks, rank, world, &lo, &n);
    hipEvent_t landed[64], packedEv[64];
    if (K > 64) K = 64;
    for (int k = 0; k < K; ++k) { HIPCK(hipEventCreateWithFlags(&landed[k], hipEventDisableTiming)); HIPCK(hipEventCreateWithFlags(&packedEv[k], hipEventDisableTiming)); }
    uint64_t pos = 0; size_t row = 0;              /* root: bytes / rows of the gathered stream so far */

#define POST_SCATTER(k) do {                                                                                                      \
        NCCLCK(ncclGroupStart());                                                                                                    \
        if (rank == 0) for (int r = selfTransfers ? 0 : 1; r < world; ++r) {                                                         \
            size_t rlo, rn; FSEHIP_shardRange(nBlocks, r, world, &rlo, &rn);                                                         \
            const size_t p0 = piece_lo(rlo, rn, (k), K), p1 = piece_lo(rlo, rn, (k) + 1, K);                                         \
            if (p1 > p0) NCCLCK(ncclSend(d_corpus + p0 * BLOCK, (p1 - p0) * BLOCK, ncclUint8, r, commIn, sIn));                      \
        }                                                                                                                            \
        if (rank != 0 || selfTransfers) {                                                                                            \
            const size_t p0 = piece_lo(lo, n, (k), K), p1 = piece_lo(lo, n, (k) + 1, K);                                             \
            if (p1 > p0) NCCLCK(ncclRecv(b->mine + (p0 - lo) * BLOCK, (p1 - p0) * BLOCK, ncclUint8, 0, commIn, sIn));                \
        }                                                                                                                            \
        NCCLCK(ncclGroupEnd());                                                                                                      \
        HIPCK(hipEventRecord(landed[k], sIn));                                                                                       \
    } while (0)

    POST_SCATTER(0);
    for (int k = 0; k <= K; ++k) {
        if (k < K) {
            if (k + 1 < K) POST_SCATTER(k + 1);
            /* ---- piece k on the compute stream: waits for its input only */
            HIPCK(hipStreamWaitEvent(sC, landed[k], 0));
            const size_t p0 = piece_lo(lo, n, k, K) - lo, p1 = piece_lo(lo, n, k + 1, K) - lo, m = p1 - p0;
            const unsigned char* src = (rank == 0 && !selfTransfers) ? d_corpus + (lo + p0) * BLOCK : b->mine + p0 * BLOCK;
            uint64_t* const off = b->offsets + p0 + (size_t)k;                      /* m + 1 entries per piece */
            FSECK(FSEHIP_FSE_compress_batch(b->slots + p0 * bound, bound, bound, b->sizes + p0, src, BLOCK, NULL, BLOCK, 255, FSEHIP_FSE_DEFAULT_TABLELOG, m,
                                            b->ws, b->wsBytes, sC));
            FSECK(FSEHIP_compact_batch(b->packed + p0 * BLOCK, m * BLOCK, off, b->slots + p0 * bound, bound, b->sizes + p0, src, BLOCK, NULL, BLOCK, m,
                                       b->ws, b->wsBytes, sC));
            HIPCK(hipMemcpyAsync(&h_pieceTotals[k], off + m, sizeof(uint64_t), hipMemcpyDeviceToHost, sC));
            HIPCK(hipEventRecord(packedEv[k], sC));
        }
        if (k >= 1) {
            /* ---- gather of piece k - 1 on the gather lane */
            const int g = k - 1;
            const size_t p0 = piece_lo(lo, n, g, K) - lo, p1 = piece_lo(lo, n, g + 1, K) - lo, m = p1 - p0;
            const uint64_t* const off = b->offsets + p0 + (size_t)g;
            HIPCK(hipEventSynchronize(packedEv[g]));                               /* the host: piece g is packed (piece k is queued behind it) */
            HIPCK(hipStreamWaitEvent(sOut, packedEv[g], 0));
            NCCLCK(ncclAllGather(off + m, d_totals, 1, ncclUint64, commOut, sOut));    /* every rank's packed size of this piece */
            HIPCK(hipMemcpyAsync(h_totals, d_totals, world * sizeof(uint64_t), hipMemcpyDeviceToHost, sOut));
            HIPCK(hipStreamSynchronize(sOut));                                     /* (the gather lane only: the compute stream runs on) */
            NCCLCK(ncclGroupStart());
            if (rank == 0) {
                for (int r = 0; r < world; ++r) {
                    size_t rlo, rn; FSEHIP_shardRange(nBlocks, r, world, &rlo, &rn);
                    const size_t q0 = piece_lo(rlo, rn, g, K), q1 = piece_lo(rlo, rn, g + 1, K), rm = q1 - q0;
                    if (r != 0 || selfTransfers) {
                        if (h_totals[r]) NCCLCK(ncclRecv(d_allPacked + pos, h_totals[r], ncclUint8, r, commOut, sOut));
                        if (rm) NCCLCK(ncclRecv(d_allOffsets + row, rm, ncclUint64, r, commOut, sOut));
                    } else {                                                        /* the root's own piece stays on the device */
                        HIPCK(hipMemcpyAsync(d_allPacked + pos, b->packed + p0 * BLOCK, h_totals[0], hipMemcpyDeviceToDevice, sOut));
                        HIPCK(hipMemcpyAsync(d_allOffsets + row, off, m * sizeof(uint64_t), hipMemcpyDeviceToDevice, sOut));
                    }
                    for (size_t i = 0; i < rm; ++i) { rowBlock[row + i] = q0 + i; h_offsets[row + i] = pos; }   /* (the base to add to row's offset) */
                    pos += h_totals[r]; row += rm;
                }
            }
            if (rank != 0 || selfTransfers) {
                if (h_pieceTotals[g]) NCCLCK(ncclSend(b->packed + p0 * BLOCK, h_pieceTotals[g], ncclUint8, 0, commOut, sOut));
                if (m) NCCLCK(ncclSend(off, m, ncclUint64, 0, commOut, sOut));
            }
            NCCLCK(ncclGroupEnd());
        }
    }
#undef POST_SCATTER
    HIPCK(hipStreamSynchronize(sIn)); HIPCK(hipStreamSynchronize(sC)); HIPCK(hipStreamSynchronize(sOut));
    if (rank == 0) {
        /* every piece's offsets count from its own first record: add the bytes in front of it (h_offsets holds them per row) */
        uint64_t* tmp = (uint64_t*)malloc(nBlocks * sizeof(uint64_t));
        HIPCK(hipMemcpy(tmp, d_allOffsets, nBlocks * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nBlocks; ++i) h_offsets[i] += tmp[i];
        h_offsets[nBlocks] = pos;
        free(tmp);
        HIPCK(hipMemcpy(d_allOffsets, h_offsets, (nBlocks + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        *packedBytes = pos;
    }
    for (int k = 0; k < K; ++k) { (void)hipEventDestroy(landed[k]); (void)hipEventDestroy(packedEv[k]); }
    return 0;
}

static int get_unique_id(ncclUniqueId* id, int rank, int world, const char* idfile0, const char* suffix)
{
    char idfile[4096];
    snprintf(idfile, sizeof idfile, "%s%s", idfile0 ? idfile0 : "", suffix);
    if (world == 1) return ncclGetUniqueId(id) == ncclSuccess ? 0 : 1;
    if (rank == 0) {
        char tmp[4200];
        if (ncclGetUniqueId(id) != ncclSuccess) return 1;
        snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(id, sizeof *id, 1, f) != 1) return 2;
        fclose(f);
        return rename(tmp, idfile) == 0 ? 0 : 3;      /* appears complete or not at all */
    }
    for (int tries = 0; tries < 6000; ++tries) {       /* up to a minute */
        FILE* f = fopen(idfile, "rb");
        if (f) { const size_t got = fread(id, sizeof *id, 1, f); fclose(f); if (got == 1) return 0; }
        usleep(10000);
    }
    return 4;
}

int main(int argc, char** argv)
{
    const size_t nBlocks = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : 4096;
    const int rank = argc > 4 ? atoi(argv[2]) : 0, world = argc > 4 ? atoi(argv[3]) : 1;
    const char* idfile = argc > 4 ? argv[4] : NULL;
    const int pieces = argc == 3 ? atoi(argv[2]) : argc > 5 ? atoi(argv[5]) : 1;
    if (!nBlocks || world < 1 || rank < 0 || rank >= world || pieces < 1) { fprintf(stderr, "usage: %s [nBlocks [rank world idfile] [pieces]]\n", argv[0]); return 2; }

    int nDev = 0;
    HIPCK(hipGetDeviceCount(&nDev));
    if (nDev < 1) { fprintf(stderr, "no GPU\n"); return 3; }
    HIPCK(hipSetDevice(rank % nDev));
    FSEHIP_DeviceInfo info;
    if (FSEHIP_deviceInfo(&info) != 0) { fprintf(stderr, "libfsehip: no gfx950 device\n"); return 3; }

    ncclUniqueId id, id2;
    if (get_unique_id(&id, rank, world, idfile, "") || get_unique_id(&id2, rank, world, idfile, ".2")) { fprintf(stderr, "rank %d: no ncclUniqueId\n", rank); return 4; }
    ncclComm_t comm, commBack;                                      /* commBack: the pipelined form's communicator of the way back */
    NCCLCK(ncclCommInitRank(&comm, world, id, rank));
    NCCLCK(ncclCommInitRank(&commBack, world, id2, rank));
    hipStream_t stream;
    HIPCK(hipStreamCreate(&stream));

    const size_t bound = FSEHIP_FSE_COMPRESSBOUND(BLOCK);
    size_t lo, n;
    FSEHIP_shardRange(nBlocks, rank, world, &lo, &n);
    const size_t nMax = (nBlocks + world - 1) / world;
    RankBufs b;
    size_t ws = FSEHIP_FSE_compress_batch_workspaceSize(nMax, FSEHIP_FSE_DEFAULT_TABLELOG);
    if (FSEHIP_compact_batch_workspaceSize(nMax) > ws) ws = FSEHIP_compact_batch_workspaceSize(nMax);
    b.wsBytes = ws;
    HIPCK(hipMalloc((void**)&b.mine, nMax * BLOCK)); HIPCK(hipMalloc((void**)&b.slots, nMax * bound)); HIPCK(hipMalloc((void**)&b.sizes, nMax * sizeof(size_t)));
    HIPCK(hipMalloc((void**)&b.packed, nMax * BLOCK)); HIPCK(hipMalloc((void**)&b.offsets, (nMax + 1 + 64) * sizeof(uint64_t)));
    HIPCK(hipMalloc((void**)&b.totals, world * sizeof(uint64_t))); HIPCK(hipMalloc(&b.ws, b.wsBytes));

    unsigned char *d_corpus = NULL, *d_allPacked = NULL, *d_back = NULL;
    uint64_t *d_allOffsets = NULL, *h_offsets = NULL;
    size_t* d_backRes = NULL;
    void* d_dws = NULL;
    size_t dwsBytes = 0;
    uint64_t* h_totals = (uint64_t*)calloc(world, sizeof(uint64_t));
    if (rank == 0) {
        /* the workload of the reference's benchmark, made on the device: block g = probagen(P = 14 %, seed g + 1) */
        uint8_t table[4096];
        FSEHIP_probagen_table(table, 0.14);
        HIPCK(hipMalloc((void**)&d_corpus, nBlocks * BLOCK));
        FSECK(FSEHIP_probagen_batch(d_corpus, BLOCK, BLOCK, nBlocks, table, 1, stream));
        HIPCK(hipMalloc((void**)&d_allPacked, nBlocks * BLOCK)); HIPCK(hipMalloc((void**)&d_allOffsets, (nBlocks + 1) * sizeof(uint64_t)));
        HIPCK(hipMalloc((void**)&d_back, nBlocks * BLOCK)); HIPCK(hipMalloc((void**)&d_backRes, nBlocks * sizeof(size_t)));
        dwsBytes = FSEHIP_FSE_decompress_batch_workspaceSize(nBlocks, FSEHIP_FSE_MAX_TABLELOG);
        HIPCK(hipMalloc(&d_dws, dwsBytes));
        h_offsets = (uint64_t*)malloc((nBlocks + 1) * sizeof(uint64_t));
    }
    HIPCK(hipStreamSynchronize(stream));

    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {           /* the second pass is the timed one (communicator and kernels warm) */
        HIPCK(hipEventRecord(e0, stream));
        const int rc = sharded_fse_compress(comm, stream, rank, world, nBlocks, world == 1, d_corpus, &b, d_allPacked, d_allOffsets, h_offsets, h_totals);
        if (rc) return rc;
        HIPCK(hipEventRecord(e1, stream));
        HIPCK(hipEventSynchronize(e1));
        HIPCK(hipEventElapsedTime(&ms, e0, e1));
    }

    int ok = 1;
    if (rank == 0) {
        /* the check: the gathered packed stream decodes, where it lies, to the corpus */
        FSECK(FSEHIP_FSE_decompress_packed_batch(d_back, BLOCK, BLOCK, d_backRes, d_allPacked, d_allOffsets, NULL, BLOCK, FSEHIP_FSE_MAX_TABLELOG, nBlocks,
                                                 d_dws, dwsBytes, stream));
        HIPCK(hipStreamSynchronize(stream));
        const size_t chunk = 1024;
        unsigned char* ha = (unsigned char*)malloc(chunk * BLOCK); unsigned char* hb = (unsigned char*)malloc(chunk * BLOCK);
        size_t* hr = (size_t*)malloc(nBlocks * sizeof(size_t));
        HIPCK(hipMemcpy(hr, d_backRes, nBlocks * sizeof(size_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nBlocks; ++i) if (hr[i] != BLOCK) { fprintf(stderr, "block %zu: decode returned %zu\n", i, hr[i]); ok = 0; break; }
        for (size_t c = 0; ok && c < nBlocks; c += chunk) {
            const size_t m = nBlocks - c < chunk ? nBlocks - c : chunk;
            HIPCK(hipMemcpy(ha, d_corpus + c * BLOCK, m * BLOCK, hipMemcpyDeviceToHost));
            HIPCK(hipMemcpy(hb, d_back + c * BLOCK, m * BLOCK, hipMemcpyDeviceToHost));
            if (memcmp(ha, hb, m * BLOCK)) { fprintf(stderr, "blocks %zu..%zu differ after the round trip\n", c, c + m); ok = 0; }
        }
        free(ha); free(hb); free(hr);
        uint64_t total = 0;
        for (int r = 0; r < world; ++r) total += h_totals[r];
        if (ok) printf("shard_rccl OK: %zu blocks of %d bytes over %d rank(s) on %s, packed %llu bytes (%.3f of the input), scatter + FSE_compress2 + pack + gather %.2f ms"
                       " (%.1f GB/s of input)%s\n", nBlocks, BLOCK, world, info.archName, (unsigned long long)total, (double)total / ((double)nBlocks * BLOCK), ms,
                       (double)nBlocks * BLOCK / ms / 1e6, world == 1 ? "; the root's shard went root -> root through ncclSend / ncclRecv" : "");
    }
    /* ---- the pipelined protocol over the same corpus, checked the same way (rows of the packed stream in arrival order) */
    if (ok && pieces > 1) {
        hipStream_t sIn, sOut;
        HIPCK(hipStreamCreateWithFlags(&sIn, hipStreamNonBlocking)); HIPCK(hipStreamCreateWithFlags(&sOut, hipStreamNonBlocking));
        uint64_t *h_pieceTotals = NULL, *h_tot2 = NULL;
        HIPCK(hipHostMalloc((void**)&h_pieceTotals, 64 * sizeof(uint64_t), hipHostMallocDefault));
        HIPCK(hipHostMalloc((void**)&h_tot2, world * sizeof(uint64_t), hipHostMallocDefault));
        size_t* rowBlock = rank == 0 ? (size_t*)malloc(nBlocks * sizeof(size_t)) : NULL;
        uint64_t packedBytes = 0;
        float pms = 0;
        for (int pass = 0; pass < 2; ++pass) {
            HIPCK(hipDeviceSynchronize());
            HIPCK(hipEventRecord(e0, stream));
            const int rc = sharded_fse_compress_pipelined(comm, commBack, stream, sIn, sOut, rank, world, nBlocks, pieces, world == 1, d_corpus, &b,
                                                          d_allPacked, d_allOffsets, h_offsets, h_pieceTotals, b.totals, h_tot2, rowBlock, &packedBytes);
            if (rc) return rc;
            HIPCK(hipEventRecord(e1, stream));
            HIPCK(hipEventSynchronize(e1));
            HIPCK(hipEventElapsedTime(&pms, e0, e1));
        }
        if (rank == 0) {
            FSECK(FSEHIP_FSE_decompress_packed_batch(d_back, BLOCK, BLOCK, d_backRes, d_allPacked, d_allOffsets, NULL, BLOCK, FSEHIP_FSE_MAX_TABLELOG, nBlocks,
                                                     d_dws, dwsBytes, stream));
            HIPCK(hipStreamSynchronize(stream));
            unsigned char* ha = (unsigned char*)malloc(BLOCK); unsigned char* hb = (unsigned char*)malloc((size_t)1024 * BLOCK);
            size_t* hr = (size_t*)malloc(nBlocks * sizeof(size_t));
            HIPCK(hipMemcpy(hr, d_backRes, nBlocks * sizeof(size_t), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nBlocks && ok; ++i) if (hr[i] != BLOCK) { fprintf(stderr, "pipelined: row %zu: decode returned %zu\n", i, hr[i]); ok = 0; }
            for (size_t c = 0; ok && c < nBlocks; c += 1024) {
                const size_t mm = nBlocks - c < 1024 ? nBlocks - c : 1024;
                HIPCK(hipMemcpy(hb, d_back + c * BLOCK, mm * BLOCK, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < mm && ok; ++i) {
                    HIPCK(hipMemcpy(ha, d_corpus + rowBlock[c + i] * BLOCK, BLOCK, hipMemcpyDeviceToHost));
                    if (memcmp(ha, hb + i * BLOCK, BLOCK)) { fprintf(stderr, "pipelined: row %zu (block %zu) differs after the round trip\n", c + i, rowBlock[c + i]); ok = 0; }
                }
            }
            free(ha); free(hb); free(hr);
            if (ok) printf("shard_rccl pipelined OK: %d pieces per shard, three streams and a communicator per direction, packed %llu bytes, %.2f ms (%.1f GB/s of input)"
                           " against %.2f ms serial\n", pieces, (unsigned long long)packedBytes, pms, (double)nBlocks * BLOCK / pms / 1e6, ms);
        }
    }
    NCCLCK(ncclCommDestroy(commBack));
    NCCLCK(ncclCommDestroy(comm));
    return ok ? 0 : 1;
}
