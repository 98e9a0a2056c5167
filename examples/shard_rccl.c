/* shard_rccl.c -- a C host around libfsehip.so for BASELINE config 5 with the corpus on rank 0:
 * one process per GPU, rank 0 scatters the raw blocks, every rank codes its contiguous range with the batched one-shot calls,
 * rank 0 gathers the fixed-stride compressed slots and sizes.  Each direction is ONE RCCL group, so the root's transfers to / from
 * its peers are in flight together (one per xGMI link: a star, not a ring).  The rank / world / ncclUniqueId exchange is the
 * launcher's business (MPI, or a file on a shared path) and left out.
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include examples/shard_rccl.c -c   (link: -L finitestateentropy_amd/csrc -lfsehip -L/opt/rocm/lib -lrccl -lamdhip64)
 * Compiled by tests/test_host_api.py (no GPU needed); runs only where several GPUs are. */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stddef.h>
#include "fsehip.h"

#define BLOCK 32768

/* returns 0 on success; d_corpus / d_allSlots / d_allSizes are used on rank 0 only */
int sharded_fse_compress(ncclComm_t comm, hipStream_t stream, int rank, int world, size_t nBlocks,
                         const unsigned char* d_corpus,           /* rank 0: nBlocks x BLOCK bytes */
                         unsigned char* d_mine,                   /* every rank: room for its shard of raw blocks */
                         unsigned char* d_slots, size_t* d_sizes, /* every rank: its shard's compressed slots (stride = bound) and results */
                         unsigned char* d_allSlots, size_t* d_allSizes,   /* rank 0: nBlocks slots and results */
                         void* d_ws, size_t wsBytes)
{
    const size_t bound = FSEHIP_FSE_COMPRESSBOUND(BLOCK);
    size_t lo, n;
    FSEHIP_shardRange(nBlocks, rank, world, &lo, &n);

    /* ---- scatter: one group, one send per peer */
    ncclGroupStart();
    if (rank == 0) {
        for (int r = 1; r < world; ++r) {
            size_t rlo, rn;
            FSEHIP_shardRange(nBlocks, r, world, &rlo, &rn);
            if (rn) ncclSend(d_corpus + rlo * BLOCK, rn * BLOCK, ncclUint8, r, comm, stream);
        }
    } else if (n) ncclRecv(d_mine, n * BLOCK, ncclUint8, 0, comm, stream);
    ncclGroupEnd();
    const unsigned char* src = rank == 0 ? d_corpus + lo * BLOCK : d_mine;

    /* ---- the hot path: FSE_compress2 of every block of the shard (no collective) */
    if (FSEHIP_FSE_compress_batch(d_slots, bound, bound, d_sizes, src, BLOCK, NULL, BLOCK, 255, FSEHIP_FSE_DEFAULT_TABLELOG, n, d_ws, wsBytes, stream)) return 1;

    /* ---- gather: one group, two receives per peer (slots, then sizes -- the same order as the peer's sends) */
    ncclGroupStart();
    if (rank == 0) {
        for (int r = 1; r < world; ++r) {
            size_t rlo, rn;
            FSEHIP_shardRange(nBlocks, r, world, &rlo, &rn);
            if (!rn) continue;
            ncclRecv(d_allSlots + rlo * bound, rn * bound, ncclUint8, r, comm, stream);
            ncclRecv(d_allSizes + rlo, rn * sizeof(size_t), ncclUint8, r, comm, stream);
        }
    } else if (n) {
        ncclSend(d_slots, n * bound, ncclUint8, 0, comm, stream);
        ncclSend(d_sizes, n * sizeof(size_t), ncclUint8, 0, comm, stream);
    }
    ncclGroupEnd();
    if (rank == 0) {   /* the root's own shard stays on the device */
        if (hipMemcpyAsync(d_allSlots + lo * bound, d_slots, n * bound, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 2;
        if (hipMemcpyAsync(d_allSizes + lo, d_sizes, n * sizeof(size_t), hipMemcpyDeviceToDevice, stream) != hipSuccess) return 2;
    }
    return hipStreamSynchronize(stream) == hipSuccess ? 0 : 3;
}
