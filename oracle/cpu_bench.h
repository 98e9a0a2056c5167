/* oracle/cpu_bench.h -- TEST INFRASTRUCTURE ONLY: the multi-core CPU baseline loop of bench.py's `cpu_baseline` leg.
 *
 * Included by ref_shim.c (codec = the unmodified reference, kind "reference") and by fse_oracle.c (codec = our restatement,
 * kind "port") after they define
 *     CPUB_NAME                      name of the exported function
 *     CPUB_COMPRESS(codec, d, cap, s, n, msv, tl)   -> size_t
 *     CPUB_DECOMPRESS(codec, d, n, s, cs)           -> size_t
 * It is the data-parallel axis of programs/bench.c:353-364,389-424 (one block per call) spread over the host cores with
 * OpenMP.  What makes the number honest on a many-core, multi-socket host:
 *   - the working buffers are allocated here and FIRST TOUCHED inside the parallel region with the same static
 *     distribution the timed loops use, so every thread works on pages of its own NUMA node;
 *   - the thread pool is warmed with the full thread count on the full sample (one untimed round trip, which is also
 *     the correctness check);
 *   - each direction is timed over whole passes until at least `minSeconds` have elapsed, best of `reps`.
 * out[0] = encode seconds per pass (best), out[1] = decode seconds per pass (best), out[2] = threads used,
 * out[3] = passes per timed repetition (encode), out[4] = same (decode), out[5] = mean compressed size.
 * Returns 0 on success, -1 allocation failure, -2 round-trip mismatch. */
#include <stdlib.h>
#include <string.h>

int CPUB_NAME(int codec, const uint8_t* sample, size_t nBlocks, size_t blockSize, size_t dstCapacity,
              unsigned maxSymbolValue, unsigned tableLog, int nthreads, int dynamicSchedule, double minSeconds, int reps, double* out)
{
    const size_t cStride = (dstCapacity + 63) & ~(size_t)63;
    uint8_t* const src = (uint8_t*)malloc(nBlocks * blockSize + 64);
    uint8_t* const comp = (uint8_t*)malloc(nBlocks * cStride + 64);
    uint8_t* const back = (uint8_t*)malloc(nBlocks * blockSize + 64);
    uint64_t* const cs = (uint64_t*)malloc(nBlocks * sizeof(uint64_t));
    long b;
    int bad = 0, threads = 1, rep;
    double bestE = 1e30, bestD = 1e30, passesE = 0, passesD = 0, sum = 0;
    if (!src || !comp || !back || !cs) { free(src); free(comp); free(back); free(cs); return -1; }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
    {
#pragma omp single
        threads = omp_get_num_threads();
    }
#else
    (void)nthreads;
#endif
    /* first touch + warm-up round trip (untimed), same distribution as the timed loops */
#pragma omp parallel for schedule(static, 16)
    for (b = 0; b < (long)nBlocks; b++) {
        memcpy(src + (size_t)b * blockSize, sample + (size_t)b * blockSize, blockSize);
        memset(comp + (size_t)b * cStride, 0, cStride);
        memset(back + (size_t)b * blockSize, 0, blockSize);
    }
#pragma omp parallel for schedule(static, 16) reduction(|:bad)
    for (b = 0; b < (long)nBlocks; b++) {
        const size_t c = CPUB_COMPRESS(codec, comp + (size_t)b * cStride, dstCapacity, src + (size_t)b * blockSize, blockSize, maxSymbolValue, tableLog);
        cs[b] = (uint64_t)c;
        if (c > 1 && c <= dstCapacity) {
            const size_t r = CPUB_DECOMPRESS(codec, back + (size_t)b * blockSize, blockSize, comp + (size_t)b * cStride, c);
            if (r != blockSize || memcmp(back + (size_t)b * blockSize, src + (size_t)b * blockSize, blockSize)) bad |= 1;
        } else bad |= 2;                 /* the baseline workload is compressible data: raw / RLE results are not expected */
    }
    for (b = 0; b < (long)nBlocks; b++) sum += (double)cs[b];
    if (bad) { free(src); free(comp); free(back); free(cs); return -2; }

    for (rep = 0; rep < reps; rep++) {
        double t0 = now_s(), t1; int passes = 0;
        do {
            if (dynamicSchedule) {
#pragma omp parallel for schedule(dynamic, 64)
                for (b = 0; b < (long)nBlocks; b++)
                    cs[b] = (uint64_t)CPUB_COMPRESS(codec, comp + (size_t)b * cStride, dstCapacity, src + (size_t)b * blockSize, blockSize, maxSymbolValue, tableLog);
            } else {
#pragma omp parallel for schedule(static, 16)
                for (b = 0; b < (long)nBlocks; b++)
                    cs[b] = (uint64_t)CPUB_COMPRESS(codec, comp + (size_t)b * cStride, dstCapacity, src + (size_t)b * blockSize, blockSize, maxSymbolValue, tableLog);
            }
            passes++; t1 = now_s();
        } while (t1 - t0 < minSeconds);
        if ((t1 - t0) / passes < bestE) { bestE = (t1 - t0) / passes; passesE = passes; }
        t0 = now_s(); passes = 0;
        do {
            if (dynamicSchedule) {
#pragma omp parallel for schedule(dynamic, 64)
                for (b = 0; b < (long)nBlocks; b++)
                    (void)CPUB_DECOMPRESS(codec, back + (size_t)b * blockSize, blockSize, comp + (size_t)b * cStride, (size_t)cs[b]);
            } else {
#pragma omp parallel for schedule(static, 16)
                for (b = 0; b < (long)nBlocks; b++)
                    (void)CPUB_DECOMPRESS(codec, back + (size_t)b * blockSize, blockSize, comp + (size_t)b * cStride, (size_t)cs[b]);
            }
            passes++; t1 = now_s();
        } while (t1 - t0 < minSeconds);
        if ((t1 - t0) / passes < bestD) { bestD = (t1 - t0) / passes; passesD = passes; }
    }
    out[0] = bestE; out[1] = bestD; out[2] = threads; out[3] = passesE; out[4] = passesD; out[5] = sum / (double)nBlocks;
    free(src); free(comp); free(back); free(cs);
    return 0;
}

/* Host memory bandwidth (parallel copy of `bytes` per pass, first-touched by the copying threads): the ceiling a
 * memory-bound multi-core figure is compared with.  Returns GB/s (read + write bytes). */
double CPUB_STREAM_NAME(size_t bytes, int nthreads, int passes)
{
    uint8_t* const a = (uint8_t*)malloc(bytes + 64);
    uint8_t* const c = (uint8_t*)malloc(bytes + 64);
    const long nChunks = (long)(bytes >> 20);
    long k; int p; double best = 1e30;
    if (!a || !c || nChunks == 0) { free(a); free(c); return 0.0; }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
    for (k = 0; k < nChunks; k++) { memset(a + ((size_t)k << 20), 1, (size_t)1 << 20); memset(c + ((size_t)k << 20), 2, (size_t)1 << 20); }
    for (p = 0; p < passes; p++) {
        const double t0 = now_s();
#pragma omp parallel for schedule(static)
        for (k = 0; k < nChunks; k++) memcpy(c + ((size_t)k << 20), a + ((size_t)k << 20), (size_t)1 << 20);
        {   const double t = now_s() - t0; if (t < best) best = t; }
    }
    {   const double gbs = 2.0 * (double)((size_t)nChunks << 20) / best / 1e9; free(a); free(c); return gbs; }
}
