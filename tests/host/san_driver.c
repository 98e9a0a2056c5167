/* tests/host/san_driver.c -- TEST INFRASTRUCTURE: the memcheck leg of the host side of libfsehip.so (the reference ships `make sanitize`
 * and `memtest` for its own host code: /root/reference/Makefile:75-79, programs/Makefile:165-170).
 *
 * A plain C program (the Python tests cannot run under AddressSanitizer on this image: the sanitizer runtime's HSA interceptor does not
 * get along with the ROCm runtime bundled inside the torch wheel) that drives, from several host threads at once, what keeps state on the
 * host side of the library -- the per-thread scratch arenas of the calls on host pointers (capi.hip), the frame calls' thread pool,
 * streams and rings (frame.hip), FSEHIP_releaseScratch, the _wksp entry points -- and compares every result with the compiled reference
 * (oracle/_ref/libfse_ref.so: the reference's own lib/ sources) byte for byte.  Built twice by scripts/sanitize.sh: against the product
 * library and against the -fsanitize=address,undefined build of the same sources (finitestateentropy_amd/csrc/variants/san).
 *
 * Exit code 0 and "san_driver OK" on success. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FSE_STATIC_LINKING_ONLY
#define HUF_STATIC_LINKING_ONLY
#include "fse.h"
#include "huf.h"
#include "hist.h"
#include "fsehip.h"

#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "%s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

/* a block with a skewed byte distribution (geometric over `alphabet` symbols), a run, or noise */
static void fill_block(uint8_t* p, size_t n, uint32_t seed, int kind)
{
    uint32_t s = seed * 2654435761u + 17;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t r = lcg(&s);
        if (kind == 0) { unsigned v = 0; uint32_t t = r & 0xFFFF; while ((t & 1) && v < 40) { ++v; t >>= 1; } p[i] = (uint8_t)(v + (r >> 20) % 3); }
        else if (kind == 1) p[i] = (uint8_t)((r >> 3) % 7 == 0 ? 1 + (r >> 9) % 5 : 0);
        else if (kind == 2) p[i] = 77;
        else p[i] = (uint8_t)(r >> 5);
    }
}

typedef struct { int id; int rounds; int failed; } Job;

static int one_round(int id, int round, uint8_t* src, uint8_t* a, uint8_t* b, uint8_t* back, size_t maxN)
{
    static const size_t sizes[] = { 32768, 4097, 300, 65536, 1, 12, 131072, 20000 };
    const size_t n = sizes[(id + round) % 8] <= maxN ? sizes[(id + round) % 8] : maxN;
    const int kind = (id * 3 + round) % 4;
    fill_block(src, n, 1000u * id + round, kind);
    const size_t cap = FSE_COMPRESSBOUND(n);

    /* HIST_count / _wksp / countFast */
    {   unsigned c1[256], c2[256], m1 = 255, m2 = 255, wk[1024];
        const size_t r1 = FSEHIP_HIST_count(c1, &m1, src, n), r2 = HIST_count(c2, &m2, src, n);
        CHECK(r1 == r2 && m1 == m2 && !memcmp(c1, c2, (m1 + 1) * 4), "HIST_count differs (thread %d round %d)", id, round);
        m1 = m2 = 255;
        CHECK(FSEHIP_HIST_count_wksp(c1, &m1, src, n, wk, sizeof wk) == HIST_count_wksp(c2, &m2, src, n, wk, sizeof wk) && m1 == m2, "HIST_count_wksp differs");
        m1 = m2 = 255;
        CHECK(FSEHIP_HIST_countFast(c1, &m1, src, n) == HIST_countFast(c2, &m2, src, n) && m1 == m2, "HIST_countFast differs");
    }
    /* FSE one-shot + wksp forms */
    {   const unsigned tl = round & 1 ? 12 : 11;
        const size_t r1 = FSEHIP_FSE_compress2(a, cap, src, n, 255, tl), r2 = FSE_compress2(b, cap, src, n, 255, tl);
        CHECK(r1 == r2, "FSE_compress2 returns %zu, reference %zu (n %zu kind %d)", r1, r2, n, kind);
        if (r1 > 1 && !FSE_isError(r1)) {
            CHECK(!memcmp(a, b, r1), "FSE_compress2 bytes differ");
            CHECK(FSEHIP_FSE_decompress(back, n, a, r1) == n && !memcmp(back, src, n), "FSE_decompress round trip");
            unsigned dt[FSE_DTABLE_SIZE_U32(12)];
            CHECK(FSEHIP_FSE_decompress_wksp(back, n, a, r1, dt, 12) == n && !memcmp(back, src, n), "FSE_decompress_wksp round trip");
            CHECK(FSEHIP_FSE_decompress(back, n - 1, a, r1) == FSE_decompress(back, n - 1, a, r1), "FSE_decompress verdict on a short destination");
        }
        unsigned wk[FSE_WKSP_SIZE_U32(12, 255)];
        const size_t w1 = FSEHIP_FSE_compress_wksp(a, cap, src, n, 255, tl, wk, sizeof wk), w2 = FSE_compress_wksp(b, cap, src, n, 255, tl, wk, sizeof wk);
        CHECK(w1 == w2 && (w1 <= 1 || FSE_isError(w1) || !memcmp(a, b, w1)), "FSE_compress_wksp differs");
    }
    /* Huff0 one-shot + wksp forms (blocks up to 128 KB) */
    if (n <= HUF_BLOCKSIZE_MAX) {
        const size_t hcap = HUF_COMPRESSBOUND(n);
        const size_t r1 = FSEHIP_HUF_compress2(a, hcap, src, n, 255, 11), r2 = HUF_compress2(b, hcap, src, n, 255, 11);
        CHECK(r1 == r2, "HUF_compress2 returns %zu, reference %zu (n %zu kind %d)", r1, r2, n, kind);
        if (r1 > 1 && !HUF_isError(r1)) {
            CHECK(!memcmp(a, b, r1), "HUF_compress2 bytes differ");
            CHECK(FSEHIP_HUF_decompress(back, n, a, r1) == n && !memcmp(back, src, n), "HUF_decompress round trip");
            HUF_CREATE_STATIC_DTABLEX1(dctx, HUF_TABLELOG_MAX);
            unsigned wk[HUF_DECOMPRESS_WORKSPACE_SIZE_U32];
            CHECK(FSEHIP_HUF_decompress4X1_DCtx_wksp(dctx, back, n, a, r1, wk, sizeof wk) == n && !memcmp(back, src, n), "HUF_decompress4X1_DCtx_wksp round trip");
        }
        unsigned wk[HUF_WORKSPACE_SIZE_U32];
        const size_t w1 = FSEHIP_HUF_compress1X_wksp(a, hcap, src, n, 255, 11, wk, sizeof wk), w2 = HUF_compress1X_wksp(b, hcap, src, n, 255, 11, wk, sizeof wk);
        CHECK(w1 == w2 && (w1 <= 1 || HUF_isError(w1) || !memcmp(a, b, w1)), "HUF_compress1X_wksp differs (%zu vs %zu)", w1, w2);
    }
    if ((round % 3) == 2) {      /* (another worker may be inside the helper pool's release at this moment: the call then says so, fsehip.h) */
        const int rs = FSEHIP_releaseScratch();
        CHECK(rs == 0 || rs == FSEHIP_SCRATCH_BUSY, "releaseScratch returns %d", rs);
    }
    return 0;
}

static void* worker(void* arg)
{
    Job* j = (Job*)arg;
    const size_t maxN = 131072;
    uint8_t* src = malloc(maxN); uint8_t* a = malloc(FSE_COMPRESSBOUND(maxN)); uint8_t* b = malloc(FSE_COMPRESSBOUND(maxN)); uint8_t* back = malloc(maxN);
    for (int r = 0; r < j->rounds && !j->failed; ++r) j->failed = one_round(j->id, r, src, a, b, back, maxN);
    free(src); free(a); free(b); free(back);
    return NULL;          /* the thread's arena goes with the thread (thread_local destructor) unless it was released above */
}

/* frames: many per call over the library's own host thread pool; every frame decodes back and equals the single-frame call's bytes */
static int frames(void)
{
    enum { NF = 24 };
    void* srcs[NF]; void* dsts[NF]; void* one; void* backs[NF];
    size_t srcSizes[NF], caps[NF], res[NF], bres[NF];
    for (int i = 0; i < NF; ++i) {
        srcSizes[i] = (size_t)(i % 5 == 4 ? 0 : 1000 + 37003 * (size_t)i);
        srcs[i] = malloc(srcSizes[i] + 1);
        fill_block(srcs[i], srcSizes[i], 77 + i, i % 4);
        caps[i] = FSEHIP_frame_compressBound(srcSizes[i], 5);
        dsts[i] = malloc(caps[i]); backs[i] = malloc(srcSizes[i] + 1);
    }
    for (int codec = 0; codec < 2; ++codec)
        for (unsigned nt = 1; nt <= 4; nt += 3) {
            CHECK(FSEHIP_frame_compress_batch(dsts, caps, (const void* const*)srcs, srcSizes, res, NF, 5, codec, nt) == 0, "frame_compress_batch");
            for (int i = 0; i < NF; ++i) {
                CHECK(!FSEHIP_isError(res[i]), "frame %d: %s", i, FSEHIP_getErrorName(res[i]));
                one = malloc(caps[i]);
                const size_t r1 = FSEHIP_frame_compress(one, caps[i], srcs[i], srcSizes[i], 5, codec);
                CHECK(r1 == res[i] && !memcmp(one, dsts[i], r1), "frame %d: batch and single call differ", i);
                free(one);
            }
            CHECK(FSEHIP_frame_decompress_batch(backs, srcSizes, (const void* const*)dsts, res, bres, NF, nt) == 0, "frame_decompress_batch");
            for (int i = 0; i < NF; ++i) CHECK(bres[i] == srcSizes[i] && !memcmp(backs[i], srcs[i], srcSizes[i]), "frame %d round trip (%zu)", i, bres[i]);
        }
    for (int i = 0; i < NF; ++i) { free(srcs[i]); free(dsts[i]); free(backs[i]); }
    return 0;
}

int main(int argc, char** argv)
{
    const int nThreads = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 12;
    FSEHIP_DeviceInfo info;
    if (FSEHIP_deviceInfo(&info) != 0) { fprintf(stderr, "no gfx950 device\n"); return 2; }
    pthread_t th[16]; Job jobs[16];
    const int nt = nThreads > 16 ? 16 : nThreads;
    for (int i = 0; i < nt; ++i) { jobs[i].id = i; jobs[i].rounds = rounds; jobs[i].failed = 0; pthread_create(&th[i], NULL, worker, &jobs[i]); }
    int failed = 0;
    for (int i = 0; i < nt; ++i) { pthread_join(th[i], NULL); failed |= jobs[i].failed; }
    if (failed) return 1;
    if (frames()) return 1;
    /* a second wave of threads after the first one's arenas are gone, then the main thread's own arena */
    for (int i = 0; i < 2; ++i) { jobs[i].id = 20 + i; jobs[i].rounds = 4; jobs[i].failed = 0; pthread_create(&th[i], NULL, worker, &jobs[i]); }
    for (int i = 0; i < 2; ++i) { pthread_join(th[i], NULL); failed |= jobs[i].failed; }
    Job me = { 31, 5, 0 };
    worker(&me);
    if (failed || me.failed) return 1;
    printf("san_driver OK: %d threads x %d rounds of host-pointer calls against the reference, %d frames x 2 codecs x 2 pool sizes, on %s\n", nt, rounds, 24, info.archName);
    return 0;
}
