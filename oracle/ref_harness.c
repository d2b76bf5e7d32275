/*
 * oracle/ref_harness.c -- replays the Haskell call sequence against the REAL
 * reference codec (cbits/lz4.c compiled where it lies under /root/reference;
 * see oracle/Makefile).  Output lives only in oracle/_ref/ (git-ignored).
 *
 * TEST INFRASTRUCTURE ONLY (oracle validation, golden-vector generation and
 * bench.py's cpu_baseline "reference" leg).  No reference source is copied:
 * this file only calls the reference's public API through its own header.
 *
 * Call sequence restated from src/Streamly/Internal/LZ4.hs:
 *   compress   :353-394 (one LZ4_stream_t per stream; each input block in its
 *               own allocation; the previous input is kept alive one step)
 *   decompress :539-567 (one LZ4_streamDecode_t per stream; each output block
 *               in its own allocation; previous output kept alive one step)
 *   headers    :177-207, 261-262
 */
#include <lz4.h>

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static void put_le32(uint8_t *p, int32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static int32_t get_le32(const uint8_t *p)
{
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

int ref_version(void) { return LZ4_versionNumber(); }

/* linked=1: reference behaviour (one context for the whole stream).
 * linked=0: fresh context per block (independent blocks). */
size_t ref_frame_stream_compress(const uint8_t *in, size_t inLen, int blockLen, int accel,
                                 int headerKind, int linked, uint8_t *out, size_t outCap)
{
    LZ4_stream_t *ctx = LZ4_createStream();
    size_t pos = 0, o = 0;
    uint8_t *prev = NULL;
    if (accel < 0) accel = 0;
    while (pos < inLen) {
        int n = (int)((inLen - pos < (size_t)blockLen) ? inLen - pos : (size_t)blockLen);
        int bound = LZ4_compressBound(n), c;
        uint8_t *blk = (uint8_t *)malloc((size_t)n + 8);
        memcpy(blk, in + pos, (size_t)n);
        if (!linked) LZ4_initStream(ctx, sizeof(*ctx));
        if (o + (size_t)headerKind + (size_t)bound > outCap) { free(blk); free(prev); LZ4_freeStream(ctx); return (size_t)-1; }
        c = LZ4_compress_fast_continue(ctx, (const char *)blk, (char *)out + o + headerKind, n, bound, accel);
        if (c <= 0) { free(blk); free(prev); LZ4_freeStream(ctx); return (size_t)-1; }
        put_le32(out + o, c);
        if (headerKind == 8) put_le32(out + o + 4, n);
        o += (size_t)headerKind + (size_t)c;
        pos += (size_t)n;
        free(prev);
        prev = blk;
    }
    free(prev);
    LZ4_freeStream(ctx);
    return o;
}

size_t ref_frame_stream_decompress(const uint8_t *in, size_t inLen, int headerKind, int fixedUncomp,
                                   int linked, uint8_t *out, size_t outCap)
{
    LZ4_streamDecode_t *ctx = LZ4_createStreamDecode();
    size_t pos = 0, o = 0, k = 0;
    uint8_t *prev = NULL;
    while (pos + (size_t)headerKind <= inLen) {
        int32_t c = get_le32(in + pos);
        int32_t u = (headerKind == 8) ? get_le32(in + pos + 4) : fixedUncomp;
        uint8_t *blk;
        int r;
        if (c <= 0 || pos + (size_t)headerKind + (size_t)c > inLen || u < 0) { free(prev); LZ4_freeStreamDecode(ctx); return (size_t)-1 - k; }
        blk = (uint8_t *)malloc((size_t)u + 8);
        if (!linked) LZ4_setStreamDecode(ctx, NULL, 0);
        r = LZ4_decompress_safe_continue(ctx, (const char *)in + pos + headerKind, (char *)blk, c, u);
        if (r < 0 || o + (size_t)r > outCap) { free(blk); free(prev); LZ4_freeStreamDecode(ctx); return (size_t)-1 - k; }
        memcpy(out + o, blk, (size_t)r);
        o += (size_t)r;
        pos += (size_t)headerKind + (size_t)c;
        free(prev);
        prev = blk;
        k++;
    }
    free(prev);
    LZ4_freeStreamDecode(ctx);
    return o;
}

/* One block with an optional external dictionary -- what
 * LZ4_decompress_safe_continue reaches for separately allocated blocks
 * (cbits/lz4.c:2347-2355).  Uses the stream API so no internal symbol is needed. */
int ref_decompress_block_dict(const uint8_t *src, int srcLen, uint8_t *dst, int cap,
                              const uint8_t *dict, int dictLen)
{
    LZ4_streamDecode_t sd;
    LZ4_setStreamDecode(&sd, (const char *)dict, dict ? dictLen : 0);
    return LZ4_decompress_safe_continue(&sd, (const char *)src, (char *)dst, srcLen, cap);
}

int ref_compress_block(const uint8_t *src, uint8_t *dst, int n, int cap, int accel)
{
    LZ4_stream_t *ctx = LZ4_createStream();
    int r = LZ4_compress_fast_continue(ctx, (const char *)src, (char *)dst, n, cap, accel);
    LZ4_freeStream(ctx);
    return r;
}

/* ---- CPU baseline timers (bench.py cpu_baseline, kind "reference") ----
 * Pre-split blocks in separate allocations; time only the codec calls. */
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Compress nBlocks blocks (blocks[i], lens[i]) through ONE context (reference
 * behaviour).  Writes compressed blocks into outs[i] (capacity bound), sizes
 * into outLens[i].  Returns seconds. */
double ref_time_compress(const uint8_t *const *blocks, const int *lens, int nBlocks, int accel,
                         uint8_t *const *outs, int *outLens)
{
    LZ4_stream_t *ctx = LZ4_createStream();
    double t0, t1;
    int i;
    t0 = now_s();
    for (i = 0; i < nBlocks; i++) {
        outLens[i] = LZ4_compress_fast_continue(ctx, (const char *)blocks[i], (char *)outs[i], lens[i],
                                                LZ4_compressBound(lens[i]), accel);
    }
    t1 = now_s();
    LZ4_freeStream(ctx);
    return t1 - t0;
}

double ref_time_decompress(const uint8_t *const *comp, const int *compLens, int nBlocks,
                           uint8_t *const *outs, const int *outCaps, int *results)
{
    LZ4_streamDecode_t *ctx = LZ4_createStreamDecode();
    double t0, t1;
    int i;
    t0 = now_s();
    for (i = 0; i < nBlocks; i++) {
        results[i] = LZ4_decompress_safe_continue(ctx, (const char *)comp[i], (char *)outs[i], compLens[i], outCaps[i]);
    }
    t1 = now_s();
    LZ4_freeStreamDecode(ctx);
    return t1 - t0;
}
