/*
 * oracle/orc_bench.c -- timing loops over the oracle restatement, used by
 * bench.py's cpu_baseline leg with kind "port" when oracle/_ref is absent.
 * TEST INFRASTRUCTURE ONLY.  Same call sequence as ref_harness.c
 * (src/Streamly/Internal/LZ4.hs:353-394, 539-567).
 */
#include "lz4_oracle.h"

#include <stdlib.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_time_compress(const uint8_t *const *blocks, const int *lens, int nBlocks, int accel,
                         uint8_t *const *outs, int *outLens)
{
    orc_cstream *s = (orc_cstream *)malloc(sizeof(*s));
    double t0, t1;
    int i;
    orc_cstream_init(s);
    t0 = now_s();
    for (i = 0; i < nBlocks; i++)
        outLens[i] = orc_compress_fast_continue(s, blocks[i], outs[i], lens[i], orc_compress_bound(lens[i]), accel);
    t1 = now_s();
    free(s);
    return t1 - t0;
}

double orc_time_decompress(const uint8_t *const *comp, const int *compLens, int nBlocks,
                           uint8_t *const *outs, const int *outCaps, int *results)
{
    orc_dstream s;
    double t0, t1;
    int i;
    orc_dstream_init(&s);
    t0 = now_s();
    for (i = 0; i < nBlocks; i++)
        results[i] = orc_decompress_safe_continue(&s, comp[i], compLens[i], outs[i], outCaps[i]);
    t1 = now_s();
    return t1 - t0;
}
