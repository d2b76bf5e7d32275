/* roundtrip.c -- the batched C ABI from plain C: compress N host blocks in one call, decompress the framed stream in
 * one call, compare.  The stream is the reference's wire format ([compLen LE32][uncompLen LE32][LZ4 block] per block),
 * so either side can be replaced by streamly-lz4 / liblz4.
 *
 *   gcc -std=c99 -I include examples/roundtrip.c -L streamly-lz4_amd/lib -lmi355lz4 -Wl,-rpath,$PWD/streamly-lz4_amd/lib -o /tmp/roundtrip
 *   /tmp/roundtrip [blocks] [linked]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355lz4.h"

int main(int argc, char **argv)
{
    const int nBlocks = argc > 1 ? atoi(argv[1]) : 256, linked = argc > 2 ? atoi(argv[2]) : 0;
    const int blockLen = 65536;
    mi355lz4_ctx *ctx = NULL;
    if (mi355lz4_create(&ctx, 0) != MI355LZ4_OK) { fprintf(stderr, "create: %s\n", mi355lz4_last_error()); return 2; }

    /* text-like input: words from a small vocabulary */
    uint8_t *raw = (uint8_t *)malloc((size_t)nBlocks * blockLen);
    uint64_t s = 88172645463325252ULL;
    for (size_t i = 0; i < (size_t)nBlocks * blockLen;) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const unsigned w = (unsigned)(s % 997), len = 2 + w % 7;
        for (unsigned k = 0; k < len && i < (size_t)nBlocks * blockLen; k++) raw[i++] = (uint8_t)('a' + (w * (k + 3)) % 26);
        if (i < (size_t)nBlocks * blockLen) raw[i++] = ' ';
    }
    const uint8_t **src = (const uint8_t **)malloc(sizeof(*src) * (size_t)nBlocks);
    int32_t *len = (int32_t *)malloc(sizeof(*len) * (size_t)nBlocks), *flen = (int32_t *)malloc(sizeof(*flen) * (size_t)nBlocks);
    int32_t *status = (int32_t *)malloc(sizeof(*status) * (size_t)nBlocks), *blen = (int32_t *)malloc(sizeof(*blen) * (size_t)nBlocks);
    for (int i = 0; i < nBlocks; i++) { src[i] = raw + (size_t)i * blockLen; len[i] = blockLen; }

    const size_t cap = (size_t)nBlocks * ((size_t)mi355lz4_compress_bound(blockLen) + 8);
    uint8_t *framed = (uint8_t *)malloc(cap), *back = (uint8_t *)malloc((size_t)nBlocks * blockLen);
    size_t framedLen = 0, outLen = 0;
    int got = 0;
    mi355lz4_set_linked_compress(ctx, linked);
    if (mi355lz4_compress_batch(ctx, src, len, nBlocks, 1, 8, framed, cap, &framedLen, flen, status) != MI355LZ4_OK) {
        fprintf(stderr, "compress: %s\n", mi355lz4_last_error());
        return 1;
    }
    if (mi355lz4_decompress_batch(ctx, framed, framedLen, 8, 0, linked, NULL, 0, back, (size_t)nBlocks * blockLen, &outLen, blen,
                                  nBlocks, &got) != MI355LZ4_OK) {
        fprintf(stderr, "decompress: %s\n", mi355lz4_last_error());
        return 1;
    }
    const int ok = got == nBlocks && outLen == (size_t)nBlocks * blockLen && memcmp(raw, back, outLen) == 0;
    printf("%d blocks of %d bytes, %s blocks: %zu -> %zu bytes (ratio %.3f), round trip %s\n", nBlocks, blockLen,
           linked ? "linked" : "independent", outLen, framedLen, (double)outLen / (double)framedLen, ok ? "ok" : "MISMATCH");
    mi355lz4_destroy(ctx);
    free(raw); free((void *)src); free(len); free(flen); free(status); free(blen); free(framed); free(back);
    return ok ? 0 : 1;
}
