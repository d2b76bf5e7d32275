/*
 * lz4.h -- legacy face of the MI355X LZ4 engine: the seven symbols (and one
 * macro) that Streamly.Internal.LZ4 imports today through
 * `foreign import ccall unsafe "lz4.h ..."` / `capi "lz4.h value ..."`
 * (reference src/Streamly/Internal/LZ4.hs:105-143).  Linking the unmodified
 * Haskell package against libmi355lz4.so with this header on the include path
 * keeps it building and running, one PCIe round trip per block.  The batched
 * entry points in mi355lz4.h are the ones meant for production.
 *
 * Semantics kept from the reference codec (cbits/lz4.c, v1.9.3):
 *   LZ4_compress_fast_continue   :1565  returns compressed size, 0 on failure;
 *                                       acceleration clamped to [1, 65537].
 *                                       Blocks are emitted INDEPENDENT (valid
 *                                       input for any linked decoder).
 *   LZ4_decompress_safe_continue :2322  returns decoded size, or the negative
 *                                       code -(ip-src)-1; the previous block's
 *                                       output is the dictionary of the next.
 *   LZ4_compressBound            :674   n + n/255 + 16, 0 if n > LZ4_MAX_INPUT_SIZE
 */
#ifndef MI355_LZ4_LEGACY_H
#define MI355_LZ4_LEGACY_H

#ifdef __cplusplus
extern "C" {
#endif

#define LZ4_MAX_INPUT_SIZE 0x7E000000 /* 2 113 929 216 bytes (cbits/lz4.h:170) */
#define LZ4_COMPRESSBOUND(isize) \
    ((unsigned)(isize) > (unsigned)LZ4_MAX_INPUT_SIZE ? 0 : (isize) + ((isize) / 255) + 16)

typedef struct LZ4_stream_u LZ4_stream_t;             /* opaque handle */
typedef struct LZ4_streamDecode_u LZ4_streamDecode_t; /* opaque handle */

LZ4_stream_t *LZ4_createStream(void);
int LZ4_freeStream(LZ4_stream_t *streamPtr);
LZ4_streamDecode_t *LZ4_createStreamDecode(void);
int LZ4_freeStreamDecode(LZ4_streamDecode_t *LZ4_stream);
int LZ4_compressBound(int inputSize);
int LZ4_compress_fast_continue(LZ4_stream_t *streamPtr, const char *src, char *dst, int srcSize,
                               int dstCapacity, int acceleration);
int LZ4_decompress_safe_continue(LZ4_streamDecode_t *LZ4_streamDecode, const char *src, char *dst,
                                 int srcSize, int dstCapacity);

#ifdef __cplusplus
}
#endif
#endif
