/*
 * oracle/lz4_oracle.h -- CPU restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product (libmi355lz4.so) never links or dlopens it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout: cbits/lz4.c, cbits/lz4.h, src/Streamly/Internal/LZ4.hs).
 *
 * Parity pin: validated against the reference itself (cbits/lz4.c compiled by
 * oracle/Makefile into oracle/_ref/) and against tests/golden/ fixtures that
 * were generated from that build (tests/golden/make_golden.py).
 */
#ifndef LZ4_ORACLE_H
#define LZ4_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cbits/lz4.h:170-171, cbits/lz4.c:674 */
#define ORC_MAX_INPUT_SIZE 0x7E000000
int orc_compress_bound(int n);

/* ---- decode: cbits/lz4.c:1737-2165 instantiated as
 * (endOnInputSize, decode_full_block, noDict|usingExtDict), i.e. what
 * LZ4_decompress_safe (:2171) and LZ4_decompress_safe_forceExtDict (:2223)
 * compute.  dict may be NULL/0.  Returns decoded size >= 0, or the reference's
 * negative code -(ip-src)-1 (:2163). */
int orc_decompress_safe_dict(const uint8_t *src, int srcLen, uint8_t *dst,
                             int cap, const uint8_t *dict, size_t dictLen);

/* ---- streaming decode context: cbits/lz4.c:2265-2277, 2322-2359,
 * cbits/lz4.h:605-610.  Each block is assumed decoded into its own allocation
 * (the Haskell call sequence, Internal/LZ4.hs:539-567), so after the first
 * block the previous output is always an external dictionary. */
typedef struct {
    const uint8_t *prevOut; /* previous block's output (caller keeps it alive) */
    size_t prevLen;
} orc_dstream;
void orc_dstream_init(orc_dstream *s);
int orc_decompress_safe_continue(orc_dstream *s, const uint8_t *src, int srcLen,
                                 uint8_t *dst, int cap);

/* ---- streaming compress context: cbits/lz4.h:596-603,
 * cbits/lz4.c:1423-1451 (create/init), 1545-1562 (renorm), 1565-1637 (driver),
 * 851-1240 (byU32 / hash5 / usingExtDict match finder + emitter). */
typedef struct {
    uint32_t table[4096];   /* LZ4_HASHLOG 12, stream offsets (lz4.h:578-580) */
    uint32_t currentOffset;
    const uint8_t *dict;    /* previous input block (caller keeps it alive) */
    uint32_t dictSize;
} orc_cstream;
void orc_cstream_init(orc_cstream *s);
/* Returns compressed size (>0) or 0 when dst is too small. */
int orc_compress_fast_continue(orc_cstream *s, const uint8_t *src, uint8_t *dst,
                               int n, int cap, int accel);

/* One independent block through a fresh context (what our GPU compressor's
 * output is compared against for size). */
int orc_compress_block(const uint8_t *src, uint8_t *dst, int n, int cap,
                       int accel);

/* ---- framing: src/Streamly/Internal/LZ4.hs:177-207,226-336 ----
 * headerKind 8 = BlockHasSize ([compLen LE32][uncompLen LE32][data]),
 * headerKind 4 = BlockMax* ([compLen LE32][data]). */
size_t orc_frame_stream_compress(const uint8_t *in, size_t inLen, int blockLen,
                                 int accel, int headerKind, int linked,
                                 uint8_t *out, size_t outCap);
/* Decode a framed stream (blocks packed back to back).  fixedUncomp is the
 * decode capacity for headerKind 4.  Returns total decoded bytes or
 * (size_t)-1-k when block k fails. */
size_t orc_frame_stream_decompress(const uint8_t *in, size_t inLen,
                                   int headerKind, int fixedUncomp, int linked,
                                   uint8_t *out, size_t outCap);

/* ---- deterministic generators ---- */
void orc_gen_random(uint8_t *dst, size_t blockLen, uint64_t blockIndex);
void orc_gen_lzsynth(uint8_t *dst, size_t blockLen, uint64_t blockIndex,
                     uint32_t litMax, uint32_t offMax);
void orc_gen_text(uint8_t *dst, size_t blockLen, uint64_t blockIndex);

#ifdef __cplusplus
}
#endif
#endif
