/*
 * mi355lz4.h -- C ABI of the MI355X (gfx950) LZ4 block engine.
 *
 * This is the drop-in boundary for the hot path of composewell/streamly-lz4:
 * the two foreign calls made once per block by Streamly.Internal.LZ4,
 *
 *   c_compressFastContinue    src/Streamly/Internal/LZ4.hs:123-131 -> cbits/lz4.c:1565
 *   c_decompressSafeContinue  src/Streamly/Internal/LZ4.hs:133-140 -> cbits/lz4.c:2322
 *
 * plus the framing those calls are wrapped in (compressChunk :226-281,
 * decompressChunk :291-336, header layout :177-207).
 *
 * Two faces:
 *   (1) include/lz4.h  -- the exact 7 legacy symbols the Haskell imports today
 *       (one block per call; source compatible, not how a GPU should be fed);
 *   (2) this header    -- the batched ABI (N blocks per call) the modified
 *       Haskell combinators in INTEGRATION.md bind with `ccall safe`.
 *
 * Plain pointers and sizes only; no C++ or torch types.  All functions return
 * MI355LZ4_OK (0) or a negative MI355LZ4_E_* code unless stated otherwise.
 * Every entry point fails with MI355LZ4_E_NO_DEVICE when no gfx950 device is
 * usable: there is NO CPU fallback.
 *
 * Framed block layout (identical to the reference, little-endian int32s):
 *   headerKind 8 (BlockHasSize, default): [compLen][uncompLen][compLen bytes]
 *   headerKind 4 (BlockMax64KB..4MB)    : [compLen][compLen bytes]
 *
 * Blocks produced by the compressor are INDEPENDENT LZ4 blocks (no reference
 * into a previous block), which the reference's linked decoder accepts
 * unchanged.  The decompressor also accepts the reference's LINKED streams:
 * see mi355lz4_decompress_batch_device (linked != 0).
 */
#ifndef MI355LZ4_H
#define MI355LZ4_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355LZ4_VERSION 100

/* ---- status codes --------------------------------------------------- */
#define MI355LZ4_OK              0
#define MI355LZ4_E_NO_DEVICE    (-1)  /* no usable HIP device / not gfx950 */
#define MI355LZ4_E_HIP          (-2)  /* a HIP runtime call failed (see mi355lz4_last_error) */
#define MI355LZ4_E_ARG          (-3)  /* bad argument */
#define MI355LZ4_E_CAPACITY     (-4)  /* output buffer too small */
#define MI355LZ4_E_BLOCK        (-5)  /* at least one block failed; see per-block status */
#define MI355LZ4_E_STREAM       (-6)  /* malformed framed stream (header chain) */

/* Per-block decode results follow the reference: >= 0 is the decoded size,
 * -1 .. -(compLen+1) is cbits/lz4.c:2163's -(ip-src)-1.  Header-level
 * rejections (what decompressChunk checks at Internal/LZ4.hs:309-318, plus the
 * short-array case it misses) use this separate range: */
#define MI355LZ4_BLK_E_COMPLEN   (-0x7F000001)  /* compLen <= 0 or > LZ4_compressBound(LZ4_MAX_INPUT_SIZE) */
#define MI355LZ4_BLK_E_TRUNCATED (-0x7F000002)  /* header/data runs past the framed buffer */
#define MI355LZ4_BLK_E_UNCOMPLEN (-0x7F000003)  /* negative uncompLen / exceeds output capacity */

#define MI355LZ4_MAX_INPUT_SIZE 0x7E000000      /* = LZ4_MAX_INPUT_SIZE, cbits/lz4.h:170 */

typedef struct mi355lz4_ctx mi355lz4_ctx;

/* ---- engine lifecycle ------------------------------------------------ */
int mi355lz4_version(void);
/* Thread-local description of the last failure in this library. */
const char *mi355lz4_last_error(void);
/* Number of usable gfx950 devices (0 when none; never fails). */
int mi355lz4_device_count(void);
/* Create an engine bound to HIP device `device` with its own stream.
 *
 * Threads: an engine is NOT thread-safe -- one engine per thread, or the caller serialises its calls (the
 * reference's contexts are the same: one per stream, one Haskell thread, Internal/LZ4.hs:105-143).  Different
 * engines may be used from different threads at the same time.  mi355lz4_last_error is thread-local.
 *
 * Streams: every *_device call only ENQUEUES work on the engine's stream (its own, or the one given to
 * mi355lz4_set_stream) and returns; the exception is a linked decode (linked != 0, or the streams call), which
 * waits on the host for its first pass before it decides whether a second one is needed.  The scratch memory of a
 * linked decode belongs to the engine: two linked decodes of ONE engine must be issued on the same stream, or the
 * caller orders them.  The host-buffer calls (mi355lz4_compress_batch, mi355lz4_decompress_batch, ..._streams)
 * are synchronous and use, besides the engine's stream, two copy streams and two compute streams created on first
 * use, plus a small pool of host threads for staging pageable memory (MI355LZ4_COPY_THREADS, default 8); for the
 * duration of such a call the engine's stream is one of its own.
 *
 * Process environment: when this is the first HIP call of the process and GPU_MAX_HW_QUEUES is not set, the first
 * mi355lz4_create sets it to 8 (the pipelines above need more than HIP's default of four hardware queues to overlap);
 * MI355LZ4_KEEP_HW_QUEUES=1 turns that off.  Nothing is changed at load time. */
int mi355lz4_create(mi355lz4_ctx **out, int device);
void mi355lz4_destroy(mi355lz4_ctx *ctx);
/* Launch on a caller-owned hipStream_t instead (e.g. torch's current stream). */
int mi355lz4_set_stream(mi355lz4_ctx *ctx, void *hipStream);
void *mi355lz4_get_stream(mi355lz4_ctx *ctx);
int mi355lz4_synchronize(mi355lz4_ctx *ctx);
/* Linked decodes without a host wait.  mi355lz4_decompress_batch_device(linked != 0) normally waits on the host for
 * its first pass, to learn whether any block needs its dictionary and which ones (a stream of independent blocks pays
 * that wait and nothing else).  With maxDecodedBlockSize > 0 (an upper bound of any block's decoded size, e.g. 65536)
 * the call only enqueues: the second pass is issued over all blocks of the call and its kernels return at once when
 * the first pass found nothing to do (about a dozen empty launches per 4096 blocks).  The call can then be captured in
 * a graph and no longer serialises a caller's pipeline; it costs more than the wait on big batches of independent
 * blocks (measured in DESIGN.md), which is why it is opt-in.  0 restores the default.  The streams call
 * (mi355lz4_decompress_streams_device) always waits.
 * Before capturing such a call in a graph, make ONE warm-up call with the largest batch the graph will see: the
 * scratch is sized for ALL blocks of the call whether or not any is dependent (lists: one byte per output byte, up
 * to 16384 blocks' worth; source pointers: four bytes per output byte of a 4096-block segment, about 1 GiB) and is
 * allocated (hipMalloc / hipFree, not capturable) the first time a call needs more than the engine holds. */
int mi355lz4_set_linked_async(mi355lz4_ctx *ctx, int maxDecodedBlockSize);
/* Small batches.  With fewer blocks in a call than the chip has wave slots, the compressor cuts every block
 * (8 KiB .. 4 MiB, independent blocks) into segments that several wavefronts compress at once (a block still
 * comes out as one valid LZ4 block; the seams cost about 1 % of size on text).  segs: -1 = automatic (default;
 * MI355LZ4_SEG in the environment overrides it), 0 = never, 2..64 = that many segments whenever possible.
 * Consequences a caller should know: (1) the compressed BYTES of a block depend on how many blocks share the call
 * (the small tail batch of a stream is cut into segments, the big batches before it are not); every form decodes to
 * the same data, and segs = 0 gives bytes that do not depend on the batch.  (2) The segment path keeps a device
 * scratch of about twice the call's input per stream it was used on (at most four streams; a fifth takes over the
 * slot used longest ago) until mi355lz4_destroy; automatic mode leaves the path alone once that scratch would pass
 * 1 GiB, a forced count does not. */
int mi355lz4_set_segments(mi355lz4_ctx *ctx, int segs);
/* Decoder variant: 0 = chosen per call (default), 1 = sequence-at-a-time kernel, 2 = lane-parallel kernel (one wavefront
 * per block: what fills the GPU when a call brings thousands of blocks), 4 = one workgroup per block (sixteen wavefronts
 * share a block's output in LDS, 32 KiB at a time: a block's latency is 1.5-2 x shorter; in a linked call it is the
 * first, standalone pass -- blocks that need their dictionary go through the second pass as ever).  Variant 0 takes
 * variant 4 for calls of up to 256 blocks -- 512 when they hold 16 KiB of compressed bytes or more on average, none when
 * less than 3 KiB -- and variant 2 otherwise (MI355LZ4_CU_BLOCKS = n in the environment: up to n blocks whatever their
 * size; 0 = never); under variant 0 the kernel itself hands a block that saves less than a sixteenth of its size to the
 * lane-parallel decoder (long literal runs end that form's segments), under variant 4 it does not.
 * Tuning/ablation knob; results are identical.  Any other value: MI355LZ4_E_ARG. */
int mi355lz4_set_decoder(mi355lz4_ctx *ctx, int variant);
/* on != 0: the compress calls treat the blocks of a call as consecutive blocks of ONE stream and use block
 * i-1 as block i's dictionary whenever it lies directly in front of it in memory -- what the reference's
 * LZ4_compress_fast_continue does with the previous chunk (cbits/lz4.c:1608-1636, kept alive by
 * Internal/LZ4.hs:376,389).  The output is then a LINKED stream like the reference's (+6 % ratio on text);
 * it must be decoded with linked != 0, in order.  Default off: independent blocks (they shard). */
int mi355lz4_set_linked_compress(mi355lz4_ctx *ctx, int on);

/* = LZ4_compressBound (cbits/lz4.c:674, lz4.h:171): n + n/255 + 16, 0 if n too large */
int mi355lz4_compress_bound(int n);
/* Bytes one worst-case framed slot needs for a block of blockLen bytes, rounded up to 16. */
size_t mi355lz4_slot_stride(int blockLen, int headerKind);

/* ---- device-resident batched API (all data pointers are DEVICE pointers;
 *      asynchronous on the engine's stream) ----------------------------- */

/* Compress nBlocks blocks.  Block i is src[srcOff[i] .. srcOff[i]+srcLen[i]);
 * srcOff == NULL means srcOff[i] = i * blockStride; srcLen == NULL means every
 * block is maxBlockLen bytes (maxBlockLen must bound every srcLen[i]: it picks
 * the hash-table entry width).  Block i's framed bytes
 * ([header][data]) are written at slots + i*slotStride and their count
 * (headerKind + compLen) to framedLen[i].  accel follows
 * LZ4_compress_fast_continue (clamped to [1,65537], cbits/lz4.c:1577-1578).
 * replaces: compressChunk, Internal/LZ4.hs:226-281. */
int mi355lz4_compress_batch_device(mi355lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff,
                                   const int32_t *srcLen, uint64_t blockStride, int maxBlockLen, int nBlocks,
                                   int accel, int headerKind, uint8_t *slots, size_t slotStride,
                                   int32_t *framedLen);

/* Pack slots into one dense framed stream: denseOff[0..nBlocks] receives the
 * exclusive scan of framedLen (denseOff[nBlocks] = total), dense the bytes.
 * Nothing is written at or past dense + denseCap: a block that does not fit is
 * skipped, and the caller sees it by denseOff[nBlocks] > denseCap (the call is
 * asynchronous, so it cannot report that itself).  nBlocks * slotStride always
 * suffices. */
int mi355lz4_compact_device(mi355lz4_ctx *ctx, const uint8_t *slots, size_t slotStride,
                            const int32_t *framedLen, int nBlocks, uint8_t *dense, size_t denseCap,
                            uint64_t *denseOff);

/* Decompress nBlocks framed blocks.  Block i's header starts at
 * framed + blockOff[i]; its output goes to out + outOff[i] with capacity
 * outCap[i] (outCap == NULL: capacity = header uncompLen, or fixedUncomp for
 * headerKind 4).  result[i] = decoded size or a negative code (see above).
 * linked == 0: every block is decoded on its own (LZ4_decompress_safe).
 * linked != 0: reference stream semantics -- block i may reference the output
 * of the last block before it that decoded to > 0 bytes, exactly as under
 * LZ4_decompress_safe_continue with separately allocated blocks
 * (cbits/lz4.c:2322-2359); blocks must be given in stream order.  Blocks that
 * decode on their own (everything this engine's compressor emits) are final
 * after the parallel kernel; blocks that reach into their predecessor are
 * resolved by a second, data-parallel pass: up to 512 big blocks (512 KiB
 * and more: BlockMax1MB / BlockMax4MB streams) by a workgroup each against a
 * guess of their dictionary, pass after pass until the guesses stand (DESIGN.md
 * 0c: scratch 64 KiB per block), short runs of them by a wave per
 * run, spans of 576 MiB and more by the run-in decode (pieces of the span,
 * each decoded from a few blocks in front of it and checked against what the
 * piece in front wrote, DESIGN.md 0b: scratch two blocks per piece, at most
 * 4096 pieces),
 * everything else through source pointers + pointer jumping (DESIGN.md 1:
 * scratch 1 + 4 bytes per output byte of up to 4096 blocks at a time).  With
 * linked != 0 the call WAITS for the first pass on the engine's stream (it
 * reads back how many blocks need the second pass and sizes it); with
 * linked == 0 it only enqueues work.  Environment knobs of the second pass
 * (read per call; for tests and measurements): MI355LZ4_LINKED_RUNS,
 * MI355LZ4_LINKED_RUNIN (0 = never, 1 = always), MI355LZ4_LINKED_RUNIN_PIECE,
 * MI355LZ4_LINKED_RUNIN_BLOCKS, MI355LZ4_LINKED_RUNIN_SPIN (polls a piece waits for the piece in front of it inside a launch),
 * MI355LZ4_LINKED_PTR, MI355LZ4_LINKED_PTR_BLOCKS, MI355LZ4_LINKED_POOL_BLOCKS,
 * MI355LZ4_LINKED_BIG (0 = never the big-block path; n = blocks from n KiB on).
 * replaces: decompressChunk, Internal/LZ4.hs:291-336. */
int mi355lz4_decompress_batch_device(mi355lz4_ctx *ctx, const uint8_t *framed, uint64_t framedLen,
                                     const uint64_t *blockOff, int nBlocks, int headerKind, int fixedUncomp,
                                     int linked, uint8_t *out, const uint64_t *outOff, const int32_t *outCap,
                                     int32_t *result);

/* Many linked streams in one call.  Stream s is the blocks
 * [streamFirst[s], streamFirst[s+1]) (device array of nStreams + 1 ascending
 * block indices); inside a stream the semantics are those of linked != 0
 * above, and no block ever sees another stream's output.  The dependency chain
 * inside a stream is serial by construction of the format, so a stream is walked
 * by one wavefront (lane-parallel inside each block) and throughput comes from
 * the number of streams in the call.
 * replaces: one decompressChunksRawD state machine per stream,
 * Internal/LZ4.hs:539-567 -> cbits/lz4.c:2322. */
int mi355lz4_decompress_streams_device(mi355lz4_ctx *ctx, const uint8_t *framed, uint64_t framedLen,
                                       const uint64_t *blockOff, int nBlocks, int headerKind, int fixedUncomp,
                                       const int32_t *streamFirst, int nStreams, uint8_t *out,
                                       const uint64_t *outOff, const int32_t *outCap, int32_t *result);

/* Read the headers of nBlocks framed blocks at blockOff[] and produce
 * outOff[0..nBlocks] = exclusive scan of their uncompressed sizes. */
int mi355lz4_index_device(mi355lz4_ctx *ctx, const uint8_t *framed, uint64_t framedLen,
                          const uint64_t *blockOff, int nBlocks, int headerKind, int fixedUncomp,
                          uint64_t *outOff);

/* ---- host-buffer batched API (what the Haskell shim binds; synchronous) -
 * A call is pipelined over groups of blocks (MI355LZ4_GROUP_MB, default 64 MiB):
 * H2D of group i+1, the kernels of group i and D2H of group i-1 overlap on
 * separate streams.  Caller buffers that are page-locked (hipHostMalloc /
 * hipHostRegister) are handed to the DMA engines directly; pageable ones are
 * staged through pinned slots by a small copy pool (MI355LZ4_COPY_THREADS).
 * A call of one or two groups has no other group to hide its staging behind:
 * its staging copies go in two pieces, the DMA engine moving one while the
 * pool copies the other (MI355LZ4_STAGE_PIECES overrides the count).  10 MiB
 * of 64 KiB blocks, decompress: 0.57 ms from pageable buffers, 0.43 ms from
 * page-locked ones (the kernel: 0.10 ms). */

/* Compress nBlocks host arrays into one dense framed stream in framedOut
 * (capacity cap).  blockFramedLen[i] (optional) = headerKind + compLen of
 * block i, so the caller can slice one Array per block like compressChunk does.
 * status[i] (optional) = compLen > 0, or 0 on failure (reference convention,
 * Internal/LZ4.hs:257-260). */
int mi355lz4_compress_batch(mi355lz4_ctx *ctx, const uint8_t *const *src, const int32_t *srcLen,
                            int nBlocks, int accel, int headerKind, uint8_t *framedOut, size_t cap,
                            size_t *outLen, int32_t *blockFramedLen, int32_t *status);

/* Walk the header chain of a dense framed stream on the host
 * (resizeChunksD's job, Internal/LZ4.hs:459-484): fills blockOff[k] and
 * uncompLen[k] for up to maxBlocks blocks; *nBlocks = blocks found.
 * MI355LZ4_E_STREAM on a malformed chain (trailing partial block). */
int mi355lz4_index_host(const uint8_t *framedIn, size_t inLen, int headerKind, int fixedUncomp,
                        uint64_t *blockOff, int32_t *uncompLen, int maxBlocks, int *nBlocks);

/* Decompress a dense framed stream held in host memory.  Output blocks are
 * written back to back into out; blockLen[k] = decoded size of block k (>= 0)
 * or its negative code.  linked as for the device call; when linked, dict /
 * dictLen (host memory, may be NULL/0) is the output of the block that preceded
 * framedIn[0] in the stream -- the array the Haskell decoder state keeps alive
 * (Internal/LZ4.hs:564) -- so a stream can be fed in several calls. */
int mi355lz4_decompress_batch(mi355lz4_ctx *ctx, const uint8_t *framedIn, size_t inLen, int headerKind,
                              int fixedUncomp, int linked, const uint8_t *dict, int dictLen, uint8_t *out,
                              size_t cap, size_t *outLen, int32_t *blockLen, int maxBlocks, int *nBlocks);

/* Host-buffer form of mi355lz4_decompress_streams_device: framedIn holds the blocks
 * of nStreams linked streams back to back, stream s = blocks
 * [streamFirst[s], streamFirst[s+1]) (host array, ascending).  Blocks outside
 * every stream are decoded on their own.  Otherwise as mi355lz4_decompress_batch. */
int mi355lz4_decompress_streams(mi355lz4_ctx *ctx, const uint8_t *framedIn, size_t inLen, int headerKind,
                                int fixedUncomp, const int32_t *streamFirst, int nStreams, uint8_t *out,
                                size_t cap, size_t *outLen, int32_t *blockLen, int maxBlocks, int *nBlocks);

/* ---- several GPUs behind one handle, one process --------
 * A host caller is bound by PCIe (one link per GPU), so for the Haskell process -- one process, host buffers -- more GPUs
 * is more links.  A multi handle owns one engine per entry of devices[] (the same device may be named more than once).
 * The two calls below take the arguments of mi355lz4_compress_batch / mi355lz4_decompress_batch (independent blocks:
 * linked = 0, no dictionary) and give the same results, byte for byte: the batch is cut into one contiguous block range
 * per device (equal shares of the uncompressed bytes), every range goes through its engine's host-buffer call on a host
 * thread of its own, and the results lie in order in the caller's one buffer -- a host consumer needs no gather.
 * (Between ranks the north star's block i -> GPU i mod n is used, with a kernel on the root that interleaves the ranks'
 * outputs, gather.py; towards host memory that would make every device-to-host copy one copy per block.)
 * A handle is not thread-safe (one call at a time), like an engine.  mi355lz4_multi_last_error is thread-local.
 * replaces: compressChunk / decompressChunk, Internal/LZ4.hs:226-281, :291-336, one FFI call per batch. */
typedef struct mi355lz4_multi mi355lz4_multi;
int mi355lz4_create_multi(mi355lz4_multi **out, const int *devices, int n);
void mi355lz4_destroy_multi(mi355lz4_multi *m);
int mi355lz4_multi_device_count(const mi355lz4_multi *m);
/* engine i of the handle (e.g. for mi355lz4_set_decoder); owned by the handle */
mi355lz4_ctx *mi355lz4_multi_engine(mi355lz4_multi *m, int i);
const char *mi355lz4_multi_last_error(void);
int mi355lz4_multi_compress_batch(mi355lz4_multi *m, const uint8_t *const *src, const int32_t *srcLen, int nBlocks,
                                  int accel, int headerKind, uint8_t *framedOut, size_t cap, size_t *outLen,
                                  int32_t *blockFramedLen, int32_t *status);
int mi355lz4_multi_decompress_batch(mi355lz4_multi *m, const uint8_t *framedIn, size_t inLen, int headerKind,
                                    int fixedUncomp, uint8_t *out, size_t cap, size_t *outLen, int32_t *blockLen,
                                    int maxBlocks, int *nBlocks);

/* ---- one linked stream over several GPUs ----------
 * A linked stream does not shard by round-robin: block k's dictionary is the
 * output of block k-1 (cbits/lz4.c:2347-2355).  It shards by CONTIGUOUS RANGES:
 * engine r decodes blocks [b_r, b_r+1) and needs one thing from engine r-1, the
 * output of block b_r - 1 (the seam: at most 64 KiB of it are ever read).
 *
 * _begin issues everything that does not READ that output: the standalone pass,
 * the tolerant re-decode of the dependent blocks, source pointers and pointer
 * jumping over the range (where a byte comes from depends on tokens only).
 * _end issues the rest: the bytes are fetched from their roots, the first of
 * which lie in the seam, and the results are set.  Between the two calls the
 * caller places the seam at out + outOff[-1] (its size in result[-1]) in the
 * engine's stream order.  lookBack = 1 says outOff[-1] / result[-1] exist (the
 * arrays handed in point at their second element); lookBack = 0: the range
 * starts the stream.  Otherwise the arguments are those of
 * mi355lz4_decompress_batch_device with linked != 0, and so are the results.
 * Spread over G engines the serial part of a stream is G fetches, not G ranges
 * (streamly_lz4_amd/linked_shard.py drives it over torch.distributed).
 * A range whose dependent blocks do not fit one pointer segment
 * (MI355LZ4_LINKED_PTR_BLOCKS, default 4096 blocks of 64 KiB) is correct but
 * leaves all its work to _end. */
int mi355lz4_decompress_linked_begin(mi355lz4_ctx *ctx, const uint8_t *framed, uint64_t framedLen,
                                     const uint64_t *blockOff, int nBlocks, int headerKind, int fixedUncomp,
                                     uint8_t *out, const uint64_t *outOff, const int32_t *outCap,
                                     int32_t *result, int lookBack);
int mi355lz4_decompress_linked_end(mi355lz4_ctx *ctx);
/* Between _begin and _end: make the LAST block of the range final ahead of the others, so that it can go to the rank
 * that holds the next range while this rank's own fetch is still to run (a stream over G GPUs then waits G times for
 * one block, not for a range).  The seam -- the output of the block in front of the range (lookBack) -- must be in
 * place before this call, exactly as for _end: the fetch reads it.  Returns 1 when the last block's BYTES are final
 * after this call (the call waits for them), 0 when they are not available this way -- call _end first -- or a
 * negative MI355LZ4_E_* code.  Only the bytes are final: result[nBlocks-1] keeps the first pass's code until _end has
 * run, so take the block's size from its header (or from the capacity handed in), not from result[].  _end must still
 * be called.  (Reference semantics as for _begin: cbits/lz4.c:2347-2355.) */
int mi355lz4_decompress_linked_end_last(mi355lz4_ctx *ctx);

/* ---- synthetic inputs (bench / test support; SURVEY.md 8d generators) ---
 * kind: 0 = xorshift64* random, 1 = lzsynth(litMax, offMax), 2 = text-like.
 * Block i of the batch is seeded by (firstBlock + i * blockStep); it is written
 * at dst + i*blockLen.  dst is a device pointer. */
int mi355lz4_generate_device(mi355lz4_ctx *ctx, int kind, uint8_t *dst, int blockLen, int nBlocks,
                             uint64_t firstBlock, uint64_t blockStep, uint32_t litMax, uint32_t offMax);

/* ---- ordered multi-GPU gather support -------------------
 * Scatter rank-local dense blocks into the global stream on the root:
 * local block j of rank g (nRanks ranks, round-robin) is global block
 * j*nRanks+g; its bytes local[localOff[j] .. localOff[j+1]) are copied to
 * global + globalOff[j*nRanks+g].  All pointers are device pointers. */
int mi355lz4_interleave_device(mi355lz4_ctx *ctx, const uint8_t *local, const uint64_t *localOff,
                               int nLocalBlocks, int rank, int nRanks, uint8_t *global,
                               const uint64_t *globalOff);

/* ---- timing support: HIP events on the engine's own stream ------------- */
int mi355lz4_event_create(void **ev);
int mi355lz4_event_destroy(void *ev);
int mi355lz4_event_record(mi355lz4_ctx *ctx, void *ev);
/* Blocks until `stop` completed; *ms = elapsed milliseconds start -> stop. */
int mi355lz4_event_elapsed_ms(void *start, void *stop, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* MI355LZ4_H */
