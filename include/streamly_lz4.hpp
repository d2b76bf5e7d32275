// streamly_lz4.hpp -- C++ host-side mirror of the reference's operator interface
// for the LZ4 hot path (GHC is not available in the build image, and the
// reference host code is compiled Haskell, so the mirror above the C ABI is C++).
//
// Same names, argument meaning and error behaviour as
//   Streamly.LZ4                 (reference src/Streamly/LZ4.hs:94-122)
//   Streamly.Internal.LZ4        (reference src/Streamly/Internal/LZ4.hs:338-651)
//   Streamly.Internal.LZ4.Config (reference src/Streamly/Internal/LZ4/Config.hs)
// over a pull stream of byte arrays (the analogue of `SerialT m (Array Word8)`).
// Errors the reference raises with `error`/`Parser.die` are thrown as
// streamly_lz4::Error carrying the same message text.
//
// All codec arithmetic happens on the GPU through include/mi355lz4.h; the
// combinators batch `batchBlocks` arrays per call.  There is no CPU codec here.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

struct mi355lz4_ctx;

namespace streamly_lz4 {

using Array = std::vector<uint8_t>;                     // Array Word8

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

// Pull stream: next() fills `out` and returns true (Yield), or returns false (Stop).
class ArrayStream {
public:
    virtual ~ArrayStream() = default;
    virtual bool next(Array &out) = 0;
};
using StreamPtr = std::unique_ptr<ArrayStream>;

StreamPtr fromList(std::vector<Array> arrays);          // Stream.fromList
std::vector<Array> toList(ArrayStream &s);              // Stream.toList

// ---- Config.hs:109-136 ------------------------------------------------------
enum class BlockSize { BlockHasSize, BlockMax64KB, BlockMax256KB, BlockMax1MB, BlockMax4MB };
struct BlockConfig { BlockSize blockSize = BlockSize::BlockHasSize; };
struct FrameConfig { bool hasEndMark = false; };
inline BlockConfig defaultBlockConfig() { return BlockConfig{}; }                       // Config.hs:159-160
inline FrameConfig defaultFrameConfig() { return FrameConfig{}; }                       // Config.hs:100-103
inline BlockConfig setBlockMaxSize(BlockSize bs, BlockConfig c) { c.blockSize = bs; return c; }   // Config.hs:139-140
inline FrameConfig setFrameEndMark(bool v, FrameConfig c) { c.hasEndMark = v; return c; }         // Config.hs:78-79

int metaSize(const BlockConfig &c);                     // Internal/LZ4.hs:177-181
int maxBlockSize(const BlockConfig &c);                 // Internal/LZ4.hs:275-281

// GPU engine handle shared by the combinators.
class Engine {
public:
    explicit Engine(int device = 0, size_t batchBlocks = 4096);
    ~Engine();
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    mi355lz4_ctx *ctx() const { return ctx_; }
    size_t batchBlocks() const { return batch_; }
    void setBatchBlocks(size_t n) { batch_ = n ? n : 1; }
    // compressChunks then writes a LINKED stream like the reference's (each block's dictionary is the block before it
    // within a batch; Internal/LZ4.hs:376,389 keeps the previous chunk alive for exactly this); decompressChunks
    // reads either kind.  Default off: independent blocks.
    void setLinkedCompress(bool on);
    bool linkedCompress() const { return linked_; }
private:
    mi355lz4_ctx *ctx_ = nullptr;
    size_t batch_;
    bool linked_ = false;
};

// ---- Streamly.LZ4 / Streamly.Internal.LZ4 -------------------------------------
// compressChunks cfg speed          (LZ4.hs:94-100, Internal/LZ4.hs:353-394)
StreamPtr compressChunks(const BlockConfig &cfg, int speed, StreamPtr in, Engine &eng);
// resizeChunksD cfg conf            (Internal/LZ4.hs:432-523)
StreamPtr resizeChunks(const BlockConfig &cfg, const FrameConfig &conf, StreamPtr in);
// decompressChunksRawD cfg          (Internal/LZ4.hs:539-567)
StreamPtr decompressChunksRaw(const BlockConfig &cfg, StreamPtr in, Engine &eng);
// decompressChunks cfg = decompressChunksRawD cfg . resizeChunksD cfg defaultFrameConfig  (LZ4.hs:114-122)
StreamPtr decompressChunks(const BlockConfig &cfg, StreamPtr in, Engine &eng);
// decompressChunks over a BATCH of arrays that lie back to back in one buffer (array i = lens[i] bytes), the result as
// slices of ONE buffer.  Same results and same errors as decompressChunks over fromList of those arrays.  Why it
// exists: an `Array Word8` of the reference is a slice of a shared buffer, so its combinators move no bytes between
// stages; Array here is an owning vector and the array-at-a-time form above pays an allocation and a copy per array
// and per stage (1.1 ms for the reference's own 10 MiB benchmark files -- more than the GPU call).  This form makes
// the copies the reference would make: none on the way in, one (device -> host) on the way out.
struct ArrayBatch {
    uint8_t *buf = nullptr;             // the arrays' bytes, back to back (not zero-filled; recycled by release())
    size_t cap = 0;
    std::vector<size_t> off;            // array i = buf[off[i] .. off[i + 1]); size() = arrays + 1
    size_t count() const { return off.empty() ? 0 : off.size() - 1; }
    void release();                     // hands the buffer back (kept for the next call: no fresh page faults)
    ArrayBatch() = default;
    ArrayBatch(ArrayBatch &&o) noexcept : buf(o.buf), cap(o.cap), off(std::move(o.off)) { o.buf = nullptr; o.cap = 0; }
    ArrayBatch &operator=(ArrayBatch &&o) noexcept { release(); buf = o.buf; cap = o.cap; off = std::move(o.off); o.buf = nullptr; o.cap = 0; return *this; }
    ArrayBatch(const ArrayBatch &) = delete;
    ArrayBatch &operator=(const ArrayBatch &) = delete;
    ~ArrayBatch() { release(); }
};
ArrayBatch decompressChunksBatch(const BlockConfig &cfg, const FrameConfig &conf, const uint8_t *data,
                                 const uint64_t *lens, size_t n, Engine &eng);
// Released result buffers are kept for the next call (at most four, 64 MiB in all); this hands them back to the allocator.
void trimBuffers();
// simpleFrameParserD                (Internal/LZ4.hs:590-651): consumes the 7-byte
// frame header from the head of the stream; returns the parsed configs and the
// stream of what follows.
std::pair<std::pair<BlockConfig, FrameConfig>, StreamPtr> simpleFrameParser(StreamPtr in);
// decompressChunksWithD simpleFrameParserD  (Internal/LZ4.hs:569-577)
StreamPtr decompressChunksWith(StreamPtr in, Engine &eng);

// ---- the standard LZ4 frame format (interop with liblz4's LZ4F_* and the lz4 CLI) ----
// What simpleFrameParserD stops short of (Internal/LZ4.hs:631-638 rejects independent blocks, checksums and
// content size; :605 skips the header checksum): csrc/lz4_frame.cpp.
struct Lz4FrameOptions {
    BlockSize blockMax = BlockSize::BlockMax64KB;       // BD byte; one of BlockMax64KB .. BlockMax4MB
    bool blockChecksum = false;                         // xxh32 behind every block
    bool contentChecksum = true;                        // xxh32 of the content behind the end mark (the CLI's default)
    bool contentSize = false;                           // 8-byte content size in the descriptor
    bool linkedBlocks = false;                          // block-dependent frame (the lz4 tool's default): smaller on text
};
uint32_t xxh32(const uint8_t *p, size_t len, uint32_t seed);
// One frame (all blocks in one batch; independent blocks unless opt.linkedBlocks).
Array lz4FrameCompress(const Array &data, int speed, Engine &eng, const Lz4FrameOptions &opt = Lz4FrameOptions());
// Host half of the reader: parses ONE frame at frame[at...] (advancing at), verifies header and block checksums and
// re-frames the blocks for the engine ([compLen LE32][LZ4 block], stored blocks as literal-only blocks).
// Returns false for a skippable frame.
struct Lz4FrameIndex {
    bool independent = false, hasContentSize = false, hasContentChecksum = false;
    size_t blockMax = 0;
    uint64_t contentSize = 0;
    uint32_t contentChecksum = 0;
    Array framed;
    std::vector<size_t> blockAt;                        // offsets of the block headers in framed, plus its size
};
bool lz4FrameParse(const Array &frame, size_t &at, Lz4FrameIndex &ix);
// Any sequence of frames and skippable frames; linked or independent blocks, stored blocks, all checksums verified.
Array lz4FrameDecompress(const Array &frame, Engine &eng);

} // namespace streamly_lz4
