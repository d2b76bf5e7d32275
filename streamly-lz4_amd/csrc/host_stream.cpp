// host_stream.cpp -- C++ mirror of Streamly.LZ4 / Streamly.Internal.LZ4 stream
// combinators (include/streamly_lz4.hpp) over the batched GPU ABI, plus a small
// C surface (slz4_*) so that tests and other languages can drive them.
//
// Host logic only (state machines, header bookkeeping, batching).  Every codec
// byte is produced by the GPU through mi355lz4_compress_batch /
// mi355lz4_decompress_batch.  Reference line numbers refer to
// src/Streamly/Internal/LZ4.hs unless stated otherwise.
#include "../../include/streamly_lz4.hpp"
#include "../../include/mi355lz4.h"

#include <cstring>
#include <deque>
#include <mutex>
#include <string>

namespace streamly_lz4 {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
static int32_t le32(const uint8_t *p)
{
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

int metaSize(const BlockConfig &c) { return c.blockSize == BlockSize::BlockHasSize ? 8 : 4; }   // :177-181

int maxBlockSize(const BlockConfig &c)                                                           // :275-281
{
    switch (c.blockSize) {
    case BlockSize::BlockHasSize: return MI355LZ4_MAX_INPUT_SIZE;
    case BlockSize::BlockMax64KB: return 64 * 1024;
    case BlockSize::BlockMax256KB: return 256 * 1024;
    case BlockSize::BlockMax1MB: return 1024 * 1024;
    case BlockSize::BlockMax4MB: return 4 * 1024 * 1024;
    }
    return 0;
}

static int fixedUncompSize(const BlockConfig &c)                                                 // :189-198
{
    return c.blockSize == BlockSize::BlockHasSize ? 0 : maxBlockSize(c);
}

namespace {
class ListStream : public ArrayStream {
public:
    explicit ListStream(std::vector<Array> a) : arrays_(std::move(a)) {}
    bool next(Array &out) override
    {
        if (pos_ >= arrays_.size()) return false;
        out = std::move(arrays_[pos_++]);
        return true;
    }
private:
    std::vector<Array> arrays_;
    size_t pos_ = 0;
};
} // namespace

StreamPtr fromList(std::vector<Array> arrays) { return StreamPtr(new ListStream(std::move(arrays))); }

std::vector<Array> toList(ArrayStream &s)
{
    std::vector<Array> out;
    Array a;
    while (s.next(a)) out.push_back(std::move(a));
    return out;
}

// ---------------------------------------------------------------------------
// Engine
// ---------------------------------------------------------------------------
Engine::Engine(int device, size_t batchBlocks) : batch_(batchBlocks ? batchBlocks : 1)
{
    if (mi355lz4_create(&ctx_, device) != MI355LZ4_OK)
        throw Error(std::string("streamly_lz4::Engine: ") + mi355lz4_last_error());
}
void Engine::setLinkedCompress(bool on) { mi355lz4_set_linked_compress(ctx_, on ? 1 : 0); linked_ = on; }

Engine::~Engine() { mi355lz4_destroy(ctx_); }

// ---------------------------------------------------------------------------
// compressChunksD (:353-394) + compressChunk (:226-281)
// ---------------------------------------------------------------------------
namespace {
class CompressStream : public ArrayStream {
public:
    CompressStream(BlockConfig cfg, int speed, StreamPtr in, Engine &eng)
        : cfg_(cfg), speed_(speed < 0 ? 0 : speed) /* speed = max speed0 0, :364 */, in_(std::move(in)), eng_(eng) {}

    bool next(Array &out) override
    {
        if (ready_.empty() && !done_) fill();
        if (ready_.empty()) return false;
        out = std::move(ready_.front());
        ready_.pop_front();
        return true;
    }

private:
    void fill()
    {
        std::vector<Array> batch;
        Array a;
        size_t bytes = 0;
        while (batch.size() < eng_.batchBlocks() && bytes < (size_t)1 << 30) {
            if (!in_->next(a)) { done_ = true; break; }
            if (a.size() >= ((size_t)2 << 30))                                   // :384-385
                throw Error("compressChunksD: Array element > 2 GB encountered");
            if (a.size() > (size_t)maxBlockSize(cfg_))                          // :237-241
                throw Error("compressChunk: Source array length " + std::to_string(a.size()) +
                            " exceeds the maximum block size of " + std::to_string(maxBlockSize(cfg_)));
            bytes += a.size();
            batch.push_back(std::move(a));
        }
        if (batch.empty()) return;
        const int n = (int)batch.size();
        const int meta = metaSize(cfg_);
        std::vector<const uint8_t *> ptrs((size_t)n);
        std::vector<int32_t> lens((size_t)n), flen((size_t)n), status((size_t)n);
        size_t cap = 0;
        for (int i = 0; i < n; i++) {
            ptrs[(size_t)i] = batch[(size_t)i].data();
            lens[(size_t)i] = (int32_t)batch[(size_t)i].size();
            cap += (size_t)mi355lz4_compress_bound(lens[(size_t)i]) + (size_t)meta;   // :244-251
        }
        Array framed(cap);
        size_t outLen = 0;
        int r = mi355lz4_compress_batch(eng_.ctx(), ptrs.data(), lens.data(), n, speed_, meta, framed.data(), cap,
                                        &outLen, flen.data(), status.data());
        if (r != MI355LZ4_OK) {
            // status[] is only filled when the call got as far as the per-block results
            for (int i = 0; r == MI355LZ4_E_BLOCK && i < n; i++)
                if (status[(size_t)i] <= 0)                                     // :257-260
                    throw Error("compressChunk: c_compressFastContinue failed. uncompLenC: " +
                                std::to_string(lens[(size_t)i]) + "compLenC: " + std::to_string(status[(size_t)i]));
            throw Error(std::string("compressChunks: ") + mi355lz4_last_error());
        }
        size_t pos = 0;
        for (int i = 0; i < n; i++) {                                           // one Array per block, :261-268
            ready_.emplace_back(framed.begin() + (long)pos, framed.begin() + (long)(pos + (size_t)flen[(size_t)i]));
            pos += (size_t)flen[(size_t)i];
        }
    }

    BlockConfig cfg_;
    int speed_;
    StreamPtr in_;
    Engine &eng_;
    std::deque<Array> ready_;
    bool done_ = false;
};
} // namespace

StreamPtr compressChunks(const BlockConfig &cfg, int speed, StreamPtr in, Engine &eng)
{
    return StreamPtr(new CompressStream(cfg, speed, std::move(in), eng));
}

// ---------------------------------------------------------------------------
// resizeChunksD (:432-523)
// ---------------------------------------------------------------------------
namespace {
class ResizeStream : public ArrayStream {
public:
    ResizeStream(BlockConfig cfg, FrameConfig conf, StreamPtr in) : cfg_(cfg), conf_(conf), in_(std::move(in)) {}

    bool next(Array &out) override
    {
        const size_t meta = (size_t)metaSize(cfg_);
        for (;;) {
            switch (st_) {
            case St::Init: {                                                    // RInit, :488-496
                if (!in_->next(buf_)) {
                    if (conf_.hasEndMark) throw Error("resizeChunksD: No end mark found");   // :495
                    st_ = St::Done;
                    return false;
                }
                st_ = St::Process;
                break;
            }
            case St::Process: {                                                 // process, :459-484
                const size_t len = buf_.size();
                if (len < 4) { st_ = St::Accumulate; break; }                   // :461-462
                if (conf_.hasEndMark && le32(buf_.data()) == 0) { st_ = St::Footer; break; }   // :451-456,464-466
                if (len <= meta) { st_ = St::Accumulate; break; }               // :468-469
                const int32_t compressedSize = le32(buf_.data());               // :471-473 (compSizeOffset = 0)
                // a negative size would make `required` wrap; the reference would then mis-slice.
                if (compressedSize < 0) throw Error("resizeChunksD: negative compressed length in block header");
                const size_t required = (size_t)compressedSize + meta;          // :474
                if (len == required) {                                          // :475-476
                    out = std::move(buf_);
                    buf_.clear();
                    st_ = St::Init;
                    return true;
                }
                if (len < required) { st_ = St::Accumulate; break; }            // :477-478
                out.assign(buf_.begin(), buf_.begin() + (long)required);        // :479-484
                buf_.erase(buf_.begin(), buf_.begin() + (long)required);
                st_ = St::Process;
                return true;
            }
            case St::Accumulate: {                                              // RAccumulate, :498-505
                Array more;
                if (!in_->next(more)) throw Error("resizeChunksD: Incomplete block");   // :505
                buf_.insert(buf_.end(), more.begin(), more.end());              // Array.splice, :502
                st_ = St::Process;
                break;
            }
            case St::Footer: {                                                  // RFooter, :506-522
                const size_t footer = conf_.hasEndMark ? 4 : 0;                 // :404-408
                if (buf_.size() < footer) {
                    Array more;
                    if (!in_->next(more)) throw Error("resizeChunksD: Incomplete footer");  // :517
                    buf_.insert(buf_.end(), more.begin(), more.end());
                    break;
                }
                // validateFooter always succeeds (:410-411); anything after the mark is ignored (:507)
                st_ = St::Done;
                return false;
            }
            case St::Done:
                return false;
            }
        }
    }

private:
    enum class St { Init, Process, Accumulate, Footer, Done };
    BlockConfig cfg_;
    FrameConfig conf_;
    StreamPtr in_;
    Array buf_;
    St st_ = St::Init;
};
} // namespace

StreamPtr resizeChunks(const BlockConfig &cfg, const FrameConfig &conf, StreamPtr in)
{
    return StreamPtr(new ResizeStream(cfg, conf, std::move(in)));
}

// ---------------------------------------------------------------------------
// decompressChunksRawD (:539-567) + decompressChunk (:291-336)
// ---------------------------------------------------------------------------
namespace {
class DecompressStream : public ArrayStream {
public:
    DecompressStream(BlockConfig cfg, StreamPtr in, Engine &eng) : cfg_(cfg), in_(std::move(in)), eng_(eng) {}

    bool next(Array &out) override
    {
        if (ready_.empty() && !done_) fill();
        if (ready_.empty()) return false;
        out = std::move(ready_.front());
        ready_.pop_front();
        return true;
    }

private:
    void fill()
    {
        const int meta = metaSize(cfg_);
        std::vector<Array> batch;
        Array a;
        size_t bytes = 0, outBytes = 0;
        while (batch.size() < eng_.batchBlocks() && bytes < (size_t)1 << 30 && outBytes < (size_t)2 << 30) {
            if (!in_->next(a)) { done_ = true; break; }
            // decompressChunk's header checks, :299-318
            if (a.size() < (size_t)meta) throw Error("decompressChunk: input array is shorter than the block header");
            const int32_t compLen = le32(a.data());
            const int64_t arrDataLen = (int64_t)a.size() - meta;
            const int64_t uncompLen = (meta == 8) ? (int64_t)le32(a.data() + 4) : (int64_t)fixedUncompSize(cfg_);
            if (compLen <= 0) throw Error("decompressChunk: compressed data length > 2GB");              // :309-310
            if ((int64_t)compLen < arrDataLen)                                                          // :311-315
                throw Error("decompressChunk: input array data length " + std::to_string(arrDataLen) +
                            " is less than the compressed data length specified in the header " + std::to_string(compLen));
            if ((int64_t)compLen > arrDataLen)   // the case the reference misses (it would read past the array)
                throw Error("decompressChunk: input array data length " + std::to_string(arrDataLen) +
                            " is shorter than the compressed data length specified in the header " + std::to_string(compLen));
            if (compLen > mi355lz4_compress_bound(MI355LZ4_MAX_INPUT_SIZE))                                 // :316-318
                throw Error("decompressChunk: compressed data length is more than the max limit: " +
                            std::to_string(mi355lz4_compress_bound(MI355LZ4_MAX_INPUT_SIZE)));
            if (uncompLen < 0) throw Error("decompressChunk: negative uncompressed length in block header");
            bytes += a.size();
            outBytes += (size_t)uncompLen;
            batch.push_back(std::move(a));
        }
        if (batch.empty()) return;
        const int n = (int)batch.size();
        Array framed;
        framed.reserve(bytes);
        for (auto &b : batch) framed.insert(framed.end(), b.begin(), b.end());
        Array out(outBytes ? outBytes : 1);
        std::vector<int32_t> blockLen((size_t)n);
        size_t outLen = 0;
        int got = 0;
        // The reference always decodes with stream (linked) semantics; prev_ is the array its
        // DecompressDo state keeps alive (:564).
        int r = mi355lz4_decompress_batch(eng_.ctx(), framed.data(), framed.size(), meta, fixedUncompSize(cfg_), 1,
                                          prev_.empty() ? nullptr : prev_.data(), (int)prev_.size(), out.data(),
                                          out.size(), &outLen, blockLen.data(), n, &got);
        if (r == MI355LZ4_E_BLOCK) {
            size_t pos = 0;
            for (int i = 0; i < got; i++) {
                const int32_t compLen = le32(framed.data() + pos);
                if (blockLen[(size_t)i] < 0)                                                             // :325-330
                    throw Error("decompressChunk: c_decompressSafeContinue failed. \narrDataLen = " +
                                std::to_string(compLen) + "\ncompLenC = " + std::to_string(compLen) +
                                "\nuncompLenC = " + std::to_string(meta == 8 ? le32(framed.data() + pos + 4) : fixedUncompSize(cfg_)) +
                                "\ndecompLenC = " + std::to_string(blockLen[(size_t)i]));
                pos += (size_t)meta + (size_t)compLen;
            }
        }
        if (r != MI355LZ4_OK) throw Error(std::string("decompressChunks: ") + mi355lz4_last_error());
        size_t pos = 0;
        for (int i = 0; i < n; i++) {                                           // one Array per block, :331-336
            const size_t len = (size_t)blockLen[(size_t)i];
            ready_.emplace_back(out.begin() + (long)pos, out.begin() + (long)(pos + len));
            if (len > 0) prev_ = ready_.back();                                 // cbits/lz4.c:2331,2353: only result > 0 moves the dictionary
            pos += len;
        }
    }

    BlockConfig cfg_;
    StreamPtr in_;
    Engine &eng_;
    std::deque<Array> ready_;
    Array prev_;
    bool done_ = false;
};
} // namespace

StreamPtr decompressChunksRaw(const BlockConfig &cfg, StreamPtr in, Engine &eng)
{
    return StreamPtr(new DecompressStream(cfg, std::move(in), eng));
}

StreamPtr decompressChunks(const BlockConfig &cfg, StreamPtr in, Engine &eng)                   // LZ4.hs:114-122
{
    return decompressChunksRaw(cfg, resizeChunks(cfg, defaultFrameConfig(), std::move(in)), eng);
}

// Result buffers of the batch form are recycled: a fresh 10 MiB allocation costs its page faults (0.3-0.6 ms, as much
// as the GPU call) every time; a buffer handed back by slz4_arrays_free / ArrayBatch's owner is kept for the next call.
namespace {
struct BufPool {
    std::mutex mu;
    std::vector<std::pair<size_t, uint8_t *>> free_;       // (capacity, buffer)
    uint8_t *get(size_t n, size_t &cap)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].first >= n && free_[i].first <= 2 * n + (1u << 20)) {
                    uint8_t *p = free_[i].second; cap = free_[i].first;
                    held -= cap;
                    free_.erase(free_.begin() + (long)i);
                    return p;
                }
        }
        cap = n;
        return new uint8_t[n];
    }
    // The pool holds at most four buffers and POOL_BYTES in all (round 3 kept up to 4 x 256 MiB for the life of the
    // process); slz4_trim() / trim() hands everything back.
    static constexpr size_t POOL_BYTES = (size_t)64 << 20;
    size_t held = 0;
    void put(uint8_t *p, size_t cap)
    {
        std::lock_guard<std::mutex> g(mu);
        if (free_.size() < 4 && held + cap <= POOL_BYTES) { free_.emplace_back(cap, p); held += cap; return; }
        delete[] p;
    }
    void trim()
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto &f : free_) delete[] f.second;
        free_.clear();
        held = 0;
    }
};
BufPool &buf_pool() { static BufPool *p = new BufPool(); return *p; }
} // namespace
void ArrayBatch::release() { if (buf) { buf_pool().put(buf, cap); buf = nullptr; cap = 0; } }
void trimBuffers() { buf_pool().trim(); }

// decompressChunks over arrays that already lie back to back (streamly_lz4.hpp): when the bytes are a well-formed
// run of whole blocks that all decode, one index walk and one GPU call do what resizeChunksD + decompressChunksRawD do
// array by array.  Anything else -- a trailing partial block, an end mark, a block the decoder rejects -- goes through
// the combinators themselves, which is what yields the reference's error for it.
ArrayBatch decompressChunksBatch(const BlockConfig &cfg, const FrameConfig &conf, const uint8_t *data,
                                 const uint64_t *lens, size_t n, Engine &eng)
{
    size_t total = 0;
    for (size_t i = 0; i < n; i++) total += (size_t)lens[i];
    ArrayBatch res;
    const int meta = metaSize(cfg);
    if (!conf.hasEndMark && total > 0 && total < ((size_t)1 << 31)) {
        // how many whole blocks? (a header walk: compLen <= 0, or a block that runs past the end, ends the fast path)
        size_t nbk = 0, p = 0;
        bool whole = true;
        while (p < total) {
            if (total - p < (size_t)meta) { whole = false; break; }
            const int32_t compLen = le32(data + p);
            if (compLen <= 0 || (size_t)compLen > total - p - (size_t)meta) { whole = false; break; }
            p += (size_t)meta + (size_t)compLen;
            nbk++;
        }
        if (whole && nbk > 0 && nbk < ((size_t)1 << 24)) {
            std::vector<uint64_t> boff(nbk + 1);
            std::vector<int32_t> ulen(nbk + 1), blen(nbk + 1);
            int nb = 0;
            if (mi355lz4_index_host(data, total, meta, fixedUncompSize(cfg), boff.data(), ulen.data(), (int)nbk, &nb) ==
                    MI355LZ4_OK && (size_t)nb == nbk) {
                size_t cap = 0;
                bool sane = true;
                for (int i = 0; i < nb; i++) { if (ulen[(size_t)i] < 0) sane = false; else cap += (size_t)ulen[(size_t)i]; }
                if (sane && cap < ((size_t)1 << 32)) {
                    res.buf = buf_pool().get(cap + 16, res.cap);
                    size_t outLen = 0;
                    int got = 0;
                    const int r = mi355lz4_decompress_batch(eng.ctx(), data, total, meta, fixedUncompSize(cfg), 1, nullptr, 0,
                                                            res.buf, cap + 16, &outLen, blen.data(), nb, &got);
                    if (r == MI355LZ4_OK && got == nb) {
                        res.off.resize((size_t)nb + 1);
                        size_t pos = 0;
                        for (int i = 0; i < nb; i++) { res.off[(size_t)i] = pos; pos += (size_t)blen[(size_t)i]; }
                        res.off[(size_t)nb] = pos;
                        return res;
                    }
                    // A block the decoder rejected: the arrays are decoded a second time below, by the combinators, because
                    // that is what raises the reference's error at the reference's block (the batch call only says that some
                    // block failed).  The double decode is paid on the error path only.
                    res.release();
                }
            }
        }
    }
    // the general path: the combinators, array by array
    std::vector<Array> v;
    v.reserve(n);
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) { v.emplace_back(data + pos, data + pos + lens[i]); pos += (size_t)lens[i]; }
    StreamPtr s = decompressChunksRaw(cfg, resizeChunks(cfg, conf, fromList(std::move(v))), eng);
    std::vector<Array> outv = toList(*s);
    size_t outTotal = 0;
    for (const Array &a : outv) outTotal += a.size();
    res.buf = buf_pool().get(outTotal + 1, res.cap);
    res.off.resize(outv.size() + 1);
    pos = 0;
    for (size_t i = 0; i < outv.size(); i++) {
        res.off[i] = pos;
        if (!outv[i].empty()) memcpy(res.buf + pos, outv[i].data(), outv[i].size());
        pos += outv[i].size();
    }
    res.off[outv.size()] = pos;
    return res;
}

// ---------------------------------------------------------------------------
// simpleFrameParserD (:590-651) and decompressChunksWithD (:569-577)
// ---------------------------------------------------------------------------
namespace {
// Puts an already-pulled array back in front of a stream.
class PrependStream : public ArrayStream {
public:
    PrependStream(Array first, StreamPtr rest) : first_(std::move(first)), rest_(std::move(rest)), hasFirst_(!first_.empty()) {}
    bool next(Array &out) override
    {
        if (hasFirst_) { out = std::move(first_); hasFirst_ = false; return true; }
        return rest_->next(out);
    }
private:
    Array first_;
    StreamPtr rest_;
    bool hasFirst_;
};
} // namespace

std::pair<std::pair<BlockConfig, FrameConfig>, StreamPtr> simpleFrameParser(StreamPtr in)
{
    // gather the 7 header bytes: magic(4) FLG(1) BD(1) HC(1)
    Array head, a;
    while (head.size() < 7) {
        if (!in->next(a)) throw Error("simpleFrameParserD: unexpected end of input in frame header");
        head.insert(head.end(), a.begin(), a.end());
    }
    const long magic = (long)head[0] | ((long)head[1] << 8) | ((long)head[2] << 16) | ((long)head[3] << 24);
    if (magic != 407708164L)                                                    // :607-618
        throw Error("The parsed magic " + std::to_string(magic) + " does not match 407708164");
    const uint8_t flg = head[4];                                                // :620-642
    const bool isVersion01 = !(flg & 0x80) && (flg & 0x40);
    if (!isVersion01) throw Error("Version is not 01");
    if (flg & 0x20) throw Error("Block independence is not yet supported");
    if (flg & 0x10) throw Error("Block checksum is not yet supported");
    if (flg & 0x08) throw Error("Content size is not yet supported");
    if (flg & 0x04) throw Error("Content checksum is not yet supported");
    if (flg & 0x01) throw Error("Dict is not yet supported");
    BlockConfig cfg;
    switch (head[5] >> 4) {                                                     // :644-651
    case 4: cfg.blockSize = BlockSize::BlockMax64KB; break;
    case 5: cfg.blockSize = BlockSize::BlockMax256KB; break;
    case 6: cfg.blockSize = BlockSize::BlockMax1MB; break;
    case 7: cfg.blockSize = BlockSize::BlockMax4MB; break;
    default: throw Error("parseBD: Unknown block max size");
    }
    // head[6] is the header checksum: read, not verified (:605)
    FrameConfig fc;
    fc.hasEndMark = true;                                                       // :598-600
    Array rest(head.begin() + 7, head.end());
    return {{cfg, fc}, StreamPtr(new PrependStream(std::move(rest), std::move(in)))};
}

StreamPtr decompressChunksWith(StreamPtr in, Engine &eng)                       // :569-577
{
    auto parsed = simpleFrameParser(std::move(in));
    const BlockConfig cfg = parsed.first.first;
    const FrameConfig conf = parsed.first.second;
    return decompressChunksRaw(cfg, resizeChunks(cfg, conf, std::move(parsed.second)), eng);
}

} // namespace streamly_lz4

// ===========================================================================
// C surface over the combinators (tests / other-language bindings)
// ===========================================================================
using namespace streamly_lz4;

struct slz4_arrays { std::vector<Array> v; ArrayBatch b; bool flat = false; };
struct slz4_engine { Engine *e; };

static thread_local std::string g_slz4_err;

static BlockConfig cfg_from_kind(int kind)
{
    BlockConfig c;
    switch (kind) {
    case 0: c.blockSize = BlockSize::BlockHasSize; break;
    case 1: c.blockSize = BlockSize::BlockMax64KB; break;
    case 2: c.blockSize = BlockSize::BlockMax256KB; break;
    case 3: c.blockSize = BlockSize::BlockMax1MB; break;
    case 4: c.blockSize = BlockSize::BlockMax4MB; break;
    default: throw Error("unknown BlockSize kind");
    }
    return c;
}

static StreamPtr list_from_c(const uint8_t *data, const uint64_t *lens, size_t n)
{
    std::vector<Array> v;
    v.reserve(n);
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        v.emplace_back(data + pos, data + pos + lens[i]);
        pos += lens[i];
    }
    return fromList(std::move(v));
}

template <typename F>
static int guarded(slz4_arrays **out, F f)
{
    try {
        StreamPtr s = f();
        slz4_arrays *r = new slz4_arrays();
        r->v = toList(*s);
        *out = r;
        return 0;
    } catch (const std::exception &e) {
        g_slz4_err = e.what();
        *out = nullptr;
        return -1;
    }
}

extern "C" {

const char *slz4_last_error(void) { return g_slz4_err.c_str(); }

int slz4_engine_create(slz4_engine **out, int device, size_t batchBlocks)
{
    try {
        slz4_engine *h = new slz4_engine();
        h->e = new Engine(device, batchBlocks);
        *out = h;
        return 0;
    } catch (const std::exception &e) {
        g_slz4_err = e.what();
        *out = nullptr;
        return -1;
    }
}
void slz4_engine_destroy(slz4_engine *h) { if (h) { delete h->e; delete h; } }
void slz4_engine_set_batch(slz4_engine *h, size_t n) { if (h) h->e->setBatchBlocks(n); }
void slz4_engine_set_linked_compress(slz4_engine *h, int on) { if (h) h->e->setLinkedCompress(on != 0); }
mi355lz4_ctx *slz4_engine_ctx(slz4_engine *h) { return h ? h->e->ctx() : nullptr; }

size_t slz4_arrays_count(const slz4_arrays *a) { return a ? (a->flat ? a->b.count() : a->v.size()) : 0; }
size_t slz4_arrays_len(const slz4_arrays *a, size_t i) { return a->flat ? a->b.off[i + 1] - a->b.off[i] : a->v[i].size(); }
const uint8_t *slz4_arrays_data(const slz4_arrays *a, size_t i) { return a->flat ? a->b.buf + a->b.off[i] : a->v[i].data(); }
void slz4_arrays_free(slz4_arrays *a) { delete a; }
// result buffers kept for reuse (<= 64 MiB in all) go back to the allocator
void slz4_trim(void) { streamly_lz4::trimBuffers(); }
// flat results (the batch forms): the shared buffer and the n + 1 offsets; returns 0 when `a` is not flat
int slz4_arrays_flat(const slz4_arrays *a, const uint8_t **base, const size_t **offsets)
{
    if (!a || !a->flat) return 0;
    *base = a->b.buf;
    *offsets = a->b.off.data();
    return 1;
}

int slz4_compress_chunks(slz4_engine *h, int blockSizeKind, int speed, const uint8_t *data, const uint64_t *lens,
                         size_t n, slz4_arrays **out)
{
    return guarded(out, [&] { return compressChunks(cfg_from_kind(blockSizeKind), speed, list_from_c(data, lens, n), *h->e); });
}

int slz4_resize_chunks(int blockSizeKind, int hasEndMark, const uint8_t *data, const uint64_t *lens, size_t n,
                       slz4_arrays **out)
{
    return guarded(out, [&] {
        FrameConfig fc; fc.hasEndMark = hasEndMark != 0;
        return resizeChunks(cfg_from_kind(blockSizeKind), fc, list_from_c(data, lens, n));
    });
}

int slz4_decompress_chunks_raw(slz4_engine *h, int blockSizeKind, const uint8_t *data, const uint64_t *lens, size_t n,
                               slz4_arrays **out)
{
    return guarded(out, [&] { return decompressChunksRaw(cfg_from_kind(blockSizeKind), list_from_c(data, lens, n), *h->e); });
}

int slz4_decompress_chunks(slz4_engine *h, int blockSizeKind, int hasEndMark, const uint8_t *data, const uint64_t *lens,
                           size_t n, slz4_arrays **out)
{
    try {
        BlockConfig cfg = cfg_from_kind(blockSizeKind);
        FrameConfig fc; fc.hasEndMark = hasEndMark != 0;
        slz4_arrays *r = new slz4_arrays();
        try { r->b = decompressChunksBatch(cfg, fc, data, lens, n, *h->e); } catch (...) { delete r; throw; }
        r->flat = true;
        *out = r;
        return 0;
    } catch (const std::exception &e) {
        g_slz4_err = e.what();
        *out = nullptr;
        return -1;
    }
}

int slz4_decompress_chunks_with(slz4_engine *h, const uint8_t *data, const uint64_t *lens, size_t n, slz4_arrays **out)
{
    return guarded(out, [&] { return decompressChunksWith(list_from_c(data, lens, n), *h->e); });
}

// frame-header parser alone: returns BlockSize kind (1..4) or -1
int slz4_simple_frame_parser(const uint8_t *data, const uint64_t *lens, size_t n, int *hasEndMark, slz4_arrays **rest)
{
    int kind = -1;
    int r = guarded(rest, [&] {
        auto parsed = simpleFrameParser(list_from_c(data, lens, n));
        kind = (int)parsed.first.first.blockSize;
        if (hasEndMark) *hasEndMark = parsed.first.second.hasEndMark ? 1 : 0;
        return std::move(parsed.second);
    });
    return r == 0 ? kind : -1;
}

// ---- standard LZ4 frames (csrc/lz4_frame.cpp) ----
uint32_t slz4_xxh32(const uint8_t *data, size_t len, uint32_t seed) { return xxh32(data, len, seed); }

// flags: bit 0 block checksums, bit 1 content checksum, bit 2 content size, bit 3 linked blocks
int slz4_lz4frame_compress(slz4_engine *h, int blockSizeKind, int flags, int speed, const uint8_t *data, size_t len,
                           slz4_arrays **out)
{
    return guarded(out, [&] {
        Lz4FrameOptions o;
        o.blockMax = cfg_from_kind(blockSizeKind).blockSize;
        o.blockChecksum = flags & 1; o.contentChecksum = flags & 2; o.contentSize = flags & 4; o.linkedBlocks = flags & 8;
        std::vector<Array> v;
        v.push_back(lz4FrameCompress(Array(data, data + len), speed, *h->e, o));
        return fromList(std::move(v));
    });
}

int slz4_lz4frame_decompress(slz4_engine *h, const uint8_t *data, size_t len, slz4_arrays **out)
{
    return guarded(out, [&] {
        std::vector<Array> v;
        v.push_back(lz4FrameDecompress(Array(data, data + len), *h->e));
        return fromList(std::move(v));
    });
}

} // extern "C"
