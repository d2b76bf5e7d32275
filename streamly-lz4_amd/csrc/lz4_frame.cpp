// lz4_frame.cpp -- the standard LZ4 frame format over the GPU block engine (second half).
//
// The reference's experimental frame parser (src/Streamly/Internal/LZ4.hs:590-651) reads the frame header of
// the LZ4 frame format but rejects every option the stock `lz4` tool uses by default: block independence
// (:631-632), block and content checksums (:633-638), content size (:635-636); it does not verify the header
// checksum (:605) and cannot read uncompressed blocks (compLen <= 0 is an error, :309-310).  This file is the
// interoperable counterpart: frames written here are read by any LZ4F decoder (liblz4's LZ4F_decompress, the
// `lz4` CLI), and frames written by those are read here -- independent or linked blocks, uncompressed blocks,
// block / content checksums (xxh32), content size, skippable frames, concatenated frames.
//
// Host side only: header and checksum handling, then ONE batched call into the engine for all blocks of a
// frame.  An uncompressed block is handed to the decoder as what it is in LZ4 terms, a block of literals only
// (token, length bytes, the bytes), so that it takes its place in the history of a linked frame without a
// special case on the device.
#include "../../include/mi355lz4.h"
#include "../../include/streamly_lz4.hpp"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace streamly_lz4 {

// ---------------------------------------------------------------------------
// xxHash32 (the checksum of the LZ4 frame format), from its published definition
// ---------------------------------------------------------------------------
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

uint32_t xxh32(const uint8_t *p, size_t len, uint32_t seed)
{
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const uint8_t *end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t *limit = end - 16;
        do {
            v1 = rotl32(v1 + rd32(p) * P2, 13) * P1;
            v2 = rotl32(v2 + rd32(p + 4) * P2, 13) * P1;
            v3 = rotl32(v3 + rd32(p + 8) * P2, 13) * P1;
            v4 = rotl32(v4 + rd32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (p <= limit);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl32(h + (uint32_t)(*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

static void put32(Array &a, uint32_t v) { for (int k = 0; k < 4; k++) a.push_back((uint8_t)(v >> (8 * k))); }

static int bd_code(BlockSize bs)
{
    switch (bs) {
    case BlockSize::BlockMax64KB: return 4;
    case BlockSize::BlockMax256KB: return 5;
    case BlockSize::BlockMax1MB: return 6;
    case BlockSize::BlockMax4MB: return 7;
    default: throw Error("lz4FrameCompress: the frame format needs one of BlockMax64KB .. BlockMax4MB");
    }
}
static size_t bd_size(int code) { return (size_t)1 << (8 + 2 * code); }       // 4 -> 64 KiB ... 7 -> 4 MiB

// ---------------------------------------------------------------------------
// writer
// ---------------------------------------------------------------------------
Array lz4FrameCompress(const Array &data, int speed, Engine &eng, const Lz4FrameOptions &opt)
{
    const int code = bd_code(opt.blockMax);
    const size_t bmax = bd_size(code);
    Array out;
    put32(out, 0x184D2204u);
    // version 01; blocks independent, or linked (every block's dictionary is the block before it: full blocks, so
    // that block IS the frame format's 64 KiB window -- what the lz4 tool writes by default)
    const uint8_t flg = (uint8_t)(0x40 | (opt.linkedBlocks ? 0 : 0x20) | (opt.blockChecksum ? 0x10 : 0) |
                                  (opt.contentSize ? 0x08 : 0) | (opt.contentChecksum ? 0x04 : 0));
    const size_t descAt = out.size();
    out.push_back(flg);
    out.push_back((uint8_t)(code << 4));
    if (opt.contentSize) { const uint64_t n = data.size(); for (int k = 0; k < 8; k++) out.push_back((uint8_t)(n >> (8 * k))); }
    out.push_back((uint8_t)(xxh32(out.data() + descAt, out.size() - descAt, 0) >> 8));

    const size_t nb = (data.size() + bmax - 1) / bmax;
    if (nb > 0) {
        if (nb > 0x7fffffffu) throw Error("lz4FrameCompress: too many blocks");
        std::vector<const uint8_t *> ptrs(nb);
        std::vector<int32_t> lens(nb), flen(nb), status(nb);
        size_t cap = 0;
        for (size_t i = 0; i < nb; i++) {
            ptrs[i] = data.data() + i * bmax;
            lens[i] = (int32_t)std::min(bmax, data.size() - i * bmax);
            cap += (size_t)mi355lz4_compress_bound(lens[i]) + 4;
        }
        Array framed(cap);
        size_t outLen = 0;
        const bool was = eng.linkedCompress();
        eng.setLinkedCompress(opt.linkedBlocks);
        const int r = mi355lz4_compress_batch(eng.ctx(), ptrs.data(), lens.data(), (int)nb, speed < 0 ? 0 : speed, 4,
                                              framed.data(), cap, &outLen, flen.data(), status.data());
        eng.setLinkedCompress(was);
        if (r != MI355LZ4_OK) throw Error(std::string("lz4FrameCompress: ") + mi355lz4_last_error());
        size_t pos = 0;
        for (size_t i = 0; i < nb; i++) {
            const uint32_t c = (uint32_t)flen[i] - 4u;                          // engine framing: [compLen LE32][data]
            const uint8_t *body = framed.data() + pos + 4;
            const uint8_t *payload;
            uint32_t n;
            if (c >= (uint32_t)lens[i]) {                                       // did not shrink: stored, high bit set
                payload = ptrs[i]; n = (uint32_t)lens[i];
                put32(out, n | 0x80000000u);
            } else {
                payload = body; n = c;
                put32(out, n);
            }
            out.insert(out.end(), payload, payload + n);
            if (opt.blockChecksum) put32(out, xxh32(payload, n, 0));
            pos += (size_t)flen[i];
        }
    }
    put32(out, 0);                                                              // end mark
    if (opt.contentChecksum) put32(out, xxh32(data.data(), data.size(), 0));
    return out;
}

// ---------------------------------------------------------------------------
// reader
// ---------------------------------------------------------------------------
namespace {
struct Cursor {
    const uint8_t *p;
    size_t n, at;
    void need(size_t k, const char *what) const { if (n - at < k) throw Error(std::string("lz4FrameDecompress: truncated ") + what); }
    uint32_t u32(const char *what) { need(4, what); const uint32_t v = rd32(p + at); at += 4; return v; }
    uint8_t u8(const char *what) { need(1, what); return p[at++]; }
};

// an uncompressed block as an LZ4 block: one sequence of literals only (cbits/lz4.c:214-235 encoding rules)
void literal_block(Array &dst, const uint8_t *src, size_t n)
{
    if (n < 15) dst.push_back((uint8_t)(n << 4));
    else {
        dst.push_back(0xF0);
        size_t rest = n - 15;
        while (rest >= 255) { dst.push_back(255); rest -= 255; }
        dst.push_back((uint8_t)rest);
    }
    dst.insert(dst.end(), src, src + n);
}
} // namespace

// One frame (or skippable frame) starting at `at`: header and block checksums verified, blocks re-framed for the
// engine.  No device work: this is the part that reads untrusted bytes, and it runs under the host sanitizers.
bool lz4FrameParse(const Array &frame, size_t &at, Lz4FrameIndex &ix)
{
    Cursor c{frame.data(), frame.size(), at};
    ix = Lz4FrameIndex();
    const uint32_t magic = c.u32("magic number");
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                                 // skippable frame
        const uint32_t len = c.u32("skippable frame size");
        c.need(len, "skippable frame");
        at = c.at + len;
        return false;
    }
    if (magic != 0x184D2204u) throw Error("lz4FrameDecompress: bad magic number " + std::to_string(magic));
    const size_t descAt = c.at;
    const uint8_t flg = c.u8("frame descriptor"), bd = c.u8("frame descriptor");
    if ((flg >> 6) != 1) throw Error("lz4FrameDecompress: frame version is not 01");
    if (flg & 0x02) throw Error("lz4FrameDecompress: reserved FLG bit set");
    const bool blockSum = flg & 0x10;
    ix.independent = flg & 0x20; ix.hasContentSize = flg & 0x08; ix.hasContentChecksum = flg & 0x04;
    if (flg & 0x01) throw Error("lz4FrameDecompress: frames that need a dictionary (DictID) are not supported");
    const int code = (bd >> 4) & 7;
    if (code < 4 || (bd & 0x8F)) throw Error("lz4FrameDecompress: bad BD byte");
    const size_t bmax = ix.blockMax = bd_size(code);
    if (ix.hasContentSize) { c.need(8, "content size"); for (int k = 0; k < 8; k++) ix.contentSize |= (uint64_t)c.p[c.at + k] << (8 * k); c.at += 8; }
    const uint8_t hc = c.u8("header checksum");
    if (hc != (uint8_t)(xxh32(c.p + descAt, c.at - 1 - descAt, 0) >> 8)) throw Error("lz4FrameDecompress: header checksum mismatch");

    // the blocks of this frame, in the engine's framing with 4-byte headers: [compLen LE32][LZ4 block]
    Array &framed = ix.framed;
    std::vector<size_t> &blockAt = ix.blockAt;
    for (;;) {
        const uint32_t word = c.u32("block size");
        if (word == 0) break;                                                   // end mark
        const uint32_t n = word & 0x7FFFFFFFu;
        const bool stored = word & 0x80000000u;
        if (n > bmax) throw Error("lz4FrameDecompress: block larger than the frame's maximum block size");
        c.need(n, "block");
        const uint8_t *payload = c.p + c.at;
        c.at += n;
        if (blockSum && c.u32("block checksum") != xxh32(payload, n, 0)) throw Error("lz4FrameDecompress: block checksum mismatch");
        if (n == 0) continue;                                                   // a stored block of nothing
        const size_t hdrAt = framed.size();
        blockAt.push_back(hdrAt);
        put32(framed, 0);
        if (stored) literal_block(framed, payload, n);
        else framed.insert(framed.end(), payload, payload + n);
        const uint32_t clen = (uint32_t)(framed.size() - hdrAt - 4);
        for (int k = 0; k < 4; k++) framed[hdrAt + (size_t)k] = (uint8_t)(clen >> (8 * k));
    }
    blockAt.push_back(framed.size());
    if (ix.hasContentChecksum) ix.contentChecksum = c.u32("content checksum");
    at = c.at;
    return true;
}

// Blocks are decoded in groups under a byte budget, so that the memory a frame can make this function reserve is
// bounded by what it actually decodes to plus one budget (a few bytes of input can NAME 4 MiB of output per block:
// sizing the output by blocks x maximum block size, as round 2 did, let a 100 KB frame ask for tens of GB on the host
// and on the device).  A block's capacity is also capped by what its compressed bytes can expand to (255 per byte).
#ifndef LZ4F_GROUP_BUDGET
#define LZ4F_GROUP_BUDGET ((size_t)256 << 20)
#endif
Array lz4FrameDecompress(const Array &frame, Engine &eng)
{
    Array out;
    Array scratch;                      // one group's output; grows to the budget at most, reused, never zero-filled twice
    size_t at = 0;
    Lz4FrameIndex ix;
    while (at < frame.size()) {
        if (!lz4FrameParse(frame, at, ix)) continue;
        const bool independent = ix.independent;
        const size_t bmax = ix.blockMax;
        const Array &framed = ix.framed;
        const std::vector<size_t> &blockAt = ix.blockAt;
        const size_t nBlocks = blockAt.size() - 1;
        const size_t base = out.size();
        if (nBlocks > 0x7fffffffu) throw Error("lz4FrameDecompress: too many blocks");
        std::vector<int32_t> blen;
        // The engine's linked mode keeps the reference's window: the output of the block before (Internal/LZ4.hs:564,
        // lz4.c:2347-2355), or the dictionary handed in for the first block of a call.  The frame format's window is
        // the last 64 KiB of the frame -- the same thing as long as blocks are full.  So a group also ends behind a
        // block that came out short (a writer that flushed in mid-frame; never liblz4's one-shot LZ4F_compressFrame
        // or the CLI): the block after it starts the next group and gets the frame's window as that call's dictionary.
        // After a cut behind a short block the next group starts small and doubles while groups come out whole: a frame
        // of N flushed short blocks then costs O(N) block decodes, not N groups of thousands of blocks each (round-3
        // advisor finding).
        size_t g0 = 0, groupLimit = (size_t)-1;
        while (g0 < nBlocks) {
            // capacity per block of this group and how many blocks fit the budget
            size_t cap = 0, g1 = g0;
            while (g1 < nBlocks && g1 - g0 < groupLimit) {
                const size_t clen = blockAt[g1 + 1] - blockAt[g1] - 4;
                const size_t c1 = std::max(cap, std::min(bmax, clen * 255 + 16));
                if (g1 > g0 && (g1 - g0 + 1) * c1 > LZ4F_GROUP_BUDGET) break;
                cap = c1;
                g1++;
            }
            if (ix.hasContentSize) {
                // a frame that declares its size cannot need more than what is left of it
                const uint64_t done = out.size() - base;
                if (done > ix.contentSize) throw Error("lz4FrameDecompress: content size mismatch");
                cap = (size_t)std::min<uint64_t>(cap, std::max<uint64_t>(ix.contentSize - done, 1));
            }
            size_t nb = g1 - g0;
            if (scratch.size() < nb * cap) scratch.resize(nb * cap);
            blen.assign(nb, 0);
            const size_t hist = independent ? 0 : std::min(out.size() - base, (size_t)65536);
            const uint8_t *dict = hist ? out.data() + out.size() - hist : nullptr;
            size_t got = 0;
            int nbOut = 0;
            int r = mi355lz4_decompress_batch(eng.ctx(), framed.data() + blockAt[g0], blockAt[g1] - blockAt[g0], 4, (int)cap,
                                              independent ? 0 : 1, dict, (int)hist, scratch.data(), nb * cap, &got, blen.data(),
                                              (int)nb, &nbOut);
            if (r != MI355LZ4_OK && r != MI355LZ4_E_BLOCK) throw Error(std::string("lz4FrameDecompress: ") + mi355lz4_last_error());
            // how many blocks of the group stand: up to the first failure, and (linked) up to and including the first
            // short block that has a successor in the frame
            size_t keep = nb;
            for (size_t k = 0; k < nb; k++) {
                if (blen[k] < 0) { keep = k; break; }
                if (!independent && (size_t)blen[k] < std::min(bmax, (size_t)65536) && g0 + k + 1 < nBlocks) { keep = k + 1; break; }
            }
            if (keep == 0) {
                // the group's first block had the whole window and still fails: the frame is damaged
                                throw Error(std::string("lz4FrameDecompress: ") + (r == MI355LZ4_OK ? "block decode failed" : mi355lz4_last_error()));
            }
            if (keep < nb && r == MI355LZ4_OK) {
                // every block decoded; the ones that stand had their whole window and lie packed at the front of the
                // scratch: keep those bytes, drop the rest (no second decode)
                got = 0;
                for (size_t k = 0; k < keep; k++) got += (size_t)blen[k];
                nb = keep;
                g1 = g0 + keep;
                groupLimit = 2;
            } else if (keep < nb || r != MI355LZ4_OK) {
                // decode exactly the blocks that stand (what a failed call leaves in the output buffer is not part of the
                // C ABI's contract)
                groupLimit = 2;
                nb = keep;
                g1 = g0 + keep;
                r = mi355lz4_decompress_batch(eng.ctx(), framed.data() + blockAt[g0], blockAt[g1] - blockAt[g0], 4, (int)cap,
                                              independent ? 0 : 1, dict, (int)hist, scratch.data(), nb * cap, &got, blen.data(),
                                              (int)nb, &nbOut);
                if (r != MI355LZ4_OK) throw Error(std::string("lz4FrameDecompress: ") + mi355lz4_last_error());
            }
            else if (groupLimit != (size_t)-1) groupLimit = (groupLimit > ((size_t)-1) / 2) ? (size_t)-1 : groupLimit * 2;
            out.insert(out.end(), scratch.data(), scratch.data() + got);
            g0 = g1;
        }
        if (ix.hasContentSize && (uint64_t)(out.size() - base) != ix.contentSize) throw Error("lz4FrameDecompress: content size mismatch");
        if (ix.hasContentChecksum && ix.contentChecksum != xxh32(out.data() + base, out.size() - base, 0))
            throw Error("lz4FrameDecompress: content checksum mismatch");
    }
    return out;
}

} // namespace streamly_lz4
