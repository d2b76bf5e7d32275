// host_san_test.cpp -- sanitizer driver for the HOST side of libmi355lz4 (api.cpp + host_stream.cpp).
//
// Built by `make asan` / `make tsan` (CPU only: the kernel launchers are stubbed by san_stubs.cpp, and
// without a gfx950 device every engine call stops at MI355LZ4_E_NO_DEVICE).  What runs under the sanitizers:
//   * the staging copy pool (api.cpp) hammered from several caller threads at once;
//   * the stream state machines that need no codec: resizeChunks at every split size the reference tests
//     (test/Main.hs:217-224), end mark, the frame-header parser, and their error paths;
//   * the legacy LZ4_* entry points' no-device behaviour (create/free, compressBound, 0 / -1 returns).
#include "../../include/lz4.h"
#include "../../include/mi355lz4.h"
#include "../../include/streamly_lz4.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

extern "C" int mi355lz4_debug_host_copy(uint8_t *dst, const uint8_t *src, size_t n);

static int failures = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); failures++; } } while (0)

static void pool_stress()
{
    std::vector<std::thread> th;
    for (int t = 0; t < 4; t++)
        th.emplace_back([t] {
            std::mt19937_64 rng(1234 + t);
            for (int it = 0; it < 12; it++) {
                const size_t n = (size_t)(rng() % (6u << 20)) + 1;
                std::vector<uint8_t> a(n), b(n, 0);
                for (size_t i = 0; i < n; i += 997) a[i] = (uint8_t)rng();
                if (mi355lz4_debug_host_copy(b.data(), a.data(), n) != 0 || memcmp(a.data(), b.data(), n) != 0) {
                    fprintf(stderr, "pool copy mismatch (thread %d, n %zu)\n", t, n);
                    __atomic_fetch_add(&failures, 1, __ATOMIC_RELAXED);
                }
            }
        });
    for (auto &x : th) x.join();
}

// a framed stream of fake blocks: resizeChunks only reads headers (Internal/LZ4.hs:459-484)
static std::vector<uint8_t> fake_stream(std::mt19937_64 &rng, int nBlocks, int meta, bool endMark, std::vector<size_t> &cuts)
{
    std::vector<uint8_t> s;
    for (int i = 0; i < nBlocks; i++) {
        const uint32_t c = (uint32_t)(rng() % 3000) + 1, u = (uint32_t)(rng() % 70000);
        cuts.push_back(s.size());
        for (int k = 0; k < 4; k++) s.push_back((uint8_t)(c >> (8 * k)));
        if (meta == 8) for (int k = 0; k < 4; k++) s.push_back((uint8_t)(u >> (8 * k)));
        for (uint32_t k = 0; k < c; k++) s.push_back((uint8_t)rng());
    }
    cuts.push_back(s.size());
    if (endMark) for (int k = 0; k < 4; k++) s.push_back(0);
    return s;
}

static void resize_checks()
{
    using namespace streamly_lz4;
    std::mt19937_64 rng(99);
    for (int meta : {8, 4})
        for (bool endMark : {false, true})
            for (size_t split : {(size_t)1, (size_t)512, (size_t)32768, (size_t)262144}) {
                std::vector<size_t> cuts;
                std::vector<uint8_t> s = fake_stream(rng, 40, meta, endMark, cuts);
                if (endMark) for (int k = 0; k < 7; k++) s.push_back(0xEE);   // trailing bytes after the end mark are ignored (:506-521)
                std::vector<Array> in;
                for (size_t o = 0; o < s.size(); o += split)
                    in.emplace_back(s.begin() + (long)o, s.begin() + (long)std::min(s.size(), o + split));
                BlockConfig cfg;
                cfg.blockSize = meta == 8 ? BlockSize::BlockHasSize : BlockSize::BlockMax64KB;
                FrameConfig fc;
                fc.hasEndMark = endMark;
                StreamPtr r = resizeChunks(cfg, fc, fromList(in));
                std::vector<Array> out = toList(*r);
                CHECK(out.size() + 1 == cuts.size());
                for (size_t i = 0; i < out.size() && i + 1 < cuts.size(); i++)
                    CHECK(out[i].size() == cuts[i + 1] - cuts[i] && memcmp(out[i].data(), s.data() + cuts[i], out[i].size()) == 0);
                // idempotence (test/Main.hs:189-201)
                if (!endMark) {
                    std::vector<Array> again = toList(*resizeChunks(cfg, fc, fromList(out)));
                    CHECK(again.size() == out.size());
                    for (size_t i = 0; i < again.size() && i < out.size(); i++) CHECK(again[i] == out[i]);
                }
            }
    // error paths: a stream cut inside a block, and a missing end mark
    {
        std::vector<size_t> cuts;
        std::vector<uint8_t> s = fake_stream(rng, 3, 8, false, cuts);
        s.resize(s.size() - 5);
        bool threw = false;
        try { toList(*resizeChunks(BlockConfig(), FrameConfig(), fromList({Array(s.begin(), s.end())}))); }
        catch (const Error &) { threw = true; }
        CHECK(threw);
        FrameConfig fc;
        fc.hasEndMark = true;
        std::vector<size_t> c2;
        std::vector<uint8_t> s2 = fake_stream(rng, 3, 8, false, c2);
        threw = false;
        try { toList(*resizeChunks(BlockConfig(), fc, fromList({Array(s2.begin(), s2.end())}))); }
        catch (const Error &) { threw = true; }
        CHECK(threw);
    }
    // frame header parser (Internal/LZ4.hs:590-651)
    {
        const uint8_t hdr[7] = {0x04, 0x22, 0x4D, 0x18, 0x40, 0x40, 0x00};
        auto r = simpleFrameParser(fromList({Array(hdr, hdr + 3), Array(hdr + 3, hdr + 7)}));
        CHECK(r.first.first.blockSize == BlockSize::BlockMax64KB && r.first.second.hasEndMark);
        uint8_t bad[7];
        memcpy(bad, hdr, 7);
        bad[4] = 0x60;                                            // block-independence flag: rejected (:631-632)
        bool threw = false;
        try { simpleFrameParser(fromList({Array(bad, bad + 7)})); } catch (const Error &) { threw = true; }
        CHECK(threw);
    }
}

static void legacy_no_device()
{
    CHECK(LZ4_compressBound(65536) == 65536 + 65536 / 255 + 16);
    CHECK(LZ4_compressBound(0x7E000001) == 0);
    mi355lz4_ctx *c = nullptr;
    const int rc = mi355lz4_create(&c, 0);
    if (rc == MI355LZ4_OK) { mi355lz4_destroy(c); return; }        // a GPU box: nothing more to check here
    CHECK(rc == MI355LZ4_E_NO_DEVICE && c == nullptr && strlen(mi355lz4_last_error()) > 0);
    void *cs = LZ4_createStream();
    void *ds = LZ4_createStreamDecode();
    char src[64] = {0}, dst[128];
    CHECK(LZ4_compress_fast_continue((LZ4_stream_t *)cs, src, dst, 64, 128, 1) == 0);     // no CPU codec: fails like the reference reports failure
    CHECK(LZ4_decompress_safe_continue((LZ4_streamDecode_t *)ds, src, dst, 1, 128) < 0);
    LZ4_freeStream((LZ4_stream_t *)cs);
    LZ4_freeStreamDecode((LZ4_streamDecode_t *)ds);
    LZ4_freeStream(nullptr);
    bool threw = false;
    try { streamly_lz4::Engine e(0); } catch (const streamly_lz4::Error &) { threw = true; }
    CHECK(threw);
}

// the frame reader's host half on valid, damaged and truncated frames: it either parses or throws, nothing else
static void frame_parser_fuzz()
{
    using namespace streamly_lz4;
    std::mt19937_64 rng(99);
    auto put32 = [](Array &a, uint32_t v) { for (int k = 0; k < 4; k++) a.push_back((uint8_t)(v >> (8 * k))); };
    for (int iter = 0; iter < 300; iter++) {
        const bool bsum = rng() & 1, csum = rng() & 1, csize = rng() & 1, indep = rng() & 1;
        const int code = 4 + (int)(rng() % 4);
        Array f, content;
        if (rng() % 4 == 0) { put32(f, 0x184D2A50u + (uint32_t)(rng() % 16)); put32(f, 5); for (int k = 0; k < 5; k++) f.push_back((uint8_t)k); }
        const size_t frameAt = f.size();
        put32(f, 0x184D2204u);
        const size_t descAt = f.size();
        f.push_back((uint8_t)(0x40 | (indep ? 0x20 : 0) | (bsum ? 0x10 : 0) | (csize ? 0x08 : 0) | (csum ? 0x04 : 0)));
        f.push_back((uint8_t)(code << 4));
        const int nb = (int)(rng() % 5);
        std::vector<Array> blocks;
        for (int b = 0; b < nb; b++) {
            Array blk(rng() % 700);
            for (auto &x : blk) x = (uint8_t)rng();
            content.insert(content.end(), blk.begin(), blk.end());
            blocks.push_back(std::move(blk));
        }
        if (csize) for (int k = 0; k < 8; k++) f.push_back((uint8_t)((uint64_t)content.size() >> (8 * k)));
        f.push_back((uint8_t)(xxh32(f.data() + descAt, f.size() - descAt, 0) >> 8));
        for (const Array &blk : blocks) {
            put32(f, (uint32_t)blk.size() | 0x80000000u);
            f.insert(f.end(), blk.begin(), blk.end());
            if (bsum) put32(f, xxh32(blk.data(), blk.size(), 0));
        }
        put32(f, 0);
        if (csum) put32(f, xxh32(content.data(), content.size(), 0));

        // intact: parses, and the re-framed literal blocks carry the content
        {
            size_t at = 0;
            Lz4FrameIndex ix;
            bool isFrame = lz4FrameParse(f, at, ix);
            if (frameAt) { CHECK(!isFrame && at == frameAt); isFrame = lz4FrameParse(f, at, ix); }
            CHECK(isFrame && at == f.size() && ix.independent == indep && ix.blockMax == ((size_t)1 << (8 + 2 * code)));
            CHECK(!csize || ix.contentSize == content.size());
            size_t bytes = 0, live = 0;
            for (const Array &blk : blocks) { bytes += blk.size(); live += !blk.empty(); }
            CHECK(ix.blockAt.size() == live + 1 && ix.framed.size() >= bytes + 5 * live);
        }
        // damaged
        for (int m = 0; m < 20; m++) {
            Array g = f;
            const int kind = (int)(rng() % 3);
            if (kind == 0 && !g.empty()) g.resize(rng() % g.size());
            else if (kind == 1 && !g.empty()) g[rng() % g.size()] ^= (uint8_t)(1u << (rng() % 8));
            else for (int k = 0; k < 4 && !g.empty(); k++) g[rng() % g.size()] = (uint8_t)rng();
            size_t at = 0;
            try {
                Lz4FrameIndex ix;
                while (at < g.size()) lz4FrameParse(g, at, ix);
            } catch (const Error &) {
            }
            CHECK(at <= g.size());
        }
    }
    CHECK(xxh32(nullptr, 0, 0) == 0x02CC5D05u);
}

int main()
{
    frame_parser_fuzz();
    pool_stress();
    resize_checks();
    legacy_no_device();
    if (failures) { fprintf(stderr, "host_san_test: %d failure(s)\n", failures); return 1; }
    printf("host_san_test ok\n");
    return 0;
}
