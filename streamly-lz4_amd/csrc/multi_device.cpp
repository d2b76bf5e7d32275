// multi_device.cpp -- several GPUs behind ONE handle of the C ABI, in one process.
//
// The Haskell host is one process and its calls are bound by PCIe (40-46 GB/s of a 54.9 GB/s link per GPU,
// BENCH host_api_pcie_inclusive): for a host caller more GPUs means more PCIe links, and nothing else is left to gain.
// A batch is cut into as many contiguous block ranges as the handle has devices (equal shares of the uncompressed bytes),
// every range goes through the ordinary host-buffer call of its own engine -- own pinned staging, own streams -- on a host
// thread of its own, and the results land in order in the caller's one buffer: a host consumer needs no gather at all
// ("alternative when the consumer is the host").  Blocks are independent (what the engine's compressor
// writes by default and what its linked decoder accepts too), so no range needs another's bytes.
//
// Why ranges and not block i -> device i mod n as between ranks (gather.py): there the ranks' outputs are interleaved by a
// kernel on the root; here the destination is host memory, where a per-block round-robin would turn every device-to-host
// copy into one copy per block.  A range per device moves the same bytes over the same links in n large copies.
//
// replaces: nothing the reference has (its API is one serial stream, src/Streamly/LZ4.hs:98,117); the calls mirror
// mi355lz4_compress_batch / mi355lz4_decompress_batch (Internal/LZ4.hs:226-281, :291-336) argument for argument.
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/mi355lz4.h"

struct mi355lz4_multi {
    std::vector<mi355lz4_ctx *> eng;
};

static thread_local char g_multi_err[512];

extern "C" const char *mi355lz4_multi_last_error(void) { return g_multi_err; }

static int mfail(int code, const char *what, const std::string &detail = std::string())
{
    snprintf(g_multi_err, sizeof(g_multi_err), "%s%s%s", what, detail.empty() ? "" : ": ", detail.c_str());
    return code;
}

extern "C" int mi355lz4_create_multi(mi355lz4_multi **out, const int *devices, int n)
{
    if (!out) return mfail(MI355LZ4_E_ARG, "mi355lz4_create_multi: null out");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return mfail(MI355LZ4_E_ARG, "mi355lz4_create_multi: 1..64 devices");
    mi355lz4_multi *m = new (std::nothrow) mi355lz4_multi();
    if (!m) return mfail(MI355LZ4_E_ARG, "out of host memory");
    for (int i = 0; i < n; i++) {
        mi355lz4_ctx *c = nullptr;
        const int r = mi355lz4_create(&c, devices[i]);       // (the same device may be named more than once: two engines on it)
        if (r != MI355LZ4_OK) {
            const std::string why = mi355lz4_last_error();
            for (mi355lz4_ctx *e : m->eng) mi355lz4_destroy(e);
            delete m;
            return mfail(r, "mi355lz4_create_multi", why);
        }
        m->eng.push_back(c);
    }
    *out = m;
    return MI355LZ4_OK;
}

extern "C" void mi355lz4_destroy_multi(mi355lz4_multi *m)
{
    if (!m) return;
    for (mi355lz4_ctx *e : m->eng) mi355lz4_destroy(e);
    delete m;
}

extern "C" int mi355lz4_multi_device_count(const mi355lz4_multi *m) { return m ? (int)m->eng.size() : 0; }
extern "C" mi355lz4_ctx *mi355lz4_multi_engine(mi355lz4_multi *m, int i)
{
    return (m && i >= 0 && i < (int)m->eng.size()) ? m->eng[(size_t)i] : nullptr;
}

// Cut n blocks of the given sizes into at most `parts` contiguous ranges of about equal bytes; first[p] .. first[p + 1].
static std::vector<int> cut_ranges(const int32_t *len, int n, int parts)
{
    uint64_t total = 0;
    for (int i = 0; i < n; i++) total += (uint64_t)(len[i] > 0 ? len[i] : 0) + 1u;     // (+ 1: empty blocks still count)
    if (parts > n) parts = n;
    if (parts < 1) parts = 1;
    std::vector<int> first;
    first.push_back(0);
    uint64_t acc = 0;
    for (int i = 0; i < n; i++) {
        acc += (uint64_t)(len[i] > 0 ? len[i] : 0) + 1u;
        const int p = (int)first.size();                     // the range being filled is p - 1; close it once its share is reached
        if (p < parts && acc * (uint64_t)parts >= total * (uint64_t)p && i + 1 < n) first.push_back(i + 1);
    }
    first.push_back(n);
    return first;
}

struct PartResult {
    int rc = MI355LZ4_OK;
    std::string err;
    size_t outLen = 0;
    int got = 0;
};

extern "C" int mi355lz4_multi_compress_batch(mi355lz4_multi *m, const uint8_t *const *src, const int32_t *srcLen, int nBlocks,
                                             int accel, int headerKind, uint8_t *framedOut, size_t cap, size_t *outLen,
                                             int32_t *blockFramedLen, int32_t *status)
{
    if (!m || m->eng.empty()) return mfail(MI355LZ4_E_ARG, "mi355lz4_multi_compress_batch: null handle");
    if (nBlocks < 0 || !outLen || (nBlocks > 0 && (!src || !srcLen || !framedOut)))
        return mfail(MI355LZ4_E_ARG, "mi355lz4_multi_compress_batch: bad arguments");
    *outLen = 0;
    if (nBlocks == 0) return MI355LZ4_OK;
    const std::vector<int> first = cut_ranges(srcLen, nBlocks, (int)m->eng.size());
    const int P = (int)first.size() - 1;
    // Every range is compressed into the caller's buffer at the place its worst case allows (the ranges in front of it at
    // their bounds), and moved down afterwards -- compressed bytes only; when the caller's capacity is tighter than the sum of
    // the bounds, into a buffer of its own.
    std::vector<size_t> bound((size_t)P, 0), at((size_t)P, 0);
    size_t sum = 0;
    for (int p = 0; p < P; p++) {
        for (int i = first[(size_t)p]; i < first[(size_t)p + 1]; i++) {
            const int b = mi355lz4_compress_bound(srcLen[i]);
            if (srcLen[i] < 0 || b <= 0) return mfail(MI355LZ4_E_ARG, "mi355lz4_multi_compress_batch: a block is larger than LZ4_MAX_INPUT_SIZE");
            bound[(size_t)p] += (size_t)b + (size_t)headerKind;
        }
        at[(size_t)p] = sum;
        sum += bound[(size_t)p];
    }
    const bool inPlace = cap >= sum;
    std::vector<std::vector<uint8_t>> own((size_t)P);
    std::vector<PartResult> res((size_t)P);
    std::vector<std::thread> th;
    for (int p = 0; p < P; p++) {
        uint8_t *dstp = framedOut + at[(size_t)p];
        size_t capp = bound[(size_t)p];
        if (!inPlace) {
            if (p == 0) capp = cap;
            else {
                try { own[(size_t)p].resize(bound[(size_t)p]); } catch (...) { return mfail(MI355LZ4_E_ARG, "out of host memory"); }
                dstp = own[(size_t)p].data();
            }
        }
        th.emplace_back([=, &res] {
            const int b0 = first[(size_t)p], nb = first[(size_t)p + 1] - b0;
            PartResult &r = res[(size_t)p];
            r.rc = mi355lz4_compress_batch(m->eng[(size_t)p], src + b0, srcLen + b0, nb, accel, headerKind, dstp, capp, &r.outLen,
                                           blockFramedLen ? blockFramedLen + b0 : nullptr, status ? status + b0 : nullptr);
            if (r.rc != MI355LZ4_OK) r.err = mi355lz4_last_error();
        });
    }
    for (std::thread &t : th) t.join();
    for (int p = 0; p < P; p++)
        if (res[(size_t)p].rc != MI355LZ4_OK) return mfail(res[(size_t)p].rc, "mi355lz4_multi_compress_batch", res[(size_t)p].err);
    size_t w = res[0].outLen;
    for (int p = 1; p < P; p++) {
        const size_t n = res[(size_t)p].outLen;
        if (w + n > cap) return mfail(MI355LZ4_E_CAPACITY, "mi355lz4_multi_compress_batch: output capacity too small");
        if (inPlace) memmove(framedOut + w, framedOut + at[(size_t)p], n);
        else memcpy(framedOut + w, own[(size_t)p].data(), n);
        w += n;
    }
    *outLen = w;
    return MI355LZ4_OK;
}

extern "C" int mi355lz4_multi_decompress_batch(mi355lz4_multi *m, const uint8_t *framedIn, size_t inLen, int headerKind,
                                               int fixedUncomp, uint8_t *out, size_t cap, size_t *outLen, int32_t *blockLen,
                                               int maxBlocks, int *nBlocksOut)
{
    if (!m || m->eng.empty()) return mfail(MI355LZ4_E_ARG, "mi355lz4_multi_decompress_batch: null handle");
    if (!outLen || !nBlocksOut || maxBlocks < 0 || (inLen > 0 && (!framedIn || !out)))
        return mfail(MI355LZ4_E_ARG, "mi355lz4_multi_decompress_batch: bad arguments");
    *outLen = 0;
    *nBlocksOut = 0;
    if (inLen == 0) return MI355LZ4_OK;
    // the header chain, once, on the host (resizeChunksD's job, Internal/LZ4.hs:459-484): where every block starts and how
    // much room it asks for
    std::vector<uint64_t> boff((size_t)maxBlocks + 1);
    std::vector<int32_t> ulen((size_t)maxBlocks + 1);
    int n = 0;
    const int ri = mi355lz4_index_host(framedIn, inLen, headerKind, fixedUncomp, boff.data(), ulen.data(), maxBlocks, &n);
    if (ri != MI355LZ4_OK) return mfail(ri, "mi355lz4_multi_decompress_batch", mi355lz4_last_error());
    if (n == 0) return MI355LZ4_OK;
    boff[(size_t)n] = inLen;
    std::vector<uint64_t> ooff((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) ooff[(size_t)i + 1] = ooff[(size_t)i] + (uint64_t)(ulen[(size_t)i] > 0 ? ulen[(size_t)i] : 0);
    if (ooff[(size_t)n] > cap) return mfail(MI355LZ4_E_CAPACITY, "mi355lz4_multi_decompress_batch: output capacity too small");
    const std::vector<int> first = cut_ranges(ulen.data(), n, (int)m->eng.size());
    const int P = (int)first.size() - 1;
    std::vector<PartResult> res((size_t)P);
    std::vector<std::thread> th;
    for (int p = 0; p < P; p++) {
        th.emplace_back([=, &res, &boff, &ooff] {
            const int b0 = first[(size_t)p], b1 = first[(size_t)p + 1];
            PartResult &r = res[(size_t)p];
            r.rc = mi355lz4_decompress_batch(m->eng[(size_t)p], framedIn + boff[(size_t)b0], (size_t)(boff[(size_t)b1] - boff[(size_t)b0]),
                                             headerKind, fixedUncomp, 0, nullptr, 0, out + ooff[(size_t)b0],
                                             (size_t)(ooff[(size_t)b1] - ooff[(size_t)b0]), &r.outLen,
                                             blockLen ? blockLen + b0 : nullptr, b1 - b0, &r.got);
            if (r.rc != MI355LZ4_OK) r.err = mi355lz4_last_error();
        });
    }
    for (std::thread &t : th) t.join();
    *nBlocksOut = n;
    int worst = MI355LZ4_OK;
    std::string why;
    for (int p = 0; p < P; p++)
        if (res[(size_t)p].rc != MI355LZ4_OK && (worst == MI355LZ4_OK || worst == MI355LZ4_E_BLOCK)) { worst = res[(size_t)p].rc; why = res[(size_t)p].err; }
    if (worst != MI355LZ4_OK) return mfail(worst, "mi355lz4_multi_decompress_batch", why);
    // a block may decode to fewer bytes than its header asks room for: every range is packed already, pack the ranges
    size_t w = res[0].outLen;
    for (int p = 1; p < P; p++) {
        const size_t at = (size_t)ooff[(size_t)first[(size_t)p]];
        if (w != at && res[(size_t)p].outLen) memmove(out + w, out + at, res[(size_t)p].outLen);
        w += res[(size_t)p].outLen;
    }
    *outLen = w;
    return MI355LZ4_OK;
}
