// api.cpp -- C ABI of the MI355X LZ4 block engine (include/mi355lz4.h, include/lz4.h).
//
// Host-side plumbing only: argument checks, device workspaces, copies and
// kernel launches.  All arithmetic of the hot path happens in kernels.hip.
// There is no CPU code path: without a gfx950 device every call fails.
#include "../../include/mi355lz4.h"
#include "../../include/lz4.h"

#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(MI355LZ4_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// ---------------------------------------------------------------------------
// Host copy pool: the staging copies between pageable caller memory and the
// pinned buffers are what bounds the host-buffer API (one thread moves ~10 GB/s,
// the Gen5 x16 link ~55 GB/s), so they are spread over a few threads.
// MI355LZ4_COPY_THREADS overrides the count (default 8, 1 = no helper threads).
// ---------------------------------------------------------------------------
struct CopyTask { uint8_t *dst; const uint8_t *src; size_t n; };

class CopyPool {
public:
    CopyPool()
    {
        int n = 8;
        if (const char *e = getenv("MI355LZ4_COPY_THREADS")) n = atoi(e);
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)n > hw) n = (int)hw;
        if (n < 1) n = 1;
        for (int i = 1; i < n; i++) workers_.emplace_back([this] { loop(); });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cvWork_.notify_all();
        for (auto &t : workers_) t.join();
    }
    // run all tasks; the calling thread works too; returns when every byte is copied
    void run(const std::vector<CopyTask> &tasks)
    {
        if (tasks.empty()) return;
        if (workers_.empty() || tasks.size() == 1) {
            for (const CopyTask &t : tasks) if (t.n) memcpy(t.dst, t.src, t.n);
            return;
        }
        std::lock_guard<std::mutex> one(callers_);   // one batch of tasks at a time
        std::unique_lock<std::mutex> lk(m_);
        tasks_ = &tasks; next_ = 0; pending_ = tasks.size();
        grab_ = tasks.size() / ((workers_.size() + 1) * 8) + 1;    // a few grabs per thread: small tasks share a lock trip
        cvWork_.notify_all();
        while (next_ < tasks.size()) {
            const size_t lo = next_, hi = (lo + grab_ < tasks.size()) ? lo + grab_ : tasks.size();
            next_ = hi;
            lk.unlock();
            for (size_t i = lo; i < hi; i++) if (tasks[i].n) memcpy(tasks[i].dst, tasks[i].src, tasks[i].n);
            lk.lock();
            pending_ -= hi - lo;
        }
        cvDone_.wait(lk, [this] { return pending_ == 0; });
        tasks_ = nullptr;
    }
    // one large range, cut into slices
    void copy(uint8_t *dst, const uint8_t *src, size_t n)
    {
        static const size_t kSlice = (size_t)1 << 20;
        std::vector<CopyTask> t;
        for (size_t off = 0; off < n; off += kSlice) t.push_back({dst + off, src + off, (n - off < kSlice) ? n - off : kSlice});
        run(t);
    }

private:
    void loop()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cvWork_.wait(lk, [this] { return stop_ || (tasks_ && next_ < tasks_->size()); });
            if (stop_) return;
            while (tasks_ && next_ < tasks_->size()) {
                const std::vector<CopyTask> &ts = *tasks_;
                const size_t lo = next_, hi = (lo + grab_ < ts.size()) ? lo + grab_ : ts.size();
                next_ = hi;
                lk.unlock();
                for (size_t i = lo; i < hi; i++) if (ts[i].n) memcpy(ts[i].dst, ts[i].src, ts[i].n);
                lk.lock();
                pending_ -= hi - lo;
                if (pending_ == 0) cvDone_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, callers_;
    std::condition_variable cvWork_, cvDone_;
    const std::vector<CopyTask> *tasks_ = nullptr;
    size_t next_ = 0, pending_ = 0, grab_ = 1;
    bool stop_ = false;
};

static CopyPool &copy_pool()
{
    static CopyPool *pool = new CopyPool();     // leaked on purpose: no thread joins during process teardown
    return *pool;
}

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct mi355lz4_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    int decoder = 0;
    int linkedCompress = 0;                 // compress calls treat their blocks as consecutive blocks of one stream
    // workspaces of the host-buffer API (grown on demand, reused across calls)
    DevBuf in, slots, dense, out, offA, offB, lenA, lenB, res, scratch;
    DevBuf tokBuf;                          // decoder variant 3: token lists
    DevBuf tolPool, tolMeta;                // deferred-copy decode of a long linked stream (linked_replay.hpp)
    DevBuf linkBuf, ptrBuf, pinStat;        // ... its failure count, control block and source pointers (linked_ptr.hpp)
    DevBuf pinIn, pinOut;   // pinned host staging
    DevBuf pinMeta;         // pinned: per-group sizes coming back from the device
    hipStream_t sIn = nullptr, sOut = nullptr;   // copy streams of the pipelined host-buffer API (created on first use)
    hipStream_t sK[2] = {nullptr, nullptr};   // compute streams: kernels of consecutive groups overlap
    // a linked decode whose data half is still to be issued (mi355lz4_decompress_linked_begin / _end)
    struct LinkedPlan {
        bool active = false, split = false;
        DecodeArgs a;
        int first = 0, last = -1, pool = 0, seg = 0;
    } plan;
    // small-batch compression: per-segment sequence lists, one scratch buffer per stream the engine has been used on
    // (the host pipelines run two groups at a time on two compute streams; work on ONE stream is ordered)
    struct SegScratch { hipStream_t s = nullptr; DevBuf b; unsigned long long tick = 0; } seg[4];
    int nSeg = 0;
    unsigned long long segTick = 0;
    int linkedAsyncCap = 0;                // > 0: linked device decodes do not wait on the host (mi355lz4_set_linked_async)
    // the run-in decode adapts to what the engine's streams are like (the calls that follow one are, as a rule, more of the same):
    bool runinLong = false;                // a call was given up with the default run-in (chains of pieces to redo): the long one from here on
    int runinLongOk = 0;                   // ... calls in a row that finished with it (after RUNIN_LONG_PROBE the default is tried again)
    int runinSkip = 0;                     // ... and given up with the long one too: this many linked decodes go straight to the pointer pass
    int linkedPath = -1;                   // diagnostics: how the last linked call was finished (mi355lz4_debug_runin_state)
    int runinShareE6 = -1;                 // diagnostics: the dictionary share the last linked call sampled, in millionths (-1: none)
    int segMode = -1;                      // small-batch segments per block: -1 auto, 0 off, k forced (mi355lz4_set_segments)
    hipEvent_t linkEvent = nullptr;        // end of the last linked decode's use of linkBuf / tolPool / tolMeta / ptrBuf
    hipStream_t linkStream = nullptr;      // ... and the stream it ran on
    bool linkBusy = false;
    unsigned long long *stats = nullptr;   // diagnostics: device counters of the lane-parallel decoder (off by default)
    uint32_t *cuDbg = nullptr;             // diagnostics: 16 words per block from the workgroup-per-block decoder (mi355lz4_debug_cu)
};

static int dev_reserve(DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (b.p) { hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}
static int pin_reserve(DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (b.p) { hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(hipHostMalloc(&b.p, want, hipHostMallocDefault));
    b.cap = want;
    return 0;
}
static void dev_release(DevBuf &b) { if (b.p) hipFree(b.p); b.p = nullptr; b.cap = 0; }
static void pin_release(DevBuf &b) { if (b.p) hipHostFree(b.p); b.p = nullptr; b.cap = 0; }

// ---------------------------------------------------------------------------
// Host <-> device transfers of pageable memory, pipelined through pinned staging:
// the CPU memcpy of chunk i+1 overlaps the DMA of chunk i.
// A plain hipMemcpy of pageable memory runs at a few GB/s; this keeps the link busy.
// ---------------------------------------------------------------------------
static size_t stage_chunk()
{
    static const size_t v = [] {
        const char *e = getenv("MI355LZ4_STAGE_CHUNK_MB");
        const long mb = e ? atol(e) : 16;
        return (size_t)((mb < 1) ? 1 : (mb > 256 ? 256 : mb)) << 20;
    }();
    return v;
}
#define kStageChunk (stage_chunk())

static int h2d_staged(mi355lz4_ctx *c, void *dstDev, const uint8_t *srcHost, size_t bytes)
{
    if (!bytes) return 0;
    int r = pin_reserve(c->pinIn, bytes);
    if (r) return r;
    uint8_t *stage = (uint8_t *)c->pinIn.p;
    for (size_t off = 0; off < bytes; off += kStageChunk) {
        const size_t n = (bytes - off < kStageChunk) ? bytes - off : kStageChunk;
        copy_pool().copy(stage + off, srcHost + off, n);
        HIP_TRY(hipMemcpyAsync((uint8_t *)dstDev + off, stage + off, n, hipMemcpyHostToDevice, c->stream));
    }
    return 0;
}

static int d2h_staged(mi355lz4_ctx *c, uint8_t *dstHost, const void *srcDev, size_t bytes)
{
    if (!bytes) return 0;
    int r = pin_reserve(c->pinOut, bytes);
    if (r) return r;
    uint8_t *stage = (uint8_t *)c->pinOut.p;
    const size_t nChunks = (bytes + kStageChunk - 1) / kStageChunk;
    std::vector<hipEvent_t> ev(nChunks, nullptr);
    int rc = 0;
    size_t issued = 0;
    for (size_t k = 0; k < nChunks && !rc; k++) {
        const size_t off = k * kStageChunk;
        const size_t n = (bytes - off < kStageChunk) ? bytes - off : kStageChunk;
        if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess ||
            hipMemcpyAsync(stage + off, (const uint8_t *)srcDev + off, n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipEventRecord(ev[k], c->stream) != hipSuccess)
            rc = fail(MI355LZ4_E_HIP, "d2h_staged: copy of chunk %zu could not be queued", k);
        else
            issued = k + 1;
    }
    for (size_t k = 0; k < nChunks; k++) {
        const size_t off = k * kStageChunk;
        const size_t n = (bytes - off < kStageChunk) ? bytes - off : kStageChunk;
        if (!rc && k < issued && hipEventSynchronize(ev[k]) != hipSuccess) rc = fail(MI355LZ4_E_HIP, "hipEventSynchronize failed");
        if (!rc && k < issued) copy_pool().copy(dstHost + off, stage + off, n);
        if (ev[k]) hipEventDestroy(ev[k]);
    }
    return rc;
}

static bool device_is_gfx950(int dev)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

// Diagnostic hook (not part of the public header): one copy through the staging copy pool, so that the
// sanitizer driver (tests/native/host_san_test.cpp) can exercise the pool without a device.
extern "C" int mi355lz4_debug_host_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
    if (n && (!dst || !src)) return fail(MI355LZ4_E_ARG, "debug_host_copy: null pointer");
    copy_pool().copy(dst, src, n);
    return MI355LZ4_OK;
}

extern "C" int mi355lz4_version(void) { return MI355LZ4_VERSION; }
extern "C" const char *mi355lz4_last_error(void) { return g_err; }

extern "C" int mi355lz4_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; d++) ok += device_is_gfx950(d) ? 1 : 0;
    return ok;
}

extern "C" int mi355lz4_create(mi355lz4_ctx **out, int device)
{
    if (!out) return fail(MI355LZ4_E_ARG, "mi355lz4_create: null out");
    *out = nullptr;
    // The host-buffer pipelines use five streams; HIP's default of four hardware queues makes two of them share one.
    // The variable is read when the HIP runtime initialises, so this only matters when this call is the process's
    // first HIP call; it never overrides a value the process has set, and MI355LZ4_KEEP_HW_QUEUES=1 leaves the
    // environment alone altogether (a process that manages its own HIP settings).  Not done at load time any more.
    {
        static std::once_flag once;
        std::call_once(once, [] {
            const char *keep = getenv("MI355LZ4_KEEP_HW_QUEUES");
            if (!(keep && atoi(keep)) && !getenv("GPU_MAX_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "8", 0);
        });
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MI355LZ4_E_NO_DEVICE, "mi355lz4: no HIP device visible (this engine has no CPU path)");
    if (device < 0 || device >= n) return fail(MI355LZ4_E_ARG, "mi355lz4_create: device %d out of range", device);
    if (!device_is_gfx950(device))
        return fail(MI355LZ4_E_NO_DEVICE, "mi355lz4: device %d is not gfx950 (MI355X); kernels are gfx950-only", device);
    mi355lz4_ctx *c = new (std::nothrow) mi355lz4_ctx();
    if (!c) return fail(MI355LZ4_E_ARG, "out of host memory");
    c->device = device;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete c; return fail(MI355LZ4_E_HIP, "hipSetDevice: %s", hipGetErrorString(e)); }
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(MI355LZ4_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->ownStream = true;
    *out = c;
    return MI355LZ4_OK;
}

extern "C" void mi355lz4_destroy(mi355lz4_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->in, &c->slots, &c->dense, &c->out, &c->offA, &c->offB, &c->lenA, &c->lenB, &c->res, &c->scratch,
                      &c->tolPool, &c->tolMeta, &c->linkBuf, &c->ptrBuf, &c->seg[0].b, &c->seg[1].b, &c->seg[2].b, &c->seg[3].b, &c->tokBuf})
        dev_release(*b);
    if (c->linkEvent) hipEventDestroy(c->linkEvent);
    pin_release(c->pinStat);
    pin_release(c->pinIn);
    pin_release(c->pinOut);
    pin_release(c->pinMeta);
    if (c->sIn) hipStreamDestroy(c->sIn);
    if (c->sOut) hipStreamDestroy(c->sOut);
    for (hipStream_t &k : c->sK) if (k) hipStreamDestroy(k);
    if (c->ownStream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mi355lz4_set_stream(mi355lz4_ctx *c, void *s)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    if (c->ownStream && c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    c->stream = (hipStream_t)s;
    c->ownStream = false;
    return MI355LZ4_OK;
}
extern "C" void *mi355lz4_get_stream(mi355lz4_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int mi355lz4_set_linked_async(mi355lz4_ctx *c, int maxDecodedBlockSize)
{
    if (!c || maxDecodedBlockSize < 0) return fail(MI355LZ4_E_ARG, "mi355lz4_set_linked_async: bad arguments");
    c->linkedAsyncCap = maxDecodedBlockSize;
    return MI355LZ4_OK;
}

extern "C" int mi355lz4_set_segments(mi355lz4_ctx *c, int segs)
{
    if (!c || segs < -1 || segs > 64) return fail(MI355LZ4_E_ARG, "mi355lz4_set_segments: -1 (auto), 0 (off) or 2..64");
    c->segMode = segs;
    return MI355LZ4_OK;
}

extern "C" int mi355lz4_synchronize(mi355lz4_ctx *c)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MI355LZ4_OK;
}

// 1 when the library was built with the shelved experiments (make lib-exp): the tests ask before they use variant 3.
// Not in the public header.
extern "C" int mi355lz4_debug_has_experiments(void)
{
#ifdef MI355LZ4_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int mi355lz4_set_decoder(mi355lz4_ctx *c, int variant)
{
    // 0 = chosen per call, 1 = sequence at a time, 2 = lane-parallel (one wavefront per block), 4 = one workgroup per block
    // (decode_cu.hpp); 3 = the parse as a pass of its own (token lists), experiment builds only (make lib-exp)
    bool ok = c && (variant == 0 || variant == 1 || variant == 2 || variant == 4);
#ifdef MI355LZ4_EXPERIMENTS
    ok = ok || (c && variant == 3);
#endif
    if (!ok) return fail(MI355LZ4_E_ARG, "bad decoder variant");
    c->decoder = variant;
    return MI355LZ4_OK;
}

extern "C" int mi355lz4_set_linked_compress(mi355lz4_ctx *c, int on)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    c->linkedCompress = on ? 1 : 0;
    return MI355LZ4_OK;
}

// Diagnostic hook (not part of the public header): enable/read the lane-parallel decoder's
// phase counters.  enable != 0 switches the STATS kernel on (slower); out receives and resets
// PAR_STATS_COUNT counters.
extern "C" int mi355lz4_debug_stats(mi355lz4_ctx *c, int enable, unsigned long long *out)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->stats && out) HIP_TRY(hipMemcpy(out, c->stats, PAR_STATS_COUNT * 8, hipMemcpyDeviceToHost));
    if (enable && !c->stats) HIP_TRY(hipMalloc((void **)&c->stats, PAR_STATS_COUNT * 8));
    if (c->stats) HIP_TRY(hipMemset(c->stats, 0, PAR_STATS_COUNT * 8));
    if (!enable && c->stats) { hipFree(c->stats); c->stats = nullptr; }
    return MI355LZ4_OK;
}

// Diagnostic hook (not part of the public header): the run-in decode's adaptive state {runinLong, runinLongOk, runinSkip} and, in
// get[3], the dictionary share the last linked call sampled (millionths; -1: none) and, in get[4], how the last linked call was
// finished: 0 no block needed its dictionary, 1 short runs walked, 2 run-in decode, 3 long run-in decode, 4 run-in decode given up
// and the lists/pointer passes, 5 the lists/pointer passes (or the walk) at once, 6 big blocks by the workgroup form against guessed dictionaries.  get (may be null, 5 ints) receives it; set (may
// be null, 3 ints) replaces the state.  Lets a test drive default -> long -> skip -> probe.
extern "C" int mi355lz4_debug_runin_state(mi355lz4_ctx *c, int *get, const int *set)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    if (get) { get[0] = c->runinLong ? 1 : 0; get[1] = c->runinLongOk; get[2] = c->runinSkip; get[3] = c->runinShareE6; get[4] = c->linkedPath; }
    if (set) { c->runinLong = set[0] != 0; c->runinLongOk = set[1]; c->runinSkip = set[2]; }
    return MI355LZ4_OK;
}

// Diagnostic hook (not part of the public header): the workgroup-per-block decoder writes 16 words per block of the
// next calls to devBuf (caller-owned device memory, 64 bytes per block; null switches it off): decode_cu.hpp, `dbg`.
extern "C" int mi355lz4_debug_cu(mi355lz4_ctx *c, uint32_t *devBuf)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    c->cuDbg = devBuf;
    return MI355LZ4_OK;
}

// Diagnostic hook (not part of the public header): what the tolerant pass of the last linked single-stream call
// left behind.  out[0] = dependent blocks that got a list, out[1] = entries in all lists, out[2] = longest
// list, out[3] = blocks whose list overflowed, out[4] = dependent blocks without a list.
extern "C" int mi355lz4_debug_tol_stats(mi355lz4_ctx *c, int nBlocks, long long *out)
{
    if (!c || !out || nBlocks <= 0 || !c->tolMeta.p) return fail(MI355LZ4_E_ARG, "debug_tol_stats: nothing to report");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<int32_t> m((size_t)nBlocks * 3 + 4);
    HIP_TRY(hipMemcpy(m.data(), c->tolMeta.p, m.size() * 4, hipMemcpyDeviceToHost));
    const int regionsUsed = m[0];
    for (int i = 0; i < 8; i++) out[i] = 0;
    // only blocks the tolerant kernel visited have meaningful entries: it hands out regions 0..regionsUsed-1
    std::vector<char> seen((size_t)(regionsUsed > 0 ? regionsUsed : 0), 0);
    for (int b = 0; b < nBlocks; b++) {
        const int reg = m[4 + (size_t)b], cnt = m[4 + (size_t)nBlocks + b];
        if (reg >= 0 && reg < regionsUsed && !seen[(size_t)reg]) {
            seen[(size_t)reg] = 1;
            out[0]++; out[1] += cnt;
            if (cnt > out[2]) out[2] = cnt;
            if (cnt > 8192) out[3]++;
        }
    }
    out[4] = regionsUsed - out[0];
    // RPL_STATS builds: rounds, super-batches, replay cycles / 16
    out[5] = (unsigned)m[1]; out[6] = (unsigned)m[2]; out[7] = (unsigned)m[3];
    return MI355LZ4_OK;
}

extern "C" int mi355lz4_compress_bound(int n)
{
    if ((unsigned)n > (unsigned)MI355LZ4_MAX_INPUT_SIZE) return 0;
    return n + n / 255 + 16;
}

extern "C" size_t mi355lz4_slot_stride(int blockLen, int headerKind)
{
    size_t b = (size_t)mi355lz4_compress_bound(blockLen) + (size_t)headerKind;
    return (b + 15) & ~(size_t)15;
}

static int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MI355LZ4_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return MI355LZ4_OK;
}

// ---------------------------------------------------------------------------
// device-resident batched API
// ---------------------------------------------------------------------------
static int encode_device(mi355lz4_ctx *c, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                         uint64_t blockStride, int maxBlockLen, int nBlocks, int accel, int headerKind, uint8_t *slots,
                         size_t slotStride, int32_t *framedLen, int lookBack)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    if (nBlocks < 0 || (headerKind != 4 && headerKind != 8) || maxBlockLen < 0 ||
        (unsigned)maxBlockLen > (unsigned)MI355LZ4_MAX_INPUT_SIZE)
        return fail(MI355LZ4_E_ARG, "compress_batch_device: bad arguments");
    if (nBlocks == 0) return MI355LZ4_OK;
    if (!src && maxBlockLen > 0) return fail(MI355LZ4_E_ARG, "compress_batch_device: null src");
    if (!slots || !framedLen) return fail(MI355LZ4_E_ARG, "compress_batch_device: null output");
    if (slotStride < (size_t)mi355lz4_compress_bound(maxBlockLen) + (size_t)headerKind)
        return fail(MI355LZ4_E_CAPACITY, "compress_batch_device: slotStride %zu < bound", slotStride);
    if (accel < 1) accel = 1;                 // cbits/lz4.c:1577
    if (accel > 65537) accel = 65537;         // cbits/lz4.c:1578
    HIP_TRY(hipSetDevice(c->device));
    EncodeArgs a;
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.blockStride = blockStride;
    a.uniformLen = maxBlockLen; a.nBlocks = nBlocks; a.accel = accel; a.headerKind = headerKind;
    a.slots = slots; a.slotStride = slotStride; a.framedLen = framedLen;
    a.stats = c->stats;
    a.linked = c->linkedCompress; a.lookBack = lookBack;
    // Small batches: with fewer blocks than the chip has wave slots (256 CUs x 16), a block is cut into segments that
    // several waves compress at once (kernels.hip, "K2, small batches").  Segments of >= 4 KiB, at most 64 per block,
    // about two waves per slot in all; blocks of up to 4 MiB (24-bit positions in the records); independent blocks only.
    // MI355LZ4_SEG=0 turns it off, MI355LZ4_SEG=k forces k segments (tests).
    {
        static const int segEnv0 = [] { const char *e = getenv("MI355LZ4_SEG"); return e ? atoi(e) : -1; }();
        const int segEnv = c->segMode >= 0 ? c->segMode : segEnv0;
        int segs = 0;
        if (!a.linked && maxBlockLen >= 8192 && maxBlockLen <= (4 << 20) && segEnv != 0) {
            const long slots_ = 2L * 256 * 16;
            long want = segEnv > 0 ? segEnv : slots_ / (long)nBlocks;
            if (want > maxBlockLen / 4096) want = maxBlockLen / 4096;
            if (want > 64) want = 64;
            if (want >= 2) segs = (int)want;
        }
        if (segs >= 2) {
            EncodeSegArgs sa;
            sa.e = a;
            sa.segs = segs;
            sa.segLen = ((maxBlockLen + segs - 1) / segs + 63) & ~63;
            sa.listStride = (size_t)maxBlockLen / 4 + (size_t)segs + 2;
            const size_t listBytes = (size_t)nBlocks * sa.listStride * sizeof(uint64_t);
            const size_t cntBytes = (size_t)nBlocks * (size_t)segs * sizeof(uint32_t);
            // The record lists are twice the input.  Automatic mode only takes the segment path while they stay under
            // SEG_SCRATCH_MAX (a call of 2048 x 4 MiB would otherwise pin 20 GiB per stream until mi355lz4_destroy:
            // round-3 advisor finding); a forced count (tests, mi355lz4_set_segments(k)) is the caller's decision.
            const size_t SEG_SCRATCH_MAX = (size_t)1 << 30;
            const bool fits = segEnv > 0 || listBytes + 3 * cntBytes <= SEG_SCRATCH_MAX;
            // One scratch per stream the engine has been used on, four at the most: a fifth stream takes over the slot
            // that was used longest ago, once the work queued on that slot's stream is done with it.
            mi355lz4_ctx::SegScratch *slot = nullptr;
            if (fits) {
                for (int i = 0; i < c->nSeg; i++) if (c->seg[i].s == c->stream) slot = &c->seg[i];
                if (!slot && c->nSeg < 4) { slot = &c->seg[c->nSeg++]; slot->s = c->stream; }
                if (!slot) {
                    slot = &c->seg[0];
                    for (int i = 1; i < c->nSeg; i++) if (c->seg[i].tick < slot->tick) slot = &c->seg[i];
                    (void)hipStreamSynchronize(slot->s);
                    slot->s = c->stream;
                }
                slot->tick = ++c->segTick;
                // a scratch that a big forced call left behind is given back when a call needs less than a quarter of it
                if (slot->b.cap > SEG_SCRATCH_MAX && (listBytes + 3 * cntBytes + 256) * 4 < slot->b.cap) {
                    (void)hipStreamSynchronize(slot->s);
                    dev_release(slot->b);
                }
            }
            DevBuf *sb = slot ? &slot->b : nullptr;
            if (sb && dev_reserve(*sb, listBytes + 3 * cntBytes + 256) == 0) {
                sa.lists = (uint64_t *)sb->p;
                sa.segCount = (uint32_t *)((uint8_t *)sb->p + ((listBytes + 63) & ~(size_t)63));
                sa.segBytes = sa.segCount + (size_t)nBlocks * (size_t)segs;
                sa.segPrevEnd = (int32_t *)(sa.segBytes + (size_t)nBlocks * (size_t)segs);
                launch_encode_seg(sa, c->stream);
                return check_launch("encode launch");
            }
            (void)hipGetLastError();          // no scratch: the one-wave-per-block path needs none
        }
    }
    launch_encode(a, maxBlockLen > 65536, c->stream);
    return check_launch("encode launch");
}

extern "C" int mi355lz4_compress_batch_device(mi355lz4_ctx *c, const uint8_t *src, const uint64_t *srcOff,
                                              const int32_t *srcLen, uint64_t blockStride, int maxBlockLen,
                                              int nBlocks, int accel, int headerKind, uint8_t *slots,
                                              size_t slotStride, int32_t *framedLen)
{
    return encode_device(c, src, srcOff, srcLen, blockStride, maxBlockLen, nBlocks, accel, headerKind, slots, slotStride,
                         framedLen, 0);
}

extern "C" int mi355lz4_compact_device(mi355lz4_ctx *c, const uint8_t *slots, size_t slotStride,
                                       const int32_t *framedLen, int nBlocks, uint8_t *dense, size_t denseCap,
                                       uint64_t *denseOff)
{
    if (!c || nBlocks < 0 || !denseOff) return fail(MI355LZ4_E_ARG, "compact_device: bad arguments");
    if (nBlocks > 0 && (!slots || !framedLen || !dense)) return fail(MI355LZ4_E_ARG, "compact_device: null pointer");
    HIP_TRY(hipSetDevice(c->device));
    // The total is only known on the device: the copy kernel never writes at or past denseCap (blocks that
    // do not fit are skipped), and denseOff[nBlocks] still reports the bytes the full stream needs, so a
    // caller that sized `dense` below the worst case compares denseOff[nBlocks] with denseCap.
    launch_compact(slots, slotStride, framedLen, nBlocks, dense, denseCap, denseOff, c->stream);
    return check_launch("compact launch");
}

// The scratch of a linked decode (linkBuf, tolPool, tolMeta, ptrBuf) belongs to the engine and its second pass is
// left in flight on the stream the call was made on.  When the next linked decode comes on ANOTHER stream (the
// Python binding re-targets the engine to torch's current stream on every call), that stream first waits for the
// previous use; on the same stream the order is already there.
static void link_scratch_acquire(mi355lz4_ctx *c)
{
    if (c->linkBusy && c->linkEvent && c->linkStream != c->stream) (void)hipStreamWaitEvent(c->stream, c->linkEvent, 0);
}
static void link_scratch_release(mi355lz4_ctx *c)
{
    if (!c->linkEvent && hipEventCreateWithFlags(&c->linkEvent, hipEventDisableTiming) != hipSuccess) { c->linkEvent = nullptr; return; }
    if (hipEventRecord(c->linkEvent, c->stream) == hipSuccess) { c->linkStream = c->stream; c->linkBusy = true; }
}

// The data half of a linked decode (see LinkedPlan): everything that reads output bytes.
static int linked_finish(mi355lz4_ctx *c)
{
    if (!c->plan.active) return MI355LZ4_OK;
    c->plan.active = false;
    DecodeArgs a = c->plan.a;
    const int first = c->plan.first, last = c->plan.last, pool = c->plan.pool, seg = c->plan.seg;
    if (c->plan.split) {
        launch_linked_resolve_b(a, c->stream);
    } else {
        for (int p0 = first; p0 <= last; p0 += pool) {
            const int p1 = (last + 1 - p0 < pool) ? last + 1 : p0 + pool;
            a.segFirst = p0; a.segEnd = p1;
            launch_linked_tolerant(a, c->stream);
            for (int b = p0; b < p1; b += seg) {
                a.segFirst = b;
                a.segEnd = (p1 - b < seg) ? p1 : b + seg;
                launch_linked_resolve(a, c->stream);
            }
        }
    }
    link_scratch_release(c);
    return check_launch("decode launch");
}

// Whether a call of decoder variant 0 takes the workgroup-per-block decoder (decode_cu.hpp): a CU decodes a 64 KiB block in 0.09-0.11 ms
// where a wavefront takes 0.2-0.3, but 19 wavefronts share a CU.  Measured (device-resident, ms, workgroup / wavefront form; lzsynth):
// 64 KiB blocks: 256: 0.10 / 0.20, 512: 0.20 / 0.21, 768: 0.29 / 0.21; 16 KiB: 256: 0.045 / 0.074, 512: 0.083 / 0.075; 4 KiB: 160: 0.042 /
// 0.037 (a workgroup's fixed costs are 28 us a block).  So: up to one block per CU when blocks are not tiny, up to two when they are big --
// judged by the compressed bytes per block, which is all the host knows.  MI355LZ4_CU_BLOCKS = n: up to n blocks whatever their size, 0 = never.
static bool cu_auto(int nBlocks, uint64_t framedLen)
{
    static const int forced = [] {
        const char *e = getenv("MI355LZ4_CU_BLOCKS");
        return e ? atoi(e) : -1;
    }();
    if (forced >= 0) return nBlocks <= forced;
    const uint64_t avg = framedLen / (uint64_t)(nBlocks > 0 ? nBlocks : 1);
    if (avg < 3072) return false;
    return nBlocks <= 256 || (nBlocks <= 512 && avg >= 16384);
}

// run-in decode of long linked streams: blocks of 64 KiB of run-in, and the span (in 64 KiB) from which it is the default
#ifndef RUNIN_DEFAULT_64K
#define RUNIN_DEFAULT_64K 11
#endif
#ifndef RUNIN_MIN_SPAN
#define RUNIN_MIN_SPAN 9216
#endif
#define RUNIN_ROUNDS 8          // launches of pieces to be redone before the call is left to the pointer pass
#define RUNIN_LONG_64K 17      // the long run-in (blocks of 64 KiB): taken after the default one gave a call up for what the data is like
#define RUNIN_LONG_PROBE 32    // calls in a row finished with the long run-in before the default is tried again
#define RUNIN_BACKOFF 16       // linked calls that skip the run-in decode after the long one gave a call up as well
#define RUNIN_SHARE_LONG 0.306  // sampled share of bytes taken directly from the block before (k_dict_share) from which the long run-in is taken ...
#define RUNIN_SHARE_NEVER 0.60  // ... and from which the stream is taken to never forget its dictionary (pointer pass)
static int decode_device(mi355lz4_ctx *c, const uint8_t *framed, uint64_t framedLen, const uint64_t *blockOff,
                         int nBlocks, int headerKind, int fixedUncomp, int linked, uint8_t *out,
                         const uint64_t *outOff, const int32_t *outCap, int32_t *result, const uint8_t *dict0,
                         uint32_t dict0Len, const int32_t *streamFirst = nullptr, int nStreams = 0, int lookBack = 0,
                         bool splitOk = false, bool deferEnd = false)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    if (c->plan.active) return fail(MI355LZ4_E_ARG, "a linked decode begun with mi355lz4_decompress_linked_begin is still open");
    if (nBlocks < 0 || (headerKind != 4 && headerKind != 8) || fixedUncomp < 0)
        return fail(MI355LZ4_E_ARG, "decompress_batch_device: bad arguments");
    if (nBlocks == 0) return MI355LZ4_OK;
    if (!framed || !blockOff || !outOff || !result) return fail(MI355LZ4_E_ARG, "decompress_batch_device: null pointer");
    HIP_TRY(hipSetDevice(c->device));
    DecodeArgs a;
    a.framed = framed; a.framedLen = framedLen; a.blockOff = blockOff; a.nBlocks = nBlocks;
    a.headerKind = headerKind; a.fixedUncomp = fixedUncomp; a.linked = linked ? 1 : 0;
    a.out = out; a.outOff = outOff; a.outCap = outCap; a.result = result;
    a.dict0 = dict0; a.dict0Len = dict0Len;
    a.streamFirst = streamFirst; a.nStreams = nStreams; a.lookBack = lookBack;
    a.tolPool = nullptr; a.tolRegions = 0; a.tolPer = 0; a.tolCounter = nullptr; a.tolRegion = a.tolCount = a.tolSize = nullptr;
    a.linkStat = nullptr; a.segFirst = 0; a.segEnd = nBlocks; a.ptr = nullptr; a.ptrCap = 0; a.ptrCtl = nullptr;
    a.ptrBad = nullptr;
    a.asyncGate = 0;
    a.onlyBlk = -1;
    a.tokList = nullptr; a.tokCnt = nullptr; a.runList = nullptr; a.runCap = 0; a.cuDbg = c->cuDbg; a.cuBail = c->decoder == 0;
    a.cuSnap = nullptr; a.cuFlags = nullptr; a.cuRes = nullptr; a.cuPass = 0;
    a.ring = nullptr; a.ringStride = 0; a.zeroPage = nullptr; a.runPiece = 0; a.runIn = 0; a.runSpin = 0; a.runRound = 0;
    a.runRes = nullptr; a.runInfo = nullptr; a.runDirty = nullptr; a.runCtl = nullptr;
    const size_t nFlags = streamFirst ? (size_t)(nStreams > 0 ? nStreams : 1) : 1;
    int r;
    if (linked) {
        link_scratch_acquire(c);
        // the standalone pass counts the blocks that need their dictionary: {count, first, last, -, largest capacity}
        if ((r = dev_reserve(c->linkBuf, 64 + ptr_ctl_bytes() + 4 * nFlags)) || (r = pin_reserve(c->pinStat, 64))) return r;
        a.linkStat = (uint32_t *)c->linkBuf.p;
        HIP_TRY(hipMemsetAsync(a.linkStat, 0, 32, c->stream));
        HIP_TRY(hipMemsetAsync(a.linkStat + 1, 0xff, 4, c->stream));
    }
    // Big linked blocks (the path behind the first pass, below): armed here, so that the first launch goes straight on with that
    // path's pass 1 for the blocks that do not decode on their own.  What the host knows beforehand is the compressed size: from half
    // the path's block size on (a stream of blocks of half that size at a ratio of 2 would otherwise pay a pass it has no use for:
    // +0.85 ms for 512 blocks of 256 KiB); big blocks that compress better than that are armed behind the first pass (bigLate).
    bool bigPre = false, bigEligible = false;
    {
        const char *envBig = getenv("MI355LZ4_LINKED_BIG");
        const long bigKiB = envBig ? atol(envBig) : 512;
        const bool plainBig = !getenv("MI355LZ4_LINKED_PTR") && !getenv("MI355LZ4_LINKED_POOL_BLOCKS") && !getenv("MI355LZ4_LINKED_RUNS") &&
                              !getenv("MI355LZ4_LINKED_RUNIN") && !getenv("MI355LZ4_LINKED_ASYNC");
        bigEligible = linked && bigKiB > 0 && plainBig && !streamFirst && !splitOk && !deferEnd && lookBack >= 0 && !dict0 && c->decoder == 0 &&
                      !c->stats && c->linkedAsyncCap <= 0 && nBlocks >= 2 && nBlocks <= 512 && cu_auto(nBlocks, framedLen);
        if (bigEligible && framedLen / (uint64_t)nBlocks >= (uint64_t)bigKiB * 1024u / 2u) {
            const size_t metaBytes = 65536 + ((size_t)nBlocks * 2 + 4) * sizeof(uint32_t);
            if (dev_reserve(c->ptrBuf, (size_t)nBlocks * 65536u) == 0 && dev_reserve(c->tolMeta, metaBytes) == 0 &&
                hipMemsetAsync(c->tolMeta.p, 0, metaBytes, c->stream) == hipSuccess) {
                uint8_t *meta = (uint8_t *)c->tolMeta.p;
                a.zeroPage = meta; a.cuSnap = (uint8_t *)c->ptrBuf.p;
                a.cuFlags = (uint32_t *)(meta + 65536); a.cuRes = (int32_t *)(meta + 65536 + ((size_t)nBlocks + 4) * sizeof(uint32_t));
                a.cuPass = 1;
                bigPre = true;
            }
            (void)hipGetLastError();
        }
    }
    if (c->decoder == 1)
        launch_decode_seq(a, c->stream);
#ifdef MI355LZ4_EXPERIMENTS
    else if (c->decoder == 3 && dev_reserve(c->tokBuf, (size_t)(framedLen >> 1) + 192 + ((size_t)nBlocks + 1) * sizeof(int32_t)) == 0) {
        // experiment: the parse as a pass of its own (token lists), then the list-driven decoder
        a.tokCnt = (int32_t *)c->tokBuf.p;
        a.tokList = (uint8_t *)c->tokBuf.p + ((((size_t)nBlocks + 1) * sizeof(int32_t) + 63) & ~(size_t)63);
        launch_decode_tok(a, c->stream);
    }
#endif
    else if (!c->stats && (c->decoder == 4 || (c->decoder == 0 && cu_auto(nBlocks, framedLen))))
        // Calls that do not fill the GPU -- one workgroup per block instead of one wavefront (decode_cu.hpp; cu_auto above says
        // which calls those are).  (Variant 4 forces it for any number of blocks: the tests.)  A linked call's first pass is this
        // same standalone decode (decompressChunks always asks for linked = 1, and what this engine's compressor writes are
        // independent blocks): a block that needs its dictionary fails here as it does there -- 50 us later -- and is counted.
        launch_decode_cu(a, c->stream);
    else
        launch_decode_par(a, c->stats, c->stream);
    if (!linked) return check_launch("decode launch");
    // Linked streams.  Whether there is a second pass at all, and over which blocks, is decided here: the
    // call waits for the standalone pass (a stream of independent blocks pays this wait and nothing else).
    uint32_t *stat = (uint32_t *)c->pinStat.p;
    // Asynchronous form (mi355lz4_set_linked_async; MI355LZ4_LINKED_ASYNC=<largest decoded block size> for the tests):
    // no wait on the host.  The second pass is enqueued over ALL blocks, its kernels return at once when the first
    // pass counted no dependent block; what the wait would have told -- the range of dependent blocks and the largest
    // block -- is replaced by the whole call and the caller's bound.  One stream only (the streams call keeps the wait:
    // its choice between walk and pointer pass needs the counts).
    int asyncCap = c->linkedAsyncCap;
    if (const char *e = getenv("MI355LZ4_LINKED_ASYNC")) asyncCap = atoi(e);
    if (asyncCap > 0 && !streamFirst) {
        a.asyncGate = 1;
        stat[0] = (uint32_t)nBlocks; stat[1] = 0; stat[2] = (uint32_t)(nBlocks - 1); stat[3] = (uint32_t)nBlocks;
        stat[4] = (uint32_t)asyncCap;
    } else {
        launch_longest_stream(a, c->stream);
        HIP_TRY(hipMemcpyAsync(stat, a.linkStat, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->runinShareE6 = -1; c->linkedPath = 0;
    if (stat[0] == 0) { link_scratch_release(c); return check_launch("decode launch"); }
    int first = (int)stat[1], last = (int)stat[2];
    if (first < 0 || last >= nBlocks || first > last) {
        link_scratch_release(c);          // the first pass is in flight on linkBuf: the next linked call must be ordered behind it
        return fail(MI355LZ4_E_HIP, "decompress: bad failure range %d..%d", first, last);
    }
    // Big blocks (BlockMax1MB / BlockMax4MB streams, Config.hs:109-116), few enough for a CU each: every dependent block by the
    // workgroup-per-block decoder against a GUESS of its dictionary -- zeros, then what its predecessor's last 64 KiB were a pass ago --
    // until a pass changes none of those (kernels.hip, k_decode_cu_linked).  A block of 1 MiB forgets a wrong dictionary long before
    // its end, so two passes do as a rule, and the first of them has been made by the first launch (bigPre, above).
    // MI355LZ4_LINKED_BIG = 0: never; = n: blocks from n KiB on (default 512: smaller blocks' ends still carry the wrong dictionary,
    // pass after pass).  Anything the form cannot take (a failing block, CU_REDO, snapshots that do not settle in BIG_PASSES)
    // leaves the call to the passes below: the results so far live in scratch.
    bool bigLate = false;
    if (bigEligible && !bigPre && !a.asyncGate) {
        const char *envBig = getenv("MI355LZ4_LINKED_BIG");
        const long bigKiB = envBig ? atol(envBig) : 512;
        const size_t metaBytes = 65536 + ((size_t)nBlocks * 2 + 4) * sizeof(uint32_t);
        if ((uint64_t)stat[4] >= (uint64_t)bigKiB * 1024u && dev_reserve(c->ptrBuf, (size_t)nBlocks * 65536u) == 0 &&
            dev_reserve(c->tolMeta, metaBytes) == 0 && hipMemsetAsync(c->tolMeta.p, 0, metaBytes, c->stream) == hipSuccess) {
            uint8_t *meta = (uint8_t *)c->tolMeta.p;
            a.zeroPage = meta; a.cuSnap = (uint8_t *)c->ptrBuf.p;
            a.cuFlags = (uint32_t *)(meta + 65536); a.cuRes = (int32_t *)(meta + 65536 + ((size_t)nBlocks + 4) * sizeof(uint32_t));
            bigLate = true;
        }
        (void)hipGetLastError();
    }
    if (bigPre || bigLate) {
        const char *envBig = getenv("MI355LZ4_LINKED_BIG");
        const long bigKiB = envBig ? atol(envBig) : 512;
        if (!a.asyncGate && (uint64_t)stat[4] >= (uint64_t)bigKiB * 1024u) {
#define BIG_PASSES 6
#define BIG_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { link_scratch_release(c); return fail(MI355LZ4_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
            bool settled = false;
            int passes = 0;
            for (int pass = 1; pass <= BIG_PASSES && !settled; pass++) {
                a.cuPass = pass; passes = pass;
                launch_cu_linked(a, pass > 1 || bigLate, c->stream);
                if (hipGetLastError() != hipSuccess) break;
                if (pass == 1) continue;                                  // (every snapshot is new after the first pass)
                BIG_TRY(hipMemcpyAsync(stat + 10, a.cuFlags, 8, hipMemcpyDeviceToHost, c->stream));
                BIG_TRY(hipStreamSynchronize(c->stream));
                if (stat[11] != 0) break;                                 // a block this form cannot take
                settled = stat[10] == 0;
            }
            if (settled) {
                launch_cu_publish(a, c->stream);
                c->linkedPath = 6; c->runinShareE6 = passes;              // (diagnostics: path 6 reports its passes where the others report the sampled share)
                link_scratch_release(c);
                return check_launch("decode launch");
            }
#undef BIG_TRY
            (void)hipGetLastError();
        }
        a.zeroPage = nullptr; a.cuSnap = nullptr; a.cuFlags = nullptr; a.cuRes = nullptr; a.cuPass = 0;
    }
    // Few dependent blocks, in short runs (stat[5] = longest run of blocks without output): every run is walked by a
    // wave of its own with the exact decoder and its dictionary; no lists, no pointers.  MI355LZ4_LINKED_RUNS = longest run
    // taken this way (default 4; 0 = never; the tests force it for whole streams).
    {
        const char *envRuns = getenv("MI355LZ4_LINKED_RUNS");
        const unsigned runMax = envRuns ? (unsigned)atoi(envRuns) : 4u;
        const bool plain = !getenv("MI355LZ4_LINKED_PTR") && !getenv("MI355LZ4_LINKED_POOL_BLOCKS");
        if (!streamFirst && !a.asyncGate && !splitOk && !deferEnd && runMax > 0 && stat[5] >= 1 && stat[5] <= runMax &&
            (envRuns || plain)) {
            // the runs' first blocks as a list taken from the first pass's results (stat[6] = how many there are in the call)
            const int runs = (int)stat[6] > 0 ? (int)stat[6] : 1;
            if ((r = dev_reserve(c->tolMeta, ((size_t)runs + 1) * sizeof(int32_t)))) { link_scratch_release(c); return r; }
            a.runList = (int32_t *)c->tolMeta.p; a.runCap = runs;
            if (hipMemsetAsync(a.runList, 0, sizeof(int32_t), c->stream) != hipSuccess) {
                link_scratch_release(c);
                return fail(MI355LZ4_E_HIP, "decompress: the run list could not be cleared");
            }
            a.segFirst = first; a.segEnd = last + 1;
            launch_linked_runs(a, c->stream);
            c->linkedPath = 1;
            link_scratch_release(c);
            return check_launch("decode launch");
        }
    }
    // Long runs of dependent blocks -- a stream written by the reference's compressor is ONE such run -- in pieces of
    // consecutive blocks, every piece decoded from a few blocks in front of it ("run-in": by the time the wave reaches
    // the piece the dictionary it carries is the true one; checked against what the piece in front wrote, redone where it
    // is not: kernels.hip, "RUN-IN DECODE").  A call's serial chain is run-in + piece blocks (0.53 ms per 64 KiB of text), one
    // wave per piece, a ring of two blocks of scratch per piece.
    // MI355LZ4_LINKED_RUNIN: 0 = never, 1 = whenever it applies (the tests); MI355LZ4_LINKED_RUNIN_BLOCKS: blocks of
    // run-in; MI355LZ4_LINKED_RUNIN_PIECE: blocks per piece (default: as many as give every piece a wave slot of its own).
    {
        const char *envRun = getenv("MI355LZ4_LINKED_RUNIN"), *envPiece = getenv("MI355LZ4_LINKED_RUNIN_PIECE"),
                   *envBlocks = getenv("MI355LZ4_LINKED_RUNIN_BLOCKS");
        const bool plain = !getenv("MI355LZ4_LINKED_PTR") && !getenv("MI355LZ4_LINKED_POOL_BLOCKS") && !getenv("MI355LZ4_LINKED_RUNS");
        const int span0 = last - first + 1;
        const uint64_t per64 = ((uint64_t)stat[4] + 65535u) / 65536u > 0 ? ((uint64_t)stat[4] + 65535u) / 65536u : 1u;
        const uint64_t stride = per64 * 65536u;
        // (the engine's own linked compressor probes every position and takes half of a text block from the block before it, the
        // reference's a third: its streams forget a dictionary after 9 to 15 blocks instead of 5 to 12 and take the long run-in,
        // which pays from twice the span)
        // (the state decays with EVERY linked call that gets here, whatever path it then takes: an engine whose later streams are
        // shorter than the long run-in's threshold would otherwise never try the default again)
        if (!envRun && c->runinLong && ++c->runinLongOk >= RUNIN_LONG_PROBE) { c->runinLong = false; c->runinLongOk = 0; }
        bool longRun = c->runinLong && !envRun;
        // (a range begun with mi355lz4_decompress_linked_begin that has no seam to wait for -- the stream's first range --
        // is finished here like a plain call: _end and _end_last then find nothing left to do)
        // (a handful of huge blocks has the bytes but not the pieces: at least 64 dependent blocks; and a piece's ring is two
        // strides, so strides beyond 1 GiB -- one piece would pass the 2 GiB the rings may take -- stay with the pointer pass)
        bool useRunIn = !streamFirst && !a.asyncGate && (!(splitOk || deferEnd) || lookBack == 0) &&
                        (envRun ? atoi(envRun) != 0
                                : (plain && span0 >= 64 && 2u * stride <= ((uint64_t)1 << 31) &&
                                   (uint64_t)span0 * per64 >= (longRun ? 2 * RUNIN_MIN_SPAN : RUNIN_MIN_SPAN)));
        if (useRunIn && !envRun && c->runinSkip > 0) { c->runinSkip--; useRunIn = false; }
        // How long the stream remembers a missing dictionary is read off the DATA before the first try (what the engine has learnt
        // from calls given up -- above -- stays as the second opinion): over 32 blocks spread over the span, the share of the bytes of
        // a block's first 1024 sequences that matches take directly from the block before it (k_dict_share: tokens only, 0.1 ms; a
        // block's head leans on the block before it more than its body: whole blocks give 0.065 / 0.077 where the heads give 0.29 /
        // 0.32).  Measured (scripts/runin_share.py): the reference's linked text 0.291-0.292 (forgotten after 5 to 12 blocks: the
        // default run-in), the engine's own linked text 0.321-0.325 (9 to 15 blocks: the long one), Python sources written by the
        // reference 0.19, noise with a period just under 64 KiB 0.9 (never: pointer pass).  The two text writers are 10 % apart --
        // a threshold between them is a calibration on two generators, not a law; a stream on the wrong side of it costs what it
        // cost before this rule (the default run-in given up once, or the long one where the default would have done).
        if (useRunIn && !envRun) {
            const int nS = span0 < 32 ? span0 : 32, step = span0 / nS;
            a.segFirst = first;
            bool sampled = hipMemsetAsync(a.linkStat + 8, 0, 8, c->stream) == hipSuccess;
            if (sampled) launch_dict_share(a, step, nS, c->stream);
            sampled = sampled && hipMemcpyAsync(stat + 8, a.linkStat + 8, 8, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                      hipStreamSynchronize(c->stream) == hipSuccess;
            (void)hipGetLastError();
            if (sampled && stat[9] > 0) {
                const double share = (double)stat[8] / (double)stat[9];
                c->runinShareE6 = (int)(share * 1e6);
                if (share >= RUNIN_SHARE_NEVER) useRunIn = false;
                else if (share >= RUNIN_SHARE_LONG) longRun = true;
                if (longRun && (uint64_t)span0 * per64 < 2 * RUNIN_MIN_SPAN) useRunIn = false;
            }
        }
        if (useRunIn) {
            // run-in length: on text the 5th to 12th block of 64 KiB is the first without a byte of the missing dictionary
            // (scripts/runin_sim.py); bigger blocks carry it further in bytes -- 256 KiB: 4 blocks, 1 MiB: 2, measured
            const uint64_t run64 = longRun ? RUNIN_LONG_64K : RUNIN_DEFAULT_64K;
            int runIn = (envBlocks && atoi(envBlocks) > 0) ? atoi(envBlocks)
                        : (per64 == 1 ? (int)run64 : (int)((run64 + per64) / per64) + 1);
            if (runIn > 64) runIn = 64;
            // pieces: one wave slot each (256 CUs x 16 waves), and at most 2 GiB of rings
            uint64_t maxPieces = ((uint64_t)1 << 31) / (2u * stride);
            if (maxPieces < 1) maxPieces = 1;
            if (maxPieces > 4096) maxPieces = 4096;
            int piece = (int)(((uint64_t)span0 + maxPieces - 1) / maxPieces);
            if (envPiece && atoi(envPiece) > 0) piece = atoi(envPiece);
            if (piece < 1) piece = 1;
            const char *envSpin = getenv("MI355LZ4_LINKED_RUNIN_SPIN");
            a.runSpin = envSpin ? atoi(envSpin) : 10000;
            int segBlocks = (int)((maxPieces * (uint64_t)piece < (uint64_t)span0) ? maxPieces * (uint64_t)piece : (uint64_t)span0);
            const size_t nPiecesMax = ((size_t)segBlocks + piece - 1) / piece;
            const size_t metaBytes = 65536 + (size_t)segBlocks * 4 + nPiecesMax * 20 + 64;
            bool done = false;
            if (dev_reserve(c->ptrBuf, nPiecesMax * 2u * stride) == 0 && dev_reserve(c->tolMeta, metaBytes) == 0) {
                uint8_t *meta = (uint8_t *)c->tolMeta.p;
                a.zeroPage = meta; a.ring = (uint8_t *)c->ptrBuf.p; a.ringStride = stride; a.runPiece = piece; a.runIn = runIn;
                a.runRes = (int32_t *)(meta + 65536);
                a.runInfo = (int32_t *)(meta + 65536 + (size_t)segBlocks * 4);
                a.runDirty = (uint32_t *)(meta + 65536 + (size_t)segBlocks * 4 + nPiecesMax * 16);
                a.runCtl = (uint32_t *)(meta + 65536 + (size_t)segBlocks * 4 + nPiecesMax * 20);
                // (a failure in here leaves the linked scratch to the next call like every other error path: RUNIN_TRY; and a kernel
                // that did not launch must not read as "nothing left to do": runCtl was zeroed by the host)
#define RUNIN_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { link_scratch_release(c); return fail(MI355LZ4_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
                RUNIN_TRY(hipMemsetAsync(meta, 0x00, 65536, c->stream));
                done = true;
                for (int s0 = first; s0 <= last && done; s0 += segBlocks) {
                    a.segFirst = s0; a.segEnd = (s0 + segBlocks < last + 1) ? s0 + segBlocks : last + 1;
                    RUNIN_TRY(hipMemsetAsync(a.runCtl, 0, 8, c->stream));
                    launch_runin_decode(a, c->stream);
                    bool segDone = false, launched = hipGetLastError() == hipSuccess;
                    for (int round = 0; round < RUNIN_ROUNDS && !segDone && launched; round++) {
                        a.runRound = round;
                        if (round) RUNIN_TRY(hipMemsetAsync(a.runCtl, 0, 4, c->stream));
                        launch_runin_fix(a, c->stream);
                        if (hipGetLastError() != hipSuccess) { launched = false; break; }
                        RUNIN_TRY(hipMemcpyAsync(stat, a.runCtl, 8, hipMemcpyDeviceToHost, c->stream));
                        RUNIN_TRY(hipStreamSynchronize(c->stream));
                        if (stat[1] != 0) break;                 // a block failed with the dictionary it got, or a chain of dirty pieces
                        segDone = stat[0] == 0;
                    }
                    if (!launched) { stat[1] = 1u; segDone = false; }   // (left to the passes below, like a broken block)
                    done = segDone;
                    // given up for what the DATA is like (chains of pieces to redo, rounds that do not end), not for a broken block:
                    // the engine's next calls take the long run-in, or -- that was the long one -- RUNIN_BACKOFF of them do not try
                    if (!segDone && !(stat[1] & 1u) && !envRun) {
                        if (!longRun) c->runinLong = true;
                        else c->runinSkip = RUNIN_BACKOFF;
                        c->runinLongOk = 0;
                    }
                    if (segDone) launch_runin_publish(a, c->stream);
                }
            }
            (void)hipGetLastError();                             // (only a failed reservation is left to swallow here)
            a.ring = nullptr; a.zeroPage = nullptr; a.runRes = nullptr; a.runCtl = nullptr; a.runInfo = nullptr; a.runDirty = nullptr;
            a.runPiece = 0; a.runIn = 0;
            c->linkedPath = done ? (longRun ? 3 : 2) : 4;
            if (done) { link_scratch_release(c); return check_launch("decode launch"); }
            // Not finished this way (a broken block, rounds that run out, no scratch): the segments that did finish are final,
            // the one that did not has the first pass's results still; its blocks go through the passes below
            a.segFirst = 0; a.segEnd = nBlocks;
            RUNIN_TRY(hipMemsetAsync(a.linkStat, 0, 32, c->stream));
            RUNIN_TRY(hipMemsetAsync(a.linkStat + 1, 0xff, 4, c->stream));
            launch_link_stat(a, c->stream);
            RUNIN_TRY(hipMemcpyAsync(stat, a.linkStat, 32, hipMemcpyDeviceToHost, c->stream));
            RUNIN_TRY(hipStreamSynchronize(c->stream));
#undef RUNIN_TRY
            if (stat[0] == 0) { link_scratch_release(c); return check_launch("decode launch"); }
            first = (int)stat[1]; last = (int)stat[2];
            if (first < 0 || last >= nBlocks || first > last) {
                link_scratch_release(c);
                return fail(MI355LZ4_E_HIP, "decompress: bad failure range %d..%d", first, last);
            }
        }
    }
    // Lists of deferred matches for up to POOL_BLOCKS dependent blocks at a time (64 KiB each: one byte per output
    // byte) and source pointers for up to PTR_BLOCKS of them (four bytes per output byte); without the lists the
    // blocks are walked one after the other.  (Read per call: the tests shrink both to reach every seam.)
    const char *envPool = getenv("MI355LZ4_LINKED_POOL_BLOCKS"), *envPtr = getenv("MI355LZ4_LINKED_PTR"),
               *envSeg = getenv("MI355LZ4_LINKED_PTR_BLOCKS");
    if (c->linkedPath != 4) c->linkedPath = 5;
    const int poolMax = envPool ? atoi(envPool) : 16384;
    const int ptrMax = (envSeg && atoi(envSeg) > 0) ? atoi(envSeg) : 4096;
    const bool usePtr = !envPtr || atoi(envPtr) != 0;
    // Many short streams are walked side by side, one wavefront per stream, faster than their bytes are resolved
    // through pointers: a walk costs ~0.42 ms per dependent block of the longest stream (up to ~5000 streams at a
    // time), the pointer passes ~0.55 ms + 1.15 us per dependent block of the call (MI355X, text-like data).
    const bool walkStreams = streamFirst && !getenv("MI355LZ4_LINKED_PTR") &&
                             0.42 * (double)(stat[3] > 0 ? stat[3] - 1 : 0) * (double)(1 + nStreams / 5000) <
                                 0.55 + 1.15e-3 * (double)stat[0];
    // The defaults are sized for 64 KiB blocks; a bigger block takes as many list regions and pointers as it has
    // 64 KiB pieces (blocks beyond 4 MiB have no list and are walked), so fewer blocks make a segment.
    const int span = last - first + 1;
    const int per = (int)((stat[4] + 65535u) / 65536u) > 0 ? (int)((stat[4] + 65535u) / 65536u) : 1;
    const int poolBlocks = envPool ? poolMax : (poolMax / per > 0 ? poolMax / per : 1);
    const int ptrBlocks = (envSeg && atoi(envSeg) > 0) ? ptrMax : (ptrMax / per > 0 ? ptrMax / per : 1);
    const int pool = (poolMax > 0 && !walkStreams) ? ((span < poolBlocks) ? span : poolBlocks) : span;
    int seg = pool;
    if (poolMax > 0 && !walkStreams && dev_reserve(c->tolPool, (size_t)pool * per * tol_region_bytes()) == 0 &&
        dev_reserve(c->tolMeta, ((size_t)nBlocks * 3 + 4) * sizeof(int32_t)) == 0) {
        a.tolPool = c->tolPool.p; a.tolRegions = pool * per; a.tolPer = per;
        a.tolCounter = (uint32_t *)c->tolMeta.p;
        a.tolRegion = (int32_t *)c->tolMeta.p + 4;
        a.tolCount = a.tolRegion + nBlocks;
        a.tolSize = a.tolCount + nBlocks;
        const int pseg = (pool < ptrBlocks) ? pool : ptrBlocks;
        const size_t ptrs = ((size_t)pseg + 1) * per * 65536 + 65536;
        if (usePtr && ptrs < ((size_t)1 << 31) && dev_reserve(c->ptrBuf, ptrs * sizeof(uint32_t)) == 0) {
            a.ptr = (uint32_t *)c->ptrBuf.p; a.ptrCap = ptrs;
            a.ptrCtl = (uint8_t *)c->linkBuf.p + 64;
            a.ptrBad = (uint32_t *)((uint8_t *)c->linkBuf.p + 64 + ptr_ctl_bytes());
            seg = pseg;
        }
    }
    (void)hipGetLastError();          // scratch that could not be had is not an error: the serial walk needs none
    c->plan.active = true;
    c->plan.a = a; c->plan.first = first; c->plan.last = last; c->plan.pool = pool; c->plan.seg = seg;
    // When one segment covers every dependent block, the half of the second pass that reads no output byte can be
    // issued now: lists, pointers and the first jump pass depend on the tokens only.
    c->plan.split = splitOk && a.ptr && span <= seg && span <= pool;
    if (c->plan.split) {
        a.segFirst = first; a.segEnd = last + 1;
        launch_linked_tolerant(a, c->stream);
        launch_linked_resolve_a(a, c->stream);
        c->plan.a = a;
    }
    if (!deferEnd) return linked_finish(c);
    return check_launch("decode launch");
}

extern "C" int mi355lz4_decompress_batch_device(mi355lz4_ctx *c, const uint8_t *framed, uint64_t framedLen,
                                                const uint64_t *blockOff, int nBlocks, int headerKind,
                                                int fixedUncomp, int linked, uint8_t *out, const uint64_t *outOff,
                                                const int32_t *outCap, int32_t *result)
{
    return decode_device(c, framed, framedLen, blockOff, nBlocks, headerKind, fixedUncomp, linked, out, outOff,
                         outCap, result, nullptr, 0);
}

extern "C" int mi355lz4_decompress_linked_begin(mi355lz4_ctx *c, const uint8_t *framed, uint64_t framedLen,
                                               const uint64_t *blockOff, int nBlocks, int headerKind, int fixedUncomp,
                                               uint8_t *out, const uint64_t *outOff, const int32_t *outCap,
                                               int32_t *result, int lookBack)
{
    if (lookBack < 0 || lookBack > 1) return fail(MI355LZ4_E_ARG, "decompress_linked_begin: lookBack must be 0 or 1");
    return decode_device(c, framed, framedLen, blockOff, nBlocks, headerKind, fixedUncomp, 1, out, outOff, outCap, result,
                         nullptr, 0, nullptr, 0, lookBack, true, true);
}

extern "C" int mi355lz4_decompress_linked_end(mi355lz4_ctx *c)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    return linked_finish(c);
}

// The LAST block of a range begun with mi355lz4_decompress_linked_begin, ahead of _end: 1 = its bytes are final (the
// caller may pass them on and call _end at leisure), 0 = not available this way (call _end first).
extern "C" int mi355lz4_decompress_linked_end_last(mi355lz4_ctx *c)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->plan.active) return 1;                           // no block of the range needed its dictionary: all final
    if (!c->plan.split || !c->plan.a.ptrCtl || c->plan.a.streamFirst) return 0;
    DecodeArgs a = c->plan.a;
    const int last = a.nBlocks - 1;
    int r;
    if ((r = pin_reserve(c->pinStat, 48))) return r;
    // {lastOpen, the stream's flag, the last block's standalone result, whether it has a list}
    uint32_t *stat = (uint32_t *)c->pinStat.p + 8;
    uint8_t *ctl = (uint8_t *)a.ptrCtl;
    stat[0] = stat[1] = 0; stat[3] = 0;
    if (last >= a.segFirst && last < a.segEnd) {
        HIP_TRY(hipMemsetAsync(ctl + ptr_ctl_last_open_offset(), 0, sizeof(uint32_t), c->stream));
        a.onlyBlk = last;
        launch_linked_fetch_block(a, c->stream);
        HIP_TRY(hipMemcpyAsync(&stat[0], ctl + ptr_ctl_last_open_offset(), 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&stat[1], a.ptrBad, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&stat[3], a.tolRegion + last, 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(&stat[2], a.result + last, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((r = check_launch("decode launch"))) return r;
    const int32_t res = (int32_t)stat[2];
    if (res > 0) return 1;                                   // decoded on its own: final since the first pass
    // a dependent block: final only if the pointer pass took it and its chasing fetch left nothing open
    return (stat[0] == 0 && stat[1] == 0 && (int32_t)stat[3] >= 0) ? 1 : 0;
}

extern "C" int mi355lz4_decompress_streams_device(mi355lz4_ctx *c, const uint8_t *framed, uint64_t framedLen,
                                                  const uint64_t *blockOff, int nBlocks, int headerKind,
                                                  int fixedUncomp, const int32_t *streamFirst, int nStreams,
                                                  uint8_t *out, const uint64_t *outOff, const int32_t *outCap,
                                                  int32_t *result)
{
    if (nStreams < 0 || (nStreams > 0 && !streamFirst))
        return fail(MI355LZ4_E_ARG, "decompress_streams_device: bad stream table");
    if (nStreams == 0)                        // no streams: every block is decoded on its own
        return decode_device(c, framed, framedLen, blockOff, nBlocks, headerKind, fixedUncomp, 0, out, outOff, outCap,
                             result, nullptr, 0);
    return decode_device(c, framed, framedLen, blockOff, nBlocks, headerKind, fixedUncomp, 1, out, outOff, outCap,
                         result, nullptr, 0, streamFirst, nStreams);
}

extern "C" int mi355lz4_index_device(mi355lz4_ctx *c, const uint8_t *framed, uint64_t framedLen,
                                     const uint64_t *blockOff, int nBlocks, int headerKind, int fixedUncomp,
                                     uint64_t *outOff)
{
    if (!c || nBlocks < 0 || !outOff || (headerKind != 4 && headerKind != 8))
        return fail(MI355LZ4_E_ARG, "index_device: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    int r = dev_reserve(c->scratch, (size_t)(nBlocks + 1) * sizeof(int32_t));
    if (r) return r;
    launch_index(framed, framedLen, blockOff, nBlocks, headerKind, fixedUncomp, (int32_t *)c->scratch.p, outOff,
                 c->stream);
    return check_launch("index launch");
}

extern "C" int mi355lz4_generate_device(mi355lz4_ctx *c, int kind, uint8_t *dst, int blockLen, int nBlocks,
                                        uint64_t firstBlock, uint64_t blockStep, uint32_t litMax, uint32_t offMax)
{
    if (!c || kind < 0 || kind > 2 || blockLen < 0 || nBlocks < 0 || (nBlocks > 0 && blockLen > 0 && !dst))
        return fail(MI355LZ4_E_ARG, "generate_device: bad arguments");
    if (kind == 1 && (litMax == 0 || offMax == 0)) return fail(MI355LZ4_E_ARG, "generate_device: litMax/offMax must be > 0");
    if (nBlocks == 0 || blockLen == 0) return MI355LZ4_OK;
    HIP_TRY(hipSetDevice(c->device));
    launch_generate(kind, dst, blockLen, nBlocks, firstBlock, blockStep, litMax, offMax, c->stream);
    return check_launch("generate launch");
}

extern "C" int mi355lz4_interleave_device(mi355lz4_ctx *c, const uint8_t *local, const uint64_t *localOff,
                                          int nLocalBlocks, int rank, int nRanks, uint8_t *global,
                                          const uint64_t *globalOff)
{
    if (!c || nLocalBlocks < 0 || nRanks <= 0 || rank < 0 || rank >= nRanks)
        return fail(MI355LZ4_E_ARG, "interleave_device: bad arguments");
    if (nLocalBlocks == 0) return MI355LZ4_OK;
    if (!local || !localOff || !global || !globalOff) return fail(MI355LZ4_E_ARG, "interleave_device: null pointer");
    HIP_TRY(hipSetDevice(c->device));
    launch_interleave(local, localOff, nLocalBlocks, rank, nRanks, global, globalOff, c->stream);
    return check_launch("interleave launch");
}

// ---------------------------------------------------------------------------
// events (bench.py times kernels on the engine's own stream)
// ---------------------------------------------------------------------------
extern "C" int mi355lz4_event_create(void **ev)
{
    if (!ev) return fail(MI355LZ4_E_ARG, "null");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = (void *)e;
    return MI355LZ4_OK;
}
extern "C" int mi355lz4_event_destroy(void *ev)
{
    if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev));
    return MI355LZ4_OK;
}
extern "C" int mi355lz4_event_record(mi355lz4_ctx *c, void *ev)
{
    if (!c || !ev) return fail(MI355LZ4_E_ARG, "null");
    HIP_TRY(hipEventRecord((hipEvent_t)ev, c->stream));
    return MI355LZ4_OK;
}
extern "C" int mi355lz4_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!start || !stop || !ms) return fail(MI355LZ4_E_ARG, "null");
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return MI355LZ4_OK;
}


// ---------------------------------------------------------------------------
// Pipelined host-buffer calls.  A call is cut into groups of blocks; the H2D copy of
// group i+1, the kernels of group i and the D2H copy of group i-1 run on three streams, and the CPU copies
// between pageable caller memory and the pinned staging slots run meanwhile on the copy pool.  Caller
// memory that is already page-locked (hipHostMalloc / hipHostRegister, e.g. a torch pinned tensor) is
// handed to the DMA engines directly.
// ---------------------------------------------------------------------------
// The pipelined calls keep the caller's stream, two copy streams and two compute streams busy; HIP multiplexes
// streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue serialize
// (measured: compress 33 -> 41 GB/s with 8).  mi355lz4_create asks for 8 when it is the first HIP user of the
// process and nobody has set the variable (see there); nothing is touched at load time.

static bool pipe_trace()
{
    static const bool v = [] { const char *e = getenv("MI355LZ4_TRACE"); return e && atoi(e); }();
    return v;
}
static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
#define PTRACE(...) do { if (pipe_trace()) { fprintf(stderr, "[%10.3f] ", now_ms()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)

// Staging copies of a call of one or two groups go in pieces (decompress_host_pipelined): how many, and where piece q begins
static size_t sub_pieces(int groups, size_t bytes)
{
    static const int forced = [] { const char *e = getenv("MI355LZ4_STAGE_PIECES"); return e ? atoi(e) : 0; }();
    if (forced > 0) return (size_t)forced;
    // (measured, 10 MiB out / 3.6 MB in, ms per call: 1 piece 0.613, 2: 0.584, 4: 0.611, 8: 0.809 -- a piece costs ~30 us of calls and waits)
    return (groups > 2 || bytes < ((size_t)1 << 20)) ? 1 : 2;
}
static size_t piece_cut(size_t bytes, size_t q, size_t pieces)
{
    if (q >= pieces) return bytes;
    return (bytes / pieces * q) & ~(size_t)4095;
}

static size_t group_bytes()
{
    static const size_t v = [] {
        const char *e = getenv("MI355LZ4_GROUP_MB");
        const long mb = e ? atol(e) : 64;
        return (size_t)((mb < 1) ? 1 : (mb > 4096 ? 4096 : mb)) << 20;
    }();
    return v;
}

static bool host_range_is_pinned(const void *p, size_t n)
{
    if (!p || !n) return false;
    if (const char *e = getenv("MI355LZ4_NO_DIRECT")) if (atoi(e)) return false;
    hipPointerAttribute_t at;
    for (const uint8_t *q : {(const uint8_t *)p, (const uint8_t *)p + (n - 1)}) {
        if (hipPointerGetAttributes(&at, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

static int pipe_streams(mi355lz4_ctx *c)
{
    if (!c->sIn) HIP_TRY(hipStreamCreateWithFlags(&c->sIn, hipStreamNonBlocking));
    if (!c->sOut) HIP_TRY(hipStreamCreateWithFlags(&c->sOut, hipStreamNonBlocking));
    for (hipStream_t &k : c->sK) if (!k) HIP_TRY(hipStreamCreateWithFlags(&k, hipStreamNonBlocking));
    return 0;
}

// run the device-API entry points on another stream of the same engine for the duration of a scope
struct StreamSwap {
    mi355lz4_ctx *c;
    hipStream_t saved;
    StreamSwap(mi355lz4_ctx *ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { c->stream = s; }
    ~StreamSwap() { c->stream = saved; }
};

// events of one pipelined call, destroyed together
struct EventSet {
    std::vector<hipEvent_t> ev;
    ~EventSet() { for (hipEvent_t e : ev) if (e) hipEventDestroy(e); }
    int make(hipEvent_t *out)
    {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ev.push_back(e);
        *out = e;
        return 0;
    }
};

// never return from a pipelined call with work in flight that still references caller or ctx buffers
struct DrainOnExit {
    mi355lz4_ctx *c;
    ~DrainOnExit()
    {
        if (c->sIn) hipStreamSynchronize(c->sIn);
        hipStreamSynchronize(c->stream);
        for (hipStream_t k : c->sK) if (k) hipStreamSynchronize(k);
        if (c->sOut) hipStreamSynchronize(c->sOut);
    }
};

// ---------------------------------------------------------------------------
// host-buffer batched API
// ---------------------------------------------------------------------------
static inline int32_t host_le32(const uint8_t *p)
{
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

extern "C" int mi355lz4_compress_batch(mi355lz4_ctx *c, const uint8_t *const *src, const int32_t *srcLen,
                                       int nBlocks, int accel, int headerKind, uint8_t *framedOut, size_t cap,
                                       size_t *outLen, int32_t *blockFramedLen, int32_t *status)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    if (nBlocks < 0 || (headerKind != 4 && headerKind != 8) || !outLen)
        return fail(MI355LZ4_E_ARG, "compress_batch: bad arguments");
    *outLen = 0;
    if (nBlocks == 0) return MI355LZ4_OK;
    if (!src || !srcLen || !framedOut) return fail(MI355LZ4_E_ARG, "compress_batch: null pointer");
    HIP_TRY(hipSetDevice(c->device));

    size_t total = 0;
    int maxLen = 0;
    bool contiguous = true;                    // blocks back to back in caller memory, every start 16-aligned
    std::vector<uint64_t> offs((size_t)nBlocks);
    for (int i = 0; i < nBlocks; i++) {
        // compressChunk's size check, Internal/LZ4.hs:237-241 (BlockHasSize limit = LZ4_MAX_INPUT_SIZE)
        if (srcLen[i] < 0 || (unsigned)srcLen[i] > (unsigned)MI355LZ4_MAX_INPUT_SIZE)
            return fail(MI355LZ4_E_ARG, "compress_batch: block %d length %d exceeds the maximum block size", i, srcLen[i]);
        if (srcLen[i] > 0 && !src[i]) return fail(MI355LZ4_E_ARG, "compress_batch: block %d is null", i);
        offs[(size_t)i] = total;
        if (i > 0 && src[i] != src[0] + total) contiguous = false;
        // 16-aligned block starts; back to back for a linked stream (a block's dictionary lies directly in front of it)
        total += c->linkedCompress ? (size_t)srcLen[i] : (((size_t)srcLen[i] + 15) & ~(size_t)15);
        if (srcLen[i] > maxLen) maxLen = srcLen[i];
    }
    const size_t stride = mi355lz4_slot_stride(maxLen, headerKind);

    // groups of consecutive blocks, about group_bytes() of input each
    std::vector<int> gFirst;
    {
        size_t acc = 0;
        for (int i = 0; i < nBlocks; i++) {
            if (i == 0 || acc >= group_bytes()) { gFirst.push_back(i); acc = 0; }
            acc += (size_t)srcLen[i];
        }
        gFirst.push_back(nBlocks);
    }
    const int G = (int)gFirst.size() - 1;
    size_t maxIn = 0, maxBlocksG = 0;
    for (int g = 0; g < G; g++) {
        const size_t lo = offs[(size_t)gFirst[g]], hi = (gFirst[g + 1] < nBlocks) ? offs[(size_t)gFirst[g + 1]] : total;
        if (hi - lo > maxIn) maxIn = hi - lo;
        if ((size_t)(gFirst[g + 1] - gFirst[g]) > maxBlocksG) maxBlocksG = (size_t)(gFirst[g + 1] - gFirst[g]);
    }
    const bool directIn = contiguous && host_range_is_pinned(src[0], total);
    const bool directOut = host_range_is_pinned(framedOut, cap);

    int r;
    if ((r = pipe_streams(c))) return r;
    if (!directIn && (r = pin_reserve(c->pinIn, 2 * (maxIn + 16)))) return r;
    if (!directOut && (r = pin_reserve(c->pinOut, 2 * maxBlocksG * stride))) return r;
    if ((r = pin_reserve(c->pinMeta, (size_t)nBlocks * 4 + (size_t)G * 8))) return r;
    if ((r = dev_reserve(c->in, total + 16))) return r;
    if ((r = dev_reserve(c->offA, (size_t)nBlocks * 8))) return r;
    if ((r = dev_reserve(c->lenA, (size_t)nBlocks * 4))) return r;
    if ((r = dev_reserve(c->lenB, (size_t)nBlocks * 4))) return r;
    if ((r = dev_reserve(c->slots, (size_t)nBlocks * stride))) return r;
    if ((r = dev_reserve(c->dense, (size_t)nBlocks * stride))) return r;
    if ((r = dev_reserve(c->offB, ((size_t)nBlocks + (size_t)G) * 8))) return r;

    DrainOnExit drain{c};
    EventSet evs;
    HIP_TRY(hipMemcpyAsync(c->offA.p, offs.data(), (size_t)nBlocks * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->lenA.p, srcLen, (size_t)nBlocks * 4, hipMemcpyHostToDevice, c->stream));
    // offs / srcLen are pageable: make sure the copies have consumed them before they go away
    HIP_TRY(hipStreamSynchronize(c->stream));

    int32_t *flenPin = (int32_t *)c->pinMeta.p;                              // framed length of every block
    uint64_t *totPin = (uint64_t *)((uint8_t *)c->pinMeta.p + (size_t)nBlocks * 4);   // compressed bytes of every group
    std::vector<hipEvent_t> evIn((size_t)G), evK((size_t)G), evOut((size_t)G);
    std::vector<size_t> outAt((size_t)G, 0), outN((size_t)G, 0);
    size_t outPos = 0;
    int bad = 0;
    bool overflow = false;

    // Phase 1 -- input.  A block takes the encoder ~2 ms whatever else runs (it is latency-bound), and a
    // group of a few hundred blocks fills a fraction of the chip, so the kernels of consecutive groups go to
    // different compute streams and overlap each other as well as the copies.
    for (int g = 0; g < G; g++) {
        const int b0 = gFirst[g], b1 = gFirst[g + 1];
        const size_t lo = offs[(size_t)b0], hi = (b1 < nBlocks) ? offs[(size_t)b1] : total;
        if ((r = evs.make(&evIn[(size_t)g])) || (r = evs.make(&evK[(size_t)g])) || (r = evs.make(&evOut[(size_t)g]))) return r;
        if (directIn) {
            HIP_TRY(hipMemcpyAsync((uint8_t *)c->in.p + lo, src[0] + lo, hi - lo, hipMemcpyHostToDevice, c->sIn));
        } else {
            uint8_t *slot = (uint8_t *)c->pinIn.p + (size_t)(g & 1) * (maxIn + 16);
            if (g >= 2) HIP_TRY(hipEventSynchronize(evIn[(size_t)g - 2]));   // the copy that last read this slot
            // (in pieces for a call of one or two groups: the copy engine moves one while the host copies the next, sub_pieces)
            const int pieces = (int)sub_pieces(G, hi - lo);
            for (int q = 0; q < pieces; q++) {
                const int q0 = b0 + (int)((int64_t)(b1 - b0) * q / pieces), q1 = b0 + (int)((int64_t)(b1 - b0) * (q + 1) / pieces);
                if (q1 <= q0) continue;
                const size_t plo = offs[(size_t)q0], phi = (q1 < nBlocks) ? offs[(size_t)q1] : total;
                std::vector<CopyTask> tasks;
                for (int i = q0; i < q1; i++)
                    if (srcLen[i] > 0) tasks.push_back({slot + (offs[(size_t)i] - lo), src[i], (size_t)srcLen[i]});
                copy_pool().run(tasks);
                if (phi > plo) HIP_TRY(hipMemcpyAsync((uint8_t *)c->in.p + plo, slot + (plo - lo), phi - plo, hipMemcpyHostToDevice, c->sIn));
            }
        }
        HIP_TRY(hipEventRecord(evIn[(size_t)g], c->sIn));
        StreamSwap on(c, c->sK[g & 1]);
        HIP_TRY(hipStreamWaitEvent(c->stream, evIn[(size_t)g], 0));
        PTRACE("compress: group %d H2D enqueued (%zu bytes, direct %d)", g, hi - lo, (int)directIn);
        // (a linked stream: the last block of the group before is this group's first dictionary)
        r = encode_device(c, (const uint8_t *)c->in.p, (const uint64_t *)c->offA.p + b0,
                          (const int32_t *)c->lenA.p + b0, 0, maxLen, b1 - b0, accel, headerKind,
                          (uint8_t *)c->slots.p + (size_t)b0 * stride, stride, (int32_t *)c->lenB.p + b0, b0);
        if (r) return r;
        uint64_t *goff = (uint64_t *)c->offB.p + b0 + g;                       // b1 - b0 + 1 offsets of this group
        r = mi355lz4_compact_device(c, (const uint8_t *)c->slots.p + (size_t)b0 * stride, stride,
                                    (const int32_t *)c->lenB.p + b0, b1 - b0, (uint8_t *)c->dense.p + (size_t)b0 * stride,
                                    (size_t)(b1 - b0) * stride, goff);
        if (r) return r;
        HIP_TRY(hipMemcpyAsync(flenPin + b0, (const int32_t *)c->lenB.p + b0, (size_t)(b1 - b0) * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(totPin + g, goff + (b1 - b0), 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(evK[(size_t)g], c->stream));
    }
    // Phase 2 -- output, once the last input copy is through: on this link both directions together run
    // at ~39 GB/s each against 57 GB/s for one alone (scripts/pcie_rate.py) and the output is the small
    // side, so it only overlaps the tail of the kernels.  D2H of group t while the pool copies group t-1 out.
    // The output copies go to the INPUT copy stream, behind the last input copy: HIP multiplexes streams onto four
    // hardware queues by default, and the caller's stream, one copy stream and two compute streams use them up.
    // Data that barely compresses sends back as much as it took in: then the two directions do overlap (own stream).
    hipStream_t so = c->sIn;
    for (int t = 0; t < G + 1; t++) {
        if (t < G) {
            const int g = t, b0 = gFirst[g], b1 = gFirst[g + 1];
            HIP_TRY(hipEventSynchronize(evK[(size_t)g]));
            if (g == 0) {
                const size_t in0 = ((gFirst[1] < nBlocks) ? offs[(size_t)gFirst[1]] : total) - offs[0];
                if ((size_t)totPin[0] * 8 > in0 * 5) so = c->sOut;
            }
            PTRACE("compress: group %d kernels done", g);
            for (int i = b0; i < b1; i++) {
                const int32_t f = flenPin[i];
                if (blockFramedLen) blockFramedLen[i] = f;
                if (status) status[i] = (f > headerKind) ? f - headerKind : 0;
                if (f <= headerKind) bad++;
            }
            outAt[(size_t)g] = outPos;
            outN[(size_t)g] = (size_t)totPin[g];
            outPos += outN[(size_t)g];
            if (outPos > cap) overflow = true;
            if (!bad && !overflow && outN[(size_t)g]) {
                uint8_t *dst = directOut ? framedOut + outAt[(size_t)g] : (uint8_t *)c->pinOut.p + (size_t)(g & 1) * maxBlocksG * stride;
                HIP_TRY(hipMemcpyAsync(dst, (const uint8_t *)c->dense.p + (size_t)b0 * stride, outN[(size_t)g], hipMemcpyDeviceToHost, so));
            }
            HIP_TRY(hipEventRecord(evOut[(size_t)g], so));
        }
        if (t >= 1) {
            const int g = t - 1;
            HIP_TRY(hipEventSynchronize(evOut[(size_t)g]));
            PTRACE("compress: group %d D2H done (%zu bytes)", g, outN[(size_t)g]);
            if (!directOut && !bad && !overflow && outN[(size_t)g])
                copy_pool().copy(framedOut + outAt[(size_t)g], (const uint8_t *)c->pinOut.p + (size_t)(g & 1) * maxBlocksG * stride, outN[(size_t)g]);
        }
    }
    if (bad) return fail(MI355LZ4_E_BLOCK, "compress_batch: %d block(s) failed", bad);
    if (overflow) return fail(MI355LZ4_E_CAPACITY, "compress_batch: need %llu bytes, have %zu", (unsigned long long)outPos, cap);
    *outLen = outPos;
    return MI355LZ4_OK;
}

extern "C" int mi355lz4_index_host(const uint8_t *framedIn, size_t inLen, int headerKind, int fixedUncomp,
                                   uint64_t *blockOff, int32_t *uncompLen, int maxBlocks, int *nBlocks)
{
    if (!nBlocks || (headerKind != 4 && headerKind != 8) || maxBlocks < 0 || (inLen && !framedIn))
        return fail(MI355LZ4_E_ARG, "index_host: bad arguments");
    size_t pos = 0;
    int k = 0;
    *nBlocks = 0;
    while (pos < inLen) {
        if (pos + (size_t)headerKind > inLen)
            return fail(MI355LZ4_E_STREAM, "index_host: incomplete block header at offset %zu", pos);
        const int32_t cl = host_le32(framedIn + pos);
        const int32_t ul = (headerKind == 8) ? host_le32(framedIn + pos + 4) : fixedUncomp;
        if (cl <= 0) return fail(MI355LZ4_E_STREAM, "index_host: block %d has compressed length %d", k, cl);
        if (pos + (size_t)headerKind + (size_t)cl > inLen)
            return fail(MI355LZ4_E_STREAM, "index_host: incomplete block %d (needs %d bytes)", k, cl);
        if (k >= maxBlocks) return fail(MI355LZ4_E_CAPACITY, "index_host: more than %d blocks", maxBlocks);
        if (blockOff) blockOff[k] = pos;
        if (uncompLen) uncompLen[k] = ul;
        pos += (size_t)headerKind + (size_t)cl;
        k++;
    }
    *nBlocks = k;
    return MI355LZ4_OK;
}

static int decompress_host(mi355lz4_ctx *c, const uint8_t *framedIn, size_t inLen, int headerKind,
                           int fixedUncomp, int linked, const uint8_t *dict, int dictLen,
                           const int32_t *streamFirst, int nStreams,
                           uint8_t *out, size_t cap, size_t *outLen, int32_t *blockLen,
                           int maxBlocks, int *nBlocksOut);
static int decompress_host_pipelined(mi355lz4_ctx *c, const uint8_t *framedIn, size_t inLen, int headerKind,
                                     int fixedUncomp, int linked, const uint8_t *dict, int dictLen,
                                     const std::vector<uint64_t> &boff, const std::vector<int32_t> &ulen,
                                     const std::vector<uint64_t> &ooff, int n, uint8_t *out, size_t *outLen,
                                     int32_t *blockLen, int *nBlocksOut);

extern "C" int mi355lz4_decompress_batch(mi355lz4_ctx *c, const uint8_t *framedIn, size_t inLen, int headerKind,
                                         int fixedUncomp, int linked, const uint8_t *dict, int dictLen,
                                         uint8_t *out, size_t cap, size_t *outLen, int32_t *blockLen,
                                         int maxBlocks, int *nBlocksOut)
{
    return decompress_host(c, framedIn, inLen, headerKind, fixedUncomp, linked, dict, dictLen, nullptr, 0, out, cap,
                           outLen, blockLen, maxBlocks, nBlocksOut);
}

extern "C" int mi355lz4_decompress_streams(mi355lz4_ctx *c, const uint8_t *framedIn, size_t inLen, int headerKind,
                                           int fixedUncomp, const int32_t *streamFirst, int nStreams,
                                           uint8_t *out, size_t cap, size_t *outLen, int32_t *blockLen,
                                           int maxBlocks, int *nBlocksOut)
{
    if (nStreams < 0 || (nStreams > 0 && !streamFirst))
        return fail(MI355LZ4_E_ARG, "decompress_streams: bad stream table");
    for (int s = 0; s < nStreams; s++)
        if (streamFirst[s] < 0 || streamFirst[s + 1] < streamFirst[s])
            return fail(MI355LZ4_E_ARG, "decompress_streams: stream table is not ascending at %d", s);
    return decompress_host(c, framedIn, inLen, headerKind, fixedUncomp, 1, nullptr, 0, streamFirst, nStreams, out, cap,
                           outLen, blockLen, maxBlocks, nBlocksOut);
}

static int decompress_host(mi355lz4_ctx *c, const uint8_t *framedIn, size_t inLen, int headerKind,
                           int fixedUncomp, int linked, const uint8_t *dict, int dictLen,
                           const int32_t *streamFirst, int nStreams,
                           uint8_t *out, size_t cap, size_t *outLen, int32_t *blockLen,
                           int maxBlocks, int *nBlocksOut)
{
    if (!c) return fail(MI355LZ4_E_ARG, "null ctx");
    if (!outLen || !nBlocksOut || maxBlocks < 0) return fail(MI355LZ4_E_ARG, "decompress_batch: bad arguments");
    *outLen = 0;
    *nBlocksOut = 0;
    std::vector<uint64_t> boff((size_t)maxBlocks + 1);
    std::vector<int32_t> ulen((size_t)maxBlocks + 1);
    int n = 0;
    int r = mi355lz4_index_host(framedIn, inLen, headerKind, fixedUncomp, boff.data(), ulen.data(), maxBlocks, &n);
    if (r) return r;
    if (n == 0) return MI355LZ4_OK;
    HIP_TRY(hipSetDevice(c->device));

    // output layout: blocks back to back at their header (or fixed) capacity
    std::vector<uint64_t> ooff((size_t)n + 1);
    uint64_t total = 0;
    for (int i = 0; i < n; i++) {
        if (ulen[(size_t)i] < 0) return fail(MI355LZ4_E_STREAM, "decompress_batch: block %d has negative size", i);
        ooff[(size_t)i] = total;
        total += (uint64_t)ulen[(size_t)i];
    }
    ooff[(size_t)n] = total;
    // the pipelined path writes every block at its capacity offset: it needs room for that layout and one stream
    if (!streamFirst && cap >= total && total > 0)
        return decompress_host_pipelined(c, framedIn, inLen, headerKind, fixedUncomp, linked, dict, dictLen, boff, ulen,
                                         ooff, n, out, outLen, blockLen, nBlocksOut);
    if ((r = dev_reserve(c->in, inLen + 16))) return r;
    if ((r = dev_reserve(c->out, (size_t)total + 16))) return r;
    if ((r = dev_reserve(c->offA, (size_t)n * 8))) return r;
    if ((r = dev_reserve(c->offB, ((size_t)n + 1) * 8))) return r;
    if ((r = dev_reserve(c->res, (size_t)n * 4))) return r;
    // dictionary in force before block 0: only its last 64 KiB can be referenced, and
    // keeping exactly 64 KiB preserves the reference's "dictSize >= 64 KB => no offset
    // check" behaviour (cbits/lz4.c:1764)
    uint32_t dlen = 0;
    if (linked && dict && dictLen > 0) {
        dlen = (dictLen > 65536) ? 65536u : (uint32_t)dictLen;
        if ((r = dev_reserve(c->scratch, 65536 + 16))) return r;
        HIP_TRY(hipMemcpyAsync(c->scratch.p, dict + (dictLen - (int)dlen), dlen, hipMemcpyHostToDevice, c->stream));
    }
    if ((r = h2d_staged(c, c->in.p, framedIn, inLen))) return r;
    HIP_TRY(hipMemcpyAsync(c->offA.p, boff.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->offB.p, ooff.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int32_t *sfDev = nullptr;
    if (streamFirst) {
        if (nStreams == 0 || streamFirst[nStreams] > n)
            return fail(MI355LZ4_E_ARG, "decompress_streams: the stream table names block %d of %d", nStreams ? streamFirst[nStreams] : 0, n);
        if ((r = dev_reserve(c->lenA, ((size_t)nStreams + 1) * 4))) return r;
        HIP_TRY(hipMemcpyAsync(c->lenA.p, streamFirst, ((size_t)nStreams + 1) * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        sfDev = (const int32_t *)c->lenA.p;
    }
    r = decode_device(c, (const uint8_t *)c->in.p, inLen, (const uint64_t *)c->offA.p, n, headerKind, fixedUncomp,
                      linked, (uint8_t *)c->out.p, (const uint64_t *)c->offB.p, nullptr, (int32_t *)c->res.p,
                      dlen ? (const uint8_t *)c->scratch.p : nullptr, dlen, sfDev, nStreams);
    if (r) return r;
    std::vector<int32_t> res((size_t)n);
    HIP_TRY(hipMemcpyAsync(res.data(), c->res.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));

    // pack the decoded blocks back to back (a block may decode to fewer bytes than its capacity)
    int bad = 0;
    uint64_t need = 0;
    for (int i = 0; i < n; i++) {
        if (blockLen) blockLen[i] = res[(size_t)i];
        if (res[(size_t)i] < 0) bad++; else need += (uint64_t)res[(size_t)i];
    }
    *nBlocksOut = n;
    if (bad) return fail(MI355LZ4_E_BLOCK, "decompress_batch: %d block(s) failed", bad);
    if (need > cap) return fail(MI355LZ4_E_CAPACITY, "decompress_batch: need %llu bytes, have %zu", (unsigned long long)need, cap);
    if (need == total) {
        if ((r = d2h_staged(c, out, c->out.p, (size_t)total))) return r;
    } else {
        uint64_t w = 0;
        for (int i = 0; i < n; i++) {
            if (res[(size_t)i] > 0)
                HIP_TRY(hipMemcpyAsync(out + w, (const uint8_t *)c->out.p + ooff[(size_t)i], (size_t)res[(size_t)i],
                                       hipMemcpyDeviceToHost, c->stream));
            w += (uint64_t)res[(size_t)i];
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    *outLen = (size_t)need;
    return MI355LZ4_OK;
}


// One stream, output laid out at capacity offsets: groups of blocks flow through H2D -> decode (-> linked
// fixup of the group, which looks back into the groups before it) -> D2H on three streams.
static int decompress_host_pipelined(mi355lz4_ctx *c, const uint8_t *framedIn, size_t inLen, int headerKind,
                                     int fixedUncomp, int linked, const uint8_t *dict, int dictLen,
                                     const std::vector<uint64_t> &boff, const std::vector<int32_t> &ulen,
                                     const std::vector<uint64_t> &ooff, int n, uint8_t *out, size_t *outLen,
                                     int32_t *blockLen, int *nBlocksOut)
{
    const uint64_t total = ooff[(size_t)n];
    std::vector<int> gFirst;
    {
        size_t acc = 0;
        for (int i = 0; i < n; i++) {
            if (i == 0 || acc >= group_bytes()) { gFirst.push_back(i); acc = 0; }
            acc += (size_t)ulen[(size_t)i];
        }
        gFirst.push_back(n);
    }
    const int G = (int)gFirst.size() - 1;
    auto in_lo = [&](int b) -> size_t { return (b < n) ? (size_t)boff[(size_t)b] : inLen; };
    size_t maxIn = 0, maxOut = 0;
    for (int g = 0; g < G; g++) {
        maxIn = std::max(maxIn, in_lo(gFirst[g + 1]) - in_lo(gFirst[g]));
        maxOut = std::max(maxOut, (size_t)(ooff[(size_t)gFirst[g + 1]] - ooff[(size_t)gFirst[g]]));
    }
    const bool directIn = host_range_is_pinned(framedIn, inLen);
    const bool directOut = host_range_is_pinned(out, (size_t)total);
    int r;
    if ((r = pipe_streams(c))) return r;
    if (!directIn && (r = pin_reserve(c->pinIn, 2 * (maxIn + 16)))) return r;
    if (!directOut && (r = pin_reserve(c->pinOut, 2 * (maxOut + 16)))) return r;
    if ((r = pin_reserve(c->pinMeta, (size_t)n * 4))) return r;
    if ((r = dev_reserve(c->in, inLen + 16))) return r;
    if ((r = dev_reserve(c->out, (size_t)total + 16))) return r;
    if ((r = dev_reserve(c->offA, (size_t)n * 8))) return r;
    if ((r = dev_reserve(c->offB, ((size_t)n + 1) * 8))) return r;
    if ((r = dev_reserve(c->res, (size_t)n * 4))) return r;

    DrainOnExit drain{c};
    EventSet evs;
    uint32_t dlen = 0;
    if (linked && dict && dictLen > 0) {                    // see decompress_host: only the last 64 KiB matter
        dlen = (dictLen > 65536) ? 65536u : (uint32_t)dictLen;
        if ((r = dev_reserve(c->scratch, 65536 + 16))) return r;
        HIP_TRY(hipMemcpyAsync(c->scratch.p, dict + (dictLen - (int)dlen), dlen, hipMemcpyHostToDevice, c->stream));
    }
    // (boff / ooff outlive everything this call enqueues -- it drains its streams before it returns --, and the kernels that read the
    // device copies are ordered behind these on the same stream: no wait here)
    HIP_TRY(hipMemcpyAsync(c->offA.p, boff.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->offB.p, ooff.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));

    int32_t *resPin = (int32_t *)c->pinMeta.p;
    std::vector<hipEvent_t> evIn((size_t)G), evK((size_t)G), evOut((size_t)G);
    std::vector<std::vector<hipEvent_t>> evPiece((size_t)G);      // device-to-host copy in pieces (small calls): an event behind every piece but the last
    // input copy of group g: enqueued one group AHEAD of its kernels, because a linked decode waits on the
    // host for its first pass (decode_device) and the copy engine should be busy meanwhile
    auto stage_in = [&](int g) -> int {
        const int b0 = gFirst[g], b1 = gFirst[g + 1];
        const size_t lo = in_lo(b0), hi = in_lo(b1);
        int rr;
        if ((rr = evs.make(&evIn[(size_t)g])) || (rr = evs.make(&evK[(size_t)g])) || (rr = evs.make(&evOut[(size_t)g]))) return rr;
        if (directIn) {
            HIP_TRY(hipMemcpyAsync((uint8_t *)c->in.p + lo, framedIn + lo, hi - lo, hipMemcpyHostToDevice, c->sIn));
        } else {
            uint8_t *slot = (uint8_t *)c->pinIn.p + (size_t)(g & 1) * (maxIn + 16);
            if (g >= 2) HIP_TRY(hipEventSynchronize(evIn[(size_t)g - 2]));
            // (a call of one or two groups has no other group's copies to hide its own staging behind: in pieces, so that the
            // copy engine moves one piece while the host copies the next)
            const size_t pieces = sub_pieces(G, hi - lo);
            for (size_t q = 0; q < pieces; q++) {
                const size_t a = piece_cut(hi - lo, q, pieces), b = piece_cut(hi - lo, q + 1, pieces);
                copy_pool().copy(slot + a, framedIn + lo + a, b - a);
                HIP_TRY(hipMemcpyAsync((uint8_t *)c->in.p + lo + a, slot + a, b - a, hipMemcpyHostToDevice, c->sIn));
            }
        }
        HIP_TRY(hipEventRecord(evIn[(size_t)g], c->sIn));
        return 0;
    };
    if (G > 0 && (r = stage_in(0))) return r;
    for (int t = 0; t < G + 1; t++) {
        if (t < G) {                                                           // ---- stage A, group t
            const int g = t, b0 = gFirst[g], b1 = gFirst[g + 1];
            if (linked && g + 1 < G && (r = stage_in(g + 1))) return r;
            HIP_TRY(hipStreamWaitEvent(c->stream, evIn[(size_t)g], 0));
            // the group's blocks, with the whole framed buffer as bounds and the blocks before it as look-back
            r = decode_device(c, (const uint8_t *)c->in.p, inLen, (const uint64_t *)c->offA.p + b0, b1 - b0, headerKind,
                              fixedUncomp, linked, (uint8_t *)c->out.p, (const uint64_t *)c->offB.p + b0, nullptr,
                              (int32_t *)c->res.p + b0, dlen ? (const uint8_t *)c->scratch.p : nullptr, dlen, nullptr, 0, b0);
            if (r) return r;
            HIP_TRY(hipMemcpyAsync(resPin + b0, (const int32_t *)c->res.p + b0, (size_t)(b1 - b0) * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipEventRecord(evK[(size_t)g], c->stream));
            HIP_TRY(hipStreamWaitEvent(c->sOut, evK[(size_t)g], 0));
            const size_t olo = (size_t)ooff[(size_t)b0], ohi = (size_t)ooff[(size_t)b1];
            if (ohi > olo) {
                if (directOut) {
                    HIP_TRY(hipMemcpyAsync(out + olo, (const uint8_t *)c->out.p + olo, ohi - olo, hipMemcpyDeviceToHost, c->sOut));
                } else {
                    // slot g & 1 was emptied by stage C of group g - 2, one iteration ago
                    uint8_t *slot = (uint8_t *)c->pinOut.p + (size_t)(g & 1) * (maxOut + 16);
                    const size_t pieces = sub_pieces(G, ohi - olo);
                    for (size_t q = 0; q < pieces; q++) {
                        const size_t a = piece_cut(ohi - olo, q, pieces), b = piece_cut(ohi - olo, q + 1, pieces);
                        HIP_TRY(hipMemcpyAsync(slot + a, (const uint8_t *)c->out.p + olo + a, b - a, hipMemcpyDeviceToHost, c->sOut));
                        if (q + 1 < pieces) {
                            hipEvent_t e;
                            if ((r = evs.make(&e))) return r;
                            HIP_TRY(hipEventRecord(e, c->sOut));
                            evPiece[(size_t)g].push_back(e);
                        }
                    }
                }
            }
            HIP_TRY(hipEventRecord(evOut[(size_t)g], c->sOut));
        }
        if (t >= 1) {                                                          // ---- stage C, group t-1
            const int g = t - 1, b0 = gFirst[g], b1 = gFirst[g + 1];
            const size_t olo = (size_t)ooff[(size_t)b0], ohi = (size_t)ooff[(size_t)b1];
            const size_t pieces = evPiece[(size_t)g].size() + 1;
            for (size_t q = 0; q < pieces; q++) {
                // (the last piece's event is the group's: the results' copy and every piece lie in front of it)
                HIP_TRY(hipEventSynchronize(q + 1 < pieces ? evPiece[(size_t)g][q] : evOut[(size_t)g]));
                if (!directOut && ohi > olo) {
                    const uint8_t *slot = (const uint8_t *)c->pinOut.p + (size_t)(g & 1) * (maxOut + 16);
                    const size_t a = piece_cut(ohi - olo, q, pieces), b = piece_cut(ohi - olo, q + 1, pieces);
                    if (b > a) copy_pool().copy(out + olo + a, slot + a, b - a);
                }
            }
        }
        if (!linked && t + 1 < G && (r = stage_in(t + 1))) return r;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));

    int bad = 0;
    bool full = true;
    uint64_t need = 0;
    for (int i = 0; i < n; i++) {
        const int32_t ri = resPin[i];
        if (blockLen) blockLen[i] = ri;
        if (ri < 0) bad++; else need += (uint64_t)ri;
        if (ri != ulen[(size_t)i]) full = false;
    }
    *nBlocksOut = n;
    if (bad) return fail(MI355LZ4_E_BLOCK, "decompress_batch: %d block(s) failed", bad);
    if (!full) {
        // a block may decode to fewer bytes than its capacity: pack the blocks back to back, in place
        uint64_t w = 0;
        for (int i = 0; i < n; i++) {
            const uint64_t len = (uint64_t)resPin[i];
            if (len && w != ooff[(size_t)i]) memmove(out + w, out + ooff[(size_t)i], (size_t)len);
            w += len;
        }
    }
    *outLen = (size_t)need;
    return MI355LZ4_OK;
}

// ===========================================================================
// Legacy face: the 7 symbols Streamly.Internal.LZ4 imports today
// (src/Streamly/Internal/LZ4.hs:105-143).  One block per call: source
// compatible, correct, and slow by construction (a PCIe round trip per block) --
// the batched calls above are what INTEGRATION.md binds instead.
// ===========================================================================
static std::mutex g_engineMu;
static mi355lz4_ctx *g_engine = nullptr;

static mi355lz4_ctx *legacy_engine()
{
    std::lock_guard<std::mutex> lk(g_engineMu);
    if (!g_engine) {
        int dev = 0;
        if (const char *e = getenv("MI355LZ4_DEVICE")) dev = atoi(e);
        if (mi355lz4_create(&g_engine, dev) != MI355LZ4_OK) {
            fprintf(stderr, "mi355lz4: %s\n", g_err);
            g_engine = nullptr;
        }
    }
    return g_engine;
}

// One call = one block.  Since round 4 a call is ONE host-to-device copy (staged in page-locked memory), ONE kernel
// launch, ONE device-to-host copy (the output with its size word behind it) and ONE synchronisation.  The decoder keeps
// the previous block's output where it was decoded -- the next call decodes into the other of two buffers -- so the
// dictionary is never copied.  (Round 3: a heap-allocated frame or output vector, three or four synchronisations and a
// device-to-device copy of the dictionary per block.)  Still one PCIe round trip per block: compatibility, not speed.
struct LegacyBuf { void *p = nullptr; size_t cap = 0; };
static bool legacy_dev(LegacyBuf &b, size_t n)
{
    if (n <= b.cap) return true;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
    if (hipMalloc(&b.p, n * 2) != hipSuccess) { b.p = nullptr; return false; }
    b.cap = n * 2;
    return true;
}
static bool legacy_pin(LegacyBuf &b, size_t n)
{
    if (n <= b.cap) return true;
    if (b.p) (void)hipHostFree(b.p);
    b.p = nullptr; b.cap = 0;
    if (hipHostMalloc(&b.p, n * 2, hipHostMallocDefault) != hipSuccess) { b.p = nullptr; return false; }
    b.cap = n * 2;
    return true;
}
struct LZ4_stream_u {
    uint32_t magic;
    LegacyBuf inDev, slotDev, inPin, outPin;
};
struct LZ4_streamDecode_u {
    uint32_t magic;
    const uint8_t *dictDev;       // last <= 64 KiB of the previous block's output, inside outDev[1 - cur]
    uint32_t dictLen;
    int cur;
    LegacyBuf inDev, outDev[2], inPin, outPin;
};
// [blockOff = 0 | outOff = 0 | result | pad] travels in front of the framed block
#define LEGACY_PRE 32

extern "C" LZ4_stream_t *LZ4_createStream(void)
{
    LZ4_stream_u *s = new (std::nothrow) LZ4_stream_u();
    if (s) s->magic = 0x4C5A3443u;
    return (LZ4_stream_t *)s;
}
extern "C" int LZ4_freeStream(LZ4_stream_t *p)
{
    LZ4_stream_u *s = (LZ4_stream_u *)p;
    if (!s) return 0;
    if (s->inDev.p) (void)hipFree(s->inDev.p);
    if (s->slotDev.p) (void)hipFree(s->slotDev.p);
    if (s->inPin.p) (void)hipHostFree(s->inPin.p);
    if (s->outPin.p) (void)hipHostFree(s->outPin.p);
    delete s;
    return 0;
}

extern "C" LZ4_streamDecode_t *LZ4_createStreamDecode(void)
{
    LZ4_streamDecode_u *s = new (std::nothrow) LZ4_streamDecode_u();
    if (s) { s->magic = 0x4C5A3444u; s->dictDev = nullptr; s->dictLen = 0; s->cur = 0; }
    return (LZ4_streamDecode_t *)s;
}
extern "C" int LZ4_freeStreamDecode(LZ4_streamDecode_t *p)
{
    LZ4_streamDecode_u *s = (LZ4_streamDecode_u *)p;
    if (!s) return 0;
    if (s->inDev.p) (void)hipFree(s->inDev.p);
    for (int k = 0; k < 2; k++) if (s->outDev[k].p) (void)hipFree(s->outDev[k].p);
    if (s->inPin.p) (void)hipHostFree(s->inPin.p);
    if (s->outPin.p) (void)hipHostFree(s->outPin.p);
    delete s;
    return 0;
}

extern "C" int LZ4_compressBound(int inputSize) { return mi355lz4_compress_bound(inputSize); }

// Emits an independent block (never references earlier blocks), which the
// reference's linked decoder accepts.  Returns 0 on failure like the reference.
extern "C" int LZ4_compress_fast_continue(LZ4_stream_t *streamPtr, const char *src, char *dst, int srcSize,
                                          int dstCapacity, int acceleration)
{
    LZ4_stream_u *s = (LZ4_stream_u *)streamPtr;
    mi355lz4_ctx *c = legacy_engine();
    if (!c || !s || srcSize < 0 || dstCapacity <= 0 || !dst || (!src && srcSize > 0)) return 0;
    if ((unsigned)srcSize > (unsigned)MI355LZ4_MAX_INPUT_SIZE) return 0;          // cbits/lz4.c:1254
    std::lock_guard<std::mutex> lk(g_engineMu);
    if (hipSetDevice(c->device) != hipSuccess) return 0;
    const size_t stride = mi355lz4_slot_stride(srcSize, 4);
    if (!legacy_dev(s->inDev, (size_t)srcSize + 16) || !legacy_dev(s->slotDev, stride + 16) ||
        !legacy_pin(s->inPin, (size_t)srcSize + 16) || !legacy_pin(s->outPin, stride + 16))
        return 0;
    if (srcSize) {
        memcpy(s->inPin.p, src, (size_t)srcSize);
        if (hipMemcpyAsync(s->inDev.p, s->inPin.p, (size_t)srcSize, hipMemcpyHostToDevice, c->stream) != hipSuccess) return 0;
    }
    int32_t *lenDev = (int32_t *)((uint8_t *)s->slotDev.p + stride);
    if (encode_device(c, (const uint8_t *)s->inDev.p, nullptr, nullptr, (uint64_t)srcSize, srcSize, 1, acceleration, 4,
                      (uint8_t *)s->slotDev.p, stride, lenDev, 0) != MI355LZ4_OK)
        return 0;
    if (hipMemcpyAsync(s->outPin.p, s->slotDev.p, stride + 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    int32_t framed = 0;
    memcpy(&framed, (const uint8_t *)s->outPin.p + stride, 4);
    const int st = framed - 4;
    if (st <= 0 || st > dstCapacity) return 0;   // limitedOutput: cbits/lz4.c:1024-1027
    memcpy(dst, (const uint8_t *)s->outPin.p + 4, (size_t)st);
    return st;
}

// Linked semantics of cbits/lz4.c:2322-2359 for separately allocated blocks: the
// previous block's output stays on the device as the external dictionary.
extern "C" int LZ4_decompress_safe_continue(LZ4_streamDecode_t *p, const char *src, char *dst, int srcSize,
                                            int dstCapacity)
{
    LZ4_streamDecode_u *s = (LZ4_streamDecode_u *)p;
    mi355lz4_ctx *c = legacy_engine();
    if (!c || !s) return -1;
    if (!src) return -1;                                          // cbits/lz4.c:1752
    if (srcSize < 0 || dstCapacity < 0) return -1;
    if (srcSize == 0) return -1;                                   // cbits/lz4.c:1787 (and :1781 for cap==0)
    std::lock_guard<std::mutex> lk(g_engineMu);
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    const size_t inBytes = LEGACY_PRE + 4 + (size_t)srcSize;
    const size_t outPad = ((size_t)dstCapacity + 15) & ~(size_t)15;      // the result word sits behind the output
    LegacyBuf &outDev = s->outDev[s->cur];
    if (!legacy_dev(s->inDev, inBytes + 16) || !legacy_dev(outDev, outPad + 32) || !legacy_pin(s->inPin, inBytes) ||
        !legacy_pin(s->outPin, outPad + 16))
        return -1;
    // (growing outDev[cur] cannot move the dictionary: that lies in the OTHER buffer)
    uint8_t *hp = (uint8_t *)s->inPin.p;
    memset(hp, 0, LEGACY_PRE);
    const int32_t preset = s->dictLen ? -1 : 0;                    // a codec error: "this block wants its dictionary"
    memcpy(hp + 16, &preset, 4);
    hp[LEGACY_PRE + 0] = (uint8_t)srcSize; hp[LEGACY_PRE + 1] = (uint8_t)(srcSize >> 8);
    hp[LEGACY_PRE + 2] = (uint8_t)(srcSize >> 16); hp[LEGACY_PRE + 3] = (uint8_t)(srcSize >> 24);
    memcpy(hp + LEGACY_PRE + 4, src, (size_t)srcSize);
    if (hipMemcpyAsync(s->inDev.p, hp, inBytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return -1;
    int32_t *resDev = (int32_t *)((uint8_t *)outDev.p + outPad);
    DecodeArgs a;
    a.framed = (const uint8_t *)s->inDev.p + LEGACY_PRE; a.framedLen = 4 + (uint64_t)srcSize;
    a.blockOff = (const uint64_t *)s->inDev.p; a.nBlocks = 1;
    a.headerKind = 4; a.fixedUncomp = dstCapacity; a.linked = 1;
    a.out = (uint8_t *)outDev.p; a.outOff = (const uint64_t *)s->inDev.p + 1; a.outCap = nullptr; a.result = resDev;
    a.dict0 = s->dictLen ? s->dictDev : nullptr; a.dict0Len = s->dictLen;
    a.streamFirst = nullptr; a.nStreams = 0; a.lookBack = 0;
    a.tolPool = nullptr; a.tolRegions = 0; a.tolPer = 0; a.tolCounter = nullptr; a.tolRegion = a.tolCount = a.tolSize = nullptr;
    a.linkStat = nullptr; a.segFirst = 0; a.segEnd = 1; a.ptr = nullptr; a.ptrCap = 0; a.ptrCtl = nullptr;
    a.ptrBad = nullptr; a.asyncGate = 0; a.onlyBlk = -1; a.tokList = nullptr; a.tokCnt = nullptr; a.runList = nullptr; a.runCap = 0;
    a.cuDbg = nullptr; a.cuBail = 0; a.cuSnap = nullptr; a.cuFlags = nullptr; a.cuRes = nullptr; a.cuPass = 0;
    a.ring = nullptr; a.ringStride = 0; a.zeroPage = nullptr; a.runPiece = 0; a.runIn = 0; a.runSpin = 0; a.runRound = 0;
    a.runRes = nullptr; a.runInfo = nullptr; a.runDirty = nullptr; a.runCtl = nullptr;
    if (s->dictLen) {
        // the exact decoder with the dictionary in force, at once (a block that does not reach back decodes the same)
        if (hipMemcpyAsync(resDev, (const uint8_t *)s->inDev.p + 16, 4, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return -1;
        launch_linked_runs(a, c->stream);
    } else {
        launch_decode_par(a, nullptr, c->stream);
    }
    if (check_launch("decode launch") != MI355LZ4_OK) return -1;
    if (hipMemcpyAsync(s->outPin.p, outDev.p, outPad + 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    int32_t res = -1;
    memcpy(&res, (const uint8_t *)s->outPin.p + outPad, 4);
    if (res <= 0) return res;                                      // :2331 / :2353: context unchanged
    if (res > dstCapacity) return -1;
    memcpy(dst, s->outPin.p, (size_t)res);
    const uint32_t keep = (res > 65536) ? 65536u : (uint32_t)res;
    s->dictDev = (const uint8_t *)outDev.p + ((size_t)res - keep);
    s->dictLen = keep;
    s->cur ^= 1;
    return res;
}
