// decode_seq.hpp -- sequence-at-a-time LZ4 block decoder, one wavefront per block.
//
// Replaces (per block) LZ4_decompress_safe / LZ4_decompress_safe_forceExtDict,
// i.e. LZ4_decompress_generic (reference cbits/lz4.c:1737-2165) instantiated
// (endOnInputSize, decode_full_block, noDict|usingExtDict), as reached from
// LZ4_decompress_safe_continue (cbits/lz4.c:2322-2359).
//
// GPU shape: the token chain is wave-uniform (scalar registers, bytes fetched
// from a 256-byte register window with v_readlane); the 64 lanes move literal
// and match bytes.  Results, including the negative error codes -(ip-src)-1
// (cbits/lz4.c:2163), are identical to the reference: the `fast` flag tracks
// which of the reference's two loops (:1797-1924 fast, :1929-2151 safe) would be
// running, because they test their error conditions at different points.
//
// This decoder is the always-correct baseline: the lane-parallel decoder
// (decode_par.hpp) hands a block over to it whenever it meets anything unusual.
#pragma once

#include "lz4_device.hpp"

namespace lz4dev {

// Overlap-safe match copy.  `match` may be negative (source starts in the
// external dictionary, cbits/lz4.c:1883-1911): byte k of the source is
// dict[dictLen + match + k] while match + k < 0, dst[match + k] afterwards.
// offset == 0 yields zero bytes, as v1.9.3 does (cbits/lz4.c:2122-2130).
__device__ __forceinline__ void wave_copy_match(uint8_t *dstGeneric, int op, int match, uint32_t ml,
                                                uint32_t offset, const uint8_t *dictGeneric, uint32_t dictLen)
{
    LZ4_GLOBAL uint8_t *dst = as_global(dstGeneric);              // output and dictionary are global memory
    const LZ4_GLOBAL uint8_t *dict = as_global(dictGeneric);
    const uint32_t lane = (uint32_t)lane_id();
    wave_fence();
    if (offset == 0) {
#pragma unroll 1
        for (uint32_t j = lane; j < ml; j += LZ4_WAVE) dst[op + j] = 0;
    } else if (offset >= LZ4_WAVE) {
        // chunk c may read what chunk c-1 wrote: stores and loads of one wave stay in order
#pragma unroll 1
        for (uint32_t c = 0; c < ml; c += LZ4_WAVE) {
            uint32_t j = c + lane;
            if (j < ml) {
                int s = match + (int)j;
                dst[op + j] = (s < 0) ? dict[(int)dictLen + s] : dst[s];
            }
            wave_fence();
        }
    } else if (match >= 0 && ml > LZ4_WAVE) {
        // Short period, long match (runs, zero pages, records): the period -- at most 63 bytes, all of it in front of this
        // sequence -- is read ONCE, a byte a lane; every output byte is then one of those, fetched from its lane through the
        // crossbar, and the stores follow each other without a wait in between.  (The loop below it reads the source again
        // for every 64 bytes and waits for each read with the store before it still in flight: 2 us per 64 bytes.)
        const uint32_t mine = (lane < offset) ? (uint32_t)dst[match + (int)lane] : 0u;
        uint32_t idx = lane % offset;                               // source lane of byte `lane` of the current 64
        const uint32_t step = LZ4_WAVE % offset;                    // ... advances by 64 mod offset per 64 bytes
#pragma unroll 1
        for (uint32_t c = 0; c < ml; c += LZ4_WAVE) {
            const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)mine);
            if (c + lane < ml) dst[op + c + lane] = (uint8_t)b;
            idx += step;
            idx -= (idx >= offset) ? offset : 0u;
        }
    } else {
        // short period: every source byte lies in [match, op), complete before this sequence
#pragma unroll 1
        for (uint32_t j = lane; j < ml; j += LZ4_WAVE) {
            int s = match + (int)(j % offset);
            dst[op + j] = (s < 0) ? dict[(int)dictLen + s] : dst[s];
        }
    }
    wave_fence();
}

// ---------------------------------------------------------------------------
// Tolerant ("deferred copy") decode, used for ONE long linked stream: a block of a linked
// stream is decoded WITHOUT its dictionary, in parallel with every other block.  A match whose source lies
// before the block (in the previous block's output, cbits/lz4.c:1883-1911) cannot be copied yet: it is
// appended to the block's deferred list and its destination is marked in a taint bitmap (one bit per
// granule of output).  A match whose source touches a tainted granule is deferred too, and taints its own
// destination.  Everything else is copied as usual.  A later pass walks the stream in order and replays each
// block's list with the previous block's final output beside it (kernels.hip: k_decode_fixup_regions).
// ---------------------------------------------------------------------------
// One deferred match in 8 bytes: destination (22 bits), length (22 bits), offset (16 bits; the source is
// destination - offset and is negative when the match starts in the previous block's output).  Blocks of up to
// TOL_MAX_BLOCK bytes can be listed; a bigger one goes to the serial path.
#define TOL_MAX_BLOCK (1 << 22)
typedef uint64_t TolEntry;
__device__ __forceinline__ TolEntry tol_entry(int dpos, int src, uint32_t ml)
{
    return (uint64_t)(uint32_t)dpos | ((uint64_t)ml << 22) | ((uint64_t)(uint32_t)(dpos - src) << 44);
}
__device__ __forceinline__ void tol_unpack(TolEntry e, int &dpos, int &ml, int &spos)
{
    dpos = (int)(uint32_t)(e & 0x3fffffu);
    ml = (int)(uint32_t)((e >> 22) & 0x3fffffu);
    spos = dpos - (int)(uint32_t)((e >> 44) & 0xffffu);
}

struct TolCtx {
    uint32_t taint[128];        // 4096 granules
    TolEntry *list;             // global memory, cap entries
    uint32_t cap, count;        // count may run past cap (overflow: the block falls back to the serial path)
    uint32_t granShift;         // log2(granule bytes), >= 4: 16-byte granules for blocks of up to 64 KiB
};

// any tainted granule in output bytes [lo, hi) ?  (lo < hi; per-lane values)
__device__ __forceinline__ bool tol_tainted(const TolCtx *t, int lo, int hi)
{
    const uint32_t g0 = (uint32_t)lo >> t->granShift, g1 = (uint32_t)(hi - 1) >> t->granShift;
    bool any = false;
    for (uint32_t d = g0 >> 5; d <= (g1 >> 5); d++) {
        uint32_t m = ~0u;
        if (d == (g0 >> 5)) m &= ~0u << (g0 & 31u);
        if (d == (g1 >> 5)) m &= ~0u >> (31u - (g1 & 31u));
        any = any || ((t->taint[d] & m) != 0u);
    }
    return any;
}
__device__ __forceinline__ void tol_taint(TolCtx *t, int lo, int hi)
{
    const uint32_t g0 = (uint32_t)lo >> t->granShift, g1 = (uint32_t)(hi - 1) >> t->granShift;
    for (uint32_t d = g0 >> 5; d <= (g1 >> 5); d++) {
        uint32_t m = ~0u;
        if (d == (g0 >> 5)) m &= ~0u << (g0 & 31u);
        if (d == (g1 >> 5)) m &= ~0u >> (31u - (g1 & 31u));
        atomicOr(&t->taint[d], m);
    }
}
// wave-uniform arguments (the sequential decoder): defer this match?  If so it is recorded and tainted.
__device__ __forceinline__ bool tol_defer_uniform(TolCtx *t, int op, int match, uint32_t ml)
{
    wave_fence();
    bool defer = match < 0;
    if (!defer && ml > 0) {
        const int hi = min(match + (int)ml, op);            // bytes from op on are this match's own output
        defer = hi > match && tol_tainted(t, match, hi);
    }
    if (!defer || ml == 0) return false;
    if (lane_id() == 0) {
        if (t->count < t->cap) t->list[t->count] = tol_entry(op, match, ml);
        t->count += 1u;
        tol_taint(t, op, op + (int)ml);
    }
    wave_fence();
    return true;
}

// Resumable decoder state (all wave-uniform).
struct SeqState {
    int ip;      // next token, relative to the block's first compressed byte
    int op;      // bytes of output produced so far
    bool fast;   // which of the reference's two loops would be running (:1791)
};
#define SEQ_CONTINUE ((int)0x80000000)   // decode_seq_run: sequence budget used up, block not finished

// Decode up to maxSeq sequences (maxSeq <= 0: until the block ends) starting at st.
// Returns SEQ_CONTINUE, or the block's final result: decoded size >= 0 or the
// reference's negative code.  [bufLo, bufHi) bounds the readable framed buffer
// (reads outside return 0 instead of faulting).
// TOL is a template parameter (not a run-time test of tol) so that the strict instantiation keeps the register
// footprint the lane-parallel kernel's occupancy is built around (it is the one non-inlined call of that kernel).
template <bool TOL>
__device__ __forceinline__ int decode_seq_body(SeqState &st, int maxSeq, const uint8_t *src, int srcLen, uint8_t *dst, int cap,
                              const uint8_t *dict, uint32_t dictLen, const uint8_t *bufLo, const uint8_t *bufHi,
                              TolCtx *tol)
{
    const int iend = srcLen, oend = cap;
    // tolerant mode: parse as if a full 64 KiB dictionary were in force (no offset check, :1764); the replay
    // pass validates every deferred source against the dictionary that really is
    const bool useDict = TOL || ((dict != nullptr) && dictLen > 0);
    const bool checkOffset = !TOL && dictLen < 65536u;             // cbits/lz4.c:1764
    int ip = st.ip, op = st.op;
    // Lengths are 64-bit: a run of 0xFF length bytes in a multi-megabyte block reaches 2^31 and more, and the
    // reference compares them as size_t (cbits/lz4.c:1811-1818); the 255-run itself accumulates in 32 bits
    // (read_variable_length returns unsigned, :1707-1729).
    uint32_t token = 0, offset = 0, s = 0, acc = 0;
    int64_t ll = 0, ml = 0;
    int match = 0;
    bool fast = st.fast;
    int budget = maxSeq;

    InWindow win;
    win.lo = bufLo; win.hi = bufHi;
    win.load(src + ip);
    auto rd = [&](int pos) -> uint32_t {
        const uint8_t *p = src + pos;
        if (!win.covers(p, 1)) win.load(p);
        return win.byte_at(p);
    };

    for (;;) {
        if (maxSeq > 0) {
            if (budget == 0 && maxSeq == 1 && fast && ip + 24 < iend) {
                // Called for ONE sequence by the lane-parallel decoder, which takes only sequences whose lengths have at most
                // one extension byte.  If the NEXT sequence is not such a one either -- a second length byte: a literal run of
                // 270 bytes and more, a match of 274 and more -- it would come straight back here after a failed batch, a
                // call and a reload of the ring (about 15 us a sequence on data made of long runs: 144 GB/s); it is decoded
                // here and now instead.  (The peek reads the token and one length byte through the window.)
                const uint32_t t = rd(ip), l = t >> 4;
                if (l == 15u) { if (rd(ip + 1) == 255u) budget = 1; }
                else if ((t & 15u) == 15u && ip + 4 + (int)l + 24 < iend && rd(ip + 3 + (int)l) == 255u) budget = 1;
            }
            if (budget == 0) { st.ip = ip; st.op = op; st.fast = fast; return SEQ_CONTINUE; }
            budget--;
        }
        token = rd(ip); ip++;
        ll = (int64_t)(token >> 4);

        if (fast) {
            // ---------------- fast loop, :1797-1924 ----------------
            if (ll == 15) {
                if (ip >= iend - 15) goto error;                 // :1809-1810 (initial_error)
                acc = 0;
                do { s = rd(ip); ip++; acc += s; } while (s == 255 && ip < iend - 15);
                ll += (int64_t)acc;
                if (op + ll > (int64_t)oend - 32 || ip + ll > (int64_t)iend - 32) { fast = false; goto safe_literal_copy; } // :1818
            } else {
                if (ip > iend - 17) { fast = false; goto safe_literal_copy; } // :1831
            }
            wave_copy_bytes(dst + op, src + ip, (uint32_t)ll);
            ip += (int)ll; op += (int)ll;
            offset = rd(ip) | (rd(ip + 1) << 8); ip += 2;        // :1844
            match = op - (int)offset;
            ml = token & 15;
            if (ml == 15) {
                if (checkOffset && match + (int)dictLen < 0) goto error;      // :1853
                acc = 0;
                do { s = rd(ip); ip++; acc += s; if (ip >= iend - 4) goto error; } while (s == 255); // :1854-1855
                ml += (int64_t)acc + LZ4_MINMATCH;
                if (op + ml >= (int64_t)oend - 64) { fast = false; goto safe_match_copy; } // :1858
            } else {
                ml += LZ4_MINMATCH;
                if (op + ml >= (int64_t)oend - 64) { fast = false; goto safe_match_copy; } // :1863
            }
            if (checkOffset && match + (int)dictLen < 0) goto error;          // :1881
            if (match < 0) {
                if (!useDict) goto error;
                if (op + ml > (int64_t)oend - LZ4_LASTLITERALS) goto error;   // :1884-1889
            }
            if (!(TOL && tol_defer_uniform(tol, op, match, (uint32_t)ml)))
                wave_copy_match(dst, op, match, (uint32_t)ml, offset, dict, dictLen);
            op += (int)ml;
            continue;
        }

        // ---------------- safe loop, :1929-2151 ----------------
        if (ll != 15 && ip < iend - 16 && op <= oend - 32) {                  // shortcut :1944-1974
            wave_copy_bytes(dst + op, src + ip, (uint32_t)ll);
            op += (int)ll; ip += (int)ll;
            ml = token & 15;
            offset = rd(ip) | (rd(ip + 1) << 8); ip += 2;
            match = op - (int)offset;
            if (ml != 15 && offset >= 8 && match >= 0) {                      // :1959-1969
                if (!(TOL && tol_defer_uniform(tol, op, match, (uint32_t)ml + LZ4_MINMATCH)))
                    wave_copy_match(dst, op, match, (uint32_t)ml + LZ4_MINMATCH, offset, dict, dictLen);
                op += (int)ml + LZ4_MINMATCH;
                continue;
            }
            goto copy_match;                                                  // :1973
        }
        if (ll == 15) {
            if (ip >= iend - 15) goto error;                                  // :1979-1980
            acc = 0;
            do { s = rd(ip); ip++; acc += s; } while (s == 255 && ip < iend - 15);
            ll += (int64_t)acc;
        }
    safe_literal_copy:
        if (op + ll > (int64_t)oend - LZ4_MFLIMIT || ip + ll > (int64_t)iend - (2 + 1 + LZ4_LASTLITERALS)) { // :1991
            if (ip + ll != (int64_t)iend || op + ll > (int64_t)oend) goto error; // :2031-2036
            wave_copy_bytes(dst + op, src + ip, (uint32_t)ll);
            ip += (int)ll; op += (int)ll;
            break;                                                            // :2046
        }
        wave_copy_bytes(dst + op, src + ip, (uint32_t)ll);                    // :2050
        ip += (int)ll; op += (int)ll;
        offset = rd(ip) | (rd(ip + 1) << 8); ip += 2;                         // :2055
        match = op - (int)offset;
        ml = token & 15;
    copy_match:
        if (ml == 15) {
            acc = 0;
            do { s = rd(ip); ip++; acc += s; if (ip >= iend - 4) goto error; } while (s == 255); // :2064-2065
            ml += (int64_t)acc;
        }
        ml += LZ4_MINMATCH;
    safe_match_copy:
        if (checkOffset && match + (int)dictLen < 0) goto error;              // :2073
        if (match < 0) {
            if (!useDict) goto error;
            if (op + ml > (int64_t)oend - LZ4_LASTLITERALS) goto error;       // :2076-2079
        } else if (op + ml > (int64_t)oend - 12) {                            // :2137
            if (op + ml > (int64_t)oend - LZ4_LASTLITERALS) goto error;       // :2139
        }
        if (!(TOL && tol_defer_uniform(tol, op, match, (uint32_t)ml)))
            wave_copy_match(dst, op, match, (uint32_t)ml, offset, dict, dictLen);
        op += (int)ml;
    }
    wave_fence();
    return op;                                                                // :2156

error:
    return -ip - 1;                                                           // :2163
}

// The two out-of-line entry points.  They are plain (non-template) functions on purpose: the strict one is the
// single non-inlined call of the lane-parallel kernel, whose occupancy is built around its register footprint.
#ifndef SEQ_RUN_ATTR
#define SEQ_RUN_ATTR __attribute__((noinline))
#endif
__device__ SEQ_RUN_ATTR int decode_seq_run(SeqState &st, int maxSeq, const uint8_t *src, int srcLen, uint8_t *dst, int cap,
                              const uint8_t *dict, uint32_t dictLen, const uint8_t *bufLo, const uint8_t *bufHi)
{
    return decode_seq_body<false>(st, maxSeq, src, srcLen, dst, cap, dict, dictLen, bufLo, bufHi, nullptr);
}
__device__ __attribute__((noinline)) int decode_seq_run_tol(SeqState &st, int maxSeq, const uint8_t *src, int srcLen, uint8_t *dst, int cap,
                              const uint8_t *bufLo, const uint8_t *bufHi, TolCtx *tol)
{
    return decode_seq_body<true>(st, maxSeq, src, srcLen, dst, cap, nullptr, 0, bufLo, bufHi, tol);
}
template <bool TOL>
__device__ __forceinline__ int decode_seq_dispatch(SeqState &st, int maxSeq, const uint8_t *src, int srcLen, uint8_t *dst, int cap,
                              const uint8_t *dict, uint32_t dictLen, const uint8_t *bufLo, const uint8_t *bufHi, TolCtx *tol)
{
    if constexpr (TOL) return decode_seq_run_tol(st, maxSeq, src, srcLen, dst, cap, bufLo, bufHi, tol);
    else return decode_seq_run(st, maxSeq, src, srcLen, dst, cap, dict, dictLen, bufLo, bufHi);
}

// Decode one whole block.  All arguments are wave-uniform.
template <bool TOL = false>
#ifndef SEQ_BLOCK_ATTR
#define SEQ_BLOCK_ATTR
#endif
__device__ SEQ_BLOCK_ATTR int decode_block_seq(const uint8_t *src, int srcLen, uint8_t *dst, int cap,
                                const uint8_t *dict, uint32_t dictLen, const uint8_t *bufLo,
                                const uint8_t *bufHi, TolCtx *tol = nullptr)
{
    if (cap == 0) {                                              // :1781-1785
        InWindow w0; w0.lo = bufLo; w0.hi = bufHi; w0.load(src);
        return (srcLen == 1 && w0.byte_at(src) == 0) ? 0 : -1;
    }
    if (srcLen == 0) return -1;                                  // :1787
    SeqState st;
    st.ip = 0; st.op = 0;
    st.fast = cap >= 64;                                         // :1791
    return decode_seq_dispatch<TOL>(st, 0, src, srcLen, dst, cap, dict, dictLen, bufLo, bufHi, tol);
}

} // namespace lz4dev
