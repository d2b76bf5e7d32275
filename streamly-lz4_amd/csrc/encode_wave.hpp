// encode_wave.hpp -- LZ4 block compressor, one wavefront per block.
//
// Replaces (per block) LZ4_compress_fast_continue -> LZ4_compress_generic
// (reference cbits/lz4.c:1565-1637, 851-1240) for INDEPENDENT blocks.
//
// It is not a transcription of the reference's serial probe loop.  The wave
// probes 64 positions at once:
//   * lane i hashes the 5 bytes at p + i*step with the reference's hash
//     (cbits/lz4.c:706-716, hashLog 12) and looks its candidate up in a
//     4096-entry table held in LDS (16-bit entries: positions for blocks <= 64 KiB,
//     positions modulo 64 Ki above; cbits/lz4.h:578-580 is the table being replaced);
//   * a ballot picks the first lane whose candidate verifies (4 equal bytes
//     within 65535, cbits/lz4.c:1003-1012); lanes up to and including it
//     publish their positions to the table -- later lanes do not, so the
//     table never holds a position ahead of the parse;
//   * the match is extended backwards ("catch up", :1019) and forwards
//     (LZ4_count, :603-626) by ballot + count-trailing-zeros, 64 bytes a step;
//   * token / lengths / offset are emitted exactly as :1022-1046, :1065-1135,
//     literals are copied by all lanes.
// Acceleration keeps the reference meaning (:634, :957-967): the probe stride
// is (accel*64 + misses) >> 6, with 64 misses charged per fruitless window; the
// first three probes after a match are adjacent, as in the reference.
// End-of-block rules (:214-221, :883-884): inputs < 13 bytes are all literals,
// no match starts within the last 12 bytes, the last 5 bytes are literals.
//
// Output is a valid LZ4 block that any LZ4 decoder (incl. the reference's
// linked decoder) accepts; bytes differ from the reference's (H3 in SURVEY.md),
// sizes are compared in tests/bench.
#pragma once

#include "lz4_device.hpp"

namespace lz4dev {

typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));

__device__ __forceinline__ uint32_t hash5(uint64_t v)
{
    return (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 12));
}
// The next four bits of the same product are kept beside the position as a TAG: a candidate whose tag differs is not
// read.  With the reference's insertion policy most table entries a probe meets are unrelated older positions; their
// verification reads were 23x the input in HBM fetches (47 scattered reads per 64-position window for 2.4 matches).
// Four bits remove 15 of 16 of them (run heads per window 47 -> 16 on lzsynth, 61 -> 17 on text); a candidate that
// passes is a real match nine times in ten, which is what lets the pipelined finder below skip the separate
// verification round trip.  Tags are nibbles behind the table (table + 4096 entries), eight per 32-bit word, written
// with two LDS atomics (and, or): two lanes that update different nibbles of one word in the same instruction both
// land, two lanes on the same bucket leave the tag of one of them.  10 KiB of LDS per wave: 16 waves per CU.  (A byte
// per tag -- plain stores -- was measured in round 3: 12 KiB, 13 waves per CU, and the rate follows the wave count.)
__device__ __forceinline__ uint32_t hash5x(uint64_t v) { return (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 16)); }
__device__ __forceinline__ uint32_t tag_get(const uint32_t *tags, uint32_t h) { return (tags[h >> 3] >> ((h & 7u) * 4u)) & 15u; }
__device__ __forceinline__ void tag_set(uint32_t *tags, uint32_t h, uint32_t t)
{
    const uint32_t sh = (h & 7u) * 4u;
    atomicAnd(&tags[h >> 3], ~(15u << sh));
    atomicOr(&tags[h >> 3], t << sh);
}
#define ENC_TABLE_ENTRIES (4096 + 1024)        /* 16-bit units: 4096 positions + 4096 nibbles */
#define ENC_TAG_DECL uint32_t *tags = (uint32_t *)(table + 4096);
#define ENC_HT(v, h, t) const uint32_t hx_ = hash5x(v); h = hx_ >> 4; t = hx_ & 15u
#define ENC_TAG_OK(h, t) (tag_get(tags, h) == (t))
#define ENC_TAG_SET(h, t) tag_set(tags, h, t)

// Emit a length >= 15 continuation (rest = len - 15): rest/255 bytes of 255 then rest%255.
__device__ __forceinline__ uint8_t *emit_ext_len(uint8_t *op, uint32_t rest)
{
    const uint32_t nff = rest / 255u;
    for (uint32_t i = (uint32_t)lane_id(); i < nff; i += LZ4_WAVE) op[i] = 255;
    if (lane_id() == 0) op[nff] = (uint8_t)(rest - nff * 255u);
    return op + nff + 1;
}

// number of leading equal bytes (0..16) of two 16-byte groups
__device__ __forceinline__ uint32_t common16(const uint4 &a, const uint4 &b)
{
    const uint32_t x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
    if (x0) return (uint32_t)__builtin_ctz(x0) >> 3;
    if (x1) return 4u + ((uint32_t)__builtin_ctz(x1) >> 3);
    if (x2) return 8u + ((uint32_t)__builtin_ctz(x2) >> 3);
    if (x3) return 12u + ((uint32_t)__builtin_ctz(x3) >> 3);
    return 16u;
}

__device__ __forceinline__ int par_free_bperm(int v, int srcLane)
{
    return __builtin_amdgcn_ds_bpermute(srcLane << 2, v);
}

__device__ __forceinline__ uint32_t ext_len_bytes(uint32_t len)   // bytes of the >= 15 continuation
{
    return (len >= 15u) ? 1u + (len - 15u) / 255u : 0u;
}

// inclusive wave scan (sum) with DPP
__device__ __forceinline__ int enc_scan_incl(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

#ifndef ENC_END2_MINLEN
#define ENC_END2_MINLEN 16        // forward length from which a match also registers (its end - 2), as the reference does behind every match
#endif

// leading zero BYTES (0..16) of a 16-byte xor, without branches: v_ffbl_b32 gives 0..31, or ~0 for a zero word, so
// "or"-ing each word's bit offset in keeps ~0 for the zero words and the minimum is the first set bit of the 128
__device__ __forceinline__ uint32_t ffbl32(uint32_t x)        // v_ffbl_b32: ~0 when x == 0
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t ffbh32(uint32_t x)        // v_ffbh_u32: ~0 when x == 0
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t common16x(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
    const uint32_t c = min(min(ffbl32(x0), ffbl32(x1) | 32u), min(ffbl32(x2) | 64u, ffbl32(x3) | 96u));
    return min(c >> 3, 16u);
}

__device__ __forceinline__ uint32_t enc_mbcnt(uint64_t m)      // set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// inclusive wave max-scan (unsigned) with DPP
__device__ __forceinline__ uint32_t enc_scan_max(uint32_t x)
{
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return x;
}

// LDS word = (word & ~mask) | bits, one atomic instruction (ds_mskor_b32): a tag nibble is replaced without touching
// its neighbours, whatever other lanes do to them in the same instruction
__device__ __forceinline__ void lds_mskor(uint32_t *w, uint32_t mask, uint32_t bits)
{
    asm volatile("ds_mskor_b32 %0, %1, %2" : : "v"((uint32_t)(uintptr_t)w), "v"(mask), "v"(bits) : "memory");
}

// diagnostics (ENC_STATS builds only): cycles per phase of the dense-window path
#ifdef ENC_STATS
#define ENC_LAP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); est[i] += now_ - etm; etm = now_; } while (0)
#else
#define ENC_LAP(i) do { } while (0)
#endif

// One sequence record of a segment's list (segment mode, small batches): match start (24 bits, block-relative),
// match length (24 bits), offset (16 bits).  A sequence's literals start where the sequence before it ends.
struct SegOut {
    uint64_t *list;      // this segment's records
    uint32_t count;      // records written so far (wave-uniform)
    int base;            // added to the finder's positions to make them block-relative
    bool last;           // the block's last segment: the block's end rules apply (:214-221); elsewhere a match may run up to the seam
    int tail;            // bytes of the block behind this segment: the block's last 5 bytes are literals whichever segment they border
};
__device__ __forceinline__ uint64_t seg_pack(int start, int len, int off)
{
    return (uint64_t)(uint32_t)start | ((uint64_t)(uint32_t)len << 24) | ((uint64_t)(uint32_t)off << 48);
}
__device__ __forceinline__ void seg_unpack(uint64_t w, int &start, int &len, int &off)
{
    start = (int)(w & 0xffffffu); len = (int)((w >> 24) & 0xffffffu); off = (int)(w >> 48);
}

// Emit up to 64 queued sequences, one per lane (token, length bytes, literals, offset: cbits/lz4.c:1022-1046,
// :1065-1135), every lane busy; returns the advanced output pointer.  qCnt is wave-uniform.
__device__ __forceinline__ uint8_t *emit_sequences(const uint8_t *src, uint8_t *op, int qPrev, int qStart, int qLen, int qOff, int qCnt)
{
    const int lane = lane_id();
    const bool act = lane < qCnt;
    const uint32_t lit = act ? (uint32_t)(qStart - qPrev) : 0u;
    const uint32_t mc = act ? (uint32_t)(qLen - LZ4_MINMATCH) : 0u;
    const uint32_t ls = (uint32_t)qPrev;
    // short literal runs: up to four 8-byte chunks per lane, requested first (they only need the queue's positions)
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const bool shortRun = act && lit <= 32u;
    if (shortRun && lit >= 8u) {
        const uint32_t last = lit - 8u;
        c0 = *(const u64_unaligned *)(src + ls);
        if (lit > 8u) c1 = *(const u64_unaligned *)(src + (ls + min(8u, last)));
        if (lit > 16u) c2 = *(const u64_unaligned *)(src + (ls + min(16u, last)));
        if (lit > 24u) c3 = *(const u64_unaligned *)(src + (ls + last));
    } else if (shortRun && lit > 0u) {
        c0 = *(const u64_unaligned *)(src + ls);     // the run ends at a match start, >= 12 bytes before the end of the input
    }
    const uint32_t esz = act ? 1u + lit + ext_len_bytes(lit) + 2u + ext_len_bytes(mc) : 0u;
    const int incl = enc_scan_incl((int)esz);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    // (everything below addresses op[] and src[] through 32-bit offsets from the two wave-uniform pointers: the stores
    // and loads then take their base from scalar registers and no 64-bit address is computed per lane)
    uint32_t o = (uint32_t)(incl - (int)esz);
    // Every store of the sequence is issued behind the one wait for the literal loads above: the vector memory counter
    // counts loads and stores together, so a store in front of that wait would have to be acknowledged before the
    // literals could move (one trip to memory and back per emission for each such place).
    const uint32_t tok = o;                                    // token, then the literal length's extra bytes
    o += 1u + ext_len_bytes(lit);
    const uint32_t ld = o;
    __builtin_amdgcn_s_waitcnt(0x0f70);                        // vmcnt(0): the literals are here; nothing below waits again
    if (shortRun) {
        if (lit >= 8u) {
            const uint32_t last = lit - 8u;
            *(u64_unaligned *)(op + ld) = c0;
            if (lit > 8u) *(u64_unaligned *)(op + (ld + min(8u, last))) = c1;
            if (lit > 16u) *(u64_unaligned *)(op + (ld + min(16u, last))) = c2;
            if (lit > 24u) *(u64_unaligned *)(op + (ld + last)) = c3;
        } else {
            uint64_t w = c0;
            uint32_t done = 0;
            if (lit >= 4u) { *(u32_unaligned *)(op + ld) = (uint32_t)w; w >>= 32; done = 4; }
            for (; done < lit; done++) { op[ld + done] = (uint8_t)w; w >>= 8; }
        }
    }
    if (act) {
        uint32_t t = tok;
        op[t++] = (uint8_t)((min(lit, 15u) << 4) | min(mc, 15u));
        if (lit >= 15u) {
            uint32_t rest = lit - 15u;
            while (rest >= 255u) { op[t++] = 255; rest -= 255u; }
            op[t++] = (uint8_t)rest;
        }
        o += lit;
        op[o] = (uint8_t)qOff; op[o + 1u] = (uint8_t)((uint32_t)qOff >> 8);
        o += 2;
        if (mc >= 15u) {
            uint32_t rest = mc - 15u;
            while (rest >= 255u) { op[o++] = 255; rest -= 255u; }
            op[o++] = (uint8_t)rest;
        }
    }
    // long literal runs are copied by the whole wave
    for (uint64_t lm = __ballot(act && lit > 32u); lm; lm &= lm - 1) {
        const int k = (int)__builtin_ctzll(lm);
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)ld, k);
        const int s0 = __builtin_amdgcn_readlane(qPrev, k);
        const uint32_t ln = (uint32_t)__builtin_amdgcn_readlane((int)lit, k);
        wave_copy_bytes(op + d, src + s0, ln);
    }
    return op + total;
}

// A table entry as a candidate position for `myPos`.  The table holds the low 16 bits of a position.  When
// positions stay below 64 Ki that IS the position; otherwise (DICT: a block above 64 KiB, or a dictionary in
// front of the block) the candidate is the nearest earlier position with those bits -- at most 65535 back by
// construction, like the reference's window; an entry older than the window names some other position inside
// it, which the 4-byte compare then rejects like any stale candidate.
template <typename TabT, bool DICT>
__device__ __forceinline__ bool tab_candidate(TabT e, int myPos, uint32_t &cand)
{
    if (DICT && sizeof(TabT) == 2) {
        const uint32_t d = ((uint32_t)myPos - (uint32_t)e) & 0xffffu;
        cand = (uint32_t)myPos - d;
        return d != 0u && d <= (uint32_t)myPos;
    }
    cand = (uint32_t)e;
    return cand < (uint32_t)myPos && (uint32_t)myPos - cand <= LZ4_MAXDIST;   // :1003-1006
}

// DICT: the block is compressed with the dictLen bytes in front of it as its dictionary -- the previous block of
// the stream, which LZ4_compress_fast_continue keeps as the window (cbits/lz4.c:1608-1636, kept alive by
// Internal/LZ4.hs:376,389).  All positions are then relative to src - dictLen; the table is seeded with the
// dictionary's positions instead of inheriting the previous call's table, which needs no order between blocks.
// SEG (segment mode, small batches): the block is cut into segments that several waves compress at once, each
// with the bytes in front of its segment as dictionary (the table is seeded from them, as for linked compression).
// The wave then writes sequence RECORDS to seg->list instead of bytes, starts no match within the last 12 bytes of its
// segment (a match may END at the seam; only the block's last segment keeps the last 5 bytes as literals) and leaves
// its trailing literals to the segment behind it: the return value is the position (block-relative) where they start.
// PAIR: the dense windows go two to a step (see "Two windows per step" below); false: one window per step.
template <typename TabT, bool DICT = false, bool SEG = false, bool PAIR = false>
__device__ int encode_block_wave(const uint8_t *src, int n, uint8_t *dst, int accel, TabT *table,
                                 unsigned long long *stats = nullptr, int dictLen = 0, SegOut *seg = nullptr)
{
    const int lane = lane_id();
    if (!DICT) dictLen = 0;
    const int blockLen = n;
    src -= dictLen;                      // position 0 is the first byte of the dictionary
    n += dictLen;
#ifdef ENC_STATS
    unsigned long long est[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long etm = __builtin_amdgcn_s_memtime();
#endif
    uint8_t *op = dst;
    int anchor = dictLen;

    // Sequence queue: selected sequences are parked one per lane (registers only) and written out
    // 64 at a time, so the emission code (:1022-1046, :1065-1135) runs with every lane busy instead of
    // once per window for the handful of lanes that own a match.
    int qPrev = 0, qStart = 0, qLen = 0, qOff = 0;     // literal start, match start, match length, offset
    int qCnt = 0;                                      // uniform
    auto flush_queue = [&]() {
        if (qCnt == 0) return;
        if (SEG) {
            // segment mode: the sequences go to the segment's list; a second kernel stitches the lists into the block
            if (lane < qCnt) seg->list[seg->count + (uint32_t)lane] = seg_pack(qStart + seg->base, qLen, qOff);
            seg->count += (uint32_t)qCnt;
            qCnt = 0;
            return;
        }
        op = emit_sequences(src, op, qPrev, qStart, qLen, qOff, qCnt);
        qCnt = 0;
    };

    if (blockLen == 0) {                // cbits/lz4.c:1263-1273: empty input -> single 0 token
        if (SEG) return seg->base + dictLen;
        if (lane == 0) dst[0] = 0;
        return 1;
    }

    // zero the table (positions are block-relative; 0 is a real position, as in the reference)
    {
        uint32_t *t32 = (uint32_t *)table;
        const int nd = (int)(4096 * sizeof(TabT) / 4) + 512;    // positions, then the tags
        for (int i = lane; i < nd; i += LZ4_WAVE) t32[i] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    ENC_TAG_DECL
    if (DICT && blockLen >= 13) {
        // seed: positions of the dictionary, in order (a later position replaces an earlier one); the
        // 8 bytes behind a position near its end run into the block itself
        // (the LDS executes one wave's stores in order: no fence between them)
#ifndef ENC_SEED_STEP
#define ENC_SEED_STEP 2           // every second position: same size as every position (the reference's own table only
#endif                           // holds the positions its parse visited), every fourth costs 0.3 % of size

        for (int q0 = 0; q0 < dictLen; q0 += 8 * LZ4_WAVE * ENC_SEED_STEP) {
            uint64_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int q = q0 + (k * LZ4_WAVE + lane) * ENC_SEED_STEP;
                v[k] = (q < dictLen) ? *(const u64_unaligned *)(src + q) : 0ull;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int q = q0 + (k * LZ4_WAVE + lane) * ENC_SEED_STEP;
                if (q < dictLen) { uint32_t h_, t_; ENC_HT(v[k], h_, t_); table[h_] = (TabT)q; ENC_TAG_SET(h_, t_); }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }

    if (blockLen >= 13) {                               // LZ4_minLength, :221,:921
        const int mfl = n - LZ4_MFLIMIT + 1;            // match start must be < mfl (:883)
        // match end must be <= matchlimit (:884): the BLOCK's last 5 bytes are literals; a segment with a few bytes
        // of the block behind it stops short of its seam by what is missing
        const int matchlimit = SEG ? n - max(0, LZ4_LASTLITERALS - seg->tail) : n - LZ4_LASTLITERALS;
        const uint32_t miss0 = (uint32_t)accel << 6;
        uint32_t missAcc = miss0;
        int64_t p = dictLen;
        uint64_t pfV8 = 0;          // dense path: this lane's 8 bytes of the window that starts at pfPos
        int pfPos = -1;


#ifndef ENC_NO_PIPE
        // ===================================================================================================
        // Dense window, one round trip (round 3).  Round 2's window (the code further down, still used for the
        // last windows of a block) is a chain of dependent memory round trips -- bytes -> table -> candidate ->
        // one to three extension steps per run head, in per-lane loops -- and a wave spent two thirds of its
        // cycles waiting.  Here every run head is handed to a GROUP OF FOUR LANES that requests, in one go,
        // 16 bytes per lane around the head's position and around its candidate: the 8 bytes before them
        // (catch up, :1019) and the 56 after.  The tags make a candidate a real match nine times in ten, so
        // there is no separate verification step: one round trip per window, every lane busy, and the lengths
        // of up to 32 heads come out of two quad-DPP reductions.  A match that reaches the 56-byte horizon is
        // extended by the whole wave, 1 KiB a step, and only if the greedy selection takes it.
        //   A (probe)   hash the 64 positions, read buckets and tags, run heads, groups, requests
        //   C (finish)  lengths per group, every lane learns its run's head (DPP max-scan), greedy selection
        //               (scalar, as below), the probed positions outside the selected matches go into the table
        //               (:998, the reference's policy), sequences are queued
        // The next window starts at the end of the last selected match; its bytes are requested as soon as
        // the selection knows that position.
        // Measured on the way (MI355X, lzsynth / text, 16 384 blocks): two windows in flight per wave (window
        // w+1 probed before window w is finished, on a fixed 64-position grid, every probed position written
        // into the table at once and positions that turn out to lie inside a selected match taking their
        // insertion back a window later; oracle/sim_encode2.c has the policy: ratio 2.915 / 1.841 against
        // 2.895 / 1.838) 132 / 112 GB/s; one window at a time, this code, 150 / 109 GB/s: the fixed grid
        // costs 17 % more windows (1022 against 875 per block) and a window costs its ~200 vector
        // instructions whatever is in flight -- at 16 waves per CU the vector ALU is 80 % busy either way.
        // ===================================================================================================
        const LZ4_GLOBAL uint8_t *gsrc = as_global(src);
        auto load16 = [&](int off) -> dev_v4 { return *(const LZ4_GLOBAL dev_v4u *)(gsrc + (uint32_t)off); };
        auto quad = [](uint32_t v, const int ctrl) -> uint32_t {      // quad_perm broadcast
            return ctrl == 0 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true)
                 : ctrl == 1 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, true)
                 : ctrl == 2 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xaa, 0xf, 0xf, true)
                             : (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xff, 0xf, 0xf, true);
        };
        const bool pipeFits = n <= (1 << 25);          // a group's head travels as a 25-bit position
        // the last request of a window ends 136 bytes behind its first position
        auto pipe_can_issue = [&](int p0) -> bool { return (missAcc >> 6) == 1u && p0 + 136 <= n && pipeFits; };
        const uint32_t lanePay = 0x80000000u | ((uint32_t)lane << 25);   // a head's message to its group: valid | lane | candidate
        const int j16m8 = (lane & 3) * 16 - 8;

        // match length of a group's head from the 4 x 16 bytes of the group (meaningful in the group's lane 0):
        // forward length (0..56) | equal bytes before the head (0..8) << 8 | four equal bytes << 12 | horizon reached << 13
        auto groupLen = [&](const dev_v4 &a, const dev_v4 &b) -> uint32_t {
            const uint32_t x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
            const uint32_t l0 = ffbl32(x0), l1 = ffbl32(x1) | 32u, l2 = ffbl32(x2), l3 = ffbl32(x3) | 32u;
            const uint32_t lo8 = min(min(l0, l1) >> 3, 8u);                        // equal bytes from byte 0 on (0..8)
            const uint32_t hi8 = min(min(l2, l3) >> 3, 8u);                        // equal bytes from byte 8 on (0..8)
            const uint32_t f16 = (lo8 == 8u) ? 8u + hi8 : lo8;
            // a group's lane 0: bytes 0..7 lie before the head (counted backwards from byte 7), bytes 8..15 are its first eight
            const uint32_t back = min(min(ffbh32(x1), ffbh32(x0) | 32u) >> 3, 8u);
            const bool first = (lane & 3) == 0;
            const uint32_t f = first ? hi8 : f16, full = first ? 8u : 16u;
            const uint32_t f1 = quad(f, 1), f2 = quad(f, 2), f3 = quad(f, 3);
            uint32_t total = f;
            total += (f == full) ? f1 : 0u;
            total += (f == full && f1 == 16u) ? f2 : 0u;
            total += (f == full && f1 == 16u && f2 == 16u) ? f3 : 0u;
            return total | (back << 8) | ((uint32_t)(f >= (uint32_t)LZ4_MINMATCH) << 12) | ((uint32_t)(total == 56u) << 13);
        };

        // heads beyond the 32nd of a window (two rounds of 16 groups are requested up front): one more round of groups at
        // a time, requested and waited for here.  Generated text has 14-17 heads per window; source code 25-27, and
        // leaving the heads past the 32nd without a match cost 2 % of its ratio (oracle/sim_encode2.c, headcap32).
        auto more_rounds = [&](const int p0w, const uint64_t headm, const bool head, const uint32_t rank4, const uint32_t cand, uint32_t r) -> uint32_t {
            const int nH = (int)__builtin_popcountll(headm);
            for (int rr = 2; rr * 16 < nH; rr++) {
                const int dest = (head && (int)(rank4 >> 8) == rr) ? (int)(rank4 & 255u) : 4;
                const uint32_t gi = quad((uint32_t)__builtin_amdgcn_ds_permute(dest, (int)(lanePay | cand)), 0);
                const bool gv = (int)gi < 0;
                const int c = (int)(gi & 0x1ffffffu), hl = (int)((gi >> 25) & 63u);
                const dev_v4 a = load16(gv ? p0w + hl + j16m8 : p0w);
                const dev_v4 b = load16(gv ? c + j16m8 : p0w);
                const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rank4 & 255u), (int)groupLen(a, b));
                if ((int)(rank4 >> 8) == rr) r = r2;
            }
            return r;
        };

        // a match that reached the horizon: the whole wave counts on, 16 bytes a lane
        auto extend_long = [&](const int pe, const int ce) -> int {
            int total = 0;
            const int maxExtra = matchlimit - pe;
            for (;;) {
                const int off = total + 16 * lane;
                uint32_t d = 0;
                bool full = false;
                if (off + 16 <= maxExtra) {
                    const dev_v4 a = load16(pe + off), b = load16(ce + off);
                    d = common16x(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
                    full = d == 16u;
                } else if (off < maxExtra) {
                    const uint32_t r = (uint32_t)(maxExtra - off);
                    while (d < r && gsrc[pe + off + (int)d] == gsrc[ce + off + (int)d]) d++;
                }
                const uint64_t ne = ~__ballot(full);
                if (ne == 0ull) { total += 16 * LZ4_WAVE; continue; }
                const int first = (int)__builtin_ctzll(ne);
                total += 16 * first + __builtin_amdgcn_readlane((int)d, first);
                break;
            }
            return total;
        };

        // The tail of a window -- catch up and parking the selected sequences in the queue, which touch neither the
        // table nor the anchor -- is put off until the NEXT window has issued its requests: it then runs while those
        // are in flight instead of in front of them (the wave's chain of dependent waits per window is what its rate
        // follows: 16 waves per CU, each waiting two thirds of the time).
        struct Tail {                                        // a window's pending tail
            uint64_t selm = 0;                               // selected lanes (0 = nothing pending)
            int p0 = 0, prevEnd = 0, ml = 0, back = 0;       // its window, and per lane: literal start, match length, equal bytes before
            uint32_t cand = 0;
            int r0 = 0, r1 = 0, r2 = 0, r3 = 0, k = 0;       // the cross-lane moves under way
        } tl[2];                                             // (two: the pair form below finishes two windows at a time)
        // ... in two halves, so that its four cross-lane moves travel with the next window's table reads (one LDS round
        // trip for both) and are only looked at once that window's requests are out.  Without a pending tail (selm = 0)
        // both halves do nothing: no branch, so that the compiler can interleave them with the window's own code.
        auto tail_issue1 = [&](Tail &t, const int qBase) {
            const uint64_t selm = t.selm;
            const bool sel = (selm >> lane) & 1ull;
            const int myPos = t.p0 + lane;
            // ---- catch up (:1019), bounded by the previous match: the bytes between my run's head and me
            // are known equal, the head's group measured up to 8 more before the head ----
            int mstart = myPos, mcand = (int)t.cand;
            if (sel) {
                const int room = min(mstart - t.prevEnd, mcand);
                const int back = min(room, t.back);
                mstart -= back; mcand -= back;
            }
            // ---- park the selected sequences in the queue (stable compaction, one ds_permute per field) ----
            t.k = (int)__builtin_popcountll(selm);
            const int rk = (int)enc_mbcnt(selm);
            const int dest = (sel ? qBase + rk : qBase + t.k + (lane - rk)) & 63;
            t.r0 = __builtin_amdgcn_ds_permute(dest << 2, t.prevEnd);
            t.r1 = __builtin_amdgcn_ds_permute(dest << 2, mstart);
            t.r2 = __builtin_amdgcn_ds_permute(dest << 2, myPos + t.ml - mstart);
            t.r3 = __builtin_amdgcn_ds_permute(dest << 2, mstart - mcand);
            t.selm = 0;
        };
        auto tail_commit1 = [&](Tail &t) {
            if (lane >= qCnt && lane < qCnt + t.k) { qPrev = t.r0; qStart = t.r1; qLen = t.r2; qOff = t.r3; }
            qCnt += t.k;
            t.k = 0;
        };
        bool tailSplit = false;                              // the two pending tails did not fit the queue together
        auto tail_issue = [&]() {
            if constexpr (!PAIR) {
                if (qCnt + (int)__builtin_popcountll(tl[0].selm) > LZ4_WAVE) flush_queue();
                tail_issue1(tl[0], qCnt);
                return;
            }
            const int k0 = (int)__builtin_popcountll(tl[0].selm), k1 = PAIR ? (int)__builtin_popcountll(tl[1].selm) : 0;
            if (qCnt + k0 + k1 > LZ4_WAVE) {
                // (rare: once per 64 sequences) one after the other, with the queue written out in between
                if (qCnt + k0 > LZ4_WAVE) flush_queue();
                tail_issue1(tl[0], qCnt); tail_commit1(tl[0]);
                if constexpr (PAIR) {
                    if (qCnt + k1 > LZ4_WAVE) flush_queue();
                    tail_issue1(tl[1], qCnt); tail_commit1(tl[1]);
                }
                tailSplit = true;
                return;
            }
            tailSplit = false;
            tail_issue1(tl[0], qCnt);
            if constexpr (PAIR) tail_issue1(tl[1], qCnt + k0);
        };
        auto tail_commit = [&]() {
            if constexpr (!PAIR) { tail_commit1(tl[0]); return; }
            if (tailSplit) { tailSplit = false; return; }
            tail_commit1(tl[0]);
            if constexpr (PAIR) tail_commit1(tl[1]);
        };
        auto window_tail = [&]() { tail_issue(); tail_commit(); };

        // one window: positions [p0, p0 + 64), p0 >= anchor; returns where the next one starts
        auto group_window = [&](const int p0) -> int {
            // ---- A: probe ----
            const int myPos = p0 + lane;
            if (pfPos != p0) pfV8 = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)myPos);
            const uint32_t hx = hash5x(pfV8);
            const uint32_t h = hx >> 4, tg = hx & 15u;
            const uint32_t tsh = (hx >> 2) & 28u;                 // (h & 7) * 4
            uint32_t *tagw = &tags[hx >> 7];                      // h >> 3
            const uint32_t oldp = table[h], tagWord = *tagw;
            tail_issue();                                      // the window before this one: its cross-lane moves ride along
            const uint32_t oldt = (tagWord >> tsh) & 15u;
            uint32_t cand = 0;
            const bool candOk = tab_candidate<TabT, DICT>((TabT)oldp, myPos, cand) && oldt == tg && cand >= 8u;
            // run heads: a candidate that continues its left neighbour's belongs to the same copied region
            // (a lane without a candidate sends ~0, and ~0 + 1 = 0 is no candidate: they start at 8)
            const uint32_t cv = candOk ? cand : 0xffffffffu;
            const uint32_t prevCand = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)cv, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const bool head = candOk && cand != prevCand + 1u;
            const uint64_t headm = __ballot(head);
            const bool twoRounds = (headm >> 16) != 0ull && __builtin_popcountll(headm) > 16;
            const uint32_t rank4 = enc_mbcnt(headm) << 4;         // byte address of lane 4 * rank
            const uint32_t payload = lanePay | cand;              // positions stay below 2^25 (blocks <= 4 MiB + dictionary)
            dev_v4 a0, b0, a1, b1;
            // round 0: heads 0..15, one per group of four lanes (lane 1 takes what the others send)
            {
                const int dest = (head && rank4 < 256u) ? (int)rank4 : 4;
                const uint32_t gi = quad((uint32_t)__builtin_amdgcn_ds_permute(dest, (int)payload), 0);
                const bool gv = (int)gi < 0;
                const int c = (int)(gi & 0x1ffffffu), hl = (int)((gi >> 25) & 63u);
                a0 = load16(gv ? p0 + hl + j16m8 : p0);
                b0 = load16(gv ? c + j16m8 : p0);
            }
            // round 1: heads 16..31 (one window in four has them)
            if (twoRounds) {
                const int dest = (head && (rank4 >> 8) == 1u) ? (int)(rank4 & 255u) : 4;
                const uint32_t gi = quad((uint32_t)__builtin_amdgcn_ds_permute(dest, (int)payload), 0);
                const bool gv = (int)gi < 0;
                const int c = (int)(gi & 0x1ffffffu), hl = (int)((gi >> 25) & 63u);
                a1 = load16(gv ? p0 + hl + j16m8 : p0);
                b1 = load16(gv ? c + j16m8 : p0);
            }
            ENC_LAP(0);
            tail_commit();                                     // (the requests are out: the moves' results are looked at now)
            // ---- C: finish ----
            uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rank4 & 255u), (int)groupLen(a0, b0));
            if (twoRounds) {
                const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rank4 & 255u), (int)groupLen(a1, b1));
                if (rank4 >= 256u) r = r1;
            }
#ifdef ENC_STATS
            est[7] += (unsigned)__builtin_popcountll(headm);
#endif
            // every lane learns its run's head (lane and result) with a max-scan: head lanes put (lane + 1) << 16 | result
            // in, the others 0, and the largest value at or below a lane belongs to the nearest head below it
#ifndef ENC_EXP_NOMORE
            if (headm >> 32 && __builtin_popcountll(headm) > 32) r = more_rounds(p0, headm, head, rank4, cand, r);
#endif
            const uint32_t hv = enc_scan_max(head ? (((uint32_t)lane + 1u) << 16) | r : 0u);
            const int delta = lane + 1 - (int)(hv >> 16);                 // lanes between my run's head and me
            const int m0 = (int)(hv & 0xffu) - delta;
            const bool hit = candOk && ((hv >> 12) & 1u) && m0 >= LZ4_MINMATCH;
            uint32_t myMl = hit ? (uint32_t)m0 : 0u;
            ENC_LAP(1);
            // ---- greedy left-to-right selection ----
            const uint64_t hitm = __ballot(hit);
            if (!hitm) {
                // nothing here: every position is registered, the miss counter widens the stride (:957-967)
                table[h] = (TabT)myPos;
                lds_mskor(tagw, 15u << tsh, tg << tsh);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                missAcc += LZ4_WAVE;
                pfPos = -1;
                return p0 + LZ4_WAVE;
            }
            const uint64_t capm = __ballot(hit && ((hv >> 13) & 1u));
            uint64_t selm = 0;
            int pEnd = anchor;
            int prevEnd = anchor;              // end of the selected match before me (selected lanes: my literal start)
            bool longBefore = false;           // ... and that match is a long one
            for (uint64_t hm = hitm; hm;) {
                const int k = (int)__builtin_ctzll(hm);
                int len = __builtin_amdgcn_readlane((int)myMl, k);
                if ((capm >> k) & 1ull) {
                    len += extend_long(p0 + k + len, __builtin_amdgcn_readlane((int)cand, k) + len);
                    if (lane == k) myMl = (uint32_t)len;
                }
                const int endk = p0 + k + len;
                selm |= 1ull << k;
                pEnd = endk;
                if (lane > k) { prevEnd = endk; longBefore = len >= ENC_END2_MINLEN; }
                const int sh = endk - p0;
                hm = (sh >= LZ4_WAVE) ? 0ull : (hm & (~0ull << sh));
            }
            const bool sel = (selm >> lane) & 1ull;
            // the next window starts at the end of the last match, or where this one ends: its bytes are requested now
            const int nextP = max(p0 + LZ4_WAVE, pEnd);
            pfPos = nextP;
            if (nextP + 72 <= n) pfV8 = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(nextP + lane));
            else pfPos = -1;
            ENC_LAP(2);
            // ---- table: the probed positions outside the selected matches (:998) ----
            // (... and, like the reference behind every match (:1146), the position two bytes in front of the end of a
            // LONG match: behind every match it costs text 1.7 % of its ratio, behind matches of 16 bytes and more it
            // costs nothing there and buys lzsynth 0.13 %)
            if (sel || myPos >= prevEnd || (longBefore && myPos == prevEnd - 2)) {
                table[h] = (TabT)myPos;
                lds_mskor(tagw, 15u << tsh, tg << tsh);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            ENC_LAP(3);
            // ---- catch up and the queue: put off (window_tail) ----
            tl[0].selm = selm; tl[0].p0 = p0; tl[0].prevEnd = prevEnd; tl[0].ml = (int)myMl; tl[0].cand = cand;
            tl[0].back = delta + (int)((hv >> 8) & 15u);
            anchor = pEnd;
            missAcc = miss0;
#ifdef ENC_STATS
            est[6] += 1;
#endif
            return nextP;
        };

        // ===================================================================================================
        // Two windows per step (positions [p0, p0 + 128)).  A window is a chain of dependent waits -- table, groups,
        // requests, lengths, selection -- and the chip has room for 16 such chains per CU (the table's LDS); the rate
        // follows the number of chains, not the instructions.  Here one chain carries two windows, step by step side by
        // side, so that each wait is paid once for both.  What the second window needs from the first is the table:
        // every position of a window is written into its bucket when it is probed (the bucket's previous entry kept in
        // a register), and positions that end up strictly inside a selected match take that back, last window first,
        // before the next pair starts -- between pairs the table is the reference's (:998), within a pair the second
        // window also sees the first one's covered positions (oracle/sim_encode2.c: the ratio goes UP, 2.895 -> 2.91).
        // ===================================================================================================
        struct PW {
            uint32_t hx, oldp, tagWord, cand, cv;
            bool candOk, head, twoRounds;
            uint64_t headm, selm;
            uint32_t rank4, gi0, hv, myMl;
            int prevEnd, delta;
            bool keep2;                      // the selected match before me is a long one: its end - 2 stays registered (:1146)
            dev_v4 a0, b0, a1, b1;
        };
        uint64_t pfV8b = 0;                                  // the second window's bytes (window at pfPos + 64)
        int pfPosB = -2;                                     // ... valid when pfPosB == pfPos (only this form requests them)
        // (What it costs: the second window meets the first one's positions in the table before the covered ones are
        // taken back -- lzsynth 2.891 -> 2.888, source text -0.5 % in the simulation -- which the registration behind
        // long matches below more than gives back on lzsynth.  The template parameter PAIR picks the form: blocks above 64 KiB, whose positions are modular, and segments run one
        // window per step.)
        auto pair_can_issue = [&](int p0) -> bool { return (missAcc >> 6) == 1u && p0 + 200 <= n && pipeFits; };
        auto pw_probe = [&](PW &W, const int p0w, const uint64_t v8) {
            const uint32_t hx = hash5x(v8);
            const uint32_t h = hx >> 4, tg = hx & 15u, tsh = (hx >> 2) & 28u;
            uint32_t *tagw = &tags[hx >> 7];
            W.hx = hx;
            W.oldp = table[h];
            W.tagWord = *tagw;
            table[h] = (TabT)(p0w + lane);                     // every position, at once: the window behind this one probes next
            lds_mskor(tagw, 15u << tsh, tg << tsh);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        };
        auto pw_heads = [&](PW &W, const int p0w, const uint32_t cvBefore) {
            const uint32_t tg = W.hx & 15u, tsh = (W.hx >> 2) & 28u;
            const uint32_t oldt = (W.tagWord >> tsh) & 15u;
            uint32_t cand = 0;
            W.candOk = tab_candidate<TabT, DICT>((TabT)W.oldp, p0w + lane, cand) && oldt == tg && cand >= 8u;
            W.cand = cand;
            W.cv = W.candOk ? cand : 0xffffffffu;
            // (lane 0 continues the run of the last lane of the window before it)
            const uint32_t prevCand = (uint32_t)__builtin_amdgcn_update_dpp((int)cvBefore, (int)W.cv, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            W.head = W.candOk && cand != prevCand + 1u;
            W.headm = __ballot(W.head);
            W.twoRounds = (W.headm >> 16) != 0ull && __builtin_popcountll(W.headm) > 16;
            W.rank4 = enc_mbcnt(W.headm) << 4;
            const int dest = (W.head && W.rank4 < 256u) ? (int)W.rank4 : 4;
            W.gi0 = (uint32_t)__builtin_amdgcn_ds_permute(dest, (int)(lanePay | cand));
        };
        auto pw_loads = [&](PW &W, const int p0w) {
            {
                const uint32_t gi = quad(W.gi0, 0);
                const bool gv = (int)gi < 0;
                const int c = (int)(gi & 0x1ffffffu), hl = (int)((gi >> 25) & 63u);
                W.a0 = load16(gv ? p0w + hl + j16m8 : p0w);
                W.b0 = load16(gv ? c + j16m8 : p0w);
            }
            if (W.twoRounds) {
                const int dest = (W.head && (W.rank4 >> 8) == 1u) ? (int)(W.rank4 & 255u) : 4;
                const uint32_t gi = quad((uint32_t)__builtin_amdgcn_ds_permute(dest, (int)(lanePay | W.cand)), 0);
                const bool gv = (int)gi < 0;
                const int c = (int)(gi & 0x1ffffffu), hl = (int)((gi >> 25) & 63u);
                W.a1 = load16(gv ? p0w + hl + j16m8 : p0w);
                W.b1 = load16(gv ? c + j16m8 : p0w);
            }
        };
        // lengths: laneBase numbers the lanes of the pair 0..127; hvBefore = the scan value of the last lane of the
        // window before (its run may go on into this window)
        auto pw_lengths = [&](PW &W, const int p0w, const int laneBase, const uint32_t hvBefore) {
            uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(W.rank4 & 255u), (int)groupLen(W.a0, W.b0));
            if (W.twoRounds) {
                const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(W.rank4 & 255u), (int)groupLen(W.a1, W.b1));
                if (W.rank4 >= 256u) r = r1;
            }
#ifdef ENC_STATS
            est[7] += (unsigned)__builtin_popcountll(W.headm);
#endif
#ifndef ENC_EXP_NOMORE
            if (W.headm >> 32 && __builtin_popcountll(W.headm) > 32) r = more_rounds(p0w, W.headm, W.head, W.rank4, W.cand, r);
#endif
            const uint32_t mine = W.head ? (((uint32_t)(laneBase + lane) + 1u) << 16) | r : 0u;
            W.hv = max(enc_scan_max(mine), hvBefore);
            W.delta = laneBase + lane + 1 - (int)(W.hv >> 16);
            const int m0 = (int)(W.hv & 0xffu) - W.delta;
            const bool hit = W.candOk && ((W.hv >> 12) & 1u) && m0 >= LZ4_MINMATCH;
            W.myMl = hit ? (uint32_t)m0 : 0u;
        };
        // greedy selection in this window, continuing from pEnd (the end of the last selected match so far)
        auto pw_select = [&](PW &W, const int p0w, int &pEnd) {
            const bool hit = W.myMl != 0u;
            uint64_t hitm = __ballot(hit);
            const uint64_t capm = __ballot(hit && ((W.hv >> 13) & 1u));
            const int lowcut = pEnd - p0w;
            if (lowcut > 0) hitm = (lowcut >= LZ4_WAVE) ? 0ull : (hitm & (~0ull << lowcut));
            uint64_t selm = 0;
            int prevEnd = pEnd;
            bool longBefore = false;
            for (uint64_t hm = hitm; hm;) {
                const int k = (int)__builtin_ctzll(hm);
                int len = __builtin_amdgcn_readlane((int)W.myMl, k);
                if ((capm >> k) & 1ull) {
                    len += extend_long(p0w + k + len, __builtin_amdgcn_readlane((int)W.cand, k) + len);
                    if (lane == k) W.myMl = (uint32_t)len;
                }
                const int endk = p0w + k + len;
                selm |= 1ull << k;
                pEnd = endk;
                if (lane > k) { prevEnd = endk; longBefore = len >= ENC_END2_MINLEN; }
                const int sh = endk - p0w;
                hm = (sh >= LZ4_WAVE) ? 0ull : (hm & (~0ull << sh));
            }
            W.selm = selm;
            W.prevEnd = prevEnd;
            W.keep2 = longBefore;
        };
        // positions strictly inside a selected match take their insertion back if the bucket still holds it
        auto pw_takeback = [&](PW &W, const int p0w) {
            const int myPos = p0w + lane;
            if (!((W.selm >> lane) & 1ull) && myPos < W.prevEnd && !(W.keep2 && myPos == W.prevEnd - 2)) {
                const uint32_t h = W.hx >> 4;
                if (table[h] == (TabT)myPos) {
                    const uint32_t tsh = (W.hx >> 2) & 28u;
                    table[h] = (TabT)W.oldp;
                    lds_mskor(&tags[W.hx >> 7], 15u << tsh, ((W.tagWord >> tsh) & 15u) << tsh);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        };
        auto pair_window = [&](const int p0) -> int {
            const int p1 = p0 + LZ4_WAVE;
            if (pfPos != p0 || pfPosB != p0) {
                pfV8 = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(p0 + lane));
                pfV8b = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(p1 + lane));
            }
            PW W0, W1;
            pw_probe(W0, p0, pfV8);
            pw_probe(W1, p1, pfV8b);
            tail_issue();                                      // the pair before this one: its cross-lane moves ride along
            pw_heads(W0, p0, 0xffffffffu);
            pw_heads(W1, p1, (uint32_t)__builtin_amdgcn_readlane((int)W0.cv, 63));
            pw_loads(W0, p0);
            pw_loads(W1, p1);
            ENC_LAP(0);
            tail_commit();
            int pEnd = anchor;
            pw_lengths(W0, p0, 0, 0u);
            pw_select(W0, p0, pEnd);
            pw_lengths(W1, p1, LZ4_WAVE, (uint32_t)__builtin_amdgcn_readlane((int)W0.hv, 63));
            pw_select(W1, p1, pEnd);
            ENC_LAP(1);
            // the next pair starts at the end of the last match, or where this one ends: its bytes are requested now
            const int nextP = max(p0 + 2 * LZ4_WAVE, pEnd);
            pfPos = nextP;
            pfPosB = -2;
            if (nextP + 136 <= n) {
                pfV8 = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(nextP + lane));
                pfV8b = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(nextP + LZ4_WAVE + lane));
                pfPosB = nextP;
            } else if (nextP + 72 <= n) {
                pfV8 = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(nextP + lane));
            } else {
                pfPos = -1;
            }
            ENC_LAP(2);
            pw_takeback(W1, p1);
            pw_takeback(W0, p0);
            ENC_LAP(3);
            tl[0].selm = W0.selm; tl[0].p0 = p0; tl[0].prevEnd = W0.prevEnd; tl[0].ml = (int)W0.myMl; tl[0].cand = W0.cand;
            tl[0].back = W0.delta + (int)((W0.hv >> 8) & 15u);
            tl[1].selm = W1.selm; tl[1].p0 = p1; tl[1].prevEnd = W1.prevEnd; tl[1].ml = (int)W1.myMl; tl[1].cand = W1.cand;
            tl[1].back = W1.delta + (int)((W1.hv >> 8) & 15u);
            if (W0.selm | W1.selm) { anchor = pEnd; missAcc = miss0; }
            else { missAcc += 2 * LZ4_WAVE; pfPos = -1; }
#ifdef ENC_STATS
            est[6] += 2;
#endif
            return nextP;
        };
#endif  // ENC_NO_PIPE
        (void)group_window; (void)pair_window; (void)pipe_can_issue; (void)pair_can_issue;   // (one form per instantiation)
        while (p < mfl) {
            const int64_t step = (int64_t)(missAcc >> 6);
            if (step == 1) {
#ifndef ENC_NO_PIPE
                if (PAIR ? pair_can_issue((int)p) : pipe_can_issue((int)p)) {
                    int np = (int)p;
                    if constexpr (PAIR) {
                        // (the last 200 bytes of a block are left to the windows below)
                        while (pair_can_issue(np)) np = pair_window(np);
                    } else {
                        do np = group_window(np); while (pipe_can_issue(np));
                    }
                    window_tail();
                    p = np;
                    continue;
                }
#endif
                // ===== dense window (acceleration 1, no miss streak): 64 consecutive positions, and EVERY
                // match found in the window is emitted in this iteration, not only the first =====
                const int p0 = (int)p;
                const int myPos = p0 + lane;
                const bool valid = myPos < mfl;
                uint32_t h = 0, tg = 0, cand = 0, myMl = 0;
                uint64_t v8 = 0;
                bool candOk = false;
                if (valid) {
                    v8 = (pfPos == p0) ? pfV8 : *(const u64_unaligned *)(src + myPos);
                    ENC_HT(v8, h, tg);
                    candOk = tab_candidate<TabT, DICT>(table[h], myPos, cand) && ENC_TAG_OK(h, tg);   // :1003-1006
                }
                ENC_LAP(0);
                // Candidates of neighbouring positions that are themselves neighbours (cand[l] == cand[l-1]+1)
                // belong to one copied region: only the HEAD of such a run reads its candidate and counts
                // bytes (:1009, LZ4_count :603-626); the others derive hit and length from it.  This keeps
                // the scattered candidate loads to roughly one per real match plus the unmatched positions.
                bool hit = false;
                uint32_t hback = 0;       // head only: equal bytes just before the match (catch up, :1019), 0..8
                int myHead = 0;
                {
                    const uint32_t prevCand = (uint32_t)__shfl_up((int)(candOk ? cand : 0xffffffffu), 1);
                    const bool contin = candOk && lane > 0 && prevCand != 0xffffffffu && cand == prevCand + 1u;
                    const bool head = candOk && !contin;
                    if (head) {
                        // one scattered request: the 8 bytes before the candidate (catch up) and the 8 at it
                        uint64_t c8, a8 = 0, b8 = 1;
                        if (cand >= 8u) {
                            uint4 bc;
                            __builtin_memcpy(&bc, src + cand - 8, 16);
                            b8 = ((uint64_t)bc.y << 32) | bc.x;
                            c8 = ((uint64_t)bc.w << 32) | bc.z;
                            a8 = *(const u64_unaligned *)(src + myPos - 8);
                        } else {
                            c8 = *(const u64_unaligned *)(src + cand);
                        }
                        const uint64_t x = v8 ^ c8;
                        if ((uint32_t)x == 0) {                                  // 4 equal bytes
                            hit = true;
                            const uint64_t xb = a8 ^ b8;
                            hback = (cand >= 8u) ? (xb ? ((uint32_t)__builtin_clzll(xb) >> 3) : 8u) : 0u;
                            const uint32_t maxLen = (uint32_t)(matchlimit - myPos);   // >= 7
                            uint32_t m = x ? ((uint32_t)__builtin_ctzll(x) >> 3) : 8u;
                            if (m == 8u) {
                                bool more = true;
                                while (more && m + 32u <= maxLen) {               // 32 bytes a step
                                    uint4 a0, a1, b0, b1;
                                    __builtin_memcpy(&a0, src + myPos + m, 16);
                                    __builtin_memcpy(&a1, src + myPos + m + 16, 16);
                                    __builtin_memcpy(&b0, src + cand + m, 16);
                                    __builtin_memcpy(&b1, src + cand + m + 16, 16);
                                    uint32_t d = common16(a0, b0);
                                    if (d == 16u) d += common16(a1, b1);
                                    m += d;
                                    more = (d == 32u);
                                }
                                while (more && m + 16u <= maxLen) {
                                    uint4 a0, b0;
                                    __builtin_memcpy(&a0, src + myPos + m, 16);
                                    __builtin_memcpy(&b0, src + cand + m, 16);
                                    const uint32_t d = common16(a0, b0);
                                    m += d;
                                    more = (d == 16u);
                                }
                                while (more && m < maxLen && src[myPos + m] == src[cand + m]) m++;
                            }
                            myMl = min(m, maxLen);
                        }
                    }
                    const uint64_t headm = __ballot(head);
#ifdef ENC_STATS
                    est[7] += (unsigned)__builtin_popcountll(headm);
#endif
                    const uint64_t below = headm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
                    myHead = below ? 63 - (int)__builtin_clzll(below) : 0;
                    const int headInfo = par_free_bperm((int)(myMl | (hback << 16)), myHead);   // 0 when the head missed
                    if (contin) {
                        const int m = (headInfo & 0xffff) - (lane - myHead);
                        hit = m >= LZ4_MINMATCH;
                        myMl = hit ? (uint32_t)m : 0u;
                        hback = (uint32_t)headInfo >> 16;
                    }
                }
                ENC_LAP(1);
                // ---- greedy left-to-right selection of non-overlapping matches ----
                uint64_t hitm = __ballot(hit);
                if (!hitm) {
                    if (valid) { table[h] = (TabT)myPos; ENC_TAG_SET(h, tg); }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    missAcc += LZ4_WAVE;
                    p += LZ4_WAVE;
                    pfPos = -1;
                    continue;
                }
                // The loop is scalar (one v_readlane per selected match); what each lane needs from it -- the end of
                // the selected match before it -- is fetched afterwards with one cross-lane read.
                uint64_t selm = 0;
                int pEnd = anchor;
                for (uint64_t hm = hitm; hm;) {
                    const int k = (int)__builtin_ctzll(hm);
                    const int endk = p0 + k + (int)__builtin_amdgcn_readlane((int)myMl, k);
                    selm |= 1ull << k;
                    pEnd = endk;
                    const int sh = endk - p0;
                    hm = (sh >= LZ4_WAVE) ? 0ull : (hm & (~0ull << sh));
                }
                int prevEnd = anchor;            // selected lanes: end of the previous selected match (my literal start)
                bool covered = false;            // my position lies strictly inside a selected match
                {
                    const bool selMe = (selm >> lane) & 1ull;
                    const uint64_t lowerIncl = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                    // selected lane: the selected lane before me; any other lane: the last selected lane below me
                    const uint64_t m = selm & (lowerIncl >> 1);          // selected lanes strictly below me
                    const int from = m ? 63 - (int)__builtin_clzll(m) : -1;
                    const int endFrom = par_free_bperm(myPos + (int)myMl, from & 63);
                    if (from >= 0) {
                        if (selMe) prevEnd = endFrom;
                        else covered = myPos < endFrom;
                    }
                }
                const int lastEnd = pEnd;
                const bool sel = (selm >> lane) & 1ull;
                // the next window starts at the end of the last match, or at the end of this window when the
                // match ends inside it: the positions after it were probed just now and all missed
                const int nextP = max(lastEnd, p0 + LZ4_WAVE);
                pfPos = nextP;
                pfV8 = (nextP + lane < mfl) ? *(const u64_unaligned *)(src + nextP + lane) : 0ull;
                ENC_LAP(2);
                // ---- table: the probed positions outside the selected matches (:998) ----
                if (valid && !covered && myPos < nextP) { table[h] = (TabT)myPos; ENC_TAG_SET(h, tg); }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int myEnd = myPos + (int)myMl;
                ENC_LAP(3);
                // ---- catch up (:1019), bounded by the previous match: the bytes between my run's head and me
                // are known equal, the head measured up to 8 more before itself ----
                int mstart = myPos, mcand = (int)cand;
                if (sel) {
                    const int room = min(mstart - prevEnd, mcand);
                    const int back = min(room, (lane - myHead) + (int)hback);
                    mstart -= back; mcand -= back;
                }
                ENC_LAP(4);
                // ---- park the selected sequences in the queue (stable compaction with one ds_permute
                // per field: selected lanes go to [qCnt, qCnt+k), the others fill the remaining lanes) ----
                {
                    const int k = (int)__builtin_popcountll(selm);
                    if (qCnt + k > LZ4_WAVE) flush_queue();
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(selm >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)selm, 0u));
                    const int dest = (sel ? qCnt + rank : qCnt + k + (lane - rank)) & 63;
                    const int r0 = __builtin_amdgcn_ds_permute(dest << 2, prevEnd);
                    const int r1 = __builtin_amdgcn_ds_permute(dest << 2, mstart);
                    const int r2 = __builtin_amdgcn_ds_permute(dest << 2, myEnd - mstart);
                    const int r3 = __builtin_amdgcn_ds_permute(dest << 2, mstart - mcand);
                    if (lane >= qCnt && lane < qCnt + k) { qPrev = r0; qStart = r1; qLen = r2; qOff = r3; }
                    qCnt += k;
                }
                anchor = lastEnd;
                p = nextP;
                missAcc = miss0;
                ENC_LAP(5);
#ifdef ENC_STATS
                est[6] += 1;
#endif
                continue;
            }
            // ===== strided window (acceleration > 1 or after a miss streak): first match only =====
            // probe positions p, p+1, p+2, then every `step`: like the reference, which probes
            // ip, ip+1, ip+2 after each match before its stride takes over (:1159, :1200, :956-967)
            const int64_t myPos64 = p + (lane < 3 ? (int64_t)lane : 2 + (int64_t)(lane - 2) * step);
            const bool valid = myPos64 < (int64_t)mfl;
            const int myPos = valid ? (int)myPos64 : 0;
            uint64_t v8 = 0;
            uint32_t h = 0, tg = 0, cand = 0;
            bool hit = false;
            if (valid) {
                v8 = *(const u64_unaligned *)(src + myPos);
                ENC_HT(v8, h, tg);
                if (tab_candidate<TabT, DICT>(table[h], myPos, cand) && ENC_TAG_OK(h, tg))
                    hit = (*(const u32_unaligned *)(src + cand) == (uint32_t)v8);
            }
            const uint64_t m = __ballot(hit);
            const int first = m ? (int)__builtin_ctzll(m) : LZ4_WAVE;
            if (valid && lane <= first) { table[h] = (TabT)myPos; ENC_TAG_SET(h, tg); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (!m) {
                if (missAcc < 0x7fffff00u) missAcc += LZ4_WAVE;
                p += 2 + 62 * step;
                continue;
            }
            int mpos = __builtin_amdgcn_readlane(myPos, first);
            int cpos = __builtin_amdgcn_readlane((int)cand, first);

            // ---- catch up (:1019): extend backwards while bytes agree ----
            {
                const int maxBack = min(mpos - anchor, cpos);
                int back = 0;
                while (back < maxBack) {
                    const int k = back + lane + 1;
                    const bool eq = (k <= maxBack) && (src[mpos - k] == src[cpos - k]);
                    const uint64_t ne = ~__ballot(eq);
                    const int run = ne ? (int)__builtin_ctzll(ne) : LZ4_WAVE;
                    back += run;
                    if (run < LZ4_WAVE) break;
                }
                mpos -= back; cpos -= back;
            }
            // ---- forward extension (:1092): first 4 bytes are known equal ----
            int ml = LZ4_MINMATCH;
#ifndef ENC_NO_PIPE
            ml += extend_long(mpos + LZ4_MINMATCH, cpos + LZ4_MINMATCH);      // the whole wave, 1 KiB a step
#else
            {
                const int maxLen = matchlimit - mpos;
                while (ml < maxLen) {
                    const int k = ml + lane;
                    const bool eq = (k < maxLen) && (src[mpos + k] == src[cpos + k]);
                    const uint64_t ne = ~__ballot(eq);
                    const int run = ne ? (int)__builtin_ctzll(ne) : LZ4_WAVE;
                    ml += run;
                    if (run < LZ4_WAVE) break;
                }
            }
#endif
            // ---- queue the sequence (written out by flush_queue) ----
            if (qCnt == LZ4_WAVE) flush_queue();
            if (lane == qCnt) { qPrev = anchor; qStart = mpos; qLen = ml; qOff = mpos - cpos; }
            qCnt++;
            anchor = mpos + ml;
            p = anchor;
            missAcc = miss0;
            // :1146 -- the reference registers ip-2 after every match
            if (anchor < mfl && lane == 0) { uint32_t h_, t_; ENC_HT(*(const u64_unaligned *)(src + anchor - 2), h_, t_); table[h_] = (TabT)(anchor - 2); ENC_TAG_SET(h_, t_); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }

#ifdef ENC_STATS
    if (stats && lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&stats[i], est[i]);
#endif
    flush_queue();
    if (SEG) return anchor + seg->base;
    // ---- last literals (:1204-1231) ----
    {
        const uint32_t lastRun = (uint32_t)(n - anchor);
        uint8_t *tok = op++;
        if (lane == 0) *tok = (uint8_t)(min(lastRun, 15u) << 4);
        if (lastRun >= 15) op = emit_ext_len(op, lastRun - 15);
        wave_copy_bytes(op, src + anchor, lastRun);
        op += lastRun;
    }
    return (int)(op - dst);
}

} // namespace lz4dev
