// encode_wave.hpp -- LZ4 block compressor, one wavefront per block.
//
// Replaces (per block) LZ4_compress_fast_continue -> LZ4_compress_generic
// (reference cbits/lz4.c:1565-1637, 851-1240) for INDEPENDENT blocks.
//
// It is not a transcription of the reference's serial probe loop.  The wave
// probes 64 positions at once:
//   * lane i hashes the 5 bytes at p + i*step with the reference's hash
//     (cbits/lz4.c:706-716, hashLog 12) and looks its candidate up in a
//     4096-entry table held in LDS (u16 positions for blocks <= 64 KiB,
//     u32 above; cbits/lz4.h:578-580 is the table being replaced);
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

// Emit a length >= 15 continuation (rest = len - 15): rest/255 bytes of 255 then rest%255.
__device__ __forceinline__ uint8_t *emit_ext_len(uint8_t *op, uint32_t rest)
{
    const uint32_t nff = rest / 255u;
    for (uint32_t i = (uint32_t)lane_id(); i < nff; i += LZ4_WAVE) op[i] = 255;
    if (lane_id() == 0) op[nff] = (uint8_t)(rest - nff * 255u);
    return op + nff + 1;
}

template <typename TabT>
__device__ int encode_block_wave(const uint8_t *src, int n, uint8_t *dst, int accel, TabT *table)
{
    const int lane = lane_id();
    uint8_t *op = dst;
    int anchor = 0;

    if (n == 0) {                       // cbits/lz4.c:1263-1273: empty input -> single 0 token
        if (lane == 0) dst[0] = 0;
        return 1;
    }

    // zero the table (positions are block-relative; 0 is a real position, as in the reference)
    {
        uint32_t *t32 = (uint32_t *)table;
        const int nd = (int)(4096 * sizeof(TabT) / 4);
        for (int i = lane; i < nd; i += LZ4_WAVE) t32[i] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    if (n >= 13) {                                      // LZ4_minLength, :221,:921
        const int mfl = n - LZ4_MFLIMIT + 1;            // match start must be < mfl (:883)
        const int matchlimit = n - LZ4_LASTLITERALS;    // match end must be <= matchlimit (:884)
        const uint32_t miss0 = (uint32_t)accel << 6;
        uint32_t missAcc = miss0;
        int64_t p = 0;

        while (p < mfl) {
            // probe positions p, p+1, p+2, then every `step`: like the reference, which probes
            // ip, ip+1, ip+2 after each match before its stride takes over (:1159, :1200, :956-967)
            const int64_t step = (int64_t)(missAcc >> 6);
            const int64_t myPos64 = p + (lane < 3 ? (int64_t)lane : 2 + (int64_t)(lane - 2) * step);
            const bool valid = myPos64 < (int64_t)mfl;
            const int myPos = valid ? (int)myPos64 : 0;
            uint64_t v8 = 0;
            uint32_t h = 0, cand = 0;
            bool hit = false;
            if (valid) {
                v8 = *(const u64_unaligned *)(src + myPos);
                h = hash5(v8);
                cand = (uint32_t)table[h];
                if (cand < (uint32_t)myPos && (uint32_t)myPos - cand <= LZ4_MAXDIST)
                    hit = (*(const u32_unaligned *)(src + cand) == (uint32_t)v8);
            }
            const uint64_t m = __ballot(hit);
            const int first = m ? (int)__builtin_ctzll(m) : LZ4_WAVE;
            if (valid && lane <= first) table[h] = (TabT)myPos;
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
            // ---- emit sequence (:1022-1046, :1065-1135) ----
            {
                const uint32_t lit = (uint32_t)(mpos - anchor);
                const uint32_t mc = (uint32_t)(ml - LZ4_MINMATCH);
                const uint32_t off = (uint32_t)(mpos - cpos);
                uint8_t *tok = op++;
                if (lane == 0) *tok = (uint8_t)((min(lit, 15u) << 4) | min(mc, 15u));
                if (lit >= 15) op = emit_ext_len(op, lit - 15);
                wave_copy_bytes(op, src + anchor, lit);
                op += lit;
                if (lane == 0) { op[0] = (uint8_t)off; op[1] = (uint8_t)(off >> 8); }
                op += 2;
                if (mc >= 15) op = emit_ext_len(op, mc - 15);
            }
            anchor = mpos + ml;
            p = anchor;
            missAcc = miss0;
            // :1146 -- the reference registers ip-2 after every match
            if (anchor < mfl && lane == 0) table[hash5(*(const u64_unaligned *)(src + anchor - 2))] = (TabT)(anchor - 2);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }

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
