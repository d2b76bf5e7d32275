// encode_wave.hpp -- LZ4 block compressor, one wavefront per block.
//
// Replaces (per block) LZ4_compress_fast_continue -> LZ4_compress_generic
// (reference cbits/lz4.c:1565-1637, 851-1240): independent blocks, or (DICT) blocks
// linked to the block in front of them; SEG: a block cut into segments for small batches.
//
// It is not a transcription of the reference's serial probe loop.  Three forms of a
// window of probes live in encode_block_wave, picked by what the block is doing:
//   * dense windows (acceleration 1, matches being found: nearly all of a compressible
//     block): 64 consecutive positions per window -- two windows per step for blocks of up
//     to 64 KiB -- hashed with the reference's hash (cbits/lz4.c:706-716, hashLog 12) into a
//     4096-entry LDS table of 16-bit positions with a 4-bit tag per entry
//     (cbits/lz4.h:578-580 is the table being replaced); every run head verified and measured
//     by a group of four (or two) lanes in ONE round trip, greedy selection by a scalar loop,
//     EVERY match of the window emitted.  The section "Dense window, one round trip" below;
//     DESIGN.md section 0a has its instruction inventory;
//   * the per-lane dense window of round 2 (a block's first window and its last 200 bytes);
//   * strided windows (acceleration > 1, or after a miss streak): a ballot picks the FIRST
//     lane whose candidate verifies (4 equal bytes within 65535, :1003-1012), lanes up to it
//     publish their positions, the match is extended backwards ("catch up", :1019) and
//     forwards (LZ4_count, :603-626) by the whole wave.
// Sequences are parked one per lane and emitted 64 at a time exactly as :1022-1046,
// :1065-1135 (emit_sequences).
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
// ENC_TAB_N: positions in the table.  4096 is what ships (with the tags: 10 KiB of LDS, 16 waves per CU).  3072 (7.5 KiB: 20 waves per
// CU, five per SIMD with at most 102 VGPRs) is round 6's occupancy experiment -- measured, not shipped: text loses 3.5 % of its
// ratio (oracle/sim_encode2.c: 1.838 -> 1.774, below the reference's 1.806) -- see DESIGN.md 0c.
#ifndef ENC_TAB_N
#define ENC_TAB_N 4096
#endif
#define ENC_STR2_(x) #x
#define ENC_STR_(x) ENC_STR2_(x)
#define ENC_TAG_OFFSET_BYTES 8192              /* = ENC_TAB_N * 2, as a literal for the instruction's offset field */
#if ENC_TAB_N != 4096
#undef ENC_TAG_OFFSET_BYTES
#define ENC_TAG_OFFSET_BYTES 6144
static_assert(ENC_TAB_N == 3072, "ENC_TAB_N: 4096 or 3072");
#endif
#define ENC_TABLE_ENTRIES (ENC_TAB_N + ENC_TAB_N / 4)        /* 16-bit units: the positions + a nibble each */
#define ENC_TAG_DECL uint32_t *tags = (uint32_t *)(table + ENC_TAB_N);
// bucket << 4 | tag of a position's first five bytes
#if ENC_TAB_N == 4096
__device__ __forceinline__ uint32_t enc_hx(uint64_t v) { return hash5x(v); }
#else
__device__ __forceinline__ uint32_t enc_hx(uint64_t v)
{
    const uint32_t x = (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 20));     // 12 bits of bucket, 4 of tag, 4 unused
    return ((((x >> 8) * (uint32_t)ENC_TAB_N) >> 12) << 4) | ((x >> 4) & 15u);
}
#endif
#define ENC_HT(v, h, t) const uint32_t hx_ = enc_hx(v); h = hx_ >> 4; t = hx_ & 15u
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

// s_setprio per phase of a pair of windows: probe + heads + requests (P), the wait for the groups' bytes + their lengths (M),
// hits + the selection's scalar loop (S), finish + table (F).  Four waves share a SIMD's issue; the wave that is in its serial
// selection, or has requests to get out, goes first.  Measured (lzsynth / text, GB/s; all 0: 196-197 / 183): P M S F =
// 1 0 3 0: 201 / 186, 2 0 3 0: 201 / 186, 3 0 3 0: 201 / 185, 2 1 3 0: 202 / 186, 3 1 3 2: 201 / 186, 2 0 3 1: 202 / 187,
// 2 0 3 2: 200 / 183, 3 1 3 1: 200 / 185; the emission at 0 or 3 instead of F: 202 / 186.
#ifndef ENC_PRIO_P
#define ENC_PRIO_P 2
#define ENC_PRIO_M 0
#define ENC_PRIO_S 3
#define ENC_PRIO_F 1
#endif
#ifndef ENC_PRIO1
#define ENC_PRIO1 1       // the same in the one-window-per-step form (blocks above 64 KiB, segments)
#endif
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
// the same on the tag word `idx` of the table that starts at `table`: the tags' distance from the table (4096 16-bit
// positions) travels in the instruction's offset field, so the word's address is the one its read already computed
__device__ __forceinline__ void lds_mskor_tag(const void *table, uint32_t idx, uint32_t mask, uint32_t bits)
{
    asm volatile("ds_mskor_b32 %0, %1, %2 offset:" ENC_STR_(ENC_TAG_OFFSET_BYTES) : : "v"((uint32_t)(uintptr_t)table + idx * 4u), "v"(mask), "v"(bits) : "memory");
}

// v_writelane_b32: lane k of v becomes the wave-uniform x.  (This clang has no builtin for it; the LLVM intrinsic is
// reached by name, and the compiler then takes care of gfx9's rule that the lane select travels in M0 when the value
// is an SGPR too.)
extern "C" __device__ int enc_llvm_writelane(int x, int k, int v) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int enc_writelane(int v, int x, int k) { return enc_llvm_writelane(x, k, v); }

// ballot of a predicate (HIP's __ballot takes an int: the predicate would travel through a register as 0 / 1 and a compare)
__device__ __forceinline__ uint64_t enc_ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// The shape of the groups that measure the run heads' matches (encode_wave.hpp, "dense window"): GL lanes of 16 bytes each
// per head -- 4: the 8 bytes before the head and the 56 after, 16 heads a round; 2: 8 + 24, 32 heads a round.
template <int GLANES> struct EncGroups {
    static constexpr uint32_t GL = GLANES, GR = 64u / GLANES;          // lanes per group, groups per round
    static constexpr int GSH = (GLANES == 4) ? 4 : 3;                  // log2 of a group's byte address step (4 bytes a lane)
};

// minimum over each group of four lanes, in all four: two v_min_u32 with a quad_perm DPP operand.  (Written out: the
// compiler keeps a copy and a v_mov_dpp per step otherwise.  The s_nop is the two wait states a DPP read of a
// freshly written register needs; the hazard pass does not look into asm.)
__device__ __forceinline__ uint32_t enc_pair_min(uint32_t g)      // ... over each pair of lanes (groups of two: ENC_GROUP_LANES 2)
{
    uint32_t r;
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(g));
    return r;
}
__device__ __forceinline__ uint32_t enc_quad_min(uint32_t g)
{
    uint32_t r;
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(g));
    return r;
}

// Greedy selection over one window's hit mask, the scalar way: take the first hit lane, read the END of its match from
// that lane (endv), continue with the hit lanes at or behind that end.  hm: hit lanes still in play (0 on return unless a
// lane of capm came up: that lane's number is returned in kcap, untouched, for the caller to extend); selm gathers the
// lanes taken, pEnd the end of the last one.  Written out because the loop is the scalar unit's largest item per
// window: 11 scalar instructions and one v_readlane per match (the compiler's form of the same loop: 22).
__device__ __forceinline__ void enc_select_run(uint64_t &hm, uint64_t &selm, int &pEnd, int &kcap, const uint64_t capm, const int endv, const int p0w)
{
    int k, endk, sh;
    uint64_t tmp;
    asm("s_mov_b32 %3, -1\n\t"
        "s_cmp_eq_u64 %0, 0\n\t"
        "s_cbranch_scc1 2f\n"
        "1:\n\t"
        "s_ff1_i32_b64 %4, %0\n\t"
        "s_bitcmp1_b64 %8, %4\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_readlane_b32 %5, %9, %4\n\t"
        "s_bitset1_b64 %1, %4\n\t"
        "s_sub_i32 %6, %5, %10\n\t"
        "s_mov_b32 %2, %5\n\t"
        "s_cmp_gt_i32 %6, 63\n\t"
        "s_cbranch_scc1 4f\n\t"
        "s_bfm_b64 %7, %6, 0\n\t"
        "s_andn2_b64 %0, %0, %7\n\t"
        "s_cmp_lg_u64 %0, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_branch 2f\n"
        "3:\n\t"
        "s_mov_b32 %3, %4\n\t"
        "s_branch 2f\n"
        "4:\n\t"
        "s_mov_b64 %0, 0\n"
        "2:\n"
        : "+s"(hm), "+s"(selm), "+s"(pEnd), "=&s"(kcap), "=&s"(k), "=&s"(endk), "=&s"(sh), "=&s"(tmp)
        : "s"(capm), "v"(endv), "s"(p0w)
        : "scc");
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
    // once per window for the handful of lanes that own a match.  Fields: literal start, match start, match
    // length, offset.  SMALLQ (positions below 64 Ki: independent blocks of up to 64 KiB): two fields to a register --
    // q0 = literal start | match start << 16, q1 = match length | offset << 16 -- so that a window moves its
    // sequences into the queue with two cross-lane operations instead of four.
    constexpr bool SMALLQ = !DICT && !SEG && sizeof(TabT) == 2;
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    int qCnt = 0;                                      // uniform
    uint64_t pendM0 = 0, pendM1 = 0;                   // dense windows: queue moves under way (commit_pending)
    int pendR[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    auto commit_pending = [&]() {
        if (__builtin_amdgcn_inverse_ballot_w64(pendM0)) { q0 = pendR[0][0]; q1 = pendR[0][1]; if (!SMALLQ) { q2 = pendR[0][2]; q3 = pendR[0][3]; } }
        if (__builtin_amdgcn_inverse_ballot_w64(pendM1)) { q0 = pendR[1][0]; q1 = pendR[1][1]; if (!SMALLQ) { q2 = pendR[1][2]; q3 = pendR[1][3]; } }
        pendM0 = 0; pendM1 = 0;
    };
    auto flush_queue = [&]() {
        commit_pending();
        if (qCnt == 0) return;
        int qPrev, qStart, qLen, qOff;
        if (SMALLQ) { qPrev = q0 & 0xffff; qStart = (int)((uint32_t)q0 >> 16); qLen = q1 & 0xffff; qOff = (int)((uint32_t)q1 >> 16); }
        else { qPrev = q0; qStart = q1; qLen = q2; qOff = q3; }
        if (SEG) {
            // segment mode: the sequences go to the segment's list; a second kernel stitches the lists into the block
            if (lane < qCnt) seg->list[seg->count + (uint32_t)lane] = seg_pack(qStart + seg->base, qLen, qOff);
            seg->count += (uint32_t)qCnt;
            qCnt = 0;
            return;
        }
#ifndef ENC_EXP_NOEMIT
        op = emit_sequences(src, op, qPrev, qStart, qLen, qOff, qCnt);
#else
        if (lane == 0) *(volatile int *)op = qPrev + qStart + qLen + qOff;      // (experiment: what the emission costs)
#endif
        qCnt = 0;
    };
    // one sequence whose fields are wave-uniform, into slot qCnt (the caller has made room)
    auto park_uniform = [&](const int prev, const int start, const int len, const int off) {
        if (SMALLQ) {
            q0 = enc_writelane(q0, prev | (start << 16), qCnt);
            q1 = enc_writelane(q1, len | (off << 16), qCnt);
        } else {
            q0 = enc_writelane(q0, prev, qCnt); q1 = enc_writelane(q1, start, qCnt);
            q2 = enc_writelane(q2, len, qCnt); q3 = enc_writelane(q3, off, qCnt);
        }
        qCnt++;
    };

    if (blockLen == 0) {                // cbits/lz4.c:1263-1273: empty input -> single 0 token
        if (SEG) return seg->base + dictLen;
        if (lane == 0) dst[0] = 0;
        return 1;
    }

    // zero the table (positions are block-relative; 0 is a real position, as in the reference)
    {
        uint32_t *t32 = (uint32_t *)table;
        const int nd = (int)(ENC_TAB_N * sizeof(TabT) / 4) + ENC_TAB_N / 8;    // positions, then the tags
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
        // Dense window, one round trip.  The window code further down (still used for the first and the last
        // windows of a block) is a chain of dependent memory round trips -- bytes -> table -> candidate -> one to
        // three extension steps per run head, in per-lane loops.  Here every run head is handed to a GROUP OF FOUR
        // LANES that requests, in one go, 16 bytes per lane around the head's position and around its candidate:
        // the 8 bytes before them (catch up, :1019) and the 56 after.  The tags make a candidate a real match nine
        // times in ten, so there is no separate verification step: one round trip per window, every lane busy.
        //   probe    hash the 64 positions, read buckets and tags, run heads, groups, requests
        //   lengths  first difference per group (one chain of v_ffbl / v_min and two quad-DPP steps), every lane
        //            learns its run's head with a DPP max-scan
        //   select   greedy, left to right: a scalar loop that only reads the END of each match it takes
        //   finish   what each selected lane needs -- the end of the selected match before it -- comes from one more
        //            max-scan; catch up, the sequence's fields and their move into the queue are vector code whose
        //            cost does not depend on the number of matches
        // Round 5 rewrote this section for instruction count (the kernel is bound by vector issue: round 4 measured
        // 228 vector instructions per window of 75 input bytes): group lengths 44 -> 28 instructions, no per-match
        // vector work in the selection loop (it was 8 per match), lane predicates taken from scalar masks
        // (inverse ballot) instead of 64-bit shifts per lane, the queue's fields packed two to a register for
        // blocks of up to 64 KiB (two cross-lane moves per window instead of four), no address selects for
        // unused groups (they read bytes that are there anyway).
        // ===================================================================================================
        const LZ4_GLOBAL uint8_t *gsrc = as_global(src);
        auto load16 = [&](uint32_t off) -> dev_v4 { return *(const LZ4_GLOBAL dev_v4u *)(gsrc + off); };
        // EXPERIMENT (round 6, ENC_STAGE=1; measured, not shipped -- DESIGN.md 0c, profiles/r06_encode_structural_attempt.txt): the
        // pair form's INPUT bytes through LDS.  A scattered 16-byte request per lane keeps a CU's one texture addresser busy for 64
        // cycles and a pair of windows makes six of them (TA busy 94 % of the kernel; one more per window: -25 %).  Three of the six
        // ask for bytes the wave is looking at anyway -- the positions' own 8 bytes and the 64 bytes around every run head.  With
        // the stage one aligned dword per lane (256 bytes from 8 bytes before the pair: a 16-cycle request) is written to 256 bytes
        // of LDS behind the table, and both come out of it with aligned dword reads and a byte funnel; only the candidates' side is
        // still scattered.  Result: requests -45 %, TA busy 94 -> 53 %, waiting 48 -> 40 % of wave cycles -- and the same time: the
        // funnels are 35 vector instructions more per pair (+12 %), which is what the waiting had been hiding, and the 256 bytes
        // take the 16th wave of a CU (10 KiB x 16 = all of LDS): -9 %.
#ifndef ENC_STAGE
#define ENC_STAGE 0
#endif
        uint8_t *const stage = (uint8_t *)table + ENC_TABLE_ENTRIES * sizeof(uint16_t);
        uint32_t pfD = 0;           // the stage's dword of this lane for the pair that starts at stNext + 8 + (0..3)
        int stNext = 0, stCur = 0;  // block-relative position of the stage's first byte: of the pair requested / of the pair in LDS
        auto stage_request = [&](const int p) {
            stNext = p - 8 - (int)(((uint32_t)(uintptr_t)gsrc + (uint32_t)p) & 3u);
            const int q = stNext + 4 * lane;
            pfD = (q < n) ? *(const LZ4_GLOBAL uint32_t *)(gsrc + q) : 0u;       // (an aligned dword with a byte of the block in it)
        };
        // 8 / 16 bytes at any byte offset of the stage: aligned dwords + v_alignbyte (an LDS access that is not naturally aligned
        // costs a cycle per lane, decode_par.hpp)
        auto stage_u64 = [&](const uint32_t addr) -> uint64_t {
            const uint32_t *q = (const uint32_t *)(stage + (addr & ~3u));
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], sh = addr & 3u;
            return ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32) | __builtin_amdgcn_alignbyte(d1, d0, sh);
        };
        auto stage_v4 = [&](const uint32_t addr) -> dev_v4 {
            const uint32_t *q = (const uint32_t *)(stage + (addr & ~3u));
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4], sh = addr & 3u;
            dev_v4 r;
            r.x = __builtin_amdgcn_alignbyte(d1, d0, sh); r.y = __builtin_amdgcn_alignbyte(d2, d1, sh);
            r.z = __builtin_amdgcn_alignbyte(d3, d2, sh); r.w = __builtin_amdgcn_alignbyte(d4, d3, sh);
            return r;
        };
        const bool pipeFits = n < (1 << 24);           // a group's candidate travels in 25 bits, an end in 24
#ifndef ENC_GROUPS
#define ENC_GROUPS 0          // 0: adaptive; 2 / 4: one shape (measurements)
#endif
        // the groups' shape (see shape_update below): windows and long matches counted, extensions made since the count was
        // last looked at
        int shapeWin = 0, shapeLong = 0, shapeExt = 0;
        bool shapeCounting = ENC_GROUPS == 0;
        // ... and whether the block has runs of equal bytes worth looking for (lw_heads, ENC_SELF_RUN): looked for in the
        // block's first 32 windows, and from there on only if ENC_SELF_RUN_MIN of those had one
        bool selfRunOn = true;
        int selfRunSeen = 0;
        // lane constants of the groups: lane j of a group holds bytes [16 j - 8, 16 j + 8) relative to the head
        // (two shapes, ENC_GROUPS in kernels: the wave switches between them with what the block's matches look like)
        // bit offsets of a lane's four words; a group's lane 0 holds the 8 bytes BEFORE the head in its first two: ~0 keeps
        // them out of the minimum
        struct GC { uint32_t c0, c1, c2, c3, j16; };
        auto group_consts = [&](const uint32_t j) -> GC {
            return GC{j ? j << 7 : 0xffffffffu, j ? (j << 7) | 32u : 0xffffffffu, (j << 7) | 64u, (j << 7) | 96u, j << 4};
        };
        const GC gc4 = group_consts((uint32_t)lane & 3u);
        // (groups of two make theirs where they are used: five registers that the groups of four's steady state does not carry)
        const uint32_t lanePay = (uint32_t)lane << 25;        // a head's message to its group: lane | candidate - 8

        // First difference of a group's 64 + 64 bytes, in every lane of the group:
        //   bits 0..7  t = 8 + equal bytes from the head on (8..64; 64 = the horizon of 56 bytes was reached)
        //   bits 8..11 equal bytes just before the head (0..8), meaningful in the group's lane 0 only
        // v_ffbl_b32 gives 0..31 or ~0: "or"-ing the word's bit offset in keeps ~0 for equal words, so the minimum over
        // the group is the first differing bit of its 512 (or ~0).
        auto group_len = [&](auto G, const dev_v4 &a, const dev_v4 &b) -> uint32_t {
            constexpr uint32_t GL = decltype(G)::GL;
            const GC gc = (GL == 4u) ? gc4 : group_consts((uint32_t)lane & 1u);
            const uint32_t x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
            uint32_t g = min(min(ffbl32(x0) | gc.c0, ffbl32(x1) | gc.c1), ffbl32(x2) | gc.c2);
            g = min(min(g, ffbl32(x3) | gc.c3), GL * 128u);                     // (the group's bits: nothing differs within the horizon)
            g = (GL == 4u) ? enc_quad_min(g) : enc_pair_min(g);
            const uint32_t t = g >> 3;
            const uint32_t back = min(min(ffbh32(x1), ffbh32(x0) | 32u) >> 3, 8u);
            return t | (back << 8);
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
                    const dev_v4 a = load16((uint32_t)(pe + off)), b = load16((uint32_t)(ce + off));
                    d = common16x(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
                    full = d == 16u;
                } else if (off < maxExtra) {
                    const uint32_t r = (uint32_t)(maxExtra - off);
                    while (d < r && gsrc[pe + off + (int)d] == gsrc[ce + off + (int)d]) d++;
                }
                const uint64_t ne = ~enc_ballot(full);
                if (ne == 0ull) {
                    total += 16 * LZ4_WAVE;
                    // A whole KiB was equal: a long run (zero pages, sparse files).  From here ENC_EXTEND_STRIDES KiB a
                    // round trip while all of it is equal; the first unequal stretch goes back to the exact loop above.
#ifndef ENC_EXTEND_STRIDES
#define ENC_EXTEND_STRIDES 2   /* (4: the pair form spills) */
#endif
                    while (total + ENC_EXTEND_STRIDES * 16 * LZ4_WAVE <= maxExtra) {
                        uint32_t diff = 0;
#pragma unroll
                        for (int u = 0; u < ENC_EXTEND_STRIDES; u++) {
                            const int o = total + u * 16 * LZ4_WAVE + 16 * lane;
                            const dev_v4 a = load16((uint32_t)(pe + o)), b = load16((uint32_t)(ce + o));
                            diff |= (a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w);
                        }
                        if (enc_ballot(diff != 0u) != 0ull) break;
                        total += ENC_EXTEND_STRIDES * 16 * LZ4_WAVE;
                    }
                    continue;
                }
                const int first = (int)__builtin_ctzll(ne);
                total += 16 * first + __builtin_amdgcn_readlane((int)d, first);
                break;
            }
            return total;
        };

        // One window's state.  Lane l stands for position p0w + l; idx = l (+ 64 in the second window of a pair).
        struct LW {
            uint32_t hx, oldp, tagWord, tmask, tbits;   // probe: hash bits, the bucket's entry and tag word as found, this position's tag in place
            uint32_t c8, off;                           // candidate - 8; position - candidate
            bool twoRounds;
            uint64_t candm, headm;                      // lanes with a candidate (right tag, 8 <= candidate < position); run heads among them
            uint32_t rank4, gi0;
            dev_v4 a0, b0, a1, b1;
            uint32_t r;                                 // head lanes: their group's result (group_len)
            uint32_t hv;                                // (head idx + 1) << 16 | back << 8 | t + head idx, of the run this lane belongs to
            uint32_t m0;                                // hit lanes: match length from this lane on (0..56; 63 once extended)
            int endv;                                   // hit lanes: end of that match
            uint64_t hitm, capm;
            uint64_t longm;                             // hit lanes whose run is at least 24 bytes long from its head: groups of two would have to extend them
        };
        auto lw_probe = [&](LW &W, const uint32_t pos, const uint64_t v8, const bool insertNow) {
            const uint32_t hx = enc_hx(v8);
            const uint32_t tsh = (hx >> 2) & 28u;                  // (h & 7) * 4
            W.hx = hx;
            W.tmask = 15u << tsh;
            W.tbits = (hx & 15u) << tsh;
            W.oldp = table[hx >> 4];
            W.tagWord = tags[hx >> 7];
            if (insertNow) {                                       // every position, at once: the window behind this one probes next
                table[hx >> 4] = (TabT)pos;
                lds_mskor_tag(table, hx >> 7, W.tmask, W.tbits);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        };
#ifndef ENC_SELF_RUN
#define ENC_SELF_RUN 1
#endif
#ifndef ENC_SELF_RUN_MIN
#define ENC_SELF_RUN_MIN 12    // windows (of the block's first 32) with such a position for the search to stay on
#endif
        // lo = the position's first four bytes; loBefore = those of the position in front of lane 0 (anything with another
        // first byte where that is not known)
        auto lw_heads = [&](auto G, LW &W, const uint32_t pos, const uint32_t pos8, const uint32_t offBefore, const uint64_t okBefore,
                            const uint32_t lo, const uint32_t loBefore) {
            constexpr uint32_t GR = decltype(G)::GR; constexpr int GSH = decltype(G)::GSH;
            // (predicates are kept as scalar masks -- one compare each, combined by the scalar unit -- and turned back
            // into lane predicates where a select needs them)
            uint64_t okm = enc_ballot((W.tagWord & W.tmask) == W.tbits);
            uint32_t c8;
            if (DICT || sizeof(TabT) != 2) {
                uint32_t cand = 0;
                okm &= enc_ballot(tab_candidate<TabT, DICT>((TabT)W.oldp, (int)pos, cand));
                okm &= enc_ballot(cand >= 8u);
                c8 = cand - 8u;
            } else {
                c8 = W.oldp - 8u;                                  // positions stay below 64 Ki: the entry IS the position,
                okm &= enc_ballot(c8 < pos8);                      // and 8 <= cand < pos in one unsigned compare (:1003-1006)
            }
            if (ENC_SELF_RUN && selfRunOn) {
                // The positions of one window cannot be each other's candidates, so a run of equal bytes -- indentation, zero
                // fill -- that starts inside a window found no match until the window behind it (the reference reaches the
                // offset-1 match at the run's second byte).  A position without a table candidate whose four bytes repeat the
                // byte in front of it takes position - 1: consecutive such positions share the distance, so a run is ONE head.
                // (oracle/sim_encode2.c, "self runs": source code 3.22 -> 3.29 against the reference's 3.36, text unchanged)
                // (two instructions for a window without four equal bytes anywhere; the rest behind a scalar branch)
                uint64_t selfm = enc_ballot(lo == __builtin_amdgcn_perm(lo, lo, 0u)) & ~okm;
                if (selfm) {
                    const uint32_t loLeft = (uint32_t)__builtin_amdgcn_update_dpp((int)loBefore, (int)lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    selfm &= enc_ballot(((lo ^ loLeft) & 0xffu) == 0u);
                    if (__builtin_amdgcn_inverse_ballot_w64(selfm)) c8 = pos8 - 1u;
                    okm |= selfm;
                    selfRunSeen += selfm != 0ull;
                }
            }
            W.candm = okm;
            W.c8 = c8;
            // run heads: a candidate that continues its left neighbour's -- the same distance back, and the neighbour has a
            // candidate -- belongs to the same copied region
            W.off = pos8 - c8;
            const uint32_t offLeft = (uint32_t)__builtin_amdgcn_update_dpp((int)offBefore, (int)W.off, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            W.headm = okm & ~(enc_ballot(W.off == offLeft) & ((okm << 1) | okBefore));
            W.twoRounds = __builtin_popcountll(W.headm) > (int)GR;
            W.rank4 = enc_mbcnt(W.headm) << GSH;                   // byte address of lane GL * rank
            // heads 0..15, one per group of four lanes (lane 1 takes what the others send)
            const int dest = __builtin_amdgcn_inverse_ballot_w64(W.headm & enc_ballot(W.rank4 < 256u)) ? (int)W.rank4 : 4;
            W.gi0 = (uint32_t)__builtin_amdgcn_ds_permute(dest, (int)(lanePay | c8));
        };
        // a group's requests.  A group without a head decodes lane 0, candidate 8: bytes that are there (p0w >= 8)
        auto lw_fetch = [&](auto G, const uint32_t giRaw, const int p0w, dev_v4 &a, dev_v4 &b) {
            constexpr uint32_t GL = decltype(G)::GL;
            const uint32_t j16 = (GL == 4u) ? gc4.j16 : ((uint32_t)lane & 1u) << 4;
            const uint32_t gi = (GL == 4u) ? (uint32_t)__builtin_amdgcn_mov_dpp((int)giRaw, 0x00 /* quad_perm 0,0,0,0 */, 0xf, 0xf, true)
                                           : (uint32_t)__builtin_amdgcn_mov_dpp((int)giRaw, 0xa0 /* quad_perm 0,0,2,2 */, 0xf, 0xf, true);
            if (PAIR && ENC_STAGE) a = stage_v4((uint32_t)(p0w - 8 - stCur) + (gi >> 25) + j16);
            else a = load16((uint32_t)(p0w - 8) + (gi >> 25) + j16);
            b = load16((gi & 0x1ffffffu) + j16);
#ifdef ENC_EXP_EXTRA_LOAD
            {   // experiment: one more 16-byte request per lane and window -- is the texture addresser the limit?  
                dev_v4 x_ = load16((uint32_t)(p0w - 8) + (gi >> 25) + (j16 ^ 16u));
                asm volatile("" : : "v"(x_.x), "v"(x_.y), "v"(x_.z), "v"(x_.w));
            }
#endif
        };
        auto lw_loads = [&](auto G, LW &W, const int p0w) {
            lw_fetch(G, W.gi0, p0w, W.a0, W.b0);
            if (W.twoRounds) {                                     // heads 16..31 (one window in four has them)
                const int dest = __builtin_amdgcn_inverse_ballot_w64(W.headm & enc_ballot((W.rank4 >> 8) == 1u)) ? (int)(W.rank4 & 255u) : 4;
                lw_fetch(G, (uint32_t)__builtin_amdgcn_ds_permute(dest, (int)(lanePay | W.c8)), p0w, W.a1, W.b1);
            }
        };
        // heads beyond the 32nd of a window: one more round of groups at a time, requested and waited for here.
        // Generated text has 14-17 heads per window; source code 25-27, and leaving the heads past the 32nd without a
        // match cost 2 % of its ratio (oracle/sim_encode2.c, headcap32).
        auto more_rounds = [&](auto G, const LW &W, const int p0w, uint32_t r) -> uint32_t {
            constexpr uint32_t GR = decltype(G)::GR;
            const int nH = (int)__builtin_popcountll(W.headm);
            for (int rr = 2; rr * (int)GR < nH; rr++) {
                const int dest = __builtin_amdgcn_inverse_ballot_w64(W.headm & enc_ballot((int)(W.rank4 >> 8) == rr)) ? (int)(W.rank4 & 255u) : 4;
                dev_v4 a, b;
                lw_fetch(G, (uint32_t)__builtin_amdgcn_ds_permute(dest, (int)(lanePay | W.c8)), p0w, a, b);
                const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(W.rank4 & 255u), (int)group_len(G, a, b));
                if ((int)(W.rank4 >> 8) == rr) r = r2;
            }
            return r;
        };
        // lengths, in two steps so that a pair's two cross-lane reads travel together.  lw_measure: each head lane fetches its
        // group's result (W.r).  lw_hits: headConst = (idx + 1) << 16 | idx, idx8 = idx + 8; hvBefore = the scan value of the
        // last lane of the window before (its run may go on into this one)
        auto lw_measure = [&](auto G, LW &W, const int p0w) {
            constexpr uint32_t GR = decltype(G)::GR;
            uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(W.rank4 & 255u), (int)group_len(G, W.a0, W.b0));
            if (W.twoRounds) {
                const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(W.rank4 & 255u), (int)group_len(G, W.a1, W.b1));
                if (W.rank4 >= 256u) r = r1;
            }
#ifdef ENC_STATS
            est[7] += (unsigned)__builtin_popcountll(W.headm);
#endif
            if (__builtin_popcountll(W.headm) > 2 * (int)GR) r = more_rounds(G, W, p0w, r);
            W.r = r;
        };
        auto lw_hits = [&](auto G, LW &W, const int p0w, const uint32_t headConst, const uint32_t idx8, const uint32_t hvBefore) {
            constexpr uint32_t GL = decltype(G)::GL;
            // every lane learns its run's head with a max-scan: head lanes put their index above their result, the others 0,
            // and the largest value at or below a lane belongs to the nearest head below it.  The low byte is t + head idx,
            // so that a lane's own length is one subtraction: (t - 8) - (idx - head idx).
            W.hv = max(enc_scan_max(__builtin_amdgcn_inverse_ballot_w64(W.headm) ? W.r + headConst : 0u), hvBefore);
            const uint32_t E = W.hv & 0xffu;                      // t + head idx
            const int m0 = (int)E - (int)idx8;
            W.m0 = (uint32_t)m0;
            W.endv = (int)E + (p0w - 8 - (int)(idx8 - 8u - (uint32_t)lane));        // position + m0 (the bracket is wave-uniform)
            W.hitm = W.candm & enc_ballot(m0 >= LZ4_MINMATCH);
            // ... and whether its run reached the horizon: t = 64
            const uint32_t tm1 = (W.hv & 0xffu) - (W.hv >> 16);    // t - 1
            W.capm = W.hitm & enc_ballot(tm1 == GL * 16u - 1u);
            W.longm = 0;
            if (GL == 4u && shapeCounting) W.longm = W.hitm & enc_ballot(tm1 >= 31u);
        };
        // greedy selection in this window, continuing from pEnd (the end of the last selected match so far): scalar, one
        // v_readlane per match taken
        auto lw_select = [&](LW &W, const int p0w, int &pEnd) -> uint64_t {
            uint64_t hm = W.hitm;
            const int lowcut = pEnd - p0w;
            if (lowcut > 0) hm = (lowcut >= LZ4_WAVE) ? 0ull : (hm & (~0ull << lowcut));
            uint64_t selm = 0;
            for (;;) {
                int k;
                enc_select_run(hm, selm, pEnd, k, W.capm, W.endv, p0w);
                if (k < 0) break;
                // lane k's run reached the horizon: the whole wave counts on
                shapeExt++;
                int endk = __builtin_amdgcn_readlane(W.endv, k);
                const int ce = __builtin_amdgcn_readlane((int)W.c8, k) + 8 + (endk - (p0w + k));
                endk += extend_long(endk, ce);
                W.endv = enc_writelane(W.endv, endk, k);
                W.m0 = (uint32_t)enc_writelane((int)W.m0, 63, k);        // (any length from ENC_END2_MINLEN up: an inline constant)
                selm |= 1ull << k;
                pEnd = endk;
                const int sh = endk - p0w;
                hm = (sh >= LZ4_WAVE) ? 0ull : (hm & (~0ull << sh));
            }
            return selm;
        };
        // The selected lanes' sequences.  P = end of the selected match before me (an exclusive max-scan over the selected
        // lanes' ends, the match's clipped length riding in the low byte); covered = my position lies strictly inside a
        // selected match and is not the one position (end - 2 of a long match, :1146) that stays registered.  The fields go
        // to queue slots qCnt.. with one ds_permute per register; their results are looked at by commit_pending.
        auto lw_finish = [&](LW &W, const uint32_t pos, const uint32_t pos8, const uint32_t idx1, const uint64_t selm, const int pStart, const int slot) -> bool {
            const bool sel = __builtin_amdgcn_inverse_ballot_w64(selm);
            const uint32_t e = sel ? ((uint32_t)W.endv << 8) | W.m0 : 0u;
            uint32_t sc = enc_scan_max((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
            sc = max(sc, (uint32_t)pStart << 8);
            const uint32_t P = sc >> 8;
            const bool covered = pos < P && !((sc & 0xffu) >= (uint32_t)ENC_END2_MINLEN && pos + 2u == P);
            // catch up (:1019), bounded by the previous match: the bytes between my run's head and me are known equal, the
            // head's group measured up to 8 more before the head
            const uint32_t backAvail = (idx1 - (W.hv >> 16)) + ((W.hv >> 8) & 15u);
            const uint32_t back = min(min(backAvail, pos - P), W.c8);
            const uint32_t start = pos - back, mlen = (uint32_t)W.endv - start, off = W.off;
            const int k = (int)__builtin_popcountll(selm);
            const int trash = ((qCnt + 40) & 63) << 2;              // a slot outside [qCnt, qCnt + k): k <= 17
            uint32_t slot4 = __builtin_amdgcn_mbcnt_hi((uint32_t)(selm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)selm, (uint32_t)qCnt)) << 2;
            asm("" : "+v"(slot4));                                  // (computed for every lane: one select below, no branch around it)
            const int dest = sel ? (int)slot4 : trash;
            if (SMALLQ) {
                pendR[slot][0] = __builtin_amdgcn_ds_permute(dest, (int)(P | (start << 16)));
                pendR[slot][1] = __builtin_amdgcn_ds_permute(dest, (int)(mlen | (off << 16)));
            } else {
                pendR[slot][0] = __builtin_amdgcn_ds_permute(dest, (int)P);
                pendR[slot][1] = __builtin_amdgcn_ds_permute(dest, (int)start);
                pendR[slot][2] = __builtin_amdgcn_ds_permute(dest, (int)mlen);
                pendR[slot][3] = __builtin_amdgcn_ds_permute(dest, (int)off);
            }
            (slot ? pendM1 : pendM0) = k ? (((1ull << k) - 1ull) << qCnt) : 0ull;
            qCnt += k;
            return covered;
        };
        // this position into its bucket / the bucket's entry as found put back if it still holds this position
        auto lw_insert = [&](const LW &W, const uint32_t pos) {
            table[W.hx >> 4] = (TabT)pos;
            lds_mskor_tag(table, W.hx >> 7, W.tmask, W.tbits);
        };
        // (Blind: the bucket is not looked at first.  Where another position of the pair has taken the bucket since, that
        // entry is lost to the older one -- oracle/sim_encode2.c, "blind takeback": ratio 2.8839 -> 2.8836 / 1.8382 -> 1.8393 --
        // and the step's chain of dependent waits is one LDS round trip shorter.)
        auto lw_takeback = [&](const LW &W) {
            table[W.hx >> 4] = (TabT)W.oldp;
            lds_mskor_tag(table, W.hx >> 7, W.tmask, W.tagWord & W.tmask);
        };
        const uint32_t hc0 = (((uint32_t)lane + 1u) << 16) | (uint32_t)lane, hc1 = hc0 + ((64u << 16) | 64u);

        // Which shape the groups have is the wave's choice, block by block: groups of four measure 56 bytes behind a head in
        // one go, groups of two 24 -- half the requests, 32 heads a round (no second round), and a match that reaches the
        // 24 is extended by the whole wave.  Measured with one shape for everything: text 165 (four) against 187 GB/s (two),
        // lzsynth -- matches of 20 bytes on average -- 201 against 153.  A block starts with groups of four and counts, over
        // its first 32 windows, the selected matches of 24 bytes and more: fewer than one in sixteen windows -> groups of two
        // from there on.  With groups of two every extension is counted (it is the slow path anyway): more than one in
        // eight windows -> back to four, for the rest of the block.  The steady state of either shape carries no counting.
        int shape = (ENC_GROUPS == 2) ? 2 : 4;
        auto shape_update = [&]() {
            if (ENC_GROUPS != 0) return;
            if (shape == 4) {
                if (shapeCounting && shapeWin >= 32) {
                    shapeCounting = false;
                    selfRunOn = selfRunSeen >= ENC_SELF_RUN_MIN;
                    if (shapeLong * 16 < shapeWin) { shape = 2; shapeWin = 0; shapeExt = 0; }
                }
            } else if (shapeWin >= 64) {
                if (shapeExt * 8 > shapeWin) shape = 4;                 // (shapeCounting stays off: four for the rest of the block)
                shapeWin = 0; shapeExt = 0;
            }
        };
        // ---- one window per step: positions [p0, p0 + 64), p0 >= max(anchor, 8); returns where the next one starts.
        // The probed positions outside the selected matches go into the table afterwards (:998, the reference's policy).
        // the last request of a window ends 136 bytes behind its first position
        auto pipe_can_issue = [&](int p0) -> bool { return (missAcc >> 6) == 1u && p0 >= 8 && p0 + 136 <= n && pipeFits; };
        auto group_window = [&](auto G, const int p0) -> int {
            const uint32_t pos = (uint32_t)(p0 + lane), pos8 = pos - 8u;
            if (pfPos != p0) pfV8 = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + pos);
            LW W;
            if (ENC_PRIO1 && ENC_PRIO_P != ENC_PRIO_F) __builtin_amdgcn_s_setprio(ENC_PRIO_P);
            lw_probe(W, pos, pfV8, false);
            lw_heads(G, W, pos, pos8, 0u, 0ull, (uint32_t)pfV8, ~(uint32_t)pfV8);
            lw_loads(G, W, p0);
            if (ENC_PRIO1 && ENC_PRIO_P != ENC_PRIO_M) __builtin_amdgcn_s_setprio(ENC_PRIO_M);
            ENC_LAP(0);
            commit_pending();                                  // (the requests are out: the last window's moves are looked at now)
            lw_measure(G, W, p0);
            if (ENC_PRIO1 && ENC_PRIO_S != ENC_PRIO_M) __builtin_amdgcn_s_setprio(ENC_PRIO_S);
            lw_hits(G, W, p0, hc0, (uint32_t)lane + 8u, 0u);
            if (!W.hitm) {
                // nothing here: every position is registered, the miss counter widens the stride (:957-967)
                lw_insert(W, pos);
                if (ENC_PRIO1 && ENC_PRIO_F != ENC_PRIO_S) __builtin_amdgcn_s_setprio(ENC_PRIO_F);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                missAcc += LZ4_WAVE;
                pfPos = -1;
                return p0 + LZ4_WAVE;
            }
            int pEnd = anchor;
            const uint64_t selm = lw_select(W, p0, pEnd);
            if (ENC_PRIO1 && ENC_PRIO_F != ENC_PRIO_S) __builtin_amdgcn_s_setprio(ENC_PRIO_F);
            shapeWin += 1;
            if (shapeCounting) shapeLong += (int)__builtin_popcountll(selm & W.longm);
            ENC_LAP(1);
            // the next window starts at the end of the last match, or where this one ends: its bytes are requested now
            const int nextP = max(p0 + LZ4_WAVE, pEnd);
            pfPos = nextP;
            if (nextP + 72 <= n) pfV8 = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(nextP + lane));
            else pfPos = -1;
            ENC_LAP(2);
            if (qCnt + (int)__builtin_popcountll(selm) > LZ4_WAVE) flush_queue();
            const bool covered = lw_finish(W, pos, pos8, (uint32_t)lane + 1u, selm, anchor, 0);
            if (!covered) lw_insert(W, pos);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            ENC_LAP(3);
            anchor = pEnd;
            missAcc = miss0;
#ifdef ENC_STATS
            est[6] += 1;
#endif
            return nextP;
        };

        // ===================================================================================================
        // Two windows per step (positions [p0, p0 + 128)).  A window is a chain of dependent waits -- table, groups,
        // requests, lengths, selection -- and the chip has room for 16 such chains per CU (the table's LDS).  Here one
        // chain carries two windows, step by step side by side, so that each wait is paid once for both.  What the second
        // window needs from the first is the table: every position of a window is written into its bucket when it is
        // probed (the bucket's previous entry kept in a register), and positions that end up strictly inside a selected
        // match take that back, last window first, before the next pair starts -- between pairs the table is the
        // reference's (:998), within a pair the second window also sees the first one's covered positions
        // (oracle/sim_encode2.c).  The template parameter PAIR picks the form: blocks above 64 KiB, whose positions are
        // modular, and segments run one window per step.
        // ===================================================================================================
        uint64_t pfV8b = 0;                                  // the second window's bytes
        auto pair_can_issue = [&](int p0) -> bool { return (missAcc >> 6) == 1u && p0 >= 8 && p0 + 200 <= n && pipeFits; };
        // (the caller has the pair's bytes in pfV8 / pfV8b; go = the next pair can start at the returned position, and its
        // bytes are on their way)
        auto pair_window = [&](auto G, const int p0, bool &go) -> int {
            const int p1 = p0 + LZ4_WAVE;
            const uint32_t pos0 = (uint32_t)(p0 + lane), pos1 = pos0 + 64u;
            LW W0, W1;
            if (ENC_STAGE) {
                // (the reads of the pair before are behind us in program order, and a wave's LDS instructions run in order)
                ((uint32_t *)stage)[lane] = pfD;
                stCur = stNext;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const uint32_t d = (uint32_t)(p0 - stCur) + (uint32_t)lane;
                pfV8 = stage_u64(d);
                pfV8b = stage_u64(d + 64u);
            }
#ifdef ENC_STATS
            { uint32_t x = (uint32_t)pfV8 ^ (uint32_t)pfV8b; asm volatile("" : "+v"(x)); }      // the bytes have arrived
            ENC_LAP(4);
#endif
            if (ENC_PRIO_P != ENC_PRIO_F) __builtin_amdgcn_s_setprio(ENC_PRIO_P);
            lw_probe(W0, pos0, pfV8, true);
            lw_probe(W1, pos1, pfV8b, true);
#ifdef ENC_STATS
            { uint32_t x = W0.oldp ^ W1.oldp ^ W0.tagWord ^ W1.tagWord; asm volatile("" : "+v"(x)); }   // the table has answered
            ENC_LAP(5);
#endif
            lw_heads(G, W0, pos0, pos0 - 8u, 0u, 0ull, (uint32_t)pfV8, ~(uint32_t)pfV8);
            lw_heads(G, W1, pos1, pos1 - 8u, (uint32_t)__builtin_amdgcn_readlane((int)W0.off, 63), W0.candm >> 63,   // (lane 0 may continue the run of the last lane of the window before it)
                     (uint32_t)pfV8b, (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pfV8, 63));
            lw_loads(G, W0, p0);
            lw_loads(G, W1, p1);
            if (ENC_PRIO_P != ENC_PRIO_M) __builtin_amdgcn_s_setprio(ENC_PRIO_M);
            ENC_LAP(0);
            commit_pending();                                  // the pair before this one: its moves' results are looked at now
            int pEnd = anchor;
#ifdef ENC_EXP_EXTRA_SALU
            {   // experiment: ENC_EXP_EXTRA_SALU scalar instructions more per pair -- is the scalar unit a limit?
                int dummy = p0;
#pragma unroll
                for (int i = 0; i < ENC_EXP_EXTRA_SALU; i++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummy));
                asm volatile("" : : "s"(dummy));
            }
#endif
#ifdef ENC_EXP_EXTRA_VALU
            {
                int dummy = lane;
#pragma unroll
                for (int i = 0; i < ENC_EXP_EXTRA_VALU; i++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy));
                asm volatile("" : : "v"(dummy));
            }
#endif
            lw_measure(G, W0, p0);
            lw_measure(G, W1, p1);
            if (ENC_PRIO_S != ENC_PRIO_M) __builtin_amdgcn_s_setprio(ENC_PRIO_S);
            lw_hits(G, W0, p0, hc0, (uint32_t)lane + 8u, 0u);
            const uint64_t sel0 = lw_select(W0, p0, pEnd);
            const int pMid = pEnd;
            lw_hits(G, W1, p1, hc1, (uint32_t)lane + 72u, (uint32_t)__builtin_amdgcn_readlane((int)W0.hv, 63));
            const uint64_t sel1 = lw_select(W1, p1, pEnd);
            if (ENC_PRIO_F != ENC_PRIO_S) __builtin_amdgcn_s_setprio(ENC_PRIO_F);
            shapeWin += 2;
            if (shapeCounting) shapeLong += (int)__builtin_popcountll(sel0 & W0.longm) + (int)__builtin_popcountll(sel1 & W1.longm);
            ENC_LAP(1);
            // the next pair starts at the end of the last match, or where this one ends: its bytes are requested now
            const int nextP = max(p0 + 2 * LZ4_WAVE, pEnd);
            go = nextP + 200 <= n && (sel0 | sel1) != 0ull;
            pfPos = -1;                                        // (the windows that follow the last pair fetch their own bytes)
            if (go) {
                if (ENC_STAGE) stage_request(nextP);
                else {
                    pfV8 = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(nextP + lane));
                    pfV8b = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(nextP + LZ4_WAVE + lane));
                }
            }
            ENC_LAP(2);
            if (qCnt + (int)__builtin_popcountll(sel0) + (int)__builtin_popcountll(sel1) > LZ4_WAVE) flush_queue();
            const bool cov0 = lw_finish(W0, pos0, pos0 - 8u, (uint32_t)lane + 1u, sel0, anchor, 0);
            const bool cov1 = lw_finish(W1, pos1, pos1 - 8u, (uint32_t)lane + 65u, sel1, pMid, 1);
            // positions strictly inside a selected match take their insertion back if the bucket still holds it
            // (last window first: where the two met in one bucket the first window's older entry is the one to come back)
            if (cov1) lw_takeback(W1);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (cov0) lw_takeback(W0);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            ENC_LAP(3);
            if (sel0 | sel1) { anchor = pEnd; missAcc = miss0; }
            else missAcc += 2 * LZ4_WAVE;
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
                        if (ENC_STAGE) stage_request(np);
                        else {
                            if (pfPos != np) pfV8 = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(np + lane));
                            pfV8b = *(const LZ4_GLOBAL u64_unaligned *)(gsrc + (uint32_t)(np + LZ4_WAVE + lane));
                        }
                        bool go = true;
                        while (go) {
                            if (shape == 4) {
                                do { np = pair_window(EncGroups<4>{}, np, go); shape_update(); } while (go && shape == 4);
                            } else {
                                do { np = pair_window(EncGroups<2>{}, np, go); shape_update(); } while (go && shape == 2);
                            }
                        }
                    } else {
                        do {
                            if (shape == 4) {
                                do { np = group_window(EncGroups<4>{}, np); shape_update(); } while (shape == 4 && pipe_can_issue(np));
                            } else {
                                do { np = group_window(EncGroups<2>{}, np); shape_update(); } while (shape == 2 && pipe_can_issue(np));
                            }
                        } while (pipe_can_issue(np));
                    }
                    commit_pending();
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
                bool longRun = false;     // the lane's own counting stopped at ENC_LANE_CAP bytes with the match still running
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
                                // (a lane counts at most ENC_LANE_CAP bytes on its own, 32 a round trip: a match that is still
                                // running there -- a zero page took one lane 2048 round trips -- is finished by the whole wave,
                                // 1 KiB a step, if the selection takes it)
#ifndef ENC_LANE_CAP
#define ENC_LANE_CAP 256u
#endif
                                while (more && m + 32u <= maxLen && m < ENC_LANE_CAP) {               // 32 bytes a step
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
#ifndef ENC_NO_PIPE
                                if (more && m >= ENC_LANE_CAP) { longRun = true; more = false; }
#endif
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
                    const int headInfo = par_free_bperm((int)(myMl | (hback << 16) | ((uint32_t)longRun << 24)), myHead);   // 0 when the head missed
                    if (contin) {
                        const int m = (headInfo & 0xffff) - (lane - myHead);
                        hit = m >= LZ4_MINMATCH;
                        myMl = hit ? (uint32_t)m : 0u;
                        hback = ((uint32_t)headInfo >> 16) & 0xffu;
                        longRun = hit && ((headInfo >> 24) & 1);
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
                const uint64_t longm = __ballot(hit && longRun);
                for (uint64_t hm = hitm; hm;) {
                    const int k = (int)__builtin_ctzll(hm);
                    int len = (int)__builtin_amdgcn_readlane((int)myMl, k);
#ifndef ENC_NO_PIPE
                    if ((longm >> k) & 1ull) {
                        len += extend_long(p0 + k + len, __builtin_amdgcn_readlane((int)cand, k) + len);
                        myMl = (uint32_t)enc_writelane((int)myMl, len, k);
                    }
#endif
                    const int endk = p0 + k + len;
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
                    if (lane >= qCnt && lane < qCnt + k) {
                        if (SMALLQ) { q0 = r0 | (r1 << 16); q1 = r2 | (r3 << 16); }
                        else { q0 = r0; q1 = r1; q2 = r2; q3 = r3; }
                    }
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
            park_uniform(anchor, mpos, ml, mpos - cpos);
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
