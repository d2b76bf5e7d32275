// decode_par.hpp -- lane-parallel LZ4 block decoder: one wavefront per block,
// up to 64 sequences per iteration.
//
// Same contract and same results as decode_seq.hpp (reference
// LZ4_decompress_generic, cbits/lz4.c:1737-2165): this file only changes HOW the
// interior of a block is decoded.  A sequence-at-a-time decoder is bound by the
// serial token chain (cbits/lz4.c:1801-1854: sequence i+1 starts where sequence i
// ends) at ~30-50 issue slots and two dependent memory round trips per ~27-byte
// sequence.  Here one iteration handles a BATCH of sequences:
//
//   1. window     1 KiB of the compressed stream is staged in LDS (16 B / lane;
//                 the next window is prefetched into registers while this one is used).
//   2. speculate  every lane parses 8 candidate token positions (512 candidates)
//                 from registers: where would the next token be if one started here?
//   3. chain      pointer jumping over that successor table: after round k lanes
//                 0..2^k-1 hold the first 2^k real token positions.  Three rounds give
//                 jump^8 for every node and lanes 0..7; then lanes 8g..8g+7 follow the
//                 real chain from lanes 8(g-1)..8g-1 (one 8-lane gather per group), so
//                 sequence r ends up on lane r.
//   4. decode     lane r decodes sequence r (lengths, offset); a DPP wave scan of the
//                 output lengths gives every sequence its output position.
//   5. far        matches whose source was already flushed to global memory are
//                 fetched in one batched pass (no dependence on this batch).
//   6. literals   lane-per-sequence copy, LDS window -> LDS output ring.
//   7. matches    dependency rounds: a match is ready when every sequence its
//                 source overlaps is complete (64-bit ballot mask); ready lanes copy
//                 two chunks (2 x 16 / 8 / 4 bytes) per step inside the LDS ring, all
//                 reads of a step before its first write.
//   8. flush      completed output leaves the ring with aligned 16-byte stores.
//
// Only "plain interior" sequences are handled here: single-byte length
// extensions, offset != 0, source inside the block, far enough from both ends
// that the reference would still be in its fast loop and could not fail.  On
// anything else the block position is handed to decode_seq_run() for ONE
// sequence (or for the rest of the block near its end), which is also what
// produces the reference's exact error codes.
#pragma once

#include "decode_seq.hpp"

namespace lz4dev {

#define PAR_NODES 512       // speculative token candidates per window in the default form (8 per lane)
#ifndef PAR_ADAPT
#define PAR_ADAPT 1         // nodes per lane chosen per batch from the compressed bytes per sequence of the batch before: 6 or 8
#endif
#ifndef PAR_TAIL_ROUND
#define PAR_TAIL_ROUND 2    // > 0: after this many dependency rounds the matches still waiting are copied one by one, in order, by the
                            // whole wave.  Measured (lzsynth / text, GB/s): 0 = rounds only 958 / 627, 1: 889 / 585, 2: 992 / 634, 3: 961 / 614,
                            // 4: 943 / 609; "as soon as fewer than 8 lanes are ready" instead of a fixed count: 984 / 602
#endif
#ifndef PAR_ADAPT6
#define PAR_ADAPT6 416      // six nodes per lane while 64 sequences of the last batch's size need at most this many compressed bytes
#endif
#ifndef PAR_ADAPT10
#define PAR_ADAPT10 0       // ... or 10 (measured: the lzsynth stream gets 62 instead of 55 sequences into a batch and decodes no faster)
#endif
#define PAR_MAXNODES (PAR_ADAPT10 ? 640 : 512)
#ifndef PAR_SQ
#define PAR_SQ 3            // squaring rounds: the chain is then followed in groups of 2^PAR_SQ lanes
#endif
#define PAR_WIN 1024        // bytes of compressed stream staged per window (16 per lane)
#ifndef PAR_RING
#define PAR_RING 6144       // LDS output staging (8.3 KiB of LDS per wave in all: 19 waves per CU)
#endif
#ifndef PAR_HIST
#define PAR_HIST 2000       // bytes of history kept in the ring across a slide: two 16-byte chunks per lane (2048 needed a third pass for one or two chunks)
#endif
#ifndef PAR_BATCH_OUT
#define PAR_BATCH_OUT 2560  // max output bytes of one batch
#endif
#ifndef PAR_FAR_WIDE
#define PAR_FAR_WIDE 0      // far matches: 1 = one 16-byte request per lane instead of two requests per length class.  Measured
                            // (round 4): vector-memory reads -50 % (text -56 %), rate unchanged, but FETCH_SIZE +5 % on text (a
                            // 16-byte request crosses more 32-byte sectors than an 8-byte one): off
#endif
#ifndef PAR_RANK
#define PAR_RANK 1          // dependency masks from a bit vector of sequence starts (rank queries) instead of binary searches
#endif
#ifndef PAR_PRIO
#define PAR_PRIO 0x33     // s_setprio per phase, two bits each: chain (bits 0-1), decode + literals + need (2-3), match rounds (4-5), flush +
                          // window + speculation (6-7).  The phases that are chains of dependent LDS round trips go first among the 19
                          // waves of a CU.  Measured (lzsynth / text, GB/s; 0: 996-1000 / 634-638): 0x03 998 / 633, 0x30 1009 / 639,
                          // 0x33 1015 / 642-643, 0x3b 1010 / 642, 0x3f 1010 / 640, 0x27 1003 / 640, 0x36 1011 / 640
#endif
#ifndef PAR_WAVES
#define PAR_WAVES 5         // occupancy target (waves per SIMD) the register allocator is held to (<= 102 VGPRs)
#endif

// jump[] is the successor table of step 3.  Entries are BYTE offsets into jump[] itself (2 x node
// index), so a gather is one ds_read_u16 with no address arithmetic.  The absorbing state of a batch is the node behind
// the last one that may hold a token (`absorb` in the speculative parse), at most 2*PAR_NODES: the entry behind the table.
struct __attribute__((aligned(16))) ParLds {
    uint8_t win[PAR_WIN + 32];
    uint16_t jump[PAR_MAXNODES + 8];
    uint8_t ring[PAR_RING + 32];
};
#define PAR_END (2 * PAR_NODES)

typedef uint64_t par_u64u __attribute__((aligned(1)));
typedef uint32_t par_u32u __attribute__((aligned(1)));
typedef uint16_t par_u16u __attribute__((aligned(1)));

typedef uint32_t par_v4 __attribute__((ext_vector_type(4)));
typedef par_v4 par_v4u __attribute__((aligned(1)));      // one ds_read/write_b128 at any byte address

// LDS cost model on gfx950 (measured, scripts/micro/lds_unaligned.hip): a naturally aligned access costs
// ~6 cycles per wave-instruction; an access that is NOT naturally aligned costs ~1 cycle PER ACTIVE LANE,
// whatever its width (4, 8 or 16 bytes).  Hence: fields are fetched with aligned dword reads + a byte
// funnel (v_alignbyte), and byte-granular copies use the widest chunk that is safe.
__device__ __forceinline__ uint32_t lds_u32_any(const uint8_t *base4, uint32_t addr)
{
    const uint32_t *p = (const uint32_t *)(base4 + (addr & ~3u));
    return __builtin_amdgcn_alignbyte(p[1], p[0], addr & 3u);
}
__device__ __forceinline__ uint64_t lds_u64_any(const uint8_t *base4, uint32_t addr)
{
    const uint32_t *p = (const uint32_t *)(base4 + (addr & ~3u));
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    const uint32_t sh = addr & 3u;
    return ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32) | __builtin_amdgcn_alignbyte(d1, d0, sh);
}

__device__ __forceinline__ int par_bperm(int v, int srcLane)
{
    return __builtin_amdgcn_ds_bpermute(srcLane << 2, v);
}

// value of lane (l - d), for d in {1,2,4,8} and l, l-d in the same row of 16 lanes (else 0)
template <int D>
__device__ __forceinline__ int par_row_shr(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x110 | D, 0xf, 0xf, true);
}

// inclusive wave scan (sum) with DPP: 4 in-row steps + 2 row broadcasts
__device__ __forceinline__ int par_scan_incl(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return x;
}

// Diagnostic counters (STATS build only; never part of a timed run)
enum { PS_BATCHES, PS_SEQS, PS_ROUNDS, PS_MATCH_ITERS, PS_LIT_ITERS, PS_HANDOVERS, PS_SLIDES, PS_FULL, PS_FAR,
       PS_T_WINDOW, PS_T_SPEC, PS_T_CHAIN, PS_T_DECODE, PS_T_LIT, PS_T_NEED, PS_T_MATCH, PS_T_FLUSH, PS_T_SEQ, PS_COUNT };

// TOL: tolerant (deferred-copy) decode of a block of a linked stream without its dictionary, see TolCtx in
// decode_seq.hpp: matches that start before the block, or whose source touches a tainted granule, are
// recorded in tol->list instead of being copied.
// LIST: the token positions of a batch come from a list made by a separate pass (kernels.hip, k_walk_tokens: one LANE per
// block walks the token chain, which costs a fraction of finding 64 tokens at once by speculation) instead of steps 2-3.
// list[k] = compressed bytes of sequence k (0 = not known from here on), listLen entries.  The list is a HINT: every
// position taken from it is checked against the successor the lane in front computes from the token bytes themselves, so
// a wrong list costs time, never correctness; from the first disagreement on the block falls back to steps 2-3.
template <bool STATS, bool DICT, bool TOL = false, bool LIST = false>
__device__ int decode_block_par(const uint8_t *src, int srcLen, uint8_t *dst, int cap, const uint8_t *dict,
                                uint32_t dictLen, const uint8_t *bufLo, const uint8_t *bufHi, ParLds &L,
                                unsigned long long *stats, TolCtx *tol = nullptr, const uint8_t *list = nullptr,
                                int listLen = 0)
{
    int li = 0;                            // LIST: index of the sequence that starts at ip
    int nodesPerLane = 8;                  // PAR_ADAPT: token candidates per lane of the next batch (6, 8 or 10)
    uint32_t sc[PS_COUNT];                 // wave-uniform (kept in scalar registers)
    uint32_t tmark = 0;
    if (STATS) {
#pragma unroll
        for (int i = 0; i < PS_COUNT; i++) sc[i] = 0;
        tmark = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_amdgcn_s_memtime());
    }
    auto lap = [&](int which) {
        if (STATS) {
            const uint32_t now = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_amdgcn_s_memtime());
            sc[which] += now - tmark;
            tmark = now;
        }
    };
    auto publish = [&]() {
        if (STATS && lane_id() == 0) {
#pragma unroll
            for (int i = 0; i < PS_COUNT; i++) atomicAdd(&stats[i], (unsigned long long)sc[i]);
        }
    };
    if (cap < 128 || srcLen < 64) return decode_block_seq<TOL>(src, srcLen, dst, cap, dict, dictLen, bufLo, bufHi, tol);
    // external dictionary (linked streams, cbits/lz4.c:2347-2355): a match that lies ENTIRELY in the
    // previous block's output is a far match with another base pointer; one that straddles the seam
    // (:1883-1911) is left to the sequential decoder
    if (!DICT) { dict = nullptr; dictLen = 0; }
    const uint8_t *dictEnd = DICT ? dict + dictLen : dst;
    const int dictLo = DICT ? -(int)min(dictLen, 65535u) : 0;

    int lane = lane_id();
    const int iend = srcLen;
    const uint32_t A = (uint32_t)((uintptr_t)dst & 15);   // ring index of output position p is p - ringBase + A
    int ip = 0, op = 0;
    int ringBase = 0;       // multiple of 16
    int flushed = 0;        // output positions < flushed are in global memory
    SeqState st;

    // 16 bytes of the compressed stream for this lane's slot of the window that starts at `base`
    auto fetch_window = [&](uintptr_t base) -> uint4 {
        const uint8_t *q = (const uint8_t *)(base + 16u * (uint32_t)lane);
        if (q >= bufLo && q + 16 <= bufHi) {
            const par_v4 v = *as_global((const par_v4 *)q);
            return make_uint4(v.x, v.y, v.z, v.w);
        }
        uint32_t w[4] = {0, 0, 0, 0};
        for (int k = 0; k < 16; k++)
            if (q + k >= bufLo && q + k < bufHi) w[k >> 2] |= (uint32_t)as_global(q)[k] << (8 * (k & 3));
        return make_uint4(w[0], w[1], w[2], w[3]);
    };

    // ring -> global for positions [flushed, upto); 16-byte aligned stores in the body.
    // PAR_FLUSH: granularity of a flush's body in bytes.  The body is written with 16-byte stores; what it leaves at its
    // end is written by the NEXT flush, so a body that stops at a 16-byte boundary cuts a 32-byte sector of the L2 / HBM
    // write path in two every time (once per ~1.5 KB batch): PMC WRITE_SIZE 4.65e6 KB per 4.29 GB of output, +11 %
    // (round-2 review).  Stopping at 64-byte boundaries writes every sector once.
#ifndef PAR_FLUSH
#define PAR_FLUSH 64
#endif
    const uint32_t AF = (uint32_t)((uintptr_t)dst & (PAR_FLUSH - 1));
    auto flush = [&](int upto, bool final) {
        wave_fence();
        int f = flushed;
        const int mis = (int)((AF + (uint32_t)f) & (PAR_FLUSH - 1));
        if (mis) {
            const int head = PAR_FLUSH - mis;
            if (upto - f >= head) {
                if (lane < head) dst[f + lane] = L.ring[f - ringBase + (int)A + lane];
                f += head;
            } else if (!final) {
                return;
            }
        }
        if (((AF + (uint32_t)f) & (PAR_FLUSH - 1)) == 0) {
            const int n16 = final ? (upto - f) >> 4 : ((upto - f) / PAR_FLUSH) * (PAR_FLUSH / 16);
            for (int c = lane; c < n16; c += LZ4_WAVE) {
                const uint4 v = *(const uint4 *)&L.ring[f - ringBase + (int)A + 16 * c];
                *(uint4 *)(dst + f + 16 * c) = v;
            }
            f += n16 << 4;
        }
        if (final) {
            for (int x = f + lane; x < upto; x += LZ4_WAVE) dst[x] = L.ring[x - ringBase + (int)A];
            f = upto;
        }
        flushed = f;
        wave_fence();
    };

    uintptr_t wbase = (uintptr_t)src & ~(uintptr_t)15;     // window base whose data is in `wnext`
    uint4 wnext = fetch_window(wbase);
    bool winStale = true;                                  // L.win does not hold the window at wbase yet
    const uint8_t *jumpB = (const uint8_t *)L.jump;
    // (the absorbing state behind the table is written with every batch's table: its place depends on the nodes per lane)

    for (;;) {
        // ================= hot loop: batches of plain interior sequences =================
        while (iend - ip >= 64 && cap - op >= 128) {
            lap(PS_T_FLUSH);

            // ---------------- 1. window ----------------
            const uint8_t *gp = src + ip;
            const uintptr_t abase = (uintptr_t)gp & ~(uintptr_t)15;
            const int wofs = (int)((uintptr_t)gp - abase);
            const int ipW0 = ip - wofs;                    // block-relative position of window byte 0
            const int iendW = iend - ipW0;                 // block end in window coordinates
            const int inLim = min(iendW - 32, PAR_WIN);    // a plain sequence must end at or before this
            // L.win already holds this window (stored at the end of the previous batch) unless this is
            // the first batch or the one after a handover
            if (abase != wbase) { wbase = abase; wnext = fetch_window(abase); winStale = true; }
            if (winStale) {
                *(uint4 *)&L.win[16 * lane] = wnext;
                winStale = false;
                wave_fence();
            }
            lap(PS_T_WINDOW);

            uint32_t c2, absorb;
            bool fromList = false;
            if (LIST && li < listLen) {
                // ---------------- 2'. token positions from the list ----------------
                fromList = true;
                absorb = 2u * (uint32_t)max(inLim, 0);
                const int k = li + lane;
                const uint32_t d = (k < listLen) ? (uint32_t)as_global(list)[k] : 0u;
                const uint64_t zm = __ballot(d == 0u);                       // lanes up to the first unknown length have a position
                const int known = zm ? (int)__builtin_ctzll(zm) : LZ4_WAVE - 1;
                const int incl0 = par_scan_incl((int)d);
                c2 = (lane <= known) ? 2u * (uint32_t)(wofs + incl0 - (int)d) : absorb;
                c2 = min(c2, absorb);
            } else {
            // ---------------- 2. speculative parse (registers only) + 3a. squaring rounds ----------------
            // NL nodes per lane (NL * 64 byte positions of the window are token candidates).  The work of both steps is
            // proportional to NL -- NL gathers per squaring round -- while a batch takes at most 64 sequences: NL = 8 covers 64
            // sequences of 8 compressed bytes; text (5.9 bytes per sequence) fills its 64 lanes from 6 nodes per lane, the
            // lzsynth stream (9.2) needs 10 to get past 55 sequences per batch.  NL follows the batch before (below).
            auto parse_square = [&](auto NC) {
                constexpr int NL = decltype(NC)::value;
                static_assert(NL % 2 == 0 && NL * LZ4_WAVE <= PAR_MAXNODES, "pairs of nodes, table size");
                uint32_t J[NL];
                // Two nodes per instruction (16-bit halves, v_pk_*).  The absorbing state is the node behind the last one
                // that may be a token, nodeLim + 1: every successor beyond nodeLim is clamped to it, its own included (a
                // successor lies at least three bytes on), so no comparison is needed; for nodeLim = NL * 64 - 1 that is the
                // extra entry behind the table.
                absorb = 2u * ((uint32_t)min(inLim, NL * LZ4_WAVE - 1) + 1u);
                {
                    typedef unsigned short par_h2 __attribute__((ext_vector_type(2)));
                    uint32_t w0, w1, w2;                      // my NL bytes and the one behind them, from byte 0 of w0 on
                    if (NL == 8) {
                        const uint64_t lo = *(const uint64_t *)&L.win[8 * lane];
                        w0 = (uint32_t)lo; w1 = (uint32_t)(lo >> 32);
                        w2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x130, 0xf, 0xf, false);   // next lane's first bytes
                    } else {
                        const uint32_t a = (uint32_t)(NL * lane);
                        const uint32_t *q = (const uint32_t *)&L.win[a & ~3u];
                        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
                        const uint32_t sh = a & 3u;
                        w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
                        w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
                        w2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
                    }
                    auto h2 = [](uint32_t x) { par_h2 r; __builtin_memcpy(&r, &x, 4); return r; };
                    auto u32 = [](par_h2 x) { uint32_t r; __builtin_memcpy(&r, &x, 4); return r; };
                    const par_h2 one = {1, 1}, lim = h2((absorb >> 1) * 0x00010001u);
                    const uint32_t base3 = ((uint32_t)(NL * lane) + 3u) * 0x00010001u + 0x00010000u;   // nodes 2p, 2p + 1 of pair p: + 2p
                    uint32_t P[NL / 2];
#pragma unroll
                    for (int p = 0; p < NL / 2; p++) {
                        // tokens of the two nodes and the byte behind each, zero-extended into the halves
                        const uint32_t ws = (p < 2) ? w0 : ((p < 4) ? w1 : w2), wn = (p < 2) ? w1 : w2;
                        const par_h2 T = h2(__builtin_amdgcn_perm(0u, ws, (p & 1) ? 0x0c030c02u : 0x0c010c00u));
                        const par_h2 B = h2((p & 1) ? __builtin_amdgcn_perm(wn, ws, 0x0c040c03u) : __builtin_amdgcn_perm(0u, ws, 0x0c020c01u));
                        const par_h2 lit0 = T >> 4;
                        const par_h2 ext = (lit0 + one) >> 4;                       // 1 when the literal nibble is 15
                        const par_h2 term = ext * (B + one) + lit0;                 // 15 + 1 + b1, or the nibble
                        const par_h2 mlx = ((T & (par_h2){15, 15}) + one) >> 4;     // 1 when the match nibble is 15
                        const par_h2 nxt = term + mlx + h2(base3 + 0x00020002u * (uint32_t)p);
                        P[p] = u32(__builtin_elementwise_min(nxt, lim) << 1);
                    }
                    uint32_t *tw = (uint32_t *)&L.jump[NL * lane];
#pragma unroll
                    for (int p = 0; p < NL / 2; p++) tw[p] = P[p];
                    if (lane == 0) L.jump[NL * LZ4_WAVE] = (uint16_t)(2 * NL * LZ4_WAVE);   // absorbing state behind the table
#pragma unroll
                    for (int p = 0; p < NL / 2; p++) { J[2 * p] = P[p] & 0xffffu; J[2 * p + 1] = P[p] >> 16; }
                }
                wave_fence();
                lap(PS_T_SPEC);
#if PAR_PRIO
            if ((PAR_PRIO >> 0 & 3) != (PAR_PRIO >> 6 & 3)) __builtin_amdgcn_s_setprio(PAR_PRIO >> 0 & 3);
#endif

                // ---------------- 3. chain: sequence r -> lane r ----------------
                c2 = (lane == 0) ? 2u * (uint32_t)wofs : absorb;   // 2 x token position
                // Three squaring rounds give jump^8 for every node and the first 8 token positions (lanes 0..7);
                // after that only the real chain is followed: lanes 8g..8g+7 are jump^8 of lanes 8(g-1)..8g-1,
                // one 8-lane gather per group instead of two more squarings of all the nodes.
#pragma unroll
                for (int k = 0; k < PAR_SQ; k++) {
                    const int d = 1 << k;
                    const int cj = (int)*(const uint16_t *)(jumpB + c2);
#pragma unroll
                    for (int j = 0; j < NL; j++) J[j] = (uint32_t)*(const uint16_t *)(jumpB + J[j]);
                    int sh;
                    if (k == 0) sh = par_row_shr<1>(cj);
                    else if (k == 1) sh = par_row_shr<2>(cj);
                    else if (k == 2) sh = par_row_shr<4>(cj);
                    else sh = par_row_shr<8>(cj);
                    if (lane >= d && lane < 2 * d) c2 = (uint32_t)sh;
                    wave_fence();
                    uint32_t *tw = (uint32_t *)&L.jump[NL * lane];
#pragma unroll
                    for (int p = 0; p < NL / 2; p++) tw[p] = J[2 * p] | (J[2 * p + 1] << 16);
                    wave_fence();
                }
            };
#ifdef PAR_FORCE_NL
            parse_square(std::integral_constant<int, PAR_FORCE_NL>());     // experiments: one fixed form
#elif PAR_ADAPT
            if (nodesPerLane == 6) parse_square(std::integral_constant<int, 6>());
#if PAR_ADAPT10
            else if (nodesPerLane == 10) parse_square(std::integral_constant<int, 10>());
#endif
            else
                parse_square(std::integral_constant<int, 8>());
#else
                parse_square(std::integral_constant<int, 8>());
#endif
            {
                constexpr int G = 1 << PAR_SQ;
#pragma unroll
                for (int g = 1; g < LZ4_WAVE / G; g++) {
                    int cj = (int)absorb;
                    if (lane >= G * (g - 1) && lane < G * g) cj = (int)*(const uint16_t *)(jumpB + c2);
                    int sh;
                    // the next group sits G lanes up: a DPP row shift while that stays inside a row of 16 lanes
                    if (G == 1 && (g & 15)) sh = par_row_shr<1>(cj);
                    else if (G == 2 && (g & 7)) sh = par_row_shr<2>(cj);
                    else if (G == 4 && (g & 3)) sh = par_row_shr<4>(cj);
                    else if (G == 8 && (g & 1)) sh = par_row_shr<8>(cj);
#ifndef PAR_GROUP_BPERM
                    // ... and across rows with the gfx950 row swaps (vector ALU) instead of a trip through the LDS crossbar:
                    // v_permlane16_swap puts rows 0 / 2 of its source into rows 1 / 3, v_permlane32_swap rows 0-1 into rows
                    // 2-3 (scripts/micro/permlane_swap.hip prints both); a rotation by 8 inside the row then brings the upper
                    // half of the row below into my lanes
                    else if (G == 8 && (g == 2 || g == 6)) {
                        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)cj, (unsigned)cj, false, false);
                        sh = __builtin_amdgcn_update_dpp(0, (int)r[0], 0x128 /* row_ror:8 */, 0xf, 0xf, false);
                    } else if (G == 8 && g == 4) {
                        const auto q = __builtin_amdgcn_permlane32_swap((unsigned)cj, (unsigned)cj, false, false);
                        const auto r = __builtin_amdgcn_permlane16_swap(q[0], q[0], false, false);
                        sh = __builtin_amdgcn_update_dpp(0, (int)r[1], 0x128 /* row_ror:8 */, 0xf, 0xf, false);
                    }
#endif
                    else sh = par_bperm(cj, (lane - G) & 63);
                    if (lane >= G * g && lane < G * g + G) c2 = (uint32_t)sh;
                }
            }
            }
            lap(PS_T_CHAIN);
#if PAR_PRIO
            if ((PAR_PRIO >> 2 & 3) != (PAR_PRIO >> 0 & 3)) __builtin_amdgcn_s_setprio(PAR_PRIO >> 2 & 3);
#endif

            // ---------------- 4. decode own sequence, place it ----------------
            const bool has = c2 < absorb;
            const uint32_t cc = has ? (c2 >> 1) : 0u;
            const uint32_t tb = lds_u32_any(L.win, cc);                          // token, next byte
            const uint32_t t = tb & 0xffu, b1 = (tb >> 8) & 0xffu;
            const bool is15 = (t >> 4) == 15u;
            const uint32_t lit = is15 ? 15u + b1 : (t >> 4);
            const uint32_t litStart = cc + 1u + (is15 ? 1u : 0u);
            const uint32_t offPos = litStart + lit;                             // <= 511 + 272: inside the window
            const uint32_t ob = lds_u32_any(L.win, offPos);                      // offset (2 bytes), match-length byte
            const uint32_t off16 = ob & 0xffffu, b2 = (ob >> 16) & 0xffu;
            const bool mlx = (t & 15u) == 15u;
            const uint32_t ml = (t & 15u) + LZ4_MINMATCH + (mlx ? b2 : 0u);
            const uint32_t nxt = offPos + 2u + (mlx ? 1u : 0u);
            bool ok = has && !(is15 && b1 == 255u) && !(mlx && b2 == 255u) && (int)nxt <= inLim && off16 != 0;
            bool listBad = false;
            if (LIST && fromList) {
                // the list is a hint: my position must be where the sequence in front of me ends
                const uint32_t prevNxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nxt, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                listBad = has && lane > 0 && prevNxt != cc;
                ok = ok && !listBad;
            }
            const int len = ok ? (int)(lit + ml) : 0;
            const int incl = par_scan_incl(len);
            const int outEnd = op + incl;
            const int outStart = outEnd - len;
            const int dpos = outStart + (int)lit;            // match destination
            const int spos = dpos - (int)off16;              // match source
            ok = ok && incl <= PAR_BATCH_OUT && outEnd + 64 < cap &&
                 ((spos >= 0 && (spos >= ringBase || spos + (int)ml <= flushed)) ||
                  (DICT && spos >= dictLo && spos + (int)ml <= 0 && (!PAR_FAR_WIDE || ml >= 16u || spos + 16 <= 0)) ||
                  (TOL && spos < 0));
            const uint64_t okm = __ballot(ok);
            const int nseq = (~okm) ? (int)__builtin_ctzll(~okm) : LZ4_WAVE;
            if (LIST && fromList) {
                // a disagreement at the lane that ended the batch: the list is not this block's chain from here on
                if (nseq < LZ4_WAVE && ((__ballot(listBad) >> nseq) & 1ull)) listLen = 0;
                li += nseq;
            }
            lap(PS_T_DECODE);
            if (STATS) { sc[PS_BATCHES]++; sc[PS_SEQS] += (unsigned)nseq; if (nseq == LZ4_WAVE) sc[PS_FULL]++; }
            if (nseq == 0) break;                            // not a plain interior sequence: slow path below

            const bool act = lane < nseq;
            const int opNext = __builtin_amdgcn_readlane(outEnd, nseq - 1);
            const int ipNext = ipW0 + __builtin_amdgcn_readlane((int)nxt, nseq - 1);
            if (PAR_ADAPT && !fromList && nseq >= 16) {
                // nodes per lane of the next batch: enough window for 64 sequences of this batch's average size
                const int need = (ipNext - ip) * LZ4_WAVE;             // compressed bytes of 64 such sequences, times nseq
                nodesPerLane = (need <= PAR_ADAPT6 * nseq) ? 6 : ((need <= 520 * nseq || !PAR_ADAPT10) ? 8 : 10);
            } else if (PAR_ADAPT && !fromList && nodesPerLane == 6 && !((__ballot(has) >> nseq) & 1ull)) {
                // a short batch because the narrow window held no further token (not because a sequence was not plain): the
                // sequences have grown -- fewer than 16 fit 384 candidates -- and the rule above would never look again
                nodesPerLane = 8;
            }
            const uint32_t mdA = (uint32_t)(dpos - ringBase) + A;   // ring index of my match destination
            const bool ext = TOL && spos < 0;                       // source starts in the previous block: deferred
            const bool nearSrc = spos >= ringBase && !ext;
            bool deferred = false;                                  // TOL: this lane's match is recorded, not copied
            const bool w8 = ml >= 8 && off16 >= 8;                  // 8-byte steps are safe
            const bool grp = w8 && (off16 >= 32 || off16 >= ml);    // 32-byte groups never read their own writes
            const bool g16 = grp && ml >= 16;                       // ... as two 16-byte chunks
            const bool g8 = grp && ml < 16;                         // ... as two 8-byte chunks (8 <= ml < 16)
            const bool g4 = ml < 8 && off16 >= ml;                  // 4 <= ml < 8, no self-overlap: two 4-byte chunks
            const bool fastc = g16 || g8 || g4;

            // ---------------- far matches, first half: the loads.  Sources already in global memory (older than
            // this batch: under TOL their taint is final) are requested FIRST, then the next window: vector memory
            // returns in order, so the far data is not held up behind the window's trip to HBM, and both are in
            // flight while the masks and the literals are done.  (Requesting the window first made every batch with
            // a far match wait for HBM right here.)
            const uint64_t farm = __ballot(act && !nearSrc);
            if (TOL && farm) {
                if (act && !nearSrc && (ext || tol_tainted(tol, spos, spos + (int)ml))) {
                    deferred = true;
                    tol_taint(tol, dpos, dpos + (int)ml);
                }
                wave_fence();
            }
            const uint8_t *gsrc = (DICT && spos < 0) ? dictEnd + spos : dst + spos;
            const bool farMine = act && !nearSrc && !deferred;
            // chunks of 16, 8 or 4 bytes; the last chunk is re-anchored at the end so that nothing past the match is
            // written.  Far sources never overlap their destination.
            const uint32_t fstep = (ml >= 16) ? 16u : ((ml >= 8) ? 8u : 4u);
            const uint32_t flast = ml - fstep;
            // Each chunk class loads into registers of its own: classes that shared destination registers made
            // the compiler wait for one class's loads before it issued the next one's (three round trips, not one).
            par_v4 f0 = {0u, 0u, 0u, 0u}, f1 = {0u, 0u, 0u, 0u};
            uint64_t fa = 0, fb = 0;
            uint32_t fw0 = 0, fw1 = 0;
            if (farm) {
                if (STATS) sc[PS_FAR] += (unsigned)__builtin_popcountll(farm);
#if PAR_FAR_WIDE
                // ONE 16-byte request per lane whatever the length (a second one, re-anchored at the end, above 16 bytes);
                // the short classes' second chunk is cut out of it in registers like the near matches' below.  The request
                // never leaves the output buffer (cap - op >= 128), and a dictionary match shorter than 16 bytes is only
                // taken when 16 bytes are left in the dictionary (`ok` above).
                if (farMine) __builtin_memcpy(&f0, gsrc, 16);
                if (farMine && fstep == 16 && ml > 16u) __builtin_memcpy(&f1, gsrc + min(16u, flast), 16);
#else
                if (farMine && fstep == 16) {
                    __builtin_memcpy(&f0, gsrc, 16);
                    __builtin_memcpy(&f1, gsrc + min(16u, flast), 16);
                }
                if (farMine && fstep == 8) {
                    fa = *(const par_u64u *)(gsrc);
                    fb = *(const par_u64u *)(gsrc + flast);
                }
                if (farMine && fstep == 4) {
                    fw0 = *(const par_u32u *)(gsrc);
                    fw1 = *(const par_u32u *)(gsrc + flast);
                }
#endif
            }
            // prefetch the next window while this batch is copied
            {
                const uintptr_t nb = (uintptr_t)(src + ipNext) & ~(uintptr_t)15;
                wbase = nb;
                wnext = fetch_window(nb);
            }

            // ---------------- dependency masks (independent of the copies below: issued first so that
            // their cross-lane traffic overlaps the literal and far copies) ----------------
            uint64_t need = 0;
#if PAR_RANK
            {
                // Which sequences of this batch does my source [spos, min(spos+ml, outStart)) overlap?  Sequence starts are
                // bits of a bit vector over the batch's output positions (relative to op; <= PAR_BATCH_OUT of them); the
                // sequence that holds position x is (number of set bits at or below x) - 1.  The vector lives where the jump
                // table was (the chain is done with it): one 16-byte record per 64 positions = {bits 0-31, bits 32-63, set
                // bits in the records before}, so a query is ONE aligned 16-byte gather and two popcounts -- instead of a
                // six-step binary search over the lanes (12 ds_bpermute and their address arithmetic per batch).
                static_assert(16 * (PAR_BATCH_OUT / 64 + 1) <= 2 * PAR_NODES, "rank records fit the jump table");
                uint8_t *rk = (uint8_t *)L.jump;
                constexpr int NREC = PAR_BATCH_OUT / 64 + 1;
                {
                    uint32_t z = 0;
                    asm volatile("" : "+v"(z));     // made here: a zero vector kept across the loop costs four registers (it was spilled)
                    if (lane < NREC) *(uint4 *)&rk[16 * lane] = make_uint4(z, z, z, z);
                }
                wave_fence();
                if (act) {
                    const uint32_t rel = (uint32_t)(outStart - op);
                    atomicOr((uint32_t *)&rk[16u * (rel >> 6) + 4u * ((rel >> 5) & 1u)], 1u << (rel & 31u));
                }
                wave_fence();
                {
                    uint2 w = make_uint2(0u, 0u);
                    if (lane < NREC) w = *(const uint2 *)&rk[16 * lane];
                    const int cnt = (int)__builtin_popcount(w.x) + (int)__builtin_popcount(w.y);
                    const int pre = par_scan_incl(cnt) - cnt;
                    if (lane < NREC) *(uint32_t *)&rk[16 * lane + 8] = (uint32_t)pre;
                }
                wave_fence();
                const int srcHi = min(spos + (int)ml, outStart);        // bytes >= outStart are my own literals
                if (act && nearSrc && srcHi > op && srcHi > spos) {
                    const uint32_t xlo = (uint32_t)(max(spos, op) - op), xhi = (uint32_t)(srcHi - 1 - op);
                    auto rank = [&](uint32_t x) -> int {
                        const uint4 r = *(const uint4 *)&rk[16u * (x >> 6)];
                        const uint32_t m = (2u << (x & 31u)) - 1u;          // bits 0 .. x & 31
                        const bool hi = (x & 32u) != 0u;
                        return (int)r.z + (int)__builtin_popcount(r.x & (hi ? ~0u : m)) + (int)__builtin_popcount(r.y & (hi ? m : 0u)) - 1;
                    };
                    const int jlo = rank(xlo), jhi = rank(xhi);
                    const uint64_t upto = (jhi >= 63) ? ~0ull : ((1ull << (jhi + 1)) - 1ull);
                    need = upto & ~((1ull << jlo) - 1ull);
                    need &= ~(1ull << lane);
                }
                wave_fence();
            }
#else
            {
                // which sequences of this batch does my source [spos, min(spos+ml, outStart)) overlap?
                const int srcHi = min(spos + (int)ml, outStart);        // bytes >= outStart are my own literals
                const int xlo = max(spos, op), xhi = max(srcHi - 1, op);
                int jlo = 0, jhi = 0;
#pragma unroll
                for (int stp = 32; stp >= 1; stp >>= 1) {
                    const int c1 = jlo + stp, cb = jhi + stp;
                    const int v1 = par_bperm(outStart, c1 & 63), v2 = par_bperm(outStart, cb & 63);
                    if (c1 < nseq && v1 <= xlo) jlo = c1;
                    if (cb < nseq && v2 <= xhi) jhi = cb;
                }
                if (act && nearSrc && srcHi > op && srcHi > spos) {
                    const uint64_t upto = (jhi >= 63) ? ~0ull : ((1ull << (jhi + 1)) - 1ull);
                    need = upto & ~((1ull << jlo) - 1ull);
                    need &= ~(1ull << lane);
                }
            }
#endif
            // ---------------- 5. literals: window -> ring ----------------
            {
                const uint32_t sA = litStart;
                const uint32_t dA = (uint32_t)(outStart - ringBase) + A;
                const uint32_t n = act ? lit : 0u;
                if (n > 16) {                                   // rare (token nibble 15 with an extension byte)
                    const uint32_t last = n - 16;
                    for (uint32_t o = 0;; o += 16) {
                        const uint32_t oo = min(o, last);
                        *(par_v4u *)&L.ring[dA + oo] = *(const par_v4u *)&L.win[sA + oo];
                        if (o >= last) break;
                    }
                } else if (n > 8) {
                    // 9..16 literals in ONE round trip: a 16-byte chunk when its tail may fall into my own match
                    // area (written afterwards), else two 8-byte chunks, the second re-anchored at the end
                    if (n + ml >= 16) {
                        *(par_v4u *)&L.ring[dA] = *(const par_v4u *)&L.win[sA];
                    } else {
                        const uint64_t a = *(const par_u64u *)&L.win[sA], b = *(const par_u64u *)&L.win[sA + n - 8];
                        *(par_u64u *)&L.ring[dA] = a;
                        *(par_u64u *)&L.ring[dA + n - 8] = b;
                    }
                } else if (n > 0) {
                    const uint64_t v = lds_u64_any(L.win, sA);  // aligned reads + funnel
                    if (n + ml >= 8) {
                        // one 8-byte store; the bytes past the literals fall into my own match area, which is
                        // written afterwards (steps 6 and 7)
                        *(par_u64u *)&L.ring[dA] = v;
                    } else {
                        uint64_t w = v;
                        for (uint32_t q = 0; q < n; q++) { L.ring[dA + q] = (uint8_t)w; w >>= 8; }
                    }
                }
            }
            wave_fence();

            // ---------------- 6. far matches, second half: into the ring (behind the literals: a literal store
            // may run over into its own sequence's match area) ----------------
            if (farm) {
#if PAR_FAR_WIDE
                if (fstep == 16 && ml <= 16u) f1 = f0;
                {
                    const uint32_t sh = ml - 8u;                                     // 8-byte class: 0..7
                    const bool up = (sh & 4u) != 0u;
                    const uint32_t lo = up ? f0.y : f0.x, mid = up ? f0.z : f0.y, hi = up ? f0.w : f0.z;
                    fa = (uint64_t)f0.x | ((uint64_t)f0.y << 32);
                    fb = (uint64_t)__builtin_amdgcn_alignbyte(mid, lo, sh & 3u) | ((uint64_t)__builtin_amdgcn_alignbyte(hi, mid, sh & 3u) << 32);
                    fw0 = f0.x;
                    fw1 = __builtin_amdgcn_alignbyte(f0.y, f0.x, (ml - 4u) & 3u);   // 4-byte class: ml - 4 = 0..3
                }
#endif
                if (farMine && fstep == 16) {
                    *(par_v4u *)&L.ring[mdA] = f0;
                    *(par_v4u *)&L.ring[mdA + min(16u, flast)] = f1;
                }
                if (farMine && fstep == 8) {
                    *(par_u64u *)&L.ring[mdA] = fa;
                    *(par_u64u *)&L.ring[mdA + flast] = fb;
                }
                if (farMine && fstep == 4) {
                    *(par_u32u *)&L.ring[mdA] = fw0;
                    *(par_u32u *)&L.ring[mdA + flast] = fw1;
                }
                // matches longer than 32 bytes: 32 more per step, every load of a step issued before its first store
                for (uint32_t base = 32; __ballot(farMine && base < ml); base += 32) {
                    par_v4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
                    const bool on = farMine && base < ml;
                    const uint32_t o0 = min(base, flast), o1 = min(base + 16u, flast);
                    if (on) {
                        __builtin_memcpy(&v0, gsrc + o0, 16);
                        __builtin_memcpy(&v1, gsrc + o1, 16);
                    }
                    if (on) {
                        *(par_v4u *)&L.ring[mdA + o0] = v0;
                        *(par_v4u *)&L.ring[mdA + o1] = v1;
                    }
                }
            }

            // ---------------- 7. near matches: dependency rounds ----------------
            lap(PS_T_NEED);
#if PAR_PRIO
            if ((PAR_PRIO >> 4 & 3) != (PAR_PRIO >> 2 & 3)) __builtin_amdgcn_s_setprio(PAR_PRIO >> 4 & 3);
#endif
            need &= ~farm;                                              // far matches are already in place
            uint64_t done = ((nseq >= LZ4_WAVE) ? 0ull : (~0ull << nseq)) | farm;
            bool pending = act && nearSrc;                              // my match still has to be copied
            const uint32_t msA = nearSrc ? (uint32_t)(spos - ringBase) + A : 0u;
            int roundNo = 0;
            while (~done) {
#if PAR_TAIL_ROUND
                if (!TOL && roundNo >= PAR_TAIL_ROUND) {
                    // The matches that are still waiting after PAR_TAIL_ROUND rounds (dependency depth beyond it: a tenth of
                    // the sequences, one to three lanes per further round) are copied ONE AFTER THE OTHER in sequence order by
                    // the whole wave, a byte per lane: in that order every source is complete, so no round -- with its two
                    // reads, six stores and class tests for a lane or two -- is needed for them.
                    for (uint64_t sm = __ballot(pending); sm; sm &= sm - 1) {
                        if (STATS) sc[PS_MATCH_ITERS]++;
                        const int k = (int)__builtin_ctzll(sm);
                        const uint32_t kd = (uint32_t)__builtin_amdgcn_readlane((int)mdA, k);
                        const uint32_t ks = (uint32_t)__builtin_amdgcn_readlane((int)msA, k);
                        const uint32_t koff = (uint32_t)__builtin_amdgcn_readlane((int)off16, k);
                        const uint32_t kml = (uint32_t)__builtin_amdgcn_readlane((int)ml, k);
                        if (koff >= kml) {
                            if ((uint32_t)lane < kml) L.ring[kd + (uint32_t)lane] = L.ring[ks + (uint32_t)lane];
                            for (uint32_t j = (uint32_t)lane + LZ4_WAVE; j < kml; j += LZ4_WAVE) L.ring[kd + j] = L.ring[ks + j];
                        } else {
                            const float rcp = 1.0f / (float)koff;
                            for (uint32_t j = (uint32_t)lane; j < kml; j += LZ4_WAVE) {
                                const uint32_t q = (uint32_t)((float)j * rcp);
                                int rem = (int)j - (int)(q * koff);
                                if (rem < 0) rem += (int)koff; else if (rem >= (int)koff) rem -= (int)koff;
                                L.ring[kd + j] = L.ring[ks + (uint32_t)rem];
                            }
                        }
                        wave_fence();
                    }
                    pending = false;
                    break;
                }
                roundNo++;
#endif
                const bool ready = pending && ((need & ~done) == 0ull);
                bool mine = ready;
                if (TOL) {
                    // every sequence my source overlaps is complete, so its taint bits are final
                    const int srcHi = min(spos + (int)ml, dpos);        // bytes from dpos on are my own output
                    if (ready && srcHi > spos && tol_tainted(tol, spos, srcHi)) {
                        deferred = true;
                        mine = false;
                        tol_taint(tol, dpos, dpos + (int)ml);
                    }
                    wave_fence();
                }
                if (STATS) sc[PS_ROUNDS]++;
                // lanes whose chunks never read their own writes: two chunks per step (2 x 16, 2 x 8 or 2 x 4
                // bytes by class), every lane's reads issued before the first write, so that a round costs
                // one LDS round trip per 32 bytes whatever mix of classes is ready
                for (uint32_t base = 0; __ballot(mine && fastc && base < ml); base += 32) {
                    if (STATS) sc[PS_MATCH_ITERS]++;
                    const bool on = mine && fastc && base < ml;
                    const uint32_t cs = g16 ? 16u : (g8 ? 8u : 4u);
                    const uint32_t lastc = ml - cs;
                    const uint32_t o0 = g16 ? min(base, lastc) : 0u;
                    const uint32_t o1 = g16 ? min(base + 16u, lastc) : lastc;
                    // one 16-byte read serves every class (the bytes past a short match are read and dropped); the second
                    // chunk of the short classes -- the match's last 8 / 4 bytes -- is cut out of it in registers
                    par_v4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
                    uint64_t na = 0, nc = 0;
                    uint32_t nw0 = 0, nw1 = 0;
                    if (on) v0 = *(const par_v4u *)&L.ring[msA + o0];
                    if (on && g16 && ml > 16u) v1 = *(const par_v4u *)&L.ring[msA + o1];
                    if (g16 && ml <= 16u) v1 = v0;                                   // (ml == 16: both chunks are the same 16 bytes)
                    {
                        const uint32_t sh = ml - 8u;                                     // g8: 0..7
                        const bool up = (sh & 4u) != 0u;
                        const uint32_t lo = up ? v0.y : v0.x, mid = up ? v0.z : v0.y, hi = up ? v0.w : v0.z;
                        na = (uint64_t)v0.x | ((uint64_t)v0.y << 32);
                        nc = (uint64_t)__builtin_amdgcn_alignbyte(mid, lo, sh & 3u) | ((uint64_t)__builtin_amdgcn_alignbyte(hi, mid, sh & 3u) << 32);
                        nw0 = v0.x;
                        nw1 = __builtin_amdgcn_alignbyte(v0.y, v0.x, (ml - 4u) & 3u);   // g4: ml - 4 = 0..3
                    }
                    if (on && g16) {
                        *(par_v4u *)&L.ring[mdA + o0] = v0;
                        *(par_v4u *)&L.ring[mdA + o1] = v1;
                    }
                    if (on && g8) {
                        *(par_u64u *)&L.ring[mdA] = na;
                        *(par_u64u *)&L.ring[mdA + o1] = nc;
                    }
                    if (on && g4) {
                        *(par_u32u *)&L.ring[mdA] = nw0;
                        *(par_u32u *)&L.ring[mdA + o1] = nw1;
                    }
                    wave_fence();
                }
                // self-overlapping matches (offset < length; rare): the output is the `offset` bytes in front of the
                // destination repeated, so the whole wave writes it at once -- byte j is source byte j mod offset --
                // one match after the other (one LDS round trip each instead of length / step of them)
                for (uint64_t sm = __ballot(mine && !fastc); sm; sm &= sm - 1) {
                    if (STATS) sc[PS_MATCH_ITERS]++;
                    const int k = (int)__builtin_ctzll(sm);
                    const uint32_t kd = (uint32_t)__builtin_amdgcn_readlane((int)mdA, k);
                    const uint32_t ks = (uint32_t)__builtin_amdgcn_readlane((int)msA, k);
                    const uint32_t koff = (uint32_t)__builtin_amdgcn_readlane((int)off16, k);
                    const uint32_t kml = (uint32_t)__builtin_amdgcn_readlane((int)ml, k);
                    const float rcp = 1.0f / (float)koff;              // j < 512, koff < 32: the quotient is off by at most one
                    for (uint32_t j = (uint32_t)lane; j < kml; j += LZ4_WAVE) {
                        const uint32_t q = (uint32_t)((float)j * rcp);
                        int rem = (int)j - (int)(q * koff);
                        if (rem < 0) rem += (int)koff; else if (rem >= (int)koff) rem -= (int)koff;
                        L.ring[kd + j] = L.ring[ks + (uint32_t)rem];
                    }
                    wave_fence();
                }
                pending = pending && !ready;
                done |= __ballot(ready);
            }
            wave_fence();
            if (TOL) {
                // the batch's deferred matches join the list in stream order (lane order)
                const uint64_t dm = __ballot(deferred);
                if (dm) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
                    const uint32_t base = tol->count;
                    if (deferred && base + rank < tol->cap) tol->list[base + rank] = tol_entry(dpos, spos, ml);
                    wave_fence();
                    if (lane == 0) tol->count = base + (uint32_t)__builtin_popcountll(dm);
                    wave_fence();
                }
            }
            lap(PS_T_MATCH);
#if PAR_PRIO
            if ((PAR_PRIO >> 6 & 3) != (PAR_PRIO >> 4 & 3)) __builtin_amdgcn_s_setprio(PAR_PRIO >> 6 & 3);
#endif

            // ---------------- advance, 8. flush, slide ----------------
            op = opNext;
            ip = ipNext;
            // the next window goes to LDS now (nobody reads the old one any more): this is where the wave
            // waits for the prefetch, BEFORE the flush issues its stores, so that no later wait for a load
            // also has to wait for those stores to be acknowledged
            *(uint4 *)&L.win[16 * lane] = wnext;
            wave_fence();
            flush(op, false);
            if (op - ringBase + (int)A + PAR_BATCH_OUT + 32 > PAR_RING) {
                if (STATS) sc[PS_SLIDES]++;
                const int newBase = (op - PAR_HIST) & ~15;
                const int delta = newBase - ringBase;
                const int n16 = (op - newBase + (int)A + 15) >> 4;
                // history + the alignment head: at most PAR_HIST + 15 + 15 + 15 bytes in chunks of 16, PAR_SLIDE chunks per lane;
                // all of them are read before the first is written (one LDS round trip, and the ranges may overlap)
                constexpr int PAR_SLIDE = (PAR_HIST + 45 + 16 * LZ4_WAVE - 1) / (16 * LZ4_WAVE);
                {
                    uint4 v[PAR_SLIDE];
#pragma unroll
                    for (int i = 0; i < PAR_SLIDE; i++) {
                        const int k = lane + i * LZ4_WAVE;
                        v[i] = (k < n16) ? *(const uint4 *)&L.ring[delta + 16 * k] : make_uint4(0u, 0u, 0u, 0u);
                    }
                    wave_fence();
#pragma unroll
                    for (int i = 0; i < PAR_SLIDE; i++) {
                        const int k = lane + i * LZ4_WAVE;
                        if (k < n16) *(uint4 *)&L.ring[16 * k] = v[i];
                    }
                }
                ringBase = newBase;
                wave_fence();
            }
        }

        // ================= slow path: the sequential decoder =================
        // Either a sequence the hot loop does not take (one sequence, then back), or the tail of the
        // block (and with it every end-of-block rule and every error code).
        flush(op, true);
        st.ip = ip; st.op = op; st.fast = true;
        const bool tail = (iend - ip < 64 || cap - op < 128);
        if (STATS && !tail) sc[PS_HANDOVERS]++;
        lap(PS_T_FLUSH);
        int r = decode_seq_dispatch<TOL>(st, tail ? 0 : 1, src, srcLen, dst, cap, dict, dictLen, bufLo, bufHi, tol);
        if (r == SEQ_CONTINUE && (!st.fast || iend - st.ip < 64 || cap - st.op < 128))
            r = decode_seq_dispatch<TOL>(st, 0, src, srcLen, dst, cap, dict, dictLen, bufLo, bufHi, tol);
        lap(PS_T_SEQ);
        r = uni(r);
        if (r != SEQ_CONTINUE) { publish(); return r; }
        ip = uni(st.ip); op = uni(st.op);                 // read back through memory: tell the compiler they are uniform
        if (LIST) li += 1;                                // (one sequence went through the sequential decoder)
        // Nothing lane-private has to survive the call: the lane id is re-read (an opaque definition, so that
        // the values derived from it are rebuilt instead of being kept in registers across the call) and the
        // window is fetched again.  decode_seq_run clobbers v0..v79; every VGPR that lives across the call
        // sits above that and used to push the kernel to 124 VGPRs = 4 waves per SIMD.
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        wbase = (uintptr_t)(src + ip) & ~(uintptr_t)15;
        wnext = fetch_window(wbase);
        winStale = true;
        // reload the ring's history from global memory
        wave_fence();
        ringBase = (op > PAR_HIST) ? ((op - PAR_HIST) & ~15) : 0;
        flushed = op;
        for (int x = ringBase + lane; x < op; x += LZ4_WAVE) L.ring[x - ringBase + (int)A] = dst[x];
        wave_fence();
        lap(PS_T_SEQ);
    }
}

} // namespace lz4dev
