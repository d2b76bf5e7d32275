// decode_cu.hpp -- one WORKGROUP (16 wavefronts = one CU) per LZ4 block: the decoder of calls that do not fill the GPU.
//
// Same contract and same results as decode_par.hpp / decode_seq.hpp (reference LZ4_decompress_generic,
// cbits/lz4.c:1737-2165, with or without an external dictionary, :2347-2355).  decode_par.hpp gives a block to ONE
// wavefront, which is what fills the chip when a call brings tens of thousands of blocks; but a wave on its own issues one
// instruction in four to eight cycles and decodes 0.2-0.3 GB/s, so a call of 160 blocks -- the reference's own benchmark
// protocol, benchmark/Main.hs:80-84 -- took one block's latency on a GPU that was 97 % idle, and a 4 MiB block
// (Config.hs:109-116) took 6 ms.  Here a block is decoded in SEGMENTS of at most 22 KiB of compressed bytes and 32 KiB of
// output; a segment's output lives in LDS and sixteen waves work on it:
//
//   1. stage      the segment's compressed bytes into LDS.
//   2. parse      EXACT and parallel.  succ(p) = where the next token is if a token starts at byte p
//                 (cbits/lz4.c:1801-1854).  Every lane takes a 16-byte chunk and computes, for every byte p of it,
//                 T[p] = the first position at or behind the chunk's end that the chain from p reaches (backwards, so
//                 T[p] = T[succ(p)] inside the chunk).  A chain can only ENTER a 256-byte super-chunk at a position some
//                 T[p] names (a bit vector; 7 to 9 positions per super-chunk).  Each of those hops T[] through its
//                 super-chunk (<= 16 hops); where they LEAVE it are the candidates (one or two per super-chunk: chains
//                 that ran for a while have met).  The candidates form a linked list of <= 384 nodes which one wave
//                 ranks by pointer doubling; the nodes reachable from the segment's first token are the true entries.
//                 They hop once more to give every chunk its entry, every chunk's lane walks its few sequences, a scan
//                 gives each sequence its index and output position.  Nothing is guessed.
//   3. literals   every sequence's literals go from the compressed bytes to the output in LDS, and so does what a match
//                 takes from in front of the segment (earlier segments' output or the dictionary, final in global
//                 memory): no dependences.
//   4. matches    BYTE-GRANULAR POINTER JUMPING.  Every output byte gets a 16-bit source pointer: itself if it is in place
//                 (step 3), else the byte its match copies (op - offset + k, cbits/lz4.c:1866-1924).  ptr[x] = ptr[ptr[x]]
//                 for every byte at once, in place, halves every chain per round: the dependence graph of a segment -- 80
//                 to 300 matches deep, 32 768 for a run of one byte -- is resolved in 8 to 10 rounds (16) with every lane
//                 of sixteen waves busy, whatever it looks like; then every byte is fetched from the byte its pointer
//                 arrived at.  (The first form walked the graph match by match through done bits: 0.9 us per level.)
//   5. flush      LDS -> global memory with 16-byte stores.
//   6. next       the first sequence that is not plain ends the segment.  If it is the output or table limit that ended it,
//                 the next segment starts there; a sequence the parse does not take (a length with more than two extension bytes, an offset
//                 of 0, a match that straddles the dictionary's end) is decoded by wave 0 with the sequential decoder and the
//                 next segment starts behind it; the block's last 512 bytes (every end-of-block rule and every error code of
//                 the reference) are decoded by the sequential decoder.
//
// Only plain interior sequences are taken by steps 2-5, as in decode_par.hpp.  A block that reports an error -- and anything
// that goes wrong inside the form -- is marked CU_REDO and decoded again from scratch by the lane-parallel decoder
// (k_decode_par_redo, the launch behind this one), which is what yields the reference's exact negative codes.  (This kernel
// calls nothing out of line that another kernel calls: decode_seq_run, the one function the lane-parallel kernel calls, takes
// its register budget from ALL its callers, and a 1024-thread caller cost that kernel a wave per SIMD.)
#pragma once

#include "decode_par.hpp"

namespace lz4dev {

#define CU_THREADS 1024
#define CU_WAVES 16
#define CU_CHUNK 16            // bytes of compressed stream per lane of the parse (32 with 64 KiB segments: a segment of half the size
                               // left half the lanes idle in the parse's two per-lane phases, which cost the same per lane)
#define CU_CHUNK_LOG 4
#define CU_SUPER 256           // bytes per super-chunk (16 chunks)
#define CU_SUPER_WORDS (CU_SUPER / 32)   // words of a candidate bit vector per super-chunk
#define CU_CAND0 16            // first-round candidates per super-chunk (every position a chunk's T[] names)
#define CU_CAND 4              // candidate entries kept per super-chunk
#define CU_NODES 384           // list nodes: super-chunks x candidates (<= 84 x 4), the last one is the list's end
#define CU_LEVELS 7            // pointer doubling: the list has at most one node per super-chunk, 2^7 > 84
#define CU_CMAX 22528          // compressed bytes of a segment: 1408 chunks (enough for 32 KiB of output down to a ratio of 1.45)
#define CU_CBIG 16384          // ... of a big block's segments: 1024 chunks
#define CU_OUTMAX 32768        // output bytes of a segment (a 16-bit source pointer per byte: 64 KiB of LDS)
#define CU_NMAX 4096           // sequences of a segment
#define CU_STOP 0xffffu        // T[]: the chain from here meets a sequence that is not plain before it leaves the chunk
#define CU_NONE 0xffffu
#define CU_REDO ((int)0x80000001)   // result of a block the form leaves to the lane-parallel decoder
#define CU_IDLE_LIMIT 2000000u   // polls without progress after which a wave gives the block up (never reached: see there)
#define CU_TAILMAX 512u         // compressed bytes left to the sequential decoder at the block's end
#ifndef CU_MLP
// quads of output bytes a thread of the match phase has in flight: the LDS round trips of a batch overlap.  Measured (160 blocks of
// 64 KiB, us: fill | rounds | gather): fill 1 / 2 / 4 quads: 4.0 / 5.6 / 6.9 (bound by vector issue: the batches' registers spill);
// rounds and gather 1 / 2 / 4 / 8: 14.2 | 1.84, 11.5 | 1.88, 12.2 | 1.96, worse
#define CU_MLPF 1u
#define CU_MLP 2u
#endif
#define CU_SWEEPS 64               // sweeps of the pointer jumping after which a thread gives the block up (to the lane-parallel decoder)
#define CU_SLOTS 5u               // sequences of a chunk the first walk keeps for the second (a chunk of 16 bytes has at most six)
#define CU_MINSEG 512u          // ... and a segment shorter than this is not worth the parse
// LDS map (bytes).  [0, 33 KiB): the segment's output (during the parse: the compressed bytes, up to 47 KiB of them with their
// padding).  [33, 97 KiB): a 16-bit source pointer per output byte (during the parse: part of T[], two bytes per compressed
// byte behind the compressed bytes, dead once every chunk has its entry).  [97, 129 KiB): sequence records, written by the
// chunks' second walk.  [136, 160 KiB): the parse's small tables, then the rank records.
// A SIMD issues one wave-instruction in four cycles and an LDS round trip is ~100 ns, so the phases are written for few
// instructions and few dependent round trips, and both big tables of the parse are laid out for the access a whole wave makes
// at once -- lane L working on chunk L:
//   * the compressed bytes are staged with 4 bytes of padding behind every 16 (chunk stride 5 dwords: lanes that read
//     "their" byte k hit 32 different banks; unpadded, chunk stride 4 dwords, they hit 8);
//   * T[] is transposed: the entry of byte p lies at (p % 16) * chunks + p / 16, so the lanes' k-th entries are neighbours.
#define CU_OFF_PTR 33792
#define CU_OFF_REC 99328
#define CU_OFF_TAB 139264
#define CU_LDS_BYTES 163840
// ... tables of the parse
#define CU_TAB_ENTRY 0         // u16[1408]   entry of every chunk
#define CU_TAB_CBITS 2816      // u32[1408]   first-round candidate bits, one word per chunk;  then u16[7][384] doubling levels
#define CU_TAB_CBITS1 8448     // u32[1408]   second-round candidate bits
#define CU_TAB_CPOS0 14080     // u16[1408]   first-round candidates, 16 per super-chunk
#define CU_TAB_CF0 16896       // u16[1408]   ... where each leaves its super-chunk
#define CU_TAB_CPOS 19712      // u16[384]    candidates
#define CU_TAB_CF 20480        // u16[384]    ... where each leaves its super-chunk
#define CU_TAB_MARK 21248      // u8[384]
#define CU_TAB_MISC 21632      // u32[32]     see CM_*
#define CU_TAB_SCAN 21760      // u32[64]     workgroup scans
#define CU_TAB_J CU_TAB_CBITS
// ... after the parse
#define CU_TAB_RANK 0          // uint4[513]
enum { CM_OVERFLOW, CM_NPAR, CM_TAIL_IP, CM_TAIL_OP, CM_TAIL_KIND, CM_ABORT, CM_RESULT, CM_NEXT_IP, CM_NEXT_OP, CM_DIFF, CM_COUNT };

// byte p of the staged segment lives at LDS offset cu_at(p)
__device__ __forceinline__ uint32_t cu_at(uint32_t p) { return p + ((p >> CU_CHUNK_LOG) << 2); }
// four bytes from byte p on (two aligned dwords + a funnel; the dwords may lie either side of a chunk's padding)
__device__ __forceinline__ uint32_t cu_u32(const uint8_t *comp, uint32_t p)
{
    const uint32_t a = p & ~3u;
    const uint32_t d0 = *(const uint32_t *)(comp + cu_at(a)), d1 = *(const uint32_t *)(comp + cu_at(a + 4u));
    return __builtin_amdgcn_alignbyte(d1, d0, p & 3u);
}

struct CuSeq { uint32_t litStart, lit, off, ml, nxt; bool odd; };   // nxt == CU_STOP: not a plain interior sequence; odd: and not for want of staged bytes

// The sequence whose token is byte p of the segment (comp = the staged bytes in LDS), p < inLim - 2.  Plain = the whole
// sequence ends at or before inLim = staged bytes - 32 (at the block's end the reference's fast loop can then neither fail on
// input nor change loops, cbits/lz4.c:1809-1831, :1854-1863), offset != 0, and neither length has more than TWO extension
// bytes (cbits/lz4.c:1707-1729: literal runs below 525, matches below 529 -- straight-line code, no loop; what is longer is
// a few big copies, which the sequential decoder makes with the whole wave).  Every byte read lies at or before inLim + 3.
// cu_lit() is the literal part from the token and the two bytes behind it alone: the parse computes the successor of EVERY
// byte of the segment from registers, with one LDS read (the match length's extension byte) where the token asks for it.
__device__ __forceinline__ void cu_lit(uint32_t t, uint32_t b1, uint32_t b1b, uint32_t &lit, uint32_t &litBytes, bool &bad)
{
    const uint32_t l0 = t >> 4;
    const bool l15 = l0 == 15u, e1 = l15 && b1 == 255u;
    lit = l15 ? 15u + b1 + (e1 ? b1b : 0u) : l0;
    litBytes = l15 ? (e1 ? 2u : 1u) : 0u;
    bad = e1 && b1b == 255u;
}
__device__ __forceinline__ CuSeq cu_parse(const uint8_t *comp, uint32_t p, uint32_t inLim)
{
    CuSeq s;
    const uint32_t tb = cu_u32(comp, p);
    const uint32_t t = tb & 0xffu;
    uint32_t litBytes;
    bool bad;
    cu_lit(t, (tb >> 8) & 0xffu, (tb >> 16) & 0xffu, s.lit, litBytes, bad);
    s.litStart = p + 1u + litBytes;
    const uint32_t offPos = s.litStart + s.lit;
    const uint32_t ob = cu_u32(comp, min(offPos, inLim));
    s.off = ob & 0xffffu;
    const bool mlx = (t & 15u) == 15u;
    const uint32_t b2 = (ob >> 16) & 0xffu, b3 = ob >> 24;
    const bool e2 = mlx && b2 == 255u;
    s.ml = (t & 15u) + LZ4_MINMATCH + (mlx ? b2 : 0u) + (e2 ? b3 : 0u);
    const uint32_t nxt = offPos + 2u + (mlx ? 1u : 0u) + (e2 ? 1u : 0u);
    s.odd = nxt <= inLim && (bad || (e2 && b3 == 255u) || s.off == 0u);
    s.nxt = (nxt <= inLim && !s.odd) ? nxt : CU_STOP;
    return s;
}

// exclusive prefix sum over the workgroup's threads (v in thread order); *total = the sum.  tmp: 16 words of LDS.
__device__ __forceinline__ uint32_t cu_scan_excl(uint32_t v, uint32_t *tmp, uint32_t *total)
{
    const int lane = lane_id(), wave = uni((int)(threadIdx.x >> 6));
    const uint32_t incl = (uint32_t)par_scan_incl((int)v);
    if (lane == LZ4_WAVE - 1) tmp[wave] = incl;
    __syncthreads();
    uint32_t w = (lane < CU_WAVES) ? tmp[lane] : 0u;
    const uint32_t wi = (uint32_t)par_scan_incl((int)w);
    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)(wi - w), wave);
    *total = (uint32_t)__builtin_amdgcn_readlane((int)wi, CU_WAVES - 1);
    __syncthreads();
    return base + incl - v;
}

// two such sums at once (one pair of barriers); tmp: 32 words of LDS
__device__ __forceinline__ void cu_scan_excl2(uint32_t va, uint32_t vb, uint32_t *tmp, uint32_t *pa, uint32_t *pb, uint32_t *ta, uint32_t *tb)
{
    const int lane = lane_id(), wave = uni((int)(threadIdx.x >> 6));
    const uint32_t ia = (uint32_t)par_scan_incl((int)va), ib = (uint32_t)par_scan_incl((int)vb);
    if (lane == LZ4_WAVE - 1) { tmp[wave] = ia; tmp[16 + wave] = ib; }
    __syncthreads();
    uint32_t wa = (lane < CU_WAVES) ? tmp[lane] : 0u, wb = (lane < CU_WAVES) ? tmp[16 + lane] : 0u;
    const uint32_t sa = (uint32_t)par_scan_incl((int)wa), sb = (uint32_t)par_scan_incl((int)wb);
    *pa = (uint32_t)__builtin_amdgcn_readlane((int)(sa - wa), wave) + ia - va;
    *pb = (uint32_t)__builtin_amdgcn_readlane((int)(sb - wb), wave) + ib - vb;
    *ta = (uint32_t)__builtin_amdgcn_readlane((int)sa, CU_WAVES - 1);
    *tb = (uint32_t)__builtin_amdgcn_readlane((int)sb, CU_WAVES - 1);
    __syncthreads();
}

// n bytes (n >= 1) from global memory to LDS by one lane, 16 at a time (the last piece re-anchored at the end)
__device__ __forceinline__ void cu_lane_fetch(uint8_t *out, uint32_t dA, const LZ4_GLOBAL uint8_t *g, uint32_t n)
{
    if (n >= 16u) {
        const uint32_t last = n - 16u;
        for (uint32_t o = 0;; o += 16u) {
            const uint32_t oo = min(o, last);
            *(par_v4u *)&out[dA + oo] = *(const LZ4_GLOBAL par_v4u *)(g + oo);
            if (o >= last) break;
        }
    } else {
        for (uint32_t o = 0; o < n; o++) out[dA + o] = g[o];
    }
}

// One match by the whole wave (uniform arguments): the ones that overlap their own output (offset < length: the output is
// the `off` bytes in front of the destination, repeated), and whatever else the lanes' own copy does not take.
__device__ __forceinline__ void cu_wave_copy(uint8_t *out, uint32_t d, uint32_t s, uint32_t off, uint32_t ml)
{
    const uint32_t lane = (uint32_t)lane_id();
    if (off >= ml && ml >= 16u) {
        const uint32_t lastc = ml - 16u;
        for (uint32_t o = 16u * lane; o < ml; o += 16u * LZ4_WAVE) {
            const uint32_t oo = min(o, lastc);
            const par_v4 v = *(const par_v4u *)&out[s + oo];
            *(par_v4u *)&out[d + oo] = v;
        }
    } else if (off >= LZ4_WAVE) {
        // (a step of 64 bytes reads what earlier steps wrote: LDS operations of one wave stay in order)
        for (uint32_t c = 0; c < ml; c += LZ4_WAVE) {
            const uint32_t j = c + lane;
            if (j < ml) out[d + j] = out[s + j];
            wave_fence();
        }
    } else {
        uint32_t idx = lane % off;
        const uint32_t step = LZ4_WAVE % off;
        for (uint32_t j = lane; j < ml; j += LZ4_WAVE) {
            out[d + j] = out[s + idx];
            idx += step;
            idx -= (idx >= off) ? off : 0u;
        }
    }
    wave_fence();
}

// Decode one block with the whole workgroup.  All arguments uniform; dict / dictLen as in decode_block_par (DICT: the previous
// block's output, linked streams).  Returns the block's result -- or CU_REDO -- in every thread.
// dbg (diagnostics, may be null): [0] why the block was left to the lane-parallel decoder (0 = it was not; 2 too many
// candidates, 3 no stop, 4 sweeps without end, 5 a sequential step failed, 6 segments that end after a few KiB again and again:
// `bail`), [1] sequences of the first segment | sweeps of its pointer jumping << 16, [2] shader clocks of the whole block, [3] the
// low halves of the 100 MHz clock behind the first segment's pointer fill and behind its sweeps, [4..15] the 100 MHz clock at the first
// segment's phase boundaries, [14] = start of the first step behind it, [15] = end of the block.
// (forced inline: out of line, `lds` is a generic pointer and every LDS access a flat one -- measured 3-4 x on the parse)
template <bool DICT>
__device__ __forceinline__ int decode_block_cu(const uint8_t *src, int srcLen, uint8_t *dst, int cap, const uint8_t *dict, uint32_t dictLen,
                               const uint8_t *bufLo, const uint8_t *bufHi, uint8_t *lds, uint32_t *dbg = nullptr, bool bail = false, int again = 0)
{
    int dbgAt = 4;
    auto stamp = [&]() {
        if (dbg && threadIdx.x == 0 && dbgAt < 16) dbg[dbgAt] = (uint32_t)wall_clock64();
        dbgAt++;
    };
    uint32_t why = 0;
    const uint32_t cyc0 = (uint32_t)clock64();
    stamp();
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id(), wave = uni((int)(tid >> 6));
    uint8_t *comp = lds;
    uint8_t *tab = lds + CU_OFF_TAB;
    uint32_t *misc = (uint32_t *)(tab + CU_TAB_MISC);
    uint32_t *scanTmp = (uint32_t *)(tab + CU_TAB_SCAN);
    if (!DICT) { dict = nullptr; dictLen = 0; }
    const uint8_t *dictEnd = DICT ? dict + dictLen : dst;
    const int dictLo = DICT ? -(int)min(dictLen, 65535u) : 0;
    int result = 0;
    bool redo = false, finished = false;
    // the block's state between segments (uniform): next token, output produced, which of the reference's loops is running
    int ipBase = 0, opBase = 0;
    bool fast = cap >= 64;                                        // cbits/lz4.c:1791
    // (small blocks, and with them the reference's special cases of empty input and output, :1781-1787: the lane-parallel path)
    if (cap < 256 || srcLen < (int)CU_MINSEG) { redo = true; why = 1; }

    // One step of the sequential decoder by wave 0: seqMode 1 = one sequence, 2 = to the block's end, from (ipBase, opBase).
    // (Called from ONE place, at the top of the loop below, and the decoder's body is expanded once inside a loop of two
    // passes: every further expansion of it costs the kernel 2000 instructions and its registers.)
    int seqMode = 0;
    uint32_t sameRun = 0u;                                       // (again) output bytes in a row that came out as they were
    auto sequential = [&]() {
        __threadfence_block();
        __syncthreads();
        if (wave == 0) {
            SeqState st;
            st.ip = ipBase; st.op = opBase; st.fast = fast;
            int r = SEQ_CONTINUE;
            for (int pass = 0; pass < 2 && r == SEQ_CONTINUE; pass++) {
                // (what decode_par.hpp does behind a single sequence: once the reference would be in its safe loop, or near
                // either end, the rest is sequential too)
                const bool all = seqMode == 2 || pass == 1;
                if (pass == 1 && !(!st.fast || srcLen - st.ip < 64 || cap - st.op < 128)) break;
                r = decode_seq_body<false>(st, all ? 0 : 1, src, srcLen, dst, cap, dict, dictLen, bufLo, bufHi, nullptr);
            }
            r = uni(r);
            if (lane == 0) { misc[CM_RESULT] = (uint32_t)r; misc[CM_NEXT_IP] = (uint32_t)uni(st.ip); misc[CM_NEXT_OP] = (uint32_t)uni(st.op); }
        }
        __threadfence_block();
        __syncthreads();
        const int r = (int)misc[CM_RESULT];
        if (r == SEQ_CONTINUE) { ipBase = (int)misc[CM_NEXT_IP]; opBase = (int)misc[CM_NEXT_OP]; }
        else if (r < 0) { redo = true; why = 5; }                  // the exact code comes from the one exact path
        else { result = r; finished = true; }
        __syncthreads();
    };

    for (int seg = 0; !redo && !finished; seg++) {
        uint32_t remIn = (uint32_t)(srcLen - ipBase), remCap = (uint32_t)(cap - opBase);
        if (!fast || remIn <= CU_TAILMAX || remIn < CU_MINSEG || remCap < 256u) seqMode = 2;
        if (seqMode) {
            sequential();
            seqMode = 0;
            if (redo || finished) break;
            remIn = (uint32_t)(srcLen - ipBase); remCap = (uint32_t)(cap - opBase);
            if (remIn <= CU_TAILMAX || remIn < CU_MINSEG || remCap < 256u) { seqMode = 2; continue; }
        }
        const uint8_t *ssrc = src + ipBase;
        uint8_t *sdst = dst + opBase;
        // bytes staged: what a segment's 32 KiB of output need at the block's ratio so far (the header's for the first segment),
        // and a quarter more -- the parse's tables cost time per staged byte whether the segment gets to use it or not
        const uint64_t est = (opBase > 4096) ? ((uint64_t)CU_OUTMAX * (uint64_t)ipBase / (uint64_t)opBase)
                                            : ((uint64_t)CU_OUTMAX * (uint64_t)srcLen / (uint64_t)cap);
        // (a big block's segments -- all but its last two -- stage at most 1024 chunks, one per thread: the parse's per-chunk phases
        // then run once instead of twice, which is worth more than the 6 % of output a text segment loses to it; 256 blocks of 1 MiB
        // of text: 151 -> 160 GB/s.  A 64 KiB block of text is 35 KiB: two segments of up to 22 KiB, not three of 16)
        const uint32_t cmax = (remIn > 2u * (uint32_t)CU_CMAX) ? (uint32_t)CU_CBIG : (uint32_t)CU_CMAX;
        const uint32_t C = min(min(remIn, cmax), (uint32_t)min(est + est / 4u + 2048u, (uint64_t)cmax));        // bytes staged
        const uint32_t capSeg = min(remCap, (uint32_t)CU_OUTMAX);
        const uint32_t inLim = C - 32u;                          // a plain sequence ends at or before this
        const uint32_t plim = inLim - 2u;                        // ... so its token lies before this
        const uint32_t nChunks = (inLim + CU_CHUNK) / CU_CHUNK;  // T[] covers [0, inLim]
        const uint32_t nSuper = (nChunks + 15u) / 16u;
        const uint32_t NCH = 16u * nSuper;                       // chunks the tables are laid out for
        const uint32_t nNodes = nSuper * CU_CAND;
        uint16_t *Tt = (uint16_t *)(lds + ((cu_at(C + 64u) + 64u + 15u) & ~15u));
        auto T_at = [&](uint32_t p) -> uint16_t & { return Tt[(p & (CU_CHUNK - 1u)) * NCH + (p >> CU_CHUNK_LOG)]; };
        uint16_t *entry = (uint16_t *)(tab + CU_TAB_ENTRY);
        uint32_t *cbits = (uint32_t *)(tab + CU_TAB_CBITS);
        uint32_t *cbits1 = (uint32_t *)(tab + CU_TAB_CBITS1);
        uint16_t *cpos0 = (uint16_t *)(tab + CU_TAB_CPOS0);
        uint16_t *cf0 = (uint16_t *)(tab + CU_TAB_CF0);
        uint16_t *cpos = (uint16_t *)(tab + CU_TAB_CPOS);
        uint16_t *cf = (uint16_t *)(tab + CU_TAB_CF);
        uint16_t *J = (uint16_t *)(tab + CU_TAB_J);
        uint8_t *mark = tab + CU_TAB_MARK;
        if (seg) dbgAt = 16;                                     // (the stamps are the first segment's)

        // ---------------- 1. stage ----------------
        // (a thread has at most two pieces of 16 bytes -- 22 KiB + 64 over 1024 threads --: both are requested before either is stored)
        {
            uint32_t w[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            const uint32_t nPieces = (C + 64u + 15u) / 16u;
#pragma unroll
            for (uint32_t u = 0; u < 2u; u++) {
                const uint32_t i = tid + u * CU_THREADS;
                const uint8_t *q = ssrc + 16u * i;
                if (i < nPieces && 16u * i < C) {
                    if (q >= bufLo && q + 16 <= bufHi) {
                        const par_v4 x = *(const LZ4_GLOBAL par_v4u *)q;
                        w[u][0] = x.x; w[u][1] = x.y; w[u][2] = x.z; w[u][3] = x.w;
                    } else {
                        for (int k = 0; k < 16; k++)
                            if (q + k >= bufLo && q + k < bufHi) w[u][k >> 2] |= (uint32_t)as_global(q)[k] << (8 * (k & 3));
                    }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < 2u; u++) {
                const uint32_t i = tid + u * CU_THREADS;
                if (i < nPieces) {
                    uint32_t *d = (uint32_t *)(comp + cu_at(16u * i));   // (16 bytes never straddle a chunk's padding)
                    d[0] = w[u][0]; d[1] = w[u][1]; d[2] = w[u][2]; d[3] = w[u][3];
                }
            }
        }
        for (uint32_t c = tid; c < NCH; c += CU_THREADS) { entry[c] = (uint16_t)CU_NONE; cbits[c] = 0u; cbits1[c] = 0u; }
        if (tid < CU_NODES) { mark[tid] = 0; }
        if (tid < CM_COUNT) misc[tid] = (tid == CM_NPAR) ? 0xffffffffu : 0u;
        __syncthreads();
        stamp();                                                 // [5] staged
        if (tid == 0) { cbits[0] = 1u; cbits1[0] = 1u; }          // the segment's first token

        // ---------------- 2a. successors, then T[] per chunk (backwards) ----------------
        // The chunk's own bytes come in nine reads at once and every successor is computed from registers (cu_succ; one more
        // read for the match length's extension byte); the backward pass -- T[p] = T[succ(p)] while succ(p) is in the chunk --
        // runs in groups of three positions, which cannot name each other (a sequence is at least three bytes), so that a
        // group is one LDS round trip.
        for (uint32_t c = tid; c < nChunks; c += CU_THREADS) {
            const uint32_t base = c * CU_CHUNK, cend = base + CU_CHUNK;
            uint32_t W[CU_CHUNK / 4 + 1];
            {
                const uint32_t *wp = (const uint32_t *)(comp + cu_at(base));
#pragma unroll
                for (int j = 0; j < CU_CHUNK / 4; j++) W[j] = wp[j];
                W[CU_CHUNK / 4] = *(const uint32_t *)(comp + cu_at(base + CU_CHUNK));
            }
            // (the chunk's upper half first, then the lower: the backward pass of the lower half finds the upper half's entries in
            // LDS like everything else it looks up, and sixteen successors are live at a time instead of thirty-two)
#pragma unroll
            for (int half = 1; half >= 0; half--) {
                uint32_t S[CU_CHUNK / 2];
#pragma unroll
                for (int kk = 0; kk < CU_CHUNK / 2; kk++) {
                    const int k = (CU_CHUNK / 2) * half + kk;
                    const uint32_t p = base + (uint32_t)k;
                    const uint32_t t = (W[k >> 2] >> (8 * (k & 3))) & 0xffu;
                    const uint32_t b1 = (W[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 0xffu;
                    const uint32_t b1b = (W[(k + 2) >> 2] >> (8 * ((k + 2) & 3))) & 0xffu;
                    uint32_t lit, litBytes;
                    bool bad;
                    cu_lit(t, b1, b1b, lit, litBytes, bad);
                    uint32_t nxt = p + 3u + litBytes + lit;
                    if ((t & 15u) == 15u) {                             // the match length's extension byte(s)
                        nxt++;
                        const uint32_t b2 = comp[cu_at(min(nxt - 1u, inLim))];
                        if (b2 == 255u) { bad = bad || comp[cu_at(min(nxt, inLim))] == 255u; nxt++; }
                    }
                    S[kk] = (!bad && p < plim && nxt <= inLim) ? nxt : CU_STOP;
                }
                // backwards, three positions at a time (kk = 15, 14, 13; 12, 11, 10; ...; 0)
#pragma unroll
                for (int g = CU_CHUNK / 2 - 1; g >= 0; g -= 3) {
                    uint32_t r[3];
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const int kk = g - j;
                        r[j] = CU_STOP;
                        if (kk >= 0 && S[kk < 0 ? 0 : kk] < cend) r[j] = T_at(S[kk < 0 ? 0 : kk]);
                    }
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const int kk = g - j;
                        if (kk >= 0) {
                            const int k = (CU_CHUNK / 2) * half + kk;
                            const uint32_t p = base + (uint32_t)k;
                            const uint32_t sv = (S[kk] < cend) ? r[j] : S[kk];
                            Tt[(uint32_t)k * NCH + c] = (uint16_t)sv;
                            if (sv != CU_STOP && (sv / CU_SUPER) != (p / CU_SUPER)) atomicOr(&cbits[sv >> 5], 1u << (sv & 31u));
                        }
                    }
                }
            }
        }
        __syncthreads();
        stamp();                                                 // [6] T[]

        // ---------------- 2b. candidates of every super-chunk, and where each leaves it ----------------
        // the j-th candidate (set bit) of super-chunk k in a bit vector with one word per chunk; *total = how many it has
        auto nth_cand = [&](const uint32_t *bitsOf, uint32_t k, uint32_t j, uint32_t *total) -> uint32_t {
            uint32_t pos = CU_NONE, seen = 0;
#pragma unroll 1
            for (uint32_t w4 = 0; w4 < CU_SUPER_WORDS / 4u; w4++) {    // (a rolled loop: unrolled, the words and the hops' state spilled)
                const uint4 v = *(const uint4 *)&bitsOf[CU_SUPER_WORDS * k + 4u * w4];
                const uint32_t b4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t pc = (uint32_t)__builtin_popcount(b4[q]);
                    if (pos == CU_NONE && seen + pc > j) {
                        uint32_t b = b4[q];
                        for (uint32_t r = j - seen; r; r--) b &= b - 1u;
                        pos = (CU_SUPER_WORDS * k + 4u * w4 + (uint32_t)q) * 32u + (uint32_t)__builtin_ctz(b);
                    }
                    seen += pc;
                }
            }
            *total = seen;
            return pos;
        };
        // First every position some T[p] names (up to CU_CAND0 per super-chunk: chains that started a few bytes in front of
        // the boundary have not met the true chain yet, 7 to 9 of them differ) hops through its super-chunk; where THOSE
        // leave it are the candidates proper: a chain that has run for 512 bytes has, as a rule, met the true one -- one or
        // two per super-chunk -- and the true entry is always among them, because the true entry of the super-chunk the
        // chain came from was a first-round candidate there.
        for (uint32_t n = tid; n < nSuper * CU_CAND0; n += CU_THREADS) {
            const uint32_t k = n / CU_CAND0, j = n % CU_CAND0;
            uint32_t total;
            const uint32_t pos = nth_cand(cbits, k, j, &total);
            if (j == 0 && total > CU_CAND0) misc[CM_OVERFLOW] = 1u;
            const uint32_t send = (k + 1u) * CU_SUPER;
            uint32_t q = pos;
            if (pos != CU_NONE) {
                for (int hop = 0; hop < 16 && q != CU_STOP && q < send; hop++) q = T_at(q);    // (a hop leaves a chunk: 16 at most)
                if (q != CU_STOP && q < send) q = CU_STOP;
                if (q != CU_STOP) atomicOr(&cbits1[q >> 5], 1u << (q & 31u));
            }
            cpos0[n] = (uint16_t)pos;
            cf0[n] = (uint16_t)q;
        }
        __syncthreads();
        // the candidates proper, with the exit the first round found for them
        for (uint32_t n = tid; n < CU_NODES; n += CU_THREADS) {
            uint32_t pos = CU_NONE, q = CU_NONE;
            if (n < nNodes) {
                const uint32_t k = n / CU_CAND, j = n % CU_CAND;
                uint32_t total;
                pos = nth_cand(cbits1, k, j, &total);
                if (j == 0 && total > CU_CAND) misc[CM_OVERFLOW] = 1u;
                if (pos != CU_NONE) {
                    q = CU_STOP;
                    uint32_t at = CU_NONE;                        // where the first round has this position
#pragma unroll 1
                    for (uint32_t h = 0; h < 2u; h++) {
                        const uint4 v = *(const uint4 *)&cpos0[k * CU_CAND0 + 8u * h];
                        const uint32_t w8[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 8; e++)
                            if (((w8[e >> 1] >> (16 * (e & 1))) & 0xffffu) == pos) at = k * CU_CAND0 + 8u * h + (uint32_t)e;
                    }
                    if (at != CU_NONE) q = cf0[at];
                    else misc[CM_OVERFLOW] = 1u;                  // (a second-round candidate is a first-round one, or the segment's first byte)
                }
            }
            cpos[n] = (uint16_t)pos;
            cf[n] = (uint16_t)q;
        }
        __syncthreads();
        // the list: node -> the node its exit is a candidate of (CU_NODES - 1 = the end)
        for (uint32_t n = tid; n < CU_NODES; n += CU_THREADS) {
            uint32_t nx = CU_NODES - 1u;
            const uint32_t f = cf[n];
            if (n < nNodes && cpos[n] != CU_NONE && f != CU_STOP) {
                const uint32_t k2 = f / CU_SUPER;
                bool found = false;
                for (uint32_t j2 = 0; j2 < CU_CAND; j2++)
                    if (k2 < nSuper && cpos[k2 * CU_CAND + j2] == f) { nx = k2 * CU_CAND + j2; found = true; }
                if (!found) misc[CM_OVERFLOW] = 1u;
            }
            J[n] = (uint16_t)nx;
        }
        __syncthreads();
        stamp();                                                 // [7] candidates, list

        // ---------------- 2c. one wave: the nodes reachable from node 0 ----------------
        // (a list over nSuper super-chunks is at most nSuper nodes long: 2^levels >= nSuper + 1 hops reach its end)
        const int levels = min((int)CU_LEVELS, 32 - __builtin_clz(nSuper));
        if (wave == 0) {
            for (int l = 1; l < levels; l++) {
                const uint16_t *Jp = J + (l - 1) * CU_NODES;
                uint16_t *Jn = J + l * CU_NODES;
                uint32_t v[CU_NODES / LZ4_WAVE];
#pragma unroll
                for (int i = 0; i < CU_NODES / LZ4_WAVE; i++) v[i] = Jp[Jp[lane + LZ4_WAVE * i]];
#pragma unroll
                for (int i = 0; i < CU_NODES / LZ4_WAVE; i++) Jn[lane + LZ4_WAVE * i] = (uint16_t)v[i];
                wave_fence();
            }
            if (lane == 0) mark[0] = 1;
            wave_fence();
            for (int l = levels - 1; l >= 0; l--) {
                const uint16_t *Jl = J + l * CU_NODES;
                uint32_t m[CU_NODES / LZ4_WAVE], t[CU_NODES / LZ4_WAVE];
#pragma unroll
                for (int i = 0; i < CU_NODES / LZ4_WAVE; i++) { m[i] = mark[lane + LZ4_WAVE * i]; t[i] = Jl[lane + LZ4_WAVE * i]; }
#pragma unroll
                for (int i = 0; i < CU_NODES / LZ4_WAVE; i++) if (m[i]) mark[t[i]] = 1;
                wave_fence();
            }
        }
        __syncthreads();
        stamp();                                                 // [8] list ranked

        // ---------------- 2d. the true entries hop through their super-chunk: every chunk's entry ----------------
        if (tid < nNodes && mark[tid] && cpos[tid] != CU_NONE) {
            uint32_t q = cpos[tid];
            const uint32_t send = (tid / CU_CAND + 1u) * CU_SUPER;
            for (int hop = 0; hop < 16 && q != CU_STOP && q < send; hop++) { entry[q / CU_CHUNK] = (uint16_t)q; q = T_at(q); }
        }
        __syncthreads();
        if (misc[CM_OVERFLOW] != 0u) { redo = true; why = 2; }
        stamp();                                                 // [9] entries
        if (redo) break;

        uint2 *rec = (uint2 *)(lds + CU_OFF_REC);
        uint8_t *out = lds;
        // (the LDS index of the segment's output position x is x: the match phase stores aligned words; the flush's 16-byte stores
        // are as aligned in global memory as the segment's first byte happens to be)
        constexpr uint32_t A = 0u;

        // ---------------- 2e. every chunk walks its sequences: counts, scan, records ----------------
        // (thread t owns chunks 2t and 2t + 1: the scan runs in chunk order; the two are walked side by side: a sequence is
        // two dependent LDS reads, and this way the two chunks' reads are in flight together)
        uint32_t n2[2] = {0u, 0u}, len2[2] = {0u, 0u};
        uint2 *walked = (uint2 *)(lds + CU_OFF_PTR);
        const uint32_t c0 = 2u * tid, c1 = 2u * tid + 1u;
        const uint32_t e0 = (c0 < nChunks) ? (uint32_t)entry[c0] : CU_NONE, e1 = (c1 < nChunks) ? (uint32_t)entry[c1] : CU_NONE;
        const uint32_t cend0 = (c0 + 1u) * CU_CHUNK, cend1 = (c1 + 1u) * CU_CHUNK;
        {
            uint32_t q0 = e0, q1 = e1;
            bool a0 = c0 < nChunks && q0 < cend0, a1 = c1 < nChunks && q1 < cend1;     // (CU_NONE is beyond every chunk there is)
            while (a0 || a1) {
                const CuSeq s0 = cu_parse(comp, a0 ? q0 : 0u, inLim), s1 = cu_parse(comp, a1 ? q1 : 0u, inLim);
                // (a chunk that stops takes one index for the token it stops at: T[] does not know everything the walk checks, so
                // chunks behind a stop may have entries of their own, and every stop must have an index no other chunk has --
                // the smallest one is the segment's end)
                // (what the walk has parsed is kept for the second one -- CU_SLOTS sequences a chunk, in the pointers' area, which is
                // free until the match phase: a parse is ~55 instructions, and this phase is bound by instruction issue)
                auto keep = [&](const CuSeq &sq, const uint32_t c, const uint32_t j, const uint32_t q) {
                    if (j >= CU_SLOTS) return;
                    if (sq.nxt == CU_STOP) walked[c * CU_SLOTS + j] = make_uint2(q, 0x80000000u | (sq.odd ? 1u : 0u));
                    else walked[c * CU_SLOTS + j] = make_uint2(sq.litStart | (sq.off << 16), sq.lit | (sq.ml << 10) | ((sq.nxt - sq.litStart) << 20));
                };
                if (a0) { keep(s0, c0, n2[0], q0); if (s0.nxt == CU_STOP) { a0 = false; n2[0]++; } else { n2[0]++; len2[0] += s0.lit + s0.ml; q0 = s0.nxt; a0 = q0 < cend0; } }
                if (a1) { keep(s1, c1, n2[1], q1); if (s1.nxt == CU_STOP) { a1 = false; n2[1]++; } else { n2[1]++; len2[1] += s1.lit + s1.ml; q1 = s1.nxt; a1 = q1 < cend1; } }
            }
        }
        uint32_t totN, totLen, seqBase, opScan;
        cu_scan_excl2(n2[0] + n2[1], len2[0] + len2[1], scanTmp, &seqBase, &opScan, &totN, &totLen);
        uint32_t myStop = 0xffffffffu, myStopIp = 0, myStopOp = 0, myStopKind = 0;
        {
            // plain on the output side too: the source lies in the block -- or entirely in the dictionary --, the reference's
            // fast loop would not change loops (cbits/lz4.c:1818, :1858-1863), and the sequence has a slot
            auto stop_at = [&](const uint32_t i, const uint32_t q, const uint32_t op, const uint32_t kind) {
                if (i < myStop) { myStop = i; myStopIp = q; myStopOp = op; myStopKind = kind; }       // (the earlier of my two chunks' stops)
            };
            // one sequence: false = the chunk's walk ends here
            auto place = [&](const uint32_t litStart, const uint32_t lit, const uint32_t off, const uint32_t ml, uint32_t &i, uint32_t &op) -> bool {
                const uint32_t outEnd = op + lit + ml;
                const int sposBlk = opBase + (int)(op + lit) - (int)off;             // the source, relative to the block's output
                const bool srcOk = sposBlk >= 0 || (DICT && sposBlk >= dictLo && sposBlk + (int)ml <= 0);
                if (!(srcOk && outEnd + 64u < capSeg && i < CU_NMAX)) {
                    // (the token: in front of the literal run's length bytes -- none below 15 literals, one below 270, else two)
                    stop_at(i, litStart - 1u - (lit < 15u ? 0u : (lit < 270u ? 1u : 2u)), op, !srcOk ? 1u : 0u);   // 1: one for the sequential decoder
                    return false;
                }
                rec[i] = make_uint2(op | (litStart << 16), lit | (off << 16));
                i++; op = outEnd;
                return true;
            };
            auto chunk = [&](const uint32_t c, const uint32_t n, uint32_t i, uint32_t op, const uint32_t cend) {
                if (n == 0u) return;
                uint2 w[CU_SLOTS];
#pragma unroll
                for (uint32_t j = 0; j < CU_SLOTS; j++) w[j] = walked[c * CU_SLOTS + min(j, n - 1u)];
                bool act = true;
                uint32_t q = 0u;
#pragma unroll
                for (uint32_t j = 0; j < CU_SLOTS; j++) {
                    if (!(act && j < n)) continue;
                    if (w[j].y & 0x80000000u) { stop_at(i, w[j].x, op, w[j].y & 1u); act = false; continue; }   // (a stop is the last one)
                    const uint32_t litStart = w[j].x & 0xffffu;
                    q = litStart + (w[j].y >> 20);
                    act = place(litStart, w[j].y & 1023u, w[j].x >> 16, (w[j].y >> 10) & 1023u, i, op);
                }
                // more sequences in 16 bytes than there are slots (six tokens three bytes apart): parsed again
                if (n > CU_SLOTS && act) {
                    act = q < cend;
                    while (act) {
                        const CuSeq sq = cu_parse(comp, q, inLim);
                        if (sq.nxt == CU_STOP) { stop_at(i, q, op, sq.odd ? 1u : 0u); break; }
                        if (!place(sq.litStart, sq.lit, sq.off, sq.ml, i, op)) break;
                        q = sq.nxt;
                        act = q < cend;
                    }
                }
            };
            chunk(c0, n2[0], seqBase, opScan, cend0);
            chunk(c1, n2[1], seqBase + n2[0], opScan + len2[0], cend1);
        }
        if (myStop != 0xffffffffu) atomicMin(&misc[CM_NPAR], myStop);
        __syncthreads();
        const uint32_t nPar = misc[CM_NPAR];
        if (myStop == nPar && nPar != 0xffffffffu) {
            misc[CM_TAIL_IP] = myStopIp; misc[CM_TAIL_OP] = myStopOp; misc[CM_TAIL_KIND] = myStopKind;
            rec[nPar] = make_uint2(myStopOp, 0u);                 // (the last match's length is the next record's start - ...)
        }
        __syncthreads();                                          // the compressed bytes and the parse's tables are dead from here on
        if (nPar == 0xffffffffu) { redo = true; why = 3; }        // (cannot happen: a segment's last sequence is never plain)
        stamp();                                                  // [10] records
        if (dbg && tid == 0 && seg == 0) dbg[1] = nPar;
        if (redo) break;
        const uint32_t tailIp = misc[CM_TAIL_IP], tailOp = misc[CM_TAIL_OP], tailKind = misc[CM_TAIL_KIND];

        if (nPar > 0u) {
            uint4 *rk = (uint4 *)(tab + CU_TAB_RANK);
            uint32_t *rkw = (uint32_t *)rk;
            if (tid <= CU_OUTMAX / 64) rk[tid] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();

            // ---------------- 3. literals, and what a match takes from in front of the segment (global memory both) ----------------
            for (uint32_t i = tid; i < nPar; i += CU_THREADS) {
                const uint2 r = rec[i];
                const uint32_t outStart = r.x & 0xffffu, litStart = r.x >> 16, lit = r.y & 0xffffu, off = r.y >> 16;
                atomicOr(&rkw[4u * (outStart >> 6) + ((outStart >> 5) & 1u)], 1u << (outStart & 31u));
                const uint32_t nextStart = rec[i + 1u].x & 0xffffu;
                const uint32_t ml = nextStart - outStart - lit;
                const LZ4_GLOBAL uint8_t *g = as_global(ssrc + litStart);
                const uint32_t dA = A + outStart;
                if (lit > 16u) {                                    // (at most 524)
                    cu_lane_fetch(out, dA, g, lit);
                } else if (lit > 8u) {
                    // (a plain sequence ends 32 bytes before the staged bytes do: 16 bytes from its literals' start are there)
                    const par_v4 v = *(const LZ4_GLOBAL par_v4u *)g;
                    if (lit + ml >= 16u) {
                        *(par_v4u *)&out[dA] = v;                  // the bytes past the literals fall into my own match area, written later
                    } else {
                        const uint64_t a = (uint64_t)v.x | ((uint64_t)v.y << 32);
                        const uint32_t sh = lit - 8u;              // 1..8: the last 8 literals start sh bytes in
                        const uint32_t w1 = sh >= 4u ? v.y : v.x, w2 = sh >= 4u ? v.z : v.y, w3 = sh >= 4u ? v.w : v.z;
                        const uint64_t b = (sh == 8u) ? ((uint64_t)v.z | ((uint64_t)v.w << 32))
                                                      : ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, sh & 3u) |
                                                         ((uint64_t)__builtin_amdgcn_alignbyte(w3, w2, sh & 3u) << 32));
                        *(par_u64u *)&out[dA] = a;
                        *(par_u64u *)&out[dA + sh] = b;
                    }
                } else if (lit > 0u) {
                    const uint64_t v = *(const LZ4_GLOBAL par_u64u *)g;
                    if (lit + ml >= 8u) {
                        *(par_u64u *)&out[dA] = v;
                    } else {
                        uint64_t w = v;
                        for (uint32_t q = 0; q < lit; q++) { out[dA + q] = (uint8_t)w; w >>= 8; }
                    }
                }
                // A match whose source starts in front of the segment: those bytes are final in global memory (an earlier
                // segment's output, or the dictionary: entirely, see `srcOk`).  They are fetched now -- behind my literals: a
                // literal store may run over into my match area -- and the record's literal count grows by as much: what is left
                // of the match, if anything, is a match whose source is the segment's first byte.
                const int spos = (int)(outStart + lit) - (int)off;
                if (spos < 0) {
                    const uint32_t k = min(ml, (uint32_t)(-spos));
                    const int sposBlk = opBase + spos;
                    const LZ4_GLOBAL uint8_t *gs = as_global((DICT && sposBlk < 0) ? dictEnd + sposBlk : dst + sposBlk);
                    cu_lane_fetch(out, dA + lit, gs, k);
                    rec[i].y = (lit + k) | (off << 16);
                }
            }
            __syncthreads();
            stamp();                                              // [11] literals
            // rank records: sequences that start before each group of 64 output positions
            {
                uint4 w = make_uint4(0u, 0u, 0u, 0u);
                if (tid < CU_OUTMAX / 64) w = rk[tid];
                const uint32_t cnt = (uint32_t)__builtin_popcount(w.x) + (uint32_t)__builtin_popcount(w.y);
                uint32_t tot;
                const uint32_t pre = cu_scan_excl(cnt, scanTmp, &tot);
                if (tid < CU_OUTMAX / 64) rkw[4u * tid + 2u] = pre;
            }
            __syncthreads();
            stamp();                                              // [12] rank records

            // ---------------- 4. matches ----------------
            // Every output byte gets a 16-bit SOURCE POINTER: itself when it is in place already (a literal, or a byte a match took
            // from in front of the segment), else the byte its match copies (cbits/lz4.c:1866-1924: op - offset + k).  Pointer jumping
            // -- ptr[x] = ptr[ptr[x]], every byte at once, all sixteen waves, in place (every value a pointer ever holds is an ancestor
            // of its byte, so the order of the updates does not matter) -- halves every chain per round: the dependence graph of a
            // segment is 80 to 300 matches deep (32 768 for a run of one byte) and is resolved in 8 to 10 rounds (16), whatever it looks
            // like; then every byte is fetched from the literal its pointer has arrived at.  The first form of this phase walked the
            // graph match by match through done bits -- eight waves polling, one to three of 64 lanes busy per step: 0.9 us a level,
            // 91-94 us of a 64 KiB segment's 134-155.
            {
                uint2 *ptr64 = (uint2 *)(lds + CU_OFF_PTR);                  // four pointers a quad: bytes 4k .. 4k + 3
                const uint16_t *ptr16 = (const uint16_t *)(lds + CU_OFF_PTR);
                const uint32_t nQ = (tailOp + 3u) >> 2;
                // (every loop of this phase is a chain of dependent LDS round trips -- ~100 ns each -- per quad, and a thread has up to
                // eight quads: the loops are unrolled in batches, every batch's reads of one kind issued together, so that a round trip
                // is paid per batch and not per quad.  Quads, not bytes or pairs: the fill is bound by vector issue -- a rank in the
                // sequence starts' bit vector per quad and one add per byte, instead of a rank per byte -- and rounds and gather by LDS
                // instructions: six per quad each)
                constexpr uint32_t PERQ = CU_OUTMAX / 4u / CU_THREADS;        // quads per thread: 8
                for (uint32_t m0_ = 0; m0_ < PERQ && tid + m0_ * CU_THREADS < nQ; m0_ += CU_MLPF) {
                    uint4 rr[CU_MLPF];
                    uint32_t ix0[CU_MLPF], ix3[CU_MLPF];
                    uint2 rlo[CU_MLPF], rhi[CU_MLPF];
#pragma unroll
                    for (uint32_t u = 0; u < CU_MLPF; u++) {
                        const uint32_t k = min(tid + (m0_ + u) * CU_THREADS, nQ - 1u);          // (a quad past the end repeats the last one)
                        rr[u] = rk[k >> 4];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < CU_MLPF; u++) {
                        const uint32_t k = min(tid + (m0_ + u) * CU_THREADS, nQ - 1u);
                        const uint32_t x0 = 4u * k, sh = x0 & 31u;                               // (sh <= 28: the quad's four bits lie in one word)
                        const bool hi = (x0 & 32u) != 0u;
                        const uint32_t w = hi ? rr[u].y : rr[u].x;
                        // the sequence that holds x0: the starts at or before it.  A sequence is at least four bytes long (its match is), so
                        // at most one more starts inside the quad: two records serve its four bytes
                        ix0[u] = rr[u].z + (hi ? (uint32_t)__builtin_popcount(rr[u].x) : 0u) + (uint32_t)__builtin_popcount(w & ((2u << sh) - 1u)) - 1u;
                        ix3[u] = ix0[u] + (((w >> sh) & 0xeu) != 0u ? 1u : 0u);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < CU_MLPF; u++) { rlo[u] = rec[ix0[u]]; rhi[u] = rec[ix3[u]]; }
#pragma unroll
                    for (uint32_t u = 0; u < CU_MLPF; u++) {
                        const uint32_t k = min(tid + (m0_ + u) * CU_THREADS, nQ - 1u);
                        // (a record: first output byte | ..., literals -- and what came from in front of the segment -- | offset << 16)
                        const uint32_t sHi = (ix3[u] != ix0[u]) ? (rhi[u].x & 0xffffu) : 0xffffffffu;
                        const uint32_t dLo = (rlo[u].x & 0xffffu) + (rlo[u].y & 0xffffu), dHi = (rhi[u].x & 0xffffu) + (rhi[u].y & 0xffffu);
                        const uint32_t oLo = rlo[u].y >> 16, oHi = rhi[u].y >> 16;
                        uint32_t pp[4];
#pragma unroll
                        for (uint32_t j = 0; j < 4u; j++) {
                            const uint32_t x = 4u * k + j;
                            const bool h = x >= sHi;
                            pp[j] = (x < (h ? dHi : dLo) || x >= tailOp) ? x : x - (h ? oHi : oLo);
                        }
                        ptr64[k] = make_uint2(pp[0] | (pp[1] << 16), pp[2] | (pp[3] << 16));
                    }
                }
                __syncthreads();
                uint32_t tPh0 = 0u, tPh1 = 0u; int nRounds = 0;
                if (dbg && tid == 0 && seg == 0) tPh0 = (uint32_t)wall_clock64();
                // rounds: a quad of pointers that does not move any more points at bytes that are in place, and is left alone
                uint32_t live = 0u;
                {
                    uint32_t m = 0u;
                    for (uint32_t k = tid; k < nQ; k += CU_THREADS) live |= 1u << m++;
                }
                // No barrier between the sweeps: a pointer that is read late or early is an ancestor of its byte either way, a quad is
                // done when all four of its pointers have arrived at bytes that point at themselves -- which it sees on its own --, and
                // pointers only ever move towards the segment's start, so every thread's loop ends (a wave whose quads are done leaves
                // the issue slots to the others).  With a barrier and a shared "something moved" word per round: 11.5 us instead of 8.6.
                while (live != 0u) {
                    for (uint32_t m0_ = 0; m0_ < PERQ && (live >> m0_) != 0u; m0_ += CU_MLP) {
                        if (((live >> m0_) & ((1u << CU_MLP) - 1u)) == 0u) continue;
                        uint2 v[CU_MLP];
                        uint32_t q[CU_MLP][4];
                        // (a quad that is not live -- settled, or past the end -- reads quad 0, which is there, and stores nothing)
#pragma unroll
                        for (uint32_t u = 0; u < CU_MLP; u++) v[u] = ptr64[((live >> (m0_ + u)) & 1u) ? tid + (m0_ + u) * CU_THREADS : 0u];
#pragma unroll
                        for (uint32_t u = 0; u < CU_MLP; u++) {
                            q[u][0] = ptr16[v[u].x & 0xffffu]; q[u][1] = ptr16[v[u].x >> 16];
                            q[u][2] = ptr16[v[u].y & 0xffffu]; q[u][3] = ptr16[v[u].y >> 16];
                        }
#pragma unroll
                        for (uint32_t u = 0; u < CU_MLP; u++) {
                            if (!((live >> (m0_ + u)) & 1u)) continue;
                            const uint32_t n0 = q[u][0] | (q[u][1] << 16), n1 = q[u][2] | (q[u][3] << 16);
                            if (n0 != v[u].x || n1 != v[u].y) ptr64[tid + (m0_ + u) * CU_THREADS] = make_uint2(n0, n1);
                            else live &= ~(1u << (m0_ + u));
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");     // (the next sweep reads what the others have stored by now)
                    // (sixteen sweeps resolve any segment when all waves sweep together; the bound is the loop's guaranteed exit)
                    if (++nRounds >= CU_SWEEPS) { misc[CM_ABORT] = 1u; break; }
                }
                __syncthreads();
                // (diagnostics: [3] = the first segment's pointer fill's end and the rounds' end as 100 MHz stamps' low halves, [1] |= rounds << 16)
                if (dbg && tid == 0 && seg == 0) { tPh1 = (uint32_t)wall_clock64(); dbg[3] = (tPh0 & 0xffffu) | (tPh1 << 16); dbg[1] = nPar | ((uint32_t)nRounds << 16); }
                // every byte from the byte its pointer has arrived at (a byte in place points at itself), a word at a time.  (All reads
                // of a batch before its stores: a byte that is read is in place and a store leaves it as it is -- every other byte of
                // a stored word is either the quad's own or in place too -- so stores never change what a later read returns; but the
                // compiler cannot know that.  The bytes of the last quad at or behind tailOp point at themselves.)
                for (uint32_t m0_ = 0; m0_ < PERQ && tid + m0_ * CU_THREADS < nQ; m0_ += CU_MLP) {
                    uint2 v[CU_MLP];
                    uint32_t bb[CU_MLP][4];
#pragma unroll
                    for (uint32_t u = 0; u < CU_MLP; u++) v[u] = ptr64[min(tid + (m0_ + u) * CU_THREADS, nQ - 1u)];
#pragma unroll
                    for (uint32_t u = 0; u < CU_MLP; u++) {
                        bb[u][0] = out[v[u].x & 0xffffu]; bb[u][1] = out[v[u].x >> 16];
                        bb[u][2] = out[v[u].y & 0xffffu]; bb[u][3] = out[v[u].y >> 16];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < CU_MLP; u++) {
                        const uint32_t k = tid + (m0_ + u) * CU_THREADS;
                        if (k < nQ) *(uint32_t *)&out[4u * k] = bb[u][0] | (bb[u][1] << 8) | (bb[u][2] << 16) | (bb[u][3] << 24);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            if (misc[CM_ABORT] != 0u) { redo = true; why = 4; }
            stamp();                                              // [13] matches
            if (redo) break;

            // ---------------- 5. flush the segment's output [0, tailOp) ----------------
            {
                LZ4_GLOBAL uint8_t *gd = as_global(sdst);
                uint32_t d = 0u;                                  // (again: does the segment differ from what the decode before this one left here?)
                for (uint32_t k = tid; k < tailOp / 16u; k += CU_THREADS) {
                    const par_v4 v = *(const par_v4 *)&out[16u * k];
                    if (again > 0) { const par_v4 o = *(const LZ4_GLOBAL par_v4u *)(gd + 16u * k); d |= (v.x ^ o.x) | (v.y ^ o.y) | (v.z ^ o.z) | (v.w ^ o.w); }
                    *(LZ4_GLOBAL par_v4u *)(gd + 16u * k) = v;
                }
                for (uint32_t x = (tailOp / 16u) * 16u + tid; x < tailOp; x += CU_THREADS) {
                    if (again > 0) d |= (uint32_t)(gd[x] ^ out[x]);
                    gd[x] = out[x];
                }
                if (d) misc[CM_DIFF] = 1u;
            }
        }
        // ---------------- 6. what comes next ----------------
        __threadfence_block();
        __syncthreads();                                          // (the segment's output is in global memory; nobody reads misc[] or the LDS output any more)
        ipBase += (int)tailIp; opBase += (int)tailOp;
        if (seg == 0) { dbgAt = 14; stamp(); }                    // [14] flushed
        // `again` (> 0: what this block decoded to the last time, with another dictionary; the bytes are still in dst): once 64 KiB in a
        // row have come out the same, everything behind them will -- no match reaches further back, the dictionary is out of reach,
        // and where the sequences lie does not depend on the bytes -- and the block is done.  (A step of the sequential decoder is not
        // compared: the count starts again behind it.)
        if (again > 0) {
            const bool same = nPar > 0u && misc[CM_DIFF] == 0u && tailKind != 1u;
            __syncthreads();                                      // (misc[] is cleared by the next segment's first step)
            sameRun = same ? sameRun + tailOp : 0u;
            if (sameRun >= 65536u && (uint32_t)opBase >= 65536u) { result = again; finished = true; }
        }
        // (bail: the caller takes blocks that do not suit this form -- segments that end after a few KiB, at a literal run of
        // hundreds of bytes, again and again -- back to the lane-parallel decoder: a segment's fixed costs are ~30 us)
        if (bail && seg >= 7 && (uint32_t)opBase < (uint32_t)(seg + 1) * 2048u && (uint32_t)(srcLen - ipBase) > 4096u) { redo = true; why = 6; }
        // a sequence the parse does not take, or a segment that took nothing: one sequence by the sequential decoder
        if (tailKind == 1u || nPar == 0u) seqMode = 1;
    }
    __syncthreads();
    if (redo) result = CU_REDO;
    dbgAt = 15;
    stamp();                                                     // [15] end
    if (dbg && tid == 0) { dbg[0] = why; dbg[2] = (uint32_t)clock64() - cyc0; }
    return result;
}

} // namespace lz4dev
