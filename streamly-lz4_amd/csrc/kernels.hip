// kernels.hip -- gfx950 kernels of the MI355X LZ4 block engine and their launchers.
//
//   K1  k_decode_par / k_decode_seq      block decode (decode_par.hpp, decode_seq.hpp)
//       k_decode_tolerant, k_ptr_*       linked streams, second pass (linked_ptr.hpp)
//       k_decode_fixup_regions / _linked ... its in-order fallbacks (linked_replay.hpp)
//   K2  k_encode<TabT, DICT>             block encode, independent or linked (encode_wave.hpp)
//   K3  k_scan_u64 + k_copy_slots        size scan + compaction into the framed stream
//       k_header_sizes                   uncompressed-size scan from block headers
//       k_generate                       synthetic inputs (bench/test support)
//
// Everything is HBM/LDS byte work; there is deliberately no MFMA anywhere.
#include "kernels.h"

#include <algorithm>

#include "decode_seq.hpp"
#include "decode_par.hpp"
#include "decode_cu.hpp"
#include "linked_replay.hpp"
#include "linked_ptr.hpp"
#include "encode_wave.hpp"

using namespace lz4dev;

// ---------------------------------------------------------------------------
// K1: decode
// ---------------------------------------------------------------------------

// Validate one block header the way decompressChunk does
// (reference src/Streamly/Internal/LZ4.hs:299-318) -- plus the short-array case
// it misses.  Returns 0 or a MI355LZ4_BLK_E_* code; fills compLen / cap.
__device__ __forceinline__ int read_block_header(const DecodeArgs &a, int blk, const uint8_t *&data,
                                                 int &compLen, int &cap)
{
    const uint64_t off = a.blockOff[blk];
    if (off + (uint64_t)a.headerKind > a.framedLen) return BLK_E_TRUNCATED;
    const uint8_t *hdr = a.framed + off;
    compLen = load_le32(hdr);
    int uncomp = (a.headerKind == 8) ? load_le32(hdr + 4) : a.fixedUncomp;
    if (compLen <= 0 || compLen > MAX_COMP_LEN) return BLK_E_COMPLEN;
    if (off + (uint64_t)a.headerKind + (uint64_t)compLen > a.framedLen) return BLK_E_TRUNCATED;
    if (uncomp < 0) return BLK_E_UNCOMPLEN;
    cap = uncomp;
    if (a.outCap) {
        if (a.headerKind == 8 && uncomp > a.outCap[blk]) return BLK_E_UNCOMPLEN;
        if (a.headerKind != 8) cap = a.outCap[blk];
    }
    data = hdr + a.headerKind;
    return 0;
}

// A codec error (not a header rejection) is what a block of a linked stream reports when it is decoded without
// its dictionary: the second pass is launched only when the standalone pass counted some.
__device__ __forceinline__ bool is_codec_error(int r) { return r < 0 && r > -0x7F000000; }

// linkStat = {dependent blocks, first, last, -, largest capacity among them}, from result[] once the standalone pass
// is done.  (Round 3 had every failing block add to these five words itself: four atomics per block on ONE cache line,
// 16 384 of them for a reference-written stream of 4096 blocks, which cost the standalone pass 0.33 of its 0.38 ms --
// the blocks themselves give up at their first sequence.)  One workgroup per 1024 blocks, one set of atomics each.
#define LINK_RUN_CAP 64
__global__ __launch_bounds__(1024) void k_link_stat(DecodeArgs a)
{
    __shared__ uint32_t sh[4];
    const int tid = (int)threadIdx.x;
    if (tid == 0) { sh[0] = 0u; sh[1] = 0xffffffffu; sh[2] = 0u; sh[3] = 0u; }
    __syncthreads();
    const int blk = (int)(blockIdx.x * 1024u) + tid;
    bool bad = false;
    int cap = 0;
    if (blk < a.nBlocks && is_codec_error(a.result[blk])) {
        const uint8_t *data = nullptr;
        int compLen = 0;
        bad = read_block_header(a, blk, data, compLen, cap) == 0;     // (a codec error means the header was accepted)
    }
    const uint64_t m = __ballot(bad);
    if (m) {
        // wave-level first: one lane per wave talks to LDS
        int wcap = bad ? cap : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wcap = max(wcap, __shfl_xor(wcap, d));
        if ((tid & 63) == 0) {
            const int w0 = blk;                                        // first lane's block
            atomicAdd(&sh[0], (uint32_t)__builtin_popcountll(m));
            atomicMin(&sh[1], (uint32_t)(w0 + (int)__builtin_ctzll(m)));
            atomicMax(&sh[2], (uint32_t)(w0 + 63 - (int)__builtin_clzll(m)));
            atomicMax(&sh[3], (uint32_t)wcap);
        }
    }
    // linkStat[5] = the longest run of consecutive blocks that produced no output (capped at LINK_RUN_CAP + 1): how long
    // the serial part is when every run is walked by a wave of its own (k_decode_fixup_runs).  Only a run's first block
    // counts it.
    if (blk < a.nBlocks && a.result[blk] <= 0 && (blk == 0 || a.result[blk - 1] > 0)) {
        int n = 1;
        while (n <= LINK_RUN_CAP && blk + n < a.nBlocks && a.result[blk + n] <= 0) n++;
        atomicMax(&a.linkStat[5], (uint32_t)n);
        atomicAdd(&a.linkStat[6], 1u);                             // ... and how many runs there are (k_run_starts' list)
    }
    __syncthreads();
    if (tid == 0 && sh[0]) {
        atomicAdd(&a.linkStat[0], sh[0]);
        atomicMin(&a.linkStat[1], sh[1]);
        atomicMax(&a.linkStat[2], sh[2]);
        atomicMax(&a.linkStat[4], sh[3]);          // the largest such block sizes the second pass's scratch
    }
}
void launch_link_stat(const DecodeArgs &a, hipStream_t s)
{
    if (a.linkStat && a.nBlocks > 0)
        hipLaunchKernelGGL(k_link_stat, dim3((unsigned)((a.nBlocks + 1023) / 1024)), dim3(1024), 0, s, a);
}

// One wavefront per block, 4 blocks per 256-thread workgroup.
__global__ __launch_bounds__(256, 6) void k_decode_seq(DecodeArgs a)
{
    const int blk = uni((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
    if (blk >= a.nBlocks) return;
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    int r = read_block_header(a, blk, data, compLen, cap);
    if (r == 0)
        r = decode_block_seq(data, compLen, a.out + a.outOff[blk], cap, nullptr, 0, a.framed,
                             a.framed + a.framedLen);
    if (lane_id() == 0) a.result[blk] = r;
}

void launch_decode_seq(const DecodeArgs &a, hipStream_t s)
{
    if (a.nBlocks <= 0) return;
    const unsigned grid = (unsigned)((a.nBlocks + 3) / 4);
    hipLaunchKernelGGL(k_decode_seq, dim3(grid), dim3(256), 0, s, a);
    launch_link_stat(a, s);
}

// ---------------------------------------------------------------------------
// K2: encode
// ---------------------------------------------------------------------------
// MOD: table entries are positions modulo 64 Ki (encode_wave.hpp, tab_candidate): needed when positions run beyond
// 64 Ki -- blocks above 64 KiB, or a dictionary in front of the block (linked compression).  The table is the same
// size either way, so every block size runs at the same occupancy (a table of 32-bit positions would halve it).
#ifndef ENC_WAVES_PER_EU
#define ENC_WAVES_PER_EU 4
#endif
// PAIR: two dense windows per step (encode_wave.hpp): blocks of up to 64 KiB, independent or linked (measured: +6 % /
// +5 %); blocks above 64 KiB run one window per step (-9 % with pairs).
template <bool MOD, bool PAIR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ENC_WAVES_PER_EU, ENC_WAVES_PER_EU))) void k_encode(EncodeArgs a)
{
#ifndef ENC_LDS_PAD
#define ENC_LDS_PAD 0
#endif
    // positions + tags (encode_wave.hpp): 10 KiB, 16 waves per CU (the ENC_STAGE experiment's 256 bytes behind them make it 15)
    __shared__ __attribute__((aligned(16))) uint16_t table[ENC_TABLE_ENTRIES + ENC_LDS_PAD + ((PAIR && ENC_STAGE) ? 128 : 0)];
    const int blk = (int)blockIdx.x;
    const uint64_t off = a.srcOff ? a.srcOff[blk] : (uint64_t)blk * a.blockStride;
    const int n = a.srcLen ? a.srcLen[blk] : a.uniformLen;
    uint8_t *slot = a.slots + (size_t)blk * a.slotStride;
    int dictLen = 0;
    if (MOD && a.linked && (blk > 0 || a.lookBack > 0)) {
        // linked stream: the block before is the dictionary when it lies directly in front of this one
        const uint64_t poff = a.srcOff ? a.srcOff[blk - 1] : (uint64_t)(blk - 1) * a.blockStride;
        const int pn = a.srcLen ? a.srcLen[blk - 1] : a.uniformLen;
        if (pn > 0 && poff + (uint64_t)pn == off) dictLen = min(pn, 65536);
    }
    int c = 0;
    if (n >= 0 && (MOD || n <= 65536))
        c = encode_block_wave<uint16_t, MOD, false, PAIR>(a.src + off, n, slot + a.headerKind, a.accel, table, a.stats, dictLen);
    if (lane_id() == 0) {
        store_le32(slot, c);                                   // Internal/LZ4.hs:262
        if (a.headerKind == 8) store_le32(slot + 4, n);        // Internal/LZ4.hs:261
        a.framedLen[blk] = (c > 0) ? a.headerKind + c : 0;
    }
}

void launch_encode(const EncodeArgs &a, bool bigBlocks, hipStream_t s)
{
    if (a.nBlocks <= 0) return;
    const dim3 grid((unsigned)a.nBlocks), wg(64);
    if (bigBlocks) hipLaunchKernelGGL((k_encode<true, false>), grid, wg, 0, s, a);
    else if (a.linked) hipLaunchKernelGGL((k_encode<true, true>), grid, wg, 0, s, a);
#ifdef ENC_EXP_NOPAIR
    else hipLaunchKernelGGL((k_encode<false, false>), grid, wg, 0, s, a);
#else
    else hipLaunchKernelGGL((k_encode<false, true>), grid, wg, 0, s, a);
#endif
}

// ---------------------------------------------------------------------------
// K2, small batches: several waves per block (encode_wave.hpp, SEG).  One wavefront per block cannot be faster than
// one block (1.5 ms for 64 KiB), however empty the chip is: a call of 160 blocks -- the reference's own benchmark
// protocol, 10 MiB per file -- left 97 % of it idle, and 16 arrays of 640 KiB took 51 ms.  Here a block is cut into
// segments; wave (b, j) seeds its table from the bytes in front of segment j (what linked compression does between
// blocks) and writes sequence records; k_emit_seg then stitches a block's lists into one valid LZ4 block: a segment's
// trailing literals simply become the first literals of the next segment's first sequence.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ENC_WAVES_PER_EU, ENC_WAVES_PER_EU))) void k_encode_seg(EncodeSegArgs a)
{
    __shared__ uint16_t table[ENC_TABLE_ENTRIES + ENC_LDS_PAD];
    const int blk = (int)(blockIdx.x / (unsigned)a.segs), j = (int)(blockIdx.x % (unsigned)a.segs);
    const uint64_t off = a.e.srcOff ? a.e.srcOff[blk] : (uint64_t)blk * a.e.blockStride;
    const int n = a.e.srcLen ? a.e.srcLen[blk] : a.e.uniformLen;
    uint32_t count = 0;
    if (n > 0) {
        const int s0 = min(n, j * a.segLen), s1 = (j == a.segs - 1) ? n : min(n, (j + 1) * a.segLen);
        if (s1 > s0) {
            SegOut so;
            so.list = a.lists + (size_t)blk * a.listStride + (size_t)(s0 / 4 + j);     // a segment has at most len/4 + 1 records
            so.count = 0;
            const int dictLen = min(s0, 65536);
            so.base = s0 - dictLen;
            so.last = s1 >= n;
            so.tail = n - s1;
            (void)encode_block_wave<uint16_t, true, true>(a.e.src + off + s0, s1 - s0, nullptr, a.e.accel, table, a.e.stats, dictLen, &so);
            count = so.count;
        }
    }
    if (lane_id() == 0) a.segCount[(size_t)blk * a.segs + j] = count;
}

// Emission of a segmented block, every segment by a wave of its own, in two steps: k_seg_sizes measures what each
// segment's records come to in bytes (a segment's first sequence takes its literals from where the last sequence
// BEFORE the segment ends), k_emit_seg places every segment behind the ones in front of it.  (One wave per block did
// this in round 3's first version: 8 ms for a 4 MiB block.)
__device__ __forceinline__ int seg_prev_end(const EncodeSegArgs &a, int blk, int j, int n)
{
    // end of the last sequence in front of segment j (0 when there is none)
    for (int i = j - 1; i >= 0; i--) {
        const int cnt = (int)a.segCount[(size_t)blk * a.segs + i];
        if (cnt > 0) {
            const int s0 = min(n, i * a.segLen);
            int start, len, mo;
            seg_unpack(a.lists[(size_t)blk * a.listStride + (size_t)(s0 / 4 + i) + (size_t)(cnt - 1)], start, len, mo);
            return start + len;
        }
    }
    return 0;
}

__global__ __launch_bounds__(64) void k_seg_sizes(EncodeSegArgs a)
{
    const int blk = (int)(blockIdx.x / (unsigned)a.segs), j = (int)(blockIdx.x % (unsigned)a.segs);
    const int lane = lane_id();
    const int n = a.e.srcLen ? a.e.srcLen[blk] : a.e.uniformLen;
    const int cnt = (n > 0) ? (int)a.segCount[(size_t)blk * a.segs + j] : 0;
    const int s0 = min(max(n, 0), j * a.segLen);
    const uint64_t *list = a.lists + (size_t)blk * a.listStride + (size_t)(s0 / 4 + j);
    int prevEnd = (n > 0) ? seg_prev_end(a, blk, j, n) : 0;
    const int prev0 = prevEnd;
    uint32_t bytes = 0;
    for (int i0 = 0; i0 < cnt; i0 += LZ4_WAVE) {
        const int k = min(LZ4_WAVE, cnt - i0);
        int start = 0, len = 0, mo = 0;
        if (lane < k) seg_unpack(list[i0 + lane], start, len, mo);
        const int end = start + len;
        int qPrev = __builtin_amdgcn_update_dpp(prevEnd, end, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (lane == 0) qPrev = prevEnd;
        const uint32_t lit = (uint32_t)(start - qPrev), mc = (uint32_t)(len - LZ4_MINMATCH);
        const uint32_t esz = (lane < k) ? 1u + lit + ext_len_bytes(lit) + 2u + ext_len_bytes(mc) : 0u;
        bytes += (uint32_t)__builtin_amdgcn_readlane(enc_scan_incl((int)esz), 63);
        prevEnd = __builtin_amdgcn_readlane(end, k - 1);
    }
    if (lane == 0) {
        a.segBytes[(size_t)blk * a.segs + j] = bytes;
        a.segPrevEnd[(size_t)blk * a.segs + j] = prev0;
    }
}

__global__ __launch_bounds__(64) void k_emit_seg(EncodeSegArgs a)
{
    const int blk = (int)(blockIdx.x / (unsigned)a.segs), j = (int)(blockIdx.x % (unsigned)a.segs);
    const int lane = lane_id();
    const uint64_t off = a.e.srcOff ? a.e.srcOff[blk] : (uint64_t)blk * a.e.blockStride;
    const int n = a.e.srcLen ? a.e.srcLen[blk] : a.e.uniformLen;
    uint8_t *slot = a.e.slots + (size_t)blk * a.e.slotStride;
    uint8_t *op0 = slot + a.e.headerKind;
    const uint8_t *src = a.e.src + off;
    const bool lastSeg = j == a.segs - 1;
    if (n <= 0) {
        if (lastSeg && lane == 0) {
            int c = 0;
            if (n == 0) { op0[0] = 0; c = 1; }                 // cbits/lz4.c:1263-1273: empty input -> single 0 token
            store_le32(slot, c);
            if (a.e.headerKind == 8) store_le32(slot + 4, n);
            a.e.framedLen[blk] = (c > 0) ? a.e.headerKind + c : 0;
        }
        return;
    }
    // where this segment's bytes go: behind the segments in front of it (at most 64: one per lane)
    const uint32_t mine = (lane < j) ? a.segBytes[(size_t)blk * a.segs + lane] : 0u;
    const uint32_t before = (uint32_t)__builtin_amdgcn_readlane(enc_scan_incl((int)mine), 63);
    uint8_t *op = op0 + before;
    const int cnt = (int)a.segCount[(size_t)blk * a.segs + j];
    const int s0 = min(n, j * a.segLen);
    const uint64_t *list = a.lists + (size_t)blk * a.listStride + (size_t)(s0 / 4 + j);
    int prevEnd = a.segPrevEnd[(size_t)blk * a.segs + j];
    for (int i0 = 0; i0 < cnt; i0 += LZ4_WAVE) {
        const int k = min(LZ4_WAVE, cnt - i0);
        int start = 0, len = 0, mo = 0;
        if (lane < k) seg_unpack(list[i0 + lane], start, len, mo);
        const int end = start + len;
        // my literals start where the sequence before me ends (lane 0: the one before this batch)
        int qPrev = __builtin_amdgcn_update_dpp(prevEnd, end, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (lane == 0) qPrev = prevEnd;
        op = emit_sequences(src, op, qPrev, start, len, mo, k);
        prevEnd = __builtin_amdgcn_readlane(end, k - 1);
    }
    if (!lastSeg) return;
    // ---- the block's last segment: last literals (:1204-1231), header ----
    const uint32_t lastRun = (uint32_t)(n - prevEnd);
    uint8_t *tok = op++;
    if (lane == 0) *tok = (uint8_t)(min(lastRun, 15u) << 4);
    if (lastRun >= 15) op = emit_ext_len(op, lastRun - 15);
    wave_copy_bytes(op, src + prevEnd, lastRun);
    op += lastRun;
    const int c = (int)(op - op0);
    if (lane == 0) {
        store_le32(slot, c);                                   // Internal/LZ4.hs:262
        if (a.e.headerKind == 8) store_le32(slot + 4, n);      // Internal/LZ4.hs:261
        a.e.framedLen[blk] = (c > 0) ? a.e.headerKind + c : 0;
    }
}

void launch_encode_seg(const EncodeSegArgs &a, hipStream_t s)
{
    if (a.e.nBlocks <= 0) return;
    const dim3 grid((unsigned)a.e.nBlocks * (unsigned)a.segs);
    hipLaunchKernelGGL(k_encode_seg, grid, dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_seg_sizes, grid, dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_emit_seg, grid, dim3(64), 0, s, a);
}

// ---------------------------------------------------------------------------
// K3: scan + ragged copy
// ---------------------------------------------------------------------------

// Exclusive scan of n int32 sizes into n+1 uint64 offsets; one 1024-thread workgroup.
// (n is the block count of a batch: at most a few million.)
__global__ __launch_bounds__(1024) void k_scan_u64(const int32_t *sizes, int n, uint64_t *offs)
{
    __shared__ uint64_t part[1024];
    const int t = (int)threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    uint64_t sum = 0;
    for (int i = lo; i < hi; i++) sum += (uint64_t)(uint32_t)max(sizes[i], 0);
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint64_t v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = part[t] - sum;
    for (int i = lo; i < hi; i++) { offs[i] = run; run += (uint64_t)(uint32_t)max(sizes[i], 0); }
    if (t == 1023) offs[n] = part[1023];
}

// Copy n bytes with a 256-thread workgroup; dst gets 16-byte aligned stores in
// the body, src is read with (possibly unaligned) 16-byte loads.
__device__ __forceinline__ void wg_copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t n)
{
    const uint32_t t = threadIdx.x, T = blockDim.x;
    uint64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
    if (head > n) head = n;
    if (t < head) dst[t] = src[t];
    const uint64_t body = (n - head) >> 4;
    uint4 *d16 = (uint4 *)(dst + head);
    const uint8_t *s16 = src + head;
    for (uint64_t i = t; i < body; i += T) {
        uint4 v;
        __builtin_memcpy(&v, s16 + (i << 4), 16);   // unaligned 16-byte global load
        d16[i] = v;
    }
    const uint64_t done = head + (body << 4);
    if (done + t < n) dst[done + t] = src[done + t];
}

__global__ __launch_bounds__(256) void k_copy_slots(const uint8_t *slots, size_t slotStride,
                                                    const int32_t *framedLen, const uint64_t *denseOff,
                                                    uint8_t *dense, uint64_t denseCap)
{
    const int blk = (int)blockIdx.x;
    const int n = framedLen[blk];
    const uint64_t at = denseOff[blk];
    if (n > 0 && at + (uint64_t)n <= denseCap) wg_copy_bytes(dense + at, slots + (size_t)blk * slotStride, (uint64_t)n);
}

void launch_compact(const uint8_t *slots, size_t slotStride, const int32_t *framedLen, int nBlocks,
                    uint8_t *dense, size_t denseCap, uint64_t *denseOff, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s, framedLen, nBlocks, denseOff);
    if (nBlocks > 0)
        hipLaunchKernelGGL(k_copy_slots, dim3((unsigned)nBlocks), dim3(256), 0, s, slots, slotStride,
                           framedLen, denseOff, dense, (uint64_t)denseCap);
}

__global__ __launch_bounds__(256) void k_interleave(const uint8_t *local, const uint64_t *localOff, int rank,
                                                    int nRanks, uint8_t *global, const uint64_t *globalOff)
{
    const int j = (int)blockIdx.x;
    const uint64_t n = localOff[j + 1] - localOff[j];
    wg_copy_bytes(global + globalOff[(size_t)j * nRanks + rank], local + localOff[j], n);
}

void launch_interleave(const uint8_t *local, const uint64_t *localOff, int nLocal, int rank, int nRanks,
                       uint8_t *global, const uint64_t *globalOff, hipStream_t s)
{
    if (nLocal > 0)
        hipLaunchKernelGGL(k_interleave, dim3((unsigned)nLocal), dim3(256), 0, s, local, localOff, rank,
                           nRanks, global, globalOff);
}

// Header gather for the output index: sizes[i] = uncompressed size of block i (0 if unreadable).
__global__ __launch_bounds__(256) void k_header_sizes(const uint8_t *framed, uint64_t framedLen,
                                                      const uint64_t *blockOff, int nBlocks, int headerKind,
                                                      int fixedUncomp, int32_t *sizes)
{
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= nBlocks) return;
    int u = fixedUncomp;
    if (headerKind == 8) {
        const uint64_t off = blockOff[i];
        u = (off + 8 <= framedLen) ? load_le32(framed + off + 4) : 0;
    }
    sizes[i] = max(u, 0);
}

void launch_index(const uint8_t *framed, uint64_t framedLen, const uint64_t *blockOff, int nBlocks,
                  int headerKind, int fixedUncomp, int32_t *scratchSizes, uint64_t *outOff, hipStream_t s)
{
    if (nBlocks > 0)
        hipLaunchKernelGGL(k_header_sizes, dim3((unsigned)((nBlocks + 255) / 256)), dim3(256), 0, s, framed,
                           framedLen, blockOff, nBlocks, headerKind, fixedUncomp, scratchSizes);
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s, scratchSizes, nBlocks, outOff);
}

// ---------------------------------------------------------------------------
// Synthetic inputs: xorshift64* seeded per block by splitmix64.
// One thread per block; setup only, never timed.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t xs64(uint64_t &st)
{
    uint64_t x = st;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    st = x;
    return x * 0x2545F4914F6CDD1DULL;
}

__global__ __launch_bounds__(64) void k_generate(int kind, uint8_t *dst, int blockLen, int nBlocks,
                                                 uint64_t firstBlock, uint64_t blockStep, uint32_t litMax,
                                                 uint32_t offMax)
{
    const int b = (int)(blockIdx.x * 64u + threadIdx.x);
    if (b >= nBlocks) return;
    uint8_t *out = dst + (size_t)b * (size_t)blockLen;
    const uint64_t index = firstBlock + (uint64_t)b * blockStep;
    const uint32_t n = (uint32_t)blockLen;
    if (kind == 0) {
        uint64_t st = splitmix64(0x9E3779B97F4A7C15ULL ^ index);
        if (!st) st = 1;
        for (uint32_t i = 0; i < n;) {
            uint64_t r = xs64(st);
            for (int k = 0; k < 8 && i < n; k++, i++) out[i] = (uint8_t)(r >> (8 * k));
        }
    } else if (kind == 1) {
        uint64_t st = splitmix64(0x9E3779B97F4A7C15ULL ^ index);
        if (!st) st = 1;
        uint32_t pos = 0;
        while (pos < n) {
            uint32_t L = 1 + (uint32_t)(xs64(st) % litMax);
            for (uint32_t i = 0; i < L && pos < n; i++) out[pos++] = (uint8_t)(32 + xs64(st) % 64);
            if (pos >= n) break;
            uint32_t M = 4 + (uint32_t)(xs64(st) % 61);
            uint32_t lim = (pos < offMax) ? pos : offMax;
            uint32_t o = 1 + (uint32_t)(xs64(st) % lim);
            for (uint32_t i = 0; i < M && pos < n; i++, pos++) out[pos] = out[pos - o];
        }
    } else {
        uint64_t st = splitmix64(0x9E3779B97F4A7C15ULL ^ (index ^ 0x7465787400000000ULL));
        if (!st) st = 1;
        uint32_t pos = 0;
        while (pos < n) {
            uint64_t r = xs64(st);
            uint32_t a = (uint32_t)(r & 4095), bq = (uint32_t)((r >> 12) & 4095);
            uint32_t c = (uint32_t)((r >> 29) & 4095), d = (uint32_t)((r >> 41) & 4095);
            uint32_t w = (((a * bq) >> 12) * ((c * d) >> 12)) >> 12;
            uint64_t h = splitmix64(0x776F7264ULL + w);
            uint32_t len = 2 + (uint32_t)(h & 7);
            uint32_t sep = (uint32_t)((r >> 24) & 31);
            for (uint32_t j = 0; j < len && pos < n; j++)
                out[pos++] = (uint8_t)('a' + ((h >> (3 + 5 * j)) & 31) % 26);
            if (pos < n) out[pos++] = (sep == 0) ? '\n' : (sep == 1) ? ',' : ' ';
            if (sep == 1 && pos < n) out[pos++] = ' ';
        }
    }
}

void launch_generate(int kind, uint8_t *dst, int blockLen, int nBlocks, uint64_t firstBlock,
                     uint64_t blockStep, uint32_t litMax, uint32_t offMax, hipStream_t s)
{
    if (nBlocks > 0)
        hipLaunchKernelGGL(k_generate, dim3((unsigned)((nBlocks + 63) / 64)), dim3(64), 0, s, kind, dst,
                           blockLen, nBlocks, firstBlock, blockStep, litMax, offMax);
}

// Lane-parallel decoder (decode_par.hpp): one wavefront (= one workgroup) per block.
#ifdef PAR_WAVES_MAX
#define PAR_OCC __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(PAR_WAVES, PAR_WAVES_MAX)))
#else
#define PAR_OCC __launch_bounds__(64, PAR_WAVES)
#endif
template <bool STATS>
__global__ PAR_OCC void k_decode_par(DecodeArgs a, unsigned long long *stats)
{
    __shared__ ParLds lds;
    const int blk = (int)blockIdx.x;
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    int r = read_block_header(a, blk, data, compLen, cap);
    if (r == 0)
        r = decode_block_par<STATS, false>(data, compLen, a.out + a.outOff[blk], cap, nullptr, 0, a.framed,
                                    a.framed + a.framedLen, lds, stats);
    if (lane_id() == 0) a.result[blk] = r;
}

void launch_decode_par(const DecodeArgs &a, unsigned long long *stats, hipStream_t s)
{
    if (a.nBlocks <= 0) return;
    if (stats)
        hipLaunchKernelGGL(k_decode_par<true>, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a, stats);
    else
        hipLaunchKernelGGL(k_decode_par<false>, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a, stats);
    launch_link_stat(a, s);
}

// The blocks the workgroup-per-block decoder left behind (result CU_REDO), by the lane-parallel decoder.  (A kernel of its
// own, not a template parameter of k_decode_par: that changed the headline kernel's register allocation.)
__global__ PAR_OCC void k_decode_par_redo(DecodeArgs a)
{
    __shared__ ParLds lds;
    const int blk = (int)blockIdx.x;
    if (uni(a.result[blk]) != CU_REDO) return;
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    int r = read_block_header(a, blk, data, compLen, cap);
    if (r == 0)
        // (LIST with an empty list: the same decoder as k_decode_par's, but an instantiation of its own -- a second user of
        // k_decode_par's instantiation turns that kernel's inlined decoder into a call)
        r = decode_block_par<false, false, false, true>(data, compLen, a.out + a.outOff[blk], cap, nullptr, 0, a.framed,
                                                        a.framed + a.framedLen, lds, nullptr, nullptr, nullptr, 0);
    if (lane_id() == 0) a.result[blk] = r;
}

// Workgroup-per-block decoder (decode_cu.hpp): sixteen wavefronts per block, for calls that do not fill the GPU.
__global__ __launch_bounds__(CU_THREADS) void k_decode_cu(DecodeArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[CU_LDS_BYTES];
    const int blk = (int)blockIdx.x;
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    int r = uni(read_block_header(a, blk, data, compLen, cap));
    // A block that hardly compresses is literal runs of hundreds of bytes: every one of them ends a segment (the parse follows two
    // extension bytes) and is copied by one wave, which is what the lane-parallel decoder does without the segments' fixed costs
    // (160 blocks of 64 KiB, ms, wavefront / workgroup form: ratio 1.00: 0.045 / 0.070; text at acceleration 64, ratio 1.01: 0.53 / 1.44;
    // lzsynth at 64, 1.03: 0.45 / 0.83; text at 16, ratio 1.14: 0.45 / 0.29 -- from there on the workgroup form is the faster one).
    if (r == 0 && a.cuBail && (int64_t)uni(compLen) * 16 > (int64_t)uni(cap) * 15) r = CU_REDO;
    else if (r == 0)
        r = decode_block_cu<false>(data, uni(compLen), a.out + a.outOff[blk], uni(cap), nullptr, 0, a.framed, a.framed + a.framedLen, lds,
                                   a.cuDbg ? a.cuDbg + 16 * (size_t)blk : nullptr, a.cuBail != 0);
    if (threadIdx.x == 0) a.result[blk] = r;
    // A linked call of big blocks (a.cuRes armed by the caller, launch_cu_linked below): a block that did not decode on its own is,
    // as a rule, one that needs its dictionary -- pass 1 of that path (the block against 64 KiB of zeros) follows at once, while the
    // stream's first block, which decodes on its own, is still at work.  (The redo launch still reports the exact code in result[].)
    // (The call's first block, when blocks lie in front of the call -- a later group of a host call, a.lookBack --, has a dictionary that
    // is FINAL: the last 64 KiB of the block in front of it; it is right after this one decode and is not looked at again.)
    const bool prevFinal = blk == 0 && a.lookBack > 0 && a.cuRes && uni(a.result[-1]) >= 65536;
    if (a.cuRes && (blk > 0 || prevFinal) && r == CU_REDO && !(a.cuBail && (int64_t)uni(compLen) * 16 > (int64_t)uni(cap) * 15)) {
        __syncthreads();
        const uint8_t *dict = prevFinal ? a.out + a.outOff[-1] + (size_t)uni(a.result[-1]) - 65536u : a.zeroPage;
        const int r2 = decode_block_cu<true>(data, uni(compLen), a.out + a.outOff[blk], uni(cap), dict, 65536u, a.framed,
                                            a.framed + a.framedLen, lds, nullptr, false, 0);
        if (threadIdx.x == 0) {
            a.cuRes[blk] = r2;
            if (r2 < 0 || (r2 < 65536 && blk + 1 < a.nBlocks)) atomicAdd(&a.cuFlags[1], 1u);   // an error, CU_REDO, or a block too short to be a whole dictionary
        }
    }
}

// ---- big linked blocks (a stream of BlockMax1MB / BlockMax4MB blocks, Config.hs:109-116, written with a dictionary carried from block
// to block, cbits/lz4.c:1608-1636): the workgroup form with a GUESSED dictionary.  A block of 1 MiB forgets a wrong dictionary long
// before its end (text: after 5 to 12 times 64 KiB), so its last 64 KiB -- all its successor can see of it -- come out right even when its
// own dictionary was wrong.  Pass 1 decodes every dependent block against 64 KiB of zeros, every later pass against a snapshot of what its
// predecessor's last 64 KiB were after the pass before, and when a pass changes no snapshot, every block has been decoded against its
// predecessor's final bytes: by induction from the stream's first block, which needs no dictionary, all of them are right.  The caller
// (api.cpp) bounds the passes and falls back to the pointer pass; results go to a.cuRes and are published at the end.
__global__ __launch_bounds__(CU_THREADS) void k_decode_cu_linked(DecodeArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[CU_LDS_BYTES];
    const int blk = (int)blockIdx.x;
    if (uni(a.result[blk]) >= 0) return;                                 // decoded on its own in the first pass: final
    const uint8_t *dict = nullptr;
    if (blk == 0) {
        // a first block that needs a dictionary: the call's own (dict0) is not this path's; the block in front of the call (a later
        // group of a host call) is final, and this block is decoded against its end once -- by the first launch, or here in pass 1
        const bool prevFinal = a.lookBack > 0 && uni(a.result[-1]) >= 65536;
        if (prevFinal && uni(a.cuRes[0]) >= 65536) return;
        if (!prevFinal || a.cuPass != 1) { if (threadIdx.x == 0) atomicAdd(&a.cuFlags[1], 1u); return; }
        dict = a.out + a.outOff[-1] + (size_t)uni(a.result[-1]) - 65536u;
    } else {
        if (a.cuPass > 2 && uni(a.cuFlags[2 + blk - 1]) == 0u) return;  // the dictionary it was decoded against last time still stands
        dict = a.cuPass == 1 ? a.zeroPage : a.cuSnap + (size_t)(blk - 1) * 65536u;
    }
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    int r = uni(read_block_header(a, blk, data, compLen, cap));
    if (r == 0)
        r = decode_block_cu<true>(data, uni(compLen), a.out + a.outOff[blk], uni(cap), dict, 65536u, a.framed, a.framed + a.framedLen, lds, nullptr, false,
                                  (blk > 0 && a.cuPass > 1 && uni(a.cuRes[blk]) >= 65536) ? uni(a.cuRes[blk]) : 0);      // (from the second pass on: stop where the bytes repeat the pass before)
    if (threadIdx.x == 0) {
        a.cuRes[blk] = r;
        if (r < 0 || (r < 65536 && blk + 1 < a.nBlocks)) atomicAdd(&a.cuFlags[1], 1u);      // an error, CU_REDO, or a block too short to be a whole dictionary
    }
}

// the last 64 KiB of every block -> its snapshot; [2 + k] = whether that changed the snapshot, [0] = how many did
__global__ __launch_bounds__(1024) void k_cu_tails(DecodeArgs a)
{
    const int blk = (int)blockIdx.x;
    if (blk + 1 >= a.nBlocks) return;                                    // (nobody looks at the last block's)
    const int32_t r = a.result[blk] >= 0 ? a.result[blk] : a.cuRes[blk];
    __shared__ uint32_t diff;
    if (threadIdx.x == 0) diff = 0u;
    __syncthreads();
    uint32_t d = 0u;
    if (r >= 65536) {
        const uint8_t *tail = a.out + a.outOff[blk] + (size_t)r - 65536u;
        uint8_t *snap = a.cuSnap + (size_t)blk * 65536u;
        for (uint32_t i = threadIdx.x * 16u; i < 65536u; i += 1024u * 16u) {
            const par_v4 v = *(const par_v4u *)(tail + i), o = *(const par_v4 *)(snap + i);
            d |= (v.x ^ o.x) | (v.y ^ o.y) | (v.z ^ o.z) | (v.w ^ o.w);
            *(par_v4 *)(snap + i) = v;
        }
    }
    if (d) atomicOr(&diff, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        // (a block that is no whole dictionary -- shorter than 64 KiB, or failed -- in front of a block that needs one: not this path's)
        if (r < 65536 && a.result[blk + 1] < 0) atomicAdd(&a.cuFlags[1], 1u);
        const uint32_t ch = (diff != 0u || a.cuPass == 1) ? 1u : 0u;
        a.cuFlags[2 + blk] = ch;
        if (ch) atomicAdd(&a.cuFlags[0], 1u);
    }
}

__global__ __launch_bounds__(256) void k_cu_publish(DecodeArgs a)
{
    const int blk = (int)(blockIdx.x * 256u + threadIdx.x);
    if (blk < a.nBlocks && a.result[blk] < 0) a.result[blk] = a.cuRes[blk];
}

void launch_cu_linked(const DecodeArgs &a, bool decode, hipStream_t s)
{
    if (a.nBlocks <= 0) return;
    hipMemsetAsync(a.cuFlags, 0, 4, s);
    if (decode) hipLaunchKernelGGL(k_decode_cu_linked, dim3((unsigned)a.nBlocks), dim3(CU_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_cu_tails, dim3((unsigned)a.nBlocks), dim3(1024), 0, s, a);
}

void launch_cu_publish(const DecodeArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_cu_publish, dim3((unsigned)((a.nBlocks + 255) / 256)), dim3(256), 0, s, a);
}

void launch_decode_cu(const DecodeArgs &a, hipStream_t s)
{
    if (a.nBlocks <= 0) return;
    hipLaunchKernelGGL(k_decode_cu, dim3((unsigned)a.nBlocks), dim3(CU_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_decode_par_redo, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a);
    launch_link_stat(a, s);
}

#ifdef MI355LZ4_EXPERIMENTS
// ---- token lists: the parse as a pass of its own (experiment of round 4) ----
// One LANE per block walks the block's token chain (cbits/lz4.c:1801-1854: token, literal length, offset, match length)
// and writes the compressed size of every sequence as one byte; 0 ends the list (a length that does not fit a byte, a
// 255-run, or the end of the block's plain part).  This first form reads the stream byte by byte from global memory.
__global__ __launch_bounds__(64) void k_walk_tokens(DecodeArgs a)
{
    const int blk = (int)(blockIdx.x * 64u + threadIdx.x);
    if (blk >= a.nBlocks) return;
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    int n = 0;
    if (read_block_header(a, blk, data, compLen, cap) == 0) {
        const LZ4_GLOBAL uint8_t *p = as_global(data);
        LZ4_GLOBAL uint8_t *out = as_global(a.tokList + (a.blockOff[blk] >> 1));
        const int limit = ((compLen + a.headerKind) >> 1) - 1;      // the list's room: half the block's framed bytes
        int ip = 0;
        while (n < limit) {
            const int tp = ip;
            if (ip + 1 > compLen) break;
            const uint32_t t = p[ip++];
            uint32_t lit = t >> 4;
            if (lit == 15u) {
                if (ip >= compLen) break;
                const uint32_t b = p[ip++];
                if (b == 255u) break;
                lit += b;
            }
            ip += (int)lit;
            if (ip + 2 > compLen) break;                         // the last sequence has no match: not listed
            ip += 2;
            if ((t & 15u) == 15u) {
                if (ip >= compLen) break;
                const uint32_t b = p[ip++];
                if (b == 255u) break;
            }
            const int d = ip - tp;
            if (d > 255) break;
            out[n++] = (uint8_t)d;
        }
    }
    a.tokCnt[blk] = n;
}

template <bool STATS>
__global__ PAR_OCC void k_decode_tok(DecodeArgs a, unsigned long long *stats)
{
    __shared__ ParLds lds;
    const int blk = (int)blockIdx.x;
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    int r = read_block_header(a, blk, data, compLen, cap);
    if (r == 0)
        r = decode_block_par<STATS, false, false, true>(data, compLen, a.out + a.outOff[blk], cap, nullptr, 0, a.framed,
                                                        a.framed + a.framedLen, lds, stats, nullptr,
                                                        a.tokList + (a.blockOff[blk] >> 1), uni(a.tokCnt[blk]));
    if (lane_id() == 0) a.result[blk] = r;
}

void launch_decode_tok(const DecodeArgs &a, hipStream_t s)
{
    if (a.nBlocks <= 0) return;
    hipLaunchKernelGGL(k_walk_tokens, dim3((unsigned)((a.nBlocks + 63) / 64)), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_decode_tok<false>, dim3((unsigned)a.nBlocks), dim3(64), 0, s, a, (unsigned long long *)nullptr);
    launch_link_stat(a, s);
}
#endif  // MI355LZ4_EXPERIMENTS

// Linked streams (reference semantics of LZ4_decompress_safe_continue with every
// block in its own allocation, cbits/lz4.c:2347-2355): block i may reference the
// output of the last block before it IN ITS STREAM that decoded to > 0 bytes.  A
// block that decodes standalone never consulted a dictionary, so its standalone
// result IS its linked result; only blocks whose standalone decode failed are
// re-decoded here, in stream order, with the dictionary in force.  The chain
// inside one stream is serial (block i needs the bytes of block i-1), so one
// wavefront walks each stream with the lane-parallel decoder; independent
// streams run side by side.  
__global__ PAR_OCC void k_decode_fixup_linked(DecodeArgs a)
{
    if (a.asyncGate && a.linkStat[0] == 0u) return;     // asynchronous linked decode: the first pass found nothing to do
    __shared__ ParLds lds;
    const int sIdx = (int)blockIdx.x;
    if (a.ptrBad && !a.ptrBad[sIdx]) return;                        // the data-parallel pass has done this stream
    int b0 = 0, b1 = a.nBlocks;
    const uint8_t *dict = nullptr;
    uint32_t dictLen = 0;
    if (a.streamFirst) {
        b0 = min(max(uni(a.streamFirst[sIdx]), 0), a.nBlocks);
        b1 = min(max(uni(a.streamFirst[sIdx + 1]), b0), a.nBlocks);
    } else if (a.dict0) {
        dict = a.dict0; dictLen = a.dict0Len;
    }
    for (int blk = b0; blk < b1; blk++) {
        int r = uni(a.result[blk]);
        uint8_t *dst = a.out + a.outOff[blk];
        if (r < 0 && r > -0x7F000000 && dictLen > 0) {   // codec error (not a header rejection)
            const uint8_t *data = nullptr;
            int compLen = 0, cap = 0;
            r = read_block_header(a, blk, data, compLen, cap);
            if (r == 0)
                r = decode_block_par<false, true>(data, compLen, dst, cap, dict, dictLen, a.framed,
                                            a.framed + a.framedLen, lds, nullptr);
            r = uni(r);
            if (lane_id() == 0) a.result[blk] = r;
        }
        if (r > 0) { dict = dst; dictLen = (uint32_t)r; }          // :2331-2333, :2353-2355
        wave_fence();       // (the next block reads this one through the pipeline that wrote it: no write-back, see k_runin_decode)
    }
}

// One stream in which FEW blocks need their dictionary (a reference-written stream of data whose matches rarely reach
// back into the block before: 229 of 16 384 blocks of the bench's lzsynth sample): every maximal run of blocks without
// output is walked by a wavefront of its own with the exact lane-parallel decoder and the previous output as external
// dictionary -- the runs are independent of each other because the block in front of a run is final.  The pointer pass
// would write and chase four bytes of pointer per output byte of the whole SPAN between the first and the last dependent
// block for them (2 ms for that sample; this: one block's latency per block of the longest run).  Chosen by the host
// when the longest run is short (linkStat[5]); same dictionary rules as k_decode_fixup_regions (:2331-2333, :2347-2355).
// The runs' first blocks are taken from the FIRST pass's results before any walker has changed them (k_run_starts: a
// list).  Round 4 let every wave decide "am I a run start" from result[blk - 1] inside the walking launch: a wave
// dispatched late could see the block in front of it already fixed by its run's walker, take itself for a run start and
// walk the same blocks a second time, racing the first walker.  (The list also shrinks the grid to one wave per run.)
__global__ __launch_bounds__(256) void k_run_starts(DecodeArgs a)
{
    const int blk = a.segFirst + (int)(blockIdx.x * 256u + threadIdx.x);
    if (blk >= a.segEnd || a.result[blk] > 0) return;
    if (blk != a.segFirst && a.result[blk - 1] <= 0) return;
    const int i = atomicAdd(&a.runList[0], 1);
    if (i < a.runCap) a.runList[1 + i] = blk;
}

__global__ PAR_OCC void k_decode_fixup_runs(DecodeArgs a)
{
    __shared__ ParLds lds;
    int blk = a.segFirst;                                           // (no list: one run, the legacy face's single block)
    if (a.runList) {
        if ((int)blockIdx.x >= min(uni(a.runList[0]), a.runCap)) return;
        blk = uni(a.runList[1 + blockIdx.x]);
    } else if (blockIdx.x != 0) return;
    if (blk >= a.segEnd || uni(a.result[blk]) > 0) return;
    const uint8_t *dict = nullptr;
    uint32_t dictLen = 0;
    if (a.dict0) { dict = a.dict0; dictLen = a.dict0Len; }
    for (int j = blk - 1; j >= -a.lookBack; j--) {
        const int rj = uni(a.result[j]);
        if (rj > 0) { dict = a.out + a.outOff[j]; dictLen = (uint32_t)rj; break; }
    }
    // (the blocks behind a run's end decoded in the first pass: no walker writes their results, reading them is safe)
    for (int f = blk; f < a.segEnd; f++) {
        int r = uni(a.result[f]);
        if (r > 0) break;                                           // the run is over
        uint8_t *dst = a.out + a.outOff[f];
        if (is_codec_error(r) && dictLen > 0) {
            const uint8_t *data = nullptr;
            int compLen = 0, cap = 0;
            r = read_block_header(a, f, data, compLen, cap);
            if (r == 0)
                r = decode_block_par<false, true>(data, compLen, dst, cap, dict, dictLen, a.framed,
                                                  a.framed + a.framedLen, lds, nullptr);
            r = uni(r);
            wave_fence();                                           // (one wave per run, nobody else looks before the launch ends)
            if (lane_id() == 0) a.result[f] = r;
        }
        if (r > 0) { dict = dst; dictLen = (uint32_t)r; }
    }
}

// the dictionary in force in front of block b0 as the first pass's results have it (blocks in front of a segment are final)
__device__ __forceinline__ void dict_before(const DecodeArgs &a, int b0, const uint8_t *&dict, uint32_t &dictLen)
{
    dict = nullptr; dictLen = 0;
    if (a.dict0) { dict = a.dict0; dictLen = a.dict0Len; }
    for (int j = b0 - 1; j >= -a.lookBack; j--) {
        const int rj = uni(a.result[j]);
        if (rj > 0) { dict = a.out + a.outOff[j]; dictLen = (uint32_t)rj; break; }
    }
}

// (a piece is a serial chain of block decodes: what counts is one wave's speed, and the dictionary form of the decoder
// spills at the 96 registers that five waves per SIMD allow -- these kernels take 128)
#ifndef RUNIN_WAVES
#define RUNIN_WAVES 4
#endif
#define RUNIN_OCC __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RUNIN_WAVES, RUNIN_WAVES)))

// ---------------------------------------------------------------------------------------------------------------------
// Long linked streams: RUN-IN DECODE (round 5).  In a stream written by the reference's compressor every block needs the
// block before it, but not much of it: on text a third of a block's bytes derive -- through chains of matches -- from
// the previous block, 5 % from the one before that, 1 % from the third, and the 6th to 12th block is the first that has
// no such byte left (scripts/runin_sim.py, byte-exact; tracked per MATCH, as the tolerant pass has to, the dependent share
// stays at 76 % in every block, which is why that pass hands practically the whole stream to the pointer machinery).
// So the stream is cut into PIECES of consecutive blocks, one wavefront per piece, and a piece does not start at its
// first block but runIn blocks in front of it, with 64 KiB of zeros as the dictionary of the block it starts at; it
// decodes those blocks with the ordinary lane-parallel decoder, every block with the one before it as its dictionary,
// into a two-block ring of scratch, and by the time it reaches its own blocks the dictionary it carries is, as a rule,
// the true one.  Whether it is needs no second decode: the block in front of a piece is decoded twice anyway -- by the
// piece in front (into the caller's buffer) and by this piece's run-in (into the ring) -- and k_runin_verify compares
// the two.  If they are equal, the piece's blocks are what the piece in front's last block makes of them, and by
// induction from the first piece (which has its true dictionary) the whole stream is exact.  A call's serial chain is
// runIn + piece blocks.  (The first half of round 5 decoded every piece TWICE, with 0x00 and 0xFF as stand-ins, and
// re-decoded what differed: exact as well, but a chain of piece + the longest re-decoded prefix with two waves per
// piece -- 1 GiB of text 12.5 ms against 9.0, 4 GiB 30.6 against 14.9.)
//   A piece whose run-in did not arrive at the true dictionary is DIRTY: k_runin_fix decodes its blocks again, in
// order, from the final dictionary, each into the ring first; a block that comes out as it was ends the work (all that
// follows depends on it alone), a piece that changes up to its last block marks the piece behind it dirty for the next
// round.  runDirty[p] holds the ROUND in which piece p is to be redone (RUNIN_CLEAN: none); a dirty piece behind a dirty
// piece waits for it inside the launch (bounded), the host launches rounds until none is marked.  result[] is not
// written before everything is final (k_runin_publish), so a call that gives up -- a block that fails with its
// dictionary, too many dirty pieces in a row, rounds that run out -- leaves the first pass's results as they were for
// the pointer pass.
// ---------------------------------------------------------------------------------------------------------------------
#define RUNIN_CLEAN 0xffffffffu
#ifndef RUNIN_CHAIN
#define RUNIN_CHAIN 4
#endif
#define RUNIN_EXACT (-2)
// the dictionary in force in front of block b0 once the blocks in front of it are final in out[] (their sizes in
// result[] -- the first pass -- or runRes[])
__device__ __forceinline__ void runin_dict_before(const DecodeArgs &a, int b0, const uint8_t *&dict, uint32_t &dictLen)
{
    for (int j = b0 - 1; j >= a.segFirst; j--) {
        const int r0 = uni(a.result[j]);
        const int rj = r0 > 0 ? r0 : uni(a.runRes[j - a.segFirst]);
        if (rj > 0) { dict = a.out + a.outOff[j]; dictLen = (uint32_t)rj; return; }
    }
    dict_before(a, a.segFirst, dict, dictLen);
}

// one wave per piece
__global__ RUNIN_OCC void k_runin_decode(DecodeArgs a)
{
    __shared__ ParLds lds;
    const int p = (int)blockIdx.x;
    const int b0 = a.segFirst + p * a.runPiece, b1 = min(b0 + a.runPiece, a.segEnd);
    if (b0 >= b1) return;
    // where the run-in starts: runIn blocks back, or behind the last block in that range that decoded on its own
    int w0 = max(b0 - a.runIn, a.segFirst);
    bool exact = w0 == a.segFirst;
    const uint8_t *dict = nullptr; uint32_t dictLen = 0;
    {
        const int j = b0 - 1 - lane_id();
        const int rj = j >= w0 ? a.result[j] : 0;
        const unsigned long long own = __ballot(rj > 0);
        if (own) { w0 = b0 - (int)__builtin_ctzll(own); exact = true; }
    }
    if (exact) runin_dict_before(a, w0, dict, dictLen);           // (blocks of the segment in front of w0: w0 - 1 decoded on its own, or there is none)
    else { dict = a.zeroPage; dictLen = 65536u; }
    uint8_t *ring = a.ring + (uint64_t)p * 2u * a.ringStride;
    int dictBlk = -1;
    for (int f = w0; f < b1; f++) {
        const bool own = f >= b0;                                 // (in front of b0: the run-in; no block of it decoded on its own)
        if (f == b0 && lane_id() == 0) {
            int32_t *info = a.runInfo + 4 * p;
            info[0] = exact ? RUNIN_EXACT : dictBlk;              // -1: the stand-in is still in force
            info[1] = (int32_t)dictLen;
            info[2] = dictBlk >= 0 ? ((dictBlk - w0) & 1) : 0;
        }
        const int r0 = uni(a.result[f]);                          // the standalone pass's result: nobody writes it before k_runin_publish
        uint8_t *dst = own ? a.out + a.outOff[f] : ring + (uint64_t)((f - w0) & 1) * a.ringStride;
        int r = r0;
        if (is_codec_error(r0) && dictLen > 0) {                  // (an empty or rejected block leaves the dictionary, :2331-2333)
            const uint8_t *data = nullptr;
            int compLen = 0, cap = 0;
            r = read_block_header(a, f, data, compLen, cap);
            if (r == 0 && !own && (uint64_t)cap > a.ringStride) r = -1;   // (cannot happen: the stride is the largest capacity)
            if (r == 0)
                r = decode_block_par<false, true>(data, compLen, dst, cap, dict, dictLen, a.framed, a.framed + a.framedLen, lds, nullptr);
            r = uni(r);
            // (the next block reads this one through the same vector memory pipeline that wrote it, in order: nothing to wait
            // for or to write back -- an agent-scope fence here is an L2 write-back per block and wave, measured below)
            wave_fence();
        }
        if (own && lane_id() == 0) {
            a.runRes[f - a.segFirst] = r;
            if (is_codec_error(r)) atomicOr(&a.runCtl[1], 1u);  // fails with the dictionary it got: the exact path decides what that means
        }
        if (r > 0) { dict = dst; dictLen = (uint32_t)r; dictBlk = f; }
    }
}

// one workgroup per piece: is the dictionary the run-in arrived with the one the piece in front left?
__global__ __launch_bounds__(256) void k_runin_verify(DecodeArgs a)
{
    const int p = (int)blockIdx.x;
    const int b0 = a.segFirst + p * a.runPiece;
    const int32_t *info = a.runInfo + 4 * p;
    __shared__ uint32_t diff;
    __shared__ int sj, slen;
    if (threadIdx.x == 0) {
        diff = 0; sj = -1; slen = 0;
        if (info[0] != RUNIN_EXACT) {
            for (int j = b0 - 1; j >= a.segFirst; j--) {
                const int r0 = a.result[j];
                const int rj = r0 > 0 ? r0 : a.runRes[j - a.segFirst];
                if (rj > 0) { sj = j; slen = rj; break; }
            }
            if (sj < 0 || sj != info[0] || slen != info[1]) diff = 1;
        }
    }
    __syncthreads();
    if (info[0] != RUNIN_EXACT && diff == 0) {
        const uint8_t *x = a.out + a.outOff[sj];
        const uint8_t *y = a.ring + ((uint64_t)p * 2u + (uint64_t)info[2]) * a.ringStride;
        uint32_t mine = 0;
        for (int i = (int)threadIdx.x * 16; i < slen; i += 256 * 16) {
            if (i + 16 <= slen && (((uintptr_t)(x + i)) & 15u) == 0) {
                const uint4 u = *(const uint4 *)(x + i), v = *(const uint4 *)(y + i);
                mine |= (uint32_t)((u.x != v.x) | (u.y != v.y) | (u.z != v.z) | (u.w != v.w));
            } else {
                for (int k = i; k < min(i + 16, slen); k++) mine |= (uint32_t)(x[k] != y[k]);
            }
        }
        if (mine) atomicOr(&diff, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) a.runDirty[p] = (info[0] != RUNIN_EXACT && diff) ? 0u : RUNIN_CLEAN;
}

// dst[0, n) = src[0, n); returns whether that changed dst (whole wave; dst and src do not overlap)
__device__ __forceinline__ bool wave_copy_changed(uint8_t *dst, const uint8_t *src, int n)
{
    uint32_t d = 0;
    const int head = min(n, (int)((16u - (uint32_t)(uintptr_t)dst) & 15u));
    if (lane_id() < head) { const uint8_t o = dst[lane_id()], v = src[lane_id()]; d |= (uint32_t)(o != v); dst[lane_id()] = v; }
    const int body = (n - head) >> 4;
    for (int i = lane_id(); i < body; i += LZ4_WAVE) {
        uint4 o = *(const uint4 *)(dst + head + 16 * i), v;
        __builtin_memcpy(&v, src + head + 16 * i, 16);
        d |= (uint32_t)((o.x != v.x) | (o.y != v.y) | (o.z != v.z) | (o.w != v.w));
        *(uint4 *)(dst + head + 16 * i) = v;
    }
    const int t0 = head + 16 * body;
    if (t0 + lane_id() < n) { const uint8_t o = dst[t0 + lane_id()], v = src[t0 + lane_id()]; d |= (uint32_t)(o != v); dst[t0 + lane_id()] = v; }
    return __ballot(d != 0) != 0ull;
}

// one wave per piece and round: redo a dirty piece from its final dictionary
__global__ RUNIN_OCC void k_runin_fix(DecodeArgs a)
{
    __shared__ ParLds lds;
    const int p = (int)blockIdx.x;
    const uint32_t round = (uint32_t)a.runRound;
    const int b0 = a.segFirst + p * a.runPiece, b1 = min(b0 + a.runPiece, a.segEnd);
    if (b0 >= b1 || (uint32_t)uni((int)__hip_atomic_load(&a.runDirty[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != round) return;
    // (p > 0: piece 0 starts at the segment's first block and is exact.)  The piece in front is being redone in this launch:
    // wait for it -- waves are dispatched in order, it is resident whenever this one is -- a bounded while; a wait that
    // gives up leaves the piece to the next round (the wave in front takes a piece marked for THIS round to be waiting for it)
    // Pieces to be redone in a row are a serial chain, and a stream whose every block is made of the block before it (a
    // 60 000-byte period of noise) has nothing but those: more than RUNIN_CHAIN in front of this one and the call is given up
    // for the pointer pass, which resolves such chains in log steps.
    {
        int k = 1;
        while (k <= RUNIN_CHAIN && p - k > 0 &&
               (uint32_t)uni((int)__hip_atomic_load(&a.runDirty[p - k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == round) k++;
        if (k > RUNIN_CHAIN) {
            if (lane_id() == 0) {
                atomicOr(&a.runCtl[1], 2u);
                __hip_atomic_store(&a.runDirty[p], RUNIN_CLEAN, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
    for (int spin = 0;; spin++) {
        if ((uint32_t)uni((int)__hip_atomic_load(&a.runDirty[p - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) != round) break;
        if (spin >= a.runSpin) {
            if (lane_id() == 0) {
                __hip_atomic_store(&a.runDirty[p], round + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&a.runCtl[0], 1u);
            }
            return;
        }
        __builtin_amdgcn_s_sleep(64);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    const uint8_t *dict = nullptr; uint32_t dictLen = 0;
    runin_dict_before(a, b0, dict, dictLen);
    uint8_t *ring = a.ring + (uint64_t)p * 2u * a.ringStride;
    bool changed = true;                                         // the dictionary in force differs from what the blocks were made with
    for (int f = b0; f < b1 && changed; f++) {
        const int r0 = uni(a.result[f]);
        if (r0 > 0) { changed = false; break; }                  // decoded on its own: what follows depends on this block alone
        if (!is_codec_error(r0)) continue;                       // empty or rejected: the dictionary passes
        int r = r0;
        if (dictLen > 0) {
            const uint8_t *data = nullptr;
            int compLen = 0, cap = 0;
            r = read_block_header(a, f, data, compLen, cap);
            if (r == 0 && (uint64_t)cap > a.ringStride) r = -1;
            if (r == 0)
                r = decode_block_par<false, true>(data, compLen, ring, cap, dict, dictLen, a.framed, a.framed + a.framedLen, lds, nullptr);
            r = uni(r);
        }
        if (r <= 0) {
            // fails with its true dictionary (or there is none): the stream is broken here, the exact path reports it
            if (lane_id() == 0) {
                atomicOr(&a.runCtl[1], 1u);
                __hip_atomic_store(&a.runDirty[p], RUNIN_CLEAN, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // (nobody waits for a call that has given up)
            }
            return;
        }
        uint8_t *dst = a.out + a.outOff[f];
        const int old = uni(a.runRes[f - a.segFirst]);
        wave_fence();                                            // (the ring's bytes are this wave's own stores)
        const bool diff = wave_copy_changed(dst, ring, r) || old != r;
        if (lane_id() == 0) a.runRes[f - a.segFirst] = r;
        if (!diff) changed = false;
        dict = dst; dictLen = (uint32_t)r;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    if (lane_id() == 0) {
        const int nPieces = (a.segEnd - a.segFirst + a.runPiece - 1) / a.runPiece;
        if (changed && p + 1 < nPieces && a.runInfo[4 * (p + 1)] != RUNIN_EXACT) {
            // the piece behind was made with another dictionary.  Dirty in this round: it is waiting for this wave; marked
            // already: nothing to add
            uint32_t seen = __hip_atomic_load(&a.runDirty[p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (seen != round && seen != round + 1u) {
                if (__hip_atomic_compare_exchange_strong(&a.runDirty[p + 1], &seen, round + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    atomicAdd(&a.runCtl[0], 1u);
                    break;
                }
            }
        }
        __hip_atomic_store(&a.runDirty[p], RUNIN_CLEAN, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void k_runin_publish(DecodeArgs a)
{
    const int f = a.segFirst + (int)(blockIdx.x * 256u + threadIdx.x);
    if (f >= a.segEnd) return;
    if (a.result[f] <= 0) a.result[f] = a.runRes[f - a.segFirst];
}

void launch_runin_decode(const DecodeArgs &a, hipStream_t s)
{
    const int n = a.segEnd - a.segFirst;
    if (n <= 0) return;
    const int nPieces = (n + a.runPiece - 1) / a.runPiece;
    hipLaunchKernelGGL(k_runin_decode, dim3((unsigned)nPieces), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_runin_verify, dim3((unsigned)nPieces), dim3(256), 0, s, a);
}
void launch_runin_fix(const DecodeArgs &a, hipStream_t s)
{
    const int n = a.segEnd - a.segFirst;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_runin_fix, dim3((unsigned)((n + a.runPiece - 1) / a.runPiece)), dim3(64), 0, s, a);
}
void launch_runin_publish(const DecodeArgs &a, hipStream_t s)
{
    const int n = a.segEnd - a.segFirst;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_runin_publish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
}

void launch_linked_runs(const DecodeArgs &a, hipStream_t s)
{
    const int n = a.segEnd - a.segFirst;
    if (n <= 0) return;
    if (!a.runList) { hipLaunchKernelGGL(k_decode_fixup_runs, dim3(1), dim3(64), 0, s, a); return; }
    hipLaunchKernelGGL(k_run_starts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_decode_fixup_runs, dim3((unsigned)min(n, a.runCap)), dim3(64), 0, s, a);
}

// Single stream (streamFirst == null): only REGIONS need the serial walk.  A region is a maximal run of
// blocks whose standalone result is <= 0; it starts behind a block with result > 0 (final: the fixup
// never touches it), which is also the dictionary in force at the region's first block (:2347-2355).
// Regions are disjoint, so each is walked by the wavefront that finds its first block; a stream in
// which every block decoded standalone has no region and the kernel costs one coalesced read of
// result[].  (A reference-written linked stream is one long region: block 0, then every block fails.)
//
// Inside a region the chain of blocks is serial, but most of each block is not: k_decode_tolerant has
// already decoded every failed block in parallel and left a list of the matches that (transitively) need the
// previous block (TolCtx, decode_seq.hpp).  The walk only replays those lists, in LDS (linked_replay.hpp);
// a block without a usable list is re-decoded by the exact serial decoder with its dictionary.
struct TolLds { ParLds p; TolCtx t; };

// (its out-of-line callee is decode_seq_run_tol, which no other kernel calls: the register bound can be its own)
__global__ __launch_bounds__(64, 4) void k_decode_tolerant(DecodeArgs a)
{
    if (a.asyncGate && a.linkStat[0] == 0u) return;     // asynchronous linked decode: the first pass found nothing to do
    __shared__ TolLds lds;
    const int blk = a.segFirst + (int)blockIdx.x;
    if (!is_codec_error(uni(a.result[blk]))) {
        if (lane_id() == 0) a.tolRegion[blk] = -1;
        return;
    }
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    int region = -1, count = 0, size = -1;
    if (read_block_header(a, blk, data, compLen, cap) == 0 && cap <= TOL_MAX_BLOCK) {
        // one list region (TOL_LIST_CAP entries) per 64 KiB of capacity, taken in one piece
        const unsigned need = (unsigned)max(1, (cap + RPL_HALF - 1) / RPL_HALF);
        // The regions of a launch are handed out by position: block segFirst + i owns [i * tolPer, (i + 1) * tolPer), tolPer
        // = what the largest dependent block of the call needs.  (Round 3 drew them from one counter: an atomic with a
        // returned value per block, all on one address, in front of every block's decode.)
        const unsigned got = (unsigned)(blk - a.segFirst) * (unsigned)a.tolPer;
        if (need <= (unsigned)a.tolPer && got + need <= (unsigned)a.tolRegions) {
            region = (int)got;
            for (int i = lane_id(); i < 128; i += LZ4_WAVE) lds.t.taint[i] = 0;
            if (lane_id() == 0) {
                lds.t.list = (TolEntry *)a.tolPool + (size_t)region * TOL_LIST_CAP;
                lds.t.cap = need * TOL_LIST_CAP;
                lds.t.count = 0;
                uint32_t gs = 4;                                   // 4096 granules cover the block
                while (((uint32_t)cap >> gs) > 4096u) gs++;
                lds.t.granShift = gs;
            }
            wave_fence();
            size = decode_block_par<false, false, true>(data, compLen, a.out + a.outOff[blk], cap, nullptr, 0, a.framed,
                                                        a.framed + a.framedLen, lds.p, nullptr, &lds.t);
            size = uni(size);
            wave_fence();
            count = (int)lds.t.count;
            if ((unsigned)count > need * TOL_LIST_CAP) region = -1;        // the list overflowed: no list
        }
    }
    if (lane_id() == 0) { a.tolRegion[blk] = region; a.tolCount[blk] = count; a.tolSize[blk] = size; }
}

// One workgroup of RPL_THREADS walks the regions that start in its 64 blocks.  Wave 0 takes every decision
// (and runs the exact serial decoder when a block has no usable list); the replay and the block copies are
// done by all waves.  The register bound matters although the LDS footprint allows one workgroup per CU anyway:
// the out-of-line sequential decoder is compiled once for all its callers, and a caller without the bound would
// relax it for every kernel.
enum { RGN_END = 0, RGN_SKIP = 1, RGN_FIX = 2 };

__global__ __attribute__((amdgpu_flat_work_group_size(RPL_THREADS, RPL_THREADS), amdgpu_waves_per_eu(PAR_WAVES, PAR_WAVES)))
void k_decode_fixup_regions(DecodeArgs a)
{
    if (a.asyncGate && a.linkStat[0] == 0u) return;     // asynchronous linked decode: the first pass found nothing to do
    __shared__ ParLds lds;
    __shared__ ReplayLds rl;
    __shared__ RplCtl ctl;
    __shared__ unsigned long long startMask;
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6;
    const int base = a.segFirst + (int)blockIdx.x * LZ4_WAVE;
    if (a.ptrBad && !a.ptrBad[0]) return;                           // the data-parallel pass has done the segment
    if (wave == 0) {
        const int blk = base + tid;
        const int r0 = (blk < a.segEnd) ? a.result[blk] : 1;
        const int rp = (blk > 0 && blk < a.segEnd) ? a.result[blk - 1] : 1;
        // the first block of a range continues whatever region the blocks before the range ended in
        const bool startsRegion = blk < a.segEnd && r0 <= 0 && (blk == 0 || blk == a.segFirst || rp > 0);
        const uint64_t m0 = __ballot(startsRegion);
        if (tid == 0) startMask = m0;
    }
    __syncthreads();
    for (uint64_t m = startMask; m; m &= m - 1) {
        int f = base + (int)__builtin_ctzll(m);
        // the dictionary in force: the last block before f that produced output (they are final), else the
        // caller's; (pointer, length) are recomputed by every thread
        const uint8_t *dict = nullptr;
        uint32_t dictLen = 0;
        if (a.dict0) { dict = a.dict0; dictLen = a.dict0Len; }
        for (int j = f - 1; j >= -a.lookBack; j--) {
            const int rj = a.result[j];
            if (rj > 0) { dict = a.out + a.outOff[j]; dictLen = (uint32_t)rj; break; }
        }
        bool dictInLds = false;                                   // rl.buf holds `dict` below RPL_HALF
        for (; f < a.segEnd; f++) {
            if (tid == 0) {
                const int r = a.result[f];
                int action = RGN_FIX;
                if (r > 0) action = RGN_END;                      // end of the region
                else if (!is_codec_error(r) || dictLen == 0) action = RGN_SKIP;   // nothing to fix, or nothing to fix it with
                ctl.action = action;
                ctl.r = r;
            }
            __syncthreads();
            const int action = ctl.action;
            __syncthreads();
            if (action == RGN_END) break;
            if (action == RGN_SKIP) continue;
            uint8_t *dst = a.out + a.outOff[f];
            if (wave == 0) {
                const uint8_t *data0 = nullptr;
                int compLen0 = 0, cap0 = 0;
                const int hdr = read_block_header(a, f, data0, compLen0, cap0);
                if (tid == 0) {
                    ctl.hdr = hdr; ctl.cap = cap0; ctl.compLen = compLen0;
                    ctl.region = -1; ctl.count = 0; ctl.size = -1;
                    if (hdr == 0 && a.tolPool) { ctl.region = a.tolRegion[f]; ctl.count = a.tolCount[f]; ctl.size = a.tolSize[f]; }
                }
            }
            __syncthreads();
            const int hdr = ctl.hdr, cap = ctl.cap, region = ctl.region, count = ctl.count, size = ctl.size;
            bool fixed = false;
            if (region >= 0 && count <= TOL_LIST_CAP && size > 0 && size <= RPL_HALF && cap <= RPL_HALF) {
                const int dl = (int)min(dictLen, (uint32_t)RPL_HALF);
                if (!dictInLds) rpl_load(rl.buf + RPL_HALF - dl, dict + (dictLen - (uint32_t)dl), dl);
                rpl_load(rl.buf + RPL_HALF, dst, size);
                __syncthreads();
                // dictLen >= 64 KiB: no offset check in the reference (:1764); every offset fits 65535 anyway
#ifdef RPL_STATS
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
                fixed = replay_block(rl, ctl, (const TolEntry *)a.tolPool + (size_t)region * TOL_LIST_CAP, count, dl, size, cap,
                                     a.tolCounter);
#ifdef RPL_STATS
                if (tid == 0) atomicAdd(&a.tolCounter[3], (unsigned)((__builtin_amdgcn_s_memtime() - t0) >> 4));
#endif
                if (fixed) {
                    rpl_store(dst, rl.buf + RPL_HALF, size);
                    __syncthreads();
                    // this block is the next one's dictionary: move it below RPL_HALF (upwards in steps that do
                    // not overlap: size <= RPL_HALF, so source and destination ranges are disjoint)
                    if ((size & 15) == 0) {
                        for (int c = tid; c < (size >> 4); c += RPL_THREADS) {
                            const par_v4 v = *(const par_v4 *)(rl.buf + RPL_HALF + 16 * c);
                            *(par_v4 *)(rl.buf + RPL_HALF - size + 16 * c) = v;
                        }
                    } else {
                        for (int x = tid; x < size; x += RPL_THREADS) rl.buf[RPL_HALF - size + x] = rl.buf[RPL_HALF + x];
                    }
                    __syncthreads();
                    dictInLds = true;
                    if (tid == 0) { ctl.r = size; a.result[f] = size; }
                }
            }
            if (!fixed) {
                // exact serial decode with the dictionary (also what yields the reference's error codes), wave 0 alone
                if (wave == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");   // replayed blocks were written by this workgroup
                    const uint8_t *data = nullptr;
                    int compLen = 0, cap1 = 0;
                    int r = read_block_header(a, f, data, compLen, cap1);
                    if (r == 0)
                        r = decode_block_par<false, true>(data, compLen, dst, cap1, dict, dictLen, a.framed,
                                                          a.framed + a.framedLen, lds, nullptr);
                    r = uni(r);
                    if (lane_id() == 0) { ctl.r = r; a.result[f] = r; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                }
                dictInLds = false;
            }
            __syncthreads();
            const int r = ctl.r;
            __syncthreads();
            if (r > 0) { dict = dst; dictLen = (uint32_t)r; }     // :2331-2333, :2353-2355
            (void)hdr;
        }
        __syncthreads();
    }
}

// ---- one long linked stream, data-parallel second pass (linked_ptr.hpp) ----
// Where the segment's pointer space starts in the output buffer: at the block before its first block (that
// block's output is the first block's dictionary), or at the first block when there is none.
__device__ __forceinline__ bool ptr_has_prev(const DecodeArgs &a) { return a.segFirst > 0 || a.lookBack > 0; }
__device__ __forceinline__ uint64_t ptr_lo(const DecodeArgs &a)
{
    return ptr_has_prev(a) ? a.outOff[a.segFirst - 1] : a.outOff[a.segFirst];
}
// The stream a block belongs to (index into ptrBad[]; -1 = none: decoded on its own) and whether the block
// before it is its dictionary.  One stream: every block but the very first has one.
__device__ __forceinline__ int ptr_stream(const DecodeArgs &a, int blk, bool &hasDict)
{
    if (!a.streamFirst) { hasDict = blk > 0 || a.lookBack > 0; return 0; }
    auto first = [&](int s) { return min(max(a.streamFirst[s], 0), a.nBlocks); };
    hasDict = false;
    if (a.nStreams <= 0 || blk < first(0) || blk >= first(a.nStreams)) return -1;
    int lo = 0, hi = a.nStreams;                       // first(lo) <= blk < first(hi)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first(mid) <= blk) lo = mid; else hi = mid;
    }
    hasDict = blk > first(lo);
    return lo;
}
// a block the tolerant pass left a usable list for (stable while the second pass runs: result[] is not)
__device__ __forceinline__ bool ptr_listed(const DecodeArgs &a, int blk)
{
    return blk >= a.segFirst && blk < a.segEnd && a.tolRegion[blk] >= 0 && a.tolSize[blk] > 0;
}
// decoded size of block blk as far as the second pass knows it, 0 = no output
__device__ __forceinline__ int ptr_size(const DecodeArgs &a, int blk)
{
    const int r = a.result[blk];
    if (r > 0) return r;
    return (is_codec_error(r) && ptr_listed(a, blk)) ? a.tolSize[blk] : 0;
}

// a dependent block the second pass is resolving: listed, and nothing in its stream was turned down
__device__ __forceinline__ bool ptr_taken(const DecodeArgs &a, int blk)
{
    if (!ptr_listed(a, blk) || !is_codec_error(a.result[blk])) return false;
    bool hd;
    const int sid = ptr_stream(a, blk, hd);
    return sid >= 0 && !a.ptrBad[sid];
}

// Workgroup i >= 1: block segFirst + i - 1 writes the pointers of its own bytes -- self, then its deferred
// matches.  Workgroup 0: the bytes in front of the segment (caller's dictionary, block before the segment).
__global__ __launch_bounds__(256) void k_ptr_expand(DecodeArgs a)
{
    if (a.asyncGate && a.linkStat[0] == 0u) return;     // asynchronous linked decode: the first pass found nothing to do
    uint32_t *P = a.ptr;
    const int tid = (int)threadIdx.x;
    const uint64_t lo = ptr_lo(a);
    if (blockIdx.x == 0) {
        uint32_t n = PTR_PRE;
        if (ptr_has_prev(a)) {
            const int rp = a.result[a.segFirst - 1];
            if (rp > 0) n += (uint32_t)rp;
        }
        n = (uint32_t)min((uint64_t)n, a.ptrCap);      // (the first block checks its own range against the capacity)
        for (uint32_t i = (uint32_t)tid; i < n; i += 256u) P[i] = i | PTR_FINAL;
        return;
    }
    const int blk = a.segFirst + (int)blockIdx.x - 1;
    bool hasDict = false;
    const int sid = ptr_stream(a, blk, hasDict);
    auto fail = [&]() { if (tid == 0 && sid >= 0) atomicOr(&a.ptrBad[sid], 1u); };
    const int r = a.result[blk];
    const bool listed = is_codec_error(r) && ptr_listed(a, blk);
    if (is_codec_error(r) && !listed) { fail(); return; }          // a dependent block without a list
    const int size = listed ? a.tolSize[blk] : r;
    if (size <= 0) return;                                          // no output: nobody points here
    if (a.outOff[blk] < lo) { fail(); return; }
    const uint64_t b64 = a.outOff[blk] - lo + PTR_PRE;
    if (b64 + (uint64_t)size > a.ptrCap || b64 + (uint64_t)size >= (uint64_t)PTR_FINAL) { fail(); return; }
    const uint32_t bLo = (uint32_t)b64;
    if (!listed) {                                                  // a block that needed nothing: all roots
        for (uint32_t i = (uint32_t)tid; i < (uint32_t)size; i += 256u) P[bLo + i] = (bLo + i) | PTR_FINAL;
        return;
    }

    // the dictionary in force (cbits/lz4.c:2347-2355 with every block in its own allocation): the block before
    uint32_t dictEnd = PTR_PRE;                                     // pointer index one past the dictionary
    int dictLen = 0;
    if (hasDict) {
        const int ps = ptr_size(a, blk - 1);
        if (ps <= 0 || a.outOff[blk - 1] < lo) { fail(); return; } // dictionary further back, or none: serial walk
        dictEnd = (uint32_t)(a.outOff[blk - 1] - lo + PTR_PRE) + (uint32_t)ps;
        dictLen = ps;
    } else if (a.dict0 && !a.streamFirst) {
        dictLen = (int)a.dict0Len;
    }
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    if (read_block_header(a, blk, data, compLen, cap) != 0) { fail(); return; }
    const TolEntry *list = (const TolEntry *)a.tolPool + (size_t)a.tolRegion[blk] * TOL_LIST_CAP;
    const int n = a.tolCount[blk];
    const int lane = tid & 63;
    bool bad = false;
    auto unpack = [](uint64_t w, int &dpos, int &ml, int &spos) { tol_unpack(w, dpos, ml, spos); };
    // The list is in stream order: destinations ascend and do not overlap.  Entry e writes the pointers of ITS range
    // of the block in one go: the clean bytes between the entry before it and itself (roots), then its own bytes.
    for (int e0 = 0; e0 < n; e0 += 256) {
        const int e = e0 + tid;
        int gs = 0, dpos = 0, ml = 0, spos = 0;
        if (e < n) {
            unpack(*(const uint64_t *)(list + e), dpos, ml, spos);
            if (e > 0) {
                int pd, pm, ps;
                unpack(*(const uint64_t *)(list + e - 1), pd, pm, ps);
                gs = pd + pm;
            }
            if (!(ml > 0 && spos < dpos && dpos + ml <= size && spos >= -dictLen && gs <= dpos)) bad = true;
            // a match that starts in the dictionary must end LASTLITERALS before the end of the output (:1884-1889)
            if (spos < 0 && dpos + ml > cap - LZ4_LASTLITERALS) bad = true;
            if (bad) { ml = 0; gs = dpos; }
        }
        // position x of the range: a root in front of dpos; behind it byte x - dpos of the match, which comes from
        // position spos + (x - dpos): in this block, or (negative) in the dictionary
        auto ptrAt = [&](int x, int d, int sp) -> uint32_t {
            if (x < d) return (bLo + (uint32_t)x) | PTR_FINAL;
            const int s1 = sp + (x - d);
            return (s1 >= 0) ? bLo + (uint32_t)s1 : dictEnd - (uint32_t)(-s1);
        };
        const int len = dpos + ml - gs;
        const int head = min(len, 8);
        for (int j = 0; j < head; j++) P[bLo + (uint32_t)(gs + j)] = ptrAt(gs + j, dpos, spos);
        for (uint64_t lm = __ballot(len > 8); lm; lm &= lm - 1) {  // the rest of a long range: by the whole wave
            const int k = (int)__builtin_ctzll(lm);
            const int kg = __builtin_amdgcn_readlane(gs, k), kd = __builtin_amdgcn_readlane(dpos, k);
            const int ks = __builtin_amdgcn_readlane(spos, k), kend = kd + __builtin_amdgcn_readlane(ml, k);
            for (int x = kg + 8 + lane; x < kend; x += LZ4_WAVE) P[bLo + (uint32_t)x] = ptrAt(x, kd, ks);
        }
    }
    {   // the clean bytes behind the last entry
        int tail = 0;
        if (n > 0) {
            int pd, pm, ps;
            unpack(*(const uint64_t *)(list + n - 1), pd, pm, ps);
            tail = min(pd + pm, size);
        }
        for (int x = tail + tid; x < size; x += 256) P[bLo + (uint32_t)x] = (bLo + (uint32_t)x) | PTR_FINAL;
    }
    if (__syncthreads_or(bad ? 1 : 0)) fail();
}

// Workgroup -> (block of the segment, part of the block) for the jump and fetch passes.  Workgroups go to the 8
// XCDs round-robin and each XCD has its own L2: a run of PTR_RUN consecutive blocks, all parts, is given to ONE
// XCD, so that the pointers a chain visits (its own block's and the block's before) are in the L2 it runs on.
#ifndef PTR_RUN
#define PTR_RUN 16
#endif
#ifndef PTR_CHASE
#define PTR_CHASE 32                // pointers the chasing fetch follows before it gives a byte up (12: an engine-written
                                   // linked text stream keeps needing the passes, 44 instead of 59 GB/s; 96: as 32)
#endif
#ifndef PTR_ILP
#define PTR_ILP 1                  // groups per thread advancing in lock step: more requests in flight LOSE (2: -8 %, 4: -15 %),
#endif                             // the passes are bound by the number of scattered requests, not by their latency
__device__ __forceinline__ void ptr_map(unsigned wg, int &blkRel, int &part)
{
#ifdef PTR_FLAT_MAP
    blkRel = (int)(wg / PTR_PARTS); part = (int)(wg % PTR_PARTS);
#else
    const unsigned xcd = wg & 7u, j = wg >> 3;                      // the j-th workgroup this XCD receives
    const unsigned per = PTR_RUN * PTR_PARTS;
    const unsigned run = j / per, within = j % per;
    blkRel = (int)((run * 8u + xcd) * PTR_RUN + within / PTR_PARTS);
    part = (int)(within % PTR_PARTS);
#endif
}
static unsigned ptr_grid(int n) { return (unsigned)((n + 8 * PTR_RUN - 1) / (8 * PTR_RUN)) * (8 * PTR_RUN) * PTR_PARTS; }

// One pass of pointer jumping over the bytes of the listed blocks (PTR_PARTS workgroups per block).  Reads of
// pointers another thread is updating are harmless: every value a pointer ever holds is an ancestor.
__global__ __launch_bounds__(256) void k_ptr_jump(DecodeArgs a, int pass, unsigned items)
{
    if (a.asyncGate && a.linkStat[0] == 0u) return;     // asynchronous linked decode: the first pass found nothing to do
    PtrCtl *ctl = (PtrCtl *)a.ptrCtl;
    // passes behind the first: only if the chasing fetch left something, and the pass before changed something
    if (pass > 0 && !(ctl->changed[PTR_MAX_PASSES] && ctl->changed[pass - 1])) return;
    uint32_t *P = a.ptr;
    bool open = false;
    int lastBlk = -1;
    // (passes behind the first are launched with a small grid: they usually find nothing to do)
    for (unsigned item = blockIdx.x; item < items; item += gridDim.x) {
    int blkRel, part;
    ptr_map(item, blkRel, part);
    const int blk = a.segFirst + blkRel;
    if (blk >= a.segEnd || !ptr_taken(a, blk)) continue;
    lastBlk = blk;
    const uint32_t bLo = (uint32_t)(a.outOff[blk] - ptr_lo(a) + PTR_PRE);
    const int size = a.tolSize[blk];
    const int per = ((size + PTR_PARTS - 1) / PTR_PARTS + 3) & ~3;
    const int x0 = part * per, x1 = min(size, x0 + per);
    auto chase = [&](uint32_t e) -> uint32_t {
#pragma unroll
        for (int k = 0; k < PTR_JUMPS; k++) {
            e = P[e];
            if (e & PTR_FINAL) break;
        }
        if (!(e & PTR_FINAL)) open = true;
        return e;
    };
    if (((bLo | (uint32_t)x0) & 3u) == 0) {
        // four pointers per thread (16-byte accesses)
        uint4 *P4 = (uint4 *)(P + bLo);
        const int q1 = x1 >> 2;
        // One hop for all four pointers of a group per step.  Neighbouring bytes of a match have neighbouring sources,
        // hop after hop, until a chain leaves its match: while the four pointers are consecutive they are fetched
        // with ONE 16-byte request; otherwise with up to four requests that are in flight together.
        auto hop = [&](uint4 &v) -> bool {                 // false: nothing left to follow
            const bool o0 = !(v.x & PTR_FINAL), o1 = !(v.y & PTR_FINAL), o2 = !(v.z & PTR_FINAL), o3 = !(v.w & PTR_FINAL);
            if (!(o0 || o1 || o2 || o3)) return false;
            if (o0 && o1 && o2 && o3 && v.y == v.x + 1u && v.z == v.x + 2u && v.w == v.x + 3u) {
                uint4 w;
                __builtin_memcpy(&w, P + v.x, 16);
                v = w;
            } else {
                const uint32_t n0 = o0 ? P[v.x] : v.x, n1 = o1 ? P[v.y] : v.y, n2 = o2 ? P[v.z] : v.z, n3 = o3 ? P[v.w] : v.w;
                v.x = n0; v.y = n1; v.z = n2; v.w = n3;
            }
            return true;
        };
        auto unresolved = [](const uint4 &v) { return !((v.x & v.y & v.z & v.w) & PTR_FINAL); };
        for (int qb = (x0 >> 2) + (int)threadIdx.x; qb < q1; qb += 256 * PTR_ILP) {
            uint4 v[PTR_ILP];
            bool live[PTR_ILP];
#pragma unroll
            for (int g = 0; g < PTR_ILP; g++) {
                const int q = qb + 256 * g;
                live[g] = q < q1;
                v[g] = live[g] ? P4[q] : make_uint4(PTR_FINAL, PTR_FINAL, PTR_FINAL, PTR_FINAL);
                live[g] = live[g] && unresolved(v[g]);
            }
            bool dirty[PTR_ILP];
#pragma unroll
            for (int g = 0; g < PTR_ILP; g++) dirty[g] = live[g];
#pragma unroll 1
            for (int k = 0; k < PTR_JUMPS; k++) {
                bool any = false;
#pragma unroll
                for (int g = 0; g < PTR_ILP; g++) {
                    if (live[g]) live[g] = hop(v[g]);
                    any = any || live[g];
                }
                if (!any) break;
            }
#pragma unroll
            for (int g = 0; g < PTR_ILP; g++) {
                if (dirty[g]) {
                    P4[qb + 256 * g] = v[g];
                    if (unresolved(v[g])) open = true;
                }
            }
        }
        for (int x = (q1 << 2) + (int)threadIdx.x; x < x1; x += 256) {
            const uint32_t e = P[bLo + (uint32_t)x];
            if (!(e & PTR_FINAL)) P[bLo + (uint32_t)x] = chase(e);
        }
    } else {
        for (int x = x0 + (int)threadIdx.x; x < x1; x += 256) {
            const uint32_t e = P[bLo + (uint32_t)x];
            if (!(e & PTR_FINAL)) P[bLo + (uint32_t)x] = chase(e);
        }
    }
    }
    if (__syncthreads_or(open ? 1 : 0) && threadIdx.x == 0) {
        ctl->changed[pass] = 1u;
        if (pass == PTR_MAX_PASSES - 1 && lastBlk >= 0) {  // cannot happen (linked_ptr.hpp); never guess
            bool hd;
            atomicOr(&a.ptrBad[ptr_stream(a, lastBlk, hd)], 1u);
        }
    }
}

// Every deferred byte is fetched from its root.  CHASE: the fetch that runs right behind the FIRST jump pass finishes
// what that pass left open by following those chains itself (up to PTR_CHASE pointers, nothing written back): on
// shallow data -- text is done after one pass and a few hops -- no further pass over the pointers is needed.  A byte it
// cannot resolve raises PtrCtl::changed[PTR_MAX_PASSES]: only then do the remaining jump passes and the plain fetch
// behind them run.
template <bool CHASE>
__global__ __launch_bounds__(256) void k_ptr_fetch(DecodeArgs a, unsigned items)
{
    if (a.asyncGate && a.linkStat[0] == 0u) return;     // asynchronous linked decode: the first pass found nothing to do
    PtrCtl *ctl = (PtrCtl *)a.ptrCtl;
    if (!CHASE && !ctl->changed[PTR_MAX_PASSES]) return;
    const uint32_t *P = a.ptr;
    const uint64_t lo = ptr_lo(a);
    const uint8_t *outLo = a.out + lo;
    const uint8_t *dictTail = a.dict0 ? a.dict0 + a.dict0Len : nullptr;      // index PTR_PRE - d is dictTail[-d]
    auto root = [&](uint32_t e) -> uint8_t {
        return (e >= PTR_PRE) ? outLo[e - PTR_PRE] : dictTail[(int)e - (int)PTR_PRE];
    };
    bool unresolved = false;
    auto follow = [&](uint32_t e) -> uint32_t {                     // CHASE, one byte: the root, or an open pointer
#pragma unroll 1
        for (int k = 0; k < PTR_CHASE && !(e & PTR_FINAL); k++) e = P[e];
        if (!(e & PTR_FINAL)) unresolved = true;
        return e;
    };
    for (unsigned item = blockIdx.x; item < items; item += gridDim.x) {
        int blkRel, part;
        ptr_map(item, blkRel, part);
        const int blk = a.segFirst + blkRel;
        if (blk >= a.segEnd || !ptr_taken(a, blk)) continue;
        if (a.onlyBlk >= 0 && blk != a.onlyBlk) continue;
        const uint32_t bLo = (uint32_t)(a.outOff[blk] - lo + PTR_PRE);
        const int size = a.tolSize[blk];
        const int per = ((size + PTR_PARTS - 1) / PTR_PARTS + 3) & ~3;
        const int x0 = part * per, x1 = min(size, x0 + per);
        uint8_t *dst = a.out + a.outOff[blk];
        if (((bLo | (uint32_t)x0) & 3u) == 0) {
            // four bytes per thread: one 16-byte load of pointers, up to four byte fetches, one 4-byte store
            const uint4 *P4 = (const uint4 *)(P + bLo);
            const int q1 = x1 >> 2;
            for (int q = (x0 >> 2) + (int)threadIdx.x; q < q1; q += 256) {
                uint4 v = P4[q];
                if (CHASE && !((v.x & v.y & v.z & v.w) & PTR_FINAL)) {
#pragma unroll 1
                    for (int k = 0; k < PTR_CHASE; k++) {
                        const bool o0 = !(v.x & PTR_FINAL), o1 = !(v.y & PTR_FINAL), o2 = !(v.z & PTR_FINAL), o3 = !(v.w & PTR_FINAL);
                        if (!(o0 || o1 || o2 || o3)) break;
                        if (o0 && o1 && o2 && o3 && v.y == v.x + 1u && v.z == v.x + 2u && v.w == v.x + 3u) {
                            uint4 w;
                            __builtin_memcpy(&w, P + v.x, 16);            // neighbours: one request for the four
                            v = w;
                        } else {
                            const uint32_t n0 = o0 ? P[v.x] : v.x, n1 = o1 ? P[v.y] : v.y, n2 = o2 ? P[v.z] : v.z, n3 = o3 ? P[v.w] : v.w;
                            v.x = n0; v.y = n1; v.z = n2; v.w = n3;
                        }
                    }
                    if (!((v.x & v.y & v.z & v.w) & PTR_FINAL)) { unresolved = true; continue; }
                }
                v.x &= ~PTR_FINAL; v.y &= ~PTR_FINAL; v.z &= ~PTR_FINAL; v.w &= ~PTR_FINAL;
                const uint32_t self = bLo + 4u * (uint32_t)q;
                const bool m0 = v.x != self, m1 = v.y != self + 1u, m2 = v.z != self + 2u, m3 = v.w != self + 3u;
                if (!(m0 || m1 || m2 || m3)) continue;
                uint32_t w;
                if (m0 && m1 && m2 && m3 && v.x >= PTR_PRE && v.y == v.x + 1u && v.z == v.x + 2u && v.w == v.x + 3u) {
                    __builtin_memcpy(&w, outLo + (v.x - PTR_PRE), 4);          // four neighbouring roots: one request
                } else {
                    __builtin_memcpy(&w, dst + 4 * q, 4);
                    if (m0) w = (w & 0xffffff00u) | (uint32_t)root(v.x);
                    if (m1) w = (w & 0xffff00ffu) | ((uint32_t)root(v.y) << 8);
                    if (m2) w = (w & 0xff00ffffu) | ((uint32_t)root(v.z) << 16);
                    if (m3) w = (w & 0x00ffffffu) | ((uint32_t)root(v.w) << 24);
                }
                __builtin_memcpy(dst + 4 * q, &w, 4);
            }
            for (int x = (q1 << 2) + (int)threadIdx.x; x < x1; x += 256) {
                const uint32_t self = bLo + (uint32_t)x;
                uint32_t e = P[self];
                if (CHASE) { e = follow(e); if (!(e & PTR_FINAL)) continue; }
                e &= ~PTR_FINAL;
                if (e != self) dst[x] = root(e);
            }
        } else {
            for (int x = x0 + (int)threadIdx.x; x < x1; x += 256) {
                const uint32_t self = bLo + (uint32_t)x;
                uint32_t e = P[self];
                if (CHASE) { e = follow(e); if (!(e & PTR_FINAL)) continue; }
                e &= ~PTR_FINAL;
                if (e != self) dst[x] = root(e);
            }
        }
    }
    if (CHASE && __syncthreads_or(unresolved ? 1 : 0) && threadIdx.x == 0) {
        if (a.onlyBlk >= 0) ctl->lastOpen = 1u;          // (the full fetch behind this one decides about the further passes)
        else ctl->changed[PTR_MAX_PASSES] = 1u;
    }
}

// ... and only then do the results change: the passes above tell a dependent block by its standalone result.
__global__ __launch_bounds__(256) void k_ptr_finish(DecodeArgs a)
{
    if (a.asyncGate && a.linkStat[0] == 0u) return;     // asynchronous linked decode: the first pass found nothing to do
    const int blk = a.segFirst + (int)(blockIdx.x * 256u + threadIdx.x);
    if (blk < a.segEnd && ptr_taken(a, blk)) a.result[blk] = a.tolSize[blk];
}

size_t tol_region_bytes() { return (size_t)TOL_LIST_CAP * sizeof(TolEntry); }
size_t ptr_ctl_bytes() { return sizeof(PtrCtl); }
size_t ptr_ctl_last_open_offset() { return offsetof(PtrCtl, lastOpen); }

// linkStat[3] = blocks of the longest stream (the serial walk of a stream costs its length)
__global__ __launch_bounds__(256) void k_longest_stream(DecodeArgs a)
{
    const int s = (int)(blockIdx.x * 256u + threadIdx.x);
    if (s >= a.nStreams) return;
    const int b0 = min(max(a.streamFirst[s], 0), a.nBlocks), b1 = min(max(a.streamFirst[s + 1], b0), a.nBlocks);
    atomicMax(&a.linkStat[3], (uint32_t)(b1 - b0));
}

// How much of a linked block's output comes DIRECTLY from the block before it: matches whose source starts in front of
// the block (cbits/lz4.c:1883-1911).  One wavefront per SAMPLED block (blocks a.segFirst, a.segFirst + step, ...: `count` of
// them) walks the block's first sequences -- tokens and lengths only, nothing is copied -- and adds {bytes from the
// dictionary, bytes walked} to linkStat[8], [9].  The run-in decode reads how long a stream remembers a missing dictionary
// off this share before its first call (api.cpp): the reference's text 0.065, the engine's own linked text 0.077, noise
// with a period just under 64 KiB 0.5-0.9 (scripts/dict_share.py).
#define DICT_SHARE_SEQS 1024
__global__ __launch_bounds__(64) void k_dict_share(DecodeArgs a, int step, int count)
{
    const int s = (int)blockIdx.x;
    if (s >= count) return;
    const int blk = a.segFirst + s * step;
    if (blk >= a.nBlocks) return;
    const uint8_t *data = nullptr;
    int compLen = 0, cap = 0;
    if (uni(read_block_header(a, blk, data, compLen, cap)) != 0) return;
    InWindow win;
    win.lo = a.framed; win.hi = a.framed + a.framedLen;
    win.load(data);
    auto rd = [&](int pos) -> uint32_t {
        const uint8_t *p = data + pos;
        if (!win.covers(p, 1)) win.load(p);
        return win.byte_at(p);
    };
    int ip = 0;
    uint32_t op = 0, direct = 0;
    for (int n = 0; n < DICT_SHARE_SEQS && ip + 3 < compLen; n++) {
        const uint32_t t = rd(ip); ip++;
        uint32_t lit = t >> 4;
        if (lit == 15u) { uint32_t x; do { x = (ip < compLen) ? rd(ip) : 0u; ip++; lit += x; } while (x == 255u && ip < compLen); }
        ip += (int)lit; op += lit;
        if (ip + 2 > compLen) break;
        const uint32_t off = rd(ip) | (rd(ip + 1) << 8); ip += 2;
        uint32_t ml = t & 15u;
        if (ml == 15u) { uint32_t x; do { x = (ip < compLen) ? rd(ip) : 0u; ip++; ml += x; } while (x == 255u && ip < compLen); }
        ml += LZ4_MINMATCH;
        if (off > op) direct += min(ml, off - op);
        op += ml;
    }
    if (lane_id() == 0) { atomicAdd(&a.linkStat[8], direct); atomicAdd(&a.linkStat[9], op); }
}

void launch_dict_share(const DecodeArgs &a, int step, int count, hipStream_t s)
{
    if (a.linkStat && count > 0) hipLaunchKernelGGL(k_dict_share, dim3((unsigned)count), dim3(64), 0, s, a, step, count);
}

void launch_longest_stream(const DecodeArgs &a, hipStream_t s)
{
    if (a.streamFirst && a.nStreams > 0 && a.linkStat)
        hipLaunchKernelGGL(k_longest_stream, dim3((unsigned)((a.nStreams + 255) / 256)), dim3(256), 0, s, a);
}

// Second pass over the blocks [a.segFirst, a.segEnd) of linked streams, in two steps so that the caller can give
// the first one a longer range than the second (lists are 1 byte per output byte, pointers are 4).
void launch_linked_tolerant(const DecodeArgs &a, hipStream_t s)
{
    const int n = a.segEnd - a.segFirst;
    if (n <= 0 || !a.tolPool) return;
    hipMemsetAsync(a.tolCounter, 0, 4 * sizeof(uint32_t), s);
    hipLaunchKernelGGL(k_decode_tolerant, dim3((unsigned)n), dim3(64), 0, s, a);
}

// The pointer pass in two halves.  The first touches pointers only -- where every byte of the segment comes from
// is known from the tokens (lists) alone; the second reads DATA: the roots, the first of which lie in the block in
// front of the segment.  mi355lz4_decompress_linked_begin / _end run them apart so that the output of that block
// (the seam of a stream that is spread over several GPUs) may arrive in between.
void launch_linked_resolve_a(const DecodeArgs &a, hipStream_t s)
{
    const int n = a.segEnd - a.segFirst;
    if (n <= 0) return;
    if (a.tolPool && a.ptr && a.ptrCtl && a.ptrBad) {
        // (the stream flags follow the control block: a stream turned down in one segment gets its chance in the next)
        hipMemsetAsync(a.ptrCtl, 0, sizeof(PtrCtl) + sizeof(uint32_t) * (size_t)(a.streamFirst ? a.nStreams : 1), s);
        hipLaunchKernelGGL(k_ptr_expand, dim3((unsigned)n + 1u), dim3(256), 0, s, a);
        const unsigned items = ptr_grid(n);
        hipLaunchKernelGGL(k_ptr_jump, dim3(items), dim3(256), 0, s, a, 0, items);
    }
}

void launch_linked_resolve_b(const DecodeArgs &a, hipStream_t s)
{
    const int n = a.segEnd - a.segFirst;
    if (n <= 0) return;
    if (a.tolPool && a.ptr && a.ptrCtl && a.ptrBad) {
        const unsigned items = ptr_grid(n), few = std::min(items, 4096u);
        hipLaunchKernelGGL(k_ptr_fetch<true>, dim3(items), dim3(256), 0, s, a, items);
        // (what follows usually finds nothing to do: small grids that stride over the items)
        for (int pass = 1; pass < PTR_MAX_PASSES; pass++)
            hipLaunchKernelGGL(k_ptr_jump, dim3(few), dim3(256), 0, s, a, pass, items);
        hipLaunchKernelGGL(k_ptr_fetch<false>, dim3(few), dim3(256), 0, s, a, items);
        hipLaunchKernelGGL(k_ptr_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    }
    // whatever the pass above did not take (ptrBad, or no pool): the walk, block after block
    if (!a.streamFirst)
        hipLaunchKernelGGL(k_decode_fixup_regions, dim3((unsigned)((n + LZ4_WAVE - 1) / LZ4_WAVE)),
                           dim3(RPL_THREADS), 0, s, a);
    else if (a.nStreams > 0)
        hipLaunchKernelGGL(k_decode_fixup_linked, dim3((unsigned)a.nStreams), dim3(64), 0, s, a);
}

// One block of the segment fetched ahead of the others (a.onlyBlk): what a rank hands to its right neighbour when ONE
// linked stream is spread over several GPUs -- the neighbour then waits for one block's fetch, not for a range's.  The
// chasing fetch is complete unless it raises PtrCtl::lastOpen (a chain deeper than the first jump pass plus PTR_CHASE).
void launch_linked_fetch_block(const DecodeArgs &a, hipStream_t s)
{
    const int n = a.segEnd - a.segFirst;
    if (n <= 0 || !(a.tolPool && a.ptr && a.ptrCtl && a.ptrBad)) return;
    const unsigned items = ptr_grid(n);
    hipLaunchKernelGGL(k_ptr_fetch<true>, dim3(std::min(items, 4096u)), dim3(256), 0, s, a, items);
}

void launch_linked_resolve(const DecodeArgs &a, hipStream_t s)
{
    launch_linked_resolve_a(a, s);
    launch_linked_resolve_b(a, s);
}
