// linked_replay.hpp -- in-order second pass of the deferred-copy decode of ONE long linked stream: the
// FALLBACK of the data-parallel pass in linked_ptr.hpp (a stream that pass turned down, or a call made
// without its scratch memory).
//
// The tolerant pass (decode_par.hpp / decode_seq.hpp, TolCtx) has decoded every dependent block of the
// stream in parallel and left, per block, a list of the matches it could not copy: those that start in the
// previous block's output (cbits/lz4.c:1883-1911, the ext-dict case of LZ4_decompress_safe_continue,
// :2347-2355) and those that read bytes such a match produces.  This pass walks the stream in order; for each
// block it puts the previous block's FINAL output (the dictionary) and the block's own output side by side
// in LDS -- positions -65536..65535 of the block's coordinate system are contiguous there -- and replays the
// list, 1024 entries at a time, with the same dependency rounds the decoder uses inside a batch.  One
// workgroup of 16 waves, 128 KiB of LDS: walked this way a stream is a serial chain of blocks, so one CU works
// on it (0.56 GB/s on text; the pointer pass does 39-55).
#pragma once

#include "decode_par.hpp"

namespace lz4dev {

#define RPL_HALF 65536
#define TOL_LIST_CAP 8192          // deferred entries per 64 KiB of block capacity (64 KiB of 8-byte entries); a block with more falls back
                                   // to the serial path.  Text-like data defers nearly every match (most bytes of a block
                                   // originate in its predecessor), ~6 500 per 64 KiB block.

struct __attribute__((aligned(16))) ReplayLds {
    uint8_t buf[2 * RPL_HALF + 64];     // [dictionary, ending at RPL_HALF | block], + slack for 16-byte chunk reads
};

#define RPL_THREADS 1024           // one workgroup of 16 waves walks a region: the replay is latency-bound, so the
                                   // entries of a block are copied 1024 at a time (dependency depth grows slower than the batch)

struct RplCtl {                    // workgroup-wide decisions and scratch of the region walk
    int32_t dposv[RPL_THREADS];    // destinations of the current super-batch (ascending)
    uint32_t doneBits[RPL_THREADS / 32];
    uint32_t doneCount;            // entries of the current super-batch copied so far
    int32_t action, r, hdr, region, count, size, cap, compLen, bad;
};

// workgroup copy of n bytes global -> LDS / LDS -> global, several loads in flight per thread
__device__ __forceinline__ void rpl_load(uint8_t *lds, const uint8_t *g, int n)
{
    const int tid = (int)threadIdx.x;
    const int n16 = n >> 4;
    int c = tid;
    for (; c + 3 * RPL_THREADS < n16; c += 4 * RPL_THREADS) {
        par_v4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) __builtin_memcpy(&v[k], as_global(g) + 16 * (c + k * RPL_THREADS), 16);
#pragma unroll
        for (int k = 0; k < 4; k++) *(par_v4u *)(lds + 16 * (c + k * RPL_THREADS)) = v[k];
    }
    for (; c < n16; c += RPL_THREADS) {
        par_v4 v;
        __builtin_memcpy(&v, as_global(g) + 16 * c, 16);
        *(par_v4u *)(lds + 16 * c) = v;
    }
    for (int x = (n16 << 4) + tid; x < n; x += RPL_THREADS) lds[x] = as_global(g)[x];
}
__device__ __forceinline__ void rpl_store(uint8_t *g, const uint8_t *lds, int n)
{
    const int tid = (int)threadIdx.x;
    const int n16 = n >> 4;
    for (int c = tid; c < n16; c += RPL_THREADS) {
        const par_v4 v = *(const par_v4u *)(lds + 16 * c);
        __builtin_memcpy(as_global(g) + 16 * c, &v, 16);
    }
    for (int x = (n16 << 4) + tid; x < n; x += RPL_THREADS) as_global(g)[x] = lds[x];
}

// Replay `n` deferred entries of a block whose tolerant output (size bytes, capacity cap) sits at
// L.buf + RPL_HALF and whose dictionary (dictLen bytes) ends at L.buf + RPL_HALF.  Called by the whole
// workgroup.  Returns false (to every thread) when an entry is not something the reference would have accepted
// (the block then goes to the exact serial decoder, which reports the reference's code).
#ifdef RPL_STATS
#define RPL_COUNT(i, v) do { if (threadIdx.x == 0 && rplStats) atomicAdd(&rplStats[i], (unsigned)(v)); } while (0)
#else
#define RPL_COUNT(i, v) do { } while (0)
#endif

__device__ bool replay_block(ReplayLds &L, RplCtl &C, const TolEntry *list, int n, int dictLen, int size, int cap,
                             unsigned *rplStats = nullptr)
{
    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    uint8_t *B = L.buf + RPL_HALF;                         // position p of the block lives at B[p]
    auto fetch = [&](int at) -> uint64_t {                 // the next super-batch is fetched while this one is copied
        return (at + tid < n) ? *as_global((const uint64_t *)(list + at + tid)) : 0ull;
    };
    uint64_t cur = fetch(0);
    if (tid == 0) C.bad = 0;
    __syncthreads();
    for (int base = 0; base < n; base += RPL_THREADS) {
        const int cnt = min(n - base, RPL_THREADS);
        const bool has = tid < cnt;
        const uint64_t nxt = fetch(base + RPL_THREADS);
        int dpos, ml, spos;
        tol_unpack(cur, dpos, ml, spos);
        cur = nxt;
        bool good = !has || (ml > 0 && spos < dpos && dpos + ml <= size && spos >= -dictLen);
        // a match that starts in the dictionary must end LASTLITERALS before the end of the output (:1884-1889)
        if (has && spos < 0 && dpos + ml > cap - LZ4_LASTLITERALS) good = false;
        if (!good) C.bad = 1;
        C.dposv[tid] = has ? dpos : 0x7fffffff;
        if (tid == 0) C.doneCount = 0;
        if (tid < RPL_THREADS / 32) {
            const int lo = tid * 32;                       // entries lo .. lo+31: bits of absent entries start out done
            C.doneBits[tid] = (cnt >= lo + 32) ? 0u : ((cnt <= lo) ? ~0u : (~0u << (cnt - lo)));
        }
        __syncthreads();
        if (C.bad) return false;
        // Entries are in stream order, so their destinations ascend: the entries my source [spos, srcHi) can
        // overlap are jlo..jhi, the last ones that start at or before its first / last byte.
        const int srcHi = min(spos + ml, dpos);            // bytes from dpos on are my own output
        int jlo = 0, jhi = -1;
        if (has && srcHi > spos && srcHi > C.dposv[0]) {
            const int xlo = max(spos, C.dposv[0]), xhi = srcHi - 1;
            int a = 0, b = 0;
#pragma unroll
            for (int stp = RPL_THREADS / 2; stp >= 1; stp >>= 1) {
                if (a + stp < cnt && C.dposv[a + stp] <= xlo) a += stp;
                if (b + stp < cnt && C.dposv[b + stp] <= xhi) b += stp;
            }
            jlo = a; jhi = min(b, tid - 1);                // only entries before me
        }
        const uint32_t off = (uint32_t)(dpos - spos);
        // A match that does not read its own output and is at most 64 bytes long is copied by its lane in ONE
        // LDS round trip (all chunk reads issued before the first write; the last chunk re-anchored at the end).
        // Anything else -- self-overlapping or long -- is copied by the whole wave, one after the other.
        const bool coop = off < (uint32_t)ml || ml > 64;
        const uint32_t cs = (ml >= 16) ? 16u : ((ml >= 8) ? 8u : 4u);        // chunk size; ml >= 4 always
        const uint32_t lastc = (uint32_t)ml - cs;
        bool pending = has;
        RPL_COUNT(2, 1);
        for (;;) {
            RPL_COUNT(1, 1);
            bool ready = pending;
            if (ready && jhi >= jlo) {                     // every entry jlo..jhi complete?
                for (int w = jlo >> 5; ready && w <= (jhi >> 5); w++) {
                    uint32_t m = ~0u;
                    if (w == (jlo >> 5)) m &= ~0u << (jlo & 31);
                    if (w == (jhi >> 5)) m &= ~0u >> (31 - (jhi & 31));
                    ready = (C.doneBits[w] & m) == m;
                }
            }
            for (uint64_t lm = __ballot(ready && coop); lm; lm &= lm - 1) {
                const int k = (int)__builtin_ctzll(lm);
                const int kd = __builtin_amdgcn_readlane(dpos, k), ks = __builtin_amdgcn_readlane(spos, k);
                const int kml = __builtin_amdgcn_readlane(ml, k);
                const uint32_t koff = (uint32_t)(kd - ks);
                wave_fence();
                if (koff >= (uint32_t)LZ4_WAVE) {
                    for (int c = 0; c < kml; c += LZ4_WAVE) {          // chunk c may read what chunk c-1 wrote
                        if (c + lane < kml) B[kd + c + lane] = B[ks + c + lane];
                        wave_fence();
                    }
                } else {
                    // short period: byte j of the match is byte j mod koff of its first period (j < 65536, koff < 64:
                    // the quotient from a float reciprocal is off by at most one)
                    const float rcp = 1.0f / (float)koff;
                    for (int j = lane; j < kml; j += LZ4_WAVE) {
                        uint32_t q = (uint32_t)((float)j * rcp);
                        int rem = j - (int)(q * koff);
                        if (rem < 0) rem += (int)koff; else if (rem >= (int)koff) rem -= (int)koff;
                        B[kd + j] = B[ks + rem];
                    }
                }
                wave_fence();
            }
            const bool mine = ready && !coop;
            if (__ballot(mine)) {
                par_v4 v0 = {0u, 0u, 0u, 0u}, v1 = v0, v2 = v0, v3 = v0;
                const uint32_t o1 = min(16u, lastc), o2 = min(32u, lastc), o3 = lastc;
                if (mine) {
                    if (cs == 16u) {
                        v0 = *(const par_v4u *)&B[spos];
                        if (ml > 16) v1 = *(const par_v4u *)&B[spos + (int)o1];
                        if (ml > 32) v2 = *(const par_v4u *)&B[spos + (int)o2];
                        if (ml > 48) v3 = *(const par_v4u *)&B[spos + (int)o3];
                    } else if (cs == 8u) {
                        const uint64_t a = *(const par_u64u *)&B[spos], c = *(const par_u64u *)&B[spos + (int)lastc];
                        v0.x = (uint32_t)a; v0.y = (uint32_t)(a >> 32); v1.x = (uint32_t)c; v1.y = (uint32_t)(c >> 32);
                    } else {
                        v0.x = *(const par_u32u *)&B[spos]; v1.x = *(const par_u32u *)&B[spos + (int)lastc];
                    }
                }
                if (mine) {
                    if (cs == 16u) {
                        *(par_v4u *)&B[dpos] = v0;
                        if (ml > 16) *(par_v4u *)&B[dpos + (int)o1] = v1;
                        if (ml > 32) *(par_v4u *)&B[dpos + (int)o2] = v2;
                        if (ml > 48) *(par_v4u *)&B[dpos + (int)o3] = v3;
                    } else if (cs == 8u) {
                        *(par_u64u *)&B[dpos] = ((uint64_t)v0.y << 32) | v0.x;
                        *(par_u64u *)&B[dpos + (int)lastc] = ((uint64_t)v1.y << 32) | v1.x;
                    } else {
                        *(par_u32u *)&B[dpos] = v0.x;
                        *(par_u32u *)&B[dpos + (int)lastc] = v1.x;
                    }
                }
                wave_fence();
            }
            pending = pending && !ready;
            // An entry is marked complete right behind its own copy: the LDS executes one wave's operations in
            // order, so whoever sees the bit also sees the bytes.  A wave that sees it in the same round merely
            // starts a dependent entry a round early.  One barrier per round: everyone has counted.
            wave_fence();
            if (ready) atomicOr(&C.doneBits[tid >> 5], 1u << (tid & 31));
            const uint64_t rm = __ballot(ready);
            if (rm && lane == 0) atomicAdd(&C.doneCount, (uint32_t)__builtin_popcountll(rm));
            __syncthreads();
            if (C.doneCount >= (uint32_t)cnt) break;
        }
        __syncthreads();                                   // nobody resets doneCount while it is still being read
    }
    __syncthreads();
    return true;
}

} // namespace lz4dev
