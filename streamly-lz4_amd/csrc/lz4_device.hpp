// lz4_device.hpp -- shared device-side helpers for the gfx950 LZ4 kernels.
//
// Everything here is written for CDNA4 wave64: one wavefront owns one LZ4
// block, wave-uniform parse state lives in SGPRs (values derived from kernel
// arguments / readlane / readfirstlane stay scalar), lanes move bytes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define LZ4_WAVE 64

// Block-format constants (reference: cbits/lz4.c:214-235, cbits/lz4.h:557).
#define LZ4_MINMATCH 4
#define LZ4_LASTLITERALS 5
#define LZ4_MFLIMIT 12
#define LZ4_MAXDIST 65535

namespace lz4dev {

// The lane's number within its wavefront, from the hardware's lane mask count rather than from threadIdx: a function that
// never reads the work-item id does not need it handed over in v31 at a call, and a kernel that never reads it does not
// keep (or spill) the register it arrives in.  (One-dimensional workgroups: a wave is 64 consecutive threads.)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int64_t uni64(int64_t v)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Orders this wave's earlier global stores before its later global loads as
// far as the compiler is concerned.  Within one wavefront the vector memory
// pipeline already executes in order, so no s_waitcnt is needed.
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// Pointers rebuilt from integers (or passed through a non-inlined call) are "flat" to the compiler:
// flat loads wait for EVERY outstanding vector-memory operation before they issue and again before
// their data is used.  These casts assert what is true of every buffer the engine touches: it is
// device global memory.
#define LZ4_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const LZ4_GLOBAL T *as_global(const T *p) { return (const LZ4_GLOBAL T *)p; }
template <typename T>
__device__ __forceinline__ LZ4_GLOBAL T *as_global(T *p) { return (LZ4_GLOBAL T *)p; }

__device__ __forceinline__ int32_t load_le32(const uint8_t *p)
{
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

__device__ __forceinline__ void store_le32(uint8_t *p, int32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

// ---------------------------------------------------------------------------
// Input window: 256 bytes of the compressed stream held one dword per lane.
// A byte at a wave-uniform position is one v_readlane + shift: no memory
// latency on the token chain.  The window is reloaded with one coalesced load
// whenever the parse position leaves it.
// ---------------------------------------------------------------------------
struct InWindow {
    const uint8_t *lo;   // first readable byte of the framed buffer
    const uint8_t *hi;   // one past the last readable byte
    uintptr_t base;      // 4-aligned absolute address of lane 0's dword
    uint32_t w;          // this lane's dword

    __device__ __forceinline__ void load(const uint8_t *p)
    {
        base = (uintptr_t)p & ~(uintptr_t)3;
        const uint8_t *q = (const uint8_t *)(base + 4u * (uint32_t)lane_id());
        uint32_t v = 0;
        if (q >= lo && q + 4 <= hi) {
            v = *as_global((const uint32_t *)q);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (q + k >= lo && q + k < hi) v |= (uint32_t)as_global(q)[k] << (8 * k);
        }
        w = v;
    }
    __device__ __forceinline__ bool covers(const uint8_t *p, int n) const
    {
        uintptr_t a = (uintptr_t)p;
        return a >= base && a + (uintptr_t)n <= base + 256;
    }
    // p must be wave-uniform and inside the window.
    __device__ __forceinline__ uint32_t byte_at(const uint8_t *p) const
    {
        uint32_t idx = (uint32_t)((uintptr_t)p - base);
        uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)(idx >> 2));
        return (d >> ((idx & 3) * 8)) & 0xffu;
    }
};

// Wave-parallel byte copy, non-overlapping (literal runs: compressed stream <-> raw bytes).
// Long runs (incompressible data is one run per block) move 16 bytes per lane with aligned
// stores; short runs one byte per lane.
typedef uint32_t dev_v4 __attribute__((ext_vector_type(4)));
typedef dev_v4 dev_v4u __attribute__((aligned(1)));

// Every buffer these helpers are handed is device global memory (never LDS): saying so keeps the
// accesses global_load/global_store instead of flat_*.
__device__ __forceinline__ void wave_copy_bytes(uint8_t *dstGeneric, const uint8_t *srcGeneric, uint32_t n)
{
    LZ4_GLOBAL uint8_t *dst = as_global(dstGeneric);
    const LZ4_GLOBAL uint8_t *src = as_global(srcGeneric);
    const uint32_t lane = (uint32_t)lane_id();
    if (n >= 512u) {
        const uint32_t head = (uint32_t)((16u - ((uintptr_t)dstGeneric & 15u)) & 15u);
        if (lane < head) dst[lane] = src[lane];
        const uint32_t body = (n - head) >> 4;
        LZ4_GLOBAL dev_v4 *d16 = (LZ4_GLOBAL dev_v4 *)(dst + head);
        const LZ4_GLOBAL uint8_t *s16 = src + head;
#pragma unroll 2
        for (uint32_t i = lane; i < body; i += LZ4_WAVE)
            d16[i] = *(const LZ4_GLOBAL dev_v4u *)(s16 + ((size_t)i << 4));     // unaligned 16-byte load
        const uint32_t done = head + (body << 4);
        if (done + lane < n) dst[done + lane] = src[done + lane];
        return;
    }
#pragma unroll 2
    for (uint32_t i = lane; i < n; i += LZ4_WAVE) dst[i] = src[i];
}

} // namespace lz4dev
