// Microbenchmark (development aid): cost of per-lane byte-misaligned 8-byte global loads on gfx950
// versus aligned dword loads.  hipcc --offload-arch=gfx950 -O3 unaligned_load.hip -o /tmp/ul && /tmp/ul
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64u __attribute__((aligned(1)));
__global__ void k_unaligned(const uint8_t *src, uint64_t *out, int iters, int stride)
{
    const uint8_t *p = src + (size_t)blockIdx.x * 65536 + threadIdx.x;       // lane l at byte offset l
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) { acc += *(const u64u *)(p + (size_t)(acc & 1) + (size_t)i * stride); }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void k_aligned3(const uint8_t *src, uint64_t *out, int iters, int stride)
{
    const uint8_t *p = src + (size_t)blockIdx.x * 65536 + threadIdx.x;
    uint64_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint8_t *q = p + (size_t)(acc & 1) + (size_t)i * stride;
        const uint32_t *a = (const uint32_t *)((uintptr_t)q & ~(uintptr_t)3);
        const uint32_t sh = ((uintptr_t)q & 3) * 8;
        const uint32_t d0 = a[0], d1 = a[1], d2 = a[2];
        const uint64_t lo = ((uint64_t)d1 << 32) | d0, hi = d2;
        acc += sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void k_scatter8(const uint8_t *src, uint64_t *out, int iters)
{
    const uint8_t *p = src + (size_t)blockIdx.x * 65536;
    uint64_t acc = threadIdx.x * 977;
    for (int i = 0; i < iters; i++) { acc += *(const u64u *)(p + ((acc * 2654435761u) >> 7 & 0xFFF7)); }   // random 8B in 64 KiB
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main()
{
    const int NB = 5120, iters = 900;
    uint8_t *src; uint64_t *out;
    hipMalloc(&src, (size_t)NB * 65536 + 4096); hipMalloc(&out, (size_t)NB * 64 * 8);
    hipMemset(src, 1, (size_t)NB * 65536 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_unaligned, dim3(NB), dim3(64), 0, 0, src, out, iters, 64); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("unaligned 8B/lane, dependent chain: %.3f ms -> %.0f cycles/iter/wave @2.4GHz (20 waves/CU)\n", ms, ms * 1e-3 * 2.4e9 / iters);
        hipEventRecord(e0); hipLaunchKernelGGL(k_aligned3, dim3(NB), dim3(64), 0, 0, src, out, iters, 64); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("3 aligned dwords + shift:           %.3f ms -> %.0f cycles/iter/wave\n", ms, ms * 1e-3 * 2.4e9 / iters);
        hipEventRecord(e0); hipLaunchKernelGGL(k_scatter8, dim3(NB), dim3(64), 0, 0, src, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("scattered 8B/lane in 64 KiB:        %.3f ms -> %.0f cycles/iter/wave\n", ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
