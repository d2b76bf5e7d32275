// Microbenchmark (development aid): LDS throughput of aligned vs byte-misaligned 8-byte accesses,
// with all 64 lanes or only a subset active.  One 64-thread workgroup per CU slot, many per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64u __attribute__((aligned(1)));
typedef uint32_t u32u __attribute__((aligned(1)));
template <int MODE>
__global__ __launch_bounds__(64) void k(uint64_t *out, int iters, int misalign, int activeLanes)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[8192];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) ((uint32_t *)buf)[i] = i;
    __syncthreads();
    uint64_t acc = 0;
    if (lane < activeLanes) {
        uint32_t a = ((uint32_t)(lane * 48) & 4080u) | ((uint32_t)misalign & 15u);          // spread over banks, optional byte misalignment
        for (int i = 0; i < iters; i++) {
            if (MODE == 0) { acc += *(const u64u *)&buf[a]; }                                   // read 8
            else if (MODE == 1) { *(u64u *)&buf[a + 4096] = acc + i; acc += a; }                // write 8
            else if (MODE == 2) { acc += *(const u32u *)&buf[a]; }                               // read 4
            else if (MODE == 3) { acc += buf[a]; }                                               // read 1
            else if (MODE == 5) { uint4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)&buf[a]) : "memory"); acc += v.x + v.w; }   // one ds_read_b128 at any alignment
            else if (MODE == 6) { uint4 v = make_uint4((uint32_t)acc, i, 3, 4); __builtin_memcpy(&buf[a + 4096], &v, 16); acc += a; }
            else if (MODE == 4) { acc += (uint64_t)__builtin_amdgcn_ds_bpermute((lane * 4 + 8) & 255, (int)acc + i); }
            a = (a + 80) & 4095u;
            if (misalign) a = (a & ~15u) | ((uint32_t)misalign & 15u); else a &= ~15u;
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}
template <int MODE>
static void run(const char *name, int misalign, int active)
{
    const int NB = 256 * 16, iters = 20000;
    static uint64_t *out = nullptr;
    if (!out) hipMalloc(&out, (size_t)NB * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(NB), dim3(64), 0, 0, out, iters, misalign, active);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(NB), dim3(64), 0, 0, out, iters, misalign, active); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 16 waves per CU: CU-level LDS cycles per wave-instruction
    printf("%-10s misalign=%d active=%2d : %.2f LDS cycles per wave-instruction per CU (2.4 GHz)\n", name, misalign, active, ms * 1e-3 * 2.4e9 / ((double)iters * 16));
}
int main()
{
    for (int active : {64, 16}) {
        run<0>("read8", 0, active); run<0>("read8", 1, active); run<0>("read8", 4, active);
        run<1>("write8", 0, active); run<1>("write8", 1, active); run<1>("write8", 4, active);
        run<2>("read4", 0, active); run<2>("read4", 1, active);
        run<3>("read1", 0, active);
        run<4>("bpermute", 0, active);
        run<5>("read16", 0, active); run<5>("read16", 1, active); run<5>("read16", 8, active);
        run<6>("write16", 0, active); run<6>("write16", 1, active);
    }
    return 0;
}
