// Development aid: what v_permlane16_swap / v_permlane32_swap do to a register holding the lane id (gfx950).
// hipcc --offload-arch=gfx950 -o /tmp/pls scripts/micro/permlane_swap.hip && /tmp/pls
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    const unsigned x = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, x + 100u, false, false);
    auto q = __builtin_amdgcn_permlane32_swap(x, x + 100u, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane16_swap[0] (old = lane)", "permlane16_swap[1] (src = lane + 100)", "permlane32_swap[0]", "permlane32_swap[1]"};
    for (int a = 0; a < 4; a++) {
        printf("%s:\n", names[a]);
        for (int i = 0; i < 64; i++) printf("%4u%s", h[64 * a + i], (i & 15) == 15 ? "\n" : "");
    }
    return 0;
}
