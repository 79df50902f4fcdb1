// Diagnostic: prints what v_permlane32_swap / v_permlane16_swap do to a (vdst, src) pair, as the lane each result
// lane received its value from (a = first operand, b = second).  hipcc --offload-arch=gfx950 -o permlane_test permlane_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    const unsigned a = lane, b = 100 + lane;
    auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r32[0]; out[64 + lane] = r32[1]; out[128 + lane] = r16[0]; out[192 + lane] = r16[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 first ", "swap32 second", "swap16 first ", "swap16 second"};
    for (int v = 0; v < 4; ++v) {
        printf("%s:", names[v]);
        for (int r = 0; r < 4; ++r) printf("  row%d <- %s row%d", r, h[v * 64 + r * 16] >= 100 ? "b" : "a", (h[v * 64 + r * 16] % 100) / 16);
        printf("\n");
    }
    return 0;
}
