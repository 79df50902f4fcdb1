// How long does a kernel that does (almost) nothing take, as a function of grid size, LDS per workgroup and a
// per-wave spin of N cycles?  Separates launch/dispatch overhead from wave time (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_DOUBLES>
__global__ __launch_bounds__(64, 2) void spin(long long* out, int cycles) {
    __shared__ double sm[LDS_DOUBLES > 0 ? LDS_DOUBLES : 1];
    sm[threadIdx.x] = (double)threadIdx.x;
    const long long t0 = (long long)__builtin_readcyclecounter();
    long long t = t0;
    while (t - t0 < cycles) t = (long long)__builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = (t - t0) + (long long)sm[7];
}
template <int L>
static void run(const char* name, long long* d) {
    const int grids[] = {16, 256, 1024, 2048, 4096, 16384};
    const int spins[] = {0, 30000};
    for (int c : spins) for (int g : grids) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 5; ++i) spin<L><<<g, 64>>>(d, c);
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) spin<L><<<g, 64>>>(d, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::printf("%-10s spin %6d cycles  grid %6d : %7.2f us per launch\n", name, c, g, ms * 1000.0f / 50);
    }
}
int main() {
    long long* d; hipMalloc(&d, 16384 * 8);
    run<64>("lds 0.5KB", d);
    run<2528>("lds 20KB", d);
    return 0;
}
