// Semantics of gfx950's global_load_lds_dwordx4 as wcqp_ik::stage_jcn (the staging experiment of round 4: profiles/r04_lds_stage.patch, profiles/r04_lds_stage_ab.txt) uses it, checked on the GPU:
//   part 1: saddr + voffset form, M0 = LDS byte address: lane l's 16 bytes land at M0 + 16 l, whatever its global address; a global base
//           that is only 8-byte aligned is fine
//   part 2: stage_jcn itself (instruction offsets 1024 / 2048 apply to the global AND the LDS address; lanes 46..63 of the third
//           instruction sit out): the staging buffer holds the 2784 bytes of the robot group's J_com and J_neck blocks, bit for bit,
//           and nothing behind them is touched
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_dma_test.hip -o /tmp/lds_dma_test && /tmp/lds_dma_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// (the helper of profiles/r04_lds_stage.patch, as it was in csrc/ik_common.h while the experiment ran)
namespace wcqp_ik {
constexpr int kNV = 29;
// A wave's share of J_com and J_neck (four robots x 3 x 29 doubles each: 2784 contiguous bytes per array) from memory STRAIGHT INTO LDS
// - gfx950's global_load_lds_dwordx4: lane l's 16 bytes land at M0 + 16 l, no register is written (tools/ubench/lds_dma_test.hip checks the
// semantics, with a base that is only 8-byte aligned too).  What that buys qp_plan_kernel: these 5.5 KB of a record's 25 KB are asked for
// while the wave is still busy with the PREVIOUS record - bytes in flight that cost none of the 255 registers the kernel runs at
// (DESIGN.md 4.5).  Three instructions per array (174 chunks of 16 bytes; lanes 46..63 of the third sit out: what lies behind the block
// belongs to other robots, or to nobody).  The loads count on vmcnt like any other; the compiler does not see them: whoever reads the
// staging buffer waits (stage_wait).  kJStage doubles per wave: [0, 384) J_com, [384, 768) J_neck, robot i's block at i * 87.
constexpr int kJStage = 768, kJStageNeck = 384;
__device__ __forceinline__ void stage_jcn(const double* JC, const double* JN, unsigned blk, unsigned lane, const double* jstage) {
    const unsigned lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)jstage;
    const unsigned voff = blk * (unsigned)(4 * 3 * kNV * 8) + lane * 16u;
    // (the previous record's reads of the buffer have returned: lgkmcnt(0) - its new contents must not overtake them)
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, %4"
                 :: "s"(lds), "s"(lds + (unsigned)kJStageNeck * 8u), "v"(voff), "s"(JC), "s"(JN) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                 "s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024"
                 :: "s"(lds), "s"(lds + (unsigned)kJStageNeck * 8u), "v"(voff), "s"(JC), "s"(JN) : "memory", "m0");
    if (lane < 174 - 128)
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %2, %3 offset:2048\n\t"
                     "s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, %4 offset:2048"
                     :: "s"(lds), "s"(lds + (unsigned)kJStageNeck * 8u), "v"(voff), "s"(JC), "s"(JN) : "memory", "m0");
}
// every load of the wave has landed - the staged blocks among them
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
}  // namespace wcqp_ik

__global__ void k1(const double* __restrict__ src, double* __restrict__ dst, int mode) {
    __shared__ __attribute__((aligned(16))) double stage[3 * 128 + 64];
    const unsigned lane = threadIdx.x;
    for (int i = lane; i < 3 * 128 + 64; i += 64) stage[i] = -1.0;
    __syncthreads();
    const unsigned lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)stage;
    const unsigned voff = (lane * 16u) ^ (mode ? 32u : 0u);      // mode 1: lanes permuted in pairs of two chunks: the LDS side goes by LANE, not by address
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds + 64u + kk * 1024u), "v"(voff + kk * 1024u), "s"(src) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 3 * 128 + 64; i += 64) dst[i] = stage[i];
}

__global__ void k2(const double* __restrict__ JC, const double* __restrict__ JN, double* __restrict__ dst, unsigned blk) {
    __shared__ __attribute__((aligned(16))) double stage[wcqp_ik::kJStage + 32];
    const unsigned lane = threadIdx.x;
    for (int i = lane; i < wcqp_ik::kJStage + 32; i += 64) stage[i] = -1.0;
    __syncthreads();
    wcqp_ik::stage_jcn(JC, JN, blk, lane, stage);
    wcqp_ik::stage_wait();
    __syncthreads();
    for (int i = lane; i < wcqp_ik::kJStage + 32; i += 64) dst[i] = stage[i];
}

int main() {
    int bad_total = 0;
    {
        const int n = 3 * 128 + 64;
        std::vector<double> h(1024), o(n);
        for (int i = 0; i < 1024; ++i) h[i] = i;
        double *s, *d;
        if (hipMalloc(&s, 1024 * 8) != hipSuccess || hipMalloc(&d, n * 8) != hipSuccess) return 2;
        (void)hipMemcpy(s, h.data(), 1024 * 8, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, 0, s + (mode ? 1 : 0), d, mode);        // mode 1: global base only 8-byte aligned
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            (void)hipMemcpy(o.data(), d, n * 8, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < n; ++i) {
                double want = -1.0;
                if (i >= 8 && i < 8 + 384) { const int c = (i - 8) / 2, w = (i - 8) % 2; const int cs = mode ? (c ^ 2) : c; want = cs * 2 + w + (mode ? 1 : 0); }
                if (o[i] != want) { if (bad < 8) printf("part 1 mode %d: [%d] = %g, want %g\n", mode, i, o[i], want); ++bad; }
            }
            printf("part 1 mode %d: %d mismatches of %d\n", mode, bad, n);
            bad_total += bad;
        }
    }
    {
        const int robots = 12, per = 3 * wcqp_ik::kNV, n = wcqp_ik::kJStage + 32;
        std::vector<double> jc(robots * per + 1), jn(robots * per + 1), o(n);
        for (size_t i = 0; i < jc.size(); ++i) { jc[i] = 1000.0 + i; jn[i] = 5000.0 + i; }
        double *dc, *dn, *d;
        if (hipMalloc(&dc, jc.size() * 8) != hipSuccess || hipMalloc(&dn, jn.size() * 8) != hipSuccess || hipMalloc(&d, n * 8) != hipSuccess) return 2;
        (void)hipMemcpy(dc, jc.data(), jc.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(dn, jn.data(), jn.size() * 8, hipMemcpyHostToDevice);
        for (int mis = 0; mis < 2; ++mis)            // mis = 1: both arrays start 8 bytes into a 16-byte unit
            for (unsigned blk = 0; blk < 3; ++blk) {
                hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, dc + mis, dn + mis, d, blk);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                (void)hipMemcpy(o.data(), d, n * 8, hipMemcpyDeviceToHost);
                int bad = 0;
                for (int i = 0; i < n; ++i) {
                    double want = -1.0;
                    if (i < 4 * per) want = jc[mis + blk * 4 * per + i];
                    else if (i >= wcqp_ik::kJStageNeck && i < wcqp_ik::kJStageNeck + 4 * per) want = jn[mis + blk * 4 * per + (i - wcqp_ik::kJStageNeck)];
                    if (o[i] != want) { if (bad < 8) printf("part 2 blk %u mis %d: [%d] = %g, want %g\n", blk, mis, i, o[i], want); ++bad; }
                }
                printf("part 2 (stage_jcn) group %u, base %s: %d mismatches of %d\n", blk, mis ? "8-byte aligned" : "16-byte aligned", bad, n);
                bad_total += bad;
            }
    }
    printf(bad_total ? "FAILED\n" : "lds dma ok\n");
    return bad_total ? 1 : 0;
}
