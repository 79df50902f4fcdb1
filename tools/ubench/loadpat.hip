// Diagnostic: how long do the INPUT LOADS of one IK launch take, as a function of the access pattern?
//   A  what ik4 does: lane j of a 16-lane group owns columns j and 16 + j, one 8-byte load per row and slot
//   B  adjacent columns: lane j owns columns 2j, 2j + 1, one 16-byte load per row
//   C  linear: the instance's block of each array read as consecutive 16-byte pieces (would need an LDS transpose)
// Same bytes, same grid (4 instances per wave64, one wave per workgroup), inputs rotated over K sets (> Infinity Cache).
// Usage: loadpat [batch] [sets] [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); std::exit(1); } } while (0)

struct Set { const double *JL, *JR, *JN, *JC, *q, *st; };

__device__ __forceinline__ double2 ldg2(const double* p) { return *reinterpret_cast<const double2*>(p); }
// 16 bytes from an address that is only 8-byte aligned
struct __attribute__((aligned(8))) d2u { double x, y; };
__device__ __forceinline__ double2 ldg2u(const double* p) {
    const d2u v = *reinterpret_cast<const d2u*>(p);
    return make_double2(v.x, v.y);
}

// RES: 0 = a small kernel; 1 = + 13.6 KB of LDS per workgroup (ik4's); 2 = + a 200-VGPR footprint as well (ik4's 214)
template <int PAT, int RES = 0>
__global__ __launch_bounds__(64, 2) void load_kernel(Set s, int batch, double* out) {
    __shared__ double lds[RES >= 1 ? 1700 : 1];
    const int lane = threadIdx.x, grp = lane >> 4, j = lane & 15;
    double pad[RES >= 2 ? 80 : 1];
    if (RES >= 2) {
#pragma unroll
        for (int k = 0; k < 80; ++k) { pad[k] = (double)(lane + k); asm volatile("" : "+v"(pad[k])); }
    }
    long inst = (long)blockIdx.x * 4 + grp;
    if (inst >= batch) inst = batch - 1;
    double acc = 0.0;
    if (PAT == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) { const int e = j + 16 * k; acc += s.st[inst * 110 + (e < 110 ? e : 0)]; }
        acc += s.q[inst * 23 + j];
        acc += s.q[inst * 23 + (j < 7 ? 16 + j : 0)];
        const int c1 = j < 13 ? 16 + j : 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) { acc += s.JL[inst * 174 + r * 29 + j]; acc += s.JL[inst * 174 + r * 29 + c1]; }
#pragma unroll
        for (int r = 0; r < 6; ++r) { acc += s.JR[inst * 174 + r * 29 + j]; acc += s.JR[inst * 174 + r * 29 + c1]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { acc += s.JN[inst * 87 + r * 29 + j]; acc += s.JN[inst * 87 + r * 29 + c1]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { acc += s.JC[inst * 87 + r * 29 + j]; acc += s.JC[inst * 87 + r * 29 + c1]; }
    } else if (PAT == 1) {
        // state 110 doubles = 55 pairs: 4 loads by lanes with pair index < 55; q: 23 doubles -> 12 pairs (the last reads one beyond: clamp)
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int e = j + 16 * k; const double2 v = ldg2(s.st + inst * 110 + 2 * (e < 55 ? e : 0)); acc += v.x + v.y; }
        { const double2 v = ldg2u(s.q + inst * 23 + 2 * (j < 11 ? j : 0)); acc += v.x + v.y; }
        const int c = 2 * (j < 14 ? j : 0);                  // columns 2j, 2j+1 (lane 14 would read column 28 and one beyond: leave it out here)
#pragma unroll
        for (int r = 0; r < 6; ++r) { const double2 v = ldg2u(s.JL + inst * 174 + r * 29 + c); acc += v.x + v.y; }
#pragma unroll
        for (int r = 0; r < 6; ++r) { const double2 v = ldg2u(s.JR + inst * 174 + r * 29 + c); acc += v.x + v.y; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { const double2 v = ldg2u(s.JN + inst * 87 + r * 29 + c); acc += v.x + v.y; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { const double2 v = ldg2u(s.JC + inst * 87 + r * 29 + c); acc += v.x + v.y; }
        // column 28 of every row: one more load instruction per array (8 bytes, lanes 0..5 / 0..2)
        acc += s.JL[inst * 174 + (j < 6 ? j : 0) * 29 + 28];
        acc += s.JR[inst * 174 + (j < 6 ? j : 0) * 29 + 28];
        acc += s.JN[inst * 87 + (j < 3 ? j : 0) * 29 + 28];
        acc += s.JC[inst * 87 + (j < 3 ? j : 0) * 29 + 28];
    } else {
        // linear pieces of 16 bytes: JL/JR 87 pairs each (6 loads), JN/JC 43.5 -> 44 (3 loads), state 55 (4), q 12 (1)
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int e = j + 16 * k; const double2 v = ldg2(s.st + inst * 110 + 2 * (e < 55 ? e : 0)); acc += v.x + v.y; }
        { const double2 v = ldg2u(s.q + inst * 23 + 2 * (j < 11 ? j : 0)); acc += v.x + v.y; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { const int e = j + 16 * k; const double2 v = ldg2(s.JL + inst * 174 + 2 * (e < 87 ? e : 0)); acc += v.x + v.y; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { const int e = j + 16 * k; const double2 v = ldg2(s.JR + inst * 174 + 2 * (e < 87 ? e : 0)); acc += v.x + v.y; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int e = j + 16 * k; const double2 v = ldg2u(s.JN + inst * 87 + 2 * (e < 43 ? e : 0)); acc += v.x + v.y; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int e = j + 16 * k; const double2 v = ldg2u(s.JC + inst * 87 + 2 * (e < 43 ? e : 0)); acc += v.x + v.y; }
    }
    if (RES >= 1) { lds[lane] = acc; __builtin_amdgcn_s_waitcnt(0); acc = lds[63 - lane]; }
    if (RES >= 2) {
#pragma unroll
        for (int k = 0; k < 80; ++k) { asm volatile("" : "+v"(pad[k])); acc += pad[k]; }
    }
    out[(long)blockIdx.x * 64 + lane] = acc;
}

int main(int argc, char** argv) {
    const int batch = argc > 1 ? std::atoi(argv[1]) : 4096;
    const int K = argc > 2 ? std::atoi(argv[2]) : 14;
    const int launches = argc > 3 ? std::atoi(argv[3]) : 280;
    const size_t per = (size_t)batch * (174 + 174 + 87 + 87 + 23 + 110) + 64;
    std::vector<Set> sets(K);
    double* all = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&all), per * K * sizeof(double)));
    CHECK(hipMemset(all, 0, per * K * sizeof(double)));
    for (int k = 0; k < K; ++k) {
        double* p = all + per * k;
        sets[k].JL = p; p += (size_t)batch * 174;
        sets[k].JR = p; p += (size_t)batch * 174;
        sets[k].JN = p; p += (size_t)batch * 87;
        sets[k].JC = p; p += (size_t)batch * 87;
        sets[k].st = p; p += (size_t)batch * 110;
        sets[k].q = p;
    }
    double* out = nullptr;
    const int grid = (batch + 3) / 4;
    CHECK(hipMalloc(reinterpret_cast<void**>(&out), (size_t)grid * 64 * sizeof(double)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[5] = {"A: 8 B per lane, columns j / 16+j (ik4)", "B: 16 B per lane, columns 2j / 2j+1", "C: linear 16 B pieces",
                            "A + 13.6 KB LDS per workgroup", "A + 13.6 KB LDS + 200 VGPRs"};
    for (int rep = 0; rep < 2; ++rep)
    for (int pat = 0; pat < 5; ++pat) {
        for (int w = 0; w < 2; ++w) {
            if (w == 1) CHECK(hipEventRecord(e0, 0));
            for (int it = 0; it < launches; ++it) {
                const Set& s = sets[it % K];
                if (pat == 0) hipLaunchKernelGGL(load_kernel<0>, dim3(grid), dim3(64), 0, 0, s, batch, out);
                else if (pat == 1) hipLaunchKernelGGL(load_kernel<1>, dim3(grid), dim3(64), 0, 0, s, batch, out);
                else if (pat == 2) hipLaunchKernelGGL(load_kernel<2>, dim3(grid), dim3(64), 0, 0, s, batch, out);
                else if (pat == 3) hipLaunchKernelGGL((load_kernel<0, 1>), dim3(grid), dim3(64), 0, 0, s, batch, out);
                else hipLaunchKernelGGL((load_kernel<0, 2>), dim3(grid), dim3(64), 0, 0, s, batch, out);
            }
            if (w == 1) CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
        }
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / launches;
        std::printf("batch %d  %-44s %8.2f us per launch  %7.1f GB/s\n", batch, names[pat], us, batch * 5240.0 / us * 1e-3);
    }
    return 0;
}
