// Diagnostic: how long do the OUTPUT STORES of one kinematics launch take, as a function of the access pattern?
//   A  what kin.hip does: 32 lanes per robot, lane i owns column i, one 8-byte store per Jacobian row (232-byte rows)
//   C  linear: every robot's block of each array written as consecutive 16-byte pieces
// Same bytes (4464 B per robot), outputs rotated over K sets.  Usage: storepat [batch] [sets] [launches] [blocks_per_cu (0 = one robot pair per workgroup)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); std::exit(1); } } while (0)
struct Set { double *JL, *JR, *JN, *JC, *st; };

template <bool NT> __device__ __forceinline__ void st1(double* p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
typedef double dvec2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ void st2a(double* p, double a, double b) {       // 16-byte aligned
    dvec2 v = {a, b};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dvec2*>(p)); else *reinterpret_cast<dvec2*>(p) = v;
}
typedef double dvec2u __attribute__((ext_vector_type(2), aligned(8)));
template <bool NT> __device__ __forceinline__ void st2u(double* p, double a, double b) {       // 8-byte aligned
    dvec2u v = {a, b};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dvec2u*>(p)); else *reinterpret_cast<dvec2u*>(p) = v;
}

template <int PATX>
__global__ __launch_bounds__(64) void store_kernel(Set s, int batch, const double* in) {
    constexpr int PAT = PATX & 1;
    constexpr bool NT = (PATX & 2) != 0;
    const int lane = threadIdx.x, half = lane >> 5, i = lane & 31;
    const long npairs = ((long)batch + 1) / 2;
    const double v = in[lane];
    for (long pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
        long inst = pair * 2 + half;
        if (inst >= batch) continue;
        if (PAT == 0) {
            if (i < 29) {
#pragma unroll
                for (int r = 0; r < 6; ++r) st1<NT>(s.JL + (inst * 174 + r * 29 + i), v + r);
#pragma unroll
                for (int r = 0; r < 6; ++r) st1<NT>(s.JR + (inst * 174 + r * 29 + i), v - r);
#pragma unroll
                for (int r = 0; r < 3; ++r) st1<NT>(s.JN + (inst * 87 + r * 29 + i), v * r);
#pragma unroll
                for (int r = 0; r < 3; ++r) st1<NT>(s.JC + (inst * 87 + r * 29 + i), v + 2 * r);
            }
            if (i < 12) { st1<NT>(s.st + (inst * 87 + i), v); st1<NT>(s.st + (inst * 87 + 12 + i), v); }
            if (i < 9) st1<NT>(s.st + (inst * 87 + 48 + i), v);
            if (i < 3) st1<NT>(s.st + (inst * 87 + 66 + i), v);
        } else {
            // pairs of doubles: JL/JR 87 pairs (3 stores by 32 lanes), JN/JC 43.5 (2 stores; 8-byte aligned), state 36 doubles
#pragma unroll
            for (int k = 0; k < 3; ++k) { const int e = i + 32 * k; if (e < 87) st2a<NT>(s.JL + inst * 174 + 2 * e, v, v + k); }
#pragma unroll
            for (int k = 0; k < 3; ++k) { const int e = i + 32 * k; if (e < 87) st2a<NT>(s.JR + inst * 174 + 2 * e, v, v - k); }
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int e = i + 32 * k; if (e < 43) st2u<NT>(s.JN + inst * 87 + 2 * e, v, v); }
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int e = i + 32 * k; if (e < 43) st2u<NT>(s.JC + inst * 87 + 2 * e, v, v); }
            if (i == 0) { st1<NT>(s.JN + (inst * 87 + 86), v); st1<NT>(s.JC + (inst * 87 + 86), v); }
            if (i < 18) st2u<NT>(s.st + inst * 87 + 2 * i, v, v);
        }
    }
}

int main(int argc, char** argv) {
    const int batch = argc > 1 ? std::atoi(argv[1]) : 65536;
    const int K = argc > 2 ? std::atoi(argv[2]) : 3;
    const int launches = argc > 3 ? std::atoi(argv[3]) : 60;
    const int per_cu = argc > 4 ? std::atoi(argv[4]) : 0;
    const size_t per = (size_t)batch * (174 + 174 + 87 + 87 + 87) + 64;
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
        sets[k].st = p;
    }
    double* in = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&in), 64 * sizeof(double)));
    CHECK(hipMemset(in, 0, 64 * sizeof(double)));
    const int grid = per_cu > 0 ? 256 * per_cu : (batch + 1) / 2;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[4] = {"A: 8 B per lane, one store per 232-byte row (kin)", "C: linear 16 B pieces", "A, nontemporal", "C, nontemporal"};
    const double bytes = (174 + 174 + 87 + 87 + 36) * 8.0;
    for (int rep = 0; rep < 2; ++rep)
    for (int pat = 0; pat < 4; ++pat) {
        for (int w = 0; w < 2; ++w) {
            if (w == 1) CHECK(hipEventRecord(e0, 0));
            for (int it = 0; it < launches; ++it) {
                const Set& s = sets[it % K];
                if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(64), 0, 0, s, batch, in);
                else if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(64), 0, 0, s, batch, in);
                else if (pat == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(64), 0, 0, s, batch, in);
                else hipLaunchKernelGGL(store_kernel<3>, dim3(grid), dim3(64), 0, 0, s, batch, in);
            }
            if (w == 1) CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
        }
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / launches;
        std::printf("batch %d grid %d  %-52s %8.2f us per launch  %7.1f GB/s\n", batch, grid, names[pat], us, batch * bytes / us * 1e-3);
    }
    return 0;
}
