// Latency microbenchmarks for the cross-lane / f64 idioms the IK kernel is built from (one wave).
// Diagnostic only.  Each test runs a dependent chain of N ops between two s_memtime stamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ik_common.h"
using namespace wcqp_ik;
#define T0 unsigned long long t0_, t1_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_), "+v"(x), "+v"(acc) :: "memory")
#define T1(slot) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_), "+v"(x), "+v"(acc) :: "memory"); if (threadIdx.x == 0) out[slot] = (long long)(t1_ - t0_)

__global__ void k(long long* out, double* sink, const double* in) {
    __shared__ double sm[256];
    const int lane = threadIdx.x, i = lane & 31, half = lane >> 5;
    double x = in[lane], y = in[64 + lane], acc = 0.0;
    sm[lane] = x; sm[64 + lane] = y;
    __syncthreads();
    { T0; T1(0); }                                                     // empty
    { T0;
#pragma unroll
      for (int n = 0; n < 64; ++n) { x = fma(x, y, 1e-9); }
      T1(1); }                                                         // 64 dependent f64 fma
    { T0; float f = (float)x;
#pragma unroll
      for (int n = 0; n < 64; ++n) { f = fmaf(f, 0.999f, 1e-9f); }
      x += f; T1(2); }                                                 // 64 dependent f32 fma
    { T0;
#pragma unroll
      for (int n = 0; n < 16; ++n) { x = wcqp::fast_rcp(x + 1.5); }
      T1(3); }                                                         // 16 dependent fast_rcp
    { T0; unsigned key = __float_as_uint((float)x);
#pragma unroll
      for (int n = 0; n < 64; ++n) { key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xf, 0xf, false)) + 1u; }
      x += key; T1(4); }                                               // 64 dependent dpp max (+add)
    { T0; unsigned key = __float_as_uint((float)x);
#pragma unroll
      for (int n = 0; n < 32; ++n) { key = rows_max_u32(key) + 1u; }
      x += key; T1(5); }                                               // 32 dependent permlane16 swap max
    { T0; int pl = 0; unsigned key;
#pragma unroll
      for (int n = 0; n < 16; ++n) { pl += group_argmax_abs(x + pl, i < 29, i, key); }
      x += pl; T1(6); }                                                // 16 dependent argmax
    { T0;
#pragma unroll
      for (int n = 0; n < 32; ++n) { const int s = __builtin_amdgcn_readlane(__double2loint(x), 5); x = fma(x, __hiloint2double(0x3ff00000, s & 1), 1e-9); }
      T1(7); }                                                         // 32 x (readlane -> sgpr -> fma)
    { T0;
#pragma unroll
      for (int n = 0; n < 32; ++n) { x = lane_gather(x, ((half << 5) + ((i + 1) & 31)) << 2) + 1e-9; }
      T1(8); }                                                         // 32 dependent bpermute pairs (+add)
    { T0;
#pragma unroll
      for (int n = 0; n < 32; ++n) { sm[lane] = x; wcqp::wave_lds_fence(); x = sm[(lane + 1) & 63] + 1e-9; wcqp::wave_lds_fence(); }
      T1(9); }                                                         // 32 dependent LDS write->read
    { T0;
#pragma unroll
      for (int n = 0; n < 32; ++n) { x = group_bcast<3>(x) + 1e-9; }
      T1(10); }                                                        // 32 dependent ds_swizzle pairs
    { T0;
#pragma unroll
      for (int n = 0; n < 32; ++n) { x = group_max(x) + 1e-9; }
      T1(11); }                                                        // 32 dependent group_max (f64)
    { T0;
#pragma unroll
      for (int n = 0; n < 32; ++n) { if (half == 0) x = fma(x, y, 1e-9); else x = fma(x, y, 2e-9); wcqp::pin_result(x); }
      T1(12); }                                                        // 32 divergent if/else fma
    { T0;   // 16 independent b128 broadcast reads then 32 fmas
      const double2* p = reinterpret_cast<const double2*>(sm + 64 * half);
      double2 v[16];
#pragma unroll
      for (int n = 0; n < 16; ++n) v[n] = p[n];
#pragma unroll
      for (int n = 0; n < 16; ++n) { acc = fma(v[n].x, y, acc); acc = fma(v[n].y, y, acc); }
      T1(13); }
    { T0;
#pragma unroll
      for (int n = 0; n < 64; ++n) { acc = fma(y, y, acc); x = fma(y, x, x); }   // 2 independent chains
      T1(14); }
    { T0;
#pragma unroll
      for (int n = 0; n < 32; ++n) { x = row_bcast<7>(x) + 1e-9; }
      T1(15); }                                                        // 32 dependent row_newbcast pairs (+add)
    sink[lane] = x + acc;
}
int main() {
    long long* d; double *s, *in; hipMalloc(&d, 16 * 8); hipMalloc(&s, 64 * 8); hipMalloc(&in, 128 * 8);
    std::vector<double> h(128); for (int n = 0; n < 128; ++n) h[n] = 0.5 + 0.001 * n;
    hipMemcpy(in, h.data(), 128 * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) k<<<1, 64>>>(d, s, in);
    long long r[16]; hipMemcpy(r, d, sizeof r, hipMemcpyDeviceToHost);
    const char* nm[] = {"empty", "fma_f64 x64", "fma_f32 x64", "fast_rcp x16", "dpp max x64", "permlane16 max x32", "argmax x16", "readlane->fma x32",
                        "bpermute pair x32", "lds write->read x32", "swizzle pair x32", "group_max f64 x32", "divergent if/else fma x32", "16 b128 reads + 32 fma", "2 chains fma x64", "row_newbcast pair x32"};
    const int cnt[] = {1, 64, 64, 16, 64, 32, 16, 32, 32, 32, 32, 32, 32, 1, 64, 32};
    for (int n = 0; n < 16; ++n) std::printf("%-28s %6lld cycles  (%.1f per op)\n", nm[n], r[n], (double)(r[n] - r[0]) / cnt[n]);
    return 0;
}
