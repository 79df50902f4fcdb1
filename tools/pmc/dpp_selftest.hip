// Self-test of the cross-lane helpers of ik_common.h (DPP/ds_swizzle control words are easy to
// get wrong and a wrong arg-max would only change pivot choices, not fail a parity test).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
#include "ik_common.h"
using namespace wcqp_ik;
__global__ void k(const double* in, double* omax, double* omin, int* ofirst, double* obc, double* ox16, int* oarg, double* ogat, double* orb) {
    const int t = threadIdx.x, half = t >> 5;
    const double v = in[blockIdx.x * 64 + t];
    omax[blockIdx.x * 64 + t] = group_max(v);
    omin[blockIdx.x * 64 + t] = group_min(v);
    ofirst[blockIdx.x * 64 + t] = group_first(v > 0.5, half);
    obc[blockIdx.x * 64 + t] = group_bcast<7>(v);
    ox16[blockIdx.x * 64 + t] = group_xor<16>(v);
    unsigned key;
    const int pl = group_argmax_abs(v - 0.5, (t & 31) < 29 && (t % 5) != 0, t & 31, key);
    oarg[blockIdx.x * 64 + t] = pl;
    ogat[blockIdx.x * 64 + t] = lane_gather(v, ((half << 5) + pl) << 2);
    orb[blockIdx.x * 64 + t] = row_bcast<11>(v) + row_bcast<0>(v) * 4.0;
}
int main() {
    const int nb = 64, n = nb * 64;
    std::vector<double> h(n);
    unsigned long long s = 1;
    for (auto& x : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = (double)(s >> 11) / 9007199254740992.0; }
    double *d, *a, *b, *e, *f, *gg, *dnb; int *c, *ga;
    hipMalloc(&d, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&e, n * 8); hipMalloc(&f, n * 8); hipMalloc(&c, n * 4); hipMalloc(&ga, n * 4); hipMalloc(&gg, n * 8); hipMalloc(&dnb, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<nb, 64>>>(d, a, b, c, e, f, ga, gg, dnb);
    std::vector<double> ra(n), rb(n), re(n), rf(n); std::vector<int> rc(n), rga(n); std::vector<double> rgg(n), rrb(n);
    hipMemcpy(ra.data(), a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(rb.data(), b, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(re.data(), e, n * 8, hipMemcpyDeviceToHost); hipMemcpy(rf.data(), f, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(rc.data(), c, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rga.data(), ga, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rgg.data(), gg, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(rrb.data(), dnb, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < n; ++t) { const int r0 = t & ~15; if (rrb[t] != h[r0 + 11] + h[r0] * 4.0) ++bad; }
    for (int g = 0; g < n / 32; ++g) {
        double mx = -1, mn = 2; int first = 32;
        for (int l = 0; l < 32; ++l) { double x = h[g * 32 + l]; mx = fmax(mx, x); mn = fmin(mn, x); if (x > 0.5 && first == 32) first = l; }
        // arg-max on float-truncated magnitudes (5 low mantissa bits dropped), ties -> lowest lane
        int arg = 31; unsigned best = 0;
        for (int l = 0; l < 29; ++l) {
            if (((g * 32 + l) % 64) % 5 == 0) continue;
            const float fv = (float)std::fabs(h[g * 32 + l] - 0.5);
            unsigned bits; std::memcpy(&bits, &fv, 4); bits &= ~31u;
            const unsigned key = bits | (unsigned)(31 - l);
            if (key > best) { best = key; arg = l; }
        }
        for (int l = 0; l < 32; ++l) {
            const int t = g * 32 + l;
            if (rga[t] != arg || rgg[t] != h[g * 32 + arg]) ++bad;
            if (ra[t] != mx || rb[t] != mn || rc[t] != first || re[t] != h[g * 32 + 7] || rf[t] != h[g * 32 + (l ^ 16)]) ++bad;
        }
    }
    std::printf("dpp_selftest: %s (%d mismatches of %d)\n", bad ? "FAIL" : "ok", bad, n);
    return bad != 0;
}
