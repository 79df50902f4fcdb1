// Device side of the DCM-MPC solve, shared by the stand-alone kernel (mpc.hip) and the tick pipeline's fused
// kernel (ik4.hip with TICK): one QP on the 16 lanes of a DPP row.  Internal, not ABI.
//
// Reference path replaced (citations relative to /root/reference/modules/Walking_module):
//   MPCSolver::{setConstraintsMatrix,setBounds,setGradient,solve,getSolution}   src/MPCSolver.cpp:76-322
//   WalkingController::solve (u0 read-out + hull-margin check)   src/WalkingDCMModelPredictiveController.cpp:491-521
// Why this is not an ADMM loop: see mpc.hip.
#pragma once
#include <limits>
#include "wcqp_internal.h"
#include "gptr.h"

namespace wcqp_mpc {

constexpr int kLanesPerInstance = 16;   // one DPP row per instance, 4 instances per wave
constexpr int kInstPerWave = 64 / kLanesPerInstance;

struct MpcDeviceConsts {
    wcqp::GPtr<const double> Gr;     // (N+1) x 2 x 2   (GPtr: the struct also travels inside TickDev, which kernels read from device memory)
    double Gx[4], Gu[4], S0[4];
    double feas_tol, hull_tol;
    int N;
};

#if defined(__HIPCC__)
// candidate 0: no row; 1..8: single row e = id-1; 9..36: row pairs (e < f)
// pair k (0..27) = rows (e < f) in the order {0,1},{0,2},...,{6,7}; four bits per entry, packed into literals: a table in memory
// would be a global load, and where the MPC rides in the shadow of the IK's Jacobian loads (ik4.hip) vmcnt's in-order
// retirement makes such a load wait for every Jacobian
//   E = 0,0,0,0,0,0,0, 1,1,1,1,1,1, 2,2,2,2,2, 3,3,3,3, 4,4,4, 5,5, 6      F = 1,2,3,4,5,6,7, 2,3,4,5,6,7, 3,4,5,6,7, 4,5,6,7, 5,6,7, 6,7, 7
__device__ __forceinline__ int pair_e(int k) { return (int)(((k < 16 ? 0x2221111110000000ull : 0x655444333322ull) >> (4 * (k & 15))) & 7ull); }
__device__ __forceinline__ int pair_f(int k) { return (int)(((k < 16 ? 0x5437654327654321ull : 0x776765765476ull) >> (4 * (k & 15))) & 7ull); }
constexpr int kNumCand = 1 + 8 + 28;

// One DPP move of both halves of a double inside a row of 16 lanes (= one instance here).
template <int CTRL>
__device__ __forceinline__ double row_move(double v) {
    const int lo = wcqp::dpp_dword<CTRL>(__double2loint(v));
    const int hi = wcqp::dpp_dword<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int row_move(int v) { return wcqp::dpp_dword<CTRL>(v); }
// the four butterfly partners inside a row: xor 1, xor 2 (quad permutes), then half-mirror and
// mirror, which pair quads / octets whose lanes already agree
#define WCQP_ROW_STEPS(X) X(0xB1) X(0x4E) X(0x141) X(0x140)

// The reference window of one QP and its gain blocks, in registers: the loads are issued by mpc_window_loads and consumed
// by mpc_row_partial, so a caller can put other loads (the tick kernel: the IK's Jacobians) behind them and run the MPC
// arithmetic while those are in flight (vmcnt retires in order: what is issued first can be waited for first).
// One pass: stages t, t+16, t+32, t+48 (covers N <= 63); longer horizons add looping passes (mpc_row_extra_passes).
struct MpcLoads {
    double2 r[4], g0[4], g1[4];
};
// (ref: the batch's reference array, w0: element index of the first stage of this instance's window - 32-bit addressing, wcqp::at32)
__device__ __forceinline__ void mpc_window_loads(const MpcDeviceConsts& c, int t, const double2* __restrict__ ref, unsigned w0, int ref_len, MpcLoads& L)
{
    const double2* gp = reinterpret_cast<const double2*>(c.Gr.get());
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = t + k * kLanesPerInstance;
        const int ic = i <= c.N ? i : c.N;                // clamped: stays in bounds, weight zeroed in mpc_row_partial
        const int ir = ic < ref_len ? ic : ref_len - 1;   // MPCSolver.cpp:200-214 (constant tail)
        L.r[k] = *wcqp::at32(ref, (w0 + (unsigned)ir) * 16u);
        const double2* g = wcqp::at32(gp, (unsigned)ic * 32u);
        L.g0[k] = g[0]; L.g1[k] = g[1];
    }
}
// this lane's partial sum of u0_unc from the loaded window (partial, extra passes, then mpc_row_add_state on lane 0: the
// order of operations of mpc_row_solve, so the results are bit-identical)
__device__ __forceinline__ void mpc_row_partial(const MpcDeviceConsts& c, int t, const MpcLoads& L, double& ux, double& uy) {
    ux = 0.0; uy = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double m = (t + k * kLanesPerInstance) <= c.N ? 1.0 : 0.0;
        ux = fma(m * L.g0[k].x, L.r[k].x, fma(m * L.g0[k].y, L.r[k].y, ux));
        uy = fma(m * L.g1[k].x, L.r[k].x, fma(m * L.g1[k].y, L.r[k].y, uy));
    }
}
// the same with the gain blocks read from a copy of Gr in LDS at the time of use instead of being held in registers from
// the time of the loads (the tick kernel with fused kinematics: 32 VGPRs less across its kinematics phase); bit-identical
__device__ __forceinline__ void mpc_window_loads_ref_only(const MpcDeviceConsts& c, int t, const double2* __restrict__ ref, unsigned w0, int ref_len, MpcLoads& L)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = t + k * kLanesPerInstance;
        const int ic = i <= c.N ? i : c.N;
        const int ir = ic < ref_len ? ic : ref_len - 1;
        L.r[k] = *wcqp::at32(ref, (w0 + (unsigned)ir) * 16u);
    }
}
__device__ __forceinline__ void mpc_row_partial_lds(const MpcDeviceConsts& c, int t, const MpcLoads& L, const double* gr_lds, double& ux, double& uy) {
    ux = 0.0; uy = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = t + k * kLanesPerInstance;
        const int ic = i <= c.N ? i : c.N;
        const double2 g0 = *reinterpret_cast<const double2*>(gr_lds + 4 * ic), g1 = *reinterpret_cast<const double2*>(gr_lds + 4 * ic + 2);
        const double m = i <= c.N ? 1.0 : 0.0;
        ux = fma(m * g0.x, L.r[k].x, fma(m * g0.y, L.r[k].y, ux));
        uy = fma(m * g1.x, L.r[k].x, fma(m * g1.y, L.r[k].y, uy));
    }
}
// stages 64 .. N of a horizon longer than one pass (the shipped controllerHorizon: N = 200), loaded on the spot
__device__ __forceinline__ void mpc_row_extra_passes(const MpcDeviceConsts& c, int t, const double2* __restrict__ rp, int ref_len, double& ux, double& uy) {
    const double2* gp = reinterpret_cast<const double2*>(c.Gr.get());
    for (int base = 4 * kLanesPerInstance; base <= c.N; base += 4 * kLanesPerInstance) {
        double2 r[4], g0[4], g1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + t + k * kLanesPerInstance;
            const int ic = i <= c.N ? i : c.N;
            const int ir = ic < ref_len ? ic : ref_len - 1;
            r[k] = rp[ir];
            g0[k] = gp[2 * ic]; g1[k] = gp[2 * ic + 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double m = (base + t + k * kLanesPerInstance) <= c.N ? 1.0 : 0.0;
            ux = fma(m * g0[k].x, r[k].x, fma(m * g0[k].y, r[k].y, ux));
            uy = fma(m * g1[k].x, r[k].x, fma(m * g1[k].y, r[k].y, uy));
        }
    }
}
// lane 0 of the row: the x0 / u_prev terms of u0_unc
__device__ __forceinline__ void mpc_row_add_state(const MpcDeviceConsts& c, double2 xs, double2 up, double& ux, double& uy) {
    ux += c.Gx[0] * xs.x + c.Gx[1] * xs.y + c.Gu[0] * up.x + c.Gu[1] * up.y;
    uy += c.Gx[2] * xs.x + c.Gx[3] * xs.y + c.Gu[2] * up.x + c.Gu[3] * up.y;
}

// The arithmetic of one MPC QP on the 16 lanes of a DPP row from loaded operands: `uxy_in` = this lane's partial sum of
// u0_unc (the gain-weighted reference stages and, on lane 0, the x0 / u_prev terms), (nc, rax, ray, rb) = row count and
// this lane's hull row.  All 64 lanes of the wave must call it together (wave-level early out, LDS fences).
// a . u - b of a hull row, with the FMAs spelled out: left to -ffp-contract the product a_x u_x is sometimes shared with another
// expression (unfused) and sometimes not, by what surrounds the inlined code - and the kernels that inline this (mpc_condensed_kernel,
// qp_pair_kernel, qp_plan_kernel, the tick kernels) must agree bit for bit
__device__ __forceinline__ double row_res(double ax, double ay, double b, double x, double y) { return fma(ay, y, fma(ax, x, -b)); }
__device__ __forceinline__ void mpc_row_finish(const MpcDeviceConsts& c, int t, double ux, double uy,
                                               int nc, double rax, double ray, double rb, double (*s_hull)[4],
                                               double& u0x, double& u0y, int& status, unsigned& active, double& margin_out);

// One MPC QP on the 16 lanes of a DPP row (t = lane inside the row).  `rp`: the instance's reference window
// (stages >= ref_len repeat the last one, MPCSolver.cpp:200-214), `hset`: index of the hull row set, `s_hull`: 8 x 4
// doubles of LDS owned by this row.  Every lane of the row returns the same u0, status, active mask and margin.
// All 64 lanes of the wave must call it together (wave-level early out, LDS fences).
__device__ __forceinline__ void mpc_row_solve(const MpcDeviceConsts& c, int t, long inst,
                                              const double* __restrict__ x0, const double2* __restrict__ rp, int ref_len,
                                              const double* __restrict__ u_prev,
                                              const double* __restrict__ hull_A, const double* __restrict__ hull_b,
                                              const int* __restrict__ hull_nc, long hset, double (*s_hull)[4],
                                              double& u0x, double& u0y, int& status, unsigned& active, double& margin_out)
{
    // ---- u0_unc = sum_i Gr_i r_i + Gx x0 + Gu u_prev ---------------------------------
    const double2* gp = reinterpret_cast<const double2*>(c.Gr.get());
    double ux = 0.0, uy = 0.0;
    // 64 stages per pass: the four reference loads of a lane (stages t, t+16, t+32, t+48) are
    // issued back to back before any is consumed, so one HBM round trip covers the whole window
    // of the N = 50 benchmark instead of four serialized ones
    for (int base = 0; base <= c.N; base += 4 * kLanesPerInstance) {
        double2 r[4], g0[4], g1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + t + k * kLanesPerInstance;
            const int ic = i <= c.N ? i : c.N;                // clamped: stays in bounds, weight zeroed below
            const int ir = ic < ref_len ? ic : ref_len - 1;   // MPCSolver.cpp:200-214 (constant tail)
            r[k] = rp[ir];
            g0[k] = gp[2 * ic]; g1[k] = gp[2 * ic + 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double m = (base + t + k * kLanesPerInstance) <= c.N ? 1.0 : 0.0;
            ux = fma(m * g0[k].x, r[k].x, fma(m * g0[k].y, r[k].y, ux));
            uy = fma(m * g1[k].x, r[k].x, fma(m * g1[k].y, r[k].y, uy));
        }
    }
    if (t == 0) {
        const double2 xs = reinterpret_cast<const double2*>(x0)[inst];
        const double2 up = reinterpret_cast<const double2*>(u_prev)[inst];
        ux += c.Gx[0] * xs.x + c.Gx[1] * xs.y + c.Gu[0] * up.x + c.Gu[1] * up.y;
        uy += c.Gx[2] * xs.x + c.Gx[3] * xs.y + c.Gu[2] * up.x + c.Gu[3] * up.y;
    }
    // hull rows (lanes 0..7 of the instance own one row each)
    const int nc = hull_nc[hset];
    double rax = 0.0, ray = 0.0, rb = 0.0;
    if (t < WCQP_HULL_ROWS) {
        const double2 a = reinterpret_cast<const double2*>(hull_A)[hset * WCQP_HULL_ROWS + t];
        rb = hull_b[hset * WCQP_HULL_ROWS + t];
        rax = a.x; ray = a.y;
    }
    mpc_row_finish(c, t, ux, uy, nc, rax, ray, rb, s_hull, u0x, u0y, status, active, margin_out);
}

__device__ __forceinline__ void mpc_row_finish(const MpcDeviceConsts& c, int t, double ux, double uy,
                                               int nc, double rax, double ray, double rb, double (*s_hull)[4],
                                               double& u0x, double& u0y, int& status, unsigned& active, double& margin_out)
{
    nc = nc < 0 ? 0 : (nc > WCQP_HULL_ROWS ? WCQP_HULL_ROWS : nc);
    // 1 / |a| of this lane's own hull row (lanes 0..7; 0 for a zero row): the margin below divides by the norm
    const double rn2 = fma(ray, ray, rax * rax);
    const double irn = (t < WCQP_HULL_ROWS && rn2 > 0.0) ? wcqp::fast_rsqrt(rn2) : 0.0;
    // butterfly over the row (DPP, no LDS-pipe round trips): every lane ends with the same sum
#define WCQP_SUM_STEP(C) ux += row_move<C>(ux); uy += row_move<C>(uy);
    WCQP_ROW_STEPS(WCQP_SUM_STEP)
#undef WCQP_SUM_STEP

    // ---- projection onto the polygon in the Sigma0^-1 metric --------------------------
    const double s00 = c.S0[0], s01 = c.S0[1], s10 = c.S0[2], s11 = c.S0[3];
    double best_cost = std::numeric_limits<double>::infinity();
    double best_x = ux, best_y = uy;
    unsigned best_mask = 0;
    int best_id = kNumCand;
    // Early out (wave-uniform): when the unconstrained optimum of every instance of this wave
    // already satisfies its hull rows, candidate 0 wins by construction (cost 0, lowest id) and the
    // 37-candidate enumeration is skipped.  Same feasibility test as the enumeration applies to
    // candidate 0, so the result is identical either way.
    const bool row_violated = t < nc && row_res(rax, ray, rb, ux, uy) > c.feas_tol;
    const bool none_violated = __ballot(row_violated) == 0ull;        // wave-uniform
    if (none_violated) {
        best_cost = 0.0; best_id = 0;
    } else {
    // the rows go to LDS only now: the enumeration is their one reader
    if (t < WCQP_HULL_ROWS) { s_hull[t][0] = rax; s_hull[t][1] = ray; s_hull[t][2] = rb; s_hull[t][3] = irn; }
    wcqp::wave_lds_fence();
    for (int id = t; id < kNumCand; id += kLanesPerInstance) {
        int e = -1, f = -1;
        if (id >= 1 && id <= 8) e = id - 1;
        else if (id > 8) { e = pair_e(id - 9); f = pair_f(id - 9); }
        if (e >= nc || f >= nc) continue;
        double px = ux, py = uy, cost = 0.0;
        unsigned mask = 0;
        bool ok = true;
        if (e >= 0) {
            const double aex = s_hull[e][0], aey = s_hull[e][1];
            const double sex = s00 * aex + s01 * aey, sey = s10 * aex + s11 * aey;   // Sigma0 a_e
            const double ree = aex * sex + aey * sey;
            const double re  = row_res(aex, aey, s_hull[e][2], ux, uy);
            mask = 1u << e;
            if (f < 0) {
                ok = ree > 0.0;
                const double mu = ok ? re * wcqp::fast_rcp(ree) : 0.0;
                px = ux - sex * mu; py = uy - sey * mu;
                cost = mu * re;
            } else {
                const double afx = s_hull[f][0], afy = s_hull[f][1];
                const double sfx = s00 * afx + s01 * afy, sfy = s10 * afx + s11 * afy;
                const double rff = afx * sfx + afy * sfy;
                const double ref_ = aex * sfx + aey * sfy;
                const double rf  = row_res(afx, afy, s_hull[f][2], ux, uy);
                const double det = ree * rff - ref_ * ref_;
                ok = det > 1e-12 * ree * rff;                 // parallel rows have no vertex
                const double idet = ok ? wcqp::fast_rcp(det) : 0.0;
                const double mue = (rff * re - ref_ * rf) * idet;
                const double muf = (ree * rf - ref_ * re) * idet;
                px = ux - sex * mue - sfx * muf; py = uy - sey * mue - sfy * muf;
                cost = mue * re + muf * rf;
                mask |= 1u << f;
            }
        }
        for (int k = 0; k < nc; ++k) {
            const double res = row_res(s_hull[k][0], s_hull[k][1], s_hull[k][2], px, py);
            ok = ok && (k == e || k == f || res <= c.feas_tol);
        }
        if (ok && (cost < best_cost || (cost == best_cost && id < best_id))) {
            best_cost = cost; best_x = px; best_y = py; best_mask = mask; best_id = id;
        }
    }
    }
#define WCQP_MIN_STEP(C) {                                                              \
        const double oc = row_move<C>(best_cost), ox = row_move<C>(best_x), oy = row_move<C>(best_y); \
        const int om = row_move<C>((int)best_mask), oi = row_move<C>(best_id);                 \
        if (oc < best_cost || (oc == best_cost && oi < best_id)) {                             \
            best_cost = oc; best_x = ox; best_y = oy; best_mask = (unsigned)om; best_id = oi;  \
        } }
    // (after the early out every lane of a row already holds the same candidate 0 - u0_unc was all-reduced above - and the
    // arg-min butterfly, 20 instructions a step, would only confirm it)
    if (!none_violated) { WCQP_ROW_STEPS(WCQP_MIN_STEP) }
#undef WCQP_MIN_STEP
    // signed distance to the hull boundary (computeMargin semantics): every row lane evaluates its
    // own row, row-min by DPP
    double margin = (t < nc && irn > 0.0) ? -row_res(rax, ray, rb, best_x, best_y) * irn : std::numeric_limits<double>::infinity();
#define WCQP_MARGIN_STEP(C) margin = fmin(margin, row_move<C>(margin));
    WCQP_ROW_STEPS(WCQP_MARGIN_STEP)
#undef WCQP_MARGIN_STEP
    int st = best_id < kNumCand ? WCQP_STATUS_SOLVED : WCQP_STATUS_INFEASIBLE;
    // WalkingController::solve: computeMargin(u0) < -tolerance => failure (cpp:513-517)
    if (st == WCQP_STATUS_SOLVED && margin < -c.hull_tol) st = WCQP_STATUS_OUTSIDE_HULL;
    u0x = best_x; u0y = best_y; status = st; active = best_mask; margin_out = margin;
    wcqp::wave_lds_fence();            // s_hull may be reused by the caller
}
#endif

}  // namespace wcqp_mpc
