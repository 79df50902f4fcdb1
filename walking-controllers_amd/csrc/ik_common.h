// Definitions shared by the two IK kernels (ik.hip: sweep on M = H + rho A'A;
// ik2.hip: null-space formulation).  Internal, not part of the ABI.
#pragma once
#include "wcqp_internal.h"
#include "mpc_device.h"

namespace wcqp_ik {

constexpr int kDof = 23;
constexpr int kNV = kDof + 6;          // 29
constexpr int kStateLen = WCQP_IK_STATE_LEN;
constexpr double kMixedTol = WCQP_IK_MIXED_TOL;      // ik4.hip: how close to [I B; 0 I] a base block has to be (include/wcqp.h)

struct IkDeviceParams {
    double lam[32];        // Lambda diagonal per variable (0 on the base)       base.cpp:64-67
    double kq[32];         // w_i * K_i per variable                              base.cpp:70-72,87-89
    double qreg[32];       // regularisation posture per variable (rad)
    double vlo[32], vhi[32];   // variable bounds; base = -/+ DBL_MAX            qp.cpp:39-49
    double Wn[9], Wc[9];
    double k_pos_com, k_pos_foot, k_att_foot, k_neck, kappa, rho, tol;
    int form, max_iter;
    // base-eliminated kernel (ik4.hip), indexed by JOINT column c = 0..22 (entries 23..31: 1.0 / 0.0)
    double sd[32], isd[32];    // sqrt(1 / Lambda_c), sqrt(Lambda_c): column scaling that turns Lambda into I
    double Lt[9];              // L' (row-major, upper triangular) of the neck weight W = L L'
    int fast_ok;               // CoM as constraint, every Lambda_c > 0, W positive definite
};

// rotation error component k of unskew(0.5 (E - E')), E = R Rd'     Utils.cpp:22-27
__device__ __forceinline__ double rot_err(const double* R, const double* Rd, int k) {
    const int a = (k + 2) % 3, b = (k + 1) % 3;
    const double eab = R[3 * a] * Rd[3 * b] + R[3 * a + 1] * Rd[3 * b + 1] + R[3 * a + 2] * Rd[3 * b + 2];
    const double eba = R[3 * b] * Rd[3 * a] + R[3 * b + 1] * Rd[3 * a + 1] + R[3 * b + 2] * Rd[3 * a + 2];
    return 0.5 * (eab - eba);
}

// Broadcast of lane `SRC` of each 32-lane group to every lane of that group: ds_swizzle in
// bit-mask mode (and = 0, or = SRC, xor = 0) moves data through the LDS crossbar without
// touching LDS memory, so no write -> read round trip is needed.
template <int SRC>
__device__ __forceinline__ double group_bcast(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), (SRC & 31) << 5);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), (SRC & 31) << 5);
    return __hiloint2double(hi, lo);
}


// xor-butterfly over the 32 lanes of a group (ds_swizzle bit-mask mode: and = 0x1f, xor = M)
template <int M>
__device__ __forceinline__ double group_xor(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x1f | (M << 10));
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x1f | (M << 10));
    return __hiloint2double(hi, lo);
}
// One DPP move of both halves of a double (dpp_ctrl must be a literal): quad_perm 0x00-0xff,
// row_mirror 0x140, row_half_mirror 0x141.  ~3 VALU instructions and no LDS-pipe round trip,
// against ~70 cycles for a ds_swizzle pair.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = wcqp::dpp_dword<CTRL>(__double2loint(v));
    const int hi = wcqp::dpp_dword<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// Row exchange inside a 32-lane group: v_permlane16_swap (new on gfx950) swaps the odd 16-lane rows
// of one operand with the even rows of the other, so with both operands equal the two results hold
// {own row, other row} in some order — a plain VALU op where ds_swizzle xor 16 is an LDS-pipe round trip.
__device__ __forceinline__ unsigned rows_max_u32(unsigned v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return max((unsigned)r[0], (unsigned)r[1]);
}
__device__ __forceinline__ void rows_pair(double v, double& d0, double& d1) {
    const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    d0 = __hiloint2double((int)rh[0], (int)rl[0]);
    d1 = __hiloint2double((int)rh[1], (int)rl[1]);
}
// Value of lane SRC of each 16-lane DPP row on every lane of that row: DPP row_newbcast (gfx90a+),
// one VALU move per dword, no LDS pipe.  SRC must be a compile-time constant.
template <int SRC>
__device__ __forceinline__ double row_bcast(double v) {
    const int lo = wcqp::dpp_dword<0x150 + (SRC & 15)>(__double2loint(v));
    const int hi = wcqp::dpp_dword<0x150 + (SRC & 15)>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// all-reduce over the 32 lanes of a group: four DPP steps inside each row of 16 (xor 1, xor 2 as
// quad permutes; then half-mirror and mirror, which pair up quads / octets that already agree),
// one row exchange between the two rows
__device__ __forceinline__ double group_max(double v) {
    v = fmax(v, dpp_move<0xB1>(v));     // quad_perm [1,0,3,2]
    v = fmax(v, dpp_move<0x4E>(v));     // quad_perm [2,3,0,1]
    v = fmax(v, dpp_move<0x141>(v));    // row_half_mirror
    v = fmax(v, dpp_move<0x140>(v));    // row_mirror
    double d0, d1;
    rows_pair(v, d0, d1);
    return fmax(d0, d1);
}
__device__ __forceinline__ double group_min(double v) {
    v = fmin(v, dpp_move<0xB1>(v));
    v = fmin(v, dpp_move<0x4E>(v));
    v = fmin(v, dpp_move<0x141>(v));
    v = fmin(v, dpp_move<0x140>(v));
    double d0, d1;
    rows_pair(v, d0, d1);
    return fmin(d0, d1);
}
// Lane of the largest |v| among the candidate lanes of a 32-lane group (ties: lowest lane).
// Pivot choice only needs the magnitude to float precision, so the search runs on a 32-bit key
// (float bits with 31 - lane in the low 5) and each butterfly step is ONE DPP-fused v_max_u32.
// `key_out` = the winning key (0 when there is no candidate); its upper 27 bits are |v| as a float.
__device__ __forceinline__ int group_argmax_abs(double v, bool candidate, int i, unsigned& key_out) {
    unsigned key = candidate ? ((__float_as_uint((float)fabs(v)) & ~31u) | (unsigned)(31 - i)) : 0u;
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xf, 0xf, false));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x4E, 0xf, 0xf, false));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x141, 0xf, 0xf, false));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x140, 0xf, 0xf, false));
    key = rows_max_u32(key);
    key_out = key;
    return 31 - (int)(key & 31u);
}
// value of `v` in wave lane `src` (wave-uniform): two v_readlane, result lives in SGPRs
__device__ __forceinline__ double lane_value(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}
// value of `v` in wave lane `src_byte / 4` (ds_bpermute: through the LDS crossbar, no LDS memory,
// one trip instead of the write -> wait -> read of a published column)
__device__ __forceinline__ double lane_gather(double v, int src_byte) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_byte, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_byte, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// all-reduce over the 16 lanes of a DPP row
__device__ __forceinline__ unsigned row_max_u32(unsigned key) {
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xf, 0xf, false));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x4E, 0xf, 0xf, false));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x141, 0xf, 0xf, false));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x140, 0xf, 0xf, false));
    return key;
}
__device__ __forceinline__ double row_min(double v) {
    v = fmin(v, dpp_move<0xB1>(v));
    v = fmin(v, dpp_move<0x4E>(v));
    v = fmin(v, dpp_move<0x141>(v));
    v = fmin(v, dpp_move<0x140>(v));
    return v;
}
__device__ __forceinline__ double row_max(double v) {
    v = fmax(v, dpp_move<0xB1>(v));
    v = fmax(v, dpp_move<0x4E>(v));
    v = fmax(v, dpp_move<0x141>(v));
    v = fmax(v, dpp_move<0x140>(v));
    return v;
}
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return v;
}
__device__ __forceinline__ unsigned mag_key(double v) { return __float_as_uint((float)fabs(v)) & ~31u; }

// lowest lane of this 32-lane group whose predicate holds (32 if none)
__device__ __forceinline__ int group_first(bool pred, int half) {
    const unsigned m = (unsigned)((__ballot(pred) >> (32 * half)) & 0xffffffffull);
    return m ? __ffs(m) - 1 : 32;
}

// ---------------------------------------------------------------------------------------------
// Goldfarb-Idnani dual active set over the joint-velocity bounds, shared by both IK kernels.
//
// Everything is expressed through full-space columns  tau_p = P e_p  of the projected inverse
// Hessian (supplied by `column_of_P`), so a working-set change never refactorises anything.
// The k x k system of the active bounds is carried as its EXPLICIT inverse Rinv, updated in
// O(k^2) by bordering (add) / rank-one downdate (drop); slot a of the working set is owned by
// lane a (its variable, sign and multiplier live in that lane's registers), so every step is a
// few lane-parallel passes instead of a serial k^3 factorisation.
// All control flow is uniform inside a 32-lane group; the two groups of a wave may diverge.
struct GiScratch {
    double* Rinv;   // [KMAX][LDR]
    double* vbuf;   // [32]
    double* zbuf;   // [32]
    double* tpb;    // [32]
    double* rvec;   // [32]  dual step per slot
    double* cvec;   // [32]
    int*    Wi;     // [32]  variable of slot a
};

template <int KMAX, int LDR, class ColumnFn>
__device__ __forceinline__ void gi_active_set(const GiScratch& w, int i, int half, bool var, double lo, double hi,
                                              double tol, int max_iter, double& nu, int& st_code, int& it,
                                              bool& in_w, double& my_sig, ColumnFn&& column_of_P) {
    const double inf = __builtin_inf();
    // slot state of lane i (slot index == lane index, slots 0..KMAX-1)
    bool s_live = false;
    int s_var = 0;
    double s_sg = 0.0, s_mu = 0.0;
    // columns sigma_a P e_{w_a} of the active bounds: entry i of slot a's column lives in lane i's
    // register tc[a] (static indices only: slot loops are unrolled over KMAX and dead slots carry a
    // zero dual step, so no predicate and no LDS image is needed)
    double tc[KMAX];
#pragma unroll
    for (int a = 0; a < KMAX; ++a) tc[a] = 0.0;
    int nW = 0, hiW = 0;
    if (i < KMAX) {
        for (int b = 0; b < KMAX; ++b) w.Rinv[i * LDR + b] = 0.0;
    }
    wcqp::wave_lds_fence();
    bool running = st_code == WCQP_STATUS_SOLVED;
    // First entering bound, empty working set, straight-line: most instances that violate a bound at
    // all violate exactly one, and the general loop below (nested, group-divergent exits) costs several
    // times what this does.  With W empty the step is the full step along tau_p: r = 0, z = tau_p,
    // Schur complement = P[p][p]; the bound takes slot 0.
    if (running) {
        const double v_hi = nu - hi, v_lo = lo - nu;
        const double viol = (var && i >= 6) ? fmax(v_hi, v_lo) : -inf;
        const double s0 = group_max(viol);
        if (s0 > tol) {
            ++it;
            const int p = group_first(viol == s0, half);
            w.zbuf[i] = v_hi >= v_lo ? 1.0 : -1.0;
            wcqp::wave_lds_fence();
            const double sig = w.zbuf[p];
            wcqp::wave_lds_fence();
            const double tp = column_of_P(p, sig);
            w.tpb[i] = tp;
            wcqp::wave_lds_fence();
            const double ppp = sig * w.tpb[p];               // P[p][p] > 0
            if (ppp > 0.0) {
                const double inz = 1.0 / ppp;
                const double t = s0 * inz;
                nu = fma(-t, tp, nu);
                if (i == 0) { w.Rinv[0] = inz; s_live = true; s_var = p; s_sg = sig; s_mu = t; w.Wi[0] = p; }
                tc[0] = tp;
                if (i == p) { in_w = true; my_sig = sig; }
                nW = 1; hiW = 1;
            } else {
                st_code = WCQP_STATUS_INFEASIBLE; running = false;
            }
            wcqp::wave_lds_fence();
        } else {
            running = false;
        }
    }
    while (running) {
        // most violated bound outside the working set
        const double v_hi = nu - hi, v_lo = lo - nu;
        const double viol = (var && i >= 6 && !in_w) ? fmax(v_hi, v_lo) : -inf;
        const double s0 = group_max(viol);
        if (!(s0 > tol)) break;
        if (it >= max_iter) { st_code = WCQP_STATUS_MAX_ITER; break; }
        ++it;
        const int p = group_first(viol == s0, half);
        w.zbuf[i] = v_hi >= v_lo ? 1.0 : -1.0;
        wcqp::wave_lds_fence();
        const double sig = w.zbuf[p];
        wcqp::wave_lds_fence();
        double s = s0;
        double tp = column_of_P(p, sig);                 // sig * P[i][p] on every variable lane, 0 elsewhere
        w.tpb[i] = tp;
        wcqp::wave_lds_fence();
        const double ppp = sig * w.tpb[p];               // P[p][p] > 0
        double mu_p = 0.0;
#pragma unroll 1
        for (int inner = 0; inner <= KMAX + 1; ++inner) {
            // dual step r = Rinv c,  c_a = sigma_a tp[w_a]
            const double c_a = s_live ? s_sg * w.tpb[s_var] : 0.0;
            w.cvec[i] = c_a;
            wcqp::wave_lds_fence();
            double r_a = 0.0;
            if (s_live) {
#pragma unroll 1
                for (int b = 0; b < hiW; ++b) r_a = fma(w.Rinv[i * LDR + b], w.cvec[b], r_a);
            }
            w.rvec[i] = r_a;
            wcqp::wave_lds_fence();
            // primal step z = tp - sum_a r_a Tc[a]
            double z = tp;
#pragma unroll
            for (int a = 0; a < KMAX; ++a) z = fma(-w.rvec[a], tc[a], z);
            w.zbuf[i] = z;
            wcqp::wave_lds_fence();
            const double nzv = sig * w.zbuf[p];          // Schur complement of the bordered system
            // a full working set (nW == n - meq) leaves no direction; otherwise dependence shows as
            // a vanishing Schur complement
            const double t2 = (nW < KMAX && nzv > 1e-10 * ppp) ? s / nzv : inf;
            const double ratio = (s_live && r_a > 0.0) ? s_mu / r_a : inf;
            const double t1 = group_min(ratio);
            const double t = fmin(t1, t2);
            if (!(t < inf)) { st_code = WCQP_STATUS_INFEASIBLE; running = false; break; }
            nu = fma(-t, z, nu);
            s_mu = s_live ? s_mu - t * r_a : s_mu;
            mu_p += t;
            s -= t * nzv;
            if (t2 <= t1) {
                // full step: p enters the first free slot; Rinv <- bordered inverse (needs r and nzv only)
                const int n = group_first(i < KMAX && !s_live, half);
                const double inz = 1.0 / nzv;
                if (s_live) {
#pragma unroll 1
                    for (int b = 0; b < hiW; ++b) w.Rinv[i * LDR + b] = fma(r_a * inz, w.rvec[b], w.Rinv[i * LDR + b]);
                    w.Rinv[i * LDR + n] = -r_a * inz;
                }
                wcqp::wave_lds_fence();
                if (i == n) {
#pragma unroll 1
                    for (int b = 0; b < KMAX; ++b) w.Rinv[n * LDR + b] = (b < hiW) ? -w.rvec[b] * inz : 0.0;
                    w.Rinv[n * LDR + n] = inz;
                    s_live = true; s_var = p; s_sg = sig; s_mu = mu_p;
                    w.Wi[n] = p;
                }
#pragma unroll
                for (int a = 0; a < KMAX; ++a) tc[a] = (a == n) ? tp : tc[a];
                if (i == p) { in_w = true; my_sig = sig; }
                ++nW;
                hiW = hiW > n + 1 ? hiW : n + 1;
                wcqp::wave_lds_fence();
                break;
            }
            // partial step: the blocking constraint leaves the working set; Rinv <- downdated inverse
            const int jd = group_first(ratio == t1, half);
            const int wdrop = w.Wi[jd];
            const double djj = w.Rinv[jd * LDR + jd];
            if (s_live && i != jd) {
                const double f = w.Rinv[i * LDR + jd] / djj;
#pragma unroll 1
                for (int b = 0; b < hiW; ++b) w.Rinv[i * LDR + b] = fma(-f, w.Rinv[jd * LDR + b], w.Rinv[i * LDR + b]);
            }
            wcqp::wave_lds_fence();
            if (i < KMAX) w.Rinv[i * LDR + jd] = 0.0;
            if (i == jd) {
#pragma unroll 1
                for (int b = 0; b < KMAX; ++b) w.Rinv[jd * LDR + b] = 0.0;
                s_live = false; s_mu = 0.0;
            }
            if (i == wdrop) { in_w = false; my_sig = 0.0; }
            --nW;
            ++it;
            wcqp::wave_lds_fence();
        }
        wcqp::wave_lds_fence();
    }
    // certificate: every bound holds and every active bound is tight, else the walk lost accuracy
    // (near-dependent working set) and the answer must not read SOLVED
    {
        const double dev = !(var && i >= 6) ? 0.0
                         : (in_w ? fabs(nu - (my_sig > 0.0 ? hi : lo)) : fmax(nu - hi, lo - nu));
        const double worst = group_max(dev == dev ? dev : inf);
        if (st_code == WCQP_STATUS_SOLVED && worst > 1e-9) st_code = WCQP_STATUS_NUMERIC;
        if (st_code == WCQP_STATUS_SOLVED && in_w) nu = my_sig > 0.0 ? hi : lo;
    }
}

// launch of the null-space kernel (ik2.hip)
int ik2_launch(const IkDeviceParams* d_prm, bool use_com, bool use_mfma, int batch,
               const double* JL, const double* JR, const double* JN, const double* JC,
               const double* q, const double* state, double* dq, int* status,
               unsigned* alo, unsigned* aup, double* ferr, int* iters, hipStream_t stream);

// launch of the 16-lanes-per-instance null-space kernel (ik3.hip; CoM-as-constraint form only)
int ik3_launch(const IkDeviceParams* d_prm, int batch,
               const double* JL, const double* JR, const double* JN, const double* JC,
               const double* q, const double* state, double* dq, int* status,
               unsigned* alo, unsigned* aup, double* ferr, int* iters, hipStream_t stream);

// launch of the base-eliminated range-space kernel (ik4.hip; MIXED free-floating Jacobians, CoM as constraint).
// Instances whose base blocks do not have the MIXED pattern come back with status WCQP_STATUS_STRUCTURE and dq = 0.
int ik4_launch(const IkDeviceParams* d_prm, int batch,
               const double* JL, const double* JR, const double* JN, const double* JC,
               const double* q, const double* state, double* dq, int* status,
               unsigned* alo, unsigned* aup, double* ferr, int* iters, hipStream_t stream);
// the general 16-lane kernel over the instances whose status reads WCQP_STATUS_STRUCTURE
int ik3_launch_list(const IkDeviceParams* d_prm, int batch,
                    const double* JL, const double* JR, const double* JN, const double* JC,
                    const double* q, const double* state, double* dq, int* status,
                    unsigned* alo, unsigned* aup, double* ferr, int* iters, hipStream_t stream);

}  // namespace wcqp_ik

namespace wcqp_tick { struct TickDev; }
namespace wcqp_ik {
// the 16-lane kernel with the tick pipeline's glue and post steps fused in (tick.hip)
int ik3_launch_tick(const void* d_prm, const wcqp_tick::TickDev& td,
                    const double* JL, const double* JR, const double* JN, const double* JC,
                    unsigned* alo, unsigned* aup, hipStream_t stream);
int ik4_launch_pair(const IkDeviceParams* d_prm, int batch,
                    const double* JL, const double* JR, const double* JN, const double* JC,
                    const double* q, const double* state, double* dq, int* status,
                    unsigned* alo, unsigned* aup, double* ferr, int* iters,
                    const wcqp_mpc::MpcDeviceConsts& c, const double* x0, const double* ref, int ref_len, const double* u_prev,
                    const double* hull_A, const double* hull_b, const int* hull_nc,
                    double* u0, int* mstatus, unsigned* mactive, double* mmargin, hipStream_t stream);
// the base-eliminated kernel with the SKEWED tick fused in: IK(t) + post step of tick t and the MPC chain of tick t + 1
// (tick_device.h); dense Jacobians, or the compact per-joint records of the tick's own kinematics kernel (td.compact)
// n_inner: ticks per launch (> 1 only without per-tick kinematics); td_dev: the same TickDev in device memory (td.phase is
// passed as a kernel argument, the copy's is not read)
int ik4_launch_tick(const void* d_prm, const wcqp_tick::TickDev& td, const wcqp_tick::TickDev* td_dev,
                    const double* JL, const double* JR, const double* JN, const double* JC,
                    unsigned* alo, unsigned* aup, int n_inner, int skip_last_mpc, hipStream_t stream, double* log_ferr = nullptr);
// a plan of steps in ONE launch (wcqp_qp_plan_*): d_recs = the records in device memory
int ik4_launch_plan(const IkDeviceParams* d_prm, int batch, const wcqp_qp_step* d_recs, int n_steps, int ways,
                    const wcqp_mpc::MpcDeviceConsts& c, hipStream_t stream, unsigned* d_queue, int queue_grid, bool ik_only);
int ik4_plan_queue_grid(int batch, int n_steps);      // work-queue form (ways = 0): as many waves as are resident at once
// its ticket counters: kPlanQueues of them + the count of finished waves, kPlanQueueStride bytes apart, all zero between launches
constexpr unsigned kPlanQueues = 32, kPlanQueueStride = 4352;
constexpr size_t kPlanQueueBytes = (size_t)(kPlanQueues + 1) * kPlanQueueStride;
// the MPC chain of tick t alone: primes the skewed tick after an upload
int ik4_launch_tick_prime(const wcqp_tick::TickDev& td, int t, hipStream_t stream);
}  // namespace wcqp_ik
