// Jacobian QP-IK: one HIP kernel, 32 lanes per robot instance (two instances per wave64).
//
// Reference path replaced (citations relative to /root/reference/modules/Walking_module):
//   constants  WalkingQPIK::initializeMatrices          src/WalkingQPInverseKinematics.cpp:25-116
//   per tick   WalkingQPIK_osqp::{setHessianMatrix,setGradientVector,setLinearConstraintMatrix,
//              setBounds,solve,getSolution,get*FootError} src/WalkingQPInverseKinematics_osqp.cpp:135-454
//              WalkingQPIK_qpOASES::{same}               src/WalkingQPInverseKinematics_qpOASES.cpp:135-401
//
// QP, nu = [v_base(6); dq(dof)], n = dof + 6 = 29:
//      min 1/2 nu'H nu + g'nu   s.t.  A nu = b,   v_min <= dq <= v_max (qpOASES form only)
//      H = Lambda + Jn'Wn Jn (+ Jc'Wc Jc)  is only PSD (rank 26/29), so the kernel works on
//      M = H + rho A'A  (PD whenever the KKT matrix is regular; identical optimum and
//      identical multipliers because A nu = b holds at every iterate).
//
// Layout: lane i of a 32-lane group owns variable i: row i of the symmetric matrices lives
// in its REGISTERS (static indices, fully unrolled), and the only cross-lane traffic is one
// published column per elimination step, written to LDS once and read back as a broadcast.
//   1. M rows        M[i][:] = Lambda_i e_i + sum_r Cl[r][i] * Cr[r][:]     (18 stacked task rows)
//   2. Minv          symmetric sweep operator over the 29 pivots
//   3. G+ = Minv [A' g~],  S+ = [A; g~'] G+,  Sinv by a second sweep (15 pivots)
//   4. equality optimum  lambda = -Sinv (A Minv g~ + b),  nu = -Minv g~ - G lambda
//   5. bounds        Goldfarb-Idnani dual active set expressed through columns of the
//                    projected inverse  P = Minv - G Sinv G'  (no refactorisation per change)
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include "ik_common.h"

// The sweep kernel below is the round-1 A/B baseline: compiled only into diagnostic builds (-DWCQP_DIAG_KERNELS,
// tools/build_variant.sh); the product library carries ik4 (default), ik3 (general fall-back) and ik2 (CoM as cost).
using namespace wcqp_ik;
#ifdef WCQP_DIAG_KERNELS
#include "../../tools/diag/ik_sweep_kernel.h"      // ik_kernel<COM_AS_CONSTRAINT>: 380 lines that only diagnostic builds compile
#endif

// ======================================================================================
struct wcqp_ik_s {
    wcqp_ik_params p{};
    IkDeviceParams hp{};
    IkDeviceParams* d_prm = nullptr;
    wcqp::DeviceScratch scratch;
};

namespace {

int ensure_device(wcqp_ik_s* h) {
    if (h->d_prm) return WCQP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        std::fprintf(stderr, "[wcqp] no HIP device: the IK solve path has no CPU fallback\n");
        return WCQP_E_HIP;
    }
    WCQP_HIP_TRY(hipMalloc(&h->d_prm, sizeof(IkDeviceParams)));
    WCQP_HIP_TRY(hipMemcpy(h->d_prm, &h->hp, sizeof(IkDeviceParams), hipMemcpyHostToDevice));
    return WCQP_OK;
}

}  // namespace

namespace wcqp {
int ik_prepare(wcqp_ik_t h) { return h ? ensure_device(h) : WCQP_E_INVALID; }
const void* ik_device_params(wcqp_ik_t h) { return h ? h->d_prm : nullptr; }
bool ik_fast_ok(wcqp_ik_t h) { return h && h->hp.fast_ok != 0; }
}  // namespace wcqp

extern "C" {

int wcqp_ik_create(const wcqp_ik_params* params, wcqp_ik_t* out) {
    if (!params || !out) return WCQP_E_INVALID;
    if (params->dof != kDof) return WCQP_E_UNSUPPORTED;      // kernels are unrolled for iCub's 23 DoF
    if (params->form != WCQP_IK_FORM_QPOASES && params->form != WCQP_IK_FORM_OSQP) return WCQP_E_INVALID;
    if (params->algorithm < 0 || params->algorithm > WCQP_IK_ALG_BASE_ELIM) return WCQP_E_INVALID;
#ifndef WCQP_DIAG_KERNELS
    if (params->algorithm == WCQP_IK_ALG_SWEEP) return WCQP_E_UNSUPPORTED;      // diagnostic builds only
#endif
    if (params->jacobian_structure < WCQP_IK_JAC_AUTO || params->jacobian_structure > WCQP_IK_JAC_GENERAL) return WCQP_E_INVALID;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < r; ++c)          // the kernels use W J and J'W interchangeably: symmetric weights only
            if (params->neck_weight[3 * r + c] != params->neck_weight[3 * c + r] ||
                params->com_weight[3 * r + c] != params->com_weight[3 * c + r]) return WCQP_E_INVALID;
    wcqp_ik_s* h = new (std::nothrow) wcqp_ik_s();
    if (!h) return WCQP_E_NOMEM;
    h->p = *params;
    IkDeviceParams& d = h->hp;
    const double big = std::numeric_limits<double>::max();
    for (int i = 0; i < 32; ++i) {
        d.lam[i] = d.kq[i] = d.qreg[i] = 0.0;
        d.vlo[i] = -big; d.vhi[i] = big;                       // qp.cpp:39-43
    }
    for (int j = 0; j < kDof; ++j) {
        d.lam[6 + j] = params->joint_reg_weights[j];           // base.cpp:64-67
        d.kq[6 + j] = params->joint_reg_weights[j] * params->joint_reg_gains[j];   // base.cpp:70-72, 87-89
        d.qreg[6 + j] = params->joint_reg_rad[j];
        d.vlo[6 + j] = params->v_min[j];                       // qp.cpp:45-49
        d.vhi[6 + j] = params->v_max[j];
        if (!(params->v_min[j] <= params->v_max[j])) { delete h; return WCQP_E_INVALID; }
    }
    std::memcpy(d.Wn, params->neck_weight, sizeof(d.Wn));
    std::memcpy(d.Wc, params->com_weight, sizeof(d.Wc));
    d.k_pos_com = params->k_pos_com; d.k_pos_foot = params->k_pos_foot;
    d.k_att_foot = params->k_att_foot; d.k_neck = params->k_neck;
    // extra k_attFoot on the neck gradient term in the osqp back-end only (osqp.cpp:183,193)
    d.kappa = params->form == WCQP_IK_FORM_OSQP ? params->k_att_foot : 1.0;
    d.rho = params->rho > 0 ? params->rho : 1.0;
    d.tol = params->tol > 0 ? params->tol : 1e-12;
    d.form = params->form;
    d.max_iter = params->max_iter > 0 ? params->max_iter : 100;   // nWSR = 100, qp.cpp:312
    // base-eliminated kernel (ik4.hip): column scaling Lam^-1/2 and the factor W_neck = L L'
    d.fast_ok = params->use_com_as_constraint ? 1 : 0;
    for (int c = 0; c < 32; ++c) { d.sd[c] = 1.0; d.isd[c] = 1.0; }
    for (int j = 0; j < kDof; ++j) {
        const double w = params->joint_reg_weights[j];
        if (!(w > 0.0) || !std::isfinite(w)) { d.fast_ok = 0; continue; }
        d.sd[j] = std::sqrt(1.0 / w); d.isd[j] = std::sqrt(w);
    }
    {
        // Cholesky W = L L' (lower), stored as L' row-major; a neck weight that is not positive definite keeps the
        // general kernels
        const double* W = params->neck_weight;
        double L[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        bool pd = true;
        for (int c = 0; c < 3 && pd; ++c) {
            double dd = W[3 * c + c];
            for (int k = 0; k < c; ++k) dd -= L[3 * c + k] * L[3 * c + k];
            if (!(dd > 0.0) || !std::isfinite(dd)) { pd = false; break; }
            L[3 * c + c] = std::sqrt(dd);
            for (int r = c + 1; r < 3; ++r) {
                double v = W[3 * r + c];
                for (int k = 0; k < c; ++k) v -= L[3 * r + k] * L[3 * c + k];
                L[3 * r + c] = v / L[3 * c + c];
            }
        }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) d.Lt[3 * r + c] = pd ? L[3 * c + r] : 0.0;
        if (!pd) d.fast_ok = 0;
    }
    *out = h;
    return WCQP_OK;
}

int wcqp_ik_set_posture(wcqp_ik_t h, const double* joint_reg_rad) {
    if (!h || !joint_reg_rad) return WCQP_E_INVALID;
    for (int j = 0; j < kDof; ++j) { h->p.joint_reg_rad[j] = joint_reg_rad[j]; h->hp.qreg[6 + j] = joint_reg_rad[j]; }
    if (h->d_prm) {
        WCQP_HIP_TRY(hipDeviceSynchronize());
        WCQP_HIP_TRY(hipMemcpy(h->d_prm, &h->hp, sizeof(IkDeviceParams), hipMemcpyHostToDevice));
    }
    return WCQP_OK;
}

int wcqp_ik_destroy(wcqp_ik_t h) {
    if (!h) return WCQP_E_INVALID;
    if (h->d_prm) (void)hipFree(h->d_prm);
    h->scratch.release();
    delete h;
    return WCQP_OK;
}

int wcqp_ik_solve_device(wcqp_ik_t h, int32_t batch,
                         const double* J_left, const double* J_right, const double* J_neck, const double* J_com,
                         const double* q, const double* state,
                         double* dq, int32_t* status, uint32_t* active_lower, uint32_t* active_upper,
                         double* foot_err, int32_t* iters, void* stream) {
    if (!h || batch < 0) return WCQP_E_INVALID;
    if (!J_left || !J_right || !J_neck || !J_com || !q || !state || !dq || !status) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    const int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    // default: base elimination + range space (ik4.hip) for MIXED-representation Jacobians, with the general 16-lane
    // kernel behind it for instances (or handles) that do not qualify
    const bool want4 = h->p.algorithm == WCQP_IK_ALG_BASE_ELIM || h->p.algorithm == WCQP_IK_ALG_DEFAULT;
    if (want4 && h->hp.fast_ok && h->p.jacobian_structure != WCQP_IK_JAC_GENERAL) {
        const int rc4 = wcqp_ik::ik4_launch(h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state, dq, status,
                                            active_lower, active_upper, foot_err, iters, (hipStream_t)stream);
        if (rc4 != WCQP_OK || h->p.jacobian_structure == WCQP_IK_JAC_MIXED) return rc4;
        return wcqp_ik::ik3_launch_list(h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state, dq, status,
                                        active_lower, active_upper, foot_err, iters, (hipStream_t)stream);
    }
    const bool use16 = want4 || h->p.algorithm == WCQP_IK_ALG_NULLSPACE_16L;
    if (use16 && h->p.use_com_as_constraint)
        return wcqp_ik::ik3_launch(h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state, dq, status,
                                   active_lower, active_upper, foot_err, iters, (hipStream_t)stream);
    if (h->p.algorithm != WCQP_IK_ALG_SWEEP)
        return wcqp_ik::ik2_launch(h->d_prm, h->p.use_com_as_constraint != 0, h->p.algorithm != WCQP_IK_ALG_NULLSPACE, batch, J_left, J_right, J_neck, J_com,
                                   q, state, dq, status, active_lower, active_upper, foot_err, iters, (hipStream_t)stream);
#ifdef WCQP_DIAG_KERNELS
    const unsigned grid = (unsigned)((batch + 1) / 2);
    if (h->p.use_com_as_constraint)
        hipLaunchKernelGGL(ik_kernel<true>, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state,
                           dq, status, active_lower, active_upper, foot_err, iters);
    else
        hipLaunchKernelGGL(ik_kernel<false>, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state,
                           dq, status, active_lower, active_upper, foot_err, iters);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
#else
    return WCQP_E_UNSUPPORTED;            // refused at create already
#endif
}

}  // extern "C"

namespace wcqp {
// wcqp_qp_enqueue_steps: both calls of a record as one launch when they go to the same stream and the IK handle runs the
// base-eliminated kernel; WCQP_E_UNSUPPORTED = "make the two calls" (not an error)
int qp_pair_enqueue(wcqp_mpc_t mpc, wcqp_ik_t h, int batch, const wcqp_qp_step& s) {
    if (!mpc || !h || batch < 1 || !s.x0 || !s.J_left || s.mpc_stream != s.ik_stream) return WCQP_E_UNSUPPORTED;
    const bool want4 = h->p.algorithm == WCQP_IK_ALG_BASE_ELIM || h->p.algorithm == WCQP_IK_ALG_DEFAULT;
    if (!(want4 && h->hp.fast_ok && h->p.jacobian_structure != WCQP_IK_JAC_GENERAL)) return WCQP_E_UNSUPPORTED;
    if (s.ref_len < 1 || !s.ref || !s.u_prev || !s.hull_A || !s.hull_b || !s.hull_nc || !s.u0 || !s.mpc_status) return WCQP_E_INVALID;
    if (!s.J_right || !s.J_neck || !s.J_com || !s.q || !s.state || !s.dq || !s.ik_status) return WCQP_E_INVALID;
    int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    rc = mpc_prepare(mpc);
    if (rc != WCQP_OK) return rc;
    wcqp_mpc::MpcDeviceConsts c;
    mpc_device_consts(mpc, &c);
    rc = wcqp_ik::ik4_launch_pair(h->d_prm, batch, s.J_left, s.J_right, s.J_neck, s.J_com, s.q, s.state, s.dq, s.ik_status,
                                  s.active_lower, s.active_upper, s.foot_err, s.iters,
                                  c, s.x0, s.ref, s.ref_len, s.u_prev, s.hull_A, s.hull_b, s.hull_nc, s.u0, s.mpc_status, s.mpc_active, s.mpc_margin,
                                  (hipStream_t)s.ik_stream);
    if (rc != WCQP_OK || h->p.jacobian_structure == WCQP_IK_JAC_MIXED) return rc;
    return wcqp_ik::ik3_launch_list(h->d_prm, batch, s.J_left, s.J_right, s.J_neck, s.J_com, s.q, s.state, s.dq, s.ik_status,
                                    s.active_lower, s.active_upper, s.foot_err, s.iters, (hipStream_t)s.ik_stream);
}
}  // namespace wcqp

// ---- plans of steps (include/wcqp.h: wcqp_qp_plan_*) ------------------------------------------------------------------
struct wcqp_qp_plan_s {
    wcqp_mpc_t mpc = nullptr;
    wcqp_ik_t ik = nullptr;
    int batch = 0, n_steps = 0, ways = 1;
    wcqp_qp_step* d_recs = nullptr;
    unsigned* d_queue = nullptr;      // ways = 0: ticket counters + waves done (ik_common.h: kPlanQueues; qp_plan_kernel zeroes them itself)
    int queue_grid = 0;
    bool mpc_only = false;            // every record without its IK part: mpc_plan_kernel (mpc.hip)
    bool ik_only = false;             // every record without its MPC part: ik_plan_kernel (ik4.hip)
};

extern "C" {

int wcqp_qp_plan_create(wcqp_mpc_t mpc, wcqp_ik_t ik, int32_t batch, int32_t n_steps, const wcqp_qp_step* steps, int32_t ways,
                        wcqp_qp_plan_t* out) {
    if (!out || batch < 1 || n_steps < 1 || !steps || ways < WCQP_PLAN_WAYS_AUTO) return WCQP_E_INVALID;
    if (ways == WCQP_PLAN_WAYS_AUTO) {
        // enough workgroups for the dispatcher to even out the launch's ends: >= 16384 (8 x the resident wavefronts of an MI355X),
        // at least 4 ways, at most one per record (bench.py's rule)
        const int groups = (batch + 3) / 4;
        ways = (16384 + groups - 1) / groups;
        if (ways < 4) ways = 4;
        if (ways > n_steps) ways = n_steps;
    }
    // MPC-only plan: NO record has an IK part (J_left == NULL everywhere; `ik` may be NULL): BASELINE config 2 on its own
    bool mpc_only = true;
    for (int k = 0; k < n_steps; ++k) mpc_only = mpc_only && !steps[k].J_left;
    // IK-only plan: NO record has an MPC part (x0 == NULL everywhere): BASELINE config 3 on its own (qp_plan_kernel without its MPC share)
    bool ik_only = true;
    for (int k = 0; k < n_steps; ++k) ik_only = ik_only && !steps[k].x0;
    if (mpc_only && ik_only) return WCQP_E_INVALID;
    if (!mpc && !ik_only) return WCQP_E_INVALID;         // (`mpc` may be NULL for an IK-only plan, `ik` for an MPC-only one)
    for (int k = 0; k < n_steps; ++k) {
        const wcqp_qp_step& s = steps[k];
        if (!ik_only && (!s.x0 || !s.ref || s.ref_len < 1 || !s.u_prev || !s.hull_A || !s.hull_b || !s.hull_nc || !s.u0 || !s.mpc_status)) return WCQP_E_INVALID;
        if (!mpc_only && (!s.J_left || !s.J_right || !s.J_neck || !s.J_com || !s.q || !s.state || !s.dq || !s.ik_status)) return WCQP_E_INVALID;
    }
    if (mpc_only) {
        if (ways < 1) return WCQP_E_UNSUPPORTED;        // the work-queue form belongs to the IK + MPC kernel
        int rc0 = wcqp::mpc_prepare(mpc);
        if (rc0 != WCQP_OK) return rc0;
        wcqp_qp_plan_s* p = new (std::nothrow) wcqp_qp_plan_s();
        if (!p) return WCQP_E_NOMEM;
        p->mpc = mpc; p->batch = batch; p->n_steps = n_steps; p->ways = ways < n_steps ? ways : n_steps; p->mpc_only = true;
        if (hipMalloc(reinterpret_cast<void**>(&p->d_recs), (size_t)n_steps * sizeof(wcqp_qp_step)) != hipSuccess) { delete p; return WCQP_E_NOMEM; }
        if (hipMemcpy(p->d_recs, steps, (size_t)n_steps * sizeof(wcqp_qp_step), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(p->d_recs); delete p; return WCQP_E_HIP;
        }
        *out = p;
        return WCQP_OK;
    }
    if (!ik) return WCQP_E_INVALID;
    // one launch walks through the records: that is the base-eliminated kernel on Jacobians the caller declares MIXED (no
    // fall-back launch behind it), with the MPC on the IK's lanes
    const bool want4 = ik->p.algorithm == WCQP_IK_ALG_BASE_ELIM || ik->p.algorithm == WCQP_IK_ALG_DEFAULT;
    if (!(want4 && ik->hp.fast_ok && ik->p.jacobian_structure == WCQP_IK_JAC_MIXED)) return WCQP_E_UNSUPPORTED;
    int rc = ensure_device(ik);
    if (rc == WCQP_OK && mpc) rc = wcqp::mpc_prepare(mpc);
    if (rc != WCQP_OK) return rc;
    wcqp_qp_plan_s* p = new (std::nothrow) wcqp_qp_plan_s();
    if (!p) return WCQP_E_NOMEM;
    p->mpc = mpc; p->ik = ik; p->batch = batch; p->n_steps = n_steps; p->ways = ways < n_steps ? ways : n_steps; p->ik_only = ik_only;
    if (hipMalloc(reinterpret_cast<void**>(&p->d_recs), (size_t)n_steps * sizeof(wcqp_qp_step)) != hipSuccess) { delete p; return WCQP_E_NOMEM; }
    if (hipMemcpy(p->d_recs, steps, (size_t)n_steps * sizeof(wcqp_qp_step), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p->d_recs); delete p; return WCQP_E_HIP;
    }
    if (ways == 0) {
        p->queue_grid = wcqp_ik::ik4_plan_queue_grid(batch, n_steps);
        if (p->queue_grid < 1 || hipMalloc(reinterpret_cast<void**>(&p->d_queue), wcqp_ik::kPlanQueueBytes) != hipSuccess ||
            hipMemset(p->d_queue, 0, wcqp_ik::kPlanQueueBytes) != hipSuccess) {
            if (p->d_queue) (void)hipFree(p->d_queue);
            (void)hipFree(p->d_recs); delete p; return WCQP_E_HIP;
        }
        (void)hipStreamSynchronize(nullptr);          // (hipMemset does not wait; wcqp_qp_plan_enqueue may name a non-blocking stream)
    }
    *out = p;
    return WCQP_OK;
}

int wcqp_qp_plan_enqueue(wcqp_qp_plan_t p, void* stream) {
    if (!p) return WCQP_E_INVALID;
    if (p->mpc_only) return wcqp::mpc_launch_plan(p->mpc, p->batch, p->d_recs, p->n_steps, p->ways, (hipStream_t)stream);
    wcqp_mpc::MpcDeviceConsts c{};
    if (p->mpc) wcqp::mpc_device_consts(p->mpc, &c);        // (an IK-only plan never reads them)
    return wcqp_ik::ik4_launch_plan(p->ik->d_prm, p->batch, p->d_recs, p->n_steps, p->ways, c, (hipStream_t)stream, p->d_queue, p->queue_grid, p->ik_only);
}

int wcqp_qp_plan_destroy(wcqp_qp_plan_t p) {
    if (!p) return WCQP_E_INVALID;
    if (p->d_recs) (void)hipFree(p->d_recs);
    if (p->d_queue) (void)hipFree(p->d_queue);
    delete p;
    return WCQP_OK;
}

}  // extern "C"

extern "C" {

int wcqp_ik_solve_host(wcqp_ik_t h, int32_t batch,
                       const double* J_left, const double* J_right, const double* J_neck, const double* J_com,
                       const double* q, const double* state,
                       double* dq, int32_t* status, uint32_t* active_lower, uint32_t* active_upper,
                       double* foot_err, int32_t* iters) {
    if (!h || batch < 0) return WCQP_E_INVALID;
    if (!J_left || !J_right || !J_neck || !J_com || !q || !state || !dq || !status) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    const size_t B = (size_t)batch;
    const size_t o_jl = 0, o_jr = o_jl + B * 6 * kNV, o_jn = o_jr + B * 6 * kNV, o_jc = o_jn + B * 3 * kNV,
                 o_q = o_jc + B * 3 * kNV, o_st = o_q + B * kDof, o_dq = o_st + B * kStateLen,
                 o_fe = o_dq + B * kDof, n_dbl = o_fe + B * 12;
    rc = h->scratch.reserve(n_dbl * 8 + B * 4 * 4);
    if (rc != WCQP_OK) return rc;
    double* d = static_cast<double*>(h->scratch.ptr);
    int32_t* d_st = reinterpret_cast<int32_t*>(d + n_dbl);
    uint32_t* d_lo = reinterpret_cast<uint32_t*>(d_st + B);
    uint32_t* d_up = d_lo + B;
    int32_t* d_it = reinterpret_cast<int32_t*>(d_up + B);
    WCQP_HIP_TRY(hipMemcpy(d + o_jl, J_left, B * 6 * kNV * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_jr, J_right, B * 6 * kNV * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_jn, J_neck, B * 3 * kNV * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_jc, J_com, B * 3 * kNV * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_q, q, B * kDof * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_st, state, B * kStateLen * 8, hipMemcpyHostToDevice));
    rc = wcqp_ik_solve_device(h, batch, d + o_jl, d + o_jr, d + o_jn, d + o_jc, d + o_q, d + o_st,
                              d + o_dq, d_st, d_lo, d_up, foot_err ? d + o_fe : nullptr, d_it, nullptr);
    if (rc != WCQP_OK) return rc;
    WCQP_HIP_TRY(hipDeviceSynchronize());
    WCQP_HIP_TRY(hipMemcpy(dq, d + o_dq, B * kDof * 8, hipMemcpyDeviceToHost));
    WCQP_HIP_TRY(hipMemcpy(status, d_st, B * 4, hipMemcpyDeviceToHost));
    if (active_lower) WCQP_HIP_TRY(hipMemcpy(active_lower, d_lo, B * 4, hipMemcpyDeviceToHost));
    if (active_upper) WCQP_HIP_TRY(hipMemcpy(active_upper, d_up, B * 4, hipMemcpyDeviceToHost));
    if (foot_err) WCQP_HIP_TRY(hipMemcpy(foot_err, d + o_fe, B * 12 * 8, hipMemcpyDeviceToHost));
    if (iters) WCQP_HIP_TRY(hipMemcpy(iters, d_it, B * 4, hipMemcpyDeviceToHost));
    return WCQP_OK;
}

}  // extern "C"
