// Device-resident tick pipeline (BASELINE configs 4/5, SURVEY.md §8f-1 tick harness and
// §8f-2 MPC->IK glue).  See include/wcqp.h for the contract and oracle/tick_spec.py for the
// CPU restatement every number is checked against.
//
// Reference call order reproduced (citations relative to /root/reference/modules/Walking_module):
//   src/WalkingModule.cpp:578-597   StableDCMModel::integrateModel        -> tick_glue_kernel (consumer)
//   src/WalkingModule.cpp:604-636   MPC bracket                           -> mpc_condensed_kernel
//   src/WalkingModule.cpp:657-695   WalkingZMPController + desired CoM    -> tick_glue_kernel
//   src/WalkingModule.cpp:709-740   IK bracket                            -> ik_kernel
//   src/WalkingModule.cpp:741-744   velocity integration                  -> tick_post_kernel
//   src/WalkingModule.cpp:816       advanceReferenceSignals               -> tick counter in HBM
#include <cstring>
#include <new>
#include <vector>
#include "wcqp_internal.h"

#include "tick_device.h"
#include "ik_common.h"

namespace {

using namespace wcqp_tick;

__global__ void tick_glue_kernel(TickDev d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.batch) return;
    const int t = d.tick2[d.phase];
    const int code = d.sel[i];
    const int st = d.mpc_status[i];
    const bool ok = st == WCQP_STATUS_SOLVED || st == WCQP_STATUS_OUTSIDE_HULL;
    if (!ok) d.mpc_fail[i] += 1;
    double* s = d.state + (size_t)i * kStateLen;
    for (int ax = 0; ax < 2; ++ax) {
        double g_com;
        tick_glue_axis(d, i, t, ax, ok, d.u0[2 * i + ax], g_com, s[69 + ax], s[72 + ax]);
        if (!d.kin_mode) s[66 + ax] = g_com;
    }
    tick_glue_height(d, i, s);
    for (int k = 0; k < 6; ++k) tick_glue_twist(d, i, code, k, t, s[75 + k], s[81 + k]);
}

__global__ void tick_post_kernel(TickDev d) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.batch * kDof) return;
    const int i = g / kDof;
    const int t = d.tick2[d.phase];
    const bool ok = d.ik_status[i] == WCQP_STATUS_SOLVED;
    tick_post_joint(d, i, t, g % kDof, ok, d.dq[g]);
    if (g % kDof == 0) tick_post_instance(d, i, t, ok);
    if (g == 0) d.tick2[1 - d.phase] = t + 1;      // advanceReferenceSignals (WalkingModule.cpp:816)
}

// external feedback: the caller's measured state into the places the next tick reads its plant state from - the skewed
// chain's per-axis records (mst: com [2], dcm [6], measured ZMP [7]) - and the measured joints into q_meas (NULL: the desired ones)
__global__ void tick_feedback_kernel(TickDev d, const double* __restrict__ dcm, const double* __restrict__ com, const double* __restrict__ zmp,
                                     const double* __restrict__ q, double* __restrict__ q_meas) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.batch * kDof) return;
    const int i = g / kDof, jj = g % kDof;
    q_meas[g] = q ? q[g] : d.q_des[g];
    if (jj < 2) {
        double* r = d.mst + ((size_t)i * 2 + jj) * 8;
        r[2] = com[2 * i + jj]; r[6] = dcm[2 * i + jj]; r[7] = zmp[2 * i + jj];
    }
}

}  // namespace

struct wcqp_tick_s {
    wcqp_tick_params p{};
    wcqp_mpc_t mpc = nullptr;
    wcqp_ik_t ik = nullptr;
    TickDev d{};
    TickDev* d_dev = nullptr;     // copy of `d` in device memory (the fused kernel reads it from there, see ik4.hip)
    std::vector<void*> allocs;
    double *J_left = nullptr, *J_right = nullptr, *J_neck = nullptr, *J_com = nullptr;
    unsigned* mpc_active = nullptr; double* mpc_margin = nullptr;
    unsigned *ik_lo = nullptr, *ik_up = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t graph_stream = nullptr;
    bool uploaded = false;
    int ticks_enqueued = 0;  // since the last upload; its parity is the `phase` of the next tick
    double* log_ferr = nullptr;   // logger rows with dense Jacobians: where the IK kernel forms the foot errors
    bool fused = false;      // glue + post inside the 16-lane IK kernel: 2 launches per tick instead of 4
    bool base_elim = false;  // the fused kernel is the base-eliminated one (ik4.hip)
    wcqp_kin_t kin = nullptr;     // use_kinematics: Jacobians, actual poses and hull rows are rebuilt every tick
    KinTick kt{};
    int phase = 0;                // which copy of the tick index the next launch reads (TickDev::tick2): toggles per LAUNCH
    int ticks_per_launch = 1;     // > 1: the fused kernel walks through that many ticks per launch (no per-tick kinematics)
    // wcqp_tick_splice_reference: the caller's host rows are staged HERE at call time (a copy stream of the handle's own, waited
    // for before the call returns), the strided device-to-device copy then runs in the caller's stream order
    double* splice_stage = nullptr; size_t splice_cap = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t splice_done = nullptr; bool splice_pending = false;
    bool external = false, feedback_set = false;     // wcqp_tick_params.plant = EXTERNAL: one tick per run call, each behind a set_feedback
    double* q_meas = nullptr;
    double* fb_stage = nullptr;   // wcqp_tick_set_feedback_host: [B][2 + 2 + 2 + dof]
};

namespace {

template <typename T>
int dev_alloc(wcqp_tick_s* h, T** out, size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, (count > 0 ? count : 1) * sizeof(T)) != hipSuccess) return WCQP_E_NOMEM;
    h->allocs.push_back(p);              // owned from here on: wcqp_tick_destroy frees it whatever happens next
    if (hipMemset(p, 0, (count > 0 ? count : 1) * sizeof(T)) != hipSuccess) return WCQP_E_HIP;
    *out = static_cast<T*>(p);
    return WCQP_OK;
}

template <typename T>
int dev_alloc(wcqp_tick_s* h, wcqp::GPtr<T>* out, size_t count) { return dev_alloc(h, &out->p, count); }

// one launch sequence of n_inner ticks (n_inner > 1: the fused base-eliminated kernel without per-tick kinematics only) with
// the given phase (which copy of the tick index it reads: see TickDev::tick2)
int enqueue_tick(wcqp_tick_s* h, int phase, hipStream_t s, int n_inner = 1, int skip_last_mpc = 0) {
    TickDev d = h->d;
    d.phase = phase & 1;
    const int B = d.batch;
    const int N = wcqp::mpc_horizon(h->mpc);
    if (n_inner > 1 && !(h->fused && h->base_elim && (!h->kin || d.kin_fused))) return WCQP_E_INVALID;
    if (h->kin && !d.kin_fused) {
        h->kt.phase = d.phase;
        const int rck = wcqp::kin_enqueue_tick(h->kin, B, h->kt, d.q_des, h->J_left, h->J_right, h->J_neck, h->J_com, d.state, s);
        if (rck != WCQP_OK) return rck;
    }
    // base-eliminated IK kernel: IK + post step of this tick and MPC + glue + plant of the NEXT one in ONE launch (skewed tick)
    if (h->fused && h->base_elim)
        return wcqp_ik::ik4_launch_tick(wcqp::ik_device_params(h->ik), d, h->d_dev, h->J_left, h->J_right, h->J_neck, h->J_com,
                                        h->ik_lo, h->ik_up, n_inner, skip_last_mpc, s, h->log_ferr);
    int rc = wcqp::mpc_enqueue(h->mpc, B, d.dcm, d.ref_traj, N + 1, d.traj_len, d.tick2 + d.phase, d.u_prev,
                               d.hull_tab_A, d.hull_tab_b, d.hull_tab_nc, d.hull_sets, d.hull_sets > 1 ? d.sel : nullptr,
                               d.u0, d.mpc_status, h->mpc_active, h->mpc_margin, s);
    if (rc != WCQP_OK) return rc;
    if (h->fused)
        return wcqp_ik::ik3_launch_tick(wcqp::ik_device_params(h->ik), d, h->J_left, h->J_right, h->J_neck, h->J_com,
                                        h->ik_lo, h->ik_up, s);
    hipLaunchKernelGGL(tick_glue_kernel, dim3((B + 127) / 128), dim3(128), 0, s, d);
    rc = wcqp_ik_solve_device(h->ik, B, h->J_left, h->J_right, h->J_neck, h->J_com, d.q_des, d.state,
                              d.dq, d.ik_status, h->ik_lo, h->ik_up, nullptr, nullptr, s);
    if (rc != WCQP_OK) return rc;
    hipLaunchKernelGGL(tick_post_kernel, dim3((B * kDof + 255) / 256), dim3(256), 0, s, d);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

}  // namespace

extern "C" {

int wcqp_tick_create(const wcqp_tick_params* params, wcqp_tick_t* out) {
    if (!params || !out || params->batch < 1 || params->max_ticks < 1) return WCQP_E_INVALID;
    if (params->step_ticks < 2 || params->ds_ticks < 0 || params->ds_ticks > params->step_ticks) return WCQP_E_INVALID;
    if (params->ik.dof != kDof) return WCQP_E_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        std::fprintf(stderr, "[wcqp] no HIP device: the tick pipeline has no CPU fallback\n");
        return WCQP_E_HIP;
    }
    wcqp_tick_s* h = new (std::nothrow) wcqp_tick_s();
    if (!h) return WCQP_E_NOMEM;
    h->p = *params;
    // the fused form exists for the 16-lane kernel only (CoM as constraint); an explicit 32-lane / sweep
    // algorithm keeps the four-launch form, which is also what the fused one is tested against
    h->fused = params->ik.use_com_as_constraint &&
               (params->ik.algorithm == WCQP_IK_ALG_DEFAULT || params->ik.algorithm == WCQP_IK_ALG_NULLSPACE_16L ||
                params->ik.algorithm == WCQP_IK_ALG_BASE_ELIM);
    int rc = wcqp_mpc_create(&params->mpc, &h->mpc);
    if (rc == WCQP_OK) rc = wcqp_ik_create(&params->ik, &h->ik);
    if (rc == WCQP_OK && params->use_kinematics) {
        if (params->kin.dof != kDof) rc = WCQP_E_UNSUPPORTED;
        if (rc == WCQP_OK) rc = wcqp_kin_create(&params->kin, &h->kin);
        if (rc == WCQP_OK) rc = wcqp::kin_prepare(h->kin);
    }
    if (rc == WCQP_OK) rc = wcqp::mpc_prepare(h->mpc);
    if (rc == WCQP_OK) rc = wcqp::ik_prepare(h->ik);
    if (rc != WCQP_OK) { wcqp_tick_destroy(h); return rc; }
    // the tick's Jacobians are MIXED free-floating ones (uploaded or from wcqp_kin_*): an instance that is not comes
    // back WCQP_STATUS_STRUCTURE and counts as an IK failure
    h->d.hot_start = params->ik_cold_start_only ? 0 : 1;
    if (params->ticks_per_launch < 0 || params->logger_ticks < 0) { wcqp_tick_destroy(h); return WCQP_E_INVALID; }
    if (params->plant != WCQP_TICK_PLANT_INTERNAL && params->plant != WCQP_TICK_PLANT_EXTERNAL) { wcqp_tick_destroy(h); return WCQP_E_INVALID; }
    h->external = params->plant == WCQP_TICK_PLANT_EXTERNAL;
    h->base_elim = h->fused && params->ik.algorithm != WCQP_IK_ALG_NULLSPACE_16L &&
                   params->ik.jacobian_structure != WCQP_IK_JAC_GENERAL && wcqp::ik_fast_ok(h->ik);
    const size_t B = (size_t)params->batch;
    const int N = params->mpc.horizon;
    TickDev& d = h->d;
    d.batch = params->batch; d.first = params->first; d.traj_len = params->max_ticks + N + 1;
    d.log_ticks = params->log_ticks > 0 ? params->log_ticks : 0;
    d.step_ticks = params->step_ticks; d.ds_ticks = params->ds_ticks;
    d.inv_ss = d.step_ticks - d.ds_ticks > 0 ? 1.0 / (double)(d.step_ticks - d.ds_ticks) : 0.0;
    d.dT = params->mpc.sampling_time; d.k_com = params->k_com; d.k_zmp = params->k_zmp;
    d.noise = params->noise; d.seed = params->seed; d.com_height = params->mpc.com_height;
    d.omega = std::sqrt(params->mpc.gravity / params->mpc.com_height);
    wcqp::mpc_dynamics(h->mpc, &d.a, &d.b);
    double *ref = nullptr, *hA = nullptr, *hb = nullptr, *sw = nullptr;
    int *hn = nullptr, *ph = nullptr;
    rc = WCQP_OK;
#define A_(ptr, n) if (rc == WCQP_OK) rc = dev_alloc(h, &(ptr), (n))
    const size_t hsets = 3;       // rows for {left, right, both} in contact: uploaded, or (kinematics mode) built at upload from the desired foot poses
    A_(ref, B * d.traj_len * 2); A_(hA, B * hsets * 16); A_(hb, B * hsets * 8); A_(hn, B * hsets); A_(ph, B); A_(sw, B * 6);
    A_(d.dcm, B * 2); A_(d.com, B * 2); A_(d.zmp_meas, B * 2); A_(d.u_prev, B * 2); A_(d.u0, B * 2);
    A_(d.c_ref, B * 2); A_(d.v_ref, B * 2); A_(d.v_ref_prev, B * 2); A_(d.p_star, B * 2); A_(d.v_star_prev, B * 2);
    A_(d.q_des, B * kDof); A_(d.dq_prev, B * kDof); A_(d.dq, B * kDof);
    A_(d.sel, B);
    A_(d.state, B * kStateLen); A_(d.mpc_status, B); A_(d.ik_status, B); A_(d.mpc_fail, B); A_(d.ik_fail, B);
    A_(d.hot_try, B); A_(d.hot_hit, B);
    A_(d.tick2, 2); A_(d.u0_log, (size_t)d.log_ticks * B * 2); A_(d.dq_log, (size_t)d.log_ticks * B * kDof);
    A_(h->J_left, B * 6 * 29); A_(h->J_right, B * 6 * 29); A_(h->J_neck, B * 3 * 29); A_(h->J_com, B * 3 * 29);
    A_(h->mpc_active, B); A_(h->mpc_margin, B); A_(h->ik_lo, B); A_(h->ik_up, B);
    // skewed tick (base-eliminated fused kernel): state of the MPC chain, MPC -> IK hand-off, one live hull row set per robot
    d.skew = (h->fused && h->base_elim) ? 1 : 0;

    double* jcomp = nullptr;
    unsigned cm[3] = {0u, 0u, 0u};
    int cstride = 0, coff_d = 0;
    if (params->kin_handoff < 0 || params->kin_handoff > 2) { wcqp_tick_destroy(h); return WCQP_E_INVALID; }
    const bool masks_ok = d.skew && h->kin && wcqp::kin_compact_layout(h->kin, cm, &cstride, &coff_d);
    // kinematics fused into the solve kernel (default), or a kinematics launch per tick handing over compact records / dense Jacobians
    std::vector<double> ktab;
    const bool fusedk = masks_ok && params->kin_handoff == WCQP_KIN_HANDOFF_FUSED && N < kGainsLdsStages &&
                        wcqp::kin_fused_tables(h->kin, ktab, &d.kin_rounds);
    const bool compact = masks_ok && !fusedk && params->kin_handoff != WCQP_KIN_HANDOFF_DENSE;
    // external feedback: the default (base-eliminated) kernel with constant Jacobians or fused kinematics, without logger rows
    if (h->external && (!d.skew || params->logger_ticks > 0 || (h->kin && !fusedk))) { wcqp_tick_destroy(h); return WCQP_E_UNSUPPORTED; }
    if (h->external) { A_(h->q_meas, B * kDof); d.q_meas = h->q_meas; A_(h->fb_stage, B * (6 + kDof)); }
    if (d.skew) {
        A_(d.mst, B * 16); A_(d.hand, 2 * B * kHandLen); A_(d.live_A, B * 16); A_(d.live_b, B * 8); A_(d.live_nc, B); A_(d.sel_built, B);
        if (compact) A_(jcomp, B * (size_t)cstride);
        if (params->logger_ticks > 0) { A_(d.log_rows, (size_t)params->logger_ticks * B * kLoggerCols); A_(h->log_ferr, B * 12); d.logger_ticks = params->logger_ticks; }
        if (fusedk) {
            double* kt = nullptr;
            A_(kt, ktab.size());
            if (rc == WCQP_OK && hipMemcpy(kt, ktab.data(), ktab.size() * 8, hipMemcpyHostToDevice) != hipSuccess) rc = WCQP_E_HIP;
            d.kin_tab = kt;
        }
    }
#undef A_
    if (rc != WCQP_OK) { wcqp_tick_destroy(h); return rc; }
    wcqp::mpc_device_consts(h->mpc, &d.mpc);
    d.horizon = N; d.hull_sets = (int)hsets;
    if (compact) { d.compact = 1; d.jcomp = jcomp; d.cmaskL = cm[0]; d.cmaskR = cm[1]; d.cmaskN = cm[2]; d.cstride = cstride; d.coff_d = coff_d; }
    if (fusedk) { d.kin_fused = 1; d.cmaskL = cm[0]; d.cmaskR = cm[1]; d.cmaskN = cm[2]; }
    // several ticks per launch: whenever a tick is ONE launch of the fused kernel (no kinematics launch in between)
    h->ticks_per_launch = (d.skew && (!h->kin || fusedk)) ? (params->ticks_per_launch > 0 ? params->ticks_per_launch : (1 << 20)) : 1;
    if (h->external) h->ticks_per_launch = 1;            // a tick cannot run ahead of its feedback
    if (h->kin) {
        double* h0 = nullptr;
        if (dev_alloc(h, &h0, B) != WCQP_OK) { wcqp_tick_destroy(h); return WCQP_E_NOMEM; }
        d.kin_mode = 1; d.com_h0 = h0;
        h->kt.tick2 = d.tick2; h->kt.phase0 = ph; h->kt.step_ticks = d.step_ticks;
        if (compact) { h->kt.jcomp = jcomp; h->kt.cstride = cstride; h->kt.coff_d = coff_d; }
    }
    d.ref_traj = ref; d.hull_tab_A = hA; d.hull_tab_b = hb; d.hull_tab_nc = hn; d.phase0 = ph; d.swing_twist = sw;
#ifdef WCQP_TICK_STAMPS
    if (d.skew && dev_alloc(h, &d.stamps, ((B + 3) / 4) * 16) != WCQP_OK) { wcqp_tick_destroy(h); return WCQP_E_NOMEM; }
#endif
    if (d.skew) {
        if (dev_alloc(h, &h->d_dev, 1) != WCQP_OK) { wcqp_tick_destroy(h); return WCQP_E_NOMEM; }
        if (hipMemcpy(h->d_dev, &d, sizeof(TickDev), hipMemcpyHostToDevice) != hipSuccess) { wcqp_tick_destroy(h); return WCQP_E_HIP; }
    }
    *out = h;
    return WCQP_OK;
}

int wcqp_tick_destroy(wcqp_tick_t h) {
    if (!h) return WCQP_E_INVALID;
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->splice_stage) (void)hipFree(h->splice_stage);
    if (h->splice_done) (void)hipEventDestroy(h->splice_done);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->kin) wcqp_kin_destroy(h->kin);
    if (h->mpc) wcqp_mpc_destroy(h->mpc);
    if (h->ik) wcqp_ik_destroy(h->ik);
    delete h;
    return WCQP_OK;
}

int wcqp_tick_upload(wcqp_tick_t h, const wcqp_tick_inputs* in) {
    if (!h || !in) return WCQP_E_INVALID;
    if (!in->ref_traj || !in->phase0 || !in->state0 || !in->swing_twist || !in->q0 || !in->dcm0 || !in->com0 || !in->u_init) return WCQP_E_INVALID;
    if (!h->kin && (!in->hull_tab_A || !in->hull_tab_b || !in->hull_tab_nc || !in->J_left || !in->J_right || !in->J_neck || !in->J_com))
        return WCQP_E_INVALID;
    TickDev& d = h->d;
    const size_t B = (size_t)d.batch;
    WCQP_HIP_TRY(hipDeviceSynchronize());
#define UP_(dst, src, n) WCQP_HIP_TRY(hipMemcpy(const_cast<void*>(static_cast<const void*>(dst)), (src), (n), hipMemcpyHostToDevice))
    UP_(d.ref_traj, in->ref_traj, B * d.traj_len * 16);
    UP_(d.phase0, in->phase0, B * 4); UP_(d.swing_twist, in->swing_twist, B * 48);
    if (d.skew) {
        // state of the MPC chain per axis: c_ref, v_ref_prev, com, u_prev (= measured ZMP), p_star, v_star_prev, dcm, spare
        std::vector<double> mst(B * 16, 0.0);
        for (size_t i = 0; i < B; ++i)
            for (int ax = 0; ax < 2; ++ax) {
                double* r = &mst[(i * 2 + ax) * 8];
                r[0] = in->com0[2 * i + ax]; r[2] = in->com0[2 * i + ax]; r[3] = in->u_init[2 * i + ax];
                r[4] = in->com0[2 * i + ax]; r[6] = in->dcm0[2 * i + ax]; r[7] = in->u_init[2 * i + ax];
            }
        WCQP_HIP_TRY(hipMemcpy(d.mst, mst.data(), B * 16 * 8, hipMemcpyHostToDevice));
        WCQP_HIP_TRY(hipMemset(d.sel_built, 0xff, B * 4));           // -1: every robot builds / copies its live rows at tick 0
        WCQP_HIP_TRY(hipMemset(d.live_nc, 0, B * 4));
    }
    if (h->kin) {
        std::vector<double> h0(B);
        for (size_t i = 0; i < B; ++i) h0[i] = in->state0[i * kStateLen + 68];       // desired CoM height = the initial one
        WCQP_HIP_TRY(hipMemcpy(const_cast<double*>(d.com_h0.get()), h0.data(), B * 8, hipMemcpyHostToDevice));
    } else {
        UP_(d.hull_tab_A, in->hull_tab_A, B * 3 * 128); UP_(d.hull_tab_b, in->hull_tab_b, B * 3 * 64); UP_(d.hull_tab_nc, in->hull_tab_nc, B * 3 * 4);
        UP_(h->J_left, in->J_left, B * 6 * 29 * 8); UP_(h->J_right, in->J_right, B * 6 * 29 * 8);
        UP_(h->J_neck, in->J_neck, B * 3 * 29 * 8); UP_(h->J_com, in->J_com, B * 3 * 29 * 8);
    }
    UP_(d.state, in->state0, B * kStateLen * 8); UP_(d.q_des, in->q0, B * kDof * 8);
    if (h->kin) {
        // setConvexHullConstraint (...PredictiveController.cpp:364-435) for the three contact pairs, from the DESIRED foot
        // poses just uploaded (the planned footsteps, WalkingModule.cpp:609-613): the MPC of a tick selects its rows by the pair
        const int rch = wcqp::hull_tables_from_state((int)B, h->p.foot_rect, d.state, kStateLen, const_cast<double*>(d.hull_tab_A.get()),
                                                     const_cast<double*>(d.hull_tab_b.get()), const_cast<int*>(d.hull_tab_nc.get()), nullptr);
        if (rch != WCQP_OK) return rch;
        WCQP_HIP_TRY(hipDeviceSynchronize());
    }
    UP_(d.dcm, in->dcm0, B * 16); UP_(d.com, in->com0, B * 16); UP_(d.c_ref, in->com0, B * 16); UP_(d.p_star, in->com0, B * 16);
    UP_(d.zmp_meas, in->u_init, B * 16); UP_(d.u_prev, in->u_init, B * 16);
#undef UP_
    WCQP_HIP_TRY(hipMemset(d.v_ref_prev, 0, B * 16)); WCQP_HIP_TRY(hipMemset(d.v_star_prev, 0, B * 16));
    WCQP_HIP_TRY(hipMemset(d.dq_prev, 0, B * kDof * 8)); WCQP_HIP_TRY(hipMemset(d.tick2, 0, 8));
    {   // contact pair of tick 0 (later ticks: tick_post_kernel)
        std::vector<int> sel(B);
        for (size_t i = 0; i < B; ++i) {
            const int cyc = in->phase0[i] % (2 * d.step_ticks), sidx = cyc % d.step_ticks;
            sel[i] = sidx < d.ds_ticks ? 2 : cyc / d.step_ticks;
        }
        WCQP_HIP_TRY(hipMemcpy(d.sel, sel.data(), B * 4, hipMemcpyHostToDevice));
    }
    WCQP_HIP_TRY(hipMemset(d.mpc_fail, 0, B * 8)); WCQP_HIP_TRY(hipMemset(d.ik_fail, 0, B * 8));
    WCQP_HIP_TRY(hipMemset(d.hot_try, 0, B * 8)); WCQP_HIP_TRY(hipMemset(d.hot_hit, 0, B * 8));
    WCQP_HIP_TRY(hipMemset(h->ik_lo, 0, B * 4)); WCQP_HIP_TRY(hipMemset(h->ik_up, 0, B * 4));      // no previous active set at tick 0
    // (hipMemset does not wait, and the copies / kernels above ran on the NULL stream: the run call that follows may name a non-blocking
    // stream - wcqp_stream_create makes such - which would not wait for them either)
    WCQP_HIP_TRY(hipDeviceSynchronize());
    h->uploaded = true;
    h->ticks_enqueued = 0;
    h->phase = 0;
    h->feedback_set = false;
    return WCQP_OK;
}

int wcqp_tick_set_feedback_device(wcqp_tick_t h, const double* dcm_meas, const double* com_meas, const double* zmp_meas, const double* q_meas, void* stream) {
    if (!h || !dcm_meas || !com_meas || !zmp_meas) return WCQP_E_INVALID;
    if (!h->external) return WCQP_E_UNSUPPORTED;
    if (!h->uploaded) return WCQP_E_INVALID;
    const int n = h->d.batch * kDof;
    hipLaunchKernelGGL(tick_feedback_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->d, dcm_meas, com_meas, zmp_meas, q_meas, h->q_meas);
    WCQP_HIP_TRY(hipGetLastError());
    h->feedback_set = true;
    return WCQP_OK;
}

int wcqp_tick_set_feedback_host(wcqp_tick_t h, const double* dcm_meas, const double* com_meas, const double* zmp_meas, const double* q_meas) {
    if (!h || !dcm_meas || !com_meas || !zmp_meas) return WCQP_E_INVALID;
    if (!h->external) return WCQP_E_UNSUPPORTED;
    const size_t B = (size_t)h->d.batch;
    double* st = h->fb_stage;
    // (synchronous copies on the NULL stream: they wait for what the handle's last tick left running on a blocking stream, and the
    // host arrays are consumed when they return)
    WCQP_HIP_TRY(hipMemcpy(st, dcm_meas, B * 16, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(st + 2 * B, com_meas, B * 16, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(st + 4 * B, zmp_meas, B * 16, hipMemcpyHostToDevice));
    if (q_meas) WCQP_HIP_TRY(hipMemcpy(st + 6 * B, q_meas, B * kDof * 8, hipMemcpyHostToDevice));
    const int rc = wcqp_tick_set_feedback_device(h, st, st + 2 * B, st + 4 * B, q_meas ? st + 6 * B : nullptr, nullptr);
    if (rc != WCQP_OK) return rc;
    // the copy kernel ran on the NULL stream; the tick that consumes the feedback may be enqueued on ANY stream - a non-blocking one
    // (wcqp_stream_create) would not wait for it - so the feedback is in place when this call returns, and the staging rows are free
    WCQP_HIP_TRY(hipStreamSynchronize(nullptr));
    return WCQP_OK;
}

int wcqp_tick_run(wcqp_tick_t h, int32_t n_ticks, int32_t use_graph, void* stream) {
    if (!h || n_ticks < 0 || !h->uploaded) return WCQP_E_INVALID;
    // the trajectories hold max_ticks + N + 1 stages per instance: a tick beyond that would read its neighbour's
    if ((long)h->ticks_enqueued + n_ticks > (long)h->p.max_ticks) return WCQP_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n_ticks == 0) return WCQP_OK;
    if (h->external && (n_ticks != 1 || !h->feedback_set)) return WCQP_E_INVALID;      // one tick per call, each behind its own feedback
    int left = n_ticks;
    // Everything that can be refused on the host is refused BEFORE anything is enqueued (the prime launch below already advances
    // the MPC chain); an enqueue that fails after that leaves device state nobody can name - the handle then wants a new upload.
    struct NeedsUpload { wcqp_tick_s* h; bool armed = true; ~NeedsUpload() { if (armed) h->uploaded = false; } } guard{h};
    if (h->ticks_per_launch > 1 && !(h->fused && h->base_elim && (!h->kin || h->d.kin_fused))) { guard.armed = false; return WCQP_E_INVALID; }
    if (h->d.skew && (!h->d_dev || !h->d.mst || !h->d.hand)) { guard.armed = false; return WCQP_E_INVALID; }
    if (h->d.skew) {
        // the fused launch of tick t carries IK(t) and MPC(t+1): the MPC of the call's first tick goes first, on its own, and
        // the call's LAST tick does not run the MPC of the tick after it - between calls nothing is ahead of anything
        const int rc = wcqp_ik::ik4_launch_tick_prime(h->d, h->ticks_enqueued, s);
        if (rc != WCQP_OK) return rc;
    }
    // the fused kernel walks through several ticks per launch (the waves need no per-tick synchronisation): no graph needed
    if (h->ticks_per_launch > 1) {
        while (left > 0) {
            const int k = left < h->ticks_per_launch ? left : h->ticks_per_launch;
            const int rc = enqueue_tick(h, h->phase, s, k, k == left ? 1 : 0);
            if (rc != WCQP_OK) return rc;
            h->phase ^= 1; h->ticks_enqueued += k; left -= k;
        }
        guard.armed = false;
        return WCQP_OK;
    }
    if (h->d.skew) left -= 1;      // the last tick of the call is a plain launch of its own (below)
    // kGraphTicks ticks per graph (the tick index lives in HBM, so the graph is tick-invariant): one
    // hipGraphLaunch costs about as much as four plain launches.  The graph is captured with phases 0, 1, 0, ...
    // and therefore replayed only from phase 0; from phase 1 a plain tick goes first.
    constexpr int kGraphTicks = 8;
    auto plain = [&]() -> int {
        const int rc = enqueue_tick(h, h->phase, s);
        if (rc == WCQP_OK) { h->phase ^= 1; ++h->ticks_enqueued; --left; }
        return rc;
    };
    if (use_graph && left >= kGraphTicks + 1 && h->phase) { const int rc = plain(); if (rc != WCQP_OK) return rc; }
    if (use_graph && !h->graph_exec && left >= kGraphTicks) {
        hipStream_t cs = nullptr;
        WCQP_HIP_TRY(hipStreamCreate(&cs));
        // (lazy device state of the solver handles exists since create: capture forbids allocations)
        WCQP_HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int rc = WCQP_OK;
        for (int k = 0; k < kGraphTicks && rc == WCQP_OK; ++k) rc = enqueue_tick(h, k & 1, cs);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(cs, &g);
        (void)hipStreamDestroy(cs);
        if (rc != WCQP_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess || !g) return WCQP_E_HIP;
        h->graph = g;
        WCQP_HIP_TRY(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
    }
    if (use_graph && h->graph_exec && !h->phase) {
        for (; left >= kGraphTicks; left -= kGraphTicks, h->ticks_enqueued += kGraphTicks) WCQP_HIP_TRY(hipGraphLaunch(h->graph_exec, s));
    }
    while (left > 0) { const int rc = plain(); if (rc != WCQP_OK) return rc; }
    if (h->d.skew) {
        const int rc = enqueue_tick(h, h->phase, s, 1, 1);
        if (rc != WCQP_OK) return rc;
        h->phase ^= 1; ++h->ticks_enqueued;
    }
    guard.armed = false;
    h->feedback_set = false;
    return WCQP_OK;
}

int wcqp_tick_splice_reference(wcqp_tick_t h, int32_t from_tick, int32_t n_stages, const double* ref_tail, void* stream) {
    if (!h || !h->uploaded || !ref_tail || n_stages < 1) return WCQP_E_INVALID;
    const TickDev& d = h->d;
    // stages the ticks already enqueued have consumed as their own reference DCM stay as they are; everything a later
    // tick's window can see may change
    if (from_tick < h->ticks_enqueued || (long)from_tick + n_stages > (long)d.traj_len) return WCQP_E_INVALID;
    // `ref_tail` is the caller's HOST memory and the copy below is ordered behind ticks that may still run for a long time: the
    // rows are therefore taken NOW - staged into device memory of the handle on a copy stream of its own, waited for before
    // this call returns - and the caller may release `ref_tail` as soon as it has.
    const size_t bytes = (size_t)d.batch * (size_t)n_stages * 16;
    if (!h->copy_stream) WCQP_HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!h->splice_done) WCQP_HIP_TRY(hipEventCreateWithFlags(&h->splice_done, hipEventDisableTiming));
    if (h->splice_pending) { WCQP_HIP_TRY(hipEventSynchronize(h->splice_done)); h->splice_pending = false; }   // the previous merge has left the staging rows
    if (bytes > h->splice_cap) {
        if (h->splice_stage) { (void)hipFree(h->splice_stage); h->splice_stage = nullptr; h->splice_cap = 0; }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return WCQP_E_NOMEM;
        h->splice_stage = static_cast<double*>(p); h->splice_cap = bytes;
    }
    WCQP_HIP_TRY(hipMemcpyAsync(h->splice_stage, ref_tail, bytes, hipMemcpyHostToDevice, h->copy_stream));
    WCQP_HIP_TRY(hipStreamSynchronize(h->copy_stream));
    // strided copy: row i of the tail goes to stages [from_tick, from_tick + n_stages) of instance i, in stream order
    // behind the ticks already enqueued (the trajectory pointer the kernels - and any captured graph - hold does not change)
    WCQP_HIP_TRY(hipMemcpy2DAsync(const_cast<double*>(d.ref_traj.get()) + (size_t)from_tick * 2, (size_t)d.traj_len * 16, h->splice_stage, (size_t)n_stages * 16,
                                  (size_t)n_stages * 16, (size_t)d.batch, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    WCQP_HIP_TRY(hipEventRecord(h->splice_done, (hipStream_t)stream));
    h->splice_pending = true;
    return WCQP_OK;
}

#ifdef WCQP_TICK_STAMPS
// diagnostic builds only: the phase stamps of every workgroup's last tick ([workgroups][16])
int wcqp_tick_debug_stamps(wcqp_tick_t h, unsigned long long* out, int32_t n) {
    if (!h || !out || !h->d.stamps || n > ((h->d.batch + 3) / 4) * 16) return WCQP_E_INVALID;
    WCQP_HIP_TRY(hipDeviceSynchronize());
    WCQP_HIP_TRY(hipMemcpy(out, h->d.stamps, (size_t)n * 8, hipMemcpyDeviceToHost));
    return WCQP_OK;
}
#endif

int wcqp_tick_download(wcqp_tick_t h, const wcqp_tick_outputs* out) {
    if (!h || !out) return WCQP_E_INVALID;
    const TickDev& d = h->d;
    const size_t B = (size_t)d.batch;
    WCQP_HIP_TRY(hipDeviceSynchronize());
#define DN_(dst, src, n) if (dst) WCQP_HIP_TRY(hipMemcpy((dst), (src), (n), hipMemcpyDeviceToHost))
    DN_(out->u0_log, d.u0_log, (size_t)d.log_ticks * B * 16); DN_(out->dq_log, d.dq_log, (size_t)d.log_ticks * B * kDof * 8);
    DN_(out->q_des, d.q_des, B * kDof * 8);
    DN_(out->active_lower, h->ik_lo, B * 4); DN_(out->active_upper, h->ik_up, B * 4);
    if (out->logger) {
        if (!d.log_rows) return WCQP_E_UNSUPPORTED;          // logger_ticks = 0, or an IK algorithm without the fused tick kernel
        WCQP_HIP_TRY(hipMemcpy(out->logger, d.log_rows, (size_t)d.logger_ticks * B * kLoggerCols * 8, hipMemcpyDeviceToHost));
    }
    if (d.skew) {
        // the state of the MPC chain lives in per-axis records (TickDev::mst): com at [2], dcm at [6]
        if (out->dcm || out->com) {
            std::vector<double> mst(B * 16);
            WCQP_HIP_TRY(hipMemcpy(mst.data(), d.mst, B * 16 * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < B; ++i)
                for (int ax = 0; ax < 2; ++ax) {
                    if (out->com) out->com[2 * i + ax] = mst[(i * 2 + ax) * 8 + 2];
                    if (out->dcm) out->dcm[2 * i + ax] = mst[(i * 2 + ax) * 8 + 6];
                }
        }
    } else {
        DN_(out->dcm, d.dcm, B * 16); DN_(out->com, d.com, B * 16);
    }
    DN_(out->mpc_fail, d.mpc_fail, B * 8); DN_(out->ik_fail, d.ik_fail, B * 8);
    DN_(out->hot_try, d.hot_try, B * 8); DN_(out->hot_hit, d.hot_hit, B * 8); DN_(out->tick, d.tick2 + h->phase, 4);
#undef DN_
    return WCQP_OK;
}

}  // extern "C"
