// DCM-MPC: batch-constant condensing on the host + one HBM-bound HIP kernel.
//
// Reference path replaced (citations relative to /root/reference/modules/Walking_module):
//   constants   WalkingController::initializeMatrices   src/WalkingDCMModelPredictiveController.cpp:170-243
//   per tick    MPCSolver::{setConstraintsMatrix,setBounds,setGradient,solve,getSolution}
//                                                       src/MPCSolver.cpp:76-322
//               WalkingController::solve (u0 read-out + hull-margin check)
//                                                       src/WalkingDCMModelPredictiveController.cpp:491-521
//
// Why this is not an ADMM loop: P, A_eq are the same for every robot and every tick, and
// the only per-instance rows of A (the support-polygon rows) touch u0 alone
// (MPCSolver.cpp:82-86).  Eliminating everything but u0 through the constant equality
// KKT K = [P A_eq'; A_eq 0] leaves
//      u0_unc = sum_i Gr_i r_i + Gx x0 + Gu u_prev          (rows of K^-1, built once)
//      u0     = argmin (u-u0_unc)' Sigma0^-1 (u-u0_unc)  s.t.  A_h u <= b_h
// i.e. ~1 KB of HBM traffic and ~1 kflop per QP; the optimum is reached exactly (it is the
// cheapest feasible point among {no row, one row, two rows active}), not to eps = 1e-3.
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include "wcqp_internal.h"
#include "mpc_device.h"

namespace {

using namespace wcqp_mpc;
constexpr int kBlock = 64;              // one wavefront per workgroup

__global__ __launch_bounds__(kBlock)
void mpc_condensed_kernel(MpcDeviceConsts c, int batch,
                          const double* __restrict__ x0, const double* __restrict__ ref, int ref_len,
                          int ref_stride, const int* __restrict__ ref_start,
                          const double* __restrict__ u_prev,
                          const double* __restrict__ hull_A, const double* __restrict__ hull_b,
                          const int* __restrict__ hull_nc, int hull_sets, const int* __restrict__ hull_sel,
                          double* __restrict__ u0_out, int* __restrict__ status_out,
                          unsigned* __restrict__ active_out, double* __restrict__ margin_out)
{
    __shared__ __attribute__((aligned(16))) double s_hull[kInstPerWave][WCQP_HULL_ROWS][4];  // ax, ay, b, |a|

    const int lane = threadIdx.x;
    const int sub  = lane / kLanesPerInstance;          // instance slot inside the wave
    const int t    = lane % kLanesPerInstance;
    const long inst_raw = (long)blockIdx.x * kInstPerWave + sub;
    const bool live = inst_raw < batch;
    const long inst = live ? inst_raw : (long)batch - 1;   // dead slots shadow the last instance, never store

    // the reference window of an instance starts `*ref_start` stages into its trajectory (the
    // deque of the reference advanced by that many ticks); instances are `ref_stride` stages apart
    const int start = ref_start ? *ref_start : 0;
    const double2* rp = reinterpret_cast<const double2*>(ref) + inst * ref_stride + start;
    // an instance may carry several precomputed row sets (one per contact pair); `hull_sel` picks
    // the live one, so a contact change costs no copy (tick pipeline)
    const long hset = inst * hull_sets + (hull_sel ? hull_sel[inst] : 0);
    double ux, uy, margin;
    int st;
    unsigned mask;
    mpc_row_solve(c, t, inst, x0, rp, ref_len, u_prev, hull_A, hull_b, hull_nc, hset, s_hull[sub], ux, uy, st, mask, margin);
    if (t == 0 && live) {
        reinterpret_cast<double2*>(u0_out)[inst] = make_double2(ux, uy);
        status_out[inst] = st;
        if (active_out) active_out[inst] = mask;
        if (margin_out) margin_out[inst] = margin;
    }
}

// A plan of MPC-only records (include/wcqp.h: wcqp_qp_plan_*; the IK + MPC plans live in ik4.hip): workgroup (way, robot group)
// solves robots 4g .. 4g+3 of records way, way + ways, ... - BASELINE config 2 (batched DCM-MPC, B = 4096) is 4.3 MB per batch, a
// launch of its own is all ramp-up (5.7 us = 0.12 of the roofline); walked through in one launch, with the kernel's 65 registers
// allowing seven waves per SIMD, the card has many batches' loads in flight.  The record's pointers come out of memory: as_global.
#ifdef WCQP_MPC_PLAN_VGPR_HALF
__attribute__((amdgpu_num_vgpr(WCQP_MPC_PLAN_VGPR_HALF)))       // (half the unified register file's count: ik4.hip, WCQP_IK_PLAN_VGPR_HALF)
#endif
__global__ __launch_bounds__(kBlock)
void mpc_plan_kernel(MpcDeviceConsts c, int batch, const wcqp_qp_step* __restrict__ recs, int n_steps, int ways, int groups)
{
    __shared__ __attribute__((aligned(16))) double s_hull[kInstPerWave][WCQP_HULL_ROWS][4];
    using wcqp::as_global;
    const int way = (int)blockIdx.x / groups, blk = (int)blockIdx.x % groups;
#pragma unroll 1
    for (int r = way; r < n_steps; r += ways) {
        __asm__ volatile("" ::: "memory");
        int lane = threadIdx.x;
        __asm__ volatile("" : "+v"(lane));          // per-lane addresses are recomputed per record, not hoisted (they would all stay live)
        const int sub = lane / kLanesPerInstance, t = lane % kLanesPerInstance;
        const long inst_raw = (long)blk * kInstPerWave + sub;
        const bool live = inst_raw < batch;
        const long inst = live ? inst_raw : (long)batch - 1;
        const wcqp_qp_step& s = recs[r];
        const double2* rp = reinterpret_cast<const double2*>(as_global(s.ref)) + inst * s.ref_len;
        double ux, uy, margin;
        int st;
        unsigned mask;
        mpc_row_solve(c, t, inst, as_global(s.x0), rp, s.ref_len, as_global(s.u_prev), as_global(s.hull_A), as_global(s.hull_b), as_global(s.hull_nc), inst,
                      s_hull[sub], ux, uy, st, mask, margin);
        if (t == 0 && live) {
            reinterpret_cast<double2*>(as_global(s.u0))[inst] = make_double2(ux, uy);
            as_global(s.mpc_status)[inst] = st;
            if (s.mpc_active) as_global(s.mpc_active)[inst] = mask;
            if (s.mpc_margin) as_global(s.mpc_margin)[inst] = margin;
        }
        wcqp::wave_lds_fence();      // s_hull is rewritten by the next record
    }
}

}  // namespace

// ======================================================================================
struct wcqp_mpc_s {
    wcqp_mpc_params p{};
    int N = 0, n = 0, nx = 0, nu = 0;
    double a = 0, b = 0;
    std::vector<double> P, Aeq, grad_sub;          // dense copies of the reference's blocks
    std::vector<double> Gr;                        // (N+1)*4
    double Gx[4]{}, Gu[4]{}, S0[4]{};
    double* d_Gr = nullptr;
    int device = -1;
    wcqp::DeviceScratch scratch;
};

namespace {

// C (r x c) = A' (A is k x r) * B (k x c)
void matmul_tn(const std::vector<double>& A, const std::vector<double>& B, std::vector<double>& C,
               int k, int r, int c) {
    C.assign((size_t)r * c, 0.0);
    for (int kk = 0; kk < k; ++kk)
        for (int i = 0; i < r; ++i) {
            const double aik = A[(size_t)kk * r + i];
            if (aik == 0.0) continue;
            for (int j = 0; j < c; ++j) C[(size_t)i * c + j] += aik * B[(size_t)kk * c + j];
        }
}
void matmul_nn(const std::vector<double>& A, const std::vector<double>& B, std::vector<double>& C,
               int r, int k, int c) {
    C.assign((size_t)r * c, 0.0);
    for (int i = 0; i < r; ++i)
        for (int kk = 0; kk < k; ++kk) {
            const double aik = A[(size_t)i * k + kk];
            if (aik == 0.0) continue;
            for (int j = 0; j < c; ++j) C[(size_t)i * c + j] += aik * B[(size_t)kk * c + j];
        }
}

int build_constants(wcqp_mpc_s& h) {
    const wcqp_mpc_params& p = h.p;
    const int N = h.N, nx = h.nx, nu = h.nu, n = h.n;
    // Theta = I - shift (cpp:23-36), Rtilde = I_N (x) R (cpp:38-49)
    std::vector<double> theta((size_t)nu * nu, 0.0), Rt((size_t)nu * nu, 0.0), tmp, Pu;
    for (int i = 0; i < nu; ++i) theta[(size_t)i * nu + i] = 1.0;
    for (int i = 2; i < nu; ++i) theta[(size_t)i * nu + (i - 2)] = -1.0;
    for (int i = 0; i < N; ++i)
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 2; ++c) Rt[(size_t)(2 * i + r) * nu + 2 * i + c] = p.R[2 * r + c];
    matmul_nn(Rt, theta, tmp, nu, nu, nu);
    matmul_tn(theta, tmp, Pu, nu, nu, nu);                       // Theta' Rtilde Theta (cpp:65-77)
    h.P.assign((size_t)n * n, 0.0);
    for (int i = 0; i <= N; ++i)                                 // Qtilde (cpp:51-62)
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 2; ++c) h.P[(size_t)(2 * i + r) * n + 2 * i + c] = p.Q[2 * r + c];
    for (int r = 0; r < nu; ++r)                                 // cpp:126-144
        for (int c = 0; c < nu; ++c) h.P[(size_t)(nx + r) * n + nx + c] = Pu[(size_t)r * nu + c];
    // gradient sub-matrix = -Theta' Rtilde e1 (cpp:148-168): first two columns of -(Theta' Rtilde)
    std::vector<double> tr;
    matmul_tn(theta, Rt, tr, nu, nu, nu);
    h.grad_sub.assign((size_t)nu * 2, 0.0);
    for (int r = 0; r < nu; ++r)
        for (int c = 0; c < 2; ++c) h.grad_sub[(size_t)r * 2 + c] = -tr[(size_t)r * nu + c];
    // dynamics (cpp:230-237) and equality block (cpp:79-124)
    const double omega = std::sqrt(p.gravity / p.com_height);
    h.a = std::exp(omega * p.sampling_time);
    h.b = 1.0 - h.a;
    h.Aeq.assign((size_t)nx * n, 0.0);
    for (int i = 0; i < nx; ++i) h.Aeq[(size_t)i * n + i] = -1.0;
    for (int i = 0; i < N; ++i)
        for (int r = 0; r < 2; ++r) {
            h.Aeq[(size_t)(2 * (i + 1) + r) * n + 2 * i + r] = h.a;
            h.Aeq[(size_t)(2 * (i + 1) + r) * n + nx + 2 * i + r] = h.b;
        }
    // condense: rows nx, nx+1 of K^-1 (K symmetric => rows == solved columns)
    const int m = n + nx;
    std::vector<double> K((size_t)m * m, 0.0);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) K[(size_t)r * m + c] = h.P[(size_t)r * n + c];
    for (int r = 0; r < nx; ++r)
        for (int c = 0; c < n; ++c) {
            const double v = h.Aeq[(size_t)r * n + c];
            K[(size_t)(n + r) * m + c] = v;
            K[(size_t)c * m + n + r] = v;
        }
    std::vector<int> piv;
    if (!wcqp::lu_factor(K, m, piv)) return WCQP_E_NUMERIC;
    std::vector<double> row0(m, 0.0), row1(m, 0.0);
    row0[nx] = 1.0; row1[nx + 1] = 1.0;
    wcqp::lu_solve(K, piv, m, row0.data());
    wcqp::lu_solve(K, piv, m, row1.data());
    const double* rows[2] = {row0.data(), row1.data()};
    h.Gr.assign((size_t)(N + 1) * 4, 0.0);
    for (int i = 0; i <= N; ++i)
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 2; ++c)      // -q_x[i] = Q r_i (MPCSolver.cpp:195-196)
                h.Gr[(size_t)i * 4 + 2 * r + c] = rows[r][2 * i] * p.Q[c] + rows[r][2 * i + 1] * p.Q[2 + c];
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) {
            // -q_u[0:2] = R u_prev (MPCSolver.cpp:244-245 with grad_sub = -Theta'Rtilde e1)
            h.Gu[2 * r + c] = rows[r][nx] * p.R[c] + rows[r][nx + 1] * p.R[2 + c];
            h.Gx[2 * r + c] = -rows[r][n + c];                    // beq[0:2] = -x0 (MPCSolver.cpp:143-146)
            h.S0[2 * r + c] = rows[r][nx + c];
        }
    return WCQP_OK;
}

int ensure_device(wcqp_mpc_s* h) {
    if (h->d_Gr) return WCQP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        std::fprintf(stderr, "[wcqp] no HIP device: the MPC solve path has no CPU fallback\n");
        return WCQP_E_HIP;
    }
    WCQP_HIP_TRY(hipGetDevice(&h->device));
    WCQP_HIP_TRY(hipMalloc(&h->d_Gr, h->Gr.size() * sizeof(double)));
    WCQP_HIP_TRY(hipMemcpy(h->d_Gr, h->Gr.data(), h->Gr.size() * sizeof(double), hipMemcpyHostToDevice));
    return WCQP_OK;
}

}  // namespace

namespace wcqp {

int mpc_enqueue(wcqp_mpc_t h, int batch, const double* x0, const double* ref, int ref_len, int ref_stride,
                const int* ref_start_dev, const double* u_prev,
                const double* hull_A, const double* hull_b, const int* hull_nc, int hull_sets, const int* hull_sel,
                double* u0, int* status, unsigned* active, double* margin, hipStream_t stream) {
    if (!h || batch < 0 || ref_len < 1 || ref_stride < ref_len || hull_sets < 1) return WCQP_E_INVALID;
    if (!x0 || !ref || !u_prev || !hull_A || !hull_b || !hull_nc || !u0 || !status) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    const int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    MpcDeviceConsts c;
    mpc_device_consts(h, &c);
    const unsigned grid = (unsigned)((batch + kInstPerWave - 1) / kInstPerWave);
    hipLaunchKernelGGL(mpc_condensed_kernel, dim3(grid), dim3(kBlock), 0, stream,
                       c, batch, x0, ref, ref_len, ref_stride, ref_start_dev, u_prev, hull_A, hull_b, hull_nc, hull_sets, hull_sel,
                       u0, status, active, margin);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

int mpc_prepare(wcqp_mpc_t h) { return h ? ensure_device(h) : WCQP_E_INVALID; }
int mpc_launch_plan(wcqp_mpc_t h, int batch, const wcqp_qp_step* d_recs, int n_steps, int ways, hipStream_t stream) {
    if (!h || !d_recs || batch < 1 || n_steps < 1 || ways < 1) return WCQP_E_INVALID;
    wcqp_mpc::MpcDeviceConsts c;
    mpc_device_consts(h, &c);
    const int groups = (batch + wcqp_mpc::kInstPerWave - 1) / wcqp_mpc::kInstPerWave;
    hipLaunchKernelGGL(mpc_plan_kernel, dim3((unsigned)(groups * ways)), dim3(kBlock), 0, stream, c, batch, d_recs, n_steps, ways, groups);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}
void mpc_device_consts(wcqp_mpc_t h, wcqp_mpc::MpcDeviceConsts* c) {
    c->Gr = h->d_Gr;
    std::memcpy(c->Gx, h->Gx, sizeof(c->Gx));
    std::memcpy(c->Gu, h->Gu, sizeof(c->Gu));
    std::memcpy(c->S0, h->S0, sizeof(c->S0));
    c->feas_tol = h->p.feas_tol;
    c->hull_tol = h->p.convex_hull_tolerance;
    c->N = h->N;
}
int mpc_horizon(wcqp_mpc_t h) { return h ? h->N : 0; }
void mpc_dynamics(wcqp_mpc_t h, double* a, double* b) { *a = h->a; *b = h->b; }

}  // namespace wcqp

extern "C" {

int wcqp_mpc_create(const wcqp_mpc_params* params, wcqp_mpc_t* out) {
    if (!params || !out) return WCQP_E_INVALID;
    if (params->horizon < 1 || params->horizon > 4096) return WCQP_E_UNSUPPORTED;
    if (!(params->sampling_time > 0) || !(params->com_height > 0) || !(params->gravity > 0)) return WCQP_E_INVALID;
    if (params->Q[1] != params->Q[2] || params->R[1] != params->R[2]) return WCQP_E_INVALID;  // symmetric weights only
    wcqp_mpc_s* h = new (std::nothrow) wcqp_mpc_s();
    if (!h) return WCQP_E_NOMEM;
    h->p = *params;
    if (!(h->p.feas_tol > 0)) h->p.feas_tol = 1e-10;
    h->N = params->horizon;
    h->nx = 2 * (h->N + 1);
    h->nu = 2 * h->N;
    h->n = h->nx + h->nu;
    const int rc = build_constants(*h);
    if (rc != WCQP_OK) { delete h; return rc; }
    *out = h;
    return WCQP_OK;
}

int wcqp_mpc_destroy(wcqp_mpc_t h) {
    if (!h) return WCQP_E_INVALID;
    if (h->d_Gr) (void)hipFree(h->d_Gr);
    h->scratch.release();
    delete h;
    return WCQP_OK;
}

int wcqp_mpc_get_condensed(wcqp_mpc_t h, double* Gr, double* Gx, double* Gu, double* Sigma0) {
    if (!h) return WCQP_E_INVALID;
    if (Gr) std::memcpy(Gr, h->Gr.data(), h->Gr.size() * sizeof(double));
    if (Gx) std::memcpy(Gx, h->Gx, sizeof(h->Gx));
    if (Gu) std::memcpy(Gu, h->Gu, sizeof(h->Gu));
    if (Sigma0) std::memcpy(Sigma0, h->S0, sizeof(h->S0));
    return WCQP_OK;
}

int wcqp_mpc_get_matrices(wcqp_mpc_t h, double* P, double* A_eq, double* grad_sub) {
    if (!h) return WCQP_E_INVALID;
    if (P) std::memcpy(P, h->P.data(), h->P.size() * sizeof(double));
    if (A_eq) std::memcpy(A_eq, h->Aeq.data(), h->Aeq.size() * sizeof(double));
    if (grad_sub) std::memcpy(grad_sub, h->grad_sub.data(), h->grad_sub.size() * sizeof(double));
    return WCQP_OK;
}

int wcqp_mpc_solve_device(wcqp_mpc_t h, int32_t batch,
                          const double* x0, const double* ref, int32_t ref_len, const double* u_prev,
                          const double* hull_A, const double* hull_b, const int32_t* hull_nc,
                          double* u0, int32_t* status, uint32_t* active, double* margin, void* stream) {
    return wcqp::mpc_enqueue(h, batch, x0, ref, ref_len, ref_len, nullptr, u_prev, hull_A, hull_b, hull_nc, 1, nullptr,
                             u0, status, active, margin, (hipStream_t)stream);
}

int wcqp_mpc_solve_host(wcqp_mpc_t h, int32_t batch,
                        const double* x0, const double* ref, int32_t ref_len, const double* u_prev,
                        const double* hull_A, const double* hull_b, const int32_t* hull_nc,
                        double* u0, int32_t* status, uint32_t* active, double* margin) {
    if (!h || batch < 0 || ref_len < 1) return WCQP_E_INVALID;
    if (!x0 || !ref || !u_prev || !hull_A || !hull_b || !hull_nc || !u0 || !status) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    const size_t B = (size_t)batch;
    const size_t o_x0 = 0, o_ref = o_x0 + B * 2, o_up = o_ref + B * ref_len * 2, o_hA = o_up + B * 2,
                 o_hb = o_hA + B * 16, o_u0 = o_hb + B * 8, o_mg = o_u0 + B * 2, n_dbl = o_mg + B;
    const size_t bytes = n_dbl * 8 + B * 4 * 3;
    rc = h->scratch.reserve(bytes);
    if (rc != WCQP_OK) return rc;
    double* d = static_cast<double*>(h->scratch.ptr);
    int32_t* d_nc = reinterpret_cast<int32_t*>(d + n_dbl);
    int32_t* d_st = d_nc + B;
    uint32_t* d_ac = reinterpret_cast<uint32_t*>(d_st + B);
    WCQP_HIP_TRY(hipMemcpy(d + o_x0, x0, B * 16, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_ref, ref, B * ref_len * 16, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_up, u_prev, B * 16, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_hA, hull_A, B * 128, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_hb, hull_b, B * 64, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d_nc, hull_nc, B * 4, hipMemcpyHostToDevice));
    rc = wcqp_mpc_solve_device(h, batch, d + o_x0, d + o_ref, ref_len, d + o_up, d + o_hA, d + o_hb, d_nc,
                               d + o_u0, d_st, d_ac, d + o_mg, nullptr);
    if (rc != WCQP_OK) return rc;
    WCQP_HIP_TRY(hipDeviceSynchronize());
    WCQP_HIP_TRY(hipMemcpy(u0, d + o_u0, B * 16, hipMemcpyDeviceToHost));
    WCQP_HIP_TRY(hipMemcpy(status, d_st, B * 4, hipMemcpyDeviceToHost));
    if (active) WCQP_HIP_TRY(hipMemcpy(active, d_ac, B * 4, hipMemcpyDeviceToHost));
    if (margin) WCQP_HIP_TRY(hipMemcpy(margin, d + o_mg, B * 8, hipMemcpyDeviceToHost));
    return WCQP_OK;
}

}  // extern "C"
