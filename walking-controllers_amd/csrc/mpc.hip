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

namespace {

constexpr int kLanesPerInstance = 16;   // one DPP row per instance, 4 instances per wave
constexpr int kInstPerWave = 64 / kLanesPerInstance;
constexpr int kBlock = 64;              // one wavefront per workgroup

// candidate 0: no row; 1..8: single row e = id-1; 9..36: row pairs (e < f)
__device__ const unsigned char kPairE[28] = {0,0,0,0,0,0,0, 1,1,1,1,1,1, 2,2,2,2,2, 3,3,3,3, 4,4,4, 5,5, 6};
__device__ const unsigned char kPairF[28] = {1,2,3,4,5,6,7, 2,3,4,5,6,7, 3,4,5,6,7, 4,5,6,7, 5,6,7, 6,7, 7};
constexpr int kNumCand = 1 + 8 + 28;

// One DPP move of both halves of a double inside a row of 16 lanes (= one instance here).
template <int CTRL>
__device__ __forceinline__ double row_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int row_move(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
// the four butterfly partners inside a row: xor 1, xor 2 (quad permutes), then half-mirror and
// mirror, which pair quads / octets whose lanes already agree
#define WCQP_ROW_STEPS(X) X(0xB1) X(0x4E) X(0x141) X(0x140)

struct MpcDeviceConsts {
    const double* Gr;     // (N+1) x 2 x 2
    double Gx[4], Gu[4], S0[4];
    double feas_tol, hull_tol;
    int N;
};

__global__ __launch_bounds__(kBlock)
void mpc_condensed_kernel(MpcDeviceConsts c, int batch,
                          const double* __restrict__ x0, const double* __restrict__ ref, int ref_len,
                          int ref_stride, const int* __restrict__ ref_start, int* __restrict__ ref_start_copy,
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

    // ---- u0_unc = sum_i Gr_i r_i + Gx x0 + Gu u_prev ---------------------------------
    // the reference window of an instance starts `*ref_start` stages into its trajectory (the
    // deque of the reference advanced by that many ticks); instances are `ref_stride` stages apart
    const int start = ref_start ? *ref_start : 0;
    // tick pipeline: the next kernel of the tick reads the tick index from this copy, so that it may
    // advance `*ref_start` itself while some of its workgroups have not started yet
    if (ref_start_copy && blockIdx.x == 0 && lane == 0) *ref_start_copy = start;
    const double2* rp = reinterpret_cast<const double2*>(ref) + inst * ref_stride + start;
    const double2* gp = reinterpret_cast<const double2*>(c.Gr);
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
    // hull rows -> LDS (lanes 0..7 of the instance own one row each)
    // an instance may carry several precomputed row sets (one per contact pair); `hull_sel` picks
    // the live one, so a contact change costs no copy (tick pipeline)
    const long hset = inst * hull_sets + (hull_sel ? hull_sel[inst] : 0);
    int nc = hull_nc[hset];
    nc = nc < 0 ? 0 : (nc > WCQP_HULL_ROWS ? WCQP_HULL_ROWS : nc);
    double rax = 0.0, ray = 0.0, rb = 0.0, rn = 0.0;    // this lane's own hull row (lanes 0..7)
    if (t < WCQP_HULL_ROWS) {
        const double2 a = reinterpret_cast<const double2*>(hull_A)[hset * WCQP_HULL_ROWS + t];
        rb = hull_b[hset * WCQP_HULL_ROWS + t];
        rax = a.x; ray = a.y; rn = sqrt(a.x * a.x + a.y * a.y);
        s_hull[sub][t][0] = rax; s_hull[sub][t][1] = ray; s_hull[sub][t][2] = rb; s_hull[sub][t][3] = rn;
    }
    // butterfly over the row (DPP, no LDS-pipe round trips): every lane ends with the same sum
#define WCQP_SUM_STEP(C) ux += row_move<C>(ux); uy += row_move<C>(uy);
    WCQP_ROW_STEPS(WCQP_SUM_STEP)
#undef WCQP_SUM_STEP
    wcqp::wave_lds_fence();

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
    const bool row_violated = t < nc && (rax * ux + ray * uy - rb) > c.feas_tol;
    if (__ballot(row_violated) == 0ull) {
        best_cost = 0.0; best_id = 0;
    } else
    for (int id = t; id < kNumCand; id += kLanesPerInstance) {
        int e = -1, f = -1;
        if (id >= 1 && id <= 8) e = id - 1;
        else if (id > 8) { e = kPairE[id - 9]; f = kPairF[id - 9]; }
        if (e >= nc || f >= nc) continue;
        double px = ux, py = uy, cost = 0.0;
        unsigned mask = 0;
        bool ok = true;
        if (e >= 0) {
            const double aex = s_hull[sub][e][0], aey = s_hull[sub][e][1];
            const double sex = s00 * aex + s01 * aey, sey = s10 * aex + s11 * aey;   // Sigma0 a_e
            const double ree = aex * sex + aey * sey;
            const double re  = aex * ux + aey * uy - s_hull[sub][e][2];
            mask = 1u << e;
            if (f < 0) {
                ok = ree > 0.0;
                const double mu = ok ? re * wcqp::fast_rcp(ree) : 0.0;
                px = ux - sex * mu; py = uy - sey * mu;
                cost = mu * re;
            } else {
                const double afx = s_hull[sub][f][0], afy = s_hull[sub][f][1];
                const double sfx = s00 * afx + s01 * afy, sfy = s10 * afx + s11 * afy;
                const double rff = afx * sfx + afy * sfy;
                const double ref_ = aex * sfx + aey * sfy;
                const double rf  = afx * ux + afy * uy - s_hull[sub][f][2];
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
            const double res = s_hull[sub][k][0] * px + s_hull[sub][k][1] * py - s_hull[sub][k][2];
            ok = ok && (k == e || k == f || res <= c.feas_tol);
        }
        if (ok && (cost < best_cost || (cost == best_cost && id < best_id))) {
            best_cost = cost; best_x = px; best_y = py; best_mask = mask; best_id = id;
        }
    }
#define WCQP_MIN_STEP(C) {                                                              \
        const double oc = row_move<C>(best_cost), ox = row_move<C>(best_x), oy = row_move<C>(best_y); \
        const int om = row_move<C>((int)best_mask), oi = row_move<C>(best_id);                 \
        if (oc < best_cost || (oc == best_cost && oi < best_id)) {                             \
            best_cost = oc; best_x = ox; best_y = oy; best_mask = (unsigned)om; best_id = oi;  \
        } }
    WCQP_ROW_STEPS(WCQP_MIN_STEP)
#undef WCQP_MIN_STEP
    // signed distance to the hull boundary (computeMargin semantics): every row lane evaluates its
    // own row, row-min by DPP
    double margin = (t < nc && rn > 0.0) ? (rb - rax * best_x - ray * best_y) / rn : std::numeric_limits<double>::infinity();
#define WCQP_MARGIN_STEP(C) margin = fmin(margin, row_move<C>(margin));
    WCQP_ROW_STEPS(WCQP_MARGIN_STEP)
#undef WCQP_MARGIN_STEP
    if (t == 0 && live) {
        int st = best_id < kNumCand ? WCQP_STATUS_SOLVED : WCQP_STATUS_INFEASIBLE;
        // WalkingController::solve: computeMargin(u0) < -tolerance => failure (cpp:513-517)
        if (st == WCQP_STATUS_SOLVED && margin < -c.hull_tol) st = WCQP_STATUS_OUTSIDE_HULL;
        reinterpret_cast<double2*>(u0_out)[inst] = make_double2(best_x, best_y);
        status_out[inst] = st;
        if (active_out) active_out[inst] = best_mask;
        if (margin_out) margin_out[inst] = margin;
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
                const int* ref_start_dev, int* ref_start_copy_dev, const double* u_prev,
                const double* hull_A, const double* hull_b, const int* hull_nc, int hull_sets, const int* hull_sel,
                double* u0, int* status, unsigned* active, double* margin, hipStream_t stream) {
    if (!h || batch < 0 || ref_len < 1 || ref_stride < ref_len || hull_sets < 1) return WCQP_E_INVALID;
    if (!x0 || !ref || !u_prev || !hull_A || !hull_b || !hull_nc || !u0 || !status) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    const int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    MpcDeviceConsts c;
    c.Gr = h->d_Gr;
    std::memcpy(c.Gx, h->Gx, sizeof(c.Gx));
    std::memcpy(c.Gu, h->Gu, sizeof(c.Gu));
    std::memcpy(c.S0, h->S0, sizeof(c.S0));
    c.feas_tol = h->p.feas_tol;
    c.hull_tol = h->p.convex_hull_tolerance;
    c.N = h->N;
    const unsigned grid = (unsigned)((batch + kInstPerWave - 1) / kInstPerWave);
    hipLaunchKernelGGL(mpc_condensed_kernel, dim3(grid), dim3(kBlock), 0, stream,
                       c, batch, x0, ref, ref_len, ref_stride, ref_start_dev, ref_start_copy_dev, u_prev, hull_A, hull_b, hull_nc, hull_sets, hull_sel,
                       u0, status, active, margin);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

int mpc_prepare(wcqp_mpc_t h) { return h ? ensure_device(h) : WCQP_E_INVALID; }
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
    return wcqp::mpc_enqueue(h, batch, x0, ref, ref_len, ref_len, nullptr, nullptr, u_prev, hull_A, hull_b, hull_nc, 1, nullptr,
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
