// batched kinematics — forward kinematics of a kinematic tree and the free-floating
// Jacobians (MIXED representation) that feed the QP-IK, for B robots at once.
//
// Reference path replaced (citations relative to /root/reference/modules/Walking_module):
//   WalkingFK::setInternalRobotState                      src/WalkingForwardKinematics.cpp:258-276
//   getLeftFootToWorldTransform / getRightFoot... / getNeckOrientation   :354-366, 402-405
//   evaluateCoM / getCoMPosition                          :312-340
//   getLeftFootJacobian / getRightFootJacobian / getNeckJacobian / getCoMJacobian   :436-454 (MIXED, :33)
// which are thin calls into iDynTree::KinDynComputations on a URDF model.  Neither is in the
// repository: the tree comes in as a table (wcqp_kin_params), and the CPU restatement the kernel is
// checked against is oracle/kin_spec.py (itself pinned by differentiating its forward kinematics).
//
// Layout: 32 lanes per instance (two per wave64); lane i owns COLUMN i of every Jacobian (0..5 base,
// 6 + j joint j), so every output row is one coalesced 232-byte segment.  The tree is walked level by
// level (parent[j] < j, depth <= 8): a joint's world frame needs its parent's, through LDS.
// HBM-bound by its output: 280 B in, 4464 B out per instance.
#include <cmath>
#include <cstring>
#include <new>
#include <vector>
#include "wcqp_internal.h"
#include "tick_device.h"
#include "hull_device.h"

namespace {

constexpr int kMaxDof = WCQP_KIN_MAX_DOF;     // 32
constexpr int kStateLen = WCQP_IK_STATE_LEN;

struct KinDev {
    int dof, max_level;
    int parent[kMaxDof], level[kMaxDof];
    unsigned desc_mask[kMaxDof];              // joints moved by joint j (itself included)
    unsigned path_mask[3];                    // joints on the path root -> frame f
    double R0[kMaxDof][9], p0[kMaxDof][3], axis[kMaxDof][3], mass[kMaxDof], com[kMaxDof][3];
    double root_mass, root_com[3], total_mass;
    int frame_joint[3];
    double frame_R[3][9], frame_p[3][3];
};

__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ __forceinline__ void mat3_vec(const double* A, const double* v, double* o) {
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = A[3 * r] * v[0] + A[3 * r + 1] * v[1] + A[3 * r + 2] * v[2];
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// LDS per instance (doubles): Tw [32][12] world frame of every joint, MC [33][4] {m c, m} per link
// (slot 32 = root link), FR [3][12] attached frames, CT [4] total first moment / mass
constexpr int OFF_TW = 0, OFF_MC = 32 * 12, OFF_FR = OFF_MC + 33 * 4, OFF_CT = OFF_FR + 36, PER_INST = OFF_CT + 4;

// TICK: the tick pipeline's per-tick call (WalkingModule.cpp:715, 396-410): base pose from the plant, support-polygon
// rows rebuilt on a contact change (tick_device.h: KinTick)
template <bool TICK>
__global__ __launch_bounds__(64)
void kin_jacobians_kernel(const KinDev* __restrict__ md, int batch,
                          const double* __restrict__ base, const double* __restrict__ q,
                          double* __restrict__ JL, double* __restrict__ JR, double* __restrict__ JN, double* __restrict__ JC,
                          double* __restrict__ state, wcqp_tick::KinTick kt)
{
    __shared__ __attribute__((aligned(16))) double smem[2][PER_INST];
    const int lane = threadIdx.x, half = lane >> 5, i = lane & 31;
    const long inst_raw = (long)blockIdx.x * 2 + half;
    const bool live = inst_raw < batch;
    const long inst = live ? inst_raw : (long)batch - 1;
    double* S = smem[half];
    const int dof = md->dof;
    const int j = i - 6;                                   // joint of this lane's column
    const bool is_joint = j >= 0 && j < dof;
    const int jc = is_joint ? j : 0;

    // base pose
    double pb[3], Rb[9];
    {
        if constexpr (TICK) {
            // the tree is walked in base coordinates first; the base pose follows from the anchor foot below
#pragma unroll
            for (int k = 0; k < 3; ++k) pb[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) Rb[k] = (k % 4 == 0) ? 1.0 : 0.0;
        } else {
            const double* b = base + inst * 12;
#pragma unroll
            for (int k = 0; k < 3; ++k) pb[k] = b[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) Rb[k] = b[3 + k];
        }
    }
    // own joint: local rotation R0 * Rot(axis, q)   (Rodrigues)
    double Rloc[9], p0[3], ax[3];
    const int par = md->parent[jc], lvl = is_joint ? md->level[jc] : 0;
    {
        const double qj = q[inst * dof + jc];
        double sn, cs;
        sincos(qj, &sn, &cs);
#pragma unroll
        for (int k = 0; k < 3; ++k) { ax[k] = md->axis[jc][k]; p0[k] = md->p0[jc][k]; }
        const double v = 1.0 - cs;
        const double Rq[9] = {cs + v * ax[0] * ax[0],         v * ax[0] * ax[1] - sn * ax[2], v * ax[0] * ax[2] + sn * ax[1],
                              v * ax[1] * ax[0] + sn * ax[2], cs + v * ax[1] * ax[1],         v * ax[1] * ax[2] - sn * ax[0],
                              v * ax[2] * ax[0] - sn * ax[1], v * ax[2] * ax[1] + sn * ax[0], cs + v * ax[2] * ax[2]};
        double R0[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) R0[k] = md->R0[jc][k];
        mat3_mul(R0, Rq, Rloc);
    }
    // world frames, level by level
    double Rw[9], pw[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rw[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) pw[k] = 0.0;
    const int max_level = md->max_level;
    for (int L = 1; L <= max_level; ++L) {
        if (lvl == L) {
            double Rp[9], pp[3];
            if (par < 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) Rp[k] = Rb[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) pp[k] = pb[k];
            } else {
                const double* T = S + OFF_TW + par * 12;
#pragma unroll
                for (int k = 0; k < 9; ++k) Rp[k] = T[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) pp[k] = T[9 + k];
            }
            mat3_mul(Rp, Rloc, Rw);
            double d[3];
            mat3_vec(Rp, p0, d);
#pragma unroll
            for (int k = 0; k < 3; ++k) pw[k] = pp[k] + d[k];
            double* T = S + OFF_TW + j * 12;
#pragma unroll
            for (int k = 0; k < 9; ++k) T[k] = Rw[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) T[9 + k] = pw[k];
        }
        wcqp::wave_lds_fence();
    }
    if constexpr (TICK) {
        // anchor foot: its sole frame in base coordinates, then world_T_base = world_T_sole,desired * (base_T_sole)^-1
        const int t_now = kt.tick2[kt.phase];
        const int side = ((t_now + kt.phase0[inst]) % (2 * kt.step_ticks)) / kt.step_ticks;     // 0: left is the stance foot
        const int jf = md->frame_joint[side];
        const double* T = S + OFF_TW + jf * 12;
        double Rj[9], fR[9], fp[3], Ra[9], pa[3], d[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) { Rj[k] = T[k]; fR[k] = md->frame_R[side][k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) fp[k] = md->frame_p[side][k];
        mat3_mul(Rj, fR, Ra);
        mat3_vec(Rj, fp, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) pa[k] = T[9 + k] + d[k];
        const double* sd = state + inst * kStateLen + (side ? 36 : 24);      // desired pose of the anchor sole: p (3), R (9)
        double Rd[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rd[k] = sd[3 + k];
        // Rb = Rd Ra'
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Rb[3 * r + c] = Rd[3 * r] * Ra[3 * c] + Rd[3 * r + 1] * Ra[3 * c + 1] + Rd[3 * r + 2] * Ra[3 * c + 2];
        mat3_vec(Rb, pa, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) pb[k] = sd[k] - d[k];
        // this lane's joint frame, now in world coordinates
        double Rl[9], pl[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rl[k] = Rw[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pl[k] = pw[k];
        mat3_mul(Rb, Rl, Rw);
        mat3_vec(Rb, pl, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) pw[k] = pb[k] + d[k];
        wcqp::wave_lds_fence();               // every lane has read the base-frame tree: it may be overwritten
        if (is_joint) {
            double* Tm = S + OFF_TW + j * 12;
#pragma unroll
            for (int k = 0; k < 9; ++k) Tm[k] = Rw[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) Tm[9 + k] = pw[k];
        }
        wcqp::wave_lds_fence();
    }
    // joint axis in world (a rotation about the axis leaves it unchanged: R_w * axis)
    double aw[3];
    mat3_vec(Rw, ax, aw);
    // link first moments; the root link goes to slot 32 (lane 0)
    if (is_joint) {
        double cl[3], cj[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) cj[k] = md->com[jc][k];
        mat3_vec(Rw, cj, cl);
        const double m = md->mass[jc];
        double* e = S + OFF_MC + j * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) e[k] = m * (pw[k] + cl[k]);
        e[3] = m;
    }
    if (i == 0) {
        double rc[3], cr[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) rc[k] = md->root_com[k];
        mat3_vec(Rb, rc, cr);
        double* e = S + OFF_MC + 32 * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) e[k] = md->root_mass * (pb[k] + cr[k]);
        e[3] = md->root_mass;
    }
    // attached frames (lanes 0..2)
    if (i < 3) {
        const int jf = md->frame_joint[i];
        const double* T = S + OFF_TW + jf * 12;
        // (T of the frame's joint is complete: all levels are done)
        double Rj[9], fR[9], fp[3], Rf[9], d[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) { Rj[k] = T[k]; fR[k] = md->frame_R[i][k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) fp[k] = md->frame_p[i][k];
        mat3_mul(Rj, fR, Rf);
        mat3_vec(Rj, fp, d);
        double* F = S + OFF_FR + i * 12;
#pragma unroll
        for (int k = 0; k < 9; ++k) F[k] = Rf[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) F[9 + k] = T[9 + k] + d[k];
    }
    wcqp::wave_lds_fence();
    // total first moment and mass: lanes 0..3 take one component each
    if (i < 4) {
        double acc = S[OFF_MC + 32 * 4 + i];
        for (int k = 0; k < dof; ++k) acc += S[OFF_MC + k * 4 + i];
        S[OFF_CT + i] = acc;
    }
    // subtree first moment and mass of this lane's joint
    double ms = 0.0, mcs[3] = {0.0, 0.0, 0.0};
    if (is_joint) {
        unsigned dm = md->desc_mask[jc];
        while (dm) {
            const int k = __ffs(dm) - 1;
            dm &= dm - 1;
            const double* e = S + OFF_MC + k * 4;
            mcs[0] += e[0]; mcs[1] += e[1]; mcs[2] += e[2]; ms += e[3];
        }
    }
    wcqp::wave_lds_fence();
    const double Mtot = S[OFF_CT + 3];
    const double iM = 1.0 / Mtot;
    double ctot[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ctot[k] = S[OFF_CT + k] * iM;

    // ---------------- Jacobian columns ---------------------------------------------------------
    if (live && i < 6 + dof) {
        const int ncol = 6 + dof;
        double e[3] = {0.0, 0.0, 0.0};
        if (i < 6) e[i % 3] = 1.0;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const double* F = S + OFF_FR + f * 12;
            double lin[3] = {0.0, 0.0, 0.0}, ang[3] = {0.0, 0.0, 0.0};
            if (i < 3) {
                lin[0] = e[0]; lin[1] = e[1]; lin[2] = e[2];
            } else if (i < 6) {
                const double d[3] = {F[9] - pb[0], F[10] - pb[1], F[11] - pb[2]};
                cross3(e, d, lin);                           // column k of -S(p_f - p_b) = e_k x (p_f - p_b)
                ang[0] = e[0]; ang[1] = e[1]; ang[2] = e[2];
            } else if ((md->path_mask[f] >> j) & 1u) {
                const double d[3] = {F[9] - pw[0], F[10] - pw[1], F[11] - pw[2]};
                cross3(aw, d, lin);
                ang[0] = aw[0]; ang[1] = aw[1]; ang[2] = aw[2];
            }
            if (f < 2) {
                double* J = (f == 0 ? JL : JR) + inst * (6 * ncol);
#pragma unroll
                for (int r = 0; r < 3; ++r) { J[r * ncol + i] = lin[r]; J[(3 + r) * ncol + i] = ang[r]; }
            } else {
                double* J = JN + inst * (3 * ncol);
#pragma unroll
                for (int r = 0; r < 3; ++r) J[r * ncol + i] = ang[r];        // the IK keeps the angular rows (setNeckJacobian)
            }
        }
        double lin[3] = {0.0, 0.0, 0.0};
        if (i < 3) {
            lin[0] = e[0]; lin[1] = e[1]; lin[2] = e[2];
        } else if (i < 6) {
            const double d[3] = {ctot[0] - pb[0], ctot[1] - pb[1], ctot[2] - pb[2]};
            cross3(e, d, lin);
        } else {
            const double d[3] = {(mcs[0] - ms * pw[0]) * iM, (mcs[1] - ms * pw[1]) * iM, (mcs[2] - ms * pw[2]) * iM};
            cross3(aw, d, lin);
        }
        double* J = JC + inst * (3 * ncol);
#pragma unroll
        for (int r = 0; r < 3; ++r) J[r * ncol + i] = lin[r];
    }
    // ---------------- actual poses into the IK state block ---------------------------------------
    if (state && live) {
        double* s = state + inst * kStateLen;
        if (i < 12) {                                         // left / right foot: p (3), R (9)
            const double* FL = S + OFF_FR, *FRt = S + OFF_FR + 12;
            s[i] = i < 3 ? FL[9 + i] : FL[i - 3];
            s[12 + i] = i < 3 ? FRt[9 + i] : FRt[i - 3];
        }
        if (i < 9) s[48 + i] = S[OFF_FR + 24 + i];            // neck orientation
        if (i < 3) s[66 + i] = ctot[i];                       // CoM position
    }
    if constexpr (TICK) {
        // setConvexHullConstraint: the rows change only when the contact pair does (cpp:369-374); they are built
        // from the DESIRED foot transforms (the planned footsteps, WalkingModule.cpp:609-613), entries 24..47 of the
        // pose block
        if (i == 0 && live) {
            const int code = kt.sel[inst];
            if (code != kt.sel_built[inst]) {
                const double* sd = state + inst * kStateLen;
                double px[8], py[8];
                int np = 0;
                if (code == 0 || code == 2) wcqp_hull::foot_points(kt.rect, sd + 24, px, py, np);
                if (code == 1 || code == 2) wcqp_hull::foot_points(kt.rect, sd + 36, px, py, np);
                kt.hull_nc[inst] = wcqp_hull::hull_rows(px, py, np, kt.hull_A + inst * 16, kt.hull_b + inst * 8);
                kt.sel_built[inst] = code;
            }
        }
    }
}

}  // namespace

struct wcqp_kin_s {
    wcqp_kin_params p{};
    KinDev hd{};
    KinDev* d_model = nullptr;
    wcqp::DeviceScratch scratch;
};

namespace {

int ensure_device(wcqp_kin_s* h) {
    if (h->d_model) return WCQP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        std::fprintf(stderr, "[wcqp] no HIP device: the kinematics kernel has no CPU fallback\n");
        return WCQP_E_HIP;
    }
    WCQP_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_model), sizeof(KinDev)));
    WCQP_HIP_TRY(hipMemcpy(h->d_model, &h->hd, sizeof(KinDev), hipMemcpyHostToDevice));
    return WCQP_OK;
}

}  // namespace

namespace wcqp {
int kin_prepare(wcqp_kin_t h) { return h ? ensure_device(h) : WCQP_E_INVALID; }
int kin_enqueue_tick(wcqp_kin_t h, int batch, const wcqp_tick::KinTick& kt, const double* q,
                     double* J_left, double* J_right, double* J_neck, double* J_com, double* state, hipStream_t stream) {
    if (!h || !h->d_model || batch < 1 || !state || !q || !J_left || !J_right || !J_neck || !J_com) return WCQP_E_INVALID;
    if (!kt.tick2 || !kt.phase0 || !kt.sel || !kt.sel_built || !kt.hull_A || !kt.hull_b || !kt.hull_nc || kt.step_ticks < 1) return WCQP_E_INVALID;
    const unsigned grid = (unsigned)((batch + 1) / 2);
    hipLaunchKernelGGL(kin_jacobians_kernel<true>, dim3(grid), dim3(64), 0, stream, h->d_model, batch, nullptr, q,
                       J_left, J_right, J_neck, J_com, state, kt);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}
}  // namespace wcqp

extern "C" {

int wcqp_kin_create(const wcqp_kin_params* params, wcqp_kin_t* out) {
    if (!params || !out) return WCQP_E_INVALID;
    const int n = params->dof;
    if (n < 1 || 6 + n > 32) return WCQP_E_UNSUPPORTED;       // one column per lane of a 32-lane group
    if (!(params->root_mass >= 0.0)) return WCQP_E_INVALID;
    wcqp_kin_s* h = new (std::nothrow) wcqp_kin_s();
    if (!h) return WCQP_E_NOMEM;
    h->p = *params;
    KinDev& d = h->hd;
    d.dof = n; d.max_level = 0; d.root_mass = params->root_mass; d.total_mass = params->root_mass;
    for (int j = 0; j < n; ++j) {
        const int par = params->parent[j];
        if (par >= j || par < -1 || !(params->mass[j] >= 0.0)) { delete h; return WCQP_E_INVALID; }
        d.parent[j] = par;
        d.level[j] = par < 0 ? 1 : d.level[par] + 1;
        if (d.level[j] > d.max_level) d.max_level = d.level[j];
        std::memcpy(d.R0[j], params->R0[j], sizeof(d.R0[j]));
        std::memcpy(d.p0[j], params->p0[j], sizeof(d.p0[j]));
        double nrm = 0.0;
        for (int k = 0; k < 3; ++k) nrm += params->axis[j][k] * params->axis[j][k];
        if (!(nrm > 0.0)) { delete h; return WCQP_E_INVALID; }
        for (int k = 0; k < 3; ++k) d.axis[j][k] = params->axis[j][k] / std::sqrt(nrm);
        d.mass[j] = params->mass[j]; d.total_mass += params->mass[j];
        std::memcpy(d.com[j], params->com[j], sizeof(d.com[j]));
        d.desc_mask[j] = 1u << j;
    }
    if (!(d.total_mass > 0.0)) { delete h; return WCQP_E_INVALID; }
    for (int j = n - 1; j >= 0; --j) if (d.parent[j] >= 0) d.desc_mask[d.parent[j]] |= d.desc_mask[j];
    std::memcpy(d.root_com, params->root_com, sizeof(d.root_com));
    for (int f = 0; f < 3; ++f) {
        const int jf = params->frame_joint[f];
        if (jf < 0 || jf >= n) { delete h; return WCQP_E_INVALID; }
        d.frame_joint[f] = jf;
        std::memcpy(d.frame_R[f], params->frame_R[f], sizeof(d.frame_R[f]));
        std::memcpy(d.frame_p[f], params->frame_p[f], sizeof(d.frame_p[f]));
        unsigned m = 0;
        for (int k = jf; k >= 0; k = d.parent[k]) m |= 1u << k;
        d.path_mask[f] = m;
    }
    *out = h;
    return WCQP_OK;
}

int wcqp_kin_destroy(wcqp_kin_t h) {
    if (!h) return WCQP_E_INVALID;
    if (h->d_model) (void)hipFree(h->d_model);
    h->scratch.release();
    delete h;
    return WCQP_OK;
}

int wcqp_kin_jacobians_device(wcqp_kin_t h, int32_t batch, const double* base, const double* q,
                              double* J_left, double* J_right, double* J_neck, double* J_com, double* state, void* stream) {
    if (!h || batch < 0) return WCQP_E_INVALID;
    if (!base || !q || !J_left || !J_right || !J_neck || !J_com) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    const int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    const unsigned grid = (unsigned)((batch + 1) / 2);
    hipLaunchKernelGGL(kin_jacobians_kernel<false>, dim3(grid), dim3(64), 0, (hipStream_t)stream, h->d_model, batch, base, q,
                       J_left, J_right, J_neck, J_com, state, wcqp_tick::KinTick{});
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

int wcqp_kin_jacobians_host(wcqp_kin_t h, int32_t batch, const double* base, const double* q,
                            double* J_left, double* J_right, double* J_neck, double* J_com, double* state) {
    if (!h || batch < 0) return WCQP_E_INVALID;
    if (!base || !q || !J_left || !J_right || !J_neck || !J_com) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    const size_t B = (size_t)batch, n = (size_t)h->hd.dof, nc = 6 + n;
    const size_t o_base = 0, o_q = o_base + B * 12, o_jl = o_q + B * n, o_jr = o_jl + B * 6 * nc, o_jn = o_jr + B * 6 * nc,
                 o_jc = o_jn + B * 3 * nc, o_st = o_jc + B * 3 * nc, total = o_st + B * kStateLen;
    rc = h->scratch.reserve(total * sizeof(double));
    if (rc != WCQP_OK) return rc;
    double* d = static_cast<double*>(h->scratch.ptr);
    WCQP_HIP_TRY(hipMemcpy(d + o_base, base, B * 12 * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_q, q, B * n * 8, hipMemcpyHostToDevice));
    if (state) WCQP_HIP_TRY(hipMemcpy(d + o_st, state, B * kStateLen * 8, hipMemcpyHostToDevice));
    rc = wcqp_kin_jacobians_device(h, batch, d + o_base, d + o_q, d + o_jl, d + o_jr, d + o_jn, d + o_jc, state ? d + o_st : nullptr, nullptr);
    if (rc != WCQP_OK) return rc;
    WCQP_HIP_TRY(hipDeviceSynchronize());
    WCQP_HIP_TRY(hipMemcpy(J_left, d + o_jl, B * 6 * nc * 8, hipMemcpyDeviceToHost));
    WCQP_HIP_TRY(hipMemcpy(J_right, d + o_jr, B * 6 * nc * 8, hipMemcpyDeviceToHost));
    WCQP_HIP_TRY(hipMemcpy(J_neck, d + o_jn, B * 3 * nc * 8, hipMemcpyDeviceToHost));
    WCQP_HIP_TRY(hipMemcpy(J_com, d + o_jc, B * 3 * nc * 8, hipMemcpyDeviceToHost));
    if (state) WCQP_HIP_TRY(hipMemcpy(state, d + o_st, B * kStateLen * 8, hipMemcpyDeviceToHost));
    return WCQP_OK;
}

}  // extern "C"
