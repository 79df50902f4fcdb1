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
// 6 + j joint j), so every output row is one coalesced 232-byte segment.  The tree is composed by pointer
// jumping through LDS (ceil(log2 depth) rounds in which every joint lane works), subtree first moments are
// differences of a prefix sum over the lanes when the joint numbering is depth-first.
// HBM-bound by its output: 280 B in, 4464 B out per instance.
#include <cmath>
#include <cstring>
#include <new>
#include <vector>
#include "wcqp_internal.h"
#include "tick_device.h"
#include "kin_device.h"

namespace {

using namespace wcqp_kin;
constexpr int kStateLen = WCQP_IK_STATE_LEN;

// inclusive prefix sum over each 32-lane half of the wave: four shifts inside the 16-lane DPP rows, then lane 15 / 47
// goes to every lane of the row above it (row_bcast:15 into rows 1 and 3)
__device__ __forceinline__ double half_scan(double v) {
    v = row_scan(v);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x142, 0xa, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x142, 0xa, 0xf, false);
    return v + __hiloint2double(hi, lo);
}

// LDS per instance (doubles): TW [32][12] every joint's frame (base coordinates) while the tree is composed; afterwards
// the same space holds PS [32][4] prefix sums of {m c, m} over the joints (or MC [33][4] for tables whose subtrees are not
// index ranges); FRB [3][12] attached frames in base coordinates, FR [3][12] the same in world coordinates; SD [24] the
// pose input (base pose, or the desired poses of the two soles in the tick pipeline)
constexpr int OFF_TW = 0, OFF_PS = 0, OFF_MC = 128, OFF_FR = 32 * 12, OFF_FRB = OFF_FR + 36, OFF_SD = OFF_FRB + 36, PER_INST = OFF_SD + 24;
static_assert(OFF_MC + 33 * 4 <= OFF_FR, "the first moments overlay the joint frames");

#ifdef WCQP_KIN_STAMPS
// diagnostic build: s_memtime at the phase boundaries, written through kt.dbg (stand-alone launches only)
#define WCQP_KSTAMP(k) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
                            if (lane == 0 && kt.dbg) reinterpret_cast<unsigned long long*>(kt.dbg)[(size_t)blockIdx.x * 8 + (k)] = t__; } while (0)
#else
#define WCQP_KSTAMP(k) do { } while (0)
#endif

// small model facts as kernel arguments: no load in front of the first address computation
struct KinShape { int dof, n_rounds, dfs_contig; };

// TICK: the tick pipeline's per-tick call (WalkingModule.cpp:715, 396-410): base pose from the stance foot (tick_device.h: KinTick)
//
// The kernel is a LATENCY chain per wave (3-4 waves per SIMD, each about a thousand instructions between two trips to
// memory), so every global load is issued at the top, in the order of use: q and the joint's constants, the
// pointer-jumping links, the pose input (one element per lane, handed round through LDS), the link and frame constants.
#ifndef WCQP_KIN_WAVES
#define WCQP_KIN_WAVES 3                       // waves per SIMD the register budget is set for (4: the tick variant spills)
#endif
// COMPACT (tick pipeline only): instead of the four dense Jacobians the kernel writes the tick-internal hand-off of
// tick_device.h - per joint one record [C lin3 | X] with X = the joint's column of the ONE frame Jacobian whose path it is
// on (every other entry of that column is a structural zero), and the three vectors p_frame - p_base the base blocks
// [I -S(p); 0 I] are made of: 1.4 KB per robot instead of 4.4 KB, on the store side here and on the load side of the IK.
template <bool TICK, bool COMPACT>
__global__ __launch_bounds__(64, WCQP_KIN_WAVES)
void kin_jacobians_kernel(const KinDev* __restrict__ md, KinShape shp, int batch,
                          const double* __restrict__ base, const double* __restrict__ q,
                          double* __restrict__ JL, double* __restrict__ JR, double* __restrict__ JN, double* __restrict__ JC,
                          double* state, wcqp_tick::KinTick kt)
{
    __shared__ __attribute__((aligned(16))) double smem[2][PER_INST];
    const int lane = threadIdx.x, half = lane >> 5, i = lane & 31;
    const long inst_raw = (long)blockIdx.x * 2 + half;
    const bool live = inst_raw < batch;
    const long inst = live ? inst_raw : (long)batch - 1;
    double* S = smem[half];
    const int dof = shp.dof;
    const int j = i - 6;                                   // joint of this lane's column
    const bool is_joint = j >= 0 && j < dof;
    const int jc = is_joint ? j : 0;

    WCQP_KSTAMP(0);
    // ---------------- every global load of the kernel ------------------------------------------------------
    const double qj = q[inst * dof + jc];
    double R0[9], pa[3], ax[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R0[k] = md->R0[jc][k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ax[k] = md->axis[jc][k]; pa[k] = md->p0[jc][k]; }
    int up[kMaxRounds];
#pragma unroll
    for (int r = 0; r < kMaxRounds; ++r) up[r] = md->up[r][jc];
    double pose_in = 0.0;
    int side = 0;
    if constexpr (TICK) {
        if (i < 24) pose_in = state[inst * kStateLen + 24 + i];                  // desired poses of the two soles: p (3), R (9) each
        const int t_now = kt.tick2[kt.phase];
        side = ((t_now + kt.phase0[inst]) % (2 * kt.step_ticks)) / kt.step_ticks;     // 0: left is the stance foot
    } else {
        if (i < 12) pose_in = base[inst * 12 + i];
    }
    double cj[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) cj[k] = md->com[jc][k];
    const double mj = md->mass[jc];
    const int sub_end = md->sub_end[jc];
    const int fi = i < 3 ? i : 0;
    const int jfi = md->frame_joint[fi];
    double fR[9], fp[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) fR[k] = md->frame_R[fi][k];
#pragma unroll
    for (int k = 0; k < 3; ++k) fp[k] = md->frame_p[fi][k];

    unsigned on_path = 0u;                                 // bit f: this lane's joint is on the path root -> frame f
    unsigned pm[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) { pm[f] = md->path_mask[f]; on_path |= ((pm[f] >> jc) & 1u) << f; }
    if (!is_joint) on_path = 0u;
    double rootc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rootc[k] = md->root_com[k];
    const double root_mass = md->root_mass;

    if (i < 24) S[OFF_SD + i] = pose_in;
    WCQP_KSTAMP(1);
    // own joint: local frame (R0 * Rot(axis, q), p0) relative to the parent's   (Rodrigues)
    double Ra[9];
    joint_rotation(R0, ax, qj, Ra);
    WCQP_KSTAMP(2);
    // the tree, in base coordinates, by POINTER JUMPING: after round r the lane's frame is relative to its 2^(r+1)-th
    // ancestor (every joint lane works in every round; a level-by-level walk runs the same code once per tree level
    // for the three or four lanes of that level)
    double* Tm = S + OFF_TW + jc * 12;
#pragma unroll
    for (int r = 0; r < kMaxRounds; ++r) {
        if (r >= shp.n_rounds) break;
        if (is_joint) {
#pragma unroll
            for (int k = 0; k < 8; k += 2) *reinterpret_cast<double2*>(Tm + k) = make_double2(Ra[k], Ra[k + 1]);
            *reinterpret_cast<double2*>(Tm + 8) = make_double2(Ra[8], pa[0]);
            *reinterpret_cast<double2*>(Tm + 10) = make_double2(pa[1], pa[2]);
        }
        wcqp::wave_lds_fence();
        const int u = up[r];
        if (is_joint && u >= 0) {
            const double* T = S + OFF_TW + u * 12;
            double Rp[9], pp[3], Rn[9], pn[3];
#pragma unroll
            for (int k = 0; k < 9; ++k) Rp[k] = T[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) pp[k] = T[9 + k];
            frame_mul(Rp, pp, Ra, pa, Rn, pn);
#pragma unroll
            for (int k = 0; k < 9; ++k) Ra[k] = Rn[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) pa[k] = pn[k];
        }
        wcqp::wave_lds_fence();               // every lane has read: the frames may be overwritten
    }
    if (is_joint) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) *reinterpret_cast<double2*>(Tm + k) = make_double2(Ra[k], Ra[k + 1]);
        *reinterpret_cast<double2*>(Tm + 8) = make_double2(Ra[8], pa[0]);
        *reinterpret_cast<double2*>(Tm + 10) = make_double2(pa[1], pa[2]);
    }
    wcqp::wave_lds_fence();
    // attached frames (lanes 0..2), base coordinates
    WCQP_KSTAMP(3);
    double Rf[9], pf[3];
    {
        const double* T = S + OFF_TW + jfi * 12;
        double Rj[9], pj[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rj[k] = T[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pj[k] = T[9 + k];
        frame_mul(Rj, pj, fR, fp, Rf, pf);
        if (i < 3) {
            double* F = S + OFF_FRB + i * 12;
#pragma unroll
            for (int k = 0; k < 9; ++k) F[k] = Rf[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) F[9 + k] = pf[k];
        }
    }
    wcqp::wave_lds_fence();
    // base pose
    double pb[3], Rb[9];
    if constexpr (TICK) {
        // anchor foot: world_T_base = world_T_sole,desired * (base_T_sole)^-1
        const double* sd = S + OFF_SD + side * 12;       // desired pose of the anchor sole: p (3), R (9)
        const double* Fs = S + OFF_FRB + side * 12;      // its frame in base coordinates: R (9), p (3)
        double Rd[9], Rs[9], ps[3], d[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) { Rd[k] = sd[3 + k]; Rs[k] = Fs[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) ps[k] = Fs[9 + k];
        // Rb = Rd Rs'
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Rb[3 * r + c] = Rd[3 * r] * Rs[3 * c] + Rd[3 * r + 1] * Rs[3 * c + 1] + Rd[3 * r + 2] * Rs[3 * c + 2];
        mat3_vec(Rb, ps, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) pb[k] = sd[k] - d[k];
    } else {
        const double* b = S + OFF_SD;
#pragma unroll
        for (int k = 0; k < 3; ++k) pb[k] = b[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rb[k] = b[3 + k];
    }
    // this lane's joint frame in world coordinates
    double Rw[9], pw[3];
    frame_mul(Rb, pb, Ra, pa, Rw, pw);
    // attached frames, world
    if (i < 3) {
        double Rg[9], pg[3];
        frame_mul(Rb, pb, Rf, pf, Rg, pg);
        double* F = S + OFF_FR + i * 12;
#pragma unroll
        for (int k = 0; k < 9; ++k) F[k] = Rg[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) F[9 + k] = pg[k];
    }
    // joint axis in world (a rotation about the axis leaves it unchanged: R_w * axis)
    double aw[3];
    mat3_vec(Rw, ax, aw);
    WCQP_KSTAMP(4);
    wcqp::wave_lds_fence();                   // FR is complete
    // ---------------- Jacobian columns of the three frames: stored NOW, so that the stores drain under the moments and the
    // CoM columns instead of all 22 store instructions leaving at the very end of the wave
    int ckind = 0;
    double* crec = nullptr;                    // COMPACT: this lane's joint record
    if constexpr (COMPACT) {
        double* jb = kt.jcomp + inst * kt.cstride;
        if (is_joint) crec = jb + wcqp_tick::compact_offset(pm[0], pm[1], pm[2], jc, ckind);
        if (live && is_joint && ckind != 0) {
            const double* F = S + OFF_FR + (ckind - 1) * 12;
            const double d[3] = {F[9] - pw[0], F[10] - pw[1], F[11] - pw[2]};
            double lin[3];
            cross3(aw, d, lin);
            if (ckind == 3) {                  // neck: the IK keeps the angular rows (setNeckJacobian)
                crec[3] = aw[0];
                *reinterpret_cast<double2*>(crec + 4) = make_double2(aw[1], aw[2]);
            } else {
                crec[3] = lin[0];
                *reinterpret_cast<double2*>(crec + 4) = make_double2(lin[1], lin[2]);
                *reinterpret_cast<double2*>(crec + 6) = make_double2(aw[0], aw[1]);
                *reinterpret_cast<double2*>(crec + 8) = make_double2(aw[2], 0.0);
            }
        }
        if (live && i < 2) {                   // p_left - p_base, p_right - p_base
            const double* F = S + OFF_FR + i * 12;
            double* dd = jb + kt.coff_d + 3 * i;
            dd[0] = F[9] - pb[0]; dd[1] = F[10] - pb[1]; dd[2] = F[11] - pb[2];
        }
    } else
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
            } else if ((on_path >> f) & 1u) {
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
    }
    // link first moment {m c, m} of the own joint's link, and of the root link
    double e4[4] = {0.0, 0.0, 0.0, 0.0};
    if (is_joint) {
        double cl[3];
        mat3_vec(Rw, cj, cl);
        const double m = mj;
#pragma unroll
        for (int k = 0; k < 3; ++k) e4[k] = m * (pw[k] + cl[k]);
        e4[3] = m;
    }
    double root4[4];
    {
        double cr[3];
        mat3_vec(Rb, rootc, cr);
#pragma unroll
        for (int k = 0; k < 3; ++k) root4[k] = root_mass * (pb[k] + cr[k]);
        root4[3] = root_mass;
    }
    wcqp::wave_lds_fence();                   // TW is dead (anchor and attached frames have been read): PS / MC overlay it
    // subtree first moment and mass of this lane's joint; total first moment and mass
    double ms = 0.0, mcs[3] = {0.0, 0.0, 0.0}, tot[4];
    if (shp.dfs_contig) {
        // the subtree of joint j is the index range j .. sub_end[j]: differences of prefix sums over the lanes
        double ps4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ps4[k] = half_scan(e4[k]);
        double* P = S + OFF_PS + i * 4;
        *reinterpret_cast<double2*>(P) = make_double2(ps4[0], ps4[1]);
        *reinterpret_cast<double2*>(P + 2) = make_double2(ps4[2], ps4[3]);
        wcqp::wave_lds_fence();
        const double* Pe = S + OFF_PS + (6 + sub_end) * 4;
        const double* Pb = S + OFF_PS + (i > 0 ? i - 1 : 0) * 4;
        const double* Pt = S + OFF_PS + 31 * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[k] = Pt[k] + root4[k];
        if (is_joint) {
            mcs[0] = Pe[0] - Pb[0]; mcs[1] = Pe[1] - Pb[1]; mcs[2] = Pe[2] - Pb[2]; ms = Pe[3] - Pb[3];
        }
    } else {
        if (is_joint) {
            double* e = S + OFF_MC + j * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = e4[k];
        }
        wcqp::wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[k] = root4[k];
        for (int k = 0; k < dof; ++k) {
            const double* e = S + OFF_MC + k * 4;
            tot[0] += e[0]; tot[1] += e[1]; tot[2] += e[2]; tot[3] += e[3];
        }
        if (is_joint) {
            unsigned dm = md->desc_mask[jc];
            while (dm) {
                const int k = __ffs(dm) - 1;
                dm &= dm - 1;
                const double* e = S + OFF_MC + k * 4;
                mcs[0] += e[0]; mcs[1] += e[1]; mcs[2] += e[2]; ms += e[3];
            }
        }
    }
    const double iM = 1.0 / tot[3];
    double ctot[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ctot[k] = tot[k] * iM;

    WCQP_KSTAMP(5);
    // ---------------- CoM Jacobian columns (the frames' went out before the moments) ---------------
    if constexpr (COMPACT) {
        if (live && is_joint) {
            const double d[3] = {(mcs[0] - ms * pw[0]) * iM, (mcs[1] - ms * pw[1]) * iM, (mcs[2] - ms * pw[2]) * iM};
            double lin[3];
            cross3(aw, d, lin);
            *reinterpret_cast<double2*>(crec) = make_double2(lin[0], lin[1]);
            crec[2] = lin[2];
            if (ckind == 0) crec[3] = 0.0;
        }
        if (live && i == 2) {                  // p_com - p_base
            double* dd = kt.jcomp + inst * kt.cstride + kt.coff_d + 6;
            dd[0] = ctot[0] - pb[0]; dd[1] = ctot[1] - pb[1]; dd[2] = ctot[2] - pb[2]; dd[3] = 0.0;
        }
    } else
    if (live && i < 6 + dof) {
        const int ncol = 6 + dof;
        double e[3] = {0.0, 0.0, 0.0};
        if (i < 6) e[i % 3] = 1.0;
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
    WCQP_KSTAMP(6);
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
// Layout of the compact kinematics -> IK hand-off for this tree (tick_device.h): the joints on the paths of the left sole,
// the right sole and the neck, doubles per robot, offset of the three p_frame - p_base vectors.  false when a joint is on
// more than one of the three paths (a record holds ONE frame's column): the tick then hands over dense Jacobians.
bool kin_compact_layout(wcqp_kin_t h, unsigned masks[3], int* stride, int* off_d) {
    if (!h) return false;
    const KinDev& d = h->hd;
    for (int f = 0; f < 3; ++f) masks[f] = d.path_mask[f];
    if ((masks[0] & masks[1]) || (masks[0] & masks[2]) || (masks[1] & masks[2])) return false;
    int kind = 0;
    const int last = wcqp_tick::compact_offset(masks[0], masks[1], masks[2], d.dof - 1, kind);
    const int end = last + (kind == 0 ? 4 : (kind == 3 ? 6 : 10));
    *off_d = end;
    // the IK reads five 16-byte pieces from a record's start whatever its length, and five from the vectors: keep both inside the block
    const int need = (last + 10 > end + 10) ? last + 10 : end + 10;
    *stride = (need + 1) & ~1;
    return true;
}
int kin_enqueue_tick(wcqp_kin_t h, int batch, const wcqp_tick::KinTick& kt, const double* q,
                     double* J_left, double* J_right, double* J_neck, double* J_com, double* state, hipStream_t stream) {
    if (!h || !h->d_model || batch < 1 || !state || !q) return WCQP_E_INVALID;
    if (!kt.jcomp && (!J_left || !J_right || !J_neck || !J_com)) return WCQP_E_INVALID;
    if (kt.jcomp && (kt.cstride < 1 || kt.coff_d < 0)) return WCQP_E_INVALID;
    if (!kt.tick2 || !kt.phase0 || kt.step_ticks < 1) return WCQP_E_INVALID;
    const unsigned grid = (unsigned)((batch + 1) / 2);
    if (kt.jcomp)
        hipLaunchKernelGGL((kin_jacobians_kernel<true, true>), dim3(grid), dim3(64), 0, stream, h->d_model,
                           KinShape{h->hd.dof, h->hd.n_rounds, h->hd.dfs_contig}, batch, nullptr, q,
                           J_left, J_right, J_neck, J_com, state, kt);
    else
        hipLaunchKernelGGL((kin_jacobians_kernel<true, false>), dim3(grid), dim3(64), 0, stream, h->d_model,
                           KinShape{h->hd.dof, h->hd.n_rounds, h->hd.dfs_contig}, batch, nullptr, q,
                           J_left, J_right, J_neck, J_com, state, kt);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}
// The model as the tick kernel's fused kinematics phase reads it (tick_device.h: TickDev::kin_tab and the int tables): false when
// the tree does not qualify (subtrees that are not index ranges, more than 3 pointer-jumping rounds, a joint on two frame paths)
bool kin_fused_tables(wcqp_kin_t h, std::vector<double>& tab, int* n_rounds) {
    if (!h) return false;
    const KinDev& d = h->hd;
    unsigned m[3]; int stride = 0, off_d = 0;
    if (!kin_compact_layout(h, m, &stride, &off_d)) return false;
    if (!d.dfs_contig || d.n_rounds > 3 || d.dof != wcqp_tick::kDof) return false;
    tab.assign(wcqp_tick::kKinTabSize, 0.0);
    for (int j = 0; j < d.dof; ++j) {
        double* r = &tab[(size_t)j * wcqp_tick::kKinTabJoint];
        for (int k = 0; k < 9; ++k) r[k] = d.R0[j][k];
        for (int k = 0; k < 3; ++k) { r[9 + k] = d.p0[j][k]; r[12 + k] = d.axis[j][k]; r[15 + k] = d.com[j][k]; }
        r[18] = d.mass[j];
        const int ints[4] = {d.up[0][j], d.up[1][j], d.up[2][j], d.sub_end[j]};
        std::memcpy(r + wcqp_tick::kKinTabInts, ints, sizeof(ints));
    }
    for (int f = 0; f < 3; ++f) {
        double* r = &tab[wcqp_tick::kKinTabFrames + (size_t)f * 12];
        for (int k = 0; k < 9; ++k) r[k] = d.frame_R[f][k];
        for (int k = 0; k < 3; ++k) r[9 + k] = d.frame_p[f][k];
    }
    for (int k = 0; k < 3; ++k) tab[wcqp_tick::kKinTabRoot + k] = d.root_com[k];
    tab[wcqp_tick::kKinTabRoot + 3] = d.root_mass;
    const int fj[4] = {d.frame_joint[0], d.frame_joint[1], d.frame_joint[2], 0};
    std::memcpy(&tab[wcqp_tick::kKinTabRoot + 4], fj, sizeof(fj));
    *n_rounds = d.n_rounds;
    return true;
}
}  // namespace wcqp

#ifdef WCQP_KIN_STAMPS
static double* g_kin_dbg = nullptr;
extern "C" void wcqp_kin_set_debug(void* p) { g_kin_dbg = static_cast<double*>(p); }
#endif

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
    d.dof = n; d.root_mass = params->root_mass; d.total_mass = params->root_mass;
    int level[kMaxDof], max_level = 0;
    for (int r = 0; r < kMaxRounds; ++r) for (int j = 0; j < kMaxDof; ++j) d.up[r][j] = -1;
    for (int j = 0; j < n; ++j) {
        const int par = params->parent[j];
        if (par >= j || par < -1 || !(params->mass[j] >= 0.0)) { delete h; return WCQP_E_INVALID; }
        d.up[0][j] = par;
        level[j] = par < 0 ? 1 : level[par] + 1;
        if (level[j] > max_level) max_level = level[j];
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
    for (int j = n - 1; j >= 0; --j) if (d.up[0][j] >= 0) d.desc_mask[d.up[0][j]] |= d.desc_mask[j];
    // pointer-jumping links: a joint at level L is complete after ceil(log2 L) rounds
    d.n_rounds = 0;
    while ((1 << d.n_rounds) < max_level) ++d.n_rounds;
    for (int r = 0; r + 1 < kMaxRounds; ++r)
        for (int j = 0; j < n; ++j) d.up[r + 1][j] = d.up[r][j] < 0 ? -1 : d.up[r][d.up[r][j]];
    // are the subtrees index ranges (a depth-first numbering, as in the shipped joint lists)?
    d.dfs_contig = 1;
    for (int j = 0; j < n; ++j) {
        const unsigned m = d.desc_mask[j] >> j;               // bit 0 = j itself
        int len = 0;
        while ((m >> len) & 1u) ++len;
        if ((m >> len) != 0u) d.dfs_contig = 0;
        d.sub_end[j] = j + len - 1;
    }
    std::memcpy(d.root_com, params->root_com, sizeof(d.root_com));
    for (int f = 0; f < 3; ++f) {
        const int jf = params->frame_joint[f];
        if (jf < 0 || jf >= n) { delete h; return WCQP_E_INVALID; }
        d.frame_joint[f] = jf;
        std::memcpy(d.frame_R[f], params->frame_R[f], sizeof(d.frame_R[f]));
        std::memcpy(d.frame_p[f], params->frame_p[f], sizeof(d.frame_p[f]));
        unsigned m = 0;
        for (int k = jf; k >= 0; k = d.up[0][k]) m |= 1u << k;
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
    wcqp_tick::KinTick kt{};
#ifdef WCQP_KIN_STAMPS
    kt.dbg = g_kin_dbg;
#endif
    hipLaunchKernelGGL((kin_jacobians_kernel<false, false>), dim3(grid), dim3(64), 0, (hipStream_t)stream, h->d_model,
                       KinShape{h->hd.dof, h->hd.n_rounds, h->hd.dfs_contig}, batch, base, q,
                       J_left, J_right, J_neck, J_com, state, kt);
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
