// Device-side pieces of the tick pipeline shared by tick.hip (stand-alone glue / post kernels) and
// ik3.hip (the same work fused into the IK kernel's prologue / epilogue).  Internal, not ABI.
// Reference call order reproduced (citations relative to /root/reference/modules/Walking_module):
//   src/WalkingModule.cpp:578-597   StableDCMModel::integrateModel        -> tick_glue_axis (consumer)
//   src/WalkingModule.cpp:657-695   WalkingZMPController + desired CoM    -> tick_glue_axis
//   src/WalkingModule.cpp:741-744   velocity integration                  -> tick_post_joint
//   src/WalkingModule.cpp:816       advanceReferenceSignals               -> tick counter in HBM
#pragma once
#include "wcqp_internal.h"
#include "mpc_device.h"
#include "hull_device.h"

namespace wcqp_tick {

constexpr int kDof = 23;
constexpr int kStateLen = WCQP_IK_STATE_LEN;

struct TickDev {
    // per-instance constants
    wcqp::GPtr<const double> ref_traj; wcqp::GPtr<const double> hull_tab_A; wcqp::GPtr<const double> hull_tab_b; wcqp::GPtr<const int> hull_tab_nc;
    wcqp::GPtr<const int> phase0; wcqp::GPtr<const double> swing_twist;
    // per-instance state
    wcqp::GPtr<double> dcm, com, zmp_meas, u_prev, u0, c_ref, v_ref, p_star, v_star_prev, v_ref_prev;
    wcqp::GPtr<double> q_des, dq_prev, dq, state;
    wcqp::GPtr<int> sel, mpc_status, ik_status;     // sel: contact pair of the CURRENT tick (0 left, 1 right, 2 both)
    wcqp::GPtr<long long> mpc_fail, ik_fail;
    wcqp::GPtr<int> tick2;         // ticks completed, kept TWICE: the kernels of a tick read tick2[phase], the last of them writes
    int phase;          // tick2[1 - phase] = tick + 1 and the next tick runs with the other phase (a host-side
                        // parity: nobody reads the word that is being written, so one kernel may do both)
    wcqp::GPtr<double> u0_log, dq_log;
    // scalars
    int batch, first, traj_len, log_ticks, step_ticks, ds_ticks;
    double omega, a, b, dT, k_com, k_zmp, noise, com_height;
    unsigned long long seed;
    // MPC fused into the IK kernel (one launch per tick): its condensed constants and hull-row tables
    wcqp_mpc::MpcDeviceConsts mpc;
    int horizon, hull_sets;
    // per-tick kinematics: the IK's ACTUAL CoM (pose block 66..68) is the forward kinematics' at the desired joint
    // state (WalkingModule.cpp:715, 373-376), not the plant's, and the desired height is the
    // instance's initial one
    int kin_mode;
    wcqp::GPtr<const double> com_h0;     // [B]
    // IK hot start (qpOASES SQProblem::hotstart, WM/src/WalkingQPInverseKinematics_qpOASES.cpp:316-318): the previous
    // tick's active bounds (the kernel's own active_lower / active_upper words, 8 B per robot) are tried first
    int hot_start;
    wcqp::GPtr<long long> hot_try, hot_hit;     // [B] ticks on which a previous active set was tried / accepted
    // ---- the SKEWED tick of the base-eliminated fused kernel (ik4.hip): the launch of tick t carries IK(t) and MPC(t+1).
    // The MPC chain MPC(t) -> ZMP-CoM law -> LIPM plant -> MPC(t+1) does not depend on the IK, so the MPC of the NEXT tick
    // runs in the shadow of this tick's Jacobian loads; its outputs reach IK(t+1) through a hand-off record.
    int skew;
    wcqp::GPtr<double> mst;        // [B][2][8] state of the MPC chain per axis: c_ref, v_ref_prev, com, u_prev (the previous command), p_star,
                        //           v_star_prev, dcm, measured ZMP (the internal plant: = the previous command; external feedback:
                        //           the caller's) - one 64-byte record per axis, loaded as four 16-byte pieces
    wcqp::GPtr<double> hand;       // [2][B][kHandLen] MPC(t) -> IK(t) at parity t & 1: p_star xy, v_star xy, com(t) xy, dcm(t) xy, mpc_ok, spare,
                        //           measured ZMP(t) xy (= the previous command), u0(t) xy   (com / dcm / ZMP: the plant at the START of
                        //           tick t; the last six entries feed the logger rows only)
    wcqp::GPtr<double> live_A; wcqp::GPtr<double> live_b; wcqp::GPtr<int> live_nc; wcqp::GPtr<int> sel_built;     // one live hull row set per robot ([B][8][2], [B][8], [B])
                        //           and the contact pair it holds: copied from the robot's three-set table on a contact change, so
                        //           that the rows of a tick are loaded from a fixed address, without waiting for the contact pair
    // ---- compact kinematics -> IK hand-off (tick-internal): per robot one record per joint, [C lin3 | X] with X = the
    // joint's column of the ONE frame Jacobian it is on the path of (left foot 6, right foot 6, neck angular 3), and the
    // three vectors p_frame - p_base that make up the base blocks.  cmask*: joints on the path of the left sole / right sole / neck.
    int compact;
    wcqp::GPtr<const double> jcomp; unsigned cmaskL, cmaskR, cmaskN; int cstride, coff_d;
    // ---- kinematics FUSED into the tick kernel (ik4.hip, JSRC = 2): no kinematics launch, no Jacobian hand-off through memory
    // at all - the wave that solves a robot's IK first evaluates its forward kinematics and Jacobian columns, 16 lanes per
    // robot, two joints per lane - and the kernel can then walk through many ticks per launch, like the constant-Jacobian form.
    // kin_tab: the model as one table of doubles (staged into LDS once per launch): [dof][22] = R0 9 | p0 3 | axis 3 | com 3 |
    // mass | four ints: the joint's pointer-jumping links of rounds 0..2, the last joint of its subtree | pad; then [3][12]
    // attached frames R 9 | p 3; then root_com 3, root_mass, three ints: the joints the frames are attached to.
    int kin_fused, kin_rounds;
    wcqp::GPtr<const double> kin_tab;
    // ---- logger rows (wcqp_tick_params.logger_ticks): the 53 values WalkingModule hands its logger per tick
    // (WM/src/WalkingModule.cpp:800-810, column names :1231-1250), kept for the first logger_ticks ticks
    wcqp::GPtr<double> log_rows;   // [logger_ticks][B][kLoggerCols]
    int logger_ticks;
    double inv_ss;      // 1 / (step_ticks - ds_ticks): the swing phase of a tick as a product (the tick kernel carries the gait cycle index
                        // of its robots from tick to tick instead of dividing its way to it: ik4.hip)
    wcqp::GPtr<unsigned long long> stamps;     // diagnostic builds (-DWCQP_TICK_STAMPS): [workgroups][16] s_memtime at the phase boundaries; else NULL
    // ---- external feedback (wcqp_tick_params.plant = EXTERNAL): the MEASURED joint positions the IK regularises towards
    // (WalkingModule.cpp:373: setRobotState gets the measured joints while the kinematics run at the desired ones, Appendix B-18);
    // NULL with the internal plant (measured = desired).  Measured DCM / CoM / ZMP go straight into the chain's state records.
    wcqp::GPtr<const double> q_meas;           // [B][dof], written by wcqp_tick_set_feedback_device
};
constexpr int kHandLen = 14;
constexpr int kLoggerCols = 53;
constexpr int kKinTabJoint = 22, kKinTabInts = 19, kKinTabFrames = kKinTabJoint * kDof, kKinTabRoot = kKinTabFrames + 36, kKinTabSize = kKinTabRoot + 6;
constexpr int kGainsLdsStages = 56;        // fused kinematics: the MPC's gain blocks Gr ((N + 1) x 2 x 2, N <= 55) sit in LDS beside the model (56: what 8 workgroups per CU leave)

// offset (doubles) of joint c's record inside a robot's compact Jacobian block, and the frame the joint belongs to
// (0 none, 1 left sole, 2 right sole, 3 neck): records are 4 (CoM column only), 6 (+ neck angular) or 10 (+ a foot's 6) long
__host__ __device__ inline int compact_offset(unsigned mL, unsigned mR, unsigned mN, int c, int& kind) {
    const unsigned below = (1u << c) - 1u;
    kind = ((mL >> c) & 1u) ? 1 : (((mR >> c) & 1u) ? 2 : (((mN >> c) & 1u) ? 3 : 0));
#if defined(__HIP_DEVICE_COMPILE__)
    return 4 * c + 6 * __popc((mL | mR) & below) + 2 * __popc(mN & below);
#else
    return 4 * c + 6 * __builtin_popcount((mL | mR) & below) + 2 * __builtin_popcount(mN & below);
#endif
}

// Per-tick kinematics (wcqp_tick_params::use_kinematics): what the kinematics kernel needs beyond the model when it
// runs inside the tick.  The joints are the integrated q_des; the floating base is ANCHORED at the stance foot of the
// current step, the way the reference does it (WalkingFK::evaluateWorldToBaseTransformation with the fixed foot,
// WM/src/WalkingForwardKinematics.cpp:160-256, called at WM/src/WalkingModule.cpp:560-576): world_T_base =
// world_T_sole,desired * (base_T_sole(q))^-1, so the stance sole sits exactly on its planned pose and the base moves
// as the stance leg's joints do.  (The support-polygon rows of the three contact pairs are built from the DESIRED foot
// poses when those are uploaded - hull.hip - and selected per tick by the MPC: WalkingController::setConvexHullConstraint,
// ...PredictiveController.cpp:364-435, switches rows only when the pair changes.)
struct KinTick {
    const int* tick2; int phase;      // tick index (TickDev::tick2)
    const int* phase0; int step_ticks;   // anchor foot = stance foot of the step: ((t + phase0) % (2 step_ticks)) / step_ticks
    // compact hand-off (TickDev::jcomp): the kernel writes the per-joint records instead of the four dense Jacobians
    double* jcomp; int cstride, coff_d;
    double* dbg;              // diagnostic builds (-DWCQP_KIN_STAMPS): where the phase stamps go
};

}  // namespace wcqp_tick
namespace wcqp {
// the kinematics kernel in tick mode (kin.hip)
int kin_enqueue_tick(wcqp_kin_t h, int batch, const wcqp_tick::KinTick& kt, const double* q,
                     double* J_left, double* J_right, double* J_neck, double* J_com, double* state, hipStream_t stream);
}  // namespace wcqp
namespace wcqp_tick {

#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
// uniform in [-1, 1): identical integer arithmetic to oracle/tick_spec.py::disturbance
__device__ __forceinline__ double disturbance(unsigned long long seed, unsigned long long inst, int tick, int axis) {
    const unsigned long long base = mix64(inst * 0x9E3779B97F4A7C15ull + seed);
    const unsigned long long h = mix64(base + ((unsigned long long)(2 * tick + axis) + 1ull) * 0x94D049BB133111EBull);
    return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
// the same with the instance's share of the hash at hand (it does not depend on the tick: the tick kernel computes it once per launch)
__device__ __forceinline__ unsigned long long disturbance_base(unsigned long long seed, unsigned long long inst) { return mix64(inst * 0x9E3779B97F4A7C15ull + seed); }
__device__ __forceinline__ double disturbance_from(unsigned long long base, int tick, int axis) {
    const unsigned long long h = mix64(base + ((unsigned long long)(2 * tick + axis) + 1ull) * 0x94D049BB133111EBull);
    return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

__device__ __forceinline__ int contact_code(int t, int phase0, int step_ticks, int ds_ticks) {
    const int cyc = (t + phase0) % (2 * step_ticks);
    const int s = cyc % step_ticks, side = cyc / step_ticks;
    return s < ds_ticks ? 2 : side;                 // 0 = left only, 1 = right only, 2 = both
}

// One horizontal axis of instance i at tick t: reference LIPM integrator, ZMP-CoM law, desired CoM
// for the IK state block (returned: the caller stores it to HBM or LDS), synthetic plant.  `mpc_ok`: this tick's MPC ended usable.
__device__ __forceinline__ void tick_glue_axis(const TickDev& d, int i, int t, int ax, bool mpc_ok, double u0_ax,
                                               double& s_com, double& s_pstar, double& s_vel) {
    // StableDCMModel::integrateModel (StableDCMModel.cpp:63-90), Tustin integrator; it precedes the
    // MPC in the reference (WalkingModule.cpp:578-597) but only the ZMP-CoM law below consumes it
    const double r = d.ref_traj[((size_t)i * d.traj_len + t) * 2 + ax];
    const double vr = -d.omega * (d.c_ref[2 * i + ax] - r);
    const double c_ref = d.c_ref[2 * i + ax] + 0.5 * d.dT * (vr + d.v_ref_prev[2 * i + ax]);
    d.c_ref[2 * i + ax] = c_ref;
    d.v_ref_prev[2 * i + ax] = vr;
    d.v_ref[2 * i + ax] = vr;
    const double u = mpc_ok ? u0_ax : d.u_prev[2 * i + ax];     // hold the last command on failure
    // WalkingZMPController::evaluateControl (WalkingZMPController.cpp:146-173)
    const double com = d.com[2 * i + ax];
    const double v = d.k_com * (c_ref - com) - d.k_zmp * (u - d.zmp_meas[2 * i + ax]) + vr;
    const double p_star = d.p_star[2 * i + ax] + 0.5 * d.dT * (v + d.v_star_prev[2 * i + ax]);
    d.p_star[2 * i + ax] = p_star;
    d.v_star_prev[2 * i + ax] = v;
    // desired CoM for the IK (WalkingModule.cpp:686-695)
    s_com = com; s_pstar = p_star; s_vel = v;      // -> state[66 + ax], [69 + ax], [72 + ax]
    // synthetic plant: LIPM with a bounded disturbance
    const double xi = d.dcm[2 * i + ax];
    d.com[2 * i + ax] = com + d.dT * (-d.omega * (com - xi));
    d.dcm[2 * i + ax] = d.a * xi + d.b * u + d.noise * disturbance(d.seed, (unsigned long long)(d.first + i), t, ax);
    d.zmp_meas[2 * i + ax] = u;
    d.u_prev[2 * i + ax] = u;
    if (t < d.log_ticks) d.u0_log[((size_t)t * d.batch + i) * 2 + ax] = u;
}
// velocity profile of the swing foot over its single-support phase x in [0, 1): f(x) = 6 sqrt(3) x (1 - x)(1 - 2x), peak 1,
// zero integral - the foot leaves its planted pose by at most 0.325 * amplitude * T_ss and is back at touch-down
// (kinematics mode only: with real kinematics a twist held constant for the whole swing drags the foot out of the
// leg's reach; the constant-Jacobian mode keeps the round-1 constant twist)
__device__ __forceinline__ double swing_profile(const TickDev& d, int i, int t) {
    if (!d.kin_mode) return 1.0;
    const int sidx = ((t + d.phase0[i]) % (2 * d.step_ticks)) % d.step_ticks;
    const int ss = d.step_ticks - d.ds_ticks;
    if (sidx < d.ds_ticks || ss < 1) return 0.0;
    const double x = (double)(sidx - d.ds_ticks) / (double)ss;
    return 10.392304845413264 * x * (1.0 - x) * (1.0 - 2.0 * x);
}
// component k of the two desired foot twists: a foot in contact keeps a zero twist
__device__ __forceinline__ void tick_glue_twist(const TickDev& d, int i, int code, int k, int t, double& tw_left, double& tw_right) {
    const double tw = d.swing_twist[(size_t)i * 6 + k] * swing_profile(d, i, t);
    tw_left = (code == 0 || code == 2) ? 0.0 : tw;      // -> state[75 + k]
    tw_right = (code == 1 || code == 2) ? 0.0 : tw;     // -> state[81 + k]
}
__device__ __forceinline__ void tick_glue_height(const TickDev& d, int i, double* s) {
    if (d.kin_mode) { s[71] = d.com_h0[i]; s[74] = 0.0; }
    else { s[68] = d.com_height; s[71] = d.com_height; s[74] = 0.0; }
}
// A robot whose IK failed once is STOPPED: updateModule returns false on an unsolved QP-IK and the module closes
// (WalkingModule.cpp:414-416, 723-739).  Here it keeps dq = 0 and every further tick counts as failed; ik_fail > 0 is the
// flag (written only by tick_post_instance, and only on ticks whose joint update is zero anyway: no ordering is needed
// between the lanes that read it and the lane that bumps it).
__device__ __forceinline__ bool tick_robot_stopped(const TickDev& d, int i) { return d.ik_fail[i] > 0; }
// joint jj of instance i after the IK of tick t: q <- Integrator(dq) (WalkingModule.cpp:741-744)
__device__ __forceinline__ void tick_post_joint(const TickDev& d, int i, int t, int jj, bool ik_ok, double dq) {
    const size_t g = (size_t)i * kDof + jj;
    const double v = (ik_ok && !tick_robot_stopped(d, i)) ? dq : 0.0;
    d.q_des[g] += 0.5 * d.dT * (v + d.dq_prev[g]);
    d.dq_prev[g] = v;
    if (t < d.log_ticks) d.dq_log[(size_t)t * d.batch * kDof + g] = v;
}
// once per instance after the IK of tick t: failure count, contact pair of the NEXT tick
// (WalkingController::setConvexHullConstraint switches rows only when the pair changes,
// …PredictiveController.cpp:369-374 — here the MPC kernel simply reads the row set this index selects)
__device__ __forceinline__ void tick_post_instance(const TickDev& d, int i, int t, bool ik_ok) {
    if (!ik_ok || tick_robot_stopped(d, i)) d.ik_fail[i] += 1;
    d.sel[i] = contact_code(t + 1, d.phase0[i], d.step_ticks, d.ds_ticks);
}

// ---------------------------------------------------------------------------------------------------------------
// Skewed tick: the MPC chain of tick t on the 16 lanes of a DPP row (j = lane in the row; lanes 0 / 1 also own one
// horizontal axis each).  tick_mpc_issue puts every load of the chain in flight; tick_mpc_finish does the arithmetic:
// contact pair -> (on a change) hull rows, condensed MPC, LIPM reference, ZMP-CoM law, synthetic plant, hand-off record.
struct TickMpcRegs {
    wcqp_mpc::MpcLoads L;
    double2 s01, s23, s45, s67;     // lanes 0 / 1: this axis' state record (TickDev::mst)
    double2 ha; double hb; int nc;  // live hull row of lane j < 8, row count
    int phase0, built;
};
// GAINS_LDS: the gain blocks Gr are read from a copy in LDS at the time of use (gr_lds), not loaded into registers here
template <bool GAINS_LDS = false>
__device__ __forceinline__ void tick_mpc_issue(const TickDev& d, int j, long inst, int t, TickMpcRegs& R) {
    // (32-bit addressing, wcqp::at32: wcqp_tick_create refuses batches whose trajectories do not fit 4 GB)
    const unsigned iu = (unsigned)inst;
    const double2* ref = reinterpret_cast<const double2*>(d.ref_traj.get());
    const unsigned w0 = iu * (unsigned)d.traj_len + (unsigned)t;
    R.phase0 = *wcqp::at32(d.phase0.get(), iu * 4u);
    R.built = *wcqp::at32(d.sel_built.get(), iu * 4u);
    const double2* sp = reinterpret_cast<const double2*>(wcqp::at32(d.mst.get(), (iu * 2u + (unsigned)(j & 1)) * 64u));
    R.s01 = sp[0]; R.s23 = sp[1]; R.s45 = sp[2]; R.s67 = sp[3];
    if constexpr (GAINS_LDS) wcqp_mpc::mpc_window_loads_ref_only(d.mpc, j, ref, w0, d.horizon + 1, R.L);
    else wcqp_mpc::mpc_window_loads(d.mpc, j, ref, w0, d.horizon + 1, R.L);
    R.nc = *wcqp::at32(d.live_nc.get(), iu * 4u);
    const unsigned jr = (unsigned)(j & 7);
    R.ha = *wcqp::at32(reinterpret_cast<const double2*>(d.live_A.get()), (iu * WCQP_HULL_ROWS + jr) * 16u);
    R.hb = *wcqp::at32(d.live_b.get(), (iu * WCQP_HULL_ROWS + jr) * 8u);
}
// tick_mpc_finish = tick_mpc_partial (this lane's share of u0_unc from the loaded window: the window registers die here) +
// tick_mpc_finish_from (everything else, from the per-axis state records, the hull row and the partial sums)
template <bool GAINS_LDS = false>
__device__ __forceinline__ void tick_mpc_partial(const TickDev& d, int j, long inst, int t, const TickMpcRegs& R, const double* gr_lds, double& ux, double& uy) {
    if constexpr (GAINS_LDS) wcqp_mpc::mpc_row_partial_lds(d.mpc, j, R.L, gr_lds, ux, uy);
    else wcqp_mpc::mpc_row_partial(d.mpc, j, R.L, ux, uy);
    if (d.horizon >= 4 * wcqp_mpc::kLanesPerInstance)
        wcqp_mpc::mpc_row_extra_passes(d.mpc, j, reinterpret_cast<const double2*>(d.ref_traj.get()) + inst * d.traj_len + t, d.horizon + 1, ux, uy);
}
// r0: stage 0 of the window (the reference DCM of tick t; meaningful on lane 0)
// EXT (external feedback: only tick_mpc_prime_kernel<true> - with such a handle every tick's MPC runs there): the measured ZMP is the
// caller's (record entry 7), not the previous command; the kernels of the internal plant keep entry 7 out of their registers
template <bool EXT = false>
__device__ __forceinline__ void tick_mpc_finish_from(const TickDev& d, int j, long inst, bool live, int t, TickMpcRegs& R, double2 r0, double ux, double uy,
                                                     double (*s_hull)[4], int code_known = -1, const unsigned long long* noise_base = nullptr) {
    // contact pair of tick t; the live row set follows it (WalkingController::setConvexHullConstraint switches rows only
    // when the pair changes, ...PredictiveController.cpp:369-374)
    const int code = code_known >= 0 ? code_known : contact_code(t, R.phase0, d.step_ticks, d.ds_ticks);
    const bool stale = code != R.built;
    if (__ballot(stale) != 0ull) {
        if (stale && live) {
            double* lA = d.live_A + inst * (2 * WCQP_HULL_ROWS);
            double* lb = d.live_b + inst * WCQP_HULL_ROWS;
            const long hset = inst * 3 + code;
            if (j < WCQP_HULL_ROWS) {
                reinterpret_cast<double2*>(lA)[j] = reinterpret_cast<const double2*>(d.hull_tab_A.get())[hset * WCQP_HULL_ROWS + j];
                lb[j] = d.hull_tab_b[hset * WCQP_HULL_ROWS + j];
            }
            if (j == 0) { d.live_nc[inst] = d.hull_tab_nc[hset]; d.sel_built[inst] = code; }
        }
        // the rows were written by other lanes of this wave: make them visible, then read them back
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        R.nc = d.live_nc[inst];
        const int jr = j & 7;
        R.ha = reinterpret_cast<const double2*>(d.live_A.get())[inst * WCQP_HULL_ROWS + jr];
        R.hb = d.live_b[inst * WCQP_HULL_ROWS + jr];
        // ... and wait for them HERE, on the rare path: left to the compiler the wait lands behind the join, where it must assume
        // that the row registers were loaded last - behind the IK's Jacobian loads - and drains those on every tick
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    }
    // ---- the condensed MPC (mpc_device.h): x0 = measured DCM, u_prev = previous output (MPCSolver.cpp:244-245)
    {
        // lane 0 holds the x axis' record, lane 1 the y axis': the y entries come over by DPP (row_shl:1)
        const double dcm_y = wcqp_mpc::row_move<0x101>(R.s67.x), up_y = wcqp_mpc::row_move<0x101>(R.s23.y);
        if (j == 0) wcqp_mpc::mpc_row_add_state(d.mpc, make_double2(R.s67.x, dcm_y), make_double2(R.s23.y, up_y), ux, uy);
    }
    double u0x, u0y, margin;
    int st;
    unsigned act;
    wcqp_mpc::mpc_row_finish(d.mpc, j, ux, uy, R.nc, j < WCQP_HULL_ROWS ? R.ha.x : 0.0, j < WCQP_HULL_ROWS ? R.ha.y : 0.0,
                             j < WCQP_HULL_ROWS ? R.hb : 0.0, s_hull, u0x, u0y, st, act, margin);
    const bool mpc_ok = st == WCQP_STATUS_SOLVED || st == WCQP_STATUS_OUTSIDE_HULL;
    const double r_y = wcqp_mpc::row_move<0x111>(r0.y);          // lane 1 <- lane 0 (row_shr:1)
    if (j < 2 && live) {
        const int ax = j;
        const double c_ref0 = R.s01.x, v_ref_prev = R.s01.y, com = R.s23.x, u_prev = R.s23.y;
        const double p_star0 = R.s45.x, v_star_prev = R.s45.y, xi = R.s67.x;
        const double zmp_meas = EXT ? R.s67.y : u_prev;
        // StableDCMModel::integrateModel (StableDCMModel.cpp:63-90), Tustin integrator
        const double rr = ax == 0 ? r0.x : r_y;        // reference DCM of tick t: stage 0 of the window (lane 0 holds it)
        const double vr = -d.omega * (c_ref0 - rr);
        const double c_ref = c_ref0 + 0.5 * d.dT * (vr + v_ref_prev);
        const double u = mpc_ok ? (ax == 0 ? u0x : u0y) : u_prev;        // hold the last command on failure
        // WalkingZMPController::evaluateControl (WalkingZMPController.cpp:146-173); the measured ZMP: with the internal plant the
        // previous command (the same number as u_prev), with external feedback what the caller measured
        const double v = d.k_com * (c_ref - com) - d.k_zmp * (u - zmp_meas) + vr;
        const double p_star = p_star0 + 0.5 * d.dT * (v + v_star_prev);
        // synthetic plant: LIPM with a bounded disturbance
        const double com1 = com + d.dT * (-d.omega * (com - xi));
        const double w_ = noise_base ? disturbance_from(*noise_base, t, ax) : disturbance(d.seed, (unsigned long long)(d.first + inst), t, ax);
        const double xi1 = d.a * xi + d.b * u + d.noise * w_;
        double2* sp = reinterpret_cast<double2*>(d.mst.get() + (inst * 2 + ax) * 8);
        sp[0] = make_double2(c_ref, vr); sp[1] = make_double2(com1, u); sp[2] = make_double2(p_star, v); sp[3] = make_double2(xi1, EXT ? u : 0.0);
        // hand-off to the IK of tick t (desired CoM position / velocity, WalkingModule.cpp:686-695) + the plant state at the start of tick t
        double* hd = d.hand + ((size_t)(t & 1) * d.batch + inst) * kHandLen;
        hd[ax] = p_star; hd[2 + ax] = v; hd[4 + ax] = com; hd[6 + ax] = xi;
        if (ax == 0) hd[8] = mpc_ok ? 1.0 : 0.0;
        hd[10 + ax] = zmp_meas; hd[12 + ax] = u;
        if (t < d.log_ticks) d.u0_log[((size_t)t * d.batch + inst) * 2 + ax] = u;
    }
}
template <bool GAINS_LDS = false, bool EXT = false>
__device__ __forceinline__ void tick_mpc_finish(const TickDev& d, int j, long inst, bool live, int t, TickMpcRegs& R, double (*s_hull)[4],
                                                const double* gr_lds = nullptr, int code_known = -1, const unsigned long long* noise_base = nullptr) {
    double ux, uy;
    tick_mpc_partial<GAINS_LDS>(d, j, inst, t, R, gr_lds, ux, uy);
    tick_mpc_finish_from<EXT>(d, j, inst, live, t, R, R.L.r[0], ux, uy, s_hull, code_known, noise_base);
}
// the same two with the gait cycle index cyc = (t + phase0) % (2 step_ticks) at hand (the tick kernel carries it from tick to tick:
// integer divisions by run-time values are ~35 instructions each, and a tick had four of them, on every lane)
__device__ __forceinline__ int contact_code_cyc(int cyc, int step_ticks, int ds_ticks) {
    const bool second = cyc >= step_ticks;
    const int s = second ? cyc - step_ticks : cyc;
    return s < ds_ticks ? 2 : (second ? 1 : 0);
}
__device__ __forceinline__ double swing_profile_cyc(const TickDev& d, int cyc) {
    if (!d.kin_mode) return 1.0;
    const int sidx = cyc >= d.step_ticks ? cyc - d.step_ticks : cyc;
    if (sidx < d.ds_ticks || d.step_ticks - d.ds_ticks < 1) return 0.0;
    const double x = (double)(sidx - d.ds_ticks) * d.inv_ss;
    return 10.392304845413264 * x * (1.0 - x) * (1.0 - 2.0 * x);
}
// swing_profile with the instance's phase offset already in a register
__device__ __forceinline__ double swing_profile_at(const TickDev& d, int phase0, int t) {
    if (!d.kin_mode) return 1.0;
    const int sidx = ((t + phase0) % (2 * d.step_ticks)) % d.step_ticks;
    const int ss = d.step_ticks - d.ds_ticks;
    if (sidx < d.ds_ticks || ss < 1) return 0.0;
    const double x = (double)(sidx - d.ds_ticks) / (double)ss;
    return 10.392304845413264 * x * (1.0 - x) * (1.0 - 2.0 * x);
}
#endif

}  // namespace wcqp_tick
