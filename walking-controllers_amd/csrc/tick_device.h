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

namespace wcqp_tick {

constexpr int kDof = 23;
constexpr int kStateLen = WCQP_IK_STATE_LEN;

struct TickDev {
    // per-instance constants
    const double* ref_traj; const double* hull_tab_A; const double* hull_tab_b; const int* hull_tab_nc;
    const int* phase0; const double* swing_twist;
    // per-instance state
    double *dcm, *com, *zmp_meas, *u_prev, *u0, *c_ref, *v_ref, *p_star, *v_star_prev, *v_ref_prev;
    double *q_des, *dq_prev, *dq, *state;
    int *sel, *mpc_status, *ik_status;     // sel: contact pair of the CURRENT tick (0 left, 1 right, 2 both)
    long long *mpc_fail, *ik_fail;
    int* tick2;         // ticks completed, kept TWICE: the kernels of a tick read tick2[phase], the last of them writes
    int phase;          // tick2[1 - phase] = tick + 1 and the next tick runs with the other phase (a host-side
                        // parity: nobody reads the word that is being written, so one kernel may do both)
    double *u0_log, *dq_log;
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
    const double* com_h0;     // [B]
    // IK hot start (qpOASES SQProblem::hotstart, WM/src/WalkingQPInverseKinematics_qpOASES.cpp:316-318): the previous
    // tick's active bounds (the kernel's own active_lower / active_upper words, 8 B per robot) are tried first
    int hot_start;
    long long *hot_try, *hot_hit;     // [B] ticks on which a previous active set was tried / accepted
};

// Per-tick kinematics (wcqp_tick_params::use_kinematics): what the kinematics kernel needs beyond the model when it
// runs inside the tick.  The joints are the integrated q_des; the floating base is ANCHORED at the stance foot of the
// current step, the way the reference does it (WalkingFK::evaluateWorldToBaseTransformation with the fixed foot,
// WM/src/WalkingForwardKinematics.cpp:160-256, called at WM/src/WalkingModule.cpp:560-576): world_T_base =
// world_T_sole,desired * (base_T_sole(q))^-1, so the stance sole sits exactly on its planned pose and the base moves
// as the stance leg's joints do.  The support-polygon rows are rebuilt from the DESIRED foot poses whenever the
// contact pair changes (WalkingController::setConvexHullConstraint, ...PredictiveController.cpp:364-435).
struct KinTick {
    const int* tick2; int phase;      // tick index (TickDev::tick2)
    const int* phase0; int step_ticks;   // anchor foot = stance foot of the step: ((t + phase0) % (2 step_ticks)) / step_ticks
    const int* sel;           // [B] contact pair of this tick (0 left, 1 right, 2 both)
    int* sel_built;           // [B] contact pair the current hull rows were built for (-1: none yet)
    double* hull_A; double* hull_b; int* hull_nc;     // [B][8][2], [B][8], [B]
    double rect[8];           // foot rectangle corners (x, y) x 4 in the foot frame
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
#endif

}  // namespace wcqp_tick
