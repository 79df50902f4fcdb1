// Jacobian QP-IK, fourth kernel: the six base unknowns eliminated in CLOSED FORM through the left-foot rows, the
// remaining 23-variable QP solved in range space.  16 lanes per instance (one DPP row), 4 instances per wave64.
//
// Same QP, inputs, outputs and reference citations as ik.hip / ik2.hip / ik3.hip
// (WM/src/WalkingQPInverseKinematics_qpOASES.cpp:135-401, _osqp.cpp:135-454).  What is different:
//
// The Jacobians the reference hands to the IK are iDynTree free-floating frame Jacobians in MIXED representation
// (WM/src/WalkingForwardKinematics.cpp:33, 436-454), whose base blocks are
//     J_left = [I B_L; 0 I | J_Lq]   J_right = [I B_R; 0 I | J_Rq]   J_com = [I B_C | J_Cq]   J_neck(angular) = [0 I | J_Nq]
// (B = -S(p_frame - p_base)).  The six left-foot rows then give the base velocity in closed form,
//     v_base = X_L^-1 (b_L - J_Lq x),   X_L^-1 = [I -B_L; 0 I],   x = joint velocities,
// and what is left is
//     min 1/2 x' Lam x + gq' x + 1/2 |Nt x - t|^2     s.t.   A x = b',   lo <= x <= hi
//     A  = [J_Rq - X_R X_L^-1 J_Lq ; J_Cq - X_C X_L^-1 J_Lq]   (9 x 23)      row operations, local to a column
//     Nt = L' (J_Nq - J_Lq,ang),  W_neck = L L',   Lam = diag(joint weights) > 0
// With Lam > 0 the Hessian needs no null-space basis: in the scaled variable x~ = Lam^1/2 x and C = [Nt; A] Lam^-1/2
// (12 x 23)
//     M y = -(C g~ + [t; b']),   M = C C' + diag(I3, 0)   (12 x 12, SPD),      x~ = -(g~ + C' y)
// i.e. ONE 12-pivot sweep without pivot search replaces the 15 searched pivots of the column-pivoted elimination
// and the 14-pivot sweep of the reduced Hessian in ik3.hip; M is one fp64 MFMA tile per instance with the SAME
// register as A and B operand.  The projected inverse Hessian P = I - C' M^-1 C feeds the same Goldfarb-Idnani
// dual active set as the other kernels (first bound straight-line, up to 4 bounds replicated in registers, bigger
// working sets slot-per-lane), so active sets stay bit-identical.
//
// Every instance checks its own base blocks for the pattern (exact 1.0 / 0.0 entries); one that does not have it
// comes back WCQP_STATUS_STRUCTURE, and the dispatcher (ik.hip) runs the general kernel (ik3.hip, list mode) over the
// flagged instances unless the handle was created for MIXED Jacobians only (include/wcqp.h: jacobian_structure).
//
// Lane j of an instance's 16 owns joint column j (slot 0) and, in slot 1: joint column 16 + j (j < 7), the
// right-hand-side column [b_L; b_R; b_C; e_neck] (j = 7), base column j - 8 (j = 8 .. 13: pattern check and B blocks).
// C^T (k-major) stays in LDS from the Gram product to the end: the columns are re-read where they are needed (x~,
// a bound's column tau_p) instead of occupying 48 VGPRs through the sweep, and the Gram tile reaches its row lanes
// through v_permlane32_swap / v_permlane16_swap (a 4 x 4 block transpose across the wave's four DPP rows) instead of
// an LDS tile.  LDS 440 doubles per instance (13.75 KB per block: 11 blocks per CU; two waves per SIMD need 8).
// Template parameter TICK: the tick pipeline's glue / post steps fused in (tick_device.h).
#include <cmath>
#include <limits>
#include <type_traits>
#include "ik_common.h"
#include "tick_device.h"
#include "kin_device.h"

namespace {

using namespace wcqp_ik;

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NR = 12;                // rows of C = [Nt (3); A_right (6); A_com (3)]
constexpr int NROWS_IN = 18;          // J_left 6, J_right 6, J_com 3, J_neck 3
constexpr int KMAX = kDof - 9;        // 14: largest working set (n - m_eq)
constexpr int LDC = 14;               // leading dimension of C^T (k-major) and of the Gram tile: b128 row accesses of 16
                                      // lanes land on 16 distinct groups of 4 banks (28 j mod 64)

// ---- LDS layout per instance (doubles) ---------------------------------------------------------
// region A, three lives:
//   set-up
constexpr int OFF_ST = 0;             // [112] state + q
constexpr int OFF_BV = 112;           // [18]  task rhs b (15) and neck target e (3)
constexpr int OFF_DB = 130;           // [18]  B_R - B_L, B_C - B_L (row-major 3x3 each)
//   Gram .. end
constexpr int OFF_CT = 0;             // [24][LDC] (+2): C^T k-major, row 12 = g~, row 13 = 0; k = 23: zero column
constexpr int OFF_PB = 0;             // [12][18] foot-error partial products (epilogue)
constexpr int A_SIZE = 340;
// region B
constexpr int OFF_COL = A_SIZE;       // [2][16] sweep columns
constexpr int OFF_YV = A_SIZE + 32;   // [16] y
constexpr int OFF_DV = A_SIZE + 48;   // [16] d = [t; b'] by row
//   active set (region B is dead after x~)
constexpr int OFF_YPV = A_SIZE;       // [16] M^-1 C v
constexpr int OFF_RV = A_SIZE + 16;   // [16] dual step per slot      (working sets of more than KS bounds)
constexpr int OFF_CV = A_SIZE + 32;   // [16]
constexpr int OFF_ROWB = A_SIZE + 48; // [16] row of the leaving slot
//   variable of slot a (working sets of more than KS bounds): an int in entry 13 of row a of C^T (the zero row of the Gram
//   tile, dead once the MFMAs have read it)
constexpr int PER_INST = 440;         // = 24 mod 32: the four instances of a wave sit 16 banks apart; 14080 B per workgroup (408 would do for the
                                      // solve; the fused kinematics' joint frames want the rest: K_* below)
// ---- fused kinematics (JSRC = 2): scratch of the kinematics phase, over the same region (everything of it is dead before the pose block
// and C^T are written).  Joint frames [23][K_FS]: 12 doubles at a stride of 14 - the b128 accesses of 16 lanes then land on 8 distinct
// groups of 4 banks (2 passes, the minimum for a b128) instead of 4 groups (4 passes) at a stride of 12: PMC showed 24 % of the tick
// kernel's LDS-active cycles as bank conflicts.  The two spare doubles behind frames 16..21 hold the anchor pose (k_sd).
constexpr int K_FS = 14, K_TW = 0, K_FRB = 322, K_FR = 358;          // joint frames, attached frames in base / world coordinates [3][12] each
constexpr int K_MS = 394, K_MH = 410;                               // stashes: the MPC chain's per-axis records [2][8], its hull rows [8][3]
__host__ __device__ constexpr int k_sd(int m) { return K_TW + (16 + (m >> 1)) * K_FS + 12 + (m & 1); }      // anchor pose [12] / CoM [3]
static_assert(kDof * K_FS <= K_FRB && K_FR + 36 <= K_MS && K_MH + 24 <= PER_INST && k_sd(11) < kDof * K_FS && k_sd(0) >= 32 * 4, "kinematics scratch fits; the prefix sums [32][4] stay clear of the anchor pose");
static_assert(OFF_CT + 24 * LDC + 2 <= A_SIZE && OFF_DB + 18 <= A_SIZE && OFF_PB + 12 * 18 <= A_SIZE, "LDS overlays");
static_assert(OFF_ROWB + 16 <= PER_INST && OFF_DV + 16 <= PER_INST && (PER_INST % 32 == 24 || PER_INST % 32 == 8), "instances 16 banks apart");
static_assert(KMAX <= 24, "one slot index per C^T row");
static_assert(PER_INST * 8 * 4 * 11 <= 160 * 1024, "11 blocks per CU");

#ifndef WCQP_IK4_WAVES
#define WCQP_IK4_WAVES 2
#endif
// Register cap of ik_plan_kernel (the IK-only plan).  gfx90a and later have ONE register file of 512 entries per SIMD lane for VGPRs and
// AGPRs together, allocated in blocks of 8, and hipcc's amdgpu_num_vgpr counts HALF of it: amdgpu_num_vgpr(108) caps the kernel at 216
// VGPRs.  Two waves of 216 leave 80 registers of a SIMD free - room for ONE wave of mpc_plan_kernel (72) beside them.
#ifdef WCQP_IK_PLAN_VGPR_HALF
#define WCQP_IK_PLAN_REGS __attribute__((amdgpu_num_vgpr(WCQP_IK_PLAN_VGPR_HALF)))
#else
#define WCQP_IK_PLAN_REGS
#endif
#ifndef WCQP_IK4_KS
#define WCQP_IK4_KS 4                 // bounds kept replicated in registers (more: the slot-per-lane loop)
#endif

#if defined(WCQP_TICK_KSTAMPS)
// diagnostic build (tools/build_variant.sh kstamps -DWCQP_TICK_KSTAMPS -DWCQP_TICK_STAMPS): the KINEMATICS phase of the fused tick in detail - stamps 0, 12 and 14 as
// below, slots 1..9 are its sub-phases (WCQP_KSTAMP); WCQP_KSTAMPS=1 tools/stamps_tick.py
#define WCQP_STAMP_AT(k) do { if constexpr (TICK) { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
                              if (lane == 0 && td.stamps) td.stamps[(size_t)blk * 16 + (k)] = t__; } } while (0)
#define WCQP_STAMP(k) do { if constexpr ((k) == 0 || (k) == 12 || (k) == 14) WCQP_STAMP_AT(k); } while (0)
#define WCQP_KSTAMP(k) WCQP_STAMP_AT(k)
#elif defined(WCQP_TICK_STAMPS)
// diagnostic build (tools/build_variant.sh tstamps -DWCQP_TICK_STAMPS): s_memtime at the phase boundaries of the TICK kernel's body, per
// workgroup, the last tick of a launch wins (TickDev::stamps; tools/stamps_tick.py)
#define WCQP_STAMP(k) do { if constexpr (TICK) { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
                           if (lane == 0 && td.stamps) td.stamps[(size_t)blk * 16 + (k)] = t__; } } while (0)
#elif defined(WCQP_IK_STAMPS)
#define WCQP_STAMP(k) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
                           if (lane == __ffsll((long long)__ballot(true)) - 1) reinterpret_cast<unsigned long long*>(ferr_out)[(size_t)blockIdx.x * 16 + (k)] = t__; } while (0)
#else
#define WCQP_STAMP(k) do { } while (0)
#endif
#ifndef WCQP_KSTAMP
#define WCQP_KSTAMP(k) do { } while (0)
#endif

// (a, b) -> (rows {a0, a1, b0, b1}, rows {a2, a3, b2, b3}) of the four 16-lane rows (checked on the GPU:
// tools/ubench/permlane_test.hip)
__device__ __forceinline__ void swap32(double& a, double& b) {
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)h[0], (int)l[0]); b = __hiloint2double((int)h[1], (int)l[1]);
}
// (a, b) -> (rows {a0, b0, a2, b2}, rows {a1, b1, a3, b3})
__device__ __forceinline__ void swap16(double& a, double& b) {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)h[0], (int)l[0]); b = __hiloint2double((int)h[1], (int)l[1]);
}

__device__ __forceinline__ void st2(double* p, double a, double b) { *reinterpret_cast<double2*>(p) = make_double2(a, b); }
__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }

// the DCM-MPC of the same robots riding along with their IK (qp_pair_kernel: in workgroups of its own; PAIR: on the IK's lanes)
struct MpcPairArgs {
    wcqp_mpc::MpcDeviceConsts c;
    const double* x0; const double* ref; int ref_len; const double* u_prev;
    const double* hull_A; const double* hull_b; const int* hull_nc;
    double* u0; int* status; unsigned* active; double* margin;
    // qp_plan_kernel's work queues: lane 0 draws the wave's next ticket from this counter (nullptr: none) BEHIND the record's loads -
    // vmcnt retires in order, and a device-scope atomic issued in front of them would hold every load of the record back by its
    // own, longer, round trip - and leaves it here for the bottom of the loop
    unsigned* ticket_from = nullptr; unsigned ticket = 0;
    bool has_mpc = true;       // false: ik_plan_kernel (an IK-only plan): the record has no MPC part
};

// PAIR (a plan of steps, wcqp_qp_plan_*): the wave also solves the DCM-MPC QP of its four robots, its loads issued in front
// of the IK's and its arithmetic running while the Jacobians are in flight (what the tick kernel does with the MPC of the next tick).
// JSRC: where the Jacobians come from - 0 the four dense arrays of the ABI, 1 the compact per-joint records of the tick's
// kinematics kernel (tick_device.h), 2 the kinematics phase of this kernel itself (no hand-off through memory at all)
// LOG (tick kernel, wcqp_tick_params.logger_ticks > 0): also writes the reference's logger row of every robot-tick; a kernel
// of its own, so that the product kernels carry none of it
// EXT (tick kernel, wcqp_tick_params.plant = EXTERNAL): the IK regularises towards the caller's MEASURED joint positions
// (TickDev::q_meas) instead of the desired ones - a kernel of its own: the two registers it holds across the kinematics phase
// cost the fused-kinematics kernel 28 B of scratch, which the product kernel does not pay
template <bool TICK, int JSRC = 0, bool PAIR = false, bool LOG = false, bool EXT = false>
__device__ __forceinline__
void ik4_body(const IkDeviceParams* __restrict__ prm, int batch,
                const double* __restrict__ JL, const double* __restrict__ JR,
                const double* __restrict__ JN, const double* __restrict__ JC,
                const double* qpos, const double* __restrict__ state,
                double* __restrict__ dq_out, int* __restrict__ status_out,
                unsigned* __restrict__ alo_out, unsigned* __restrict__ aup_out,
                double* __restrict__ ferr_out, int* __restrict__ iters_out, const wcqp_tick::TickDev& td, double (*smem)[PER_INST], int blk,
                const int tick_now = 0, const bool do_mpc = true, const double* kmodel = nullptr, const double* gr_lds = nullptr,
                MpcPairArgs* pm = nullptr, double* carry = nullptr, int* gait = nullptr, const unsigned long long* noise_base = nullptr)
{
    static_assert(!(TICK && PAIR), "the tick kernel carries its own MPC chain");
    constexpr bool COMPACT = JSRC == 1;
    constexpr bool KINF = JSRC == 2;
    int lane_id = threadIdx.x;
    // inside the tick kernel's loop over ticks: keeps hipcc from hoisting every per-lane address and constant of the body
    // out of the loop (they would all be live across the whole body: +100 VGPRs and spills)
    if constexpr (TICK || PAIR) __asm__ volatile("" : "+v"(lane_id));
    const int lane = lane_id;
    const int grp = lane >> 4;
    const int j = lane & 15;
    const long inst_raw = (long)blk * 4 + grp;
    const bool live = inst_raw < batch;
    const long inst = live ? inst_raw : (long)batch - 1;
    // 32-bit addressing (wcqp::at32): uniform array base + this lane's BYTE offset - the host entry points refuse batches whose
    // arrays do not fit 4 GB
    using wcqp::at32;
    const unsigned iu = (unsigned)inst, j8 = (unsigned)j * 8u;
    double* S = smem[grp];
    double* st = S + OFF_ST;
    const double inf = std::numeric_limits<double>::infinity();
    const bool var1 = j < kDof - 16;                // slot 1 is joint 16 + j
    const bool rhs1 = j == kDof - 16;               // slot 1 is the right-hand-side column
    const bool base1 = j >= 8 && j < 14;            // slot 1 is base column j - 8
    const int col1 = j + 16;

    unsigned prev_lo = 0u, prev_up = 0u;            // hot start: the previous tick's active bounds of this instance
    bool stopped = false;                           // tick pipeline: the robot's IK failed on an earlier tick (tick_device.h)
    auto load_previous_set = [&]() {
        if (td.hot_start && alo_out && aup_out) { prev_lo = *at32(alo_out, iu * 4u); prev_up = *at32(aup_out, iu * 4u); }
        stopped = wcqp_tick::tick_robot_stopped(td, (int)inst);
        if (stopped) { prev_lo = 0u; prev_up = 0u; }
    };
    if constexpr (TICK && JSRC != 2) load_previous_set();

    WCQP_STAMP(0);
    // ---------------- phase 0: loads ------------------------------------------------------------
    const int v0i = 6 + j, v1i = var1 ? 22 + j : 0;           // index into the per-variable constant tables
    // per-lane constants: up front, in the shadow of the input loads - or (fused kinematics) behind the kinematics phase,
    // in the shadow of the MPC arithmetic: across that phase every register counts
    double sd0, sd1, isd0, isd1, kq0, kq1, qreg0, qreg1;
    auto load_lane_constants = [&]() {
        sd0 = prm->sd[j]; sd1 = prm->sd[col1]; isd0 = prm->isd[j]; isd1 = prm->isd[col1];
        kq0 = prm->kq[v0i]; kq1 = prm->kq[v1i]; qreg0 = prm->qreg[v0i]; qreg1 = prm->qreg[v1i];
    };
    if constexpr (!KINF) load_lane_constants();
    double a0[NROWS_IN], a1[NROWS_IN];     // columns of [J_left; J_right; J_com; J_neck]
    double q0, q1;
    double qm0, qm1;                       // the joint positions the IK's regularisation sees (tick with external feedback: measured ones)
    bool osqp_form;
    double k_pos_foot, k_att_foot, k_pos_com, kap;
    int fast_ok;
    // tick pipeline, SKEWED: this launch carries IK(t) and the MPC chain of tick t + 1 (tick_device.h).  The MPC chain
    // MPC -> ZMP-CoM law -> LIPM plant does not depend on the IK, so the loads of MPC(t+1) are issued FIRST, the IK's loads
    // behind them, and its arithmetic runs while the Jacobians are in flight; IK(t) reads what MPC(t) left in the hand-off
    // record one launch ago (the first launch after an upload is primed by tick_mpc_prime_kernel, tick.hip).
    wcqp_tick::TickMpcRegs mreg;
    double2 p_xs = make_double2(0.0, 0.0), p_up = make_double2(0.0, 0.0);
    if constexpr (PAIR) {
        if (pm->has_mpc) {          // (an IK-only plan has no MPC part: a compile-time constant in either plan kernel)
        wcqp_mpc::mpc_window_loads(pm->c, j, reinterpret_cast<const double2*>(pm->ref), iu * (unsigned)pm->ref_len, pm->ref_len, mreg.L);
        if (j == 0) { p_xs = *at32(reinterpret_cast<const double2*>(pm->x0), iu * 16u); p_up = *at32(reinterpret_cast<const double2*>(pm->u_prev), iu * 16u); }
        mreg.nc = *at32(pm->hull_nc, iu * 4u);
        mreg.ha = make_double2(0.0, 0.0); mreg.hb = 0.0;
        if (j < WCQP_HULL_ROWS) {
            mreg.ha = *at32(reinterpret_cast<const double2*>(pm->hull_A), (iu * WCQP_HULL_ROWS + (unsigned)j) * 16u);
            mreg.hb = *at32(pm->hull_b, iu * (WCQP_HULL_ROWS * 8u) + j8);
        }
        }
    }
    if constexpr (TICK) { if (do_mpc) wcqp_tick::tick_mpc_issue<KINF>(td, j, inst, tick_now + 1, mreg); else mreg.phase0 = td.phase0[inst]; }
    double2 cr0[5], cr1[5], cdv[5];        // COMPACT: the two joint records and the three vectors p_frame - p_base, as loaded
    int ckind0 = 0, ckind1 = 0;
    double m_ux = 0.0, m_uy = 0.0;         // ... the MPC chain's partial sums, reduced early
    double2 m_r0 = make_double2(0.0, 0.0);
    {
        // the state block first: vmcnt retires in order, and the rhs phase only needs the state, so the 36
        // Jacobian loads stay in flight underneath it
        const unsigned so = iu * (unsigned)(kStateLen * 8);
        const double* sp = at32(state, so + j8);                                         // entry m * 16 + j of the block: this lane's offset + an immediate
        const double* sp5 = at32(state, so + 640u + (j < kStateLen - 80 ? j8 : 0u));     // entries 80 .. 86
        double sreg[6];
        if constexpr (!KINF) {
#pragma unroll
            for (int m = 0; m < 5; ++m) sreg[m] = sp[m * 16];
            sreg[5] = sp5[0];
        }
        if constexpr (TICK) {
            // the tick kernel walks through the ticks: this lane's joint positions (and previous velocities) are carried from the
            // post step of one tick to the next in registers - what the post step stores is never re-read inside a launch
            q0 = carry[0]; q1 = carry[1];
        } else {
            q0 = *at32(qpos, iu * (unsigned)(kDof * 8) + j8);
            q1 = *at32(qpos, iu * (unsigned)(kDof * 8) + (var1 ? j8 + 128u : 0u));
        }
        auto load_measured_joints = [&]() {
            if constexpr (EXT) { const double* qm = td.q_meas; qm0 = *at32(qm, iu * (unsigned)(kDof * 8) + j8); qm1 = *at32(qm, iu * (unsigned)(kDof * 8) + (var1 ? j8 + 128u : 0u)); }
        };
        if constexpr (TICK && !KINF) load_measured_joints();
        double g_com = 0.0, g_pstar = 0.0, g_vel = 0.0, g_twl = 0.0, g_twr = 0.0, g_ok = 1.0, g_sw = 0.0, g_h0 = 0.0;
        auto load_handoff = [&]() {
            // hand-off of MPC(t): desired CoM position / velocity (WalkingModule.cpp:686-695), the plant's CoM, did the MPC end usable
            const double* hb_ = at32(td.hand.get(), ((unsigned)(tick_now & 1) * (unsigned)td.batch + iu) * (unsigned)(wcqp_tick::kHandLen * 8));
            const double* hd = at32(hb_, (unsigned)(j & 1) * 8u);
            g_pstar = hd[0]; g_vel = hd[2]; g_com = hd[4]; g_ok = hb_[8];
            g_sw = *at32(td.swing_twist.get(), iu * 48u + (j < 6 ? j8 : 0u));
            g_h0 = td.kin_mode ? *at32(td.com_h0.get(), iu * 8u) : td.com_height;
        };
        if constexpr (TICK && !KINF) load_handoff();
        // the state / q loads above must ISSUE before the 36 column loads (vmcnt retires in order): hipcc otherwise sinks
        // one of them below the Jacobian loads and the state's LDS stores then wait for everything (vmcnt(0))
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KINF) {
            // ================= kinematics phase (wcqp_tick_params.use_kinematics, fused): forward kinematics at the integrated
            // joint state with the base anchored at the stance foot (WalkingFK::evaluateWorldToBaseTransformation,
            // WM/src/WalkingForwardKinematics.cpp:160-256; WM/src/WalkingModule.cpp:715, 396-410) and this lane's two columns of
            // the four MIXED Jacobians, straight into the registers the row operations read.  Same algebra as
            // kin_jacobians_kernel (kin.hip), laid out for the IK's 16 lanes per robot: lane j owns joints j and 16 + j.
            using namespace wcqp_kin;
            // every register counts across this phase: what the MPC chain of tick t + 1 has loaded is reduced to this lane's share
            // of u0_unc now (its loads were issued first: they have landed when the pose block below has) and its per-axis records
            // and hull row wait in LDS; the pose block is re-read behind the kinematics (L2) instead of being held
            if (do_mpc) {
                wcqp_tick::tick_mpc_partial<true>(td, j, inst, tick_now + 1, mreg, gr_lds, m_ux, m_uy);
                m_r0 = mreg.L.r[0];
                if (j < 2) {
                    double* ms = S + K_MS + j * 8;
                    st2(ms, mreg.s01.x, mreg.s01.y); st2(ms + 2, mreg.s23.x, mreg.s23.y); st2(ms + 4, mreg.s45.x, mreg.s45.y); st2(ms + 6, mreg.s67.x, mreg.s67.y);
                }
                if (j < 8) { double* mh = S + K_MH + j * 3; mh[0] = mreg.ha.x; mh[1] = mreg.ha.y; mh[2] = mreg.hb; }
            }
            WCQP_KSTAMP(1);          // MPC loads landed, partial sums stashed
            const int side = *gait >= td.step_ticks ? 1 : 0;          // (gait: this robot's cycle index (tick + phase0) % (2 step_ticks), carried from tick to tick) 0: left is the stance foot
            if (j < 12) S[k_sd(j)] = *at32(state, iu * (unsigned)(kStateLen * 8) + (unsigned)(24 + side * 12) * 8u + j8);                // desired pose of the anchor sole: p (3), R (9)
            const int cs[2] = {j, var1 ? col1 : 0};
            double* TW = S + K_TW;
            int kup[2][3], ksub[2];                 // the joints' pointer-jumping links and subtree ends: from the model table in LDS
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const int* ip = reinterpret_cast<const int*>(kmodel + cs[s_] * wcqp_tick::kKinTabJoint + wcqp_tick::kKinTabInts);
                kup[s_][0] = ip[0]; kup[s_][1] = ip[1]; kup[s_][2] = ip[2]; ksub[s_] = ip[3];
            }
            const int kfj = reinterpret_cast<const int*>(kmodel + wcqp_tick::kKinTabRoot + 4)[j < 3 ? j : 0];
            {
            double Ra[2][9], pa[2][3];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const double* mt = kmodel + cs[s_] * wcqp_tick::kKinTabJoint;
                double R0[9], axl[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) R0[k] = mt[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) { pa[s_][k] = mt[9 + k]; axl[k] = mt[12 + k]; }
                joint_rotation(R0, axl, s_ == 0 ? q0 : q1, Ra[s_]);
            }
            WCQP_KSTAMP(2);          // joint rotations (sin / cos) done
            // the tree in base coordinates by pointer jumping (kin.hip): after round r a frame is relative to its 2^(r+1)-th ancestor
            const int n_rounds = td.kin_rounds;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (r >= n_rounds) break;
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    if (s_ == 0 || var1) {
                        double* Tm = TW + cs[s_] * K_FS;
#pragma unroll
                        for (int k = 0; k < 8; k += 2) st2(Tm + k, Ra[s_][k], Ra[s_][k + 1]);
                        st2(Tm + 8, Ra[s_][8], pa[s_][0]); st2(Tm + 10, pa[s_][1], pa[s_][2]);
                    }
                }
                wcqp::wave_lds_fence();
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    const int u = kup[s_][r];
                    if (u >= 0 && (s_ == 0 || var1)) {
                        const double* T = TW + u * K_FS;
                        double Rp[9], pp[3], Rn[9], pn[3];
#pragma unroll
                        for (int k = 0; k < 9; ++k) Rp[k] = T[k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) pp[k] = T[9 + k];
                        frame_mul(Rp, pp, Ra[s_], pa[s_], Rn, pn);
#pragma unroll
                        for (int k = 0; k < 9; ++k) Ra[s_][k] = Rn[k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) pa[s_][k] = pn[k];
                    }
                }
                wcqp::wave_lds_fence();
            }
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                if (s_ == 0 || var1) {
                    double* Tm = TW + cs[s_] * K_FS;
#pragma unroll
                    for (int k = 0; k < 8; k += 2) st2(Tm + k, Ra[s_][k], Ra[s_][k + 1]);
                    st2(Tm + 8, Ra[s_][8], pa[s_][0]); st2(Tm + 10, pa[s_][1], pa[s_][2]);
                }
            }
            }
            wcqp::wave_lds_fence();
            WCQP_KSTAMP(3);          // pointer jumping done, frames stored
            // attached frames (left sole, right sole, neck) in base coordinates: lanes 0..2
            const int fi = j < 3 ? j : 0;
            double Rf[9], pf[3];
            {
                const double* T = TW + kfj * K_FS;
                const double* ft = kmodel + wcqp_tick::kKinTabFrames + fi * 12;
                double Rj[9], pj[3], fR[9], fp[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) { Rj[k] = T[k]; fR[k] = ft[k]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { pj[k] = T[9 + k]; fp[k] = ft[9 + k]; }
                frame_mul(Rj, pj, fR, fp, Rf, pf);
                if (j < 3) {
                    double* F = S + K_FRB + j * 12;
#pragma unroll
                    for (int k = 0; k < 9; ++k) F[k] = Rf[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) F[9 + k] = pf[k];
                }
            }
            wcqp::wave_lds_fence();
            WCQP_KSTAMP(4);          // attached frames in base coordinates
            // base pose from the anchor foot: world_T_base = world_T_sole,desired * (base_T_sole)^-1
            double pb[3], Rb[9];
            {
                const double* Fs = S + K_FRB + side * 12;
                double Rs[9], ps[3], d3[3], sdp[3], sdR[9];
#pragma unroll
                for (int k = 0; k < 3; ++k) sdp[k] = S[k_sd(k)];
#pragma unroll
                for (int k = 0; k < 9; ++k) sdR[k] = S[k_sd(3 + k)];
#pragma unroll
                for (int k = 0; k < 9; ++k) Rs[k] = Fs[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) ps[k] = Fs[9 + k];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) Rb[3 * r + c] = sdR[3 * r] * Rs[3 * c] + sdR[3 * r + 1] * Rs[3 * c + 1] + sdR[3 * r + 2] * Rs[3 * c + 2];
                mat3_vec(Rb, ps, d3);
#pragma unroll
                for (int k = 0; k < 3; ++k) pb[k] = sdp[k] - d3[k];
            }
            // attached frames in world coordinates
            if (j < 3) {
                double Rg[9], pg[3];
                frame_mul(Rb, pb, Rf, pf, Rg, pg);
                double* F = S + K_FR + j * 12;
#pragma unroll
                for (int k = 0; k < 9; ++k) F[k] = Rg[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) F[9 + k] = pg[k];
            }
            WCQP_KSTAMP(5);          // base pose, attached frames in world coordinates
            // own joints in world coordinates, their axes, link first moments {m c, m}
            double pw[2][3], aw[2][3], e4[2][4];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const double* mt = kmodel + cs[s_] * wcqp_tick::kKinTabJoint;
                double Rw[9], cl[3], Rl[9], pl[3];
                {   // the joint's frame in base coordinates, back from LDS (not held in registers across the frames / base pose above)
                    const double* Tm = TW + cs[s_] * K_FS;
#pragma unroll
                    for (int k = 0; k < 9; ++k) Rl[k] = Tm[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) pl[k] = Tm[9 + k];
                }
                frame_mul(Rb, pb, Rl, pl, Rw, pw[s_]);
                const double axl[3] = {mt[12], mt[13], mt[14]};
                mat3_vec(Rw, axl, aw[s_]);
                const double cj[3] = {mt[15], mt[16], mt[17]};
                const double mj = (s_ == 0 || var1) ? mt[18] : 0.0;
                mat3_vec(Rw, cj, cl);
#pragma unroll
                for (int k = 0; k < 3; ++k) e4[s_][k] = mj * (pw[s_][k] + cl[k]);
                e4[s_][3] = mj;
            }
            wcqp::wave_lds_fence();          // FR is complete; the joint frames are dead: the prefix sums overlay them
            WCQP_KSTAMP(6);          // own joints in world coordinates
            // ---- frame columns: joint c is on the path of at most one of the three frames (compact_offset: kind)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                int kind;
                (void)wcqp_tick::compact_offset(td.cmaskL, td.cmaskR, td.cmaskN, cs[s_], kind);
                if (s_ == 1 && !var1) kind = 0;
                const double* F = S + K_FR + (kind > 0 ? kind - 1 : 0) * 12;
                const double d3[3] = {F[9] - pw[s_][0], F[10] - pw[s_][1], F[11] - pw[s_][2]};
                double lin[3];
                cross3(aw[s_], d3, lin);
                const double mL = kind == 1 ? 1.0 : 0.0, mR = kind == 2 ? 1.0 : 0.0, mN = kind == 3 ? 1.0 : 0.0;
                double (&a)[NROWS_IN] = s_ == 0 ? a0 : a1;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    a[r] = mL * lin[r]; a[3 + r] = mL * aw[s_][r];
                    a[6 + r] = mR * lin[r]; a[9 + r] = mR * aw[s_][r];
                    a[15 + r] = mN * aw[s_][r];
                }
            }
            WCQP_KSTAMP(7);          // frame columns
            // ---- subtree first moments: the joint numbering is depth-first, a subtree is an index range; inclusive prefix sums
            // over joints 0..15 (slot 0, a DPP row scan) and 16.. (slot 1, offset by the row's total)
            double* PS = S + K_TW;               // [32][4]
            {
                double p0s[4], p1s[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { p0s[k] = row_scan(e4[0][k]); p1s[k] = row_scan(e4[1][k]); }
                st2(PS + j * 4, p0s[0], p0s[1]); st2(PS + j * 4 + 2, p0s[2], p0s[3]);
                wcqp::wave_lds_fence();
                const double2 t01 = ld2(PS + 15 * 4), t23 = ld2(PS + 15 * 4 + 2);
                st2(PS + (16 + j) * 4, p1s[0] + t01.x, p1s[1] + t01.y); st2(PS + (16 + j) * 4 + 2, p1s[2] + t23.x, p1s[3] + t23.y);
                wcqp::wave_lds_fence();
            }
            WCQP_KSTAMP(8);          // prefix sums in LDS
            double tot[4], ctot[3];
            {
                const double* rt = kmodel + wcqp_tick::kKinTabRoot;
                const double rootc[3] = {rt[0], rt[1], rt[2]};
                const double root_mass = rt[3];
                double cr[3];
                mat3_vec(Rb, rootc, cr);
                const double* Pt = PS + (kDof - 1) * 4;
#pragma unroll
                for (int k = 0; k < 3; ++k) tot[k] = Pt[k] + root_mass * (pb[k] + cr[k]);
                tot[3] = Pt[3] + root_mass;
            }
            const double iM = 1.0 / tot[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) ctot[k] = tot[k] * iM;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const int c = cs[s_];
                const double* Pe = PS + ksub[s_] * 4;
                const double* Pb = PS + (c > 0 ? c - 1 : 0) * 4;
                const double z = c > 0 ? 1.0 : 0.0;
                const double ms = Pe[3] - z * Pb[3];
                const double d3[3] = {(Pe[0] - z * Pb[0] - ms * pw[s_][0]) * iM, (Pe[1] - z * Pb[1] - ms * pw[s_][1]) * iM, (Pe[2] - z * Pb[2] - ms * pw[s_][2]) * iM};
                double lin[3];
                cross3(aw[s_], d3, lin);
                double (&a)[NROWS_IN] = s_ == 0 ? a0 : a1;
                const double mv = (s_ == 0 || var1) ? 1.0 : 0.0;
#pragma unroll
                for (int r = 0; r < 3; ++r) a[12 + r] = mv * lin[r];
            }
            WCQP_KSTAMP(9);          // CoM columns
            // the vectors the base blocks [I -S(p); 0 I] are made of: p_left - p_base, p_right - p_base, p_com - p_base
            double kdv[9];
            {
                const double* FL = S + K_FR, *FRt = S + K_FR + 12;
#pragma unroll
                for (int k = 0; k < 3; ++k) { kdv[k] = FL[9 + k] - pb[k]; kdv[3 + k] = FRt[9 + k] - pb[k]; kdv[6 + k] = ctot[k] - pb[k]; }
            }
            // (the attached frames in world coordinates stay in LDS - they are the ACTUAL poses the pose block gets below - and the
            // CoM joins them in the anchor pose's stash, which is dead)
            wcqp::wave_lds_fence();
            if (j < 3) S[k_sd(j)] = ctot[j];
            // (pinned: hipcc otherwise hoists these loads - 44 registers of results - to the top of the kinematics phase)
            __builtin_amdgcn_sched_barrier(0);
            // the pose block, the per-lane constants and the hand-off record: on their way under the MPC arithmetic below
#pragma unroll
            for (int m = 0; m < 5; ++m) sreg[m] = sp[m * 16];
            sreg[5] = sp5[0];
            WCQP_STAMP(12);
            load_lane_constants();
            load_handoff();
            load_previous_set();
            load_measured_joints();
            wcqp::wave_lds_fence();              // everything of the kinematics scratch has been read
            if (j >= 11 && j < 14) {
                // B_R - B_L, B_C - B_L for the row operations, column cm: B_f = -S(p_f - p_base), column cm = e_cm x (p_f - p_base) -
                // the products kin_jacobians_kernel forms for the dense base columns
                double* db = S + OFF_DB;
                const int cm = j - 11;
                const double e0 = cm == 0 ? 1.0 : 0.0, e1 = cm == 1 ? 1.0 : 0.0, e2 = cm == 2 ? 1.0 : 0.0;
                double Bc[3][3];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const double d0 = kdv[3 * f], d1 = kdv[3 * f + 1], d2 = kdv[3 * f + 2];
                    Bc[f][0] = e1 * d2 - e2 * d1; Bc[f][1] = e2 * d0 - e0 * d2; Bc[f][2] = e0 * d1 - e1 * d0;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) { db[r * 3 + cm] = Bc[1][r] - Bc[0][r]; db[9 + r * 3 + cm] = Bc[2][r] - Bc[0][r]; }
            }
            if (do_mpc) {                        // the MPC chain's stash back into registers
                if (j < 2) {
                    const double* ms = S + K_MS + j * 8;
                    mreg.s01 = ld2(ms); mreg.s23 = ld2(ms + 2); mreg.s45 = ld2(ms + 4); mreg.s67 = ld2(ms + 6);
                }
                if (j < 8) { const double* mh = S + K_MH + j * 3; mreg.ha.x = mh[0]; mreg.ha.y = mh[1]; mreg.hb = mh[2]; }
            }
        } else if constexpr (COMPACT) {
            // compact kinematics -> IK hand-off (tick_device.h): one record per joint, [C lin3 | X ...], X = the joint's column of
            // the one frame Jacobian it is on the path of; every other entry of the four Jacobians is a structural zero and the
            // base blocks follow from the three vectors p_frame - p_base.  Five 16-byte loads per slot, whatever the record's
            // length (what lies behind a short record is the next one: read and masked off), instead of 18 column loads.
            // Unpacked into the dense columns further down, behind the MPC arithmetic that runs under these loads.
            const double* jb = td.jcomp + inst * td.cstride;
            const int off0 = wcqp_tick::compact_offset(td.cmaskL, td.cmaskR, td.cmaskN, j, ckind0);
            const int off1 = wcqp_tick::compact_offset(td.cmaskL, td.cmaskR, td.cmaskN, var1 ? col1 : 0, ckind1);
#pragma unroll
            for (int m = 0; m < 5; ++m) { cr0[m] = ld2(jb + off0 + 2 * m); cr1[m] = ld2(jb + off1 + 2 * m); }
#pragma unroll
            for (int m = 0; m < 5; ++m) cdv[m] = ld2(jb + td.coff_d + 2 * m);
        } else {
        const int fc0 = 6 + j;
        const int fc1 = var1 ? 22 + j : (base1 ? j - 8 : 0);
        // per array two lane offsets (this lane's two columns of the instance's block), the rows behind them as immediates
        const unsigned o6 = iu * (unsigned)(6 * kNV * 8), o3 = iu * (unsigned)(3 * kNV * 8), c0b = (unsigned)fc0 * 8u, c1b = (unsigned)fc1 * 8u;
        const double* jl0 = at32(JL, o6 + c0b), *jl1 = at32(JL, o6 + c1b);
        const double* jr0 = at32(JR, o6 + c0b), *jr1 = at32(JR, o6 + c1b);
        const double* jc0 = at32(JC, o3 + c0b), *jc1 = at32(JC, o3 + c1b);
        const double* jn0 = at32(JN, o3 + c0b), *jn1 = at32(JN, o3 + c1b);
#pragma unroll
        for (int r = 0; r < 6; ++r) { a0[r] = jl0[r * kNV]; a1[r] = jl1[r * kNV]; }
#pragma unroll
        for (int r = 0; r < 6; ++r) { a0[6 + r] = jr0[r * kNV]; a1[6 + r] = jr1[r * kNV]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { a0[12 + r] = jc0[r * kNV]; a1[12 + r] = jc1[r * kNV]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { a0[15 + r] = jn0[r * kNV]; a1[15 + r] = jn1[r * kNV]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PAIR) {
            if (pm->ticket_from && threadIdx.x == 0) pm->ticket = __hip_atomic_fetch_add(pm->ticket_from, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PAIR) if (pm->has_mpc) {
            // the DCM-MPC QP of the same four robots while the Jacobians are on their way: the operations of mpc_row_solve, in its order
            double ux, uy, u0x, u0y, margin;
            int mst_;
            unsigned mact;
            wcqp_mpc::mpc_row_partial(pm->c, j, mreg.L, ux, uy);
            if (pm->c.N >= 4 * wcqp_mpc::kLanesPerInstance)          // a horizon beyond one 64-stage pass (the shipped N = 200): the rest of the window, loaded on the spot
                wcqp_mpc::mpc_row_extra_passes(pm->c, j, reinterpret_cast<const double2*>(pm->ref) + inst * pm->ref_len, pm->ref_len, ux, uy);
            if (j == 0) wcqp_mpc::mpc_row_add_state(pm->c, p_xs, p_up, ux, uy);
            wcqp_mpc::mpc_row_finish(pm->c, j, ux, uy, mreg.nc, mreg.ha.x, mreg.ha.y, mreg.hb, reinterpret_cast<double (*)[4]>(S + OFF_COL), u0x, u0y, mst_, mact, margin);
            if (j == 0 && live) {
                *at32(reinterpret_cast<double2*>(pm->u0), iu * 16u) = make_double2(u0x, u0y);
                *at32(pm->status, iu * 4u) = mst_;
                if (pm->active) *at32(pm->active, iu * 4u) = mact;
                if (pm->margin) *at32(pm->margin, iu * 8u) = margin;
            }
        }
        if constexpr (TICK) {
            // MPC(t+1), ZMP-CoM law and plant of tick t + 1 for the same four robots, while the Jacobians are on their way
            if (do_mpc) {
                // (hull rows in the MPC stash's place, just read back: the attached frames at 312..347 are still needed)
                const int cyc1 = *gait + 1 == 2 * td.step_ticks ? 0 : *gait + 1;
                const int code1 = wcqp_tick::contact_code_cyc(cyc1, td.step_ticks, td.ds_ticks);
                if constexpr (KINF) wcqp_tick::tick_mpc_finish_from(td, j, inst, live, tick_now + 1, mreg, m_r0, m_ux, m_uy, reinterpret_cast<double (*)[4]>(S + K_MS), code1, noise_base);
                else wcqp_tick::tick_mpc_finish(td, j, inst, live, tick_now + 1, mreg, reinterpret_cast<double (*)[4]>(S + OFF_COL), nullptr, code1, noise_base);
            }
            if (j < 6) {
                const int code = wcqp_tick::contact_code_cyc(*gait, td.step_ticks, td.ds_ticks);
                const double tw = g_sw * wcqp_tick::swing_profile_cyc(td, *gait);
                g_twl = (code == 0 || code == 2) ? 0.0 : tw;
                g_twr = (code == 1 || code == 2) ? 0.0 : tw;
            }
            if (live && j == 0 && g_ok == 0.0) td.mpc_fail[inst] += 1;
            WCQP_STAMP(13);
        }
        // the scalar settings are read only now: in front of the column loads their (cold) scalar-cache misses would sit in
        // the same s_waitcnt as the Jacobian pointers and hold the 36 loads back
        osqp_form = prm->form == WCQP_IK_FORM_OSQP;
        k_pos_foot = prm->k_pos_foot; k_att_foot = prm->k_att_foot; k_pos_com = prm->k_pos_com;
        kap = prm->kappa * (-prm->k_neck);
        fast_ok = prm->fast_ok;
#pragma unroll
        for (int m = 0; m < 5; ++m) st[m * 16 + j] = sreg[m];
        st[80 + j] = sreg[5];        // unconditional (slots 87..95 are spare): a predicated store makes hipcc sink the LOAD into the branch, behind the column loads
        if constexpr (KINF) {
            // the ACTUAL poses the kinematics phase produced (WalkingModule.cpp:396-410) over the pose block's
            wcqp::wave_lds_fence();
            if (j < 12) {
                const double* FL = S + K_FR, *FRt = S + K_FR + 12;
                st[j] = j < 3 ? FL[9 + j] : FL[j - 3];
                st[12 + j] = j < 3 ? FRt[9 + j] : FRt[j - 3];
            }
            if (j < 9) st[48 + j] = S[K_FR + 24 + j];
            if (j < 3) st[66 + j] = S[k_sd(j)];
        }
        if constexpr (TICK) {
            wcqp::wave_lds_fence();
            if (j < 2) { if (!td.kin_mode) st[66 + j] = g_com; st[69 + j] = g_pstar; st[72 + j] = g_vel; }
            if (j < 6) { st[75 + j] = g_twl; st[81 + j] = g_twr; }
            if (j == 0) { if (!td.kin_mode) st[68] = g_h0; st[71] = g_h0; st[74] = 0.0; }      // tick_glue_height
        }
    }
    wcqp::wave_lds_fence();
    if constexpr (LOG) {
        // the logger row of this robot-tick (WM/src/WalkingModule.cpp:800-810; columns :1231-1250): measured / desired DCM, desired
        // DCM velocity, measured / desired ZMP, measured CoM, desired CoM position / velocity, actual and desired foot poses
        // (position + roll-pitch-yaw), foot errors (written with the IK's result below)
        if (tick_now < td.logger_ticks && live) {
            double* row = td.log_rows + ((size_t)tick_now * td.batch + inst) * wcqp_tick::kLoggerCols;
            if (j < 2) {
                const double* hd = td.hand + ((size_t)(tick_now & 1) * td.batch + inst) * wcqp_tick::kHandLen;
                const double r0 = td.ref_traj[((size_t)inst * td.traj_len + tick_now) * 2 + j];
                const double r1 = td.ref_traj[((size_t)inst * td.traj_len + tick_now + 1) * 2 + j];
                row[j] = hd[6 + j]; row[2 + j] = r0; row[4 + j] = (r1 - r0) / td.dT;        // the planner's DCM velocity: finite difference of the reference
                row[6 + j] = hd[10 + j]; row[8 + j] = hd[12 + j];
                row[13 + j] = hd[j]; row[15 + j] = hd[2 + j];
            }
            if (j < 3) {
                row[10 + j] = st[66 + j];
                row[17 + j] = st[j]; row[23 + j] = st[12 + j]; row[29 + j] = st[24 + j]; row[35 + j] = st[36 + j];
            }
            if (j < 4) {
                // iDynTree::Rotation::asRPY (upstream): roll = atan2(R21, R22), pitch = asin(-R20), yaw = atan2(R10, R00)
                const double* R = st + (j == 0 ? 3 : (j == 1 ? 15 : (j == 2 ? 27 : 39)));
                double* o = row + (j == 0 ? 20 : (j == 1 ? 26 : (j == 2 ? 32 : 38)));
                const double s_ = fmin(1.0, fmax(-1.0, -R[6]));
                o[0] = atan2(R[7], R[8]); o[1] = asin(s_); o[2] = atan2(R[3], R[0]);
            }
        }
    }
    if constexpr (COMPACT) {
        auto unpack = [&](const double2 (&r)[5], int kind, double (&a)[NROWS_IN]) {
            const double x[6] = {r[1].y, r[2].x, r[2].y, r[3].x, r[3].y, r[4].x};
            const double mL = kind == 1 ? 1.0 : 0.0, mR = kind == 2 ? 1.0 : 0.0, mN = kind == 3 ? 1.0 : 0.0;
#pragma unroll
            for (int r_ = 0; r_ < 6; ++r_) { a[r_] = mL * x[r_]; a[6 + r_] = mR * x[r_]; }
            a[12] = r[0].x; a[13] = r[0].y; a[14] = r[1].x;
#pragma unroll
            for (int r_ = 0; r_ < 3; ++r_) a[15 + r_] = mN * x[r_];
        };
        unpack(cr0, ckind0, a0);
        unpack(cr1, ckind1, a1);
        if (!var1) {
#pragma unroll
            for (int r_ = 0; r_ < NROWS_IN; ++r_) a1[r_] = 0.0;
        }
    }
#ifdef WCQP_IK4_EXIT_AFTER_LOADS
    {   // diagnostic build: the launch up to the point where every input has landed
        double acc = q0 + q1 + st[j];
#pragma unroll
        for (int r = 0; r < NROWS_IN; ++r) acc += a0[r] + a1[r];
        if (live) dq_out[inst * kDof + j] = acc + sd0 + sd1 + isd0 + isd1 + kq0 + kq1 + qreg0 + qreg1 + (double)fast_ok + k_pos_foot + k_att_foot + k_pos_com + kap + (osqp_form ? 1.0 : 0.0);
        return;
    }
#endif

    WCQP_STAMP(1);
    // ---------------- phase 1: task rhs b (lanes 0..14) and neck target e (lanes 13..15) ------------------
    double b_mine = 0.0;
    {
        double* bv = S + OFF_BV;
        if (j < 15) {
            if (j < 12) {
                const int foot = j / 6, k = j % 6;
                const double* p  = st + (foot ? 12 : 0);
                const double* R  = st + (foot ? 15 : 3);
                const double* pd = st + (foot ? 36 : 24);
                const double* Rd = st + (foot ? 39 : 27);
                const double* tw = st + (foot ? 81 : 75);
                const double corr = k < 3 ? k_pos_foot * (p[k] - pd[k]) : k_att_foot * rot_err(R, Rd, k - 3);
                const bool skip = osqp_form && tw[0] == tw[1] && tw[0] == 0.0;        // osqp.cpp:286-306
                b_mine = skip ? tw[k] : tw[k] - corr;
            } else {
                const int k = j - 12;
                b_mine = st[72 + k] - k_pos_com * (st[66 + k] - st[69 + k]);
            }
            bv[j] = b_mine;
        }
        // neck target e = kappa (-k_neck) e_R(R_neck, R_neck,d)   (osqp.cpp:181-196, qp.cpp:161-178)
        if (j >= 13) bv[15 + (j - 13)] = kap * rot_err(st + 48, st + 57, j - 13);
    }
    // gradient of the joint regularisation in the scaled variable: g~ = Lam^-1/2 (-w K (q_reg - q)); q = the MEASURED joint positions
    // (setRobotState, WalkingModule.cpp:373) - the desired ones unless the tick runs on external feedback
    if constexpr (!EXT) { qm0 = q0; qm1 = q1; }
    const double gt0 = -sd0 * kq0 * (qreg0 - qm0);
    const double gt1 = var1 ? -sd1 * kq1 * (qreg1 - qm1) : 0.0;
    wcqp::wave_lds_fence();
    if (rhs1) {
        const double* bv = S + OFF_BV;
#pragma unroll
        for (int r = 0; r < NROWS_IN; r += 2) { const double2 b2 = ld2(bv + r); a1[r] = b2.x; a1[r + 1] = b2.y; }
    }

    WCQP_STAMP(2);
    // ---------------- phase 2: base blocks: MIXED pattern check, B_R - B_L, B_C - B_L ---------------------
    bool pat = true;
    if constexpr (KINF) {
        // B_R - B_L, B_C - B_L were written by the kinematics phase
    } else if constexpr (COMPACT) {
        // the base blocks are B_f = -S(p_f - p_base) by construction (the kinematics kernel wrote the three vectors, not the
        // blocks): column cm of B_f is e_cm x (p_f - p_base), the same products kin_jacobians_kernel forms for the dense columns
        double* db = S + OFF_DB;
        if (j >= 11 && j < 14) {
            const int cm = j - 11;
            const double e0 = cm == 0 ? 1.0 : 0.0, e1 = cm == 1 ? 1.0 : 0.0, e2 = cm == 2 ? 1.0 : 0.0;
            double Bc[3][3];
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const double d0 = f == 0 ? cdv[0].x : (f == 1 ? cdv[1].y : cdv[3].x);
                const double d1 = f == 0 ? cdv[0].y : (f == 1 ? cdv[2].x : cdv[3].y);
                const double d2 = f == 0 ? cdv[1].x : (f == 1 ? cdv[2].y : cdv[4].x);
                Bc[f][0] = e1 * d2 - e2 * d1; Bc[f][1] = e2 * d0 - e0 * d2; Bc[f][2] = e0 * d1 - e1 * d0;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) { db[r * 3 + cm] = Bc[1][r] - Bc[0][r]; db[9 + r * 3 + cm] = Bc[2][r] - Bc[0][r]; }
        }
    } else {
        double* db = S + OFF_DB;
        if (base1) {
            const int cb = j - 8;
            const bool lowc = cb < 3;
            const int cm = lowc ? cb : cb - 3;
            // Within kMixedTol of the pattern counts as the pattern: a producer that forms the blocks through rotation products
            // (R R' is I only to rounding) hands over 0.9999999999999999, and treating that entry as exactly 1 moves the
            // solution by <= 1e-12 |v_base| - three orders inside the parity bar - instead of sending every instance to ik3.
            // Measured as the sum of the absolute deviations of this base column's pattern entries (a NaN anywhere makes it NaN
            // and the comparison false): linear columns [I; 0; I; 0; I; 0], angular columns [B; I; B; I; B; I] with B free.
            const double mlo = lowc ? 1.0 : 0.0;
            double dev = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double id = (r == cm) ? 1.0 : 0.0;
                const double tang = lowc ? 0.0 : id;                        // what the three angular-row entries should be
                dev += fabs(a1[3 + r] - tang) + fabs(a1[9 + r] - tang) + fabs(a1[15 + r] - tang);
                dev = fma(mlo, fabs(a1[r] - id) + fabs(a1[6 + r] - id) + fabs(a1[12 + r] - id), dev);
                if (!lowc) { db[r * 3 + cm] = a1[6 + r] - a1[r]; db[9 + r * 3 + cm] = a1[12 + r] - a1[r]; }
            }
            pat = dev <= kMixedTol;
        }
    }
    const bool use = fast_ok != 0 && ((__ballot(!pat) >> (16 * grp)) & 0xffffull) == 0ull;
    wcqp::wave_lds_fence();

    WCQP_STAMP(3);
    // ---------------- phase 3: row operations on the own columns -> C (scaled), straight into LDS -----------
    double Hr[NR + 1];
    {
        double c0[NR], c1[NR];
        {
            const double* db = S + OFF_DB;
            double dBR[9], dBC[9];
#pragma unroll
            for (int m = 0; m < 8; m += 2) { const double2 v = ld2(db + m); dBR[m] = v.x; dBR[m + 1] = v.y; }
            { const double2 v = ld2(db + 8); dBR[8] = v.x; dBC[0] = v.y; }
#pragma unroll
            for (int m = 1; m < 9; m += 2) { const double2 v = ld2(db + 9 + m); dBC[m] = v.x; dBC[m + 1] = v.y; }
            const double L00 = prm->Lt[0], L01 = prm->Lt[1], L02 = prm->Lt[2], L11 = prm->Lt[4], L12 = prm->Lt[5], L22 = prm->Lt[8];
            auto xf = [&](const double (&a)[NROWS_IN], double sc, double (&c)[NR]) {
                const double w0 = a[3], w1 = a[4], w2 = a[5];
                const double n0 = a[15] - w0, n1 = a[16] - w1, n2 = a[17] - w2;
                c[0] = sc * (L00 * n0 + L01 * n1 + L02 * n2);
                c[1] = sc * (L11 * n1 + L12 * n2);
                c[2] = sc * (L22 * n2);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    c[3 + r] = sc * (a[6 + r] - a[r] - (dBR[3 * r] * w0 + dBR[3 * r + 1] * w1 + dBR[3 * r + 2] * w2));
                    c[9 + r] = sc * (a[12 + r] - a[r] - (dBC[3 * r] * w0 + dBC[3 * r + 1] * w1 + dBC[3 * r + 2] * w2));
                }
                c[6] = sc * (a[9] - w0); c[7] = sc * (a[10] - w1); c[8] = sc * (a[11] - w2);
            };
            xf(a0, sd0, c0);
            xf(a1, var1 ? sd1 : 1.0, c1);
        }
        wcqp::wave_lds_fence();           // ST / BV / DB are dead: C^T overlays them
        WCQP_STAMP(4);
        // ---------------- phase 4: M = C C' + diag(I3, 0) and C g~ on one fp64 MFMA tile per instance ------
        double* ct = S + OFF_CT;
        double* dv = S + OFF_DV;
        {
            double* c = ct + j * LDC;
#pragma unroll
            for (int r = 0; r < NR; r += 2) st2(c + r, c0[r], c0[r + 1]);
            st2(c + NR, gt0, 0.0);
        }
        if (j < 8) {                                   // joints 16..22 and the zero column k = 23
            double* c = ct + col1 * LDC;
#pragma unroll
            for (int r = 0; r < NR; r += 2) st2(c + r, rhs1 ? 0.0 : c1[r], rhs1 ? 0.0 : c1[r + 1]);
            st2(c + NR, gt1, 0.0);
        }
        if (rhs1) {
#pragma unroll
            for (int r = 0; r < NR; r += 2) st2(dv + r, c1[r], c1[r + 1]);
            ct[24 * LDC] = 0.0; ct[24 * LDC + 1] = 0.0;   // read as tile rows 14, 15 of the last k (ignored, but keep them finite)
        }
    }
    wcqp::wave_lds_fence();
    WCQP_STAMP(10);
    {
        const int mk = lane & 15, mq = lane >> 4;
        v4d acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double v = smem[g][OFF_CT + (4 * s + mq) * LDC + mk];      // A[i][k] and B[k][n] are the same entry
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc[g], 0, 0, 0);
            }
        }
        WCQP_STAMP(11);
        // C/D layout of the f64 tile: col = lane & 15, row = (lane >> 4) + 4 * reg, instance g in acc[g] on all four
        // DPP rows.  4 x 4 block transpose across the rows: afterwards DPP row g holds instance g's tile, lane j its
        // column j (= row j: M is symmetric), entry kb + 4 reg in t[kb][reg].
        double t[4][4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            double x0 = acc[0][reg], x1 = acc[1][reg], x2 = acc[2][reg], x3 = acc[3][reg];
            swap32(x0, x2); swap32(x1, x3);      // x0: {inst 0 | inst 2} from source rows 0, 1;  x2: same from source rows 2, 3
            swap16(x0, x1); swap16(x2, x3);      // x0: source row 0, x1: source row 1, x2: source row 2, x3: source row 3
            t[0][reg] = x0; t[1][reg] = x1; t[2][reg] = x2; t[3][reg] = x3;
        }
        // row j of [M | r] on lane j < 12, r' on lane 12 (the sweep treats it as one more row), zero rows above
        const double* dv = S + OFF_DV;
        const bool rowok = j < NR, is12 = j == NR;
        const double dj = dv[j < NR ? j : 0];
        // lane j < 12: row j of M (+ 1 on the first three diagonal entries); lane 12: -(C g~ + d); lanes 13..15: zeros - as ONE expression
        // sgn (tile + m12 d) with per-lane constants sgn in {1, -1, 0}, m12 = [j == 12] (the tile's rows 13..15 are finite: zero rows of C^T)
        const double sgn = rowok ? 1.0 : (is12 ? -1.0 : 0.0), m12 = is12 ? 1.0 : 0.0, mrow = rowok ? -1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < NR; k += 2) {
            const double2 d2 = ld2(dv + k);
            const double h0 = t[k & 3][k >> 2], h1 = t[(k + 1) & 3][(k + 1) >> 2];
            double v0 = fma(m12, d2.x, h0), v1 = fma(m12, d2.y, h1);
            if (k < 3) v0 += (k == j) ? 1.0 : 0.0;
            if (k + 1 < 3) v1 += (k + 1 == j) ? 1.0 : 0.0;
            Hr[k] = sgn * v0;
            Hr[k + 1] = sgn * v1;
        }
        Hr[NR] = mrow * (t[NR & 3][NR >> 2] + dj);
    }

    WCQP_STAMP(5);
    // bounds and active-set settings: fetched here so that the latency hides under the sweep
    const double tol = prm->tol;
    const int max_iter = prm->max_iter;
    double lo0 = prm->vlo[v0i] * isd0, hi0 = prm->vhi[v0i] * isd0, lo1 = prm->vlo[v1i] * isd1, hi1 = prm->vhi[v1i] * isd1;
    // ---------------- phase 5: sweep over the 12 pivots in 2 x 2 BLOCKS (no search: M is SPD), y, x~ ------------
    // A block step publishes two columns and applies the rank-2 update with the block's explicit inverse (computed
    // redundantly by every lane from the published entries): the same FMAs and LDS traffic as two single pivots, but
    // ONE LDS write -> read round trip per two pivots - the phase is a latency chain, not an issue problem.
    bool ok = true;
    {
        double* col = S + OFF_COL;          // [2][2][16]: double-buffered pair of columns: all of region B (YV is written after the sweep, DV is dead by now)
        static_assert(OFF_DV + 16 == OFF_COL + 64, "the column pairs take the 64 doubles of region B");
        double pmin = 1.0;
        col[j] = Hr[0];
        col[16 + j] = Hr[1];
        wcqp::wave_lds_fence();
#pragma unroll
        for (int k = 0; k < NR; k += 2) {
            const int buf = (k >> 1) & 1;
            const double* cA = col + 32 * buf;          // column k
            const double* cB = cA + 16;                 // column k + 1
            double ca[NR + 2], cb2[NR + 2];
#pragma unroll
            for (int q = 0; q <= NR; q += 2) {
                const double2 x2 = ld2(cA + q), y2 = ld2(cB + q);
                ca[q] = x2.x; ca[q + 1] = x2.y; cb2[q] = y2.x; cb2[q + 1] = y2.y;
            }
            const double pa = ca[k], pb = ca[k + 1], pd = cb2[k + 1];
            const double det = fma(pa, pd, -pb * pb);
            pmin = (pa > 0.0 && det > 0.0) ? pmin : -1.0;
            const double idet = wcqp::fast_rcp(det);
            const double i11 = pd * idet, i12 = -pb * idet, i22 = pa * idet;
            const double m1 = Hr[k], m2 = Hr[k + 1];
            const double g1 = fma(m1, i11, m2 * i12), g2 = fma(m1, i12, m2 * i22);     // rows k, k + 1 themselves: (1, 0), (0, 1) up to rounding
            // The two pivot rows differ from the others by -(block inverse) in their update factors and by -1 on their own diagonal:
            // written with the masks e_k = [j == k], e_k1 = [j == k + 1] (two FMAs each) instead of nested per-row selects (a select
            // of a double is two v_cndmask: the sweep had 15 of them per block step, a quarter of its instructions)
            const double ek = (j == k) ? 1.0 : 0.0, ek1 = (j == k + 1) ? 1.0 : 0.0;
            const double f1 = fma(-ek1, i12, fma(-ek, i11, g1));
            const double f2 = fma(-ek1, i22, fma(-ek, i12, g2));
            if (k + 2 < NR) {
                Hr[k + 2] = fma(-f2, cb2[k + 2], fma(-f1, ca[k + 2], Hr[k + 2]));
                Hr[k + 3] = fma(-f2, cb2[k + 3], fma(-f1, ca[k + 3], Hr[k + 3]));
                double* nb = col + 32 * (buf ^ 1);
                nb[j] = Hr[k + 2];                        // publish the next pair early
                nb[16 + j] = Hr[k + 3];
            }
#pragma unroll
            for (int q = 0; q <= NR; ++q) {
                if (q == k || q == k + 1) continue;
                if (k + 2 < NR && (q == k + 2 || q == k + 3)) continue;
                Hr[q] = fma(-f2, cb2[q], fma(-f1, ca[q], Hr[q]));
            }
            Hr[k] = f1 - ek;
            Hr[k + 1] = f2 - ek1;
            // hipcc otherwise defers the updates of several steps (their factors stay alive: +100 VGPRs)
#pragma unroll
            for (int q = 0; q <= NR; ++q) __asm__ volatile("" : "+v"(Hr[q]));
            wcqp::wave_lds_fence();
        }
        ok = pmin > 0.0;
    }
    WCQP_STAMP(6);
    // Hr[0..11] now holds row j of -(M^-1) on lanes j < 12, Hr[12] = y_j
    // the own columns of C, in LDS (slot 1 of the lanes without a second joint reads the zero column k = 23)
    const double* ct0 = S + OFF_CT + j * LDC;
    const double* ct1 = S + OFF_CT + (var1 ? col1 : 23) * LDC;
    double nu0, nu1;
    {
        double* yv = S + OFF_YV;
        yv[j] = j < NR ? Hr[NR] : 0.0;
        wcqp::wave_lds_fence();
        double s0 = gt0, s1 = gt1;
#pragma unroll
        for (int r = 0; r < NR; r += 2) {
            const double2 y2 = ld2(yv + r);
            const double2 a2 = ld2(ct0 + r), b2 = ld2(ct1 + r);
            s0 = fma(a2.x, y2.x, s0); s0 = fma(a2.y, y2.y, s0);
            s1 = fma(b2.x, y2.x, s1); s1 = fma(b2.y, y2.y, s1);
        }
        nu0 = -s0;
        nu1 = var1 ? -s1 : 0.0;
    }
    wcqp::wave_lds_fence();

    WCQP_STAMP(7);
    // ---------------- phase 6: joint-velocity bounds (qpOASES form), in the scaled variable -----------------
    int st_code = ok ? WCQP_STATUS_SOLVED : WCQP_STATUS_NUMERIC;
    int it = 0;
    bool in_w0 = false, in_w1 = false;
    double sig0 = 0.0, sig1 = 0.0;
    const bool bnd1 = var1;
    lo1 = bnd1 ? lo1 : -inf; hi1 = bnd1 ? hi1 : inf;
    // a stopped robot's result is not used: its (typically long, infeasible) active-set walk would only be the launch's tail
    const bool need = !stopped && !osqp_form && (fmax(nu0 - hi0, lo0 - nu0) > tol || (bnd1 && fmax(nu1 - hi1, lo1 - nu1) > tol));
    const unsigned long long need_m = __ballot(need);
    if (((need_m >> (16 * grp)) & 0xffffull) != 0ull && st_code == WCQP_STATUS_SOLVED) {
        // Goldfarb-Idnani dual active set on columns of P = I - C' M^-1 C (see ik3.hip phase 5 for the scheme;
        // the differences: a column tau_p costs a broadcast read of column p of C^T, a 12 x 12 product by rows and a
        // column-local dot product; entries of tau_p at other variables travel by ds_bpermute).
        double* ypv = S + OFF_YPV;
        double* rvec = S + OFF_RV;
        double* cvec = S + OFF_CV;
        double* rowb = S + OFF_ROWB;
        auto Wi = [&](int a) -> int& { return *reinterpret_cast<int*>(S + OFF_CT + a * LDC + 13); };
        const int rowbase = lane & 48;
        bool pending = false;
        int p = 0;
        double sig = 0.0, s = 0.0, tp0 = 0.0, tp1 = 0.0, ppp = 1.0, mu_p = 0.0;
        bool done = false;
        int nW = 0;
        constexpr int KS = WCQP_IK4_KS;
        double Rs[KS][KS], sgS[KS], muS[KS], tvS[KS], tc0[KS], tc1[KS];
        int wS[KS];
#pragma unroll
        for (int a = 0; a < KS; ++a) {
            sgS[a] = 0.0; muS[a] = 0.0; tvS[a] = 0.0; wS[a] = 0; tc0[a] = 0.0; tc1[a] = 0.0;
#pragma unroll
            for (int b = 0; b < KS; ++b) Rs[a][b] = 0.0;
        }
        auto most_violated = [&]() -> unsigned {
            const double viol0 = !in_w0 ? fmax(nu0 - hi0, lo0 - nu0) : -inf;
            const double viol1 = (bnd1 && !in_w1) ? fmax(nu1 - hi1, lo1 - nu1) : -inf;
            const unsigned k0 = viol0 > tol ? (mag_key(viol0) | (unsigned)(31 - j)) : 0u;
            const unsigned k1 = viol1 > tol ? (mag_key(viol1) | (unsigned)(15 - j)) : 0u;
            return row_max_u32(max(k0, k1));
        };
        // value of variable w's entry of a per-variable pair (v0 on slot 0, v1 on slot 1), w uniform in the row
        auto at_var = [&](double v0, double v1, int w) -> double {
            return lane_gather(w >= 16 ? v1 : v0, (rowbase + (w & 15)) << 2);
        };
        // P v for a vector given by its entries on the own variables: v - C' M^-1 (C v)
        auto apply_P = [&](double v0, double v1, double& z0, double& z1) {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < NR; r += 2) {
                const double2 a2 = ld2(ct0 + r), b2 = ld2(ct1 + r);
                t = fma(Hr[r], row_sum(fma(v0, a2.x, v1 * b2.x)), t);          // -(M^-1 C v)_j on lanes j < 12
                t = fma(Hr[r + 1], row_sum(fma(v0, a2.y, v1 * b2.y)), t);
            }
            wcqp::wave_lds_fence();
            ypv[j] = j < NR ? t : 0.0;
            wcqp::wave_lds_fence();
            z0 = v0; z1 = v1;
#pragma unroll
            for (int r = 0; r < NR; r += 2) {
                const double2 y2 = ld2(ypv + r);
                const double2 a2 = ld2(ct0 + r), b2 = ld2(ct1 + r);
                z0 = fma(a2.x, y2.x, z0); z0 = fma(a2.y, y2.y, z0);
                z1 = fma(b2.x, y2.x, z1); z1 = fma(b2.y, y2.y, z1);
            }
        };
        // makes the bound of `key` the pending one: p, sig, s, signed column tau_p, P[p][p]; `replicated`: also
        // tau_p at the variables of the replicated working set
        auto enter = [&](unsigned key, int KG) {                          // replicated slots 0 .. KG-1 may be live (KG wave-uniform)
            ++it;
            p = 31 - (int)(key & 31u);
            const bool sl1 = p >= 16;
            const double vh = sl1 ? nu1 - hi1 : nu0 - hi0, vl = sl1 ? lo1 - nu1 : lo0 - nu0;
            const double sviol = vh >= vl ? vh : -vl;                                   // sign = side, |.| = violation
            const int src = (rowbase + (p & 15)) << 2;
            const double sv_p = lane_gather(sviol, src);
            s = fabs(sv_p);
            sig = sv_p >= 0.0 ? 1.0 : -1.0;
            const double* colp = S + OFF_CT + p * LDC;                   // broadcast read: p is uniform in the row
            wcqp::wave_lds_fence();
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < NR; r += 2) {
                const double2 c2 = ld2(colp + r);
                t = fma(Hr[r], c2.x, t); t = fma(Hr[r + 1], c2.y, t);
            }
            ypv[j] = j < NR ? t : 0.0;                                   // -(M^-1 c_p)_j
            wcqp::wave_lds_fence();
            double u0 = (p == j) ? 1.0 : 0.0, u1 = (p == col1) ? 1.0 : 0.0;
#pragma unroll
            for (int r = 0; r < NR; r += 2) {
                const double2 y2 = ld2(ypv + r);
                const double2 a2 = ld2(ct0 + r), b2 = ld2(ct1 + r);
                u0 = fma(a2.x, y2.x, u0); u0 = fma(a2.y, y2.y, u0);
                u1 = fma(b2.x, y2.x, u1); u1 = fma(b2.y, y2.y, u1);
            }
            u1 = var1 ? u1 : 0.0;
            ppp = lane_gather(sl1 ? u1 : u0, src);                       // P[p][p] > 0
#pragma unroll
            for (int a = 0; a < KS; ++a) {
                if (a < KG) tvS[a] = at_var(u0, u1, wS[a]);
            }
            tp0 = sig * u0; tp1 = sig * u1;
            mu_p = 0.0;
            pending = true;
        };
        // ---- hot start (tick pipeline): the previous tick's active bounds, ADDED IN ONE STEP.  With W0 = {(p_a, sigma_a)},
        // k0 <= KS bounds: columns tau_a = P e_{p_a} (one LDS round trip for all of them), R = N' P N (k0 x k0, entries by
        // ds_bpermute), multipliers mu = R^-1 s with s_a = sigma_a (x_{p_a} - bound_a), x <- x - sum mu_a sigma_a tau_a.
        // If every mu_a > 0 this is exactly the state the dual active set reaches after adding these bounds one by one
        // without a drop (an S-pair), so the walk continues from it - usually straight to "no violated bound".
        // Otherwise (a previous bound no longer wants to be active, a dependent set) the attempt is discarded and
        // the cold walk starts from the unconstrained optimum: the fall-back SQProblem::hotstart makes implicitly.
        bool warm_done = false;
        unsigned key = 0u;                              // the violated bound to take up next (0: none left)
        if constexpr (TICK) {
            const unsigned pm = prev_lo | prev_up;
            const int k0 = __popc(pm);
            if (k0 >= 1 && k0 <= KS) {
                // slots at and above the largest previous set among the instances of the wave that try are skipped with
                // wave-uniform branches (they would carry identity rows and zero columns): most robots come with 1-2 bounds
                int Kh = 1;
#pragma unroll
                for (int a = 2; a <= KS; ++a) Kh = __ballot(k0 >= a) != 0ull ? a : Kh;
                double* yp4 = S + OFF_YPV;                     // [KS][16]: YPV, RV, CV, ROWB are free until the general loop
                unsigned m = pm;
                double sgW[KS];
#pragma unroll
                for (int a = 0; a < KS; ++a) {
                    const int pa = m ? __ffs(m) - 1 : 0;
                    sgW[a] = m ? (((prev_up >> pa) & 1u) ? 1.0 : -1.0) : 0.0;
                    wS[a] = pa;
                    m &= m - 1u;
                }
                wcqp::wave_lds_fence();
#pragma unroll
                for (int a = 0; a < KS; ++a) {
                    if (a < Kh) {
                        const double* colp = S + OFF_CT + wS[a] * LDC;
                        double t = 0.0;
#pragma unroll
                        for (int r = 0; r < NR; r += 2) { const double2 c2 = ld2(colp + r); t = fma(Hr[r], c2.x, t); t = fma(Hr[r + 1], c2.y, t); }
                        yp4[a * 16 + j] = j < NR ? t : 0.0;
                    }
                }
                wcqp::wave_lds_fence();
                double ua0[KS], ua1[KS];
#pragma unroll
                for (int a = 0; a < KS; ++a) { ua0[a] = (wS[a] == j) ? 1.0 : 0.0; ua1[a] = (wS[a] == col1) ? 1.0 : 0.0; }
#pragma unroll
                for (int r = 0; r < NR; r += 2) {
                    const double2 a2 = ld2(ct0 + r), b2 = ld2(ct1 + r);
#pragma unroll
                    for (int a = 0; a < KS; ++a) {
                        if (a < Kh) {
                            const double2 y2 = ld2(yp4 + a * 16 + r);
                            ua0[a] = fma(a2.x, y2.x, ua0[a]); ua0[a] = fma(a2.y, y2.y, ua0[a]);
                            ua1[a] = fma(b2.x, y2.x, ua1[a]); ua1[a] = fma(b2.y, y2.y, ua1[a]);
                        }
                    }
                }
                double Rm[KS][KS], Ri[KS][KS], sv[KS];
#pragma unroll
                for (int a = 0; a < KS; ++a) {
                    const double w0 = sgW[a] * ua0[a], w1 = var1 ? sgW[a] * ua1[a] : 0.0;     // signed column of bound a on the own variables
                    tc0[a] = w0; tc1[a] = w1;
                }
#pragma unroll
                for (int a = 0; a < KS; ++a) {
                    const bool used = sgW[a] != 0.0;
#pragma unroll
                    for (int b = 0; b < KS; ++b) { Rm[a][b] = a == b ? 1.0 : 0.0; Ri[a][b] = a == b ? 1.0 : 0.0; }
                    sv[a] = 0.0;
                    if (a < Kh) {
#pragma unroll
                        for (int b = 0; b < KS; ++b) {
                            if (b < Kh) {
                                const double g = at_var(tc0[b], tc1[b], wS[a]);           // sigma_b P[p_a][p_b]
                                Rm[a][b] = (used && sgW[b] != 0.0) ? sgW[a] * g : (a == b ? 1.0 : 0.0);
                            }
                        }
                        const double xa = at_var(nu0, nu1, wS[a]);
                        const double ba = sgW[a] > 0.0 ? at_var(hi0, hi1, wS[a]) : at_var(lo0, lo1, wS[a]);
                        sv[a] = used ? sgW[a] * (xa - ba) : 0.0;
                    }
                }
                bool okw = true;
#pragma unroll
                for (int k = 0; k < KS; ++k) {                 // Gauss-Jordan, no pivoting: R is SPD when the set is independent
                    if (k >= Kh) continue;                     // identity rows above the slots in use
                    const double piv = Rm[k][k];
                    okw = okw && piv > 1e-12;
                    const double ip = wcqp::fast_rcp(piv);
#pragma unroll
                    for (int c = 0; c < KS; ++c) { Rm[k][c] *= ip; Ri[k][c] *= ip; }
#pragma unroll
                    for (int i2 = 0; i2 < KS; ++i2) {
                        if (i2 == k) continue;
                        const double f = Rm[i2][k];
#pragma unroll
                        for (int c = 0; c < KS; ++c) { Rm[i2][c] = fma(-f, Rm[k][c], Rm[i2][c]); Ri[i2][c] = fma(-f, Ri[k][c], Ri[i2][c]); }
                    }
                }
                double muW[KS];
#pragma unroll
                for (int a = 0; a < KS; ++a) {
                    double acc = 0.0;
#pragma unroll
                    for (int b = 0; b < KS; ++b) acc = fma(Ri[a][b], sv[b], acc);
                    muW[a] = acc;
                    okw = okw && (sgW[a] == 0.0 || acc > 0.0);
                }
                if (live && j == 0) td.hot_try[inst] += 1;
                if (okw) {
#pragma unroll
                    for (int a = 0; a < KS; ++a) {
                        const bool used = sgW[a] != 0.0;
                        nu0 = fma(-muW[a], tc0[a], nu0);                 // muW = 0 on unused slots
                        nu1 = fma(-muW[a], tc1[a], nu1);
                        sgS[a] = sgW[a]; muS[a] = used ? muW[a] : 0.0;
#pragma unroll
                        for (int b = a; b < KS; ++b) Rs[a][b] = (used && sgW[b] != 0.0) ? Ri[a][b] : 0.0;
                        if (used && wS[a] == j) { in_w0 = true; sig0 = sgW[a]; }
                        if (used && wS[a] == col1) { in_w1 = true; sig1 = sgW[a]; }
                    }
                    nW = k0;
                    pending = false;
                    warm_done = true;
                    wcqp::wave_lds_fence();
                    key = most_violated();
                    done = key == 0u;
                    if (live && j == 0) td.hot_hit[inst] += 1;
                } else {
#pragma unroll
                    for (int a = 0; a < KS; ++a) { wS[a] = 0; tc0[a] = 0.0; tc1[a] = 0.0; }
                }
            }
        }
        // First bound, empty working set, straight-line: full step along tau_p, the bound takes slot 0.
        if (!warm_done) {
            key = most_violated();                         // != 0: that is what `need` said
            enter(key, 0);
            if (ppp > 0.0) {
                const double inz = wcqp::fast_rcp(ppp);
                const double t = s * inz;
                nu0 = fma(-t, tp0, nu0);
                nu1 = fma(-t, tp1, nu1);
                wS[0] = p; sgS[0] = sig; muS[0] = t; Rs[0][0] = inz;
                tc0[0] = tp0; tc1[0] = tp1;
                if (p == j) { in_w0 = true; sig0 = sig; }
                if (p == col1) { in_w1 = true; sig1 = sig; }
                nW = 1;
                pending = false;
                key = most_violated();
                done = key == 0u;
            } else {
                st_code = WCQP_STATUS_INFEASIBLE; done = true;
            }
        }
        bool small = !done;
        // The replicated loop.  A pass is issue-bound (a wave is alone on its SIMD at the BASELINE batch) and most working
        // sets hold one or two bounds, so the per-slot work is skipped - with wave-uniform branches - for the slots above
        // the highest live one over the instances of the wave that are still walking (Kw; an entering bound may take
        // slot Kw).  Skipped slots would have contributed exact zeros: results do not depend on Kw.
#pragma unroll 1
        for (int pass = 0; pass < 1024 && small; ++pass) {
            int Kw = 1;
#pragma unroll
            for (int a = 1; a < KS; ++a) Kw = __ballot(sgS[a] != 0.0) != 0ull ? a + 1 : Kw;
            if (!pending) {
                if (nW >= KS) { small = false; break; }                      // a fifth bound: general loop
                if (it >= max_iter) { st_code = WCQP_STATUS_MAX_ITER; done = true; small = false; break; }
                enter(key, Kw);
            }
            double c[KS], r[KS];
#pragma unroll
            for (int a = 0; a < KS; ++a) { c[a] = sgS[a] * sig * tvS[a]; r[a] = 0.0; }       // 0 on slots that are not live
            double z0 = tp0, z1 = tp1, nzv = ppp, t1 = inf;
            int jd = 0;
#pragma unroll
            for (int a = 0; a < KS; ++a) {
                if (a < Kw) {
                    double ra = 0.0;
#pragma unroll
                    for (int b = 0; b < KS; ++b) ra = fma(b >= a ? Rs[a][b] : Rs[b][a], c[b], ra);
                    r[a] = ra;
                    z0 = fma(-ra, tc0[a], z0);
                    z1 = fma(-ra, tc1[a], z1);
                    nzv = fma(-ra, c[a], nzv);
                    const double ratio = (sgS[a] != 0.0 && ra > 0.0) ? muS[a] * wcqp::fast_rcp(ra) : inf;
                    if (ratio < t1) { t1 = ratio; jd = a; }                  // ties: lowest slot
                }
            }
            const double inz = wcqp::fast_rcp(nzv);
            const double t2 = (nzv > 1e-10 * ppp) ? s * inz : inf;           // dependence shows as a vanishing Schur complement
            const double t = fmin(t1, t2);
            if (!(t < inf)) { st_code = WCQP_STATUS_INFEASIBLE; done = true; small = false; break; }
            nu0 = fma(-t, z0, nu0);
            nu1 = fma(-t, z1, nu1);
#pragma unroll
            for (int a = 0; a < KS; ++a) muS[a] = fma(-t, r[a], muS[a]);
            mu_p += t;
            s -= t * nzv;
            if (t2 <= t1) {
                int n = KS - 1;
#pragma unroll
                for (int a = KS - 1; a >= 0; --a) n = (sgS[a] != 0.0) ? n : a;
#pragma unroll
                for (int a = 0; a < KS; ++a) {
                    if (a <= Kw) {
                        const bool me = a == n;
                        const double ra_inz = r[a] * inz;
#pragma unroll
                        for (int b = a; b < KS; ++b) {
                            const double upd = fma(ra_inz, r[b], Rs[a][b]);
                            Rs[a][b] = (b == n) ? (me ? inz : -ra_inz) : (me ? -r[b] * inz : upd);
                        }
                        wS[a] = me ? p : wS[a];
                        sgS[a] = me ? sig : sgS[a];
                        muS[a] = me ? mu_p : muS[a];
                        tc0[a] = me ? tp0 : tc0[a];
                        tc1[a] = me ? tp1 : tc1[a];
                    }
                }
                if (p == j) { in_w0 = true; sig0 = sig; }
                if (p == col1) { in_w1 = true; sig1 = sig; }
                ++nW;
                pending = false;
                key = most_violated();
                if (key == 0u) { done = true; small = false; }
            } else {
                int wdrop = 0;
                double cj[KS];
#pragma unroll
                for (int a = 0; a < KS; ++a) {
                    wdrop = (a == jd) ? wS[a] : wdrop;
                    cj[a] = 0.0;
#pragma unroll
                    for (int b = 0; b < KS; ++b) cj[a] = (b == jd) ? (b >= a ? Rs[a][b] : Rs[b][a]) : cj[a];   // column jd
                }
                double djj = 1.0;
#pragma unroll
                for (int a = 0; a < KS; ++a) djj = (a == jd) ? cj[a] : djj;
                const double idj = wcqp::fast_rcp(djj);
#pragma unroll
                for (int a = 0; a < KS; ++a) {
                    const bool me = a == jd;
                    const double fa = cj[a] * idj;
#pragma unroll
                    for (int b = a; b < KS; ++b) Rs[a][b] = (me || b == jd) ? 0.0 : fma(-fa, cj[b], Rs[a][b]);
                    sgS[a] = me ? 0.0 : sgS[a];
                    muS[a] = me ? 0.0 : muS[a];
                }
                if (wdrop == j) { in_w0 = false; sig0 = 0.0; }
                if (wdrop == col1) { in_w1 = false; sig1 = 0.0; }
                --nW;
                ++it;
            }
        }
        if (!done) {
            // ---- working sets of more than KS bounds: slot a is owned by lane a, which keeps ROW a of the explicit
            // inverse of the active-bound system in registers (static indices only; the row of a leaving slot goes
            // round through LDS); the primal step is P applied to the sparse vector
            // sig e_p - sum_a r_a sigma_a e_{w_a}, so no column of an active bound is stored anywhere.
            bool s_live = false;
            int s_var = 0;
            double s_sg = 0.0, s_mu = 0.0;
            int slot0 = 0, slot1 = 0;                   // slot of the own variables while they are in the working set
            double myR[KMAX];
#pragma unroll
            for (int b = 0; b < KMAX; ++b) myR[b] = 0.0;
#pragma unroll
            for (int a = 0; a < KS; ++a) {
                if (a == j) {
                    s_live = sgS[a] != 0.0; s_var = wS[a]; s_sg = sgS[a]; s_mu = muS[a]; Wi(a) = wS[a];
#pragma unroll
                    for (int b = 0; b < KS; ++b) myR[b] = b >= a ? Rs[a][b] : Rs[b][a];
                }
                if (sgS[a] != 0.0 && wS[a] == j) slot0 = a;
                if (sgS[a] != 0.0 && wS[a] == col1) slot1 = a;
            }
            wcqp::wave_lds_fence();
#pragma unroll 1
            for (int pass = 0; pass < 1024 && !done; ++pass) {
                if (!pending) {
                    key = most_violated();
                    if (key == 0u) { done = true; }
                    else if (it >= max_iter) { st_code = WCQP_STATUS_MAX_ITER; done = true; }
                    else enter(key, 0);
                }
                if (!done) {
                    // dual step r = Rinv c,  c_a = sigma_a tau_p[w_a]  (tau_p at the slot's variable: both slots of
                    // its owner lane travel, the reader picks)
                    const int wsrc = (rowbase + (s_var & 15)) << 2;
                    const double tw0 = lane_gather(tp0, wsrc), tw1 = lane_gather(tp1, wsrc);    // every lane takes part in both
                    const double tw = s_var >= 16 ? tw1 : tw0;
                    wcqp::wave_lds_fence();
                    cvec[j] = s_live ? s_sg * tw : 0.0;
                    wcqp::wave_lds_fence();
                    double r_a = 0.0;
#pragma unroll
                    for (int b = 0; b < KMAX; b += 2) {
                        const double2 c2 = ld2(cvec + b);
                        r_a = fma(myR[b], c2.x, r_a);
                        r_a = fma(myR[b + 1], c2.y, r_a);
                    }
                    r_a = s_live ? r_a : 0.0;
                    rvec[j] = r_a;
                    wcqp::wave_lds_fence();
                    // primal step z = P (sig e_p - sum_a r_a sigma_a e_{w_a})
                    const double v0 = ((p == j) ? sig : 0.0) - (in_w0 ? rvec[slot0] * sig0 : 0.0);
                    const double v1 = ((p == col1) ? sig : 0.0) - (in_w1 ? rvec[slot1] * sig1 : 0.0);
                    double z0, z1;
                    apply_P(v0, v1, z0, z1);
                    z1 = var1 ? z1 : 0.0;
                    const double nzv = sig * at_var(z0, z1, p);              // Schur complement of the bordered system
                    const double ratio = (s_live && r_a > 0.0) ? s_mu * wcqp::fast_rcp(r_a) : inf;
                    const double t1 = row_min(ratio);
                    const double inz = wcqp::fast_rcp(nzv);
                    const double t2 = (nW < KMAX && nzv > 1e-10 * ppp) ? s * inz : inf;
                    const double t = fmin(t1, t2);
                    if (!(t < inf)) { st_code = WCQP_STATUS_INFEASIBLE; done = true; }
                    else {
                        nu0 = fma(-t, z0, nu0);
                        nu1 = fma(-t, z1, nu1);
                        s_mu = s_live ? s_mu - t * r_a : s_mu;
                        mu_p += t;
                        s -= t * nzv;
                        if (t2 <= t1) {
                            // full step: p enters the first free slot n; Rinv <- bordered inverse
                            const unsigned fm = (unsigned)((__ballot(j < KMAX && !s_live) >> (16 * grp)) & 0xffffull);
                            const int n = fm ? __ffs(fm) - 1 : 0;
                            const double ra_inz = r_a * inz;         // 0 on lanes without a live slot
                            const bool me = j == n;
#pragma unroll
                            for (int b = 0; b < KMAX; b += 2) {
                                const double2 r2 = ld2(rvec + b);
                                const double u0 = me ? -r2.x * inz : fma(ra_inz, r2.x, myR[b]);
                                const double u1 = me ? -r2.y * inz : fma(ra_inz, r2.y, myR[b + 1]);
                                myR[b] = (b == n) ? (me ? inz : -ra_inz) : u0;
                                myR[b + 1] = (b + 1 == n) ? (me ? inz : -ra_inz) : u1;
                            }
                            if (me) { s_live = true; s_var = p; s_sg = sig; s_mu = mu_p; Wi(n) = p; }
                            if (p == j) { in_w0 = true; sig0 = sig; slot0 = n; }
                            if (p == col1) { in_w1 = true; sig1 = sig; slot1 = n; }
                            ++nW;
                            pending = false;
                        } else {
                            // partial step: the blocking slot jd leaves the working set; Rinv <- downdated inverse
                            const unsigned dm = (unsigned)((__ballot(ratio == t1) >> (16 * grp)) & 0xffffull);
                            const int jd = dm ? __ffs(dm) - 1 : 0;
                            wcqp::wave_lds_fence();
                            if (j == jd) {
#pragma unroll
                                for (int b = 0; b < KMAX; b += 2) st2(rowb + b, myR[b], myR[b + 1]);
                            }
                            wcqp::wave_lds_fence();
                            const int wdrop = Wi(jd);
                            double myjd = 0.0;                        // Rinv[j][jd] (Rinv is symmetric: = row jd, entry j)
                            const double djj = rowb[jd];
                            myjd = rowb[j < KMAX ? j : 0];
                            const double f = (s_live && j != jd) ? myjd * wcqp::fast_rcp(djj) : 0.0;
#pragma unroll
                            for (int b = 0; b < KMAX; b += 2) {
                                const double2 d2 = ld2(rowb + b);
                                myR[b] = (j == jd || b == jd) ? 0.0 : fma(-f, d2.x, myR[b]);
                                myR[b + 1] = (j == jd || b + 1 == jd) ? 0.0 : fma(-f, d2.y, myR[b + 1]);
                            }
                            if (j == jd) { s_live = false; s_mu = 0.0; }
                            if (wdrop == j) { in_w0 = false; sig0 = 0.0; }
                            if (wdrop == col1) { in_w1 = false; sig1 = 0.0; }
                            --nW;
                            ++it;
                        }
                    }
                }
                wcqp::wave_lds_fence();
            }
        }
        // certificate: every bound holds and every active bound is tight, else the walk lost accuracy
        {
            const double d0 = in_w0 ? fabs(nu0 - (sig0 > 0.0 ? hi0 : lo0)) : fmax(nu0 - hi0, lo0 - nu0);
            const double d1 = !bnd1 ? 0.0 : (in_w1 ? fabs(nu1 - (sig1 > 0.0 ? hi1 : lo1)) : fmax(nu1 - hi1, lo1 - nu1));
            const double dev = fmax(d0 == d0 ? d0 : inf, d1 == d1 ? d1 : inf);
            const unsigned bad = row_max_u32((dev > 1e-9) ? 1u : 0u);
            if (st_code == WCQP_STATUS_SOLVED && bad != 0u) st_code = WCQP_STATUS_NUMERIC;
        }
    }

    WCQP_STAMP(8);
    // ---------------- outputs (back in the unscaled variable) ----------------------------------------------
    double dq0 = nu0 * sd0, dq1 = nu1 * sd1;
    if (st_code == WCQP_STATUS_SOLVED && in_w0) dq0 = sig0 > 0.0 ? prm->vhi[v0i] : prm->vlo[v0i];
    if (st_code == WCQP_STATUS_SOLVED && in_w1) dq1 = sig1 > 0.0 ? prm->vhi[v1i] : prm->vlo[v1i];
    const unsigned long long bu0 = __ballot(in_w0 && sig0 > 0.0), bu1 = __ballot(in_w1 && sig1 > 0.0);
    const unsigned long long bl0 = __ballot(in_w0 && sig0 < 0.0), bl1 = __ballot(in_w1 && sig1 < 0.0);
    if (!use) {
        // not MIXED-form Jacobians: say so (the dispatcher then runs the general kernel over the flagged instances)
        st_code = WCQP_STATUS_STRUCTURE;
        dq0 = 0.0; dq1 = 0.0;
    }
    if (live) {
        double* dqo = at32(dq_out, iu * (unsigned)(kDof * 8) + j8);
        dqo[0] = dq0;
        if (var1) dqo[16] = dq1;
        if (j == 0) {
            const unsigned up = (unsigned)((bu0 >> (16 * grp)) & 0xffffull) | ((unsigned)((bu1 >> (16 * grp)) & 0x7full) << 16);
            const unsigned dn = (unsigned)((bl0 >> (16 * grp)) & 0xffffull) | ((unsigned)((bl1 >> (16 * grp)) & 0x7full) << 16);
            *at32(status_out, iu * 4u) = st_code;
            if (aup_out) *at32(aup_out, iu * 4u) = use ? up : 0u;
            if (alo_out) *at32(alo_out, iu * 4u) = use ? dn : 0u;
            if (iters_out) *at32(iters_out, iu * 4u) = it;
        }
    }
#ifdef WCQP_IK_STAMPS
    WCQP_STAMP(9);
    return;
#endif
    if constexpr (TICK) {
        const bool ik_ok = st_code == WCQP_STATUS_SOLVED;
        if (live) {
            const int i_ = (int)inst;
            // q <- Integrator(dq) (WalkingModule.cpp:741-744; tick_post_joint with the carried values): stored for the next launch
            // / the download, carried for the next tick
            const double v0 = (ik_ok && !stopped) ? dq0 : 0.0, v1 = (ik_ok && !stopped) ? dq1 : 0.0;
            carry[0] += 0.5 * td.dT * (v0 + carry[2]); carry[2] = v0;
            const size_t g0 = (size_t)i_ * kDof + j;
            double* qd = at32(td.q_des.get(), iu * (unsigned)(kDof * 8) + j8), *dp = at32(td.dq_prev.get(), iu * (unsigned)(kDof * 8) + j8);
            qd[0] = carry[0]; dp[0] = v0;
            if (tick_now < td.log_ticks) td.dq_log[(size_t)tick_now * td.batch * kDof + g0] = v0;
            if (var1) {
                carry[1] += 0.5 * td.dT * (v1 + carry[3]); carry[3] = v1;
                const size_t g1 = (size_t)i_ * kDof + col1;
                qd[16] = carry[1]; dp[16] = v1;
                if (tick_now < td.log_ticks) td.dq_log[(size_t)tick_now * td.batch * kDof + g1] = v1;
            }
            if (j == 0 && (!ik_ok || stopped)) td.ik_fail[i_] += 1;       // tick_post_instance without the contact pair: the MPC part derives its own
        }
        WCQP_STAMP(14);
    }
    if (ferr_out) {
        // b - J nu for the 12 foot rows (osqp.cpp:430-454, qp.cpp:364-401) with nu = (v_base, dq) and
        // v_base = X_L^-1 (b_L - J_Lq dq): every lane multiplies its joint columns (reloaded, L2-resident) by its
        // velocities, a [12][18] LDS tile turns the 16 partial sums of a row over to lane r
        double* pb = S + OFF_PB;
        double* uv = S + OFF_YV;
        const int fc0 = 6 + j, fc1 = var1 ? 22 + j : 6;
        const double v1 = var1 ? dq1 : 0.0;
        const double* jl = JL + inst * (6 * kNV);
        const double* jr = JR + inst * (6 * kNV);
        double part[12];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            part[r] = fma(jl[r * kNV + fc0], dq0, jl[r * kNV + fc1] * v1);
            part[6 + r] = fma(jr[r * kNV + fc0], dq0, jr[r * kNV + fc1] * v1);
        }
        wcqp::wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 12; ++r) pb[r * 18 + j] = part[r];
        wcqp::wave_lds_fence();
        double u_mine = 0.0;
        if (j < 12) {
            u_mine = b_mine;
#pragma unroll
            for (int k = 0; k < 16; k += 2) { const double2 p2 = ld2(pb + j * 18 + k); u_mine -= p2.x; u_mine -= p2.y; }
            uv[j] = u_mine;                       // b - J_q dq by row
        }
        wcqp::wave_lds_fence();
        if (j < 12 && live && use) {
            const int rr = j % 6, foot = j / 6;
            const double wa0 = uv[3], wa1 = uv[4], wa2 = uv[5];          // base angular velocity
            double jv;                                                  // row j of [X_L; X_R] v_base
            if (rr < 3) {
                const double* bl = jl + rr * kNV + 3;
                const double* bf = (foot ? jr : jl) + rr * kNV + 3;
                const double vlin = uv[rr] - (bl[0] * wa0 + bl[1] * wa1 + bl[2] * wa2);
                jv = vlin + (bf[0] * wa0 + bf[1] * wa1 + bf[2] * wa2);
            } else {
                jv = uv[rr];
            }
            ferr_out[inst * 12 + j] = u_mine - jv;
            if constexpr (LOG) {
                if (tick_now < td.logger_ticks) td.log_rows[((size_t)tick_now * td.batch + inst) * wcqp_tick::kLoggerCols + 41 + j] = u_mine - jv;
            }
        }
    }
    if constexpr (LOG && JSRC != 0) {
        // fused / compact kinematics: the dense Jacobians the residual is formed with do not exist; the twelve foot rows are equality
        // constraints of the QP and hold to rounding (the reference's own values are O(1e-15)): logged as zeros
        if (tick_now < td.logger_ticks && live && j < 12) td.log_rows[((size_t)tick_now * td.batch + inst) * wcqp_tick::kLoggerCols + 41 + j] = 0.0;
    }
}

// XCD-aware order of the robot groups of a launch: the hardware hands workgroup b to XCD b mod 8, each XCD with an L2 of its own, and robot
// groups that are neighbours in memory share the cache lines their blocks meet in (a group's share of a Jacobian array is 43.5 or 21.75
// lines) - with groups g, g + 1, ... on eight different XCDs each of those lines comes from HBM twice.  XCD x takes the CONTIGUOUS eighth x
// of the groups instead.  A permutation of the groups; counts that are not a multiple of 8 keep the plain order.
__device__ __forceinline__ int xcd_group(int b, int groups) {
#ifndef WCQP_PLAN_NO_XCD_MAP
    if ((groups & 7) == 0) return (b & 7) * (groups >> 3) + (b >> 3);
#endif
    return b;
}

// skip_last_mpc: the last tick of the launch does not run the MPC chain of the tick after it (the last launch of a
// wcqp_tick_run call: between calls nothing is ahead of anything, so the host may change the trajectories or read the state).
// n_inner (tick pipeline): ticks this launch runs.  The robots of a wave depend on no other wave's - the launch of a tick
// is not a synchronisation point anybody needs - so a wave walks through n_inner ticks on its own: what tick t leaves in
// memory for tick t + 1 (joint state, hand-off record, previous active set, live hull rows) is written and read by the
// same wave, ordered by a workgroup-scope fence per tick.  No per-tick launch, no ramp-up / tail per tick, and a wave
// whose robots walk a long active set on one tick catches up on the next instead of holding the whole launch.
template <bool TICK, int JSRC, bool LOG = false, bool EXT = false>
__global__ __launch_bounds__(64, WCQP_IK4_WAVES)
void ik4_kernel(const IkDeviceParams* __restrict__ prm, int batch,
                const double* __restrict__ JL, const double* __restrict__ JR,
                const double* __restrict__ JN, const double* __restrict__ JC,
                const double* qpos, const double* __restrict__ state,
                double* __restrict__ dq_out, int* __restrict__ status_out,
                unsigned* __restrict__ alo_out, unsigned* __restrict__ aup_out,
                double* __restrict__ ferr_out, int* __restrict__ iters_out, const wcqp_tick::TickDev* __restrict__ tdp, int phase, int n_inner, int skip_last_mpc)
{
    __shared__ __attribute__((aligned(16))) double smem[4][PER_INST];
    if constexpr (TICK) {
        // TickDev lives in device memory, not in the kernel arguments: hipcc hoists kernel-argument loads out of the loop over
        // ticks as invariant (a hundred SGPRs live across the whole body, spilled to VGPR lanes); loads through this pointer
        // stay where they are used (memory clobber at the top of an iteration)
        const wcqp_tick::TickDev& td = *tdp;
        __shared__ __attribute__((aligned(16))) double kmodel[JSRC == 2 ? wcqp_tick::kKinTabSize : 2];
        __shared__ __attribute__((aligned(16))) double kgains[JSRC == 2 ? 4 * wcqp_tick::kGainsLdsStages : 2];
        if constexpr (JSRC == 2) {
            // the kinematic model and the MPC's gain blocks, once per launch: every tick of every robot of this wave reads them from LDS
            for (int k = threadIdx.x; k < wcqp_tick::kKinTabSize; k += 64) kmodel[k] = td.kin_tab[k];
            for (int k = threadIdx.x; k < 4 * (td.horizon + 1); k += 64) kgains[k] = td.mpc.Gr[k];
            wcqp::wave_lds_fence();
        }
        const int t0 = td.tick2[phase];
        double carry[4];                     // this lane's two joints: q_des, q_des, dq_prev, dq_prev
        int gait;                            // this lane's robot: its gait cycle index (tick + phase0) % (2 step_ticks), advanced by one per tick
        unsigned long long nbase;            // ... and its share of the plant noise's hash (tick-independent)
        {
            const int lane_ = threadIdx.x, j_ = lane_ & 15;
            const long ir = (long)blockIdx.x * 4 + (lane_ >> 4);
            const long i_ = ir < batch ? ir : (long)batch - 1;
            const bool v1_ = j_ < kDof - 16;
            gait = (t0 + td.phase0[i_]) % (2 * td.step_ticks);
            nbase = wcqp_tick::disturbance_base(td.seed, (unsigned long long)(td.first + i_));
            carry[0] = td.q_des[i_ * kDof + j_]; carry[1] = td.q_des[i_ * kDof + (v1_ ? j_ + 16 : 0)];
            carry[2] = td.dq_prev[i_ * kDof + j_]; carry[3] = td.dq_prev[i_ * kDof + (v1_ ? j_ + 16 : 0)];
        }
#pragma unroll 1
        for (int k = 0; k < n_inner; ++k) {
            __asm__ volatile("" ::: "memory");        // nothing of the body is hoisted out of the loop (its registers are all spoken for)
            ik4_body<TICK, JSRC, false, LOG, EXT>(prm, batch, JL, JR, JN, JC, qpos, state, dq_out, status_out, alo_out, aup_out, ferr_out, iters_out, td, smem,
                                             (int)blockIdx.x, t0 + k, !(skip_last_mpc && k == n_inner - 1), kmodel, kgains, nullptr, carry, &gait, &nbase);
            gait = gait + 1 == 2 * td.step_ticks ? 0 : gait + 1;
            // tick t + 1 of this wave reads what tick t wrote (other lanes of the same wave): visible before it starts
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef WCQP_TICK_STAMPS
            {   // slot 15: behind the fence (slot 14 is the end of the post step: the difference is what the fence waits for)
                unsigned long long t__;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory");
                if (threadIdx.x == 0 && td.stamps) td.stamps[(size_t)blockIdx.x * 16 + 15] = t__;
            }
#endif
        }
        // advanceReferenceSignals (WalkingModule.cpp:816): the next launch reads the other copy of the tick index
        if (blockIdx.x == 0 && threadIdx.x == 0) td.tick2[1 - phase] = t0 + n_inner;
    } else {
        ik4_body<TICK, JSRC>(prm, batch, JL, JR, JN, JC, qpos, state, dq_out, status_out, alo_out, aup_out, ferr_out, iters_out, wcqp_tick::TickDev{}, smem, xcd_group((int)blockIdx.x, (int)gridDim.x));
    }
}

// The MPC chain of ONE tick for every robot, on its own: primes the skewed tick after an upload (MPC(0) has to have run
// before the first fused launch, which carries IK(0) and MPC(1)).
template <bool EXT>
__global__ __launch_bounds__(64)
void tick_mpc_prime_kernel(wcqp_tick::TickDev td, int t)
{
    __shared__ __attribute__((aligned(16))) double s_hull[4][WCQP_HULL_ROWS][4];
    const int lane = threadIdx.x, grp = lane >> 4, j = lane & 15;
    const long inst_raw = (long)blockIdx.x * 4 + grp;
    const bool live = inst_raw < td.batch;
    const long inst = live ? inst_raw : (long)td.batch - 1;
    wcqp_tick::TickMpcRegs mreg;
    wcqp_tick::tick_mpc_issue(td, j, inst, t, mreg);
    wcqp_tick::tick_mpc_finish<false, EXT>(td, j, inst, live, t, mreg, s_hull[grp]);
}

// Both QPs of a batch of robot-ticks in ONE launch (wcqp_qp_enqueue_steps, a record whose two calls go to the same
// stream): workgroups 0 .. ik_blocks-1 are the IK kernel above, the rest the DCM-MPC kernel of mpc.hip (same device
// functions, same results).  At the BASELINE batch each is one wave per SIMD, so the MPC waves run in the slots the IK
// waves leave idle while their inputs are on the way, and the host pays one launch per step instead of two.
__global__ __launch_bounds__(64, WCQP_IK4_WAVES)
void qp_pair_kernel(const IkDeviceParams* __restrict__ prm, int batch,
                    const double* __restrict__ JL, const double* __restrict__ JR,
                    const double* __restrict__ JN, const double* __restrict__ JC,
                    const double* qpos, const double* __restrict__ state,
                    double* __restrict__ dq_out, int* __restrict__ status_out,
                    unsigned* __restrict__ alo_out, unsigned* __restrict__ aup_out,
                    double* __restrict__ ferr_out, int* __restrict__ iters_out, int ik_blocks, MpcPairArgs m)
{
    __shared__ __attribute__((aligned(16))) double smem[4][PER_INST];
    // IK workgroups first: they are the long ones.  (Alternating the two kinds in dispatch order was measured: the MPC waves
    // then take slots from IK waves that have not started yet - 17.7 vs 14.5 us at 4096 robots, 158 vs 116 us at 65536.)
    const bool is_mpc = (int)blockIdx.x >= ik_blocks;
    const int blk = is_mpc ? (int)blockIdx.x - ik_blocks : xcd_group((int)blockIdx.x, ik_blocks);
    if (is_mpc) {
        static_assert(wcqp_mpc::kInstPerWave * WCQP_HULL_ROWS * 4 <= 4 * PER_INST, "the hull rows fit the IK's LDS");
        double (*s_hull)[WCQP_HULL_ROWS][4] = reinterpret_cast<double (*)[WCQP_HULL_ROWS][4]>(&smem[0][0]);
        const int lane = threadIdx.x;
        const int sub = lane / wcqp_mpc::kLanesPerInstance, t = lane % wcqp_mpc::kLanesPerInstance;
        const long inst_raw = (long)blk * wcqp_mpc::kInstPerWave + sub;
        const bool live = inst_raw < batch;
        const long inst = live ? inst_raw : (long)batch - 1;
        const double2* rp = reinterpret_cast<const double2*>(m.ref) + inst * m.ref_len;
        double ux, uy, margin;
        int st;
        unsigned mask;
        wcqp_mpc::mpc_row_solve(m.c, t, inst, m.x0, rp, m.ref_len, m.u_prev, m.hull_A, m.hull_b, m.hull_nc, inst, s_hull[sub], ux, uy, st, mask, margin);
        if (t == 0 && live) {
            reinterpret_cast<double2*>(m.u0)[inst] = make_double2(ux, uy);
            m.status[inst] = st;
            if (m.active) m.active[inst] = mask;
            if (m.margin) m.margin[inst] = margin;
        }
        return;
    }
    ik4_body<false>(prm, batch, JL, JR, JN, JC, qpos, state, dq_out, status_out, alo_out, aup_out, ferr_out, iters_out, wcqp_tick::TickDev{}, smem, blk);
}

// A PLAN of steps (wcqp_qp_plan_*): every record is one batch of robot-ticks - one DCM-MPC QP and one QP-IK per robot, as
// wcqp_qp_enqueue_steps would launch it - and ONE launch walks through all of them.  Consecutive records are independent
// batches, so nothing orders them: workgroup (way w, robot group g) solves robots 4g .. 4g+3 of records w, w + ways, ... on
// its own - no launch, ramp-up or tail per step, the MPC of a record in the shadow of its IK's Jacobian loads, and with
// `ways` workgroups per robot group the BASELINE batch (1024 robot groups) fills both wave slots of every SIMD.
template <bool WITH_MPC>
__device__ __forceinline__
void plan_walk(const IkDeviceParams* __restrict__ prm, int batch, const wcqp_qp_step* __restrict__ recs, int n_steps, int ways, int groups,
               const wcqp_mpc::MpcDeviceConsts& c, unsigned* queue, double (*smem)[PER_INST])
{
    // ways > 0: workgroup (way, robot group) walks through records way, way + ways, ... of its group.
    // ways = 0, work queues: unit u = (robot group u / n_steps, record u % n_steps), GROUP-major: the resident waves then work inside a
    // window of a few dozen robot groups (record-major at 65536 robots every unit of a wave lies 11 MB further on in each of the
    // twelve input arrays - a new page per array per record: 15 % slower than the fixed ways; group-major it is on a par).  kPlanQueues ticket counters,
    // kPlanQueueStride bytes apart (one counter serves ~8e7 tickets/s - measured: every wave behind ONE counter ran at half the
    // rate of the fixed ways, at every batch size - and the launch needs 1.5e8); ticket k of queue q is unit k kPlanQueues + q.
    // A wave draws from its home queue, asking for the next ticket behind the record's loads (ik4_body; the answer is needed at the
    // bottom: the atomic's round trip rides under the record's arithmetic), and goes round the other queues once its own has run out; it leaves
    // when all of them have.  queue[kPlanQueues * stride] counts the waves that are done: the last one zeroes everything.
    const bool dynamic = ways == 0;
    unsigned z = threadIdx.x;
    __asm__ volatile("" : "+v"(z));
    z >>= 6;          // an opaque per-lane zero in the ticket address: with a wave-uniform address hipcc's atomic optimizer rewrites the add into
                      // "count the lanes, one atomic, s_waitcnt vmcnt(0), redistribute" - a synchronous round trip at the top of every record
    constexpr unsigned QS = wcqp_ik::kPlanQueueStride / 4u;
    const unsigned total = (unsigned)n_steps * (unsigned)groups;
    unsigned home = blockIdx.x % wcqp_ik::kPlanQueues;
    // the next unit of any queue, starting at `home` (synchronous): total when every queue has run out
    auto draw = [&]() -> unsigned {
#ifndef WCQP_PLAN_STEAL
#define WCQP_PLAN_STEAL (wcqp_ik::kPlanQueues - 1u)
#endif
        // (every queue is the HOME of gridDim.x / kPlanQueues waves, which draw from it until it is empty: a queue is drained whether or
        // not anybody else visits it, so how many OTHER queues a wave tries before it leaves is a matter of balance, not of correctness)
        for (unsigned tried = 0; tried <= WCQP_PLAN_STEAL; ++tried) {
            unsigned k = 0;
            if (threadIdx.x == 0) k = __hip_atomic_fetch_add(queue + home * QS + z, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((int)k) * wcqp_ik::kPlanQueues + home;
            if (u < total) return u;
            home = (home + 1u) % wcqp_ik::kPlanQueues;
        }
        return total;
    };
    int r = (int)blockIdx.x / groups, blk = (int)blockIdx.x % groups;
#ifndef WCQP_PLAN_NO_XCD_MAP
    // XCD-aware (xcd_group above): XCD x walks the contiguous eighth x of the robot groups - its k-th workgroup is way k / n, group x n + k mod n.
    // PMC: the launch fetches 3 % fewer bytes (the 4 % by which its traffic exceeded its algorithmic bytes), +0.5 to +1.5 % QP/s
    // (profiles/r04_xcd_map_ab.txt).
    if ((groups & 7) == 0) {
        const int n = groups >> 3, x = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
        r = k / n; blk = x * n + k % n;
    }
#endif
    if (dynamic) {
        const unsigned u = draw();
        r = u < total ? (int)(u % (unsigned)n_steps) : n_steps; blk = u < total ? (int)(u / (unsigned)n_steps) : 0;
    }
#pragma unroll 1
    while (r < n_steps) {
        __asm__ volatile("" ::: "memory");        // nothing of the body is hoisted out of the loop
        // (the record's pointers come out of memory: as_global says what a kernel argument would have said - gptr.h)
        using wcqp::as_global;
        const wcqp_qp_step& s = recs[r];
        MpcPairArgs m{c, WITH_MPC ? as_global(s.x0) : nullptr, WITH_MPC ? as_global(s.ref) : nullptr, s.ref_len, WITH_MPC ? as_global(s.u_prev) : nullptr,
                      WITH_MPC ? as_global(s.hull_A) : nullptr, WITH_MPC ? as_global(s.hull_b) : nullptr, WITH_MPC ? as_global(s.hull_nc) : nullptr,
                      WITH_MPC ? as_global(s.u0) : nullptr, WITH_MPC ? as_global(s.mpc_status) : nullptr, WITH_MPC ? as_global(s.mpc_active) : nullptr,
                      WITH_MPC ? as_global(s.mpc_margin) : nullptr,
                      dynamic ? queue + home * QS + z : nullptr, 0u, WITH_MPC};
        ik4_body<false, 0, true>(prm, batch, as_global(s.J_left), as_global(s.J_right), as_global(s.J_neck), as_global(s.J_com), as_global(s.q), as_global(s.state),
                                 as_global(s.dq), as_global(s.ik_status), as_global(s.active_lower), as_global(s.active_upper),
                                 as_global(s.foot_err), as_global(s.iters), wcqp_tick::TickDev{}, smem, blk, 0, true, nullptr, nullptr, &m);
        wcqp::wave_lds_fence();
        if (dynamic) {
            unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((int)m.ticket) * wcqp_ik::kPlanQueues + home;
            if (u >= total) { home = (home + 1u) % wcqp_ik::kPlanQueues; u = draw(); }
            r = u < total ? (int)(u % (unsigned)n_steps) : n_steps; blk = u < total ? (int)(u / (unsigned)n_steps) : 0;
        } else {
            r += ways;
        }
    }
    if (dynamic && threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(queue + wcqp_ik::kPlanQueues * QS, 1u) == gridDim.x - 1) {
            for (unsigned q = 0; q <= wcqp_ik::kPlanQueues; ++q) queue[q * QS] = 0u;
        }
    }
}

__global__ __launch_bounds__(64, WCQP_IK4_WAVES)
void qp_plan_kernel(const IkDeviceParams* __restrict__ prm, int batch, const wcqp_qp_step* __restrict__ recs, int n_steps, int ways, int groups,
                    wcqp_mpc::MpcDeviceConsts c, unsigned* queue)
{
    __shared__ __attribute__((aligned(16))) double smem[4][PER_INST];
    plan_walk<true>(prm, batch, recs, n_steps, ways, groups, c, queue, smem);
}
// an IK-only plan (no record has an MPC part: BASELINE config 3 on its own): the same walk without the MPC share, under a name of its
// own so that profiles of the two do not mix
__global__ __launch_bounds__(64, WCQP_IK4_WAVES) WCQP_IK_PLAN_REGS
void ik_plan_kernel(const IkDeviceParams* __restrict__ prm, int batch, const wcqp_qp_step* __restrict__ recs, int n_steps, int ways, int groups,
                    wcqp_mpc::MpcDeviceConsts c, unsigned* queue)
{
    __shared__ __attribute__((aligned(16))) double smem[4][PER_INST];
    plan_walk<false>(prm, batch, recs, n_steps, ways, groups, c, queue, smem);
}

}  // namespace

namespace wcqp_ik {

int ik4_plan_queue_grid(int batch, int n_steps) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return -1;
    const long long total = (long long)((batch + 3) / 4) * n_steps, slots = (long long)cus * 4 * WCQP_IK4_WAVES;   // waves that are resident at once
    return (int)(total < slots ? total : slots);
}

int ik4_launch_plan(const IkDeviceParams* d_prm, int batch, const wcqp_qp_step* d_recs, int n_steps, int ways,
                    const wcqp_mpc::MpcDeviceConsts& c, hipStream_t stream, unsigned* d_queue, int queue_grid, bool ik_only) {
    if (!d_prm || !d_recs || batch < 1 || n_steps < 1 || ways < 0 || (ways == 0 && (!d_queue || queue_grid < 1))) return WCQP_E_INVALID;
    const int groups = (batch + 3) / 4;
    if ((long long)groups * n_steps >= (1ll << 31)) return WCQP_E_INVALID;
    const unsigned grid = ways == 0 ? (unsigned)queue_grid : (unsigned)(groups * ways);
    if (ik_only) hipLaunchKernelGGL(ik_plan_kernel, dim3(grid), dim3(64), 0, stream, d_prm, batch, d_recs, n_steps, ways, groups, c, d_queue);
    else hipLaunchKernelGGL(qp_plan_kernel, dim3(grid), dim3(64), 0, stream, d_prm, batch, d_recs, n_steps, ways, groups, c, d_queue);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

int ik4_launch_pair(const IkDeviceParams* d_prm, int batch,
                    const double* JL, const double* JR, const double* JN, const double* JC,
                    const double* q, const double* state, double* dq, int* status,
                    unsigned* alo, unsigned* aup, double* ferr, int* iters,
                    const wcqp_mpc::MpcDeviceConsts& c, const double* x0, const double* ref, int ref_len, const double* u_prev,
                    const double* hull_A, const double* hull_b, const int* hull_nc,
                    double* u0, int* mstatus, unsigned* mactive, double* mmargin, hipStream_t stream) {
    const int ik_blocks = (batch + 3) / 4;
    const int mpc_blocks = (batch + wcqp_mpc::kInstPerWave - 1) / wcqp_mpc::kInstPerWave;
    MpcPairArgs m{c, x0, ref, ref_len, u_prev, hull_A, hull_b, hull_nc, u0, mstatus, mactive, mmargin};
    hipLaunchKernelGGL(qp_pair_kernel, dim3((unsigned)(ik_blocks + mpc_blocks)), dim3(64), 0, stream, d_prm, batch, JL, JR, JN, JC, q, state,
                       dq, status, alo, aup, ferr, iters, ik_blocks, m);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

int ik4_launch(const IkDeviceParams* d_prm, int batch,
               const double* JL, const double* JR, const double* JN, const double* JC,
               const double* q, const double* state, double* dq, int* status,
               unsigned* alo, unsigned* aup, double* ferr, int* iters, hipStream_t stream) {
    const unsigned grid = (unsigned)((batch + 3) / 4);
    hipLaunchKernelGGL((ik4_kernel<false, 0>), dim3(grid), dim3(64), 0, stream, d_prm, batch, JL, JR, JN, JC, q, state,
                       dq, status, alo, aup, ferr, iters, nullptr, 0, 1, 0);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

int ik4_launch_tick(const void* d_prm, const wcqp_tick::TickDev& td, const wcqp_tick::TickDev* td_dev,
                    const double* JL, const double* JR, const double* JN, const double* JC,
                    unsigned* alo, unsigned* aup, int n_inner, int skip_last_mpc, hipStream_t stream, double* log_ferr) {
    if (n_inner < 1 || (n_inner > 1 && td.kin_mode && !td.kin_fused)) return WCQP_E_INVALID;      // kinematics in a launch of their own: the Jacobians of tick t + 1 come from another launch
    if (!d_prm || !td_dev || !td.skew || !td.mst || !td.hand || !td.live_A || !td.live_b || !td.live_nc || !td.sel_built) return WCQP_E_INVALID;
    if (td.compact && (!td.jcomp || td.cstride < 1)) return WCQP_E_INVALID;
    if (td.kin_fused && (!td.kin_tab || !td.kin_mode || td.kin_rounds < 0 || td.kin_rounds > 3 || td.horizon >= wcqp_tick::kGainsLdsStages)) return WCQP_E_INVALID;
    const unsigned grid = (unsigned)((td.batch + 3) / 4);
    const IkDeviceParams* prm = static_cast<const IkDeviceParams*>(d_prm);
    if (td.logger_ticks > 0) {
        // the logging kernels (a debugging aid like the reference's dumpData): dense Jacobians also produce the foot errors
        if (!td.log_rows) return WCQP_E_INVALID;
        if (td.kin_fused)
            hipLaunchKernelGGL((ik4_kernel<true, 2, true>), dim3(grid), dim3(64), 0, stream, prm, td.batch,
                               JL, JR, JN, JC, td.q_des, td.state, td.dq, td.ik_status, alo, aup, nullptr, nullptr, td_dev, td.phase, n_inner, skip_last_mpc);
        else if (td.compact)
            hipLaunchKernelGGL((ik4_kernel<true, 1, true>), dim3(grid), dim3(64), 0, stream, prm, td.batch,
                               JL, JR, JN, JC, td.q_des, td.state, td.dq, td.ik_status, alo, aup, nullptr, nullptr, td_dev, td.phase, n_inner, skip_last_mpc);
        else
            hipLaunchKernelGGL((ik4_kernel<true, 0, true>), dim3(grid), dim3(64), 0, stream, prm, td.batch,
                               JL, JR, JN, JC, td.q_des, td.state, td.dq, td.ik_status, alo, aup, log_ferr, nullptr, td_dev, td.phase, n_inner, skip_last_mpc);
        WCQP_HIP_TRY(hipGetLastError());
        return WCQP_OK;
    }
    if (td.q_meas) {
        // external feedback (wcqp_tick_params.plant = EXTERNAL): measured joints in the IK's regularisation; one tick per launch
        if (n_inner != 1 || td.compact) return WCQP_E_INVALID;
        if (td.kin_fused)
            hipLaunchKernelGGL((ik4_kernel<true, 2, false, true>), dim3(grid), dim3(64), 0, stream, prm, td.batch,
                               JL, JR, JN, JC, td.q_des, td.state, td.dq, td.ik_status, alo, aup, nullptr, nullptr, td_dev, td.phase, n_inner, skip_last_mpc);
        else
            hipLaunchKernelGGL((ik4_kernel<true, 0, false, true>), dim3(grid), dim3(64), 0, stream, prm, td.batch,
                               JL, JR, JN, JC, td.q_des, td.state, td.dq, td.ik_status, alo, aup, nullptr, nullptr, td_dev, td.phase, n_inner, skip_last_mpc);
        WCQP_HIP_TRY(hipGetLastError());
        return WCQP_OK;
    }
    if (td.kin_fused)
        hipLaunchKernelGGL((ik4_kernel<true, 2>), dim3(grid), dim3(64), 0, stream, prm, td.batch,
                           JL, JR, JN, JC, td.q_des, td.state, td.dq, td.ik_status, alo, aup, nullptr, nullptr, td_dev, td.phase, n_inner, skip_last_mpc);
    else if (td.compact)
        hipLaunchKernelGGL((ik4_kernel<true, 1>), dim3(grid), dim3(64), 0, stream, prm, td.batch,
                           JL, JR, JN, JC, td.q_des, td.state, td.dq, td.ik_status, alo, aup, nullptr, nullptr, td_dev, td.phase, n_inner, skip_last_mpc);
    else
        hipLaunchKernelGGL((ik4_kernel<true, 0>), dim3(grid), dim3(64), 0, stream, prm, td.batch,
                           JL, JR, JN, JC, td.q_des, td.state, td.dq, td.ik_status, alo, aup, nullptr, nullptr, td_dev, td.phase, n_inner, skip_last_mpc);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

// MPC chain of tick t alone (see tick_mpc_prime_kernel)
int ik4_launch_tick_prime(const wcqp_tick::TickDev& td, int t, hipStream_t stream) {
    if (!td.skew || !td.mst || !td.hand) return WCQP_E_INVALID;
    const unsigned grid = (unsigned)((td.batch + 3) / 4);
    if (td.q_meas) hipLaunchKernelGGL(tick_mpc_prime_kernel<true>, dim3(grid), dim3(64), 0, stream, td, t);      // external feedback: the caller's measured ZMP
    else hipLaunchKernelGGL(tick_mpc_prime_kernel<false>, dim3(grid), dim3(64), 0, stream, td, t);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

}  // namespace wcqp_ik
