/*
 * wcqp.h — C ABI of the MI355X-native batched QP solve path that replaces the
 * osqp-eigen / qpOASES calls behind the reference's WalkingController
 * (DCM-MPC) and WalkingQPIK (Jacobian QP-IK) solver interfaces.
 *
 * Citations are relative to /root/reference/modules/Walking_module ("WM/").
 *
 * The reference has no FFI: its boundary is two C++ class surfaces
 * (WM/include/WalkingDCMModelPredictiveController.hpp:190-250 with
 * WM/include/MPCSolver.hpp:57-138, and WM/include/WalkingQPInverseKinematics.hpp:81-208).
 * Every entry point below names the reference call(s) it stands in for.  Batch = 1
 * reproduces the per-robot call; batch = B solves B independent robot instances.
 *
 * Conventions
 *   - plain C, no exceptions cross the ABI; every function returns WCQP_OK (0) or
 *     a negative WCQP_E_* code; per-instance results carry a WCQP_STATUS_* code
 *     (the batch analogue of the reference's `bool` returns).
 *   - all arithmetic is IEEE fp64 like the reference.
 *   - `*_device` entry points take DEVICE pointers and a hipStream_t (as void*); they
 *     enqueue work and return without synchronising (graph-capturable: no
 *     allocation, no sync inside).  `*_host` entry points take HOST pointers,
 *     stage through the handle's own device buffers and synchronise.
 *   - the handle owns every buffer it allocates; the caller owns inputs/outputs.
 *     A solver handle (wcqp_mpc_t, wcqp_ik_t, wcqp_kin_t) and wcqp_qp_enqueue_steps retain no caller pointer past a
 *     call.  Two objects DO, by design: a wcqp_qp_plan_t keeps the device pointers of every record it was created
 *     from, and both solver handles, until wcqp_qp_plan_destroy (the caller keeps those arrays and handles alive and
 *     their addresses unchanged for as long as the plan may be enqueued); a wcqp_tick_t owns its whole robot state on
 *     the device and copies HOST inputs at the call that receives them (wcqp_tick_upload,
 *     wcqp_tick_splice_reference: the host array may be released when the call returns).  On the HOST side a handle is single-caller
 *     (like the reference's solvers, which are only touched under WalkingModule's m_mutex,
 *     WM/src/WalkingModule.cpp:429): one thread at a time calls into it.  On the DEVICE
 *     side the work a `*_device` call (or wcqp_qp_enqueue_steps) enqueues only READS the
 *     handle's device state (constants uploaded at create / wcqp_ik_set_posture), so solves
 *     of one wcqp_mpc_t / wcqp_ik_t pair enqueued on several streams may run concurrently -
 *     the caller keeps their input / output arrays apart.  (wcqp_ik_set_posture and the
 *     `*_host` entry points synchronise the device first; a wcqp_tick_t owns its state and
 *     takes one stream at a time.)
 */
#ifndef WCQP_H
#define WCQP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WCQP_VERSION 400

/* return codes */
#define WCQP_OK              0
#define WCQP_E_INVALID      (-1)   /* bad argument / NULL pointer                      */
#define WCQP_E_UNSUPPORTED  (-2)   /* size outside what the kernels are built for      */
#define WCQP_E_NUMERIC      (-3)   /* host-side constant precomputation failed         */
#define WCQP_E_HIP          (-4)   /* HIP runtime error (no device, launch failure...) */
#define WCQP_E_NOMEM        (-5)

/* per-instance status */
#define WCQP_STATUS_SOLVED        0
#define WCQP_STATUS_MAX_ITER      1   /* active-set iteration budget exhausted            */
#define WCQP_STATUS_INFEASIBLE    2   /* constraints admit no point                       */
#define WCQP_STATUS_OUTSIDE_HULL  3   /* MPC: margin(u0) < -convex_hull_tolerance
                                         (WM/src/WalkingDCMModelPredictiveController.cpp:513-517) */
#define WCQP_STATUS_NUMERIC       4   /* non-positive pivot (KKT not regular)             */
#define WCQP_STATUS_STRUCTURE     5   /* IK: handle created with WCQP_IK_JAC_MIXED, but this instance's Jacobians do
                                         not have MIXED-representation base blocks          */

#define WCQP_HULL_ROWS   8            /* hull rows are padded to 8 per instance           */
#define WCQP_MAX_DOF     32
#define WCQP_IK_STATE_LEN 87          /* doubles in the packed per-instance pose block    */

const char* wcqp_strerror(int code);
int wcqp_version(void);
/* number of visible HIP devices (0 on a CPU-only host; never initialises a context) */
int wcqp_device_count(void);
/* HIP streams for hosts that reach this library through an FFI and have no HIP binding of their own (ctypes, cgo, JNI):
 * what the `stream` argument of the `*_device` entry points takes.  Non-blocking streams of the current device;
 * wcqp_stream_synchronize(NULL) waits for the whole device.  (The reference has no counterpart: its solvers are synchronous.) */
int wcqp_stream_create(void** out);
int wcqp_stream_destroy(void* stream);
int wcqp_stream_synchronize(void* stream);

/* =====================================================================================
 * DCM-MPC  — replaces  WalkingController::{initialize, setConvexHullConstraint,
 *            setFeedback, setReferenceSignal, solve, getControllerOutput}
 *            (WM/src/WalkingDCMModelPredictiveController.cpp:311-535) and the
 *            MPCSolver / OsqpEigen::Solver object they drive (WM/src/MPCSolver.cpp:16-322).
 * ===================================================================================== */
typedef struct wcqp_mpc_params {
    int32_t horizon;              /* N = round(controllerHorizon / sampling_time), cpp:182-187 */
    double  sampling_time;        /* cpp:182                                                   */
    double  com_height;           /* cpp:224-229                                               */
    double  gravity;              /* cpp:230 (default 9.81)                                    */
    double  Q[4];                 /* state weight, row-major 2x2 (stateWeightTriplets)         */
    double  R[4];                 /* input weight, row-major 2x2 (inputWeightTriplets)         */
    double  convex_hull_tolerance;/* cpp:306                                                   */
    double  feas_tol;             /* feasibility slack of a candidate vertex; 0 -> 1e-10       */
} wcqp_mpc_params;

typedef struct wcqp_mpc_s* wcqp_mpc_t;

/* WalkingController::initialize/initializeMatrices (cpp:170-243, 311-362): builds P,
 * A_eq and the gradient sub-matrix, then condenses the batch-constant equality KKT
 * once on the host (u0 = sum_i Gr_i r_i + Gx x0 + Gu u_prev - Sigma0 A_h' mu). */
int wcqp_mpc_create(const wcqp_mpc_params* params, wcqp_mpc_t* out);
int wcqp_mpc_destroy(wcqp_mpc_t h);

/* Host-side introspection of the condensed constants (tests):
 * Gr[(N+1)*4], Gx[4], Gu[4], Sigma0[4], all row-major 2x2 blocks. */
int wcqp_mpc_get_condensed(wcqp_mpc_t h, double* Gr, double* Gx, double* Gu, double* Sigma0);
/* Dense copies of the reference's constant blocks for parity checks against
 * initializeMatrices: P[n*n], A_eq[n_x*n], grad_sub[n_u*2]; n = 4N+2. */
int wcqp_mpc_get_matrices(wcqp_mpc_t h, double* P, double* A_eq, double* grad_sub);

/*
 * One MPC tick for `batch` instances — the contents of the reference's "MPC" profiler
 * bracket (WM/src/WalkingModule.cpp:604-636):
 *   hull_A[B][8][2], hull_b[B][8], hull_nc[B]  <- setConvexHullConstraint / MPCSolver::setConstraintsMatrix
 *                                                 + the hull part of setBounds (MPCSolver.cpp:76-123,149-153);
 *                                                 rows >= hull_nc[i] are ignored
 *   x0[B][2]                                   <- setFeedback / setBounds rows 0..1 (MPCSolver.cpp:143-146)
 *   ref[B][ref_len][2], ref_len >= 1           <- setReferenceSignal / setGradient (MPCSolver.cpp:183-239);
 *                                                 stages >= ref_len repeat the last one (:200-214)
 *   u_prev[B][2]                               <- m_output fed back as previousControllerOutput (:244-245)
 * outputs
 *   u0[B][2]      first input = desired ZMP    -> getControllerOutput (cpp:510-511, 523-535)
 *   status[B]     WCQP_STATUS_*                -> the bool of solve() (cpp:491-521)
 *   active[B]     bit e set <=> hull row e is in the optimal active set       (may be NULL)
 *   margin[B]     signed distance of u0 to the hull boundary, + inside         (may be NULL)
 */
int wcqp_mpc_solve_device(wcqp_mpc_t h, int32_t batch,
                          const double* x0, const double* ref, int32_t ref_len,
                          const double* u_prev,
                          const double* hull_A, const double* hull_b, const int32_t* hull_nc,
                          double* u0, int32_t* status, uint32_t* active, double* margin,
                          void* stream);
int wcqp_mpc_solve_host(wcqp_mpc_t h, int32_t batch,
                        const double* x0, const double* ref, int32_t ref_len,
                        const double* u_prev,
                        const double* hull_A, const double* hull_b, const int32_t* hull_nc,
                        double* u0, int32_t* status, uint32_t* active, double* margin);

/* =====================================================================================
 * Support polygon rows from foot poses — the batch analogue of
 * WalkingController::setConvexHullConstraint -> buildConvexHull
 * (WM/src/WalkingDCMModelPredictiveController.cpp:364-489): feet rectangles
 * (foot_size, cpp:295-303) transformed by the foot poses, projected on the XY plane, 2-D
 * convex hull.  iDynTree's row order / normalisation is upstream and unpinned, so the
 * convention is this library's own: CCW hull, unit outward normals,
 * rows a.u <= b, padded to 8 rows with 0.u <= 1e30.
 *   foot_rect[8]            corners (x, y) x 4 of the foot rectangle in the foot frame
 *   left_T / right_T[B][12] foot-to-world transforms: position (3) + row-major rotation (9)
 *   contact[B]              bit 0 = left foot in contact, bit 1 = right foot in contact
 * outputs hull_A[B][8][2], hull_b[B][8], hull_nc[B] (0 when no foot is in contact — the
 * reference refuses that case, cpp:406-410; the MPC kernel then reports the unconstrained u0).
 * ===================================================================================== */
int wcqp_hull_from_feet_device(int32_t batch, const double* foot_rect,
                               const double* left_T, const double* right_T, const uint8_t* contact,
                               double* hull_A, double* hull_b, int32_t* hull_nc, void* stream);
int wcqp_hull_from_feet_host(int32_t batch, const double* foot_rect,
                             const double* left_T, const double* right_T, const uint8_t* contact,
                             double* hull_A, double* hull_b, int32_t* hull_nc);

/* =====================================================================================
 * QP-IK — replaces WalkingQPIK_osqp::solve / WalkingQPIK_qpOASES::solve and the
 *         OsqpEigen::Solver / qpOASES::SQProblem objects behind them
 *         (WM/src/WalkingQPInverseKinematics_osqp.cpp:340-428,
 *          WM/src/WalkingQPInverseKinematics_qpOASES.cpp:284-362).
 * ===================================================================================== */
#define WCQP_IK_FORM_QPOASES 0   /* bounds enforced, kappa = 1, feet always corrected        */
#define WCQP_IK_ALG_DEFAULT   0
#define WCQP_IK_ALG_SWEEP     1        /* round-1 A/B baseline; diagnostic builds only (-DWCQP_DIAG_KERNELS), else WCQP_E_UNSUPPORTED */
#define WCQP_IK_ALG_NULLSPACE 2        /* null-space kernel, reduced Hessian on the fp64 VALU                 */
#define WCQP_IK_ALG_NULLSPACE_MFMA 3   /* same, reduced-Hessian Gram product as one v_mfma_f64_16x16x4 tile per
                                          instance: ~5 % faster than 2 in interleaved A/B runs;
                                          needs use_com_as_constraint (one 16x16 tile), else runs as 2        */
#define WCQP_IK_ALG_NULLSPACE_16L 4    /* null-space kernel on 16 lanes per instance, 4 instances per wave
                                          (csrc/ik3.hip): 1.6x the throughput of 3 on a full chip; the fall-back of
                                          algorithm 5 for Jacobians without the MIXED pattern and the kernel of
                                          WCQP_IK_JAC_GENERAL; needs use_com_as_constraint, else runs as 2          */
#define WCQP_IK_ALG_BASE_ELIM 5        /* base unknowns eliminated in closed form through the left-foot rows, 23-variable
                                          QP in range space (csrc/ik4.hip): what the default resolves to when the CoM is a
                                          constraint, every joint weight is > 0 and the neck weight is positive definite;
                                          otherwise runs as 4.  See `jacobian_structure`.                        */
/* jacobian_structure: what the caller promises about the base blocks (columns 0..5) of the four Jacobians.
 * The reference always passes iDynTree free-floating Jacobians in MIXED representation
 * (WM/src/WalkingForwardKinematics.cpp:33, 436-454): J_left/J_right = [I B; 0 I | .], J_com = [I B | .],
 * J_neck (angular rows) = [0 I | .].  Algorithm 5 relies on that pattern and CHECKS it per instance: for each of the six base
 * columns the absolute deviations of its identity / zero entries from 1.0 / 0.0, summed over the four Jacobians, must not exceed
 * WCQP_IK_MIXED_TOL (a producer that forms the blocks through rotation products hands over 0.9999999999999999; such entries are
 * then TREATED as exact, which moves the solution by at most WCQP_IK_MIXED_TOL x |base velocity|).  Beyond the tolerance - or with
 * a NaN in a base block - the instance does not have the pattern. */
#define WCQP_IK_MIXED_TOL 1e-12
#define WCQP_IK_JAC_AUTO    0    /* default: instances without the pattern are re-solved by the general kernel (one
                                    more, nearly empty, launch per call)                                          */
#define WCQP_IK_JAC_MIXED   1    /* instances without the pattern come back WCQP_STATUS_STRUCTURE (dq = 0); one launch */
#define WCQP_IK_JAC_GENERAL 2    /* arbitrary Jacobians: the general kernel (algorithm 4) only                     */
#define WCQP_IK_FORM_OSQP    1   /* joint-limit rows are zero rows (never bind), extra
                                    k_attFoot on the neck gradient term, zero-twist rule
                                                               */

typedef struct wcqp_ik_params {
    int32_t dof;                         /* actuated DoF (23 on iCub); n = dof + 6            */
    int32_t use_com_as_constraint;       /* qpInverseKinematics.ini:2                          */
    int32_t form;                        /* WCQP_IK_FORM_*                                     */
    int32_t max_iter;                    /* active-set changes budget; 0 -> 100 (nWSR, qp.cpp:312) */
    double  com_weight[9];               /* row-major 3x3, used when !use_com_as_constraint    */
    double  neck_weight[9];              /* row-major 3x3                                      */
    double  joint_reg_weights[WCQP_MAX_DOF];
    double  joint_reg_gains[WCQP_MAX_DOF];
    double  joint_reg_rad[WCQP_MAX_DOF]; /* jointRegularization already in rad (osqp.cpp:101-102) */
    double  v_min[WCQP_MAX_DOF];         /* joint velocity limits (WalkingModule.cpp:237-243)  */
    double  v_max[WCQP_MAX_DOF];
    double  k_pos_com, k_pos_foot, k_att_foot, k_neck;
    double  rho;                         /* weight of the A'A term that regularises H; 0 -> 1  */
    double  tol;                         /* bound-violation tolerance; 0 -> 1e-12              */
    int32_t algorithm;                   /* WCQP_IK_ALG_*: 0 -> default (5; 2 for CoM-as-cost), 1 = sweep on H + rho A'A (csrc/ik.hip),
                                            2 / 3 = null-space (csrc/ik2.hip) without / with MFMA, 4 = null-space on
                                            16 lanes per instance (csrc/ik3.hip), 5 = base elimination + range space
                                            (csrc/ik4.hip); same optimum */
    int32_t jacobian_structure;          /* WCQP_IK_JAC_* (algorithms 0 / 5 only)               */
} wcqp_ik_params;

typedef struct wcqp_ik_s* wcqp_ik_t;

/* WalkingQPIK_*::initialize + WalkingQPIK::initializeMatrices
 * (osqp.cpp:54-133, qp.cpp:53-133, WalkingQPInverseKinematics.cpp:25-116). */
int wcqp_ik_create(const wcqp_ik_params* params, wcqp_ik_t* out);
int wcqp_ik_destroy(wcqp_ik_t h);
/* WalkingQPIK::setDesiredJointPosition (WalkingQPInverseKinematics.cpp:246-256): replaces the regularisation posture
 * (rad, `dof` entries) that enters the gradient of every later solve (osqp.cpp:185, qp.cpp:166).  Synchronises the
 * device; not graph-capturable. */
int wcqp_ik_set_posture(wcqp_ik_t h, const double* joint_reg_rad);

/*
 * One IK tick for `batch` instances — the solver part of the reference's "IK" bracket
 * (WM/src/WalkingModule.cpp:367-425, 721-740).  n = dof + 6, all matrices row-major:
 *   J_left[B][6][n], J_right[B][6][n]   <- setLeftFootJacobian / setRightFootJacobian
 *   J_neck[B][3][n]                     <- rows 3..5 of the 6 x n neck Jacobian, as kept by
 *                                          setNeckJacobian (WalkingQPInverseKinematics.cpp:214-215)
 *   J_com[B][3][n]                      <- setCoMJacobian
 *   q[B][dof]                           <- setRobotState joint positions
 *   state[B][87]                        <- setRobotState / setDesired* poses, packed:
 *        p_left 0..2 | R_left 3..11 | p_right 12..14 | R_right 15..23
 *        pd_left 24..26 | Rd_left 27..35 | pd_right 36..38 | Rd_right 39..47
 *        R_neck 48..56 | Rd_neck 57..65 (already multiplied by additional_rotation, cpp:143-146)
 *        com 66..68 | com_des 69..71 | com_vel_des 72..74 | twist_left 75..80 | twist_right 81..86
 * outputs
 *   dq[B][dof]                          -> getSolution (osqp.cpp:410-428, qp.cpp:341-362)
 *   status[B]                           -> the bool of solve()
 *   active_lower[B], active_upper[B]    bit i set <=> joint i sits on its lower/upper velocity
 *                                       limit with a positive multiplier          (may be NULL)
 *   foot_err[B][12]                     -> getLeftFootError | getRightFootError
 *                                          (osqp.cpp:430-454, qp.cpp:364-401)     (may be NULL)
 *   iters[B]                            active-set changes performed              (may be NULL)
 */
int wcqp_ik_solve_device(wcqp_ik_t h, int32_t batch,
                         const double* J_left, const double* J_right,
                         const double* J_neck, const double* J_com,
                         const double* q, const double* state,
                         double* dq, int32_t* status,
                         uint32_t* active_lower, uint32_t* active_upper,
                         double* foot_err, int32_t* iters,
                         void* stream);
int wcqp_ik_solve_host(wcqp_ik_t h, int32_t batch,
                       const double* J_left, const double* J_right,
                       const double* J_neck, const double* J_com,
                       const double* q, const double* state,
                       double* dq, int32_t* status,
                       uint32_t* active_lower, uint32_t* active_upper,
                       double* foot_err, int32_t* iters);

/* =====================================================================================
 * Several solve calls in one host call.  A robot-tick batch at the BASELINE size is two kernels of
 * about 5 and 15 us, which is what their two launches cost a host that reaches this library through
 * an FFI (ctypes, cgo, JNI): wcqp_qp_enqueue_steps takes an array of argument records and makes, for
 * each record in order, exactly the calls
 *     wcqp_mpc_solve_device(mpc, batch, <the record's MPC arguments>, mpc_stream)
 *     wcqp_ik_solve_device (ik,  batch, <the record's IK arguments>,  ik_stream)
 * (a record whose x0 is NULL skips the MPC call, one whose J_left is NULL the IK call).  Same
 * streams, same results; a record whose two calls name the SAME stream is enqueued as one launch
 * whose workgroups split between the two problems (the IK handle's default kernel permitting) - the
 * MPC waves then run in the slots the IK waves leave idle while they wait for their inputs.  The
 * first failing call's code is returned and *n_done (may be NULL) says how many records were
 * enqueued completely.
 */
typedef struct wcqp_qp_step {
    /* wcqp_mpc_solve_device */
    const double* x0; const double* ref; int32_t ref_len; const double* u_prev;
    const double* hull_A; const double* hull_b; const int32_t* hull_nc;
    double* u0; int32_t* mpc_status; uint32_t* mpc_active; double* mpc_margin; void* mpc_stream;
    /* wcqp_ik_solve_device */
    const double* J_left; const double* J_right; const double* J_neck; const double* J_com;
    const double* q; const double* state;
    double* dq; int32_t* ik_status; uint32_t* active_lower; uint32_t* active_upper;
    double* foot_err; int32_t* iters; void* ik_stream;
} wcqp_qp_step;
int wcqp_qp_enqueue_steps(wcqp_mpc_t mpc, wcqp_ik_t ik, int32_t batch,
                          int32_t n_steps, const wcqp_qp_step* steps, int32_t* n_done);

/* A PLAN of steps: the records of wcqp_qp_enqueue_steps, uploaded once and replayed as ONE launch.  Consecutive steps of the
 * BASELINE workload are independent cold-start batches, so nothing has to order them on the device: a wavefront owns four
 * robots and walks through the records on its own - no launch, ramp-up or tail per step, the MPC of a record solved on the
 * IK's lanes while its Jacobians are in flight - and `ways` wavefronts share a robot group, way w taking records w, w + ways,
 * ... (two ways fill both wave slots of every SIMD at the BASELINE batch of 4096).  Records of DIFFERENT ways run
 * concurrently: they must not share output arrays (give each way its own, like the pipelines of separate streams).
 * ways = WCQP_PLAN_WAYS_AUTO picks the number of ways (enough workgroups for the hardware's dispatcher to even out the launch's ends: 16 at
 * 4096 robots, 4 from 16384 on, never more than one per record); every record then needs output arrays of its own.
 * ways = 0 is the WORK-QUEUE form: the launch has as many wavefronts as are resident at once (2 per SIMD), and each takes the
 * next (record, robot group) unit - robot-group-major - from a device-side queue when it is done with one, so that no wave slot
 * idles while another wavefront still has records left (the tail of the fixed ways).  Any two records may then be in flight
 * together and in any order: NO two records of the plan may share an output array.  The queue is re-armed by the launch itself;
 * launches of ONE plan must be ordered (one stream at a time), different plans are independent.
 * A plan in which NO record has an IK part (J_left == NULL in every record; `ik` may then be NULL, ways >= 1) is an MPC-only plan -
 * BASELINE config 2 on its own: one launch walks through the DCM-MPC batches; one in which NO record has an MPC part (x0 == NULL in
 * every record) is an IK-only plan - config 3 on its own.  Otherwise every
 * record needs both parts; the stream fields of the records are ignored (wcqp_qp_plan_enqueue names the stream).
 * WCQP_E_UNSUPPORTED unless the IK handle runs its default kernel with jacobian_structure = WCQP_IK_JAC_MIXED: use
 * wcqp_qp_enqueue_steps then.  Same results as the single calls, bit for bit.  The arrays the
 * records point to must stay valid while the plan is used; the handles must outlive the plan. */
#define WCQP_PLAN_WAYS_AUTO (-1)
typedef struct wcqp_qp_plan_s* wcqp_qp_plan_t;
int wcqp_qp_plan_create(wcqp_mpc_t mpc, wcqp_ik_t ik, int32_t batch, int32_t n_steps, const wcqp_qp_step* steps, int32_t ways,
                        wcqp_qp_plan_t* out);
int wcqp_qp_plan_enqueue(wcqp_qp_plan_t plan, void* stream);      /* enqueue only; graph-capturable */
int wcqp_qp_plan_destroy(wcqp_qp_plan_t plan);

/* =====================================================================================
 * Shard slabs: the exchange format of the multi-GPU path (rank 0 scatters the inputs of every rank's block of
 * robots, the ranks solve, rank 0 gathers the solutions).  All input arrays of a block live in ONE contiguous device buffer -
 *     [x0 | ref | u_prev | hull_A | hull_b | hull_nc | J_left | J_right | J_neck | J_com | q | state]
 * each in the layout of wcqp_mpc_solve_device / wcqp_ik_solve_device, every array starting on a 256-byte boundary - and all
 * outputs in another - [u0 | mpc_margin | dq | mpc_status | mpc_active | ik_status | active_lower | active_upper | iters] - so
 * that ONE ncclScatter (ncclSend / ncclRecv per peer) and ONE ncclGather move a step's data whatever the backend, and the solve
 * kernels read the received bytes where they landed: wcqp_qp_step_from_slabs fills a step record with pointers INTO the two
 * slabs (no unpack, no copy).  The reference has no counterpart (one robot per process, WM/include/WalkingModule.hpp:65-77).
 * Pure host arithmetic: no device call, usable from any host language next to RCCL. */
#define WCQP_SLAB_IN_ARRAYS  12
#define WCQP_SLAB_OUT_ARRAYS 9
typedef struct wcqp_slab_layout {
    int32_t batch, ref_len;
    int64_t in_offset[WCQP_SLAB_IN_ARRAYS];    /* byte offsets, in the order listed above */
    int64_t in_bytes;                          /* size of an input slab (a multiple of 256)  */
    int64_t out_offset[WCQP_SLAB_OUT_ARRAYS];
    int64_t out_bytes;
} wcqp_slab_layout;
int wcqp_slab_layout_for(int32_t batch, int32_t ref_len, wcqp_slab_layout* out);
/* step <- pointers into the slabs (streams NULL; foot_err NULL).  in_slab / out_slab: device addresses of buffers of at least
 * in_bytes / out_bytes, at least 16-byte aligned (every hipMalloc is 256-byte aligned). */
int wcqp_qp_step_from_slabs(const wcqp_slab_layout* layout, const void* in_slab, void* out_slab, wcqp_qp_step* step);

/* =====================================================================================
 * Batched kinematics: forward kinematics of a kinematic tree and the free-floating
 * Jacobians in MIXED representation that the QP-IK consumes - what the reference obtains from
 * iDynTree::KinDynComputations through WalkingFK:
 *   setInternalRobotState        WM/src/WalkingForwardKinematics.cpp:258-276
 *   get{Left,Right}FootToWorldTransform, getNeckOrientation, getCoMPosition   :312-340, 354-366, 402-405
 *   get{Left,Right}FootJacobian, getNeckJacobian, getCoMJacobian              :436-454 (MIXED, :33)
 * The robot model of the reference is an external URDF (WalkingModule.cpp:107) that is not in the
 * repository: the tree comes in as a table.  Joint j has a parent joint (-1 = root link, always < j), a
 * fixed transform (R0, p0) from the parent joint frame to its own frame at q = 0, a unit axis in its own
 * frame and carries one link; three frames are attached: left sole, right sole, neck.
 * Generalised velocity: (v of the base origin in world, omega of the base in world, dq).
 * Outputs use the batch layouts of wcqp_ik_solve_*: J_left/J_right [B][6][6+dof] (linear rows, then
 * angular), J_neck [B][3][6+dof] (angular rows), J_com [B][3][6+dof]; if `state` is given, the ACTUAL
 * poses are written into the packed pose block (foot positions/rotations, neck rotation, CoM position:
 * offsets 0..23, 48..56, 66..68), the desired entries are left alone.
 * ===================================================================================== */
#define WCQP_KIN_MAX_DOF 32
typedef struct wcqp_kin_params {
    int32_t dof;                        /* 6 + dof <= 32                                          */
    int32_t parent[WCQP_KIN_MAX_DOF];   /* parent joint, -1 = root link; parent[j] < j             */
    double  R0[WCQP_KIN_MAX_DOF][9];    /* row-major                                               */
    double  p0[WCQP_KIN_MAX_DOF][3];
    double  axis[WCQP_KIN_MAX_DOF][3];  /* in the joint's own frame; normalised at create          */
    double  mass[WCQP_KIN_MAX_DOF];     /* link carried by joint j                                 */
    double  com[WCQP_KIN_MAX_DOF][3];   /* its centre of mass in the joint frame                   */
    double  root_mass, root_com[3];
    int32_t frame_joint[3];             /* left sole, right sole, neck: joint the frame is fixed to */
    double  frame_R[3][9], frame_p[3][3];
} wcqp_kin_params;

typedef struct wcqp_kin_s* wcqp_kin_t;

int wcqp_kin_create(const wcqp_kin_params* params, wcqp_kin_t* out);
int wcqp_kin_destroy(wcqp_kin_t h);
/* base [B][12] = position, row-major rotation of the root link; q [B][dof].  DEVICE pointers. */
int wcqp_kin_jacobians_device(wcqp_kin_t h, int32_t batch, const double* base, const double* q,
                              double* J_left, double* J_right, double* J_neck, double* J_com,
                              double* state /* [B][87] or NULL */, void* stream);
/* same with HOST pointers (copies in and out; `state` is read-modify-write) */
int wcqp_kin_jacobians_host(wcqp_kin_t h, int32_t batch, const double* base, const double* q,
                            double* J_left, double* J_right, double* J_neck, double* J_com, double* state);

/* =====================================================================================
 * Device-resident tick pipeline — BASELINE configs 4/5 and SURVEY.md §8f-1/2: the call
 * order of WalkingModule::updateModule around the two solvers (WM/src/WalkingModule.cpp:
 * 578-745) for a batch of synthetic robots, kept entirely on the GPU:
 *   MPC   hull rows of this tick's contact pair (= setConvexHullConstraint; the kernel indexes one
 *         of the three precomputed row sets, so a contact change costs nothing),
 *         window [t, t+N] of the per-instance DCM reference trajectory (the deque that
 *         advances one stage per tick, WalkingModule.cpp:35-96), x0 = measured DCM,
 *         u_prev = previous output (MPCSolver.cpp:244-245)
 *   glue  LIPM reference (StableDCMModel.cpp:63-90), ZMP-CoM law + integrator
 *         (WalkingZMPController.cpp:146-173) -> desired CoM position /
 *         velocity into the IK pose block (WalkingModule.cpp:686-695); synthetic LIPM plant
 *   IK    joint velocities
 *   post  q <- Integrator(dq) (WalkingModule.cpp:741-744), contact pair of the next tick, tick += 1
 * With the default IK kernel (base elimination) a tick is ONE kernel - forward kinematics (use_kinematics), the MPC chain, the glue,
 * the IK and the post step on the 16 lanes a robot's IK runs on - and, the robots of a wavefront depending on no other wavefront's,
 * a launch walks through ALL the ticks of a wcqp_tick_run call (ticks_per_launch).  The MPC -> ZMP-CoM law -> plant chain does not
 * depend on the IK, so the tick is SKEWED inside a call: the kernel solves IK(t) and, in the shadow of its loads, the MPC chain of
 * tick t + 1 (a call starts with the MPC of its first tick alone and its last tick runs no MPC ahead: between calls nothing is
 * ahead of anything).  Same results as the in-order forms that remain for the other IK algorithms: the general 16-lane kernel
 * (algorithm 4) takes an MPC launch and an IK launch per tick, an explicit 32-lane / sweep IK algorithm or the CoM-as-cost
 * variant four (stand-alone glue / post kernels); for those, and with ticks_per_launch = 1, `use_graph` replays hipGraphs of 8
 * ticks each (captured ONCE: the tick index lives in device memory), remaining ticks go as plain launches.
 * ===================================================================================== */
#define WCQP_KIN_HANDOFF_FUSED   0
#define WCQP_KIN_HANDOFF_DENSE   1
#define WCQP_KIN_HANDOFF_COMPACT 2
typedef struct wcqp_tick_params {
    int32_t batch;              /* instances on this device                                   */
    int32_t first;              /* global index of instance 0 (disturbance stream)             */
    int32_t max_ticks;          /* trajectories hold max_ticks + horizon + 1 stages            */
    int32_t log_ticks;          /* > 0: keep u0/dq of the first log_ticks ticks for parity     */
    int32_t step_ticks, ds_ticks;
    double  k_com, k_zmp;       /* zmpControllerParams.ini:7-8                                 */
    double  noise;              /* amplitude of the bounded DCM disturbance                    */
    uint64_t seed;
    wcqp_mpc_params mpc;
    wcqp_ik_params ik;
    /* Per-tick kinematics (SURVEY.md 8f-4 inside the tick, WM/src/WalkingModule.cpp:715, 396-410): when set, every tick
     * first evaluates the forward kinematics of `kin` at the integrated joint positions q_des with the floating base
     * anchored at the stance foot of the current step (world_T_base = desired sole pose x inverse of the sole's pose in
     * the base frame: WalkingFK::evaluateWorldToBaseTransformation, WM/src/WalkingForwardKinematics.cpp:160-256) and gives
     * the IK of the tick the four MIXED Jacobians and the actual foot / neck poses and CoM (`kin_handoff` says how); the
     * support-polygon rows of the three contact pairs are built from the DESIRED foot poses (state0 entries 24..47,
     * `foot_rect`) when those are uploaded, and a tick selects by its contact pair (setConvexHullConstraint,
     * ...PredictiveController.cpp:364-435, switches rows only when the pair changes).  The J_* and hull_tab_* inputs are
     * then ignored (may be NULL). */
    /* IK hot start (SQProblem::hotstart, WM/src/WalkingQPInverseKinematics_qpOASES.cpp:312-335): by default every tick
     * first tries the previous tick's active joint-velocity bounds of the robot (added in one step, accepted when all
     * their multipliers are positive) and falls back to the cold active-set walk otherwise; 1 = always cold.
     * Same optimum either way (the QP is strictly convex); base-eliminated kernel only. */
    int32_t ik_cold_start_only;
    int32_t use_kinematics;
    wcqp_kin_params kin;
    double  foot_rect[8];       /* corners (x, y) x 4 of the foot rectangle in the foot frame (foot_size, cpp:295-303) */
    /* How the per-tick kinematics reach the IK of the same tick (WCQP_KIN_HANDOFF_*; same results):
     * FUSED (0, default)  no hand-off at all: the wavefront that solves a robot's IK first evaluates its forward kinematics and
     *           its Jacobian columns (16 lanes per robot, two joints per lane) - one launch per tick, or many ticks per launch
     *           (ticks_per_launch).  Needs a tree whose joints each lie on the path of at most ONE of the three frames (left sole,
     *           right sole, neck: a humanoid whose legs and torso branch at the root link), depth-first joint numbering, depth
     *           <= 8 and an MPC horizon <= 55; otherwise COMPACT is taken.
     * DENSE (1)   a kinematics launch per tick writes the four dense Jacobians (the layouts of wcqp_kin_jacobians_* /
     *           wcqp_ik_solve_*), 4.4 KB per robot of which ~70 % are structural zeros.
     * COMPACT (2) a kinematics launch per tick writes, per joint, its CoM column and its column of the ONE frame Jacobian
     *           it is on the path of, plus the three vectors p_frame - p_base that make up the base blocks [I -S(p); 0 I]:
     *           1.4 KB per robot (same condition on the tree as FUSED, else DENSE). */
    int32_t kin_handoff;
    /* Ticks per launch of the fused kernel (default IK kernel; constant Jacobians or FUSED kinematics): the robots of a wavefront depend on
     * no other wavefront's, so a wave walks through the ticks of a wcqp_tick_run call on its own - no launch, ramp-up or tail
     * per tick, and a wave whose robots walk a long active set falls behind without holding anybody up.
     * 0 -> all the ticks of a wcqp_tick_run call in one launch; k > 0: at most k per launch; 1 = one launch per tick (what
     * `use_graph` then replays from a hipGraph of 8 ticks).  Same results whatever the value. */
    int32_t ticks_per_launch;
    /* > 0: keep, for the first logger_ticks ticks, the row WalkingModule hands its logger per tick (WM/src/WalkingModule.cpp:800-810;
     * the 53 values behind "record" of the column list :1231-1250), per robot: wcqp_tick_outputs.logger.  Columns:
     *   0-1 dcm (measured)   2-3 dcm_des   4-5 dcm_des_d (finite difference of the uploaded reference: the planner's DCM velocity is
     *   not an input of this pipeline)   6-7 zmp (measured = the previous command)   8-9 zmp_des (the MPC's u0)   10-12 com (measured:
     *   the kinematics' CoM with use_kinematics, else the plant's)   13-14 com_des   15-16 com_des_d   17-19 lf position   20-22 lf
     *   roll pitch yaw (iDynTree::Rotation::asRPY)   23-28 rf   29-34 lf_des   35-40 rf_des   41-46 lf_err   47-52 rf_err (the IK's
     *   getLeftFootError / getRightFootError; with kinematics in the tick the dense Jacobians they are formed with do not exist and
     *   the twelve - residuals of equality constraints, O(1e-15) in the reference too - are logged as zeros).
     * Runs a logging build of the tick kernel (the default IK kernel only); a debugging aid like the reference's dumpData. */
    int32_t logger_ticks;
    /* Where a tick's MEASURED state comes from (WCQP_TICK_PLANT_*).  INTERNAL (0, default): the synthetic LIPM plant of the harness
     * (measured DCM follows xi+ = a xi + b u0 + w, measured CoM c+ = c + dT (-omega (c - xi)), measured ZMP = the previous command,
     * measured joints = the desired ones).  EXTERNAL: the caller's - B robots simulated or measured by something else - handed over on
     * the device before every tick with wcqp_tick_set_feedback_device; what the reference reads from the robot every tick:
     * setFeedback(measuredDCM) WM/src/WalkingModule.cpp:612, WalkingZMPController::setFeedback(measuredZMP, measuredCoM) :665, the
     * measured joint positions of WalkingQPIK::setRobotState :373.  A tick then cannot run ahead of its feedback: wcqp_tick_run takes
     * exactly ONE tick per call, in order (MPC(t), then IK(t): 2 launches + the feedback copy).  Default IK kernel only. */
    int32_t plant;
} wcqp_tick_params;
#define WCQP_TICK_PLANT_INTERNAL 0
#define WCQP_TICK_PLANT_EXTERNAL 1

typedef struct wcqp_tick_inputs {   /* HOST pointers, copied at upload */
    const double* ref_traj;     /* [B][max_ticks+N+1][2]                                      */
    const double* hull_tab_A;   /* [B][3][8][2]  rows for {left, right, both} in contact       */
    const double* hull_tab_b;   /* [B][3][8]                                                   */
    const int32_t* hull_tab_nc; /* [B][3]                                                      */
    const int32_t* phase0;      /* [B] offset into the step cycle                              */
    const double* J_left; const double* J_right; const double* J_neck; const double* J_com;
    const double* state0;       /* [B][87] poses; CoM entries and twists are rewritten per tick */
    const double* swing_twist;  /* [B][6] desired twist of whichever foot is in the air        */
    const double* q0;           /* [B][dof]                                                    */
    const double* dcm0; const double* com0; const double* u_init;   /* [B][2] each             */
} wcqp_tick_inputs;

typedef struct wcqp_tick_outputs {  /* HOST pointers, any may be NULL */
    double* u0_log;             /* [log_ticks][B][2]                                           */
    double* dq_log;             /* [log_ticks][B][dof]                                         */
    double* q_des;              /* [B][dof]                                                    */
    double* dcm; double* com;   /* [B][2]                                                      */
    int64_t* mpc_fail; int64_t* ik_fail;   /* [B] ticks whose QP did not end SOLVED; a robot whose IK failed once is
                                              stopped (the reference's updateModule returns false, WalkingModule.cpp:723-739):
                                              dq = 0 from then on and every further tick counts                      */
    int64_t* hot_try; int64_t* hot_hit;    /* [B] ticks on which the previous active set was tried / accepted (IK hot start) */
    int32_t* tick;              /* ticks executed so far                                       */
    double* logger;             /* [logger_ticks][B][53] logger rows (wcqp_tick_params.logger_ticks)   */
    uint32_t* active_lower; uint32_t* active_upper;   /* [B] the IK's active joint-velocity bounds of the LAST tick (bit i = joint i):
                                                         what the next tick's hot start begins from                               */
} wcqp_tick_outputs;

typedef struct wcqp_tick_s* wcqp_tick_t;
int wcqp_tick_create(const wcqp_tick_params* params, wcqp_tick_t* out);
int wcqp_tick_destroy(wcqp_tick_t h);
int wcqp_tick_upload(wcqp_tick_t h, const wcqp_tick_inputs* in);                 /* also rewinds to tick 0 */
/* enqueue only; use_graph: hipGraph replays of 8 ticks each, remainder as plain launches - IGNORED (no graph is built) whenever
 * the fused kernel runs several ticks per launch, i.e. for every wcqp_tick_params.ticks_per_launch != 1 including the default 0,
 * which makes a whole call ONE launch: a caller that needs the device back within a bound (another stream's work, a watchdog)
 * caps it - ticks_per_launch = 256 keeps a launch of 8192 robots under 5 ms at no measurable cost.  WCQP_E_INVALID when the
 * ticks enqueued since the last upload + n_ticks would exceed max_ticks (the trajectories end there).  A call that fails AFTER
 * it has started to enqueue leaves the handle without a defined state: it then refuses to run until the next wcqp_tick_upload. */
int wcqp_tick_run(wcqp_tick_t h, int32_t n_ticks, int32_t use_graph, void* stream);
/* Trajectory merge (WM/src/WalkingModule.cpp:500-535, 1263-1308: a newly planned trajectory is spliced into the deques at a
 * merge point - 20 ticks ahead in the shipped configuration - and `resetTrajectory` is raised for exactly one tick): replaces
 * stages [from_tick, from_tick + n_stages) of every instance's DCM reference trajectory with ref_tail[B][n_stages][2] (HOST
 * pointer; its rows are staged into device memory of the handle BEFORE the call returns - the caller may release or reuse
 * ref_tail at once, whatever is still running on `stream`), in stream order behind the ticks already enqueued, while everything
 * else of the pipeline stays as it is; later
 * ticks see the new stages through their windows [t, t + N].  from_tick >= the ticks enqueued so far, from_tick + n_stages <=
 * max_ticks + N + 1.  The reset flag itself has no counterpart: it makes MPCSolver::setGradient rebuild the gradient instead of
 * shifting it (MPCSolver.cpp:188-239), and this library always evaluates the gradient's contribution from the current window.
 * Valid between wcqp_tick_run calls (a call leaves nothing running ahead); captured graphs stay valid - the trajectory
 * buffer does not move. */
int wcqp_tick_splice_reference(wcqp_tick_t h, int32_t from_tick, int32_t n_stages, const double* ref_tail, void* stream);
/* External feedback (wcqp_tick_params.plant = WCQP_TICK_PLANT_EXTERNAL): the measured state the NEXT tick is to use - DEVICE
 * pointers, dcm_meas / com_meas / zmp_meas [B][2], q_meas [B][dof] or NULL (= the desired joint positions, as with the internal
 * plant).  Enqueues one small copy kernel on `stream`: the arrays may be reused once it has run (stream order), nothing is
 * retained.  Required before every wcqp_tick_run call of such a handle (which then takes n_ticks = 1: WCQP_E_INVALID otherwise,
 * or when no feedback has been set since the last tick); WCQP_E_UNSUPPORTED on a handle with the internal plant.
 * Replaces: WalkingController::setFeedback (:612), WalkingZMPController::setFeedback (:665), the joint part of
 * WalkingQPIK::setRobotState (:373) of WM/src/WalkingModule.cpp. */
int wcqp_tick_set_feedback_device(wcqp_tick_t h, const double* dcm_meas, const double* com_meas, const double* zmp_meas,
                                  const double* q_meas, void* stream);
/* the same from HOST pointers: staged through device memory of the handle and IN PLACE when the call returns (it synchronises: the host
 * arrays may be released at once, and the tick may then be run on any stream, a non-blocking one included).  With the device form the
 * caller orders the copy kernel's stream before the stream of the run call (the same stream does). */
int wcqp_tick_set_feedback_host(wcqp_tick_t h, const double* dcm_meas, const double* com_meas, const double* zmp_meas, const double* q_meas);
int wcqp_tick_download(wcqp_tick_t h, const wcqp_tick_outputs* out);             /* synchronises     */

#ifdef __cplusplus
}
#endif
#endif /* WCQP_H */
