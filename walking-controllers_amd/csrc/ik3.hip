// Jacobian QP-IK, third kernel: the null-space formulation of ik2.hip on 16 lanes per instance
// (four instances per wave64, one DPP row each).
//
// Same QP, inputs, outputs and reference citations as ik.hip / ik2.hip.  Why another layout:
// s_memtime stamps and latency micro-benchmarks (tools/ubench/lat.hip) showed that the kernel is
// a chain of ~45 serial cross-lane steps (arg-max ~110 cycles, crossbar / LDS round trip ~75,
// v_readlane -> use ~40) in which a dependent fp64 FMA costs 4 cycles: the steps cost the same
// whether they serve 2 instances per wave or 4, and the FMAs between them are almost free.  So
//   * lane j of an instance's 16 owns TWO columns of [A | b]: j and j + 16 (column 29 = b);
//   * an instance is one DPP row: every reduction is 4 DPP steps, no row exchange;
//   * the reduced Hessian row k lives on lane k (k = compact index), not on the variable's lane;
//   * the Gram product is one fp64 MFMA tile per instance, X' = D_B F formed on the fly.
//   * bounds: Goldfarb-Idnani dual active set; first bound straight-line, working sets of up to 4 bounds
//     replicated in registers, larger ones slot-per-lane (see phase 5).
// LDS is 632 doubles per instance (20.2 KB per block: 8 blocks per CU); the default IK kernel of the library.
// Template parameter TICK: the tick pipeline's glue / post steps fused in (tick_device.h).
#include <cmath>
#include <limits>
#include "ik_common.h"
#include "tick_device.h"

namespace {

using namespace wcqp_ik;

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int MEQ = 15;               // equality rows (CoM as constraint)
constexpr int NCOST = 3;              // cost rows: J_neck
constexpr int NN = kNV - MEQ;         // 14 free variables
constexpr int KMAX = NN;
constexpr int LDR = KMAX | 1;         // 15: odd, row-per-lane b64 accesses spread over the banks
constexpr int LDH = 18;

// ---- LDS layout per instance (doubles) ---------------------------------------------------------
// Leading dimensions of 18 doubles make the row-per-lane b128 reads (lane j reads row j) conflict
// free: 36 j mod 64 are 16 distinct multiples of 4 banks.  PER_INST = 24 mod 32 doubles puts the
// four instances of a wave 16 banks apart, so a broadcast read that serves two instances in one
// lane group does not collide either (at a multiple of 32 doubles every such read was 2-way).
constexpr int LDF = 18;
constexpr int OFF_F = 0;              // [15][LDF]  F[r][k] = (A_B^-1 A_N)[r][k], column NN = b'
constexpr int OFF_P = 15 * LDF;       // 270
//   set-up
constexpr int OFF_ST = OFF_P;         // [112] state + q (dead after the gradient)
constexpr int OFF_CB = OFF_P + 112;   // [4][16] entries of a panel's 4 pivot columns; before that the task rhs b
constexpr int OFF_RD = OFF_P + 176;   // [16][8] per row r: {g, 3 neck-row entries, D} of its basic variable
constexpr int OFF_GRV = OFF_P + 304;  // [16] reduced gradient by compact index
constexpr int OFF_DN = OFF_P + 320;   // [16] Lambda entry of the free variable with compact index k
constexpr int OFF_YTT = OFF_P;        // [5][16] rows 15..19 of Y^T: (W N Z)' and zero padding  (over ST)
constexpr int OFF_XTT = OFF_P + 80;   // [5][16] rows 15..19 of X^T: (N Z)' and zero padding    (over ST/CB)
constexpr int OFF_HM = OFF_P;         // [16][LDH] Gram tile (over everything up to RD, all dead by then)
constexpr int OFF_COL = OFF_P;        // [2][16] sweep columns (over the tile once its rows are in registers)
constexpr int OFF_XNV = OFF_P + 32;   // [16] x_N by compact index
constexpr int OFF_XBV = OFF_P + 48;   // [16] x_B by row
//   active set (over the set-up area)
constexpr int OFF_RINV = OFF_P;       // [KMAX][LDR]
constexpr int OFF_TPB = OFF_P + 212;  // [32] column tau_p by variable
constexpr int OFF_ZB = OFF_P + 244;   // [32] primal step by variable
constexpr int OFF_RV = OFF_P + 276;   // [16] dual step per slot
constexpr int OFF_CV = OFF_P + 292;   // [16]
constexpr int OFF_TKB = OFF_P + 308;  // [16] t by compact index
constexpr int OFF_TBV = OFF_P + 324;  // [16] -F t by row
constexpr int OFF_WI = OFF_P + 340;   // [16] ints: variable of slot a
constexpr int P_SIZE = 352;
constexpr int PER_INST = 632;         // >= OFF_P + P_SIZE = 622, = 24 mod 32
static_assert(OFF_WI + 8 <= OFF_P + P_SIZE && OFF_DN + 16 <= OFF_P + P_SIZE && OFF_HM + 16 * LDH <= OFF_GRV && OFF_P + P_SIZE <= PER_INST, "LDS overlays");
static_assert(PER_INST % 32 == 24 || PER_INST % 32 == 8, "instances 16 banks apart");
static_assert(PER_INST * 8 * 4 * 8 <= 160 * 1024, "8 blocks per CU");

#ifdef WCQP_IK_STAMPS
#define WCQP_STAMP(k) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
                           if (lane == __ffsll((long long)__ballot(true)) - 1) reinterpret_cast<unsigned long long*>(ferr_out)[(size_t)blockIdx.x * 16 + (k)] = t__; } while (0)
#else
#define WCQP_STAMP(k) do { } while (0)
#endif

// TICK: the receding-horizon pipeline's glue (ZMP-CoM law, plant; tick_device.h) runs in the prologue
// and its post step (joint integration, next contact pair, tick counter) in the epilogue, so that a
// tick is two launches (MPC, this) instead of four.
#ifndef WCQP_IK3_WAVES
#define WCQP_IK3_WAVES 2
#endif
// LIST: solves only the instances whose status reads WCQP_STATUS_STRUCTURE (what ik4.hip leaves behind for Jacobians
// whose base blocks are not in MIXED form): a workgroup scans 64 status words, compacts the flagged ones into LDS and
// loops over them four at a time; without any, the launch is batch / 64 workgroups that load one word and exit.
template <bool TICK, bool LIST>
__global__ __launch_bounds__(64, WCQP_IK3_WAVES)
void ik3_kernel(const IkDeviceParams* __restrict__ prm, int batch,
                const double* __restrict__ JL, const double* __restrict__ JR,
                const double* __restrict__ JN, const double* __restrict__ JC,
                const double* qpos, const double* __restrict__ state,
                double* __restrict__ dq_out, int* __restrict__ status_out,
                unsigned* __restrict__ alo_out, unsigned* __restrict__ aup_out,
                double* __restrict__ ferr_out, int* __restrict__ iters_out,
                wcqp_tick::TickDev td)
{
    __shared__ __attribute__((aligned(16))) double smem[4][PER_INST];
    __shared__ int s_list[64];

    int n_work = batch;
    long w0 = (long)blockIdx.x * 4, wstep = (long)gridDim.x * 4;
    if constexpr (LIST) {
        const long cand = (long)blockIdx.x * 64 + threadIdx.x;
        const bool flag = cand < batch && status_out[cand] == WCQP_STATUS_STRUCTURE;
        const unsigned long long m = __ballot(flag);
        n_work = __popcll(m);
        if (n_work == 0) return;
        if (flag) s_list[__popcll(m & ((1ull << threadIdx.x) - 1ull))] = (int)cand;
        wcqp::wave_lds_fence();
        w0 = 0; wstep = 4;
    }
  for (long wbase = w0; wbase < n_work; wbase += wstep) {
    const int lane = threadIdx.x;
    const int grp = lane >> 4;
    const int j = lane & 15;                        // owns columns j (slot 0) and j + 16 (slot 1)
    const long inst_raw = wbase + grp;
    const bool live = inst_raw < n_work;
    const long inst_idx = live ? inst_raw : (long)n_work - 1;
    const long inst = LIST ? (long)s_list[inst_idx] : inst_idx;
    double* S = smem[grp];
    double* F = S + OFF_F;
    double* st = S + OFF_ST;
    double* bvec = S + OFF_CB;       // task rhs, handed to the rhs column before the elimination starts
    const double inf = std::numeric_limits<double>::infinity();
    const bool var1 = j < kNV - 16;                 // column j + 16 is a variable (j < 13)
    const bool rhs1 = j == kNV - 16;                // column 29 = b
    const int col1 = j + 16;

    int tick_now = 0;
    if constexpr (TICK) tick_now = td.tick2[td.phase];

    WCQP_STAMP(0);
    // ---------------- phase 0: loads ------------------------------------------------------------
    const double Di0 = prm->lam[j], Di1 = prm->lam[col1];
    // batch constants of the rhs / gradient phase: scalar loads issued here, ahead of the global loads,
    // so that their latency is not paid where they are used
    const bool osqp_form = prm->form == WCQP_IK_FORM_OSQP;
    const double k_pos_foot = prm->k_pos_foot, k_att_foot = prm->k_att_foot, k_pos_com = prm->k_pos_com;
    const double kap = prm->kappa * (-prm->k_neck);
    const double kq0 = prm->kq[j], kq1 = prm->kq[col1], qreg0 = prm->qreg[j], qreg1 = prm->qreg[col1];
    double a0[MEQ], a1[MEQ];    // columns j and j + 16 of A = [J_left; J_right; J_com]; lane 13 slot 1: b
    double cn0[NCOST], cn1[NCOST];
    {
        // the state block first: vmcnt retires in order, and the rhs / gradient phase only needs the
        // state, q and the neck rows, so the 30 Jacobian loads stay in flight underneath it
        const double* sp = state + inst * kStateLen;
        double sreg[6];
#pragma unroll
        for (int m = 0; m < 5; ++m) sreg[m] = sp[m * 16 + j];
        sreg[5] = sp[80 + (j < kStateLen - 80 ? j : 0)];
        const double q0 = qpos[inst * kDof + j];
        const double q1 = qpos[inst * kDof + (16 + j < kDof ? 16 + j : 0)];
        // tick pipeline: this tick's desired CoM (lanes 0, 1: one axis each) and foot twists (lanes 0..5)
        // replace what the stored state block holds; issued ahead of the Jacobian loads for the same reason
        double g_com = 0.0, g_pstar = 0.0, g_vel = 0.0, g_twl = 0.0, g_twr = 0.0;
        if constexpr (TICK) {
            if (live) {
                const int i_ = (int)inst;
                const int mst = td.mpc_status[i_];
                const bool mpc_ok = mst == WCQP_STATUS_SOLVED || mst == WCQP_STATUS_OUTSIDE_HULL;
                if (j < 2) wcqp_tick::tick_glue_axis(td, i_, tick_now, j, mpc_ok, td.u0[2 * i_ + j], g_com, g_pstar, g_vel);
                if (j < 6) wcqp_tick::tick_glue_twist(td, i_, td.sel[i_], j, tick_now, g_twl, g_twr);
                if (j == 0 && !mpc_ok) td.mpc_fail[i_] += 1;
            }
        }
        const int c1 = var1 ? col1 : kNV - 1;       // lanes 13..15 reload column 28 (never used)
        const double* jl = JL + inst * (6 * kNV);
        const double* jr = JR + inst * (6 * kNV);
        const double* jc = JC + inst * (3 * kNV);
        const double* jn = JN + inst * (3 * kNV);
#pragma unroll
        for (int r = 0; r < 3; ++r) { cn0[r] = jn[r * kNV + j]; cn1[r] = jn[r * kNV + c1]; }
#pragma unroll
        for (int r = 0; r < 6; ++r) { a0[r] = jl[r * kNV + j]; a1[r] = jl[r * kNV + c1]; }
#pragma unroll
        for (int r = 0; r < 6; ++r) { a0[6 + r] = jr[r * kNV + j]; a1[6 + r] = jr[r * kNV + c1]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { a0[12 + r] = jc[r * kNV + j]; a1[12 + r] = jc[r * kNV + c1]; }
#pragma unroll
        for (int m = 0; m < 5; ++m) st[m * 16 + j] = sreg[m];
        if (80 + j < kStateLen) st[80 + j] = sreg[5];
        st[kStateLen + j] = q0;
        if (16 + j < kDof) st[kStateLen + 16 + j] = q1;
        if constexpr (TICK) {
            wcqp::wave_lds_fence();
            if (j < 2) { if (!td.kin_mode) st[66 + j] = g_com; st[69 + j] = g_pstar; st[72 + j] = g_vel; }
            if (j < 6) { st[75 + j] = g_twl; st[81 + j] = g_twr; }
            if (j == 0) wcqp_tick::tick_glue_height(td, (int)inst, st);
        }
    }
    wcqp::wave_lds_fence();

    WCQP_STAMP(1);
    // ---------------- phase 1: task rhs b (lane r < MEQ) and gradient g ---------------------------
    double b_mine = 0.0;
    if (j < MEQ) {
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
        bvec[j] = b_mine;
    }
    double g0, g1;              // gradient entries (osqp.cpp:181-196, qp.cpp:161-178)
    {
        const double e0 = kap * rot_err(st + 48, st + 57, 0);
        const double e1 = kap * rot_err(st + 48, st + 57, 1);
        const double e2 = kap * rot_err(st + 48, st + 57, 2);
        const double y0 = prm->Wn[0] * e0 + prm->Wn[1] * e1 + prm->Wn[2] * e2;
        const double y1 = prm->Wn[3] * e0 + prm->Wn[4] * e1 + prm->Wn[5] * e2;
        const double y2 = prm->Wn[6] * e0 + prm->Wn[7] * e1 + prm->Wn[8] * e2;
        g0 = -(cn0[0] * y0 + cn0[1] * y1 + cn0[2] * y2);
        if (j >= 6) g0 -= kq0 * (qreg0 - st[kStateLen + j - 6]);
        g1 = -(cn1[0] * y0 + cn1[1] * y1 + cn1[2] * y2);
        g1 -= kq1 * (qreg1 - st[kStateLen + (var1 ? j + 10 : 0)]);
        g1 = var1 ? g1 : 0.0;
    }
    wcqp::wave_lds_fence();
    if (rhs1) {
#pragma unroll
        for (int r = 0; r < MEQ; ++r) a1[r] = bvec[r];
    }

    WCQP_STAMP(2);
    // ---------------- phase 2: Gauss-Jordan with column pivoting, panels of 4 rows ---------------
    int myrow0 = -1, myrow1 = -1;      // row in which column j / j + 16 is basic (-1: free)
    unsigned kmin = 0xffffffffu;
    {
        double* cb = S + OFF_CB;
        const int rowbase = lane & 48;
        const unsigned var1_mask = var1 ? 0xffffffffu : 0u;
#pragma unroll
        for (int r0 = 0; r0 < MEQ; r0 += 4) {
            const int pw = (MEQ - r0 < 4) ? MEQ - r0 : 4;
#pragma unroll
            for (int s = 0; s < pw; ++s) {
                const int r = r0 + s;
                // this lane's better candidate; its panel entries are what the lane would publish
                // (arithmetic masks, not selects: hipcc turns a select around the float conversion into a branch)
                const unsigned k0 = (mag_key(a0[r]) | (unsigned)(31 - j)) & (unsigned)(myrow0 >> 31);
                const unsigned k1 = (mag_key(a1[r]) | (unsigned)(15 - j)) & (unsigned)(myrow1 >> 31) & var1_mask;
                const bool best1 = k1 > k0;
                double m[4];
#pragma unroll
                for (int u = 0; u < pw; ++u) m[u] = best1 ? a1[r0 + u] : a0[r0 + u];
                m[s] = wcqp::fast_rcp(m[s]);            // 1 / pivot, speculatively on every lane
                const unsigned key = row_max_u32(max(k0, k1));
                kmin = min(kmin, key);
                const int p = 31 - (int)(key & 31u);    // pivot column
                myrow0 = (p == j) ? r : myrow0;
                myrow1 = (p == col1) ? r : myrow1;
                wcqp::pin_value(kmin);
                wcqp::pin_value(myrow0);
                wcqp::pin_value(myrow1);
                double c[4];
                const int src = (rowbase + (p & 15)) << 2;
                // 1 / pivot and the entry of the next pivot row first: they are what the next arg-max waits for
                c[s] = lane_gather(m[s], src);
                if (s + 1 < pw) c[s + 1] = lane_gather(m[s + 1], src);
#pragma unroll
                for (int u = 0; u < pw; ++u) { if (u != s && u != s + 1) c[u] = lane_gather(m[u], src); }
                const double t0 = a0[r] * c[s], t1 = a1[r] * c[s];
#pragma unroll
                for (int u = 0; u < pw; ++u) {
                    if (u != s) { a0[r0 + u] = fma(-c[u], t0, a0[r0 + u]); a1[r0 + u] = fma(-c[u], t1, a1[r0 + u]); }
                }
                a0[r] = t0; a1[r] = t1;
            }
            // rank-pw update of the other rows; the pivot lanes publish their untouched entries of those
            // rows.  A store costs per instruction, so both slots share one sequence (a lane that holds
            // two pivots of the same panel is rare and gets a second one).
            {
                const bool piv0 = myrow0 >= r0, piv1 = myrow1 >= r0;
                if (piv0 || piv1) {
                    double* c = cb + ((piv0 ? myrow0 : myrow1) - r0) * 16;
#pragma unroll
                    for (int q = 0; q < MEQ; q += 2) {
                        if (q >= r0 && q < r0 + 4) continue;
                        *reinterpret_cast<double2*>(c + q) = make_double2(piv0 ? a0[q] : a1[q], q + 1 < MEQ ? (piv0 ? a0[q + 1] : a1[q + 1]) : 0.0);
                    }
                }
                if (__ballot(piv0 && piv1) != 0ull) {
                    if (piv0 && piv1) {
                        double* c = cb + (myrow1 - r0) * 16;
#pragma unroll
                        for (int q = 0; q < MEQ; q += 2) {
                            if (q >= r0 && q < r0 + 4) continue;
                            *reinterpret_cast<double2*>(c + q) = make_double2(a1[q], q + 1 < MEQ ? a1[q + 1] : 0.0);
                        }
                    }
                }
            }
            wcqp::wave_lds_fence();
#pragma unroll
            for (int q = 0; q < MEQ; q += 2) {
                if (q >= r0 && q < r0 + 4) continue;
                double x0 = a0[q], x1 = q + 1 < MEQ ? a0[q + 1] : 0.0;
                double y0 = a1[q], y1 = q + 1 < MEQ ? a1[q + 1] : 0.0;
#pragma unroll
                for (int s = 0; s < pw; ++s) {
                    const double2 c2 = *reinterpret_cast<const double2*>(cb + s * 16 + q);
                    x0 = fma(-c2.x, a0[r0 + s], x0); x1 = fma(-c2.y, a0[r0 + s], x1);
                    y0 = fma(-c2.x, a1[r0 + s], y0); y1 = fma(-c2.y, a1[r0 + s], y1);
                }
                a0[q] = x0; a1[q] = y0;
                if (q + 1 < MEQ) { a0[q + 1] = x1; a1[q + 1] = y1; }
                if (q == (r0 < 8 ? 8 : 4)) wcqp::pin_result(a0[q]);
            }
            wcqp::wave_lds_fence();
        }
    }
    bool ok = __uint_as_float(kmin & ~31u) > 1e-12f;
    // compact index of the free columns, in column order; the rhs column takes slot NN
    const bool free0 = myrow0 < 0;
    const bool free1 = var1 && myrow1 < 0;
    int kap0, kap1;
    {
        const unsigned g0b = (unsigned)((__ballot(free0) >> (16 * grp)) & 0xffffull);
        const unsigned g1b = (unsigned)((__ballot(free1) >> (16 * grp)) & 0xffffull);
        const unsigned below = (1u << j) - 1u;
        const int n0 = __popc(g0b);
        kap0 = free0 ? __popc(g0b & below) : 31;
        kap1 = free1 ? n0 + __popc(g1b & below) : (rhs1 ? NN : 31);
        ok = ok && (n0 + __popc(g1b) == NN);
        // a failed elimination must not index the tables out of range
        kap0 = (kap0 < NN) ? kap0 : (free0 ? NN - 1 : 31);
        kap1 = (kap1 <= NN) ? kap1 : (free1 ? NN - 1 : 31);
    }
    const bool own1 = free1 || rhs1;   // slot 1 owns a compact column (slot 0: free0)

    WCQP_STAMP(3);
    // ---------------- phase 3: tables for the reduced Hessian -------------------------------------
    double* rd = S + OFF_RD;
    if (!free0) {
        double* d = rd + myrow0 * 8;
        d[0] = g0; d[1] = cn0[0]; d[2] = cn0[1]; d[3] = cn0[2]; d[4] = Di0;
    }
    if (var1 && !free1) {
        double* d = rd + myrow1 * 8;
        d[0] = g1; d[1] = cn1[0]; d[2] = cn1[1]; d[3] = cn1[2]; d[4] = Di1;
    }
    if (free0) {
#pragma unroll
        for (int r = 0; r < MEQ; ++r) F[r * LDF + kap0] = a0[r];
        S[OFF_DN + kap0] = Di0;
    }
    if (own1) {
#pragma unroll
        for (int r = 0; r < MEQ; ++r) F[r * LDF + kap1] = a1[r];
        S[OFF_DN + kap1] = Di1;
    }
    wcqp::wave_lds_fence();
    // one pass over the rows for both columns: nz = column of N Z (rhs: -N x_p), reduced gradient
    double nz0[NCOST], nz1[NCOST];
#pragma unroll
    for (int s = 0; s < NCOST; ++s) { nz0[s] = cn0[s]; nz1[s] = rhs1 ? 0.0 : cn1[s]; }
    double gr0 = g0, gr1 = g1;
#pragma unroll
    for (int r = 0; r < MEQ; ++r) {
        const double2 gn = *reinterpret_cast<const double2*>(rd + r * 8);          // {g, n0}
        const double2 nn = *reinterpret_cast<const double2*>(rd + r * 8 + 2);      // {n1, n2}
        nz0[0] = fma(-gn.y, a0[r], nz0[0]); nz0[1] = fma(-nn.x, a0[r], nz0[1]); nz0[2] = fma(-nn.y, a0[r], nz0[2]);
        nz1[0] = fma(-gn.y, a1[r], nz1[0]); nz1[1] = fma(-nn.x, a1[r], nz1[1]); nz1[2] = fma(-nn.y, a1[r], nz1[2]);
        gr0 = fma(-a0[r], gn.x, gr0);
        gr1 = fma(-a1[r], gn.x, gr1);
        if ((r & 3) == 3) wcqp::pin_result(gr0);
    }
    wcqp::wave_lds_fence();           // ST / CB are dead: the operand tails overlay them
    {
        double* ytt = S + OFF_YTT;
        double* xtt = S + OFF_XTT;
        if (free0) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                xtt[s * 16 + kap0] = nz0[s];
                ytt[s * 16 + kap0] = prm->Wn[3 * s] * nz0[0] + prm->Wn[3 * s + 1] * nz0[1] + prm->Wn[3 * s + 2] * nz0[2];
            }
            S[OFF_GRV + kap0] = gr0;
        }
        if (own1) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                xtt[s * 16 + kap1] = nz1[s];
                ytt[s * 16 + kap1] = prm->Wn[3 * s] * nz1[0] + prm->Wn[3 * s + 1] * nz1[1] + prm->Wn[3 * s + 2] * nz1[2];
            }
            S[OFF_GRV + kap1] = gr1;
        }
        // zero padding: reduction rows 18, 19 and the unused compact slot 15
        xtt[3 * 16 + j] = 0.0; xtt[4 * 16 + j] = 0.0;
        ytt[3 * 16 + j] = 0.0; ytt[4 * 16 + j] = 0.0;
        if (j < 3) { xtt[j * 16 + 15] = 0.0; ytt[j * 16 + 15] = 0.0; }
        if (j < MEQ) F[j * LDF + 15] = 0.0;
    }
    wcqp::wave_lds_fence();

    WCQP_STAMP(4);
    // [Hr | h_rhs] = X Y',  X = [F D_B | (N Z)'],  Y = [F | (W N Z)'],  K = 18 (padded to 20):
    // one 16x16 fp64 MFMA tile per instance, fed by all 64 lanes (A[i = lane & 15][k = lane >> 4])
    double Hr[NN + 1];
    {
        const int mk = lane & 15, mq = lane >> 4;
        v4d acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int kk = 4 * s + mq;
            const bool tail = kk >= MEQ;
            const int oy = tail ? OFF_YTT + (kk - MEQ) * 16 + mk : OFF_F + kk * LDF + mk;
            const int ox = tail ? OFF_XTT + (kk - MEQ) * 16 + mk : OFF_F + kk * LDF + mk;
            const int od = tail ? OFF_RD + 4 : OFF_RD + kk * 8 + 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double yv = smem[g][oy];
                const double xs = smem[g][ox];
                const double dv = tail ? 1.0 : smem[g][od];
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs * dv, yv, acc[g], 0, 0, 0);
            }
        }
        // C/D layout of the f64 tile: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) smem[g][OFF_HM + (mq + 4 * reg) * LDH + mk] = acc[g][reg];
        }
        wcqp::wave_lds_fence();
        const double fm = j < NN ? 1.0 : 0.0;          // lanes 14, 15 carry zero rows
        const double dn = S[OFF_DN + (j < NN ? j : 0)];
        const double* hrow = S + OFF_HM + j * LDH;
#pragma unroll
        for (int k = 0; k <= NN; k += 2) {
            const double2 h2 = *reinterpret_cast<const double2*>(hrow + k);
            Hr[k] = fm * (h2.x + (k == j ? dn : 0.0));
            if (k + 1 <= NN) Hr[k + 1] = fm * (h2.y + (k + 1 == j ? dn : 0.0));
        }
    }
    const double gr = (j < NN ? S[OFF_GRV + j] : 0.0) - Hr[NN];   // g_r = g_j - F_j' g_B - (b'-dependent column)
    wcqp::wave_lds_fence();

    WCQP_STAMP(5);
    // bounds and active-set settings: fetched here so that the (L2 / scalar cache) latency hides under the sweep
    const double tol = prm->tol;
    const int max_iter = prm->max_iter;
    double lo0 = prm->vlo[j], hi0 = prm->vhi[j], lo1 = prm->vlo[col1], hi1 = prm->vhi[col1];
    // ---------------- phase 4: Hr^-1 (sweep over the NN pivots), x_N, x_B -------------------------
    // (a variant that hands lane k's row to the others by DPP row_newbcast - no LDS instruction, 28 VALU
    // moves per pivot - measured 3 % slower on the whole kernel: the phase is issue-bound)
    {
        double* col = S + OFF_COL;
        double pmin = 1.0;
        col[j] = Hr[0];
        wcqp::wave_lds_fence();
#pragma unroll
        for (int k = 0; k < NN; ++k) {
            double* cb = col + 16 * (k & 1);
            double* nb = col + 16 * ((k + 1) & 1);
            const double piv = cb[k];
            pmin = (piv > 0.0) ? pmin : -1.0;   // NaN-safe flag carried in a register: hipcc otherwise keeps all 14 pivots alive to test them at the end
            const double d = wcqp::fast_rcp(piv);
            const double f0 = Hr[k] * d;
            const double f = (j == k) ? (1.0 - d) : f0;
            if (k + 1 < NN) {
                Hr[k + 1] = fma(-f, cb[k + 1], Hr[k + 1]);
                nb[j] = Hr[k + 1];                        // publish the next column early
            }
#pragma unroll
            for (int q = 0; q < NN; q += 2) {
                const double2 c2 = *reinterpret_cast<const double2*>(cb + q);
                if (q != k && q != k + 1) Hr[q] = fma(-f, c2.x, Hr[q]);
                if (q + 1 < NN && q + 1 != k && q + 1 != k + 1) Hr[q + 1] = fma(-f, c2.y, Hr[q + 1]);
            }
            Hr[k] = (j == k) ? -d : f0;
            wcqp::wave_lds_fence();
        }
        ok = ok && (pmin > 0.0);
    }
    WCQP_STAMP(6);
    // Hr now holds row j of -(Hr^-1) on lanes j < NN
    double nu0, nu1;
    {
        double* xnv = S + OFF_XNV;
        double* xbv = S + OFF_XBV;
        xnv[j] = gr;                                      // reduced gradient by compact index
        wcqp::wave_lds_fence();
        double xn = 0.0;
#pragma unroll
        for (int k = 0; k < NN; k += 2) {
            const double2 g2 = *reinterpret_cast<const double2*>(xnv + k);
            xn = fma(Hr[k], g2.x, xn);
            xn = fma(Hr[k + 1], g2.y, xn);
        }
        wcqp::wave_lds_fence();
        xnv[j] = xn;                                      // x_N = -Hinv g_r
        wcqp::wave_lds_fence();
        {
            const double* frow = F + (j < MEQ ? j : 0) * LDF;
            double acc = frow[NN];                        // b'
#pragma unroll
            for (int k = 0; k < NN; k += 2) {
                const double2 f2 = *reinterpret_cast<const double2*>(frow + k);
                const double2 x2 = *reinterpret_cast<const double2*>(xnv + k);
                acc = fma(-f2.x, x2.x, acc);
                acc = fma(-f2.y, x2.y, acc);
            }
            xbv[j] = acc;
        }
        wcqp::wave_lds_fence();
        nu0 = free0 ? xnv[kap0] : xbv[myrow0];
        nu1 = var1 ? (free1 ? xnv[kap1] : xbv[myrow1]) : 0.0;
    }
    wcqp::wave_lds_fence();

    WCQP_STAMP(7);
    // ---------------- phase 5: joint-velocity bounds (qpOASES form) --------------------------------
    int st_code = ok ? WCQP_STATUS_SOLVED : WCQP_STATUS_NUMERIC;
    int it = 0;
    bool in_w0 = false, in_w1 = false;
    double sig0 = 0.0, sig1 = 0.0;
    const bool bnd0 = j >= 6;                             // the base (columns 0..5) is unbounded
    const bool bnd1 = var1;
    lo0 = bnd0 ? lo0 : -inf; hi0 = bnd0 ? hi0 : inf;
    lo1 = bnd1 ? lo1 : -inf; hi1 = bnd1 ? hi1 : inf;
    const bool need = !osqp_form && ((bnd0 && fmax(nu0 - hi0, lo0 - nu0) > tol) || (bnd1 && fmax(nu1 - hi1, lo1 - nu1) > tol));
    const unsigned long long need_m = __ballot(need);
    if (((need_m >> (16 * grp)) & 0xffffull) != 0ull && st_code == WCQP_STATUS_SOLVED) {
        // Goldfarb-Idnani dual active set as in ik_common.h (gi_active_set), two variables per lane.
        // Slot a of the working set is owned by lane a; row a of the explicit inverse Rinv of the
        // active-bound system sits in LDS (bordering on add, rank-one downdate on drop: registers
        // are what this phase is short of).  ONE flat loop, one step per pass: a pass first picks
        // the entering bound and its column tau_p if none is pending, then takes the primal/dual
        // step.  All control flow is uniform inside a DPP row; an empty working set - by far the
        // most common state when a bound enters - skips the dual-step machinery altogether.
        double* Rinv = S + OFF_RINV;
        double* tpb = S + OFF_TPB;
        double* zb = S + OFF_ZB;
        double* rvec = S + OFF_RV;
        double* cvec = S + OFF_CV;
        double* tkb = S + OFF_TKB;
        double* tbv = S + OFF_TBV;
        int* Wi = reinterpret_cast<int*>(S + OFF_WI);
        bool s_live = false;
        int s_var = 0;
        double s_sg = 0.0, s_mu = 0.0;
        double tc0[KMAX], tc1[KMAX];
        double* myR = Rinv + (j < KMAX ? j : 0) * LDR;
        int nW = 0;
        WCQP_STAMP(10);
        // pending entering bound: variable p, sign, remaining violation s, column tau_p, P[p][p], multiplier
        bool pending = false;
        int p = 0;
        double sig = 0.0, s = 0.0, tp0 = 0.0, tp1 = 0.0, ppp = 1.0, mu_p = 0.0;
        int p_info = 0;                 // entering column: compact index if free, 32 + row if basic
        bool done = false;
        // Working sets of up to KS bounds are kept REPLICATED on every lane of the instance (uniform inside
        // the DPP row): variable, sign, multiplier, the explicit inverse Rs of the active-bound system.
        // The dual step, the Schur complement and the ratio test are then plain register arithmetic
        // (k^2 FMAs, no slot-lane exchange, no reduction, no LDS), and a working-set drop costs no LDS
        // trip at all; only tau_p itself still travels through LDS.  Bigger working sets fall back to the
        // slot-per-lane loop below.
        constexpr int KS = 4;
        // (Rs is symmetric: only b >= a is stored and updated; a slot is live iff its sign is not 0)
        double Rs[KS][KS], sgS[KS], muS[KS], tvS[KS];
        int wS[KS], infoS[KS];
#pragma unroll
        for (int a = 0; a < KS; ++a) {
            sgS[a] = 0.0; muS[a] = 0.0; tvS[a] = 0.0; wS[a] = 0; infoS[a] = 0;
#pragma unroll
            for (int b = 0; b < KS; ++b) Rs[a][b] = 0.0;
        }
        // key of the most violated bound outside the working set (0: none); the choice runs on float
        // keys, the value itself is read back exactly from the owner
        auto most_violated = [&]() -> unsigned {
            const double viol0 = (bnd0 && !in_w0) ? fmax(nu0 - hi0, lo0 - nu0) : -inf;
            const double viol1 = (bnd1 && !in_w1) ? fmax(nu1 - hi1, lo1 - nu1) : -inf;
            const unsigned k0 = viol0 > tol ? (mag_key(viol0) | (unsigned)(31 - j)) : 0u;
            const unsigned k1 = viol1 > tol ? (mag_key(viol1) | (unsigned)(15 - j)) : 0u;
            return row_max_u32(max(k0, k1));
        };
        // makes the bound of `key` the pending one: p, sig, s, tau_p = Z Hr^-1 Z' e_p (t over the
        // compact indices first, then the basic rows through F), P[p][p]
        auto enter = [&](unsigned key) {
            ++it;
            p = 31 - (int)(key & 31u);
            // the owner lane's signed violation and (free / basic, index) of column p: every lane prepares
            // the values of its slot p >> 4, ds_bpermute fetches lane p & 15's (no stores, no branches)
            const bool sl1 = p >= 16;
            const double vh = sl1 ? nu1 - hi1 : nu0 - hi0, vl = sl1 ? lo1 - nu1 : lo0 - nu0;
            const double sviol = vh >= vl ? vh : -vl;                                   // sign = side, |.| = violation
            const int myinfo = sl1 ? (free1 ? kap1 : 32 + myrow1) : (free0 ? kap0 : 32 + myrow0);
            const int src = ((lane & 48) + (p & 15)) << 2;
            const double sv_p = lane_gather(sviol, src);
            p_info = __builtin_amdgcn_ds_bpermute(src, myinfo);
            s = fabs(sv_p);
            sig = sv_p >= 0.0 ? 1.0 : -1.0;
            const bool p_basic = p_info >= 32;
            const int p_idx = p_basic ? p_info - 32 : p_info;
            // t = Hr^-1 Z' e_p by compact index: column p_idx of the inverse (= entry p_idx of this lane's
            // row, picked by a select chain: the index is uniform in the row but not a constant), or
            // -Hinv F[row_p,:]' for a basic variable
            double t = 0.0;
            if (p_basic) {
                const double* frow = F + p_idx * LDF;
#pragma unroll
                for (int k = 0; k < NN; k += 2) {
                    const double2 f2 = *reinterpret_cast<const double2*>(frow + k);
                    t = fma(Hr[k], f2.x, t);
                    t = fma(Hr[k + 1], f2.y, t);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NN; ++k) t = (k == p_idx) ? -Hr[k] : t;
            }
            tkb[j] = t;                                                  // lanes >= NN: 0
            wcqp::wave_lds_fence();
            {
                const double* frow = F + (j < MEQ ? j : 0) * LDF;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NN; k += 2) {
                    const double2 f2 = *reinterpret_cast<const double2*>(frow + k);
                    const double2 t2 = *reinterpret_cast<const double2*>(tkb + k);
                    acc = fma(-f2.x, t2.x, acc);
                    acc = fma(-f2.y, t2.y, acc);
                }
                tbv[j] = acc;
            }
            wcqp::wave_lds_fence();
            tp0 = sig * (free0 ? tkb[kap0] : tbv[myrow0]);
            tp1 = var1 ? sig * (free1 ? tkb[kap1] : tbv[myrow1]) : 0.0;
            ppp = p_basic ? tbv[p_idx] : tkb[p_idx];                 // P[p][p] > 0 (sig * sig * tau_p[p])
            // tau_p at the variables of the replicated working set (slots that are not live read entry 0)
#pragma unroll
            for (int a = 0; a < KS; ++a) tvS[a] = (infoS[a] < 32) ? tkb[infoS[a]] : tbv[infoS[a] - 32];
            tpb[j] = tp0; tpb[col1] = tp1;                           // read by the slot lanes of the general loop
            mu_p = 0.0;
            pending = true;
        };
        // First bound, empty working set, straight-line (no loop bookkeeping): full step along tau_p,
        // the bound takes slot 0.  Three instances in four need nothing else.
        {
            const unsigned key = most_violated();          // != 0: that is what `need` said
            enter(key);
            if (ppp > 0.0) {
                const double inz = wcqp::fast_rcp(ppp);
                const double t = s * inz;
                nu0 = fma(-t, tp0, nu0);
                nu1 = fma(-t, tp1, nu1);
                wS[0] = p; infoS[0] = p_info; sgS[0] = sig; muS[0] = t; Rs[0][0] = inz;
                tc0[0] = tp0; tc1[0] = tp1;
                if (p == j) { in_w0 = true; sig0 = sig; }
                if (p == col1) { in_w1 = true; sig1 = sig; }
                nW = 1;
                pending = false;
                wcqp::wave_lds_fence();
                done = most_violated() == 0u;
            } else {
                st_code = WCQP_STATUS_INFEASIBLE; done = true;
            }
        }
        WCQP_STAMP(14);
#pragma unroll
        for (int a = 1; a < KS; ++a) { tc0[a] = 0.0; tc1[a] = 0.0; }
        bool small = !done;
#pragma unroll 1
        for (int pass = 0; pass < 1024 && small; ++pass) {
            if (!pending) {
                if (nW >= KS) { small = false; break; }                      // a fifth bound: general loop
                const unsigned key = most_violated();
                if (key == 0u) { done = true; small = false; break; }
                if (it >= max_iter) { st_code = WCQP_STATUS_MAX_ITER; done = true; small = false; break; }
                enter(key);
            }
            // dual step r = Rs c with c_a = sigma_a sigma_p tau_p[w_a]; primal step z = tp - sum_a r_a Tc[a];
            // Schur complement of the bordered system = P[p][p] - r'c (P is symmetric)
            double c[KS], r[KS];
#pragma unroll
            for (int a = 0; a < KS; ++a) c[a] = sgS[a] * sig * tvS[a];       // 0 on slots that are not live
            double z0 = tp0, z1 = tp1, nzv = ppp, t1 = inf;
            int jd = 0;
#pragma unroll
            for (int a = 0; a < KS; ++a) {
                double ra = 0.0;
#pragma unroll
                for (int b = 0; b < KS; ++b) ra = fma(b >= a ? Rs[a][b] : Rs[b][a], c[b], ra);
                r[a] = ra;
                z0 = fma(-ra, tc0[a], z0);
                z1 = fma(-ra, tc1[a], z1);
                nzv = fma(-ra, c[a], nzv);
            }
#pragma unroll
            for (int a = 0; a < KS; ++a) {
                const double ratio = (sgS[a] != 0.0 && r[a] > 0.0) ? muS[a] * wcqp::fast_rcp(r[a]) : inf;
                if (ratio < t1) { t1 = ratio; jd = a; }                      // ties: lowest slot
            }
            const double inz = wcqp::fast_rcp(nzv);
            const double t2 = (nzv > 1e-10 * ppp) ? s * inz : inf;           // dependence shows as a vanishing Schur complement
            const double t = fmin(t1, t2);
            if (!(t < inf)) { st_code = WCQP_STATUS_INFEASIBLE; done = true; small = false; break; }
            nu0 = fma(-t, z0, nu0);
            nu1 = fma(-t, z1, nu1);
#pragma unroll
            for (int a = 0; a < KS; ++a) muS[a] = fma(-t, r[a], muS[a]);      // r = 0 on slots that are not live
            mu_p += t;
            s -= t * nzv;
            if (t2 <= t1) {
                // full step: p takes the first free slot n; Rs <- bordered inverse
                int n = KS - 1;
#pragma unroll
                for (int a = KS - 1; a >= 0; --a) n = (sgS[a] != 0.0) ? n : a;
#pragma unroll
                for (int a = 0; a < KS; ++a) {
                    const bool me = a == n;
                    const double ra_inz = r[a] * inz;
#pragma unroll
                    for (int b = a; b < KS; ++b) {
                        // r[n] = 0 (the slot was empty), so the plain update leaves row / column n alone and the
                        // border is one select per entry
                        const double upd = fma(ra_inz, r[b], Rs[a][b]);
                        Rs[a][b] = (b == n) ? (me ? inz : -ra_inz) : (me ? -r[b] * inz : upd);
                    }
                    wS[a] = me ? p : wS[a];
                    infoS[a] = me ? p_info : infoS[a];
                    sgS[a] = me ? sig : sgS[a];
                    muS[a] = me ? mu_p : muS[a];
                    tc0[a] = me ? tp0 : tc0[a];
                    tc1[a] = me ? tp1 : tc1[a];
                }
                if (p == j) { in_w0 = true; sig0 = sig; }
                if (p == col1) { in_w1 = true; sig1 = sig; }
                ++nW;
                pending = false;
            } else {
                // partial step: slot jd leaves the working set; Rs <- downdated inverse.  The same entering
                // bound stays pending and the next pass needs nothing from LDS.
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
            // hand-over to the slot-per-lane representation: lane a owns slot a, row a of the inverse goes to LDS
#pragma unroll
            for (int a = KS; a < KMAX; ++a) { tc0[a] = 0.0; tc1[a] = 0.0; }
            double myrow[KS];
#pragma unroll
            for (int b = 0; b < KS; ++b) {
                myrow[b] = 0.0;
#pragma unroll
                for (int a = 0; a < KS; ++a) myrow[b] = (a == j) ? (b >= a ? Rs[a][b] : Rs[b][a]) : myrow[b];
            }
#pragma unroll
            for (int a = 0; a < KS; ++a) {
                if (a == j) { s_live = sgS[a] != 0.0; s_var = wS[a]; s_sg = sgS[a]; s_mu = muS[a]; Wi[a] = wS[a]; }
            }
            if (j < KMAX) {
#pragma unroll
                for (int b = 0; b < KMAX; ++b) myR[b] = (b < KS) ? myrow[b < KS ? b : 0] : 0.0;
            }
            wcqp::wave_lds_fence();
        }
#pragma unroll 1
        for (int pass = 0; pass < 1024 && !done; ++pass) {
            if (!pending) {
                const unsigned key = most_violated();
                if (key == 0u) { done = true; }
                else if (it >= max_iter) { st_code = WCQP_STATUS_MAX_ITER; done = true; }
                else enter(key);
            }
            if (!done) {
                double r_a = 0.0, z0 = tp0, z1 = tp1, nzv = ppp, t1 = inf, ratio = inf;
                if (nW > 0) {
                    // dual step r = Rinv c,  c_a = sigma_a tp[w_a]
                    wcqp::wave_lds_fence();
                    cvec[j] = s_live ? s_sg * tpb[s_var] : 0.0;
                    wcqp::wave_lds_fence();
#pragma unroll
                    for (int b = 0; b < KMAX; b += 2) {
                        const double2 c2 = *reinterpret_cast<const double2*>(cvec + b);
                        r_a = fma(myR[b], c2.x, r_a);
                        r_a = fma(myR[b + 1], c2.y, r_a);
                    }
                    r_a = s_live ? r_a : 0.0;
                    rvec[j] = r_a;
                    wcqp::wave_lds_fence();
                    // primal step z = tp - sum_a r_a Tc[a]
#pragma unroll
                    for (int a = 0; a < KMAX; a += 2) {
                        const double2 r2 = *reinterpret_cast<const double2*>(rvec + a);
                        z0 = fma(-r2.x, tc0[a], z0); z1 = fma(-r2.x, tc1[a], z1);
                        z0 = fma(-r2.y, tc0[a + 1], z0); z1 = fma(-r2.y, tc1[a + 1], z1);
                    }
                    zb[j] = z0; zb[col1] = z1;
                    wcqp::wave_lds_fence();
                    nzv = sig * zb[p];                       // Schur complement of the bordered system
                    ratio = (s_live && r_a > 0.0) ? s_mu * wcqp::fast_rcp(r_a) : inf;
                    t1 = row_min(ratio);
                }
                // a full working set (nW == n - meq) leaves no direction; otherwise dependence shows as
                // a vanishing Schur complement
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
                        // full step: p enters the first free slot; Rinv <- bordered inverse
                        const unsigned fm = (unsigned)((__ballot(j < KMAX && !s_live) >> (16 * grp)) & 0xffffull);
                        const int n = fm ? __ffs(fm) - 1 : 0;
                        const double ra_inz = r_a * inz;         // 0 on lanes without a live slot
                        const bool me = j == n;
                        if (nW > 0) {
#pragma unroll
                            for (int b = 0; b < KMAX; b += 2) {
                                const double2 r2 = *reinterpret_cast<const double2*>(rvec + b);
                                const double u0 = me ? -r2.x * inz : fma(ra_inz, r2.x, myR[b]);
                                const double u1 = me ? -r2.y * inz : fma(ra_inz, r2.y, myR[b + 1]);
                                if (j < KMAX) { myR[b] = u0; myR[b + 1] = u1; }
                            }
                            wcqp::wave_lds_fence();
                            if (j < KMAX) myR[n] = me ? inz : -ra_inz;
                        } else if (me) {
                            myR[n] = inz;
                        }
                        if (me) { s_live = true; s_var = p; s_sg = sig; s_mu = mu_p; Wi[n] = p; }
#pragma unroll
                        for (int a = 0; a < KMAX; ++a) { tc0[a] = (a == n) ? tp0 : tc0[a]; tc1[a] = (a == n) ? tp1 : tc1[a]; }
                        if (p == j) { in_w0 = true; sig0 = sig; }
                        if (p == col1) { in_w1 = true; sig1 = sig; }
                        ++nW;
                        pending = false;
                    } else {
                        // partial step: the blocking constraint leaves the working set; Rinv <- downdated inverse
                        const unsigned dm = (unsigned)((__ballot(ratio == t1) >> (16 * grp)) & 0xffffull);
                        const int jd = dm ? __ffs(dm) - 1 : 0;
                        const int wdrop = Wi[jd];
                        const double* dR = Rinv + jd * LDR;
                        const double djj = dR[jd];
                        const double f = (s_live && j != jd) ? dR[j < KMAX ? j : 0] * wcqp::fast_rcp(djj) : 0.0;      // Rinv is symmetric
                        double u[KMAX];
#pragma unroll
                        for (int b = 0; b < KMAX; ++b) u[b] = (j == jd) ? 0.0 : fma(-f, dR[b], myR[b]);
                        wcqp::wave_lds_fence();
                        if (j < KMAX) {
#pragma unroll
                            for (int b = 0; b < KMAX; ++b) myR[b] = u[b];
                            myR[jd] = 0.0;
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
        WCQP_STAMP(15);
        // certificate: every bound holds and every active bound is tight, else the walk lost accuracy
        {
            const double d0 = !bnd0 ? 0.0 : (in_w0 ? fabs(nu0 - (sig0 > 0.0 ? hi0 : lo0)) : fmax(nu0 - hi0, lo0 - nu0));
            const double d1 = !bnd1 ? 0.0 : (in_w1 ? fabs(nu1 - (sig1 > 0.0 ? hi1 : lo1)) : fmax(nu1 - hi1, lo1 - nu1));
            const double dev = fmax(d0 == d0 ? d0 : inf, d1 == d1 ? d1 : inf);
            const unsigned bad = row_max_u32((dev > 1e-9) ? 1u : 0u);
            if (st_code == WCQP_STATUS_SOLVED && bad != 0u) st_code = WCQP_STATUS_NUMERIC;
            if (st_code == WCQP_STATUS_SOLVED && in_w0) nu0 = sig0 > 0.0 ? hi0 : lo0;
            if (st_code == WCQP_STATUS_SOLVED && in_w1) nu1 = sig1 > 0.0 ? hi1 : lo1;
        }
    }

    WCQP_STAMP(8);
    // ---------------- outputs ------------------------------------------------------------------------
    const unsigned long long bu0 = __ballot(in_w0 && sig0 > 0.0), bu1 = __ballot(in_w1 && sig1 > 0.0);
    const unsigned long long bl0 = __ballot(in_w0 && sig0 < 0.0), bl1 = __ballot(in_w1 && sig1 < 0.0);
    if (live) {
        if (j >= 6) dq_out[inst * kDof + (j - 6)] = nu0;
        if (var1) dq_out[inst * kDof + (j + 10)] = nu1;
        if (j == 0) {
            const unsigned up = (unsigned)((bu0 >> (16 * grp)) & 0xffffull) | ((unsigned)((bu1 >> (16 * grp)) & 0xffffull) << 16);
            const unsigned dn = (unsigned)((bl0 >> (16 * grp)) & 0xffffull) | ((unsigned)((bl1 >> (16 * grp)) & 0xffffull) << 16);
            status_out[inst] = st_code;
            if (aup_out) aup_out[inst] = up >> 6;
            if (alo_out) alo_out[inst] = dn >> 6;
            if (iters_out) iters_out[inst] = it;
        }
    }
#ifdef WCQP_IK_STAMPS
    WCQP_STAMP(9);
    return;
#endif
    static_assert(!(TICK && LIST), "the tick pipeline does not use the list mode");
    if constexpr (TICK) {
        const bool ik_ok = st_code == WCQP_STATUS_SOLVED;
        if (live) {
            const int i_ = (int)inst;
            if (j >= 6) wcqp_tick::tick_post_joint(td, i_, tick_now, j - 6, ik_ok, nu0);
            if (var1) wcqp_tick::tick_post_joint(td, i_, tick_now, j + 10, ik_ok, nu1);
            if (j == 0) wcqp_tick::tick_post_instance(td, i_, tick_now, ik_ok);
        }
        // advanceReferenceSignals (WalkingModule.cpp:816): the next tick reads the other copy of the tick index
        if (blockIdx.x == 0 && lane == 0) td.tick2[1 - td.phase] = tick_now + 1;
    }
    if (ferr_out) {
        // b - J nu for the 12 foot rows (osqp.cpp:430-454, qp.cpp:364-401): every lane multiplies its two
        // columns (reloaded, L2-resident) by its two velocities, a [12][18] LDS tile turns the 16 partial
        // sums of a row over to lane r
        double* pb = S + OFF_P;
        const int c1 = var1 ? col1 : kNV - 1;
        const double v1 = var1 ? nu1 : 0.0;
        const double* jl = JL + inst * (6 * kNV);
        const double* jr = JR + inst * (6 * kNV);
        double part[12];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            part[r] = fma(jl[r * kNV + j], nu0, jl[r * kNV + c1] * v1);
            part[6 + r] = fma(jr[r * kNV + j], nu0, jr[r * kNV + c1] * v1);
        }
        wcqp::wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 12; ++r) pb[r * 18 + j] = part[r];
        wcqp::wave_lds_fence();
        if (j < 12 && live) {
            double acc = b_mine;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                const double2 p2 = *reinterpret_cast<const double2*>(pb + j * 18 + k);
                acc -= p2.x; acc -= p2.y;
            }
            ferr_out[inst * 12 + j] = acc;
        }
    }
    if constexpr (!LIST) break;        // one pass: keeps the plain kernel's code what it was before the list mode
    wcqp::wave_lds_fence();
  }
}

}  // namespace

namespace wcqp_ik {

int ik3_launch(const IkDeviceParams* d_prm, int batch,
               const double* JL, const double* JR, const double* JN, const double* JC,
               const double* q, const double* state, double* dq, int* status,
               unsigned* alo, unsigned* aup, double* ferr, int* iters, hipStream_t stream) {
    const unsigned grid = (unsigned)((batch + 3) / 4);
    hipLaunchKernelGGL((ik3_kernel<false, false>), dim3(grid), dim3(64), 0, stream, d_prm, batch, JL, JR, JN, JC, q, state,
                       dq, status, alo, aup, ferr, iters, wcqp_tick::TickDev{});
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

int ik3_launch_list(const IkDeviceParams* d_prm, int batch,
                    const double* JL, const double* JR, const double* JN, const double* JC,
                    const double* q, const double* state, double* dq, int* status,
                    unsigned* alo, unsigned* aup, double* ferr, int* iters, hipStream_t stream) {
    const unsigned grid = (unsigned)((batch + 63) / 64);
    hipLaunchKernelGGL((ik3_kernel<false, true>), dim3(grid), dim3(64), 0, stream, d_prm, batch, JL, JR, JN, JC, q, state,
                       dq, status, alo, aup, ferr, iters, wcqp_tick::TickDev{});
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

int ik3_launch_tick(const void* d_prm, const wcqp_tick::TickDev& td,
                    const double* JL, const double* JR, const double* JN, const double* JC,
                    unsigned* alo, unsigned* aup, hipStream_t stream) {
    if (!d_prm) return WCQP_E_INVALID;
    const unsigned grid = (unsigned)((td.batch + 3) / 4);
    hipLaunchKernelGGL((ik3_kernel<true, false>), dim3(grid), dim3(64), 0, stream, static_cast<const IkDeviceParams*>(d_prm), td.batch,
                       JL, JR, JN, JC, td.q_des, td.state, td.dq, td.ik_status, alo, aup, nullptr, nullptr, td);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

}  // namespace wcqp_ik
