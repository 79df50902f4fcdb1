// Jacobian QP-IK, second kernel: NULL-SPACE formulation (32 lanes per instance, two per wave).
//
// Same QP, same inputs/outputs and same reference citations as ik.hip; different algebra,
// chosen after profiling ik.hip (53 % of wave cycles parked on LDS round trips, 6 k
// instructions and 256 VGPRs per instance pair):
//
//   1. Gauss-Jordan on [A | b] with column pivoting, rows in panels of 4.  Lane j holds COLUMN j
//      of A in registers (lane 29 holds b, so the right-hand side rides along for free).  After
//      meq steps the basic variables are  x_B = b' - F x_N  and only nN = 29 - meq = 14 are free.
//   2. Reduced Hessian  Hr = Z'HZ = D_N + F'D_B F + (N Z)'W(N Z)  (14 x 14, SPD whenever the
//      KKT matrix is regular — H itself is only PSD) built row-per-lane over the free lanes,
//      addressed through a compact index (prefix count of free lanes).
//   3. Hr^-1 by the symmetric sweep (14 pivots instead of 29 + 15), x_N = -Hr^-1 g_r.
//   4. Bounds: the same Goldfarb-Idnani dual active set as ik.hip, fed with full-space columns
//      tau_p = Z Hr^-1 Z' e_p.
// About 2.2 k instructions per instance pair instead of 6 k, and ~1/2 the registers.
#include <cmath>
#include <limits>
#include "ik_common.h"

namespace {

using namespace wcqp_ik;

// Diagnostic builds only (tools/build_variant.sh stamps -DWCQP_IK_STAMPS, tools/stamps.py): s_memtime at phase boundaries, written by lane 0 of
// each wave into the foot-error buffer (never an output in that build).  No stamp exists in the
// product build.
#ifdef WCQP_IK_STAMPS
#define WCQP_STAMP(k) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
                           if (lane == 0) reinterpret_cast<unsigned long long*>(ferr_out)[(size_t)blockIdx.x * 16 + (k)] = t__; } while (0)
#else
#define WCQP_STAMP(k) do { } while (0)
#endif

constexpr int kLDF = 16;      // leading dim of a stored F column (meq <= 15 entries)
typedef double v4d __attribute__((ext_vector_type(4)));

template <bool USE_COM, bool USE_MFMA>
struct Ik2Layout {
    static constexpr int MEQ = USE_COM ? 15 : 12;       // equality rows
    static constexpr int NCOST = USE_COM ? 3 : 6;       // cost rows: [J_com;] J_neck
    static constexpr int NN = kNV - MEQ;                // free (non-basic) variables
    static constexpr int NK = NN + 1;                   // compact slots: free variables + the rhs lane
    static constexpr int KMAX = NN;
    static_assert(!USE_MFMA || (USE_COM && NK <= 16 && MEQ + NCOST <= 20), "one 16x16x20 MFMA tile");
    // persistent: F = A_B^-1 A_N with the rhs column b' in slot NN.
    //   VALU variant: columns by compact index, F_STORE[k * kLDF + r]
    //   MFMA variant: the K-major operand table Y^T [20][16] of the Gram product doubles as the
    //   store of F (rows 0..MEQ-1: Y^T[r][k] = F[r][k]); rows MEQ.. hold (W N Z)' and zero padding
    static constexpr int OFF_F = 0;
    static constexpr int F_SIZE = USE_MFMA ? 20 * 16 : NK * kLDF;
    static constexpr int OFF_YT = OFF_F;
    static constexpr int OFF_P = OFF_F + F_SIZE;
    // phase A (set-up), all inside the region that the active set reuses afterwards
    static constexpr int OFF_ST = OFF_P;                // [112] state + q (dead after the gradient)
    static constexpr int OFF_CB = OFF_ST + 112;         // [4][16] entries of a panel's 4 pivot columns
    static constexpr int OFF_RD = OFF_CB + 64;          // [16][8] per row r: {D, g, cost-row entries} of its basic variable
    // MFMA variant: X^T [20][16] is written after the last read of ST / CB / RD and overlays them;
    // the 16x16 result tile [16][LDH] overlays X^T again (its writes depend on the MFMA results,
    // i.e. follow every read); the sweep's buffers overlay the tile once the rows are in registers
    static constexpr int LDH = 18;
    static constexpr int OFF_XT = OFF_P;
    static constexpr int OFF_HM = OFF_XT;
    static constexpr int LDW = NCOST + (NCOST & 1);
    static constexpr int OFF_WNZ = OFF_RD + 16 * 8;     // [NK][LDW]  (VALU variant only)
    static constexpr int OFF_COL = OFF_P;               // [2][32] sweep columns, double-buffered (ST / CB / tile are dead by then)
    static constexpr int OFF_GR = OFF_COL + 64;         // [32] reduced gradient / x_N by compact index
    static constexpr int END_A = USE_MFMA ? OFF_XT + 20 * 16 : OFF_WNZ + NK * LDW;
    static_assert(OFF_GR + 32 <= OFF_RD && OFF_RD + 128 <= END_A, "sweep buffers overlay ST/CB only; RD inside the region");
    // phase B (active set) reuses the phase-A area
    static constexpr int LDR = KMAX | 1;                // odd leading dim: row-per-lane accesses spread over banks
    static constexpr int OFF_RINV = OFF_P;              // [KMAX][LDR]
    static constexpr int OFF_V0 = OFF_RINV + KMAX * LDR + ((KMAX * LDR) & 1);
    static constexpr int OFF_V1 = OFF_V0 + 32;          // sign / z
    static constexpr int OFF_V2 = OFF_V1 + 32;          // tp
    static constexpr int OFF_V3 = OFF_V2 + 32;          // published Hinv row / t by compact index
    static constexpr int OFF_R = OFF_V3 + 32;           // [32] dual step per slot
    static constexpr int OFF_C = OFF_R + 32;            // [32]
    static constexpr int OFF_WI = OFF_C + 32;           // [32] ints
    static constexpr int OFF_INFO = OFF_WI + 16;        // [4]
    static constexpr int END_B = OFF_INFO + 4;
    static constexpr int OFF_B = (END_A > END_B ? END_A : END_B);   // [16] task rhs (kept for foot errors)
    static constexpr int PER_INST = ((OFF_B + 16) + 1) & ~1;
};

template <bool USE_COM, bool USE_MFMA>
// 2 waves/SIMD: at 3 the allocator spills ~50 B/lane to scratch, which costs more than the extra wave buys
#ifndef WCQP_IK2_WAVES
#define WCQP_IK2_WAVES 2
#endif
__global__ __launch_bounds__(64, WCQP_IK2_WAVES)
void ik2_kernel(const IkDeviceParams* __restrict__ prm, int batch,
                const double* __restrict__ JL, const double* __restrict__ JR,
                const double* __restrict__ JN, const double* __restrict__ JC,
                const double* __restrict__ qpos, const double* __restrict__ state,
                double* __restrict__ dq_out, int* __restrict__ status_out,
                unsigned* __restrict__ alo_out, unsigned* __restrict__ aup_out,
                double* __restrict__ ferr_out, int* __restrict__ iters_out)
{
    using L = Ik2Layout<USE_COM, USE_MFMA>;
    constexpr int MEQ = L::MEQ, NCOST = L::NCOST, NN = L::NN, NK = L::NK, KMAX = L::KMAX, LDW = L::LDW;
    (void)NK;
    __shared__ __attribute__((aligned(16))) double smem[2][L::PER_INST];

    const int lane = threadIdx.x;
    const int half = lane >> 5;
    const int i = lane & 31;                       // variable owned by this lane; lane 29 = rhs column
    const long inst_raw = (long)blockIdx.x * 2 + half;
    const bool live = inst_raw < batch;
    const long inst = live ? inst_raw : (long)batch - 1;
    double* S = smem[half];
    double* Fst = S + L::OFF_F;
    // entry (row r, compact column k) of F; see Ik2Layout
    auto F_at = [&](int k, int r) -> double& { return USE_MFMA ? Fst[r * 16 + k] : Fst[k * kLDF + r]; };
    double* st = S + L::OFF_ST;
    double* bvec = S + L::OFF_B;
    const double inf = std::numeric_limits<double>::infinity();
    const bool var = i < kNV;
    const bool rhs_lane = i == kNV;

    WCQP_STAMP(0);
    // ---------------- phase 0: loads (column i of every task Jacobian) -------------------
    // per-variable constants first: their L2 latency hides under the Jacobian loads instead of
    // being exposed where they are consumed
    double Di = prm->lam[i], kq_i = prm->kq[i], qreg_i = prm->qreg[i];
    double a[MEQ];          // column i of A = [J_left; J_right; (J_com)]; on lane 29: b
    double cn[NCOST];       // column i of the cost rows [ (J_com;) J_neck ]
    {
        // lanes 29..31 own no variable: they load column 28 again (no predicated loads, which hipcc
        // turns into one branch each); lane 29's copy is replaced by b below and nothing ever
        // reads what lanes 30/31 compute
        const int ic = var ? i : kNV - 1;
        const double* jl = JL + inst * (6 * kNV) + ic;
        const double* jr = JR + inst * (6 * kNV) + ic;
        const double* jc = JC + inst * (3 * kNV) + ic;
        const double* jn = JN + inst * (3 * kNV) + ic;
#pragma unroll
        for (int r = 0; r < 6; ++r) a[r] = jl[r * kNV];
#pragma unroll
        for (int r = 0; r < 6; ++r) a[6 + r] = jr[r * kNV];
        if constexpr (USE_COM) {
#pragma unroll
            for (int r = 0; r < 3; ++r) a[12 + r] = jc[r * kNV];
#pragma unroll
            for (int r = 0; r < 3; ++r) cn[r] = jn[r * kNV];
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) cn[r] = jc[r * kNV];
#pragma unroll
            for (int r = 0; r < 3; ++r) cn[3 + r] = jn[r * kNV];
        }
        const double* sp = state + inst * kStateLen;
        st[i] = sp[i];
        st[i + 32] = sp[i + 32];
        if (i + 64 < kStateLen) st[i + 64] = sp[i + 64];
        if (i < kDof) st[kStateLen + i] = qpos[inst * kDof + i];
    }
    wcqp::wave_lds_fence();

    WCQP_STAMP(1);
    // ---------------- phase 1: task rhs b (lane r < MEQ) and gradient g ------------------
    const bool osqp_form = prm->form == WCQP_IK_FORM_OSQP;
    double b_mine = 0.0;
    if (i < MEQ) {
        if (i < 12) {
            const int foot = i / 6, k = i % 6;
            const double* p  = st + (foot ? 12 : 0);
            const double* R  = st + (foot ? 15 : 3);
            const double* pd = st + (foot ? 36 : 24);
            const double* Rd = st + (foot ? 39 : 27);
            const double* tw = st + (foot ? 81 : 75);
            const double corr = k < 3 ? prm->k_pos_foot * (p[k] - pd[k]) : prm->k_att_foot * rot_err(R, Rd, k - 3);
            const bool skip = osqp_form && tw[0] == tw[1] && tw[0] == 0.0;        // osqp.cpp:286-306
            b_mine = skip ? tw[k] : tw[k] - corr;
        } else {
            const int k = i - 12;
            b_mine = st[72 + k] - prm->k_pos_com * (st[66 + k] - st[69 + k]);
        }
        bvec[i] = b_mine;
    }
    double g = 0.0;             // gradient entry of this variable (osqp.cpp:181-196, qp.cpp:161-178)
    {
        const double kap = prm->kappa * (-prm->k_neck);
        const double e0 = kap * rot_err(st + 48, st + 57, 0);
        const double e1 = kap * rot_err(st + 48, st + 57, 1);
        const double e2 = kap * rot_err(st + 48, st + 57, 2);
        const double y0 = prm->Wn[0] * e0 + prm->Wn[1] * e1 + prm->Wn[2] * e2;
        const double y1 = prm->Wn[3] * e0 + prm->Wn[4] * e1 + prm->Wn[5] * e2;
        const double y2 = prm->Wn[6] * e0 + prm->Wn[7] * e1 + prm->Wn[8] * e2;
        constexpr int NO = NCOST - 3;                      // neck rows offset inside cn
        if (var) {
            g = -(cn[NO] * y0 + cn[NO + 1] * y1 + cn[NO + 2] * y2);
            if (i >= 6) g -= kq_i * (qreg_i - st[kStateLen + i - 6]);
            if constexpr (!USE_COM) {
                const double w0 = prm->Wc[0] * st[72] + prm->Wc[1] * st[73] + prm->Wc[2] * st[74];
                const double w1 = prm->Wc[3] * st[72] + prm->Wc[4] * st[73] + prm->Wc[5] * st[74];
                const double w2 = prm->Wc[6] * st[72] + prm->Wc[7] * st[73] + prm->Wc[8] * st[74];
                g -= cn[0] * w0 + cn[1] * w1 + cn[2] * w2;
            }
        }
    }
    wcqp::wave_lds_fence();
    if (rhs_lane) {
#pragma unroll
        for (int r = 0; r < MEQ; ++r) a[r] = bvec[r];
    }

    WCQP_STAMP(2);
    // ---------------- phase 2: Gauss-Jordan with column pivoting --------------------------
    int myrow = -1;             // row in which this lane's variable is basic (-1: free)
    unsigned kmin = 0xffffffffu; // smallest pivot key met (|pivot| as float bits)
    {
        // Blocked: rows are taken in panels of 4.  Inside a panel only the panel's own rows are
        // reduced, so a step moves 4 doubles (3 entries + 1/pivot) of the pivot lane, by v_readlane
        // into SGPRs (VALU has headroom, the CU-shared LDS pipe has none); 1/pivot is computed by
        // EVERY lane for its own entry while the arg-max runs, off the serial chain.  The
        // other 11 rows then get the panel's rank-4 update  a[q] -= sum_s C[q][s] T[s]  at once:
        // the 4 pivot lanes publish their (still untouched) entries C of those rows in the SAME
        // store instructions - an LDS store costs per instruction, not per lane, so 24 stores per
        // instance pair replace the 120 of a column-at-a-time Gauss-Jordan (they were 2/3 of the
        // kernel's LDS-pipe time, and the LDS pipe is what bounds this kernel at scale).
        double* cb = S + L::OFF_CB;
#pragma unroll
        for (int r0 = 0; r0 < MEQ; r0 += 4) {
            const int pw = (MEQ - r0 < 4) ? MEQ - r0 : 4;
#pragma unroll
            for (int s = 0; s < pw; ++s) {
                const int r = r0 + s;
                const double rinv_mine = wcqp::fast_rcp(a[r]);
                unsigned key;
                const int pl = group_argmax_abs(a[r], var && myrow < 0, i, key);
                kmin = min(kmin, key);
                myrow = (i == pl) ? r : myrow;
                wcqp::pin_value(kmin);
                wcqp::pin_value(myrow);
                // the pivot lane of each instance is wave-uniform once read into an SGPR, and so are
                // its panel entries: v_readlane puts them in SGPRs that feed the FMAs directly
                const int pl0 = __builtin_amdgcn_readlane(pl, 0);
                const int pl1 = __builtin_amdgcn_readlane(pl, 32) + 32;
                if (half == 0) {
                    const double t = a[r] * lane_value(rinv_mine, pl0);
#pragma unroll
                    for (int u = 0; u < pw; ++u) {
                        if (u != s) a[r0 + u] = fma(-lane_value(a[r0 + u], pl0), t, a[r0 + u]);
                    }
                    a[r] = t;
                } else {
                    const double t = a[r] * lane_value(rinv_mine, pl1);
#pragma unroll
                    for (int u = 0; u < pw; ++u) {
                        if (u != s) a[r0 + u] = fma(-lane_value(a[r0 + u], pl1), t, a[r0 + u]);
                    }
                    a[r] = t;
                }
            }
            if (myrow >= r0) {
                double* c = cb + (myrow - r0) * 16;
#pragma unroll
                for (int q = 0; q < MEQ; q += 2) {
                    if (q >= r0 && q < r0 + 4) continue;                  // the panel's own rows are done
                    *reinterpret_cast<double2*>(c + q) = make_double2(a[q], q + 1 < MEQ ? a[q + 1] : 0.0);
                }
            }
            wcqp::wave_lds_fence();
#pragma unroll
            for (int q = 0; q < MEQ; q += 2) {
                if (q >= r0 && q < r0 + 4) continue;
                double acc0 = a[q], acc1 = q + 1 < MEQ ? a[q + 1] : 0.0;
#pragma unroll
                for (int s = 0; s < pw; ++s) {
                    const double2 c2 = *reinterpret_cast<const double2*>(cb + s * 16 + q);
                    acc0 = fma(-c2.x, a[r0 + s], acc0);
                    acc1 = fma(-c2.y, a[r0 + s], acc1);
                }
                a[q] = acc0;
                if (q + 1 < MEQ) a[q + 1] = acc1;
                // at most half a panel's column reads in flight: all 24 at once cost 96 VGPRs
                if (q == (r0 < 8 ? 8 : 4)) wcqp::pin_result(a[q]);
            }
            wcqp::wave_lds_fence();
        }
    }
    const bool basic = myrow >= 0;
    bool ok = __uint_as_float(kmin & ~31u) > 1e-12f;
    // compact index of the free lanes (prefix count inside the 32-lane group); rhs lane -> slot NN
    const bool free_var = var && !basic;
    const unsigned long long fm = __ballot(free_var);
    const unsigned gm = (unsigned)((fm >> (32 * half)) & 0xffffffffull);
    const int kap_i = free_var ? __popc(gm & ((1u << i) - 1u)) : (rhs_lane ? NN : 31);
    const bool rowlane = free_var || rhs_lane;          // lanes that own a row of [Hr | h_rhs]
    ok = ok && (__popc(gm) == NN);

    WCQP_STAMP(3);
    // ---------------- phase 3: reduced Hessian rows ---------------------------------------
    double* rd = S + L::OFF_RD;
    double* wnz = S + L::OFF_WNZ;
    if (basic) {
        rd[myrow * 8] = Di; rd[myrow * 8 + 1] = g;
#pragma unroll
        for (int s = 0; s < NCOST; ++s) rd[myrow * 8 + 2 + s] = cn[s];
    }
    if (rowlane) {
#pragma unroll
        for (int r = 0; r < MEQ; ++r) F_at(kap_i, r) = a[r];                    // MFMA: Y^T rows 0..MEQ-1 = F
    }
    wcqp::wave_lds_fence();
    // one pass over the rows: nz = column of N Z (rhs lane: -N x_p), g_r partial, a := F_j[r] D_B[r]
    double nz[NCOST];
#pragma unroll
    for (int s = 0; s < NCOST; ++s) nz[s] = rhs_lane ? 0.0 : cn[s];
    double gr = g;
#pragma unroll
    for (int r = 0; r < MEQ; ++r) {
        const double2 dg = *reinterpret_cast<const double2*>(rd + r * 8);
        const double ar = a[r];
#pragma unroll
        for (int s = 0; s < NCOST; s += 2) {
            const double2 n2 = *reinterpret_cast<const double2*>(rd + r * 8 + 2 + s);
            nz[s] = fma(-n2.x, ar, nz[s]);
            if (s + 1 < NCOST) nz[s + 1] = fma(-n2.y, ar, nz[s + 1]);
        }
        gr = fma(-ar, dg.y, gr);
        a[r] = ar * dg.x;       // F itself now lives in LDS
        if (r & 1) wcqp::pin_result(gr);
    }
    {
        double w[NCOST];
        if constexpr (USE_COM) {
#pragma unroll
            for (int s = 0; s < 3; ++s) w[s] = prm->Wn[3 * s] * nz[0] + prm->Wn[3 * s + 1] * nz[1] + prm->Wn[3 * s + 2] * nz[2];
        } else {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                w[s] = prm->Wc[3 * s] * nz[0] + prm->Wc[3 * s + 1] * nz[1] + prm->Wc[3 * s + 2] * nz[2];
                w[3 + s] = prm->Wn[3 * s] * nz[3] + prm->Wn[3 * s + 1] * nz[4] + prm->Wn[3 * s + 2] * nz[5];
            }
        }
        if constexpr (USE_MFMA) {
            wcqp::wave_lds_fence();      // X^T overlays RD: every lane's row pass has read it by now
            if (rowlane) {
#pragma unroll
                for (int r = 0; r < MEQ; ++r) S[L::OFF_XT + r * 16 + kap_i] = a[r];   // X^T rows 0..MEQ-1 = F D_B
#pragma unroll
                for (int s = 0; s < NCOST; ++s) { S[L::OFF_XT + (MEQ + s) * 16 + kap_i] = nz[s]; S[L::OFF_YT + (MEQ + s) * 16 + kap_i] = w[s]; }
            }
            // zero padding: reduction rows MEQ+NCOST..19 and the unused slot column 15
            if (i < 16) {
#pragma unroll
                for (int r = MEQ + NCOST; r < 20; ++r) { S[L::OFF_XT + r * 16 + i] = 0.0; S[L::OFF_YT + r * 16 + i] = 0.0; }
            }
            if (i < 20) {
#pragma unroll
                for (int c = NK; c < 16; ++c) { S[L::OFF_XT + i * 16 + c] = 0.0; S[L::OFF_YT + i * 16 + c] = 0.0; }
            }
        } else {
            if (rowlane) {
#pragma unroll
                for (int s = 0; s < NCOST; ++s) wnz[kap_i * LDW + s] = w[s];
            }
        }
    }
    wcqp::wave_lds_fence();
    WCQP_STAMP(4);
    double Hr[NK];
    const double fmask = free_var ? 1.0 : 0.0;
    if constexpr (USE_MFMA) {
        // The one genuinely dense contraction of the path: the 15 x 15 Gram product
        //   [Hr | h_rhs] = X Y',   X = [F D_B | (N Z)'],  Y = [F | (W N Z)'],  K = 18 (padded to 20)
        // is exactly one 16x16 fp64 MFMA tile per instance: 5 x v_mfma_f64_16x16x4 instead of
        // 270 VALU FMAs + 150 LDS broadcasts per instance pair.  All 64 lanes feed each
        // instance's tile (A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]).
        const int mk = lane & 15, mq = lane >> 4;
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double* x0 = smem[0] + L::OFF_XT; const double* y0 = smem[0] + L::OFF_YT;
        const double* x1 = smem[1] + L::OFF_XT; const double* y1 = smem[1] + L::OFF_YT;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int o = (4 * s + mq) * 16 + mk;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[o], y0[o], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[o], y1[o], acc1, 0, 0, 0);
        }
        // C/D layout of the f64 tile: col = lane & 15, row = (lane >> 4) + 4 * reg
        double* h0 = smem[0] + L::OFF_HM; double* h1 = smem[1] + L::OFF_HM;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            h0[(mq + 4 * reg) * L::LDH + mk] = acc0[reg];
            h1[(mq + 4 * reg) * L::LDH + mk] = acc1[reg];
        }
        wcqp::wave_lds_fence();
        const double* hrow = S + L::OFF_HM + (kap_i < 16 ? kap_i : 15) * L::LDH;
#pragma unroll
        for (int k = 0; k < NK; k += 2) {
            const double2 h2 = *reinterpret_cast<const double2*>(hrow + k);
            Hr[k] = fmask * (h2.x + (k == kap_i ? Di : 0.0));
            if (k + 1 < NK) Hr[k + 1] = fmask * (h2.y + (k + 1 == kap_i ? Di : 0.0));
        }
        wcqp::wave_lds_fence();
    } else {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < MEQ; r += 2) {
            const double2 f2 = *reinterpret_cast<const double2*>(Fst + k * kLDF + r);
            acc = fma(a[r], f2.x, acc);
            if (r + 1 < MEQ) acc = fma(a[r + 1], f2.y, acc);
        }
#pragma unroll
        for (int s = 0; s < NCOST; s += 2) {
            const double2 w2 = *reinterpret_cast<const double2*>(wnz + k * LDW + s);
            acc = fma(nz[s], w2.x, acc);
            if (s + 1 < NCOST) acc = fma(nz[s + 1], w2.y, acc);
        }
        // non-free lanes are zero padding; a multiply (not a select) keeps hipcc from turning the row
        // into a branch and hoisting every LDS read of the loop in front of it (they then spill)
        Hr[k] = fmask * (acc + (k == kap_i ? Di : 0.0));
        wcqp::pin_result(Hr[k]);
    }
    }
    gr -= Hr[NN];               // g_r = g_j - F_j' g_B - (b'-dependent column)

    WCQP_STAMP(5);
    // bounds of this variable: fetched here so that the (L2) latency hides under the sweep and
    // the two registers are not carried through the set-up phases
    double lo = prm->vlo[i], hi = prm->vhi[i];
    // ---------------- phase 4: Hr^-1 (sweep over the NN compact pivots), x_N, x_B ---------
    {
        double* col = S + L::OFF_COL;
        col[kap_i] = Hr[0];
        wcqp::wave_lds_fence();
#pragma unroll
        for (int k = 0; k < NN; ++k) {
            double* cb = col + 32 * (k & 1);
            double* nb = col + 32 * ((k + 1) & 1);
            const double piv = cb[k];
            ok = ok && (piv > 0.0);
            const double d = wcqp::fast_rcp(piv);
            const double ck = Hr[k];
            const double f0 = ck * d;
            const double f = (kap_i == k) ? (1.0 - d) : f0;
            if (k + 1 < NN) {
                Hr[k + 1] = fma(-f, cb[k + 1], Hr[k + 1]);
                nb[kap_i] = Hr[k + 1];                       // publish the next column early
            }
#pragma unroll
            for (int j = 0; j < NN; j += 2) {
                const double2 c2 = *reinterpret_cast<const double2*>(cb + j);
                if (j != k && j != k + 1) Hr[j] = fma(-f, c2.x, Hr[j]);
                if (j + 1 < NN && j + 1 != k && j + 1 != k + 1) Hr[j + 1] = fma(-f, c2.y, Hr[j + 1]);
            }
            Hr[k] = (kap_i == k) ? -d : f0;
            wcqp::wave_lds_fence();
        }
    }
    WCQP_STAMP(6);
    // Hr now holds -(Hr^-1) rows on the free lanes
    double* grv = S + L::OFF_GR;
    if (rowlane) grv[kap_i] = gr;
    wcqp::wave_lds_fence();
    double xn = 0.0;
#pragma unroll
    for (int k = 0; k < NN; ++k) xn = fma(Hr[k], grv[k], xn);   // x_N = -Hinv g_r
    wcqp::wave_lds_fence();
    if (free_var) grv[kap_i] = xn;
    wcqp::wave_lds_fence();
    double nu = free_var ? xn : 0.0;
    if (basic) {
        double acc = F_at(NN, myrow);                            // b'
#pragma unroll
        for (int k = 0; k < NN; ++k) acc = fma(-F_at(k, myrow), grv[k], acc);
        nu = acc;
    }
    wcqp::wave_lds_fence();

    WCQP_STAMP(7);
    // ---------------- phase 5: joint-velocity bounds (qpOASES form) ------------------------
    int st_code = ok ? WCQP_STATUS_SOLVED : WCQP_STATUS_NUMERIC;
    int it = 0;
    bool in_w = false;
    double my_sig = 0.0;
    const double tol = prm->tol;
    lo = var ? lo : -inf;
    hi = var ? hi : inf;
    const bool need = !osqp_form && var && i >= 6 && fmax(nu - hi, lo - nu) > tol;
    if (__ballot(need) != 0ull) {
        const GiScratch w{S + L::OFF_RINV, S + L::OFF_V0, S + L::OFF_V1, S + L::OFF_V2,
                          S + L::OFF_R, S + L::OFF_C, reinterpret_cast<int*>(S + L::OFF_WI)};
        double* tkb = S + L::OFF_V3;
        int* info = reinterpret_cast<int*>(S + L::OFF_INFO);
        // tau_p = Z Hr^-1 Z' e_p: t over the free lanes first, then the basic lanes through F
        auto column_of_P = [&](int p, double sig) -> double {
            if (i == p) {
                info[0] = basic ? 1 : 0;
                info[1] = basic ? myrow : kap_i;
                if (!basic) {
#pragma unroll
                    for (int k = 0; k < NN; ++k) tkb[k] = -Hr[k];          // row == column (symmetric)
                }
            }
            wcqp::wave_lds_fence();
            const bool p_basic = info[0] != 0;
            const int p_idx = info[1];
            double tfree = 0.0;
            if (p_basic) {
#pragma unroll
                for (int k = 0; k < NN; ++k) tfree = fma(Hr[k], F_at(k, p_idx), tfree);   // -Hinv F[row_p,:]'
            } else if (free_var) {
                tfree = tkb[kap_i];
            }
            wcqp::wave_lds_fence();
            if (free_var) tkb[kap_i] = tfree;
            wcqp::wave_lds_fence();
            double tp = free_var ? tfree : 0.0;
            if (basic) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NN; ++k) acc = fma(-F_at(k, myrow), tkb[k], acc);
                tp = acc;
            }
            wcqp::wave_lds_fence();
            return tp * sig;
        };
        gi_active_set<KMAX, L::LDR>(w, i, half, var, lo, hi, tol, prm->max_iter, nu, st_code, it, in_w, my_sig,
                                            column_of_P);
    }

    WCQP_STAMP(8);
    // ---------------- outputs ---------------------------------------------------------------
    const unsigned long long bu = __ballot(in_w && my_sig > 0.0);
    const unsigned long long bl = __ballot(in_w && my_sig < 0.0);
    if (live) {
        if (i >= 6 && var) dq_out[inst * kDof + (i - 6)] = nu;
        if (i == 0) {
            status_out[inst] = st_code;
            if (aup_out) aup_out[inst] = (unsigned)((bu >> (32 * half)) & 0xffffffffull) >> 6;
            if (alo_out) alo_out[inst] = (unsigned)((bl >> (32 * half)) & 0xffffffffull) >> 6;
            if (iters_out) iters_out[inst] = it;
        }
    }
#ifdef WCQP_IK_STAMPS
    WCQP_STAMP(9);
    return;
#endif
    if (ferr_out) {
        // b - J nu for the 12 foot rows (osqp.cpp:430-454, qp.cpp:364-401): every lane multiplies its
        // column (reloaded, L2-resident) by its velocity, a [12][34] LDS tile turns the 32 partial
        // products of a row over to lane r
        static_assert(12 * 34 <= L::OFF_B, "the tile fits in front of the task rhs");
        double* pb = S;
        const int ic = var ? i : kNV - 1;
        const double v = var ? nu : 0.0;
        const double* jl = JL + inst * (6 * kNV) + ic;
        const double* jr = JR + inst * (6 * kNV) + ic;
        double part[12];
#pragma unroll
        for (int r = 0; r < 6; ++r) { part[r] = jl[r * kNV] * v; part[6 + r] = jr[r * kNV] * v; }
        wcqp::wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 12; ++r) pb[r * 34 + i] = part[r];
        wcqp::wave_lds_fence();
        if (i < 12 && live) {
            double acc = bvec[i];
#pragma unroll
            for (int k = 0; k < 32; k += 2) {
                const double2 p2 = *reinterpret_cast<const double2*>(pb + i * 34 + k);
                acc -= p2.x; acc -= p2.y;
            }
            ferr_out[inst * 12 + i] = acc;
        }
    }
}

}  // namespace

namespace wcqp_ik {

int ik2_launch(const IkDeviceParams* d_prm, bool use_com, bool use_mfma, int batch,
               const double* JL, const double* JR, const double* JN, const double* JC,
               const double* q, const double* state, double* dq, int* status,
               unsigned* alo, unsigned* aup, double* ferr, int* iters, hipStream_t stream) {
    const unsigned grid = (unsigned)((batch + 1) / 2);
    if (use_com && use_mfma)
        hipLaunchKernelGGL((ik2_kernel<true, true>), dim3(grid), dim3(64), 0, stream, d_prm, batch, JL, JR, JN, JC, q, state,
                           dq, status, alo, aup, ferr, iters);
    else if (use_com)
        hipLaunchKernelGGL((ik2_kernel<true, false>), dim3(grid), dim3(64), 0, stream, d_prm, batch, JL, JR, JN, JC, q, state,
                           dq, status, alo, aup, ferr, iters);
    else
        hipLaunchKernelGGL((ik2_kernel<false, false>), dim3(grid), dim3(64), 0, stream, d_prm, batch, JL, JR, JN, JC, q, state,
                           dq, status, alo, aup, ferr, iters);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
}

}  // namespace wcqp_ik
