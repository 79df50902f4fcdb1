// Jacobian QP-IK: one HIP kernel, 32 lanes per robot instance (two instances per wave64).
//
// Reference path replaced (citations relative to /root/reference/modules/Walking_module):
//   constants  WalkingQPIK::initializeMatrices          src/WalkingQPInverseKinematics.cpp:25-116
//   per tick   WalkingQPIK_osqp::{setHessianMatrix,setGradientVector,setLinearConstraintMatrix,
//              setBounds,solve,getSolution,get*FootError} src/WalkingQPInverseKinematics_osqp.cpp:135-454
//              WalkingQPIK_qpOASES::{same}               src/WalkingQPInverseKinematics_qpOASES.cpp:135-401
//
// QP, nu = [v_base(6); dq(dof)], n = dof + 6 = 29:
//      min 1/2 nu'H nu + g'nu   s.t.  A nu = b,   v_min <= dq <= v_max (qpOASES form only)
//      H = Lambda + Jn'Wn Jn (+ Jc'Wc Jc)  is only PSD (rank 26/29), so the kernel works on
//      M = H + rho A'A  (PD whenever the KKT matrix is regular; identical optimum and
//      identical multipliers because A nu = b holds at every iterate).
//
// Layout: lane i of a 32-lane group owns variable i: row i of the symmetric matrices lives
// in its REGISTERS (static indices, fully unrolled), and the only cross-lane traffic is one
// published column per elimination step, written to LDS once and read back as a broadcast.
//   1. M rows        M[i][:] = Lambda_i e_i + sum_r Cl[r][i] * Cr[r][:]     (18 stacked task rows)
//   2. Minv          symmetric sweep operator over the 29 pivots
//   3. G+ = Minv [A' g~],  S+ = [A; g~'] G+,  Sinv by a second sweep (15 pivots)
//   4. equality optimum  lambda = -Sinv (A Minv g~ + b),  nu = -Minv g~ - G lambda
//   5. bounds        Goldfarb-Idnani dual active set expressed through columns of the
//                    projected inverse  P = Minv - G Sinv G'  (no refactorisation per change)
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include "ik_common.h"

// The sweep kernel below is the round-1 A/B baseline: compiled only into diagnostic builds (-DWCQP_DIAG_KERNELS,
// tools/build_variant.sh); the product library carries ik4 (default), ik3 (general fall-back) and ik2 (CoM as cost).
using namespace wcqp_ik;
#ifdef WCQP_DIAG_KERNELS
namespace {


constexpr int kLD = 30;                // leading dim of 29-wide LDS rows: even (16-B aligned b128
                                       // broadcasts) and 60 dwords mod 64 -> at most 2-way on row-per-lane reads
constexpr int kLDG = 18;               // leading dim of G+ rows (<= 16 columns)
constexpr int kRows = 18;              // stacked task rows: J_left 6 | J_right 6 | J_com 3 | J_neck 3

template <bool USE_COM>
struct IkLayout {
    static constexpr int MEQ = USE_COM ? 15 : 12;       // equality rows
    static constexpr int NC1 = MEQ + 1;                 // columns of G+ = [A' g~]
    static constexpr int KMAX = kNV - MEQ;              // most bounds that can be active at once
    static constexpr int LDS_S = MEQ + (MEQ & 1);       // leading dim of Sinv rows
    // ---- per-instance LDS map (doubles) ----
    static constexpr int OFF_CR = 0;                    // [18][kLD]   phases 1-5
    static constexpr int OFF_GM = OFF_CR + kRows * kLD; // [29][kLDG]  phases 4-5
    static constexpr int END_MAT = OFF_GM + kNV * kLDG;
    // phase 6 reuses the matrix area
    static constexpr int OFF_SV = 0;                    // [MEQ][LDS_S]
    static constexpr int LDR = KMAX | 1;
    static constexpr int OFF_RINV = OFF_SV + MEQ * LDS_S;   // [KMAX][LDR]
    static constexpr int OFF_GROW = OFF_RINV + KMAX * LDR + ((KMAX * LDR) & 1); // [16] one row of G
    static constexpr int OFF_R = OFF_GROW + 16;         // [32] dual step per slot
    static constexpr int OFF_C = OFF_R + 32;            // [32]
    static constexpr int OFF_WI = OFF_C + 32;           // [32] ints
    static constexpr int END_AS = OFF_WI + 16;
    static_assert(END_AS <= END_MAT, "active-set scratch must fit in the dead matrix area");
    static constexpr int OFF_ST = END_MAT;              // [112] state 87 + q 23; later 4 x [32] vectors
    static constexpr int OFF_V0 = OFF_ST;               // vbuf   (violations)
    static constexpr int OFF_V1 = OFF_ST + 32;          // sgbuf / zbuf
    static constexpr int OFF_V2 = OFF_ST + 64;          // tpbuf
    static constexpr int OFF_V3 = OFF_ST + 96;          // rowbuf [32]
    static constexpr int OFF_COL = OFF_ST;              // [2][32] published column, double-buffered; the sweeps
                                                        // run after the state block is dead and before phase 6
    static constexpr int OFF_B = OFF_ST + 128;          // [16] task rhs
    static constexpr int OFF_LAM = OFF_B + 16;          // [16] multipliers / rhs
    static constexpr int PER_INST = ((OFF_LAM + 16) + 1) & ~1;
    // two instances per workgroup, 8 workgroups per CU (2 waves per SIMD) must fit in 160 KiB
    static_assert(2 * PER_INST * 8 <= 20480, "LDS budget: 8 workgroups per CU");
};

// Symmetric sweep over pivots 0..SZ-1 of the matrix whose row i sits in `row` of lane i.
// On exit row = -(A^-1) row.  Lanes >= SZ must hold zero rows (they act as padding).
//
// Software-pipelined: the only serial chain is pivot -> reciprocal -> multiplier -> next
// pivot, and it runs entirely in registers (two crossbar broadcasts per step): column k+1
// is updated first and its pivot broadcast at once, while the bulk of step k's rank-1
// update waits for the published column k to come back from LDS.  `col` is double-buffered
// (2 x 32 doubles) so step k+1's publish never races step k's reads.
template <int SZ, int K, int NR>
__device__ __forceinline__ void sweep_step(double (&row)[NR], double* col, int i, double& piv, bool& ok) {
    double* cb = col + 32 * (K & 1);
    const double ck = row[K];
    cb[i] = ck;                                        // column K == row K (symmetry)
    ok = ok && (piv > 0.0);
    const double d = wcqp::fast_rcp(piv);
    const double f0 = ck * d;
    // lane K holds row K == the column itself: row - (1-d) col = d col, so one multiplier
    // serves every lane and no per-element select is needed
    const double f = (i == K) ? (1.0 - d) : f0;
    if constexpr (K + 1 < SZ) {
        const double cn = group_bcast<K + 1>(ck);      // M[K+1][K]
        row[K + 1] = fma(-f, cn, row[K + 1]);
        piv = group_bcast<K + 1>(row[K + 1]);          // next pivot, off the LDS round trip
    }
    wcqp::wave_lds_fence();
#pragma unroll
    for (int j = 0; j < SZ; j += 2) {
        const double2 c2 = *reinterpret_cast<const double2*>(cb + j);
        if (j != K && j != K + 1) row[j] = fma(-f, c2.x, row[j]);
        if (j + 1 < SZ && j + 1 != K && j + 1 != K + 1) row[j + 1] = fma(-f, c2.y, row[j + 1]);
    }
    row[K] = (i == K) ? -d : f0;
    if constexpr (K + 1 < SZ) sweep_step<SZ, K + 1, NR>(row, col, i, piv, ok);
}

template <int SZ, int NR>
__device__ __forceinline__ bool sweep_rows(double (&row)[NR], double* col, int i) {
    bool ok = true;
    double piv = group_bcast<0>(row[0]);
    sweep_step<SZ, 0, NR>(row, col, i, piv, ok);
    wcqp::wave_lds_fence();
    return ok;
}

template <bool USE_COM>
__global__ __launch_bounds__(64, 2)
void ik_kernel(const IkDeviceParams* __restrict__ prm, int batch,
               const double* __restrict__ JL, const double* __restrict__ JR,
               const double* __restrict__ JN, const double* __restrict__ JC,
               const double* __restrict__ qpos, const double* __restrict__ state,
               double* __restrict__ dq_out, int* __restrict__ status_out,
               unsigned* __restrict__ alo_out, unsigned* __restrict__ aup_out,
               double* __restrict__ ferr_out, int* __restrict__ iters_out)
{
    using L = IkLayout<USE_COM>;
    constexpr int MEQ = L::MEQ, NC1 = L::NC1, KMAX = L::KMAX;
    __shared__ __attribute__((aligned(16))) double smem[2][L::PER_INST];

    const int lane = threadIdx.x;
    const int half = lane >> 5;
    const int i = lane & 31;                       // variable owned by this lane
    const long inst_raw = (long)blockIdx.x * 2 + half;
    const bool live = inst_raw < batch;
    const long inst = live ? inst_raw : (long)batch - 1;
    double* S = smem[half];
    double* Cr = S + L::OFF_CR;
    double* Gm = S + L::OFF_GM;
    double* st = S + L::OFF_ST;
    double* col = S + L::OFF_COL;
    double* bvec = S + L::OFF_B;
    double* lamv = S + L::OFF_LAM;
    const double inf = std::numeric_limits<double>::infinity();
    const bool var = i < kNV;

    // ---------------- phase 0: loads --------------------------------------------------
    // column i of the stacked task Jacobian [J_left; J_right; J_com; J_neck]: consecutive
    // lanes read consecutive doubles of each 29-wide row.
    double cl[kRows];
    {
        const double* jl = JL + inst * (6 * kNV);
        const double* jr = JR + inst * (6 * kNV);
        const double* jc = JC + inst * (3 * kNV);
        const double* jn = JN + inst * (3 * kNV);
#pragma unroll
        for (int r = 0; r < 6; ++r) cl[r] = var ? jl[r * kNV + i] : 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) cl[6 + r] = var ? jr[r * kNV + i] : 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) cl[12 + r] = var ? jc[r * kNV + i] : 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) cl[15 + r] = var ? jn[r * kNV + i] : 0.0;
        const double* sp = state + inst * kStateLen;
        st[i] = sp[i];
        st[i + 32] = sp[i + 32];
        if (i + 64 < kStateLen) st[i + 64] = sp[i + 64];
        if (i < kDof) st[kStateLen + i] = qpos[inst * kDof + i];
    }
    // right operands of the M build: equality rows as they are, cost rows pre-multiplied by W
    if (i < kLD) {
#pragma unroll
        for (int r = 0; r < MEQ; ++r) Cr[r * kLD + i] = cl[r];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            Cr[(15 + r) * kLD + i] = prm->Wn[3 * r] * cl[15] + prm->Wn[3 * r + 1] * cl[16] + prm->Wn[3 * r + 2] * cl[17];
        if (!USE_COM) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
                Cr[(12 + r) * kLD + i] = prm->Wc[3 * r] * cl[12] + prm->Wc[3 * r + 1] * cl[13] + prm->Wc[3 * r + 2] * cl[14];
        }
    }
    wcqp::wave_lds_fence();

#if defined(WCQP_IK_PHASE_STOP) && WCQP_IK_PHASE_STOP == 1   /* diagnostic timing builds only (tools/phase_timing.sh) */
    { double accx = 0.0; _Pragma("unroll") for (int r = 0; r < kRows; ++r) accx += cl[r];  if (live && var) dq_out[inst * kDof + (i % kDof)] = accx; return; }
#endif
    // ---------------- phase 1: task rhs b (lane r < MEQ) and gradient g~ ----------------
    const bool osqp_form = prm->form == WCQP_IK_FORM_OSQP;
    double b_mine = 0.0;
    if (i < MEQ) {
        if (i < 12) {
            const int foot = i / 6, k = i % 6;           // 0 = left, 1 = right
            const double* p  = st + (foot ? 12 : 0);
            const double* R  = st + (foot ? 15 : 3);
            const double* pd = st + (foot ? 36 : 24);
            const double* Rd = st + (foot ? 39 : 27);
            const double* tw = st + (foot ? 81 : 75);
            const double corr = k < 3 ? prm->k_pos_foot * (p[k] - pd[k]) : prm->k_att_foot * rot_err(R, Rd, k - 3);
            // osqp back-end skips the correction when twist[0] == twist[1] == 0 (osqp.cpp:286-306)
            const bool skip = osqp_form && tw[0] == tw[1] && tw[0] == 0.0;
            b_mine = skip ? tw[k] : tw[k] - corr;
        } else {
            const int k = i - 12;                        // CoM rows (osqp.cpp:307-313, qp.cpp:273-279)
            b_mine = st[72 + k] - prm->k_pos_com * (st[66 + k] - st[69 + k]);
        }
        bvec[i] = b_mine;
    }
    wcqp::wave_lds_fence();
    double gt;   // g~_i = g_i - rho (A'b)_i
    {
        const double kap = prm->kappa * (-prm->k_neck);
        const double e0 = kap * rot_err(st + 48, st + 57, 0);
        const double e1 = kap * rot_err(st + 48, st + 57, 1);
        const double e2 = kap * rot_err(st + 48, st + 57, 2);
        // g = -Jn' Wn kappa(-k_neck e) - Lambda_g K (q_reg - q) [- Jc' Wc v_c]   osqp.cpp:181-196, qp.cpp:161-178
        // Cr rows 15..17 hold (Wn Jn)[:, i] for this lane's column
        double g = 0.0;
        if (var) {
            g = -(Cr[15 * kLD + i] * e0 + Cr[16 * kLD + i] * e1 + Cr[17 * kLD + i] * e2);
            if (i >= 6) g -= prm->kq[i] * (prm->qreg[i] - st[kStateLen + i - 6]);
            if (!USE_COM)
                g -= Cr[12 * kLD + i] * st[72] + Cr[13 * kLD + i] * st[73] + Cr[14 * kLD + i] * st[74];
        }
        double atb = 0.0;
#pragma unroll
        for (int r = 0; r < MEQ; ++r) atb = fma(cl[r], bvec[r], atb);
        gt = g - prm->rho * atb;
    }

#if defined(WCQP_IK_PHASE_STOP) && WCQP_IK_PHASE_STOP == 2   /* diagnostic timing builds only (tools/phase_timing.sh) */
    { double accx = 0.0; _Pragma("unroll") for (int r = 0; r < kRows; ++r) accx += cl[r]; accx += gt + b_mine; if (live && var) dq_out[inst * kDof + (i % kDof)] = accx; return; }
#endif
    // ---------------- phase 2: M rows ---------------------------------------------------
    double Mr[kNV];
#pragma unroll
    for (int j = 0; j < kNV; ++j) Mr[j] = 0.0;
    {
        const double rho = prm->rho;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const double own = (r < MEQ) ? rho * cl[r] : cl[r];
#pragma unroll
            for (int j = 0; j < kNV; j += 2) {
                const double2 c2 = *reinterpret_cast<const double2*>(Cr + r * kLD + j);
                Mr[j] = fma(own, c2.x, Mr[j]);
                if (j + 1 < kNV) Mr[j + 1] = fma(own, c2.y, Mr[j + 1]);
            }
            wcqp::pin_result(Mr[0]);   // keep the scheduler from hoisting every row's LDS reads
        }
        const double lam_i = prm->lam[i];
#pragma unroll
        for (int j = 0; j < kNV; ++j) Mr[j] += (i == j) ? lam_i : 0.0;
    }

#if defined(WCQP_IK_PHASE_STOP) && WCQP_IK_PHASE_STOP == 3   /* diagnostic timing builds only (tools/phase_timing.sh) */
    { double accx = 0.0; _Pragma("unroll") for (int r = 0; r < kRows; ++r) accx += cl[r]; accx += gt + b_mine; _Pragma("unroll") for (int j = 0; j < kNV; ++j) accx += Mr[j]; if (live && var) dq_out[inst * kDof + (i % kDof)] = accx; return; }
#endif
    // ---------------- phase 3: Minv -----------------------------------------------------
    bool ok = sweep_rows<kNV>(Mr, col, i);
#pragma unroll
    for (int j = 0; j < kNV; ++j) Mr[j] = -Mr[j];

#if defined(WCQP_IK_PHASE_STOP) && WCQP_IK_PHASE_STOP == 4   /* diagnostic timing builds only (tools/phase_timing.sh) */
    { double accx = 0.0; _Pragma("unroll") for (int r = 0; r < kRows; ++r) accx += cl[r]; accx += gt + b_mine; _Pragma("unroll") for (int j = 0; j < kNV; ++j) accx += Mr[j]; if (live && var) dq_out[inst * kDof + (i % kDof)] = accx; return; }
#endif
    // ---------------- phase 4: G+ = Minv [A' g~] ----------------------------------------
    if (i < kLD) Cr[MEQ * kLD + i] = var ? gt : 0.0;     // row MEQ of C+ := g~ (cost rows are dead)
    wcqp::wave_lds_fence();
    double Gr[NC1];
#pragma unroll
    for (int c = 0; c < NC1; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < kNV; j += 2) {
            const double2 c2 = *reinterpret_cast<const double2*>(Cr + c * kLD + j);
            acc = fma(Mr[j], c2.x, acc);
            if (j + 1 < kNV) acc = fma(Mr[j + 1], c2.y, acc);
        }
        Gr[c] = acc;
        wcqp::pin_result(Gr[c]);
    }
    if (var) {
#pragma unroll
        for (int c = 0; c < NC1; ++c) Gm[i * kLDG + c] = Gr[c];
    }
    wcqp::wave_lds_fence();

#if defined(WCQP_IK_PHASE_STOP) && WCQP_IK_PHASE_STOP == 5   /* diagnostic timing builds only (tools/phase_timing.sh) */
    { double accx = 0.0; _Pragma("unroll") for (int r = 0; r < kRows; ++r) accx += cl[r]; accx += gt + b_mine; _Pragma("unroll") for (int j = 0; j < kNV; ++j) accx += Mr[j]; _Pragma("unroll") for (int c = 0; c < NC1; ++c) accx += Gr[c]; if (live && var) dq_out[inst * kDof + (i % kDof)] = accx; return; }
#endif
    // ---------------- phase 5: S+ rows, Sinv, lambda, equality optimum -------------------
    double Sr[NC1];
    {
        const int cs = i <= MEQ ? i : MEQ;
#pragma unroll
        for (int d = 0; d < NC1; ++d) Sr[d] = 0.0;
#pragma unroll 1
        for (int k = 0; k < kNV; ++k) {
            const double a = Cr[cs * kLD + k];
#pragma unroll
            for (int d = 0; d < NC1; d += 2) {
                const double2 g2 = *reinterpret_cast<const double2*>(Gm + k * kLDG + d);
                Sr[d] = fma(a, g2.x, Sr[d]);
                if (d + 1 < NC1) Sr[d + 1] = fma(a, g2.y, Sr[d + 1]);
            }
        }
    }
    // rhs_d = (A Minv g~)_d + b_d sits in lane d < MEQ
    if (i < MEQ) lamv[i] = Sr[MEQ] + b_mine;
    if (i >= MEQ) {
#pragma unroll
        for (int d = 0; d < NC1; ++d) Sr[d] = 0.0;
    }
    ok = sweep_rows<MEQ>(Sr, col, i) && ok;              // Sr = -Sinv rows on lanes < MEQ
    {
        double lam_c = 0.0;                              // lambda_c = -sum_d Sinv[c][d] rhs_d
#pragma unroll
        for (int d = 0; d < MEQ; ++d) lam_c = fma(Sr[d], lamv[d], lam_c);
        wcqp::wave_lds_fence();
        if (i < MEQ) lamv[i] = lam_c;
        wcqp::wave_lds_fence();
    }
    double nu = -Gr[MEQ];
#pragma unroll
    for (int c = 0; c < MEQ; ++c) nu = fma(-Gr[c], lamv[c], nu);

#if defined(WCQP_IK_PHASE_STOP) && WCQP_IK_PHASE_STOP == 6   /* diagnostic timing builds only (tools/phase_timing.sh) */
    { double accx = 0.0; _Pragma("unroll") for (int r = 0; r < kRows; ++r) accx += cl[r]; accx += nu; _Pragma("unroll") for (int j = 0; j < kNV; ++j) accx += Mr[j]; _Pragma("unroll") for (int c = 0; c < NC1; ++c) accx += Gr[c] + Sr[c]; if (live && var) dq_out[inst * kDof + (i % kDof)] = accx; return; }
#endif
    // ---------------- phase 6: joint-velocity bounds (qpOASES form) ----------------------
    int st_code = ok ? WCQP_STATUS_SOLVED : WCQP_STATUS_NUMERIC;
    int it = 0;
    bool in_w = false;
    double my_sig = 0.0;
    const double lo = var ? prm->vlo[i] : -inf, hi = var ? prm->vhi[i] : inf;
    const double tol = prm->tol;
    const bool need = !osqp_form && var && i >= 6 && fmax(nu - hi, lo - nu) > tol;
    if (__ballot(need) != 0ull) {
        double* Sv = S + L::OFF_SV;
        double* rowbuf = S + L::OFF_V3;
        double* grow = S + L::OFF_GROW;
        const GiScratch w{S + L::OFF_RINV, S + L::OFF_V0, S + L::OFF_V1, S + L::OFF_V2,
                          S + L::OFF_R, S + L::OFF_C, reinterpret_cast<int*>(S + L::OFF_WI)};
        // E = G Sinv (row i in registers): publish Sinv rows once
        if (i < MEQ) {
#pragma unroll
            for (int d = 0; d < MEQ; ++d) Sv[i * L::LDS_S + d] = -Sr[d];
        }
        wcqp::wave_lds_fence();
        double Er[MEQ];
#pragma unroll
        for (int c = 0; c < MEQ; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < MEQ; ++d) acc = fma(Gr[d], Sv[d * L::LDS_S + c], acc);
            Er[c] = acc;
            wcqp::pin_result(Er[c]);
        }
        // tau_p = sig * P[:, p],  P = Minv - E G'
        auto column_of_P = [&](int p, double sig) -> double {
            if (i == p) {
#pragma unroll
                for (int j = 0; j < kNV; ++j) rowbuf[j] = Mr[j];
#pragma unroll
                for (int c = 0; c < MEQ; ++c) grow[c] = Gr[c];
            }
            wcqp::wave_lds_fence();
            double tp = var ? rowbuf[i] : 0.0;
#pragma unroll
            for (int c = 0; c < MEQ; ++c) tp = fma(-Er[c], grow[c], tp);
            wcqp::wave_lds_fence();
            return tp * sig;
        };
        gi_active_set<KMAX, L::LDR>(w, i, half, var, lo, hi, tol, prm->max_iter, nu, st_code, it, in_w, my_sig,
                                            column_of_P);
    }

    // ---------------- outputs ------------------------------------------------------------
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
    if (ferr_out) {
        // "foot errors" = b - J nu  (osqp.cpp:430-454, qp.cpp:364-401)
        double* nub = S + L::OFF_V0;
        wcqp::wave_lds_fence();
        nub[i] = var ? nu : 0.0;
        wcqp::wave_lds_fence();
        if (i < 12 && live) {
            const double* jrow = (i < 6 ? JL + inst * (6 * kNV) + i * kNV : JR + inst * (6 * kNV) + (i - 6) * kNV);
            double acc = b_mine;
            for (int k = 0; k < kNV; ++k) acc = fma(-jrow[k], nub[k], acc);
            ferr_out[inst * 12 + i] = acc;
        }
    }
}

}  // namespace
#endif  // WCQP_DIAG_KERNELS

// ======================================================================================
struct wcqp_ik_s {
    wcqp_ik_params p{};
    IkDeviceParams hp{};
    IkDeviceParams* d_prm = nullptr;
    wcqp::DeviceScratch scratch;
};

namespace {

int ensure_device(wcqp_ik_s* h) {
    if (h->d_prm) return WCQP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        std::fprintf(stderr, "[wcqp] no HIP device: the IK solve path has no CPU fallback\n");
        return WCQP_E_HIP;
    }
    WCQP_HIP_TRY(hipMalloc(&h->d_prm, sizeof(IkDeviceParams)));
    WCQP_HIP_TRY(hipMemcpy(h->d_prm, &h->hp, sizeof(IkDeviceParams), hipMemcpyHostToDevice));
    return WCQP_OK;
}

}  // namespace

namespace wcqp {
int ik_prepare(wcqp_ik_t h) { return h ? ensure_device(h) : WCQP_E_INVALID; }
const void* ik_device_params(wcqp_ik_t h) { return h ? h->d_prm : nullptr; }
bool ik_fast_ok(wcqp_ik_t h) { return h && h->hp.fast_ok != 0; }
}  // namespace wcqp

extern "C" {

int wcqp_ik_create(const wcqp_ik_params* params, wcqp_ik_t* out) {
    if (!params || !out) return WCQP_E_INVALID;
    if (params->dof != kDof) return WCQP_E_UNSUPPORTED;      // kernels are unrolled for iCub's 23 DoF
    if (params->form != WCQP_IK_FORM_QPOASES && params->form != WCQP_IK_FORM_OSQP) return WCQP_E_INVALID;
    if (params->algorithm < 0 || params->algorithm > WCQP_IK_ALG_BASE_ELIM) return WCQP_E_INVALID;
#ifndef WCQP_DIAG_KERNELS
    if (params->algorithm == WCQP_IK_ALG_SWEEP) return WCQP_E_UNSUPPORTED;      // diagnostic builds only
#endif
    if (params->jacobian_structure < WCQP_IK_JAC_AUTO || params->jacobian_structure > WCQP_IK_JAC_GENERAL) return WCQP_E_INVALID;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < r; ++c)          // the kernels use W J and J'W interchangeably: symmetric weights only
            if (params->neck_weight[3 * r + c] != params->neck_weight[3 * c + r] ||
                params->com_weight[3 * r + c] != params->com_weight[3 * c + r]) return WCQP_E_INVALID;
    wcqp_ik_s* h = new (std::nothrow) wcqp_ik_s();
    if (!h) return WCQP_E_NOMEM;
    h->p = *params;
    IkDeviceParams& d = h->hp;
    const double big = std::numeric_limits<double>::max();
    for (int i = 0; i < 32; ++i) {
        d.lam[i] = d.kq[i] = d.qreg[i] = 0.0;
        d.vlo[i] = -big; d.vhi[i] = big;                       // qp.cpp:39-43
    }
    for (int j = 0; j < kDof; ++j) {
        d.lam[6 + j] = params->joint_reg_weights[j];           // base.cpp:64-67
        d.kq[6 + j] = params->joint_reg_weights[j] * params->joint_reg_gains[j];   // base.cpp:70-72, 87-89
        d.qreg[6 + j] = params->joint_reg_rad[j];
        d.vlo[6 + j] = params->v_min[j];                       // qp.cpp:45-49
        d.vhi[6 + j] = params->v_max[j];
        if (!(params->v_min[j] <= params->v_max[j])) { delete h; return WCQP_E_INVALID; }
    }
    std::memcpy(d.Wn, params->neck_weight, sizeof(d.Wn));
    std::memcpy(d.Wc, params->com_weight, sizeof(d.Wc));
    d.k_pos_com = params->k_pos_com; d.k_pos_foot = params->k_pos_foot;
    d.k_att_foot = params->k_att_foot; d.k_neck = params->k_neck;
    // extra k_attFoot on the neck gradient term in the osqp back-end only (osqp.cpp:183,193)
    d.kappa = params->form == WCQP_IK_FORM_OSQP ? params->k_att_foot : 1.0;
    d.rho = params->rho > 0 ? params->rho : 1.0;
    d.tol = params->tol > 0 ? params->tol : 1e-12;
    d.form = params->form;
    d.max_iter = params->max_iter > 0 ? params->max_iter : 100;   // nWSR = 100, qp.cpp:312
    // base-eliminated kernel (ik4.hip): column scaling Lam^-1/2 and the factor W_neck = L L'
    d.fast_ok = params->use_com_as_constraint ? 1 : 0;
    for (int c = 0; c < 32; ++c) { d.sd[c] = 1.0; d.isd[c] = 1.0; }
    for (int j = 0; j < kDof; ++j) {
        const double w = params->joint_reg_weights[j];
        if (!(w > 0.0) || !std::isfinite(w)) { d.fast_ok = 0; continue; }
        d.sd[j] = std::sqrt(1.0 / w); d.isd[j] = std::sqrt(w);
    }
    {
        // Cholesky W = L L' (lower), stored as L' row-major; a neck weight that is not positive definite keeps the
        // general kernels
        const double* W = params->neck_weight;
        double L[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        bool pd = true;
        for (int c = 0; c < 3 && pd; ++c) {
            double dd = W[3 * c + c];
            for (int k = 0; k < c; ++k) dd -= L[3 * c + k] * L[3 * c + k];
            if (!(dd > 0.0) || !std::isfinite(dd)) { pd = false; break; }
            L[3 * c + c] = std::sqrt(dd);
            for (int r = c + 1; r < 3; ++r) {
                double v = W[3 * r + c];
                for (int k = 0; k < c; ++k) v -= L[3 * r + k] * L[3 * c + k];
                L[3 * r + c] = v / L[3 * c + c];
            }
        }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) d.Lt[3 * r + c] = pd ? L[3 * c + r] : 0.0;
        if (!pd) d.fast_ok = 0;
    }
    *out = h;
    return WCQP_OK;
}

int wcqp_ik_set_posture(wcqp_ik_t h, const double* joint_reg_rad) {
    if (!h || !joint_reg_rad) return WCQP_E_INVALID;
    for (int j = 0; j < kDof; ++j) { h->p.joint_reg_rad[j] = joint_reg_rad[j]; h->hp.qreg[6 + j] = joint_reg_rad[j]; }
    if (h->d_prm) {
        WCQP_HIP_TRY(hipDeviceSynchronize());
        WCQP_HIP_TRY(hipMemcpy(h->d_prm, &h->hp, sizeof(IkDeviceParams), hipMemcpyHostToDevice));
    }
    return WCQP_OK;
}

int wcqp_ik_destroy(wcqp_ik_t h) {
    if (!h) return WCQP_E_INVALID;
    if (h->d_prm) (void)hipFree(h->d_prm);
    h->scratch.release();
    delete h;
    return WCQP_OK;
}

int wcqp_ik_solve_device(wcqp_ik_t h, int32_t batch,
                         const double* J_left, const double* J_right, const double* J_neck, const double* J_com,
                         const double* q, const double* state,
                         double* dq, int32_t* status, uint32_t* active_lower, uint32_t* active_upper,
                         double* foot_err, int32_t* iters, void* stream) {
    if (!h || batch < 0) return WCQP_E_INVALID;
    if (!J_left || !J_right || !J_neck || !J_com || !q || !state || !dq || !status) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    const int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    // default: base elimination + range space (ik4.hip) for MIXED-representation Jacobians, with the general 16-lane
    // kernel behind it for instances (or handles) that do not qualify
    const bool want4 = h->p.algorithm == WCQP_IK_ALG_BASE_ELIM || h->p.algorithm == WCQP_IK_ALG_DEFAULT;
    if (want4 && h->hp.fast_ok && h->p.jacobian_structure != WCQP_IK_JAC_GENERAL) {
        const int rc4 = wcqp_ik::ik4_launch(h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state, dq, status,
                                            active_lower, active_upper, foot_err, iters, (hipStream_t)stream);
        if (rc4 != WCQP_OK || h->p.jacobian_structure == WCQP_IK_JAC_MIXED) return rc4;
        return wcqp_ik::ik3_launch_list(h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state, dq, status,
                                        active_lower, active_upper, foot_err, iters, (hipStream_t)stream);
    }
    const bool use16 = want4 || h->p.algorithm == WCQP_IK_ALG_NULLSPACE_16L;
    if (use16 && h->p.use_com_as_constraint)
        return wcqp_ik::ik3_launch(h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state, dq, status,
                                   active_lower, active_upper, foot_err, iters, (hipStream_t)stream);
    if (h->p.algorithm != WCQP_IK_ALG_SWEEP)
        return wcqp_ik::ik2_launch(h->d_prm, h->p.use_com_as_constraint != 0, h->p.algorithm != WCQP_IK_ALG_NULLSPACE, batch, J_left, J_right, J_neck, J_com,
                                   q, state, dq, status, active_lower, active_upper, foot_err, iters, (hipStream_t)stream);
#ifdef WCQP_DIAG_KERNELS
    const unsigned grid = (unsigned)((batch + 1) / 2);
    if (h->p.use_com_as_constraint)
        hipLaunchKernelGGL(ik_kernel<true>, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state,
                           dq, status, active_lower, active_upper, foot_err, iters);
    else
        hipLaunchKernelGGL(ik_kernel<false>, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           h->d_prm, batch, J_left, J_right, J_neck, J_com, q, state,
                           dq, status, active_lower, active_upper, foot_err, iters);
    WCQP_HIP_TRY(hipGetLastError());
    return WCQP_OK;
#else
    return WCQP_E_UNSUPPORTED;            // refused at create already
#endif
}

}  // extern "C"

namespace wcqp {
// wcqp_qp_enqueue_steps: both calls of a record as one launch when they go to the same stream and the IK handle runs the
// base-eliminated kernel; WCQP_E_UNSUPPORTED = "make the two calls" (not an error)
int qp_pair_enqueue(wcqp_mpc_t mpc, wcqp_ik_t h, int batch, const wcqp_qp_step& s) {
    if (!mpc || !h || batch < 1 || !s.x0 || !s.J_left || s.mpc_stream != s.ik_stream) return WCQP_E_UNSUPPORTED;
    const bool want4 = h->p.algorithm == WCQP_IK_ALG_BASE_ELIM || h->p.algorithm == WCQP_IK_ALG_DEFAULT;
    if (!(want4 && h->hp.fast_ok && h->p.jacobian_structure != WCQP_IK_JAC_GENERAL)) return WCQP_E_UNSUPPORTED;
    if (s.ref_len < 1 || !s.ref || !s.u_prev || !s.hull_A || !s.hull_b || !s.hull_nc || !s.u0 || !s.mpc_status) return WCQP_E_INVALID;
    if (!s.J_right || !s.J_neck || !s.J_com || !s.q || !s.state || !s.dq || !s.ik_status) return WCQP_E_INVALID;
    int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    rc = mpc_prepare(mpc);
    if (rc != WCQP_OK) return rc;
    wcqp_mpc::MpcDeviceConsts c;
    mpc_device_consts(mpc, &c);
    rc = wcqp_ik::ik4_launch_pair(h->d_prm, batch, s.J_left, s.J_right, s.J_neck, s.J_com, s.q, s.state, s.dq, s.ik_status,
                                  s.active_lower, s.active_upper, s.foot_err, s.iters,
                                  c, s.x0, s.ref, s.ref_len, s.u_prev, s.hull_A, s.hull_b, s.hull_nc, s.u0, s.mpc_status, s.mpc_active, s.mpc_margin,
                                  (hipStream_t)s.ik_stream);
    if (rc != WCQP_OK || h->p.jacobian_structure == WCQP_IK_JAC_MIXED) return rc;
    return wcqp_ik::ik3_launch_list(h->d_prm, batch, s.J_left, s.J_right, s.J_neck, s.J_com, s.q, s.state, s.dq, s.ik_status,
                                    s.active_lower, s.active_upper, s.foot_err, s.iters, (hipStream_t)s.ik_stream);
}
}  // namespace wcqp

// ---- plans of steps (include/wcqp.h: wcqp_qp_plan_*) ------------------------------------------------------------------
struct wcqp_qp_plan_s {
    wcqp_mpc_t mpc = nullptr;
    wcqp_ik_t ik = nullptr;
    int batch = 0, n_steps = 0, ways = 1;
    wcqp_qp_step* d_recs = nullptr;
    unsigned* d_queue = nullptr;      // ways = 0: ticket counters + waves done (ik_common.h: kPlanQueues; qp_plan_kernel zeroes them itself)
    int queue_grid = 0;
    bool mpc_only = false;            // every record without its IK part: mpc_plan_kernel (mpc.hip)
    bool ik_only = false;             // every record without its MPC part: ik_plan_kernel (ik4.hip)
};

extern "C" {

int wcqp_qp_plan_create(wcqp_mpc_t mpc, wcqp_ik_t ik, int32_t batch, int32_t n_steps, const wcqp_qp_step* steps, int32_t ways,
                        wcqp_qp_plan_t* out) {
    if (!out || batch < 1 || n_steps < 1 || !steps || ways < WCQP_PLAN_WAYS_AUTO) return WCQP_E_INVALID;
    if (ways == WCQP_PLAN_WAYS_AUTO) {
        // enough workgroups for the dispatcher to even out the launch's ends: >= 16384 (8 x the resident wavefronts of an MI355X),
        // at least 4 ways, at most one per record (bench.py's rule)
        const int groups = (batch + 3) / 4;
        ways = (16384 + groups - 1) / groups;
        if (ways < 4) ways = 4;
        if (ways > n_steps) ways = n_steps;
    }
    // MPC-only plan: NO record has an IK part (J_left == NULL everywhere; `ik` may be NULL): BASELINE config 2 on its own
    bool mpc_only = true;
    for (int k = 0; k < n_steps; ++k) mpc_only = mpc_only && !steps[k].J_left;
    // IK-only plan: NO record has an MPC part (x0 == NULL everywhere): BASELINE config 3 on its own (qp_plan_kernel without its MPC share)
    bool ik_only = true;
    for (int k = 0; k < n_steps; ++k) ik_only = ik_only && !steps[k].x0;
    if (mpc_only && ik_only) return WCQP_E_INVALID;
    if (!mpc && !ik_only) return WCQP_E_INVALID;         // (`mpc` may be NULL for an IK-only plan, `ik` for an MPC-only one)
    for (int k = 0; k < n_steps; ++k) {
        const wcqp_qp_step& s = steps[k];
        if (!ik_only && (!s.x0 || !s.ref || s.ref_len < 1 || !s.u_prev || !s.hull_A || !s.hull_b || !s.hull_nc || !s.u0 || !s.mpc_status)) return WCQP_E_INVALID;
        if (!mpc_only && (!s.J_left || !s.J_right || !s.J_neck || !s.J_com || !s.q || !s.state || !s.dq || !s.ik_status)) return WCQP_E_INVALID;
    }
    if (mpc_only) {
        if (ways < 1) return WCQP_E_UNSUPPORTED;        // the work-queue form belongs to the IK + MPC kernel
        int rc0 = wcqp::mpc_prepare(mpc);
        if (rc0 != WCQP_OK) return rc0;
        wcqp_qp_plan_s* p = new (std::nothrow) wcqp_qp_plan_s();
        if (!p) return WCQP_E_NOMEM;
        p->mpc = mpc; p->batch = batch; p->n_steps = n_steps; p->ways = ways < n_steps ? ways : n_steps; p->mpc_only = true;
        if (hipMalloc(reinterpret_cast<void**>(&p->d_recs), (size_t)n_steps * sizeof(wcqp_qp_step)) != hipSuccess) { delete p; return WCQP_E_NOMEM; }
        if (hipMemcpy(p->d_recs, steps, (size_t)n_steps * sizeof(wcqp_qp_step), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(p->d_recs); delete p; return WCQP_E_HIP;
        }
        *out = p;
        return WCQP_OK;
    }
    if (!ik) return WCQP_E_INVALID;
    // one launch walks through the records: that is the base-eliminated kernel on Jacobians the caller declares MIXED (no
    // fall-back launch behind it), with the MPC on the IK's lanes
    const bool want4 = ik->p.algorithm == WCQP_IK_ALG_BASE_ELIM || ik->p.algorithm == WCQP_IK_ALG_DEFAULT;
    if (!(want4 && ik->hp.fast_ok && ik->p.jacobian_structure == WCQP_IK_JAC_MIXED)) return WCQP_E_UNSUPPORTED;
    int rc = ensure_device(ik);
    if (rc == WCQP_OK && mpc) rc = wcqp::mpc_prepare(mpc);
    if (rc != WCQP_OK) return rc;
    wcqp_qp_plan_s* p = new (std::nothrow) wcqp_qp_plan_s();
    if (!p) return WCQP_E_NOMEM;
    p->mpc = mpc; p->ik = ik; p->batch = batch; p->n_steps = n_steps; p->ways = ways < n_steps ? ways : n_steps; p->ik_only = ik_only;
    if (hipMalloc(reinterpret_cast<void**>(&p->d_recs), (size_t)n_steps * sizeof(wcqp_qp_step)) != hipSuccess) { delete p; return WCQP_E_NOMEM; }
    if (hipMemcpy(p->d_recs, steps, (size_t)n_steps * sizeof(wcqp_qp_step), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p->d_recs); delete p; return WCQP_E_HIP;
    }
    if (ways == 0) {
        p->queue_grid = wcqp_ik::ik4_plan_queue_grid(batch, n_steps);
        if (p->queue_grid < 1 || hipMalloc(reinterpret_cast<void**>(&p->d_queue), wcqp_ik::kPlanQueueBytes) != hipSuccess ||
            hipMemset(p->d_queue, 0, wcqp_ik::kPlanQueueBytes) != hipSuccess) {
            if (p->d_queue) (void)hipFree(p->d_queue);
            (void)hipFree(p->d_recs); delete p; return WCQP_E_HIP;
        }
    }
    *out = p;
    return WCQP_OK;
}

int wcqp_qp_plan_enqueue(wcqp_qp_plan_t p, void* stream) {
    if (!p) return WCQP_E_INVALID;
    if (p->mpc_only) return wcqp::mpc_launch_plan(p->mpc, p->batch, p->d_recs, p->n_steps, p->ways, (hipStream_t)stream);
    wcqp_mpc::MpcDeviceConsts c{};
    if (p->mpc) wcqp::mpc_device_consts(p->mpc, &c);        // (an IK-only plan never reads them)
    return wcqp_ik::ik4_launch_plan(p->ik->d_prm, p->batch, p->d_recs, p->n_steps, p->ways, c, (hipStream_t)stream, p->d_queue, p->queue_grid, p->ik_only);
}

int wcqp_qp_plan_destroy(wcqp_qp_plan_t p) {
    if (!p) return WCQP_E_INVALID;
    if (p->d_recs) (void)hipFree(p->d_recs);
    if (p->d_queue) (void)hipFree(p->d_queue);
    delete p;
    return WCQP_OK;
}

}  // extern "C"

extern "C" {

int wcqp_ik_solve_host(wcqp_ik_t h, int32_t batch,
                       const double* J_left, const double* J_right, const double* J_neck, const double* J_com,
                       const double* q, const double* state,
                       double* dq, int32_t* status, uint32_t* active_lower, uint32_t* active_upper,
                       double* foot_err, int32_t* iters) {
    if (!h || batch < 0) return WCQP_E_INVALID;
    if (!J_left || !J_right || !J_neck || !J_com || !q || !state || !dq || !status) return WCQP_E_INVALID;
    if (batch == 0) return WCQP_OK;
    int rc = ensure_device(h);
    if (rc != WCQP_OK) return rc;
    const size_t B = (size_t)batch;
    const size_t o_jl = 0, o_jr = o_jl + B * 6 * kNV, o_jn = o_jr + B * 6 * kNV, o_jc = o_jn + B * 3 * kNV,
                 o_q = o_jc + B * 3 * kNV, o_st = o_q + B * kDof, o_dq = o_st + B * kStateLen,
                 o_fe = o_dq + B * kDof, n_dbl = o_fe + B * 12;
    rc = h->scratch.reserve(n_dbl * 8 + B * 4 * 4);
    if (rc != WCQP_OK) return rc;
    double* d = static_cast<double*>(h->scratch.ptr);
    int32_t* d_st = reinterpret_cast<int32_t*>(d + n_dbl);
    uint32_t* d_lo = reinterpret_cast<uint32_t*>(d_st + B);
    uint32_t* d_up = d_lo + B;
    int32_t* d_it = reinterpret_cast<int32_t*>(d_up + B);
    WCQP_HIP_TRY(hipMemcpy(d + o_jl, J_left, B * 6 * kNV * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_jr, J_right, B * 6 * kNV * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_jn, J_neck, B * 3 * kNV * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_jc, J_com, B * 3 * kNV * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_q, q, B * kDof * 8, hipMemcpyHostToDevice));
    WCQP_HIP_TRY(hipMemcpy(d + o_st, state, B * kStateLen * 8, hipMemcpyHostToDevice));
    rc = wcqp_ik_solve_device(h, batch, d + o_jl, d + o_jr, d + o_jn, d + o_jc, d + o_q, d + o_st,
                              d + o_dq, d_st, d_lo, d_up, foot_err ? d + o_fe : nullptr, d_it, nullptr);
    if (rc != WCQP_OK) return rc;
    WCQP_HIP_TRY(hipDeviceSynchronize());
    WCQP_HIP_TRY(hipMemcpy(dq, d + o_dq, B * kDof * 8, hipMemcpyDeviceToHost));
    WCQP_HIP_TRY(hipMemcpy(status, d_st, B * 4, hipMemcpyDeviceToHost));
    if (active_lower) WCQP_HIP_TRY(hipMemcpy(active_lower, d_lo, B * 4, hipMemcpyDeviceToHost));
    if (active_upper) WCQP_HIP_TRY(hipMemcpy(active_upper, d_up, B * 4, hipMemcpyDeviceToHost));
    if (foot_err) WCQP_HIP_TRY(hipMemcpy(foot_err, d + o_fe, B * 12 * 8, hipMemcpyDeviceToHost));
    if (iters) WCQP_HIP_TRY(hipMemcpy(iters, d_it, B * 4, hipMemcpyDeviceToHost));
    return WCQP_OK;
}

}  // extern "C"
