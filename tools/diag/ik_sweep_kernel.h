// DIAGNOSTIC ONLY - the round-1 IK kernel (sweep on M = H + rho A'A, 32 lanes per instance), kept as an A/B baseline.
// Included by walking-controllers_amd/csrc/ik.hip when the library is built with -DWCQP_DIAG_KERNELS (tools/build_variant.sh);
// the product library does not carry it (wcqp_ik_create refuses WCQP_IK_ALG_SWEEP there).  Algorithm: see the header of ik.hip.
#pragma once
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
