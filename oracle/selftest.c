/* ORACLE self-test under AddressSanitizer / UBSan (CPU only; test infrastructure).
 * Builds wc_oracle.c with -fsanitize=address,undefined and runs both batch entry points on a
 * small deterministic problem set, so that memory errors in the checker itself cannot hide. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int N; double dT, com_height, gravity, Q[4], R[4]; } orc_mpc_params;
typedef struct { int dof, use_com, form; double Wc[9], Wn[9]; double w[32], gains[32], qreg[32], vmin[32], vmax[32];
                 double k_pos_com, k_pos_foot, k_att_foot, k_neck; } orc_ik_params;
int orc_mpc_batch_osqp(const orc_mpc_params*, int, const double*, const double*, int, const double*, const double*,
                       const double*, const int*, double*, int*, int*, int);
int orc_ik_batch(const orc_ik_params*, int, const double*, const double*, const double*, const double*, const double*,
                 const double*, double*, int*, uint32_t*, uint32_t*, int*, int);

int orc_mpc_batch_condensed(int, const double*, const double*, const double*, const double*, double, int, const double*, const double*, int,
                            const double*, const double*, const double*, const int*, double*, uint32_t*, int*, int);
int orc_ik_batch_range_space(const orc_ik_params*, int, const double*, const double*, const double*, const double*, const double*,
                             const double*, double*, int*, uint32_t*, uint32_t*, int*, int);

static unsigned long long s_ = 88172645463325252ull;
static double rnd(void) { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return (double)(s_ >> 11) / 9007199254740992.0 * 2.0 - 1.0; }

int main(void) {
    enum { B = 6, N = 20, DOF = 23, NV = 29 };
    orc_mpc_params mp = {N, 0.01, 0.53, 9.81, {7500, 0, 0, 7500}, {9e6, 0, 0, 9e6}};
    double x0[B * 2], ref[B * (N + 1) * 2], up[B * 2], hA[B * 16], hb[B * 8], u0[B * 2];
    int nc[B], it[B], st[B];
    for (int i = 0; i < B; ++i) {
        x0[2 * i] = 0.01 * rnd(); x0[2 * i + 1] = 0.01 * rnd(); up[2 * i] = 0.04 * rnd(); up[2 * i + 1] = 0.03 * rnd();
        for (int k = 0; k <= N; ++k) { ref[(i * (N + 1) + k) * 2] = x0[2 * i] + 0.001 * k; ref[(i * (N + 1) + k) * 2 + 1] = x0[2 * i + 1]; }
        const double A4[8] = {1, 0, 0, 1, -1, 0, 0, -1}, b4[4] = {0.05, 0.025, 0.02, 0.025};
        memset(hA + i * 16, 0, sizeof(double) * 16);
        for (int k = 0; k < 8; ++k) hb[i * 8 + k] = 1e30;
        memcpy(hA + i * 16, A4, sizeof(A4)); memcpy(hb + i * 8, b4, sizeof(b4));
        nc[i] = 4;
    }
    int fail = orc_mpc_batch_osqp(&mp, B, x0, ref, N + 1, up, hA, hb, nc, u0, it, st, 2);
    printf("mpc fail %d u0[0] %.6f %.6f iters %d\n", fail, u0[0], u0[1], it[0]);
    {   /* part (4): the condensed MPC with made-up (but well-formed) gains: this run is about memory safety, not values */
        double Gr[(N + 1) * 4], Gx[4] = {0.6, 0, 0, 0.6}, Gu[4] = {0.1, 0, 0, 0.1}, S0[4] = {1e-7, 0, 0, 1e-7};
        uint32_t act[B];
        for (int k = 0; k <= N; ++k) { Gr[4 * k] = Gr[4 * k + 3] = 0.3 / (N + 1); Gr[4 * k + 1] = Gr[4 * k + 2] = 0.0; }
        up[0] = 0.3;                                        /* pushes one instance out of its polygon */
        fail = orc_mpc_batch_condensed(N, Gr, Gx, Gu, S0, 1e-10, B, x0, ref, N + 1, up, hA, hb, nc, u0, act, st, 2);
        printf("mpc condensed fail %d u0[0] %.6f %.6f active %u\n", fail, u0[0], u0[1], act[0]);
    }
    orc_ik_params ip; memset(&ip, 0, sizeof(ip));
    ip.dof = DOF; ip.use_com = 1; ip.k_pos_com = 1; ip.k_pos_foot = 4; ip.k_att_foot = 2; ip.k_neck = 1;
    for (int k = 0; k < 3; ++k) { ip.Wn[4 * k] = 5; ip.Wc[4 * k] = 100; }
    for (int j = 0; j < DOF; ++j) { ip.w[j] = 1; ip.gains[j] = 5; ip.qreg[j] = 0.1 * rnd(); ip.vmin[j] = -0.4; ip.vmax[j] = 0.4; }
    static double JL[B * 6 * NV], JR[B * 6 * NV], JN[B * 3 * NV], JC[B * 3 * NV], q[B * DOF], state[B * 87], dq[B * DOF];
    uint32_t lo[B], up2[B];
    for (int i = 0; i < B; ++i) {
        for (int r = 0; r < 6; ++r) for (int c = 0; c < NV; ++c) {
            JL[(i * 6 + r) * NV + c] = (c == r) ? 1.0 : (c < 6 ? 0.0 : 0.3 * rnd());
            JR[(i * 6 + r) * NV + c] = (c == r) ? 1.0 : (c < 6 ? 0.0 : 0.3 * rnd());
        }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < NV; ++c) {
            JN[(i * 3 + r) * NV + c] = (c == 3 + r) ? 1.0 : (c < 6 ? 0.0 : 0.3 * rnd());
            JC[(i * 3 + r) * NV + c] = (c == r) ? 1.0 : (c < 6 ? 0.0 : 0.05 * rnd());
        }
        for (int j = 0; j < DOF; ++j) q[i * DOF + j] = 0.2 * rnd();
        double* s = state + i * 87;
        memset(s, 0, sizeof(double) * 87);
        for (int k = 0; k < 6; ++k) { const int o[6] = {3, 15, 27, 39, 48, 57}; s[o[k]] = s[o[k] + 4] = s[o[k] + 8] = 1.0; }
        for (int k = 0; k < 3; ++k) { s[k] = 0.005 * rnd(); s[72 + k] = 0.05 * rnd(); s[81 + k] = 0.2 * rnd(); }
    }
    for (int form = 0; form < 2; ++form) {
        ip.form = form;
        fail = orc_ik_batch(&ip, B, JL, JR, JN, JC, q, state, dq, st, lo, up2, it, 2);
        printf("ik form %d fail %d dq[0] %.6f iters %d\n", form, fail, dq[0], it[0]);
        for (int j = 0; j < DOF; ++j) { ip.vmin[j] = form ? -0.4 : -0.08; ip.vmax[j] = form ? 0.4 : 0.08; }      /* tight: the active set walks */
        fail = orc_ik_batch_range_space(&ip, B, JL, JR, JN, JC, q, state, dq, st, lo, up2, it, 2);
        printf("ik range space form %d fail %d dq[0] %.6f iters %d\n", form, fail, dq[0], it[0]);
        for (int j = 0; j < DOF; ++j) { ip.vmin[j] = -0.4; ip.vmax[j] = 0.4; }
    }
    return 0;
}
