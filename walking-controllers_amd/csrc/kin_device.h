// Kinematic-tree model in the form the device code reads, and the small 3-D helpers of the kinematics code: shared by
// the stand-alone kinematics kernel (kin.hip) and the kinematics phase fused into the tick kernel (ik4.hip).  Internal, not ABI.
#pragma once
#include "wcqp_internal.h"

namespace wcqp_kin {

constexpr int kMaxDof = WCQP_KIN_MAX_DOF;     // 32
constexpr int kMaxRounds = 5;                 // pointer jumping covers 2^5 = 32 >= kMaxDof levels

struct KinDev {
    int dof, n_rounds, dfs_contig;
    int up[kMaxRounds][kMaxDof];              // up[0] = parent, up[r + 1][j] = up[r][up[r][j]] (-1: above the root)
    int sub_end[kMaxDof];                     // last joint of j's subtree when the subtrees are index ranges (dfs_contig)
    unsigned desc_mask[kMaxDof];              // joints moved by joint j (itself included)
    unsigned path_mask[3];                    // joints on the path root -> frame f
    double R0[kMaxDof][9], p0[kMaxDof][3], axis[kMaxDof][3], mass[kMaxDof], com[kMaxDof][3];
    double root_mass, root_com[3], total_mass;
    int frame_joint[3];
    double frame_R[3][9], frame_p[3][3];
};

#if defined(__HIPCC__)
__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ __forceinline__ void mat3_vec(const double* A, const double* v, double* o) {
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = A[3 * r] * v[0] + A[3 * r + 1] * v[1] + A[3 * r + 2] * v[2];
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
// (Ro, po) = (Ra, pa) o (Rb, pb)
__device__ __forceinline__ void frame_mul(const double* Ra, const double* pa, const double* Rb, const double* pb, double* Ro, double* po) {
    mat3_mul(Ra, Rb, Ro);
    double d[3];
    mat3_vec(Ra, pb, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) po[k] = pa[k] + d[k];
}
// local frame of a revolute joint: R0 * Rot(axis, q)   (Rodrigues)
// sin and cos of a joint angle: Cody-Waite reduction by pi/2 in two FMA steps and the fdlibm kernels on |r| <= pi/4 (their
// published minimax coefficients; < 1 ulp each).  Joint angles are a few radians at most: the reduction's error is
// |k| x 2^-107, nothing the Payne-Hanek branch of ocml's sincos (154 VALU instructions and two branches per call) would add to.
__device__ __forceinline__ void joint_sincos(double x, double& sn, double& cs) {
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    const double s = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * pc);
    const int n = (int)k & 3;
    const double a = (n & 1) ? c : s, b = (n & 1) ? s : c;
    sn = (n & 2) ? -a : a;
    cs = ((n + 1) & 2) ? -b : b;
}
__device__ __forceinline__ void joint_rotation(const double* R0, const double* ax, double q, double* Ra) {
    double sn, cs;
    joint_sincos(q, sn, cs);
    const double v = 1.0 - cs;
    const double Rq[9] = {cs + v * ax[0] * ax[0],         v * ax[0] * ax[1] - sn * ax[2], v * ax[0] * ax[2] + sn * ax[1],
                          v * ax[1] * ax[0] + sn * ax[2], cs + v * ax[1] * ax[1],         v * ax[1] * ax[2] - sn * ax[0],
                          v * ax[2] * ax[0] - sn * ax[1], v * ax[2] * ax[1] + sn * ax[0], cs + v * ax[2] * ax[2]};
    mat3_mul(R0, Rq, Ra);
}
// lane i of a DPP row receives lane i - N (0 below the row start)
template <int N>
__device__ __forceinline__ double row_shr0(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over each 16-lane DPP row
__device__ __forceinline__ double row_scan(double v) {
    v += row_shr0<1>(v);
    v += row_shr0<2>(v);
    v += row_shr0<4>(v);
    v += row_shr0<8>(v);
    return v;
}
#endif

}  // namespace wcqp_kin
