// Internal helpers shared by the translation units of libwcqp (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "wcqp.h"

#define WCQP_HIP_TRY(expr)                                                         \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) {                                                   \
            std::fprintf(stderr, "[wcqp] %s failed: %s (%s:%d)\n", #expr,          \
                         hipGetErrorString(e__), __FILE__, __LINE__);              \
            return WCQP_E_HIP;                                                     \
        }                                                                          \
    } while (0)

namespace wcqp_mpc { struct MpcDeviceConsts; }
namespace wcqp {
void mpc_device_consts(wcqp_mpc_t h, wcqp_mpc::MpcDeviceConsts* c);   // kernel-argument copy of the condensed constants (after mpc_prepare)

// Dense LU with partial pivoting, row-major, in place; returns false when singular.
bool lu_factor(std::vector<double>& a, int n, std::vector<int>& piv);
void lu_solve(const std::vector<double>& lu, const std::vector<int>& piv, int n, double* b);

// A growable device scratch buffer owned by a handle (used only by *_host entry points).
struct DeviceScratch {
    void*  ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
};

// internal launch of the MPC kernel with a strided / offset reference window (tick pipeline)
int mpc_enqueue(wcqp_mpc_t h, int batch, const double* x0, const double* ref, int ref_len, int ref_stride,
                const int* ref_start_dev, const double* u_prev,
                const double* hull_A, const double* hull_b, const int* hull_nc, int hull_sets, const int* hull_sel,
                double* u0, int* status, unsigned* active, double* margin, hipStream_t stream);
int mpc_horizon(wcqp_mpc_t h);
int mpc_prepare(wcqp_mpc_t h);     // uploads the handle's device constants now (graph capture forbids it later)
int mpc_launch_plan(wcqp_mpc_t h, int batch, const wcqp_qp_step* d_recs, int n_steps, int ways, hipStream_t stream);   // MPC-only plan (mpc.hip)
int ik_prepare(wcqp_ik_t h);
const void* ik_device_params(wcqp_ik_t h);     // IkDeviceParams* in HBM (after ik_prepare)
int qp_pair_enqueue(wcqp_mpc_t mpc, wcqp_ik_t ik, int batch, const wcqp_qp_step& s);   // WCQP_E_UNSUPPORTED: make the two calls instead
bool ik_fast_ok(wcqp_ik_t h);                  // the handle qualifies for the base-eliminated kernel (ik4.hip)
void mpc_dynamics(wcqp_mpc_t h, double* a, double* b);
int kin_prepare(wcqp_kin_t h);
// hull.hip: the three support-polygon row sets (left, right, both feet in contact) of every robot from the desired foot poses of its pose block
int hull_tables_from_state(int batch, const double* foot_rect_host, const double* state_dev, int state_len,
                           double* tab_A, double* tab_b, int* tab_nc, hipStream_t stream);
bool kin_compact_layout(wcqp_kin_t h, unsigned masks[3], int* stride, int* off_d);
bool kin_fused_tables(wcqp_kin_t h, std::vector<double>& tab, int* n_rounds);   // kin.hip: the model for the tick kernel's own kinematics phase   // kin.hip: layout of the tick's compact Jacobian hand-off

// ---- wave-level helpers used by the kernels (gfx950, wave64) -------------------------
#if defined(__HIPCC__)
// Orders LDS traffic between lanes of ONE wavefront: the LDS pipeline executes a
// wave's DS instructions in program order, so only the compiler has to be told
// not to move them across this point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// In a fully unrolled loop of "broadcast LDS reads -> FMA chain -> one result" hipcc (ROCm 7.2)
// issues every iteration's reads up front and sinks the FMA chains to the first use of the
// results, so hundreds of loaded values are live at once and get spilled to scratch
// (sched_barrier does not help: the IR is already reordered).  Passing each iteration's result
// through an empty volatile asm with a memory clobber pins both: the reads cannot cross it and
// the chain has to finish in front of it.
__device__ __forceinline__ void pin_result(double& x) { __asm__ volatile("" : "+v"(x) :: "memory"); }
// Same for a 32-bit value, without the memory clobber: makes hipcc materialise `x` HERE instead of
// keeping everything it was computed from alive to rebuild it later (it otherwise carries the 15
// pivot keys of the Gauss-Jordan through the whole elimination to derive each lane's row at the end).
__device__ __forceinline__ void pin_value(int& x) { __asm__ volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin_value(unsigned& x) { __asm__ volatile("" : "+v"(x)); }

// One dword through DPP.  For the controls whose every lane has a source inside its own row (quad_perm 0x00-0xff, row_mirror 0x140,
// row_half_mirror 0x141, row_newbcast 0x150-0x15f) the "old" operand is never read: with update_dpp(0, ...) hipcc still materialises
// that 0 in the destination in front of EVERY DPP move (two v_mov per double moved: a third of a butterfly step's instructions);
// mov_dpp leaves it undefined.  Shifts (row_shl / row_shr: lanes without a source keep `old`) stay on update_dpp with old = 0.
template <int CTRL>
__device__ __forceinline__ int dpp_dword(int v) {
    if constexpr (CTRL <= 0xFF || CTRL == 0x140 || CTRL == 0x141 || (CTRL >= 0x150 && CTRL <= 0x15F)) return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false);
    else return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

// `base` + a 32-bit BYTE offset.  With a wave-uniform base (an SGPR pair: a kernel argument, or a pointer out of a record through
// gptr.h) and a zero-extended 32-bit per-lane offset the access is `global_load ... v_off, s[base:base+1] offset:imm` - no 64-bit
// VALU arithmetic at all, and constant element offsets behind it fold into the immediate.  Written as `ptr + long_index * stride`
// the same address costs a 64-bit multiply-add and a 64-bit add per array (~8 VALU instructions each, ~150 per robot-tick record
// in the solve kernels' load phase).  The caller guarantees that the offset fits 32 bits (the host entry points check batch x stride).
template <class T>
__device__ __forceinline__ const T* at32(const T* base, unsigned byte_off) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off); }
template <class T>
__device__ __forceinline__ T* at32(T* base, unsigned byte_off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off); }

// 1/x: v_rcp_f64 seed + two Newton steps (error <= ~1 ulp; the QPs need 1e-9, not
// correctly-rounded division, and the IEEE division sequence is ~3x longer).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
// 1/sqrt(x), x > 0: v_rsq_f64 seed + two Newton steps y <- y (1.5 - 0.5 x y^2) (error ~1 ulp; the IEEE sqrt + division pair it
// replaces is ~4x longer)
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * __builtin_fma(-hx * y, y, 1.5);
    y = y * __builtin_fma(-hx * y, y, 1.5);
    return y;
}
#endif

}  // namespace wcqp
