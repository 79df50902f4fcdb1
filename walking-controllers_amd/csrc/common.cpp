// Host-side helpers shared by libwcqp's translation units.
#include <cmath>
#include "wcqp_internal.h"

namespace wcqp {

bool lu_factor(std::vector<double>& a, int n, std::vector<int>& piv) {
    piv.resize(n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(a[(size_t)k * n + k]);
        for (int r = k + 1; r < n; ++r) {
            const double v = std::fabs(a[(size_t)r * n + k]);
            if (v > best) { best = v; p = r; }
        }
        if (!(best > 0.0)) return false;
        piv[k] = p;
        if (p != k)
            for (int c = 0; c < n; ++c) std::swap(a[(size_t)k * n + c], a[(size_t)p * n + c]);
        const double inv = 1.0 / a[(size_t)k * n + k];
        for (int r = k + 1; r < n; ++r) {
            const double f = a[(size_t)r * n + k] * inv;
            a[(size_t)r * n + k] = f;
            if (f == 0.0) continue;
            const double* src = &a[(size_t)k * n];
            double* dst = &a[(size_t)r * n];
            for (int c = k + 1; c < n; ++c) dst[c] -= f * src[c];
        }
    }
    return true;
}

void lu_solve(const std::vector<double>& lu, const std::vector<int>& piv, int n, double* b) {
    // rows were swapped whole (LAPACK getrf convention): permute b completely first
    for (int k = 0; k < n; ++k)
        if (piv[k] != k) std::swap(b[k], b[piv[k]]);
    for (int k = 0; k < n; ++k) {
        const double bk = b[k];
        if (bk != 0.0)
            for (int r = k + 1; r < n; ++r) b[r] -= lu[(size_t)r * n + k] * bk;
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = b[k];
        for (int c = k + 1; c < n; ++c) s -= lu[(size_t)k * n + c] * b[c];
        b[k] = s / lu[(size_t)k * n + k];
    }
}

int DeviceScratch::reserve(size_t bytes) {
    if (bytes <= cap) return WCQP_OK;
    if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
    if (hipMalloc(&ptr, bytes) != hipSuccess) { ptr = nullptr; return WCQP_E_NOMEM; }
    cap = bytes;
    return WCQP_OK;
}

void DeviceScratch::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

}  // namespace wcqp

extern "C" {

const char* wcqp_strerror(int code) {
    switch (code) {
        case WCQP_OK: return "ok";
        case WCQP_E_INVALID: return "invalid argument";
        case WCQP_E_UNSUPPORTED: return "unsupported problem size";
        case WCQP_E_NUMERIC: return "constant precomputation failed (singular KKT)";
        case WCQP_E_HIP: return "HIP runtime error or no device (this path has no CPU fallback)";
        case WCQP_E_NOMEM: return "out of memory";
        default: return "unknown wcqp error";
    }
}

int wcqp_version(void) { return WCQP_VERSION; }

int wcqp_qp_enqueue_steps(wcqp_mpc_t mpc, wcqp_ik_t ik, int32_t batch, int32_t n_steps, const wcqp_qp_step* steps, int32_t* n_done) {
    if (n_done) *n_done = 0;
    if (n_steps < 0 || (n_steps > 0 && !steps)) return WCQP_E_INVALID;
    for (int32_t k = 0; k < n_steps; ++k) {
        const wcqp_qp_step& s = steps[k];
        {
            const int rc = wcqp::qp_pair_enqueue(mpc, ik, batch, s);      // one launch for both calls when they share a stream
            if (rc == WCQP_OK) { if (n_done) *n_done = k + 1; continue; }
            if (rc != WCQP_E_UNSUPPORTED) return rc;
        }
        if (s.x0) {
            const int rc = wcqp_mpc_solve_device(mpc, batch, s.x0, s.ref, s.ref_len, s.u_prev, s.hull_A, s.hull_b, s.hull_nc,
                                                 s.u0, s.mpc_status, s.mpc_active, s.mpc_margin, s.mpc_stream);
            if (rc != WCQP_OK) return rc;
        }
        if (s.J_left) {
            const int rc = wcqp_ik_solve_device(ik, batch, s.J_left, s.J_right, s.J_neck, s.J_com, s.q, s.state,
                                                s.dq, s.ik_status, s.active_lower, s.active_upper, s.foot_err, s.iters, s.ik_stream);
            if (rc != WCQP_OK) return rc;
        }
        if (n_done) *n_done = k + 1;
    }
    return WCQP_OK;
}

// ---- shard slabs (include/wcqp.h): the multi-GPU exchange format; host arithmetic only
int wcqp_slab_layout_for(int32_t batch, int32_t ref_len, wcqp_slab_layout* out) {
    if (!out || batch < 1 || ref_len < 1) return WCQP_E_INVALID;
    const int64_t B = batch;
    const int64_t in_sz[WCQP_SLAB_IN_ARRAYS] = {B * 2 * 8, B * ref_len * 2 * 8, B * 2 * 8, B * WCQP_HULL_ROWS * 2 * 8, B * WCQP_HULL_ROWS * 8, B * 4,
                                                B * 6 * 29 * 8, B * 6 * 29 * 8, B * 3 * 29 * 8, B * 3 * 29 * 8, B * 23 * 8, B * WCQP_IK_STATE_LEN * 8};
    const int64_t out_sz[WCQP_SLAB_OUT_ARRAYS] = {B * 2 * 8, B * 8, B * 23 * 8, B * 4, B * 4, B * 4, B * 4, B * 4, B * 4};
    out->batch = batch; out->ref_len = ref_len;
    int64_t off = 0;
    for (int k = 0; k < WCQP_SLAB_IN_ARRAYS; ++k) { out->in_offset[k] = off; off += (in_sz[k] + 255) / 256 * 256; }
    out->in_bytes = off;
    off = 0;
    for (int k = 0; k < WCQP_SLAB_OUT_ARRAYS; ++k) { out->out_offset[k] = off; off += (out_sz[k] + 255) / 256 * 256; }
    out->out_bytes = off;
    return WCQP_OK;
}

int wcqp_qp_step_from_slabs(const wcqp_slab_layout* L, const void* in_slab, void* out_slab, wcqp_qp_step* s) {
    if (!L || !in_slab || !out_slab || !s || L->batch < 1 || L->ref_len < 1) return WCQP_E_INVALID;
    if (((uintptr_t)in_slab | (uintptr_t)out_slab) & 15u) return WCQP_E_INVALID;        // 16-byte vector loads; the arrays inside sit on 256-byte offsets
    const char* in = static_cast<const char*>(in_slab);
    char* o = static_cast<char*>(out_slab);
    auto d = [&](int k) { return reinterpret_cast<const double*>(in + L->in_offset[k]); };
    *s = wcqp_qp_step{};
    s->x0 = d(0); s->ref = d(1); s->ref_len = L->ref_len; s->u_prev = d(2); s->hull_A = d(3); s->hull_b = d(4);
    s->hull_nc = reinterpret_cast<const int32_t*>(in + L->in_offset[5]);
    s->J_left = d(6); s->J_right = d(7); s->J_neck = d(8); s->J_com = d(9); s->q = d(10); s->state = d(11);
    s->u0 = reinterpret_cast<double*>(o + L->out_offset[0]); s->mpc_margin = reinterpret_cast<double*>(o + L->out_offset[1]);
    s->dq = reinterpret_cast<double*>(o + L->out_offset[2]);
    s->mpc_status = reinterpret_cast<int32_t*>(o + L->out_offset[3]); s->mpc_active = reinterpret_cast<uint32_t*>(o + L->out_offset[4]);
    s->ik_status = reinterpret_cast<int32_t*>(o + L->out_offset[5]);
    s->active_lower = reinterpret_cast<uint32_t*>(o + L->out_offset[6]); s->active_upper = reinterpret_cast<uint32_t*>(o + L->out_offset[7]);
    s->iters = reinterpret_cast<int32_t*>(o + L->out_offset[8]);
    return WCQP_OK;
}

int wcqp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int wcqp_stream_create(void** out) {
    if (!out) return WCQP_E_INVALID;
    hipStream_t s = nullptr;
    WCQP_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = s;
    return WCQP_OK;
}

int wcqp_stream_destroy(void* stream) {
    if (!stream) return WCQP_E_INVALID;
    WCQP_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return WCQP_OK;
}

int wcqp_stream_synchronize(void* stream) {
    if (stream) WCQP_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    else WCQP_HIP_TRY(hipDeviceSynchronize());
    return WCQP_OK;
}

}  // extern "C"
