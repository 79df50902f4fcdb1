// Device pointers that kernels read out of RECORDS in device memory (the tick pipeline's TickDev, the plan's step records).
//
// hipcc knows the address space of a pointer that is a kernel argument; one it has loaded from memory is a generic ("flat")
// pointer unless it can prove where it points, and every access through it becomes a flat_load / flat_store.  A flat access
// counts on BOTH vmcnt and lgkmcnt and may return out of order, so the compiler can only wait for "everything" behind it
// (s_waitcnt vmcnt(0) lgkmcnt(0)): the kernels' "state block first, the Jacobians stay in flight under the next phase" and every
// LDS wait while loads are outstanding turn into full drains.  GPtr<T> is a T* that says "global memory" when it is used.
#pragma once
#include <hip/hip_runtime.h>

namespace wcqp {

// `p` must be WAVE-UNIFORM (it is read with readfirstlane: free for a pointer that already sits in SGPRs, which is where a
// pointer out of a uniform record lives).  A plain cast to address space 1 and back is folded away before it can tell the
// compiler anything; the empty asm on the SGPR pair keeps it.
template <class T>
__host__ __device__ __forceinline__ T* as_global(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long u = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    u = ((unsigned long long)hi << 32) | lo;
    __attribute__((address_space(1))) T* q = (__attribute__((address_space(1))) T*)u;
    __asm__("" : "+s"(q));
    return (T*)q;
#else
    return p;
#endif
}

template <class T>
struct GPtr {
    T* p;
    GPtr() = default;
    __host__ __device__ GPtr(T* q) : p(q) {}
    __host__ __device__ __forceinline__ operator T*() const { return as_global(p); }
    __host__ __device__ __forceinline__ T* get() const { return as_global(p); }
};

}  // namespace wcqp
