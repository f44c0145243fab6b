// Shared device helpers for libevk (gfx950 / CDNA4 only; wavefront = 64).
// Build flags that matter for parity (see csrc/build.py): -ffp-contract=off (the reference evaluates
// x - dt*v, w*(1-dx)*(1-dy), ... as separate roundings; a fused multiply-add would change per-event values) and
// -munsafe-fp-atomics (hardware global_atomic_add_f32/f64 and ds_add_f32 instead of CAS loops; the outputs are
// ordinary hipMalloc'ed device memory, never fine-grained host memory).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/evk.h"

#define EVK_WAVE 64
#define EVK_BLOCK 256
#define EVK_NUM_CU 256

namespace evk {

__device__ __forceinline__ void atomic_add(float *p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add(int32_t *p, int32_t v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void count_oob(uint32_t *oob) {
    if (oob) __hip_atomic_fetch_add(oob, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 4 consecutive elements of a column, as one (float) or two (double) 16-byte loads per lane.
template <typename T>
struct Vec4 {
    T v[4];
};
template <typename T>
__device__ __forceinline__ Vec4<T> load4(const T *p, int64_t i4) {
    Vec4<T> r;
    if constexpr (sizeof(T) == 4) {
        const uint4 q = reinterpret_cast<const uint4 *>(p)[i4];
        uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) r.v[k] = __builtin_bit_cast(T, u[k]);
    } else {
        const double2 a = reinterpret_cast<const double2 *>(p)[2 * i4];
        const double2 b = reinterpret_cast<const double2 *>(p)[2 * i4 + 1];
        double d[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
        for (int k = 0; k < 4; ++k) r.v[k] = __builtin_bit_cast(T, d[k]);
    }
    return r;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Streaming launch shape: enough 256-thread blocks to fill 256 CUs x 8 blocks, grid-stride beyond that.
inline int stream_grid(int64_t work_items, int per_thread = 1) {
    int64_t blocks = (work_items + (int64_t)EVK_BLOCK * per_thread - 1) / ((int64_t)EVK_BLOCK * per_thread);
    if (blocks < 1) blocks = 1;
    const int64_t cap = (int64_t)EVK_NUM_CU * 8;
    return (int)(blocks < cap ? blocks : cap);
}

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? EVK_OK : (int)e;
}

}  // namespace evk
