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

// The in-kernel hand-overs of this library (the partition's ticket, the pieces of a cut tile, the host publish of a result)
// carry NO release / acquire fence: everything the receiving side reads was written with agent- (or system-) scope atomics,
// which on gfx950 are performed at the level all XCDs share and complete -- vmcnt -- only then, and every wave drains its own
// with `s_waitcnt vmcnt(0)` before the workgroup's ticket (DESIGN.md section 3, K1': a fence there is a write-back of the
// XCD's whole L2, 4.5 us of a 47 us kernel).  That reasoning is specific to gfx9-family memory pipelines (one counter, vmcnt,
// covering loads, stores and atomics): refuse to compile the device code for anything else.  -DEVK_SAFE_HANDOVER puts the
// textbook fences back (release before the ticket, acquire after it) so that the parity tests can be run with both forms.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "libevk's fence-free hand-overs assume a gfx9-family memory pipeline (built and validated for gfx950 only)"
#endif
#ifdef EVK_SAFE_HANDOVER
#define EVK_HANDOVER_RELEASE() __atomic_thread_fence(__ATOMIC_RELEASE)   /* (HIP: agent scope) */
#define EVK_HANDOVER_ACQUIRE() __atomic_thread_fence(__ATOMIC_ACQUIRE)
#else
#define EVK_HANDOVER_RELEASE() do {} while (0)
#define EVK_HANDOVER_ACQUIRE() do {} while (0)
#endif
// every wave drains its own stores / atomics, [fence], workgroup barrier: what precedes a ticket
#define EVK_HANDOVER_DRAIN()                              \
    do {                                                  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  \
        EVK_HANDOVER_RELEASE();                           \
        __syncthreads();                                  \
    } while (0)

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
// columns of the one-pass entry points: 16-byte aligned, or -- with EVK_COLUMNS_UNALIGNED (evk.h) in `flags` -- aligned to their
// element (4 bytes; 8 for int64 pixels): the kernels' 16-byte loads are dword-aligned loads (evk_part.h, load_col16)
inline bool column_ok(const void *p, int flags, unsigned elem = 4) {
    return (reinterpret_cast<uintptr_t>(p) & ((flags & EVK_COLUMNS_UNALIGNED) ? elem - 1u : 15u)) == 0;
}

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
