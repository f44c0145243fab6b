// Small element-wise helpers of the host layer (round 6): the last pieces of arithmetic the Python side still did with torch on
// the product path -- the polarity weights of events_to_neg_pos_voxel_torch's two-voxelisation route (voxel_grid.py:173-174),
// max |p| of an event set (the bound of the IWE kernels' fixed-point accumulators), |p| for use_polarity=False
// (objectives.py:184-185).  Streaming kernels, 16-byte loads where the pointers allow.
#include "evk_common.h"

namespace evk {

__global__ void __launch_bounds__(EVK_BLOCK) k_polarity_weights(const float *__restrict__ p, int64_t n, float *__restrict__ pos,
                                                               float *__restrict__ neg) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = p[i];
        if (pos) pos[i] = v > 0.0f ? 1.0f : 0.0f;     // torch.where(ps > 0, 1.0, 0.0): a NaN polarity is in neither grid
        if (neg) neg[i] = v <= 0.0f ? 1.0f : 0.0f;    // torch.where(ps <= 0, 1.0, 0.0)
    }
}

// max |p| as a BIT PATTERN: the patterns of non-negative floats order like the values, and a NaN's pattern lies above every
// finite one -- so an unsigned atomic max propagates NaN as torch's max() does
template <typename T>
__global__ void __launch_bounds__(EVK_BLOCK) k_abs_max(const T *__restrict__ p, int64_t n, unsigned long long *__restrict__ out) {
    unsigned long long m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long b;
        if constexpr (sizeof(T) == 4) b = (unsigned long long)(__float_as_uint((float)p[i]) & 0x7FFFFFFFu);
        else b = (unsigned long long)__double_as_longlong((double)p[i]) & 0x7FFFFFFFFFFFFFFFull;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) __hip_atomic_fetch_max(out, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T>
__global__ void __launch_bounds__(EVK_BLOCK) k_abs(const T *__restrict__ in, int64_t n, T *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = in[i] < (T)0 ? -in[i] : (in[i] == (T)0 ? (T)0 : in[i]);   // |x| with -0.0 -> +0.0 (and NaN kept)
}

// float64 column -> float32 column of (in[i] - offset), the subtraction in float64; *inexact |= 1 when some value does not
// survive the narrowing exactly ((double)(float)v != v: also a NaN, as the host-side policy it replaces treats it)
__global__ void __launch_bounds__(EVK_BLOCK) k_narrow_f64(const double *__restrict__ in, int64_t n, double offset, float *__restrict__ out,
                                                         uint32_t *__restrict__ inexact) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = in[i] - offset;
        const float f = (float)v;
        out[i] = f;
        bad = bad || !((double)f == v);
    }
    if (inexact && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(inexact, 1u);
}

// the four planes of the average-timestamp images before / after their events (image.py:266-283): time planes 0, count
// planes ONE (upstream quirk: img_*_cnt = torch.ones); then cnt[cnt == 0] = 1 and time / cnt, float32
__global__ void __launch_bounds__(EVK_BLOCK) k_ts_planes_init(float *__restrict__ out4, int64_t plane) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 4 * plane; i += (int64_t)gridDim.x * blockDim.x)
        out4[i] = ((i / plane) & 1) ? 1.0f : 0.0f;
}
__global__ void __launch_bounds__(EVK_BLOCK) k_ts_finalise(const float *__restrict__ in4, int64_t plane, float *__restrict__ pos,
                                                          float *__restrict__ neg) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
        const float cp = in4[plane + i], cn = in4[3 * plane + i];
        pos[i] = in4[i] / (cp == 0.0f ? 1.0f : cp);
        neg[i] = in4[2 * plane + i] / (cn == 0.0f ? 1.0f : cn);
    }
}

// np.searchsorted(a, v) (side 'left') for a sorted float32 / float64 device column and float64 keys, one thread per key: the
// comparison in float64, as numpy promotes a float32 array against python floats (voxel_grid.py:104-105); a NaN key sorts
// behind every number, as in numpy
template <typename T>
__global__ void __launch_bounds__(EVK_BLOCK) k_searchsorted_left(const T *__restrict__ a, int64_t n, const double *__restrict__ v, int64_t m,
                                                                int64_t *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const double key = v[j];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const double av = (double)a[mid];
        if (av < key || (key != key && av == av)) lo = mid + 1;
        else hi = mid;
    }
    out[j] = lo;
}

}  // namespace evk

using namespace evk;

extern "C" int evk_searchsorted_left(const void *a, int elem_bytes, int64_t n, const double *keys, int64_t m, int64_t *out,
                                     void *stream) {
    if (n < 0 || m < 0 || (elem_bytes != 4 && elem_bytes != 8) || (n > 0 && !a) || (m > 0 && (!keys || !out))) return EVK_EINVAL;
    if (m == 0) return EVK_OK;
    const int blocks = (int)((m + EVK_BLOCK - 1) / EVK_BLOCK);
    if (elem_bytes == 4) k_searchsorted_left<float><<<blocks, EVK_BLOCK, 0, (hipStream_t)stream>>>((const float *)a, n, keys, m, out);
    else k_searchsorted_left<double><<<blocks, EVK_BLOCK, 0, (hipStream_t)stream>>>((const double *)a, n, keys, m, out);
    return launch_status();
}

extern "C" int evk_timestamp_planes_init_f32(float *out4, int64_t plane_elems, void *stream) {
    if (plane_elems <= 0 || !out4) return EVK_EINVAL;
    k_ts_planes_init<<<stream_grid(4 * plane_elems), EVK_BLOCK, 0, (hipStream_t)stream>>>(out4, plane_elems);
    return launch_status();
}

extern "C" int evk_timestamp_finalise_f32(const float *planes4, int64_t plane_elems, float *pos, float *neg, void *stream) {
    if (plane_elems <= 0 || !planes4 || !pos || !neg) return EVK_EINVAL;
    k_ts_finalise<<<stream_grid(plane_elems), EVK_BLOCK, 0, (hipStream_t)stream>>>(planes4, plane_elems, pos, neg);
    return launch_status();
}

extern "C" int evk_narrow_f64_f32(const double *in, int64_t n, double offset, float *out, uint32_t *inexact, void *stream) {
    if (n < 0 || (n > 0 && (!in || !out))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_narrow_f64<<<stream_grid(n), EVK_BLOCK, 0, (hipStream_t)stream>>>(in, n, offset, out, inexact);
    return launch_status();
}

extern "C" int evk_polarity_weights_f32(const float *p, int64_t n, float *pos, float *neg, void *stream) {
    if (n < 0 || (n > 0 && (!p || (!pos && !neg)))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_polarity_weights<<<stream_grid(n), EVK_BLOCK, 0, (hipStream_t)stream>>>(p, n, pos, neg);
    return launch_status();
}

// out8: 8 bytes of device memory; receives the bit pattern of max |p| (float32 pattern in the low word for elem_bytes 4,
// float64 pattern for 8), 0 for n == 0
extern "C" int evk_abs_max(const void *p, int elem_bytes, int64_t n, void *out8, void *stream) {
    if (n < 0 || !out8 || (n > 0 && !p) || (elem_bytes != 4 && elem_bytes != 8)) return EVK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out8, 0, 8, s);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return EVK_OK;
    if (elem_bytes == 4) k_abs_max<float><<<stream_grid(n, 4), EVK_BLOCK, 0, s>>>((const float *)p, n, (unsigned long long *)out8);
    else k_abs_max<double><<<stream_grid(n, 4), EVK_BLOCK, 0, s>>>((const double *)p, n, (unsigned long long *)out8);
    return launch_status();
}

extern "C" int evk_abs(const void *in, int elem_bytes, int64_t n, void *out, void *stream) {
    if (n < 0 || (n > 0 && (!in || !out)) || (elem_bytes != 4 && elem_bytes != 8)) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (elem_bytes == 4) k_abs<float><<<stream_grid(n), EVK_BLOCK, 0, s>>>((const float *)in, n, (float *)out);
    else k_abs<double><<<stream_grid(n), EVK_BLOCK, 0, s>>>((const double *)in, n, (double *)out);
    return launch_status();
}
