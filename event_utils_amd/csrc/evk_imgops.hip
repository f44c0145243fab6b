// Image-sized kernels of the contrast-maximisation objective: separable Gaussian blur with scipy's 'reflect'
// boundary (objectives.py:233,253) and the variance / gradient reductions (objectives.py:234,256-264).
// The images are <= a few MB (L2 / MALL resident); these kernels are latency-, not bandwidth-, bound.
#include "evk_common.h"

namespace evk {

#define EVK_MAX_RADIUS 32

struct BlurWeights {
    double w[2 * EVK_MAX_RADIUS + 1];
    int radius;
};

// One axis of scipy.ndimage.correlate1d(mode='reflect') on an array viewed as (outer, len, inner), filtered along
// `len`.  Evaluated in float64 with the symmetric-kernel summation order of scipy's NI_Correlate1D
// (centre tap first, then (x[-j] + x[+j]) * w[j] from the outermost pair inwards), stored as float32.
__global__ void __launch_bounds__(EVK_BLOCK) k_blur_axis(const float *__restrict__ src, float *__restrict__ dst,
                                                         int64_t total, int len, int64_t inner, BlurWeights bw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int r = bw.radius;
    const int period = 2 * len;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t i = idx % inner;
        const int64_t ol = idx / inner;
        const int l = (int)(ol % len);
        const int64_t o = ol / len;
        const float *line = src + o * len * inner + i;
        auto at = [&](int q) -> double {  // reflect ('symmetric'): d c b a | a b c d | d c b a
            int m = q % period;
            if (m < 0) m += period;
            if (m >= len) m = period - 1 - m;
            return (double)line[(int64_t)m * inner];
        };
        double acc = at(l) * bw.w[r];
        for (int j = r; j >= 1; --j) acc += (at(l - j) + at(l + j)) * bw.w[r - j];
        dst[idx] = (float)acc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// deterministic two-stage reductions (per-block partials in fixed order, then one block)
// ---------------------------------------------------------------------------------------------------------

#define EVK_REDUCE_MAX_BLOCKS 512
#define EVK_REDUCE_K 5

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int K>
__device__ __forceinline__ void block_sum(double (&acc)[K], double *partial_out) {
    __shared__ double lds[EVK_BLOCK / EVK_WAVE][K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) lds[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double s = 0.0;
        for (int w = 0; w < EVK_BLOCK / EVK_WAVE; ++w) s += lds[w][threadIdx.x];
        partial_out[threadIdx.x] = s;
    }
}

// MODE 0: sum(a), sum(a^2).   MODE 1: sum(a), sum(d0), sum(d1), sum(a*d0), sum(a*d1)
template <int MODE>
__global__ void __launch_bounds__(EVK_BLOCK) k_reduce_partial(const float *__restrict__ a,
                                                              const float *__restrict__ d, int64_t n,
                                                              double *__restrict__ partials) {
    double acc[EVK_REDUCE_K] = {0, 0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = (double)a[i];
        acc[0] += v;
        if constexpr (MODE == 0) {
            acc[1] += v * v;
        } else {
            const double d0 = (double)d[i], d1 = (double)d[n + i];
            acc[1] += d0;
            acc[2] += d1;
            acc[3] += v * d0;
            acc[4] += v * d1;
        }
    }
    block_sum<EVK_REDUCE_K>(acc, partials + (int64_t)blockIdx.x * EVK_REDUCE_K);
}

template <int MODE>
__global__ void __launch_bounds__(EVK_BLOCK) k_reduce_final(const double *__restrict__ partials, int nblocks,
                                                            int64_t n, double *__restrict__ out) {
    double acc[EVK_REDUCE_K] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += EVK_BLOCK)
#pragma unroll
        for (int k = 0; k < EVK_REDUCE_K; ++k) acc[k] += partials[(int64_t)b * EVK_REDUCE_K + k];
    __shared__ double tot[EVK_REDUCE_K];
    block_sum<EVK_REDUCE_K>(acc, tot);
    __syncthreads();
    if (threadIdx.x == 0) {
        const double inv = 1.0 / (double)n;
        const double mean = tot[0] * inv;
        if constexpr (MODE == 0) {
            out[0] = mean;
            out[1] = tot[1] * inv - mean * mean;
            out[2] = tot[0];
            out[3] = tot[1];
        } else {
            // mean(2*(a-mean)*d_i) = 2/n * (sum(a*d_i) - mean*sum(d_i))
            out[0] = 2.0 * inv * (tot[3] - mean * tot[1]);
            out[1] = 2.0 * inv * (tot[4] - mean * tot[2]);
            out[2] = mean;
            out[3] = tot[0];
        }
    }
}

}  // namespace evk

using namespace evk;

extern "C" int64_t evk_reduce_scratch_bytes(void) {
    return (int64_t)EVK_REDUCE_MAX_BLOCKS * EVK_REDUCE_K * sizeof(double);
}

extern "C" int evk_gaussian_filter_f32(const float *src, float *dst, float *tmp, int ndim, const int *host_dims,
                                       const double *host_weights, int radius, void *stream) {
    if (!src || !dst || !tmp || !host_dims || !host_weights || (ndim != 2 && ndim != 3)) return EVK_EINVAL;
    if (radius < 0 || radius > EVK_MAX_RADIUS || src == dst || src == tmp || dst == tmp) return EVK_EINVAL;
    int64_t total = 1;
    for (int a = 0; a < ndim; ++a) {
        if (host_dims[a] <= 0) return EVK_EINVAL;
        total *= host_dims[a];
    }
    BlurWeights bw;
    bw.radius = radius;
    for (int j = 0; j < 2 * radius + 1; ++j) bw.w[j] = host_weights[j];
    // ping-pong so that the last pass writes dst: passes alternate tmp/dst starting such that pass ndim-1 -> dst
    const float *in = src;
    hipStream_t s = (hipStream_t)stream;
    for (int a = 0; a < ndim; ++a) {
        float *out = ((ndim - 1 - a) % 2 == 0) ? dst : tmp;
        int64_t inner = 1;
        for (int b = a + 1; b < ndim; ++b) inner *= host_dims[b];
        k_blur_axis<<<stream_grid(total), EVK_BLOCK, 0, s>>>(in, out, total, host_dims[a], inner, bw);
        in = out;
    }
    return launch_status();
}

template <int MODE>
static int launch_reduce(const float *a, const float *d, int64_t n, double *out, void *scratch, int64_t scratch_bytes,
                         void *stream) {
    if (!a || n <= 0 || !out || !scratch || (MODE == 1 && !d)) return EVK_EINVAL;
    if (scratch_bytes < evk_reduce_scratch_bytes()) return EVK_ESCRATCH;
    int grid = stream_grid(n);
    if (grid > EVK_REDUCE_MAX_BLOCKS) grid = EVK_REDUCE_MAX_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    k_reduce_partial<MODE><<<grid, EVK_BLOCK, 0, s>>>(a, d, n, (double *)scratch);
    k_reduce_final<MODE><<<1, EVK_BLOCK, 0, s>>>((const double *)scratch, grid, n, out);
    return launch_status();
}

extern "C" int evk_variance_f32(const float *img, int64_t n, double *out, void *scratch, int64_t scratch_bytes,
                                void *stream) {
    return launch_reduce<0>(img, nullptr, n, out, scratch, scratch_bytes, stream);
}

extern "C" int evk_variance_grad_f32(const float *iwe, const float *diwe, int64_t n, double *out, void *scratch,
                                     int64_t scratch_bytes, void *stream) {
    return launch_reduce<1>(iwe, diwe, n, out, scratch, scratch_bytes, stream);
}
