// The two stateful image classes of lib/representations/image.py:355-396.
//   TimestampImage.add_events: image[int(y), int(x)] = t for every event IN STREAM ORDER -- the last event of a pixel wins.
//     Two kernels: the largest event position per pixel (one global atomicMax per event on a uint32 image), then one thread
//     per pixel fetches that event's time stamp.  int() truncates toward zero; a negative index wraps once (numpy
//     indexing); anything else is counted (the wrapper raises IndexError, as the reference's assignment does).
//   TimestampImage.get_image: scipy.stats.rankdata(image, method='dense') - 1, divided by its maximum: a radix sort of the
//     pixel values (hipCUB, as a library GEMM would be), a flag per change of value, an inclusive scan, a scatter.
//   EventImage.add_event(s): image[int(y), int(x)] += p (float64 global atomics; p == NULL only checks the indices, which is
//     what upstream's add_events does: it passes a literal 0 for the polarity); get_image: (image - min) / (max - min).
// Float64 throughout, as the reference's numpy images.  None of this is on the headline path: global atomics at ~21 G/s.
#include <hipcub/hipcub.hpp>

#include "evk_common.h"

namespace evk {

__device__ __forceinline__ bool class_pixel(double x, double y, int h, int w, int64_t &pix) {
    if (!(fabs(x) < 4.0e18) || !(fabs(y) < 4.0e18)) return false;   // NaN / infinity: int() raises
    long long xi = (long long)x, yi = (long long)y;                 // int(): toward zero
    if (xi < 0) xi += w;
    if (yi < 0) yi += h;
    if (xi < 0 || xi >= w || yi < 0 || yi >= h) return false;
    pix = (int64_t)yi * w + xi;
    return true;
}

__global__ void __launch_bounds__(EVK_BLOCK) k_last_writer_mark(const double *__restrict__ x, const double *__restrict__ y,
                                                                int64_t n, int h, int w, uint32_t *__restrict__ last,
                                                                uint32_t *oob) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t pix;
        if (class_pixel(x[i], y[i], h, w, pix)) atomicMax(last + pix, (uint32_t)(i + 1));
        else count_oob(oob);
    }
}
__global__ void __launch_bounds__(EVK_BLOCK) k_last_writer_apply(const double *__restrict__ t, const uint32_t *__restrict__ last,
                                                                 int64_t npix, double *__restrict__ image) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t k = last[i];
        if (k) image[i] = t[k - 1];
    }
}
__global__ void __launch_bounds__(EVK_BLOCK) k_class_add(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ p, int64_t n, int h, int w,
                                                         double *__restrict__ image, uint32_t *oob) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t pix;
        if (!class_pixel(x[i], y[i], h, w, pix)) count_oob(oob);
        else if (p) atomic_add(image + pix, p[i]);
    }
}

__global__ void __launch_bounds__(EVK_BLOCK) k_iota(uint32_t *v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(EVK_BLOCK) k_rank_flags(const double *__restrict__ sorted, int64_t n, uint32_t *__restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        flag[i] = (i > 0 && sorted[i] != sorted[i - 1]) ? 1u : 0u;      // (numeric: -0.0 and +0.0 share a rank)
}
__global__ void __launch_bounds__(EVK_BLOCK) k_rank_scatter(const uint32_t *__restrict__ rank, const uint32_t *__restrict__ where,
                                                            const double *__restrict__ sorted, int64_t n, double *__restrict__ out) {
    double top = (double)rank[n - 1];      // the largest dense rank (0 for a constant image: 0 / 0 = NaN, as upstream)
    // A NaN pixel: scipy.stats.rankdata (image.py:371) propagates it to EVERY rank.  The radix sort orders bit patterns, so a
    // NaN sits at one end of the sorted keys (sign bit clear: behind +inf; set: in front of -inf)
    if (sorted[0] != sorted[0] || sorted[n - 1] != sorted[n - 1]) top = __builtin_nan("");
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[where[i]] = (double)rank[i] / top;
}

// min / max with numpy's NaN propagation, two stages; then (image - min) / (max - min)
__global__ void __launch_bounds__(EVK_BLOCK) k_minmax_partial(const double *__restrict__ a, int64_t n, double *__restrict__ part) {
    double lo = __builtin_inf(), hi = -__builtin_inf();
    bool nan = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = a[i];
        nan |= v != v;
        lo = fmin(lo, v), hi = fmax(hi, v);
    }
    __shared__ double slo[EVK_BLOCK], shi[EVK_BLOCK];
    __shared__ int snan;
    if (threadIdx.x == 0) snan = 0;
    __syncthreads();
    slo[threadIdx.x] = lo, shi[threadIdx.x] = hi;
    if (nan) snan = 1;
    __syncthreads();
    for (int s = EVK_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + s]), shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[3 * blockIdx.x] = slo[0], part[3 * blockIdx.x + 1] = shi[0], part[3 * blockIdx.x + 2] = snan ? 1.0 : 0.0;
}
__global__ void __launch_bounds__(EVK_BLOCK) k_minmax_normalise(const double *__restrict__ a, int64_t n, const double *__restrict__ part,
                                                                int nparts, double *__restrict__ out) {
    double lo = __builtin_inf(), hi = -__builtin_inf();
    bool nan = false;
    for (int k = 0; k < nparts; ++k) lo = fmin(lo, part[3 * k]), hi = fmax(hi, part[3 * k + 1]), nan |= part[3 * k + 2] != 0.0;
    if (nan) lo = hi = __builtin_nan("");
    const double span = hi - lo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (a[i] - lo) / span;
}

static inline int64_t al(int64_t b) { return (b + 255) & ~(int64_t)255; }
static size_t rank_temp_bytes(int64_t n) {
    size_t a = 0, b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const double *)nullptr, (double *)nullptr, (const uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (int)n);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
    return a > b ? a : b;
}

}  // namespace evk

using namespace evk;

#define EVK_CLASS_MINMAX_BLOCKS 256

extern "C" int evk_timestamp_image_add_f64(const double *x, const double *y, const double *t, int64_t n, int h, int w,
                                           double *image, uint32_t *last_scratch, uint32_t *oob, void *stream) {
    if (n < 0 || n >= (int64_t)4294967295LL || h <= 0 || w <= 0 || !image || !last_scratch || (n > 0 && (!x || !y || !t)))
        return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)h * w;
    hipError_t e = hipMemsetAsync(last_scratch, 0, npix * sizeof(uint32_t), s);
    if (e != hipSuccess) return (int)e;
    k_last_writer_mark<<<stream_grid(n), EVK_BLOCK, 0, s>>>(x, y, n, h, w, last_scratch, oob);
    k_last_writer_apply<<<stream_grid(npix), EVK_BLOCK, 0, s>>>(t, last_scratch, npix, image);
    return launch_status();
}

extern "C" int evk_event_image_add_f64(const double *x, const double *y, const double *p, int64_t n, int h, int w, double *image,
                                       uint32_t *oob, void *stream) {
    if (n < 0 || h <= 0 || w <= 0 || !image || (n > 0 && (!x || !y))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_class_add<<<stream_grid(n), EVK_BLOCK, 0, (hipStream_t)stream>>>(x, y, p, n, h, w, image, oob);
    return launch_status();
}

extern "C" int64_t evk_dense_rank_scratch_bytes(int64_t npix) {
    if (npix <= 0 || npix > (int64_t)1 << 30) return 0;
    return al(npix * 8) + 3 * al(npix * 4) + al((int64_t)rank_temp_bytes(npix)) + 256;
}

extern "C" int evk_dense_rank_f64(const double *image, int64_t npix, double *out, void *scratch, int64_t scratch_bytes,
                                  void *stream) {
    if (!image || !out || !scratch || npix <= 0 || npix > (int64_t)1 << 30) return EVK_EINVAL;
    if (scratch_bytes < evk_dense_rank_scratch_bytes(npix)) return EVK_ESCRATCH;
    if ((uintptr_t)scratch & 255u) return EVK_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    char *sb = (char *)scratch;
    double *keys = (double *)sb;
    uint32_t *iota = (uint32_t *)(sb + al(npix * 8)), *where = iota + al(npix * 4) / 4, *flag = where + al(npix * 4) / 4;
    void *temp = (char *)(flag) + al(npix * 4);
    size_t tb = rank_temp_bytes(npix);
    k_iota<<<stream_grid(npix), EVK_BLOCK, 0, s>>>(iota, npix);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, tb, image, keys, (const uint32_t *)iota, where, (int)npix, 0, 64, s);
    if (e != hipSuccess) return (int)e;
    k_rank_flags<<<stream_grid(npix), EVK_BLOCK, 0, s>>>(keys, npix, flag);
    tb = rank_temp_bytes(npix);
    e = hipcub::DeviceScan::InclusiveSum(temp, tb, (const uint32_t *)flag, iota, (int)npix, s);   // iota now holds the ranks
    if (e != hipSuccess) return (int)e;
    k_rank_scatter<<<stream_grid(npix), EVK_BLOCK, 0, s>>>(iota, where, keys, npix, out);
    return launch_status();
}

extern "C" int evk_minmax_normalise_f64(const double *image, int64_t npix, double *out, double *scratch3n, void *stream) {
    if (!image || !out || !scratch3n || npix <= 0) return EVK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int blocks = stream_grid(npix);
    if (blocks > EVK_CLASS_MINMAX_BLOCKS) blocks = EVK_CLASS_MINMAX_BLOCKS;
    k_minmax_partial<<<blocks, EVK_BLOCK, 0, s>>>(image, npix, scratch3n);
    k_minmax_normalise<<<stream_grid(npix), EVK_BLOCK, 0, s>>>(image, npix, scratch3n, blocks, out);
    return launch_status();
}
extern "C" int64_t evk_minmax_scratch_bytes(void) { return (int64_t)EVK_CLASS_MINMAX_BLOCKS * 3 * sizeof(double); }
