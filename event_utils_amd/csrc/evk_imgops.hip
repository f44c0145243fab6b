// Image-sized kernels of the contrast-maximisation objective: separable Gaussian blur with scipy's 'reflect'
// boundary (objectives.py:233,253) and the variance / gradient reductions (objectives.py:234,256-264).
// The images are <= a few MB (L2 / MALL resident); these kernels are latency-, not bandwidth-, bound.
#include "evk_img.h"

namespace evk {

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

// MODE 0: sum(a), sum(a^2).   MODE 1: sum(a), sum(d0), sum(d1), sum(a*d0), sum(a*d1)
template <int MODE>
__global__ void __launch_bounds__(EVK_BLOCK) k_reduce_partial(const float *__restrict__ a,
                                                              const float *__restrict__ d, int64_t n,
                                                              double *__restrict__ partials) {
    double acc[EVK_REDUCE_K] = {};
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

// ---------------------------------------------------------------------------------------------------------
// fused objective post-pass: [channel mix ->] blur axis 0 -> blur axis 1 -> per-block partial sums, one launch.
// Arithmetic and summation order are those of k_blur_axis (bit-identical blurred values); the blurred images are
// never written to memory.  MODE 0: variance of blur(iwe).  MODE 1: gradient sums with blur3d(diwe).
// ---------------------------------------------------------------------------------------------------------
// weight function g(a) of the generic gradient sums  sum_pix g(a) * blur(d_iwe)[i]
#define EVK_G_IDENT 0   // a                 (variance, sos)
#define EVK_G_EXP 1     // exp(a)            (soe)
#define EVK_G_STEP 2    // a > gparam ? 1:0  (isoa)
#define EVK_G_EXPNEG 3  // exp((double)(float)(-gparam*a))   (sosa; the reference forms -p*iwe in float32 first)
struct PostParams {
    uint32_t flags;
    int gfun;
    double gparam;  // MODE 1: parameter of g;  MODE 2: p of sum(exp(-p v))
    double thresh;  // MODE 2: threshold of count(v > thresh)
    unsigned int *max_bits;  // MODE 2: running max of v as order-preserving uint bits (atomicMax)
    int batch_planes;        // MODE 1 / 3: the LDS holds the patches of all three planes (launches whose 3 x LDS fits)
    int y_lo, y_hi;          // rows whose pixels enter the sums (the whole image: 0, ch).  A row block of an image that is
                             // sharded by rows (evk_objective_variance_rows_f32) carries halo rows that are blurred FROM
                             // but not summed
};

__device__ __forceinline__ unsigned int float_order_bits(float f) {  // monotone map float -> uint
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// MODE 0: sum v, sum v^2 of the blurred image.  MODE 1: sum g(a), sum d0, sum d1, sum g(a) d0, sum g(a) d1.
// MODE 2: sum v, sum v^2, sum exp(v), sum exp(-p v), count(v > thresh) and max v of the blurred image.
// MODE 3: MODE 1 and MODE 0 together (function value AND gradient of one evaluation): the 5 sums of MODE 1, then
//         sum v, sum v^2 of the blurred IWE.
template <int MODE, int RC>
__global__ void __launch_bounds__(EVK_BLOCK) k_post_fused(const float *__restrict__ iwe,
                                                          const float *__restrict__ diwe, int ch, int cw,
                                                          BlurWeights bw, PostParams pp,
                                                          double *__restrict__ partials) {
    const uint32_t flags = pp.flags;
    extern __shared__ float sm[];
    const int r = RC ? RC : bw.radius, PW = EVK_POST_T + 2 * r;
    float *patch = sm, *inter = sm + ((MODE == 1 || MODE == 3) && pp.batch_planes ? 3 : 1) * PW * PW;
    const int tiles_x = (cw + EVK_POST_T - 1) / EVK_POST_T, ntile = tiles_x * ((ch + EVK_POST_T - 1) / EVK_POST_T);
    const int64_t plane = (int64_t)ch * cw;
    iwe += blockIdx.y * plane;                                         // MODE 0 batched over image planes
    partials += (int64_t)blockIdx.y * gridDim.x * EVK_REDUCE_K;
    double acc[EVK_REDUCE_K] = {};
    // one 32x32 output tile per workgroup, or several (grid-stride) when the image has more tiles than the reduction
    // scratch has slots (1080p x 3 planes, 4K): the partial sums of a workgroup's tiles are added in tile order
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int y0 = (tile / tiles_x) * EVK_POST_T, x0 = (tile % tiles_x) * EVK_POST_T;
    if constexpr (MODE == 0 || MODE == 2) {
        float v[4];
        blur_tile<RC>(patch, inter, bw, y0, x0, ch, cw, [&](int gy, int gx) { return iwe[(int64_t)gy * cw + gx]; }, v);
        float vmax = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = threadIdx.x + k * EVK_BLOCK, y = y0 + o / EVK_POST_T, x = x0 + o % EVK_POST_T;
            if (y >= pp.y_lo && y < pp.y_hi && x < cw) {
                acc[0] += (double)v[k];
                acc[1] += (double)v[k] * (double)v[k];
                if constexpr (MODE == 2) {
                    acc[2] += exp((double)v[k]);
                    acc[3] += exp(-pp.gparam * (double)v[k]);
                    acc[4] += ((double)v[k] > pp.thresh) ? 1.0 : 0.0;
                    vmax = fmaxf(vmax, v[k]);
                }
            }
        }
        if constexpr (MODE == 2) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
            if ((threadIdx.x & 63) == 0 && vmax > -__builtin_inff()) atomicMax(pp.max_bits, float_order_bits(vmax));
        }
    } else {
        // (One workgroup per PART of a tile -- channel 0, channel 1, blurred IWE; grid.y = 3 -- measured slower: 37.4 vs
        // 29.5 us for value + gradient at 720p, and the finalise has three times the partial sums to add.)
        float d[2][4], a[4];
        const bool need_a = MODE == 3 || (flags & EVK_POST_BLUR_IWE);
        if (pp.batch_planes) {
            // (round 6) all planes of the tile in ONE fill | rows | columns sweep (evk_img.h, blur_tiles_fill): the two dIWE
            // channels' loads serve both mixed channels, the IWE's loads fly with them
            uint64_t lo[2] = {0, 0}, hi[2] = {0, 0};
            for (int c = 0; c < 2; ++c)
                for (int j = 1; j <= r; ++j) {
                    lo[c] |= (uint64_t)reflect_idx(c - j, 2) << j;
                    hi[c] |= (uint64_t)reflect_idx(c + j, 2) << j;
                }
            float res3[3][4];
            blur_tiles_fill<RC, 3>(patch, inter, bw, need_a ? 3 : 2, [&](int py, int px, float (&v)[3]) {
                const int64_t pix = (int64_t)reflect_idx(y0 - r + py, ch) * cw + reflect_idx(x0 - r + px, cw);
                const float dv[2] = {diwe[pix], diwe[plane + pix]};
                v[2] = need_a ? iwe[pix] : 0.0f;
                if (!(flags & EVK_POST_MIX)) {
                    v[0] = dv[0], v[1] = dv[1];
                } else {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        double s = (double)dv[c] * bw.w[r];
                        for (int j = r; j >= 1; --j)
                            s += ((double)((lo[c] >> j) & 1 ? dv[1] : dv[0]) + (double)((hi[c] >> j) & 1 ? dv[1] : dv[0])) * bw.w[r - j];
                        v[c] = (float)s;
                    }
                }
            }, res3);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[0][k] = res3[0][k], d[1][k] = res3[1][k], a[k] = res3[2][k];
        } else {
        for (int c = 0; c < 2; ++c) {
            // channel reflected at offsets -j / +j of the length-2 channel axis: the same for every pixel, so the two
            // reflections per tap are taken once per tile (bit j of lo / hi), not once per tap and pixel
            uint64_t lo = 0, hi = 0;
            for (int j = 1; j <= r; ++j) {
                lo |= (uint64_t)reflect_idx(c - j, 2) << j;
                hi |= (uint64_t)reflect_idx(c + j, 2) << j;
            }
            auto load_mixed = [&](int gy, int gx) -> float {
                const int64_t pix = (int64_t)gy * cw + gx;
                if (!(flags & EVK_POST_MIX)) return diwe[c * plane + pix];
                // axis-0 pass of scipy's 3-D filter over the length-2 channel axis (reflect), f64 -> f32
                const float dv[2] = {diwe[pix], diwe[plane + pix]};
                double s = (double)dv[c] * bw.w[r];
                for (int j = r; j >= 1; --j)
                    s += ((double)((lo >> j) & 1 ? dv[1] : dv[0]) + (double)((hi >> j) & 1 ? dv[1] : dv[0])) * bw.w[r - j];
                return (float)s;
            };
            blur_tile<RC>(patch, inter, bw, y0, x0, ch, cw, load_mixed, d[c]);
        }
        if (need_a) {
            blur_tile<RC>(patch, inter, bw, y0, x0, ch, cw, [&](int gy, int gx) { return iwe[(int64_t)gy * cw + gx]; }, a);
        }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = threadIdx.x + k * EVK_BLOCK, y = y0 + o / EVK_POST_T, x = x0 + o % EVK_POST_T;
            if (y >= pp.y_lo && y < pp.y_hi && x < cw) {
                const float af = (flags & EVK_POST_BLUR_IWE) ? a[k] : iwe[(int64_t)y * cw + x];
                double av = (double)af;
                if (pp.gfun == EVK_G_EXP) av = exp(av);
                else if (pp.gfun == EVK_G_STEP) av = (av > pp.gparam) ? 1.0 : 0.0;
                else if (pp.gfun == EVK_G_EXPNEG) av = exp((double)((float)(-pp.gparam) * af));
                acc[0] += av;
                acc[1] += (double)d[0][k];
                acc[2] += (double)d[1][k];
                acc[3] += av * (double)d[0][k];
                acc[4] += av * (double)d[1][k];
                if constexpr (MODE == 3) {
                    acc[5] += (double)a[k];
                    acc[6] += (double)a[k] * (double)a[k];
                }
            }
        }
    }
    }
    block_sum<EVK_REDUCE_K>(acc, partials + (int64_t)blockIdx.x * EVK_REDUCE_K);
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

// LDS of a post-pass workgroup; MODE 1 / 3 hold the patches of all three planes (one fill | rows | columns sweep) while that fits
// the 64 KB a launch gets without asking
static size_t post_lds(int mode, int radius, int *batch) {
    const int PW = EVK_POST_T + 2 * radius;
    const size_t one = (size_t)(PW * PW + EVK_POST_T * PW) * sizeof(float);
    *batch = ((mode == 1 || mode == 3) && 3 * one <= (size_t)60 * 1024) ? 1 : 0;
#ifdef EVK_POST_NO_BATCH
    *batch = 0;   // (A/B)
#endif
    return *batch ? 3 * one : one;
}

template <int MODE>
static int launch_post(const float *iwe, const float *diwe, int h, int w, const double *host_weights, int radius,
                       uint32_t flags, double *out, void *scratch, int64_t scratch_bytes, void *stream,
                       int nplanes = 1, const HostPublish *pub = nullptr, bool *published = nullptr) {
    if (!iwe || h <= 0 || w <= 0 || !out || !scratch || (MODE == 1 && !diwe) || nplanes < 1 || nplanes > 8)
        return EVK_EINVAL;
    if (published) *published = false;
    if (scratch_bytes < evk_reduce_scratch_bytes()) return EVK_ESCRATCH;
    static const double identity[1] = {1.0};
    if (MODE == 3 && radius < 0) host_weights = identity, radius = 0, flags &= ~EVK_POST_MIX;  // no blur = 1-tap kernel
    if (radius < 0) {  // blur_sigma <= 0: plain reductions
        for (int k = 0; k < nplanes; ++k) {
            const int rc = launch_reduce<MODE>(iwe + (int64_t)k * h * w, diwe, (int64_t)h * w, out + 4 * k, scratch,
                                               scratch_bytes, stream);
            if (rc != EVK_OK) return rc;
        }
        return EVK_OK;
    }
    if (!host_weights || radius > EVK_MAX_RADIUS) return EVK_EINVAL;
    int grid = ((h + EVK_POST_T - 1) / EVK_POST_T) * ((w + EVK_POST_T - 1) / EVK_POST_T);
    if (grid > EVK_REDUCE_MAX_BLOCKS / nplanes) grid = EVK_REDUCE_MAX_BLOCKS / nplanes;  // the kernel strides over the tiles
    BlurWeights bw;
    bw.radius = radius;
    for (int j = 0; j < 2 * radius + 1; ++j) bw.w[j] = host_weights[j];
    PostParams pp;
    const size_t lds = post_lds(MODE, radius, &pp.batch_planes);
    hipStream_t s = (hipStream_t)stream;
    pp.flags = flags, pp.gfun = EVK_G_IDENT, pp.gparam = 0.0, pp.thresh = 0.0, pp.max_bits = nullptr, pp.y_lo = 0, pp.y_hi = h;
    if (radius == 4) k_post_fused<MODE, 4><<<dim3(grid, nplanes), EVK_BLOCK, lds, s>>>(iwe, diwe, h, w, bw, pp, (double *)scratch);
    else k_post_fused<MODE, 0><<<dim3(grid, nplanes), EVK_BLOCK, lds, s>>>(iwe, diwe, h, w, bw, pp, (double *)scratch);
    const bool publish = pub && pub->slot && pub->flag && nplanes <= 3;
    k_reduce_final<MODE><<<nplanes, EVK_BLOCK, 0, s>>>((const double *)scratch, grid, (int64_t)h * w, out,
                                                      publish ? *pub : HostPublish{nullptr, nullptr, 0u});
    if (published) *published = publish;
    return launch_status();
}

// evk_cmax.hip: the post-pass of the one-call evaluation; mode 0 value (nplanes images), 1 gradient, 3 value + gradient.
// *published tells whether the finalise kernel also delivers the results to pub (it does on the blurred path).
int evk_post_variance_publish(int mode, const float *iwe, const float *diwe, int h, int w, const double *host_weights,
                              int radius, uint32_t flags, double *out, void *scratch, int64_t scratch_bytes, void *stream,
                              int nplanes, const HostPublish *pub, bool *published) {
    if (mode == 0)
        return launch_post<0>(iwe, nullptr, h, w, host_weights, radius, 0u, out, scratch, scratch_bytes, stream, nplanes, pub,
                              published);
    if (mode == 1)
        return launch_post<1>(iwe, diwe, h, w, host_weights, radius, flags, out, scratch, scratch_bytes, stream, 1, pub, published);
    if (mode == 3)
        return launch_post<3>(iwe, diwe, h, w, host_weights, radius, flags, out, scratch, scratch_bytes, stream, 1, pub, published);
    return EVK_EINVAL;
}

extern "C" int evk_objective_variance_f32(const float *iwe, int h, int w, const double *host_weights, int radius,
                                          double *out, void *scratch, int64_t scratch_bytes, void *stream) {
    return launch_post<0>(iwe, nullptr, h, w, host_weights, radius, 0u, out, scratch, scratch_bytes, stream);
}

extern "C" int evk_objective_variance_grad_f32(const float *iwe, const float *diwe, int h, int w,
                                               const double *host_weights, int radius, uint32_t flags, double *out,
                                               void *scratch, int64_t scratch_bytes, void *stream) {
    return launch_post<1>(iwe, diwe, h, w, host_weights, radius, flags, out, scratch, scratch_bytes, stream);
}

extern "C" int evk_objective_variance_fg_f32(const float *iwe, const float *diwe, int h, int w,
                                             const double *host_weights, int radius, uint32_t flags, double *out,
                                             void *scratch, int64_t scratch_bytes, void *stream) {
    return launch_post<3>(iwe, diwe, h, w, host_weights, radius, flags, out, scratch, scratch_bytes, stream);
}

extern "C" int evk_objective_variance_planes_f32(const float *imgs, int nplanes, int h, int w,
                                                 const double *host_weights, int radius, double *out, void *scratch,
                                                 int64_t scratch_bytes, void *stream) {
    return launch_post<0>(imgs, nullptr, h, w, host_weights, radius, 0u, out, scratch, scratch_bytes, stream, nplanes);
}

// ---- row-sharded post-pass (multi-GPU, optional): the image is cut into row blocks, one per rank; a rank holds the rows of
// its block plus `radius` halo rows on each interior side, already summed over the ranks (an all-to-all of row blocks
// moves half the bytes of an all-reduce), blurs them and adds up its OWN rows only.  The raw sums of all ranks are then
// all-reduced (8 doubles) and finalised by the caller: mean = S0/N, var = S1/N - mean^2, g_i = 2/N (S(3+i) - mean S(1+i)).
__global__ void __launch_bounds__(EVK_BLOCK) k_reduce_raw(const double *__restrict__ partials, int nblocks,
                                                          double *__restrict__ out) {
    double acc[EVK_REDUCE_K] = {};
    for (int b = threadIdx.x; b < nblocks; b += EVK_BLOCK)
#pragma unroll
        for (int k = 0; k < EVK_REDUCE_K; ++k) acc[k] += partials[(int64_t)b * EVK_REDUCE_K + k];
    __shared__ double tot[EVK_REDUCE_K];
    block_sum<EVK_REDUCE_K>(acc, tot);
    __syncthreads();
    if (threadIdx.x < 8) out[threadIdx.x] = threadIdx.x < EVK_REDUCE_K ? tot[threadIdx.x] : 0.0;
}

template <int MODE>
static int launch_post_rows(const float *img, int hb, int w, int y_lo, int y_hi, const double *host_weights, int radius,
                            uint32_t flags, double *sums, void *scratch, int64_t scratch_bytes, void *stream) {
    if (!img || hb <= 0 || w <= 0 || y_lo < 0 || y_hi < y_lo || y_hi > hb || !sums || !scratch) return EVK_EINVAL;
    if (scratch_bytes < evk_reduce_scratch_bytes()) return EVK_ESCRATCH;
    static const double identity[1] = {1.0};
    if (radius < 0) host_weights = identity, radius = 0, flags &= ~EVK_POST_MIX;  // no blur = 1-tap kernel
    if (!host_weights || radius > EVK_MAX_RADIUS) return EVK_EINVAL;
    int grid = ((hb + EVK_POST_T - 1) / EVK_POST_T) * ((w + EVK_POST_T - 1) / EVK_POST_T);
    if (grid > EVK_REDUCE_MAX_BLOCKS) grid = EVK_REDUCE_MAX_BLOCKS;
    BlurWeights bw;
    bw.radius = radius;
    for (int j = 0; j < 2 * radius + 1; ++j) bw.w[j] = host_weights[j];
    PostParams pp;
    const size_t lds = post_lds(MODE, radius, &pp.batch_planes);
    hipStream_t s = (hipStream_t)stream;
    pp.flags = flags, pp.gfun = EVK_G_IDENT, pp.gparam = 0.0, pp.thresh = 0.0, pp.max_bits = nullptr, pp.y_lo = y_lo, pp.y_hi = y_hi;
    const float *diwe = img + (int64_t)hb * w;
    if (radius == 4) k_post_fused<MODE, 4><<<dim3(grid, 1), EVK_BLOCK, lds, s>>>(img, diwe, hb, w, bw, pp, (double *)scratch);
    else k_post_fused<MODE, 0><<<dim3(grid, 1), EVK_BLOCK, lds, s>>>(img, diwe, hb, w, bw, pp, (double *)scratch);
    k_reduce_raw<<<1, EVK_BLOCK, 0, s>>>((const double *)scratch, grid, sums);
    return launch_status();
}

extern "C" int evk_objective_variance_rows_f32(const float *img, int mode, int hb, int w, int y_lo, int y_hi,
                                               const double *host_weights, int radius, uint32_t flags, double *sums,
                                               void *scratch, int64_t scratch_bytes, void *stream) {
    if (mode == 0) return launch_post_rows<0>(img, hb, w, y_lo, y_hi, host_weights, radius, 0u, sums, scratch, scratch_bytes, stream);
    if (mode == 1) return launch_post_rows<1>(img, hb, w, y_lo, y_hi, host_weights, radius, flags, sums, scratch, scratch_bytes, stream);
    if (mode == 3) return launch_post_rows<3>(img, hb, w, y_lo, y_hi, host_weights, radius, flags, sums, scratch, scratch_bytes, stream);
    return EVK_EINVAL;
}

// Generic objective reductions (the other objectives of objectives.py:266-596 differ from the variance objective only
// in these scalars): one fused blur + partial-sum launch and a finalise.
__global__ void k_stats_finish(const double *__restrict__ wide, const unsigned int *__restrict__ max_bits,
                               double *__restrict__ out) {
    // wide = k_reduce_final<0, true> output: [mean, var, sum v, sum v^2, sum exp v, sum exp(-p v), count, -]
    for (int k = 0; k < 7; ++k) out[k] = wide[k];
    const unsigned int b = *max_bits;
    const unsigned int u = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b;
    out[7] = (double)__uint_as_float(u);
}

static int blur_setup(int h, int w, const double *host_weights, int radius, BlurWeights &bw, int &grid, size_t &lds) {
    if (!host_weights || radius < 0 || radius > EVK_MAX_RADIUS) return EVK_EINVAL;
    grid = ((h + EVK_POST_T - 1) / EVK_POST_T) * ((w + EVK_POST_T - 1) / EVK_POST_T);
    if (grid > EVK_REDUCE_MAX_BLOCKS - 8) grid = EVK_REDUCE_MAX_BLOCKS - 8;  // the kernel strides over the tiles
    bw.radius = radius;
    for (int j = 0; j < 2 * radius + 1; ++j) bw.w[j] = host_weights[j];
    const int PW = EVK_POST_T + 2 * radius;
    lds = (size_t)(PW * PW + EVK_POST_T * PW) * sizeof(float);
    return EVK_OK;
}

static const double kIdentityWeight[1] = {1.0};

extern "C" int evk_objective_stats_f32(const float *img, int h, int w, const double *host_weights, int radius, double p,
                                       double thresh, double *out8, void *scratch, int64_t scratch_bytes,
                                       void *stream) {
    if (!img || h <= 0 || w <= 0 || !out8 || !scratch) return EVK_EINVAL;
    if (scratch_bytes < evk_reduce_scratch_bytes()) return EVK_ESCRATCH;
    if (radius < 0) host_weights = kIdentityWeight, radius = 0;  // no blur = a 1-tap kernel
    BlurWeights bw;
    int grid;
    size_t lds;
    int rc = blur_setup(h, w, host_weights, radius, bw, grid, lds);
    if (rc != EVK_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    // tail of the scratch: 8 doubles of wide sums + the running max
    double *wide = (double *)scratch + (int64_t)EVK_REDUCE_MAX_BLOCKS * EVK_REDUCE_K - 16;
    unsigned int *max_bits = (unsigned int *)(wide + 8);
    hipError_t e = hipMemsetAsync(max_bits, 0, sizeof(unsigned int), s);
    if (e != hipSuccess) return (int)e;
    PostParams pp;
    pp.flags = 0, pp.gfun = 0, pp.gparam = p, pp.thresh = thresh, pp.max_bits = max_bits, pp.y_lo = 0, pp.y_hi = h, pp.batch_planes = 0;
    if (radius == 4) k_post_fused<2, 4><<<dim3(grid, 1), EVK_BLOCK, lds, s>>>(img, nullptr, h, w, bw, pp, (double *)scratch);
    else k_post_fused<2, 0><<<dim3(grid, 1), EVK_BLOCK, lds, s>>>(img, nullptr, h, w, bw, pp, (double *)scratch);
    k_reduce_final<0, true><<<1, EVK_BLOCK, 0, s>>>((const double *)scratch, grid, (int64_t)h * w, wide);
    k_stats_finish<<<1, 1, 0, s>>>(wide, max_bits, out8);
    return launch_status();
}

extern "C" int evk_objective_gradsums_f32(const float *iwe, const float *diwe, int h, int w,
                                          const double *host_weights, int radius, uint32_t flags, int gfun,
                                          double gparam, double *out8, void *scratch, int64_t scratch_bytes,
                                          void *stream) {
    if (!iwe || !diwe || h <= 0 || w <= 0 || !out8 || !scratch || gfun < 0 || gfun > 3) return EVK_EINVAL;
    if (scratch_bytes < evk_reduce_scratch_bytes()) return EVK_ESCRATCH;
    if (radius < 0) host_weights = kIdentityWeight, radius = 0, flags &= ~EVK_POST_MIX;  // sigma <= 0: nothing is blurred
    BlurWeights bw;
    int grid;
    size_t lds;
    int rc = blur_setup(h, w, host_weights, radius, bw, grid, lds);
    if (rc != EVK_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    PostParams pp;
    pp.flags = flags, pp.gfun = gfun, pp.gparam = gparam, pp.thresh = 0.0, pp.max_bits = nullptr, pp.y_lo = 0, pp.y_hi = h;
    lds = post_lds(1, radius, &pp.batch_planes);
    if (radius == 4) k_post_fused<1, 4><<<dim3(grid, 1), EVK_BLOCK, lds, s>>>(iwe, diwe, h, w, bw, pp, (double *)scratch);
    else k_post_fused<1, 0><<<dim3(grid, 1), EVK_BLOCK, lds, s>>>(iwe, diwe, h, w, bw, pp, (double *)scratch);
    k_reduce_final<1, true><<<1, EVK_BLOCK, 0, s>>>((const double *)scratch, grid, (int64_t)h * w, out8);
    return launch_status();
}
