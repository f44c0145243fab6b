// Image-side helpers shared by evk_imgops.hip (standalone blur / reductions / fused post-pass) and evk_tiled.hip (the
// post-pass fused with the gather of the staged IWE windows): blur weights, scipy's 'reflect' index, deterministic
// two-stage reductions.
#pragma once
#include "evk_common.h"

namespace evk {

#define EVK_MAX_RADIUS 32

struct BlurWeights {
    double w[2 * EVK_MAX_RADIUS + 1];
    int radius;
};

#define EVK_REDUCE_MAX_BLOCKS 4096
#define EVK_REDUCE_K 7

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

// WIDE: additionally out[4..7] = the raw sums tot[1..4] (generic objectives); stride of `out` per plane stays 4 or 8.
// host_slot / host_flag (optional, pinned host memory): the block also stores its 4 results at host_slot[4 * plane ..] and
// then `seq` at host_flag[plane] (system-scope stores, the store counter drained in between), so that a host thread polling the flag has the results without
// a copy command or a stream synchronisation (evk_cmax.hip).
struct HostPublish {
    double *slot;
    uint32_t *flag;
    uint32_t seq;
};
// The results of one image plane from its total sums (thread 0 of the finalising workgroup), and their delivery to the host.
// The four doubles and then the sequence number go out as system-scope stores with the store counter drained in between: the
// host cannot see the flag before the values, and no release fence -- a write-back of the XCD's L2, on the path the host is
// waiting on -- is needed to order them.
template <int MODE, bool WIDE>
__device__ __forceinline__ void finalise_plane(const double *tot, int64_t n, double *out, const HostPublish &pub, int plane) {
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
        if constexpr (MODE == 3) {  // value + gradient: [g0, g1, mean v, var v] of the blurred image v
            const double mv = tot[5] * inv;
            out[2] = mv;
            out[3] = tot[6] * inv - mv * mv;
        }
    }
    if constexpr (WIDE && MODE == 0) {  // stats: [mean, var, sum v, sum v^2, sum exp v, sum exp(-p v), count, -]
        out[4] = tot[2], out[5] = tot[3], out[6] = tot[4], out[7] = 0.0;
    }
    if constexpr (WIDE && MODE == 1) {  // gradient sums: [.., .., mean, sum g, sum d0, sum d1, sum g d0, sum g d1]
        out[4] = tot[1], out[5] = tot[2], out[6] = tot[3], out[7] = tot[4];
    }
    if (pub.slot) {
        for (int k = 0; k < 4; ++k)
            __hip_atomic_store(pub.slot + 4 * plane + k, out[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(pub.flag + plane, pub.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int MODE, bool WIDE = false>
__global__ void __launch_bounds__(EVK_BLOCK) k_reduce_final(const double *__restrict__ partials, int nblocks,
                                                            int64_t n, double *__restrict__ out,
                                                            HostPublish pub = HostPublish{nullptr, nullptr, 0u}) {
    double acc[EVK_REDUCE_K] = {};
    partials += (int64_t)blockIdx.x * nblocks * EVK_REDUCE_K;  // one block per image plane (batched evaluation)
    for (int b = threadIdx.x; b < nblocks; b += EVK_BLOCK)
#pragma unroll
        for (int k = 0; k < EVK_REDUCE_K; ++k) acc[k] += partials[(int64_t)b * EVK_REDUCE_K + k];
    __shared__ double tot[EVK_REDUCE_K];
    block_sum<EVK_REDUCE_K>(acc, tot);
    __syncthreads();
    if (threadIdx.x == 0) finalise_plane<MODE, WIDE>(tot, n, out + (WIDE ? 8 : 4) * blockIdx.x, pub, (int)blockIdx.x);
}

#define EVK_POST_T 32
#define EVK_POST_MIX 1u       // scipy's 3-D gaussian_filter on (2, H, W): also filter across the channel axis (Q4)
#define EVK_POST_BLUR_IWE 2u  // gradient: use the blurred IWE (reference_exact=False); default is the raw IWE (Q5)

__device__ __forceinline__ int reflect_idx(int q, int len) {
    if ((unsigned)q < (unsigned)len) return q;                 // inside: the common case, no division
    if (q >= -len && q < 2 * len) return q < 0 ? -q - 1 : 2 * len - 1 - q;   // one reflection
    const int period = 2 * len;                                // a radius larger than the image
    int m = q % period;
    if (m < 0) m += period;
    return m >= len ? period - 1 - m : m;
}

// Blurs one plane over a 32x32 output tile (scipy gaussian_filter arithmetic: per-axis float64 accumulation in scipy's
// symmetric summation order, stored as float32 between the axes).  FILL(py, px) returns the source value at patch position
// (py, px), i.e. image pixel (y0 - r + py, x0 - r + px) under the 'reflect' boundary rule.
// RC = the radius when it is known at compile time (4: sigma = 1, the default blur of every objective), 0 = any radius
// (bw.radius).  With a constant radius the patch width is a constant (the flat index splits by a multiply-shift, not an
// integer division) and the taps are unrolled with immediate LDS offsets -- index arithmetic was most of this kernel.
template <int RC, typename FILL>
__device__ __forceinline__ void blur_tile_fill(float *patch, float *inter, const BlurWeights &bw, FILL fill, float (&res)[4]) {
    const int r = RC ? RC : bw.radius, PW = EVK_POST_T + 2 * r, PH = PW;
    for (int i = threadIdx.x; i < PH * PW; i += EVK_BLOCK) {
        const int py = i / PW, px = i - py * PW;
        patch[i] = fill(py, px);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < EVK_POST_T * PW; i += EVK_BLOCK) {  // axis 0 (rows)
        const int y = i / PW, xx = i - y * PW;
        const float *col = patch + (y + r) * PW + xx;
        double acc = (double)col[0] * bw.w[r];
#pragma unroll
        for (int j = r; j >= 1; --j) acc += ((double)col[-j * PW] + (double)col[j * PW]) * bw.w[r - j];
        inter[i] = (float)acc;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // axis 1 (columns)
        const int o = threadIdx.x + k * EVK_BLOCK;
        const int y = o / EVK_POST_T, x = o - y * EVK_POST_T;
        const float *row = inter + y * PW + x + r;
        double acc = (double)row[0] * bw.w[r];
#pragma unroll
        for (int j = r; j >= 1; --j) acc += ((double)row[-j] + (double)row[j]) * bw.w[r - j];
        res[k] = (float)acc;
    }
    __syncthreads();
}
// NP planes at once (round 6): FILL(py, px, v) delivers the source values of ALL planes at a patch position, so their global
// loads are in flight together, and every phase (fill | rows | columns) runs over all planes between ONE pair of barriers --
// three planes cost three barriers and one exposed load latency instead of nine and three.  Same arithmetic per pixel as
// blur_tile_fill (bit-identical values).  patch: NP * PW * PW floats, inter: NP * EVK_POST_T * PW floats.
template <int RC, int NPMAX, typename FILL>
__device__ __forceinline__ void blur_tiles_fill(float *patch, float *inter, const BlurWeights &bw, int np, FILL fill,
                                                float (&res)[NPMAX][4]) {
    const int r = RC ? RC : bw.radius, PW = EVK_POST_T + 2 * r, PH = PW, psz = PH * PW, isz = EVK_POST_T * PW;
    for (int i = threadIdx.x; i < psz; i += EVK_BLOCK) {
        const int py = i / PW, px = i - py * PW;
        float v[NPMAX];
        fill(py, px, v);
#pragma unroll
        for (int p = 0; p < NPMAX; ++p)
            if (p < np) patch[p * psz + i] = v[p];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < isz; i += EVK_BLOCK) {  // axis 0 (rows)
        const int y = i / PW, xx = i - y * PW;
#pragma unroll
        for (int p = 0; p < NPMAX; ++p) {
            if (p >= np) continue;
            const float *col = patch + p * psz + (y + r) * PW + xx;
            double acc = (double)col[0] * bw.w[r];
#pragma unroll
            for (int j = r; j >= 1; --j) acc += ((double)col[-j * PW] + (double)col[j * PW]) * bw.w[r - j];
            inter[p * isz + i] = (float)acc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // axis 1 (columns)
        const int o = threadIdx.x + k * EVK_BLOCK;
        const int y = o / EVK_POST_T, x = o - y * EVK_POST_T;
#pragma unroll
        for (int p = 0; p < NPMAX; ++p) {
            if (p >= np) continue;
            const float *row = inter + p * isz + y * PW + x + r;
            double acc = (double)row[0] * bw.w[r];
#pragma unroll
            for (int j = r; j >= 1; --j) acc += ((double)row[-j] + (double)row[j]) * bw.w[r - j];
            res[p][k] = (float)acc;
        }
    }
    __syncthreads();
}
// the same with the source read from global memory: LOAD(gy, gx) returns image pixel (gy, gx)
template <int RC, typename LOAD>
__device__ __forceinline__ void blur_tile(float *patch, float *inter, const BlurWeights &bw, int y0, int x0, int ch,
                                          int cw, LOAD load, float (&res)[4]) {
    const int r = RC ? RC : bw.radius;
    blur_tile_fill<RC>(patch, inter, bw,
                       [&](int py, int px) { return load(reflect_idx(y0 - r + py, ch), reflect_idx(x0 - r + px, cw)); }, res);
}

}  // namespace evk
