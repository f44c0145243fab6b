// Shared pieces of the tile-bucketed kernels (evk_tiled.hip: three-pass counting sort + IWE windows;
// evk_voxel2.hip: one-pass sub-chunk partition + voxel tiles): tile geometry, the key of an event, the column
// sources (float32 SoA / the reference's on-disk dtypes), LDS accumulation helpers.
#pragma once
#include <cstdlib>

#include "evk_common.h"

namespace evk {

struct TileGrid {
    int tw_log2, th_log2;  // tile size (pixels), powers of two
    int tiles_x, tiles_y;  // tiles covering the key domain
    int dom_w, dom_h;      // key domain in pixels
};

#define EVK_KEY_NEAREST 0  // voxel / nearest image: trunc toward zero, negative wrap, out-of-domain -> dropped + counted
#define EVK_KEY_FLOOR_CLAMP 1  // IWE: floor, clamped into the domain (the tile only seeds the window; any event is legal)

__device__ __forceinline__ int tile_key(float x, float y, const TileGrid &g, int mode) {
    int xi, yi;
    if (mode == EVK_KEY_NEAREST) {
        // .long() truncation; a single saturating v_cvt_i32_f32 is enough: anything beyond int32 is out of the domain
        // either way (NaN would convert to 0, so it is rejected explicitly: torch gives INT64_MIN -> IndexError)
        if (x != x || y != y) return -1;
        xi = (int)x;
        yi = (int)y;
        if (xi < 0) xi += g.dom_w;
        if (yi < 0) yi += g.dom_h;
        if (xi < 0 || xi >= g.dom_w || yi < 0 || yi >= g.dom_h) return -1;
    } else {
        const float fx = floorf(x), fy = floorf(y);
        xi = fx > 0.0f ? (fx < (float)(g.dom_w - 1) ? (int)fx : g.dom_w - 1) : 0;  // NaN -> 0
        yi = fy > 0.0f ? (fy < (float)(g.dom_h - 1) ? (int)fy : g.dom_h - 1) : 0;
    }
    return (yi >> g.th_log2) * g.tiles_x + (xi >> g.tw_log2);
}

// ---------------------------------------------------------------------------------------------------------
// column sources of the bucketing kernels: quads of 4 consecutive events (base index lo % 4 == 0) and single events
// ---------------------------------------------------------------------------------------------------------
struct ColsF32 {  // four float32 SoA columns, 16 B / event
    const float *x, *y, *t, *p;
    __device__ __forceinline__ void xy4(int64_t lo, int64_t q, Vec4<float> &xv, Vec4<float> &yv) const {
        xv = load4(x + lo, q), yv = load4(y + lo, q);
    }
    __device__ __forceinline__ void tp4(int64_t lo, int64_t q, Vec4<float> &tv, Vec4<float> &pv) const {
        tv = load4(t + lo, q), pv = load4(p + lo, q);
    }
    // quad with only nv (1..4) valid events: the 16-byte load stays inside its aligned block, so reading the unused tail
    // cannot leave the page the valid elements live in
    // (qb + ql = quad index; qb is workgroup-uniform, ql the thread's 32-bit part: base in SGPRs, one VGPR of offset)
    __device__ __forceinline__ void xy4n(int64_t lo, int64_t qb, uint32_t ql, int, Vec4<float> &xv, Vec4<float> &yv) const {
        xv = load4(x + lo + 4 * qb, ql), yv = load4(y + lo + 4 * qb, ql);
    }
    __device__ __forceinline__ void tp4n(int64_t lo, int64_t qb, uint32_t ql, int, Vec4<float> &tv, Vec4<float> &pv) const {
        tv = load4(t + lo + 4 * qb, ql), pv = load4(p + lo + 4 * qb, ql);
    }
    __device__ __forceinline__ void p4(int64_t lo, int64_t q, Vec4<float> &pv) const { pv = load4(p + lo, q); }
    __device__ __forceinline__ float x1(int64_t i) const { return x[i]; }
    __device__ __forceinline__ float y1(int64_t i) const { return y[i]; }
    __device__ __forceinline__ float t1(int64_t i) const { return t[i]; }
    __device__ __forceinline__ float p1(int64_t i) const { return p[i]; }
};

// The on-disk dtypes of the reference's event files (event_packagers.py:90-93: xs, ys int16, ts float64, ps bool;
// h5_to_memmap.py:119-121: xy int16 (N, 2), t float64, p uint8): 13 B / event, converted in registers.
// xy_stride 1: separate x / y columns; 2: one interleaved (N, 2) array (y = x + 1).
// t: float64 or float32; the record holds (float)(t - t_offset), the subtraction in float64.
// p: EVK_P_U8_PM1 uint8/bool {0,1} -> 2p - 1 (what the loaders' get_events does, hdf5_dataset.py:22,
// memmap_dataset.py:23), EVK_P_U8 uint8 as is, EVK_P_I8 int8 as is.
struct ColsNative {
    const int16_t *x, *y;
    const void *t;
    const uint8_t *p;
    double t_offset;
    int xy_stride, t_f64, p_kind;
    __device__ __forceinline__ float pol(uint32_t b) const {
        return p_kind == EVK_P_U8_PM1 ? (float)(2 * (int)b - 1) : (p_kind == EVK_P_I8 ? (float)(int8_t)b : (float)b);
    }
    __device__ __forceinline__ void xy4(int64_t lo, int64_t q, Vec4<float> &xv, Vec4<float> &yv) const {
        if (xy_stride == 2) {
            const uint4 w = reinterpret_cast<const uint4 *>(x + 2 * lo)[q];  // x0 y0 | x1 y1 | x2 y2 | x3 y3
            const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) xv.v[k] = (float)(int16_t)(u[k] & 0xffffu), yv.v[k] = (float)(int16_t)(u[k] >> 16);
        } else {
            const uint2 a = reinterpret_cast<const uint2 *>(x + lo)[q], b = reinterpret_cast<const uint2 *>(y + lo)[q];
            const uint32_t ua[2] = {a.x, a.y}, ub[2] = {b.x, b.y};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                xv.v[2 * k] = (float)(int16_t)(ua[k] & 0xffffu), xv.v[2 * k + 1] = (float)(int16_t)(ua[k] >> 16);
                yv.v[2 * k] = (float)(int16_t)(ub[k] & 0xffffu), yv.v[2 * k + 1] = (float)(int16_t)(ub[k] >> 16);
            }
        }
    }
    __device__ __forceinline__ void tp4(int64_t lo, int64_t q, Vec4<float> &tv, Vec4<float> &pv) const {
        if (t_f64) {
            const Vec4<double> d = load4(static_cast<const double *>(t) + lo, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) tv.v[k] = (float)(d.v[k] - t_offset);
        } else {
            const Vec4<float> f = load4(static_cast<const float *>(t) + lo, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) tv.v[k] = (float)((double)f.v[k] - t_offset);
        }
        const uint32_t w = reinterpret_cast<const uint32_t *>(p + lo)[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) pv.v[k] = pol((w >> (8 * k)) & 0xffu);
    }
    // nv (1..4) valid events: every load is one naturally aligned block except the 2 x 16 bytes of a float64 t quad,
    // whose second half is skipped when it holds no valid event (it may start a new page)
    __device__ __forceinline__ void xy4n(int64_t lo, int64_t qb, uint32_t ql, int, Vec4<float> &xv, Vec4<float> &yv) const {
        xy4(lo + 4 * qb, ql, xv, yv);
    }
    __device__ __forceinline__ void tp4n(int64_t lo, int64_t qb, uint32_t ql, int nv, Vec4<float> &tv, Vec4<float> &pv) const {
        lo += 4 * qb;
        const uint32_t q = ql;
        if (t_f64) {
            const double2 a = reinterpret_cast<const double2 *>(static_cast<const double *>(t) + lo)[2 * q];
            double2 b = a;
            if (nv > 2) b = reinterpret_cast<const double2 *>(static_cast<const double *>(t) + lo)[2 * q + 1];
            tv.v[0] = (float)(a.x - t_offset), tv.v[1] = (float)(a.y - t_offset);
            tv.v[2] = (float)(b.x - t_offset), tv.v[3] = (float)(b.y - t_offset);
        } else {
            const Vec4<float> f = load4(static_cast<const float *>(t) + lo, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) tv.v[k] = (float)((double)f.v[k] - t_offset);
        }
        const uint32_t w = reinterpret_cast<const uint32_t *>(p + lo)[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) pv.v[k] = pol((w >> (8 * k)) & 0xffu);
    }
    __device__ __forceinline__ void p4(int64_t lo, int64_t q, Vec4<float> &pv) const {
        const uint32_t w = reinterpret_cast<const uint32_t *>(p + lo)[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) pv.v[k] = pol((w >> (8 * k)) & 0xffu);
    }
    __device__ __forceinline__ float x1(int64_t i) const { return (float)x[i * xy_stride]; }
    __device__ __forceinline__ float y1(int64_t i) const { return (float)y[i * xy_stride]; }
    __device__ __forceinline__ float t1(int64_t i) const {
        return (float)((t_f64 ? static_cast<const double *>(t)[i] : (double)static_cast<const float *>(t)[i]) - t_offset);
    }
    __device__ __forceinline__ float p1(int64_t i) const { return pol(p[i]); }
};

// LDS accumulators are float64: on gfx950 ds_add_f32 sustains only ~0.33 lane-ops/clk/CU (204 G/s chip-wide) while
// ds_add_f64 runs at ~2.9 (1.8 T/s) -- measured with tools/lds_probe.hip.  The per-event f32 values are exactly the
// reference's; summing them in f64 and rounding once at the flush is also closer to the true sum than f32 atomics.
typedef double acc_t;
__device__ __forceinline__ void lds_add(acc_t *p, float v) {
    __hip_atomic_fetch_add(p, (acc_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void voxel_bins_lds(acc_t *acc, int tpix, int local, int B, float tn, float p) {
    if (tn != tn) {  // dt == 0 (Q9): NaN in every bin of the pixel
        for (int b = 0; b < B; ++b) lds_add(acc + b * tpix + local, tn * p);
        return;
    }
    if (!(fabsf(p) <= 3.0e38f)) {  // a polarity that is not finite reaches EVERY bin (p * 0 = NaN), as in the reference
        for (int b = 0; b < B; ++b) lds_add(acc + b * tpix + local, p * fmaxf(0.0f, 1.0f - fabsf(tn - (float)b)));
        return;
    }
    const float fl = floorf(tn);
    const int b0 = (int)fmaxf(fminf(fl, (float)(B + 1)), -2.0f);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int b = b0 + k;
        if (b < 0 || b >= B) continue;
        const float val = p * fmaxf(0.0f, 1.0f - fabsf(tn - (float)b));
        if (val != 0.0f) lds_add(acc + b * tpix + local, val);
    }
}

__host__ __device__ inline int64_t bucket_cap(int64_t n, int ntiles) {
    const int64_t c = 4 * (n / (ntiles > 0 ? ntiles : 1));
    return c > 32768 ? c : 32768;  // uniform data is never split; at most ntiles/4 extra items
}
__host__ __device__ inline int bucket_max_items(int64_t n, int ntiles) {
    return ntiles + (int)(n / bucket_cap(n, ntiles)) + 1;
}
// Work-item budget of the three-pass bucketing (evk_tiled.hip), whose plan BALANCES non-uniform scenes: the tile kernels
// run one workgroup per item, all of them resident at once, so a launch lasts as long as its largest item.  When the
// fullest tile holds more than 1.25 x the mean, k_tile_scan_totals lowers the split threshold (by bisection, not below
// max(4096, 5/8 of the mean)) as far as this budget of 2 items per tile allows.
__host__ __device__ inline int bucket_item_budget(int ntiles) { return 2 * ntiles; }
__host__ __device__ inline int bucket_max_items_balanced(int64_t n, int ntiles) {
    const int a = bucket_max_items(n, ntiles), b = bucket_item_budget(ntiles) + 1;
    return a > b ? a : b;
}
static inline int make_grid(TileGrid &g, int dom_h, int dom_w, int tw_log2, int th_log2) {
    if (dom_h <= 0 || dom_w <= 0 || tw_log2 < 2 || tw_log2 > 8 || th_log2 < 2 || th_log2 > 8) return EVK_EINVAL;
    g.tw_log2 = tw_log2;
    g.th_log2 = th_log2;
    g.dom_w = dom_w;
    g.dom_h = dom_h;
    g.tiles_x = (dom_w + (1 << tw_log2) - 1) >> tw_log2;
    g.tiles_y = (dom_h + (1 << th_log2) - 1) >> th_log2;
    return EVK_OK;
}

#define EVK_MAX_TILES 8192

}  // namespace evk

static inline int native_cols(evk::ColsNative &c, const int16_t *x, const int16_t *y, int xy_stride, const void *t, int t_kind,
                       double t_offset, const void *p, int p_kind, int64_t n) {
    if ((xy_stride != 1 && xy_stride != 2) || (t_kind != EVK_T_F32 && t_kind != EVK_T_F64) ||
        (p_kind != EVK_P_U8_PM1 && p_kind != EVK_P_U8 && p_kind != EVK_P_I8) || !(t_offset == t_offset))
        return EVK_EINVAL;
    if (n > 0 && (!x || !t || !p || (xy_stride == 1 && !y))) return EVK_EINVAL;
    c.x = x, c.y = xy_stride == 2 ? x + 1 : y, c.t = t, c.p = static_cast<const uint8_t *>(p);
    c.t_offset = t_offset, c.xy_stride = xy_stride, c.t_f64 = t_kind == EVK_T_F64, c.p_kind = p_kind;
    return EVK_OK;
}
