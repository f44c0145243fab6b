// Event images on the one-pass partition + LDS-tile design (round 4): events_to_image (image.py:5-44, nearest, numpy
// path), events_to_image_torch (image.py:46-100, nearest and bilinear) and with it interpolate_to_image (image.py:102-115).
//
// The global-atomic kernels of evk_scatter.hip cap these functions at ~21 G atomics/s -- 1 (nearest) or 4 (bilinear) per
// event -- i.e. 1-3.5 % of the 12 B/event HBM roofline.  Here the events take the voxel grid's route: k_part_sorted
// (evk_part2.h) sorts sub-chunks of 8 K consecutive events by output tile in LDS and writes each one back as a contiguous
// run of records, plus a (sub-chunk, tile) table; a tile kernel then pulls every tile's ~16-record segments out of the runs
// and accumulates in LDS.  12 B/event are read (x, y, weight), 4 (nearest) or 8 (bilinear) written and read once more -- plus
// 4 for the exact float32 weight, only in sub-chunks that hold a weight the record cannot carry.
//
// Nearest: record = [31:11] top 21 bits of the weight | [10] wide | [9:0] cell (evk_part2.h, V2_FMT_IMGN).  The tile
// kernel adds integers (int32 LDS atomics) whenever it can -- the integer image of the numpy path, which stays BIT-EXACT
// with np.bincount, and float32 calls all of whose weights are +1, -1 or +0 (the partition kernel counts the others) -- and
// float64 otherwise; every pixel of a tile is then written once (plain stores, no global atomics).
// Bilinear: record = {x - tile x0, y - tile y0} as exact float32 (both >= 0), the weight's code -- +1, -1, +0, other -- in their
// two sign bits; a sub-chunk with an "other" weight writes its float32 weights as a second dense run (V2_FMT_IMGB, evk_part2.h:
// 8 B/event for the unit weights of real event streams instead of 12).  The tile's accumulator is a window one pixel
// wider and higher than the tile (px + 1, py + 1 of its last column / row); the four products are evaluated in float32
// in the reference's order (image.py:111-114) and added as int64 fixed point (2^-30 steps, unit weights) or float64; the
// window's interior is added to the image with plain read-modify-writes, its one-pixel ring -- shared with the
// neighbouring tiles' windows -- with global float atomics (2 (tw + th) of them per tile).
//
// A tile's ~16-record segments are cut into 8-record chunks, listed per wave in LDS and handed to groups of 4 lanes, the next
// round's loads in flight while a round is accumulated (img_records below: the scheme of k_voxel_tiles2 as a function template
// over the record type).  (The first version gave every LANE one segment: 40 % idle lanes, the bilinear kernel VALU-bound at
// 72 us against 30 us.)  Hot tiles are cut by the partition kernel's plan exactly as for the voxel grid (pieces = ranges of
// sub-chunks, partial tiles summed by the last piece to arrive, in piece order; integer accumulators hand over exact values).
#include "evk_part2.h"
#include "evk_splat.h"

namespace evk {

// ---- column sources (evk_part.h) -----------------------------------------------------------------------------------
// x, y, w float32: events_to_image_torch.  Nearest (key_of): image.py:87-95 -- events with x >= clipx or y >= clipy go to
// pixel (0, 0) WITH their weight (quirk Q8), coordinates are truncated toward zero, negative indices wrap once, NaN and
// anything still outside is counted (IndexError).  Bilinear (key_rel): image.py:79-86.
struct SrcImgF32 {
    static constexpr int G = 4, XYW = 8, TPW = 4;
    const float *x, *y, *w;
    float clipx, clipy;
    float *img;   // (h, wd): the rare path of the bilinear format adds to it directly
    int h, wd;
    template <bool NT = false>
    __device__ __forceinline__ void load_xy(int64_t ev0, uint32_t gl, uint32_t *r) const {
        const uint4 a = load_col16<NT>(reinterpret_cast<const float *>(x), ev0, gl), b = load_col16<NT>(reinterpret_cast<const float *>(y), ev0, gl);
        r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
    }
    template <bool NT = false>
    __device__ __forceinline__ void load_tp(int64_t ev0, uint32_t gl, uint32_t *r) const {
        const uint4 a = load_col16<NT>(w, ev0, gl);
        r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w;
    }
    __device__ __forceinline__ int key_of(const uint32_t *r, int e, const TileGridG &g, uint32_t &cell) const {
        const float xf = __uint_as_float(r[e]), yf = __uint_as_float(r[4 + e]);
        const bool keep = !(xf >= clipx) & !(yf >= clipy);   // the mask multiplies the INDICES only (image.py:93-95)
        // (a masked event's indices are x.long() * 0 = 0 also when its other coordinate is NaN; an unmasked NaN raises)
        return nearest_key_cell_int(keep ? (int)xf : 0, keep ? (int)yf : 0, !keep | ((xf == xf) & (yf == yf)), g, cell);
    }
    // Bilinear: tile of (floor(x), floor(y)) and the coordinates relative to that tile's origin.  -3 (xr, yr = x, y):
    // an event the LDS windows cannot take -- masked (it lands on pixel (0, 0) with weight w * 0), a pixel or its right /
    // lower neighbour outside the image (negative: wraps; beyond: IndexError), NaN -- goes to rare() below.
    __device__ __forceinline__ int key_rel(const uint32_t *r, int e, const TileGridG &g, float &xr, float &yr) const {
        const float xf = __uint_as_float(r[e]), yf = __uint_as_float(r[4 + e]);
        const float fx = floorf(xf), fy = floorf(yf);
        const bool ok = !(xf >= clipx) & !(yf >= clipy) & (fx >= 0.0f) & (fx <= (float)(g.dom_w - 2)) & (fy >= 0.0f) &
                        (fy <= (float)(g.dom_h - 2));
        const int px = ok ? (int)fx : 0, py = ok ? (int)fy : 0;
        const int tx = tile_of(px, g.ix), ty = tile_of(py, g.iy);
        xr = ok ? xf - (float)__mul24(tx, g.tw) : xf;   // exact: a multiple of ulp(x) below x
        yr = ok ? yf - (float)__mul24(ty, g.th) : yf;
        return ok ? __mul24(ty, g.tiles_x) + tx : -3;
    }
    __device__ __forceinline__ uint32_t w_bits(const uint32_t *r, int e) const { return r[e]; }
    __device__ __forceinline__ uint32_t payload(const uint32_t *r, int e, bool &wide, bool &unit) const {
        const uint32_t pbits = r[e];
        wide = ((pbits & ~V2_P_MASK) != 0u) | ((pbits & 0x7F800000u) == 0x7F800000u);   // (not finite: always wide)
        unit = ((pbits & 0x7FFFFFFFu) == 0x3F800000u) | (pbits == 0u);
        return pbits;
    }
    // the direct kernel's per-event code (evk_scatter.hip, k_image_bilinear_f32); false = IndexError
    __device__ __forceinline__ bool rare(float xf, float yf, float wv) const {
        const float mask = (!(xf >= clipx) && !(yf >= clipy)) ? 1.0f : 0.0f;
        // a masked event adds w * 0 * (...) to the pixels (0..1, 0..1): nothing, unless a factor is not finite
        if (mask == 0.0f && fabsf(wv) <= 3.0e38f && fabsf(xf) <= 3.0e38f && fabsf(yf) <= 3.0e38f) return wd >= 2 && h >= 2;
        const float fx = floorf(xf), fy = floorf(yf);
        Splat s;
        s.dx = xf - fx;
        s.dy = yf - fy;
        s.px = (long long)(fx * mask);
        s.py = (long long)(fy * mask);
        return splat_iwe(img, h, wd, s, wv * mask);
    }
};

// x, y, w int32 on the (H+1, W+1) canvas: events_to_image, numpy path (np.ravel_multi_index + np.bincount, image.py:28-38).
// No wrap: a negative coordinate is a ValueError there.  w == NULL: every weight is 1 (the meanval count image).
struct SrcImgI32 {
    static constexpr int G = 4, XYW = 8, TPW = 4;
    const int32_t *x, *y, *w;
    template <bool NT = false>
    __device__ __forceinline__ void load_xy(int64_t ev0, uint32_t gl, uint32_t *r) const {
        const uint4 a = load_col16<NT>(reinterpret_cast<const float *>(x), ev0, gl), b = load_col16<NT>(reinterpret_cast<const float *>(y), ev0, gl);
        r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
    }
    template <bool NT = false>
    __device__ __forceinline__ void load_tp(int64_t ev0, uint32_t gl, uint32_t *r) const {
        uint4 a = make_uint4(1u, 1u, 1u, 1u);
        if (w) a = load_col16<NT>(reinterpret_cast<const float *>(w), ev0, gl);
        r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w;
    }
    __device__ __forceinline__ int key_of(const uint32_t *r, int e, const TileGridG &g, uint32_t &cell) const {
        return nearest_key_cell_int<false>((int)r[e], (int)r[4 + e], true, g, cell);
    }
    __device__ __forceinline__ uint32_t w_bits(const uint32_t *r, int e) const { return r[e]; }
    __device__ __forceinline__ uint32_t payload(const uint32_t *r, int e, bool &wide, bool &unit) const {
        const int wv = (int)r[e];
        wide = ((wv << (V2_LB + 1)) >> (V2_LB + 1)) != wv;   // does not fit the record's 21 signed bits
        unit = (uint32_t)(wv + 1) <= 2u;
        return (uint32_t)wv << (V2_LB + 1);
    }
};

// x, y, t, p float32: the average-timestamp images (image.py:219-353; V2_FMT_IMGT).  Every event is splat bilinearly, with
// weight nts (its normalised time stamp: mode 0 (t - ta) / td, 1 (-t + ta) / td, 2 t) into the time image of its class and
// with weight 1 into the count image of its class -- positive / non-positive polarity.  Upstream quirk kept (as in the direct
// kernel, evk_scatter.hip:k_timestamp_images_f32): an event with x >= clipx or y >= clipy is moved to pixel (0, 0) -- the mask
// multiplies the INDICES -- but keeps its fractions and its weights (masked_ps is never used, image.py:264,336).
struct SrcTsF32 {
    static constexpr int G = 4, XYW = 8, TPW = 8;
    const float *x, *y, *t, *p;
    float clipx, clipy;
    int mode;
    float ta, td;
    float *out4;   // (4, h, wd): time+ | count+ | time- | count-; the rare path adds to it directly
    int h, wd;
    template <bool NT = false>
    __device__ __forceinline__ void load_xy(int64_t ev0, uint32_t gl, uint32_t *r) const {
        const uint4 a = load_col16<NT>(x, ev0, gl), b = load_col16<NT>(y, ev0, gl);
        r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
    }
    template <bool NT = false>
    __device__ __forceinline__ void load_tp(int64_t ev0, uint32_t gl, uint32_t *r) const {
        const uint4 a = load_col16<NT>(t, ev0, gl), b = load_col16<NT>(p, ev0, gl);
        r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
    }
    // Tile of (floor(x), floor(y)) -- of pixel (0, 0) for a masked event -- and the coordinates relative to that tile's origin
    // (a masked event: its fractions, i.e. pixel (0, 0) of tile 0 with dx, dy intact).  -3 (xr, yr = x, y): an event the LDS
    // windows cannot take (a pixel or its right / lower neighbour outside the image, non-finite coordinates) -> rare_ts().
    __device__ __forceinline__ int key_rel(const uint32_t *r, int e, const TileGridG &g, float &xr, float &yr) const {
        const float xf = __uint_as_float(r[e]), yf = __uint_as_float(r[4 + e]);
        const float fx = floorf(xf), fy = floorf(yf);
        const bool masked = (xf >= clipx) | (yf >= clipy);
        const bool finite = fabsf(xf) <= 3.0e38f && fabsf(yf) <= 3.0e38f;
        const bool inwin = (fx >= 0.0f) & (fx <= (float)(g.dom_w - 2)) & (fy >= 0.0f) & (fy <= (float)(g.dom_h - 2));
        const bool ok = finite & (masked | inwin);
        const int px = (ok & !masked) ? (int)fx : 0, py = (ok & !masked) ? (int)fy : 0;
        const int tx = tile_of(px, g.ix), ty = tile_of(py, g.iy);
        xr = ok ? (masked ? xf - fx : xf - (float)__mul24(tx, g.tw)) : xf;   // exact either way
        yr = ok ? (masked ? yf - fy : yf - (float)__mul24(ty, g.th)) : yf;
        return ok ? __mul24(ty, g.tiles_x) + tx : -3;
    }
    // from_events (modes 0 / 1): ta = ts[0] (mode 1: ts[-1]), td = (ts[-1] - ts[0]) + 1e-6 in float32 (image.py:326-329)
    __device__ __forceinline__ void time_constants(int64_t n, int from_events, float &a, float &d) const {
        a = ta, d = td;
        if (from_events && mode != 2) {
            const float t0 = t[0], t1 = t[n - 1];
            a = mode == 0 ? t0 : t1;
            d = (t1 - t0) + 1e-6f;
        }
    }
    __device__ __forceinline__ float nts_of(float tf, float a, float d) const {
        return mode == 0 ? (tf - a) / d : (mode == 1 ? (-tf + a) / d : tf);
    }
    __device__ __forceinline__ uint32_t nts_bits(const uint32_t *r, int e, float a, float d) const {
        return __float_as_uint(nts_of(__uint_as_float(r[e]), a, d));
    }
    // pos_events_mask = ps > 0, neg_events_mask = ps <= 0 (image.py:258-259); a NaN polarity is in neither
    __device__ __forceinline__ uint32_t cls(const uint32_t *r, int e) const {
        const float pv = __uint_as_float(r[4 + e]);
        return pv > 0.0f ? 0u : (pv <= 0.0f ? 1u : 2u);
    }
    // the direct kernel's per-event code (evk_scatter.hip, k_timestamp_images_f32); false = IndexError
    __device__ __forceinline__ bool rare_ts(float xf, float yf, const uint32_t *r, int e, float a, float d) const {
        const uint32_t k = cls(r, e);
        if (k == 2u) return true;
        const float mask = (!(xf >= clipx) && !(yf >= clipy)) ? 1.0f : 0.0f;
        const float fx = floorf(xf), fy = floorf(yf);
        Splat s;
        s.dx = xf - fx;
        s.dy = yf - fy;
        s.px = (long long)(fx * mask);
        s.py = (long long)(fy * mask);
        float *val = out4 + (int64_t)(2u * k) * h * wd, *cnt = val + (int64_t)h * wd;
        return splat_iwe(val, h, wd, s, nts_of(__uint_as_float(r[e]), a, d)) && splat_iwe(cnt, h, wd, s, 1.0f);
    }
};

// pxs, pys int64, dxs, dys, w float32: interpolate_to_image on caller-computed pixels and fractions (image.py:102-115;
// V2_FMT_IMGX).  The bilinear record holds {x - tile x0, y - tile y0} with x = px + dx: it can carry the event when that sum IS
// a float32 value with floor(x) == px and x - px == dx -- always the case when the caller took px, dx from a float32 coordinate,
// as every upstream caller does (image.py:79-82, 199-202, 253-256) -- and the pixel and its right / lower neighbour are inside
// the image.  Anything else (a sum that rounds, fractions outside [0, 1), NaN, pixels that wrap or raise in index_put_) is
// re-read by its index and takes the direct kernel's code (rare_at).
struct SrcIdxF32 {
    static constexpr int G = 4, XYW = 24, TPW = 4;
    const long long *px, *py;
    const float *dx, *dy, *w;
    float *img;
    int h, wd;
    template <bool NT = false>
    __device__ __forceinline__ void load_xy(int64_t ev0, uint32_t gl, uint32_t *r) const {
        // (16-byte loads at dword alignment, as load_col16: 2 gl and 2 gl + 1 are the two halves of this lane's four int64)
        const float *fx = reinterpret_cast<const float *>(px + ev0), *fy = reinterpret_cast<const float *>(py + ev0);
        const uint4 a0 = load_col16<NT>(fx, 0, 2 * gl), a1 = load_col16<NT>(fx, 0, 2 * gl + 1);
        const uint4 b0 = load_col16<NT>(fy, 0, 2 * gl), b1 = load_col16<NT>(fy, 0, 2 * gl + 1);
        const uint4 c = load_col16<NT>(dx, ev0, gl), d = load_col16<NT>(dy, ev0, gl);
        r[0] = a0.x, r[1] = a0.y, r[2] = a0.z, r[3] = a0.w, r[4] = a1.x, r[5] = a1.y, r[6] = a1.z, r[7] = a1.w;
        r[8] = b0.x, r[9] = b0.y, r[10] = b0.z, r[11] = b0.w, r[12] = b1.x, r[13] = b1.y, r[14] = b1.z, r[15] = b1.w;
        r[16] = c.x, r[17] = c.y, r[18] = c.z, r[19] = c.w, r[20] = d.x, r[21] = d.y, r[22] = d.z, r[23] = d.w;
    }
    template <bool NT = false>
    __device__ __forceinline__ void load_tp(int64_t ev0, uint32_t gl, uint32_t *r) const {
        const uint4 a = load_col16<NT>(w, ev0, gl);
        r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w;
    }
    __device__ __forceinline__ int key_rel(const uint32_t *r, int e, const TileGridG &g, float &xr, float &yr) const {
        // (a pixel inside the image has a zero high word; its low word is compared as unsigned, so a negative one fails too)
        const uint32_t pxl = r[2 * e], pxh = r[2 * e + 1], pyl = r[8 + 2 * e], pyh = r[8 + 2 * e + 1];
        const float fdx = __uint_as_float(r[16 + e]), fdy = __uint_as_float(r[20 + e]);
        const bool inwin = ((pxh | pyh) == 0u) & (pxl <= (uint32_t)(g.dom_w - 2)) & (pyl <= (uint32_t)(g.dom_h - 2));
        const int ipx = inwin ? (int)pxl : 0, ipy = inwin ? (int)pyl : 0;
        const float fpx = (float)ipx, fpy = (float)ipy;
        const float xf = fpx + fdx, yf = fpy + fdy;
        const float flx = floorf(xf), fly = floorf(yf);
        const bool ok = inwin & (flx == fpx) & (xf - fpx == fdx) & (fly == fpy) & (yf - fpy == fdy);
        const int tx = tile_of(ipx, g.ix), ty = tile_of(ipy, g.iy);
        xr = ok ? xf - (float)__mul24(tx, g.tw) : 0.0f;
        yr = ok ? yf - (float)__mul24(ty, g.th) : 0.0f;
        return ok ? __mul24(ty, g.tiles_x) + tx : -3;
    }
    __device__ __forceinline__ uint32_t w_bits(const uint32_t *r, int e) const { return r[e]; }
    // the direct kernel's per-event code (evk_scatter.hip, k_splat_indexed_f32); false = IndexError
    __device__ __forceinline__ bool rare_at(int64_t i) const {
        Splat s;
        s.px = px[i], s.py = py[i], s.dx = dx[i], s.dy = dy[i];
        return splat_iwe(img, h, wd, s, w[i]);
    }
};

// ---- the derivative splats (V2_FMT_IMGD: the side run holds the event's index, the tile kernel fetches its weights) ----------
// pxs, pys int64, dxs, dys float32; w1, w2 float32 (2, n): interpolate_to_derivative_img (image.py:117-136).  Keys as SrcIdxF32.
struct SrcIdxDrvF32 {
    static constexpr int G = 4, XYW = 24, TPW = 1;
    const long long *px, *py;
    const float *dx, *dy, *w1, *w2;
    int64_t n;
    float *dimg;
    int h, wd;
    template <bool NT = false>
    __device__ __forceinline__ void load_xy(int64_t ev0, uint32_t gl, uint32_t *r) const {
        SrcIdxF32{px, py, dx, dy, nullptr, nullptr, h, wd}.template load_xy<NT>(ev0, gl, r);
    }
    template <bool NT = false>
    __device__ __forceinline__ void load_tp(int64_t, uint32_t, uint32_t *r) const { r[0] = 0u; }
    __device__ __forceinline__ int key_rel(const uint32_t *r, int e, const TileGridG &g, float &xr, float &yr) const {
        return SrcIdxF32{px, py, dx, dy, nullptr, nullptr, h, wd}.key_rel(r, e, g, xr, yr);
    }
    // the direct kernel's per-event code (evk_scatter.hip, k_splat_drv_indexed_f32, C = 2); false = IndexError
    __device__ __forceinline__ bool rare_at(int64_t i) const {
        long long x0 = px[i], x1 = x0 + 1, y0 = py[i], y1 = y0 + 1;
        if (!(wrap_index(x0, wd) && wrap_index(x1, wd) && wrap_index(y0, h) && wrap_index(y1, h))) return false;
        const float fx = dx[i], fy = dy[i], ax = 1.0f - fx, ay = 1.0f - fy;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float a = w1[(int64_t)c * n + i], b = w2[(int64_t)c * n + i];
            float *d = dimg + (int64_t)c * h * wd;
            atomic_add(d + y0 * wd + x0, a * (-ay) + b * (-ax));
            atomic_add(d + y0 * wd + x1, a * ay + b * (-fx));
            atomic_add(d + y1 * wd + x0, a * (-fy) + b * ax);
            atomic_add(d + y1 * wd + x1, a * fy + b * fx);
        }
        return true;
    }
};
// x, y, p float64, jx, jy float64 (2, n) or NULL: events_to_image_drv (image.py:162-217) -- coordinates and weights are cast to
// float32 BEFORE floor (Q7); the clip mask multiplies indices AND weights (:203-206), so a masked event adds p * 0 at pixel (0, 0).
struct SrcDrvF64 {
    static constexpr int G = 4, XYW = 16, TPW = 1;
    const double *x, *y, *p, *jx, *jy;
    int64_t n;
    float clipx, clipy;
    float *img, *dimg;
    int h, wd;
    template <bool NT = false>
    __device__ __forceinline__ void load_xy(int64_t ev0, uint32_t gl, uint32_t *r) const {
        const float *fx = reinterpret_cast<const float *>(x + ev0), *fy = reinterpret_cast<const float *>(y + ev0);
        const uint4 a0 = load_col16<NT>(fx, 0, 2 * gl), a1 = load_col16<NT>(fx, 0, 2 * gl + 1);
        const uint4 b0 = load_col16<NT>(fy, 0, 2 * gl), b1 = load_col16<NT>(fy, 0, 2 * gl + 1);
        r[0] = a0.x, r[1] = a0.y, r[2] = a0.z, r[3] = a0.w, r[4] = a1.x, r[5] = a1.y, r[6] = a1.z, r[7] = a1.w;
        r[8] = b0.x, r[9] = b0.y, r[10] = b0.z, r[11] = b0.w, r[12] = b1.x, r[13] = b1.y, r[14] = b1.z, r[15] = b1.w;
    }
    template <bool NT = false>
    __device__ __forceinline__ void load_tp(int64_t, uint32_t, uint32_t *r) const { r[0] = 0u; }
    __device__ __forceinline__ int key_rel(const uint32_t *r, int e, const TileGridG &g, float &xr, float &yr) const {
        const float xf = (float)__hiloint2double((int)r[2 * e + 1], (int)r[2 * e]);
        const float yf = (float)__hiloint2double((int)r[8 + 2 * e + 1], (int)r[8 + 2 * e]);
        const float fx = floorf(xf), fy = floorf(yf);
        const bool ok = !(xf >= clipx) & !(yf >= clipy) & (fx >= 0.0f) & (fx <= (float)(g.dom_w - 2)) & (fy >= 0.0f) &
                        (fy <= (float)(g.dom_h - 2));
        const int px = ok ? (int)fx : 0, py = ok ? (int)fy : 0;
        const int tx = tile_of(px, g.ix), ty = tile_of(py, g.iy);
        xr = ok ? xf - (float)__mul24(tx, g.tw) : 0.0f;
        yr = ok ? yf - (float)__mul24(ty, g.th) : 0.0f;
        return ok ? __mul24(ty, g.tiles_x) + tx : -3;
    }
    // the direct kernel's per-event code (evk_scatter.hip, k_image_drv_f64); false = IndexError
    __device__ __forceinline__ bool rare_at(int64_t i) const {
        const float xf = (float)x[i], yf = (float)y[i], pf = (float)p[i];
        const float mask = (!(xf >= clipx) && !(yf >= clipy)) ? 1.0f : 0.0f;
        // a masked event adds p * 0 * (...) to the pixels (0..1, 0..1): nothing, unless a factor is not finite
        if (mask == 0.0f && fabsf(pf) <= 3.0e38f && fabsf(xf) <= 3.0e38f && fabsf(yf) <= 3.0e38f &&
            (!jx || (fabs(jx[i]) <= 3.0e38 && fabs(jx[n + i]) <= 3.0e38 && fabs(jy[i]) <= 3.0e38 && fabs(jy[n + i]) <= 3.0e38)))
            return wd >= 2 && h >= 2;
        const float fx = floorf(xf), fy = floorf(yf);
        Splat s;
        s.dx = xf - fx;
        s.dy = yf - fy;
        s.px = (long long)(fx * mask);
        s.py = (long long)(fy * mask);
        const float mp = pf * mask;
        if (!splat_iwe(img, h, wd, s, mp)) return false;
        if (jx) {
            long long x0 = s.px, x1 = s.px + 1, y0 = s.py, y1 = s.py + 1;
            wrap_index(x0, wd), wrap_index(x1, wd), wrap_index(y0, h), wrap_index(y1, h);
            const float ax = 1.0f - s.dx, ay = 1.0f - s.dy;
            const float w1[2] = {(float)jx[i] * mp, (float)jx[n + i] * mp};
            const float w2[2] = {(float)jy[i] * mp, (float)jy[n + i] * mp};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float *d = dimg + (int64_t)c * h * wd;
                atomic_add(d + y0 * wd + x0, w1[c] * (-ay) + w2[c] * (-ax));
                atomic_add(d + y0 * wd + x1, w1[c] * ay + w2[c] * (-s.dx));
                atomic_add(d + y1 * wd + x0, w1[c] * (-s.dy) + w2[c] * ax);
                atomic_add(d + y1 * wd + x1, w1[c] * s.dy + w2[c] * s.dx);
            }
        }
        return true;
    }
};
// the weights of an event by its index: mp (the image's weight), a = w1[0..1], b = w2[0..1]
struct WgtIdxDrvF32 {
    static constexpr bool IWE = false;
    const float *w1, *w2;
    int64_t n;
    __device__ __forceinline__ void get(uint32_t i, bool grad, float &mp, float (&a)[2], float (&b)[2]) const {
        mp = 0.0f;
        a[0] = w1[i], a[1] = w1[n + i], b[0] = w2[i], b[1] = w2[n + i];
    }
};
struct WgtDrvF64 {
    static constexpr bool IWE = true;
    const double *p, *jx, *jy;
    int64_t n;
    __device__ __forceinline__ void get(uint32_t i, bool grad, float &mp, float (&a)[2], float (&b)[2]) const {
        mp = (float)p[i];
        a[0] = a[1] = b[0] = b[1] = 0.0f;
        if (grad) {
            a[0] = (float)jx[i] * mp, a[1] = (float)jx[n + i] * mp;
            b[0] = (float)jy[i] * mp, b[1] = (float)jy[n + i] * mp;
        }
    }
};

#ifndef IMG_WG
#define IMG_WG 512    // threads of a tile workgroup (768 and up: the chunk lists no longer fit the 64 KB of static LDS)
#endif
#ifndef IMG_U
#define IMG_U 1       // chunk loads per lane group in flight (A/B, 10 M events: bilinear tile kernel 31.8 us with 1, 34.0 with 2,
                      // 37 with 4; nearest 15.0 / 15.0 / 17.0)
#endif
#define IMG_FIX_ONE 1073741824.0f   // 2^30: fixed-point unit of the bilinear window (unit weights: |product| <= 1)

// 16 / 8 bytes at any dword boundary (global loads need no more alignment than that)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) Quad4 {
    u32x4 v;
};
struct __attribute__((packed, aligned(4))) Pair2 {
    u32x2 v;
};
__device__ __forceinline__ uint4 load_u4(const void *p) {
    const u32x4 v = static_cast<const Quad4 *>(p)->v;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 load_u2(const void *p) {
    const u32x2 v = static_cast<const Pair2 *>(p)->v;
    return make_uint2(v.x, v.y);
}

// The plan of a tile workgroup (as k_voxel_tiles2): work item -> tile, piece of a cut tile, its range of sub-chunks.
struct ImgItem {
    int tile;
    uint32_t item, first_item, nparts, part_id;
    int sc_lo, sc_hi;
};
__device__ __forceinline__ bool img_item(const uint32_t *index, int ntiles, const Part2 &q, int flags, ImgItem &it) {
    const uint32_t *part_start = index + V2_PART, *item_tile = index + V2_ITEM(ntiles);
    const uint32_t nitems = part_start[ntiles];
    if (blockIdx.x >= nitems) return false;
    uint32_t item = blockIdx.x;
    // XCD-aware order (workgroup b runs on XCD b % 8): XCD k takes a contiguous range of the tile-ordered items, whose
    // segments are neighbours in every run -- but only when no tile was cut (evk_voxel2.hip)
    if (!(flags & EVK_VOXEL2_NO_XCD_ORDER) && nitems == (uint32_t)ntiles) {
        const uint32_t k = blockIdx.x & 7u, j = blockIdx.x >> 3, q8 = nitems >> 3, r8 = nitems & 7u;
        item = k * q8 + (k < r8 ? k : r8) + j;
    }
    it.item = item, it.tile = (int)item, it.first_item = item, it.nparts = 1u;
    if (nitems != (uint32_t)ntiles) {
        it.tile = (int)item_tile[item];
        it.first_item = part_start[it.tile], it.nparts = part_start[it.tile + 1] - it.first_item;
    }
    it.part_id = item - it.first_item;
    it.sc_lo = (int)(((int64_t)q.nsc * it.part_id) / it.nparts);
    it.sc_hi = (int)(((int64_t)q.nsc * (it.part_id + 1)) / it.nparts);
    return true;
}

// The last piece of a cut tile to arrive (relaxed agent-scope ticket; the partial tiles were stored with agent-scope
// stores and every wave has drained them: no fence, evk_voxel2.hip)
__device__ __forceinline__ bool img_last_part(uint32_t *index, int ntiles, const ImgItem &it) {
    __shared__ int is_last;
    EVK_HANDOVER_DRAIN();
    if (threadIdx.x == 0) {
        uint32_t *counter = index + V2_COUNTER(ntiles) + it.tile;
        const uint32_t prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == it.nparts - 1);
        if (is_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (is_last) EVK_HANDOVER_ACQUIRE();
    return is_last != 0;
}

// The records of a tile sit in ~16-record segments, one per sub-chunk (evk_part2.h).  As in k_voxel_tiles2 every thread
// fetches the table entry of one sub-chunk -- in equal batches interleaved over the waves (slot = lane * NW + wave), so that
// a short range (a piece of a hot tile) still gives every wave its share --, each wave cuts its 64 segments into CHUNKS of
// 8 records starting at the segment's first record, lists them in LDS ({first record, end of the segment}) and hands them to
// groups of 4 lanes, 2 records per lane: every lane has work whatever the segment lengths are (one segment per LANE, the
// first version of these kernels, left 40 % of the lanes idle and made the bilinear kernel VALU-bound at 72 us per 10 M
// events).  A wave lists all chunks of its segments, in as many passes over its lanes as its list needs; only a segment
// that alone overflows the list (> 3584 records) is streamed by the whole wave.  load(pos) -> L fetches the records pos,
// pos + 1 (unconditionally: `pos` is always inside the record buffer); use(L, pos, end) accumulates them.  The loads of
// the next round are in flight while a round is accumulated.
#define IMG_CAP 448   // chunk descriptors per wave
// (round 6: entry [IMG_CAP] of every list is a ZERO entry -- a lane group without a chunk reads it: one v_min and an
// unconditional LDS read instead of a compare, two zero moves and an exec-masked read per list access, as in k_voxel_tiles2)
template <int WG, int U, typename L, typename LoadF, typename UseF>
__device__ __forceinline__ void img_records(const uint32_t *table, const Part2 &q, const ImgItem &it, uint2 (*cseg)[IMG_CAP + 1],
                                            LoadF load, UseF use) {
    constexpr int NW = WG / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane * NW + wave, sub = lane & 3, grp = lane >> 2;
    const uint32_t *col = table + it.tile;
    auto wave_scan = [&](uint32_t v) {   // inclusive
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        return incl;
    };
    auto rounds = [&](const uint32_t total) {
        auto meta = [&](uint32_t j0, uint2(&cs)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t j = j0 + 16u * u + grp;
                j = j < total ? j : (uint32_t)IMG_CAP;
                cs[u] = cseg[wave][j];
            }
        };
        auto fire = [&](const uint2(&cs)[U], L(&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pos = cs[u].x + 2u * sub;
                v[u] = load(pos < cs[u].y ? pos : 2u * sub);
            }
        };
        auto eat = [&](const uint2(&cs)[U], const L(&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pos = cs[u].x + 2u * sub;
                if (pos < cs[u].y) use(v[u], pos, cs[u].y);
            }
        };
        constexpr uint32_t step = 16u * U;
        uint2 ca[U], cb[U];
        L va[U], vb[U];
        meta(0u, ca);
        fire(ca, va);
        for (uint32_t j0 = 0; j0 < total; j0 += 2u * step) {
            meta(j0 + step, cb);
            fire(cb, vb);          // round j0 + step in flight
            eat(ca, va);           // round j0
            meta(j0 + 2u * step, ca);
            fire(ca, va);          // round j0 + 2 step in flight
            eat(cb, vb);           // round j0 + step
        }
    };
    auto entry = [&](int my) -> uint32_t { return my < it.sc_hi ? col[(int64_t)my * q.nt_pad] : 0u; };
    if (threadIdx.x < NW) cseg[threadIdx.x][IMG_CAP] = make_uint2(0u, 0u);   // (ordered before the first rounds by the first batch's barrier)
    uint32_t ent_next = entry(it.sc_lo + slot);
    for (int base = it.sc_lo; base < it.sc_hi; base += WG) {
        const uint32_t ent = ent_next;
        ent_next = entry(base + WG + slot);   // in flight while this batch is processed
        const uint32_t cnt = ent >> 16, p0 = (uint32_t)(base + slot) * (uint32_t)q.S + (ent & 0xFFFFu), e0 = p0 + cnt;
        const uint32_t nch = (cnt + 7u) >> 3;
        const bool is_long = nch > (uint32_t)IMG_CAP;
        const uint32_t mych = is_long ? 0u : nch;
        const uint32_t incl = wave_scan(mych), total = __shfl(incl, 63, 64);
        // (all tiles walking the runs in step keeps each run L2-hot while its segments are pulled: evk_voxel2.hip; the
        // barrier also separates this batch's list from the previous batch's rounds)
        __syncthreads();
        uint32_t done = 0;
        while (done < total) {   // (wave-uniform) lanes whose chunks fit the list, in scan order
            const bool take = mych != 0u && incl - mych >= done && incl <= done + (uint32_t)IMG_CAP;
            const uint64_t tm = __ballot(take);
            const uint32_t n_this = __shfl(incl, 63 - __builtin_clzll(tm), 64) - done;
            if (take)
                for (uint32_t k = 0; k < mych; ++k) cseg[wave][incl - mych - done + k] = make_uint2(p0 + 8u * k, e0);
            rounds(n_this);
            done += n_this;
        }
        uint64_t m = __ballot(is_long);
        while (m) {   // a segment that alone overflows the list: the whole wave, 2 records per lane, four loads in flight
            const int s = __builtin_ctzll(m);
            m &= m - 1;
            const uint32_t b2 = __shfl(p0, s, 64), e2 = __shfl(e0, s, 64);
            for (uint32_t p2 = b2 + 2u * lane; p2 < e2; p2 += 512u) {
                L v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = load(p2 + 128u * u < e2 ? p2 + 128u * u : p2);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (p2 + 128u * u < e2) use(v[u], p2 + 128u * u, e2);
            }
        }
    }
}

// ---- nearest: one accumulator per pixel ---------------------------------------------------------------------------
// INT: int32 weights, int32 canvas (bit-exact).  Else float32 weights and image: integer counts while every weight of the
// call is +1, -1 or +0 (index[7], set by the partition kernel), float64 cells otherwise.  A call with weights that do not
// fit their record (index[1]) reads them from the side runs.
struct RecN {
    uint2 r, w;   // two one-word records; their exact weights (only loaded when the call has wide ones)
};
template <int WG, bool INT>
__global__ void __launch_bounds__(WG) k_image_tiles_n(const uint32_t *__restrict__ rec, const uint32_t *__restrict__ side,
                                                      const uint32_t *__restrict__ table, uint32_t *__restrict__ index,
                                                      TileGridG g, Part2 q, int flags, void *__restrict__ out_,
                                                      void *__restrict__ staging_) {
    __shared__ int cnt32[EVK_GRIDG_MAX_CELLS];
    __shared__ acc_t acc64[INT ? 1 : EVK_GRIDG_MAX_CELLS];
    __shared__ uint2 cseg[WG / 64][IMG_CAP + 1];
    const int ntiles = g.tiles_x * g.tiles_y;
    ImgItem it;
    if (!img_item(index, ntiles, q, flags, it)) return;
    const bool unit = INT || index[7] == 0u;
    const bool has_wide = index[1] != 0u;
    const int tw = g.tw, th = g.th, tpix = tw * th, ppix = g.pitch * th;
    for (int i = threadIdx.x; i < ppix; i += WG) {
        cnt32[i] = 0;
        if constexpr (!INT) acc64[i] = 0.0;
    }
    // (the first batch's barrier in img_records orders the zeroing before the first adds)
    auto one = [&](uint32_t word, uint32_t exact) {
        const int local = (int)(word & V2_LOCAL_MASK);
        const uint32_t bits = (word & V2_WIDE) ? exact : (word & V2_P_MASK);
        if constexpr (INT) {
            const int wv = (word & V2_WIDE) ? (int)bits : (int)bits >> (V2_LB + 1);
            __hip_atomic_fetch_add(cnt32 + local, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (unit) {
            __hip_atomic_fetch_add(cnt32 + local, (int)__uint_as_float(bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            lds_add(acc64 + local, __uint_as_float(bits));
        }
    };
    img_records<WG, IMG_U, RecN>(
        table, q, it, cseg,
        [&](uint32_t pos) -> RecN {
            RecN v;
            v.r = load_u2(rec + pos);
            v.w = has_wide ? load_u2(side + pos) : make_uint2(0u, 0u);
            return v;
        },
        [&](const RecN &v, uint32_t pos, uint32_t end) {
            one(v.r.x, v.w.x);
            if (pos + 1 < end) one(v.r.y, v.w.y);
        });
    __syncthreads();
    typedef typename std::conditional<INT, int, float>::type Out;
    Out *const out = static_cast<Out *>(out_);
    const int overwrite = flags & EVK_VOXEL_OVERWRITE;
    const int tx0 = (it.tile % g.tiles_x) * tw, ty0 = (it.tile / g.tiles_x) * th;
    auto lds_cell = [&](int c) -> Out {   // dense cell c of the tile -> padded LDS layout
        const int row = (int)div_magic((uint32_t)c, g.mx), col = c - row * tw, l = row * g.pitch + col;
        if constexpr (INT) return cnt32[l];
        else return unit ? (float)cnt32[l] : (float)acc64[l];
    };
    auto flush = [&](auto value_of) {
        for (int c = threadIdx.x; c < tpix; c += WG) {
            const int row = (int)div_magic((uint32_t)c, g.mx), col = c - row * tw;
            const int X = tx0 + col, Y = ty0 + row;
            if (X < g.dom_w && Y < g.dom_h) {
                Out *o = out + (int64_t)Y * g.dom_w + X;
                const Out v = value_of(c);
                *o = overwrite ? v : *o + v;
            }
        }
    };
    if (it.nparts == 1) {
        flush(lds_cell);
        return;
    }
    // a piece of a cut tile: partial tile to the staging area (agent-scope stores), the last piece to arrive sums them
    const int64_t stride = v2_staging_stride(tpix);
    Out *const staging = static_cast<Out *>(staging_);
    Out *mine = staging + (int64_t)it.item * stride;
    for (int c = threadIdx.x; c < tpix; c += WG) __hip_atomic_store(mine + c, lds_cell(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!img_last_part(index, ntiles, it)) return;
    const Out *parts = staging + (int64_t)it.first_item * stride;
    flush([&](int c) {
        Out sum = 0;
        for (uint32_t p = 0; p < it.nparts; ++p)
            sum += __hip_atomic_load(parts + (int64_t)p * stride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return sum;
    });
}

// ---- bilinear: a window one pixel wider and higher than the tile --------------------------------------------------
#define IMG_WIN_MAX 2048   // cells of the window: (tw + 2) * (th + 1) with (tw | 1) * th <= 1024
struct RecB {
    uint4 r;   // {x - x0, y - y0} of two records
    uint2 w;   // their weights
};
template <int WG>
__global__ void __launch_bounds__(WG) k_image_tiles_b(const uint2 *__restrict__ rec, const uint32_t *__restrict__ side,
                                                      const uint32_t *__restrict__ table, uint32_t *__restrict__ index,
                                                      TileGridG g, Part2 q, int flags, float *__restrict__ img,
                                                      float *__restrict__ staging) {
    // win: float64, or int64 multiples of 2^-30 (unit weights) / 2^-27 (MIXED: weights of ordinary size); win2 (MIXED): float64,
    // the events whose weight is not of ordinary size
    __shared__ acc_t win[IMG_WIN_MAX], win2[IMG_WIN_MAX];
    __shared__ uint2 cseg[WG / 64][IMG_CAP + 1];
    const int ntiles = g.tiles_x * g.tiles_y;
    ImgItem it;
    if (!img_item(index, ntiles, q, flags, it)) return;
    const bool has_side = index[7] != 0u;   // the call has weights other than +-1 / +0: their records point to the side runs
    const bool unit = !(flags & EVK_IMAGE2_NO_FIXED) && !has_side;
    const bool mixed = !(flags & EVK_IMAGE2_NO_FIXED) && has_side;
    const int tw = g.tw, th = g.th;
    const int ww = tw + 1, wh = th + 1, wpitch = ww | 1, wcells = wpitch * wh;   // odd pitch (evk_part.h)
    for (int i = threadIdx.x; i < wcells; i += WG) win[i] = 0.0, win2[i] = 0.0;
    unsigned long long *const winq = reinterpret_cast<unsigned long long *>(win);
    auto one = [&](auto mode_tag, uint32_t xb, uint32_t yb, uint32_t wside) {
        constexpr int MODE = decltype(mode_tag)::value;   // 0 float64, 1 unit weights (2^-30 steps), 2 MIXED
        constexpr bool UNIT = MODE == 1;
        // the weight's code in the two sign bits: 0 / 1 / 2 = +1.0 / -1.0 / +0.0, 3 = the side run's value
        const uint32_t code = (xb >> 31) | ((yb >> 31) << 1);
        const uint32_t wb = code == 3u ? wside : (code == 2u ? 0u : (0x3F800000u | (code << 31)));
        const float xr = __uint_as_float(xb & 0x7FFFFFFFu), yr = __uint_as_float(yb & 0x7FFFFFFFu);
        const float fx = floorf(xr), fy = floorf(yr);
        const float dx = xr - fx, dy = yr - fy;                  // = x - floor(x), y - floor(y) (image.py:81-82), exactly
        const float ax = 1.0f - dx, ay = 1.0f - dy;
        const int c0 = __mul24((int)fy, wpitch) + (int)fx;
        if constexpr (UNIT) {
            // image.py:111-114, the products in the order written there, on the weight times 2^30: a power of two commutes
            // with every rounding (|product| <= 1: no overflow, and nothing below 2^-126 matters at 2^-30 resolution)
            const float wv = __uint_as_float(wb) * IMG_FIX_ONE;
            const float wa = wv * ax, wd = wv * dx;
            auto addq = [&](int c, float v) {
                __hip_atomic_fetch_add(winq + c, (unsigned long long)(long long)__float2int_rn(v), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            addq(c0, wa * ay), addq(c0 + 1, wd * ay), addq(c0 + wpitch, wa * dy), addq(c0 + wpitch + 1, wd * dy);
        } else if constexpr (MODE == 2) {
            // (round 6) arbitrary float32 weights: 64-bit integer LDS atomics run at 4.1 lane-operations per clock and CU, float64
            // ones at 2.9.  A weight of ORDINARY size -- 2^-9 <= |w| <= 8, or 0 -- goes in as 2^-27 steps (|product| <= 8: no
            // overflow; a step is 2^-18 of the smallest such weight); any other (tiny, large, not finite) adds float64 values
            // to the second window -- wave-divergent only where the two kinds meet (with |w| <= 2 as the bound, N(0, 1) weights
            // had an outsider in 95 % of the waves: tile kernel 38.2 us instead of 45); the flush adds the windows
            const float wv = __uint_as_float(wb), aw = fabsf(wv);
            if ((aw <= 8.0f) & ((aw >= 0.001953125f) | (aw == 0.0f))) {
                const float ws = wv * (0.125f * IMG_FIX_ONE), wa = ws * ax, wd = ws * dx;
                auto addq = [&](int c, float v) {
                    __hip_atomic_fetch_add(winq + c, (unsigned long long)(long long)__float2int_rn(v), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                };
                addq(c0, wa * ay), addq(c0 + 1, wd * ay), addq(c0 + wpitch, wa * dy), addq(c0 + wpitch + 1, wd * dy);
            } else {
                const float wa = wv * ax, wd = wv * dx;
                lds_add(win2 + c0, wa * ay), lds_add(win2 + c0 + 1, wd * ay), lds_add(win2 + c0 + wpitch, wa * dy),
                    lds_add(win2 + c0 + wpitch + 1, wd * dy);
            }
        } else {
            const float wv = __uint_as_float(wb);
            const float wa = wv * ax, wd = wv * dx;
            lds_add(win + c0, wa * ay), lds_add(win + c0 + 1, wd * ay), lds_add(win + c0 + wpitch, wa * dy),
                lds_add(win + c0 + wpitch + 1, wd * dy);
        }
    };
    auto run = [&](auto unit_tag) {
        img_records<WG, IMG_U, RecB>(
            table, q, it, cseg,
            [&](uint32_t pos) -> RecB {
                RecB v;
                v.r = load_u4(rec + pos);
                v.w = has_side ? load_u2(side + pos) : make_uint2(0u, 0u);
                return v;
            },
            [&](const RecB &v, uint32_t pos, uint32_t end) {
                one(unit_tag, v.r.x, v.r.y, v.w.x);
                if (pos + 1 < end) one(unit_tag, v.r.z, v.r.w, v.w.y);
            });
    };
    if (unit) run(std::integral_constant<int, 1>{});
    else if (mixed) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 0>{});
    __syncthreads();
    const int tx0 = (it.tile % g.tiles_x) * tw, ty0 = (it.tile / g.tiles_x) * th;
    const int dcells = ww * wh;   // dense cells of the window
    const uint32_t mw = magic_div((uint32_t)ww);
    auto lds_cell = [&](int c) -> float {
        const int row = (int)div_magic((uint32_t)c, mw), col = c - row * ww, l = row * wpitch + col;
        if (unit) return (float)((double)(long long)winq[l] * (1.0 / (double)IMG_FIX_ONE));
        if (mixed) return (float)((double)(long long)winq[l] * (8.0 / (double)IMG_FIX_ONE) + win2[l]);
        return (float)win[l];
    };
    // Interior pixels belong to this window alone: plain read-modify-write.  The ring (first / last row and column) is also
    // covered by the neighbouring tiles' windows: global float atomics.
    auto flush = [&](auto value_of) {
        for (int c = threadIdx.x; c < dcells; c += WG) {
            const int row = (int)div_magic((uint32_t)c, mw), col = c - row * ww;
            const int X = tx0 + col, Y = ty0 + row;
            if (X < g.dom_w && Y < g.dom_h) {
                float *o = img + (int64_t)Y * g.dom_w + X;
                const float v = value_of(c);
                if (row == 0 || row == th || col == 0 || col == tw) {
                    if (v != 0.0f || v != v) atomic_add(o, v);
                } else {
                    *o += v;
                }
            }
        }
    };
    if (it.nparts == 1) {
        flush(lds_cell);
        return;
    }
    const int64_t stride = v2_staging_stride(dcells);
    float *mine = staging + (int64_t)it.item * stride;
    for (int c = threadIdx.x; c < dcells; c += WG) __hip_atomic_store(mine + c, lds_cell(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!img_last_part(index, ntiles, it)) return;
    const float *parts = staging + (int64_t)it.first_item * stride;
    flush([&](int c) {
        float sum = 0.0f;
        for (uint32_t p = 0; p < it.nparts; ++p)
            sum += __hip_atomic_load(parts + (int64_t)p * stride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return sum;
    });
}

// ---- average-timestamp images: four windows per tile (time and count of the positive / of the non-positive events) ----
// The bilinear kernel with the record's two sign bits read as the event's class and the side run as its normalised time
// stamp: four float64 LDS atomics into the class's time window (nts (1 - dx) (1 - dy) ..., float32 products in the reference's
// order, image.py:111-114) and four fixed-point ones (2^-30 steps, as k_image_tiles_b's unit weights) into its count window.
template <int WG>
__global__ void __launch_bounds__(WG) k_image_tiles_ts(const uint2 *__restrict__ rec, const uint32_t *__restrict__ side,
                                                       const uint32_t *__restrict__ table, uint32_t *__restrict__ index,
                                                       TileGridG g, Part2 q, int flags, float *__restrict__ out4,
                                                       float *__restrict__ staging) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ts_smem[];
    // [4][wcells]: time+ | count+ | time- | count-; fixed-point mode: [4] and [5] take the time of events with |nts| > 1 (float64)
    acc_t *const win = reinterpret_cast<acc_t *>(ts_smem);
    __shared__ uint2 cseg[WG / 64][IMG_CAP + 1];
    const int ntiles = g.tiles_x * g.tiles_y;
    ImgItem it;
    if (!img_item(index, ntiles, q, flags, it)) return;
    const bool fixed = !(flags & EVK_IMAGE2_NO_FIXED);
    const int tw = g.tw, th = g.th;
    const int ww = tw + 1, wh = th + 1, wpitch = ww | 1, wcells = wpitch * wh;
    for (int i = threadIdx.x; i < (fixed ? 6 : 4) * wcells; i += WG) win[i] = 0.0;
    unsigned long long *const winq = reinterpret_cast<unsigned long long *>(win);
    auto one = [&](auto fixed_tag, uint32_t xb, uint32_t yb, uint32_t wside) {
        constexpr bool FIXED = decltype(fixed_tag)::value;
        const uint32_t k = (xb >> 31) | ((yb >> 31) << 1);
        if (k >= 2u) return;   // a NaN polarity: in neither class
        const float xr = __uint_as_float(xb & 0x7FFFFFFFu), yr = __uint_as_float(yb & 0x7FFFFFFFu);
        const float fx = floorf(xr), fy = floorf(yr);
        const float dx = xr - fx, dy = yr - fy;
        const float ax = 1.0f - dx, ay = 1.0f - dy;
        const int c0 = __mul24((int)fy, wpitch) + (int)fx;
        acc_t *const val = win + __mul24((int)(2u * k), wcells);
        const float nts = __uint_as_float(wside);
        if constexpr (FIXED) {
            // 64-bit integer LDS atomics run at 4.1 lane-operations per clock and CU, float64 ones at 2.9: the
            // normalised time of a SORTED stream lies in [0, 1], so its four products go in as 2^-30 steps like the count's (a
            // power of two commutes with every rounding); an event beyond that -- unsorted time stamps, mode 2 with large t, NaN
            // -- adds float64 values to a window of its own (wave-divergent only there)
            auto addq = [&](unsigned long long *base, int c, float v) {
                __hip_atomic_fetch_add(base + c, (unsigned long long)(long long)__float2int_rn(v), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            if (fabsf(nts) <= 1.0f) {
                unsigned long long *const tq = winq + __mul24((int)(2u * k), wcells);
                const float ns = nts * IMG_FIX_ONE, ta_ = ns * ax, td_ = ns * dx;
                addq(tq, c0, ta_ * ay), addq(tq, c0 + 1, td_ * ay), addq(tq, c0 + wpitch, ta_ * dy), addq(tq, c0 + wpitch + 1, td_ * dy);
            } else {
                acc_t *const esc = win + __mul24((int)(4u + k), wcells);
                const float ta_ = nts * ax, td_ = nts * dx;
                lds_add(esc + c0, ta_ * ay), lds_add(esc + c0 + 1, td_ * ay), lds_add(esc + c0 + wpitch, ta_ * dy),
                    lds_add(esc + c0 + wpitch + 1, td_ * dy);
            }
            unsigned long long *const cnt = winq + __mul24((int)(2u * k + 1u), wcells);
            const float wa = IMG_FIX_ONE * ax, wd = IMG_FIX_ONE * dx;   // the weight is 1.0: 1.0f * ax == ax
            addq(cnt, c0, wa * ay), addq(cnt, c0 + 1, wd * ay), addq(cnt, c0 + wpitch, wa * dy), addq(cnt, c0 + wpitch + 1, wd * dy);
        } else {
            const float ta_ = nts * ax, td_ = nts * dx;
            lds_add(val + c0, ta_ * ay), lds_add(val + c0 + 1, td_ * ay), lds_add(val + c0 + wpitch, ta_ * dy),
                lds_add(val + c0 + wpitch + 1, td_ * dy);
            acc_t *const cnt = val + wcells;
            lds_add(cnt + c0, ax * ay), lds_add(cnt + c0 + 1, dx * ay), lds_add(cnt + c0 + wpitch, ax * dy),
                lds_add(cnt + c0 + wpitch + 1, dx * dy);
        }
    };
    auto run = [&](auto fixed_tag) {
        img_records<WG, IMG_U, RecB>(
            table, q, it, cseg,
            [&](uint32_t pos) -> RecB {
                RecB v;
                v.r = load_u4(rec + pos);
                v.w = load_u2(side + pos);
                return v;
            },
            [&](const RecB &v, uint32_t pos, uint32_t end) {
                one(fixed_tag, v.r.x, v.r.y, v.w.x);
                if (pos + 1 < end) one(fixed_tag, v.r.z, v.r.w, v.w.y);
            });
    };
    if (fixed) run(std::true_type{});
    else run(std::false_type{});
    __syncthreads();
    const int tx0 = (it.tile % g.tiles_x) * tw, ty0 = (it.tile / g.tiles_x) * th;
    const int dcells = ww * wh;   // dense cells of one window
    const uint32_t mw = magic_div((uint32_t)ww);
    const int64_t plane = (int64_t)g.dom_h * g.dom_w;
    auto lds_cell = [&](int pl, int c) -> float {
        const int row = (int)div_magic((uint32_t)c, mw), col = c - row * ww, l = pl * wcells + row * wpitch + col;
        if (fixed) {   // count: integer steps; time: integer steps + the float64 window of the events beyond [-1, 1]
            const double q30 = (double)(long long)winq[l] * (1.0 / (double)IMG_FIX_ONE);
            return (float)((pl & 1) ? q30 : q30 + win[l + (4 - pl / 2) * wcells]);     // time plane 0 -> window 4, 2 -> 5
        }
        return (float)win[l];
    };
    // interior pixels belong to this window alone (plain read-modify-write), the ring is shared with the neighbours' windows
    auto flush = [&](auto value_of) {
        for (int i = threadIdx.x; i < 4 * dcells; i += WG) {
            const int pl = i / dcells, c = i - pl * dcells;
            const int row = (int)div_magic((uint32_t)c, mw), col = c - row * ww;
            const int X = tx0 + col, Y = ty0 + row;
            if (X < g.dom_w && Y < g.dom_h) {
                float *o = out4 + pl * plane + (int64_t)Y * g.dom_w + X;
                const float v = value_of(pl, c);
                if (row == 0 || row == th || col == 0 || col == tw) {
                    if (v != 0.0f || v != v) atomic_add(o, v);
                } else {
                    *o += v;
                }
            }
        }
    };
    if (it.nparts == 1) {
        flush(lds_cell);
        return;
    }
    const int64_t stride = v2_staging_stride(4 * dcells);
    float *mine = staging + (int64_t)it.item * stride;
    for (int i = threadIdx.x; i < 4 * dcells; i += WG) {
        const int pl = i / dcells;
        __hip_atomic_store(mine + i, lds_cell(pl, i - pl * dcells), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!img_last_part(index, ntiles, it)) return;
    const float *parts = staging + (int64_t)it.first_item * stride;
    flush([&](int pl, int c) {
        float sum = 0.0f;
        for (uint32_t pp = 0; pp < it.nparts; ++pp)
            sum += __hip_atomic_load(parts + (int64_t)pp * stride + pl * dcells + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return sum;
    });
}

// ---- derivative splats: the image of warped events and / or its two derivative planes (round 6) ------------------------------
// Per record the event's weights come from the caller's columns by the index the side run holds (dependent loads: the record's
// four or five weights do not fit a sub-chunk's LDS); float64 LDS atomics; the products and sums of every corner in float32
// exactly as the direct kernels write them (image.py:111-114, 131-135).
template <int WG, typename WF>
__global__ void __launch_bounds__(WG) k_image_tiles_drv(const uint2 *__restrict__ rec, const uint32_t *__restrict__ side,
                                                        const uint32_t *__restrict__ table, uint32_t *__restrict__ index,
                                                        TileGridG g, Part2 q, int flags, const WF wf, int grad,
                                                        float *__restrict__ img, float *__restrict__ dimg, float *__restrict__ staging) {
    extern __shared__ __attribute__((aligned(16))) unsigned char drv_smem[];
    acc_t *const win = reinterpret_cast<acc_t *>(drv_smem);   // [planes][wcells]: (image |) derivative 0 | derivative 1
    __shared__ uint2 cseg[WG / 64][IMG_CAP + 1];
    const int ntiles = g.tiles_x * g.tiles_y;
    ImgItem it;
    if (!img_item(index, ntiles, q, flags, it)) return;
    constexpr int P0 = WF::IWE ? 1 : 0;
    const int planes = P0 + (grad ? 2 : 0);
    const int tw = g.tw, th = g.th;
    const int ww = tw + 1, wh = th + 1, wpitch = ww | 1, wcells = wpitch * wh;
    for (int i = threadIdx.x; i < planes * wcells; i += WG) win[i] = 0.0;
    auto one = [&](uint32_t xb, uint32_t yb, uint32_t idx) {
        const float xr = __uint_as_float(xb), yr = __uint_as_float(yb);
        const float fx = floorf(xr), fy = floorf(yr);
        const float dx = xr - fx, dy = yr - fy;
        const float ax = 1.0f - dx, ay = 1.0f - dy;
        const int c0 = __mul24((int)fy, wpitch) + (int)fx;
        float mp, a[2], b[2];
        wf.get(idx, grad != 0, mp, a, b);
        if constexpr (WF::IWE) {
            const float wa = mp * ax, wd = mp * dx;
            lds_add(win + c0, wa * ay), lds_add(win + c0 + 1, wd * ay), lds_add(win + c0 + wpitch, wa * dy),
                lds_add(win + c0 + wpitch + 1, wd * dy);
        }
        if (grad) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc_t *const d = win + __mul24(P0 + c, wcells) + c0;
                lds_add(d, a[c] * (-ay) + b[c] * (-ax)), lds_add(d + 1, a[c] * ay + b[c] * (-dx)),
                    lds_add(d + wpitch, a[c] * (-dy) + b[c] * ax), lds_add(d + wpitch + 1, a[c] * dy + b[c] * dx);
            }
        }
    };
    img_records<WG, IMG_U, RecB>(
        table, q, it, cseg,
        [&](uint32_t pos) -> RecB {
            RecB v;
            v.r = load_u4(rec + pos);
            v.w = load_u2(side + pos);
            return v;
        },
        [&](const RecB &v, uint32_t pos, uint32_t end) {
            one(v.r.x, v.r.y, v.w.x);
            if (pos + 1 < end) one(v.r.z, v.r.w, v.w.y);
        });
    __syncthreads();
    const int tx0 = (it.tile % g.tiles_x) * tw, ty0 = (it.tile / g.tiles_x) * th;
    const int dcells = ww * wh;
    const uint32_t mw = magic_div((uint32_t)ww);
    const int64_t plane = (int64_t)g.dom_h * g.dom_w;
    auto lds_cell = [&](int pl, int c) -> float {
        const int row = (int)div_magic((uint32_t)c, mw), col = c - row * ww;
        return (float)win[pl * wcells + row * wpitch + col];
    };
    auto flush = [&](auto value_of) {
        for (int i = threadIdx.x; i < planes * dcells; i += WG) {
            const int pl = i / dcells, c = i - pl * dcells;
            const int row = (int)div_magic((uint32_t)c, mw), col = c - row * ww;
            const int X = tx0 + col, Y = ty0 + row;
            if (X < g.dom_w && Y < g.dom_h) {
                float *o = ((WF::IWE && pl == 0) ? img : dimg + (pl - P0) * plane) + (int64_t)Y * g.dom_w + X;
                const float v = value_of(pl, c);
                if (row == 0 || row == th || col == 0 || col == tw) {
                    if (v != 0.0f || v != v) atomic_add(o, v);
                } else {
                    *o += v;
                }
            }
        }
    };
    if (it.nparts == 1) {
        flush(lds_cell);
        return;
    }
    const int64_t stride = v2_staging_stride(3 * dcells);
    float *mine = staging + (int64_t)it.item * stride;
    for (int i = threadIdx.x; i < planes * dcells; i += WG) {
        const int pl = i / dcells;
        __hip_atomic_store(mine + i, lds_cell(pl, i - pl * dcells), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!img_last_part(index, ntiles, it)) return;
    const float *parts = staging + (int64_t)it.first_item * stride;
    flush([&](int pl, int c) {
        float sum = 0.0f;
        for (uint32_t pp = 0; pp < it.nparts; ++pp)
            sum += __hip_atomic_load(parts + (int64_t)pp * stride + pl * dcells + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return sum;
    });
}

// ---- host side ------------------------------------------------------------------------------------------------------
// One geometry for the images: 1024 threads x 8 events, sub-chunks of 8 K events (68 KB of LDS with 4-byte records,
// 100 KB with the 12 bytes of the bilinear format).
static inline int img_window_cells(int tw, int th) { return (tw + 1) * (th + 1); }

// sub-chunks of 8 K events, or -- more than 680 tiles (720p and up): longer segments for the tile kernel -- 12 K, when the
// bilinear format's 12 bytes of LDS per event and the tile counters still fit
static inline bool img_small(int ntiles) {
    return ntiles <= 680 || v2_part_lds(1024, 12, V2_FMT_IMGB, ntiles) > (size_t)V2_LDS_LIMIT;
}
struct ImgCall {
    TileGridG g;
    Part2 q;
    V2Layout L;
    int ntiles;
    bool small;   // 8 K-event sub-chunks (1024 x 8)
};
#define IMG_TS_PLANES 6   // staging of a cut tile's piece, in tw x th floats x 2: four (tw + 1) x (th + 1) float windows fit
static int img_setup(ImgCall &ic, int64_t n, int h, int wd, int tile_w, int tile_h, int flags, const void *out, uint32_t *index,
                     void *scratch, int64_t scratch_bytes, uint32_t *host_report, int planes = 2, bool small_only = false) {
    const int known = EVK_VOXEL_OVERWRITE | EVK_VOXEL2_PARTITION_ONLY | EVK_VOXEL2_TILES_ONLY | EVK_VOXEL2_NO_XCD_ORDER |
                      EVK_IMAGE2_NO_FIXED;
    if (make_grid_g(ic.g, h, wd, tile_w, tile_h) != EVK_OK || !out || !index || !scratch || n <= 0 ||
        n > (int64_t)4000000000LL || (flags & ~known))
        return EVK_EINVAL;
    if (host_report && ((uintptr_t)host_report & 7u)) return EVK_EALIGN;
    ic.ntiles = ic.g.tiles_x * ic.g.tiles_y;
    if (ic.ntiles > evk_voxel2_max_tiles() || !evk_voxel2_num_tiles(h, wd, tile_w, tile_h) ||
        (tile_w + 2) * (tile_h + 1) > IMG_WIN_MAX)
        return EVK_EINVAL;
    // (small_only: the column sources that load 24 words per four events -- int64 pixels -- spill in the 12-event geometry)
    const bool small = small_only || img_small(ic.ntiles);
    ic.small = small;
    ic.L = v2_layout(ic.ntiles, n, planes, tile_w, tile_h, small);   // (2 planes of tw x th floats hold a (tw + 1) x (th + 1) window)
    if (scratch_bytes < ic.L.total) return EVK_ESCRATCH;
    if (!aligned16(scratch)) return EVK_EALIGN;
    ic.q = v2_geometry(n, ic.ntiles, small);
    return EVK_OK;
}

template <int FMT, typename C>
static void img_partition(const C &c, int64_t n, const ImgCall &ic, uint32_t *index, void *scratch, uint32_t *oob,
                          uint32_t *host_report, uint32_t seq, hipStream_t s, int t_from_events = 0) {
    char *sb = (char *)scratch;
    if constexpr (FMT == V2_FMT_IMGX || FMT == V2_FMT_IMGD) {   // (small_only callers: no 12-event instantiation of these)
        launch_part<1024, 8, FMT>(c, n, ic.g, ic.ntiles, ic.q, 0.0f, 0.0f, 0.0f, t_from_events, sb + ic.L.rec, sb + ic.L.pw,
                                  (uint32_t *)(sb + ic.L.bases), (uint32_t *)(sb + ic.L.table), index, oob, host_report, seq, s);
    } else if (ic.small) {
        launch_part<1024, 8, FMT>(c, n, ic.g, ic.ntiles, ic.q, 0.0f, 0.0f, 0.0f, t_from_events, sb + ic.L.rec, sb + ic.L.pw,
                                  (uint32_t *)(sb + ic.L.bases), (uint32_t *)(sb + ic.L.table), index, oob, host_report, seq, s);
    } else {
        launch_part<1024, 12, FMT>(c, n, ic.g, ic.ntiles, ic.q, 0.0f, 0.0f, 0.0f, t_from_events, sb + ic.L.rec, sb + ic.L.pw,
                                   (uint32_t *)(sb + ic.L.bases), (uint32_t *)(sb + ic.L.table), index, oob, host_report, seq, s);
    }
}

}  // namespace evk

using namespace evk;

extern "C" int64_t evk_image2_scratch_bytes(int ntiles, int64_t n, int tile_w, int tile_h) {
    if (ntiles <= 0 || n < 0 || tile_w <= 0 || tile_h <= 0) return 0;
    return v2_layout(ntiles, n, 2, tile_w, tile_h, img_small(ntiles)).total;
}

extern "C" int evk_image2_nearest_i32(const int32_t *x, const int32_t *y, const int32_t *w, int64_t n, int canvas_h,
                                      int canvas_w, int tile_w, int tile_h, int flags, int32_t *canvas, uint32_t *index,
                                      void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report,
                                      uint32_t seq, void *stream) {
    if (n > 0 && (!x || !y)) return EVK_EINVAL;
    if (!(aligned16(x) && aligned16(y) && aligned16(w))) return EVK_EALIGN;
    ImgCall ic;
    const int rc = img_setup(ic, n, canvas_h, canvas_w, tile_w, tile_h, flags, canvas, index, scratch, scratch_bytes, host_report);
    if (rc != EVK_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    char *sb = (char *)scratch;
    if (!(flags & EVK_VOXEL2_TILES_ONLY)) img_partition<V2_FMT_IMGN>(SrcImgI32{x, y, w}, n, ic, index, scratch, oob, host_report, seq, s);
    if (!(flags & EVK_VOXEL2_PARTITION_ONLY))
        k_image_tiles_n<IMG_WG, true><<<v2_max_items(n, ic.ntiles), IMG_WG, 0, s>>>(
            (const uint32_t *)(sb + ic.L.rec), (const uint32_t *)(sb + ic.L.pw), (const uint32_t *)(sb + ic.L.table), index, ic.g,
            ic.q, flags, canvas, sb + ic.L.staging);
    return launch_status();
}

extern "C" int evk_image2_nearest_f32(const float *x, const float *y, const float *w, int64_t n, int h, int wd, float clipx,
                                      float clipy, int tile_w, int tile_h, int flags, float *img, uint32_t *index,
                                      void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report,
                                      uint32_t seq, void *stream) {
    if (n > 0 && (!x || !y || !w)) return EVK_EINVAL;
    if (!(column_ok(x, flags) && column_ok(y, flags) && column_ok(w, flags))) return EVK_EALIGN;
    flags &= ~EVK_COLUMNS_UNALIGNED;
    ImgCall ic;
    const int rc = img_setup(ic, n, h, wd, tile_w, tile_h, flags, img, index, scratch, scratch_bytes, host_report);
    if (rc != EVK_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    char *sb = (char *)scratch;
    if (!(flags & EVK_VOXEL2_TILES_ONLY))
        img_partition<V2_FMT_IMGN>(SrcImgF32{x, y, w, clipx, clipy, img, h, wd}, n, ic, index, scratch, oob, host_report, seq, s);
    if (!(flags & EVK_VOXEL2_PARTITION_ONLY))
        k_image_tiles_n<IMG_WG, false><<<v2_max_items(n, ic.ntiles), IMG_WG, 0, s>>>(
            (const uint32_t *)(sb + ic.L.rec), (const uint32_t *)(sb + ic.L.pw), (const uint32_t *)(sb + ic.L.table), index, ic.g,
            ic.q, flags, img, sb + ic.L.staging);
    return launch_status();
}

extern "C" int evk_image2_bilinear_f32(const float *x, const float *y, const float *w, int64_t n, int h, int wd, float clipx,
                                       float clipy, int tile_w, int tile_h, int flags, float *img, uint32_t *index,
                                       void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report,
                                       uint32_t seq, void *stream) {
    if (n > 0 && (!x || !y || !w)) return EVK_EINVAL;
    if (!(column_ok(x, flags) && column_ok(y, flags) && column_ok(w, flags))) return EVK_EALIGN;
    flags &= ~EVK_COLUMNS_UNALIGNED;
    if (flags & EVK_VOXEL_OVERWRITE) return EVK_EINVAL;   // the ring of a window is ADDED to the image: always accumulates
    ImgCall ic;
    const int rc = img_setup(ic, n, h, wd, tile_w, tile_h, flags, img, index, scratch, scratch_bytes, host_report);
    if (rc != EVK_OK) return rc;
    if (h < 2 || wd < 2) return EVK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    char *sb = (char *)scratch;
    if (!(flags & EVK_VOXEL2_TILES_ONLY))
        img_partition<V2_FMT_IMGB>(SrcImgF32{x, y, w, clipx, clipy, img, h, wd}, n, ic, index, scratch, oob, host_report, seq, s);
    if (!(flags & EVK_VOXEL2_PARTITION_ONLY))
        k_image_tiles_b<IMG_WG><<<v2_max_items(n, ic.ntiles), IMG_WG, 0, s>>>(
            (const uint2 *)(sb + ic.L.rec), (const uint32_t *)(sb + ic.L.pw), (const uint32_t *)(sb + ic.L.table), index, ic.g,
            ic.q, flags, img, (float *)(sb + ic.L.staging));
    return launch_status();
}

/* scratch of evk_image2_splat_indexed_f32, evk_image2_splat_drv_indexed_f32 and evk_image2_drv_f64 (8 K-event sub-chunks whatever
 * the tile count) */
extern "C" int64_t evk_image2_indexed_scratch_bytes(int ntiles, int64_t n, int tile_w, int tile_h) {
    if (ntiles <= 0 || n < 0 || tile_w <= 0 || tile_h <= 0) return 0;
    return v2_layout(ntiles, n, IMG_TS_PLANES, tile_w, tile_h, true).total;
}

extern "C" int64_t evk_timestamp_images2_scratch_bytes(int ntiles, int64_t n, int tile_w, int tile_h) {
    if (ntiles <= 0 || n < 0 || tile_w <= 0 || tile_h <= 0) return 0;
    return v2_layout(ntiles, n, IMG_TS_PLANES, tile_w, tile_h, img_small(ntiles)).total;
}

extern "C" int evk_timestamp_images2_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int h, int wd,
                                         float clipx, float clipy, int mode, float ta, float td, int tile_w, int tile_h,
                                         int flags, float *out4, uint32_t *index, void *scratch, int64_t scratch_bytes,
                                         uint32_t *oob, uint32_t *host_report, uint32_t seq, void *stream) {
    if (n > 0 && (!x || !y || !t || !p)) return EVK_EINVAL;
    if (mode < 0 || mode > 2 || (flags & EVK_VOXEL_OVERWRITE)) return EVK_EINVAL;   // (the windows are ADDED to the images)
    const int from_events = (flags & EVK_VOXEL_T_FROM_EVENTS) ? 1 : 0;
    flags &= ~EVK_VOXEL_T_FROM_EVENTS;
    if (!(column_ok(x, flags) && column_ok(y, flags) && column_ok(t, flags) && column_ok(p, flags))) return EVK_EALIGN;
    flags &= ~EVK_COLUMNS_UNALIGNED;
    ImgCall ic;
    const int rc = img_setup(ic, n, h, wd, tile_w, tile_h, flags, out4, index, scratch, scratch_bytes, host_report, IMG_TS_PLANES);
    if (rc != EVK_OK) return rc;
    if (h < 2 || wd < 2) return EVK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    char *sb = (char *)scratch;
    if (!(flags & EVK_VOXEL2_TILES_ONLY))
        img_partition<V2_FMT_IMGT>(SrcTsF32{x, y, t, p, clipx, clipy, mode, ta, td, out4, h, wd}, n, ic, index, scratch, oob,
                                   host_report, seq, s, from_events);
    if (!(flags & EVK_VOXEL2_PARTITION_ONLY)) {
        const int wcells = ((tile_w + 1) | 1) * (tile_h + 1);
        const size_t lds = (size_t)((flags & EVK_IMAGE2_NO_FIXED) ? 4 : 6) * wcells * sizeof(acc_t);
        static std::once_flag once[64];   // per device: the attribute belongs to the loaded code object
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::call_once(once[dev & 63], [] {
            (void)hipFuncSetAttribute((const void *)k_image_tiles_ts<IMG_WG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      6 * IMG_WIN_MAX * (int)sizeof(acc_t));
        });
        k_image_tiles_ts<IMG_WG><<<v2_max_items(n, ic.ntiles), IMG_WG, lds, s>>>(
            (const uint2 *)(sb + ic.L.rec), (const uint32_t *)(sb + ic.L.pw), (const uint32_t *)(sb + ic.L.table), index, ic.g,
            ic.q, flags, out4, (float *)(sb + ic.L.staging));
    }
    return launch_status();
}

/* interpolate_to_image (image.py:102-115) on caller-computed pixels and fractions, on the one-pass design: evk_splat_indexed_f32's
 * arguments and semantics (negative pixels wrap once, anything else outside raises: *oob), the rest as evk_image2_bilinear_f32 */
extern "C" int evk_image2_splat_indexed_f32(const int64_t *px, const int64_t *py, const float *dx, const float *dy, const float *w,
                                            int64_t n, int h, int wd, int tile_w, int tile_h, int flags, float *img,
                                            uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                                            uint32_t *host_report, uint32_t seq, void *stream) {
    if (n > 0 && (!px || !py || !dx || !dy || !w)) return EVK_EINVAL;
    if (!(column_ok(px, flags, 8) && column_ok(py, flags, 8) && column_ok(dx, flags) && column_ok(dy, flags) && column_ok(w, flags)))
        return EVK_EALIGN;
    flags &= ~EVK_COLUMNS_UNALIGNED;
    if (flags & EVK_VOXEL_OVERWRITE) return EVK_EINVAL;
    ImgCall ic;
    const int rc = img_setup(ic, n, h, wd, tile_w, tile_h, flags, img, index, scratch, scratch_bytes, host_report, IMG_TS_PLANES, true);
    if (rc != EVK_OK) return rc;
    if (h < 2 || wd < 2) return EVK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    char *sb = (char *)scratch;
    if (!(flags & EVK_VOXEL2_TILES_ONLY))
        img_partition<V2_FMT_IMGX>(SrcIdxF32{(const long long *)px, (const long long *)py, dx, dy, w, img, h, wd}, n, ic, index, scratch,
                                   oob, host_report, seq, s);
    if (!(flags & EVK_VOXEL2_PARTITION_ONLY))
        k_image_tiles_b<IMG_WG><<<v2_max_items(n, ic.ntiles), IMG_WG, 0, s>>>(
            (const uint2 *)(sb + ic.L.rec), (const uint32_t *)(sb + ic.L.pw), (const uint32_t *)(sb + ic.L.table), index, ic.g,
            ic.q, flags, img, (float *)(sb + ic.L.staging));
    return launch_status();
}

template <typename C, typename WF>
static int drv_call(const C &c, const WF &wf, int grad, int64_t n, int h, int wd, int tile_w, int tile_h, int flags, float *img,
                    float *dimg, uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report,
                    uint32_t seq, void *stream) {
    if (flags & EVK_VOXEL_OVERWRITE) return EVK_EINVAL;   // (the windows are ADDED to the images)
    ImgCall ic;
    const int rc = img_setup(ic, n, h, wd, tile_w, tile_h, flags, WF::IWE ? img : dimg, index, scratch, scratch_bytes, host_report,
                             IMG_TS_PLANES, true);
    if (rc != EVK_OK) return rc;
    if (h < 2 || wd < 2) return EVK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    char *sb = (char *)scratch;
    if (!(flags & EVK_VOXEL2_TILES_ONLY)) img_partition<V2_FMT_IMGD>(c, n, ic, index, scratch, oob, host_report, seq, s);
    if (!(flags & EVK_VOXEL2_PARTITION_ONLY)) {
        const int wcells = ((tile_w + 1) | 1) * (tile_h + 1);
        const size_t lds = (size_t)((WF::IWE ? 1 : 0) + (grad ? 2 : 0)) * wcells * sizeof(acc_t);
        static std::once_flag once[64];   // per device and instantiation
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::call_once(once[dev & 63], [] {
            (void)hipFuncSetAttribute((const void *)k_image_tiles_drv<IMG_WG, WF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      3 * IMG_WIN_MAX * (int)sizeof(acc_t));
        });
        k_image_tiles_drv<IMG_WG, WF><<<v2_max_items(n, ic.ntiles), IMG_WG, lds, s>>>(
            (const uint2 *)(sb + ic.L.rec), (const uint32_t *)(sb + ic.L.pw), (const uint32_t *)(sb + ic.L.table), index, ic.g,
            ic.q, flags, wf, grad, img, dimg, (float *)(sb + ic.L.staging));
    }
    return launch_status();
}

/* interpolate_to_derivative_img (image.py:117-136), two channels, on the one-pass design: evk_splat_drv_indexed_f32's arguments
 * (C = 2) and semantics, the rest as evk_image2_splat_indexed_f32; scratch: evk_timestamp_images2_scratch_bytes */
extern "C" int evk_image2_splat_drv_indexed_f32(const int64_t *px, const int64_t *py, const float *dx, const float *dy,
                                                const float *w1, const float *w2, int64_t n, int h, int wd, int tile_w, int tile_h,
                                                int flags, float *d_img, uint32_t *index, void *scratch, int64_t scratch_bytes,
                                                uint32_t *oob, uint32_t *host_report, uint32_t seq, void *stream) {
    if (n > 0 && (!px || !py || !dx || !dy || !w1 || !w2)) return EVK_EINVAL;
    if (n > (int64_t)4000000000LL || !d_img) return EVK_EINVAL;
    if (!(column_ok(px, flags, 8) && column_ok(py, flags, 8) && column_ok(dx, flags) && column_ok(dy, flags))) return EVK_EALIGN;
    flags &= ~EVK_COLUMNS_UNALIGNED;
    const SrcIdxDrvF32 c{(const long long *)px, (const long long *)py, dx, dy, w1, w2, n, d_img, h, wd};
    return drv_call(c, WgtIdxDrvF32{w1, w2, n}, 1, n, h, wd, tile_w, tile_h, flags, nullptr, d_img, index, scratch, scratch_bytes, oob,
                    host_report, seq, stream);
}

/* events_to_image_drv (image.py:162-217) on the one-pass design: evk_image_drv_f64's arguments and semantics (jx, jy NULL: the
 * image alone), the rest as evk_image2_bilinear_f32; columns 16-byte aligned; scratch: evk_timestamp_images2_scratch_bytes */
extern "C" int evk_image2_drv_f64(const double *x, const double *y, const double *p, const double *jx, const double *jy, int64_t n,
                                  int h, int wd, float clipx, float clipy, int tile_w, int tile_h, int flags, float *img,
                                  float *d_img, uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                                  uint32_t *host_report, uint32_t seq, void *stream) {
    if (n > 0 && (!x || !y || !p)) return EVK_EINVAL;
    if ((jx == nullptr) != (jy == nullptr) || (jx && !d_img) || !img || n > (int64_t)4000000000LL) return EVK_EINVAL;
    if (!(aligned16(x) && aligned16(y))) return EVK_EALIGN;
    const SrcDrvF64 c{x, y, p, jx, jy, n, clipx, clipy, img, d_img, h, wd};
    return drv_call(c, WgtDrvF64{p, jx, jy, n}, jx ? 1 : 0, n, h, wd, tile_w, tile_h, flags, img, d_img, index, scratch,
                    scratch_bytes, oob, host_report, seq, stream);
}
