// Direct (global-atomic) event scatter kernels: one streaming pass over the SoA event columns, every contribution
// is one hardware global atomic add.  These are the always-correct baseline path; the LDS-tiled kernels in
// evk_tiled.hip replace them on the hot configurations.
#include "evk_splat.h"

namespace evk {

// Visit events as aligned quads (one 16 B load per lane per f32 column) when VEC, else one event per lane.
// f(base, cnt, vec): events [base, base+cnt); vec => base % 4 == 0, cnt == 4 and columns are 16 B aligned.
template <bool VEC, typename F>
__device__ __forceinline__ void foreach_events(int64_t n, F f) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const int64_t nq = n >> 2;
        for (int64_t q = tid; q < nq; q += stride) f(q << 2, 4, true);
        if (tid == 0 && (n & 3)) f(nq << 2, (int)(n & 3), false);
    } else {
        for (int64_t i = tid; i < n; i += stride) f(i, 1, false);
    }
}

template <typename T>
__device__ __forceinline__ Vec4<T> load_col(const T *p, int64_t base, int cnt, bool vec) {
    if (vec) return load4(p, base >> 2);
    Vec4<T> r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v[k] = (k < cnt) ? p[base + k] : T(0);
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// nearest-pixel images
// ---------------------------------------------------------------------------------------------------------

template <bool VEC, typename TW>
__global__ void __launch_bounds__(EVK_BLOCK) k_image_nearest_int(const int32_t *__restrict__ x,
                                                                 const int32_t *__restrict__ y,
                                                                 const TW *__restrict__ w, int64_t n, int ch, int cw,
                                                                 TW *__restrict__ canvas, uint32_t *oob) {
    foreach_events<VEC>(n, [&](int64_t base, int cnt, bool vec) {
        Vec4<int32_t> xv = load_col(x, base, cnt, vec), yv = load_col(y, base, cnt, vec);
        Vec4<TW> wv;
        if (w) wv = load_col(w, base, cnt, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= cnt) break;
            const int32_t xi = xv.v[k], yi = yv.v[k];
            if ((unsigned)xi < (unsigned)cw && (unsigned)yi < (unsigned)ch)
                atomic_add(canvas + (int64_t)yi * cw + xi, w ? wv.v[k] : TW(1));
            else
                count_oob(oob);
        }
    });
}

template <bool VEC>
__global__ void __launch_bounds__(EVK_BLOCK) k_image_nearest_f32(const float *__restrict__ x,
                                                                 const float *__restrict__ y,
                                                                 const float *__restrict__ w, int64_t n, int h, int wd,
                                                                 float clipx, float clipy, float *__restrict__ img,
                                                                 uint32_t *oob) {
    foreach_events<VEC>(n, [&](int64_t base, int cnt, bool vec) {
        Vec4<float> xv = load_col(x, base, cnt, vec), yv = load_col(y, base, cnt, vec),
                    wv = load_col(w, base, cnt, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= cnt) break;
            const float xf = xv.v[k], yf = yv.v[k];
            // mask multiplies the INDICES only; the weight survives (image.py:93-95)
            const bool keep = !(xf >= clipx) && !(yf >= clipy);
            long long xi = keep ? (long long)xf : 0, yi = keep ? (long long)yf : 0;  // .long(): toward zero
            // (a MASKED event's indices are x.long() * 0 = 0 whatever x was -- also NaN, whose .long() is INT64_MIN: it lands
            // on pixel (0, 0); an unmasked NaN is an index out of range)
            if ((!keep || (xf == xf && yf == yf)) && wrap_index(xi, wd) && wrap_index(yi, h))
                atomic_add(img + yi * wd + xi, wv.v[k]);
            else
                count_oob(oob);
        }
    });
}

// ---------------------------------------------------------------------------------------------------------
// bilinear splat (image.py:79-86, 102-115, 117-136)
// ---------------------------------------------------------------------------------------------------------

// interpolate_to_image / interpolate_to_derivative_img on caller-computed pixels and fractions (image.py:102-136).
__global__ void __launch_bounds__(EVK_BLOCK) k_splat_indexed_f32(const int64_t *__restrict__ px,
                                                                 const int64_t *__restrict__ py,
                                                                 const float *__restrict__ dx,
                                                                 const float *__restrict__ dy,
                                                                 const float *__restrict__ w, int64_t n, int h, int wd,
                                                                 float *__restrict__ img, uint32_t *oob) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Splat s{(long long)px[i], (long long)py[i], dx[i], dy[i]};
        if (!splat_iwe(img, h, wd, s, w[i])) count_oob(oob);
    }
}

__global__ void __launch_bounds__(EVK_BLOCK) k_splat_drv_indexed_f32(const int64_t *__restrict__ px,
                                                                     const int64_t *__restrict__ py,
                                                                     const float *__restrict__ dx,
                                                                     const float *__restrict__ dy,
                                                                     const float *__restrict__ w1,
                                                                     const float *__restrict__ w2, int C, int64_t n,
                                                                     int h, int wd, float *__restrict__ dimg,
                                                                     uint32_t *oob) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long x0 = px[i], x1 = x0 + 1, y0 = py[i], y1 = y0 + 1;
        if (!(wrap_index(x0, wd) && wrap_index(x1, wd) && wrap_index(y0, h) && wrap_index(y1, h))) {
            count_oob(oob);
            continue;
        }
        const float fx = dx[i], fy = dy[i], ax = 1.0f - fx, ay = 1.0f - fy;
        for (int c = 0; c < C; ++c) {
            const float a = w1[(int64_t)c * n + i], b = w2[(int64_t)c * n + i];
            float *d = dimg + (int64_t)c * h * wd;
            atomic_add(d + y0 * wd + x0, a * (-ay) + b * (-ax));
            atomic_add(d + y0 * wd + x1, a * ay + b * (-fx));
            atomic_add(d + y1 * wd + x0, a * (-fy) + b * ax);
            atomic_add(d + y1 * wd + x1, a * fy + b * fx);
        }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(EVK_BLOCK) k_image_bilinear_f32(const float *__restrict__ x,
                                                                  const float *__restrict__ y,
                                                                  const float *__restrict__ w, int64_t n, int h,
                                                                  int wd, float clipx, float clipy,
                                                                  float *__restrict__ img, uint32_t *oob) {
    foreach_events<VEC>(n, [&](int64_t base, int cnt, bool vec) {
        Vec4<float> xv = load_col(x, base, cnt, vec), yv = load_col(y, base, cnt, vec),
                    wv = load_col(w, base, cnt, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= cnt) break;
            const float xf = xv.v[k], yf = yv.v[k];
            const float mask = (!(xf >= clipx) && !(yf >= clipy)) ? 1.0f : 0.0f;
            const float fx = floorf(xf), fy = floorf(yf);
            Splat s;
            s.dx = xf - fx;
            s.dy = yf - fy;
            s.px = (long long)(fx * mask);
            s.py = (long long)(fy * mask);
            if (!splat_iwe(img, h, wd, s, wv.v[k] * mask)) count_oob(oob);
        }
    });
}

template <bool VEC>
__global__ void __launch_bounds__(EVK_BLOCK) k_image_drv_f64(const double *__restrict__ x,
                                                             const double *__restrict__ y,
                                                             const double *__restrict__ p,
                                                             const double *__restrict__ jx,
                                                             const double *__restrict__ jy, int64_t n, int h, int wd,
                                                             float clipx, float clipy, float *__restrict__ img,
                                                             float *__restrict__ dimg, uint32_t *oob) {
    foreach_events<VEC>(n, [&](int64_t base, int cnt, bool vec) {
        Vec4<double> xv = load_col(x, base, cnt, vec), yv = load_col(y, base, cnt, vec),
                     pv = load_col(p, base, cnt, vec);
        Vec4<double> jx0, jx1, jy0, jy1;
        if (jx) {
            jx0 = load_col(jx, base, cnt, vec);
            jy0 = load_col(jy, base, cnt, vec);
            const bool vec1 = vec && ((n & 3) == 0);  // second row starts at n: 32 B aligned only if n % 4 == 0
            jx1 = load_col(jx + n, base, cnt, vec1);
            jy1 = load_col(jy + n, base, cnt, vec1);
            if (vec && !vec1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    jx1.v[k] = jx[n + base + k];
                    jy1.v[k] = jy[n + base + k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= cnt) break;
            const float xf = (float)xv.v[k], yf = (float)yv.v[k], pf = (float)pv.v[k];  // cast BEFORE floor (Q7)
            const float mask = (!(xf >= clipx) && !(yf >= clipy)) ? 1.0f : 0.0f;
            const float fx = floorf(xf), fy = floorf(yf);
            Splat s;
            s.dx = xf - fx;
            s.dy = yf - fy;
            s.px = (long long)(fx * mask);
            s.py = (long long)(fy * mask);
            const float mp = pf * mask;
            if (!splat_iwe(img, h, wd, s, mp)) {
                count_oob(oob);
                continue;
            }
            if (jx) {
                long long x0 = s.px, x1 = s.px + 1, y0 = s.py, y1 = s.py + 1;
                wrap_index(x0, wd), wrap_index(x1, wd), wrap_index(y0, h), wrap_index(y1, h);
                const float ax = 1.0f - s.dx, ay = 1.0f - s.dy;
                const float w1[2] = {(float)jx0.v[k] * mp, (float)jx1.v[k] * mp};
                const float w2[2] = {(float)jy0.v[k] * mp, (float)jy1.v[k] * mp};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float *d = dimg + (int64_t)c * h * wd;
                    atomic_add(d + y0 * wd + x0, w1[c] * (-ay) + w2[c] * (-ax));
                    atomic_add(d + y0 * wd + x1, w1[c] * ay + w2[c] * (-s.dx));
                    atomic_add(d + y1 * wd + x0, w1[c] * (-s.dy) + w2[c] * ax);
                    atomic_add(d + y1 * wd + x1, w1[c] * s.dy + w2[c] * s.dx);
                }
            }
        }
    });
}

// image_to_event_weights (image.py:138-160): bilinear GATHER of a float32 (the IWE) or float64 image at float64 event
// coordinates; the products are float64 either way, as numpy's promotion makes them upstream.
template <typename IMG>
__global__ void __launch_bounds__(EVK_BLOCK) k_image_gather_f64(const double *__restrict__ x,
                                                                const double *__restrict__ y, int64_t n,
                                                                const IMG *__restrict__ img, int h, int wd,
                                                                double *__restrict__ out, uint32_t *oob) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double clipx = (double)(wd - 1), clipy = (double)(h - 1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double xv = x[i], yv = y[i];
        const double mask = ((xv >= clipx) ? 0.0 : 1.0) * ((yv >= clipy) ? 0.0 : 1.0);
        long long x0 = (long long)floor(xv * mask), y0 = (long long)floor(yv * mask);
        const double dx = xv - (double)x0, dy = yv - (double)y0;  // fractions of the UNMASKED coordinate (upstream)
        const double wx = 1.0 - dx, wy = 1.0 - dy;
        long long x1 = x0 + 1, y1 = y0 + 1;
        if (!(wrap_index(x0, wd) && wrap_index(x1, wd) && wrap_index(y0, h) && wrap_index(y1, h))) {
            count_oob(oob);
            out[i] = 0.0;
            continue;
        }
        double w = (double)img[y0 * wd + x0] * wx * wy;
        w += (double)img[y0 * wd + x1] * dx * wy;
        w += (double)img[y1 * wd + x0] * wx * dy;
        w += (double)img[y1 * wd + x1] * dx * dy;
        out[i] = w * mask;
    }
}

// Average-timestamp images (image.py:219-353): four bilinear splats per event -- normalised timestamp and count, for
// positive and non-positive events.  nts = (t - ta) / td (mode 0), (-t + ta) / td (mode 1), t (mode 2), float32.
// Upstream quirk kept: clipped events are moved to pixel (0, 0) but keep their weights (masked_ps is never used).
__global__ void __launch_bounds__(EVK_BLOCK) k_timestamp_images_f32(const float *__restrict__ x,
                                                                    const float *__restrict__ y,
                                                                    const float *__restrict__ t,
                                                                    const float *__restrict__ p, int64_t n, int h,
                                                                    int wd, float clipx, float clipy, int mode,
                                                                    float ta, float td, float *__restrict__ out,
                                                                    uint32_t *oob) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t plane = (int64_t)h * wd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float xf = x[i], yf = y[i], tf = t[i];
        const float mask = (!(xf >= clipx) && !(yf >= clipy)) ? 1.0f : 0.0f;
        const float fx = floorf(xf), fy = floorf(yf);
        Splat s;
        s.dx = xf - fx;
        s.dy = yf - fy;
        s.px = (long long)(fx * mask);
        s.py = (long long)(fy * mask);
        const float nts = mode == 0 ? (tf - ta) / td : (mode == 1 ? (-tf + ta) / td : tf);
        const bool pos = p[i] > 0.0f;  // pos_events_mask = ps > 0, neg_events_mask = ps <= 0 (NaN: neither)
        const bool neg = p[i] <= 0.0f;
        if (!pos && !neg) continue;
        float *val = out + (pos ? 0 : 2) * plane, *cnt = val + plane;
        if (!splat_iwe(val, h, wd, s, nts) || !splat_iwe(cnt, h, wd, s, 1.0f)) count_oob(oob);
    }
}

// warp_events_flow_torch (lib/transforms/optic_flow.py:5-46): bilinear sample of a dense (2, h, wd) flow field at the
// event position (F.grid_sample semantics: align_corners=True, zero padding; the coordinate is normalised to [-1, 1] and
// un-normalised again, both in float32, :38-39), then x' = x + flow_x * (t - t0), y' = y + flow_y * (t - t0).
__global__ void __launch_bounds__(EVK_BLOCK) k_warp_flow_field_f32(const float *__restrict__ x,
                                                                   const float *__restrict__ y,
                                                                   const float *__restrict__ t, int64_t n,
                                                                   const float *__restrict__ flow, int h, int wd,
                                                                   float t0, float *__restrict__ xo,
                                                                   float *__restrict__ yo) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t plane = (int64_t)h * wd;
    const float wm1 = (float)(wd - 1), hm1 = (float)(h - 1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float xv = x[i], yv = y[i];
        const float gx = xv / wm1 * 2.0f - 1.0f, gy = yv / hm1 * 2.0f - 1.0f;
        const float ix = (gx + 1.0f) / 2.0f * wm1, iy = (gy + 1.0f) / 2.0f * hm1;
        const float xw = floorf(ix), yn = floorf(iy);
        const float w = ix - xw, e = 1.0f - w, nn = iy - yn, ss = 1.0f - nn;
        const int x0 = (int)xw, y0 = (int)yn;
        float f[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float *fl = flow + c * plane;
            auto at = [&](int yy, int xx) -> float {
                return (xx >= 0 && xx < wd && yy >= 0 && yy < h) ? fl[(int64_t)yy * wd + xx] : 0.0f;
            };
            float acc = at(y0, x0) * (e * ss);
            acc = acc + at(y0, x0 + 1) * (w * ss);
            acc = acc + at(y0 + 1, x0) * (e * nn);
            acc = acc + at(y0 + 1, x0 + 1) * (w * nn);
            f[c] = acc;
        }
        const float dt = t[i] - t0;
        xo[i] = xv + f[0] * dt;
        yo[i] = yv + f[1] * dt;
    }
}

// ---------------------------------------------------------------------------------------------------------
// voxel grid: single pass, <= 2 bins per event
// ---------------------------------------------------------------------------------------------------------

// Adds p*max(0, 1-|t_norm-b|) for every bin b where it is non-zero (b = floor(t_norm), floor(t_norm)+1); a NaN
// t_norm (dt == 0, quirk Q9) poisons all B bins of the pixel exactly as the reference does, and so does a polarity that is
// not finite: the reference adds p * weight to EVERY bin (voxel_grid.py:138-142), and NaN * 0 = inf * 0 = NaN.
template <typename T>
__device__ __forceinline__ void voxel_bins(T *__restrict__ vox, int64_t plane, int64_t pix, int B, T tn, T p) {
    if (tn != tn) {
        for (int b = 0; b < B; ++b) atomic_add(vox + b * plane + pix, tn * p);
        return;
    }
    if (p - p != (T)0) {   // NaN or +-inf (finite p: p - p == 0)
        for (int b = 0; b < B; ++b) atomic_add(vox + b * plane + pix, p * fmax((T)0, (T)1 - fabs(tn - (T)b)));
        return;
    }
    const T fl = floor(tn);
    // clamp before the int conversion so that +-inf / huge values cannot overflow; they touch no bin anyway
    const int b0 = (int)fmax(fmin(fl, (T)(B + 1)), (T)-2);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int b = b0 + k;
        if (b < 0 || b >= B) continue;
        const T wgt = fmax((T)0, (T)1 - fabs(tn - (T)b));
        const T val = p * wgt;
        if (val != (T)0) atomic_add(vox + b * plane + pix, val);
    }
}

template <bool VEC>
__global__ void __launch_bounds__(EVK_BLOCK) k_voxel_f32(const float *__restrict__ x, const float *__restrict__ y,
                                                         const float *__restrict__ t, const float *__restrict__ p,
                                                         int64_t n, float t_first, float dt, float bm1, int B, int h,
                                                         int wd, float *__restrict__ vox, uint32_t *oob, int t_from_events) {
    const int64_t plane = (int64_t)h * wd;
    if (t_from_events) t_first = t[0], dt = t[n - 1] - t_first;   // ts[0], ts[-1] (voxel_grid.py:133): no transfer before the launch
    foreach_events<VEC>(n, [&](int64_t base, int cnt, bool vec) {
        Vec4<float> xv = load_col(x, base, cnt, vec), yv = load_col(y, base, cnt, vec),
                    tv = load_col(t, base, cnt, vec), pv = load_col(p, base, cnt, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= cnt) break;
            long long xi = (long long)xv.v[k], yi = (long long)yv.v[k];
            // NaN.long() is INT64_MIN in torch (IndexError); the hardware conversion gives 0, so reject it explicitly
            if (xv.v[k] != xv.v[k] || yv.v[k] != yv.v[k] || !(wrap_index(xi, wd) && wrap_index(yi, h))) {
                count_oob(oob);
                continue;
            }
            const float tn = (tv.v[k] - t_first) / dt * bm1;  // voxel_grid.py:134, float32, IEEE divide
            voxel_bins<float>(vox, plane, yi * wd + xi, B, tn, pv.v[k]);
        }
    });
}

template <bool VEC>
__global__ void __launch_bounds__(EVK_BLOCK) k_voxel_f64(const int32_t *__restrict__ x, const int32_t *__restrict__ y,
                                                         const double *__restrict__ t, const double *__restrict__ p,
                                                         int64_t n, double t_first, double dt, double bm1, int B,
                                                         int h, int wd, double *__restrict__ vox, uint32_t *oob) {
    const int64_t plane = (int64_t)h * wd;
    foreach_events<VEC>(n, [&](int64_t base, int cnt, bool vec) {
        Vec4<int32_t> xv = load_col(x, base, cnt, vec), yv = load_col(y, base, cnt, vec);
        Vec4<double> tv = load_col(t, base, cnt, vec), pv = load_col(p, base, cnt, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= cnt) break;
            const int xi = xv.v[k], yi = yv.v[k];
            if (xi < 0 || xi > wd || yi < 0 || yi > h) {  // outside the (h+1, wd+1) canvas: ValueError
                count_oob(oob);
                continue;
            }
            if (xi == wd || yi == h) continue;  // pad row / column, cropped away (image.py:44)
            const double tn = (tv.v[k] - t_first) / dt * bm1;
            voxel_bins<double>(vox, plane, (int64_t)yi * wd + xi, B, tn, pv.v[k]);
        }
    });
}

// Windowed voxelisation (voxel_grids_fixed_n_torch / voxel_grids_fixed_t_torch, voxel_grid.py:37-112): S consecutive
// event ranges [seg[s], seg[s+1]) -> S independent voxel grids in ONE launch.  blockIdx.y = window; each window
// normalises time with ITS first / last event exactly as the per-window reference call does (voxel_grid.py:133-134).
__global__ void __launch_bounds__(EVK_BLOCK) k_voxel_segments_f32(const float *__restrict__ x,
                                                                  const float *__restrict__ y,
                                                                  const float *__restrict__ t,
                                                                  const float *__restrict__ p,
                                                                  const int64_t *__restrict__ seg, int B, int h, int wd,
                                                                  float *__restrict__ vox, uint32_t *oob) {
    const int s = blockIdx.y;
    const int64_t lo = seg[s], hi = seg[s + 1];
    if (hi <= lo) return;
    const float t_first = t[lo], dt = t[hi - 1] - t_first, bm1 = (float)(B - 1);
    const int64_t plane = (int64_t)h * wd;
    float *out = vox + (int64_t)s * B * plane;
    for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (int64_t)gridDim.x * blockDim.x) {
        const float xf = x[i], yf = y[i];
        long long xi = (long long)xf, yi = (long long)yf;
        if (xf != xf || yf != yf || !(wrap_index(xi, wd) && wrap_index(yi, h))) {
            count_oob(oob);
            continue;
        }
        const float tn = (t[i] - t_first) / dt * bm1;
        voxel_bins<float>(out, plane, yi * wd + xi, B, tn, p[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// warp / mask (elementwise, float64) and the fused linear-flow IWE
// ---------------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(EVK_BLOCK) k_warp_linvel_f64(const double *__restrict__ x,
                                                               const double *__restrict__ y,
                                                               const double *__restrict__ t, int64_t n, double t0,
                                                               double vx, double vy, double *__restrict__ xo,
                                                               double *__restrict__ yo, double *__restrict__ jx,
                                                               double *__restrict__ jy) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double dt = t[i] - t0;
        xo[i] = x[i] - dt * vx;
        yo[i] = y[i] - dt * vy;
        if (jx) {
            jx[i] = -dt;
            jx[n + i] = 0.0;
            jy[i] = 0.0;
            jy[n + i] = -dt;
        }
    }
}

__global__ void __launch_bounds__(EVK_BLOCK) k_bounds_mask_f64(const double *__restrict__ x,
                                                               const double *__restrict__ y, int64_t n, double xmin,
                                                               double xmax, double ymin, double ymax,
                                                               double *__restrict__ mask) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double mx = (x[i] <= xmin || x[i] > xmax) ? 0.0 : 1.0;
        const double my = (y[i] <= ymin || y[i] > ymax) ? 0.0 : 1.0;
        mask[i] = mx * my;
    }
}

// Per-event part of get_iwe shared by the direct and the tiled kernels: everything up to the splat coordinates.
// Returns false when the event contributes nothing (rejected by either mask: its splat would add +-0 at (0,0)).
template <typename T>
__device__ __forceinline__ bool iwe_event(T x, T y, T t, T p, double t_ref, double vx, double vy, double bw,
                                          double bh, float clipx, float clipy, bool abs_p, double p_scale, int &px,
                                          int &py, float &dx, float &dy, float &mp, float &jf) {
    const double dt = (double)t - t_ref;           // warps.py:52
    const double xw = (double)x - dt * vx;         // warps.py:53 (two roundings: no FMA)
    const double yw = (double)y - dt * vy;         // warps.py:54
    // events_bounds_mask(xs, ys, 0, W, 0, H) (event_util.py:26-27, quirk Q2)
    if (xw <= 0.0 || xw > bw || yw <= 0.0 || yw > bh) return false;
    const double ps = (double)p * p_scale;         // adaptive lifespan: ps*100 (objectives.py:225); else 1
    const double pd = abs_p ? fabs(ps) : ps;
    const float xf = (float)xw, yf = (float)yw;    // image.py:179-180 (cast before floor, Q7)
    if (xf >= clipx || yf >= clipy) return false;  // image.py:195-197
    const float fx = floorf(xf), fy = floorf(yf);
    dx = xf - fx;
    dy = yf - fy;
    px = (int)fx;
    py = (int)fy;
    mp = (float)pd;
    jf = (float)(-dt);                             // jacobian_x[0,:] = jacobian_y[1,:] = -dt (warps.py:59-60)
    return true;
}

template <typename T, bool VEC, bool GRAD>
__global__ void __launch_bounds__(EVK_BLOCK) k_iwe_linvel(const T *__restrict__ x, const T *__restrict__ y,
                                                          const T *__restrict__ t, const T *__restrict__ p, int64_t n,
                                                          double t_ref, double vx, double vy, double bw, double bh,
                                                          int ch, int cw, bool abs_p, double p_scale,
                                                          float *__restrict__ iwe, float *__restrict__ diwe) {
    const float clipx = (float)(cw - 1), clipy = (float)(ch - 1);
    const int64_t plane = (int64_t)ch * cw;
    foreach_events<VEC>(n, [&](int64_t base, int cnt, bool vec) {
        Vec4<T> xv = load_col(x, base, cnt, vec), yv = load_col(y, base, cnt, vec), tv = load_col(t, base, cnt, vec),
                pv = load_col(p, base, cnt, vec);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= cnt) break;
            int px, py;
            float dx, dy, mp, jf;
            if (!iwe_event<T>(xv.v[k], yv.v[k], tv.v[k], pv.v[k], t_ref, vx, vy, bw, bh, clipx, clipy, abs_p, p_scale,
                              px, py, dx, dy, mp, jf))
                continue;
            const float ax = 1.0f - dx, ay = 1.0f - dy;
            float *c = iwe + (int64_t)py * cw + px;
            atomic_add(c, mp * ax * ay);
            atomic_add(c + 1, mp * dx * ay);
            atomic_add(c + cw, mp * ax * dy);
            atomic_add(c + cw + 1, mp * dx * dy);
            if constexpr (GRAD) {
                const float a = jf * mp;  // w1[0] = w2[1]; w1[1] = w2[0] = 0 (image.py:211-212)
                float *d0 = diwe + (int64_t)py * cw + px, *d1 = d0 + plane;
                atomic_add(d0, a * (-ay));
                atomic_add(d0 + 1, a * ay);
                atomic_add(d0 + cw, a * (-dy));
                atomic_add(d0 + cw + 1, a * dy);
                atomic_add(d1, a * (-ax));
                atomic_add(d1 + 1, a * (-dx));
                atomic_add(d1 + cw, a * ax);
                atomic_add(d1 + cw + 1, a * dx);
            }
        }
    });
}

}  // namespace evk

// =============================================================================================================
// C ABI
// =============================================================================================================
using namespace evk;

#define EVK_STREAM(s) ((hipStream_t)(s))

extern "C" int evk_version(void) { return EVK_VERSION; }

extern "C" const char *evk_error_string(int code) {
    switch (code) {
        case EVK_OK: return "ok";
        case EVK_EINVAL: return "invalid argument";
        case EVK_ESCRATCH: return "scratch buffer too small";
        case EVK_EALIGN: return "pointer not sufficiently aligned";
        case EVK_ECOMM: return "RCCL unavailable or collective failed";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown evk error";
    }
}

extern "C" int evk_image_nearest_i32(const int32_t *x, const int32_t *y, const int32_t *w, int64_t n, int ch, int cw,
                                     int32_t *canvas, uint32_t *oob, void *stream) {
    if (n < 0 || ch <= 0 || cw <= 0 || !canvas || (n > 0 && (!x || !y))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    const bool vec = aligned16(x) && aligned16(y) && (!w || aligned16(w));
    if (vec)
        k_image_nearest_int<true, int32_t><<<stream_grid(n, 4), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, w, n, ch, cw,
                                                                                                 canvas, oob);
    else
        k_image_nearest_int<false, int32_t><<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, w, n, ch, cw,
                                                                                               canvas, oob);
    return launch_status();
}

extern "C" int evk_image_nearest_f64(const int32_t *x, const int32_t *y, const double *w, int64_t n, int ch, int cw,
                                     double *canvas, uint32_t *oob, void *stream) {
    if (n < 0 || ch <= 0 || cw <= 0 || !canvas || (n > 0 && (!x || !y))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    const bool vec = aligned16(x) && aligned16(y) && (!w || aligned16(w));
    if (vec)
        k_image_nearest_int<true, double><<<stream_grid(n, 4), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, w, n, ch, cw,
                                                                                                canvas, oob);
    else
        k_image_nearest_int<false, double><<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, w, n, ch, cw,
                                                                                              canvas, oob);
    return launch_status();
}

extern "C" int evk_image_nearest_f32(const float *x, const float *y, const float *w, int64_t n, int h, int wd,
                                     float clipx, float clipy, float *img, uint32_t *oob, void *stream) {
    if (n < 0 || h <= 0 || wd <= 0 || !img || (n > 0 && (!x || !y || !w))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    if (aligned16(x) && aligned16(y) && aligned16(w))
        k_image_nearest_f32<true><<<stream_grid(n, 4), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, w, n, h, wd, clipx,
                                                                                        clipy, img, oob);
    else
        k_image_nearest_f32<false><<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, w, n, h, wd, clipx,
                                                                                      clipy, img, oob);
    return launch_status();
}

extern "C" int evk_image_bilinear_f32(const float *x, const float *y, const float *w, int64_t n, int h, int wd,
                                      float clipx, float clipy, float *img, uint32_t *oob, void *stream) {
    if (n < 0 || h <= 1 || wd <= 1 || !img || (n > 0 && (!x || !y || !w))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    if (aligned16(x) && aligned16(y) && aligned16(w))
        k_image_bilinear_f32<true><<<stream_grid(n, 4), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, w, n, h, wd, clipx,
                                                                                         clipy, img, oob);
    else
        k_image_bilinear_f32<false><<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, w, n, h, wd, clipx,
                                                                                       clipy, img, oob);
    return launch_status();
}

extern "C" int evk_splat_indexed_f32(const int64_t *px, const int64_t *py, const float *dx, const float *dy,
                                     const float *w, int64_t n, int h, int wd, float *img, uint32_t *oob,
                                     void *stream) {
    if (n < 0 || h <= 0 || wd <= 0 || !img || (n > 0 && (!px || !py || !dx || !dy || !w))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_splat_indexed_f32<<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(px, py, dx, dy, w, n, h, wd, img, oob);
    return launch_status();
}

extern "C" int evk_splat_drv_indexed_f32(const int64_t *px, const int64_t *py, const float *dx, const float *dy,
                                         const float *w1, const float *w2, int C, int64_t n, int h, int wd,
                                         float *d_img, uint32_t *oob, void *stream) {
    if (n < 0 || C <= 0 || h <= 0 || wd <= 0 || !d_img || (n > 0 && (!px || !py || !dx || !dy || !w1 || !w2)))
        return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_splat_drv_indexed_f32<<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(px, py, dx, dy, w1, w2, C, n, h, wd,
                                                                               d_img, oob);
    return launch_status();
}

extern "C" int evk_image_drv_f64(const double *x, const double *y, const double *p, const double *jx,
                                 const double *jy, int64_t n, int h, int wd, float clipx, float clipy, float *img,
                                 float *d_img, uint32_t *oob, void *stream) {
    if (n < 0 || h <= 1 || wd <= 1 || !img || (n > 0 && (!x || !y || !p))) return EVK_EINVAL;
    if ((jx == nullptr) != (jy == nullptr) || (jx && !d_img)) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    const bool vec = aligned16(x) && aligned16(y) && aligned16(p) && (!jx || (aligned16(jx) && aligned16(jy)));
    if (vec)
        k_image_drv_f64<true><<<stream_grid(n, 4), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, p, jx, jy, n, h, wd, clipx,
                                                                                    clipy, img, d_img, oob);
    else
        k_image_drv_f64<false><<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, p, jx, jy, n, h, wd, clipx,
                                                                                  clipy, img, d_img, oob);
    return launch_status();
}

extern "C" int evk_image_gather_bilinear_f64(const double *x, const double *y, int64_t n, const float *img, int h,
                                             int wd, double *out, uint32_t *oob, void *stream) {
    if (n < 0 || h <= 1 || wd <= 1 || !img || (n > 0 && (!x || !y || !out))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_image_gather_f64<float><<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, n, img, h, wd, out, oob);
    return launch_status();
}

extern "C" int evk_image_gather_bilinear_f64img(const double *x, const double *y, int64_t n, const double *img, int h,
                                                int wd, double *out, uint32_t *oob, void *stream) {
    if (n < 0 || h <= 1 || wd <= 1 || !img || (n > 0 && (!x || !y || !out))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_image_gather_f64<double><<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, n, img, h, wd, out, oob);
    return launch_status();
}

extern "C" int evk_timestamp_images_f32(const float *x, const float *y, const float *t, const float *p, int64_t n,
                                        int h, int wd, float clipx, float clipy, int mode, float ta, float td,
                                        float *out4, uint32_t *oob, void *stream) {
    if (n < 0 || h <= 1 || wd <= 1 || !out4 || mode < 0 || mode > 2 || (n > 0 && (!x || !y || !t || !p)))
        return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_timestamp_images_f32<<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, t, p, n, h, wd, clipx, clipy, mode,
                                                                              ta, td, out4, oob);
    return launch_status();
}

extern "C" int evk_warp_flow_field_f32(const float *x, const float *y, const float *t, int64_t n, const float *flow,
                                       int h, int wd, float t0, float *xo, float *yo, void *stream) {
    if (n < 0 || h <= 1 || wd <= 1 || !flow || (n > 0 && (!x || !y || !t || !xo || !yo))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_warp_flow_field_f32<<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, t, n, flow, h, wd, t0, xo, yo);
    return launch_status();
}

static int voxel_direct(const float *x, const float *y, const float *t, const float *p, int64_t n, float t_first, float t_last,
                        int from_events, int B, int h, int wd, float *vox, uint32_t *oob, void *stream) {
    if (n < 0 || B <= 0 || h <= 0 || wd <= 0 || !vox || (n > 0 && (!x || !y || !t || !p))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    const float dt = t_last - t_first, bm1 = (float)(B - 1);
    if (aligned16(x) && aligned16(y) && aligned16(t) && aligned16(p))
        k_voxel_f32<true><<<stream_grid(n, 4), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, t, p, n, t_first, dt, bm1, B,
                                                                                h, wd, vox, oob, from_events);
    else
        k_voxel_f32<false><<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, t, p, n, t_first, dt, bm1, B, h,
                                                                              wd, vox, oob, from_events);
    return launch_status();
}
extern "C" int evk_voxel_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, float t_first,
                             float t_last, int B, int h, int wd, float *vox, uint32_t *oob, void *stream) {
    return voxel_direct(x, y, t, p, n, t_first, t_last, 0, B, h, wd, vox, oob, stream);
}
extern "C" int evk_voxel_from_events_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int B,
                                         int h, int wd, float *vox, uint32_t *oob, void *stream) {
    return voxel_direct(x, y, t, p, n, 0.0f, 0.0f, 1, B, h, wd, vox, oob, stream);
}

extern "C" int evk_voxel_f64(const int32_t *x, const int32_t *y, const double *t, const double *p, int64_t n,
                             double t_first, double t_last, int B, int h, int wd, double *vox, uint32_t *oob,
                             void *stream) {
    if (n < 0 || B <= 0 || h <= 0 || wd <= 0 || !vox || (n > 0 && (!x || !y || !t || !p))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    const double dt = t_last - t_first, bm1 = (double)(B - 1);
    if (aligned16(x) && aligned16(y) && aligned16(t) && aligned16(p))
        k_voxel_f64<true><<<stream_grid(n, 4), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, t, p, n, t_first, dt, bm1, B,
                                                                                h, wd, vox, oob);
    else
        k_voxel_f64<false><<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, t, p, n, t_first, dt, bm1, B, h,
                                                                              wd, vox, oob);
    return launch_status();
}

extern "C" int evk_voxel_segments_f32(const float *x, const float *y, const float *t, const float *p,
                                      const int64_t *seg, int nseg, int64_t max_seg_len, int B, int h, int wd,
                                      float *vox, uint32_t *oob, void *stream) {
    if (nseg < 0 || B <= 0 || h <= 0 || wd <= 0 || !vox || (nseg > 0 && (!x || !y || !t || !p || !seg))) return EVK_EINVAL;
    if (nseg == 0 || max_seg_len <= 0) return EVK_OK;
    if (nseg > 65535) return EVK_EINVAL;
    int bx = stream_grid(max_seg_len);
    const int cap = (EVK_NUM_CU * 8 + nseg - 1) / nseg;   // keep the whole grid around 8 blocks per CU
    if (bx > cap) bx = cap < 1 ? 1 : cap;
    k_voxel_segments_f32<<<dim3(bx, nseg), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, t, p, seg, B, h, wd, vox, oob);
    return launch_status();
}

extern "C" int evk_warp_linvel_f64(const double *x, const double *y, const double *t, int64_t n, double t0, double vx,
                                   double vy, double *xo, double *yo, double *jx, double *jy, void *stream) {
    if (n < 0 || (n > 0 && (!x || !y || !t || !xo || !yo)) || ((jx == nullptr) != (jy == nullptr))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_warp_linvel_f64<<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, t, n, t0, vx, vy, xo, yo, jx, jy);
    return launch_status();
}

extern "C" int evk_bounds_mask_f64(const double *x, const double *y, int64_t n, double xmin, double xmax, double ymin,
                                   double ymax, double *mask, void *stream) {
    if (n < 0 || (n > 0 && (!x || !y || !mask))) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    k_bounds_mask_f64<<<stream_grid(n), EVK_BLOCK, 0, EVK_STREAM(stream)>>>(x, y, n, xmin, xmax, ymin, ymax, mask);
    return launch_status();
}

template <typename T>
static int launch_iwe(const T *x, const T *y, const T *t, const T *p, int64_t n, double t_ref, double vx, double vy,
                      double bw, double bh, int ch, int cw, uint32_t flags, double p_scale, float *iwe, float *diwe,
                      void *stream) {
    if (n < 0 || ch <= 1 || cw <= 1 || !iwe || (n > 0 && (!x || !y || !t || !p))) return EVK_EINVAL;
    const bool grad = flags & EVK_IWE_GRADIENT;
    if (grad && !diwe) return EVK_EINVAL;
    if (n == 0) return EVK_OK;
    const bool abs_p = flags & EVK_IWE_ABS_POLARITY;
    const bool vec = aligned16(x) && aligned16(y) && aligned16(t) && aligned16(p);
    hipStream_t s = EVK_STREAM(stream);
#define EVK_LAUNCH_IWE(V, G)                                                                                     \
    k_iwe_linvel<T, V, G><<<stream_grid(n, V ? 4 : 1), EVK_BLOCK, 0, s>>>(x, y, t, p, n, t_ref, vx, vy, bw, bh, ch, \
                                                                        cw, abs_p, p_scale, iwe, diwe)
    if (vec && grad) EVK_LAUNCH_IWE(true, true);
    else if (vec) EVK_LAUNCH_IWE(true, false);
    else if (grad) EVK_LAUNCH_IWE(false, true);
    else EVK_LAUNCH_IWE(false, false);
#undef EVK_LAUNCH_IWE
    return launch_status();
}

extern "C" int evk_iwe_linvel_f32(const float *x, const float *y, const float *t, const float *p, int64_t n,
                                  double t_ref, double vx, double vy, double bounds_w, double bounds_h, int canvas_h,
                                  int canvas_w, uint32_t flags, double p_scale, float *iwe, float *diwe, void *stream) {
    return launch_iwe<float>(x, y, t, p, n, t_ref, vx, vy, bounds_w, bounds_h, canvas_h, canvas_w, flags, p_scale, iwe, diwe,
                             stream);
}

extern "C" int evk_iwe_linvel_f64(const double *x, const double *y, const double *t, const double *p, int64_t n,
                                  double t_ref, double vx, double vy, double bounds_w, double bounds_h, int canvas_h,
                                  int canvas_w, uint32_t flags, double p_scale, float *iwe, float *diwe, void *stream) {
    return launch_iwe<double>(x, y, t, p, n, t_ref, vx, vy, bounds_w, bounds_h, canvas_h, canvas_w, flags, p_scale, iwe, diwe,
                              stream);
}
