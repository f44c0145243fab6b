// Index rules and the four-neighbour splat of the global-atomic image kernels (evk_scatter.hip), shared with the one-pass
// image path (evk_image2.hip), whose partition kernel hands the rare events -- pixels that wrap or raise in index_put_ --
// to exactly this code.
#pragma once
#include "evk_common.h"

namespace evk {

// torch index semantics: negative indices wrap once, anything else out of [0, dim) is an error.
__device__ __forceinline__ bool wrap_index(long long &i, int dim) {
    if (i < 0) i += dim;
    return i >= 0 && i < dim;
}

struct Splat {
    long long px, py;
    float dx, dy;
};

// Four IWE atomics, products evaluated in the reference's order (image.py:111-114).
__device__ __forceinline__ bool splat_iwe(float *img, int h, int wd, const Splat &s, float w) {
    long long x0 = s.px, x1 = s.px + 1, y0 = s.py, y1 = s.py + 1;
    if (!(wrap_index(x0, wd) && wrap_index(x1, wd) && wrap_index(y0, h) && wrap_index(y1, h))) return false;
    const float ax = 1.0f - s.dx, ay = 1.0f - s.dy;
    atomic_add(img + y0 * wd + x0, w * ax * ay);
    atomic_add(img + y0 * wd + x1, w * s.dx * ay);
    atomic_add(img + y1 * wd + x0, w * ax * s.dy);
    atomic_add(img + y1 * wd + x1, w * s.dx * s.dy);
    return true;
}

}  // namespace evk
