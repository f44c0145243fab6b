// One-call objective evaluation: memset -> tiled IWE (+dIWE) -> gather -> fused blur + reductions -> finalise, all
// enqueued back to back from C so that the host (Python) pays for one call instead of four and the GPU never waits
// for the interpreter between the kernels of one evaluation.
#include "evk_common.h"

extern "C" int evk_cmax_variance_tiled_f32(const float *records, const uint32_t *bucket_start, int64_t n, int dom_h, int dom_w,
                                           int tw_log2, int th_log2, int slices, int win_w, int win_h, double t_first,
                                           double t_ref, double vx, double vy, double bounds_w, double bounds_h,
                                           int canvas_h, int canvas_w, uint32_t iwe_flags, double p_scale,
                                           double p_bound, double dt_bound, const double *host_weights, int radius, uint32_t post_flags, void *staging,
                                           int64_t staging_bytes, float *iwe_buf, double *out, void *scratch,
                                           int64_t scratch_bytes, void *stream) {
    if (!iwe_buf || canvas_h <= 1 || canvas_w <= 1) return EVK_EINVAL;
    const bool grad = iwe_flags & EVK_IWE_GRADIENT;
    const size_t plane = (size_t)canvas_h * canvas_w;
    hipError_t e = hipMemsetAsync(iwe_buf, 0, (grad ? 3 : 1) * plane * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    float *diwe = grad ? iwe_buf + plane : nullptr;
    int rc = evk_iwe_linvel_tiled_f32(records, bucket_start, n, dom_h, dom_w, tw_log2, th_log2, slices, win_w, win_h,
                                      t_first, t_ref, vx, vy, bounds_w, bounds_h, canvas_h, canvas_w, iwe_flags, p_scale,
                                      p_bound, dt_bound, staging, staging_bytes, iwe_buf, diwe, stream);
    if (rc != EVK_OK) return rc;
    if (grad && (post_flags & EVK_POST_VALUE))
        return evk_objective_variance_fg_f32(iwe_buf, diwe, canvas_h, canvas_w, host_weights, radius,
                                             post_flags & ~EVK_POST_VALUE, out, scratch, scratch_bytes, stream);
    if (grad)
        return evk_objective_variance_grad_f32(iwe_buf, diwe, canvas_h, canvas_w, host_weights, radius, post_flags, out,
                                               scratch, scratch_bytes, stream);
    return evk_objective_variance_f32(iwe_buf, canvas_h, canvas_w, host_weights, radius, out, scratch, scratch_bytes,
                                      stream);
}

extern "C" int evk_cmax_variance_batch3_tiled_f32(const float *records, const uint32_t *bucket_index, int64_t n,
                                                  int dom_h, int dom_w, int tw_log2, int th_log2, int slices, int win_w,
                                                  int win_h, double t_first, double t_ref, const double *host_vx,
                                                  const double *host_vy, double bounds_w, double bounds_h,
                                                  int canvas_h, int canvas_w, uint32_t iwe_flags, double p_scale,
                                                  double p_bound, double dt_bound, const double *host_weights, int radius, void *staging,
                                                  int64_t staging_bytes, float *iwe3, double *out12, void *scratch,
                                                  int64_t scratch_bytes, void *stream) {
    if (!iwe3 || canvas_h <= 1 || canvas_w <= 1) return EVK_EINVAL;
    const size_t plane = (size_t)canvas_h * canvas_w;
    hipError_t e = hipMemsetAsync(iwe3, 0, 3 * plane * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    int rc = evk_iwe_linvel_tiled_batch3_f32(records, bucket_index, n, dom_h, dom_w, tw_log2, th_log2, slices, win_w,
                                             win_h, t_first, t_ref, host_vx, host_vy, bounds_w, bounds_h, canvas_h,
                                             canvas_w, iwe_flags, p_scale, p_bound, dt_bound, staging, staging_bytes, iwe3, stream);
    if (rc != EVK_OK) return rc;
    return evk_objective_variance_planes_f32(iwe3, 3, canvas_h, canvas_w, host_weights, radius, out12, scratch,
                                             scratch_bytes, stream);
}
