// One-call objective evaluation: tiled IWE (+dIWE) -> gather -> fused blur + reductions -> finalise, all enqueued back
// to back from C so that the host (Python) pays for one call and the GPU never waits for the interpreter between the
// kernels of one evaluation.  With a spill pair (two zeroed (planes, ch, cw) images the caller keeps between calls) there is
// no memset: the rare out-of-window atomics of k_iwe_tiled go to spill[parity], the gather WRITES spill + windows and
// zeroes what the previous call left in spill[parity ^ 1].  host_out (4 / 12 doubles, host memory) makes the call
// synchronous: the results are copied through a pinned slot and the stream is synchronised inside the call.
// (Fusing the gather into the post-pass as well -- one kernel gathering a 40x40 halo patch per 32x32 tile -- measured
// 35.6 us against 8.1 + 9.2 us for the two kernels at 640x480: every halo pixel is gathered 1.6 times from a window
// list five times longer.  Removed.)
#include "evk_common.h"

int evk_iwe_tiled_spill(int mode, const float *records, const uint32_t *bucket_index, int64_t n, int dom_h, int dom_w,
                        int tw_log2, int th_log2, int slices, int win_w, int win_h, double t_first, double t_ref,
                        const double *vx, const double *vy, double bounds_w, double bounds_h, int canvas_h, int canvas_w,
                        uint32_t flags, double p_scale, double p_bound, double dt_bound, void *staging, int64_t staging_bytes,
                        float *iwe_buf, const float *spill, float *spill_clean, void *stream);

#include "evk_img.h"
using evk::HostPublish;
int evk_post_variance_publish(int mode, const float *iwe, const float *diwe, int h, int w, const double *host_weights,
                              int radius, uint32_t flags, double *out, void *scratch, int64_t scratch_bytes, void *stream,
                              int nplanes, const HostPublish *pub, bool *published);

// Results to the host.  The finalise kernel stores them in a pinned slot and then a sequence number (system scope); the
// host polls that flag: no copy command, no stream synchronisation (their completion signal + wake-up cost ~15 us per
// evaluation, a fifth of a 10 M-event evaluation).  Every 16 K polls the stream is queried so that a failed launch
// cannot hang the caller.  A post-pass that cannot publish takes the copy + synchronise route.
// (Letting the post-pass kernel finalise as well -- last workgroup of a plane by ticket, one launch less -- is SLOWER, with
// or without fences: round 2, an agent-scope release per workgroup: 69.9 vs 66.6 us per 10 M-event evaluation; round 3, the
// partial sums handed over with agent-scope stores and loads and a relaxed ticket, no fence: 70.1 vs 65.6 us -- every
// one of the 300 short workgroups waits for its ticket to come back, and the last one then adds the partials alone.)
struct HostSlot {
    double *vals = nullptr;    // 12 doubles
    uint32_t *flags = nullptr; // 3 sequence numbers, one per plane
    uint32_t seq = 0;
};
static HostSlot *host_slot() {
    static thread_local HostSlot hs;
    if (!hs.vals) {
        void *p = nullptr;
        if (hipHostMalloc(&p, 256, hipHostMallocDefault) != hipSuccess) return nullptr;
        hs.vals = (double *)p;
        hs.flags = (uint32_t *)((char *)p + 128);
        for (int k = 0; k < 3; ++k) hs.flags[k] = 0;
    }
    return &hs;
}

static int fetch_results(const double *out, int count, double *host_out, void *stream, const HostPublish *pub = nullptr) {
    if (!host_out) return EVK_OK;
    HostSlot *hs = host_slot();
    if (!hs) return EVK_EINVAL;
    if (pub) {
        const int nplanes = count / 4;
        for (uint64_t spins = 1;; ++spins) {
            bool all = true;
            for (int k = 0; k < nplanes; ++k) all = all && __atomic_load_n(&hs->flags[k], __ATOMIC_ACQUIRE) == pub->seq;
            if (all) break;
            if ((spins & 0x3FFF) == 0) {
                const hipError_t e = hipStreamQuery((hipStream_t)stream);
                if (e != hipSuccess && e != hipErrorNotReady) return (int)e;
                if (e == hipSuccess) {  // everything ran: the flags are there now, or the finalise kernel never was
                    bool done = true;
                    for (int k = 0; k < nplanes; ++k) done = done && __atomic_load_n(&hs->flags[k], __ATOMIC_ACQUIRE) == pub->seq;
                    if (!done) return EVK_EINVAL;
                    break;
                }
            }
            __builtin_ia32_pause();
        }
        for (int k = 0; k < count; ++k) host_out[k] = ((volatile double *)hs->vals)[k];  // ordered after the acquire of the flags
        return EVK_OK;
    }
    hipError_t e = hipMemcpyAsync(hs->vals, out, count * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    for (int k = 0; k < count; ++k) host_out[k] = hs->vals[k];
    return EVK_OK;
}

// post-pass + delivery: with host_out the finalise kernel publishes to the pinned slot when it can
static int post_and_fetch(int mode, const float *iwe, const float *diwe, int h, int w, const double *host_weights, int radius,
                          uint32_t flags, double *out, void *scratch, int64_t scratch_bytes, void *stream, int nplanes,
                          double *host_out) {
    HostPublish pub{nullptr, nullptr, 0u};
    HostSlot *hs = host_out ? host_slot() : nullptr;
    if (hs) {
        if (++hs->seq == 0) hs->seq = 1;
        pub = HostPublish{hs->vals, hs->flags, hs->seq};
    }
    bool published = false;
    const int rc = evk_post_variance_publish(mode, iwe, diwe, h, w, host_weights, radius, flags, out, scratch,
                                                  scratch_bytes, stream, nplanes, hs ? &pub : nullptr, &published);
    if (rc != EVK_OK) return rc;
    return fetch_results(out, 4 * nplanes, host_out, stream, published ? &pub : nullptr);
}

extern "C" int evk_cmax_variance_tiled_f32(const float *records, const uint32_t *bucket_start, int64_t n, int dom_h, int dom_w,
                                           int tw_log2, int th_log2, int slices, int win_w, int win_h, double t_first,
                                           double t_ref, double vx, double vy, double bounds_w, double bounds_h,
                                           int canvas_h, int canvas_w, uint32_t iwe_flags, double p_scale,
                                           double p_bound, double dt_bound, const double *host_weights, int radius, uint32_t post_flags, void *staging,
                                           int64_t staging_bytes, float *iwe_buf, double *out, void *scratch,
                                           int64_t scratch_bytes, float *spill_pair, int parity, double *host_out,
                                           void *stream) {
    if (!iwe_buf || canvas_h <= 1 || canvas_w <= 1) return EVK_EINVAL;
    const bool grad = iwe_flags & EVK_IWE_GRADIENT;
    const size_t plane = (size_t)canvas_h * canvas_w;
    int rc;
    float *diwe = grad ? iwe_buf + plane : nullptr;
    if (spill_pair) {
        const size_t img = (grad ? 3 : 1) * plane;
        rc = evk_iwe_tiled_spill(grad ? 1 : 0, records, bucket_start, n, dom_h, dom_w, tw_log2, th_log2, slices, win_w, win_h,
                                 t_first, t_ref, &vx, &vy, bounds_w, bounds_h, canvas_h, canvas_w, iwe_flags, p_scale, p_bound,
                                 dt_bound, staging, staging_bytes, iwe_buf, spill_pair + (parity & 1) * img,
                                 spill_pair + ((parity & 1) ^ 1) * img, stream);
    } else {
        hipError_t e = hipMemsetAsync(iwe_buf, 0, (grad ? 3 : 1) * plane * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
        rc = evk_iwe_linvel_tiled_f32(records, bucket_start, n, dom_h, dom_w, tw_log2, th_log2, slices, win_w, win_h,
                                      t_first, t_ref, vx, vy, bounds_w, bounds_h, canvas_h, canvas_w, iwe_flags, p_scale,
                                      p_bound, dt_bound, staging, staging_bytes, iwe_buf, diwe, stream);
    }
    if (rc != EVK_OK || (post_flags & EVK_POST_NONE)) return rc;
    const int mode = grad ? ((post_flags & EVK_POST_VALUE) ? 3 : 1) : 0;
    return post_and_fetch(mode, iwe_buf, diwe, canvas_h, canvas_w, host_weights, radius, post_flags & ~EVK_POST_VALUE, out,
                          scratch, scratch_bytes, stream, 1, host_out);
}

extern "C" int evk_cmax_variance_batch3_tiled_f32(const float *records, const uint32_t *bucket_index, int64_t n,
                                                  int dom_h, int dom_w, int tw_log2, int th_log2, int slices, int win_w,
                                                  int win_h, double t_first, double t_ref, const double *host_vx,
                                                  const double *host_vy, double bounds_w, double bounds_h,
                                                  int canvas_h, int canvas_w, uint32_t iwe_flags, double p_scale,
                                                  double p_bound, double dt_bound, const double *host_weights, int radius, void *staging,
                                                  int64_t staging_bytes, float *iwe3, double *out12, void *scratch,
                                                  int64_t scratch_bytes, float *spill_pair, int parity, double *host_out,
                                                  void *stream) {
    if (!iwe3 || canvas_h <= 1 || canvas_w <= 1 || !host_vx || !host_vy || (iwe_flags & EVK_IWE_GRADIENT)) return EVK_EINVAL;
    const size_t plane = (size_t)canvas_h * canvas_w;
    int rc;
    if (spill_pair) {
        rc = evk_iwe_tiled_spill(2, records, bucket_index, n, dom_h, dom_w, tw_log2, th_log2, slices, win_w, win_h, t_first,
                                 t_ref, host_vx, host_vy, bounds_w, bounds_h, canvas_h, canvas_w, iwe_flags, p_scale, p_bound,
                                 dt_bound, staging, staging_bytes, iwe3, spill_pair + (parity & 1) * 3 * plane,
                                 spill_pair + ((parity & 1) ^ 1) * 3 * plane, stream);
    } else {
        hipError_t e = hipMemsetAsync(iwe3, 0, 3 * plane * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
        rc = evk_iwe_linvel_tiled_batch3_f32(records, bucket_index, n, dom_h, dom_w, tw_log2, th_log2, slices, win_w,
                                             win_h, t_first, t_ref, host_vx, host_vy, bounds_w, bounds_h, canvas_h,
                                             canvas_w, iwe_flags, p_scale, p_bound, dt_bound, staging, staging_bytes, iwe3, stream);
    }
    if (rc != EVK_OK) return rc;
    return post_and_fetch(0, iwe3, nullptr, canvas_h, canvas_w, host_weights, radius, 0u, out12, scratch, scratch_bytes, stream,
                          3, host_out);
}
