/*
 * evk.h -- C ABI of libevk.so: MI355X (gfx950 / CDNA4) kernels for the data-parallel core of
 * TimoStoff/event_utils (event -> image / voxel scatter-add binning, linear-flow contrast maximisation).
 *
 * Boundary rules:
 *   - plain C: pointers, sizes, scalars.  No torch / C++ types.  `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).  Every entry point only ENQUEUES work on `stream` and returns.
 *   - every pointer is a DEVICE pointer unless the parameter is named host_*.
 *   - the library never allocates or frees caller-visible memory; outputs are ACCUMULATED INTO (the caller zero-
 *     or default-initialises them, as the reference does with its `default` image, image.py:77).
 *   - return value: 0 = ok, <0 = EVK_E* argument error, >0 = hipError_t from the launch.  Never throws / exits.
 *   - data-dependent errors the reference raises as Python exceptions (index out of range: image.py:30-36,96-99) are
 *     counted on the device in `oob` (uint32, caller-zeroed, may be NULL = unchecked); offending events are
 *     dropped; the host wrapper raises the reference's exception type when the counter is non-zero.
 *
 * Reference citations are file:line relative to the reference checkout.
 */
#ifndef EVK_H
#define EVK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVK_VERSION 100

#define EVK_OK 0
#define EVK_EINVAL (-1)   /* bad size / null pointer / unsupported parameter            */
#define EVK_ESCRATCH (-2) /* caller-provided scratch too small                           */
#define EVK_EALIGN (-3)   /* pointer not aligned as documented                            */
#define EVK_ECOMM (-4)    /* RCCL not available / collective failed                       */

/* flags for evk_iwe_* */
#define EVK_IWE_ABS_POLARITY 1u /* get_iwe(use_polarity=False): ps = |ps|  (objectives.py:184-185)   */
#define EVK_IWE_GRADIENT 2u     /* also accumulate dIWE/dparams (2 planes) (objectives.py:189-192)    */

int evk_version(void);
const char *evk_error_string(int code);

/* ------------------------------------------------------------------------------------------------------------
 * Event image, nearest pixel
 * ---------------------------------------------------------------------------------------------------------- */

/* events_to_image(..., interpolation=None), numpy path: np.ravel_multi_index + np.bincount on the (H+1, W+1)
 * canvas (image.py:28-38).  canvas[y, x] += w (w == NULL -> 1).  Integer accumulate => bit-exact.
 * Events outside [0,canvas_w) x [0,canvas_h) are counted in *oob (reference: ValueError, image.py:30-36). */
int evk_image_nearest_i32(const int32_t *x, const int32_t *y, const int32_t *w, int64_t n, int canvas_h,
                          int canvas_w, int32_t *canvas, uint32_t *oob, void *stream);

/* Same call site with floating-point weights (np.bincount(weights=ps) accumulates in float64, image.py:37). */
int evk_image_nearest_f64(const int32_t *x, const int32_t *y, const double *w, int64_t n, int canvas_h,
                          int canvas_w, double *canvas, uint32_t *oob, void *stream);

/* events_to_image_torch(..., interpolation=None) (image.py:87-99): coordinates truncated toward zero (.long()),
 * events with x >= clipx or y >= clipy are sent to pixel (0,0) WITH their weight (quirk Q8; pass +inf to disable
 * clipping = clip_out_of_range=False), negative indices wrap as torch's index_put_ does, anything still outside
 * the image is counted in *oob (reference: IndexError). img is (h, w) float32. */
int evk_image_nearest_f32(const float *x, const float *y, const float *w, int64_t n, int h, int wd, float clipx,
                          float clipy, float *img, uint32_t *oob, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Event image, bilinear 4-neighbour splat
 * ---------------------------------------------------------------------------------------------------------- */

/* events_to_image_torch(..., interpolation='bilinear') (image.py:79-86) + interpolate_to_image (image.py:102-115):
 * px=floor(x), dx=x-px, events with x >= clipx or y >= clipy get index (0,0) and weight 0;
 * img[py,px]+=w(1-dx)(1-dy), img[py,px+1]+=w dx(1-dy), img[py+1,px]+=w(1-dx)dy, img[py+1,px+1]+=w dx dy (f32). */
int evk_image_bilinear_f32(const float *x, const float *y, const float *w, int64_t n, int h, int wd, float clipx,
                           float clipy, float *img, uint32_t *oob, void *stream);

/* interpolate_to_image (image.py:102-115) on caller-computed integer pixels px, py (int64, torch .long()) and
 * fractions dx, dy: the four accumulates into img (h, wd).  Negative indices wrap, out-of-range counts in *oob. */
int evk_splat_indexed_f32(const int64_t *px, const int64_t *py, const float *dx, const float *dy, const float *w,
                          int64_t n, int h, int wd, float *img, uint32_t *oob, void *stream);

/* interpolate_to_derivative_img (image.py:117-136): d_img is (C, h, wd), w1 and w2 are (C, n) row-major float32. */
int evk_splat_drv_indexed_f32(const int64_t *px, const int64_t *py, const float *dx, const float *dy,
                              const float *w1, const float *w2, int C, int64_t n, int h, int wd, float *d_img,
                              uint32_t *oob, void *stream);

/* events_to_image_drv (image.py:162-217) for arbitrary per-event Jacobians: float64 inputs are cast to float32
 * BEFORE floor/frac (image.py:179-183), IWE splat as above, and if jx/jy != NULL the derivative splat of
 * interpolate_to_derivative_img (image.py:117-136) into d_img (2, h, wd) with w1 = jx*p*mask, w2 = jy*p*mask
 * (image.py:211-212).  jx, jy are (2, n) row-major float64. */
int evk_image_drv_f64(const double *x, const double *y, const double *p, const double *jx, const double *jy,
                      int64_t n, int h, int wd, float clipx, float clipy, float *img, float *d_img, uint32_t *oob,
                      void *stream);

/* image_to_event_weights (image.py:138-160): out[i] = bilinear interpolation of img (h, wd) float32 at (x[i], y[i])
 * (float64), 0 for events with x >= wd-1 or y >= h-1.  Used by get_iwe(return_per_event_contrast=True)
 * (objectives.py:196-198). */
int evk_image_gather_bilinear_f64(const double *x, const double *y, int64_t n, const float *img, int h, int wd,
                                  double *out, uint32_t *oob, void *stream);
/* the same for a float64 image (upstream takes any numpy array: img[...] * weights promotes to float64 either way) */
int evk_image_gather_bilinear_f64img(const double *x, const double *y, int64_t n, const double *img, int h, int wd,
                                     double *out, uint32_t *oob, void *stream);

/* events_to_timestamp_image[_torch] (image.py:219-353): out4 = (4, h, wd) float32 = [sum of normalised timestamps of
 * positive events, count of positive events, the same two for non-positive events], bilinear splat, accumulated into
 * (the caller initialises the count planes to ONE as upstream does, image.py:269,271).  Normalised timestamp
 * nts = (t - ta)/td (mode 0), (-t + ta)/td (mode 1, timestamp_reverse), t (mode 2), in float32. */
int evk_timestamp_images_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int h, int wd,
                             float clipx, float clipy, int mode, float ta, float td, float *out4, uint32_t *oob,
                             void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Voxel grid (temporal-bilinear, spatially nearest)
 * ---------------------------------------------------------------------------------------------------------- */

/* events_to_voxel_torch (voxel_grid.py:114-153): t_norm = (t - t_first)/(t_last - t_first)*(B-1) in float32,
 * bin weights p*max(0, 1-|t_norm-b|), one nearest-pixel accumulate per touched bin (image.py:87-95 with
 * clip_out_of_range=False).  ONE pass over the events (each event touches <= 2 bins) instead of the reference's B.
 * vox is (B, h, wd) float32.  t_first / t_last are ts[0] / ts[-1] (the caller reads them; they are also what every
 * rank of an event-sharded run must agree on). */
int evk_voxel_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, float t_first,
                  float t_last, int B, int h, int wd, float *vox, uint32_t *oob, void *stream);
/* the same with t_first = t[0], t_last = t[n-1] read by the kernel itself (voxel_grid.py:133-134 takes them from the same
 * column): a caller whose events are on the device needs no device-to-host transfer before the launch */
int evk_voxel_from_events_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int B, int h,
                              int wd, float *vox, uint32_t *oob, void *stream);

/* Windowed voxelisation: voxel_grids_fixed_n_torch / voxel_grids_fixed_t_torch (voxel_grid.py:37-80) call
 * events_to_voxel_torch once per window; this builds all nseg grids in ONE launch.  seg = nseg+1 device int64 offsets
 * (window s = events [seg[s], seg[s+1])), each window normalises time with its own first / last event
 * (voxel_grid.py:133-134).  vox is (nseg, B, h, wd) float32, accumulated into.  max_seg_len sizes the grid. */
int evk_voxel_segments_f32(const float *x, const float *y, const float *t, const float *p, const int64_t *seg,
                           int nseg, int64_t max_seg_len, int B, int h, int wd, float *vox, uint32_t *oob,
                           void *stream);

/* events_to_voxel (voxel_grid.py:184-217), numpy path: integer coordinates on the (h+1, wd+1) canvas of
 * events_to_image (x == wd / y == h are legal and cropped away, image.py:17,44), float64 arithmetic and output. */
int evk_voxel_f64(const int32_t *x, const int32_t *y, const double *t, const double *p, int64_t n, double t_first,
                  double t_last, int B, int h, int wd, double *vox, uint32_t *oob, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Contrast maximisation: warp, mask, IWE, blur, objective
 * ---------------------------------------------------------------------------------------------------------- */

/* linvel_warp.warp (warps.py:51-61): dt=t-t0; xo=x-dt*vx; yo=y-dt*vy; if jx/jy != NULL they are (2, n) float64:
 * jx=[-dt; 0], jy=[0; -dt]. */
int evk_warp_linvel_f64(const double *x, const double *y, const double *t, int64_t n, double t0, double vx,
                        double vy, double *xo, double *yo, double *jx, double *jy, void *stream);

/* warp_events_flow_torch (lib/transforms/optic_flow.py:5-46): per-pixel flow field (2, h, wd) float32 sampled bilinearly
 * at every event (grid_sample, align_corners=True, zero padding), xo = x + flow_x(x, y) * (t - t0), same for y. */
int evk_warp_flow_field_f32(const float *x, const float *y, const float *t, int64_t n, const float *flow, int h, int wd,
                            float t0, float *xo, float *yo, void *stream);

/* events_bounds_mask (event_util.py:15-28): mask = !(x<=xmin || x>xmax) * !(y<=ymin || y>ymax) as 0.0/1.0. */
int evk_bounds_mask_f64(const double *x, const double *y, int64_t n, double xmin, double xmax, double ymin,
                        double ymax, double *mask, void *stream);

/* get_iwe (objectives.py:165-199) fused for the linear-flow model: warp at t_ref (= ts[-1], :186) in float64 ->
 * events_bounds_mask(0, bounds_w, 0, bounds_h) (:187) -> multiply by mask (:188-190) -> cast to float32, inner mask
 * x >= canvas_w-1 / y >= canvas_h-1, floor/frac (image.py:179-205) -> IWE splat (image.py:102-115) and, with
 * EVK_IWE_GRADIENT, dIWE splat (image.py:117-136; for this model only w1[0] = w2[1] = -dt*p are non-zero).
 * iwe is (canvas_h, canvas_w) f32, diwe is (2, canvas_h, canvas_w) f32.  Per-event values are bit-identical to the
 * reference's; only the summation order differs.  p_scale multiplies the polarity in float64 before everything
 * else (1.0 normally; 100.0 on the adaptive-lifespan path, objectives.py:225).  Columns are float32 (evk_iwe_linvel_f32, 16 B/event) or float64
 * (evk_iwe_linvel_f64, 32 B/event, exact for arbitrary float64 inputs). */
int evk_iwe_linvel_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, double t_ref,
                       double vx, double vy, double bounds_w, double bounds_h, int canvas_h, int canvas_w,
                       uint32_t flags, double p_scale, float *iwe, float *diwe, void *stream);
int evk_iwe_linvel_f64(const double *x, const double *y, const double *t, const double *p, int64_t n, double t_ref,
                       double vx, double vy, double bounds_w, double bounds_h, int canvas_h, int canvas_w,
                       uint32_t flags, double p_scale, float *iwe, float *diwe, void *stream);

/* scipy.ndimage.gaussian_filter(a, sigma) (objectives.py:233,253), mode='reflect': separable correlation with the
 * symmetric kernel host_weights[0..2*radius] (float64, HOST pointer; computed by the caller exactly as scipy does),
 * filtering axes 0..ndim-1 in order, each pass evaluated in float64 and stored as float32.  ndim is 2 or 3; a 3-D
 * (2, H, W) input is also filtered across its first axis (quirk Q4).  src is not modified; tmp is a scratch array of
 * the same size; the result lands in dst.  src, dst, tmp must be distinct. */
int evk_gaussian_filter_f32(const float *src, float *dst, float *tmp, int ndim, const int *host_dims,
                            const double *host_weights, int radius, void *stream);

/* variance_objective.evaluate_function (objectives.py:234-236): out[0] = mean(img), out[1] = var(img - mean(img))
 * (population variance over all n pixels, float64 accumulation), out[2] = sum(img).
 * scratch: >= evk_reduce_scratch_bytes() bytes. out: 4 doubles on the device. */
int evk_variance_f32(const float *img, int64_t n, double *out, void *scratch, int64_t scratch_bytes, void *stream);

/* variance_objective.evaluate_gradient (objectives.py:256-264): for i in {0,1}
 * out[i] = mean( 2*(iwe - mean(iwe)) * diwe[i] ), out[2] = mean(iwe); the caller negates. diwe is (2, n). */
int evk_variance_grad_f32(const float *iwe, const float *diwe, int64_t n, double *out, void *scratch,
                          int64_t scratch_bytes, void *stream);

/* Fused objective post-pass (one launch + a 1-block finalise; the blurred images are never materialised):
 * evaluate_function (objectives.py:231-236): out as evk_variance_f32 but of gaussian_filter(iwe) (host_weights /
 * radius as in evk_gaussian_filter_f32; radius < 0 = no blur).  Bit-identical to the unfused sequence. */
int evk_objective_variance_f32(const float *iwe, int h, int w, const double *host_weights, int radius, double *out,
                               void *scratch, int64_t scratch_bytes, void *stream);

#define EVK_POST_MIX 1u      /* scipy's 3-D filter of the (2, H, W) dIWE also mixes the two channels (quirk Q4) */
#define EVK_POST_BLUR_IWE 2u /* use the blurred IWE in the gradient (reference_exact=False); default raw (Q5)    */
#define EVK_POST_VALUE 4u    /* evk_cmax_variance_tiled_f32 with EVK_IWE_GRADIENT: also the function value (out as
                                evk_objective_variance_fg_f32)                                                   */
#define EVK_POST_NONE 8u     /* evk_cmax_variance_tiled_f32: stop after the gather -- iwe_buf holds the raw IWE [, dIWE] of
                                this rank's events, ready for the all-reduce of an event-sharded run; `out` untouched  */
/* evaluate_gradient (objectives.py:252-264): out as evk_variance_grad_f32 with diwe replaced by its blurred version. */
int evk_objective_variance_grad_f32(const float *iwe, const float *diwe, int h, int w, const double *host_weights,
                                    int radius, uint32_t flags, double *out, void *scratch, int64_t scratch_bytes,
                                    void *stream);

/* evaluate_function AND evaluate_gradient of one parameter vector from one IWE / dIWE (what a BFGS line search asks for
 * at every trial point, events_cmax.py:345): out = [g0, g1, mean v, var v] with g as evk_objective_variance_grad_f32
 * (same flags) and v = gaussian_filter(iwe) as evk_objective_variance_f32; values identical to the two separate calls. */
int evk_objective_variance_fg_f32(const float *iwe, const float *diwe, int h, int w, const double *host_weights,
                                  int radius, uint32_t flags, double *out, void *scratch, int64_t scratch_bytes,
                                  void *stream);

/* ---- generic objective reductions: the objectives of objectives.py:266-596 other than the variance one differ only in
 * the scalar they take from the (blurred) IWE -----------------------------------------------------------------------
 * evk_objective_stats_f32: v = gaussian_filter(img) (radius < 0: v = img);
 *   out8 = [mean v, var v, sum v, sum v^2, sum exp(v), sum exp(-p v), count(v > thresh), max v]
 *   (sos :329, soe :373-374, moa :421, isoa :450, sosa :497-498, r1 :584-586; exponentials in float64). */
int evk_objective_stats_f32(const float *img, int h, int w, const double *host_weights, int radius, double p,
                            double thresh, double *out8, void *scratch, int64_t scratch_bytes, void *stream);

#define EVK_G_IDENT 0  /* g(a) = a                                    variance :257, sos :349                        */
#define EVK_G_EXP 1    /* g(a) = exp(a)                               soe :395                                        */
#define EVK_G_STEP 2   /* g(a) = a > gparam ? 1 : 0                   isoa :468-469                                   */
#define EVK_G_EXPNEG 3 /* g(a) = exp((double)(float)(-gparam * a))    sosa :513 (forms -p*iwe in float32 first)       */
/* evk_objective_gradsums_f32: d = gaussian_filter(diwe) (3-D with EVK_POST_MIX), a = iwe or, with EVK_POST_BLUR_IWE,
 * gaussian_filter(iwe);  out8 = [.., .., .., sum g(a), sum d0, sum d1, sum g(a) d0, sum g(a) d1] (out8[0..2] as
 * evk_objective_variance_grad_f32). */
int evk_objective_gradsums_f32(const float *iwe, const float *diwe, int h, int w, const double *host_weights, int radius,
                               uint32_t flags, int gfun, double gparam, double *out8, void *scratch,
                               int64_t scratch_bytes, void *stream);

int64_t evk_reduce_scratch_bytes(void);

/* ------------------------------------------------------------------------------------------------------------
 * Tile-bucketed path (the fast path).  Global float atomics sustain only ~21 G/s on MI355X, so
 * the hot configurations bucket the events by output tile once and accumulate per tile in LDS.
 * ---------------------------------------------------------------------------------------------------------- */

#define EVK_KEY_NEAREST 0     /* tile of (trunc(x), trunc(y)) with torch's negative wrap; out-of-domain events are
                                 dropped and counted in *oob (voxel_grid.py:140-142 -> image.py:87-99)            */
#define EVK_KEY_FLOOR_CLAMP 1 /* tile of (floor(x), floor(y)) clamped into the domain (IWE: the tile only seeds the
                                 LDS window, every event stays legal)                                             */

/* number of tiles of a (dom_h, dom_w) domain cut into 2^th_log2 x 2^tw_log2 tiles; <0 if unsupported (> 8192). */
int evk_bucket_num_tiles(int dom_h, int dom_w, int tw_log2, int th_log2);
int64_t evk_bucket_scratch_bytes(int ntiles);
/* length (uint32 entries) of the bucket index for n events: tile offsets (ntiles+1), work-item offsets (ntiles+1),
 * per-tile arrival counters (ntiles), the item -> tile map (evk_bucket_max_items entries), the `scene` word and, LAST, the
 * two words of EVK_STAGE_STATS (below); scene = index[len - 3].
 * A tile holding more than max(32768, 4n/ntiles) events is split into several work items, so clustered event data cannot
 * serialise on one CU.  When the fullest tile holds more than 1.25 x the mean tile population the scene word is 1 (a
 * structured scene; 0 otherwise) and the plan is BALANCED: the split threshold is lowered, not below
 * max(4096, 5/8 of the mean), as far as 2 work items per tile allow -- the tile kernels run one workgroup per item, all
 * resident at once, so a launch lasts as long as its largest item. */
int64_t evk_bucket_index_len(int ntiles, int64_t n);
int evk_bucket_max_items(int ntiles, int64_t n);

#define EVK_STAGE_HIST 1    /* per-block tile histograms (reads x, y)                                            */
#define EVK_STAGE_SCAN 2    /* prefix sums -> bucket index                                                       */
#define EVK_STAGE_SCATTER 4 /* write-combining scatter into records (reads x, y, t, p; needs stages 1|2 done)    */
#define EVK_STAGE_ALL 7
#define EVK_STAGE_SHARE_CU 8 /* modifier of the scatter stage: keep its LDS rings <= 96 KB so that workgroups of another,
                                concurrently running kernel (an overlapped RCCL collective) still fit on every CU    */
/* Round 6.  EVK_STAGE_STATS: the histogram pass also reads the polarity column and leaves, in the LAST TWO words of the
 * bucket index, [len - 2] = 0 when every event has a compact 8-byte record (integer pixel coordinates inside the domain, a
 * polarity without low mantissa bits), else 1, and -- filled by the scatter stage (the LDS-sorting one) -- [len - 1] = the
 * float32 bit pattern of max |p| over the bucketed events (0xFFFFFFFF when the ring scatter ran instead): what callers needed a pass over the records
 * (evk_compact_records_f32) and a reduction over p for.  EVK_STAGE_COMPACT (with STATS, IWE key,
 * tiles of <= 1024 pixels): when that verdict is 0 the scatter writes the COMPACT records (see evk_compact_records_f32:
 * same records, same order) into the first 8 n bytes of `records` instead of the 16-byte ones -- the caller learns which
 * from index[len - 2] (also 1 when the tiling forces the ring scatter, which writes 16-byte records only).  EVK_STAGE_LEGACY_SCATTER: the write-combining ring scatter of rounds 1-5 instead of the LDS-sorting
 * one (A/B, tests; both produce identical records). */
#define EVK_STAGE_STATS 16
#define EVK_STAGE_COMPACT 32
#define EVK_STAGE_LEGACY_SCATTER 64

/* Counting sort of the SoA columns by tile: records = n x (x, y, t, p) float4 (16 B, contiguous per tile, time order
 * preserved across the 256 partition blocks), bucket_index = evk_bucket_index_len(ntiles, n) uint32 (see above).
 * Columns and records must be 16-byte aligned (EVK_EALIGN otherwise); n < 2^32.  `stages` = EVK_STAGE_ALL normally;
 * the stages can be launched one by one (same arguments, same scratch) to time them separately. */
int evk_bucket_events_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int key_mode,
                          int dom_h, int dom_w, int tw_log2, int th_log2, float *records, uint32_t *bucket_index,
                          void *scratch, int64_t scratch_bytes, uint32_t *oob, int stages, void *stream);

/* ---- native on-disk dtypes --------------------------------------------------------------
 * The reference's event files hold xs, ys int16, ts float64, ps bool (HDF5, lib/data_formats/event_packagers.py:90-93)
 * or xy int16 (N, 2), t float64, p uint8 (memmap, lib/data_formats/h5_to_memmap.py:119-121); its loaders widen them on
 * the host (lib/data_loaders/memmap_dataset.py:19-24, hdf5_dataset.py:18-23: float32 coordinates, p * 2.0 - 1.0).
 * These two entry points take the columns as stored -- 13 B/event instead of 16 -- and widen in registers:
 *   x, y      int16; xy_stride 1 = two separate columns, 2 = one interleaved (N, 2) array passed as `x` (y ignored)
 *   t         EVK_T_F64 / EVK_T_F32; the value used is (float)((double)t - t_offset): pass t_offset = ts[0] so that
 *             epoch-scale float64 timestamps survive the narrowing ("t pre-offset by t[0] in f64")
 *   p         EVK_P_U8_PM1: uint8 / bool {0, 1} -> 2p - 1;  EVK_P_U8: uint8 as is;  EVK_P_I8: int8 as is
 * Everything downstream (records, tile kernels) is identical to the float32 path on the widened columns. */
#define EVK_T_F32 0
#define EVK_T_F64 1
#define EVK_P_U8_PM1 0
#define EVK_P_U8 1
#define EVK_P_I8 2

/* evk_bucket_events_f32 reading native columns (each 16-byte aligned, EVK_EALIGN otherwise). */
int evk_bucket_events_native_f32(const int16_t *x, const int16_t *y, int xy_stride, const void *t, int t_kind,
                                 double t_offset, const void *p, int p_kind, int64_t n, int key_mode, int dom_h,
                                 int dom_w, int tw_log2, int th_log2, float *records, uint32_t *bucket_index,
                                 void *scratch, int64_t scratch_bytes, uint32_t *oob, int stages, void *stream);

/* Native columns -> the four float32 SoA columns every other entry point takes (any alignment); replaces the host
 * casts of memmap_dataset.py:21-23 / hdf5_dataset.py:19-22. */
int evk_native_to_columns_f32(const int16_t *x, const int16_t *y, int xy_stride, const void *t, int t_kind,
                              double t_offset, const void *p, int p_kind, int64_t n, float *out_x, float *out_y,
                              float *out_t, float *out_p, void *stream);

/* events_to_voxel_torch on bucketed records (EVK_KEY_NEAREST over the (h, wd) image; n = the event count that was
 * bucketed): one workgroup per work item, LDS accumulators (B x tile, float64), exclusive plain-store flush: vox += tile,
 * or vox = tile with EVK_VOXEL_OVERWRITE (the caller then needs no memset: every cell is written).  The parts of a
 * split tile meet in `staging` (evk_voxel_tiled_staging_bytes) and the last one to arrive sums them in part order.
 * Same per-event arithmetic as evk_voxel_f32.
 * EVK_VOXEL_SPLIT_POLARITY: events_to_neg_pos_voxel_torch (voxel_grid.py:155-182) in ONE pass instead of two:
 * vox = (2, B, h, wd), [0] = grid of the events with p > 0, [1] = of those with p <= 0, each event with weight 1
 * (2B accumulator planes: pass 2B to evk_voxel_tiled_staging_bytes). */
#define EVK_VOXEL_OVERWRITE 1
#define EVK_VOXEL_SPLIT_POLARITY 2
int64_t evk_voxel_tiled_staging_bytes(int ntiles, int64_t n, int B, int tw_log2, int th_log2);
int evk_voxel_tiled_f32(const float *records, uint32_t *bucket_index, int64_t n, int h, int wd, int tw_log2, int th_log2,
                        float t_first, float t_last, int B, int flags, float *vox, void *staging,
                        int64_t staging_bytes, void *stream);

/* ---- one-pass voxel path (evk_voxel2.hip; DESIGN.md section 3, K1') ---------------------------------------------
 * events_to_voxel_torch (voxel_grid.py:114-153) straight from the event columns in TWO launches: a partition kernel
 * that sorts sub-chunks of <= 16 K events by tile in LDS and writes them back as contiguous runs of 8-byte records
 * (float32 t | 21-bit polarity | pixel in tile; polarities that need more bits go, exactly, to a side array) with a
 * (tile, sub-chunk) table, and a tile kernel that pulls every tile's segments and accumulates in LDS (float64) --
 * the events are read once and written once (24 B/event + the grid instead of 56).  Same per-event arithmetic and
 * results as evk_voxel_f32 / evk_voxel_tiled_f32 -- with one exception: a call all of whose polarities are +1, -1 or +0
 * (and whose accumulators fit, see DESIGN.md K2') is accumulated as an integer count and one int64 fixed-point sum per bin,
 * grid[b] = S0[b] - G[b] + G[b-1], i.e. without rounding p (1 - f) to float32 per event (differences < 6e-8 per event,
 * the parity bar is 1e-5 of the grid's maximum), and such a grid does not depend on the order of the events;
 * EVK_VOXEL2_NO_COUNT keeps the two float64 atomics.
 *   index    evk_voxel2_index_len(ntiles, n) uint32, ZEROED ONCE by the caller when it is allocated; the library
 *            leaves its counters at zero after every call (persistent across calls on one stream)
 *   scratch  evk_voxel2_scratch_bytes(...) bytes, 16-byte aligned, uninitialised
 *   host_report  optional: two uint32, 8-byte aligned, in PINNED host memory the device can write (hipHostMalloc).  As soon as
 *            every partition workgroup has counted the dropped events of its LAST sub-chunk (round 5: ~8 us before the kernel
 *            ends; the bilinear image format: when its last workgroup finishes) the kernel stores {seq, *oob} there (one 8-byte system-scope store: the pair
 *            is never seen torn; EVK_EALIGN if the pointer is not 8-byte aligned), so a caller that checks for
 *            dropped events lazily needs neither a device-to-host copy nor an event on the stream: the call numbered
 *            `seq` has counted all its events once host_report[0] == seq (sequence numbers grow by one per call).
 *            Independent of `oob`: without a counter the report is {seq, 0}.
 *   flags    EVK_VOXEL_OVERWRITE, EVK_VOXEL_SPLIT_POLARITY as evk_voxel_tiled_f32;
 *            EVK_VOXEL_T_FROM_EVENTS: t_first / t_last are read on the device from t[0] / t[n-1] (voxel_grid.py:133-134
 *            takes them from the same column), so the caller needs no device-to-host transfer before the launch;
 *            EVK_VOXEL2_PARTITION_ONLY / EVK_VOXEL2_TILES_ONLY: launch one of the two kernels (timing).
 *            EVK_VOXEL_DETERMINISTIC: the tile kernel accumulates int64 multiples of 2^-32 instead of float64 (integer
 *            adds commute: the grid is bit-identical from run to run and for any order of the events); a NaN t_norm
 *            (t_last == t_first, Q9) marks its cell NaN in every bin as in the reference; every other contribution must be
 *            finite and below 2^30, anything else is counted in index[4] (the caller reads and clears it) and left out.
 * Tiles: tile_w x tile_h PIXELS, any size with (tile_w | 1) * tile_h <= 1024 (evk_voxel2_num_tiles() > 0), at most
 * evk_voxel2_max_tiles() of them.  The tile kernel runs one workgroup per tile, all resident at once, so a tile COUNT
 * that is a multiple of the 256 CUs (640x480: 512 tiles of 20x30) keeps every CU equally busy. */
#define EVK_VOXEL_T_FROM_EVENTS 4
#define EVK_VOXEL_DETERMINISTIC 256
#define EVK_VOXEL2_PARTITION_ONLY 16
#define EVK_VOXEL2_TILES_ONLY 32
#define EVK_VOXEL2_SHARE_CU 128    /* partition with 64 KB of LDS per CU instead of 128 KB, so that workgroups of another,
                                      concurrently running kernel (an overlapped RCCL collective) still fit on every CU */
#define EVK_VOXEL2_NO_XCD_ORDER 64 /* A/B switch: tile kernel work items in plain order instead of one contiguous range per XCD */
/* The library picks the record size (4 bytes above 16 M events, else 8), the unit-polarity counting mode and the tile
 * workgroup size (768 threads where two such workgroups fit a CU) by itself; these flags force the other choice so that tests
 * and measurements can run every shipped kernel shape at any size (results are the same within the parity bar). */
#define EVK_VOXEL2_REC4 1024
#define EVK_VOXEL2_REC8 2048
#define EVK_VOXEL2_NO_COUNT 4096
#define EVK_VOXEL2_WG512 8192
#define EVK_VOXEL2_NO_COUNT2 32768   /* (A/B, tests) where the counting mode's planes do not fit (1280x720 tiles) unit polarities are
                                        still counted in integers, two int64 LDS atomics per event in the float64 mode's planes
                                        (same bits as the counting mode); this flag keeps the float64 atomics there */
/* EVK_VOXEL2_LIVE (round 5): the tiles are accumulated WHILE the partition sorts -- a consumer kernel on a second stream of the
 * library's own (one per device, created on first use) takes every run as soon as it is written, two tiles per workgroup, one
 * workgroup beside the partition's on every CU; the tile kernel proper still follows on `stream` and accumulates only what
 * the consumers leave (polarities other than +1 / -1 / +0, hot tiles, rounds that did not arrive in time), so the call is
 * complete in `stream`'s order exactly as without the flag (round 6: `stream` also WAITS for the consumer kernel's end behind
 * the tile kernel -- an event on the side stream -- so index / scratch / the columns may be freed or rewritten by anything
 * ordered after the call on `stream`, as for every other entry point), and the grid is bit-identical to the counting mode's.  A request:
 * calls the consumer kernel is not written for (more than 512 tiles, accumulators beyond 52 KB per pair of tiles, fewer than
 * two sub-chunks per partition workgroup, 4-byte records, split polarities, EVK_VOXEL2_SHARE_CU, single stages, a stream that
 * is being captured into a graph) run as without it.  evk_voxel2_f32 only. */
#define EVK_VOXEL2_LIVE 16384
/* EVK_COLUMNS_UNALIGNED (round 6; evk_voxel2_f32, evk_image2_nearest_f32 / _bilinear_f32, evk_image2_splat_indexed_f32,
 * evk_image2_splat_drv_indexed_f32 (pixels and fractions), evk_timestamp_images2_f32): the event columns need only the alignment of their elements (4 bytes; 8 for int64 pixels) instead
 * of 16 bytes -- a device SLICE xs[a:b] of a resident stream is read where it lies.  The kernels load 16 bytes at a time and a
 * group of four events that is only partly inside the stream is loaded whole: the caller guarantees that the 12 bytes (24 for
 * int64) behind every column's last event are readable (inside the same allocation).  Without the flag a column that is not
 * 16-byte aligned is refused with EVK_EALIGN, as before. */
#define EVK_COLUMNS_UNALIGNED 65536
int evk_voxel2_max_tiles(void);
int64_t evk_voxel2_index_len(int ntiles, int64_t n);
int evk_voxel2_num_tiles(int h, int wd, int tile_w, int tile_h);   /* 0 = this tiling is not supported */
/* 1 when evk_voxel2_f32 takes this tiling with `planes` accumulator planes (B; 2 B with EVK_VOXEL_SPLIT_POLARITY; 1 for the
 * event images): grid, partition LDS, tile-kernel LDS and tile count all fit.  evk_num_cu(): the CU count the tile count
 * should be a multiple of (256 on MI355X). */
int evk_voxel2_fits(int h, int wd, int tile_w, int tile_h, int planes);
int evk_num_cu(void);
int64_t evk_voxel2_scratch_bytes(int ntiles, int64_t n, int planes, int tile_w, int tile_h);
int evk_voxel2_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int h, int wd,
                   int tile_w, int tile_h, float t_first, float t_last, int B, int flags, float *vox,
                   uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report,
                   uint32_t seq, void *stream);
/* ROW BANDS for event-sharded runs: after an evk_voxel2_f32 call with EVK_VOXEL2_PARTITION_ONLY (same n, h,
 * wd, tile size, B, flags, index, scratch), accumulate only the tile rows [tile_row_lo, tile_row_hi) and WRITE them to `band`,
 * a contiguous (B, rows, wd) float32 buffer (rows = min(tile_row_hi * tile_h, h) - tile_row_lo * tile_h; 2B planes with
 * EVK_VOXEL_SPLIT_POLARITY).  A sharded caller launches the bands one after the other and all-reduces band k (one contiguous
 * buffer) while band k + 1 is being accumulated; event_utils_amd/distributed.py does. */
int evk_voxel2_band_f32(int64_t n, int h, int wd, int tile_w, int tile_h, int B, int flags, int tile_row_lo, int tile_row_hi,
                        float *band, uint32_t *index, void *scratch, int64_t scratch_bytes, void *stream);
/* out[i] = (t[i] - t_first) / (t_last - t_first) * (B - 1) in float32: the normalised time of voxel_grid.py:134, by the very
 * function the partition kernel calls -- bit-identical to numpy's float32 arithmetic; the tests compare it bit for bit. */
int evk_normalise_time_f32(const float *t, int64_t n, float t_first, float t_last, int B, float *out, void *stream);
/* the same from the reference's on-disk dtypes (see evk_bucket_events_native_f32) */
int evk_voxel2_native_f32(const int16_t *x, const int16_t *y, int xy_stride, const void *t, int t_kind, double t_offset,
                          const void *p, int p_kind, int64_t n, int h, int wd, int tile_w, int tile_h, float t_first,
                          float t_last, int B, int flags, float *vox, uint32_t *index, void *scratch,
                          int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report, uint32_t seq, void *stream);

/* ---- event images on the one-pass partition (evk_image2.hip) -------------------------------
 * The three event-image entry points above (evk_image_nearest_i32 / _f32, evk_image_bilinear_f32: one global atomic per
 * contribution, ~21 G atomics/s) on the voxel grid's partition + LDS-tile design: the same results -- the integer image
 * bit for bit, the float32 images up to the order of the additions -- from two launches that read 12 B/event and move a
 * 4-byte (nearest) or 12-byte (bilinear) record once.  Semantics, clip thresholds, the treatment of negative /
 * out-of-range / non-finite coordinates and the meaning of *oob are those of the corresponding direct entry point
 * (the partition kernel hands the events an LDS tile cannot take to the direct kernel's own code).
 *   tile_w, tile_h   as evk_voxel2_f32 (evk_voxel2_num_tiles(h, wd, tile_w, tile_h) > 0)
 *   index            evk_voxel2_index_len(ntiles, n) uint32, zeroed once by the caller, persistent (as evk_voxel2_f32;
 *                    the two paths may share one index on a stream)
 *   scratch          evk_image2_scratch_bytes(ntiles, n, tile_w, tile_h) bytes, 16-byte aligned
 *   host_report/seq  as evk_voxel2_f32
 *   flags            EVK_VOXEL_OVERWRITE (nearest only): every pixel of the image is WRITTEN, the caller needs no memset;
 *                    without it the result is added to the image (the reference's `default` image, image.py:77).  The
 *                    bilinear call always adds.  EVK_VOXEL2_PARTITION_ONLY / _TILES_ONLY / _NO_XCD_ORDER as evk_voxel2_f32.
 *                    EVK_IMAGE2_NO_FIXED: bilinear windows in float64 even when every weight is +1, -1 or +0 (A/B).
 * Columns 16-byte aligned (EVK_EALIGN otherwise).  evk_image2_nearest_i32: w == NULL counts the events (weight 1). */
#define EVK_IMAGE2_NO_FIXED 512
int64_t evk_image2_scratch_bytes(int ntiles, int64_t n, int tile_w, int tile_h);
int evk_image2_nearest_i32(const int32_t *x, const int32_t *y, const int32_t *w, int64_t n, int canvas_h, int canvas_w,
                           int tile_w, int tile_h, int flags, int32_t *canvas, uint32_t *index, void *scratch,
                           int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report, uint32_t seq, void *stream);
int evk_image2_nearest_f32(const float *x, const float *y, const float *w, int64_t n, int h, int wd, float clipx,
                           float clipy, int tile_w, int tile_h, int flags, float *img, uint32_t *index, void *scratch,
                           int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report, uint32_t seq, void *stream);
int evk_image2_bilinear_f32(const float *x, const float *y, const float *w, int64_t n, int h, int wd, float clipx,
                            float clipy, int tile_w, int tile_h, int flags, float *img, uint32_t *index, void *scratch,
                            int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report, uint32_t seq, void *stream);
/* evk_splat_indexed_f32 (above; interpolate_to_image on caller-computed pixels and fractions, image.py:102-115) on the same design
 * (round 6): an event whose px + dx, py + dy are float32 values with floor() == px, py -- every event whose pixel and fraction
 * were taken from a float32 coordinate, as upstream's callers do -- travels as a bilinear record; any other (a sum that rounds,
 * fractions outside [0, 1), NaN, pixels that wrap or raise) is re-read by its index and takes the direct kernel's code inside
 * the partition kernel.  Same arguments and *oob as evk_splat_indexed_f32, the rest as evk_image2_bilinear_f32 (always adds). */
int evk_image2_splat_indexed_f32(const int64_t *px, const int64_t *py, const float *dx, const float *dy, const float *w, int64_t n,
                                 int h, int wd, int tile_w, int tile_h, int flags, float *img, uint32_t *index, void *scratch,
                                 int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report, uint32_t seq, void *stream);
/* The derivative splats on the same design (round 6): evk_splat_drv_indexed_f32 (interpolate_to_derivative_img, image.py:117-136;
 * two channels) and evk_image_drv_f64 (events_to_image_drv, image.py:162-217).  An event's record carries its coordinates
 * relative to the tile and its INDEX in the stream; the tile kernel fetches the event's four or five weights from the caller's
 * columns by that index (they do not fit the partition's LDS) and accumulates the image and / or its two derivative planes in LDS
 * windows.  Arguments and *oob as the direct entry points', the rest as evk_image2_bilinear_f32 (always add; n < 2^32);
 * evk_image2_drv_f64: x, y 16-byte aligned, jx == jy == NULL for the image alone.  `scratch` of these two and of
 * evk_image2_splat_indexed_f32: evk_image2_indexed_scratch_bytes (8 K-event sub-chunks whatever the tile count). */
int64_t evk_image2_indexed_scratch_bytes(int ntiles, int64_t n, int tile_w, int tile_h);
int evk_image2_splat_drv_indexed_f32(const int64_t *px, const int64_t *py, const float *dx, const float *dy, const float *w1,
                                     const float *w2, int64_t n, int h, int wd, int tile_w, int tile_h, int flags, float *d_img,
                                     uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report,
                                     uint32_t seq, void *stream);
int evk_image2_drv_f64(const double *x, const double *y, const double *p, const double *jx, const double *jy, int64_t n, int h,
                       int wd, float clipx, float clipy, int tile_w, int tile_h, int flags, float *img, float *d_img,
                       uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report, uint32_t seq,
                       void *stream);
/* evk_timestamp_images_f32 (above; events_to_timestamp_image[_torch], image.py:219-353: eight global atomics per event) on
 * the same design (round 6): one partition -- 16 B/event read, a 12-byte record {x, y relative to the tile; the event's
 * class in their sign bits; its normalised time stamp} moved once -- and a tile kernel with four LDS windows per tile
 * (time / count of the positive and of the non-positive events).  Arguments x .. td and the meaning of *oob as
 * evk_timestamp_images_f32, including the upstream quirk (a clipped event lands on pixel (0, 0) with its weights intact);
 * the rest as evk_image2_bilinear_f32, with evk_timestamp_images2_scratch_bytes for `scratch`.  out4 = (4, h, wd), ADDED to. */
int64_t evk_timestamp_images2_scratch_bytes(int ntiles, int64_t n, int tile_w, int tile_h);
int evk_timestamp_images2_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int h, int wd,
                              float clipx, float clipy, int mode, float ta, float td, int tile_w, int tile_h, int flags,
                              float *out4, uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob,
                              uint32_t *host_report, uint32_t seq, void *stream);

/* get_iwe (linear flow) on bucketed records (EVK_KEY_FLOOR_CLAMP over a (dom_h, dom_w) domain covering the events):
 * one workgroup per (work item, time slice) accumulates a (win_h x win_w) LDS window (tile + flow halo, origin shifted
 * by the slice's displacement), stores it to `staging`; a gather kernel adds the windows covering each canvas pixel to
 * iwe / diwe.  Events falling outside their window use a global atomic (correct for any flow; slices / win_* only
 * tune speed).  t_first = earliest event time (bounds the window shifts).  Same per-event arithmetic as
 * evk_iwe_linvel_f32.
 * p_bound / dt_bound: upper bounds of |p * p_scale| and of |t - t_ref| over the events (0 / negative = unknown).  With
 * bounds the LDS windows accumulate in FIXED POINT -- this kernel is LDS-atomic bound, ds_add_u64 is 1.5x faster than
 * ds_add_f64 on gfx950, and integer sums are order-independent (bit-reproducible):
 *   - 64-bit cells, value * 2^k with k = min(40, 61 - ceil(log2(n * p_bound * max(1, dt_bound)))), used when k >= 26;
 * Quantisation <= 2^-(k+1) per contribution for the 64-bit cells.  Without bounds: float64 accumulation. */
/* COMPACT RECORDS.  The evaluation kernels stream the bucketed records once per evaluation, and an optimisation evaluates
 * ~10^2 times, so the record stream is what the function evaluation is bound by.  Sensor events have integer pixel
 * coordinates and +-1 polarities: evk_compact_records_f32 rewrites the (x, y, t, p) float32 records of a bucketing
 * (EVK_KEY_FLOOR_CLAMP, tiles of <= 1024 pixels) as 8-byte records {t, polarity bits [31:11] | pixel in tile [9:0]} in the
 * same order, and ORs 1 into *not_compact if some record does not survive that exactly (non-integer or out-of-domain
 * x / y, a polarity with any of its low 11 mantissa bits set).  When *not_compact stays 0 the tiled IWE entry points take
 * the compact buffer as `records` with EVK_IWE_COMPACT in `flags`: x, y are rebuilt from the tile origin, every
 * result is bit-identical to the 16-byte records'.  `compact` holds evk_compact_records_bytes(n) bytes, 16-byte aligned. */
#define EVK_IWE_COMPACT 16u
int64_t evk_compact_records_bytes(int64_t n);
int evk_compact_records_f32(const float *records, int64_t n, int dom_h, int dom_w, int tw_log2, int th_log2,
                            void *compact, uint32_t *not_compact, void *stream);
int64_t evk_iwe_tiled_staging_bytes(int ntiles, int64_t n, int slices, int planes, int win_w, int win_h);
int evk_iwe_linvel_tiled_f32(const float *records, const uint32_t *bucket_index, int64_t n, int dom_h, int dom_w,
                             int tw_log2, int th_log2, int slices, int win_w, int win_h, double t_first, double t_ref,
                             double vx, double vy, double bounds_w, double bounds_h, int canvas_h, int canvas_w,
                             uint32_t flags, double p_scale, double p_bound, double dt_bound, void *staging, int64_t staging_bytes,
                             float *iwe, float *diwe, void *stream);

/* variance_objective.evaluate_function / evaluate_gradient (objectives.py:211-264) in ONE call on bucketed records:
 * memset(iwe_buf) -> evk_iwe_linvel_tiled_f32 -> evk_objective_variance[_grad]_f32.  iwe_buf is (1, ch, cw) or, with
 * EVK_IWE_GRADIENT, (3, ch, cw) float32 = IWE followed by the two dIWE planes (left filled, un-blurred).
 * out: as evk_objective_variance_f32 / evk_objective_variance_grad_f32 / (post_flags & EVK_POST_VALUE)
 * evk_objective_variance_fg_f32.
 * spill_pair (optional): 2 x (1 | 3, ch, cw) float32, ZEROED ONCE by the caller and kept between calls, with `parity`
 * alternating 0 / 1 from call to call on one stream.  The evaluation then needs no memset: the rare events outside their
 * LDS window go to spill[parity], the gather writes iwe_buf = spill[parity] + windows and zeroes what the previous call
 * left in spill[parity ^ 1].
 * host_out (optional, HOST pointer to 4 doubles): the call also delivers `out` there and returns only when it has
 * (the finalise kernel stores the results and a sequence number in a pinned slot that the call polls: no copy command, no
 * stream synchronisation).  Everything enqueued on the stream before the call has completed when it returns. */
int evk_cmax_variance_tiled_f32(const float *records, const uint32_t *bucket_index, int64_t n, int dom_h, int dom_w,
                                int tw_log2, int th_log2, int slices, int win_w, int win_h, double t_first, double t_ref, double vx,
                                double vy, double bounds_w, double bounds_h, int canvas_h, int canvas_w,
                                uint32_t iwe_flags, double p_scale, double p_bound, double dt_bound, const double *host_weights,
                                int radius, uint32_t post_flags, void *staging, int64_t staging_bytes, float *iwe_buf,
                                double *out, void *scratch, int64_t scratch_bytes, float *spill_pair, int parity,
                                double *host_out, void *stream);

/* ---- batched evaluation: three NEARBY flows in one pass over the events ----------------
 * The reference's default optimiser path (numeric_grads=True, events_cmax.py:343) lets scipy estimate the gradient by
 * forward differences with epsilon = 1: f(v), f(v + e1), f(v + e2) = three full get_iwe passes.  These entry points
 * read every event once and accumulate the three IWEs side by side (iwe3 = (3, ch, cw)); host_vx / host_vy are 3
 * doubles each (host pointers).  The flows must be close (they share one LDS window per workgroup; an event outside it
 * still lands through a global atomic, so distant flows are merely slower). */
int evk_iwe_linvel_tiled_batch3_f32(const float *records, const uint32_t *bucket_index, int64_t n, int dom_h, int dom_w,
                                    int tw_log2, int th_log2, int slices, int win_w, int win_h, double t_first,
                                    double t_ref, const double *host_vx, const double *host_vy, double bounds_w,
                                    double bounds_h, int canvas_h, int canvas_w, uint32_t flags, double p_scale,
                                    double p_bound, double dt_bound, void *staging, int64_t staging_bytes, float *iwe3,
                                    void *stream);
/* evk_objective_variance_f32 for nplanes stacked images: out = nplanes x 4 doubles. */
int evk_objective_variance_planes_f32(const float *imgs, int nplanes, int h, int w, const double *host_weights,
                                      int radius, double *out, void *scratch, int64_t scratch_bytes, void *stream);

/* Row-sharded post-pass (multi-GPU option).  img = (1 | 3, hb, w) float32: the rows of ONE row block of the (ch, cw)
 * image -- plane 0 the IWE, planes 1, 2 the dIWE for mode 1 / 3 -- including `radius` halo rows on every side that is not
 * an edge of the whole image, already summed over the ranks.  The block is blurred with the 'reflect' rule at its own
 * edges (exact: an edge of the buffer is either an edge of the image or lies beyond the halo) and the pixels of rows
 * [y_lo, y_hi) of the buffer are added up.  mode 0: sums = [S v, S v^2]; mode 1: [S a, S d0, S d1, S a d0, S a d1];
 * mode 3: those five, then S v, S v^2 of the blurred IWE (flags as evk_objective_variance_grad_f32).  `sums` = 8 doubles
 * (device); all-reduce them over the ranks and finalise: mean = S0 / N, var = S1 / N - mean^2,
 * g_i = 2 / N (S(3+i) - (S0 / N) S(1+i)) with N = ch * cw. */
int evk_objective_variance_rows_f32(const float *img, int mode, int hb, int w, int y_lo, int y_hi,
                                    const double *host_weights, int radius, uint32_t flags, double *sums, void *scratch,
                                    int64_t scratch_bytes, void *stream);
/* memset -> batch3 IWE -> gather -> fused blur + variance of each plane: out12 = 3 x 4 doubles. */
int evk_cmax_variance_batch3_tiled_f32(const float *records, const uint32_t *bucket_index, int64_t n, int dom_h,
                                       int dom_w, int tw_log2, int th_log2, int slices, int win_w, int win_h,
                                       double t_first, double t_ref, const double *host_vx, const double *host_vy,
                                       double bounds_w, double bounds_h, int canvas_h, int canvas_w, uint32_t iwe_flags,
                                       double p_scale, double p_bound, double dt_bound, const double *host_weights, int radius,
                                       void *staging, int64_t staging_bytes, float *iwe3, double *out12, void *scratch,
                                       int64_t scratch_bytes, float *spill_pair, int parity, double *host_out,
                                       void *stream);

/* ---- the whole optimisation in one call (round 6; evk_optim.hip) -------------------------------------------------
 * optimize_contrast for the linear-flow warp and the variance objective (events_cmax.py:313-346, where scipy's fmin_bfgs
 * drives one Python callback per evaluation): the quasi-Newton iteration of event_utils_amd.contrast_max.evk_bfgs with its
 * arithmetic, the LDS window of every pass and the result poll inside the library.  Every pass is one
 * evk_cmax_variance_tiled_f32 (value + gradient at one flow; EVK_POST_VALUE is added to post_flags) or one
 * evk_cmax_variance_batch3_tiled_f32 (three step lengths of the line search); the arguments up to `spill_pair` mean what
 * they mean there (iwe_flags WITHOUT EVK_IWE_GRADIENT; iwe_buf: 3 planes; out12: 12 doubles, device; spill_pair required).
 * Time slices and window are chosen per pass from the flow; `staging_bytes` is the capacity the loop may use
 * (evk_iwe_tiled_staging_bytes of the largest plan the caller wants it to take).
 * parity (host, in/out): the spill parity of the LAST successful evaluation on this pair; updated.
 * x0: 2 doubles.  opts: 6 doubles [xtol (px/s), gtol, ftol, maxiter, numeric_grads (0 | 1: forward differences with
 * epsilon = 1 from one three-flow pass, the reference's default), unit_first (0 | 1)].
 * result (host): 6 + 5 * trace_cap doubles: [x0, x1, f(x), accepted points, event passes, status], then one row
 * [x0, x1, f, g0, g1] per accepted point (the first trace_cap of them).  status 0: finished (converged or maxiter);
 * 1: a trial flow needs a plan this call cannot take (more than 64 time slices, more than 128 candidate windows, more
 * staging than staging_bytes, a non-finite flow) -- x is the last accepted point, the caller continues with its own loop.
 * Returns when the last pass has delivered its results (synchronous, like the evaluation calls with host_out). */
int evk_cmax_bfgs_variance_tiled_f32(const float *records, const uint32_t *bucket_index, int64_t n, int dom_h, int dom_w,
                                     int tw_log2, int th_log2, double t_first, double t_ref, double bounds_w, double bounds_h,
                                     int canvas_h, int canvas_w, uint32_t iwe_flags, double p_scale, double p_bound,
                                     double dt_bound, const double *host_weights, int radius, uint32_t post_flags,
                                     void *staging, int64_t staging_bytes, float *iwe_buf, double *out12, void *scratch,
                                     int64_t scratch_bytes, float *spill_pair, int *parity, const double *x0,
                                     const double *opts, double *result, int trace_cap, void *stream);
/* The same iteration on an evaluator of the caller's (any two-parameter objective; no GPU work of its own -- a host function):
 * fg(user, x[2], &f, g[2]) = value and gradient at x; f3(user, pts[6] = {x0a, x1a, x0b, x1b, x0c, x1c}, fs[3]) = the values at
 * three points (one pass over the events when the objective can do that).  Both return 0, or 1 to decline the point (the
 * run ends with status 1, as above), or any other code, which is returned as it is.  fg may be NULL when opts[4]
 * (numeric_grads) is set: value and gradient then come from f3 at x, x + e0, x + e1.  opts / result / trace_cap as above. */
typedef int (*evk_bfgs2_fg_fn)(void *user, const double *x, double *f, double *g);
typedef int (*evk_bfgs2_f3_fn)(void *user, const double *pts, double *fs);
int evk_bfgs2_minimize(evk_bfgs2_fg_fn fg, evk_bfgs2_f3_fn f3, void *user, const double *x0, const double *opts,
                       double *result, int trace_cap);

/* Largest singular value SQUARED of a float32 (h, w) image, float64 -> out[0]: what the "rms" objective needs
 * (objectives.py:282: np.linalg.norm(iwe, 2) of a 2-D array is the spectral norm).  Lanczos on the Gram operator with full
 * re-orthogonalisation in one workgroup, then a bisection; h, w <= 4096.  scratch: evk_spectral_scratch_bytes(h, w).
 * Round 6: `out` holds TWO doubles.  The Ritz pair's true residual is measured after every sweep of <= 96 steps and the
 * iteration restarts from the Ritz vector until it is below 1e-10 of the value (at most 8 sweeps); out[1] = the final relative
 * residual, a bound on the relative distance of out[0] (a lower bound of sigma_max^2) from an eigenvalue -- a caller that needs
 * more than that checks it. */
int64_t evk_spectral_scratch_bytes(int h, int w);
int evk_spectral_norm_sq_f32(const float *img, int h, int w, double *out, void *scratch, int64_t scratch_bytes, void *stream);

/* ---- the stateful image classes of image.py:355-396 (float64 images, as the reference's numpy arrays) ----------
 * TimestampImage.add_events (image.py:366-368): image[int(y), int(x)] = t per event IN STREAM ORDER -- for every pixel the
 * LAST event that hits it wins.  int() truncates toward zero, a negative index wraps once (numpy indexing), anything else
 * (also NaN / infinity) is counted in *oob (the reference raises IndexError).  last_scratch: h * w uint32, any content.
 * n < 2^32 - 1. */
int evk_timestamp_image_add_f64(const double *x, const double *y, const double *t, int64_t n, int h, int w, double *image,
                                uint32_t *last_scratch, uint32_t *oob, void *stream);
/* EventImage.add_event (image.py:384-385): image[int(y), int(x)] += p.  p == NULL: the indices are only checked -- what
 * upstream's add_events does, which passes a literal 0 for the polarity (image.py:387-389). */
int evk_event_image_add_f64(const double *x, const double *y, const double *p, int64_t n, int h, int w, double *image,
                            uint32_t *oob, void *stream);
/* TimestampImage.get_image (image.py:370-375): out = (scipy.stats.rankdata(image, method='dense') - 1) / its maximum
 * (0 / 0 = NaN for a constant image, as upstream).  scratch: evk_dense_rank_scratch_bytes(npix) bytes, 256-byte aligned. */
int64_t evk_dense_rank_scratch_bytes(int64_t npix);
int evk_dense_rank_f64(const double *image, int64_t npix, double *out, void *scratch, int64_t scratch_bytes, void *stream);
/* EventImage.get_image (image.py:391-394): out = (image - min) / (max - min), NaN-propagating like np.min / np.max.
 * scratch: evk_minmax_scratch_bytes() bytes. */
int64_t evk_minmax_scratch_bytes(void);
int evk_minmax_normalise_f64(const double *image, int64_t npix, double *out, double *scratch, void *stream);

/* ---- element-wise helpers of the host layer (evk_elem.hip, round 6) ---------------------------------------------
 * evk_polarity_weights_f32: pos[i] = p[i] > 0 ? 1 : 0, neg[i] = p[i] <= 0 ? 1 : 0 -- the two weight columns of
 *   events_to_neg_pos_voxel_torch (voxel_grid.py:173-174); either output may be NULL.
 * evk_abs_max: the BIT PATTERN of max |p| over a float32 (elem_bytes 4) or float64 (8) column into 8 bytes of device memory
 *   (a NaN anywhere gives a NaN pattern, as torch's max()); 0 for n == 0.
 * evk_abs: out[i] = |in[i]| (get_iwe's use_polarity=False, objectives.py:184-185); float32 or float64. */
int evk_polarity_weights_f32(const float *p, int64_t n, float *pos, float *neg, void *stream);
int evk_abs_max(const void *p, int elem_bytes, int64_t n, void *out8, void *stream);
/* the average-timestamp images around their events (image.py:266-283): evk_timestamp_planes_init_f32 sets the (4, plane_elems)
 * planes [time+, count+, time-, count-] to 0 / 1 / 0 / 1 (the counts START AT ONE upstream); evk_timestamp_finalise_f32 forms
 * pos = time+ / (count+ == 0 ? 1 : count+) and neg likewise (image.py:278-282), float32 divisions */
/* out[j] = np.searchsorted(a, keys[j]) (side 'left') for a sorted float32 (elem_bytes 4) / float64 (8) device column of n values
 * and m float64 device keys, compared in float64 as numpy does: the window bounds of events_to_voxel_timesync_torch /
 * voxel_grids_fixed_t_torch (voxel_grid.py:104-105) without copying the time column to the host */
int evk_searchsorted_left(const void *a, int elem_bytes, int64_t n, const double *keys, int64_t m, int64_t *out, void *stream);
int evk_timestamp_planes_init_f32(float *out4, int64_t plane_elems, void *stream);
int evk_timestamp_finalise_f32(const float *planes4, int64_t plane_elems, float *pos, float *neg, void *stream);
int evk_abs(const void *in, int elem_bytes, int64_t n, void *out, void *stream);
/* evk_narrow_f64_f32: out[i] = (float)(in[i] - offset), the subtraction in float64; *inexact (optional, caller-zeroed) |= 1
 * when some value is not exactly representable in float32 (also a NaN).  How the host layer takes the reference's float64
 * host arrays (xs, ys, ts, ps of objectives.py / events_cmax.py) onto the float32 path: columns whose values are float32-exact
 * as they are, the time stamps as differences from ts[-1] (DeviceEvents.from_arrays). */
int evk_narrow_f64_f32(const double *in, int64_t n, double offset, float *out, uint32_t *inexact, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Event-sharded data parallelism: the path's only exchange step
 * Every accumulator is a sum over events (image.py:95,111-114,132-135: index_put_(accumulate=True); image.py:37:
 * np.bincount), so ranks holding disjoint event shards compute partial grids and ONE in-place SUM all-reduce of the grid
 * (voxel grid, IWE + dIWE buffer, integer event image) gives every rank the result.  These entry points wrap RCCL
 * (xGMI inside a node), bound at run time with dlopen: a process that already has an RCCL loaded (PyTorch) uses that
 * copy.  One process per GPU; rank 0 creates the id (evk_comm_unique_id_bytes() bytes), the application ships it to
 * the other ranks by whatever means it has (MPI, a file, torch.distributed), every rank calls evk_comm_init.
 * ---------------------------------------------------------------------------------------------------------- */
int evk_comm_unique_id_bytes(void);
int evk_comm_unique_id(void *host_id);
int evk_comm_init(const void *host_id, int rank, int world, void **comm_out);
int evk_comm_destroy(void *comm);
/* buf (device) <- sum over ranks of buf, enqueued on `stream`; int32 for the bit-exact integer event image */
int evk_allreduce_f32(float *buf, int64_t count, void *comm, void *stream);
int evk_allreduce_i32(int32_t *buf, int64_t count, void *comm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EVK_H */
