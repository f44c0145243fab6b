/* Entry points that exist only in experiments builds of libevk.so (-DEVK_EXPERIMENTS, tools/exp_build.sh): measured
 * alternatives that the product does not ship.  Not part of the C ABI of include/evk.h. */
#pragma once
#include "../../include/evk.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- one-pass voxel path, 4-byte records (evk_voxel3.hip) ----------------------------------------
 * Same contract, arguments, flags and results as evk_voxel2_f32 (replaces the B index_put_ passes of
 * voxel_grid.py:136-152), but a record is ONE 32-bit word: [31:12] the float32 t_norm of voxel_grid.py:134 as a
 * bit-pattern delta from its sub-chunk's first event (exact), [11:10] polarity code (+1 / -1 / +0 / escape), [9:0] pixel in
 * the tile; records that do not fit (other polarity values, unsorted or sparse time stamps, NaN) escape, exactly, to an
 * 8-byte side array.  24 B/event moved instead of 32, sub-chunks of <= 16 K events.
 *   index    evk_voxel3_index_len(ntiles, n) uint32, ZEROED ONCE by the caller (self-resetting counters);
 *            index[3] counts escaped records (information), index[4] the contributions EVK_VOXEL_DETERMINISTIC could not
 *            represent (the caller reads and clears it)
 *   scratch  evk_voxel3_scratch_bytes(...) bytes, 16-byte aligned, uninitialised
 *   flags    those of evk_voxel2_f32, plus
 *            EVK_VOXEL_DETERMINISTIC: the tile kernel accumulates int64 multiples of 2^-32 (integer adds commute: the
 *            grid is bit-identical from run to run and for any order of the events); |p * weight| must stay below 2^30
 *            and be finite, anything else is counted in index[4] and left out.
 * Tiles: 2^tw_log2 x 2^th_log2 <= 1024 pixels, at most evk_voxel3_max_tiles() of them; n * (1 + 3 * ntiles / 16384) < 2^32. */
int evk_voxel3_max_tiles(void);
int64_t evk_voxel3_index_len(int ntiles, int64_t n);
int64_t evk_voxel3_scratch_bytes(int ntiles, int64_t n, int planes, int tile_w, int tile_h);
int evk_voxel3_f32(const float *x, const float *y, const float *t, const float *p, int64_t n, int h, int wd,
                   int tile_w, int tile_h, float t_first, float t_last, int B, int flags, float *vox,
                   uint32_t *index, void *scratch, int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report,
                   uint32_t seq, void *stream);
/* the same from the reference's on-disk dtypes (event_packagers.py:90-93, h5_to_memmap.py:119-121; see
 * evk_bucket_events_native_f32): 13 B/event read */
int evk_voxel3_native_f32(const int16_t *x, const int16_t *y, int xy_stride, const void *t, int t_kind, double t_offset,
                          const void *p, int p_kind, int64_t n, int h, int wd, int tile_w, int tile_h, float t_first,
                          float t_last, int B, int flags, float *vox, uint32_t *index, void *scratch,
                          int64_t scratch_bytes, uint32_t *oob, uint32_t *host_report, uint32_t seq, void *stream);


#ifdef __cplusplus
}
#endif
