"""ctypes binding of libevk.so (include/evk.h).  There is NO fallback: if the HIP library is missing or cannot be
loaded the product fails loudly -- build it with `python -m event_utils_amd.csrc.build` (or __graft_entry__.build())."""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_ENV_DATA = getattr(os.environ, "_data", None)     # posix: {bytes: bytes}, the dict behind os.environ (kept in step by setenv / monkeypatch)
_ENV_KEYS = {}


def getenv(name, default):
    """os.environ.get(name, default) without its per-call encode / decode machinery (~0.7 us each, seven of them on the
    path of one public call: with synchronous error reports the host's share of a call is on the critical path)."""
    if _ENV_DATA is None or type(_ENV_DATA) is not dict:
        return os.environ.get(name, default)
    k = _ENV_KEYS.get(name)
    if k is None:
        k = _ENV_KEYS[name] = os.fsencode(name)
    v = _ENV_DATA.get(k)
    return default if v is None else os.fsdecode(v)


LIB_PATH = os.environ.get("EVK_LIB_PATH") or os.path.join(_HERE, "csrc", "libevk.so")   # EVK_LIB_PATH: A/B builds

EVK_IWE_ABS_POLARITY = 1
EVK_IWE_GRADIENT = 2
EVK_IWE_COMPACT = 16
EVK_POST_MIX, EVK_POST_BLUR_IWE, EVK_POST_VALUE, EVK_POST_NONE = 1, 2, 4, 8
EVK_VOXEL_OVERWRITE, EVK_VOXEL_SPLIT_POLARITY, EVK_VOXEL_T_FROM_EVENTS = 1, 2, 4
EVK_VOXEL2_PARTITION_ONLY, EVK_VOXEL2_TILES_ONLY = 16, 32
EVK_VOXEL_DETERMINISTIC = 256
EVK_IMAGE2_NO_FIXED = 512
EVK_VOXEL2_REC4, EVK_VOXEL2_REC8, EVK_VOXEL2_NO_COUNT, EVK_VOXEL2_WG512 = 1024, 2048, 4096, 8192
EVK_VOXEL2_LIVE = 16384
EVK_COLUMNS_UNALIGNED = 65536
EVK_VOXEL2_NO_COUNT2 = 32768
EVK_STAGE_STATS, EVK_STAGE_COMPACT, EVK_STAGE_LEGACY_SCATTER = 16, 32, 64

P = c_void_p  # every device / host pointer crosses as void*

# name -> argtypes (restype is int unless noted); mirrors include/evk.h one-to-one
SIGNATURES = {
    "evk_image_nearest_i32": [P, P, P, c_int64, c_int, c_int, P, P, P],
    "evk_image_nearest_f64": [P, P, P, c_int64, c_int, c_int, P, P, P],
    "evk_image_nearest_f32": [P, P, P, c_int64, c_int, c_int, c_float, c_float, P, P, P],
    "evk_image_bilinear_f32": [P, P, P, c_int64, c_int, c_int, c_float, c_float, P, P, P],
    "evk_splat_indexed_f32": [P, P, P, P, P, c_int64, c_int, c_int, P, P, P],
    "evk_splat_drv_indexed_f32": [P, P, P, P, P, P, c_int, c_int64, c_int, c_int, P, P, P],
    "evk_image_drv_f64": [P, P, P, P, P, c_int64, c_int, c_int, c_float, c_float, P, P, P, P],
    "evk_image_gather_bilinear_f64": [P, P, c_int64, P, c_int, c_int, P, P, P],
    "evk_image_gather_bilinear_f64img": [P, P, c_int64, P, c_int, c_int, P, P, P],
    "evk_timestamp_images_f32": [P, P, P, P, c_int64, c_int, c_int, c_float, c_float, c_int, c_float, c_float, P, P, P],
    "evk_voxel_f32": [P, P, P, P, c_int64, c_float, c_float, c_int, c_int, c_int, P, P, P],
    "evk_voxel_from_events_f32": [P, P, P, P, c_int64, c_int, c_int, c_int, P, P, P],
    "evk_voxel_segments_f32": [P, P, P, P, P, c_int, c_int64, c_int, c_int, c_int, P, P, P],
    "evk_voxel_f64": [P, P, P, P, c_int64, c_double, c_double, c_int, c_int, c_int, P, P, P],
    "evk_warp_linvel_f64": [P, P, P, c_int64, c_double, c_double, c_double, P, P, P, P, P],
    "evk_warp_flow_field_f32": [P, P, P, c_int64, P, c_int, c_int, c_float, P, P, P],
    "evk_bounds_mask_f64": [P, P, c_int64, c_double, c_double, c_double, c_double, P, P],
    "evk_iwe_linvel_f32": [P, P, P, P, c_int64, c_double, c_double, c_double, c_double, c_double, c_int, c_int,
                           c_uint32, c_double, P, P, P],
    "evk_iwe_linvel_f64": [P, P, P, P, c_int64, c_double, c_double, c_double, c_double, c_double, c_int, c_int,
                           c_uint32, c_double, P, P, P],
    "evk_gaussian_filter_f32": [P, P, P, c_int, P, P, c_int, P],
    "evk_variance_f32": [P, c_int64, P, P, c_int64, P],
    "evk_variance_grad_f32": [P, P, c_int64, P, P, c_int64, P],
    "evk_objective_variance_f32": [P, c_int, c_int, P, c_int, P, P, c_int64, P],
    "evk_objective_variance_grad_f32": [P, P, c_int, c_int, P, c_int, c_uint32, P, P, c_int64, P],
    "evk_cmax_variance_tiled_f32": [P, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_double, c_double, c_double,
                                    c_double, c_double, c_double, c_int, c_int, c_uint32, c_double, c_double, c_double, P, c_int,
                                    c_uint32, P, c_int64, P, P, P, c_int64, P, c_int, P, P],
    "evk_cmax_bfgs_variance_tiled_f32": [P, P, c_int64, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_double,
                                         c_int, c_int, c_uint32, c_double, c_double, c_double, P, c_int, c_uint32, P, c_int64,
                                         P, P, P, c_int64, P, P, P, P, P, c_int, P],
    "evk_bfgs2_minimize": [P, P, P, P, P, P, c_int],
    "evk_iwe_linvel_tiled_batch3_f32": [P, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_double,
                                        c_double, P, P, c_double, c_double, c_int, c_int, c_uint32, c_double, c_double,
                                        c_double, P, c_int64, P, P],
    "evk_objective_variance_planes_f32": [P, c_int, c_int, c_int, P, c_int, P, P, c_int64, P],
    "evk_cmax_variance_batch3_tiled_f32": [P, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_double,
                                           c_double, P, P, c_double, c_double, c_int, c_int, c_uint32, c_double, c_double,
                                           c_double, P, c_int, P, c_int64, P, P, P, c_int64, P, c_int, P, P],
    "evk_objective_stats_f32": [P, c_int, c_int, P, c_int, c_double, c_double, P, P, c_int64, P],
    "evk_objective_variance_fg_f32": [P, P, c_int, c_int, P, c_int, c_uint32, P, P, c_int64, P],
    "evk_objective_gradsums_f32": [P, P, c_int, c_int, P, c_int, c_uint32, c_int, c_double, P, P, c_int64, P],
    "evk_spectral_norm_sq_f32": [P, c_int, c_int, P, P, c_int64, P],
    "evk_timestamp_image_add_f64": [P, P, P, c_int64, c_int, c_int, P, P, P, P],
    "evk_event_image_add_f64": [P, P, P, c_int64, c_int, c_int, P, P, P],
    "evk_dense_rank_f64": [P, c_int64, P, P, c_int64, P],
    "evk_minmax_normalise_f64": [P, c_int64, P, P, P],
    "evk_polarity_weights_f32": [P, c_int64, P, P, P],
    "evk_narrow_f64_f32": [P, c_int64, c_double, P, P, P],
    "evk_abs_max": [P, c_int, c_int64, P, P],
    "evk_searchsorted_left": [P, c_int, c_int64, P, c_int64, P, P],
    "evk_timestamp_planes_init_f32": [P, c_int64, P],
    "evk_timestamp_finalise_f32": [P, c_int64, P, P, P],
    "evk_abs": [P, c_int, c_int64, P, P],
    "evk_bucket_num_tiles": [c_int, c_int, c_int, c_int],
    "evk_bucket_events_f32": [P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P, c_int, P],
    "evk_bucket_events_native_f32": [P, P, c_int, P, c_int, c_double, P, c_int, c_int64, c_int, c_int, c_int, c_int, c_int,
                                     P, P, P, c_int64, P, c_int, P],
    "evk_native_to_columns_f32": [P, P, c_int, P, c_int, c_double, P, c_int, c_int64, P, P, P, P, P],
    "evk_voxel_tiled_f32": [P, P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, P, P, c_int64, P],
    "evk_voxel2_f32": [P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, P, P, P, c_int64,
                       P, P, c_uint32, P],
    "evk_voxel2_native_f32": [P, P, c_int, P, c_int, c_double, P, c_int, c_int64, c_int, c_int, c_int, c_int, c_float,
                              c_float, c_int, c_int, P, P, P, c_int64, P, P, c_uint32, P],
    "evk_normalise_time_f32": [P, c_int64, c_float, c_float, c_int, P, P],
    "evk_voxel2_band_f32": [c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P],
    "evk_image2_nearest_i32": [P, P, P, c_int64, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P, P, c_uint32, P],
    "evk_image2_nearest_f32": [P, P, P, c_int64, c_int, c_int, c_float, c_float, c_int, c_int, c_int, P, P, P, c_int64, P, P,
                               c_uint32, P],
    "evk_image2_bilinear_f32": [P, P, P, c_int64, c_int, c_int, c_float, c_float, c_int, c_int, c_int, P, P, P, c_int64, P, P,
                                c_uint32, P],
    "evk_image2_splat_indexed_f32": [P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P, P, c_uint32, P],
    "evk_image2_splat_drv_indexed_f32": [P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P, P, c_uint32,
                                         P],
    "evk_image2_drv_f64": [P, P, P, P, P, c_int64, c_int, c_int, c_float, c_float, c_int, c_int, c_int, P, P, P, P, c_int64, P, P,
                           c_uint32, P],
    "evk_timestamp_images2_f32": [P, P, P, P, c_int64, c_int, c_int, c_float, c_float, c_int, c_float, c_float, c_int, c_int,
                                  c_int, P, P, P, c_int64, P, P, c_uint32, P],
    "evk_comm_unique_id": [P],
    "evk_comm_init": [P, c_int, c_int, P],
    "evk_comm_destroy": [P],
    "evk_objective_variance_rows_f32": [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_uint32, P, P, c_int64, P],
    "evk_compact_records_f32": [P, c_int64, c_int, c_int, c_int, c_int, P, P, P],
    "evk_allreduce_f32": [P, c_int64, P, P],
    "evk_allreduce_i32": [P, c_int64, P, P],
    "evk_iwe_linvel_tiled_f32": [P, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_double, c_double, c_double,
                                 c_double, c_double, c_double, c_int, c_int, c_uint32, c_double, c_double, c_double, P, c_int64,
                                 P, P, P],
}
_SPECIAL = {
    "evk_version": ([], c_int),
    "evk_error_string": ([c_int], c_char_p),
    "evk_reduce_scratch_bytes": ([], c_int64),
    "evk_bucket_scratch_bytes": ([c_int], c_int64),
    "evk_compact_records_bytes": ([c_int64], c_int64),
    "evk_iwe_tiled_staging_bytes": ([c_int, c_int64, c_int, c_int, c_int, c_int], c_int64),
    "evk_voxel_tiled_staging_bytes": ([c_int, c_int64, c_int, c_int, c_int], c_int64),
    "evk_bucket_index_len": ([c_int, c_int64], c_int64),
    "evk_bucket_max_items": ([c_int, c_int64], c_int),
    "evk_comm_unique_id_bytes": ([], c_int),
    "evk_voxel2_max_tiles": ([], c_int),
    "evk_voxel2_index_len": ([c_int, c_int64], c_int64),
    "evk_voxel2_scratch_bytes": ([c_int, c_int64, c_int, c_int, c_int], c_int64),
    "evk_voxel2_num_tiles": ([c_int, c_int, c_int, c_int], c_int),
    "evk_voxel2_fits": ([c_int, c_int, c_int, c_int, c_int], c_int),
    "evk_num_cu": ([], c_int),
    "evk_image2_scratch_bytes": ([c_int, c_int64, c_int, c_int], c_int64),
    "evk_timestamp_images2_scratch_bytes": ([c_int, c_int64, c_int, c_int], c_int64),
    "evk_image2_indexed_scratch_bytes": ([c_int, c_int64, c_int, c_int], c_int64),
    "evk_dense_rank_scratch_bytes": ([c_int64], c_int64),
    "evk_spectral_scratch_bytes": ([c_int, c_int], c_int64),
    "evk_minmax_scratch_bytes": ([], c_int64),
}


class EvkError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library (loads on first use; raises if it is not built -- never falls back to a CPU path)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise EvkError("libevk.so is not built (%s missing): run `python -m event_utils_amd.csrc.build`; "
                           "event_utils_amd has no CPU fallback" % LIB_PATH)
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise EvkError("cannot load %s: %s" % (LIB_PATH, e))
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        for name, (argtypes, restype) in _SPECIAL.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().evk_error_string(rc)
        raise EvkError("%s failed: %s (code %d)" % (what or "evk call", msg.decode() if msg else "?", rc))


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
