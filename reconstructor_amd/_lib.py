"""ctypes binding of librcn.so (the C ABI declared in include/rcn.h).

The HIP library IS the product: if it is missing or cannot be loaded this module raises --
there is no CPU fallback anywhere in reconstructor_amd.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("RCN_LIB", os.path.join(_HERE, "librcn.so"))   # RCN_LIB: A/B another build of the same ABI

RCN_OK = 0
ERRORS = {-1: "RCN_ERR_ARG", -2: "RCN_ERR_HIP", -3: "RCN_ERR_NO_DEVICE",
          -4: "RCN_ERR_UNSUPPORTED", -5: "RCN_ERR_NOT_FOUND", -6: "RCN_ERR_NUMERIC",
          -7: "RCN_ERR_COMM", -8: "RCN_ERR_IO"}

# every symbol include/rcn.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "rcn_create", "rcn_destroy", "rcn_last_error", "rcn_version", "rcn_set_stream",
    "rcn_synchronize", "rcn_desc_upload", "rcn_desc_upload_device", "rcn_desc_upload_batch_device", "rcn_desc_upload_batch", "rcn_desc_remove", "rcn_desc_sample_device", "rcn_desc_sample_errors", "rcn_desc_clear",
    "rcn_desc_count", "rcn_match_pair", "rcn_match_grid", "rcn_match_grid_device",
    "rcn_match_last_stats", "rcn_match_profile", "rcn_match_set_workspace_rows", "rcn_ba_default_options", "rcn_ba_solve", "rcn_ba_factor_plan",
    "rcn_landmark_validity", "rcn_landmark_validity_device",
    "rcn_fmat_filter", "rcn_fmat_filter_grid", "rcn_fmat_filter_grid_device",
    "rcn_coords_upload", "rcn_coords_upload_batch", "rcn_coords_clear", "rcn_match_table_filter_device",
    "rcn_host_alloc", "rcn_host_free", "rcn_match_compact_begin", "rcn_match_compact_wait",
    "rcn_shard_owned_images", "rcn_shard_pair_count", "rcn_shard_pairs", "rcn_shard_unique_id",
    "rcn_shard_create", "rcn_shard_destroy", "rcn_shard_ctx", "rcn_shard_reserve", "rcn_shard_put_image", "rcn_shard_exchange",
    "rcn_shard_match", "rcn_shard_lists", "rcn_shard_info", "rcn_device_count",
    "rcn_shard_fail", "rcn_shard_set_timeout", "rcn_shard_gather_lists", "rcn_shard_merge_lists", "rcn_shard_profile", "rcn_shard_profile_read", "rcn_shard_filter", "rcn_match_grid_filtered",
    "rcn_ba_session_create", "rcn_ba_session_destroy", "rcn_ba_session_add_camera", "rcn_ba_session_cameras",
    "rcn_ba_session_add_points", "rcn_ba_session_add_observations", "rcn_ba_session_counts", "rcn_ba_session_graph",
    "rcn_ba_session_solve", "rcn_ba_session_read_points", "rcn_ba_session_points_device", "rcn_ba_session_validity",
    "rcn_ba_session_remove_outliers",
    "rcn_store_save", "rcn_store_open", "rcn_store_contents_of", "rcn_store_close", "rcn_store_upload",
]
SHARD_ID_BYTES = 128


class RcnError(RuntimeError):
    def __init__(self, code, text=""):
        self.code = code
        super().__init__("%s (%d)%s" % (ERRORS.get(code, "RCN_ERR"), code, ": " + text if text else ""))


class MatchStats(C.Structure):
    _fields_ = [("rows_total", C.c_int64), ("rows_reranked", C.c_int64), ("rows_exact_fallback", C.c_int64),
                ("pair_distances", C.c_int64), ("err_bound_d2", C.c_double),
                ("used_mfma_path", C.c_int32), ("profiled_calls", C.c_int32),
                ("coarse_ms", C.c_double), ("rerank_ms", C.c_double), ("unique_ms", C.c_double),
                ("rows_brute_force", C.c_int64), ("chunks", C.c_int32), ("coarse_launches", C.c_int32)]


class ShardStats(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("n_images", C.c_int32), ("images_per_rank", C.c_int32),
                ("n_pairs", C.c_int64), ("exchange_bytes_f16", C.c_int64), ("exchange_bytes_f32", C.c_int64),
                ("comm_ranks", C.c_int32), ("reserved", C.c_int32)]


class ShardTimes(C.Structure):
    _fields_ = [("exchanges", C.c_int32), ("matches", C.c_int32), ("exchange_ms", C.c_double),
                ("f32_gather_ms", C.c_double), ("match_ms", C.c_double)]


class StoreContents(C.Structure):
    _fields_ = [("n_images", C.c_int32), ("D", C.c_int32), ("has_coords", C.c_int32), ("n_pairs", C.c_int32),
                ("img_ids", C.c_void_p), ("img_K", C.c_void_p), ("desc", C.c_void_p), ("coords", C.c_void_p),
                ("pairs", C.c_void_p), ("offsets", C.c_void_p), ("qt", C.c_void_p)]


class BaProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32),
                ("reserved", C.c_int32),
                ("poses", C.c_void_p), ("intrinsics", C.c_void_p), ("points", C.c_void_p),
                ("obs_uv", C.c_void_p), ("obs_cam", C.c_void_p), ("obs_pt", C.c_void_p)]


class LandmarkProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32),
                ("reserved", C.c_int32),
                ("poses34", C.c_void_p), ("intrinsics", C.c_void_p), ("points", C.c_void_p),
                ("pt_off", C.c_void_p), ("obs_cam", C.c_void_p), ("obs_xy", C.c_void_p)]


class BaOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("intrinsics_mode", C.c_int32),
                ("fix_cam0_pose", C.c_int32), ("fix_cam1_translation", C.c_int32),
                ("focal_upper_bound", C.c_double),
                ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double),
                ("max_consecutive_invalid_steps", C.c_int32), ("jacobi_scaling", C.c_int32)]


class BaSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("initial_rms_px", C.c_double), ("final_rms_px", C.c_double),
                ("iterations", C.c_int32), ("successful_steps", C.c_int32),
                ("unsuccessful_steps", C.c_int32), ("invalid_steps", C.c_int32),
                ("termination", C.c_int32), ("line_search_backtracks", C.c_int32),
                ("bound_projections", C.c_int32), ("reduced_dim", C.c_int32),
                ("pair_lists_reused", C.c_int32), ("reserved", C.c_int32),
                ("solve_seconds", C.c_double), ("schur_seconds", C.c_double),
                ("cholesky_seconds", C.c_double), ("trisolve_seconds", C.c_double),
                ("cost_trace", C.c_double * 160),
                ("jacobian_seconds", C.c_double), ("jacobian_evals", C.c_int32), ("factor_schedule", C.c_int32)]


_LIB = None


def load():
    """Load librcn.so.  torch (when used in the same process) must be imported first so that
    both share one HIP runtime (same SONAME libamdhip64.so.7); callers in this package do so."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "reconstructor_amd/librcn.so is missing: run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        import torch  # noqa: F401  -- FIRST: torch ships its own ROCm runtime + RCCL under the same SONAMEs; whichever
    except ImportError:             # copy is mapped first serves both, and two copies in one process corrupt the heap at exit
        pass
    L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.rcn_create.restype = C.c_int
    L.rcn_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rcn_destroy.restype = None
    L.rcn_destroy.argtypes = [vp]
    L.rcn_last_error.restype = C.c_char_p
    L.rcn_last_error.argtypes = [vp]
    L.rcn_version.restype = C.c_char_p
    L.rcn_version.argtypes = []
    L.rcn_set_stream.restype = C.c_int
    L.rcn_set_stream.argtypes = [vp, vp]
    L.rcn_synchronize.restype = C.c_int
    L.rcn_synchronize.argtypes = [vp]
    L.rcn_desc_upload.restype = C.c_int
    L.rcn_desc_upload.argtypes = [vp, i32, vp, i32, i32]
    L.rcn_desc_upload_device.restype = C.c_int
    L.rcn_desc_upload_device.argtypes = [vp, i32, vp, i32, i32]
    L.rcn_desc_upload_batch_device.restype = C.c_int
    L.rcn_desc_upload_batch_device.argtypes = [vp, i32, i32, vp, i32, i32]
    L.rcn_desc_upload_batch.restype = C.c_int
    L.rcn_desc_upload_batch.argtypes = [vp, i32, i32, vp, vp, i32]
    L.rcn_desc_remove.restype = C.c_int
    L.rcn_desc_remove.argtypes = [vp, i32]
    L.rcn_desc_sample_device.restype = C.c_int
    L.rcn_desc_sample_device.argtypes = [vp, vp, i64, i64, i64, i32, i32, vp, i32, i32, vp]
    L.rcn_desc_sample_errors.restype = C.c_int
    L.rcn_desc_sample_errors.argtypes = [vp, C.POINTER(i32)]
    L.rcn_desc_clear.restype = C.c_int
    L.rcn_desc_clear.argtypes = [vp]
    L.rcn_desc_count.restype = C.c_int
    L.rcn_desc_count.argtypes = [vp]
    L.rcn_match_pair.restype = C.c_int
    L.rcn_match_pair.argtypes = [vp, vp, i32, vp, i32, i32, f32, vp, C.POINTER(i32)]
    L.rcn_match_grid.restype = C.c_int
    L.rcn_match_grid.argtypes = [vp, vp, i32, f32, vp, i64, vp]
    L.rcn_match_grid_device.restype = C.c_int
    L.rcn_match_grid_device.argtypes = [vp, vp, i32, f32, vp, i64, vp]
    L.rcn_match_last_stats.restype = C.c_int
    L.rcn_match_last_stats.argtypes = [vp, C.POINTER(MatchStats)]
    L.rcn_match_profile.restype = C.c_int
    L.rcn_match_profile.argtypes = [vp, C.c_int]
    L.rcn_ba_default_options.restype = None
    L.rcn_ba_default_options.argtypes = [i32, C.POINTER(BaOptions)]
    L.rcn_ba_solve.restype = C.c_int
    L.rcn_ba_solve.argtypes = [vp, C.POINTER(BaProblem), C.POINTER(BaOptions), C.POINTER(BaSummary)]
    for fn in (L.rcn_landmark_validity, L.rcn_landmark_validity_device):
        fn.restype = C.c_int
        fn.argtypes = [vp, C.POINTER(LandmarkProblem), C.c_double, C.c_double, vp, vp, vp]
    L.rcn_fmat_filter.restype = C.c_int
    L.rcn_fmat_filter.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    for fn in (L.rcn_fmat_filter_grid, L.rcn_fmat_filter_grid_device):
        fn.restype = C.c_int
        fn.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.rcn_coords_upload.restype = C.c_int
    L.rcn_coords_upload.argtypes = [vp, i32, vp, i32]
    L.rcn_coords_upload_batch.restype = C.c_int
    L.rcn_coords_upload_batch.argtypes = [vp, i32, i32, vp, vp]
    L.rcn_coords_clear.restype = C.c_int
    L.rcn_coords_clear.argtypes = [vp]
    L.rcn_match_table_filter_device.restype = C.c_int
    L.rcn_match_table_filter_device.argtypes = [vp, vp, i32, vp, C.c_int64, vp, vp]
    L.rcn_host_alloc.restype = C.c_int
    L.rcn_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.rcn_host_free.restype = None
    L.rcn_host_free.argtypes = [vp]
    L.rcn_match_compact_begin.restype = C.c_int
    L.rcn_match_compact_begin.argtypes = [vp, vp, i64, vp, i32, vp, vp, i64, C.POINTER(i64)]
    L.rcn_match_compact_wait.restype = C.c_int
    L.rcn_match_compact_wait.argtypes = [vp]
    L.rcn_shard_owned_images.restype = C.c_int
    L.rcn_shard_owned_images.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.rcn_shard_pair_count.restype = i64
    L.rcn_shard_pair_count.argtypes = [i32, i32, i32]
    L.rcn_shard_pairs.restype = C.c_int
    L.rcn_shard_pairs.argtypes = [i32, i32, i32, vp]
    L.rcn_shard_unique_id.restype = C.c_int
    L.rcn_shard_unique_id.argtypes = [vp]
    L.rcn_shard_create.restype = C.c_int
    L.rcn_shard_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    L.rcn_shard_destroy.restype = None
    L.rcn_shard_destroy.argtypes = [vp]
    L.rcn_shard_ctx.restype = vp
    L.rcn_shard_ctx.argtypes = [vp]
    L.rcn_shard_reserve.restype = C.c_int
    L.rcn_shard_reserve.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.rcn_shard_exchange.restype = C.c_int
    L.rcn_shard_exchange.argtypes = [vp, vp, vp]
    L.rcn_shard_put_image.restype = C.c_int
    L.rcn_shard_put_image.argtypes = [vp, i32, vp, i32]
    L.rcn_shard_lists.restype = C.c_int
    L.rcn_shard_lists.argtypes = [vp, vp, vp, i64, C.POINTER(i64)]
    L.rcn_device_count.restype = C.c_int
    L.rcn_device_count.argtypes = []
    L.rcn_shard_match.restype = C.c_int
    L.rcn_shard_match.argtypes = [vp, f32, vp, i64, vp]
    L.rcn_shard_info.restype = C.c_int
    L.rcn_shard_info.argtypes = [vp, C.POINTER(ShardStats)]
    L.rcn_shard_filter.restype = C.c_int
    L.rcn_shard_filter.argtypes = [vp, vp]
    L.rcn_match_grid_filtered.restype = C.c_int
    L.rcn_match_grid_filtered.argtypes = [vp, vp, i32, f32, i32, vp, i64, vp, vp]
    L.rcn_shard_fail.restype = C.c_int
    L.rcn_shard_fail.argtypes = [vp, i32]
    L.rcn_shard_set_timeout.restype = C.c_int
    L.rcn_shard_set_timeout.argtypes = [vp, C.c_double]
    L.rcn_shard_gather_lists.restype = C.c_int
    L.rcn_shard_gather_lists.argtypes = [vp, i32, vp, i64, vp, vp, vp, i64, C.POINTER(i64)]
    L.rcn_shard_merge_lists.restype = C.c_int
    L.rcn_shard_merge_lists.argtypes = [i32, i32, vp, vp, vp, vp, i64, C.POINTER(i64)]
    L.rcn_shard_profile.restype = C.c_int
    L.rcn_shard_profile.argtypes = [vp, C.c_int]
    L.rcn_shard_profile_read.restype = C.c_int
    L.rcn_shard_profile_read.argtypes = [vp, C.POINTER(ShardTimes)]
    L.rcn_ba_session_create.restype = C.c_int
    L.rcn_ba_session_create.argtypes = [vp, C.POINTER(vp)]
    L.rcn_ba_session_destroy.restype = None
    L.rcn_ba_session_destroy.argtypes = [vp]
    L.rcn_ba_session_add_camera.restype = C.c_int
    L.rcn_ba_session_add_camera.argtypes = [vp, vp, vp, C.POINTER(i32)]
    L.rcn_ba_session_cameras.restype = C.c_int
    L.rcn_ba_session_cameras.argtypes = [vp, vp, vp, vp, vp]
    L.rcn_ba_session_add_points.restype = C.c_int
    L.rcn_ba_session_add_points.argtypes = [vp, i32, vp, C.POINTER(i32)]
    L.rcn_ba_session_add_observations.restype = C.c_int
    L.rcn_ba_session_add_observations.argtypes = [vp, i32, vp, vp, vp]
    L.rcn_ba_session_counts.restype = C.c_int
    L.rcn_ba_session_counts.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64)]
    L.rcn_ba_session_graph.restype = C.c_int
    L.rcn_ba_session_graph.argtypes = [vp, vp, vp, vp]
    L.rcn_ba_session_solve.restype = C.c_int
    L.rcn_ba_session_solve.argtypes = [vp, C.POINTER(BaOptions), C.POINTER(BaSummary)]
    L.rcn_ba_session_read_points.restype = C.c_int
    L.rcn_ba_session_read_points.argtypes = [vp, vp]
    L.rcn_ba_session_points_device.restype = vp
    L.rcn_ba_session_points_device.argtypes = [vp]
    L.rcn_ba_session_validity.restype = C.c_int
    L.rcn_ba_session_validity.argtypes = [vp, vp, C.c_double, C.c_double, vp, C.POINTER(i32), C.POINTER(i32)]
    L.rcn_ba_session_remove_outliers.restype = C.c_int
    L.rcn_ba_session_remove_outliers.argtypes = [vp, vp, C.POINTER(i32)]
    L.rcn_store_save.restype = C.c_int
    L.rcn_store_save.argtypes = [C.c_char_p, C.POINTER(StoreContents)]
    L.rcn_store_open.restype = C.c_int
    L.rcn_store_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.rcn_store_contents_of.restype = C.c_int
    L.rcn_store_contents_of.argtypes = [vp, C.POINTER(StoreContents)]
    L.rcn_store_close.restype = None
    L.rcn_store_close.argtypes = [vp]
    L.rcn_store_upload.restype = C.c_int
    L.rcn_store_upload.argtypes = [vp, vp]
    _LIB = L
    return L


class Context:
    """One rcn_ctx (one GPU)."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.rcn_create(int(device), C.byref(h))
        if rc != RCN_OK:
            raise RcnError(rc, "rcn_create(device=%d): no usable gfx950 device" % device)
        self.h = h
        self.device = device

    def check(self, rc):
        if rc != RCN_OK:
            raise RcnError(rc, self.lib.rcn_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rcn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
