"""ctypes declarations for libppf_hip.so (include/ppf_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950)
into ``yolo_ppf_pose_estimation_amd/csrc/libppf_hip.so``.  Loading fails loudly when it is missing:
there is no Python or CPU fallback for the engine.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPF_HIP_LIB") or os.path.join(HERE, "csrc", "libppf_hip.so")  # override: diagnostic builds only

PPF_OPT_HIT_FRACTION, PPF_OPT_GROUP_ROUND_BUCKETS, PPF_OPT_CLUSTER_SERIAL, PPF_OPT_ACC32, PPF_OPT_TABLE_FRACTION, PPF_OPT_BATCH_REFS, PPF_OPT_RUN_STAGING = 1, 2, 3, 4, 5, 6, 7
PPF_ICP_NO_SMALL_LEVELS, PPF_ICP_ONE_STREAM, PPF_ICP_LEGACY, PPF_ICP_GRID_ALWAYS = 1, 2, 4, 8
PPF_OK, PPF_ERR_INVALID, PPF_ERR_NOT_TRAINED, PPF_ERR_HIP, PPF_ERR_NOMEM, PPF_ERR_IO, PPF_ERR_CAPACITY = range(7)
STATUS_NAMES = {0: "PPF_OK", 1: "PPF_ERR_INVALID", 2: "PPF_ERR_NOT_TRAINED", 3: "PPF_ERR_HIP", 4: "PPF_ERR_NOMEM",
                5: "PPF_ERR_IO", 6: "PPF_ERR_CAPACITY"}


class PPFError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class TrainParams(C.Structure):
    _fields_ = [("relative_sampling_step", C.c_double), ("relative_distance_step", C.c_double),
                ("num_angles", C.c_double), ("presampled", C.c_int32), ("distance_from_distance_step", C.c_int32),
                ("max_tile_refs", C.c_int32), ("key_equality", C.c_int32), ("feature", C.c_int32),
                ("reserved", C.c_int32)]


class MatchParams(C.Structure):
    _fields_ = [("relative_scene_sample_step", C.c_double), ("relative_scene_distance", C.c_double),
                ("position_threshold", C.c_double), ("rotation_threshold", C.c_double),
                ("use_weighted_avg", C.c_int32), ("presampled", C.c_int32), ("ref_offset", C.c_int32),
                ("ref_stride", C.c_int32), ("skip_clustering", C.c_int32), ("vote_mode", C.c_int32),
                ("pair_radius", C.c_double), ("rot_metric_relative", C.c_int32), ("alpha_range_2pi", C.c_int32)]


class IcpParams(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("tolerance", C.c_float), ("rejection_scale", C.c_float),
                ("num_levels", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32 * 3)]


class Pose(C.Structure):
    _fields_ = [("pose", C.c_double * 16), ("q", C.c_double * 4), ("t", C.c_double * 3), ("angle", C.c_double),
                ("alpha", C.c_double), ("residual", C.c_double), ("model_index", C.c_uint32),
                ("num_votes", C.c_uint32)]


class Vote(C.Structure):
    _fields_ = [("ref_ind_max", C.c_uint32), ("alpha_ind_max", C.c_uint32), ("max_votes", C.c_uint32)]


class ModelInfo(C.Structure):
    _fields_ = [("n_ref", C.c_int32), ("num_angles", C.c_int32), ("slots", C.c_uint32), ("n_buckets", C.c_uint32),
                ("n_entries", C.c_uint64), ("n_tiles", C.c_int32), ("tile_refs", C.c_int32),
                ("angle_step", C.c_double), ("distance_step", C.c_double), ("diameter", C.c_double),
                ("position_threshold_default", C.c_double), ("rotation_threshold_default", C.c_double),
                ("device_bytes", C.c_uint64)]


class MatchStats(C.Structure):
    _fields_ = [("n_scene_sampled", C.c_int32), ("n_paired", C.c_int32), ("n_ref", C.c_int32),
                ("n_poses", C.c_int32), ("n_pairs", C.c_uint64), ("n_votes", C.c_uint64),
                ("ms_vote_kernel", C.c_float), ("ms_pair_kernel", C.c_float), ("ms_total_device", C.c_float),
                ("ms_group_kernel", C.c_float), ("n_hits", C.c_uint64), ("n_lds_atomics", C.c_uint64),
                ("scratch_bytes", C.c_uint64), ("n_batches", C.c_int32), ("n_retries", C.c_int32),
                ("n_acc32_items", C.c_uint64), ("n_tables", C.c_uint64), ("phase_clocks", C.c_uint64 * 8)]


def stats_dict(st):
    """A stats structure as a plain dict (array fields as lists)."""
    out = {}
    for name, _ in st._fields_:
        v = getattr(st, name)
        out[name] = list(v) if isinstance(v, C.Array) else v
    return out


class BatchStats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_votes", C.c_uint64), ("n_hits", C.c_uint64), ("n_lds_atomics", C.c_uint64),
                ("n_matches", C.c_int32), ("n_retries", C.c_int32), ("lanes", C.c_int32), ("ms_wall", C.c_float),
                ("ms_vote_kernel", C.c_float), ("ms_pair_kernel", C.c_float), ("ms_group_kernel", C.c_float),
                ("reserved", C.c_int32)]


# every symbol include/ppf_hip.h declares (tests/test_capi_symbols.py checks the header against this)
_SIGNATURES = {
    "ppf_default_train_params": (None, [C.POINTER(TrainParams)]),
    "ppf_default_match_params": (None, [C.POINTER(MatchParams)]),
    "ppf_default_icp_params": (None, [C.POINTER(IcpParams)]),
    "ppf_abi_version": (C.c_int, []),
    "ppf_last_error": (C.c_int, [C.c_char_p, C.c_int]),
    "ppf_device_count": (C.c_int, []),
    "ppf_model_train": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(TrainParams), C.POINTER(C.c_void_p)]),
    "ppf_pair_features": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "ppf_model_retain": (C.c_int, [C.c_void_p]),
    "ppf_model_release": (C.c_int, [C.c_void_p]),
    "ppf_model_get_info": (C.c_int, [C.c_void_p, C.POINTER(ModelInfo)]),
    "ppf_model_get_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "ppf_model_trim_contexts": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "ppf_model_nearest_pairs": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_int)]),
    "ppf_model_get_sampled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ppf_model_get_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppf_model_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ppf_model_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ppf_model_check_file": (C.c_int, [C.c_char_p]),
    "ppf_model_save_mem": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ppf_model_load_mem": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ppf_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                            C.POINTER(MatchParams), C.POINTER(Pose), C.c_int, C.POINTER(C.c_int)]),
    "ppf_match_batch": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                  C.c_int, C.POINTER(MatchParams), C.POINTER(Pose), C.c_int, C.POINTER(C.c_int)]),
    "ppf_batch_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ppf_batch_destroy": (C.c_int, [C.c_void_p]),
    "ppf_batch_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ppf_batch_run": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int,
                                C.c_int, C.c_int, C.c_int, C.POINTER(MatchParams), C.POINTER(Pose), C.c_int, C.POINTER(C.c_int),
                                C.POINTER(BatchStats)]),
    "ppf_batch_device_block": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "ppf_batch_copy_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ppf_raw_votes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.POINTER(MatchParams), C.POINTER(Vote), C.POINTER(Pose), C.c_int,
                                C.POINTER(C.c_int), C.POINTER(MatchStats)]),
    "ppf_workspace_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ppf_workspace_destroy": (C.c_int, [C.c_void_p]),
    "ppf_workspace_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ppf_workspace_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "ppf_match_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.POINTER(MatchParams), C.c_void_p]),
    "ppf_workspace_results": (C.c_int, [C.c_void_p, C.POINTER(Vote), C.POINTER(Pose), C.c_int, C.POINTER(C.c_int),
                                        C.POINTER(Pose), C.c_int, C.POINTER(C.c_int), C.POINTER(MatchStats)]),
    "ppf_workspace_ref_counters": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ppf_debug_accumulators": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(MatchParams), C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "ppf_debug_block_size": (C.c_size_t, [C.c_size_t]),
    "ppf_debug_device_math": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ppf_workspace_device_poses": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "ppf_workspace_copy_top_poses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ppf_workspace_copy_raw_poses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ppf_cluster_poses_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(MatchParams),
                                           C.c_void_p]),
    "ppf_cluster_poses": (C.c_int, [C.c_void_p, C.POINTER(Pose), C.c_int, C.c_int, C.POINTER(MatchParams),
                                    C.POINTER(Pose), C.c_int, C.POINTER(C.c_int)]),
    "ppf_sample_cloud": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                                   C.POINTER(C.c_int)]),
    "ppf_transform_pc_pose": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    "ppf_icp_refine": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(IcpParams),
                                 C.POINTER(Pose), C.c_int, C.POINTER(C.c_int)]),
    "ppf_icp_refine_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(IcpParams),
                                        C.POINTER(Pose), C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "ppf_cloud_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ppf_cloud_release": (C.c_int, [C.c_void_p]),
    "ppf_cloud_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "ppf_cloud_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ppf_cloud_device_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "ppf_prep_crop": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double),
                                C.POINTER(C.c_void_p)]),
    "ppf_prep_voxel_grid": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(C.c_void_p)]),
    "ppf_prep_outlier_removal": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_void_p)]),
    "ppf_prep_normals": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ppf_prep_edges": (C.c_int, [C.c_void_p, C.c_float, C.POINTER(C.c_void_p)]),
    "ppf_prep_to_mat": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ppf_match_clouds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(MatchParams), C.POINTER(Pose), C.c_int,
                                   C.POINTER(C.c_int)]),
    "ppf_icp_refine_clouds": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(IcpParams), C.POINTER(Pose), C.c_int,
                                        C.POINTER(C.c_int)]),
    "ppf_prep_knn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ppf_icp_register": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(IcpParams),
                                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}

PPF_FEATURE_PPF, PPF_FEATURE_DARBOUX = 0, 1
PPF_ABI_VERSION = 4   # include/ppf_hip.h
PPF_NOFF_MAT = 3      # x y z nx ny nz rows (the N x 6 Mat of CloudProcessing.h:163-190)
PPF_NOFF_PCL = 4      # pcl::PointNormal rows: x y z 1 | nx ny nz 0 | curvature pad pad pad (stride 12)

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as the system
    one libppf_hip.so is linked against).  Whichever copy is loaded first serves every later user of that SONAME -- but
    torch opens its copy by path, so when the system runtime came first the process ends up with two runtimes and
    torch sees "No HIP GPUs".  Loading torch's copy first (by path, without importing torch) makes both agree,
    whatever the import order.  Without torch installed nothing happens and the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libppf_hip.so; raises (never falls back) when the extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()').  There is no CPU fallback.")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.ppf_abi_version() != PPF_ABI_VERSION:
            raise ImportError("libppf_hip.so ABI version mismatch")
        _lib = L
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib().ppf_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


def check(status: int):
    if status != PPF_OK:
        raise PPFError(status, last_error())
