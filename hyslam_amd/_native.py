"""ctypes binding of libhyslam_amd.so (the C ABI in include/hyslam_amd.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HYSLAM_AMD_LIB: development only — an instrumented build of the same sources (make BUILD=_build_prof OUT=../libhyslam_amd_prof.so EXTRA=-D...)
LIB_PATH = os.environ.get("HYSLAM_AMD_LIB") or os.path.join(_HERE, "libhyslam_amd.so")

HS_OK, HS_ERR_INVALID, HS_ERR_HIP, HS_ERR_CAPACITY, HS_ERR_NO_DEVICE = 0, 1, 2, 3, 4

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4")])


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("cell_px", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("fast_threshold", C.c_int32),
                ("blur_taps", C.c_uint16 * 7), ("_pad", C.c_uint16)]


class PreprocessParams(C.Structure):
    _fields_ = [("channels", C.c_int32), ("rgb", C.c_int32), ("scale", C.c_float), ("_pad", C.c_int32)]


class StereoParams(C.Structure):
    _fields_ = [("fx", C.c_float), ("mbf", C.c_float), ("n_rows", C.c_int32), ("th_high", C.c_float),
                ("th_low", C.c_float), ("size_ref", C.c_float)]


LM_DTYPE = np.dtype([("pos", "<f4", 3), ("size", "<f4"), ("min_dist", "<f4"), ("max_dist", "<f4"), ("normal", "<f4", 3),
                     ("assoc_kp", "<i4"), ("prev_angle", "<f4"), ("skip", "<i4"), ("desc", "u1", 32)])


class FrameView(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mbf", C.c_float),
                ("sensor", C.c_int32), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("size_ref", C.c_float), ("n", C.c_int32),
                ("kps", C.c_void_p), ("desc", C.c_void_p), ("uR", C.c_void_p), ("kp_lm_obs", C.c_void_p)]


class VocabTree(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("levels", C.c_int32), ("child_begin", C.c_void_p), ("child_count", C.c_void_p),
                ("desc", C.c_void_p), ("word_id", C.c_void_p), ("weight", C.c_void_p), ("orig_id", C.c_void_p)]


class ProjParams(C.Structure):
    _fields_ = [("th", C.c_float), ("score_threshold", C.c_float), ("second_best_ratio", C.c_float),
                ("frac_smaller", C.c_float), ("frac_larger", C.c_float),
                ("use_distance", C.c_int32), ("use_stereo", C.c_int32), ("check_rotation", C.c_int32),
                ("use_prev_matched", C.c_int32), ("use_viewing_angle", C.c_int32), ("max_view_angle", C.c_float),
                ("use_reprojection", C.c_int32), ("reproj_threshold", C.c_float), ("sigma_ref", C.c_float), ("first_wins", C.c_int32),
                ("dist_is_invariance_range", C.c_int32)]

    def __init__(self, th=3.0, score_threshold=100.0, second_best_ratio=0.6, frac_smaller=0.5, frac_larger=1.5, use_distance=1, use_stereo=1,
                 check_rotation=0, use_prev_matched=1, use_viewing_angle=0, max_view_angle=1.047, use_reprojection=0, reproj_threshold=5.99,
                 sigma_ref=1.0, first_wins=0, dist_is_invariance_range=0):
        super().__init__(th, score_threshold, second_best_ratio, frac_smaller, frac_larger, use_distance, use_stereo, check_rotation,
                         use_prev_matched, use_viewing_angle, max_view_angle, use_reprojection, reproj_threshold, sigma_ref, first_wins, dist_is_invariance_range)


class HsError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("hyslam_amd: %s (status %d)" % (msg, status))
        self.status = status


# every symbol include/hyslam_amd.h declares (tests check the library exports all of them)
EXPORTS = [
    "hs_version", "hs_status_string", "hs_device_count", "hs_orb_default_params", "hs_orb_create", "hs_orb_destroy",
    "hs_orb_last_error", "hs_orb_get_levels", "hs_orb_get_device", "hs_orb_get_scale_factor", "hs_orb_get_scale_tables",
    "hs_orb_max_keypoints", "hs_orb_reserve", "hs_orb_extract", "hs_orb_extract_batch", "hs_orb_extract_batch_device",
    "hs_preprocess_size", "hs_preprocess_device", "hs_orb_extract_camera_batch", "hs_orb_submit_camera_batch",
    "hs_host_alloc", "hs_host_free", "hs_orb_submit_batch", "hs_orb_wait", "hs_orb_cancel", "hs_ticket_frames_copied",
    "hs_stereo_match", "hs_stereo_match_batch_device", "hs_stereo_frontend_batch_device", "hs_orb_set_lanes", "hs_orb_set_split", "hs_orb_synchronize",
    "hs_frame_grid", "hs_search_by_projection", "hs_search_by_projection_device", "hs_frame_publish", "hs_frame_find", "hs_frame_release", "hs_frame_info", "hs_frame_cache_clear", "hs_search_by_projection_frame", "hs_stereo_match_frames", "hs_search_by_projection_sim3", "hs_search_by_sim3", "hs_search_by_bow", "hs_search_by_bow_ex", "hs_search_by_bow_legacy", "hs_search_for_initialization",
    "hs_vocab_last_error", "hs_vocab_load", "hs_vocab_from_tree", "hs_vocab_save", "hs_vocab_destroy", "hs_vocab_get_tree", "hs_vocab_info",
    "hs_vocab_upload", "hs_vocab_dev_destroy", "hs_vocab_dev_groups", "hs_bow_transform_device", "hs_records_bow_match_device", "hs_bow_transform", "hs_hamming_knn2", "hs_hamming_knn2_device",
    "hs_record_bytes", "hs_record_offsets", "hs_records_knn2_device",
    "hs_comm_available", "hs_comm_unavailable_reason", "hs_orb_borrowers", "hs_comm_get_unique_id", "hs_comm_create", "hs_comm_destroy", "hs_comm_rccl_ranks", "hs_comm_rccl_rank", "hs_comm_rccl_version", "hs_comm_world", "hs_comm_rank", "hs_comm_last_error", "hs_comm_allgather_records",
    "hs_orb_stage_launches", "hs_orb_profile_begin", "hs_orb_profile_pause", "hs_orb_profile_end", "hs_debug_stream_copy",
    "hs_orb_debug_level", "hs_orb_set_debug", "hs_orb_debug_candidates", "hs_orb_debug_selected",
]

_lib = None

# source files each kernel family is compiled from: the committed rocprofv3 counter passes (profiles/*_{hbm_traffic,sq_counters}.json) are stamped
# with these digests and bench.py only replays a counter whose kernel sources are unchanged
KERNEL_SOURCES = {
    "k_fast_rows": ["kernels_fast.hip", "hs_internal.h"],
    "k_resize": ["kernels_pyramid.hip", "hs_internal.h"],
    "k_describe": ["kernels_describe.hip", "lean_sincos.h", "hs_internal.h"],
    "k_quadtree": ["kernels_quadtree.hip", "hs_internal.h"],
    "k_qt": ["kernels_quadtree.hip", "hs_internal.h"],
    "k_stereo": ["kernels_stereo.hip", "hs_internal.h"],
}


def source_digests():
    """{kernel-name prefix: sha256[:16] of the source files it is compiled from} (hyslam_amd/csrc travels with the repository snapshot)"""
    import hashlib
    out = {}
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for k, files in KERNEL_SOURCES.items():
        h = hashlib.sha256()
        for f in files:
            try:
                h.update(open(os.path.join(src, f), "rb").read())
            except OSError:
                h.update(b"missing:" + f.encode())
        out[k] = h.hexdigest()[:16]
    return out



def lib():
    """Load the HIP library (once).  Raises if it is absent — build it with `python -c 'import __graft_entry__ as g; g.build()'`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("hyslam_amd: %s not found; the HIP extension must be built (make -C hyslam_amd/csrc). "
                          "There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
    L.hs_version.restype = C.c_char_p
    L.hs_status_string.restype = C.c_char_p
    L.hs_status_string.argtypes = [C.c_int]
    L.hs_device_count.argtypes = [C.POINTER(C.c_int)]
    L.hs_orb_default_params.argtypes = [C.POINTER(OrbParams)]
    L.hs_orb_default_params.restype = None
    L.hs_orb_create.argtypes = [C.POINTER(OrbParams), C.c_int, C.POINTER(vp)]
    L.hs_orb_destroy.argtypes = [vp]
    L.hs_orb_destroy.restype = None
    L.hs_orb_last_error.argtypes = [vp]
    L.hs_orb_last_error.restype = C.c_char_p
    L.hs_orb_get_levels.argtypes = [vp]
    L.hs_orb_get_device.argtypes = [vp]
    L.hs_orb_set_split.argtypes = [vp, C.c_int]
    L.hs_orb_get_scale_factor.argtypes = [vp]
    L.hs_orb_get_scale_factor.restype = f32
    L.hs_orb_get_scale_tables.argtypes = [vp, vp, vp, vp, vp, vp]
    L.hs_orb_max_keypoints.argtypes = [vp]
    L.hs_orb_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.hs_orb_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]
    L.hs_orb_extract_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]
    L.hs_orb_extract_batch_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, sz, sz, vp, vp, vp, C.c_int, vp]
    L.hs_preprocess_size.argtypes = [C.c_int, C.c_int, f32, vp, vp]
    L.hs_preprocess_size.restype = None
    L.hs_preprocess_device.argtypes = [vp, vp, C.c_int, C.c_int, sz, sz, C.c_int, C.POINTER(PreprocessParams), vp, sz, sz, vp]
    L.hs_orb_extract_camera_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, sz, C.POINTER(PreprocessParams), vp, vp, C.c_int, vp, vp]
    L.hs_orb_submit_camera_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, sz, C.POINTER(PreprocessParams), vp, C.POINTER(i32)]
    L.hs_stereo_match.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(StereoParams), vp, vp]
    L.hs_stereo_match_batch_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(StereoParams), vp, vp, vp]
    L.hs_stereo_frontend_batch_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, sz, sz,
                                                  vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(StereoParams), vp, vp, vp]
    L.hs_orb_set_lanes.argtypes = [vp, C.c_int]
    L.hs_orb_synchronize.argtypes = [vp, vp]
    L.hs_frame_grid.argtypes = [vp, C.POINTER(FrameView), vp]
    L.hs_search_by_projection.argtypes = [vp, C.POINTER(FrameView), vp, C.c_int, C.POINTER(ProjParams), vp, vp, vp]
    L.hs_search_by_projection_device.argtypes = [vp, C.POINTER(FrameView), vp, C.c_int, C.POINTER(ProjParams), vp, vp, vp, vp]
    u64 = C.c_uint64
    L.hs_frame_publish.argtypes = [vp, C.c_int, vp, C.c_int, C.POINTER(u64)]
    L.hs_frame_find.argtypes = [C.c_int, vp, C.c_int, C.POINTER(u64)]
    L.hs_frame_release.argtypes = [C.c_int, u64]
    L.hs_frame_info.argtypes = [C.c_int, u64, vp]
    L.hs_frame_cache_clear.argtypes = [C.c_int]
    L.hs_search_by_projection_frame.argtypes = [vp, u64, C.POINTER(FrameView), vp, C.c_int, C.POINTER(ProjParams), vp, vp, vp]
    L.hs_stereo_match_frames.argtypes = [vp, u64, u64, C.POINTER(StereoParams), vp, vp]
    L.hs_search_by_projection_sim3.argtypes = [vp, C.POINTER(FrameView), vp, vp, C.c_int, C.c_int, f32, vp, vp, vp]
    L.hs_search_by_sim3.argtypes = [vp, C.POINTER(FrameView), vp, C.POINTER(FrameView), vp, f32, vp, vp, f32, f32, vp, vp]
    L.hs_search_by_bow.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                   vp, f32, f32, C.c_int, vp, vp]
    L.hs_search_by_bow_ex.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                      vp, vp, vp, f32, f32, f32, f32, C.c_int, vp, vp]
    L.hs_search_by_bow_legacy.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                          vp, vp, f32, f32, C.c_int, vp, vp]
    L.hs_search_for_initialization.argtypes = [vp, vp, vp, C.c_int, C.POINTER(FrameView), vp, C.c_int, f32, f32, vp, vp]
    L.hs_vocab_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.hs_vocab_from_tree.argtypes = [C.POINTER(VocabTree), C.c_int, C.POINTER(vp)]
    L.hs_vocab_save.argtypes = [vp, C.c_char_p]
    L.hs_vocab_destroy.argtypes = [vp]
    L.hs_vocab_destroy.restype = None
    L.hs_vocab_get_tree.argtypes = [vp, C.POINTER(VocabTree)]
    L.hs_vocab_info.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.hs_vocab_upload.argtypes = [vp, C.POINTER(VocabTree), C.c_int, C.POINTER(vp)]
    L.hs_vocab_dev_destroy.argtypes = [vp]
    L.hs_vocab_dev_destroy.restype = None
    L.hs_vocab_dev_groups.argtypes = [vp]
    L.hs_bow_transform_device.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]
    L.hs_records_bow_match_device.argtypes = [vp, vp, vp, sz, C.c_int, C.c_int, C.c_int, f32, f32, C.c_int, vp, vp, vp]
    L.hs_bow_transform.argtypes = [vp, C.POINTER(VocabTree), vp, C.c_int, C.c_int, vp, vp, vp]
    L.hs_hamming_knn2.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
    L.hs_hamming_knn2_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    L.hs_record_bytes.argtypes = [C.c_int]
    L.hs_record_bytes.restype = sz
    L.hs_record_offsets.argtypes = [C.c_int, vp, vp, vp]
    L.hs_record_offsets.restype = None
    L.hs_records_knn2_device.argtypes = [vp, vp, sz, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.hs_host_alloc.argtypes = [sz, C.POINTER(vp)]
    L.hs_host_free.argtypes = [vp]
    L.hs_host_free.restype = None
    L.hs_orb_submit_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.POINTER(i32)]
    L.hs_orb_wait.argtypes = [vp, i32, vp, vp, vp, C.c_int, vp, vp]
    L.hs_orb_cancel.argtypes = [vp, i32]
    L.hs_ticket_frames_copied.argtypes = [vp, i32]
    L.hs_comm_get_unique_id.argtypes = [vp]
    L.hs_vocab_last_error.argtypes = []
    L.hs_vocab_last_error.restype = C.c_char_p
    L.hs_comm_available.argtypes = []
    L.hs_comm_unavailable_reason.argtypes = []
    L.hs_comm_unavailable_reason.restype = C.c_char_p
    L.hs_orb_borrowers.argtypes = [vp]
    L.hs_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.hs_comm_destroy.argtypes = [vp]
    L.hs_comm_destroy.restype = None
    L.hs_comm_world.argtypes = [vp]
    L.hs_comm_rank.argtypes = [vp]
    L.hs_comm_last_error.argtypes = [vp]
    L.hs_comm_last_error.restype = C.c_char_p
    L.hs_comm_allgather_records.argtypes = [vp, vp, vp, sz, vp]
    L.hs_debug_stream_copy.argtypes = [vp, vp, vp, sz, C.c_int, vp]
    L.hs_orb_stage_launches.argtypes = [vp, C.c_int]
    L.hs_orb_profile_begin.argtypes = [vp]
    L.hs_orb_profile_pause.argtypes = [vp]
    L.hs_orb_profile_end.argtypes = [vp, vp, vp]
    L.hs_orb_debug_level.argtypes = [vp, C.c_int, C.c_int, vp, sz, vp, vp]
    L.hs_orb_set_debug.argtypes = [vp, C.c_int]
    L.hs_orb_debug_candidates.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
    L.hs_orb_debug_selected.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
    _lib = L
    return L


def check(handle, status):
    if status != HS_OK:
        L = lib()
        msg = L.hs_status_string(status).decode()
        if handle:
            detail = L.hs_orb_last_error(handle).decode()
            if detail:
                msg += ": " + detail
        raise HsError(status, msg)
