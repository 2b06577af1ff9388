"""ctypes loader of flame_ros_amd/libflame_hip.so (the C ABI of include/flame_hip.h).

The HIP library is the product: there is no Python/CPU fallback.  `load()` raises if the shared
library is missing; compute entry points return FLAME_HIP_ERR_NODEVICE when no GPU is present.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (FLAME_HIP_LIB: dev aid -- A/B of kernel variants built side by side by tools/exp/build_variant.sh)
LIB_PATH = os.environ.get("FLAME_HIP_LIB") or os.path.join(HERE, "libflame_hip.so")

ERR_ARG, ERR_STATE, ERR_NAN, ERR_ALLOC, ERR_NODEVICE, ERR_NORCCL, ERR_HIP, ERR_RCCL = -1, -2, -3, -4, -5, -6, -1000, -3000
PATH_AUTO, PATH_GLOBAL, PATH_TILE = 0, 1, 2
IMG_WIREFRAME, IMG_FEATURES, IMG_NORMALS, IMG_IDEPTHMAP = 0, 1, 2, 3


class Params(C.Structure):
    """flame_hip_params == flame::Params::rparams (reference src/flame_offline_tum.cc:242-245)."""
    _fields_ = [("data_factor", C.c_float), ("step_x", C.c_float), ("step_q", C.c_float),
                ("theta", C.c_float), ("x_min", C.c_float), ("x_max", C.c_float)]


class TriParams(C.Structure):
    """flame_hip_tri_params (reference src/flame_offline_tum.cc:168-192)."""
    _fields_ = [("do_oblique_triangle_filter", C.c_int32), ("oblique_normal_thresh", C.c_float),
                ("oblique_idepth_diff_factor", C.c_float), ("oblique_idepth_diff_abs", C.c_float),
                ("do_edge_length_filter", C.c_int32), ("edge_length_thresh", C.c_float),
                ("do_idepth_triangle_filter", C.c_int32), ("min_triangle_idepth", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class SyncParams(C.Structure):
    """flame_hip_sync_params (reference src/flame_offline_tum.cc:234-249, yaml :89-92)."""
    _fields_ = [("adaptive_data_weights", C.c_int32), ("rescale_data", C.c_int32),
                ("init_with_prediction", C.c_int32), ("idepth_var_max_graph", C.c_float),
                # [UPSTREAM-RECALL] switches, all-zero = default (include/flame_hip.h)
                ("edge_weight_rule", C.c_int32), ("alpha_gain", C.c_float), ("beta_gain", C.c_float)]


class TileDesc(C.Structure):
    """Mirror of flamehip::TileDesc (csrc/common.h) for the plan debug hook."""
    _fields_ = [(n, C.c_int32) for n in
                ("vstart", "n_own", "n_ext", "estart", "e_own", "e_loc", "n_upd", "depth",
                 "vmap_off", "emap_off", "erec_off", "srow_off", "nslots")] + \
               [("ring_end", C.c_int32 * 17), ("level_end", C.c_int32 * 17)]


# name -> (restype, argtypes): every symbol include/flame_hip.h declares
_VP, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
SYMBOLS = {
    "flame_hip_graph_create": (C.c_int, [C.POINTER(_VP), C.c_int, _I32, _I32, _I32]),
    "flame_hip_graph_destroy": (None, [_VP]),
    "flame_hip_graph_resize": (C.c_int, [_VP, _I32, _I32, _I32]),
    "flame_hip_set_option": (C.c_int, [_VP, C.c_char_p, _I32]),
    "flame_hip_get_info": (C.c_int, [_VP, C.c_char_p, C.POINTER(_I64)]),
    "flame_hip_graph_upload": (C.c_int, [_VP] + [_VP] * 8),
    "flame_hip_graph_sync": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, C.POINTER(C.c_float)]),
    "flame_hip_feature_gate": (_I32, [_I32, _VP, C.c_float, _VP]),
    "flame_hip_delaunay": (C.c_int, [_VP, _I32, _VP, _I32, _VP, C.POINTER(_I32)]),
    "flame_hip_delaunay_list": (C.c_int, [_VP, _I32, _VP]),
    "flame_hip_graph_edges": (C.c_int, [_VP, _VP]),
    "flame_hip_scale_state": (C.c_int, [_VP, C.c_float]),
    "flame_hip_graph_update_data": (C.c_int, [_VP, _VP, _VP, _VP]),
    "flame_hip_graph_upload_batch": (C.c_int, [_VP, _I32, _VP] + [_VP] * 8),
    "flame_hip_set_state": (C.c_int, [_VP] + [_VP] * 7),
    "flame_hip_solve": (C.c_int, [_VP, C.POINTER(Params), _I32, _VP]),
    "flame_hip_graph_filter": (C.c_int, [_VP, _I32, _I32]),
    "flame_hip_sync": (C.c_int, [_VP]),
    "flame_hip_last_solve_ms": (C.c_int, [_VP, C.POINTER(C.c_float), C.POINTER(_I32)]),
    "flame_hip_costs": (C.c_int, [_VP, C.POINTER(Params), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "flame_hip_costs_masked": (C.c_int, [_VP, C.POINTER(Params), _VP, _VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "flame_hip_triangles": (C.c_int, [_VP, _VP, C.POINTER(TriParams), _VP, _VP, _VP]),
    "flame_hip_frame_results": (C.c_int, [_VP, C.POINTER(Params), C.c_float, _VP, C.POINTER(TriParams),
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), _VP, _VP, _VP, _VP,
                                          C.POINTER(C.c_float)]),
    "flame_hip_debug_image": (C.c_int, [_VP, _I32, _VP, C.POINTER(TriParams), C.c_float, _I32, _VP, _VP, _VP]),
    "flame_hip_mesh": (C.c_int, [_VP, _VP, C.POINTER(TriParams), _VP, _VP, C.POINTER(_I32)]),
    "flame_hip_depthmaps": (C.c_int, [_VP, _VP, C.POINTER(TriParams), _I32, C.c_float, C.c_float, _VP, _VP, _VP]),
    "flame_hip_download": (C.c_int, [_VP] + [_VP] * 4),
    "flame_hip_download_bar": (C.c_int, [_VP] + [_VP] * 3),
    "flame_hip_halo_register": (C.c_int, [_VP, _I32, _VP, _I32, _VP, _I32, _VP, _I32, _VP]),
    "flame_hip_halo_bytes": (C.c_int, [_VP, C.POINTER(_I64), C.POINTER(_I64)]),
    "flame_hip_halo_pack": (C.c_int, [_VP, _VP, _VP]),
    "flame_hip_halo_unpack": (C.c_int, [_VP, _VP, _VP]),
    "flame_hip_halo_view_get": (C.c_int, [_VP, _VP, _VP]),
    "flame_hip_halo_written": (C.c_int, [_VP]),
    "flame_hip_comm_create_local": (C.c_int, [C.POINTER(_VP), C.c_int, C.c_int, C.c_int]),
    "flame_hip_part_peer_blob": (C.c_int, [_VP, _VP]),
    "flame_hip_part_peer_connect": (C.c_int, [_VP, _VP]),
    "flame_hip_rccl_available": (C.c_int, []),
    "flame_hip_comm_get_unique_id": (C.c_int, [_VP]),
    "flame_hip_comm_create": (C.c_int, [C.POINTER(_VP), C.c_int, C.c_int, C.c_int, _VP]),
    "flame_hip_comm_destroy": (None, [_VP]),
    "flame_hip_comm_stream": (_VP, [_VP]),
    "flame_hip_comm_info": (C.c_int, [_VP, C.c_char_p, C.POINTER(_I64)]),
    "flame_hip_state_snapshot": (C.c_int, [_VP, _VP]),
    "flame_hip_state_rollback": (C.c_int, [_VP, _VP]),
    "flame_hip_persist_take_error": (C.c_int, [_VP, C.POINTER(_I32)]),
    "flame_hip_part_create": (C.c_int, [C.POINTER(_VP), _VP, _I32, _I32, _I32, _I32, _I32, _I32] + [_VP] * 7),
    "flame_hip_part_destroy": (None, [_VP]),
    "flame_hip_part_solve": (C.c_int, [_VP, C.POINTER(Params), _I32]),
    "flame_hip_part_sync": (C.c_int, [_VP]),
    "flame_hip_part_update_data": (C.c_int, [_VP, _VP, _VP, _VP]),
    "flame_hip_part_costs": (C.c_int, [_VP, C.POINTER(Params), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "flame_hip_part_gather": (C.c_int, [_VP] + [_VP] * 4),
    "flame_hip_part_set_option": (C.c_int, [_VP, C.c_char_p, _I32]),
    "flame_hip_part_info": (C.c_int, [_VP, C.c_char_p, _I32, C.POINTER(_I64)]),
    "flame_hip_part_array": (_I64, [_VP, C.c_char_p, _I32, _VP, _I64]),
    "flame_hip_debug_plan_array": (_I64, [_VP, C.c_char_p, _VP, _I64]),
    "flame_hip_strerror": (C.c_char_p, [C.c_int]),
    "flame_hip_version": (C.c_int, []),
}

_lib = None


class FlameHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = load().flame_hip_strerror(code)
        super().__init__("%s failed: %d (%s)" % (where, code, msg.decode() if msg else "?"))


def load():
    """Load libflame_hip.so.  torch (if used in this process) must be imported first so that both
    resolve the same libamdhip64.so.7; this function imports it when it is installed."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `python flame_ros_amd/build.py` (hipcc, gfx950). There is no CPU "
                          "fallback." % LIB_PATH)
    try:  # share torch's HIP runtime instead of loading a second copy of libamdhip64
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError = missing export
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(code, where):
    if code != 0:
        raise FlameHipError(code, where)
