"""ctypes binding of libreal3daug_hip.so (the C ABI declared in include/real3daug_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at load time, and
every operation raises when no GPU is visible.  PyTorch-ROCm is used for device memory, streams
and nothing else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# R3D_LIB: a diagnostic build of the same library (e.g. `make STAMPS=1`), never a different implementation
LIB_PATH = os.environ.get("R3D_LIB") or os.path.join(_HERE, "libreal3daug_hip.so")

R3D_OK = 0
E_ARG, E_HIP, E_WORKSPACE, E_IO = -1, -2, -3, -4
S_NONFINITE, S_ROW_RANGE, S_COL_RANGE, S_SAMPLE_TOO_LARGE, S_CAPACITY, S_FAR_OVERFLOW, S_WINDOW_TOO_LARGE = 1, 2, 4, 8, 16, 32, 64
S_CHAIN_TIMEOUT = 128
S_ORDER_PROMISE = 256
# limits of the batched kernels that the reference does not have (insertion.py:455-482 takes a sample of any size and any
# number of returns beyond 500 m): a frame flagged with one of these -- and with nothing else -- is run once more, alone,
# through the Level-1 kernels by every mirror that drives Level 2 (batch.augment_batch, streaming.StreamedAugmenter)
S_REDO_LEVEL1 = S_WINDOW_TOO_LARGE | S_SAMPLE_TOO_LARGE | S_FAR_OVERFLOW
STATUS_TEXT = {
    S_NONFINITE: "NaN/Inf coordinate or a point at the origin (reference: int() raises, insertion.py:104)",
    S_ROW_RANGE: "Rows in FoV went something wrong (assert insertion.py:110)",
    S_COL_RANGE: "Column in FoV went something wrong (assert insertion.py:112)",
    S_SAMPLE_TOO_LARGE: "sample has more than R3D_MAX_SAMPLE points (Level 2 only; the mirrors redo such a frame through Level 1)",
    S_CAPACITY: "merged cloud or insert log exceeds its capacity",
    S_FAR_OVERFLOW: "more than R3D_FAR_CAP pixels deeper than 500 m (Level 2 only; the mirrors redo such a frame through Level 1)",
    S_WINDOW_TOO_LARGE: "the insert's window of the range image and the sample's arrays do not fit one CU's LDS",
    S_CHAIN_TIMEOUT: "insert_many: the chain of a scene's slots was left unfinished",
    S_ORDER_PROMISE: "R3D_B_FILE_ORDER was set between a begin that numbered this scene's points anew and its finish / export",
}
K_BOUNDS, K_PREPARE, K_PROJECT, K_ALIVE_WRITE = 1, 2, 3, 5
NUMROW, NUMCOLUMN = 112, 1440
MAX_SAMPLE = 65535
FAR_CAP = 1024
B_FILE_ORDER = 2048      # r3d_batch_t.reserved: the clouds come in a LiDAR file order (no look at the chunk boxes, no virtual order)
B_SLOT_LAUNCHES = 65536  # r3d_batch_t.reserved: r3d_batch_insert_many makes one launch per slot (no chain inside a kernel)
MAX_CHAIN = 64           # insert slots one launch of the chain kernel takes (kMaxChain in csrc/r3d_batch.hpp); insert_many splits longer lists


class R3DError(RuntimeError):
    pass


class BatchDesc(C.Structure):
    """Mirror of r3d_batch_t."""
    _fields_ = [
        ("B", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("reserved", C.c_int32),
        ("cap", C.c_int64), ("log_cap", C.c_int64),
        ("xyzi", C.c_void_p), ("label", C.c_void_p), ("pix", C.c_void_p),
        ("n_head", C.c_void_p), ("n_total", C.c_void_p), ("tail_ref", C.c_void_p),
        ("log5", C.c_void_p), ("log_birth", C.c_void_p), ("n_log", C.c_void_p),
        ("bounds", C.c_void_p), ("far_pix", C.c_void_p),
        ("n_far", C.c_void_p), ("rebase", C.c_void_p), ("status", C.c_void_p),
        ("out_xyzi", C.c_void_p), ("out_label", C.c_void_p), ("n_out", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class PlaceQuery(C.Structure):
    """Mirror of r3d_place_query_t."""
    _fields_ = [
        ("scene", C.c_void_p), ("orig", C.c_void_p), ("boxes", C.c_void_p), ("sample", C.c_void_p), ("map", C.c_void_p),
        ("scene_ranges", C.c_void_p), ("orig_ranges", C.c_void_p),
        ("n_scene", C.c_int64), ("n_orig", C.c_int64),
        ("scene_ld", C.c_int32), ("scene_label_col", C.c_int32), ("orig_ld", C.c_int32), ("orig_label_col", C.c_int32),
        ("n_boxes", C.c_int32), ("m", C.c_int32), ("map_rows", C.c_int32), ("map_cols", C.c_int32),
        ("n_ok_labels", C.c_int32), ("cand_cap", C.c_int32),
        ("ok_labels", C.c_int32 * 32), ("ok_map", C.c_uint64 * 4),
        ("anno", C.c_double * 10), ("pose", C.c_double * 8), ("map_move", C.c_double * 2),
        ("cand_off", C.c_int64), ("cand_stride", C.c_int64),
        ("flavour", C.c_int32), ("collide_label", C.c_int32), ("collide_dz", C.c_double),
        ("scene_label", C.c_void_p), ("scene_alive", C.c_void_p), ("scene_tail_ref", C.c_void_p), ("scene_log5", C.c_void_p),
        ("scene_head", C.c_int64),
        ("orig_label", C.c_void_p),
    ]


PLACE_ROTATIONS, PLACE_SURFACE_CAP, PLACE_MAX_OK_LABELS = 360, 128, 32
PS_SURFACE_OVERFLOW, PS_NONFINITE, PS_BAD_DESCRIPTOR = 1, 2, 4
PQ_POINTWISE_ROTATION, PQ_MAP_NEEDS_POINT, PQ_COLLIDE_LABEL, PQ_COLLIDE_ABOVE, PQ_SCENE_SLAB, PQ_ORIG_SLAB = 1, 2, 4, 8, 16, 32
PF_ON_SURFACE, PF_NEAR_ROAD, PF_SCENE_IN_BOX, PF_SAMPLE_IN_BOX, PF_POSSIBLE = 1, 2, 4, 8, 16

_P = C.c_void_p
_SIGNATURES = {
    "r3d_version": (C.c_int, []),
    "r3d_last_error": (C.c_char_p, []),
    "r3d_build_info": (C.c_char_p, []),
    "r3d_add_space_for_spherical": (C.c_int, [_P, C.c_int64, _P, _P]),
    "r3d_fill_spherical": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "r3d_front_view_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "r3d_geometrical_front_view": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                             C.c_int32, _P, _P, _P, C.c_size_t, _P, _P]),
    "r3d_geometrical_front_view_grid": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                                  C.c_int32, _P, _P, _P, C.c_size_t, _P, _P]),
    "r3d_class_closing": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "r3d_smooth_out": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "r3d_occlusion_merge_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "r3d_occlusion_merge": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int32, C.c_int32,
                                      _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "r3d_occlusion_merge_grid": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int32, C.c_int32, C.c_int32,
                                           _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "r3d_remove_space_for_spherical": (C.c_int, [_P, C.c_int64, _P, _P, _P, C.c_int32, _P]),
    "r3d_batch_workspace_bytes": (C.c_size_t, [C.POINTER(BatchDesc)]),
    "r3d_batch_create": (C.c_int, [C.POINTER(BatchDesc), _P]),
    "r3d_batch_begin": (C.c_int, [C.POINTER(BatchDesc), _P, _P]),
    "r3d_batch_begin_f64": (C.c_int, [C.POINTER(BatchDesc), _P, _P, _P]),
    "r3d_batch_begin_xyz": (C.c_int, [C.POINTER(BatchDesc), _P, _P, _P]),
    "r3d_batch_insert": (C.c_int, [C.POINTER(BatchDesc), _P, _P, _P, _P, C.c_int32, _P, _P, _P]),
    "r3d_batch_insert_first": (C.c_int, [C.POINTER(BatchDesc), _P, C.c_int64, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         _P, _P, _P]),
    "r3d_batch_finish": (C.c_int, [C.POINTER(BatchDesc), _P, C.c_int32, _P]),
    "r3d_batch_launch_one": (C.c_int, [C.POINTER(BatchDesc), C.c_int32, _P]),
    "r3d_batch_debug_counters": (C.c_int, [C.POINTER(BatchDesc), _P, C.c_int32, _P]),
    "r3d_batch_insert_many": (C.c_int, [C.POINTER(BatchDesc), C.c_int32, _P, _P, _P, _P, C.c_int32, _P, _P, _P]),
    "r3d_batch_export_rows": (C.c_int, [C.POINTER(BatchDesc), _P, _P, _P]),
    "r3d_batch_export_pix": (C.c_int, [C.POINTER(BatchDesc), _P, _P]),
    "r3d_batch_export_alive": (C.c_int, [C.POINTER(BatchDesc), _P, _P]),
    "r3d_batch_point_order": (C.c_int, [C.POINTER(BatchDesc), _P, _P]),
    "r3d_batch_adopt_rejected": (C.c_int, [C.POINTER(BatchDesc), _P, _P]),
    "r3d_places_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "r3d_places_release": (C.c_int, []),
    "r3d_cut_boxes_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "r3d_cut_boxes": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, C.c_int64,
                                _P, C.c_size_t, _P]),
    "r3d_map_bounds": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_double), _P, _P]),
    "r3d_map_splat": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int32,
                                C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_double, C.c_double,
                                C.c_int32, C.c_int32, C.c_int64, _P, _P, _P]),
    "r3d_map_finish": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "r3d_od_maps": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "r3d_places_chunk_ranges": (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P]),
    "r3d_places_chunk_ranges_f32": (C.c_int, [_P, C.c_int64, _P, _P]),
    "r3d_host_pack_frames": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, _P, _P, C.c_int32, C.c_int32]),
    "r3d_host_read_frames": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P, C.c_int32, C.c_int32]),
    "r3d_host_pack_frames_xyz": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, _P, _P, _P, C.c_int32, C.c_int32]),
    "r3d_host_read_frames_xyz": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, _P, _P, C.c_int32, C.c_int32]),
    "r3d_host_write_frames": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, C.c_int64, _P, _P, C.c_int64, C.c_int32, _P, C.c_int32]),
    "r3d_host_write_delta_frames": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int64, _P, C.c_int32, _P,
                                              C.c_int32]),
    "r3d_host_append_text_files": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32]),
    "r3d_batch_export_delta": (C.c_int, [C.POINTER(BatchDesc), _P, _P, _P, C.c_int64, _P, _P]),
    "r3d_host_merge_frames": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int64, _P, C.c_int32, _P, _P, C.c_int64, _P, _P,
                                        C.c_int64, C.c_int32, C.c_int32]),
    "r3d_find_possible_places": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                           C.POINTER(C.c_double), C.c_int32, _P, _P, _P, _P, _P, C.c_int32, _P,
                                           _P, C.c_size_t, _P]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load the shared library (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7; importing torch
        # first makes the loader bind this library's NEEDED libamdhip64.so.7 to that same copy, so
        # tensors, streams and our kernels share one runtime (loaded the other way round, two
        # runtimes coexist and ours reports "no ROCm-capable device").
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != R3D_OK:
        raise R3DError(f"{what or 'r3d call'} failed (code {rc}): {load().r3d_last_error().decode()}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise R3DError("no GPU visible: the Real3D-Aug HIP path has no CPU fallback")
    return torch


def stream_ptr():
    """The current stream of the CURRENT device: callers make the device of their tensors current
    first (``on(device)``), the C library launches on HIP's current device."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on(device):
    """Context manager that makes `device` (a torch.device, string, index or CUDA tensor) the current
    one: every library call must run with the device of its pointers current."""
    import torch
    if isinstance(device, torch.Tensor):
        device = device.device
    return torch.cuda.device(torch.device(device))


def on_own_device(method):
    """Decorator for methods of objects with a ``.device``: run the method with that device current."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        with on(self.device):
            return method(self, *args, **kwargs)
    return wrapped


def needs_level1(bits) -> bool:
    """A Level-2 status that only says "beyond this path's limits" (see S_REDO_LEVEL1)."""
    bits = int(bits)
    return bool(bits & S_REDO_LEVEL1) and not (bits & ~S_REDO_LEVEL1)


def raise_status(bits: int, where: str):
    """Device status bits -> the exception the reference would have raised."""
    if not bits:
        return
    msgs = [t for b, t in STATUS_TEXT.items() if bits & b]
    if bits & (S_ROW_RANGE | S_COL_RANGE):
        raise AssertionError(f"{where}: " + "; ".join(msgs))
    raise ValueError(f"{where}: " + "; ".join(msgs))
