"""ctypes binding of ``libeyoc_hip.so`` (declared in ``include/eyoc_hip.h``).

The product path has no CPU fallback: if the shared library is missing or a call fails this module
raises.  ``torch`` is used only to own device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EYOC_HIP_LIB") or os.path.join(_HERE, "lib", "libeyoc_hip.so")   # override: diagnostics builds

MAX_LEVELS = 4
MAP_S1, MAP_DOWN, MAP_UP = 0, 1, 2


class MapsInfo(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("rows", C.c_int32 * MAX_LEVELS),
                ("pairs_s1", C.c_int64 * MAX_LEVELS), ("pairs_down", C.c_int64 * MAX_LEVELS),
                ("pairs_up", C.c_int64 * MAX_LEVELS), ("pairs_conv1", C.c_int64)]


class ModelDesc(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("conv1_kernel_size", C.c_int32),
                ("normalize_feature", C.c_int32), ("channels", C.c_int32 * 5), ("tr_channels", C.c_int32 * 5),
                ("bn_eps", C.c_float), ("expanded", C.c_int32)]


class LayerParams(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kernel", C.POINTER(C.c_float)), ("K", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("bias", C.POINTER(C.c_float)), ("bn_weight", C.POINTER(C.c_float)),
                ("bn_bias", C.POINTER(C.c_float)), ("bn_mean", C.POINTER(C.c_float)),
                ("bn_var", C.POINTER(C.c_float))]


class RansacParams(C.Structure):
    _fields_ = [("max_distance", C.c_float), ("edge_similarity", C.c_float), ("max_iteration", C.c_int32),
                ("seed", C.c_uint32)]


class RansacResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("inliers", C.c_int32), ("best_hypothesis", C.c_int32),
                ("survivors", C.c_int32), ("inlier_rmse", C.c_float)]


class Sc2pcrParams(C.Structure):
    _fields_ = [("inlier_threshold", C.c_float), ("d_thre", C.c_float), ("ratio", C.c_float),
                ("nms_radius", C.c_float), ("num_iterations", C.c_int32), ("max_points", C.c_int32),
                ("k1", C.c_int32), ("k2", C.c_int32)]


_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
# name -> (restype, argtypes); every symbol include/eyoc_hip.h declares
PROTOTYPES = {
    "eyoc_version": (_i, []),
    "eyoc_last_error": (C.c_char_p, []),
    "eyoc_create": (_i, [_i, C.POINTER(_vp)]),
    "eyoc_destroy": (_i, [_vp]),
    "eyoc_maps_workspace_bytes": (_sz, [_i]),
    "eyoc_maps_build": (_i, [_vp, _vp, _i, _vp, _sz, _vp, C.POINTER(_vp)]),
    "eyoc_maps_build_ordered": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _i, C.POINTER(_vp)]),
    "eyoc_maps_free": (_i, [_vp]),
    "eyoc_maps_internal_order": (_i, [_vp, _i]),
    "eyoc_maps_lazy_tables": (_i, [_vp, _i]),
    "eyoc_maps_fused_levels": (_i, [_vp, _i]),
    "eyoc_spconv_select_down_kernel": (_i, [_vp, _i]),
    "eyoc_maps_order_window_shift": (_i, [_vp, _i]),
    "eyoc_maps_row_order": (_vp, [_vp]),
    "eyoc_maps_copy_row_order": (_i, [_vp, _vp, _vp]),
    "eyoc_maps_rows": (_i, [_vp, _i]),
    "eyoc_maps_coords": (_vp, [_vp, _i]),
    "eyoc_maps_table": (_vp, [_vp, _i, _i]),
    "eyoc_maps_copy_coords": (_i, [_vp, _i, _vp, _vp]),
    "eyoc_maps_copy_table": (_i, [_vp, _i, _i, _vp, _vp]),
    "eyoc_maps_copy_up_order": (_i, [_vp, _i, _vp, _vp]),
    "eyoc_maps_order_min_rows": (_i, [_vp, _i]),
    "eyoc_maps_info": (_i, [_vp, _vp, _i, _vp, C.POINTER(MapsInfo)]),
    "eyoc_voxelize_workspace_bytes": (_sz, [_i]),
    "eyoc_voxelize": (_i, [_vp, _vp, _i, _i, C.c_float, _i, _vp, _vp, C.POINTER(C.c_int), _vp, _sz, _vp]),
    "eyoc_gather_rows": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, C.c_float, _vp, _vp]),
    "eyoc_dotmax": (_i, [_vp, _vp, _vp, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i, _vp, _vp, _vp]),
    "eyoc_knn2": (_i, [_vp, _vp, _vp, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i, _vp, _vp, _vp, _vp]),
    "eyoc_lowe_topk": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "eyoc_pair_filter": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, C.c_float, _vp, _vp, _vp]),
    "eyoc_pair_filter_similarity": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, C.c_float, C.c_float, C.c_double, _vp, _vp,
                                         _vp]),
    "eyoc_spconv_packed_floats": (_sz, [_i, _i, _i]),
    "eyoc_spconv_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "eyoc_spconv_select_kernel": (_i, [_vp, _i]),
    "eyoc_spconv_pack_weights_transposed": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "eyoc_spconv_grad_weight_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "eyoc_spconv_grad_weight": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "eyoc_spconv_pack_weights_split16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "eyoc_spconv_ex": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "eyoc_spconv_select_split16_kernel": (_i, [_vp, _i]),
    "eyoc_spconv_select_st_kernel": (_i, [_vp, _i]),
    "eyoc_spconv_st_split_below": (_i, [_vp, _i]),
    "eyoc_spconv_st_group_rows": (_i, [_vp, _i]),
    "eyoc_spconv_st_ksplit": (_i, [_vp, _i]),
    "eyoc_spconv_local_rulebook_bytes": (_sz, [_i]),
    "eyoc_spconv_build_local_rulebook": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "eyoc_spconv_staged": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "eyoc_split16_encode": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "eyoc_split16_decode": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "eyoc_model_set_math": (_i, [_vp, _i]),
    "eyoc_model_last_math": (_i, [_vp]),
    "eyoc_model_range_check": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "eyoc_model_range_snapshot": (_i, [_vp, _vp, _vp]),
    "eyoc_model_set_progress_event": (_i, [_vp, _i, _vp]),
    "eyoc_model_set_probe": (_i, [_vp, _i]),
    "eyoc_spconv": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "eyoc_spconv_sum": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp]),
    "eyoc_model_blob_floats": (_sz, [C.POINTER(ModelDesc)]),
    "eyoc_model_pack_host": (_i, [C.POINTER(ModelDesc), C.POINTER(LayerParams), _i, _vp, _sz]),
    "eyoc_model_create": (_i, [_vp, C.POINTER(ModelDesc), C.POINTER(LayerParams), _i, _vp, _sz, C.POINTER(_vp)]),
    "eyoc_model_destroy": (_i, [_vp]),
    "eyoc_model_fuse_tail": (_i, [_vp, _i]),
    "eyoc_model_workspace_bytes": (_sz, [_vp, _vp]),
    "eyoc_model_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "eyoc_model_num_layers": (_i, [_vp]),
    "eyoc_model_layer_work": (_i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "eyoc_model_set_timing": (_i, [_vp, _i]),
    "eyoc_model_layer_ms": (_i, [_vp, C.POINTER(C.c_float)]),
    "eyoc_model_timing_slot": (_i, [_vp, _i]),
    "eyoc_bn_workspace_bytes": (_sz, [_i, _i]),
    "eyoc_bn_train_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, C.c_float, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "eyoc_bn_train_forward_running": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, C.c_float, _i, _vp, _i, _vp, _vp, _vp, C.c_float, _vp, _sz, _vp]),
    "eyoc_bn_train_backward": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, C.c_float, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "eyoc_maps_gather_window": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "eyoc_maps_gather_window_internal": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "eyoc_knn_prefilter": (_i, [_vp, _i]),
    "eyoc_spconv_select_up_kernel": (_i, [_vp, _i]),
    "eyoc_spconv_upc_min_rows": (_i, [_vp, _i]),
    "eyoc_spconv_upc_tile_rows": (_i, [_i, _i]),
    "eyoc_spconv_upc_bytes": (_sz, [_i]),
    "eyoc_spconv_upc_build": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "eyoc_spconv_upc": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "eyoc_spconv_select_conv1_kernel": (_i, [_vp, _i]),
    "eyoc_knn1": (_i, [_vp, _vp, _vp, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i, _i, _vp, _vp, _vp]),
    "eyoc_pdist": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "eyoc_kabsch_batched": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "eyoc_irls_quad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "eyoc_ransac": (_i, [_vp, _vp, _vp, _vp, _i, C.POINTER(RansacParams), _vp, _vp]),
    "eyoc_maps_select_orders": (_i, [_vp, _i, _i]),
    "eyoc_ransac_select_pruning": (_i, [_vp, _i]),
    "eyoc_ransac_transform_store": (_i, [_vp, _i]),
    "eyoc_ransac_workspace_bytes": (_sz, [_vp, _i, _i, _i, _sz]),
    "eyoc_ransac_batched_ws": (_i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i,
                                    C.POINTER(RansacParams), _vp, _vp, _sz, _vp]),
    "eyoc_ransac_batched": (_i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i,
                                 C.POINTER(RansacParams), _vp, _vp]),
    "eyoc_sc2pcr_workspace_bytes": (_sz, [_i, C.POINTER(Sc2pcrParams)]),
    "eyoc_sc2pcr": (_i, [_vp, _vp, _vp, _i, C.POINTER(Sc2pcrParams), _vp, _vp, _vp, _sz, _vp]),
    "eyoc_sc2pcr_batched_workspace_bytes": (_sz, [_i, C.POINTER(Sc2pcrParams)]),
    "eyoc_sc2pcr_batched_workspace_bytes_n": (_sz, [_i, _i, C.POINTER(Sc2pcrParams)]),
    "eyoc_sc2pcr_set_shortlist_cap": (_i, [_vp, _i]),
    "eyoc_sc2pcr_set_dense_threshold": (_i, [_vp, _i]),
    "eyoc_sc2pcr_select_kernels": (_i, [_vp, _i]),
    "eyoc_sc2pcr_batched": (_i, [_vp, _vp, _vp, C.POINTER(C.c_int32), _i, C.POINTER(Sc2pcrParams), _vp, _vp, _i, _vp, _sz,
                                 _vp]),
}


ERR_INVALID, ERR_HIP, ERR_WORKSPACE, ERR_DUPLICATE, ERR_RANGE = -1, -2, -3, -4, -5     # eyoc_status


class EyocError(RuntimeError):
    """``code``: the ``eyoc_status`` the library returned (``None`` for errors raised on the Python side)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


_lib = None
_lock = threading.Lock()
_ctx = {}


def load():
    """Load the shared library (no GPU needed) and attach prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise EyocError(
                    f"{LIB_PATH} not found: build it with `python -m eyoc_amd.build` (hipcc, gfx950). "
                    "There is no CPU fallback for this path.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in PROTOTYPES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().eyoc_last_error().decode("utf-8", "replace")
        raise EyocError(f"{what or 'libeyoc_hip'} failed ({rc}): {msg}", rc)


def ctx(device_index: int | None = None):
    """The per-(process, device) ``eyoc_ctx*``."""
    import torch
    if device_index is not None:                       # the hot path: a dozen calls per forward, none of them needs torch
        h = _ctx.get(device_index)
        if h is not None:
            return h
    if not torch.cuda.is_available():
        raise EyocError("no GPU visible: the EYOC hot path runs on MI355X only (no CPU fallback)")
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index not in _ctx:
        lib = load()
        h = C.c_void_p()
        check(lib.eyoc_create(int(device_index), C.byref(h)), "eyoc_create")
        _ctx[device_index] = h
    return _ctx[device_index]


def knob(name: str, *values, device=None) -> int:
    """A kernel-selection / tiling switch of the device's ctx (``eyoc_spconv_select_*``, ``eyoc_maps_*``, ``eyoc_ransac_*`` ...; tests and
    diagnostics): ``knob("eyoc_spconv_select_kernel", 1)`` sets it and returns the previous value, an out-of-range value only queries."""
    import torch
    if isinstance(device, torch.device):
        device = device.index
    return getattr(load(), name)(ctx(device), *values)


def stream_ptr():
    """The current HIP stream of the current device as a ``void*``.  ``torch.cuda.current_stream().cuda_stream`` builds a Stream object
    per call (4 us, seven calls per registered pair); the raw getter torch's own compiled code uses returns the handle directly."""
    import torch
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
    except AttributeError:                             # (a torch without the raw getter)
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(device):
    """``torch.cuda.device(device)`` when the current device is another one, a no-op context otherwise (the context manager's own
    enter / exit is ~5 us - a tenth of one autograd layer's host time, paid twice per layer per direction)."""
    import torch
    idx = device.index if isinstance(device, torch.device) else device
    if idx is None or idx == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(idx)


_SCRATCH: dict = {}
_SCRATCH_MAX_STREAMS = 8


def scratch(nbytes: int, device):
    """A grow-only 256-byte aligned device buffer per (device, current stream) for kernels that need their workspace only until they
    finish (batch-norm partial sums, gradient partials): consecutive calls on a stream reuse it in stream order - no allocation, no
    slicing per call (``workspace`` = a fresh tensor every time: three torch ops, ~8 us of a ~45 us layer call).

    Keyed by the raw stream handle, which the runtime may hand to a NEW stream after the old one was destroyed: a buffer is therefore
    tied to its torch stream with ``record_stream`` (the caching allocator then orders its reuse on that stream whoever holds the
    handle), grows by a quarter over the request (not 2 x), and the cache keeps the ``_SCRATCH_MAX_STREAMS`` most recently used streams
    (``scratch_clear()`` drops everything, e.g. between training and serving phases)."""
    import torch
    idx = device.index if isinstance(device, torch.device) else int(device)
    stream = torch.cuda.current_stream(idx)
    key = (idx, stream.cuda_stream)
    t = _SCRATCH.pop(key, None)
    if t is None or t.numel() < nbytes:
        t = workspace(max(int(nbytes) + int(nbytes) // 4, 1 << 20), torch.device("cuda", idx))
        t.record_stream(stream)
    _SCRATCH[key] = t                                             # re-inserted: dict order = least recently used first
    while len(_SCRATCH) > _SCRATCH_MAX_STREAMS:
        _SCRATCH.pop(next(iter(_SCRATCH)))
    return t


def scratch_clear():
    """Drops every cached scratch buffer (they go back to torch's caching allocator once the kernels using them have finished)."""
    _SCRATCH.clear()


def workspace(nbytes: int, device):
    """256-byte aligned device workspace owned by torch's caching allocator."""
    import torch
    t = torch.empty(max(int(nbytes), 256) + 256, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % 256
    return t[off:off + max(int(nbytes), 256)]
