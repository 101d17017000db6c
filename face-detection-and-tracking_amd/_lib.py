"""ctypes binding of libfdt_hip.so (the C ABI declared in include/fdt.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C face-detection-and-tracking_amd/csrc -j8`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDT_LIB: another build of the same library (kernel experiments: tools/experiments/*_variants.sh); default = the in-tree one
LIB_PATH = os.environ.get("FDT_LIB") or os.path.join(_HERE, "csrc", "libfdt_hip.so")

FDT_OK = 0
FDT_ERR_ARG, FDT_ERR_HIP, FDT_ERR_STATE, FDT_ERR_NAME = -1, -2, -3, -4
ARCH_RES50, ARCH_TRY3, ARCH_FACEBOX, ARCH_TRY4, ARCH_TRY5, ARCH_TRY1, ARCH_TRY2 = 0, 1, 2, 3, 4, 5, 6
FRAME_U8_HWC_BGR, FRAME_F32_NCHW = 0, 1
F32, F64 = 0, 1


class FdtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfdt_hip error %d: %s" % (code, msg))
        self.code = code


_c_int_p = C.POINTER(C.c_int)
_c_f32_p = C.POINTER(C.c_float)
_c_f64_p = C.POINTER(C.c_double)
_c_i64_p = C.POINTER(C.c_longlong)
_vp = C.c_void_p

# name -> (restype, argtypes): every symbol include/fdt.h declares
SIGNATURES = {
    "fdt_last_error": (C.c_char_p, []),
    "fdt_version": (C.c_int, []),
    "fdt_device_count": (C.c_int, [_c_int_p]),
    "fdt_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "fdt_set_device": (C.c_int, [C.c_int]),
    "fdt_device_synchronize": (C.c_int, []),
    "fdt_thread_stream": (C.c_int, [C.POINTER(_vp)]),
    "fdt_stream_create_partition": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "fdt_stream_destroy": (C.c_int, [_vp]),
    "fdt_device_mem_info": (C.c_int, [_c_i64_p, _c_i64_p]),
    "fdt_priorbox": (C.c_int, [C.c_int] * 5 + [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fdt_decode": (C.c_int, [_vp, _vp, C.c_int, C.c_float, C.c_float, _vp]),
    "fdt_nms": (C.c_int, [_vp, _vp, C.c_int, C.c_float, C.c_int, _vp, _c_int_p]),
    "fdt_detect": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                             C.c_float, C.c_int, C.c_float, C.c_float, _vp, _vp]),
    "fdt_detect_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                 C.c_float, C.c_int, C.c_float, C.c_float, _vp, _vp, _vp,
                                 C.c_longlong, _vp]),
    "fdt_detect_workspace_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "fdt_pairwise_iou": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "fdt_pairwise_distance": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "fdt_conv2d": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "fdt_expand_dw": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "fdt_ir_block": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int,
                               _vp]),
    "fdt_facebox_anchors": (C.c_int, [_vp]),
    "fdt_facebox_decode": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_float, C.c_float, _vp, _vp, _c_int_p]),
    "fdt_tracker_create": (_vp, [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]),
    "fdt_tracker_destroy": (None, [_vp]),
    "fdt_tracker_reset": (C.c_int, [_vp]),
    "fdt_tracker_step": (C.c_int, [_vp, _vp, C.c_int]),
    "fdt_tracker_step_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "fdt_tracker_step_dev_multi": (C.c_int, [_vp, _vp, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                             _vp]),
    "fdt_tracker_finish": (C.c_int, [_vp]),
    "fdt_tracker_num_tracks": (C.c_int, [_vp, _c_int_p]),
    "fdt_tracker_track_info": (C.c_int, [_vp, C.c_int, _c_int_p, _c_f64_p, _c_int_p]),
    "fdt_tracker_track_boxes": (C.c_int, [_vp, C.c_int, _vp]),
    "fdt_tracker_stats": (C.c_int, [_vp, _c_i64_p, _c_i64_p]),
    "fdt_model_create": (_vp, [C.c_int, C.c_int]),
    "fdt_model_destroy": (None, [_vp]),
    "fdt_model_clone": (_vp, [_vp]),
    "fdt_model_enable_graph": (C.c_int, [_vp, C.c_int]),
    "fdt_comm_unique_id": (C.c_int, [C.c_char_p]),
    "fdt_comm_unique_id_local": (C.c_int, [C.c_char_p]),
    "fdt_comm_init_rank": (_vp, [C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "fdt_comm_init_all": (_vp, [C.c_int, _c_int_p]),
    "fdt_comm_world": (C.c_int, [_vp, _c_int_p, _c_int_p]),
    "fdt_comm_group_begin": (C.c_int, []),
    "fdt_comm_group_end": (C.c_int, []),
    "fdt_allgather_dets": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_longlong, _vp]),
    "fdt_comm_destroy": (None, [_vp]),
    "fdt_model_set_tensor": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int, _c_i64_p]),
    "fdt_model_missing": (C.c_int, [_vp, _c_int_p]),
    "fdt_model_missing_name": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int]),
    "fdt_model_finalize": (C.c_int, [_vp]),
    "fdt_model_set_priorbox": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _c_int_p, _c_int_p]),
    "fdt_model_set_detect": (C.c_int, [_vp, C.c_int, C.c_float, C.c_float, C.c_int]),
    "fdt_model_forward": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "fdt_model_forward_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "fdt_model_forward_resized": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                            _vp]),
    "fdt_model_forward_async": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _c_int_p]),
    "fdt_model_async_record": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _vp]),
    "fdt_model_wait": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "fdt_model_release": (C.c_int, [_vp, C.c_int, _vp]),
    "fdt_model_forward_raw": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "fdt_model_detect_facebox": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                           _vp, _vp, _vp]),
    "fdt_model_detect_facebox_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                               C.c_float, _vp, _vp]),
    "fdt_model_detect_facebox_resized": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                                   _vp, _vp, _vp, _vp]),
    "fdt_model_traffic": (C.c_int, [_vp, _c_f64_p, _c_f64_p, C.c_int, _vp, _c_int_p]),
    "fdt_model_num_priors": (C.c_int, [_vp, _c_int_p]),
    "fdt_model_get_tensor": (C.c_int, [_vp, C.c_char_p, _vp, C.c_longlong, _c_i64_p]),
    "fdt_model_autotune": (C.c_int, [_vp, C.c_int]),
    "fdt_model_export_plan": (C.c_int, [_vp, C.c_char_p, C.c_int, _c_int_p]),
    "fdt_model_import_plan": (C.c_int, [_vp, C.c_char_p]),
    "fdt_model_profile_enable": (C.c_int, [_vp, C.c_int]),
    "fdt_model_profile_read": (C.c_int, [_vp, C.c_int, C.c_char_p, _vp, _vp, _c_int_p]),
    "fdt_model_profile_segment": (C.c_int, [_vp, C.c_int, C.c_int]),
    "fdt_model_profile_segment_ms": (C.c_int, [_vp, _vp]),
    "fdt_model_flops": (C.c_int, [_vp, _c_f64_p]),
    "fdt_model_get_detect": (C.c_int, [_vp, _c_int_p, _vp, _vp, _c_int_p]),
    "fdt_model_fuse_ingest": (C.c_int, [_vp, C.c_int]),
    "fdt_pipeline_create": (_vp, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, _vp, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_float, C.c_double, C.c_double, C.c_int, C.c_int]),
    "fdt_pipeline_destroy": (None, [_vp]),
    "fdt_pipeline_prime": (C.c_int, [_vp, _vp]),
    "fdt_pipeline_step": (C.c_int, [_vp, C.c_longlong, _vp]),
    "fdt_pipeline_step_host": (C.c_int, [_vp, C.c_longlong, _vp, C.c_int]),
    "fdt_pipeline_step_frame": (C.c_int, [_vp, C.c_longlong, _vp]),
    "fdt_pipeline_flush": (C.c_int, [_vp]),
    "fdt_pipeline_sync": (C.c_int, [_vp]),
    "fdt_pipeline_tracker": (_vp, [_vp]),
    "fdt_pipeline_slot": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "fdt_pipeline_mark": (C.c_int, [_vp, C.c_int]),
    "fdt_pipeline_elapsed_ms": (C.c_int, [_vp, _vp]),
    "fdt_pipeline_stamps_enable": (C.c_int, [_vp, C.c_int]),
    "fdt_pipeline_stamps_read": (C.c_int, [_vp, _vp, C.c_int, _c_int_p]),
    "fdt_dev_malloc": (C.c_int, [C.POINTER(_vp), C.c_longlong]),
    "fdt_dev_free": (C.c_int, [_vp]),
    "fdt_dev_upload": (C.c_int, [_vp, _vp, C.c_longlong]),
    "fdt_dev_download": (C.c_int, [_vp, _vp, C.c_longlong]),
}

_lib = None


def lib():
    """The loaded CDLL (loads on first use; raises if the library was not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FdtError(FDT_ERR_STATE,
                           "%s not found -- build it first (make -C %s); there is no CPU fallback"
                           % (LIB_PATH, os.path.dirname(LIB_PATH)))
        # Load order: the PyTorch-ROCm wheel ships its own libamdhip64 / librccl and refers to them by un-versioned
        # names, while this library links the versioned sonames of /opt/rocm.  If this library came first, a later
        # `import torch` would bring a SECOND HIP runtime into the process (its torch.cuda then finds no GPU and stream /
        # device pointers could not be shared); with torch first, the loader satisfies our sonames from the copies torch
        # loaded.  Every host module of this package that hands out tensors imports torch anyway.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)        # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != FDT_OK:
        raise FdtError(rc, (lib().fdt_last_error() or b"").decode(errors="replace"))
    return rc


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_vp)


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def device_count():
    n = C.c_int(0)
    check(lib().fdt_device_count(C.byref(n)))
    return n.value


def device_name(dev=0):
    buf = C.create_string_buffer(256)
    check(lib().fdt_device_name(dev, buf, 256))
    return buf.value.decode()
