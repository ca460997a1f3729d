"""ctypes binding of libet_hip.so (C ABI: include/et_hip.h).

The HIP library is the product: if it is missing (not built) or a tensor is not on a GPU, every
entry point raises -- there is no CPU fallback.  ``tests/`` may substitute the SIMT-emulator build
of the *same kernel sources* (tests/simt_emu) through ``_use_library_for_tests`` to exercise the
kernels and the host glue on CPU before a GPU run; nothing in the package does that by itself.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ET_HIP_LIB") or os.path.join(_HERE, "libet_hip.so")   # ET_HIP_LIB: experiment builds

ET_F32, ET_BF16, ET_F16 = 0, 1, 2

P = c_void_p
# name -> (restype, argtypes); kept in the order of include/et_hip.h
SIGNATURES = {
    "et_build_arch": (c_char_p, []),
    "et_abi_version": (c_int, []),
    "et_nms_ssod_workspace_bytes": (c_int, [c_int, c_int, ctypes.POINTER(c_size_t)]),
    "et_nms_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "et_nms": (c_int, [P, c_int, c_int, c_int, c_float, c_float, c_int, c_int, ctypes.c_uint64, ctypes.c_uint64, c_int,
                       c_float, c_int, P, P, P, P, P, c_size_t, P]),
    "et_nms_ssod": (c_int, [P, c_int, c_int, c_int, c_float, c_float, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "et_detect_decode": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
                                 P, c_float, P, c_int64, c_int64, P]),
    "et_ema_update": (c_int, [P, P, c_int64, c_float, c_float, P]),
    "et_adamw": (c_int, [P, P, P, P, P, c_int, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, P, P]),
    "et_adamw_tick": (c_int, [P, c_float, c_float, P, P]),
    "et_adamw_dev": (c_int, [P, P, P, P, P, c_int, c_int64, c_float, c_float, c_float, c_float, c_float, P, c_float, P, P]),
    "et_sgd_nesterov": (c_int, [P, P, P, P, c_int, c_int64, c_float, c_float, c_float, c_int, c_float, P, P]),
    "et_cast_f32_to_lp": (c_int, [P, P, c_int, c_int64, P]),
    "et_scaler_check": (c_int, [P, c_int64, P, P]),
    "et_scaler_update": (c_int, [P, c_float, c_float, c_int, P]),
    "et_ema_update_dev": (c_int, [P, P, c_int64, P, P]),
    "et_sgd_nesterov_dev": (c_int, [P, P, P, P, c_int, c_int64, P, c_int, P, P]),
    "et_conv2d_stats_rows": (c_int, [c_int, c_int, c_int]),
    "et_conv2d_stats_rows_for": (c_int, [c_int] * 12),
    "et_conv2d_stats_adds_for": (c_int, [c_int] * 12),
    "et_conv2d_fwd": (c_int, [P, P, P, c_int] + [c_int] * 11 + [P, P, c_int, P, c_int, P, c_int, P, P]),
    "et_conv2d_dgrad": (c_int, [P, P, P, c_int] + [c_int] * 11 + [c_int, P, c_int, P, P]),
    "et_conv2d_dgrad_bn": (c_int, [P, P, P, c_int] + [c_int] * 10 + [P, c_int, P, c_int, P, P, c_int, P, c_int, P, P]),
    "et_conv2d_wgrad": (c_int, [P, P, P, c_int] + [c_int] * 11 + [P, P]),
    "et_conv2d_wgrad_grouped": (c_int, [P, c_int, c_int] + [c_int] * 9 + [P, P]),
    "et_weight_transpose_all": (c_int, [P, P, c_int, P, c_int, ctypes.c_longlong, P]),
    "et_weight_transpose": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "et_conv2d_kernel_name": (c_int, [c_int] * 13 + [c_char_p, c_int]),
    "et_env_knobs": (c_int, [c_char_p, c_int]),
    "et_colsum": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "et_bn_reduce_rows": (c_int, [c_int, c_int, c_int]),
    "et_bn_finalize": (c_int, [P, c_int, c_int, c_double, P, P, c_float, c_float, P, P, P, P, P, P, P, P]),
    "et_bn_eval_affine": (c_int, [c_int, P, P, P, P, c_float, P, P, P]),
    "et_bn_act_fwd": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P, c_int, P]),
    "et_bn_act_fwd_sharded": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, c_int, c_double, P, P, c_float, c_float,
                                      P, P, P, P, P, P, c_int, P]),
    "et_bn_act_bwd_sharded": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P, c_int, c_int, P]),
    "et_bn_act_bwd": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P, P,
                              c_size_t, P]),
    "et_bn_act_bwd_from_partials": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P,
                                            P, c_int, P, P]),
    "et_act_bwd": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "et_pack_input": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "et_pack_input_u8": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    "et_maxpool5_fwd": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "et_maxpool5_bwd": (c_int, [P, c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "et_upsample2x_fwd": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "et_upsample2x_bwd": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "et_strong_view_u8": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, c_int, P]),
    "et_mosaic4_u8": (c_int, [P, P, c_int, c_int, c_int, P]),
    "et_pseudo_label_transform": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "et_score_log_append": (c_int, [P, P, c_int, c_int, P, P, P, c_int64, P]),
    "et_yolo_loss": (c_int, [P, P]),
    "et_ota_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "et_ota_assign": (c_int, [P, P, c_float, c_int, P, P, P]),
    "et_select_targets": (c_int, [P, P, c_int, P, P, c_int, c_int, P, P]),
    "et_scale_cast": (c_int, [P, P, c_int, c_int64, c_float, P, P]),
    "et_domain_focal": (c_int, [P, c_int, c_int, c_int64, c_int, c_float, P, c_int, P, P]),
    "et_scale_inplace": (c_int, [P, c_int, c_int64, c_float, P, P]),
    "et_v8_decode": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, P, c_int64, c_int64, P]),
    "et_tal_loss": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, P, P, P, P, P]),
    "et_tal_assign_workspace_bytes": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "et_tal_assign": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "et_tal_targets_pad": (c_int, [P, c_int, c_int, c_int, c_float, c_float, P, P, P]),
    "et_tal_pseudo_split": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, P, P, P, P, P, P, P, P, P]),
    "et_tal_assigned_gt": (c_int, [P, c_int, c_int, c_int, P, P]),
    "et_tal_merge_pseudo": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P]),
}



class WgradItem(ctypes.Structure):
    """et_wgrad_item (include/et_hip.h)"""
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("ldx", ctypes.c_int),
                ("ldy", ctypes.c_int)]


class LossLevel(ctypes.Structure):
    _fields_ = [("p", P), ("dp", P), ("tobj_ws", P), ("sb", c_int64), ("sa", c_int64), ("sy", c_int64),
                ("sx", c_int64), ("ny", c_int), ("nx", c_int), ("anchors", c_float * 6), ("balance", c_float)]


class LossDesc(ctypes.Structure):
    _fields_ = [("dtype", c_int), ("B", c_int), ("na", c_int), ("nc", c_int), ("NT", c_int), ("nl", c_int),
                ("anchor_t", c_float), ("gr", c_float), ("cp", c_float), ("cn", c_float), ("cls_pw", c_float),
                ("obj_pw", c_float), ("box_w", c_float), ("obj_w", c_float), ("cls_w", c_float),
                ("pass_mask", c_int), ("ignore_obj", c_int), ("targets", P), ("acc_ws", P), ("out", P),
                ("ota_match", P), ("obj_channel", c_int), ("balance_dev", P), ("autobalance_ssi", c_int), ("fl_gamma", c_float), ("level", LossLevel * 4)]


_dll = None
_emulated = False
# bench.py's per-family time budget (ops.FamilyTimer): when set, every launching entry point is bracketed by two HIP events on
# the launching stream.  None (always, outside that one instrumented step) = the library object itself, no wrapper.
CALL_TIMER = None
_NO_LAUNCH = frozenset(n for n in (
    "et_build_arch", "et_abi_version", "et_nms_ssod_workspace_bytes", "et_nms_workspace_bytes", "et_conv2d_stats_rows", "et_conv2d_stats_rows_for",
    "et_conv2d_kernel_name", "et_env_knobs", "et_bn_reduce_rows", "et_ota_workspace_bytes", "et_tal_assign_workspace_bytes"))


class _TimedDll:
    """the loaded library with every launching entry point wrapped: timer.begin(name) / timer.end(token) around the call"""

    def __init__(self, dll, timer):
        self._dll, self._timer, self._cache = dll, timer, {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            raw = getattr(self._dll, name)
            if name in _NO_LAUNCH or not name.startswith("et_"):
                fn = raw
            else:
                timer = self._timer

                def fn(*a, _raw=raw, _name=name):
                    tok = timer.begin(_name)
                    rc = _raw(*a)
                    timer.end(tok)
                    return rc
            self._cache[name] = fn
        return fn


class EtHipError(RuntimeError):
    pass


def _header_abi_version():
    """ET_ABI_VERSION of include/et_hip.h -- the contract SIGNATURES above was written against"""
    import re
    with open(os.path.join(os.path.dirname(_HERE), "include", "et_hip.h")) as f:
        m = re.search(r"^#define\s+ET_ABI_VERSION\s+(\d+)", f.read(), re.M)
    if m is None:
        raise EtHipError("include/et_hip.h does not define ET_ABI_VERSION")
    return int(m.group(1))


ABI_VERSION = _header_abi_version()


def _declare(dll):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(dll, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = dll.et_abi_version()
    if got != ABI_VERSION:
        # a stale build: argument lists differ (an added int would be read as the stream, kernels would run unordered on the null stream)
        raise EtHipError(f"{getattr(dll, '_name', 'libet_hip.so')} has ABI version {got}, this binding needs {ABI_VERSION}: "
                         f"rebuild it (python -m efficientteacher_amd.csrc.build)")
    return dll


def load(path=None):
    """Load (once) and return the HIP library.  Raises if it has not been built."""
    global _dll
    if _dll is None:
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise EtHipError(
                f"{path} not found: build the gfx950 kernels first "
                f"(python -m efficientteacher_amd.csrc.build).  There is no CPU fallback.")
        _dll = _declare(ctypes.CDLL(path))
    if CALL_TIMER is not None:
        w = getattr(CALL_TIMER, "_wrapped", None)
        if w is None or w._dll is not _dll:
            w = CALL_TIMER._wrapped = _TimedDll(_dll, CALL_TIMER)
        return w
    return _dll


def _use_library_for_tests(path, emulated):
    """TEST HOOK: point the binding at another build of the same C ABI (the SIMT emulator)."""
    global _dll, _emulated
    _dll = _declare(ctypes.CDLL(path)) if path else None
    _emulated = bool(emulated) if path else False


def is_emulated():
    return _emulated


def check(rc, what):
    if rc != 0:
        raise EtHipError(f"{what} failed with code {rc}")


def ptr(t):
    """Device pointer of a tensor (or NULL)."""
    if t is None:
        return None
    if not t.is_cuda and not _emulated:
        raise EtHipError("efficientteacher_amd kernels need CUDA/HIP tensors (got a CPU tensor); "
                         "there is no CPU fallback")
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream(t=None):
    """the current HIP stream handle of t's device.  Through torch's raw-stream accessor where it exists: constructing a
    torch.cuda.Stream object per launch (current_stream().cuda_stream) cost 4.7 us x 340 launches = 1.6 ms of host time per SSOD step
    (profiles/r06_adapter_host_cost.txt)"""
    if _emulated:
        return None
    if _RAW_STREAM is not None:
        idx = t.device.index if t is not None else None
        return _RAW_STREAM(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(t.device if t is not None else None).cuda_stream
