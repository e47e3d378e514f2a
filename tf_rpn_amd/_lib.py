"""ctypes binding of ``librpn_hip.so`` (C ABI declared in ``include/rpn_hip.h``).

There is no CPU fallback: if the HIP library is missing this module raises at first use,
and every compute entry point returns ``RPN_ERR_NO_DEVICE`` without a GPU, which is
re-raised here as ``RuntimeError``.  torch is used only for device memory and streams.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RPN_HIP_LIB: another build of the same library (kernel A/B runs); still no fallback of any kind.
LIB_PATH = os.environ.get("RPN_HIP_LIB") or os.path.join(_HERE, "csrc", "librpn_hip.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
vp = ctypes.c_void_p

RPN_OK, RPN_ERR_INVALID, RPN_ERR_NO_DEVICE, RPN_ERR_WORKSPACE, RPN_ERR_UNSUPPORTED = 0, -1, -2, -3, -4
PRECISIONS = {"f32": 0, "fp32": 0, "float32": 0, "bf16x3": 1, "f16x3": 2, "fp16x3": 2, "f32w": 3}
BACKBONES = {"vgg16": 0, "mobilenet_v2": 1}
STATUS_F16_RANGE = 1
ACTS = {None: 0, "linear": 0, "relu": 1, "sigmoid": 2, "relu6": 3}

# name -> (restype, argtypes); mirrors include/rpn_hip.h one to one
_SIGNATURES = {
    "rpn_abi_version": (ctypes.c_int, []),
    "rpn_last_error": (ctypes.c_char_p, []),
    "rpn_device_count": (ctypes.c_int, []),
    "rpn_stream_spin": (ctypes.c_int, [vp, ctypes.c_int]),
    "rpn_generate_anchors": (ctypes.c_int, [ctypes.c_double, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p,
                                            ctypes.c_int, vp, vp]),
    "rpn_decode": (ctypes.c_int, [vp, ctypes.c_int, vp, c_float_p, ctypes.c_int, ctypes.c_int, vp, vp]),
    "rpn_encode": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "rpn_scale_boxes": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_float, ctypes.c_float, ctypes.c_int, vp, vp]),
    "rpn_targets_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "rpn_rpn_targets": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       c_float_p, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "rpn_preprocess_image": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "rpn_iou_map": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "rpn_nms_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "rpn_combined_nms": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                        vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "rpn_decode_nms": (ctypes.c_int, [vp, vp, c_float_p, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_int, vp, vp, vp, vp, vp,
                                      ctypes.c_size_t, vp]),
    "rpn_model_create": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(vp)]),
    "rpn_model_destroy": (None, [vp]),
    "rpn_model_feature_map_shape": (ctypes.c_int, [vp]),
    "rpn_model_num_layers": (ctypes.c_int, [vp]),
    "rpn_model_layer_info": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, c_int_p, c_int_p]),
    "rpn_model_layer_bn_name": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    "rpn_model_memory_bytes": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "rpn_model_keep_activations": (ctypes.c_int, [vp, ctypes.c_int]),
    "rpn_model_set_layer": (ctypes.c_int, [vp, ctypes.c_char_p] + [c_float_p] * 6),
    "rpn_model_forward": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp, vp]),
    "rpn_model_status": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_uint), ctypes.c_int, vp]),
    "rpn_model_get_activation": (ctypes.c_int, [vp, ctypes.c_char_p, vp, ctypes.c_size_t, c_int_p, vp]),
    "rpn_model_flops_per_image": (ctypes.c_double, [vp]),
    "rpn_model_set_profiling": (ctypes.c_int, [vp, ctypes.c_int]),
    "rpn_model_set_profiling_mask": (ctypes.c_int, [vp, vp, ctypes.c_int]),
    "rpn_model_set_profiling_rotate": (ctypes.c_int, [vp, ctypes.c_int]),
    "rpn_model_num_ops": (ctypes.c_int, [vp]),
    "rpn_model_op_arith": (ctypes.c_int, [vp, ctypes.c_int]),
    "rpn_model_op_info": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "rpn_model_get_profile": (ctypes.c_int, [vp, c_float_p, ctypes.c_int, c_int_p]),
    "rpn_conv2d": (ctypes.c_int, [vp] + [ctypes.c_int] * 4 + [vp, vp] + [ctypes.c_int] * 10 + [vp, vp]),
    "rpn_maxpool2x2": (ctypes.c_int, [vp] + [ctypes.c_int] * 4 + [vp, vp]),
    "rpn_dwconv3x3": (ctypes.c_int, [vp] + [ctypes.c_int] * 4 + [vp, vp] + [ctypes.c_int] * 6 + [vp, vp]),
}

_lib = None


def exported_symbols():
    """Names every entry point declared in include/rpn_hip.h (used by the CPU test-suite)."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "librpn_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C tf_rpn_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so does not export it
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(status, what):
    if status != RPN_OK:
        msg = lib().rpn_last_error().decode("utf-8", "replace")
        exc = ValueError if status == RPN_ERR_INVALID else RuntimeError
        raise exc("%s failed (%d): %s" % (what, status, msg))


def require_gpu():
    if not torch.cuda.is_available() or lib().rpn_device_count() < 1:
        raise RuntimeError("tf_rpn_amd needs a HIP device (MI355X / gfx950); there is no CPU fallback")


def stream_ptr():
    return vp(torch.cuda.current_stream().cuda_stream)


def to_device(x, dtype=torch.float32):
    """numpy / torch (any device) -> contiguous torch tensor on the current HIP device.
    Returns (tensor, was_numpy)."""
    was_numpy = not isinstance(x, torch.Tensor)
    if was_numpy:
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    require_gpu()
    x = x.to(device="cuda", dtype=dtype).contiguous()
    return x, was_numpy


def from_device(t, as_numpy):
    return t.cpu().numpy() if as_numpy else t


def ptr(t):
    return vp(t.data_ptr()) if t is not None else vp(0)


def host_floats(values):
    arr = np.ascontiguousarray(np.asarray(values, dtype=np.float32))
    return arr, arr.ctypes.data_as(c_float_p)
