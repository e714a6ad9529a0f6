"""oracle/roi_align.py -- TEST INFRASTRUCTURE.

ctypes front for oracle/roi_align_oracle.c (the plain-C restatement of
mmcv-1.4.7/mmcv/ops/csrc/pytorch/cpu/roi_align.cpp:23-382) and a loader for
oracle/_ref (the reference's own CPU sources compiled unmodified by build_ref.py).
"""
import ctypes
import importlib.util
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleError(RuntimeError):
    pass


_ERR = {1: "ROIs in ROIAlign cannot have negative size", 2: "allocation failed",
        3: "roi batch index out of range"}


def build():
    """Compile the C restatement (gcc, <1 s)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle_roi_align.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle_roi_align.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _sfx(dtype):
    if dtype == np.float32:
        return "f32", ctypes.c_float
    if dtype == np.float64:
        return "f64", ctypes.c_double
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def forward(x, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg",
            aligned=True):
    """x [B,C,H,W], rois [n,5] -> (out [n,C,ph,pw], argmax_y, argmax_x)."""
    x = np.ascontiguousarray(x)
    sfx, cty = _sfx(x.dtype)
    rois = np.ascontiguousarray(rois, dtype=x.dtype)
    assert rois.ndim == 2 and rois.shape[1] == 5, "RoI must be (idx, x1, y1, x2, y2)!"
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    B, C, H, W = x.shape
    n = rois.shape[0]
    mode = {"max": 0, "avg": 1}[pool_mode]
    out = np.zeros((n, C, ph, pw), x.dtype)
    amy = np.zeros((n, C, ph, pw) if mode == 0 else (0,), x.dtype)
    amx = np.zeros_like(amy)
    rc = getattr(_lib(), f"oracle_roi_align_forward_{sfx}")(
        _ptr(x), _ptr(rois), _ptr(out), _ptr(amy), _ptr(amx), B, n, C, H, W, ph, pw,
        cty(spatial_scale), int(sampling_ratio), mode, int(bool(aligned)))
    if rc:
        raise OracleError(_ERR.get(rc, f"oracle error {rc}"))
    return out, amy, amx


def backward(grad_out, rois, input_shape, output_size, spatial_scale=1.0, sampling_ratio=0,
             pool_mode="avg", aligned=True, argmax_y=None, argmax_x=None):
    g = np.ascontiguousarray(grad_out)
    sfx, cty = _sfx(g.dtype)
    rois = np.ascontiguousarray(rois, dtype=g.dtype)
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    B, C, H, W = input_shape
    n = rois.shape[0]
    mode = {"max": 0, "avg": 1}[pool_mode]
    gin = np.zeros(input_shape, g.dtype)
    amy = np.zeros((0,), g.dtype) if argmax_y is None else np.ascontiguousarray(argmax_y)
    amx = np.zeros((0,), g.dtype) if argmax_x is None else np.ascontiguousarray(argmax_x)
    rc = getattr(_lib(), f"oracle_roi_align_backward_{sfx}")(
        _ptr(g), _ptr(rois), _ptr(amy), _ptr(amx), _ptr(gin), B, n, C, H, W, ph, pw,
        cty(spatial_scale), int(sampling_ratio), mode, int(bool(aligned)))
    if rc:
        raise OracleError(_ERR.get(rc, f"oracle error {rc}"))
    return gin


def load_ref():
    """Import oracle/_ref/mmcv_roi_align_ref.so (prebuilt); None when it does not exist."""
    path = os.path.join(HERE, "_ref", "mmcv_roi_align_ref.so")
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401  (the extension links libtorch)
    spec = importlib.util.spec_from_file_location("mmcv_roi_align_ref", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_forward(x, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg",
                aligned=True):
    """Same contract as forward() but through the reference's own compiled CPU code,
    allocating exactly as mmcv/ops/roi_align.py:83-104 does."""
    import torch
    ref = load_ref()
    assert ref is not None, "oracle/_ref not built (python oracle/build_ref.py)"
    xt = torch.as_tensor(np.ascontiguousarray(x))
    rt = torch.as_tensor(np.ascontiguousarray(rois, dtype=x.dtype))
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    mode = {"max": 0, "avg": 1}[pool_mode]
    shape = (rt.size(0), xt.size(1), ph, pw)
    out = xt.new_zeros(shape)
    amy = xt.new_zeros(shape) if mode == 0 else xt.new_zeros(0)
    amx = xt.new_zeros(shape) if mode == 0 else xt.new_zeros(0)
    ref.roi_align_forward(xt, rt, out, amy, amx, aligned_height=ph, aligned_width=pw,
                          spatial_scale=float(spatial_scale), sampling_ratio=int(sampling_ratio),
                          pool_mode=mode, aligned=bool(aligned))
    return out.numpy(), amy.numpy(), amx.numpy()


def ref_backward(grad_out, rois, input_shape, output_size, spatial_scale=1.0, sampling_ratio=0,
                 pool_mode="avg", aligned=True, argmax_y=None, argmax_x=None):
    import torch
    ref = load_ref()
    assert ref is not None, "oracle/_ref not built (python oracle/build_ref.py)"
    g = torch.as_tensor(np.ascontiguousarray(grad_out))
    rt = torch.as_tensor(np.ascontiguousarray(rois, dtype=grad_out.dtype))
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    mode = {"max": 0, "avg": 1}[pool_mode]
    gin = g.new_zeros(tuple(input_shape))
    amy = g.new_zeros(0) if argmax_y is None else torch.as_tensor(np.ascontiguousarray(argmax_y))
    amx = g.new_zeros(0) if argmax_x is None else torch.as_tensor(np.ascontiguousarray(argmax_x))
    ref.roi_align_backward(g, rt, amy, amx, gin, aligned_height=ph, aligned_width=pw,
                           spatial_scale=float(spatial_scale), sampling_ratio=int(sampling_ratio),
                           pool_mode=mode, aligned=bool(aligned))
    return gin.numpy()
