"""Drop-in for `mmcv.ops.roi_align` on MI355X.

Mirrors /root/reference/mmcv-1.4.7/mmcv/ops/roi_align.py: `RoIAlignFunction` (same 7
forward arguments :64-72, same 7-tuple backward :128), `roi_align = RoIAlignFunction.apply`
(:131) and the `RoIAlign` module (:134-224, including the deprecated `out_size` /
`sample_num` aliases :171-176 and `output_size -> _pair` :186 that MlvlRoIExtractor reads
back at gpt4roi/models/layers.py:270,286).  The native call goes to the hand-written
gfx950 kernels through the C ABI of include/g4r_roi_align.h instead of `mmcv._ext`.
"""
import ctypes
import warnings

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.float16: "f16"}


def _suffix(t):
    try:
        return _SUFFIX[t.dtype]
    except KeyError:
        # the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF (roi_align_cuda.cu:17)
        raise RuntimeError(f'"roi_align" not implemented for {t.dtype}') from None


def roi_align_forward(input, rois, output, argmax_y, argmax_x, aligned_height, aligned_width,
                      spatial_scale, sampling_ratio, pool_mode, aligned):
    """Same signature as `mmcv._ext.roi_align_forward` (pybind.cpp:611-615)."""
    _lib.require_gpu(input, rois, output)
    if not (input.device == rois.device == output.device):
        raise RuntimeError("roi_align_forward: all tensors must be on the same device")
    fn = getattr(_lib.lib(), f"g4r_roi_align_forward_{_suffix(input)}")
    rc = fn(_lib.ptr(input), _lib.ptr(rois), _lib.ptr(output),
            _lib.ptr(argmax_y if argmax_y.numel() else None),
            _lib.ptr(argmax_x if argmax_x.numel() else None),
            input.size(0), input.size(1), input.size(2), input.size(3), rois.size(0),
            int(aligned_height), int(aligned_width), ctypes.c_float(spatial_scale),
            int(sampling_ratio), int(pool_mode), int(bool(aligned)), _lib.stream_of(input))
    _lib.check(rc, "roi_align_forward")


def roi_align_backward(grad_output, rois, argmax_y, argmax_x, grad_input, aligned_height,
                       aligned_width, spatial_scale, sampling_ratio, pool_mode, aligned):
    """Same signature as `mmcv._ext.roi_align_backward` (pybind.cpp:616-620)."""
    _lib.require_gpu(grad_output, rois, grad_input)
    fn = getattr(_lib.lib(), f"g4r_roi_align_backward_{_suffix(grad_output)}")
    rc = fn(_lib.ptr(grad_output), _lib.ptr(rois),
            _lib.ptr(argmax_y if argmax_y.numel() else None),
            _lib.ptr(argmax_x if argmax_x.numel() else None), _lib.ptr(grad_input),
            grad_input.size(0), grad_input.size(1), grad_input.size(2), grad_input.size(3),
            rois.size(0), int(aligned_height), int(aligned_width), ctypes.c_float(spatial_scale),
            int(sampling_ratio), int(pool_mode), int(bool(aligned)), _lib.stream_of(grad_output))
    _lib.check(rc, "roi_align_backward")


class RoIAlignFunction(Function):

    @staticmethod
    def forward(ctx, input, rois, output_size, spatial_scale=1.0, sampling_ratio=0,
                pool_mode='avg', aligned=True):
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        assert pool_mode in ('max', 'avg')
        ctx.pool_mode = 0 if pool_mode == 'max' else 1
        ctx.aligned = aligned
        ctx.input_shape = input.size()

        assert rois.size(1) == 5, 'RoI must be (idx, x1, y1, x2, y2)!'

        output_shape = (rois.size(0), input.size(1), ctx.output_size[0], ctx.output_size[1])
        output = input.new_zeros(output_shape)
        if ctx.pool_mode == 0:
            argmax_y = input.new_zeros(output_shape)
            argmax_x = input.new_zeros(output_shape)
        else:
            argmax_y = input.new_zeros(0)
            argmax_x = input.new_zeros(0)

        # the reference assumes (does not check) contiguity; the kernels index dense NCHW
        roi_align_forward(input.contiguous(), rois.contiguous().to(input.dtype), output,
                          argmax_y, argmax_x, ctx.output_size[0], ctx.output_size[1],
                          ctx.spatial_scale, ctx.sampling_ratio, ctx.pool_mode, ctx.aligned)

        ctx.save_for_backward(rois, argmax_y, argmax_x)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rois, argmax_y, argmax_x = ctx.saved_tensors
        grad_input = grad_output.new_zeros(ctx.input_shape)
        grad_output = grad_output.contiguous()
        roi_align_backward(grad_output, rois.contiguous().to(grad_output.dtype), argmax_y,
                           argmax_x, grad_input, ctx.output_size[0], ctx.output_size[1],
                           ctx.spatial_scale, ctx.sampling_ratio, ctx.pool_mode, ctx.aligned)
        return grad_input, None, None, None, None, None, None


roi_align = RoIAlignFunction.apply


class RoIAlign(nn.Module):
    """RoI align pooling layer (API of mmcv.ops.RoIAlign, roi_align.py:134-224).

    Args:
        output_size (tuple): h, w
        spatial_scale (float): scale the input boxes by this number
        sampling_ratio (int): samples per bin side; 0 = adaptive
        pool_mode (str, 'avg' or 'max'): pooling mode in each bin.
        aligned (bool): pixel-centre convention (True) or the legacy one.
        use_torchvision (bool): accepted for signature compatibility; torchvision is not
            part of this stack, so True raises at call time.
    """

    def __init__(self, output_size=None, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg',
                 aligned=True, use_torchvision=False, **deprecated):
        super().__init__()
        # deprecated aliases handled like mmcv's deprecated_api_warning (roi_align.py:171-176)
        for old, new in (('out_size', 'output_size'), ('sample_num', 'sampling_ratio')):
            if old in deprecated:
                warnings.warn(f'"{old}" is deprecated in `RoIAlign`, please use "{new}" instead')
                if new == 'output_size':
                    output_size = deprecated.pop(old)
                else:
                    sampling_ratio = deprecated.pop(old)
        if deprecated:
            raise TypeError(f"RoIAlign got unexpected arguments {sorted(deprecated)}")
        if output_size is None:
            raise TypeError("RoIAlign missing required argument 'output_size'")
        self.output_size = _pair(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.pool_mode = pool_mode
        self.aligned = aligned
        self.use_torchvision = use_torchvision

    def forward(self, input, rois):
        """input: NCHW images; rois: Bx5 (batch index, x1, y1, x2, y2)."""
        if self.use_torchvision:
            raise RuntimeError("use_torchvision=True is not supported by the MI355X build")
        return roi_align(input, rois, self.output_size, self.spatial_scale,
                         self.sampling_ratio, self.pool_mode, self.aligned)

    def __repr__(self):
        s = self.__class__.__name__
        s += f'(output_size={self.output_size}, '
        s += f'spatial_scale={self.spatial_scale}, '
        s += f'sampling_ratio={self.sampling_ratio}, '
        s += f'pool_mode={self.pool_mode}, '
        s += f'aligned={self.aligned}, '
        s += f'use_torchvision={self.use_torchvision})'
        return s
