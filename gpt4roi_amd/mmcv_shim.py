"""Optional name-level drop-in for the reference's op lookup (seam B1 of SURVEY.md 8b).

The reference resolves the op BY NAME: `getattr(mmcv.ops, 'RoIAlign')`
(/root/reference/mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:54-60) and
`ext_loader.load_ext('_ext', ['roi_align_forward', 'roi_align_backward'])`
(mmcv-1.4.7/mmcv/ops/roi_align.py:10-11).  `install()` registers lightweight `mmcv.ops` and
`mmcv._ext` modules exposing exactly those names, backed by the gfx950 kernels, for processes
where the real mmcv (and its ~45 other CUDA ops) is not installed.  It refuses to shadow a real
mmcv that is already imported.
"""
import sys
import types

from . import roi_align as _ra


def install(force=False):
    if "mmcv" in sys.modules and not getattr(sys.modules["mmcv"], "__g4r_shim__", False) and not force:
        raise RuntimeError("a real mmcv is already imported; patch mmcv.ops.roi_align.ext_module instead "
                           "(see INTEGRATION.md)")
    mmcv = sys.modules.get("mmcv") or types.ModuleType("mmcv")
    mmcv.__g4r_shim__ = True
    ops = types.ModuleType("mmcv.ops")
    ops.RoIAlign, ops.roi_align, ops.RoIAlignFunction = _ra.RoIAlign, _ra.roi_align, _ra.RoIAlignFunction
    ext = types.ModuleType("mmcv._ext")
    ext.roi_align_forward, ext.roi_align_backward = _ra.roi_align_forward, _ra.roi_align_backward
    mmcv.ops, mmcv._ext = ops, ext
    sys.modules.update({"mmcv": mmcv, "mmcv.ops": ops, "mmcv._ext": ext})
    return ops
