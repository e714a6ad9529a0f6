"""CPU: the drop-in boundary without a GPU -- the C-ABI library loads and exports every symbol the
headers declare, the Python mirrors keep the reference's signatures / state_dict keys, and the
product path refuses to run on CPU tensors (no fallback)."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gpt4roi_amd", "lib", "libgpt4roi_hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        from gpt4roi_amd import build
        build.build()
    return ctypes.CDLL(LIB)


def test_library_exports_every_declared_symbol(lib):
    names = set()
    for h in ("g4r_roi_align.h", "g4r_kernels.h", "g4r_train.h"):
        names |= set(re.findall(r"\b(g4r_\w+)\s*\(", open(os.path.join(ROOT, "include", h)).read()))
    assert len(names) >= 45
    for n in sorted(names):
        assert hasattr(lib, n), n
    assert lib.g4r_abi_version() == 5


def test_argument_validation_without_a_gpu(lib):
    # status codes come back before any launch: bad pool_mode, bad head_dim
    rc = lib.g4r_roi_align_forward_f32(None, None, None, None, None, 1, 1, 4, 4, 1, 2, 2, ctypes.c_float(1.0), 2, 7, 1, None)
    assert rc == 1
    lib.g4r_last_error.restype = ctypes.c_char_p
    assert b"pool_mode" in lib.g4r_last_error()
    rc = lib.g4r_roi_align_forward_f32(None, None, None, None, None, 1, 1, 4, 4, 0, 2, 2, ctypes.c_float(1.0), 2, 1, 1, None)
    assert rc == 0  # zero RoIs: nothing to do, as the reference's empty launch


def test_roi_align_signatures_match_mmcv():
    from gpt4roi_amd.roi_align import RoIAlign, RoIAlignFunction
    # mmcv/ops/roi_align.py:64-72 and :177-191
    assert list(inspect.signature(RoIAlignFunction.forward).parameters) == [
        'ctx', 'input', 'rois', 'output_size', 'spatial_scale', 'sampling_ratio', 'pool_mode', 'aligned']
    m = RoIAlign(14, spatial_scale=1 / 7, sampling_ratio=2)
    assert m.output_size == (14, 14) and m.pool_mode == 'avg' and m.aligned is True
    with pytest.warns(UserWarning):
        m2 = RoIAlign(out_size=7, sample_num=2)           # deprecated aliases, roi_align.py:171-176
    assert m2.output_size == (7, 7) and m2.sampling_ratio == 2
    assert repr(m).startswith("RoIAlign(output_size=(14, 14), spatial_scale=0.14285714285714285, sampling_ratio=2")


def test_no_cpu_fallback():
    from gpt4roi_amd._lib import HipKernelError
    from gpt4roi_amd.roi_align import roi_align
    with pytest.raises(HipKernelError):
        roi_align(torch.zeros(1, 1, 4, 4), torch.zeros(1, 5), 2, 1.0, 2, 'avg', True)


def test_region_module_keeps_reference_state_dict_keys():
    from gpt4roi_amd.layers import MLVLROIQueryModule
    from oracle.spi_oracle import MLVLROIQueryOracle
    ours = MLVLROIQueryModule(embed_dims=64, out_dims=4096, num_levels=4).state_dict()
    ref = MLVLROIQueryOracle(embed_dims=64).state_dict()           # keys checked against the reference
    assert sorted(ours.keys()) == sorted(ref.keys())
    assert all(ours[k].shape == ref[k].shape for k in ref)


def test_model_forward_signature_matches_reference():
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel, SPILlavaMPTForCausalLM
    # gpt4roi/models/spi_llava.py:23-36
    want = ['self', 'input_ids', 'attention_mask', 'img_metas', 'bboxes', 'past_key_values', 'inputs_embeds',
            'use_cache', 'output_attentions', 'output_hidden_states', 'images', 'return_dict']
    assert list(inspect.signature(SPILlavaLlamaModel.forward).parameters)[:len(want)] == want
    p = inspect.signature(SPILlavaMPTForCausalLM.forward).parameters
    assert p['img_metas'].kind is inspect.Parameter.KEYWORD_ONLY and p['bboxes'].kind is inspect.Parameter.KEYWORD_ONLY


def test_mmcv_shim_resolves_the_op_by_name():
    import sys
    from gpt4roi_amd import mmcv_shim
    saved = {k: sys.modules.pop(k, None) for k in ("mmcv", "mmcv.ops", "mmcv._ext")}
    try:
        ops = mmcv_shim.install()
        import mmcv.ops                                         # noqa: F401
        layer_cls = getattr(ops, 'RoIAlign')                    # base_roi_extractor.py:56-57
        layer = layer_cls(spatial_scale=1 / 14, output_size=14, sampling_ratio=2)
        assert layer.output_size[0] == 14                       # read back at layers.py:270,286
        assert hasattr(sys.modules["mmcv._ext"], "roi_align_forward")
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def test_conv_wgrad_slice_planner_on_the_host():
    """g4r_conv3x3_wgrad_nhwc_slices is host code (no GPU): the slice count the workspace is sized by -- every level cut
    into slices of about `slice_tiles` K tiles of 32 bordered pixels; unsupported shapes answer -1."""
    import ctypes

    from gpt4roi_amd import _lib
    lib = _lib.lib()
    f = lib.g4r_conv3x3_wgrad_nhwc_slices
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4

    def slices(sizes, B, cin, cout, st):
        h = (ctypes.c_int * len(sizes))(*[s[0] for s in sizes])
        w = (ctypes.c_int * len(sizes))(*[s[1] for s in sizes])
        return f(len(sizes), ctypes.cast(h, ctypes.c_void_p), ctypes.cast(w, ctypes.c_void_p), B, cin, cout, st)

    def expect(sizes, B, st):
        n = 0
        for hh, ww in sizes:
            nk = -(-(B * (hh + 2) * (ww + 2)) // 32)
            n += max(1, -(-nk // st))
        return n
    pyr = [(192, 192), (96, 96), (48, 48), (24, 24)]
    assert slices(pyr, 8, 1024, 1024, 589) == expect(pyr, 8, 589) == 24
    assert slices([(14, 14)], 64, 1024, 1024, 64) == expect([(14, 14)], 64, 64)
    assert slices([(7, 33)], 3, 256, 768, 8) == expect([(7, 33)], 3, 8)
    assert slices(pyr, 8, 1000, 1024, 589) == -1            # channels must be multiples of 256
    assert slices(pyr + [(12, 12)], 8, 1024, 1024, 589) == -1        # at most four levels
    assert slices(pyr, 8, 1024, 1024, 4) == -1              # slices of at least 8 K tiles
