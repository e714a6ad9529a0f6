"""tests/golden/make_spi_golden.py -- golden outputs of the REFERENCE'S OWN region module.

Runs only in the build container (needs /root/reference).  Imports
/root/reference/gpt4roi/models/layers.py unmodified; its two unavailable leaf imports
(`mmcv.cnn`, `mmdet.models` -- mmcv cannot be imported here: addict/yapf/cv2/_ext missing) are
satisfied by stub modules exposing the restated leaves of oracle/spi_oracle.py
(ConvModule = conv/GN/ReLU, Linear, normal_init, BaseRoIExtractor with the C-oracle RoIAlign).
Everything above the leaves -- MLVLROIQueryModule / MLVLFuseModule / MlvlRoIExtractor forward,
the level shuffle, coordinate channels, flatten order, pos-embed -- is the reference's code.

Weights and inputs are NOT stored: oracle.spi_oracle.synthetic_state / synthetic_inputs rebuild
them from the seeds recorded in the fixture.  Stored: the module outputs [sum n_i, 4096] (fp32)
and per-stage statistics.

  spi_module_ref_c64.npz   embed_dims 64,  B 2, rois (3, 2)   -> CPU pin of the restatement
  spi_module_ref_c512.npz  embed_dims 512, B 1, rois (5,)     -> GPU parity fixture (224^2, P 16)
  spi_module_ref_grads_c64.npz  digests of every parameter GRADIENT of the reference module (loss = sum(out^2)/2)
                           -> CPU pin of the oracle's backward (the reference the training rows are checked against)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import spi_oracle as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LAYERS = "/root/reference/gpt4roi/models/layers.py"


def import_reference_layers():
    def conv_module(cin, cout, k, stride=1, padding=0, conv_cfg=None, norm_cfg=None):
        assert k == 3 and stride == 1 and padding == 1 and conv_cfg is None
        assert norm_cfg['type'] == 'GN'
        return S.ConvModuleOracle(cin, cout, groups=norm_cfg['num_groups'])

    def normal_init(module, mean=0, std=1, bias=0):
        if getattr(module, 'weight', None) is not None:
            nn.init.normal_(module.weight, mean, std)
        if getattr(module, 'bias', None) is not None:
            nn.init.constant_(module.bias, bias)

    mmcv = types.ModuleType("mmcv")
    cnn = types.ModuleType("mmcv.cnn")
    cnn.ConvModule, cnn.Linear, cnn.normal_init = conv_module, nn.Linear, normal_init
    mmcv.cnn = cnn
    mmdet = types.ModuleType("mmdet")
    models = types.ModuleType("mmdet.models")
    models.BaseRoIExtractor = S.BaseRoIExtractorOracle
    mmdet.models = models
    saved = {k: sys.modules.get(k) for k in ("mmcv", "mmcv.cnn", "mmdet", "mmdet.models")}
    sys.modules.update({"mmcv": mmcv, "mmcv.cnn": cnn, "mmdet": mmdet, "mmdet.models": models})
    try:
        spec = importlib.util.spec_from_file_location("ref_gpt4roi_layers", REF_LAYERS)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def run(ref, embed_dims, B, n_rois, wseed, iseed, name):
    torch.manual_seed(0)
    m = ref.MLVLROIQueryModule(embed_dims=embed_dims, out_dims=4096, num_levels=4)
    probe = S.MLVLROIQueryOracle(embed_dims=embed_dims, P=16)
    assert sorted(m.state_dict().keys()) == sorted(probe.state_dict().keys()), "state_dict keys differ"
    m.load_state_dict(S.synthetic_state(m, wseed))
    m.eval()
    feats, boxes = S.synthetic_inputs(iseed, B, 16, embed_dims, n_rois)
    with torch.no_grad():
        out = m([f.clone() for f in feats], boxes)
    out = torch.cat(out, 0).numpy()
    np.savez_compressed(os.path.join(HERE, name), out=out, embed_dims=embed_dims, B=B, n_rois=np.array(n_rois),
                        wseed=wseed, iseed=iseed, P=16)
    print(name, out.shape, "mean", out.mean(), "std", out.std(), "absmax", np.abs(out).max())


def grad_digest(g):
    """Small, order-sensitive digest of a gradient tensor: norms + 64 leading + 64 strided elements."""
    f = g.detach().double().flatten()
    n = f.numel()
    idx = (torch.arange(64) * max(1, n // 64)).clamp(max=n - 1)
    return np.concatenate([[float(f.norm()), float(f.sum()), float(f.abs().max()), float(n)],
                           f[:64].numpy() if n >= 64 else np.pad(f.numpy(), (0, 64 - n)), f[idx].numpy()])


def run_grads(ref, embed_dims, B, n_rois, wseed, iseed, name):
    """Parameter gradients of the REFERENCE'S module code under autograd (training rows): loss = sum(out^2)/2.
    The RoIAlign leaf is oracle.spi_oracle.RoIAlignOracle, whose autograd node is the C oracle's forward and
    backward -- both bit-exact against the reference's compiled CPU op (tests/test_oracle_roi_align.py)."""
    torch.manual_seed(0)
    m = ref.MLVLROIQueryModule(embed_dims=embed_dims, out_dims=4096, num_levels=4)
    m.load_state_dict(S.synthetic_state(m, wseed))
    m.train()
    feats, boxes = S.synthetic_inputs(iseed, B, 16, embed_dims, n_rois)
    out = torch.cat(m([f.clone() for f in feats], boxes), 0)
    (0.5 * out.pow(2).sum()).backward()
    digests = {k.replace(".", "__"): grad_digest(p.grad) for k, p in m.named_parameters()}
    np.savez_compressed(os.path.join(HERE, name), embed_dims=embed_dims, B=B, n_rois=np.array(n_rois), wseed=wseed,
                        iseed=iseed, P=16, loss=float(0.5 * out.detach().pow(2).sum()), **digests)
    print(name, len(digests), "parameter gradients; loss", float(0.5 * out.detach().pow(2).sum()))


if __name__ == "__main__":
    ref = import_reference_layers()
    torch.set_num_threads(os.cpu_count())
    run(ref, 64, 2, (3, 2), 11, 12, "spi_module_ref_c64.npz")
    run_grads(ref, 64, 2, (3, 2), 11, 12, "spi_module_ref_grads_c64.npz")
    if "--small" not in sys.argv:
        run(ref, 512, 1, (5,), 21, 22, "spi_module_ref_c512.npz")
