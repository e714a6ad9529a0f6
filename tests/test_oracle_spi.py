"""CPU: pins the restated region module (oracle/spi_oracle.py) against outputs of the reference's
own gpt4roi/models/layers.py (tests/golden/make_spi_golden.py), and the restated splice against a
hand-built case."""
import os

import numpy as np
import pytest
import torch

from oracle import spi_oracle as S


def _run_fixture(golden_dir, name, emulate=False):
    z = np.load(os.path.join(golden_dir, name))
    C, B, P = int(z["embed_dims"]), int(z["B"]), int(z["P"])
    m = S.MLVLROIQueryOracle(embed_dims=C, P=P)
    m.load_state_dict(S.synthetic_state(m, int(z["wseed"])))
    m.eval()
    feats, boxes = S.synthetic_inputs(int(z["iseed"]), B, P, C, [int(n) for n in z["n_rois"]])
    with torch.no_grad():
        out = torch.cat(m(feats, boxes, emulate=emulate), 0).numpy()
    return out, z["out"]


def test_restated_module_matches_reference_code(golden_dir):
    got, want = _run_fixture(golden_dir, "spi_module_ref_c64.npz")
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * np.abs(want).max())


def test_bf16_emulation_stays_close_to_fp32(golden_dir):
    got, want = _run_fixture(golden_dir, "spi_module_ref_c64.npz", emulate=True)
    rel = np.abs(got - want).max() / np.abs(want).max()
    assert rel < 5e-2, rel


def test_state_dict_keys_are_the_reference_checkpoint_keys():
    # SURVEY.md section 5 (checkpoint row): key names the module must keep
    keys = set(S.MLVLROIQueryOracle(embed_dims=64).state_dict().keys())
    for k in ["mlvl_fuse.input_conv.0.weight", "mlvl_fuse.input_conv.3.bias", "mlvl_fuse.fuse_convs.0.conv.weight",
              "mlvl_fuse.fuse_convs.4.gn.weight", "mlvl_fuse.fuse_convs.4.gn.bias", "roi_align.pconvs.3.weight",
              "roi_align.pconvs.0.bias", "roi_align.pos_embedd.0.weight", "roi_align.pos_embedd.2.weight",
              "roi_align.pos_embedd.3.bias", "roi_align.pos_embedd.5.bias", "roi_align.updims.weight",
              "roi_align.flatten_linear.weight"]:
        assert k in keys, k
    assert not any("fuse_convs" in k and k.endswith("conv.bias") for k in keys)  # conv_module.py:104-105


def test_splice_restatement():
    # spi_llava.py:99-196: patches go between <im_start>/<im_end>, <bbox> rows take region features
    IMS, IME, BBOX, T, C, NP = 50, 51, 52, 12, 4, 3
    ids = torch.tensor([[1, IMS, 9, 9, 9, IME, 7, BBOX, 8, BBOX, 2, 2]])
    emb = torch.arange(T * C, dtype=torch.float32).reshape(1, T, C)
    img = -torch.ones(1, NP, C)
    spi = [torch.full((2, C), 100.0) * torch.tensor([[1.0], [2.0]])]
    out = S.splice(ids, emb, img, spi, IMS, IME, BBOX)
    assert torch.equal(out[0, 2:5], img[0]) and torch.equal(out[0, 7], spi[0][0]) and torch.equal(out[0, 9], spi[0][1])
    assert torch.equal(out[0, [0, 1, 5, 6, 8, 10, 11]], emb[0, [0, 1, 5, 6, 8, 10, 11]])
    bad = ids.clone()
    bad[0, 5] = 3
    with pytest.raises(ValueError):
        S.splice(bad, emb, img, spi, IMS, IME, BBOX)


def test_restated_module_backward_matches_reference_code(golden_dir):
    """Training rows: every parameter gradient of the restated module (autograd, RoIAlign = the C oracle's forward +
    backward) against digests of the gradients the REFERENCE'S own layers.py produces on the same seeds
    (tests/golden/make_spi_golden.py::run_grads).  This is what makes the oracle a pinned reference for
    MLVLROIQueryModule.backward (tests/test_train_gpu.py)."""
    z = np.load(os.path.join(golden_dir, "spi_module_ref_grads_c64.npz"))
    C, B, P = int(z["embed_dims"]), int(z["B"]), int(z["P"])
    m = S.MLVLROIQueryOracle(embed_dims=C, P=P)
    m.load_state_dict(S.synthetic_state(m, int(z["wseed"])))
    feats, boxes = S.synthetic_inputs(int(z["iseed"]), B, P, C, [int(n) for n in z["n_rois"]])
    out = torch.cat(m(feats, boxes), 0)
    loss = 0.5 * out.pow(2).sum()
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < 1e-4 * float(z["loss"])
    names = [k for k, _ in m.named_parameters()]
    assert len(names) == 43
    for k, p in m.named_parameters():
        d = z[k.replace(".", "__")]
        f = p.grad.detach().double().flatten()
        n = f.numel()
        assert n == int(d[3]), k
        scale = max(d[2], 1e-12)                                          # largest |gradient| of this tensor
        assert abs(float(f.norm()) - d[0]) < 2e-4 * d[0] + 1e-9, (k, float(f.norm()), d[0])
        head = f[:64].numpy() if n >= 64 else np.pad(f.numpy(), (0, 64 - n))
        idx = (torch.arange(64) * max(1, n // 64)).clamp(max=n - 1)
        np.testing.assert_allclose(head, d[4:68], atol=2e-4 * scale, err_msg=k)
        np.testing.assert_allclose(f[idx].numpy(), d[68:132], atol=2e-4 * scale, err_msg=k)


def _splice_case():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_splice_golden",
                                                  os.path.join(os.path.dirname(__file__), "golden", "make_splice_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_splice_and_level_selection_match_reference_code(golden_dir):
    """Rows a5 + a15: the restated level selection and splice against the output of the reference's own
    SPILlavaLlamaModel.forward (tests/golden/make_splice_golden.py: the decoder parent is stubbed to return the
    spliced inputs_embeds).  Pins oracle.spi_oracle.splice, which the GPU splice kernel is tested against."""
    from oracle import transformer_oracle as T
    z = np.load(os.path.join(golden_dir, "splice_ref.npz"))
    G = _splice_case()
    case = G.build_case(int(z["seed"]))
    ids = case["input_ids"]
    assert np.array_equal(ids.numpy(), z["input_ids"])
    img_feat, levels = T.select_spi_levels(case["hidden"], -2, 4)
    assert [int(round(float(t.mean()))) for t in levels] == z["levels"].tolist() == [14, 17, 20, 23]
    assert [tuple(t.shape) for t in levels] == [tuple(s) for s in z["level_shapes"].tolist()]
    proj = torch.nn.functional.linear(img_feat, case["proj_w"], case["proj_b"])      # as nn.Linear does in the reference
    emb = case["embed"][ids]
    got = S.splice(ids, emb, proj, case["spi"], G.IDS["im_start"], G.IDS["im_end"], G.IDS["bbox"])
    np.testing.assert_allclose(got.numpy(), z["out"], rtol=1e-5, atol=1e-3)   # projector rows are O(1e3) in fp32
    # the two malformed prompts the reference rejects (spi_llava.py:115-128) are rejected with the same messages
    end = 3 + case["n_patch"] + 1
    bad = ids.clone()
    bad[0, end] = 55
    with pytest.raises(ValueError, match="should be the same"):
        S.splice(bad, emb, proj, case["spi"], G.IDS["im_start"], G.IDS["im_end"], G.IDS["bbox"])
    bad = ids.clone()
    bad[0, end], bad[0, end + 1] = bad[0, end + 1].item(), G.IDS["im_end"]
    with pytest.raises(ValueError, match="should follow"):
        S.splice(bad, emb, proj, case["spi"], G.IDS["im_start"], G.IDS["im_end"], G.IDS["bbox"])
    assert "should be the same" in str(z["malformed_error"]) and "should follow" in str(z["malformed_error"])
