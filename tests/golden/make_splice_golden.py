"""tests/golden/make_splice_golden.py -- golden output of the REFERENCE'S OWN model-level forward up to the decoder
call: level selection (spi_llava.py:58-82) and the patch splice + <bbox> injection (spi_llava.py:99-196).

Runs only in the build container (needs /root/reference).  Imports /root/reference/gpt4roi/models/spi_llava.py
unmodified; its two unimportable imports (`gpt4roi.models.layers` -> mmcv/mmdet, `llava.model.llava` -> clashes with
transformers 5.x) are stub modules: the LLaVA base class is an nn.Module whose parent `forward` simply returns the
`inputs_embeds` it is handed, so calling the reference's `SPILlavaLlamaModel.forward` yields exactly the spliced
embedding tensor the decoder would receive.  Vision tower, projector, region module and tokenizer are seeded stubs
(the region module returns given features and records which hidden states it was handed).

Stored (tests/golden/splice_ref.npz): the seeds, the token ids, the reference's spliced embeddings [B, T, C] and the
hidden-state indices the reference passed to the region module.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gpt4roi/models/spi_llava.py"
IDS = dict(im_patch=100, bbox=101, im_start=103, im_end=104)


def import_reference():
    layers = types.ModuleType("gpt4roi.models.layers")
    layers.MLVLROIQueryModule = object
    llava_mod = types.ModuleType("llava.model.llava")

    class _Decoder(nn.Module):                     # stands in for transformers' LlamaModel
        def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, **kw):
            return inputs_embeds

    class LlavaLlamaModel(_Decoder):
        pass

    class LlavaLlamaForCausalLM(nn.Module):
        pass

    llava_mod.LlavaLlamaModel, llava_mod.LlavaLlamaForCausalLM = LlavaLlamaModel, LlavaLlamaForCausalLM
    llava_mod.DEFAULT_IMAGE_PATCH_TOKEN, llava_mod.DEFAULT_IM_START_TOKEN, llava_mod.DEFAULT_IM_END_TOKEN = \
        '<im_patch>', '<im_start>', '<im_end>'
    stubs = {"gpt4roi": types.ModuleType("gpt4roi"), "gpt4roi.models": types.ModuleType("gpt4roi.models"),
             "gpt4roi.models.layers": layers, "llava": types.ModuleType("llava"),
             "llava.model": types.ModuleType("llava.model"), "llava.model.llava": llava_mod}
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location("ref_spi_llava", REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def build_case(seed, B=2, P=3, C=1024, D=24, n_rois=(3, 1)):   # C = 1024: the reference projects a hard-wired zeros(256, 1024)
    """Seeded inputs shared by this script and tests/test_oracle_spi.py."""
    g = torch.Generator().manual_seed(seed)
    n_patch = P * P
    hidden = [torch.randn(B, n_patch + 1, C, generator=g) + 1.0 * i for i in range(25)]   # the mean encodes the index
    embed = torch.randn(120, D, generator=g)
    proj_w, proj_b = torch.randn(D, C, generator=g), torch.randn(D, generator=g)
    spi = [torch.randn(n, D, generator=g) for n in n_rois]
    rows = []
    for b, n in enumerate(n_rois):
        seq = [1, 7 + b, 9] + [IDS["im_start"]] + [IDS["im_patch"]] * n_patch + [IDS["im_end"]]
        for r in range(n):
            seq += [20 + r, IDS["bbox"], 30]
        rows.append(seq)
    T = max(len(r) for r in rows)
    rows = [r + [40 + i for i in range(T - len(r))] for r in rows]           # same T (the reference stacks)
    return dict(hidden=hidden, embed=embed, proj_w=proj_w, proj_b=proj_b, spi=spi,
                input_ids=torch.tensor(rows, dtype=torch.int64), n_patch=n_patch)


def run_reference(ref, case):
    m = ref.SPILlavaLlamaModel.__new__(ref.SPILlavaLlamaModel)
    nn.Module.__init__(m)
    m.num_level_spi_features = 4
    m.embed_tokens = nn.Embedding.from_pretrained(case["embed"])
    m.mm_projector = nn.Linear(case["proj_w"].size(1), case["proj_w"].size(0))
    with torch.no_grad():
        m.mm_projector.weight.copy_(case["proj_w"])
        m.mm_projector.bias.copy_(case["proj_b"])
    seen = {}

    class Tower(nn.Module):
        config = types.SimpleNamespace(im_patch_token=IDS["im_patch"], use_im_start_end=True,
                                       im_start_token=IDS["im_start"], im_end_token=IDS["im_end"])

        def forward(self, images, output_hidden_states=True):
            return types.SimpleNamespace(hidden_states=case["hidden"])

    def spi_module(mlvl, bboxes):
        seen["levels"] = [int(round(float(t.mean()))) for t in mlvl]
        seen["shapes"] = [tuple(t.shape) for t in mlvl]
        return case["spi"]

    object.__setattr__(m, "vision_tower", [Tower()])
    object.__setattr__(m, "spi_module", spi_module)
    m.config = types.SimpleNamespace(mm_vision_select_layer=-2)
    m.tokenizer = types.SimpleNamespace(convert_tokens_to_ids=lambda toks: [IDS["bbox"]])
    m.eval()
    images = torch.zeros(case["input_ids"].size(0), 3, 4, 4)
    boxes = [torch.zeros(s.size(0), 4) for s in case["spi"]]
    with torch.no_grad():
        out = m.forward(input_ids=case["input_ids"], images=images, bboxes=boxes)
    return out, seen


if __name__ == "__main__":
    ref = import_reference()
    case = build_case(7)
    out, seen = run_reference(ref, case)
    assert seen["levels"] == [14, 17, 20, 23], seen
    # malformed prompts -> the reference raises (spi_llava.py:115-128)
    def error_of(ids):
        bad = dict(case)
        bad["input_ids"] = ids
        try:
            run_reference(ref, bad)
            return ""
        except ValueError as e:
            return str(e)
    end = 3 + case["n_patch"] + 1                                   # position of <im_end> in row 0
    ids = case["input_ids"].clone()
    ids[0, end] = 55                                                # <im_end> missing
    err_count = error_of(ids)
    ids = case["input_ids"].clone()
    ids[0, end], ids[0, end + 1] = ids[0, end + 1].item(), IDS["im_end"]   # <im_end> one position late
    err_place = error_of(ids)
    assert "should be the same" in err_count and "should follow" in err_place, (err_count, err_place)
    raised = err_count + " | " + err_place
    np.savez_compressed(os.path.join(HERE, "splice_ref.npz"), seed=7, out=out.numpy(), input_ids=case["input_ids"].numpy(),
                        levels=np.array(seen["levels"]), level_shapes=np.array(seen["shapes"]), malformed_error=raised)
    print("splice_ref.npz", tuple(out.shape), "levels", seen["levels"], "| malformed ->", raised)
