"""tests/golden/make_collator_golden.py -- golden output of the REFERENCE'S OWN DataCollatorForDetDataset
(/root/reference/gpt4roi/datasets/data_modules.py:22-56), imported unmodified with its dataset / mmcv / llava imports
stubbed (only `IGNORE_INDEX` from gpt4roi/train/train.py:34 is real: -100).  Runs only in the build container."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gpt4roi/datasets/data_modules.py"


def instances():
    g = torch.Generator().manual_seed(5)
    lens = [9, 4, 6]
    out = []
    for i, n in enumerate(lens):
        ids = torch.randint(1, 50, (n,), generator=g)
        lab = ids.clone()
        lab[: n // 2] = -100
        out.append(dict(input_ids=ids, labels=lab, image=torch.randn(3, 8, 8, generator=g), img_metas=dict(idx=i),
                        bboxes=torch.rand(i + 1, 4, generator=g)))
    return out


def import_reference():
    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        return m
    stubs = {n: pkg(n) for n in ("gpt4roi", "gpt4roi.datasets", "gpt4roi.models", "gpt4roi.train", "llava", "llava.train", "mmcv")}
    leaf = {"gpt4roi.datasets.det_llava": ["DetLLava"], "gpt4roi.datasets.refcoco": ["RefCOCO", "RefCOCOG", "RefCOCOP"],
            "gpt4roi.datasets.vg": ["VGDATA"], "gpt4roi.models.spi_llava": ["add_spatial_token"],
            "llava.train.train": ["LazySupervisedDataset"], "gpt4roi.datasets.coco_det": ["CocoDet"],
            "gpt4roi.datasets.flickr30k": ["Flickr30k"], "gpt4roi.datasets.vcr": ["MultiVCRDataset", "SingleVCRDataset", "VCRDataset"]}
    for name, attrs in leaf.items():
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, object)
        stubs[name] = m
    tr = types.ModuleType("gpt4roi.train.train")
    tr.IGNORE_INDEX = -100                                  # gpt4roi/train/train.py:34
    stubs["gpt4roi.train.train"] = tr
    stubs["mmcv"].Config = object
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location("gpt4roi.datasets.data_modules", REF)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["gpt4roi.datasets.data_modules"] = mod
        spec.loader.exec_module(mod)
    finally:
        sys.modules.pop("gpt4roi.datasets.data_modules", None)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


if __name__ == "__main__":
    ref = import_reference()
    coll = ref.DataCollatorForDetDataset(tokenizer=types.SimpleNamespace(pad_token_id=0))
    b = coll(instances())
    ragged = instances()
    ragged[1]["image"] = torch.zeros(3, 4, 4)
    b2 = coll(ragged)
    np.savez_compressed(os.path.join(HERE, "collator_ref.npz"), input_ids=b["input_ids"].numpy(), labels=b["labels"].numpy(),
                        attention_mask=b["attention_mask"].numpy(), images=b["images"].numpy(),
                        n_boxes=np.array([x.shape[0] for x in b["bboxes"]]), metas=np.array([m["idx"] for m in b["img_metas"]]),
                        ragged_images_is_list=isinstance(b2["images"], list), keys=np.array(sorted(b.keys())))
    print("collator_ref.npz", sorted(b.keys()), b["input_ids"].shape, "ragged images ->", type(b2["images"]).__name__)
