"""Batch contract of the path's callers: what sits immediately BEFORE `forward(input_ids, images,
bboxes, labels)`.

Mirrors /root/reference/gpt4roi/datasets/data_modules.py:22-56 (`DataCollatorForDetDataset`): instances are dicts
with `input_ids`, `labels` (1-D int64), optional `image` ([3,S,S] float), `img_metas`, `bboxes` ([n_i,4] normalised
xyxy, refcoco.py:296); the collator right-pads ids with the pad token and labels with IGNORE_INDEX = -100
(gpt4roi/train/train.py:34), builds the attention mask, stacks the images when they share a shape and keeps the
boxes as a list.  `to_device` is the hand-over to the gfx950 path: ids/labels/images move to the GPU once and the
boxes become a layers.PreparedBoxes (RoI table, prefix sums) so that the step itself has no host traffic.

The token-level prompt construction (`preprocess_multimodal`: one `<image>` -> <im_start> + P^2 x <im_patch> +
<im_end>, llava/train/train.py; `<bbox>` placeholders, gpt4roi/train/train.py:185-208) works on tokenizer output
and stays with the tokenizer; `expand_image_tokens` below is its id-level core, used by the synthetic workloads.
"""
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

IGNORE_INDEX = -100


@dataclass
class DataCollatorForDetDataset:
    pad_token_id: int = 0

    def __call__(self, instances: Sequence[dict]) -> dict:
        input_ids, labels, img_metas, bboxes = ([inst.get(k, None) for inst in instances]
                                                for k in ('input_ids', 'labels', 'img_metas', 'bboxes'))
        input_ids = torch.nn.utils.rnn.pad_sequence(input_ids, batch_first=True, padding_value=self.pad_token_id)
        labels = torch.nn.utils.rnn.pad_sequence(labels, batch_first=True, padding_value=IGNORE_INDEX)
        batch = dict(input_ids=input_ids, labels=labels, attention_mask=input_ids.ne(self.pad_token_id),
                     img_metas=img_metas, bboxes=bboxes)
        if 'image' in instances[0]:
            images = [inst['image'] for inst in instances]
            if all(x is not None and x.shape == images[0].shape for x in images):
                batch['images'] = torch.stack(images)
            else:
                batch['images'] = images
        return batch


def expand_image_tokens(ids, image_token_id, token_ids, n_patch):
    """Replace every `image_token_id` in a 1-D id tensor by <im_start> + n_patch x <im_patch> + <im_end>
    (mm_use_im_start_end=True, the only mode GPT4RoI trains with: train_stage1.sh)."""
    out = []
    for t in ids.tolist():
        if t == image_token_id:
            out += [token_ids.im_start_token] + [token_ids.im_patch_token] * n_patch + [token_ids.im_end_token]
        else:
            out.append(t)
    return torch.tensor(out, dtype=torch.int64)


def to_device(batch: dict, device, image_size: Optional[int] = None) -> dict:
    """Collated host batch -> what RegionTrainer.step / SPILlavaLlamaModel.forward take on the GPU."""
    from .layers import PreparedBoxes
    out = dict(input_ids=batch['input_ids'].to(device, non_blocking=True),
               labels=batch['labels'].to(device, non_blocking=True) if batch.get('labels') is not None else None)
    images = batch.get('images')
    if images is not None:
        if isinstance(images, (list, tuple)):
            images = torch.stack([im.to(device, non_blocking=True) for im in images])
        else:
            images = images.to(device, non_blocking=True)
        out['images'] = images
        image_size = image_size or images.size(-1)
    if batch.get('bboxes') is not None and image_size is not None:
        boxes = [b if b is not None else torch.zeros(0, 4) for b in batch['bboxes']]
        out['bboxes'] = PreparedBoxes(boxes, image_size, device)
    return out
