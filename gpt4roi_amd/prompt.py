"""Prompt assembly for the region-feature path: the step immediately BEFORE the path.

Mirrors the two functions every caller of the reference runs on a conversation before the model sees it
(gpt4roi/datasets/*.py `__getitem__`, gpt4roi/app.py:171-175):
  preprocess_multimodal(sources, multimodal_cfg, cur_token_len)   gpt4roi/train/train.py:185-208
  preprocess(sources, tokenizer)                                   gpt4roi/train/train.py:354-386  (the default, version-less
      conversation `conv_v1_2` of llava/conversation.py:201-229,349: "### Human: ...\\n### Assistant: ...\\n###")
with their helpers `_add_speaker_and_signal` (:165-183), `_tokenize_fn` (:125-148) and `_mask_targets` (:151-162).
Same contracts: `sources` is a list of conversations (lists of {'from': 'human'|'gpt', 'value': str}), mutated in place like
the reference does; `<image>` becomes `<im_start>` + cur_token_len x `<im_patch>` + `<im_end>`; labels are the input ids with
the header and every human turn masked to IGNORE_INDEX (-100), the two tokens of the "###"-signal of a human turn kept
(`cur_idx + 2`).  The tokenizer is whatever the caller uses (HF call signature); the output feeds
gpt4roi_amd.data.DataCollatorForDetDataset and, through it, the splice kernel -- no further host work per sample.
"""
import copy

import torch

IGNORE_INDEX = -100
DEFAULT_IMAGE_TOKEN = '<image>'
DEFAULT_IMAGE_PATCH_TOKEN = '<im_patch>'
DEFAULT_IM_START_TOKEN = '<im_start>'
DEFAULT_IM_END_TOKEN = '<im_end>'

# llava/conversation.py:201-229 (conv_v1_2, the `default_conversation` of :349): what `preprocess` reads of it
SYSTEM = ("A chat between a curious human and an artificial intelligence assistant. "
          "The assistant gives helpful, detailed, and polite answers to the human's questions.")
ROLES = ("Human", "Assistant")
SEP = "###"


def preprocess_multimodal(sources, multimodal_cfg, cur_token_len):
    """`<image>` -> [<im_start>] + cur_token_len x <im_patch> + [<im_end>] in every turn (train.py:185-208); with
    `sep_image_conv_front` the image placeholder is first moved in front of the first human turn."""
    if not multimodal_cfg['is_multimodal']:
        return sources
    patches = DEFAULT_IMAGE_PATCH_TOKEN * cur_token_len
    if multimodal_cfg['use_im_start_end']:
        patches = DEFAULT_IM_START_TOKEN + patches + DEFAULT_IM_END_TOKEN
    for turns in sources:
        if multimodal_cfg['sep_image_conv_front']:
            first = turns[0]
            assert DEFAULT_IMAGE_TOKEN in first['value']
            text = first['value'].replace(DEFAULT_IMAGE_TOKEN, '').strip()
            first['value'] = f"{DEFAULT_IMAGE_TOKEN}{SEP}{ROLES[0]}: {text}"
        for turn in turns:
            turn['value'] = turn['value'].replace(DEFAULT_IMAGE_TOKEN, patches)
    return sources


def _ids_and_length(text, tokenizer):
    """One string through the caller's tokenizer (HF call signature, train.py:125-148): (ids, number of non-pad ids)."""
    enc = tokenizer(text, return_tensors='pt', padding='longest', max_length=tokenizer.model_max_length, truncation=True)
    ids = enc.input_ids[0]
    return ids, int(enc.input_ids.ne(tokenizer.pad_token_id).sum().item())


def _render_turns(turns):
    """Rewrites every turn IN PLACE to its rendered form "### <Role>: <text>\n" (train.py:165-183: roles of conv_v1_2,
    'unknown' for anything else) and returns the rendered turns."""
    names = {'human': ROLES[0], 'gpt': ROLES[1]}
    for turn in turns:
        turn['value'] = f"{SEP} {names.get(turn['from'].lower(), 'unknown')}: {turn['value']}\n"
    return [t['value'] for t in turns]


def preprocess(sources, tokenizer):
    """The version-less conversation format (train.py:354-386 with `default_conversation = conv_v1_2`): header + rendered
    turns + a trailing "### ", tokenised as ONE string; labels = ids with the header and the body of every human turn set
    to IGNORE_INDEX.  Span lengths come from tokenising each piece separately, as the reference does (:151-162): the first
    two ids of a human turn (its "###" signal) stay unmasked.
    -> dict(input_ids=[LongTensor], labels=[LongTensor]), one entry per conversation."""
    header = f'{SYSTEM}\n\n'
    header_len = None
    all_ids, all_labels = [], []
    for turns in sources:
        rendered = _render_turns(turns)
        ids, _ = _ids_and_length(header + ''.join(rendered) + f'{SEP} ', tokenizer)
        labels = ids.clone()
        if header_len is None:
            header_len = _ids_and_length(header, tokenizer)[1]
        labels[:header_len] = IGNORE_INDEX
        pos = header_len
        for turn, text in zip(turns, rendered):
            n = _ids_and_length(text, tokenizer)[1]
            if turn['from'] == 'human':
                labels[pos + 2:pos + n] = IGNORE_INDEX
            pos += n
        all_ids.append(ids)
        all_labels.append(labels)
    return dict(input_ids=all_ids, labels=all_labels)


def region_question(question, boxes_present=True):
    """gpt4roi/app.py:145-151: `<regionN>` / `<N>` / `<>` in the user's text become `regionN <bbox>` / `<bbox>`."""
    import re
    if boxes_present:
        question = re.sub(r'<region(\d+)>', r'region\g<1> <bbox>', question)
        question = re.sub(r'\<(\d+)\>', r'region\g<1> <bbox>', question)
        question = question.replace('<>', '<bbox>')
    return question


def build_sample(conversation, tokenizer, image_token_len, use_im_start_end=True):
    """The three dataset lines every reference `__getitem__` ends with (e.g. refcoco.py:262-275):
    preprocess_multimodal -> preprocess -> {input_ids, labels} of the single conversation."""
    sources = preprocess_multimodal(copy.deepcopy([conversation]),
                                    dict(is_multimodal=True, sep_image_conv_front=False, use_im_start_end=use_im_start_end),
                                    image_token_len)
    out = preprocess(sources, tokenizer)
    return dict(input_ids=torch.as_tensor(out['input_ids'][0]), labels=torch.as_tensor(out['labels'][0]))
