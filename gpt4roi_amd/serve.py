"""Headless conversation loop around the region-feature path: what `ConversationBot` of the reference's
demo does between the UI and the model, without the UI.

Mirrors gpt4roi/app.py:
  count_num_bboxes            :47-50    `<regionN>` mentions of a question
  ConversationBot.build_model :62-108   -> `ConversationBot.from_pretrained` (checkpoint.from_pretrained + bind_tokenizer)
  ConversationBot.init_inputs :110-188  boxes normalised by the image size, image -> 224^2 pixel tensor, first-round preamble,
                                        `<regionN>` -> `regionN <bbox>`, preprocess_multimodal + preprocess, box carry-over
  ConversationBot.check_input :190-239  number of `<regionN>` in the text vs boxes drawn this round; re-mentions of a known
                                        region lose their brackets
  ConversationBot.run         :243-328  KeywordsStoppingCriteria('###'), forward bound with partial(bboxes=...), sampling
                                        generate(temperature 0.2, <= 1024 new tokens), answer clean-up, history update
Not mirrored: Gradio widgets, the box overlay drawing (`visualize`), chat-history HTML escaping.

The history cache has the reference's structure (a list whose last entry holds 'sources', 'region_name_set', 'bboxes') and
the same aliasing: from the second round on the stored conversation is the object the prompt functions rewrite in place,
so a third round sees the already rendered turns exactly like the reference does (pinned by tests/golden/bot_ref.json,
generated from the reference's own class).

Image step: the reference resizes with CLIPImageProcessor (bicubic, shortest edge 224, no crop) and then bilinearly to
224 x 224 (:125-136); here `kernels.image_preprocess` does normalise + bilinear resize to 224 x 224 in one launch from the
uint8 image.  `preprocess_image` can be replaced by the caller (e.g. with the HF processor) -- everything
after it only needs a float [3, S, S] tensor.
"""
import copy
import re
from functools import partial

import numpy as np
import torch

from .generation import KeywordsStoppingCriteria
from .prompt import preprocess, preprocess_multimodal

MULTIMODAL_CFG = dict(is_multimodal=True, sep_image_conv_front=False, image_token_len=256, image_aspect_ratio='square',
                      use_im_start_end=True)                       # app.py:33-39
PREAMBLE = "The <image> provides an overview of the picture.\n"     # app.py:141
STOP_STR = '###'
EMPTY_ANSWER = "There is internal error. Please click 'Clear All' and try again."     # app.py:316 (what the caller shows)


def count_num_bboxes(text):
    """-> (number of `<regionN>` mentions, the mentions)  (app.py:47-50)"""
    found = re.findall(r'<region\d+>', text)
    return len(found), found


def _mark_regions(text):
    """`<regionN>` / `<N>` -> `regionN <bbox>`, `<>` -> `<bbox>`  (app.py:148-150, 164-166)"""
    text = re.sub(r'<region(\d+)>', r'region\g<1> <bbox>', text)
    text = re.sub(r'\<(\d+)\>', r'region\g<1> <bbox>', text)
    return text.replace('<>', '<bbox>')


def _image_size(image):
    """(width, height) of a PIL image / HWC array / HWC tensor."""
    if hasattr(image, 'size') and not isinstance(image, (np.ndarray, torch.Tensor)):
        return image.size
    return int(image.shape[1]), int(image.shape[0])


class ConversationBot:
    def __init__(self, model, tokenizer, image_size=224, device='cuda', patch=14):
        self.model, self.tokenizer = model, tokenizer
        self.image_size, self.device, self.patch = image_size, device, patch

    @classmethod
    def from_pretrained(cls, model_name, tokenizer, vision_tower=None, device='cuda', **kw):
        """build_model (app.py:62-108): checkpoint directory -> model on the GPU, special tokens bound."""
        from .checkpoint import from_pretrained
        from .spi_llava import SPILlavaMPTForCausalLM
        model = from_pretrained(SPILlavaMPTForCausalLM, model_name, device=device, vision_tower=vision_tower,
                                tokenizer=tokenizer, **kw)
        return cls(model, tokenizer, device=device)

    # ---- image -----------------------------------------------------------------------------------------------------
    def preprocess_image(self, image):
        """uint8 RGB image (PIL / HWC array / HWC tensor) -> float [3, S, S] on the device, CLIP-normalised."""
        from . import kernels as K
        if not isinstance(image, torch.Tensor):
            arr = np.asarray(image.convert('RGB') if hasattr(image, 'convert') else image)
            image = torch.from_numpy(np.ascontiguousarray(arr))
        return K.image_preprocess(image.to(self.device), self.image_size)

    # ---- one round: host-side assembly ---------------------------------------------------------------------------
    def init_inputs(self, input_dict, question_str, history_cache):
        """-> (dict(input_ids, labels, sources, image, bboxes), history_cache)   (app.py:110-188)"""
        boxes = input_dict['boxes']
        have_boxes = len(boxes) > 0
        width, height = _image_size(input_dict['image'])
        if have_boxes:
            norm = np.array(boxes, dtype=np.float64) / np.array([width, height, width, height])
        image = self.preprocess_image(input_dict['image'])
        n_patch = (image.shape[1] // self.patch) * (image.shape[2] // self.patch)
        if not history_cache:
            question = PREAMBLE + question_str
            names = count_num_bboxes(question)[1]
            if have_boxes:
                question = _mark_regions(question)
            sources = {'conversations': [{'from': 'human', 'value': question}]}
            history_cache.append({'sources': copy.deepcopy(sources), 'region_name_set': set(names)})
        else:
            sources = history_cache[-1]['sources']           # the stored object itself: rewritten in place below
            sources['conversations'].append({'from': 'human', 'value': _mark_regions(question_str)})
        turns = preprocess_multimodal([sources['conversations']], MULTIMODAL_CFG, n_patch)
        rendered = copy.deepcopy(turns)
        enc = preprocess(turns, self.tokenizer)
        data = dict(input_ids=enc['input_ids'][0], labels=enc['labels'][0], sources=rendered, image=image)
        data['bboxes'] = torch.Tensor(norm) if have_boxes else history_cache[-1]['bboxes']
        history_cache[-1]['bboxes'] = copy.deepcopy(data['bboxes'])
        return data, history_cache

    def check_input(self, text, image, history_cache):
        """-> (error message or None, text with re-mentioned regions un-bracketed)   (app.py:190-239)"""
        if image is None:
            return 'No image: upload an image first.', text
        first = len(history_cache) == 0
        n_boxes = len(image['boxes'])
        if first:
            if n_boxes == 0:
                return 'No region of interest: draw at least one box on the image.', text
            drawn = n_boxes
        else:
            drawn = 0 if n_boxes == 0 else n_boxes - len(history_cache[-1]['bboxes'])
            known = copy.deepcopy(history_cache[-1]['region_name_set'])
            for name in count_num_bboxes(text)[1]:
                if name in known:
                    text = text.replace(name, name[1:-1])     # an old region is referred to by its plain name
                else:
                    known.add(name)
            history_cache[-1]['region_name_set'] = known
        mentioned = count_num_bboxes(text)[0]
        if mentioned != drawn:
            if mentioned == 0:
                return (f'The question {text!r} does not refer to the drawn boxes (use the <regionN> format).'), text
            return (f'The question {text!r} mentions {mentioned} <regionN>, but {drawn} boxes were drawn this round.'), text
        return None, text

    # ---- one round: model ------------------------------------------------------------------------------------------
    def run(self, text, image, history_cache, do_sample=True, temperature=0.2, max_new_tokens=1024, seed=None):
        """One question -> (answer text or None, error message or None, history_cache)   (app.py:243-328, no UI state)."""
        error, text = self.check_input(text, image, history_cache)
        if error is not None:
            return None, error, history_cache
        text = text.strip()
        if not text:
            text = 'hello, world!'                           # app.py:256-258
        inputs, history_cache = self.init_inputs(image, text, history_cache)
        bboxes = inputs['bboxes']
        if bboxes is not None:
            bboxes = [bboxes.to(self.device)]
        input_ids = inputs['input_ids'].to(self.device)[None]
        criteria = KeywordsStoppingCriteria([STOP_STR], self.tokenizer, input_ids)
        model = self.model
        with torch.inference_mode():
            model.orig_forward = model.forward
            model.forward = partial(model.orig_forward, img_metas=[None], bboxes=bboxes)
            try:
                output_ids = model.generate(input_ids, images=inputs['image'][None].to(self.device), do_sample=do_sample,
                                            temperature=temperature, max_new_tokens=max_new_tokens,
                                            stopping_criteria=[criteria], seed=seed)
            finally:
                model.forward = model.orig_forward
        T = input_ids.shape[1]
        answer = self.tokenizer.batch_decode(output_ids[:, T:], skip_special_tokens=True)[0].strip()
        if answer.endswith(STOP_STR):
            answer = answer[:-len(STOP_STR)]
        answer = answer.strip() or EMPTY_ANSWER
        shown = answer
        answer = answer.replace('Assistant: ', '').replace('Assistant:', '')
        history_cache[-1]['sources']['conversations'].append({'from': 'gpt', 'value': answer})
        return shown, None, history_cache
