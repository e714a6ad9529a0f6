"""tests/golden/make_bot_golden.py -- golden outputs of the REFERENCE'S OWN ConversationBot (gpt4roi/app.py:53-239):
`check_input` and `init_inputs` over a three-round conversation, for gpt4roi_amd/serve.py.

Runs only in the build container (needs /root/reference); writes tests/golden/bot_ref.json.  The class is the reference's
file imported unmodified; stubbed are the packages it cannot import here (gradio, cv2), `llava.utils.disable_torch_init`,
and the three collaborators a bot instance holds: the tokenizer (HFToyTokenizer of make_setup_golden.py), the image
processor (returns a fixed-size tensor: only its SHAPE enters the prompt) and the model (unused by these two methods).
The import runs with a temporary working directory because app.py creates an `image/` directory where it is imported.
"""
import copy
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_setup_golden import REF, HFToyTokenizer, _import, ref_utils  # noqa: E402


def import_ref_app():
    gr = types.ModuleType("gradio")
    gr.Image = type("Image", (), {})
    themes = types.ModuleType("gradio.themes")
    base = types.ModuleType("gradio.themes.base")
    gr.themes, themes.base = themes, base
    conv = _import(f"{REF}/llava/conversation.py", "llava.conversation", {})
    llava_pkg = types.ModuleType("llava")
    llava_pkg.conversation = conv
    model_pkg = types.ModuleType("llava.model")
    model_pkg.__all__ = []
    trainer = types.ModuleType("gpt4roi.train.llava_trainer")
    trainer.LLaVATrainer = object
    g_pkg, g_train = types.ModuleType("gpt4roi"), types.ModuleType("gpt4roi.train")
    base_stubs = {"llava": llava_pkg, "llava.conversation": conv, "llava.model": model_pkg, "gpt4roi": g_pkg,
                  "gpt4roi.train": g_train, "gpt4roi.train.llava_trainer": trainer}
    train_mod = _import(f"{REF}/gpt4roi/train/train.py", "gpt4roi.train.train", base_stubs)
    utils_mod = ref_utils()
    llava_utils = types.ModuleType("llava.utils")
    llava_utils.disable_torch_init = lambda: None
    stubs = dict(base_stubs)
    stubs.update({"gradio": gr, "gradio.themes": themes, "gradio.themes.base": base, "cv2": types.ModuleType("cv2"),
                  "gpt4roi.train.train": train_mod, "llava.model.utils": utils_mod, "llava.utils": llava_utils})
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            return _import(f"{REF}/gpt4roi/app.py", "ref_app", stubs)
        finally:
            os.chdir(cwd)


class FakeProcessor:
    """image_processor.preprocess(...)['pixel_values'][0]: a [3, H, W] tensor; the prompt only depends on the 224 x 224
    the bot interpolates it to."""

    def preprocess(self, image, do_center_crop=False, return_tensors='pt'):
        w, h = image.size
        return {'pixel_values': [torch.zeros(3, h, w)]}


def conversation_script():
    """(question, boxes drawn SO FAR in pixels, answer the model is assumed to give).  Image 640 x 480."""
    return [
        ("What is <region1> doing next to <region2> ?", [[10, 20, 200, 220], [300, 40, 620, 460]], "Assistant: sitting ."),
        ("Is <region1> bigger than <region3> ?", [[10, 20, 200, 220], [300, 40, 620, 460], [5, 5, 60, 70]], "yes ."),
        ("And what colour is <region2> ?", [], "brown ."),
    ]


def check_cases():
    """(text, n boxes in the image dict, rounds of history before) -> check_input decisions."""
    return [("Describe <region1> .", 1, 0), ("Describe it .", 1, 0), ("Describe <region1> .", 0, 0),
            ("<region1> and <region2> ?", 1, 0), ("What about <region1> ?", 2, 1), ("Compare <region3> with <region1> .", 3, 1),
            ("Compare <region3> with <region4> .", 3, 1), ("And now ?", 0, 1)]


def run_ref_bot():
    from PIL import Image
    app = import_ref_app()
    pil = Image.fromarray(np.zeros((480, 640, 3), dtype=np.uint8))

    def new_bot():
        bot = object.__new__(app.ConversationBot)
        bot.tokenizer = HFToyTokenizer(model_max_length=2048)
        bot.image_processor = FakeProcessor()
        return bot

    rounds = []
    bot, history = new_bot(), []
    import contextlib
    import io
    for question, boxes, answer in conversation_script():
        image = {'image': pil, 'boxes': boxes}
        with contextlib.redirect_stdout(io.StringIO()):
            err, text = bot.check_input(question, image, history)
            assert err is None, err
            data, history = bot.init_inputs(image, text.strip(), history)
        rounds.append(dict(text_after_check=text, input_ids=data['input_ids'].tolist(), labels=data['labels'].tolist(),
                           sources=data['sources'], bboxes=data['bboxes'].tolist(),
                           image_shape=list(data['image'].shape),
                           region_name_set=sorted(history[-1]['region_name_set'])))
        # what run() does with the answer (app.py:320-325)
        cleaned = answer.replace('Assistant: ', '').replace('Assistant:', '')
        history[-1]['sources']['conversations'].append({'from': 'gpt', 'value': cleaned})
    checks = []
    for text, n_boxes, n_hist in check_cases():
        bot, history = new_bot(), []
        with contextlib.redirect_stdout(io.StringIO()):
            if n_hist:
                q, b, _ = conversation_script()[0]
                bot.init_inputs({'image': pil, 'boxes': b}, q, history)
            image = {'image': pil, 'boxes': [[1, 2, 30, 40]] * n_boxes}
            err, out = bot.check_input(text, image, copy.deepcopy(history))
        checks.append(dict(text=text, n_boxes=n_boxes, n_hist=n_hist, ok=err is None, text_after=out))
    none_case = new_bot().check_input("hi", None, [])
    return dict(rounds=rounds, checks=checks, no_image_is_error=none_case[0] is not None)


if __name__ == "__main__":
    out = run_ref_bot()
    with open(os.path.join(HERE, "bot_ref.json"), "w") as f:
        json.dump(out, f, indent=1)
    for r in out["rounds"]:
        print(len(r["input_ids"]), "ids;", r["text_after_check"], "| boxes", len(r["bboxes"]))
    print([c["ok"] for c in out["checks"]])
    print("wrote bot_ref.json")
