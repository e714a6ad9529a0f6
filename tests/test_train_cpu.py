"""CPU: host logic of the training rows -- the learning-rate schedule against HF's own scheduler, the tile /
wave-split planners of the GEMM front, argument validation of the training C ABI without a GPU, and the oracle's
differentiable RoIAlign node (forward and backward are the pinned C oracle)."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gpt4roi_amd", "lib", "libgpt4roi_hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        from gpt4roi_amd import build
        build.build()
    l = ctypes.CDLL(LIB)
    l.g4r_last_error.restype = ctypes.c_char_p
    return l


def test_cosine_schedule_matches_hf_trainer():
    # train_stage1.sh: --lr_scheduler_type cosine --warmup_ratio 0.003 --learning_rate 2e-5
    from transformers import get_cosine_schedule_with_warmup
    from gpt4roi_amd.train import cosine_lr
    total, base, ratio = 5000, 2e-5, 0.003
    warm = math.ceil(total * ratio)                       # HF TrainingArguments.get_warmup_steps
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=base)
    sch = get_cosine_schedule_with_warmup(opt, warm, total)
    for step in range(total):
        if step in (0, 1, warm - 1, warm, warm + 1, 777, 2500, 4998, 4999):
            assert abs(sch.get_last_lr()[0] - cosine_lr(step, total, base, ratio)) < 1e-12, step
        opt.step()
        sch.step()


def test_tile_planners():
    from gpt4roi_amd import kernels as K
    # the ring ping-pong tile only where whole waves of 256x256 tiles come out; never for skinny or short-K problems
    assert K.pick_tile(767, 12288, 4096) == K.BIG_TILE and K.pick_tile(4096, 4096, 4096) == K.BIG_TILE   # the 256-row production tile
    assert all(K.pick_tile(*shp) not in (24, 34) for shp in [(767, 22016, 4096), (577, 4096, 1024), (1, 4096, 4096)])
    assert K.wave_split(767, 22016, 4096) == 85 * 256      # 3 x 85 = 255 tiles = one wave; tail of 256 columns
    assert K.wave_split(767, 32006, 4096) == 85 * 256
    assert K.wave_split(767, 12288, 4096) is None          # fewer than one wave: no main part
    assert K.wave_split(4096, 4096, 4096) is None          # exactly one wave: nothing to split
    for (M, N, Kd) in [(767, 22016, 4096), (1534, 22016, 4096), (700, 33280, 2048), (3000, 9000, 8192)]:
        n = K.wave_split(M, N, Kd)
        if n is not None:
            assert 0 < n < N and n % 256 == 0 and (-(-M // 256) * (n // 256)) % 256 <= -(-M // 256) * 256 - 1
            assert (-(-M // 256) * (n // 256)) <= (-(-M // 256) * -(-N // 256))
    tile, splits = K.pick_conv_tile(192 * 192, 1024, 9216)
    assert (tile, splits) == (K.BIG_TILE, 1)
    tile, splits = K.pick_conv_tile(24 * 24, 1024, 9216)
    assert tile == 4 and splits > 1


def test_training_abi_validates_arguments_without_a_gpu(lib):
    f = ctypes.c_float
    assert lib.g4r_flash_attn_bwd_bf16(*([None] * 10), 1, 2, 8, 8, 96, *([ctypes.c_long(128)] * 16), f(1.0), 1, None) == 1
    assert b"head_dim" in lib.g4r_last_error()
    assert lib.g4r_rmsnorm_bwd_bf16(None, None, None, None, None, None, 4, 100, ctypes.c_long(104), ctypes.c_long(104),
                                    ctypes.c_long(0), ctypes.c_long(104), f(1e-6), None) == 1     # cols % 8
    assert lib.g4r_rmsnorm_bwd_bf16(None, None, None, None, None, None, 0, 128, ctypes.c_long(128), ctypes.c_long(128),
                                    ctypes.c_long(0), ctypes.c_long(128), f(1e-6), None) == 0     # zero rows: no-op
    assert lib.g4r_cross_entropy_f32(None, None, None, None, None, 3, 10, ctypes.c_long(10), ctypes.c_long(0), 8, None) == 1
    assert lib.g4r_adamw_f32(None, None, 0, None, None, None, ctypes.c_long(16), f(1e-3), f(0.9), f(0.999), f(1e-8), f(0.0),
                             0, f(1.0), None) == 1
    assert b"step" in lib.g4r_last_error()
    assert lib.g4r_nhwc_to_cm_padded_bf16(None, None, 1, 4, 4, 8, 5, ctypes.c_long(30), ctypes.c_long(8), ctypes.c_long(64),
                                          1, None) == 1                                          # Wp < W + 2
    assert lib.g4r_gather_rows_bf16(None, None, None, 0, 64, ctypes.c_long(64), ctypes.c_long(64), None) == 0


def test_oracle_roi_align_autograd_node():
    """oracle/spi_oracle._RoIAlignOracleFn: backward is the C oracle's backward, which is the transpose of its forward."""
    from oracle import spi_oracle as S
    layer = S.RoIAlignOracle(7, spatial_scale=0.5, sampling_ratio=2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 12, 10, generator=g, requires_grad=True)
    rois = torch.tensor([[0, 1.5, 2.0, 14.0, 17.5], [1, 0.0, 0.0, 19.0, 23.0], [0, 6.2, 3.3, 9.9, 8.1]])
    y = layer(x, rois)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    # <A x, w> = <x, A^T w> for the linear map A = roi_align(., rois)
    x2 = torch.randn(x.shape, generator=g)
    y2 = layer(x2, rois)
    lhs = float((y2 * w).sum())
    rhs = float((x2 * x.grad).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    assert np.isfinite(x.grad.numpy()).all() and float(x.grad.abs().sum()) > 0


def test_trainer_refuses_cpu():
    # no CPU fallback anywhere in the product path: kernels.* raise on CPU tensors
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd._lib import HipKernelError
    with pytest.raises(HipKernelError):
        K.rmsnorm_bwd(torch.zeros(2, 64, dtype=torch.bfloat16), torch.ones(64), torch.zeros(2, 64, dtype=torch.bfloat16))
    with pytest.raises(HipKernelError):
        K.cross_entropy(torch.zeros(2, 8), torch.zeros(2, dtype=torch.int64), torch.zeros(1))


def test_collator_contract():
    # gpt4roi/datasets/data_modules.py:22-56
    from types import SimpleNamespace
    from gpt4roi_amd.data import IGNORE_INDEX, DataCollatorForDetDataset, expand_image_tokens, to_device
    coll = DataCollatorForDetDataset(pad_token_id=0)
    inst = [dict(input_ids=torch.tensor([1, 5, 6, 7]), labels=torch.tensor([-100, -100, 6, 7]), image=torch.zeros(3, 28, 28),
                 img_metas=dict(i=0), bboxes=torch.tensor([[0.1, 0.2, 0.5, 0.6]])),
            dict(input_ids=torch.tensor([1, 9]), labels=torch.tensor([-100, 9]), image=torch.ones(3, 28, 28),
                 img_metas=dict(i=1), bboxes=torch.zeros(0, 4))]
    b = coll(inst)
    assert b['input_ids'].tolist() == [[1, 5, 6, 7], [1, 9, 0, 0]]
    assert b['labels'].tolist() == [[-100, -100, 6, 7], [-100, 9, IGNORE_INDEX, IGNORE_INDEX]]
    assert b['attention_mask'].tolist() == [[True] * 4, [True, True, False, False]]
    assert b['images'].shape == (2, 3, 28, 28) and len(b['bboxes']) == 2 and b['img_metas'][1]['i'] == 1
    inst[1]['image'] = torch.ones(3, 14, 14)                         # ragged images stay a list
    assert isinstance(coll(inst)['images'], list)
    ids = SimpleNamespace(im_patch_token=100, im_start_token=103, im_end_token=104)
    assert expand_image_tokens(torch.tensor([1, 50, 2]), 50, ids, 3).tolist() == [1, 103, 100, 100, 100, 104, 2]
    dev = to_device(b, "cpu")
    assert dev['bboxes'].counts == [1, 0] and dev['bboxes'].n == 1 and dev['bboxes'].offsets.tolist() == [0, 1, 1]
    assert torch.allclose(dev['bboxes'].rois5, torch.tensor([[0.0, 2.8, 5.6, 14.0, 16.8]]))


def test_collator_matches_reference_code(golden_dir):
    """gpt4roi_amd.data.DataCollatorForDetDataset against the batch the reference's own collator builds from the same
    seeded instances (tests/golden/make_collator_golden.py imports data_modules.py:22-56 unmodified)."""
    import importlib.util
    from gpt4roi_amd.data import DataCollatorForDetDataset
    spec = importlib.util.spec_from_file_location("make_collator_golden", os.path.join(golden_dir, "make_collator_golden.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    z = np.load(os.path.join(golden_dir, "collator_ref.npz"))
    b = DataCollatorForDetDataset(pad_token_id=0)(G.instances())
    assert sorted(b.keys()) == z["keys"].tolist()
    for k in ("input_ids", "labels", "attention_mask", "images"):
        assert np.array_equal(b[k].numpy(), z[k]), k
    assert [x.shape[0] for x in b["bboxes"]] == z["n_boxes"].tolist()
    assert [m["idx"] for m in b["img_metas"]] == z["metas"].tolist()
    ragged = G.instances()
    ragged[1]["image"] = torch.zeros(3, 4, 4)
    assert isinstance(DataCollatorForDetDataset(0)(ragged)["images"], list) == bool(z["ragged_images_is_list"])


def test_gradient_slots_save_the_copy_into_the_exchange_bucket():
    """GradBucketReducer.slot(param) hands the producer the bucket view itself: a gradient written there is reported with
    no copy (ready() recognises its own storage), any other tensor is copied as before; finish() returns the same views."""
    from gpt4roi_amd.grad_reduce import GradBucketReducer
    p1, p2 = torch.nn.Parameter(torch.zeros(4, 6)), torch.nn.Parameter(torch.zeros(10))
    red = GradBucketReducer([p1, p2], bucket_bytes=1 << 20, comm_dtype=torch.float32)
    red.reset()
    s1 = red.slot(p1)
    assert s1.shape == (4, 6) and s1.dtype == torch.float32
    s1.copy_(torch.arange(24.).view(4, 6))               # the "kernel" writes its result into the bucket
    calls = []
    orig = torch.Tensor.copy_

    def spy(self, *a, **k):
        calls.append(self.data_ptr())
        return orig(self, *a, **k)
    torch.Tensor.copy_ = spy
    try:
        red.ready(p1, s1)                                 # produced in place: no copy
        assert calls == []
        g2 = torch.full((10,), 2.0)
        red.ready(p2, g2)                                 # a foreign tensor: copied
        assert calls == [red.slot(p2).data_ptr()]
    finally:
        torch.Tensor.copy_ = orig
    out = red.finish()
    assert torch.equal(out[id(p1)], torch.arange(24.).view(4, 6)) and torch.equal(out[id(p2)], torch.full((10,), 2.0))
    assert out[id(p1)].data_ptr() == s1.data_ptr()
