"""tests/golden/make_golden.py -- regenerates the committed RoIAlign fixtures.

Run in the build container (where /root/reference exists and oracle/_ref has been built
by oracle/build_ref.py):   python tests/golden/make_golden.py

Outputs (committed, small):
  roi_align_known.npz   the three hand-computed known-answer cases of
                        /root/reference/mmcv-1.4.7/tests/test_ops/test_roi_align.py:14-32
                        (inputs, rois, expected output, expected input-gradient under
                        grad_output = ones; pool 2x2, scale 1.0, sr 2, avg, aligned -- :35-38,:89-90)
  roi_align_seeded.npz  seeded cases in the regime GPT4RoI uses (14x14 bins, sr 2,
                        fractional scales 1/1.75..1/14, gpt4roi/models/layers.py:206-214)
                        plus the edge cases; outputs/gradients produced by oracle/_ref =
                        the reference's own CPU implementation compiled unmodified.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import roi_align as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def known():
    d = {}
    ins = [
        ([[[[1., 2.], [3., 4.]]]], [[0., 0., 0., 1., 1.]]),
        ([[[[1., 2.], [3., 4.]], [[4., 3.], [2., 1.]]]], [[0., 0., 0., 1., 1.]]),
        ([[[[1., 2., 5., 6.], [3., 4., 7., 8.], [9., 10., 13., 14.], [11., 12., 15., 16.]]]],
         [[0., 0., 0., 3., 3.]]),
    ]
    outs = [
        ([[[[1.0, 1.25], [1.5, 1.75]]]], [[[[3.0625, 0.4375], [0.4375, 0.0625]]]]),
        ([[[[1.0, 1.25], [1.5, 1.75]], [[4.0, 3.75], [3.5, 3.25]]]],
         [[[[3.0625, 0.4375], [0.4375, 0.0625]], [[3.0625, 0.4375], [0.4375, 0.0625]]]]),
        ([[[[1.9375, 4.75], [7.5625, 10.375]]]],
         [[[[0.47265625, 0.42968750, 0.42968750, 0.04296875],
            [0.42968750, 0.39062500, 0.39062500, 0.03906250],
            [0.42968750, 0.39062500, 0.39062500, 0.03906250],
            [0.04296875, 0.03906250, 0.03906250, 0.00390625]]]]),
    ]
    for i, ((x, r), (o, g)) in enumerate(zip(ins, outs)):
        d[f"x{i}"] = np.array(x, np.float64)
        d[f"rois{i}"] = np.array(r, np.float64)
        d[f"out{i}"] = np.array(o, np.float64)
        d[f"grad{i}"] = np.array(g, np.float64)
    np.savez(os.path.join(HERE, "roi_align_known.npz"), **d)


def boxes(rng, n, batch, img):
    xy = rng.uniform(0, 0.6, (n, 2))
    wh = rng.uniform(0.05, 0.35, (n, 2))
    idx = rng.integers(0, batch, (n, 1)).astype(np.float64)
    return np.concatenate([idx, xy * img, (xy + wh) * img], 1).astype(np.float32)


# name, B, C, H, W, n, out, scale, sr, mode, aligned, img
CASES = [
    ("lvl0", 2, 8, 64, 64, 12, 14, 1 / 1.75, 2, "avg", True, 112),
    ("lvl1", 2, 8, 32, 32, 12, 14, 1 / 3.5, 2, "avg", True, 112),
    ("lvl2", 2, 8, 16, 16, 12, 14, 1 / 7.0, 2, "avg", True, 112),
    ("lvl3", 2, 8, 8, 8, 12, 14, 1 / 14.0, 2, "avg", True, 112),
    ("rect", 1, 5, 13, 29, 7, (3, 5), 0.25, 2, "avg", True, 100),
    ("adaptive", 1, 4, 20, 20, 6, 4, 0.5, 0, "avg", True, 40),
    ("legacy", 1, 4, 20, 20, 6, 7, 0.5, 2, "avg", False, 40),
    ("maxpool", 2, 6, 16, 16, 9, 7, 1 / 7.0, 2, "max", True, 112),
]


def seeded():
    rng = np.random.default_rng(20260925)
    d = {"names": np.array([c[0] for c in CASES] + ["edges"])}
    for name, B, C, H, W, n, out, scale, sr, mode, aligned, img in CASES:
        x = rng.standard_normal((B, C, H, W)).astype(np.float32)
        r = boxes(rng, n, B, img)
        o, ay, ax = O.ref_forward(x, r, out, scale, sr, mode, aligned)
        g = rng.standard_normal(o.shape).astype(np.float32)
        gi = O.ref_backward(g, r, x.shape, out, scale, sr, mode, aligned,
                            ay if mode == "max" else None, ax if mode == "max" else None)
        ph, pw = (out, out) if isinstance(out, int) else out
        d.update({f"{name}.x": x, f"{name}.rois": r, f"{name}.out": o, f"{name}.gout": g,
                  f"{name}.gin": gi, f"{name}.cfg": np.array([ph, pw, sr, mode == "avg", aligned], np.int64),
                  f"{name}.scale": np.array(scale, np.float64)})
        if mode == "max":
            d[f"{name}.argmax_y"] = ay
            d[f"{name}.argmax_x"] = ax
    # edge cases the reference's sampling rules define (cpu/roi_align.cpp:42-86):
    # box hanging off every border, box fully outside, zero-area box, box covering the map.
    x = rng.standard_normal((1, 3, 10, 12)).astype(np.float32)
    r = np.array([[0, -6., -6., 5., 5.], [0, 8., 7., 20., 18.], [0, 30., 30., 40., 44.],
                  [0, 4., 4., 4., 4.], [0, 0., 0., 12., 10.], [0, -1.2, 3., 3.3, 3.5]], np.float32)
    o, _, _ = O.ref_forward(x, r, 14, 1.0, 2, "avg", True)
    g = rng.standard_normal(o.shape).astype(np.float32)
    gi = O.ref_backward(g, r, x.shape, 14, 1.0, 2, "avg", True)
    d.update({"edges.x": x, "edges.rois": r, "edges.out": o, "edges.gout": g, "edges.gin": gi,
              "edges.cfg": np.array([14, 14, 2, 1, 1], np.int64), "edges.scale": np.array(1.0)})
    np.savez_compressed(os.path.join(HERE, "roi_align_seeded.npz"), **d)


def edges_mlvl():
    """Edge boxes for the PRODUCTION kernel (g4r_roi_align_mlvl_nhwc_*: four levels, NHWC, one launch -- VERDICT r03
    weak-2: its edge handling was only ever compared with itself).  The reference's compiled CPU op (oracle/_ref) runs each
    of the four levels of a P = 4 pyramid (maps 32/16/8/4, strides 14/8 .. 14 as layers.py:206-214) on boxes that hang
    off every border, lie fully outside, have zero area, cover the whole image, are sub-pixel thin, and sit on the last
    row / column.  Map values are multiples of 1/16 below 8: exact in fp32, bf16 and fp16, so the 16-bit instantiations
    are checked against the same file with only their OUTPUT rounding as tolerance."""
    rng = np.random.default_rng(77)
    P, C, B = 4, 16, 2
    img = 14.0 * P
    sizes = [8 * P, 4 * P, 2 * P, P]
    strides = [14 / 8, 14 / 4, 14 / 2, 14.0]
    r = np.array([[0, -20., -20., 30., 40.], [1, 0., 0., img, img], [0, 25., 25., 25., 25.], [1, 70., 70., 90., 95.],
                  [0, img - 3, img - 3, img + 30, img + 30], [1, -40., -40., -10., -5.], [0, 10.3, 20.7, 17.1, 25.2],
                  [1, 5., 30., 50., 30.4], [0, img - 1, 0., img, img], [1, 0., img - 0.5, img, img],
                  [0, -0.4, -0.4, 0.4, 0.4], [1, 12.25, 3.5, 40.75, 52.0]], np.float32)
    d = {"rois": r, "strides": np.array(strides, np.float64), "cfg": np.array([14, 14, 2, 1, 1], np.int64)}
    for l, (sz, st) in enumerate(zip(sizes, strides)):
        x = np.clip(np.round(rng.standard_normal((B, C, sz, sz)) * 16) / 16, -7.9375, 7.9375).astype(np.float32)
        o, _, _ = O.ref_forward(x, r, 14, np.float32(1.0 / st), 2, "avg", True)
        d[f"x{l}"], d[f"out{l}"] = x, o
    np.savez_compressed(os.path.join(HERE, "roi_align_edges_mlvl.npz"), **d)


if __name__ == "__main__":
    known()
    seeded()
    edges_mlvl()
    for f in ("roi_align_known.npz", "roi_align_seeded.npz", "roi_align_edges_mlvl.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
