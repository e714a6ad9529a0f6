"""CPU: pins the plain-C oracle (oracle/roi_align_oracle.c) against
  (1) the reference's hand-computed known-answer vectors
      (mmcv-1.4.7/tests/test_ops/test_roi_align.py:14-32, atol 1e-3 there; exact here), and
  (2) fixtures produced by the reference's own CPU code compiled unmodified (oracle/_ref),
      committed under tests/golden/ by tests/golden/make_golden.py.
"""
import os

import numpy as np
import pytest

from oracle import roi_align as O


def _cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "roi_align_seeded.npz"))
    return z, [str(n) for n in z["names"]]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_known_answer_vectors(golden_dir, dtype):
    z = np.load(os.path.join(golden_dir, "roi_align_known.npz"))
    for i in range(3):
        x, r = z[f"x{i}"].astype(dtype), z[f"rois{i}"].astype(dtype)
        out, _, _ = O.forward(x, r, (2, 2), 1.0, 2, "avg", True)
        np.testing.assert_allclose(out, z[f"out{i}"], rtol=0, atol=1e-6)
        gin = O.backward(np.ones_like(out), r, x.shape, (2, 2), 1.0, 2, "avg", True)
        np.testing.assert_allclose(gin, z[f"grad{i}"], rtol=0, atol=1e-6)


def test_seeded_fixtures_bit_exact(golden_dir):
    z, names = _cases(golden_dir)
    for name in names:
        ph, pw, sr, avg, aligned = [int(v) for v in z[f"{name}.cfg"]]
        mode = "avg" if avg else "max"
        scale = float(z[f"{name}.scale"])
        x, r = z[f"{name}.x"], z[f"{name}.rois"]
        out, ay, ax = O.forward(x, r, (ph, pw), scale, sr, mode, bool(aligned))
        assert np.array_equal(out, z[f"{name}.out"]), name
        if mode == "max":
            assert np.array_equal(ay, z[f"{name}.argmax_y"]) and np.array_equal(ax, z[f"{name}.argmax_x"])
        gin = O.backward(z[f"{name}.gout"], r, x.shape, (ph, pw), scale, sr, mode, bool(aligned),
                         ay if mode == "max" else None, ax if mode == "max" else None)
        assert np.array_equal(gin, z[f"{name}.gin"]), name


def test_gradcheck_fp64():
    """Mirror of the reference's gradcheck (test_roi_align.py:41-64): numeric vs analytic
    input-gradient in fp64, eps 1e-5 / atol 1e-5, on its third known-answer input."""
    x = np.array([[[[1., 2., 5., 6.], [3., 4., 7., 8.], [9., 10., 13., 14.], [11., 12., 15., 16.]]]])
    r = np.array([[0., 0., 0., 3., 3.]])
    w = np.random.default_rng(1).standard_normal((1, 1, 2, 2))
    ana = O.backward(w, r, x.shape, 2, 1.0, 2)
    num = np.zeros_like(x)
    eps = 1e-5
    for idx in np.ndindex(*x.shape):
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        num[idx] = ((O.forward(xp, r, 2, 1.0, 2)[0] - O.forward(xm, r, 2, 1.0, 2)[0]) * w).sum() / (2 * eps)
    np.testing.assert_allclose(ana, num, atol=1e-5)


def test_negative_roi_is_an_error():
    # cpu/roi_align.cpp:137-139: aligned RoIs with negative extent raise.
    x = np.zeros((1, 1, 4, 4), np.float32)
    with pytest.raises(O.OracleError):
        O.forward(x, np.array([[0, 3., 3., 1., 1.]], np.float32), 2, 1.0, 2)


def test_empty_rois():
    x = np.ones((1, 2, 4, 4), np.float32)
    out, _, _ = O.forward(x, np.zeros((0, 5), np.float32), 14, 1.0, 2)
    assert out.shape == (0, 2, 14, 14)


def test_ref_build_agrees_when_present():
    if O.load_ref() is None:
        pytest.skip("oracle/_ref not built in this checkout")
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 4, 24, 24)).astype(np.float32)
    r = np.array([[0, 3.3, 4.1, 60.2, 70.9], [1, 10., 10., 50., 30.], [1, 0., 0., 96., 96.]], np.float32)
    a = O.forward(x, r, 14, 0.25, 2)[0]
    b = O.ref_forward(x, r, 14, 0.25, 2)[0]
    assert np.array_equal(a, b)


def test_random_configurations_against_the_compiled_reference_and_adjoint():
    """Hypothesis sweep over shapes / scales / sampling ratios / aligned / pool modes: the C restatement equals the
    reference's own compiled CPU op bit for bit (when oracle/_ref is built) and its backward is the exact transpose of
    its forward in fp64 (avg mode): <f(x), w> = <x, f^T(w)>."""
    from hypothesis import given, settings, strategies as st
    ref = O.load_ref()

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 2), st.integers(1, 5), st.integers(3, 17), st.integers(3, 17), st.integers(1, 4),
           st.integers(1, 5), st.integers(1, 5), st.sampled_from([0.25, 0.5, 1.0 / 7.0, 1.0]), st.integers(0, 3),
           st.booleans(), st.sampled_from(["avg", "max"]), st.integers(0, 2 ** 31 - 1))
    def check(B, C, H, W, n, ph, pw, scale, sr, aligned, mode, seed):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((B, C, H, W)).astype(np.float32)
        lo = rng.uniform(-2, 0.7 * max(H, W) / scale, (n, 2))
        rois = np.concatenate([rng.integers(0, B, (n, 1)).astype(np.float64), lo,
                               lo + rng.uniform(0.0 if not aligned else 0.0, 0.6 * max(H, W) / scale, (n, 2))], 1).astype(np.float32)
        out, amy, amx = O.forward(x, rois, (ph, pw), np.float32(scale), sr, mode, aligned)
        assert out.shape == (n, C, ph, pw) and np.isfinite(out).all()
        if ref is not None:
            r_out = O.ref_forward(x, rois, (ph, pw), np.float32(scale), sr, mode, aligned)[0]
            assert np.array_equal(out, r_out)
        if mode == "avg":
            x64, r64 = x.astype(np.float64), rois.astype(np.float64)
            w = rng.standard_normal(out.shape)
            f = O.forward(x64, r64, (ph, pw), scale, sr, mode, aligned)[0]
            gt = O.backward(w, r64, x64.shape, (ph, pw), scale, sr, mode, aligned)
            lhs, rhs = float((f * w).sum()), float((x64 * gt).sum())
            assert abs(lhs - rhs) <= 1e-9 * (1.0 + abs(lhs))

    check()
