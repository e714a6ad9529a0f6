"""CPU: the tile / split dispatch of the production shapes (pure host logic in gpt4roi_amd/kernels.py).  Every decision below
was taken from an A/B measurement on MI355X (profiles/r02_gemm_tiles.md, r02_gemm_small_m.txt, r02_conv_k_order.jsonl,
r05_w4k64_epilogue.txt: tile 34 replaces the ring ping-pong tile 24 on the 256 x 256 shapes); this test pins them so that a heuristic edit for one shape cannot silently move another."""
import pytest

from gpt4roi_amd import kernels as K


@pytest.mark.parametrize("M,N,Kd,tile,main", [
    (767, 12288, 4096, 34, None),      # LLaMA fused qkv: 144 one-wave-per-SIMD tiles (round 5: 82.6 us vs 95.5 on 192 x 256 ring tiles)
    (767, 22016, 4096, 0, 21760),      # gate|up: whole-wave column split, 255 tiles on the ring kernel + 256 columns
    (767, 32006, 4096, 0, 21760),      # lm_head
    (767, 4096, 4096, 7, None),        # o_proj: 128x128 x 8 waves, ring of 4
    (767, 256, 4096, 14, None),        # the gate|up remainder: 64x64 ring-4 (+ 2 K slices in gemm())
    (577, 3072, 1024, 14, None), (577, 1024, 1024, 14, None), (577, 4096, 1024, 13, None), (577, 1024, 4096, 14, None),  # ViT, batch 1
    (4616, 3072, 1024, 34, None), (4616, 1024, 4096, 0, None),                                                           # ViT, batch 8
    (12272, 12288, 4096, 34, None), (12272, 4096, 4096, 34, None), (12272, 22016, 4096, 34, 21760), (12272, 4096, 11008, 34, None),   # 16 merged requests (bench default): the one-wave-per-SIMD 256 x 256 tile (round 5) in whole waves
    (9232, 3072, 1024, 34, None), (9232, 4096, 1024, 34, None), (9232, 1024, 4096, 34, None),                                          # ViT, batch 16
    (8, 12288, 4096, 14, None), (8, 4096, 4096, 14, None), (8, 22016, 4096, 13, None), (16, 4096, 11008, 14, None),      # batched decode
])
def test_gemm_tile_dispatch(M, N, Kd, tile, main):
    assert K.pick_tile(M, N, Kd) == tile
    assert K.wave_split(M, N, Kd) == main


@pytest.mark.parametrize("M,N,Kd,want", [
    (5592, 1280, 11008, (34, 2)), (5592, 1280, 22016, (34, 2)), (5592, 1280, 12288, (34, 2)),   # training-batch column remainders
    (5592, 1280, 4096, None),          # K = 4096: the two-stage tile is as fast
    (767, 4096, 11008, None),          # down_proj at batch 1: long_k_plan's shape (48 tiles), not this one
    (767, 10246, 4096, None),          # lm_head remainder
    (5592, 2816, 11008, None),         # 242 tiles: a full wave on its own
])
def test_partial_wave_plan(M, N, Kd, want):
    assert K.partial_wave_plan(M, N, Kd) == want


@pytest.mark.parametrize("M,Cout,Kd,want", [
    (36864, 1024, 9216, (34, 1)),      # a 192^2 level on its own (the fuse rounds use conv3x3_mlvl: one launch for all levels)
    (2304, 1024, 9216, (4, 1)),
    (6272, 1024, 36864, (34, 2)),      # pconv, 32 RoIs: 100 tiles x 2 K slices
    (15288, 1024, 36864, (34, 1)),     # pconv, training batch (78 RoIs)
])
def test_conv_tile_dispatch(M, Cout, Kd, want):
    assert K.pick_conv_tile(M, Cout, Kd) == want


def test_mlvl_geometry_of_the_336_pyramid_is_three_full_waves():
    rows = sum(h * h for h in (192, 96, 48, 24))
    assert rows == 48960 and -(-rows // 256) * (1024 // 256) == 768 == 3 * 256
    for start in (192 * 192, 192 * 192 + 96 * 96, 192 * 192 + 96 * 96 + 48 * 48):
        assert start % 256 == 0                                   # level boundaries fall on tile boundaries at P = 24


def test_mlvl_maps_layout_is_level_major_with_contiguous_per_level_views():
    """kernels.MlvlMaps: the buffer conv3x3_mlvl reads / writes -- [level][b][y][x][C] in one allocation, each level a
    contiguous NHWC view at the row offset the kernel's level table assumes."""
    import torch
    m = K.MlvlMaps(2, [(6, 5), (3, 3), (2, 1)], 8, "cpu", dtype=torch.float32)
    rows = [2 * 6 * 5, 2 * 3 * 3, 2 * 2 * 1]
    assert m.flat.shape == (sum(rows), 8) and [tuple(v.shape) for v in m.levels] == [(2, 6, 5, 8), (2, 3, 3, 8), (2, 2, 1, 8)]
    off = 0
    for v, n in zip(m.levels, rows):
        assert v.is_contiguous() and v.data_ptr() == m.flat[off].data_ptr()
        off += n
    m.levels[1][1, 2, 0, 3] = 7.0                                 # level 1, image 1, y 2, x 0 -> flat row 60 + 9 + 6
    assert float(m.flat[60 + 9 + 6, 3]) == 7.0
    assert list(m._hw[0]) == [6, 3, 2] and list(m._hw[1]) == [5, 3, 1]
