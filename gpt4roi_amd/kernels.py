"""Typed Python fronts of the C ABI (include/g4r_kernels.h, include/g4r_roi_align.h).

Thin: every function checks devices/dtypes, passes raw device pointers + the current HIP
stream, and raises on a non-zero status.  Output tensors are allocated by the caller or
here with torch.empty (PyTorch is only the allocator / stream provider).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p

import torch

from . import _lib

P = c_void_p
_SIGS = {
    "g4r_gemm_bf16_nt": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_int, c_int, P],
    "g4r_conv3x3_nhwc_bf16": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_long, c_int,
                              c_int, c_int, c_int, P],
    "g4r_flash_attn_fwd_bf16": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_long,
                                c_long, c_long, c_long, c_long, c_float, c_int, P, P, P],
    "g4r_conv3x3_mlvl_nhwc_bf16": [P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, P],
    "g4r_batch_advance": [P, c_int, P, P, P, P, P, c_int, P],
    "g4r_gemv_rmsnorm_bf16": [P, P, c_float, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "g4r_gemv_batch_bf16": [P, c_int, c_long, P, c_float, P, P, c_long, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "g4r_attn_decode_bf16": [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_long, c_long, c_float, c_int, P, c_int,
                             c_int, c_long, c_long, c_long, c_long, P],
    "g4r_attn_decode_ragged_bf16": [P, P, P, P, P, P, P, P, c_int, c_int, c_long, c_long, c_float, c_int, P, P, c_int,
                                    c_long, c_long, c_long, c_long, P],
    "g4r_batch_advance_ragged": [P, c_int, P, P, P, P, P, c_int, c_int, P],
    "g4r_gemv_attn_merge_bf16": [P, c_int, c_int, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "g4r_flash_attn_bwd_bf16": [P] * 10 + [c_int] * 5 + [c_long] * 16 + [c_float, c_int, P],
    "g4r_rmsnorm_bwd_bf16": [P, P, P, P, P, P, c_int, c_int, c_long, c_long, c_long, c_long, c_float, P],
    "g4r_layernorm_bwd_bf16": [P, P, P, P, P, P, c_int, c_int, c_long, c_long, c_long, c_float, c_int, P],
    "g4r_swiglu_il_bf16": [P, P, c_int, c_int, P],
    "g4r_swiglu_il_bwd_bf16": [P, P, P, c_int, c_int, P],
    "g4r_rope_qkv_bwd_bf16": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_long, c_long, c_long, P],
    "g4r_rope_qkv_bwd_batch_bf16": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long, P],
    "g4r_cross_entropy_f32": [P, P, P, P, P, c_int, c_int, c_long, c_long, c_int, P],
    "g4r_transpose_bf16": [P, P, c_int, c_int, c_long, c_long, c_int, P],
    "g4r_colsum_bf16": [P, P, c_int, c_int, c_long, P],
    "g4r_relu_bwd_bf16": [P, P, P, c_long, P],
    "g4r_gather_rows_bf16": [P, P, P, c_int, c_int, c_long, c_long, P],
    "g4r_scatter_add_rows_f32": [P, P, P, c_int, c_int, c_long, c_long, P],
    "g4r_adamw_f32": [P, P, c_int, P, P, P, c_long, c_float, c_float, c_float, c_float, c_float, c_int, c_float, P],
    "g4r_multi_sumsq": [P, P, P, P, c_int, c_int, P, P, P],
    "g4r_multi_adamw_f32": [P, P, P, P, P, P, P, P, c_int, c_int, P, c_float, c_float, c_float, c_float, c_float, c_float,
                            c_float, c_int, P],
    "g4r_groupnorm_stats_nhwc_bf16": [P, P, P, c_int, c_int, c_int, c_int, c_float, P],
    "g4r_gn_relu_bwd_nhwc_bf16": [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "g4r_fuse_shuffle_bwd_nhwc_bf16": [P, c_int, c_int, P, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P],
    "g4r_fuse_shuffle_bwd_gather_nhwc_bf16": [P, P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int,
                                              c_int, P],
    "g4r_nhwc_pad_bf16": [P, P, c_int, c_int, c_int, c_int, c_long, P],
    "g4r_conv3x3_wgrad_nhwc_bf16": [P, P, c_int, P, P, c_int, c_int, c_int, c_int, P, P, c_int, P],
    "g4r_conv3x3_weight_layout_bf16": [P, P, c_int, c_int, c_long, c_long, c_int, P],
    "g4r_gemm_tn_bf16": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P],
    "g4r_conv3x3_wgrad_nhwc_slices": [c_int, P, P, c_int, c_int, c_int, c_int],
    "g4r_nhwc_to_cm_padded_bf16": [P, P, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_int, P],
    "g4r_roi_align_mlvl_nhwc_bwd_bf16": [P, c_long, c_long, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_int, P],
    "g4r_roi_align_mlvl_nhwc_bwd_gather_bf16": [P, c_long, c_long, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int,
                                                c_int, c_int, c_int, P],
    "g4r_layernorm_bf16": [P, P, P, P, c_int, c_int, c_long, c_long, c_float, c_int, P],
    "g4r_rmsnorm_bf16": [P, P, P, c_int, c_int, c_long, c_long, c_float, P],
    "g4r_groupnorm_affine_nhwc_bf16": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P],
    "g4r_upsample_coord_nhwc_bf16": [P, P, c_int, c_int, c_int, c_long, c_int, c_int, c_int, c_int, c_int, P],
    "g4r_fuse_shuffle_nhwc_bf16": [P, P, c_int, c_int, P, P, c_int, c_int, P, P, c_int, c_int, P, c_int,
                                   c_int, P],
    "g4r_im2col_patch14_f32": [P, P, c_int, c_int, c_int, P],
    "g4r_vit_assemble_bf16": [P, P, P, P, c_int, c_int, c_int, P],
    "g4r_rope_qkv_bf16": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P],
    "g4r_greedy_advance_f32": [P, c_int, P, P, P, P, c_int, P],
    "g4r_sample_advance_f32": [P, c_int, c_float, c_int, c_float, P, P, P, P, P, c_int, P, P],
    "g4r_swiglu_bf16": [P, P, c_int, c_int, P],
    "g4r_splice_embed_bf16": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_long, c_long, c_long,
                              c_long, c_int, P],
    "g4r_argmax_rows_f32": [P, c_long, c_int, c_int, P, P],
    "g4r_add_rows_bf16": [P, P, P, c_long, c_int, c_long, P],
    "g4r_cast_f32_to_bf16": [P, P, c_long, P],
    "g4r_image_preprocess_u8_f32": [P, c_int, c_int, c_long, c_int, P, c_int, c_int, c_float, c_float, c_float, c_float,
                                    c_float, c_float, P],
    "g4r_roi_align_mlvl_nhwc_bf16": [P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, P],
    "g4r_roi_align_mlvl_nhwc_f32": [P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_int, P],
    "g4r_groupnorm_affine_mlvl_nhwc_bf16": [P, P, P, P, P, c_int, P, c_int, c_int, c_int, c_float, P],
    "g4r_gemm_bf16_nt_partials": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "g4r_gemm_qkv_rope_bf16": [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_long, c_long, P, P, c_int,
                               c_int, P],
    "g4r_rmsnorm_splitk_bf16": [P, c_int, P, c_long, P, c_long, P, P, c_long, c_int, c_int, c_float, P],
    "g4r_layernorm_splitk_bf16": [P, c_int, P, P, c_long, P, c_long, P, P, P, c_long, c_int, c_int, c_float, P],
    "g4r_fuse_shuffle_mlvl_nhwc_bf16": [P, P, P, P, P, P, P, c_int, c_int, c_int, P],
}
_bound = {}

ACT = {None: 0, "none": 0, "relu": 1, "quick_gelu": 2, "silu": 3, "swiglu": 4}


def _fn(name, sym=None):
    sym = sym or name
    f = _bound.get(sym)
    if f is None:
        f = getattr(_lib.lib(), sym)
        f.argtypes = _SIGS[name]           # the fp16 instantiation has the signature of the bf16 entry point
        f.restype = c_int
        _bound[sym] = f
    return f


class _Profiler:
    """Optional per-launch HIP-event timing (bench.py's roofline leg).  Events are recorded on the
    stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.enabled = False
        self.detail = False
        self.records = []

    def start(self, detail=False):
        """detail: GEMM / conv tags also carry the problem shape (tools/train_step_bench.py --by-shape)."""
        self.records, self.enabled, self.detail = [], True, bool(detail)

    def stop(self):
        """-> {tag: dict(calls, ms, flops, bytes)}"""
        self.enabled = False
        torch.cuda.synchronize()
        agg = {}
        for tag, flops, nbytes, e0, e1 in self.records:
            a = agg.setdefault(tag, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            a["calls"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["bytes"] += nbytes
        self.records = []
        return agg


PROFILER = _Profiler()
TILE_NAMES = {0: "128x128", 1: "256x128", 2: "128x128reg", 4: "64x128", 5: "128x128s3", 6: "128x128s4",
              7: "128x128w8s4", 8: "256x128s3", 9: "256x256", 10: "128x128w8", 11: "128x64", 12: "64x128s4", 13: "64x128s3", 14: "64x64s4",
              15: "128x64s4", 22: "256x256pp", 24: "256x256pp32", 26: "256x256w4", 27: "128x384pp32", 28: "192x256pp32",
              34: "256x256w4k64", 36: "256x256w4k64"}
# the production 256 x 256 tile: 34 = one wave per SIMD, K tiles of 64, straight-line epilogues (round 5: 1.38-1.41 PF/s on the
# merged LLaMA shapes, +14 % over the ring ping-pong tile, profiles/r05_w4k64_epilogue.txt); G4R_BIG_TILE=24 = the ring
# ping-pong tile of rounds 2-4 (A/B runs)
BIG_TILE = int(os.environ.get("G4R_BIG_TILE", "34"))
if BIG_TILE != 34:
    _lib.lib().g4r_gemm_debug_mode(60)          # the one-launch-per-round conv follows (debug mode 60 = its ring ping-pong arm)


def _sym(name, dt):
    """Entry point of `name` for the 16-bit storage type dt: include/g4r_f16_names.h (bf16 -> f16 in the name, or an _f16
    suffix where the name carries no dtype) for torch.float16, the name itself otherwise."""
    if dt is torch.float16:
        return name.replace("bf16", "f16") if "bf16" in name else name + "_f16"
    return name


def _launch(name, args, tag=None, flops=0.0, nbytes=0.0, dt=None):
    f = _fn(name, _sym(name, dt))
    if PROFILER.enabled:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = f(*args)
        e1.record()
        PROFILER.records.append((tag or name, flops, nbytes, e0, e1))
    else:
        rc = f(*args)
    _lib.check(rc, name)


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _bf16(*ts):
    for t in ts:
        if t is not None:
            _lib.require_gpu(t)
            if t.dtype != torch.bfloat16:
                raise TypeError(f"expected bfloat16, got {t.dtype}")


H16 = (torch.bfloat16, torch.float16)


def _h16(*ts):
    """The inference kernels exist for both 16-bit storage types (bf16 = the reference's training dtype, fp16 = its serving
    dtype, app.py:74-98).  Checks that every tensor is on the GPU and that all share ONE of the two; returns it (None when
    every argument is None).  A torch.dtype among the arguments (the `dt` of an earlier call) joins the check."""
    dt = None
    for t in ts:
        if t is None:
            continue
        if isinstance(t, torch.dtype):
            d = t
        else:
            _lib.require_gpu(t)
            d = t.dtype
        if d not in H16:
            raise TypeError(f"expected bfloat16 or float16, got {d}")
        if dt is not None and d != dt:
            raise TypeError(f"mixed 16-bit storage types in one call: {dt} and {d}")
        dt = d
    return dt


def _f32(*ts):
    for t in ts:
        if t is not None:
            _lib.require_gpu(t)
            if t.dtype != torch.float32:
                raise TypeError(f"expected float32, got {t.dtype}")


_zeros = {}


def zeros_line(device):
    z = _zeros.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.bfloat16, device=device)
        _zeros[device] = z
    return z


def pick_tile(M, N, K=0):
    """Tile heuristic for a 256-CU part, from tools/microbench.py on MI355X: 128x128 two-stage when
    the grid fills the chip; 64x128 (more workgroups) when it would not; for long-K / narrow-N shapes
    (LLaMA down_proj 767x4096x11008) the 8-wave ring variant."""
    # 256x256 ring ping-pong (tile 24, ~1.0 PF/s on full waves): when its tiles fill >= 55 % of one wave of the 256
    # CUs (LLaMA fused qkv 767x12288: 144 tiles, 92 us vs 108 us for 128x128) or >= 85 % of several, and the M
    # padding costs < 10 %.  768x22016 (258 tiles = one wave + 2) and the N = 4096 projections (48 tiles) stay on
    # the 128-wide tiles.
    if M >= 1024 and K >= 2048 and K % 64 == 0:
        # Several requests in one launch sequence (round 4: bench --batch 4 -> M = 3068).  The ring ping-pong tiles run one
        # workgroup per CU, so a launch costs whole waves of 256 workgroups; a 192 x 256 wave takes 0.92 of a 256 x 256 wave
        # (0.75 of the MFMAs, 0.875 of the operand bytes: the K loop is bound by the CU's load path, profiles/
        # r04_gemm_ablation.txt).  M = 3068 (tools/gemm_bench.cpp, profiles/r04_gemm_batch4_tiles.txt): q|k|v 768 tiles of
        # 192 x 256 = 3 waves 283 us vs 303 (256 x 256: 2.25 -> 3 waves) vs 350 (128 x 128); o_proj 256 tiles = ONE wave 99 us vs
        # 105 / 132; down_proj 247 vs 252 / 345; gate|up 256 x 256 (4.03 waves, see wave_split) 521 vs 560; lm_head 868 vs 1088.
        t24 = -(-M // 256) * -(-N // 256)
        t28 = -(-M // 192) * -(-N // 256)
        if max(t24, t28) >= 192:                    # narrow outputs (ViT fc2 at batch 8: 100 tiles) stay on the smaller tiles below
            # (a 192 x 256 wave takes 0.92 of a ring ping-pong 256 x 256 wave, 1.05 of a one-wave-per-SIMD one)
            return 28 if -(-t28 // 256) * (1.05 if BIG_TILE == 34 else 0.92) < -(-t24 // 256) else BIG_TILE
    t256 = -(-M // 256) * -(-N // 256)
    if K >= 1024 and K % 64 == 0 and (-(-M // 256) * 256) <= 1.1 * M:     # K = 1024: ViT at batch 8 (4616x3072x1024: 715 vs 552 TF/s)
        eff = t256 / (-(-t256 // 256) * 256)
        # (round 5, profiles/r05_vit_gemm_sweep.txt: the one-wave-per-SIMD tile also wins at 0.77 of its last wave -- ViT fc1 at
        #  batch 16, 9232 x 4096 x 1024: 108 us vs 122-128 on the 128 x 128 tile; at 0.59 -- batch 8 -- they are equal)
        if (t256 <= 256 and eff >= 0.55) or eff >= (0.75 if BIG_TILE == 34 else 0.85):
            # one partial wave: 192-row tiles when they give more of the 256 CUs a (smaller) tile -- LLaMA fused qkv
            # 767x12288x4096: 4 x 48 = 192 workgroups of 0.75 the work, 101.6 vs 114.6 us (profiles/r02_gemm_tiles.md)
            t192 = -(-M // 192) * -(-N // 256)
            # (round 5: the one-wave-per-SIMD tile beats the 192-row ring tile there too -- 82.6 vs 95.5 us, profiles/r05_m767_sweep.txt)
            if BIG_TILE != 34 and t256 <= 256 and eff < 0.75 and t256 < t192 <= 256 and (-(-M // 192) * 192) <= 1.1 * M:
                return 28
            return BIG_TILE
    t128 = -(-M // 128) * -(-N // 128)
    if t128 >= 192:
        # o_proj 767x4096x4096 (192 tiles): 8-wave ring 45.7 us vs 65.9 on the two-stage tile; with more than one tile per CU
        # the two-stage tile wins again (ViT fc2 at batch 8, 4616x1024x4096 = 296 tiles: 71.2 vs 83.6 us)
        return 7 if (K >= 4096 and t128 <= 256) else 0
    # Small M (CLIP ViT at batch 1, M = 577): fewer workgroups than CUs and only 16-64 K tiles, so ONE workgroup's
    # latency is the time -> deep LDS-DMA rings (tools/gemm_bench.cpp on MI355X, profiles/r02_gemm_tiles.md):
    # 577x3072x1024 15.1 us on 64x64 ring-4 vs 22.1 on the two-stage 64x128; 577x1024x1024 12.3 vs 20.5;
    # 577x4096x1024 18.1 on 64x128 ring-3 vs 21.2.
    if K % 64 == 0 and K >= 512 and M > 64:
        t64 = -(-M // 64) * -(-N // 64)
        return 14 if t64 <= 512 else 13
    # 1 < M <= 64: the decode step of a small batch of sequences -- still weight streaming, one 64-row tile in M.  Ring-4
    # 64x64 tiles up to N = 16 k (qkv 8x12288x4096: 25.4 us = 4.0 TB/s vs 58.2 on the two-stage 64x128), ring-3 64x128
    # beyond (gate|up 8x22016x4096: 40.1 us = 4.5 TB/s); `gemm` adds K slices when N is small (profiles/r02_gemm_small_m.txt)
    if K % 64 == 0 and K >= 512 and M > 1:
        return 14 if N <= 16384 else 13
    return 4


def pick_conv_tile(M, Cout, K):
    """Implicit-GEMM conv (tools/conv_tiles.py on MI355X, N-fastest tile order): the 256x256 ping-pong
    kernel wins on the large maps (same box, 192^2: 898 TF/s vs 750 for the two-barrier 256x256 and 702 for
    128x128; 96^2: 692 vs 556 / 588) because the other wave group's MFMAs hide the per-piece halo/bounds
    address work; the 256x128 two-stage kernel serves the K = 4*9*C pconv, and small maps want 64x128 tiles +
    split-K to cover the 256 CUs.  Returns (tile_cfg, splits)."""
    if K >= 18432:
        # the K = 4*9*C pconv (6272 x 1024 x 36864 for 32 RoIs: 100 tiles of 256 x 256): K-slices on the ring ping-pong
        # kernel so that tiles x slices stays within one wave of the 256 CUs.  (The one-wave-per-SIMD kernel wins the dense
        # proxy of this shape, 452 vs 492 us, but loses the real implicit conv -- 841 us: a lone wave per SIMD has nobody to
        # hide the per-piece halo / bounds arithmetic behind.)
        # Round 5: the K-64 one-wave-per-SIMD kernel has no VALU address work left in its loop and wins the real pconv as well
        # (32 RoIs x 2 slices 417 vs 459 us; 78 RoIs 887 vs 976; 512 RoIs 6154 vs 6887: profiles/r05_m767_sweep.txt).
        t256 = -(-M // 256) * -(-Cout // 256)
        return BIG_TILE, max(1, min(5, 256 // t256))
    if M >= 8192:
        return BIG_TILE, 1
    blocks = -(-M // 64) * -(-Cout // 128)
    splits = max(1, min(8, round(384 / blocks), K // 1024))
    return 4, splits


def wave_split(M, N, K):
    """Columns [0, N_main) that fill WHOLE waves of 256x256 tiles on the 256 CUs, when the full problem does not
    (LLaMA gate|up 767x22016: 3 x 86 = 258 tiles = one wave + 2; lm_head 767x32006: 378 tiles).  The main part
    then runs on the ring ping-pong kernel (tile 24) with one tile per CU per wave and the remaining columns as a
    second, small GEMM.  Returns N_main or None."""
    if not (K >= 2048 and K % 64 == 0 and M > 1 and (-(-M // 256) * 256) <= 1.1 * M):
        return None
    mt, nt = -(-M // 256), -(-N // 256)
    t256 = mt * nt
    waves, rem = divmod(t256, 256)
    if waves < 1 or rem == 0:
        return None
    n_main = (waves * 256 // mt) * 256
    if n_main <= 0 or n_main >= N:
        return None
    return n_main


def long_k_plan(M, N, K):
    """(tile_cfg, K slices) for the long-K / few-tiles shape of the LLaMA down_proj (767 x 4096 x 11008: 48 tiles of 256 x 256),
    or None: the 192 x 256 ring ping-pong tile x 4 K-slices = 64 x 4 = 256 workgroups."""
    if K >= 8192 and K % 64 == 0 and (-(-M // 256) * 256) <= 1.1 * M and 32 <= -(-M // 256) * -(-N // 256) <= 64:
        # round 5: 48 tiles of 256 x 256 x 4 K slices on the one-wave-per-SIMD tile 78.3 us vs 83.9 (profiles/r05_m767_sweep.txt)
        return (34, 4) if BIG_TILE == 34 else (28, 4)
    return None


def partial_wave_plan(M, N, K):
    """(tile_cfg, K slices) for a long-K shape that covers a quarter to a half of one wave of 256 x 256 tiles -- the column
    remainder `wave_split` leaves behind in the training batch (5592 x 1280 x {11008, 12288, 22016}: 110 tiles) -- or None.
    K slices on the ring ping-pong kernel until tiles x slices fills the 256 CUs: 153 vs 186 us at K = 11008, 169 vs 215 at
    12288, 300 vs 377 at 22016 against the two-stage 128 x 128 tile (profiles/r03_gemm_partial_wave.txt).  Not for K = 4096
    (equal) and not for fp32 outputs (lm_head remainder: 118 vs 91 us -- the fp32 partials cost more than the wave gains)."""
    if not (K >= 8192 and K % 64 == 0 and (-(-M // 256) * 256) <= 1.1 * M):
        return None
    t256 = -(-M // 256) * -(-N // 256)
    if 64 < t256 <= 128:
        return BIG_TILE, 256 // t256
    return None


def gemm_partials(a, w, splits, tile_cfg):
    """a [M,K] @ w[N,K]^T as fp32 K-slice partials [n_slices, M, N] WITHOUT the reduce launch (the consumer combines them:
    rmsnorm_splitk).  Returns (partials, n_slices)."""
    dt = _h16(a, w)
    M, K = a.shape
    N = w.size(0)
    assert a.stride(1) == 1 and w.stride(1) == 1 and splits >= 2
    ws = torch.empty((splits, M, N), dtype=torch.float32, device=a.device)
    n = c_int(0)
    _launch("g4r_gemm_bf16_nt_partials", (_p(a), _p(w), _p(ws), M, N, K, a.stride(0), w.stride(0), int(splits), int(tile_cfg),
                                          ctypes.byref(n), _stream(a),),
            tag=f"gemm_bf16_nt<{TILE_NAMES.get(tile_cfg, tile_cfg)}>+splitk" + (f" {M}x{N}x{K}/{splits}" if PROFILER.detail else ""),
            flops=2.0 * M * N * K, nbytes=2.0 * (M * K + N * K) + 4.0 * splits * M * N, dt=dt)
    return ws, n.value


def rmsnorm_splitk(partials, n_slices, residual, gamma, eps=1e-6):
    """x = bf16(sum of the first n_slices partials + residual); y = rmsnorm(x; gamma) -> (x, y): the split-K reduce of a
    residual GEMM and the RMSNorm that follows it in one pass (bit-identical to gemm(..., residual=) + rmsnorm())."""
    _f32(partials, gamma)
    dt = _h16(residual)
    _, M, N = partials.shape
    x = torch.empty((M, N), dtype=dt, device=partials.device)
    y = torch.empty_like(x)
    _launch("g4r_rmsnorm_splitk_bf16", (_p(partials), int(n_slices), _p(residual), residual.stride(0) if residual is not None else 0,
                                        _p(x), x.stride(0), _p(gamma), _p(y), y.stride(0), M, N, float(eps), _stream(x),),
            tag="g4r_rmsnorm_bf16", dt=dt)
    return x, y


def small_m_split_plan(M, N, K):
    """(tile_cfg, K slices) when gemm() would run this shape as K slices on the 64x64 ring tile (CLIP fc2 at batch 1:
    577 x 1024 x 4096), or None."""
    if pick_tile(M, N, K) == 14 and K >= 4096 and -(-M // 64) * -(-N // 64) <= 256 and M > 64:
        return 14, (4 if -(-M // 64) * -(-N // 64) <= 64 else 2)
    return None


def decode_split_plan(M, N, K):
    """(tile_cfg, K slices) when gemm() runs a batched-decode projection (a handful of rows) as K slices on the 64x64 ring tile
    followed by a reduce launch -- o_proj / down_proj of LlamaDecoder._decode_step_batch: 8 x 4096 x {4096, 11008} -- or None.
    The step then hands the partials to rmsnorm_splitk (reduce + residual + the next RMSNorm in one launch, same bits)."""
    if 1 < M <= 64 and not gemv_batch_wins(M, N, K) and pick_tile(M, N, K) == 14 and K >= 4096 and K % 64 == 0 \
            and -(-M // 64) * -(-N // 64) <= 64:
        return 14, 4                                     # (the branch of gemm() with the same conditions)
    return None


def layernorm_splitk(partials, n_slices, bias, residual, gamma, beta, eps=1e-5):
    """x = bf16(sum of the first n_slices partials + bias + residual); y = layernorm(x) -> (x, y) in one pass: the arithmetic of
    gemm(..., bias=, residual=) with K slices + layernorm() in the same order (x is bit-identical; y is equal on the tested
    case and within ONE ulp in ~4 of a million outputs on random rows: hipcc fuses the two kernels' sums of squares differently,
    DESIGN.md section 7)."""
    _f32(partials, bias, gamma, beta)
    dt = _h16(residual)
    _, M, N = partials.shape
    x = torch.empty((M, N), dtype=dt, device=partials.device)
    y = torch.empty_like(x)
    _launch("g4r_layernorm_splitk_bf16", (_p(partials), int(n_slices), _p(bias), _p(residual),
                                          residual.stride(0) if residual is not None else 0, _p(x), x.stride(0), _p(gamma),
                                          _p(beta), _p(y), y.stride(0), M, N, float(eps), _stream(x),),
            tag="g4r_layernorm_bf16", dt=dt)
    return x, y


def gemm_qkv_rope(h, wqkv, B, T, heads, head_dim, q_out, k_cache, v_cache, cos, sin, pos0, tile_cfg=None):
    """The fused q|k|v projection of a LLaMA layer with RoPE and the KV-cache append in the GEMM epilogue: h [B*T, K] ->
    q_out [B, T, heads*D] (rotated), k_cache / v_cache [B, maxT, heads*D] rows pos0 .. pos0+T-1 (rotated keys, values).
    Same rounding points as gemm() + rope_qkv().  Returns None when the shape is not one the ring ping-pong tiles serve
    (the caller then runs the two launches)."""
    dt = _h16(h, wqkv, q_out, k_cache, v_cache)
    _f32(cos, sin)
    M, Kd = h.shape
    HD = heads * head_dim
    if tile_cfg is None:
        tile_cfg = pick_tile(M, 3 * HD, Kd)
    if head_dim != 128 or HD % 256 != 0 or tile_cfg not in (24, 28, 34, 36) or M != B * T:
        return None
    assert wqkv.shape == (3 * HD, Kd) and q_out.is_contiguous() and cos.size(1) == 64 and cos.is_contiguous() and sin.is_contiguous()
    assert k_cache.stride(2) == 1 and v_cache.stride() == k_cache.stride() and pos0 + T <= k_cache.size(1)
    _launch("g4r_gemm_qkv_rope_bf16", (_p(h), _p(wqkv), B, T, Kd, h.stride(0), wqkv.stride(0), heads, head_dim, _p(q_out),
                                       _p(k_cache), _p(v_cache), k_cache.stride(1), k_cache.stride(0), _p(cos), _p(sin), int(pos0),
                                       int(tile_cfg), _stream(h),),
            tag=f"gemm_bf16_nt<{TILE_NAMES.get(tile_cfg, tile_cfg)}>", flops=2.0 * M * 3 * HD * Kd,
            nbytes=2.0 * (M * Kd + 3 * HD * Kd + 3 * M * HD), dt=dt)
    return q_out


def gemm(a, w, bias=None, residual=None, act=None, out=None, out_dtype=None, splits=1,
         tile_cfg=None, workspace=None):
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias) + residual.  a may be row-strided (last dim dense)."""
    dt = _h16(a, w, residual)
    _f32(bias)
    assert a.dim() == 2 and w.dim() == 2 and a.size(1) == w.size(1), (a.shape, w.shape)
    assert a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.size(0)
    n_out = N // 2 if act == "swiglu" else N          # swiglu: interleaved (gate, up) weight rows
    if out is None:
        out = torch.empty((M, n_out), dtype=out_dtype or dt, device=a.device)
    assert out.stride(1) == 1 and out.shape == (M, n_out)
    assert out.dtype in (dt, torch.float32), f"gemm: out is {out.dtype}, operands are {dt}"     # (the epilogue writes dt or fp32 bits)
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
    if tile_cfg is None and splits == 1 and workspace is None and gemv_batch_wins(M, N, K) and a.stride(0) % 8 == 0 \
            and a.stride(0) >= K and w.stride(0) % 8 == 0 and (residual is None or residual.data_ptr() != out.data_ptr()) \
            and a.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0:        # (the kernel reads both operands in 16-byte pieces)
        # a handful of rows (the decode step of a batch of sequences): weight streaming with the products on the matrix pipe
        return gemv_batch(a, w, bias=bias, residual=residual, act=act, out=out, variant=GEMV_BATCH_VARIANT)
    thin_tail = False
    if tile_cfg is None and splits == 1 and M >= 1024 and pick_tile(M, N, K) == BIG_TILE and out.dtype != torch.float32:
        # several whole waves of 256 x 256 tiles plus a THIN last one (batch-4 gate|up 3068 x 22016: 1032 tiles = 4 waves + 8
        # tiles): the last columns go to a second launch as K slices instead of costing a fifth wave
        t256 = -(-M // 256) * -(-N // 256)
        thin_tail = t256 > 256 and 0 < t256 % 256 <= 48
    if tile_cfg is None and splits == 1 and (pick_tile(M, N, K) not in (24, 28, 34) or thin_tail):
        n_main = wave_split(M, N, K)
        if n_main is not None:
            o_main = n_main // 2 if act == "swiglu" else n_main
            gemm(a, w[:n_main], bias[:n_main] if bias is not None else None,
                 residual[:, :n_main] if residual is not None else None, act, out[:, :o_main], tile_cfg=BIG_TILE)
            rem_tiles = -(-M // 256) * -(-(N - n_main) // 256)
            if thin_tail and rem_tiles <= 64:
                gemm(a, w[n_main:], bias[n_main:] if bias is not None else None,
                     residual[:, n_main:] if residual is not None else None, act, out[:, o_main:], tile_cfg=BIG_TILE,
                     splits=max(2, min(16, 256 // rem_tiles, K // 256)))
            else:
                gemm(a, w[n_main:], bias[n_main:] if bias is not None else None,
                     residual[:, n_main:] if residual is not None else None, act, out[:, o_main:])
            return out
    if tile_cfg is None and splits == 1 and long_k_plan(M, N, K) is not None:
        # LLaMA down_proj 767x4096x11008 (48 tiles of 256x256): 5 K-slices on the one-wave-per-SIMD kernel, 100.2 us
        # (incl. the reduce) vs 110.4 for the ring ping-pong kernel x 4 slices and 115.3 for the 128x128 ring on the
        # same box (tools/gemm_bench.cpp, profiles/r02_gemm_tiles.md)
        # round 3: with the pieces as buffer loads the 192x256 ring ping-pong tile x 4 K-slices (64 x 4 = 256 workgroups) is
        # the best form: 85.0 us vs 88.0 (256x256 x 5) and 87.6 (one wave per SIMD x 5), reduce included
        tile_cfg, splits = long_k_plan(M, N, K)
    if tile_cfg is None and splits == 1 and out.dtype != torch.float32 and partial_wave_plan(M, N, K) is not None:
        tile_cfg, splits = partial_wave_plan(M, N, K)
    if tile_cfg is None:
        tile_cfg = pick_tile(M, N, K)
        if splits == 1 and tile_cfg == 4 and K >= 2048 and K % 64 == 0 and M > 1:
            # few tiles and a long K: split K so the grid covers the 256 CUs
            tiles = -(-M // 64) * -(-N // 128)
            if tiles < 160:
                splits = max(1, min(4, 256 // tiles, K // 1024))
        if splits == 1 and tile_cfg == 14 and K >= 4096 and -(-M // 64) * -(-N // 64) <= 256:
            splits = 2                 # ViT fc2 577x1024x4096: 24.2 us vs 27.1 unsplit (29.4 before: 64x128 x 3 splits)
            if -(-M // 64) * -(-N // 64) <= 64:
                splits = 4             # <= 64 tiles: the gate|up remainder 767x256x4096 13.8 vs 17.2 us (x2) / 23.6 (x1);
                #                        batched decode o_proj / down_proj 8x4096x4096 13.4 vs 23.6, x11008 24.9 vs 54.8
            elif M <= 64:
                splits = 1             # batched decode, wide projections: measured unsplit (25.4 / 40.1 us)
    if splits > 1 and workspace is None:
        workspace = torch.empty((splits, M, N), dtype=torch.float32, device=a.device)
    _launch("g4r_gemm_bf16_nt", (
        _p(a), _p(w), _p(out), _p(bias), _p(residual), _p(workspace), M, N, K, a.stride(0), w.stride(0),
        out.stride(0), residual.stride(0) if residual is not None else 0, ACT[act],
        1 if out.dtype == torch.float32 else 0, splits, tile_cfg, _stream(a),),
        tag="gemv_bf16" if (M == 1 and splits == 1 and K >= 512 and K % 8 == 0) else
        ((f"gemm_bf16_nt<{TILE_NAMES.get(tile_cfg, tile_cfg)}>" + ("+splitk" if splits > 1 else "")) if K % 64 == 0
         else "small_linear") + (f" {M}x{N}x{K}" if PROFILER.detail else ""),
        flops=2.0 * M * N * K, nbytes=2.0 * (M * K + N * K) + out.element_size() * M * N, dt=dt)
    return out


def gemv(x, w, norm_weight=None, eps=1e-6, bias=None, residual=None, act=None, out=None, out_dtype=None):
    """out[N] = act(w[N,K] . rmsnorm(x; norm_weight, eps) + bias) + residual for ONE row x [K] (any shape with K
    elements); norm_weight None = no norm.  The decode-step projections: weight streaming, x staged in LDS, the
    preceding RMSNorm fused in (bit-identical to rmsnorm() followed by gemm())."""
    dt = _h16(x, w, residual)
    _f32(bias, norm_weight)
    N, K = w.shape
    assert x.numel() == K and x.is_contiguous() and w.stride(1) == 1
    n_out = N // 2 if act == "swiglu" else N
    if out is None:
        out = torch.empty((1, n_out), dtype=out_dtype or dt, device=x.device)
    assert out.numel() == n_out and out.is_contiguous() and out.dtype in (dt, torch.float32)
    if residual is not None:
        assert residual.numel() == N and residual.is_contiguous()
    _launch("g4r_gemv_rmsnorm_bf16", (_p(x), _p(norm_weight), float(eps), _p(w), _p(out), _p(bias), _p(residual), N, K,
                                      w.stride(0), ACT[act], 1 if out.dtype == torch.float32 else 0, _stream(x),),
            tag="gemv_bf16", flops=2.0 * N * K, nbytes=2.0 * (K + N * K) + out.element_size() * n_out, dt=dt)
    return out


GEMV_BATCH_MAX = 16


def gemv_batch(x, w, norm_weight=None, eps=1e-6, bias=None, residual=None, act=None, out=None, out_dtype=None, variant=0):
    """out[B, N] = act(w[N, K] . rmsnorm(x_b; norm_weight, eps) + bias) + residual_b for B = 2..16 rows x [B, K] sharing ONE pass
    over the weights: the projections of a batched decode step (round 5, csrc/gemv_mfma.hip; `gemv` is the one-row form).
    norm_weight None = no norm; the normalised rows are bit-identical to rmsnorm()'s."""
    dt = _h16(x, w, residual)
    _f32(bias, norm_weight)
    assert x.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1 and x.size(1) == w.size(1)
    B, K = x.shape
    N = w.size(0)
    assert 2 <= B <= GEMV_BATCH_MAX
    n_out = N // 2 if act == "swiglu" else N
    if out is None:
        out = torch.empty((B, n_out), dtype=out_dtype or dt, device=x.device)
    assert out.shape == (B, n_out) and out.stride(1) == 1 and out.dtype in (dt, torch.float32)
    if residual is not None:
        assert residual.shape == (B, N) and residual.stride(1) == 1 and residual.data_ptr() != out.data_ptr()
    _launch("g4r_gemv_batch_bf16", (_p(x), B, x.stride(0), _p(norm_weight), float(eps), _p(w), _p(out), out.stride(0), _p(bias),
                                    _p(residual), residual.stride(0) if residual is not None else 0, N, K, w.stride(0), ACT[act],
                                    1 if out.dtype == torch.float32 else 0, int(variant), _stream(x),),
            tag="gemv_batch", flops=2.0 * B * N * K, nbytes=2.0 * (B * K + N * K) + out.element_size() * B * n_out, dt=dt)
    return out


def gemv_batch_wins(M, N, K):
    """Where gemm() hands a few-row problem to the weight-streaming MFMA kernel instead of the small-M GEMM tiles
    (profiles/r05_gemv_batch.txt).  Launch by launch, weights out of cache, the kernel wins up to 8 rows on q|k|v, o_proj,
    gate|up and lm_head (o_proj 12 vs 18 us) and loses on down_proj (K = 11008: two staging passes); INSIDE the batched decode
    step (hipGraph replay, LLaMA-7B, 767-token prompts) it wins at 2-4 sequences (3.66 / 3.74 / 3.83 vs 3.91 / 4.00 / 4.02 ms per
    step) and loses at 8 (4.51 vs 4.36) -- each of its N / 16 workgroups stages all the rows, which the tiles amortise over 64
    output columns.  So: 2..4 rows, one staging pass, K <= 8192.  Round 6 (profiles/r06_decode_batch_ab.txt): with the fused
    RMSNorm in its one-round-trip form 3.51 / 3.65 / 3.76 ms at 2 / 3 / 4 sequences; 8 sequences still lose (4.47 vs 4.30)."""
    return _GEMV_BATCH_ON and 2 <= M <= GEMV_BATCH_ROWS and 512 <= K <= 8192 and K % 64 == 0 and M * (2 * K + 16) <= 98304


_GEMV_BATCH_ON = os.environ.get("G4R_GEMV_BATCH", "1") != "0"      # (read once: gemm() sits in the eager decode loop)
GEMV_BATCH_ROWS = 4          # rows up to which gemm() routes to the kernel
GEMV_BATCH_VARIANT = 0       # tools (decode_batch_ab.py): 6 / 7 = never / always issue the first weight block before the staging
DECODE_SPLITK_NORM = True       # the batched decode step: K-slice reduce + residual + next RMSNorm as one launch (rmsnorm_splitk)
GEMV_BATCH_FUSED_NORM = True    # the batched decode step: RMSNorm inside the q|k|v / gate|up / lm_head launches (LlamaDecoder._decode_step_batch)


def conv3x3(x, w, bias=None, act=None, groups=1, out=None, out_dtype=None, splits=1, tile_cfg=None,
            workspace=None):
    """x [groups?, B, H, W, Cin] NHWC bf16 (groups dim present iff groups > 1);
    w [Cout, groups*9*Cin] prepared by prep_conv3x3_weight; returns [B, H, W, Cout]."""
    dt = _h16(x, w)
    _f32(bias)
    x = x.contiguous()
    if groups > 1:
        assert x.dim() == 5 and x.size(0) == groups
        B, H, W, Cin = x.shape[1:]
        gstride = x.stride(0)
    else:
        assert x.dim() == 4
        B, H, W, Cin = x.shape
        gstride = 0
    Cout = w.size(0)
    assert w.size(1) == groups * 9 * Cin and w.is_contiguous()
    if out is None:
        out = torch.empty((B, H, W, Cout), dtype=out_dtype or dt, device=x.device)
    assert out.dtype in (dt, torch.float32), f"conv3x3: out is {out.dtype}, operands are {dt}"
    if tile_cfg is None:
        tile_cfg, auto_splits = pick_conv_tile(B * H * W, Cout, groups * 9 * Cin)
        if splits == 1:
            splits = auto_splits
    if splits > 1 and workspace is None:
        workspace = torch.empty((splits, B * H * W, Cout), dtype=torch.float32, device=x.device)
    _launch("g4r_conv3x3_nhwc_bf16", (
        _p(x), _p(w), _p(out), _p(bias), _p(zeros_line(x.device)), _p(workspace), B, H, W, Cin, Cout, groups,
        gstride, ACT[act], 1 if out.dtype == torch.float32 else 0, splits, tile_cfg, _stream(x),),
        tag=f"conv3x3_igemm<{TILE_NAMES.get(tile_cfg, tile_cfg)}>" + (f" {B}x{H}x{W} {Cin}->{Cout} g{groups}/{splits}" if PROFILER.detail else ""),
        flops=2.0 * B * H * W * Cout * groups * 9 * Cin,
        nbytes=2.0 * (groups * B * H * W * Cin + Cout * groups * 9 * Cin + B * H * W * Cout), dt=dt)
    return out


class MlvlMaps:
    """The NHWC maps of all pyramid levels in ONE buffer, stacked [level][b][y][x][C], with per-level views: the layout
    conv3x3_mlvl reads and writes (one implicit GEMM over every level of a fuse round)."""

    def __init__(self, B, sizes, C, device, dtype=torch.bfloat16):
        self.dtype = dtype
        self.B, self.sizes, self.C = B, [(int(h), int(w)) for h, w in sizes], C
        rows = [B * h * w for h, w in self.sizes]
        self.flat = torch.empty((sum(rows), C), dtype=dtype, device=device)
        self.levels, off = [], 0
        for (h, w), n in zip(self.sizes, rows):
            self.levels.append(self.flat[off:off + n].view(B, h, w, C))
            off += n
        self._hw = (ctypes.c_int * len(self.sizes))(*[h for h, _ in self.sizes]), \
            (ctypes.c_int * len(self.sizes))(*[w for _, w in self.sizes])


def conv3x3_mlvl(x, w, bias=None, act=None, out=None):
    """One 3x3 / pad 1 convolution with the SAME weights over every level of a pyramid (MlvlMaps in, MlvlMaps out): the
    fuse round of gpt4roi/models/layers.py:218-236 as a single implicit GEMM launch."""
    assert isinstance(x, MlvlMaps)
    dt = _h16(x.flat, w)
    _f32(bias)
    Cout = w.size(0)
    assert w.size(1) == 9 * x.C and w.is_contiguous() and len(x.sizes) <= 4
    if out is None:
        out = MlvlMaps(x.B, x.sizes, Cout, x.flat.device, dtype=dt)
    assert out.sizes == x.sizes and out.B == x.B and out.C == Cout
    M = x.flat.size(0)
    _launch("g4r_conv3x3_mlvl_nhwc_bf16", (
        _p(x.flat), _p(w), _p(out.flat), _p(bias), _p(zeros_line(x.flat.device)), len(x.sizes), x._hw[0], x._hw[1], x.B,
        x.C, Cout, ACT[act], _stream(x.flat),),
        tag=f"conv3x3_igemm<{TILE_NAMES[BIG_TILE] if BIG_TILE == 34 else '256x256pp32'}>", flops=2.0 * M * Cout * 9 * x.C,
        nbytes=2.0 * (M * x.C + Cout * 9 * x.C + M * Cout), dt=dt)
    return out


def prep_conv3x3_weight(ws, dtype=torch.bfloat16):
    """list of torch conv weights [Cout, Cin, 3, 3] (one per group) -> [Cout, groups*9*Cin] in the 16-bit storage type."""
    if isinstance(ws, torch.Tensor):
        ws = [ws]
    if _layout_kernel_ok(ws):
        co, ci, G = ws[0].size(0), ws[0].size(1), len(ws)
        out = torch.empty((co, G * 9 * ci), dtype=dtype, device=ws[0].device)
        for g, w in enumerate(ws):
            _launch("g4r_conv3x3_weight_layout_bf16", (_p(w), _p(out), co, ci, out.stride(0), g * 9 * ci, 0, _stream(w),),
                    tag="g4r_conv3x3_weight_layout", dt=_h16(dtype))
        return out
    parts = [w.permute(0, 2, 3, 1).reshape(w.size(0), 1, 9 * w.size(1)) for w in ws]
    return torch.cat(parts, 1).reshape(ws[0].size(0), -1).to(dtype).contiguous()


def _layout_kernel_ok(ws):
    """fp32 contiguous [Co, Ci, 3, 3] weights of one shape on the GPU: the one-pass layout kernel applies."""
    w0 = ws[0]
    return all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 4 and w.shape == w0.shape
               and w.shape[2:] == (3, 3) for w in ws)


def flash_attn(q, k, v, heads, scale, causal=False, out=None, kv_len_dev=None, lse=None):
    """q [B, Tq, heads*D], k/v [B, Tk, heads*D] (row-strided views allowed) -> [B, Tq, heads*D].
    lse (optional fp32 [B, heads, Tq]) receives the log2-domain log-sum-exp for flash_attn_bwd."""
    dt = _h16(q, k, v)
    B, Tq, HD = q.shape
    Tk = k.size(1)
    D = HD // heads
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty((B, Tq, HD), dtype=dt, device=q.device)
    assert out.dtype == dt, f"flash_attn: out is {out.dtype}, operands are {dt}"
    _launch("g4r_flash_attn_fwd_bf16", (
        _p(q), _p(k), _p(v), _p(out), B, heads, Tq, Tk, D, q.stride(1), k.stride(1), v.stride(1), out.stride(1),
        q.stride(0), k.stride(0), v.stride(0), out.stride(0), float(scale), int(bool(causal)), _p(kv_len_dev),
        _p(lse), _stream(q),),
        tag=f"flash_attn<{D}>", flops=4.0 * B * heads * Tq * Tk * D * (0.5 if causal and Tq == Tk else 1.0),
        nbytes=2.0 * B * HD * (2 * Tq + 2 * Tk), dt=dt)
    return out


class DecodeAttnWorkspace:
    """Partials + arrival counters of attn_decode (one per decoder: the layers run back to back on one stream)."""

    def __init__(self, heads, head_dim, device, splits=8, batch=1):
        self.splits, self.batch = splits, batch
        self.ws = torch.empty(batch * heads * splits * (head_dim + 2), dtype=torch.float32, device=device)
        self.cnt = torch.zeros(batch * heads, dtype=torch.int32, device=device)      # every call leaves it zero again


def attn_decode(q, k, v, heads, scale, work, kv_len_dev=None, kv_len=None, out=None, qkv=None, cos=None, sin=None,
                defer_merge=False, kv_lens_dev=None, rope_pos_dev=None):
    """One query row per sequence against its KV cache.  Single sequence: q [heads*D] (any shape with that many elements),
    k/v [T_max, heads*D] row-strided views.  Batch of B equal-length sequences: q [B, heads*D] (or qkv [B, 3*heads*D]),
    k/v [B, T_max, heads*D] views -> out [B, heads*D].  Attends the first *kv_len_dev + 1 (or kv_len) rows.
    qkv (instead of q): the raw q|k|v projection rows of the new tokens -- RoPE (cos/sin tables) and the cache append at row
    kv_len - 1 happen inside the launch (rope_qkv + attention in one).
    defer_merge (single sequence): returns None; the per-split partials stay in work.ws for gemv_attn_merge (the o_proj).
    kv_lens_dev (int32 [B], with qkv): RAGGED batch -- sequence b attends its first kv_lens_dev[b] + 1 rows and appends at
    row kv_lens_dev[b]; rope_pos_dev (int32 [1]) = the RoPE position of the new tokens (the padded-layout position)."""
    dt = _h16(q, k, v, qkv)
    batched = k.dim() == 3
    B = k.size(0) if batched else 1
    HD = k.size(-1)
    D = HD // heads
    assert k.stride(-1) == 1 and v.stride(-1) == 1 and v.dim() == k.dim()
    assert (q is None) != (qkv is None) and work.batch >= B
    src = q if q is not None else qkv
    assert src.numel() == B * (HD if q is not None else 3 * HD) and src.is_contiguous()
    if qkv is not None:
        _f32(cos, sin)
        assert cos.size(1) == D // 2 and cos.is_contiguous()
    assert kv_len_dev is not None or kv_len is not None or kv_lens_dev is not None
    assert not (defer_merge and batched)
    if out is None and not defer_merge:
        out = torch.empty((B, HD) if batched else HD, dtype=dt, device=k.device)
    if out is not None:
        assert out.numel() == B * HD and out.is_contiguous()
    if kv_lens_dev is not None:
        assert qkv is not None and batched and kv_lens_dev.dtype == torch.int32 and kv_lens_dev.numel() >= B
        assert rope_pos_dev is None or rope_pos_dev.dtype == torch.int32
        _launch("g4r_attn_decode_ragged_bf16", (_p(qkv), _p(cos), _p(sin), _p(k), _p(v), _p(out), _p(work.ws), _p(work.cnt),
                                                heads, D, k.stride(-2), v.stride(-2), float(scale), work.splits,
                                                _p(kv_lens_dev), _p(rope_pos_dev), B, src.numel() // B, k.stride(0),
                                                v.stride(0), HD, _stream(k),), tag=f"attn_decode<{D}>", dt=dt)
        return out
    _launch("g4r_attn_decode_bf16", (_p(q), _p(qkv), _p(cos), _p(sin), _p(k), _p(v), _p(out), _p(work.ws), _p(work.cnt),
                                     heads, D, int(kv_len or 0), k.stride(-2), v.stride(-2), float(scale), work.splits,
                                     _p(kv_len_dev), int(bool(defer_merge)), B, src.numel() // B,
                                     k.stride(0) if batched else 0, v.stride(0) if batched else 0, HD, _stream(k),),
            tag=f"attn_decode<{D}>", dt=dt)
    return out


def gemv_attn_merge(work, heads, head_dim, w, bias=None, residual=None, out=None, out_dtype=None):
    """out[N] = w[N, heads*head_dim] . (attention output assembled from work.ws) + bias + residual: the o_proj of the decode
    step after attn_decode(..., defer_merge=True)."""
    dt = _h16(w, residual)
    _f32(bias)
    N, K = w.shape
    assert K == heads * head_dim and w.stride(1) == 1
    if out is None:
        out = torch.empty((1, N), dtype=out_dtype or dt, device=w.device)
    assert out.numel() == N and out.is_contiguous()
    if residual is not None:
        assert residual.numel() == N and residual.is_contiguous()
    _launch("g4r_gemv_attn_merge_bf16", (_p(work.ws), work.splits, head_dim, _p(w), _p(out), _p(bias), _p(residual), N, K,
                                         w.stride(0), 1 if out.dtype == torch.float32 else 0, _stream(w),),
            tag="gemv_bf16", flops=2.0 * N * K, nbytes=2.0 * N * K, dt=dt)
    return out


def layernorm(x, gamma, beta, eps=1e-5, relu_in=False, out=None):
    dt = _h16(x)
    _f32(gamma, beta)
    x2 = x.reshape(-1, x.size(-1)) if x.is_contiguous() else x
    assert x2.dim() == 2 and x2.stride(1) == 1
    if out is None:
        out = torch.empty((x2.size(0), x2.size(1)), dtype=dt, device=x.device)
    _launch("g4r_layernorm_bf16", (
        _p(x2), _p(gamma), _p(beta), _p(out), x2.size(0), x2.size(1), x2.stride(0),
                                   out.stride(0), float(eps), int(relu_in), _stream(x),), dt=dt)
    return out.view(x.shape) if x.is_contiguous() else out


def rmsnorm(x, gamma, eps=1e-6, out=None):
    dt = _h16(x)
    _f32(gamma)
    x2 = x.reshape(-1, x.size(-1))
    if out is None:
        out = torch.empty_like(x2)
    _launch("g4r_rmsnorm_bf16", (
        _p(x2), _p(gamma), _p(out), x2.size(0), x2.size(1), x2.stride(0), out.stride(0),
                                 float(eps), _stream(x),), dt=dt)
    return out.view(x.shape)


def groupnorm_affine(x, gamma, beta, groups, eps=1e-5):
    """x [B, H, W, C] bf16 -> scale_shift [B, 2, C] fp32 (deferred GN: y = a*x + s)."""
    dt = _h16(x)
    _f32(gamma, beta)
    B, H, W, C = x.shape
    acc = torch.empty((B, 256, groups, 2), dtype=torch.float32, device=x.device)
    ss = torch.empty((B, 2, C), dtype=torch.float32, device=x.device)
    _launch("g4r_groupnorm_affine_nhwc_bf16", (
        _p(x), _p(gamma), _p(beta), _p(acc), _p(ss), B, H * W, C, groups,
                                               float(eps), _stream(x),), dt=dt)
    return ss


def groupnorm_affine_mlvl(z, gamma, beta, groups, eps=1e-5):
    """z: MlvlMaps (bf16) -> list over levels of scale_shift [B, 2, C] fp32 (views of one [L, B, 2, C] tensor): the deferred
    GN of every level of a fuse round in two launches (bit-identical per level to groupnorm_affine)."""
    assert isinstance(z, MlvlMaps)
    dt = _h16(z.flat)
    _f32(gamma, beta)
    L, B, C = len(z.sizes), z.B, z.C
    acc = torch.empty((L, B, 256, groups, 2), dtype=torch.float32, device=z.flat.device)
    ss = torch.empty((L, B, 2, C), dtype=torch.float32, device=z.flat.device)
    hw = (c_int * L)(*[h * w for h, w in z.sizes])
    _launch("g4r_groupnorm_affine_mlvl_nhwc_bf16", (_p(z.flat), _p(gamma), _p(beta), _p(acc), _p(ss), L, ctypes.cast(hw, P), B, C,
                                                    groups, float(eps), _stream(z.flat),),
            tag="g4r_groupnorm_affine_nhwc_bf16", dt=dt)
    return [ss[l] for l in range(L)]


def fuse_shuffle_mlvl(maps, affs, lvl_list, out):
    """Every target level of a fuse round in one launch.  maps: list over levels of NHWC bf16 maps; affs: list of [B,2,C] fp32
    or None per level; lvl_list: [(target, top, down)] covering every level once (layers.py:108-112); out: MlvlMaps."""
    assert isinstance(out, MlvlMaps)
    dt = _h16(*maps)
    L = len(maps)
    B, _, _, C = maps[0].shape
    # every level exactly once as a target, neighbours inside the pyramid, and `out` laid out for THESE maps: the merged launch
    # writes all levels through one table, so a missing / duplicated target or a mismatched MlvlMaps would silently use level
    # 0 as the neighbour or write out of bounds
    assert sorted(t for t, _, _ in lvl_list) == list(range(L)), "fuse_shuffle_mlvl: lvl_list must name every level once"
    assert len(affs) == L and out.B == B and out.C == C and len(out.sizes) == L
    top, down = [0] * L, [0] * L
    for tar, tp, dn in lvl_list:
        assert 0 <= tp < L and 0 <= dn < L
        top[tar], down[tar] = tp, dn
    for l, m in enumerate(maps):
        assert m.is_contiguous() and m.size(0) == B and m.size(3) == C
        assert out.sizes[l] == (m.size(1), m.size(2)), "fuse_shuffle_mlvl: `out` was built for other map sizes"
    has_aff = any(a is not None for a in affs)
    if has_aff:
        _f32(*[a for a in affs if a is not None])
    PA = c_void_p * L
    ma = PA(*[m.data_ptr() for m in maps])
    aa = PA(*[(a.data_ptr() if a is not None else None) for a in affs]) if has_aff else None
    oa = PA(*[o.data_ptr() for o in out.levels])
    ha = (c_int * L)(*[m.size(1) for m in maps])
    wa = (c_int * L)(*[m.size(2) for m in maps])
    ta = (c_int * L)(*top)
    da = (c_int * L)(*down)
    _launch("g4r_fuse_shuffle_mlvl_nhwc_bf16", (ctypes.cast(ma, P), ctypes.cast(aa, P) if aa is not None else None,
                                                ctypes.cast(ha, P), ctypes.cast(wa, P), ctypes.cast(ta, P), ctypes.cast(da, P),
                                                ctypes.cast(oa, P), L, B, C, _stream(maps[0]),),
            tag="g4r_fuse_shuffle_nhwc_bf16", dt=dt)
    return out


def upsample_coord(tokens, hin, win, H, W, cpad):
    """tokens [B, hin*win, C] bf16 (row/batch strided ok) -> [B, H, W, cpad] with coord channels."""
    dt = _h16(tokens)
    B, n, C = tokens.shape
    assert n == hin * win and tokens.stride(2) == 1
    out = torch.empty((B, H, W, cpad), dtype=dt, device=tokens.device)
    _launch("g4r_upsample_coord_nhwc_bf16", (
        _p(tokens), _p(out), B, hin, win, tokens.stride(0), tokens.stride(1),
                                             H, W, C, cpad, _stream(tokens),), dt=dt)
    return out


def fuse_shuffle(own, top, down, own_aff=None, top_aff=None, down_aff=None, out=None):
    dt = _h16(own, top, down)
    _f32(own_aff, top_aff, down_aff)
    B, H, W, C = own.shape
    if out is None:
        out = torch.empty_like(own)
    _launch("g4r_fuse_shuffle_nhwc_bf16", (
        _p(own), _p(own_aff), H, W, _p(top), _p(top_aff), top.size(1),
                                           top.size(2), _p(down), _p(down_aff), down.size(1), down.size(2),
                                           _p(out), B, C, _stream(own),), dt=dt)
    return out


def im2col_patch14(img, kpad=640, dtype=torch.bfloat16):
    """fp32 image [B,3,S,S] -> the [B*P*P, kpad] patch rows in the 16-bit storage type `dtype`."""
    _f32(img)
    dt = _h16(dtype)
    img = img.contiguous()
    B, _, S, _ = img.shape
    Pn = S // 14
    out = torch.empty((B * Pn * Pn, kpad), dtype=dt, device=img.device)
    _launch("g4r_im2col_patch14_f32", (
        _p(img), _p(out), B, S, kpad, _stream(img),), dt=dt)
    return out


def vit_assemble(patch, cls, pos, B):
    dt = _h16(patch, cls, pos)
    n = patch.size(0) // B
    C = patch.size(1)
    tok = torch.empty((B, n + 1, C), dtype=dt, device=patch.device)
    _launch("g4r_vit_assemble_bf16", (
        _p(patch), _p(cls), _p(pos), _p(tok), B, n, C, _stream(patch),), dt=dt)
    return tok


def rope_qkv(qkv, cos, sin, q_out, k_cache, v_cache, heads, head_dim, pos0, pos_dev=None):
    dt = _h16(qkv, q_out, k_cache, v_cache)
    _f32(cos, sin)
    T = qkv.size(0)
    _launch("g4r_rope_qkv_bf16", (
        _p(qkv), _p(cos), _p(sin), _p(q_out), _p(k_cache), _p(v_cache), T, heads,
                                  head_dim, pos0, _p(pos_dev), _stream(qkv),), dt=dt)


def interleave_gate_up(gate_w, up_w):
    """[F,K],[F,K] -> [2F,K] with rows g0,u0,g1,u1,... for gemm(act="swiglu")."""
    return torch.stack([gate_w, up_w], 1).reshape(2 * gate_w.size(0), gate_w.size(1)).contiguous()


def swiglu(gate_up, out=None):
    dt = _h16(gate_up)
    T, F2 = gate_up.shape
    if out is None:
        out = torch.empty((T, F2 // 2), dtype=dt, device=gate_up.device)
    _launch("g4r_swiglu_bf16", (
        _p(gate_up), _p(out), T, F2 // 2, _stream(gate_up),), dt=dt)
    return out


def splice_embed(ids, embed, img, spi, spi_offset, n_patch, patch_id, bbox_id, im_start_id, im_end_id):
    """ids [B,T] int64; embed [V,C]; img [B,n_patch,C] or None; spi [N,C] or None;
    spi_offset int32 [B+1] or None -> (inputs_embeds [B,T,C], status int32 [B])."""
    dt = _h16(embed, img, spi)
    _lib.require_gpu(ids)
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    B, T = ids.shape
    C = embed.size(1)
    out = torch.empty((B, T, C), dtype=dt, device=ids.device)
    status = torch.empty((B,), dtype=torch.int32, device=ids.device)
    _launch("g4r_splice_embed_bf16", (
        _p(ids), _p(embed), _p(img), _p(spi), _p(spi_offset), _p(out), _p(status),
                                      B, T, C, n_patch if img is not None else 0, patch_id, bbox_id,
                                      im_start_id, im_end_id, embed.size(0), _stream(ids),), dt=dt)
    return out, status


def greedy_advance(logits_row, tok, out_ids, step, pos):
    """tok[0] = argmax(logits_row); out_ids[step[0]] = tok; step += 1; pos += 1 -- all on the device."""
    _f32(logits_row)
    _lib.require_gpu(tok, out_ids, step, pos)
    assert tok.dtype == torch.int64 and out_ids.dtype == torch.int64 and step.dtype == torch.int32 and pos.dtype == torch.int32
    _launch("g4r_greedy_advance_f32", (_p(logits_row), logits_row.numel(), _p(tok), _p(out_ids), _p(step), _p(pos),
                                       out_ids.numel(), _stream(logits_row),))


def batch_advance(nxt, tok, tok32, out_ids, step, pos):
    """tok[b] = tok32[b] = nxt[b]; out_ids[b, step[0]] = nxt[b]; step += 1; pos += 1 -- the batched greedy step.
    pos with more than one element (ragged batch: B cache lengths + the RoPE position): every counter advances."""
    B = nxt.numel()
    assert nxt.dtype == torch.int64 and tok.dtype == torch.int64 and tok32.dtype == torch.int32 and out_ids.dtype == torch.int64
    assert tok.numel() == B and tok32.numel() == B and out_ids.dim() == 2 and out_ids.size(0) == B and out_ids.is_contiguous()
    if pos.numel() > 1:
        assert pos.dtype == torch.int32 and pos.is_contiguous()
        _launch("g4r_batch_advance_ragged", (_p(nxt), B, _p(tok), _p(tok32), _p(out_ids), _p(step), _p(pos), pos.numel(),
                                             out_ids.size(1), _stream(nxt),))
        return
    _launch("g4r_batch_advance", (_p(nxt), B, _p(tok), _p(tok32), _p(out_ids), _p(step), _p(pos), out_ids.size(1),
                                  _stream(nxt),))


def sample_advance(logits_row, tok, out_ids, step, pos, seed, temperature=1.0, top_k=50, top_p=1.0, u_out=None):
    """One sampling step on the device (include/g4r_kernels.h: g4r_sample_advance_f32): tok[0] ~ softmax of the
    temperature / top-k / top-p warped logits, drawn with the Philox uniform of (*step, *seed); counters advance."""
    _f32(logits_row, u_out)
    _lib.require_gpu(tok, out_ids, step, pos, seed)
    assert tok.dtype == torch.int64 and out_ids.dtype == torch.int64 and step.dtype == torch.int32
    assert pos.dtype == torch.int32 and seed.dtype == torch.int64 and seed.numel() == 1
    _launch("g4r_sample_advance_f32", (_p(logits_row), logits_row.numel(), float(temperature), int(top_k or 0),
                                       float(top_p), _p(seed), _p(tok), _p(out_ids), _p(step), _p(pos),
                                       out_ids.numel(), _p(u_out), _stream(logits_row),))


def argmax_rows(logits):
    _f32(logits)
    rows, N = logits.shape
    out = torch.empty((rows,), dtype=torch.int64, device=logits.device)
    _launch("g4r_argmax_rows_f32", (
        _p(logits), logits.stride(0), rows, N, _p(out), _stream(logits),))
    return out


def add_rows(a, b, out=None):
    dt = _h16(a, b)
    a2 = a.reshape(-1, a.size(-1))
    b2 = b.reshape(-1, b.size(-1))
    if out is None:
        out = torch.empty_like(a2)
    _launch("g4r_add_rows_bf16", (
        _p(a2), _p(b2), _p(out), a2.size(0), a2.size(1), b2.size(0), _stream(a),), dt=dt)
    return out.view(a.shape)


def cast_bf16(x, dtype=torch.bfloat16):
    """fp32 -> the 16-bit storage type (round to nearest even)."""
    _f32(x)
    dt = _h16(dtype)
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=dt, device=x.device)
    _launch("g4r_cast_f32_to_bf16", (
        _p(x), _p(y), x.numel(), _stream(x),), dt=dt)
    return y


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # openai/clip-vit-large-patch14 image processor;
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)       # refcoco.py:69-72 uses the same numbers * 255


def image_preprocess(image, size, mean=CLIP_MEAN, std=CLIP_STD, bgr=False, out=None):
    """image uint8 [H, W, 3] on the GPU (row-strided ok) -> fp32 [3, size, size] normalised and bilinearly resized
    (align_corners=False), ready for ClipVisionTower.forward (stack several for a batch)."""
    _lib.require_gpu(image)
    assert image.dtype == torch.uint8 and image.dim() == 3 and image.size(2) == 3 and image.stride(2) == 1 \
        and image.stride(1) == 3
    oh, ow = (size, size) if isinstance(size, int) else size
    if out is None:
        out = torch.empty((3, oh, ow), dtype=torch.float32, device=image.device)
    _launch("g4r_image_preprocess_u8_f32", (_p(image), image.size(0), image.size(1), image.stride(0), int(bool(bgr)),
                                            _p(out), oh, ow, *[float(m) for m in mean], *[float(s) for s in std],
                                            _stream(image),), tag="g4r_image_preprocess_u8_f32")
    return out


def roi_align_mlvl(feats, rois, output_size, scales, sampling_ratio=2, aligned=True, affines=None, out=None):
    """feats: list of NHWC [B,H_l,W_l,C] (bf16 or fp32); rois [N,5] fp32 (batch idx, x1,y1,x2,y2 px);
    -> [L, N, ph, pw, C] in the dtype of feats."""
    L = len(feats)
    dt = feats[0].dtype
    name = {torch.bfloat16: "g4r_roi_align_mlvl_nhwc_bf16", torch.float16: "g4r_roi_align_mlvl_nhwc_bf16",
            torch.float32: "g4r_roi_align_mlvl_nhwc_f32"}[dt]
    for f in feats:
        _lib.require_gpu(f)
        assert f.dtype == dt and f.is_contiguous() and f.dim() == 4
    _f32(rois)
    rois = rois.contiguous()
    B, _, _, C = feats[0].shape
    N = rois.size(0)
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    if out is None:
        out = torch.empty((L, N, ph, pw, C), dtype=dt, device=rois.device)
    PA = c_void_p * L
    fa = PA(*[f.data_ptr() for f in feats])
    aa = None
    if affines is not None:
        _f32(*[a for a in affines if a is not None])
        aa = PA(*[(a.data_ptr() if a is not None else None) for a in affines])
    ha = (c_int * L)(*[f.size(1) for f in feats])
    wa = (c_int * L)(*[f.size(2) for f in feats])
    sa = (c_float * L)(*[float(s) for s in scales])
    # algorithmic bytes: every map texel once + every output element once
    alg = sum(f.numel() for f in feats) * feats[0].element_size() + out.numel() * out.element_size() + rois.numel() * 4
    _launch(name, (ctypes.cast(fa, P), ctypes.cast(aa, P) if aa is not None else None, ctypes.cast(ha, P),
                   ctypes.cast(wa, P), ctypes.cast(sa, P), L, _p(rois), _p(out), B, C, N, ph, pw,
                   int(sampling_ratio), int(bool(aligned)), _stream(rois)),
            tag="roi_align_mlvl_nhwc", flops=0.0, nbytes=float(alg), dt=dt)
    return out


# ---- training rows (include/g4r_train.h) ------------------------------------------------------------------
def flash_attn_bwd(q, k, v, o, do, lse, heads, scale, causal=True, out=None):
    """-> dq, dk, dv (bf16, [B, T, heads*D] contiguous; or written into the three row-strided views of `out`)."""
    _bf16(q, k, v, o, do)
    _f32(lse)
    B, Tq, HD = q.shape
    Tk = k.size(1)
    D = HD // heads
    for t in (q, k, v, o, do):
        assert t.stride(2) == 1
    if out is not None:
        dq, dk, dv = out
        _bf16(dq, dk, dv)
        assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape
        assert dq.stride(2) == 1 and dk.stride(2) == 1 and dv.stride(2) == 1
    else:
        dq = torch.empty((B, Tq, HD), dtype=torch.bfloat16, device=q.device)
        dk = torch.empty((B, Tk, HD), dtype=torch.bfloat16, device=q.device)
        dv = torch.empty((B, Tk, HD), dtype=torch.bfloat16, device=q.device)
    delta = torch.empty((B, heads, Tq), dtype=torch.float32, device=q.device)
    _launch("g4r_flash_attn_bwd_bf16", (
        _p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, heads, Tq, Tk, D,
        q.stride(1), k.stride(1), v.stride(1), o.stride(1), do.stride(1), dq.stride(1), dk.stride(1), dv.stride(1),
        q.stride(0), k.stride(0), v.stride(0), o.stride(0), do.stride(0), dq.stride(0), dk.stride(0), dv.stride(0),
        float(scale), int(bool(causal)), _stream(q),),
        tag=f"flash_attn_bwd<{D}>", flops=14.0 * B * heads * Tq * Tk * D * (0.5 if causal and Tq == Tk else 1.0),
        nbytes=2.0 * B * HD * (4 * Tq + 4 * Tk))
    return dq, dk, dv


def rmsnorm_bwd(x, gamma, dy, dres=None, dgamma=None, eps=1e-6):
    """dx = dres + d rmsnorm(x; gamma)/dx . dy; x, dy, dres [rows, cols] bf16 (row-strided ok)."""
    _bf16(x, dy, dres)
    _f32(gamma, dgamma)
    rows, cols = x.shape
    dx = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    _launch("g4r_rmsnorm_bwd_bf16", (_p(x), _p(gamma), _p(dy), _p(dres), _p(dx), _p(dgamma), rows, cols, x.stride(0),
                                     dy.stride(0), dres.stride(0) if dres is not None else 0, dx.stride(0),
                                     float(eps), _stream(x),), tag="g4r_rmsnorm_bwd_bf16")
    return dx


def layernorm_bwd(x, gamma, dy, dgamma=None, dbeta=None, eps=1e-5, relu_in=False, need_dx=True):
    _bf16(x, dy)
    _f32(gamma, dgamma, dbeta)
    rows, cols = x.shape
    dx = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device) if need_dx else None
    _launch("g4r_layernorm_bwd_bf16", (_p(x), _p(gamma), _p(dy), _p(dx), _p(dgamma), _p(dbeta), rows, cols,
                                       x.stride(0), dy.stride(0), dx.stride(0) if need_dx else 0, float(eps),
                                       int(bool(relu_in)), _stream(x),), tag="g4r_layernorm_bwd_bf16")
    return dx


def swiglu_il(gu):
    """gu [T, 2F] with interleaved (gate, up) columns -> silu(gate) * up [T, F]."""
    _bf16(gu)
    T, F2 = gu.shape
    out = torch.empty((T, F2 // 2), dtype=torch.bfloat16, device=gu.device)
    assert gu.is_contiguous()
    _launch("g4r_swiglu_il_bf16", (_p(gu), _p(out), T, F2 // 2, _stream(gu),), tag="g4r_swiglu_il_bf16")
    return out


def swiglu_il_bwd(gu, dy):
    _bf16(gu, dy)
    T, F2 = gu.shape
    assert gu.is_contiguous() and dy.is_contiguous() and dy.shape == (T, F2 // 2)
    dgu = torch.empty_like(gu)
    _launch("g4r_swiglu_il_bwd_bf16", (_p(gu), _p(dy), _p(dgu), T, F2 // 2, _stream(gu),),
            tag="g4r_swiglu_il_bwd_bf16")
    return dgu


def rope_qkv_bwd(dq, dk, dv, cos, sin, heads, head_dim, pos0=0, out=None, period=0):
    """dq, dk, dv [T, heads*D] (row-strided ok) -> d(qkv) [T, 3*heads*D] (`out`: a dense [T, 3*heads*D] view to fill).
    period > 0: the rows are B stacked sequences of `period` tokens each (positions restart): one launch for the batch."""
    _bf16(dq, dk, dv)
    T, HD = dq.shape
    if period:
        if out is None:
            out = torch.empty((T, 3 * HD), dtype=torch.bfloat16, device=dq.device)
        _bf16(out)
        assert out.shape == (T, 3 * HD) and out.is_contiguous() and T % period == 0
        _launch("g4r_rope_qkv_bwd_batch_bf16", (_p(dq), _p(dk), _p(dv), _p(cos), _p(sin), _p(out), T, int(period), heads,
                                                head_dim, pos0, dq.stride(0), dk.stride(0), dv.stride(0), _stream(dq),),
                tag="g4r_rope_qkv_bwd_bf16")
        return out
    if out is None:
        out = torch.empty((T, 3 * HD), dtype=torch.bfloat16, device=dq.device)
    _bf16(out)
    assert out.shape == (T, 3 * HD) and out.is_contiguous()
    _launch("g4r_rope_qkv_bwd_bf16", (_p(dq), _p(dk), _p(dv), _p(cos), _p(sin), _p(out), T, heads, head_dim, pos0,
                                      dq.stride(0), dk.stride(0), dv.stride(0), _stream(dq),),
            tag="g4r_rope_qkv_bwd_bf16")
    return out


def cross_entropy(logits, labels, loss_sum, grad_scale=None, dlogits=None, n_pad=None):
    """logits fp32 [R, N] (row-strided), labels int64 [R] (< 0 = ignored).  Adds the summed loss to loss_sum
    (fp32 [1]); when dlogits (bf16 [R, >= n_pad]) is given it receives (softmax - onehot) * grad_scale[0]."""
    _f32(logits, loss_sum, grad_scale)
    R, N = logits.shape
    assert labels.dtype == torch.int64 and labels.numel() == R and labels.is_contiguous()
    if dlogits is not None:
        _bf16(dlogits)
        n_pad = n_pad or dlogits.size(1)
    _launch("g4r_cross_entropy_f32", (_p(logits), _p(labels), _p(dlogits), _p(loss_sum), _p(grad_scale), R, N,
                                      logits.stride(0), dlogits.stride(0) if dlogits is not None else 0,
                                      n_pad or N, _stream(logits),), tag="g4r_cross_entropy_f32")
    return dlogits


def transpose(x, r_pad=None, out=None):
    """x [R, C] bf16 (row-strided) -> [C, R_pad] with zero columns R..R_pad."""
    _bf16(x)
    R, C = x.shape
    r_pad = r_pad or R
    if out is None:
        out = torch.empty((C, r_pad), dtype=torch.bfloat16, device=x.device)
    _launch("g4r_transpose_bf16", (_p(x), _p(out), R, C, x.stride(0), out.stride(0), r_pad, _stream(x),),
            tag="g4r_transpose_bf16", nbytes=2.0 * (R * C + C * r_pad))
    return out


def colsum(x, out=None):
    _bf16(x)
    M, N = x.shape
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=x.device)
    _launch("g4r_colsum_bf16", (_p(x), _p(out), M, N, x.stride(0), _stream(x),), tag="g4r_colsum_bf16")
    return out


def relu_bwd(y, dy):
    _bf16(y, dy)
    assert y.is_contiguous() and dy.is_contiguous() and y.numel() == dy.numel()
    dx = torch.empty_like(dy)
    _launch("g4r_relu_bwd_bf16", (_p(y), _p(dy), _p(dx), y.numel(), _stream(y),), tag="g4r_relu_bwd_bf16")
    return dx


def gather_rows(src, idx, out=None):
    """src [R, C] 16-bit (bf16 or fp16; row-strided), idx int32 [n] (negative -> zero row) -> [n, C].  A pure 16-byte row
    copy: one kernel serves both storage types (the batched decode gathers its next embedding rows with it)."""
    dt = _h16(src, out)
    assert idx.dtype == torch.int32 and idx.is_contiguous()
    n, C = idx.numel(), src.size(1)
    if out is None:
        out = torch.empty((n, C), dtype=dt, device=src.device)
    _launch("g4r_gather_rows_bf16", (_p(src), _p(idx), _p(out), n, C, src.stride(0), out.stride(0), _stream(src),),
            tag="g4r_gather_rows_bf16")
    return out


def scatter_add_rows(src, idx, out):
    """out[idx[r]] += src[r] (src bf16 [n, C], idx int32 [n] with -1 = skip, out fp32 [V, C])."""
    _bf16(src)
    _f32(out)
    assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == src.size(0)
    assert out.stride(1) == 1 and src.stride(1) == 1 and out.size(1) == src.size(1)
    _launch("g4r_scatter_add_rows_f32", (_p(src), _p(idx), _p(out), src.size(0), src.size(1), src.stride(0),
                                         out.stride(0), _stream(src),), tag="g4r_scatter_add_rows_f32")
    return out


def adamw(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
          grad_scale=1.0, param_bf16=None):
    _f32(param, exp_avg, exp_avg_sq)
    assert grad.dtype in (torch.bfloat16, torch.float32) and grad.numel() == param.numel()
    assert param.is_contiguous() and grad.is_contiguous() and exp_avg.is_contiguous() and exp_avg_sq.is_contiguous()
    if param_bf16 is not None:
        _bf16(param_bf16)
        assert param_bf16.is_contiguous() and param_bf16.numel() == param.numel()
    _launch("g4r_adamw_f32", (_p(param), _p(grad), int(grad.dtype == torch.bfloat16), _p(exp_avg), _p(exp_avg_sq),
                              _p(param_bf16), param.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                              float(weight_decay), int(step), float(grad_scale), _stream(param),),
            tag="g4r_adamw_f32")


class MultiTensorAdamW:
    """clip_grad_norm_ + AdamW over a fixed list of fp32 master tensors in two launches (include/g4r_train.h:
    g4r_multi_sumsq / g4r_multi_adamw_f32).  The pointer table of the masters / moments / bf16 copies is built once; the
    gradient pointers are re-uploaded only when they change (they are stable when the bucketed reducer owns them)."""
    CHUNK = 4096

    def __init__(self, params, bf16_copies=None, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = list(params)
        _f32(*self.params)
        for p in self.params:
            assert p.is_contiguous()
        dev = self.params[0].device
        self.device = dev
        self.copies = list(bf16_copies) if bf16_copies is not None else [None] * len(self.params)
        for p, c in zip(self.params, self.copies):
            if c is not None:
                _bf16(c)
                assert c.is_contiguous() and c.numel() == p.numel()
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.steps = 0
        i64 = lambda xs: torch.tensor(xs, dtype=torch.int64, device=dev)  # noqa: E731
        self._p = i64([p.data_ptr() for p in self.params])
        self._m = i64([t.data_ptr() for t in self.exp_avg])
        self._v = i64([t.data_ptr() for t in self.exp_avg_sq])
        self._pb = i64([c.data_ptr() if c is not None else 0 for c in self.copies])
        numel = [p.numel() for p in self.params]
        self._numel = i64(numel)
        starts = [0]
        for n in numel:
            starts.append(starts[-1] + -(-n // self.CHUNK))
        self.n_chunks = starts[-1]
        self._starts = torch.tensor(starts, dtype=torch.int32, device=dev)
        self._partial = torch.empty(max(self.n_chunks, 1), dtype=torch.float64, device=dev)
        self.total_sq = torch.zeros(1, dtype=torch.float64, device=dev)
        self._g_key, self._g, self._gflag = None, None, None

    def _grad_table(self, grads):
        key = tuple((g.data_ptr(), g.dtype) for g in grads)
        if key != self._g_key:
            for g, p in zip(grads, self.params):
                assert g.is_cuda and g.is_contiguous() and g.numel() == p.numel() and g.dtype in (torch.float32, torch.bfloat16)
            self._g = torch.tensor([g.data_ptr() for g in grads], dtype=torch.int64, device=self.device)
            self._gflag = torch.tensor([int(g.dtype == torch.bfloat16) for g in grads], dtype=torch.int32, device=self.device)
            self._g_key = key
        return self._g, self._gflag

    def grad_norm_sq(self, grads):
        """Device fp64 [1]: sum of squares over every gradient (no host sync)."""
        g, flag = self._grad_table(grads)
        _launch("g4r_multi_sumsq", (_p(g), _p(self._numel), _p(flag), _p(self._starts), len(self.params), self.n_chunks,
                                    _p(self._partial), _p(self.total_sq), _stream(self.total_sq)), tag="g4r_multi_sumsq")
        return self.total_sq

    def step(self, grads, lr, max_grad_norm=None, pre_scale=1.0, total_sq=None):
        """One update.  max_grad_norm > 0 clips by the global norm like torch.nn.utils.clip_grad_norm_ (computed and
        applied on the device).  Returns the device tensor holding the squared gradient norm (before clipping).
        `total_sq` (device fp64 [1]) overrides the norm this object would compute: the sharded optimizer passes the
        all-reduced sum over every rank's shards."""
        grads = list(grads)
        assert len(grads) == len(self.params)
        clip = max_grad_norm is not None and max_grad_norm > 0
        if clip and total_sq is not None:
            assert total_sq.dtype == torch.float64 and total_sq.is_cuda
            total = total_sq
        else:
            total = self.grad_norm_sq(grads) if clip else None
        g, flag = self._grad_table(grads)
        self.steps += 1
        _launch("g4r_multi_adamw_f32", (_p(self._p), _p(g), _p(self._m), _p(self._v), _p(self._pb), _p(self._numel), _p(flag),
                                        _p(self._starts), len(self.params), self.n_chunks, _p(total),
                                        float(max_grad_norm) if clip else 0.0, float(pre_scale), float(lr),
                                        float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                                        int(self.steps), _stream(self.params[0])), tag="g4r_multi_adamw_f32",
                nbytes=float(sum(p.numel() for p in self.params)) * 30.0)
        return total

    def state_dict(self):
        return dict(step=self.steps, exp_avg=[t.clone() for t in self.exp_avg], exp_avg_sq=[t.clone() for t in self.exp_avg_sq])

    def load_state_dict(self, sd):
        self.steps = int(sd["step"])
        for dst, src in zip(self.exp_avg, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            dst.copy_(src)


def linear_dgrad(dy, w_t, out=None, residual=None):
    """dx [M, K] = dy [M, N] @ W [N, K], given W^T ([K, N_pad], zero padded to a multiple of 64) as the NT weight."""
    return gemm(dy, w_t, residual=residual, out=out)


def gemm_tn(a, b, out=None, accumulate=False, slices=None):
    """C [M, N] fp32 (+)= a^T b with a [K, M], b [K, N] bf16 row-strided: the reduction runs over the ROWS of both operands,
    read as they lie (csrc/gemm_tn.hip).  Few output tiles and a long K: the K axis is cut into slices (fp32 partials)."""
    _bf16(a, b)
    Kd, M = a.shape
    N = b.size(1)
    assert b.size(0) == Kd and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        assert not accumulate
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _f32(out)
    assert out.shape == (M, N) and out.stride(1) == 1
    tiles = -(-M // 256) * -(-N // 256)
    nk = -(-Kd // 32)
    if slices is None:
        slices = 1 if tiles >= 128 else max(1, min(512 // tiles, nk // 16, 64))
    ws = None
    if slices > 1 or accumulate:
        assert out.is_contiguous()
        ws = _wgrad_partials(slices * M * N, a.device)
    _launch("g4r_gemm_tn_bf16", (_p(a), _p(b), _p(out), M, N, Kd, a.stride(0), b.stride(0), out.stride(0), _p(ws),
                                 int(slices), int(bool(accumulate)), _stream(a),),
            tag="gemm_tn" + (f" {M}x{N}x{Kd}/{slices}" if PROFILER.detail else ""), flops=2.0 * M * N * Kd,
            nbytes=2.0 * Kd * (M + N) + 4.0 * M * N)
    return out


def linear_wgrad(dy, x, out_dtype=torch.float32, splits=None, out=None):
    """dW [N, K] = dy^T [N, M] . x [M, K] (torch autograd's grad_weight of a Linear).  Operands whose widths and row strides
    are multiples of 8: the TN kernel, straight from the row-major operands.  Otherwise (the 4-wide box embedding, an
    unpadded vocabulary): transposed copies + one NT GEMM over the zero-padded token axis.
    out: a dense fp32 [N, K] tensor to receive the gradient (a view of an exchange bucket: no copy afterwards)."""
    if out_dtype == torch.float32 and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.dim() == 2 and \
            x.dim() == 2 and dy.size(1) % 8 == 0 and x.size(1) % 8 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 \
            and dy.stride(1) == 1 and x.stride(1) == 1 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 \
            and os.environ.get("G4R_WGRAD_CM", "0") != "1" and (out is None or (out.dtype == torch.float32 and out.is_contiguous())) \
            and dy.size(0) * dy.stride(0) * 2 < 2 ** 31 - 1 and x.size(0) * x.stride(0) * 2 < 2 ** 31 - 1:
        # (operands of 2 GiB and more -- dgu [tokens, 22016] above ~48.7 k tokens -- exceed the TN kernel's 32-bit buffer
        #  offsets: they take the transposed-copy path below instead of raising)
        return gemm_tn(dy, x, out=out)
    if out is not None:
        out.copy_(linear_wgrad(dy, x, out_dtype=out_dtype, splits=splits))
        return out
    M = dy.size(0)
    m_pad = -(-M // 64) * 64
    dyt = transpose(dy, m_pad)
    xt = transpose(x, m_pad)
    if splits is None:
        tiles = -(-dyt.size(0) // 128) * -(-xt.size(0) // 128)
        splits = max(1, min(8, 512 // max(tiles, 1), m_pad // 256))
    return gemm(dyt, xt, out_dtype=out_dtype, splits=splits)


# ---- region-module backward ----------------------------------------------------------------------------------
def groupnorm_stats(z, groups, eps=1e-5):
    """z [B, H, W, C] bf16 -> (mean, rstd) [B, groups, 2] fp32."""
    _bf16(z)
    B, H, W, C = z.shape
    part = torch.empty((B, 256, groups, 2), dtype=torch.float32, device=z.device)
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=z.device)
    _launch("g4r_groupnorm_stats_nhwc_bf16", (_p(z), _p(part), _p(stats), B, H * W, C, groups, float(eps), _stream(z),),
            tag="g4r_groupnorm_stats_nhwc_bf16")
    return stats


def gn_relu_bwd(z, dy, affine, gamma, stats, dgamma, dbeta, groups, out=None):
    """y = relu(GN(z)): dy fp32 [B,H,W,C] -> dz bf16; dgamma / dbeta (fp32 [C]) accumulated."""
    _bf16(z)
    _f32(dy, affine, gamma, stats, dgamma, dbeta)
    B, H, W, C = z.shape
    assert dy.shape == z.shape and dy.is_contiguous() and z.is_contiguous()
    gsum = torch.empty((B, groups, 2), dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z) if out is None else out
    assert dz.shape == z.shape and dz.is_contiguous() and dz.dtype == torch.bfloat16
    _launch("g4r_gn_relu_bwd_nhwc_bf16", (_p(z), _p(dy), _p(affine), _p(gamma), _p(stats), _p(dgamma), _p(dbeta),
                                          _p(gsum), _p(dz), B, H * W, C, groups, _stream(z),),
            tag="g4r_gn_relu_bwd_nhwc_bf16", nbytes=float(z.numel() * (2 * 2 + 2 * 4 + 2)))
    return dz


def fuse_shuffle_bwd(dinp, d_own, d_top, d_down):
    """dinp bf16 [B,H,W,C]; d_* fp32 NHWC gradient maps (accumulated)."""
    _bf16(dinp)
    _f32(d_own, d_top, d_down)
    B, H, W, C = dinp.shape
    assert dinp.is_contiguous() and d_own.shape == dinp.shape
    for d in (d_own, d_top, d_down):
        assert d.is_contiguous() and d.size(0) == B and d.size(3) == C
    _launch("g4r_fuse_shuffle_bwd_nhwc_bf16", (_p(dinp), H, W, _p(d_own), _p(d_top), d_top.size(1), d_top.size(2),
                                               _p(d_down), d_down.size(1), d_down.size(2), B, C, _stream(dinp),),
            tag="g4r_fuse_shuffle_bwd_nhwc_bf16")


def fuse_shuffle_bwd_gather(level, dinps):
    """Gradient w.r.t. the (post GN+ReLU) map of source level `level` from the conv-input gradients `dinps` (list over
    levels, bf16 NHWC) of one fuse round -> fp32 [B, H, W, C], every element written once (no atomics)."""
    L = len(dinps)
    own = dinps[level]
    _bf16(*dinps)
    B, H, W, C = own.shape
    fine = dinps[level - 1] if level >= 1 else None          # target level-1 read this level as `top`
    coarse = dinps[level + 1] if level + 1 < L else None     # target level+1 read this level as `down`
    for d in dinps:
        assert d.is_contiguous()
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=own.device)
    _launch("g4r_fuse_shuffle_bwd_gather_nhwc_bf16", (
        _p(out), _p(own), H, W, _p(fine), fine.size(1) if fine is not None else 0,
        fine.size(2) if fine is not None else 0, _p(coarse), coarse.size(1) if coarse is not None else 0,
        coarse.size(2) if coarse is not None else 0, int(level == L - 1), int(level == 0), B, C, _stream(own),),
        tag="g4r_fuse_shuffle_bwd_gather", nbytes=float(own.numel() * (4 + 2 * 2)))
    return out


def roi_align_mlvl_bwd(dout, lvl_stride, pix_stride, grads, rois, output_size, scales, sampling_ratio=2, aligned=True,
                       roi_offsets=None, atomic=False):
    """dout bf16 (any layout described by the two strides, see g4r_train.h); grads: list of fp32 NHWC maps.
    Default: the atomic-free gather kernel -- every element of `grads` is written once (the maps may be uninitialised),
    bit-reproducible.  atomic=True: the reference-style atomicAdd scatter into ZEROED maps (kept for A/B)."""
    _bf16(dout)
    L = len(grads)
    for g in grads:
        _f32(g)
        assert g.is_contiguous() and g.dim() == 4
    _f32(rois)
    rois = rois.contiguous()
    B, _, _, C = grads[0].shape
    N = rois.size(0)
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    PA = c_void_p * L
    ga = PA(*[g.data_ptr() for g in grads])
    ha = (c_int * L)(*[g.size(1) for g in grads])
    wa = (c_int * L)(*[g.size(2) for g in grads])
    sa = (c_float * L)(*[float(s) for s in scales])
    nbytes = float(sum(g.numel() for g in grads) * 4 + L * N * ph * pw * C * 2)
    if atomic:
        _launch("g4r_roi_align_mlvl_nhwc_bwd_bf16", (_p(dout), int(lvl_stride), int(pix_stride), ctypes.cast(ga, P),
                                                     ctypes.cast(ha, P), ctypes.cast(wa, P), ctypes.cast(sa, P), L,
                                                     _p(rois), B, C, N, ph, pw, int(sampling_ratio), int(bool(aligned)),
                                                     _stream(rois)), tag="roi_align_mlvl_nhwc_bwd", nbytes=nbytes)
        return
    if roi_offsets is not None:
        assert roi_offsets.dtype == torch.int32 and roi_offsets.numel() == B + 1 and roi_offsets.is_cuda
    _launch("g4r_roi_align_mlvl_nhwc_bwd_gather_bf16", (
        _p(dout), int(lvl_stride), int(pix_stride), ctypes.cast(ga, P), ctypes.cast(ha, P), ctypes.cast(wa, P),
        ctypes.cast(sa, P), L, _p(rois), _p(roi_offsets), B, C, N, ph, pw, int(sampling_ratio), int(bool(aligned)),
        _stream(rois)), tag="roi_align_mlvl_nhwc_bwd_gather", nbytes=nbytes)


def conv3x3_dgrad_weight(ws):
    """Conv weights [Cout, Cin, 3, 3] (one per group) -> prepared weight of the transposed convolution
    dX = conv3x3(dY, W^T rotated by 180 degrees); the groups' input gradients are stacked on the output channels:
    result [groups*Cin, 9*Cout]."""
    if isinstance(ws, torch.Tensor):
        ws = [ws]
    if _layout_kernel_ok(ws):
        co, ci, G = ws[0].size(0), ws[0].size(1), len(ws)
        out = torch.empty((G * ci, 9 * co), dtype=torch.bfloat16, device=ws[0].device)
        for g, w in enumerate(ws):
            _launch("g4r_conv3x3_weight_layout_bf16", (_p(w), _p(out[g * ci:]), co, ci, out.stride(0), 0, 1, _stream(w),),
                    tag="g4r_conv3x3_weight_layout")
        return out
    wt = torch.cat([w.flip(2, 3).permute(1, 0, 2, 3) for w in ws], 0)
    return prep_conv3x3_weight(wt)


_WGRAD_PARTIALS = {}


def _wgrad_partials(n_floats, device):
    """One fp32 workspace per device for the pixel-slice partials of g4r_conv3x3_wgrad_nhwc (the launches of a step run back
    to back on one stream; the largest one sizes it)."""
    key = str(device)
    ws = _WGRAD_PARTIALS.get(key)
    if ws is None or ws.numel() < n_floats:
        ws = _WGRAD_PARTIALS[key] = torch.empty(n_floats, dtype=torch.float32, device=device)
    return ws


class ConvWgradNHWC:
    """3x3 weight gradient from NHWC operands for the map geometries that share ONE weight (the levels of a fuse round, or a
    single conv): zero-bordered copies of each level's input and output gradient (buffers reused across steps, the borders
    stay zero) and one launch of the TN kernel over all levels (csrc/gemm_tn.hip).  Channels: multiples of 256."""

    def __init__(self, B, sizes, cin, cout, device):
        self.B, self.sizes, self.cin, self.cout = B, [(int(h), int(w)) for h, w in sizes], cin, cout
        assert cin % 256 == 0 and cout % 256 == 0 and 1 <= len(self.sizes) <= 4
        self.xp, self.dp, self.guard, nks = [], [], [], []
        for h, w in self.sizes:
            krows = -(-(B * (h + 2) * (w + 2)) // 32) * 32
            self.guard.append(w + 3)
            self.xp.append(torch.zeros((krows + 2 * (w + 3), cin), dtype=torch.bfloat16, device=device))
            self.dp.append(torch.zeros((krows, cout), dtype=torch.bfloat16, device=device))
            nks.append(krows // 32)
        L = len(self.sizes)
        self._h = (c_int * L)(*[h for h, _ in self.sizes])
        self._w = (c_int * L)(*[w for _, w in self.sizes])
        # slices of equal length over all levels, about 16 for the largest level: 144 workgroups (16 tiles x 9 taps at
        # 1024 x 1024) per slice -> ~3000 work items of ~600 K tiles for the 336^2 pyramid at 8 images (measured: 12 slices of
        # the largest level 6.20 ms per round, 16: 6.03, 24: 6.14)
        self.slice_tiles = max(64, -(-max(nks) // int(os.environ.get("G4R_TN_LEVEL0_SLICES", 16))))      # (env: sweeps only)
        self.slices = _fn("g4r_conv3x3_wgrad_nhwc_slices")(L, ctypes.cast(self._h, P), ctypes.cast(self._w, P), B, cin, cout,
                                                            self.slice_tiles)
        assert self.slices > 0, "conv3x3_wgrad_nhwc: shape not supported"
        self.flops = 2.0 * 9 * cout * cin * sum(nks) * 32

    def wgrad(self, xs, dys, accumulate_into=None):
        """xs[l] [B,H_l,W_l,Cin], dys[l] [B,H_l,W_l,Cout] bf16 NHWC -> dW [Cout, Cin, 3, 3] fp32, summed over the levels."""
        L = len(self.sizes)
        assert len(xs) == L and len(dys) == L
        for l, (x, dy) in enumerate(zip(xs, dys)):
            _bf16(x, dy)
            h, w = self.sizes[l]
            assert x.is_contiguous() and dy.is_contiguous()
            assert x.shape == (self.B, h, w, self.cin) and dy.shape == (self.B, h, w, self.cout)
            _launch("g4r_nhwc_pad_bf16", (_p(x), _p(self.xp[l]), self.B, h, w, self.cin, self.guard[l], _stream(x),),
                    tag="g4r_nhwc_pad_bf16", nbytes=4.0 * x.numel())
            _launch("g4r_nhwc_pad_bf16", (_p(dy), _p(self.dp[l]), self.B, h, w, self.cout, 0, _stream(x),),
                    tag="g4r_nhwc_pad_bf16", nbytes=4.0 * dy.numel())
        dev = xs[0].device
        ws = _wgrad_partials(self.slices * 9 * self.cout * self.cin, dev)
        dw = accumulate_into
        if dw is None:
            dw = torch.empty((self.cout, self.cin, 3, 3), dtype=torch.float32, device=dev)
        _f32(dw)
        assert dw.shape == (self.cout, self.cin, 3, 3) and dw.is_contiguous()
        PA = c_void_p * L
        da, xa = PA(*[t.data_ptr() for t in self.dp]), PA(*[t.data_ptr() for t in self.xp])
        _launch("g4r_conv3x3_wgrad_nhwc_bf16", (ctypes.cast(da, P), ctypes.cast(xa, P), L, ctypes.cast(self._h, P),
                                                ctypes.cast(self._w, P), self.B, self.cin, self.cout, self.slice_tiles,
                                                _p(ws), _p(dw), int(accumulate_into is not None), _stream(xs[0]),),
                tag="conv3x3_wgrad_tn" + (f" {self.B}x{self.sizes} {self.cin}->{self.cout}/{self.slices}" if PROFILER.detail else ""),
                flops=self.flops, nbytes=2.0 * self.flops / (2.0 * 9 * max(self.cin, self.cout)))
        return dw


class ConvWgradPlan:
    """Buffers for the 3x3 weight gradient of ONE map geometry [B, H, W], reused across steps (the borders stay zero).
    Channel counts that are multiples of 256 (every conv of the region module): the NHWC form above.  Otherwise:
    channel-major zero-bordered copies (3 column shifts of the input) for the NT GEMM."""

    def __init__(self, B, H, W, cin, cout, device):
        self.B, self.H, self.W, self.cin, self.cout = B, H, W, cin, cout
        self.nhwc = cin % 256 == 0 and cout % 256 == 0 and os.environ.get("G4R_WGRAD_CM", "0") != "1"
        if self.nhwc:
            self.tn = ConvWgradNHWC(B, [(H, W)], cin, cout, device)
            return
        self.Wp = -(-(W + 2) // 8) * 8
        self.seg = (H + 2) * self.Wp
        self.base = self.Wp + 8
        self.kp = -(-(B * self.seg) // 64) * 64
        self.ltot = -(-(self.base + self.kp + self.Wp + 8) // 64) * 64
        self.xt = torch.zeros((3, cin, self.ltot), dtype=torch.bfloat16, device=device)
        self.dt = torch.zeros((1, cout, self.ltot), dtype=torch.bfloat16, device=device)
    def _fill(self, src, dst, n_shift):
        B, H, W, C = src.shape
        _launch("g4r_nhwc_to_cm_padded_bf16", (_p(src), _p(dst), B, H, W, C, self.Wp, self.seg, self.base, self.ltot,
                                               n_shift, _stream(src),), tag="g4r_nhwc_to_cm_padded_bf16",
                nbytes=2.0 * src.numel() * (1 + n_shift))

    def wgrad(self, x, dy, accumulate_into=None):
        """x [B,H,W,Cin], dy [B,H,W,Cout] bf16 NHWC -> dW [Cout, Cin, 3, 3] fp32 (torch conv layout).
        accumulate_into: a previous result to add this geometry's gradient to (the levels of a pyramid share the weight)."""
        _bf16(x, dy)
        assert x.is_contiguous() and dy.is_contiguous()
        assert x.shape == (self.B, self.H, self.W, self.cin) and dy.shape == (self.B, self.H, self.W, self.cout)
        if self.nhwc:
            return self.tn.wgrad([x], [dy], accumulate_into=accumulate_into)
        self._fill(x, self.xt, 3)
        self._fill(dy, self.dt, 1)
        a = self.dt[0][:, self.base:self.base + self.kp]
        # [Cout x Cin x pixels] GEMMs: few output tiles, very long K -> split-K over the pixel axis.  MI355X,
        # 1024 x 1024 (tools/wgrad_tiles.py): K = 38.8k: 256x256 ping-pong x16 splits 707 TF/s (64x128 x8: 451);
        # K = 9.9k: 128x128 x8 413; K <= 2.6k: 128x128 x4.
        tiles = -(-self.cout // 128) * -(-self.cin // 128)
        if self.kp >= 16384 and self.cout >= 512 and self.cin >= 512:
            tile, splits = 24, 16
        elif self.kp >= 4096:
            tile, splits = 0, max(1, min(8, 512 // tiles))
        else:
            tile, splits = 0, max(1, min(4, 512 // tiles, self.kp // 256))
        taps = []
        for ky in range(3):
            for kx in range(3):
                o = self.base + (ky - 1) * self.Wp
                taps.append(gemm(a, self.xt[kx][:, o:o + self.kp], out_dtype=torch.float32, splits=splits,
                                 tile_cfg=tile))
        dw = torch.stack(taps, 2).view(self.cout, self.cin, 3, 3)
        return dw if accumulate_into is None else accumulate_into.add_(dw)
