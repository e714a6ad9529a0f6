#!/usr/bin/env python3
"""tools/persist_ab.py -- round 6: the PERSISTENT form of the one-wave-per-SIMD GEMM (gemm_bf16_w4k64p_kernel) against the
per-tile form (gemm_bf16_w4k64_kernel, debug mode 61) and the vendor library (torch.mm, calibration only), ONE process, the same
operands, on the GEMMs of the merged LLaMA step with their PRODUCTION epilogues (fused RoPE + cache append, residual, SwiGLU) and
the batch-16 ViT block GEMMs (bias, QuickGELU, residual).  bf16 (the debug switch lives in the bf16 instantiation).

Timing as tools/vendor_ab.py: burst = N launches per HIP-event pair, median of R rounds, sides interleaved; sustained = each side
alone >= --sustain seconds, mean of the second half.

    python tools/persist_ab.py [--sustain 1.5] [--burst 8] [--rounds 5] [--out FILE]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402
from gpt4roi_amd._lib import lib  # noqa: E402
from vendor_ab import burst_time, sustained_time  # noqa: E402

DEV = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sustain", type=float, default=1.5)
    ap.add_argument("--burst", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dt = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(1)
    mk = lambda *s, sc=0.5: (torch.randn(*s, device=DEV, generator=g) * sc).to(dt)      # noqa: E731
    st = torch.cuda.current_stream()
    M, B, T, heads = 12272, 16, 767, 32
    x = mk(M, 4096)
    cases = []
    # q|k|v + RoPE + cache append
    wqkv = mk(12288, 4096, sc=0.02)
    ang = torch.rand(2048, 64, device=DEV) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    q = torch.empty(B, T, 4096, dtype=dt, device=DEV)
    kc = torch.empty(B, 1024, 4096, dtype=dt, device=DEV)
    vc = torch.empty(B, 1024, 4096, dtype=dt, device=DEV)
    cases.append(("q|k|v + RoPE 12272x12288x4096", 2.0 * M * 12288 * 4096,
                  lambda: K.gemm_qkv_rope(x, wqkv, B, T, heads, 128, q, kc, vc, cos, sin, 0, tile_cfg=34),
                  lambda: torch.mm(x, wqkv.t())))
    wq_plain = wqkv
    out_qkv = torch.empty(M, 12288, dtype=dt, device=DEV)
    cases.append(("plain 12272x12288x4096", 2.0 * M * 12288 * 4096, lambda: K.gemm(x, wq_plain, out=out_qkv, tile_cfg=34),
                  lambda: torch.mm(x, wq_plain.t(), out=out_qkv)))
    wo = mk(4096, 4096, sc=0.02)
    res = mk(M, 4096)
    out_o = torch.empty(M, 4096, dtype=dt, device=DEV)
    cases.append(("o_proj + residual 12272x4096x4096", 2.0 * M * 4096 * 4096,
                  lambda: K.gemm(x, wo, residual=res, out=out_o, tile_cfg=34), lambda: torch.addmm(res, x, wo.t(), out=out_o)))
    wgu = mk(21760, 4096, sc=0.02)
    out_gu = torch.empty(M, 10880, dtype=dt, device=DEV)
    tmp_gu = torch.empty(M, 21760, dtype=dt, device=DEV)
    cases.append(("gate|up + SwiGLU 12272x21760x4096", 2.0 * M * 21760 * 4096,
                  lambda: K.gemm(x, wgu, act="swiglu", out=out_gu, tile_cfg=34), lambda: torch.mm(x, wgu.t(), out=tmp_gu)))
    f = mk(M, 11008)
    wd = mk(4096, 11008, sc=0.02)
    cases.append(("down_proj + residual 12272x4096x11008", 2.0 * M * 4096 * 11008,
                  lambda: K.gemm(f, wd, residual=res, out=out_o, tile_cfg=34), lambda: torch.addmm(res, f, wd.t(), out=out_o)))
    # ViT block GEMMs at batch 16
    Mv = 9232
    xv = mk(Mv, 1024)
    w1, b1 = mk(4096, 1024, sc=0.03), torch.randn(4096, device=DEV)
    o1 = torch.empty(Mv, 4096, dtype=dt, device=DEV)
    cases.append(("ViT fc1 + bias + QuickGELU 9232x4096x1024", 2.0 * Mv * 4096 * 1024,
                  lambda: K.gemm(xv, w1, bias=b1, act="quick_gelu", out=o1, tile_cfg=34), lambda: torch.mm(xv, w1.t(), out=o1)))
    wq_, bq_ = mk(3072, 1024, sc=0.03), torch.randn(3072, device=DEV)
    oq = torch.empty(Mv, 3072, dtype=dt, device=DEV)
    cases.append(("ViT qkv + bias 9232x3072x1024", 2.0 * Mv * 3072 * 1024,
                  lambda: K.gemm(xv, wq_, bias=bq_, out=oq, tile_cfg=34), lambda: torch.mm(xv, wq_.t(), out=oq)))
    rows = []
    for name, flops, hand, vendor in cases:
        def arm(mode):
            def run():
                lib().g4r_gemm_debug_mode(mode)
                hand()
            return run
        arms = {"persistent": arm(0), "per_tile": arm(61), "vendor": vendor}
        for fn in arms.values():
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        burst = {k: [] for k in arms}
        for _ in range(a.rounds):
            for k, fn in arms.items():
                burst[k].append(burst_time(fn, a.burst, st))
        bmed = {k: sorted(v)[len(v) // 2] for k, v in burst.items()}
        sus = {}
        for k, fn in arms.items():
            sus[k] = sustained_time(fn, a.sustain, st, bmed[k])[0]
        lib().g4r_gemm_debug_mode(0)
        row = {"case": name, "burst_us": {k: round(v, 1) for k, v in bmed.items()}, "sustained_us": {k: round(v, 1) for k, v in sus.items()},
               "sustained_TFs": {k: round(flops / v / 1e6, 1) for k, v in sus.items()},
               "persistent_over_per_tile": round(sus["per_tile"] / sus["persistent"], 3),
               "persistent_over_vendor": round(sus["vendor"] / sus["persistent"], 3)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if a.out:
        with open(a.out, "w") as fh:
            for r in rows:
                fh.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
