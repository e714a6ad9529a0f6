#!/usr/bin/env python3
"""tools/gemv_batch_check.py -- g4r_gemv_batch (csrc/gemv_mfma.hip): B = 2..16 rows through one projection against fp32 torch
(every epilogue, fused RMSNorm == the separate launch bit for bit, one and several K passes, ragged N), then timings of the LLaMA-7B
projections of a batched decode step against the small-M MFMA GEMM tiles K.gemm picks (the path of rounds 2-4)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(7)
bad = 0


def R(*s, scale=0.5, dtype=torch.bfloat16):
    return (torch.randn(*s, device=dev, generator=g) * scale).to(dtype)


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-9)).item()


def check(name, got, ref, tol):
    global bad
    e = rel(got, ref)
    ok = e < tol
    bad += not ok
    print(json.dumps({"case": name, "err": round(e, 5), "ok": ok}), flush=True)


for dtype, tol in ((torch.bfloat16, 6e-3), (torch.float16, 1e-3)):
    for (B, N, Kd) in [(2, 1000, 512), (3, 4096, 4096), (8, 12288, 4096), (8, 4096, 11008), (16, 2050, 11008), (5, 32006, 4096), (16, 4096, 4096)]:
        xw = R(B, Kd + 64, dtype=dtype)
        x = xw[:, :Kd]
        w = R(N, Kd, scale=0.05, dtype=dtype)
        bias = torch.randn(N, device=dev, generator=g)
        res = R(B, N, dtype=dtype)
        ref = x.float() @ w.float().t()
        tag = f"{str(dtype)[6:]} {B}x{N}x{Kd}"
        check(tag + " plain", K.gemv_batch(x, w), ref, tol)
        check(tag + " bias+silu+residual", K.gemv_batch(x, w, bias=bias, residual=res, act="silu"), F.silu(ref + bias) + res.float(), 2 * tol)
        check(tag + " fp32 out", K.gemv_batch(x, w, bias=bias, out_dtype=torch.float32), ref + bias, 2e-4)
        if N % 4 == 0:
            check(tag + " swiglu", K.gemv_batch(x, w, act="swiglu"), F.silu(ref[:, 0::2]) * ref[:, 1::2], 3 * tol)
        for v in (2, 3, 5):
            check(tag + f" variant {v}", K.gemv_batch(x, w, variant=v), ref, tol)
    for B in (2, 7, 16):
        x = R(B, 4096, dtype=dtype)
        gamma = 1 + 0.1 * torch.randn(4096, device=dev, generator=g)
        w = R(1536, 4096, scale=0.05, dtype=dtype)
        fused = K.gemv_batch(x, w, norm_weight=gamma, eps=1e-6)
        sep = K.gemv_batch(K.rmsnorm(x, gamma, 1e-6), w)
        same = torch.equal(fused, sep)
        bad += not same
        print(json.dumps({"case": f"{str(dtype)[6:]} fused rmsnorm B={B}: bit-identical to the separate launch", "ok": same}), flush=True)
print("ALL OK" if bad == 0 else f"{bad} FAILED", flush=True)
if bad or "--no-time" in sys.argv:
    sys.exit(1 if bad else 0)


def burst(fn, n=20, rounds=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


# weights of TWO layers alternate so that no launch finds its weights in the Infinity Cache
for B in (2, 4, 8, 16):
    for (N, Kd, what, kw) in [(12288, 4096, "q|k|v", {}), (4096, 4096, "o_proj", {"res": True}), (22016, 4096, "gate|up", {"act": "swiglu"}),
                              (4096, 11008, "down_proj", {"res": True}), (32006, 4096, "lm_head", {"f32": True})]:
        ws = [R(N, Kd, scale=0.05) for _ in range(3)]
        x = R(B, Kd)
        res = R(B, N) if kw.get("res") else None
        act = kw.get("act")
        od = torch.float32 if kw.get("f32") else None
        row = {"B": B, "gemm": what, "MB": round((N * Kd * 2) / 1e6, 1)}
        it = {"i": 0}

        def tile():
            it["i"] += 1
            return K.gemm(x, ws[it["i"] % 3], residual=res, act=act, out_dtype=od)
        os.environ["G4R_GEMV_BATCH"] = "0"
        tile(); torch.cuda.synchronize()
        us = burst(tile)
        row["mfma_tiles"] = [round(us, 1), round(N * Kd * 2 / us / 1e6, 2)]
        for v in (0, 2, 5):
            def gb():
                it["i"] += 1
                return K.gemv_batch(x, ws[it["i"] % 3], residual=res, act=act, out_dtype=od, variant=v)
            gb(); torch.cuda.synchronize()
            us = burst(gb)
            row[f"gemv_mfma_v{v}"] = [round(us, 1), round(N * Kd * 2 / us / 1e6, 2)]
        print(json.dumps(row), flush=True)
