#!/usr/bin/env python3
"""tools/vendor_ab.py -- same-process A/B of the hand-written GEMM (K.gemm) against the vendor library
(torch.mm -> hipBLASLt).  CALIBRATION ONLY: the vendor library is never on the product path; this answers one question:
"is the hand kernel at the chip's sustained ceiling on random operands, or is there headroom the vendor kernel shows?"

Protocol: ONE process, the SAME randn operands for both sides, two timings per shape
  burst      N launches per HIP-event pair, median of R rounds, the two sides interleaved round by round
  sustained  each side alone in a loop of >= --sustain seconds (DVFS equilibrium under the power cap); the figure is the mean
             launch time over the second half of the loop
Shapes: the merged-16 LLaMA prefill GEMMs (M = 12272), the batch-16 ViT block GEMMs (M = 9232, K or N = 1024), 4096^3.

    python tools/vendor_ab.py [--sustain 2.0] [--burst 8] [--rounds 5] [--fill n|u|z] [--shapes all|llama|vit] [--out FILE]
    python tools/vendor_ab.py --trace      # a few launches per shape only: run under rocprofv3 --kernel-trace --stats to
                                           # read the vendor kernel's NAME (macro tile, MFMA shape, staging) per shape
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402

SHAPES = {
    "llama": [(12272, 12288, 4096, "q|k|v"), (12272, 4096, 4096, "o_proj"), (12272, 22016, 4096, "gate|up"),
              (12272, 4096, 11008, "down_proj"), (12272, 32006, 4096, "lm_head(all rows)")],
    "vit": [(9232, 3072, 1024, "vit qkv"), (9232, 4096, 1024, "vit fc1"), (9232, 1024, 1024, "vit o"),
            (9232, 1024, 4096, "vit fc2"), (577, 3072, 1024, "vit qkv b1"), (577, 1024, 4096, "vit fc2 b1")],
    "square": [(4096, 4096, 4096, "4096^3"), (8192, 8192, 4096, "8192^2x4096")],
    "pyramid": [(48960, 1024, 1088, "input 1x1 conv")],
}


def operands(M, N, Kd, fill, dtype, dev):
    g = torch.Generator(device=dev).manual_seed(M * 7 + N * 3 + Kd)
    if fill == "z":
        return torch.zeros(M, Kd, dtype=dtype, device=dev), torch.zeros(N, Kd, dtype=dtype, device=dev)
    if fill == "u":
        mk = lambda *s: (torch.rand(*s, device=dev, generator=g) - 0.5).to(dtype)      # noqa: E731
    else:
        mk = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dtype)     # noqa: E731
    return mk(M, Kd), mk(N, Kd)


def burst_time(fn, burst, stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(burst):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / burst            # us per launch


def sustained_time(fn, seconds, stream, est_us):
    """Loop >= `seconds`; events every `chunk` launches; mean over the chunks of the second half."""
    chunk = max(4, int(50e3 / max(est_us, 1.0)))        # ~50 ms of work per chunk
    evs = [torch.cuda.Event(enable_timing=True)]
    evs[0].record(stream)
    t0 = time.perf_counter()
    while True:
        for _ in range(chunk):
            fn()
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        evs.append(e)
        if len(evs) % 4 == 0:
            e.synchronize()                               # keep the queue bounded, the GPU never idles: 3 chunks ahead
            if time.perf_counter() - t0 >= seconds:
                break
    evs[-1].synchronize()
    per = [evs[i].elapsed_time(evs[i + 1]) * 1e3 / chunk for i in range(len(evs) - 1)]
    half = per[len(per) // 2:]
    return sum(half) / len(half), per[0], len(per) * chunk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sustain", type=float, default=2.0)
    ap.add_argument("--burst", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--fill", default="n")
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    try:
        torch.backends.cuda.preferred_blas_library("hipblaslt")
    except Exception as ex:                                # pragma: no cover
        print("# preferred_blas_library:", ex)
    stream = torch.cuda.current_stream()
    groups = list(SHAPES) if a.shapes == "all" else a.shapes.split(",")
    out = open(a.out, "w") if a.out else None

    def emit(s):
        print(s, flush=True)
        if out:
            out.write(s + "\n")
            out.flush()

    emit(f"# tools/vendor_ab.py: hand kernel (K.gemm, production dispatch) vs torch.mm (hipBLASLt), one process, same operands; "
         f"fill={a.fill} dtype={a.dtype} burst={a.burst}x{a.rounds} sustained>={a.sustain}s; torch {torch.__version__}")
    for grp in groups:
        for (M, N, Kd, what) in SHAPES[grp]:
            x, w = operands(M, N, Kd, a.fill, dtype, dev)
            y_h = torch.empty(M, N, dtype=dtype, device=dev)
            y_v = torch.empty(M, N, dtype=dtype, device=dev)
            wt = w.t()
            hand = lambda: K.gemm(x, w, out=y_h)                      # noqa: E731
            vend = lambda: torch.mm(x, wt, out=y_v)                   # noqa: E731
            for _ in range(3):
                hand()
                vend()
            torch.cuda.synchronize()
            err = (y_h.float() - y_v.float()).abs().max().item() / max(y_v.float().abs().max().item(), 1e-9)
            if a.trace:
                emit(json.dumps({"shape": [M, N, Kd], "what": what, "rel_diff_hand_vs_vendor": round(err, 5)}))
                continue
            flops = 2.0 * M * N * Kd
            bh, bv = [], []
            for _ in range(a.rounds):
                bh.append(burst_time(hand, a.burst, stream))
                bv.append(burst_time(vend, a.burst, stream))
            bh.sort()
            bv.sort()
            mh, mv = bh[len(bh) // 2], bv[len(bv) // 2]
            sh, sh0, nh = sustained_time(hand, a.sustain, stream, mh)
            time.sleep(0.5)
            sv, sv0, nv = sustained_time(vend, a.sustain, stream, mv)
            time.sleep(0.5)
            emit(json.dumps({
                "shape": [M, N, Kd], "what": what, "rel_diff_hand_vs_vendor": round(err, 5),
                "burst_us": {"hand": round(mh, 1), "vendor": round(mv, 1)},
                "burst_TFs": {"hand": round(flops / mh / 1e6, 1), "vendor": round(flops / mv / 1e6, 1)},
                "sustained_us": {"hand": round(sh, 1), "vendor": round(sv, 1)},
                "sustained_TFs": {"hand": round(flops / sh / 1e6, 1), "vendor": round(flops / sv / 1e6, 1)},
                "first_chunk_us": {"hand": round(sh0, 1), "vendor": round(sv0, 1)},
                "launches": {"hand": nh, "vendor": nv},
                "hand_over_vendor": {"burst": round(mv / mh, 3), "sustained": round(sv / sh, 3)}}))
            del x, w, y_h, y_v
    if out:
        out.close()


if __name__ == "__main__":
    main()
