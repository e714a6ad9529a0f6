#!/usr/bin/env python3
"""tools/mfma_power.py -- round 6: matrix-pipe throughput under the socket power cap by MFMA SHAPE, nothing but MFMAs in the loop
(tools/probe/mfma_power_probe.hip: one wave per SIMD, 256 accumulator registers, random bf16 operands in registers):
v_mfma_f32_32x32x16_bf16 (the hand kernels, 16 products per pass) against v_mfma_f32_16x16x32_bf16 (the vendor library's kernels, 64
products per pass = twice the FLOPs of the other arm's pass; the first version of this script counted them as equal).
Each arm runs alone for --seconds; socket power and shader clock are sampled from sysfs (tools/power_ab.py).

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/mfma_power_probe.so tools/probe/mfma_power_probe.hip
    python tools/mfma_power.py [--seconds 4]
"""
import argparse, ctypes, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from power_ab import _sysfs, sample

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=4.0)
a = ap.parse_args()
so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "mfma_power_probe.so"))
so.mfma_probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
so.mfma_probe_launch.restype = ctypes.c_int
out = torch.zeros(4096, device="cuda")
src = _sysfs()
blocks, iters = 256, 20000
flops0 = blocks * 4 * iters * 16 * 32 * 32 * 16 * 2.0        # mode 0: 16 products of 32 x 32 x 16 per pass
flops1 = blocks * 4 * iters * 64 * 16 * 16 * 32 * 2.0        # mode 1: 64 products of 16 x 16 x 32 per pass = TWICE mode 0's
st = torch.cuda.current_stream().cuda_stream
for name, mode in (("v_mfma_f32_32x32x16_bf16 x 16 accumulators", 0), ("v_mfma_f32_16x16x32_bf16 x 64 accumulators", 1),
                   ("v_mfma_f32_32x32x16_bf16 x 16 accumulators (again)", 0)):
    for _ in range(2):
        assert so.mfma_probe_launch(mode, out.data_ptr(), iters, blocks, st) == 0
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            samples.append(sample(src))
            time.sleep(0.05)
    th = threading.Thread(target=poll, daemon=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.perf_counter()
    th.start()
    e0.record()
    while time.perf_counter() - t0 < a.seconds:
        for _ in range(8):
            so.mfma_probe_launch(mode, out.data_ptr(), iters, blocks, st)
        n += 8
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    half = samples[len(samples) // 2:]
    pw = [s["power_W"] for s in half if "power_W" in s]
    ck = [s["sclk_MHz"] for s in half if "sclk_MHz" in s]
    tf = (flops1 if mode == 1 else flops0) / us / 1e6
    clk = sum(ck) / len(ck) if ck else None
    print(json.dumps({"arm": name, "us_per_launch": round(us, 1), "TFs": round(tf, 1), "power_W": round(sum(pw) / len(pw), 1) if pw else None,
                      "sclk_MHz": round(clk, 1) if clk else None,
                      "fraction_of_peak_at_that_clock": round(tf / (2500.0 * clk / 2400.0), 3) if clk else None}), flush=True)
    time.sleep(1.0)
