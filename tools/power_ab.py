#!/usr/bin/env python3
"""tools/power_ab.py -- round 6: WHY the hand GEMM trails the vendor kernel on random operands: socket power and shader clock
sampled (sysfs hwmon / pp_dpm_sclk, or rocm-smi) while each arm runs alone for --seconds: the hand kernel's persistent and
per-tile forms and torch.mm (hipBLASLt; calibration only), same operands, bf16 and fp16.  If every arm sits at the same power
and the faster arm runs the higher clock, the difference is ENERGY PER FLOP (the chip clocks to its power budget,
MI355X_MICROARCH.md "DVFS give-back"), not a schedule with idle slots to fill.

    python tools/power_ab.py [--seconds 4] [--out FILE]
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402
from gpt4roi_amd._lib import lib  # noqa: E402


def _sysfs():
    """every amdgpu card's (power, clock) files: the box exposes all the node's cards in sysfs while the process sees ONE GPU;
    the busy card is the one whose power rises under load (sample() returns the card with the highest power)"""
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
        p = [f for h in hw for f in glob.glob(os.path.join(h, "power1_*")) if f.endswith(("average", "input"))]
        fq = [f for h in hw for f in glob.glob(os.path.join(h, "freq1_input"))]
        if p:
            out.append({"card": card, "power": p[0], "freq": fq[0] if fq else None})
    return out


def sample(src):
    best = {}
    for c in src:
        try:
            r = {"power_W": int(open(c["power"]).read()) / 1e6, "card": c["card"].split("/")[4]}
            if c.get("freq"):
                r["sclk_MHz"] = int(open(c["freq"]).read()) / 1e6
            if r["power_W"] > best.get("power_W", -1):
                best = r
        except Exception:                                     # noqa: BLE001
            pass
    return best


def smi_sample():
    try:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(o)
        c = d[sorted(d)[0]]
        r = {}
        for k, v in c.items():
            if "power" in k.lower() and "W" in k:
                try:
                    r["power_W"] = float(v)
                except Exception:
                    pass
            if "sclk" in k.lower():
                r["sclk_MHz"] = float(str(v).strip("()").lower().replace("mhz", ""))
        return r
    except Exception as ex:                                   # noqa: BLE001
        return {"error": repr(ex)[:100]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--probe16", action="store_true", help="tools build only: add the per-tile kernel with every 32x32x16 product replaced by two "
                    "16x16x32 ones (debug mode 65, wrong results, same traffic and FLOPs): what the other MFMA shape buys under the power cap")
    a = ap.parse_args()
    src = _sysfs()
    print(json.dumps({"n_cards": len(src), "first_sysfs": sample(src) if src else None, "first_smi": smi_sample()}), flush=True)
    use_smi = not src or not sample(src)
    M, N, Kd = 12272, 12288, 4096
    rows = []
    for dt in (torch.bfloat16, torch.float16):
        g = torch.Generator(device="cuda").manual_seed(1)
        x = (torch.randn(M, Kd, device="cuda", generator=g) * 0.5).to(dt)
        w = (torch.randn(N, Kd, device="cuda", generator=g) * 0.02).to(dt)
        out = torch.empty(M, N, dtype=dt, device="cuda")

        def hand(mode):
            def f():
                lib().g4r_gemm_debug_mode(mode)
                K.gemm(x, w, out=out, tile_cfg=34)
            return f
        arms = {"hand_persistent": hand(0), "vendor": lambda: torch.mm(x, w.t(), out=out)}
        if dt is torch.bfloat16:
            arms["hand_per_tile"] = hand(61)
        if a.probe16:
            arms = {"hand_per_tile": hand(61), "hand_per_tile_16x16x32_PROBE_wrong_results": hand(65), "hand_per_tile_again": hand(61),
                    "vendor": arms["vendor"]}
        for name, fn in arms.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            samples, stop = [], threading.Event()

            def poll():
                while not stop.is_set():
                    samples.append(smi_sample() if use_smi else sample(src))
                    time.sleep(0.05 if not use_smi else 0.2)
            th = threading.Thread(target=poll, daemon=True)
            n, t0 = 0, time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            th.start()
            e0.record()
            while time.perf_counter() - t0 < a.seconds:
                for _ in range(16):
                    fn()
                n += 16
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
            stop.set()
            th.join()
            lib().g4r_gemm_debug_mode(0)
            us = e0.elapsed_time(e1) * 1e3 / n
            half = samples[len(samples) // 2:]
            pw = [s["power_W"] for s in half if "power_W" in s]
            ck = [s["sclk_MHz"] for s in half if "sclk_MHz" in s]
            row = {"dtype": str(dt).split(".")[-1], "arm": name, "shape": [M, N, Kd], "us": round(us, 1),
                   "TFs": round(2.0 * M * N * Kd / us / 1e6, 1), "samples": len(samples),
                   "power_W_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_W_max": round(max(pw), 1) if pw else None,
                   "sclk_MHz_mean": round(sum(ck) / len(ck), 1) if ck else None,
                   "J_per_TFLOP": round((sum(pw) / len(pw)) * us * 1e-6 / (2.0 * M * N * Kd / 1e12), 3) if pw else None}
            rows.append(row)
            print(json.dumps(row), flush=True)
            time.sleep(1.0)
    if a.out:
        with open(a.out, "w") as fh:
            for r in rows:
                fh.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
