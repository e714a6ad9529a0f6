"""tools/decode_batch_ab.py -- the batched decode step (LlamaDecoder.decode_graph_batch, LLaMA-7B shapes, hipGraph replay) under the
knobs of the weight-streaming MFMA kernel (csrc/gemv_mfma.hip): rows routed to it, RMSNorm fused into its launches, first weight
block issued before the staging is waited for.  One decoder; every configuration captures its own graph.
    python tools/decode_batch_ab.py [--batches 2,4,8] [--tokens 48]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="2,4,8")
    ap.add_argument("--tokens", type=int, default=48)
    ap.add_argument("--prompt", type=int, default=767)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.llama import LlamaDecoder
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    l = syn.LLAMA_7B
    lsd = syn.llama_state(l["hidden"], l["inter"], 32, 32006, seed=1, device=dev, dtype=torch.bfloat16)
    dec = LlamaDecoder(lsd, heads=l["heads"], max_positions=2048, device=dev)
    del lsd
    # name: (rows routed to the kernel, fused norm, variant (6 / 7 = never / always early loads; 0: only without a fused norm))
    configs = {"tiles_separate_reduce_and_norm": (0, False, 0, False), "tiles": (0, False, 0), "kernel_noearly_sepnorm": (16, False, 6), "kernel_early_sepnorm": (16, False, 7),
               "kernel_noearly_fused": (16, True, 6), "kernel_early_fused": (16, True, 7), "kernel_production_fused": (16, True, 0)}
    for B in [int(x) for x in a.batches.split(",")]:
        emb = (torch.randn(B, a.prompt, l["hidden"], device=dev) * 0.02).to(torch.bfloat16)
        ids, ms = {}, {k: [] for k in configs}
        for r in range(a.rounds):
            for name, cfg in configs.items():
                rows, fused, variant = cfg[:3]
                K.GEMV_BATCH_ROWS, K.GEMV_BATCH_FUSED_NORM, K.GEMV_BATCH_VARIANT = rows, fused, variant
                K.DECODE_SPLITK_NORM = cfg[3] if len(cfg) > 3 else True
                dec._bstate = {}
                out = dec.decode_graph_batch(emb, a.tokens + 2)
                torch.cuda.synchronize()
                ids.setdefault(name, out)
                ts = []
                for n in (a.tokens + 2, 2):
                    best = 1e9
                    for _ in range(2):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        dec.decode_graph_batch(emb, n)
                        torch.cuda.synchronize()
                        best = min(best, time.perf_counter() - t0)
                    ts.append(best)
                ms[name].append(round((ts[0] - ts[1]) / a.tokens * 1e3, 3))
        ref = ids["kernel_noearly_sepnorm"]
        print({"B": B, "ms_per_step": ms, "ids_equal_to_kernel_noearly_sepnorm": {k: v == ref for k, v in ids.items()}}, flush=True)


if __name__ == "__main__":
    main()
