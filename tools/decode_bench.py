"""tools/decode_bench.py -- the KV-cache decode step of row a17 in isolation (LLaMA-7B shapes, synthetic weights, batch 1):
ms per token of the captured hipGraph, and the weight-streaming rate it corresponds to.  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/runs/gpu_run10.sh).
    python tools/decode_bench.py [--tokens 64] [--prompt 767] [--layers 32] [--sample]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--prompt", type=int, default=767)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--sample", action="store_true")
    ap.add_argument("--batch", type=int, default=1, help="B equal-length sequences decoded together (greedy)")
    ap.add_argument("--ab", type=int, default=0, help="A/B against g4r_gemm_debug_mode(N) (62: GEMV without the early weight loads): "
                    "both forms timed in turn, ids compared")
    a = ap.parse_args()
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.llama import LlamaDecoder
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    l = syn.LLAMA_7B
    lsd = syn.llama_state(l["hidden"], l["inter"], a.layers, 32006, seed=1, device=dev, dtype=torch.bfloat16)
    dec = LlamaDecoder(lsd, heads=l["heads"], max_positions=2048, device=dev)
    del lsd
    emb = (torch.randn(a.batch, a.prompt, l["hidden"], device=dev) * 0.02).to(torch.bfloat16)
    sampler = (0.2, 50, 1.0) if a.sample else None

    def run(n):
        if a.batch > 1:
            return dec.decode_graph_batch(emb, n)
        return dec.decode_graph(emb, n, sampler=sampler, seed=1)
    ids_ship = run(a.tokens + 2)                       # warm-up + graph capture (the batched graph is keyed by length)
    torch.cuda.synchronize()

    def timed(n):
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    timed(4)
    t_long, t_short = min(timed(a.tokens + 2) for _ in range(3)), min(timed(2) for _ in range(3))
    dt = (t_long - t_short) / a.tokens
    wbytes = sum(L[k].numel() * 2 for L in dec.layers for k in ("wqkv", "wo", "wgu", "wd")) + dec.lm_head.numel() * 2
    print({"batch": a.batch, "ms_per_step": round(1e3 * dt, 3), "tokens_per_s_all_sequences": round(a.batch / dt, 1),
           "ms_per_token": round(1e3 * dt, 3), "tokens_per_s": round(1 / dt, 1), "weight_GB": round(wbytes / 1e9, 2),
           "weight_stream_GBps": round(wbytes / dt / 1e9, 1), "layers": a.layers, "prompt": a.prompt, "sampled": a.sample})


    if a.ab:
        # the debug mode is read at LAUNCH time by the host side of the C ABI, i.e. at capture: a second decoder captures its own
        # graph under it
        from gpt4roi_amd._lib import lib
        lib().g4r_gemm_debug_mode(a.ab)
        lsd = syn.llama_state(l["hidden"], l["inter"], a.layers, 32006, seed=1, device=dev, dtype=torch.bfloat16)
        dec2 = LlamaDecoder(lsd, heads=l["heads"], max_positions=2048, device=dev)
        del lsd

        def run2(n):
            return dec2.decode_graph_batch(emb, n) if a.batch > 1 else dec2.decode_graph(emb, n, sampler=sampler, seed=1)
        ids_dbg = run2(a.tokens + 2)
        torch.cuda.synchronize()
        lib().g4r_gemm_debug_mode(0)
        res = {"ship": [], "dbg": []}
        for _ in range(5):
            for name, fn in (("ship", run), ("dbg", run2)):
                ts = []
                for n in (a.tokens + 2, 2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fn(n)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                res[name].append((ts[0] - ts[1]) / a.tokens * 1e3)
        same = torch.equal(torch.as_tensor(ids_ship), torch.as_tensor(ids_dbg))
        print({"ab_mode": a.ab, "ms_per_token_shipped": [round(x, 3) for x in res["ship"]],
               "ms_per_token_debug_mode": [round(x, 3) for x in res["dbg"]], "ids_identical": bool(same)})


if __name__ == "__main__":
    main()
