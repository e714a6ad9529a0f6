#!/usr/bin/env python3
"""tools/persist_stress.py -- race screen of the persistent GEMM: every epilogue mode at the merged LLaMA shapes, N repeats, each
output compared bit for bit with the first one and with the per-tile form (debug mode 61) / the ring tile 24, while a second
stream keeps the memory system busy (uneven load).  Counted vmcnt waits that leave stores in flight must never let a fragment read
overtake its LDS-DMA piece."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402
from gpt4roi_amd._lib import lib  # noqa: E402

DEV = "cuda"
N_REP = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def main():
    bad = 0
    for dt in (torch.bfloat16, torch.float16):
        g = torch.Generator(device=DEV).manual_seed(3)
        mk = lambda *s, sc=0.5: (torch.randn(*s, device=DEV, generator=g) * sc).to(dt)      # noqa: E731
        M, B, T, heads = 12272, 16, 767, 32
        x = mk(M, 4096)
        wqkv, wo, wgu, wd = mk(12288, 4096, sc=0.02), mk(4096, 4096, sc=0.02), mk(21760, 4096, sc=0.02), mk(4096, 11008, sc=0.02)
        f, res = mk(M, 11008), mk(M, 4096)
        ang = torch.rand(2048, 64, device=DEV) * 6.28
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        noise_a, noise_b = torch.randn(64 << 20, device=DEV), torch.empty(64 << 20, device=DEV)
        side = torch.cuda.Stream()

        def qkv(tile):
            q = torch.zeros(B, T, 4096, dtype=dt, device=DEV)
            kc = torch.zeros(B, 1024, 4096, dtype=dt, device=DEV)
            vc = torch.zeros(B, 1024, 4096, dtype=dt, device=DEV)
            K.gemm_qkv_rope(x, wqkv, B, T, heads, 128, q, kc, vc, cos, sin, 0, tile_cfg=tile)
            return torch.cat([q.view(-1), kc.view(-1), vc.view(-1)])
        cases = {"qkv+rope": qkv, "plain": lambda t: K.gemm(x, wqkv, tile_cfg=t), "o_proj+res": lambda t: K.gemm(x, wo, residual=res, tile_cfg=t),
                 "gate|up swiglu": lambda t: K.gemm(x, wgu, act="swiglu", tile_cfg=t), "down+res": lambda t: K.gemm(f, wd, residual=res, tile_cfg=t)}
        for name, fn in cases.items():
            ref24 = fn(24)
            first = fn(34)
            ok24 = bool(torch.equal(first, ref24))
            mism = 0
            for r in range(N_REP):
                if r % 2 == 0:
                    with torch.cuda.stream(side):                       # uneven background load on some repeats
                        noise_b.copy_(noise_a)
                out = fn(34)
                if not torch.equal(out, first):
                    mism += 1
                    d = (out.float() - first.float()).abs()
                    print(f"  MISMATCH {name} rep {r}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}", flush=True)
            torch.cuda.synchronize()
            print(f"{str(dt).split('.')[-1]:9s} {name:16s} equals tile 24: {ok24}; {N_REP} repeats, mismatching repeats: {mism}", flush=True)
            bad += mism + (0 if ok24 else 1)
    print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
