"""Phase stamps (s_memtime) of one wave per key group inside csrc/attention_v2.hip, step 2 of the heaviest workgroup.
The stamps are compiled in only with  G4R_EXTRA_HIPCC_FLAGS=-DG4R_ATTN2_PROBE python -m gpt4roi_amd.build --force
(rebuild without the flag afterwards: the production library carries no probe code)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K, _lib
lib = _lib.lib()
import ctypes
R = lambda *s: (torch.randn(*s, device="cuda") * 0.7).to(torch.bfloat16)
# stamp slots of A2_STAMP in program order: 0 A start, 1 QK done, 2 (unused), 3 max/alpha done, 4 B start, 9 V reads + pieces issued,
# 5 exp/rescale done, 6 PV done, 7 vmcnt(0) done, 8 barrier passed
order = [0, 1, 3, 4, 9, 5, 6, 7, 8]
names = {1: "QK done", 3: "max/alpha done (A end)", 4: "B start", 9: "V reads + pieces issued", 5: "exp/rescale done", 6: "PV done",
         7: "vmcnt(0) done", 8: "barrier passed"}
for (B, H, D, T, c, var) in [(1, 32, 128, 767, True, 142), (1, 32, 128, 767, True, 42), (1, 16, 64, 577, False, 124), (1, 16, 64, 577, False, 24)]:
    q, k, v = R(B, T, H * D), R(B, T, H * D), R(B, T, H * D)
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    lib.g4r_attn_debug_variant(var)
    for _ in range(3): K.flash_attn(q, k, v, H, 1 / math.sqrt(D), c)
    lib.g4r_attn2_debug_probe(ctypes.c_void_p(buf.data_ptr()))
    K.flash_attn(q, k, v, H, 1 / math.sqrt(D), c)
    torch.cuda.synchronize()
    lib.g4r_attn2_debug_probe(ctypes.c_void_p(0))
    st = buf.cpu().tolist()
    print(f"== B{B} H{H} D{D} T{T} variant {var}")
    for g in range(2):
        s = st[g * 16:g * 16 + 16]
        print(f"   group {g}: " + "  ".join(f"{names[order[i]]} +{s[order[i]] - s[order[i - 1]]}" for i in range(1, len(order)))
              + f"  | step {s[8] - s[0]}")
lib.g4r_attn_debug_variant(0)
