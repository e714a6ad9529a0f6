"""Builds gpt4roi_amd/lib/libgpt4roi_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m gpt4roi_amd.build [-v] [--force]

One object per .hip file under gpt4roi_amd/csrc (rebuilt when the source or any header is
newer), linked into ONE shared library with a plain C ABI (include/*.h).  The files of the inference path
(F16_SOURCES) are compiled a second time with -DG4R_F16: the same kernels storing IEEE half instead of bfloat16 -- the
reference's serving dtype (gpt4roi/app.py:74-98) -- under the entry-point names of include/g4r_f16_names.h.  The .so is built
in-tree so that it travels with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
_TAG = os.environ.get("G4R_BUILD_TAG", "")        # tools: a second library next to the shipped one (A/B of compile-time switches), see _lib.py G4R_LIB
OBJ = os.path.join(HERE, "lib", "obj" + ("_" + _TAG if _TAG else ""))
LIB = os.path.join(HERE, "lib", "libgpt4roi_hip" + ("_" + _TAG if _TAG else "") + ".so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(os.path.dirname(HERE), "include"),
         "-I", CSRC] + os.environ.get("G4R_EXTRA_HIPCC_FLAGS", "").split()      # tools: e.g. -DG4R_ATTN2_PROBE


# the inference path (ViT, region module, projector, splice, LLaMA prefill + decode); training-only files stay bf16
F16_SOURCES = ("gemm_bf16.hip", "gemv_mfma.hip", "attention.hip", "attention_v2.hip", "norm.hip", "elementwise.hip", "roi_align.hip")
F16_NAMES = os.path.join(os.path.dirname(HERE), "include", "g4r_f16_names.h")


def _newest_header():
    t = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr_t = _newest_header()
    jobs, objs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([HIPCC, *FLAGS, "-c", src, "-o", obj])
        if s in F16_SOURCES:
            obj16 = os.path.join(OBJ, s[:-4] + ".f16.o")
            objs.append(obj16)
            if force or not os.path.exists(obj16) or os.path.getmtime(obj16) < max(os.path.getmtime(src), hdr_t):
                jobs.append([HIPCC, *FLAGS, "-DG4R_F16", "-include", F16_NAMES, "-c", src, "-o", obj16])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv, force="--force" in sys.argv))
