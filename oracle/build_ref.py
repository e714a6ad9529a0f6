"""oracle/build_ref.py -- TEST INFRASTRUCTURE.

Compiles the reference's own CPU RoIAlign (unmodified, from where it lies under
/root/reference) into oracle/_ref/mmcv_roi_align_ref.so.  No reference source is
copied into this repository: two one-line `#include "<abs path>"` wrappers are
generated inside oracle/_ref/ (git-ignored) because both reference files are named
roi_align.cpp.  The build does not run the reference's build system.

The resulting .so travels to the GPU box with the gpurun snapshot; nothing at run
time reads /root/reference.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CSRC = "/root/reference/mmcv-1.4.7/mmcv/ops/csrc"
OUT = os.path.join(HERE, "_ref")
NAME = "mmcv_roi_align_ref"


def build(verbose=False):
    if not os.path.isdir(REF_CSRC):
        print("build_ref: /root/reference not present; keeping prebuilt oracle/_ref as is")
        return None
    os.makedirs(OUT, exist_ok=True)
    wrappers = []
    for tag, rel in (("dispatch", "pytorch/roi_align.cpp"), ("cpu", "pytorch/cpu/roi_align.cpp")):
        w = os.path.join(OUT, f"wrap_{tag}.cpp")
        text = f'#include "{os.path.join(REF_CSRC, rel)}"\n'
        if not os.path.exists(w) or open(w).read() != text:
            with open(w, "w") as f:
                f.write(text)
        wrappers.append(w)
    from torch.utils.cpp_extension import load

    mod = load(
        name=NAME,
        sources=[os.path.join(HERE, "ref_shim.cpp")] + wrappers,
        extra_include_paths=[os.path.join(REF_CSRC, "common")],
        extra_cflags=["-O2", "-ffp-contract=off"],
        build_directory=OUT,
        verbose=verbose,
    )
    return mod


if __name__ == "__main__":
    m = build(verbose="-v" in sys.argv)
    print("built" if m is not None else "skipped", os.path.join(OUT, NAME + ".so"))
