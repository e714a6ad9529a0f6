"""ctypes binding of libgpt4roi_hip.so (the C ABI declared in include/*.h).

The product path has no CPU fallback: if the HIP library is missing or a tensor is not on
the GPU, calls raise.  Build with `python -m gpt4roi_amd.build`.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("G4R_LIB") or os.path.join(HERE, "lib", "libgpt4roi_hip.so")     # (G4R_LIB: tools, a library built with G4R_BUILD_TAG)
ABI_VERSION = 5
_lib = None


class HipKernelError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipKernelError(
                f"{LIB_PATH} not found: the gfx950 kernels are not built "
                "(run `python -m gpt4roi_amd.build`); there is no CPU fallback")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.g4r_last_error.restype = ctypes.c_char_p
        v = _lib.g4r_abi_version()
        if v != ABI_VERSION:
            raise HipKernelError(f"libgpt4roi_hip.so ABI {v} != expected {ABI_VERSION}; rebuild")
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().g4r_last_error().decode("utf-8", "replace")
        raise HipKernelError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipKernelError(
                "gpt4roi_amd ops run only on an MI355X device tensor; got a CPU tensor "
                "(no CPU fallback exists by design)")
