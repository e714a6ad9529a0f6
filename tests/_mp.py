"""Helpers for the world-size-2 gloo tests.

Results travel from the spawned ranks to the parent BY VALUE: a torch tensor put on an mp.Queue is sent as a handle to a
shared-memory file descriptor that the parent opens later -- if the worker has exited by then, the parent fails with
ConnectionResetError / FileNotFoundError (2 of 5 runs in round 3).  `plain()` turns every tensor in a nested result into a
numpy array (pickled into the pipe itself), `tensors()` turns them back."""
import numpy as np
import torch


def plain(obj):
    if isinstance(obj, torch.Tensor):
        t = obj.detach().cpu()
        # numpy has no bfloat16: widen to fp32 (exact) and narrow again on the other side
        return ("__tensor__", (t.float() if t.dtype == torch.bfloat16 else t).numpy().copy(), str(obj.dtype))
    if isinstance(obj, dict):
        return {k: plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(plain(v) for v in obj)
    return obj


def tensors(obj):
    if isinstance(obj, tuple) and len(obj) == 3 and isinstance(obj[0], str) and obj[0] == "__tensor__":
        return torch.from_numpy(np.asarray(obj[1])).to(getattr(torch, obj[2].split(".")[-1]))
    if isinstance(obj, dict):
        return {k: tensors(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(tensors(v) for v in obj)
    return obj
