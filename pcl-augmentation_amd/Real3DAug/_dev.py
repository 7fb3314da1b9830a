"""numpy <-> device plumbing shared by the function-level mirrors."""
from __future__ import annotations

import numpy as np

from .. import _lib


def to_device(arr, dtype=None):
    """(tensor on cuda, True if the caller handed a numpy array)."""
    torch = _lib.require_gpu()
    if isinstance(arr, torch.Tensor):
        t = arr if arr.is_cuda else arr.cuda()
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous(), False
    a = np.ascontiguousarray(arr, dtype=np.float64 if dtype is None else None)
    t = torch.from_numpy(a).cuda()
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t, True


def back(t, was_numpy):
    return t.cpu().numpy() if was_numpy else t


def ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() else 0) if t is not None else None
