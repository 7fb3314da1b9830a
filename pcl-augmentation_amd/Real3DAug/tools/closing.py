"""Mirror of the reference's ``Real3DAug/tools/closing.py`` (same names, HIP underneath)."""
from __future__ import annotations

from ... import _lib
from .._dev import back, ptr, to_device


def class_closing(original_label):
    """SS Real3DAug/tools/closing.py:9-23: uint8 closing of clip(label, 0, 1) with rectangle(5, 3)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    lab, was_np = to_device(original_label, torch.float64)
    rows, cols = lab.shape
    closed = torch.empty((rows, cols), dtype=torch.uint8, device=lab.device)
    _lib.check(lib.r3d_class_closing(ptr(lab), rows, cols, ptr(closed), _lib.stream_ptr()), "class_closing")
    return back(closed, was_np)


def smooth_out(original_train, original_label):
    """SS Real3DAug/tools/closing.py:26-62: returns (train, label) with closed holes filled."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    tr, was_np = to_device(original_train, torch.float64)
    lab, _ = to_device(original_label, torch.float64)
    rows, cols = lab.shape
    tr_out = torch.empty_like(tr)
    lab_out = torch.empty_like(lab)
    _lib.check(lib.r3d_smooth_out(ptr(tr), ptr(lab), rows, cols, ptr(tr_out), ptr(lab_out), _lib.stream_ptr()),
               "smooth_out")
    return back(tr_out, was_np), back(lab_out, was_np)
