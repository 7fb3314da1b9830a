"""Function-level mirror of Real3DAug/tools/cut_bbox.py (identical in both trees of the reference)
on the HIP path: ``cut_bounding_box`` (:7-68) and ``separate_bbox`` (:71-123), plus ``cut_boxes``
for all annotated objects of a frame in one call (what cut_object/cut_out.py:100-157 loops over).
The tests run in ``r3d_cut_boxes``; there is no CPU fallback.
"""
from __future__ import annotations

import numpy as np

from ... import _lib


def _box10(annotation, annotation_move):
    a = annotation
    return [a["center"]["x"] - annotation_move[0], a["center"]["y"] - annotation_move[1],
            a["center"]["z"] - annotation_move[2], a["rotation"]["x"], a["rotation"]["y"], a["rotation"]["z"],
            a["rotation"]["w"], a["length"], a["width"], a["height"]]


def cut_indices(point_cloud, boxes10, labels=None, label_col=-1, strict=True):
    """Row numbers (cloud order) of the points inside each box: list of int32 arrays."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    rows = point_cloud if isinstance(point_cloud, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(point_cloud, dtype=np.float64))
    rows = rows.to(device="cuda", dtype=torch.float64).contiguous()
    n, ld = rows.shape
    boxes = np.ascontiguousarray(boxes10, dtype=np.float64).reshape(-1, 10)
    k = len(boxes)
    if n == 0 or k == 0:
        return [np.zeros(0, dtype=np.int32) for _ in range(k)]
    d_boxes = torch.from_numpy(boxes).cuda()
    d_labels = None if labels is None else torch.from_numpy(np.ascontiguousarray(labels, dtype=np.float64)).cuda()
    counts = torch.zeros(k, dtype=torch.int32, device="cuda")
    index = torch.empty((k, n), dtype=torch.int32, device="cuda")
    ws_bytes = lib.r3d_cut_boxes_workspace_bytes(n, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    _lib.check(lib.r3d_cut_boxes(rows.data_ptr(), n, ld, label_col, d_boxes.data_ptr(),
                                 d_labels.data_ptr() if d_labels is not None else None, k, 1 if strict else 0,
                                 counts.data_ptr(), index.data_ptr(), n, ws.data_ptr(), ws_bytes, _lib.stream_ptr()),
               "r3d_cut_boxes")
    c = counts.cpu().numpy()
    idx = index.cpu().numpy()
    return [idx[b, :c[b]].copy() for b in range(k)]


def cut_bounding_box(point_cloud, annotation, annotation_move=[0, 0, 0]):
    """tools/cut_bbox.py:7-68: the rows strictly inside the annotation's box, in order."""
    pc = np.asarray(point_cloud)
    return pc[cut_indices(pc, [_box10(annotation, annotation_move)])[0]]


def separate_bbox(point_cloud, annotation, annotation_move=[0, 0, 0]):
    """tools/cut_bbox.py:71-123: (rows outside the box, rows inside it); rows on a face are inside."""
    pc = np.asarray(point_cloud)
    inside = np.zeros(len(pc), dtype=bool)
    inside[cut_indices(pc, [_box10(annotation, annotation_move)], strict=False)[0]] = True
    return pc[~inside], pc[inside]


def cut_boxes(point_cloud, annotations, classes=None, label_col=4, annotation_move=[0, 0, 0]):
    """All annotated objects of a frame at once: per annotation the rows inside its box (and, with
    ``classes``, of the object's label: cut_object/cut_out.py:123-125)."""
    pc = np.asarray(point_cloud)
    boxes = [_box10(a, annotation_move) for a in annotations]
    idx = cut_indices(pc, boxes, labels=classes, label_col=label_col if classes is not None else -1)
    return [pc[i] for i in idx]
