"""Rich-map rasterisation on the GPU (include/real3daug_hip.h, r3d_map_*):
what the ``__main__`` block of semantic_segmentation/rich_map/drivable_area_map.py:122-206 does for
one sequence -- world extremes of all frames, then the map cell under every placement-surface
point (road 1 / sidewalk 2 overwrite each other in frame and point order, parking 3 stays).
No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _decode(key):
    key = int(key)
    u = key & 0x7FFFFFFFFFFFFFFF if key >> 63 else (~key) & 0xFFFFFFFFFFFFFFFF
    return float(np.frombuffer(np.uint64(u).tobytes(), dtype=np.float64)[0])


def build_rich_map(frames, placement_labels, device="cuda:0", as_uint8=False):
    with _lib.on(device):
        return _build_rich_map(frames, placement_labels, device, as_uint8)


def _build_rich_map(frames, placement_labels, device, as_uint8):
    """frames: sequence of (xyzi float32 [n,4], label uint32 [n], transform_matrix 4x4) in processing
    order (tools/datasets.py:45-60 per frame); placement_labels: config['insertion']
    ['placement_labels'] ({1: road, 2: sidewalk, 3: parking}).  Returns (map, move) as the
    reference stores them: map float64 [size_x, size_y] (uint8 with ``as_uint8``), move (4, 1)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    frames = list(frames)
    dev = []
    for xyzi, label, t in frames:
        dev.append((torch.from_numpy(np.ascontiguousarray(xyzi, dtype=np.float32)).to(device),
                    torch.from_numpy(np.ascontiguousarray(label, dtype=np.uint32).view(np.int32)).to(device),
                    (C.c_double * 16)(*np.asarray(t, dtype=np.float64).reshape(16))))
    minmax = torch.tensor([-1, 0, -1, 0], dtype=torch.int64, device=device)       # {~0, 0, ~0, 0}
    for x, _, t in dev:
        _lib.check(lib.r3d_map_bounds(x.data_ptr(), x.shape[0], t, minmax.data_ptr(), _lib.stream_ptr()), "r3d_map_bounds")
    mm = [_decode(int(v) & 0xFFFFFFFFFFFFFFFF) for v in minmax.cpu().numpy()]
    min_x, min_y = int(np.floor(mm[0])), int(np.floor(mm[2]))                       # :160-161
    size_x, size_y = int(mm[1]) + 1 - min_x, int(mm[3]) + 1 - min_y                 # :163-168
    keys = torch.zeros(size_x * size_y, dtype=torch.int64, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    lists = [(C.c_int32 * len(placement_labels[c]))(*placement_labels[c]) for c in (1, 2, 3)]
    for f, (x, lab, t) in enumerate(dev):
        _lib.check(lib.r3d_map_splat(x.data_ptr(), lab.data_ptr(), x.shape[0], t, lists[0], len(lists[0]), lists[1],
                                     len(lists[1]), lists[2], len(lists[2]), float(min_x), float(min_y), size_x, size_y,
                                     f, keys.data_ptr(), status.data_ptr(), _lib.stream_ptr()), "r3d_map_splat")
    out = torch.empty((size_x, size_y), dtype=torch.uint8 if as_uint8 else torch.float64, device=device)
    _lib.check(lib.r3d_map_finish(keys.data_ptr(), size_x * size_y, None if as_uint8 else out.data_ptr(),
                                  out.data_ptr() if as_uint8 else None, _lib.stream_ptr()), "r3d_map_finish")
    if int(status.item()):
        raise AssertionError("Indexing error: a surface point outside the map (drivable_area_map.py:180)")
    return out.cpu().numpy(), np.array([[min_x], [min_y], [0], [1]])


def build_od_maps(xyzi, label, road_label, device="cuda:0"):
    """One frame of object_detection/rich_map/single_drivable_area_map.py:113-194: (road_map uint8
    [size_x, size_y], pedestrian_map uint8, min_x, min_y) as its two np.savez calls store them (:157, :193).
    xyzi float32 [n,4], label uint32 [n] as read from the frame's files (tools/datasets.py:56-62)."""
    with _lib.on(device):
        torch = _lib.require_gpu()
        lib = _lib.load()
        x = torch.from_numpy(np.ascontiguousarray(xyzi, dtype=np.float32)).to(device)
        lab = torch.from_numpy(np.ascontiguousarray(label, dtype=np.uint32).view(np.int32)).to(device)
        ident = (C.c_double * 16)(*np.eye(4).reshape(16))
        minmax = torch.tensor([-1, 0, -1, 0], dtype=torch.int64, device=device)
        _lib.check(lib.r3d_map_bounds(x.data_ptr(), x.shape[0], ident, minmax.data_ptr(), _lib.stream_ptr()), "r3d_map_bounds")
        mm = [_decode(int(v) & 0xFFFFFFFFFFFFFFFF) for v in minmax.cpu().numpy()]
        min_x, min_y = int(mm[0]), int(mm[2])                                           # :118-119 (truncation)
        size_x, size_y = int(mm[1]) + 1 - min_x, int(mm[3]) + 1 - min_y                 # :121-127
        road = torch.empty((size_x, size_y), dtype=torch.uint8, device=device)
        ped = torch.empty((size_x, size_y), dtype=torch.uint8, device=device)
        scratch = torch.empty(2 * size_x * size_y, dtype=torch.uint8, device=device)
        _lib.check(lib.r3d_od_maps(x.data_ptr(), lab.data_ptr(), x.shape[0], int(road_label), min_x, min_y, size_x, size_y,
                                   road.data_ptr(), ped.data_ptr(), scratch.data_ptr(), _lib.stream_ptr()), "r3d_od_maps")
        return road.cpu().numpy(), ped.cpu().numpy(), min_x, min_y
