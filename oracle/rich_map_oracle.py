"""CPU restatement of the rich-map rasterisation (TEST INFRASTRUCTURE ONLY).

Follows the ``__main__`` block of semantic_segmentation/rich_map/drivable_area_map.py:122-206 for one
sequence.  Pinned by tests/golden/rich_map.npz, the map the reference's own script wrote for a small
synthetic sequence (tests/golden/make_golden_map.py).  The 4x4 product ``t_matrix @ points.T`` goes
through BLAS; its accumulation order is restated as in oracle/find_spot_oracle.py.
"""
import numpy as np

from .find_spot_oracle import blas_matmul


def world_points(xyzi, transform):
    """drivable_area_map.py:137-141: homogeneous transform, then the division by the last row."""
    hom = np.hstack((xyzi[:, :3].astype(np.float64), np.ones((len(xyzi), 1))))
    p = blas_matmul(np.asarray(transform, dtype=np.float64), hom.T)
    return (p / p[3, :]).T


def build_rich_map(frames, labels_road, labels_sidewalk, labels_parking):
    """frames: list of (xyzi float32 [n,4], label uint32 [n], transform 4x4).  Returns
    (map float64 [size_x, size_y], move int [4,1]) as np.savez stores them (:205-206)."""
    max_x = max_y = -np.inf
    min_x = min_y = np.inf
    for xyzi, _, t in frames:                                              # :129-150
        w = world_points(xyzi, t)
        max_x, min_x = max(max_x, w[:, 0].max()), min(min_x, w[:, 0].min())
        max_y, min_y = max(max_y, w[:, 1].max()), min(min_y, w[:, 1].min())
    min_x, min_y = int(np.floor(min_x)), int(np.floor(min_y))              # :160-161
    size_x, size_y = int(max_x) + 1 - min_x, int(max_y) + 1 - min_y        # :163-168
    area = np.zeros((size_x, size_y))
    for xyzi, label, t in frames:                                          # :172-200, point by point
        w = world_points(xyzi, t)
        sem = (label & 0xFFFF).astype(np.int64)
        for i in range(len(w)):
            lab = sem[i]
            if not (lab in labels_road or lab in labels_sidewalk or lab in labels_parking):
                continue
            px, py = int(w[i, 0] - min_x), int(w[i, 1] - min_y)
            assert w[i, 0] - min_x >= 0 and w[i, 1] - min_y >= 0
            if lab in labels_road:
                if area[px][py] != 3:
                    area[px][py] = 1
            elif lab in labels_parking:
                area[px][py] = 3
            elif area[px][py] != 3:
                area[px][py] = 2
    return area, np.array([[min_x], [min_y], [0], [1]])


# ---- object-detection flavour (object_detection/rich_map/single_drivable_area_map.py:113-194) ---------------------
def _disk_offsets(radius):
    return [(dr, dc) for dr in range(-radius, radius + 1) for dc in range(-radius, radius + 1) if dr * dr + dc * dc <= radius * radius]


def _morph(img, radius, erode):
    """Binary dilation / erosion with disk(radius), windows clipped at the borders (scikit-image's grey
    closing / dilation with a symmetric convex footprint and reflecting borders)."""
    sx, sy = img.shape
    out = np.ones_like(img) if erode else np.zeros_like(img)
    for dr, dc in _disk_offsets(radius):
        r0, r1, c0, c1 = max(0, -dr), min(sx, sx - dr), max(0, -dc), min(sy, sy - dc)
        if r0 >= r1 or c0 >= c1:
            continue
        src = img[r0 + dr:r1 + dr, c0 + dc:c1 + dc]
        if erode:
            out[r0:r1, c0:c1] &= src
        else:
            out[r0:r1, c0:c1] |= src
    return out


def od_maps(xyzi, label, road_label):
    """One frame: (road_map uint8, pedestrian_map uint8, min_x, min_y) as the script saves them (:157, :193)."""
    pc = xyzi[:, :3].astype(np.float64)
    sem = (label & 0xFFFF).astype(np.int64)
    min_x, min_y = int(pc[:, 0].min()), int(pc[:, 1].min())                               # :118-119
    size_x, size_y = int(pc[:, 0].max()) + 1 - min_x, int(pc[:, 1].max()) + 1 - min_y      # :121-127
    raster = np.zeros((size_x, size_y), dtype=np.uint8)
    road = pc[sem == road_label]
    raster[(road[:, 0] - min_x).astype(np.int64), (road[:, 1] - min_y).astype(np.int64)] = 1   # :129-139 (int() truncates)
    closed = _morph(_morph(raster, 4, False), 4, True)                                     # :145-151
    ring = np.zeros_like(closed)
    near = _morph(closed, 1, False) | np.pad(closed, 1)[:-2, :-2] | np.pad(closed, 1)[:-2, 2:] | np.pad(closed, 1)[2:, :-2] | np.pad(closed, 1)[2:, 2:]
    ring[(closed == 0) & (near == 1)] = 1                                                  # :160-178 (8-neighbourhood)
    return closed, _morph(ring, 2, False), min_x, min_y                                    # :180-188
