"""Seeded synthetic scans and insert candidates.

There is no network and no dataset in the build or GPU environment, so every benchmark and
parity input is generated here: a 64-beam, ~120k-point spinning-LiDAR scan in KITTI point
order (ring-major, then azimuth) with float32 x y z intensity and a uint32 semantic label --
the exact content of ``velodyne/*.bin`` + ``labels/*.label`` as the reference reads them
(SS tools/datasets.py:45-60) -- and float64 box-shaped objects standing in for the reference's
cut-out object database (``np.load(sample)['pcl']``, insertion.py:431), placed where the
reference's placement search would have put them.
"""
from __future__ import annotations

import numpy as np

SENSOR_HEIGHT = 1.73

# name -> (length, width, height, points, semantic label)
INSERT_KINDS = {
    "pedestrian": (0.6, 0.6, 1.7, 400, 30),
    "cyclist": (1.8, 0.6, 1.7, 600, 31),
    "car": (4.2, 1.8, 1.5, 1500, 10),
}


def make_scene(seed: int, n_beams: int = 64, n_az: int = 1875, shuffle: bool = False,
               collapse_labels_to_road: bool = False):
    """Return (xyzi float32 [N,4], label uint32 [N]) with N = n_beams * n_az.

    Beam elevations linspace(-24.8 deg, +2 deg) + N(0, 2e-4 rad); azimuth steps over [-pi, pi)
    + N(0, 1e-4); range = ground hit at sensor height 1.73 m for beams below -0.02 rad, else a
    40 m wall, capped at 60 m, times (1 + N(0, 0.002)).  Labels 40 (ground) / 50 (wall);
    ``collapse_labels_to_road`` applies the object-detection collapse to {40, 1}
    (OD insertion.py:353-355).
    """
    rng = np.random.default_rng(seed)
    beam = np.linspace(np.deg2rad(-24.8), np.deg2rad(2.0), n_beams)
    az0 = -np.pi + 2 * np.pi * np.arange(n_az) / n_az
    el = beam[:, None] + rng.normal(0.0, 2e-4, size=(n_beams, n_az))
    az = az0[None, :] + rng.normal(0.0, 1e-4, size=(n_beams, n_az))
    ground = beam[:, None] < -0.02
    with np.errstate(divide="ignore"):
        rng_ground = SENSOR_HEIGHT / np.sin(-np.minimum(el, -1e-3))
    rad = np.where(ground, np.minimum(rng_ground, 60.0), 40.0)
    rad = rad * (1.0 + rng.normal(0.0, 0.002, size=rad.shape))
    x = rad * np.cos(el) * np.cos(az)
    y = rad * np.cos(el) * np.sin(az)
    z = rad * np.sin(el)
    inten = rng.random(size=rad.shape, dtype=np.float32)
    label = np.where(np.broadcast_to(ground, rad.shape), 40, 50).astype(np.uint32)
    xyzi = np.stack([x.astype(np.float32), y.astype(np.float32), z.astype(np.float32), inten],
                    axis=-1).reshape(-1, 4)
    label = label.reshape(-1)
    if collapse_labels_to_road:
        label = np.where(label == 40, 40, 1).astype(np.uint32)
    if shuffle:
        perm = rng.permutation(len(xyzi))
        xyzi, label = xyzi[perm], label[perm]
    return np.ascontiguousarray(xyzi), np.ascontiguousarray(label)


def make_insert(seed: int, kind: str = "pedestrian", rng_range=(5.0, 30.0), points: int | None = None,
                centre_range: float | None = None, centre_az: float | None = None):
    """Return an M x 5 float64 array (x y z intensity label) for one placed object.

    Points lie on the surface of an axis-rotated box resting on the ground plane
    z = -SENSOR_HEIGHT, centre at range U[5, 30] m and azimuth U[-pi, pi); coordinates are
    genuine float64 (the reference rotates samples by accumulated float64 rotations,
    find_spot.py:233), intensity is float32-representable like the object database's.
    """
    length, width, height, m, label = INSERT_KINDS[kind]
    if points is not None:
        m = points
    rng = np.random.default_rng(seed)
    dist = rng.uniform(*rng_range) if centre_range is None else centre_range
    phi = rng.uniform(-np.pi, np.pi) if centre_az is None else centre_az
    yaw = rng.uniform(-np.pi, np.pi)
    # surface sampling: choose a face by area, then a uniform point on it
    dims = np.array([length, width, height])
    areas = np.array([dims[1] * dims[2], dims[0] * dims[2], dims[0] * dims[1]])
    face_axis = rng.choice(3, size=m, p=areas / areas.sum())
    side = rng.integers(0, 2, size=m) * 2.0 - 1.0
    p = rng.uniform(-0.5, 0.5, size=(m, 3)) * dims
    p[np.arange(m), face_axis] = side * dims[face_axis] / 2
    c, s = np.cos(yaw), np.sin(yaw)
    xr = c * p[:, 0] - s * p[:, 1] + dist * np.cos(phi)
    yr = s * p[:, 0] + c * p[:, 1] + dist * np.sin(phi)
    zr = p[:, 2] + height / 2 - SENSOR_HEIGHT
    inten = rng.random(size=m, dtype=np.float32).astype(np.float64)
    return np.stack([xr, yr, zr, inten, np.full(m, float(label))], axis=-1)


def make_inserts(scene_seed: int, kinds):
    """K inserts for one scene, seeds derived from the scene seed (one accepted candidate each)."""
    return [make_insert(scene_seed * 1000 + 17 * k + 1, kind) for k, kind in enumerate(kinds)]


# The insert mixes of BASELINE.json's configs.
CONFIG_INSERTS = {
    "C1": ["pedestrian"],
    "C2": ["pedestrian", "cyclist", "car", "pedestrian", "cyclist"],
    "C3": ["car", "pedestrian", "cyclist", "car", "pedestrian", "cyclist", "car", "pedestrian",
           "cyclist", "pedestrian"],
    "C4": ["pedestrian", "cyclist", "car", "pedestrian", "cyclist", "car", "pedestrian", "cyclist"],
}


def scene5_from_packed(xyzi: np.ndarray, label: np.ndarray) -> np.ndarray:
    """float32 xyzi + uint32 label -> the N x 5 float64 array the reference's ``__getitem__``
    hands to the driver (SS tools/datasets.py:51-56: hstack promotes to float64)."""
    return np.hstack((xyzi.astype(np.float64), (label & 0xFFFF).astype(np.float64)[:, None]))


# ---- placement-search inputs ----------------------------------------------
# semantic-kitti.yaml:22-34 for the three classes of INSERT_KINDS
PLACEMENT = {18: [1, 3], 30: [2], 31: [1, 3]}
PLACEMENT_LABELS = {1: [40, 60], 2: [48], 3: [44]}
KIND_CLASS = {"pedestrian": 30, "cyclist": 31, "car": 18}


def make_place_frame(seed: int, n_boxes: int = 6, n_beams: int = 64, n_az: int = 1875):
    """One frame for the placement search: the scan of ``make_scene`` with part of its ground
    relabelled sidewalk (48, y > 4 m) and parking (44, x < -8 m), a pose, the 1 m rich map
    rasterised from the ground labels (road 1, sidewalk 2, parking 3) with its ``move`` and
    ``n_boxes`` annotated objects.  Returns a dict: original (N x 5 float64), rich (uint8), move
    (4 x 1), pose (4 x 4), boxes (k x 10: centre, quaternion xyzw, length, width, height)."""
    xyzi, label = make_scene(seed, n_beams, n_az)
    label = label.copy()
    ground = label == 40
    label[ground & (xyzi[:, 1] > 4.0)] = 48
    label[ground & (xyzi[:, 0] < -8.0) & (xyzi[:, 1] <= 4.0)] = 44
    original = scene5_from_packed(xyzi, label)
    pose = np.eye(4)
    pose[:3, 3] = [500.5 + seed, -200.25, 1.7]
    half = 70
    move = np.array([[int(np.floor(pose[0, 3])) - half], [int(np.floor(pose[1, 3])) - half], [0], [1]])
    rich = np.zeros((2 * half + 1, 2 * half + 1), dtype=np.uint8)
    world = (pose @ np.hstack((original[:, :3], np.ones((len(original), 1)))).T - move).astype(int)
    inside = (world[0] >= 0) & (world[0] < rich.shape[0]) & (world[1] >= 0) & (world[1] < rich.shape[1])
    for value, labels in ((1, (40,)), (2, (48,)), (3, (44,))):
        sel = inside & np.isin(original[:, 4], labels)
        rich[world[0][sel], world[1][sel]] = value
    rng = np.random.default_rng(seed)
    boxes = []
    for ang in rng.uniform(-np.pi, np.pi, size=n_boxes):
        d = rng.uniform(6, 25)
        boxes.append([d * np.cos(ang), d * np.sin(ang), -SENSOR_HEIGHT, 0, 0, np.sin(ang / 2), np.cos(ang / 2), 4.2, 1.8, 1.5])
    return {"xyzi": xyzi, "label": label, "original": original, "rich": rich, "move": move, "pose": pose,
            "boxes": np.asarray(boxes, dtype=np.float64).reshape(-1, 10)}


def make_place_sample(seed: int, kind: str):
    """A sample of the object database for the placement search: (M x 5 points, label line
    'class x y z height length width rot_z' as cut_object/cut_out.py:113-121 writes it)."""
    smp = make_insert(seed, kind)
    length, width, height, _, _ = INSERT_KINDS[kind]
    centre = [smp[:, 0].mean(), smp[:, 1].mean(), smp[:, 2].min()]
    line = " ".join([str(KIND_CLASS[kind])] + [repr(float(v)) for v in (*centre, height, length, width, 0.3)])
    return smp, line
