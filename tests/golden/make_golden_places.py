#!/usr/bin/env python3
"""Golden fixtures for the placement search, made by running the
REFERENCE's own ``find_possible_places`` (SS tools/find_spot.py:192-273) in this container:

    python tests/golden/make_golden_places.py            # semantic_segmentation tree
    python tests/golden/make_golden_places.py --od       # object_detection tree (OD tools/find_spot.py:227-304)
    python tests/golden/make_golden_places.py --cut      # tools/cut_bbox.py: cut_bounding_box, separate_bbox

``/root/reference`` is read-only and never travels to the GPU box; only arrays are written.
Two API renames stand between the reference and today's libraries and are bridged here, in the
generator only:

* ``np.int`` (find_spot.py:238) was removed in NumPy 1.24 -> bound to ``int`` (what it aliased);
* ``Rotation.as_dcm`` / ``from_dcm`` (find_spot.py:55,63,179; cut_bbox.py:28) were renamed
  ``as_matrix`` / ``from_matrix`` in SciPy 1.4 and removed in 1.6 -> the modules' ``R`` is
  replaced by a thin class forwarding the old names to the new ones.

Everything else (rotation chain, map test, height correction, collision tests, ``deepcopy`` of
the outputs) is the reference's code, unmodified.  Inputs are synthetic (tests/golden has no
dataset): a small spinning-LiDAR scene with road / sidewalk / parking labels, a rich map
rasterised from it, a tilted pose, annotated boxes and an earlier insert standing on the circle
the sample is rotated along, so that every outcome of the search occurs.
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/semantic_segmentation/Real3DAug"
sys.path.insert(0, ROOT)

PLACEMENT = {11: [1, 3], 15: [1, 3], 18: [1, 3], 30: [2], 31: [1, 3], 32: [1, 3], 253: [1, 3], 255: [1, 3]}
PLACEMENT_LABELS = {1: [40, 60], 2: [48], 3: [44]}      # semantic-kitti.yaml:22-34


def import_reference():
    from scipy.spatial.transform import Rotation

    class OldRotation:
        """SciPy < 1.4 names on today's Rotation."""

        def __init__(self, r):
            self._r = r

        @staticmethod
        def from_quat(q):
            return OldRotation(Rotation.from_quat(q))

        @staticmethod
        def from_dcm(m):
            return OldRotation(Rotation.from_matrix(m))

        def as_dcm(self):
            return self._r.as_matrix()

        def as_quat(self):
            return self._r.as_quat()

    if not hasattr(np, "int"):
        np.int = int
    sys.path.insert(0, REF)
    from tools import cut_bbox, find_spot      # noqa: E402  (reference modules)
    find_spot.R = OldRotation
    cut_bbox.R = OldRotation
    return find_spot


def label_line(cls, centre, height, length, width, yaw):
    """cut_object/cut_out.py:113-121 / find_spot.py:161-172: 'class x y z height length width rot_z'."""
    return " ".join([str(cls)] + [repr(float(v)) for v in (*centre, height, length, width, yaw)])


def build_case(seed, cls, beams=32, n_az=600, m_points=60, tilt=True, small_map=False):
    synth = __import__("importlib").import_module("pcl-augmentation_amd.synth")
    rng = np.random.default_rng(seed)
    xyzi, label = synth.make_scene(seed, beams, n_az)
    label = label.copy()
    ground = label == 40
    label[ground & (xyzi[:, 1] > 5.0)] = 48                      # sidewalk
    label[ground & (xyzi[:, 0] < -9.0) & (xyzi[:, 1] <= 5.0)] = 44   # parking
    label[ground & (xyzi[:, 0] > 14.0) & (xyzi[:, 1] <= 5.0)] = 60   # lane marking (road)
    # a hole in the ground so that some placements have no surface point nearby
    hole = ground & ((xyzi[:, 0] - 2.0) ** 2 + (xyzi[:, 1] + 9.0) ** 2 < 5.2 ** 2)
    label[hole] = 70
    original = synth.scene5_from_packed(xyzi, label)

    # pose and rich map (1 m cells, world frame)
    T = np.eye(4)
    if tilt:
        a, b, c = 0.31, 0.012, -0.008
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
        T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [412.37, -128.61, 1.9]
    half = 14 if small_map else 70
    move = np.array([[int(np.floor(T[0, 3])) - half], [int(np.floor(T[1, 3])) - half], [0], [1]])
    rich = np.zeros((2 * half + 1, 2 * half + 1))
    world = (T @ np.hstack((original[:, :3], np.ones((len(original), 1)))).T - move).astype(int)
    inside = (world[0] >= 0) & (world[0] < rich.shape[0]) & (world[1] >= 0) & (world[1] < rich.shape[1])
    for value, labels in ((1, (40, 60)), (2, (48,)), (3, (44,)), (2, (70,))):   # the hole counts as sidewalk
        sel = inside & np.isin(original[:, 4], labels)
        rich[world[0][sel], world[1][sel]] = value

    # the sample: points on a box, annotation = its bottom centre, standing 0.13 m too high
    kind = {30: "pedestrian", 31: "cyclist", 18: "car"}[cls]
    length, width, height, _, _ = synth.INSERT_KINDS[kind]
    dist, phi, yaw = 9.0 + rng.uniform(-1, 1), rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi, np.pi)
    p = rng.uniform(-0.5, 0.5, size=(m_points, 3)) * [length, width, height]
    centre = np.array([dist * np.cos(phi), dist * np.sin(phi), -synth.SENSOR_HEIGHT + 0.13])
    cy, sy = np.cos(yaw), np.sin(yaw)
    pts = np.stack([cy * p[:, 0] - sy * p[:, 1] + centre[0], sy * p[:, 0] + cy * p[:, 1] + centre[1],
                    p[:, 2] + height / 2 + centre[2]], axis=1)
    sample = np.column_stack([pts, rng.random(m_points, dtype=np.float32).astype(np.float64), np.full(m_points, float(cls))])
    sample_line = label_line(cls, centre, height, length, width, yaw)

    # annotated scene objects on the sample's circle + an earlier insert (points with label 10)
    anno_lines, extra = [], []
    for k, ang in enumerate(rng.uniform(-np.pi, np.pi, size=3)):
        c = np.array([dist * np.cos(ang), dist * np.sin(ang), -synth.SENSOR_HEIGHT])
        anno_lines.append(label_line(10, c, 1.5, 4.2, 1.8, rng.uniform(-np.pi, np.pi)))
    ang = rng.uniform(-np.pi, np.pi)
    blob_c = np.array([dist * np.cos(ang), dist * np.sin(ang), -synth.SENSOR_HEIGHT + 0.6])
    blob = blob_c + rng.normal(0, 0.25, size=(80, 3))
    extra = np.column_stack([blob, rng.random(80), np.full(80, 10.0)])

    scene5 = np.vstack([original, extra])
    return dict(xyzi=xyzi, label=label, extra=extra, sample=sample, sample_line=sample_line, anno_lines=anno_lines,
                rich=rich.astype(np.uint8), move=move, T=T, original=original, scene5=scene5)


def run_case(fs, case):
    from oracle import real3d_oracle as O
    scene9 = O.add_space_for_spherical(case["scene5"])                 # insertion.py:362: N x 9, label in column 7
    annos = [fs.read_label_line(l) for l in case["anno_lines"]]
    config = {"insertion": {"placement": PLACEMENT, "placement_labels": PLACEMENT_LABELS}}
    sample_data = {"pcl": case["sample"].copy(), "anno": np.array(case["sample_line"])}
    pcl, anno, rot = fs.find_possible_places(scene9, annos, sample_data, case["rich"].astype(np.float64), case["move"],
                                             case["original"].copy(), case["T"], config)
    m = len(case["sample"])
    return dict(out_rot=np.array(rot, dtype=np.int32),
                out_pcl=np.array(pcl, dtype=np.float64).reshape(len(rot), m, 5),
                out_centre=np.array([[a["center"]["x"], a["center"]["y"], a["center"]["z"]] for a in anno]).reshape(len(rot), 3),
                out_quat=np.array([[a["rotation"]["x"], a["rotation"]["y"], a["rotation"]["z"], a["rotation"]["w"]]
                                   for a in anno]).reshape(len(rot), 4))


CASES = {
    "places_cyclist": dict(seed=11, cls=31),
    "places_pedestrian": dict(seed=12, cls=30),
    "places_car_smallmap": dict(seed=13, cls=18, small_map=True, tilt=False),
    "places_one_point": dict(seed=12, cls=30, m_points=1),      # one column: numpy takes the matrix x vector routine (:72, :236)
}


REF_OD = "/root/reference/object_detection/Real3DAug"


def import_reference_od():
    """The object-detection copy (its ``tools`` is a namespace package with relative imports);
    run in its own process (``--od``): both trees call their package ``tools``."""
    from scipy.spatial.transform import Rotation

    class OldRotation:
        def __init__(self, r):
            self._r = r

        @staticmethod
        def from_quat(q):
            return OldRotation(Rotation.from_quat(q))

        @staticmethod
        def from_dcm(m):
            return OldRotation(Rotation.from_matrix(m))

        def as_dcm(self):
            return self._r.as_matrix()

        def as_quat(self):
            return self._r.as_quat()

    sys.path.insert(0, REF_OD)
    from tools import cut_bbox, find_spot      # noqa: E402
    find_spot.R = OldRotation
    cut_bbox.R = OldRotation
    return find_spot


def kitti_line(cls, centre_lidar, height, length, width, yaw_lidar):
    """A label_2 line whose LiDAR-frame reading (OD find_spot.py:186-224) is the given box, up to the
    constant offsets the reference adds (0.1 m on the sizes, 0.27 / -0.08 m on the centre)."""
    x_cam, y_cam, z_cam = -centre_lidar[1], -(centre_lidar[2] + 0.08), centre_lidar[0] - 0.27
    vals = [0.0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, height - 0.1, length - 0.1, width - 0.1, x_cam, y_cam, z_cam, -yaw_lidar]
    return " ".join([cls] + [repr(float(v)) if not isinstance(v, int) else str(v) for v in vals])


def build_case_od(seed, cls, beams=32, n_az=600, m_points=60):
    synth = __import__("importlib").import_module("pcl-augmentation_amd.synth")
    rng = np.random.default_rng(seed)
    xyzi, label = synth.make_scene(seed, beams, n_az, collapse_labels_to_road=True)          # labels {40, 1}
    label = label.copy()
    hole = (label == 40) & ((xyzi[:, 0] - 2.0) ** 2 + (xyzi[:, 1] + 9.0) ** 2 < 5.3 ** 2)
    label[hole] = 1                                                     # no road points there
    original = synth.scene5_from_packed(xyzi, label)
    half = 45
    move = np.array([-half + 0.37, -half - 0.21])
    rich = np.zeros((2 * half, 2 * half))
    g = (original[:, :2] - move)
    ok = (original[:, 4] == 40) | hole                                  # the hole still counts as road in the map
    gi = g[ok].astype(int)
    keep = (gi[:, 0] >= 0) & (gi[:, 0] < rich.shape[0]) & (gi[:, 1] >= 0) & (gi[:, 1] < rich.shape[1])
    rich[gi[keep, 0], gi[keep, 1]] = 1
    rich[:, :30] = 0                                                    # a strip that is not road
    kind = {"Pedestrian": "pedestrian", "Cyclist": "cyclist", "Car": "car"}[cls]
    length, width, height, _, _ = synth.INSERT_KINDS[kind]
    dist, phi, yaw = 9.0 + rng.uniform(-1, 1), rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi, np.pi)
    p = rng.uniform(-0.5, 0.5, size=(m_points, 3)) * [length, width, height]
    centre = np.array([dist * np.cos(phi), dist * np.sin(phi), -synth.SENSOR_HEIGHT + 0.13])
    cy, sy = np.cos(yaw), np.sin(yaw)
    pts = np.stack([cy * p[:, 0] - sy * p[:, 1] + centre[0], sy * p[:, 0] + cy * p[:, 1] + centre[1],
                    p[:, 2] + height / 2 + centre[2]], axis=1)
    sample = np.column_stack([pts, rng.random(m_points, dtype=np.float32).astype(np.float64), np.full(m_points, 7.0)])
    sample_line = kitti_line(cls, centre, height, length, width, yaw)
    anno_lines = []
    for ang in rng.uniform(-np.pi, np.pi, size=3):
        c = np.array([dist * np.cos(ang), dist * np.sin(ang), -synth.SENSOR_HEIGHT])
        anno_lines.append(kitti_line("Car", c, 1.5, 4.2, 1.8, rng.uniform(-np.pi, np.pi)))
    ang = rng.uniform(-np.pi, np.pi)
    blob = np.array([dist * np.cos(ang), dist * np.sin(ang), -synth.SENSOR_HEIGHT + 0.6]) + rng.normal(0, 0.25, size=(80, 3))
    extra = np.column_stack([blob, rng.random(80), np.full(80, 1.0)])
    return dict(xyzi=xyzi, label=label, extra=extra, sample=sample, sample_line=sample_line, anno_lines=anno_lines,
                rich=rich.astype(np.uint8), move=move, original=original, scene5=np.vstack([original, extra]))


def run_case_od(fs, case):
    from oracle import real3d_oracle as O
    scene9 = O.add_space_for_spherical(case["scene5"])
    annos = [fs.read_label_line(l) for l in case["anno_lines"]]
    config = {"labels": {"Road": 40}}
    map_data = {"map": case["rich"].astype(np.float64), "min_x": case["move"][0], "min_y": case["move"][1]}
    sample_data = {"pcl": case["sample"].copy(), "anno": np.array(case["sample_line"])}
    pcl, anno, rot = fs.find_possible_places(scene9, annos, sample_data, map_data, case["original"].copy(), config)
    m = len(case["sample"])
    return dict(out_rot=np.array(rot, dtype=np.int32),
                out_pcl=np.array(pcl, dtype=np.float64).reshape(len(rot), m, 5),
                out_centre=np.array([[a["center"]["x"], a["center"]["y"], a["center"]["z"]] for a in anno]).reshape(len(rot), 3),
                out_quat=np.array([[a["rotation"]["x"], a["rotation"]["y"], a["rotation"]["z"], a["rotation"]["w"]]
                                   for a in anno]).reshape(len(rot), 4))


CASES_OD = {"places_od_car": dict(seed=21, cls="Car"), "places_od_pedestrian": dict(seed=22, cls="Pedestrian")}


def main_od():
    fs = import_reference_od()
    for name, kw in CASES_OD.items():
        case = build_case_od(**kw)
        out = run_case_od(fs, case)
        keep = {k: case[k] for k in ("xyzi", "label", "extra", "sample", "rich", "move")}
        keep["sample_line"] = np.array(case["sample_line"])
        keep["anno_lines"] = np.array(case["anno_lines"])
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **keep, **out)
        print(name, "possible:", len(out["out_rot"]), "first:", out["out_rot"][:8])


def main_cut():
    """cut_bounding_box / separate_bbox of tools/cut_bbox.py on a cloud with points on the faces."""
    fs = import_reference()
    from tools import cut_bbox
    rng = np.random.default_rng(5)
    n = 2500
    pc = np.column_stack([rng.uniform(-6, 6, n), rng.uniform(-6, 6, n), rng.uniform(-2, 2, n), rng.random(n),
                          rng.choice([10.0, 30.0, 40.0], n)])
    pc[:40, 0] = 1.0 + 2.0                          # on the +length face of box 0 (identity orientation)
    pc[40:80, 2] = -1.0                             # on its bottom face
    pc[80:90, 1] = np.nan
    quats = [[0, 0, 0, 1], [0, 0, np.sin(0.35), np.cos(0.35)], list(rng.normal(size=4)), [0.1, -0.2, 0.3, 0.9]]
    boxes = [[1.0, 0.5, -1.0] + quats[0] + [4.0, 2.0, 1.5], [-2.0, 1.0, -1.5] + quats[1] + [4.2, 1.8, 1.6],
             [0.5, -2.0, -0.5] + quats[2] + [3.0, 3.0, 2.0], [2.0, 2.0, 0.0] + quats[3] + [1.0, 5.0, 1.0]]
    move = [0.25, -0.5, 0.125]
    out = {"pc": pc, "boxes": np.array(boxes), "move": np.array(move)}
    for b, bx in enumerate(boxes):
        anno = {"center": {"x": bx[0], "y": bx[1], "z": bx[2]}, "rotation": {"x": bx[3], "y": bx[4], "z": bx[5], "w": bx[6]},
                "length": bx[7], "width": bx[8], "height": bx[9]}
        out[f"cut{b}"] = cut_bbox.cut_bounding_box(pc, anno)
        out[f"cutm{b}"] = cut_bbox.cut_bounding_box(pc, anno, move)
        with np.errstate(invalid="ignore"):
            scene, box = cut_bbox.separate_bbox(pc, anno)
        out[f"sep_scene{b}"], out[f"sep_box{b}"] = scene, box
        print("box", b, "cut", len(out[f"cut{b}"]), "moved", len(out[f"cutm{b}"]), "separate", len(scene), len(box))
    np.savez_compressed(os.path.join(HERE, "cut_bbox.npz"), **out)


def main():
    if "--od" in sys.argv:
        return main_od()
    if "--cut" in sys.argv:
        return main_cut()
    fs = import_reference()
    for name, kw in CASES.items():
        case = build_case(**kw)
        out = run_case(fs, case)
        keep = {k: case[k] for k in ("xyzi", "label", "extra", "sample", "rich", "move", "T")}
        keep["sample_line"] = np.array(case["sample_line"])
        keep["anno_lines"] = np.array(case["anno_lines"])
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **keep, **out)
        print(name, "possible:", len(out["out_rot"]), "first:", out["out_rot"][:8])


if __name__ == "__main__":
    main()
