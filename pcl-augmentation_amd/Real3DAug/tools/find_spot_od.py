"""Function-level mirror of object_detection/Real3DAug/tools/find_spot.py on the HIP path (the
semantic_segmentation flavour is ``find_spot.py``).  Same names, arguments and return values as
the reference; the 360-step search runs in ``r3d_find_possible_places`` with the object-detection
flavour bits set (include/real3daug_hip.h, R3D_PQ_*) and has no CPU fallback.
"""
from __future__ import annotations

import math

import numpy as np

from ... import _lib
from ... import places as _places


def make_dictionary(annotation_array):
    """OD find_spot.py:43-54 (the class is kept as a string)."""
    center = {"x": annotation_array[0][0], "y": annotation_array[0][1], "z": annotation_array[0][2]}
    rotation = {"x": annotation_array[1][0], "y": annotation_array[1][1], "z": annotation_array[1][2],
                "w": annotation_array[1][3]}
    return {"center": center, "rotation": rotation, "length": annotation_array[2][0], "width": annotation_array[2][1],
            "height": annotation_array[2][2], "class": annotation_array[3][0]}


def dictionary2array(annotation_dictionary):
    """OD find_spot.py:57-69."""
    a = annotation_dictionary
    return [[a["center"]["x"], a["center"]["y"], a["center"]["z"]],
            [a["rotation"]["x"], a["rotation"]["y"], a["rotation"]["z"], a["rotation"]["w"]],
            [a["length"], a["width"], a["height"]], [a["class"]]]


def read_label_line(line):
    """OD find_spot.py:179-224: KITTI label_2 line (camera frame) -> annotation in the LiDAR frame."""
    from scipy.spatial.transform import Rotation
    it = line.split(" ")
    height, width, length = float(it[8]), float(it[9]), float(it[10])
    x, y, z = float(it[11]), float(it[12]), float(it[13])
    a = float(it[14]) * -1
    m = [[math.cos(a), -1 * math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]]
    q = Rotation.from_matrix(m).as_quat()
    return make_dictionary([[float(z) + 0.27, float(x) * -1, float(y) * -1 - 0.08], [q[0], q[1], q[2], q[3]],
                            [width + 0.1, length + 0.1, height + 0.1], [it[0]]])


def _anno10(a):
    return [a["center"]["x"], a["center"]["y"], a["center"]["z"], a["rotation"]["x"], a["rotation"]["y"],
            a["rotation"]["z"], a["rotation"]["w"], a["length"], a["width"], a["height"]]


def place_query(scene, sample_pcl, sample_annotation, road_label):
    """The query dict ``places.find_places`` takes, with the object-detection rules switched on."""
    flavour = _lib.PQ_POINTWISE_ROTATION | _lib.PQ_MAP_NEEDS_POINT | _lib.PQ_COLLIDE_LABEL
    if sample_annotation["class"] == "Pedestrian":                       # OD find_spot.py:123-124
        flavour |= _lib.PQ_COLLIDE_ABOVE
    smp = np.array(sample_pcl, dtype=np.float64, copy=True)
    smp[:, 4] = 1                                                        # :249
    return {"scene": scene, "sample": smp, "anno": _anno10(sample_annotation), "ok_labels": [road_label],
            "ok_map": [1], "flavour": flavour, "collide_label": 1, "collide_dz": 0.1}


def find_possible_places(point_cloud, scene_annotation, sample_data, map_data, original_pcl, config):
    """OD find_spot.py:227-304."""
    sample_annotation = sample_data["anno"]
    sample_annotation = read_label_line(sample_annotation.item() if hasattr(sample_annotation, "item") else sample_annotation)
    scene = _places.PlaceScene(point_cloud, original_pcl, [_anno10(a) for a in scene_annotation], map_data["map"],
                               [map_data["min_x"], map_data["min_y"]], np.eye(4))
    res = _places.find_places([place_query(scene, sample_data["pcl"], sample_annotation, config["labels"]["Road"])])[0]
    output_pcl = [c for c in res["clouds"]]
    output_annotation = [make_dictionary([[a[0], a[1], a[2]], [a[3], a[4], a[5], a[6]],
                                          [sample_annotation["length"], sample_annotation["width"],
                                           sample_annotation["height"]], [sample_annotation["class"]]])
                         for a in res["anno"]]
    return output_pcl, output_annotation, [int(r) for r in res["rotations"]]
