"""Function-level mirror of Real3DAug/tools/find_spot.py on the HIP path, both trees.

Module level: the semantic_segmentation flavour (SS find_spot.py) -- same names, arguments and return values as the
reference: ``find_possible_places`` (:192-273) returns the list of possible sample clouds, their annotation dictionaries
and rotation numbers.  ``od``: the object_detection flavour (OD find_spot.py) with the same function names
(``od.find_possible_places`` :227-304, ``od.read_label_line`` :179-224, ...); the two differ in how a label line is read, in
keeping the class as a string instead of a one-element list, and in the rules of the search (``od.place_query``: the
R3D_PQ_* flavour bits of include/real3daug_hip.h).  The annotation helpers are host code like the reference's (text parsing
and one scipy call); the 360-step search itself runs in ``r3d_find_possible_places`` and has no CPU fallback.
"""
from __future__ import annotations

import math
import types

import numpy as np

from ... import _lib
from ... import places as _places


def _make_dictionary(annotation_array, class_of):
    center = {"x": annotation_array[0][0], "y": annotation_array[0][1], "z": annotation_array[0][2]}
    rotation = {"x": annotation_array[1][0], "y": annotation_array[1][1], "z": annotation_array[1][2],
                "w": annotation_array[1][3]}
    return {"center": center, "rotation": rotation, "length": annotation_array[2][0], "width": annotation_array[2][1],
            "height": annotation_array[2][2], "class": class_of(annotation_array[3])}


def _dictionary2array(a, class_of):
    return [[a["center"]["x"], a["center"]["y"], a["center"]["z"]],
            [a["rotation"]["x"], a["rotation"]["y"], a["rotation"]["z"], a["rotation"]["w"]],
            [a["length"], a["width"], a["height"]], class_of(a["class"])]


def _quaternion_of_yaw(a):
    """``Rotation.from_dcm`` of the reference is today's ``from_matrix``."""
    from scipy.spatial.transform import Rotation
    m = [[math.cos(a), -1 * math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]]
    return Rotation.from_matrix(m).as_quat()


def _anno10(a):
    return [a["center"]["x"], a["center"]["y"], a["center"]["z"], a["rotation"]["x"], a["rotation"]["y"],
            a["rotation"]["z"], a["rotation"]["w"], a["length"], a["width"], a["height"]]


def _outputs(res, sample_annotation, make, class_of):
    output_annotation = [make([[a[0], a[1], a[2]], [a[3], a[4], a[5], a[6]],
                               [sample_annotation["length"], sample_annotation["width"], sample_annotation["height"]],
                               class_of(sample_annotation["class"])]) for a in res["anno"]]
    return [c for c in res["clouds"]], output_annotation, [int(r) for r in res["rotations"]]


# ---- semantic_segmentation ---------------------------------------------------------------------------------------------
def make_dictionary(annotation_array):
    """find_spot.py:15-27."""
    return _make_dictionary(annotation_array, lambda c: c)


def dictionary2array(annotation_dictionary):
    """find_spot.py:30-39."""
    return _dictionary2array(annotation_dictionary, lambda c: c)


def read_label_line(line):
    """find_spot.py:155-189: 'class x y z height length width rot_z' -> annotation dictionary."""
    it = line.split(" ")
    q = _quaternion_of_yaw(float(it[7]))
    return make_dictionary([[float(it[1]), float(it[2]), float(it[3])], [q[0], q[1], q[2], q[3]],
                            [float(it[6]), float(it[5]), float(it[4])], [it[0]]])


def placement_surfaces(sample_annotation, config):
    """find_spot.py:218-223."""
    ok_map_surface = config["insertion"]["placement"][int(sample_annotation["class"][0])]
    ok_surface = []
    for map_surface in ok_map_surface:
        ok_surface = ok_surface + config["insertion"]["placement_labels"][map_surface]
    return ok_map_surface, ok_surface


def find_possible_places(point_cloud, scene_annotation, sample_data, map, map_move, original_pcl,
                         transformation_matrix, config):
    """find_spot.py:192-273."""
    sample_annotation = sample_data["anno"]
    sample_annotation = read_label_line(sample_annotation.item() if hasattr(sample_annotation, "item") else sample_annotation)
    ok_map_surface, ok_surface = placement_surfaces(sample_annotation, config)
    scene = _places.PlaceScene(point_cloud, original_pcl, [_anno10(a) for a in scene_annotation], map, map_move,
                               transformation_matrix)
    res = _places.find_places([{"scene": scene, "sample": sample_data["pcl"], "anno": _anno10(sample_annotation),
                                "ok_labels": ok_surface, "ok_map": ok_map_surface}])[0]
    return _outputs(res, sample_annotation, make_dictionary, lambda c: c)


# ---- object_detection --------------------------------------------------------------------------------------------------
def _od_make_dictionary(annotation_array):
    """OD find_spot.py:43-54 (the class is kept as a string)."""
    return _make_dictionary(annotation_array, lambda c: c[0])


def _od_dictionary2array(annotation_dictionary):
    """OD find_spot.py:57-69."""
    return _dictionary2array(annotation_dictionary, lambda c: [c])


def _od_read_label_line(line):
    """OD find_spot.py:179-224: KITTI label_2 line (camera frame) -> annotation in the LiDAR frame."""
    it = line.split(" ")
    height, width, length = float(it[8]), float(it[9]), float(it[10])
    x, y, z = float(it[11]), float(it[12]), float(it[13])
    q = _quaternion_of_yaw(float(it[14]) * -1)
    return _od_make_dictionary([[float(z) + 0.27, float(x) * -1, float(y) * -1 - 0.08], [q[0], q[1], q[2], q[3]],
                                [width + 0.1, length + 0.1, height + 0.1], [it[0]]])


def _od_place_query(scene, sample_pcl, sample_annotation, road_label):
    """The query dict ``places.find_places`` takes, with the object-detection rules switched on."""
    flavour = _lib.PQ_POINTWISE_ROTATION | _lib.PQ_MAP_NEEDS_POINT | _lib.PQ_COLLIDE_LABEL
    if sample_annotation["class"] == "Pedestrian":                       # OD find_spot.py:123-124
        flavour |= _lib.PQ_COLLIDE_ABOVE
    smp = np.array(sample_pcl, dtype=np.float64, copy=True)
    smp[:, 4] = 1                                                        # :249
    return {"scene": scene, "sample": smp, "anno": _anno10(sample_annotation), "ok_labels": [road_label],
            "ok_map": [1], "flavour": flavour, "collide_label": 1, "collide_dz": 0.1}


def _od_find_possible_places(point_cloud, scene_annotation, sample_data, map_data, original_pcl, config):
    """OD find_spot.py:227-304."""
    sample_annotation = sample_data["anno"]
    sample_annotation = _od_read_label_line(sample_annotation.item() if hasattr(sample_annotation, "item") else sample_annotation)
    scene = _places.PlaceScene(point_cloud, original_pcl, [_anno10(a) for a in scene_annotation], map_data["map"],
                               [map_data["min_x"], map_data["min_y"]], np.eye(4))
    res = _places.find_places([_od_place_query(scene, sample_data["pcl"], sample_annotation, config["labels"]["Road"])])[0]
    return _outputs(res, sample_annotation, _od_make_dictionary, lambda c: [c])


od = types.SimpleNamespace(make_dictionary=_od_make_dictionary, dictionary2array=_od_dictionary2array,
                           read_label_line=_od_read_label_line, place_query=_od_place_query,
                           find_possible_places=_od_find_possible_places, _anno10=_anno10)
