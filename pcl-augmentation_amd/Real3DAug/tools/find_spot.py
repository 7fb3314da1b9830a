"""Function-level mirror of semantic_segmentation/Real3DAug/tools/find_spot.py on the HIP path.

Same names, arguments and return values as the reference: ``find_possible_places`` (:192-273)
returns the list of possible sample clouds, their annotation dictionaries and rotation numbers.
The annotation helpers are host code like the reference's (text parsing and one scipy call);
the 360-step search itself runs in ``r3d_find_possible_places`` and has no CPU fallback.
"""
from __future__ import annotations

import math

from ... import places as _places


def make_dictionary(annotation_array):
    """find_spot.py:15-27."""
    center = {"x": annotation_array[0][0], "y": annotation_array[0][1], "z": annotation_array[0][2]}
    rotation = {"x": annotation_array[1][0], "y": annotation_array[1][1], "z": annotation_array[1][2],
                "w": annotation_array[1][3]}
    return {"center": center, "rotation": rotation, "length": annotation_array[2][0], "width": annotation_array[2][1],
            "height": annotation_array[2][2], "class": annotation_array[3]}


def dictionary2array(annotation_dictionary):
    """find_spot.py:30-39."""
    a = annotation_dictionary
    return [[a["center"]["x"], a["center"]["y"], a["center"]["z"]],
            [a["rotation"]["x"], a["rotation"]["y"], a["rotation"]["z"], a["rotation"]["w"]],
            [a["length"], a["width"], a["height"]], a["class"]]


def read_label_line(line):
    """find_spot.py:155-189: 'class x y z height length width rot_z' -> annotation dictionary
    (``Rotation.from_dcm`` is today's ``from_matrix``)."""
    from scipy.spatial.transform import Rotation
    it = line.split(" ")
    a = float(it[7])
    m = [[math.cos(a), -1 * math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]]
    q = Rotation.from_matrix(m).as_quat()
    return make_dictionary([[float(it[1]), float(it[2]), float(it[3])], [q[0], q[1], q[2], q[3]],
                            [float(it[6]), float(it[5]), float(it[4])], [it[0]]])


def _anno10(a):
    return [a["center"]["x"], a["center"]["y"], a["center"]["z"], a["rotation"]["x"], a["rotation"]["y"],
            a["rotation"]["z"], a["rotation"]["w"], a["length"], a["width"], a["height"]]


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
    output_pcl = [c for c in res["clouds"]]
    output_annotation = [make_dictionary([[a[0], a[1], a[2]], [a[3], a[4], a[5], a[6]],
                                          [sample_annotation["length"], sample_annotation["width"],
                                           sample_annotation["height"]], sample_annotation["class"]])
                         for a in res["anno"]]
    output_rotation = [int(r) for r in res["rotations"]]
    return output_pcl, output_annotation, output_rotation
