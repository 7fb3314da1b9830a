"""Object-database creation: the per-frame body of
semantic_segmentation/cut_object/cut_out.py:86-157 and the clean-up pass of filter_objects.py:85-115.

For every annotated object of a frame whose class is inserted (config['insertion']['classes']) the
points inside its box AND of its label are cut out (``r3d_cut_boxes``: every box of the frame in one
call, on the device), objects with fewer than ``min_points`` points are dropped, the rest are saved as
``{shortcut}{sequence}-{frame}_{count:02d}_{distance:03d}_m.npz`` with ``anno`` = the annotation line
and ``pcl`` = the N x 5 rows, exactly like the reference's ``np.savez`` (:156-157).  File walking and
the statistics of ``filter_objects`` are host bookkeeping (Python, like the reference).  The
object-detection script (object_cut_out.py) additionally asks the camera calibration whether the
enlarged box is in view (``cutout_frame``): ``in_view`` below is that predicate, supplied by the caller.
"""
from __future__ import annotations

import glob
import math
import os

import numpy as np

from .Real3DAug.tools.cut_bbox import cut_boxes


def box_from_bbox_line(items):
    """cut_out.py:116-138: centre, yaw about z as a quaternion, (length, width, height) = items 6, 5, 4."""
    from scipy.spatial.transform import Rotation as R
    yaw = float(items[7])
    rot = [[math.cos(yaw), -1 * math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]]
    q = R.from_matrix(rot).as_quat()                       # the reference's R.from_dcm, renamed in SciPy 1.4
    return {"center": {"x": float(items[1]), "y": float(items[2]), "z": float(items[3])},
            "rotation": {"x": q[0], "y": q[1], "z": q[2], "w": q[3]},
            "length": float(items[6]), "width": float(items[5]), "height": float(items[4])}


def cut_frame_objects(points, anno_file, config, sequence, save_path):
    """One frame (cut_out.py:90-157): points N x 5 float64 (x y z intensity label), anno_file = its bbox
    text file.  Returns the list of files written."""
    if not os.path.exists(anno_file):
        return []
    classes = config["insertion"]["classes"]
    with open(anno_file, "r") as fh:
        lines = [ln for ln in fh.readlines() if len(ln)]
    picked = [ln for ln in lines if int(ln.split(" ")[0]) in classes]
    if not picked:
        return []
    annos = [box_from_bbox_line(ln.split(" ")) for ln in picked]
    labels = [float(int(ln.split(" ")[0])) for ln in picked]
    cuts = cut_boxes(points, annos, classes=labels)                      # all boxes of the frame: one device call
    frame = anno_file.split("/")[-1].split(".")[0]
    counts = np.zeros(len(classes))
    written = []
    for ln, cut, anno in zip(picked, cuts, annos):
        cls = int(ln.split(" ")[0])
        counts[classes.index(cls)] += 1                                  # counted before the min_points test (:112)
        if len(cut) < config["insertion"]["min_points"][cls]:
            continue
        name = config["labels"][cls]
        short = config["insertion"]["labels_shortcut"][cls]
        dist = int(np.sqrt(anno["center"]["x"] ** 2 + anno["center"]["y"] ** 2))
        path = f"{save_path}/{name}/{short}{sequence}-{frame}_{int(counts[classes.index(cls)]):02d}_{dist:03d}_m"
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez(path, anno=ln, pcl=cut)
        written.append(path + ".npz")
    return written


def filter_objects(save_path, config):
    """filter_objects.py:85-115: per class and 1 m distance bin, the samples are grouped by their yaw
    (1 degree bins); a sample with fewer points than the average of its group is deleted."""
    removed = []
    for c in config["insertion"]["classes"]:
        cl = config["labels"][c]
        for i in range(100):
            files = glob.glob(f"{save_path}/{cl}/*_{i:03d}_m.npz")
            if not files:
                continue
            total, number, info = np.zeros(360), np.zeros(360), []
            for f in files:
                d = np.load(f, allow_pickle=True)
                rot = int(np.rad2deg(float(str(d["anno"]).split(" ")[7])) + 180)
                total[rot] += len(d["pcl"])
                number[rot] += 1
                info.append((f, rot, len(d["pcl"])))
            avg = np.where(number != 0, total / np.where(number != 0, number, 1), np.inf)
            for f, rot, n in info:
                if avg[rot] > n:
                    os.remove(f)
                    removed.append(f)
    return removed
