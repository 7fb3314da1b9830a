"""Object-database creation: the per-frame body of
semantic_segmentation/cut_object/cut_out.py:86-157 and the clean-up pass of filter_objects.py:85-115.

For every annotated object of a frame whose class is inserted (config['insertion']['classes']) the
points inside its box AND of its label are cut out (``r3d_cut_boxes``: every box of the frame in one
call, on the device), objects with fewer than ``min_points`` points are dropped, the rest are saved as
``{shortcut}{sequence}-{frame}_{count:02d}_{distance:03d}_m.npz`` with ``anno`` = the annotation line
and ``pcl`` = the N x 5 rows, exactly like the reference's ``np.savez`` (:156-157).  File walking and
the statistics of ``filter_objects`` are host bookkeeping (Python, like the reference).  The
``cut_frame_objects_od`` is the per-frame body of object_detection/cut_object/object_cut_out.py:75-168 (KITTI
label_2 lines in camera coordinates, unoccluded objects only, the enlarged box entirely inside the camera
image -- ``camera_fov_flags`` restates cut_object/cutout.py:52-127 --, ground-labelled points dropped).
"""
from __future__ import annotations

import glob
import math
import os
import struct

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


# ---- object-detection flavour (KITTI) --------------------------------------------------------------------
def kitti_box_from_label_line(items):
    """object_cut_out.py:108-137: a label_2 line (camera coordinates: items 8-10 = height width length,
    11-13 = x y z, 14 = rotation_y) -> the box in the LiDAR frame, 0.2 / 0.2 / 0.1 m larger than annotated."""
    from scipy.spatial.transform import Rotation as R
    height, width, length = float(items[8]), float(items[9]), float(items[10])
    cx, cy, cz = float(items[13]) + 0.27, float(items[11]) * -1, float(items[12]) * -1 - 0.08
    z_rot = float(items[14]) * -1
    rot = [[math.cos(z_rot), -1 * math.sin(z_rot), 0], [math.sin(z_rot), math.cos(z_rot), 0], [0, 0, 1]]
    q = R.from_matrix(rot).as_quat()                       # the reference's R.from_dcm, renamed in SciPy 1.4
    return {"center": {"x": cx, "y": cy, "z": cz}, "rotation": {"x": q[0], "y": q[1], "z": q[2], "w": q[3]},
            "length": width + 0.2, "width": length + 0.2, "height": height + 0.1}


def image_shape(img_file):
    """(rows, columns) of the camera image -- all cutout.py:116-117 takes from it.  PNG (KITTI's format) is read
    from the file header; anything else goes through Pillow."""
    with open(img_file, "rb") as fh:
        head = fh.read(24)
    if head[:8] == b"\x89PNG\r\n\x1a\n" and head[12:16] == b"IHDR":
        w, h = struct.unpack(">II", head[16:24])
        return np.array([h, w], dtype=np.int32)
    from PIL import Image
    with Image.open(img_file) as im:
        return np.array([im.height, im.width], dtype=np.int32)


def read_calibration(calib_file):
    """cutout.py:35-51: P2 (3x4), R0 (3x3), Tr_velo_to_cam (3x4) of a KITTI calib file, float32."""
    with open(calib_file) as fh:
        lines = fh.readlines()
    row = lambda i: np.array(lines[i].strip().split(" ")[1:], dtype=np.float32)
    return row(2).reshape(3, 4), row(4).reshape(3, 3), row(5).reshape(3, 4)


def camera_fov_flags(xyz, calib_file, img_shape):
    """cutout.py:70-74, :88-93, :101-108: LiDAR -> rectified camera -> image; a point is in view when its pixel lies
    inside the image and its depth is not negative.  Same NumPy operations and dtypes as the reference (float32
    matrices, float64 points), so the flags are the reference's."""
    P2, R0, V2C = read_calibration(calib_file)
    one = lambda a: np.hstack((a, np.ones((a.shape[0], 1), dtype=np.float32)))
    rect = np.dot(one(np.asarray(xyz)), np.dot(V2C.T, R0.T))
    rect_h = one(rect)
    hom = np.dot(rect_h, P2.T)
    img = (hom[:, 0:2].T / rect_h[:, 2]).T
    depth = hom[:, 2] - P2.T[3, 2]
    in_cols = np.logical_and(img[:, 0] >= 0, img[:, 0] < img_shape[1])
    in_rows = np.logical_and(img[:, 1] >= 0, img[:, 1] < img_shape[0])
    return np.logical_and(np.logical_and(in_cols, in_rows), depth >= 0)


def cut_frame_objects_od(points, label_file, calib_file, img_file, config, save_path):
    """One frame (object_cut_out.py:83-168): points N x 5 (x y z intensity semantic label), label_file = its
    label_2 text file.  Every box of the frame (enlarged and annotated size) is cut in one device call.
    Returns the list of files written."""
    classes = config["insertion"]["classes"]
    with open(label_file, "r") as fh:
        lines = [ln for ln in fh.readlines() if len(ln)]
    picked = [ln for ln in lines if ln.split(" ")[0] in classes and int(ln.split(" ")[2]) == 0]
    if not picked:
        return []
    annos = [kitti_box_from_label_line(ln.split(" ")) for ln in picked]
    wide = [dict(a, length=a["length"] + 0.2, width=a["width"] + 0.2, height=a["height"] + 0.2) for a in annos]
    cuts = cut_boxes(points, wide + annos)
    frame = label_file.split("/")[-1].split(".")[0]
    shape = image_shape(img_file)
    ground = [config["labels"][k] for k in ("Road", "Parking", "Sidewalk")]
    counts = {c: 0 for c in classes}
    written = []
    for k, (ln, anno) in enumerate(zip(picked, annos)):
        cls = ln.split(" ")[0]
        around, inside = cuts[k], cuts[len(annos) + k]
        if len(around) and not camera_fov_flags(around[:, 0:3], calib_file, shape).all():
            continue                                                     # partly outside the camera image (:147)
        counts[cls] += 1                                                 # counted before the min_points test (:152)
        inside = inside[~np.isin(inside[:, 4], ground)][:, 0:4]
        if len(inside) < config["insertion"]["min_points"][cls]:
            continue
        pcl = np.hstack((inside, np.ones((len(inside), 1))))
        c = anno["center"]
        path = (f"{save_path}/{cls}/{config['insertion']['labels_shortcut'][cls]}{frame}_{counts[cls]}_"
                f"{int(np.sqrt(c['x'] ** 2 + c['y'] ** 2))}_m")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez(path, anno=ln, pcl=pcl)
        written.append(path + ".npz")
    return written
