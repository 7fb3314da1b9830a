#!/usr/bin/env python3
"""Golden fixture for the object-detection half of the object-database creation (SURVEY.md par.8 row
f-3), made by running the REFERENCE's own script object_detection/cut_object/object_cut_out.py:

    python tests/golden/make_golden_cut_od.py

The script is run as it is (``runpy``) on a small synthetic KITTI tree (velodyne/*.bin, pseudo-label
*.label, label_2/*.txt, calib/*.txt, image_2/*.png, train.txt) that this generator writes to a temporary
directory.  Stood in / bridged in this generator only:
  * ``skimage.io.imread`` (scikit-image is not installed here; the script only takes the image's shape)
    is served by Pillow;
  * ``Rotation.from_dcm`` (renamed in SciPy 1.4, removed in 1.6) is bound to ``from_matrix``.
The fixture holds, per frame, the points as the reference's dataset class hands them to the script, the
text of the label and calibration files, the image size, and every sample file the script left (name,
``anno``, ``pcl``).
"""
import glob
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
import yaml

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden_cut import bridge_scipy     # noqa: E402

IMG_W, IMG_H = 1242, 375
CALIB = """P0: 7.215377e+02 0.000000e+00 6.095593e+02 0.000000e+00 0.000000e+00 7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P1: 7.215377e+02 0.000000e+00 6.095593e+02 -3.875744e+02 0.000000e+00 7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P2: 7.215377e+02 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 7.215377e+02 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 1.000000e+00 2.745884e-03
P3: 7.215377e+02 0.000000e+00 6.095593e+02 -3.395242e+02 0.000000e+00 7.215377e+02 1.728540e+02 2.199936e+00 0.000000e+00 0.000000e+00 1.000000e+00 2.729905e-03
R0_rect: 9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01
Tr_velo_to_cam: 7.533745e-03 -9.999714e-01 -6.166020e-04 -4.069766e-03 1.480249e-02 7.280733e-04 -9.998902e-01 -7.631618e-02 9.998621e-01 7.523790e-03 1.480755e-02 -2.717806e-01
Tr_imu_to_velo: 9.999976e-01 7.553071e-04 -2.035826e-03 -8.086759e-01 -7.854027e-04 9.998898e-01 -1.482298e-02 3.195559e-01 2.024406e-03 1.482454e-02 9.998881e-01 -7.997231e-01
"""


def skimage_stand_in():
    """`from skimage import io` inside cutout.py: io.imread through Pillow (only .shape[:2] is used)."""
    from PIL import Image
    sk, io = types.ModuleType("skimage"), types.ModuleType("skimage.io")
    io.imread = lambda path: np.asarray(Image.open(path))
    sk.io = io
    sys.modules["skimage"], sys.modules["skimage.io"] = sk, io


def make_frames(n_frames=2):
    synth = __import__("importlib").import_module("pcl-augmentation_amd.synth")
    frames = []
    for f in range(n_frames):
        xyzi, label = synth.make_scene(610 + f, 24, 400)
        parts, labs, lines = [xyzi.astype(np.float32)], [label.astype(np.uint32)], []
        # (class, kind, lidar position, occluded, keep every n-th point, label of its points)
        spec = [("Pedestrian", "pedestrian", (9.0, 1.5), 0, 1, 30),       # in view, kept
                ("Cyclist", "cyclist", (14.0, -3.0), 0, 1, 31),           # in view, kept
                ("Pedestrian", "pedestrian", (7.0, 9.5), 0, 1, 30),       # beside the camera's field of view: skipped (:147)
                ("Cyclist", "cyclist", (-8.0, 1.0), 0, 1, 31),            # behind the car: skipped
                ("Pedestrian", "pedestrian", (12.0, 3.0), 1, 1, 30),      # occluded != 0: skipped (:106)
                ("Car", "car", (18.0, 4.0), 0, 1, 10),                    # a class that is not inserted (:101)
                ("Pedestrian", "pedestrian", (16.0, -1.0), 0, 6, 30),     # too few points (:159)
                ("Cyclist", "cyclist", (11.0 + f, -5.0), 0, 1, 31)]       # in view, kept; some of its points are road
        for k, (cls, kind, (px, py), occ, thin, lab) in enumerate(spec):
            obj = synth.make_insert(1200 + 10 * f + k, kind, rng_range=(6.0, 7.0))
            obj = obj[::thin]
            c = obj[:, :3].mean(0)
            obj = obj.copy()
            obj[:, 0] += px - c[0]
            obj[:, 1] += py - c[1]
            c = obj[:, :3].mean(0)
            lo, hi = obj[:, :3].min(0), obj[:, :3].max(0)
            olab = np.full(len(obj), lab, dtype=np.uint32)
            if k == 7:
                olab[obj[:, 2] < lo[2] + 0.15] = 40                        # its lowest points carry the road label (:153-155)
            parts.append(obj[:, :4].astype(np.float32))
            labs.append(olab)
            h = float(hi[2] - lo[2]) + 0.1
            ext = 1.0 if kind == "pedestrian" else (2.2 if kind == "cyclist" else 4.4)
            # KITTI camera coordinates of the box centre (object_cut_out.py:118-120 inverts this), rotation about the camera's y
            cam_x, cam_y, cam_z = -c[1], -(float(c[2]) + 0.08), c[0] - 0.27
            ry = np.random.default_rng(300 + 10 * f + k).uniform(-np.pi, np.pi)
            lines.append(f"{cls} 0.00 {occ} -1.57 100.00 100.00 200.00 200.00 {h:.2f} {ext:.2f} {ext:.2f} "
                         f"{cam_x:.2f} {cam_y:.2f} {cam_z:.2f} {ry:.2f}\n")
        frames.append((np.vstack(parts), np.concatenate(labs), lines))
    return frames


def main():
    from PIL import Image
    bridge_scipy()
    skimage_stand_in()
    tmp = tempfile.mkdtemp(prefix="r3d_cut_od_")
    try:
        frames = make_frames()
        data, labels, save = os.path.join(tmp, "kitti"), os.path.join(tmp, "pseudo"), os.path.join(tmp, "objects")
        for sub in ("velodyne", "label_2", "calib", "image_2"):
            os.makedirs(os.path.join(data, sub))
        os.makedirs(labels)
        os.makedirs(save)
        for f, (xyzi, label, lines) in enumerate(frames):
            xyzi.astype(np.float32).tofile(os.path.join(data, "velodyne", f"{f:06d}.bin"))
            label.astype(np.uint32).tofile(os.path.join(labels, f"{f:06d}.label"))
            with open(os.path.join(data, "label_2", f"{f:06d}.txt"), "w") as fh:
                fh.writelines(lines)
            with open(os.path.join(data, "calib", f"{f:06d}.txt"), "w") as fh:
                fh.write(CALIB)
            Image.new("RGB", (IMG_W, IMG_H)).save(os.path.join(data, "image_2", f"{f:06d}.png"))
        with open(os.path.join(tmp, "train.txt"), "w") as fh:
            fh.writelines(f"{f}\n" for f in range(len(frames)))
        with open(os.path.join(REF, "object_detection", "config", "KITTI.yaml")) as fh:
            config = yaml.safe_load(fh)
        config["path"].update(dataset_path=data, label_path=labels, sample_path=save, output_path=tmp,
                              train_txt_path=os.path.join(tmp, "train.txt"))
        run_dir = os.path.join(tmp, "object_detection", "cut_object")
        os.makedirs(run_dir)
        os.makedirs(os.path.join(tmp, "object_detection", "config"))
        with open(os.path.join(tmp, "object_detection", "config", "KITTI.yaml"), "w") as fh:
            yaml.safe_dump(config, fh)
        sys.path.insert(0, REF)
        sys.path.insert(0, os.path.join(REF, "object_detection", "cut_object"))
        cwd = os.getcwd()
        os.chdir(run_dir)
        keep = {}
        try:
            runpy.run_path(os.path.join(REF, "object_detection", "cut_object", "object_cut_out.py"), run_name="__main__")
            files = sorted(glob.glob(os.path.join(save, "*", "*.npz")))
            for i, p in enumerate(files):
                d = np.load(p, allow_pickle=True)
                keep[f"name{i}"], keep[f"anno{i}"], keep[f"pcl{i}"] = np.array(os.path.relpath(p, save)), np.array(str(d["anno"])), d["pcl"]
            from object_detection.Real3DAug.tools.datasets import KITTI
            ds = KITTI(config)
            for f in range(len(ds)):
                pts, label_file, _, calib_file, img_file = ds[f]
                keep[f"points{f}"] = pts
                keep[f"label_2_{f}"] = np.array(open(label_file).read())
        finally:
            os.chdir(cwd)
        print()
        keep["n_files"], keep["n_frames"] = np.array(len(files)), np.array(len(frames))
        keep["calib"], keep["img_size"] = np.array(CALIB), np.array([IMG_W, IMG_H])
        ins = config["insertion"]
        keep["classes"] = np.array(ins["classes"])
        keep["min_points"] = np.array([ins["min_points"][c] for c in ins["classes"]])
        keep["shortcuts"] = np.array([ins["labels_shortcut"][c] for c in ins["classes"]])
        keep["ground_labels"] = np.array([config["labels"][k] for k in ("Road", "Parking", "Sidewalk")])
        np.savez_compressed(os.path.join(HERE, "cut_objects_od.npz"), **keep)
        print(f"{len(files)} samples written by object_cut_out.py:")
        for p in files:
            print("  ", os.path.relpath(p, save), np.load(p, allow_pickle=True)["pcl"].shape)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
