#!/usr/bin/env python3
"""Golden fixture for the object-database creation, made by running the
REFERENCE's own scripts semantic_segmentation/cut_object/cut_out.py and filter_objects.py:

    python tests/golden/make_golden_cut.py

Both scripts are run as they are (``runpy``) on a small synthetic sequence in the Waymo layout of the
reference's dataset class (lidar/*.npy, labels_v3_2/*.npy, poses/*.npy, bbox/*.txt) -- the SemanticKITTI
branch of cut_out.py cannot run (:55 formats a str with ``:02d``, :92 passes an argument
``SemanticKITTI.delete_item`` does not take); the per-frame body (:86-157) is the same for both.  Bridged in
this generator only: ``Rotation.from_dcm`` / ``as_dcm`` (renamed in SciPy 1.4, removed in 1.6) are bound to
``from_matrix`` / ``as_matrix``.  The fixture holds the frames as the dataset class hands them to the script,
the box files, the configuration values used and every sample file the scripts left (name, ``anno``,
``pcl``), before and after filter_objects.
"""
import builtins
import glob
import os
import runpy
import shutil
import sys
import tempfile

import numpy as np
import yaml

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def bridge_scipy():
    """`from scipy.spatial.transform import Rotation` inside the scripts gets SciPy < 1.4 names on today's class."""
    import scipy.spatial.transform as T
    Rotation = T.Rotation

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

    T.Rotation = OldRotation


def make_sequence(n_frames=3):
    synth = __import__("importlib").import_module("pcl-augmentation_amd.synth")
    rng = np.random.default_rng(5)
    frames = []
    for f in range(n_frames):
        xyzi, label = synth.make_scene(350 + f, 24, 400)
        label = np.where(label == 40, 18, 14).astype(np.int64)                 # road / building in Waymo's label set
        parts, lines, labs = [xyzi.astype(np.float64)], [], [label]
        # objects of the inserted classes 5, 6, 7 standing in the scene, plus their box lines
        for k, (kind, cls) in enumerate([("cyclist", 5), ("pedestrian", 7), ("cyclist", 6), ("cyclist", 5), ("pedestrian", 7)]):
            obj = synth.make_insert(900 + 10 * (f % 2) + k, kind, rng_range=(4.0, 20.0))   # frame 2 sees frame 0's objects again,
            if f == 2:
                obj = obj[::2]                                                # thinner: filter_objects removes the poorer twin
            obj = obj[:[len(obj), len(obj), len(obj), 12, len(obj)][k]]       # one object too sparse for min_points
            parts.append(obj[:, :4])
            labs.append(np.full(len(obj), cls if k != 4 else 18, dtype=np.int64))    # last box: around road-labelled points
            c = obj[:, :3].mean(0)
            height, ext = (1.9, 1.0) if kind == "pedestrian" else (1.9, 2.2)
            yaw = np.random.default_rng(70 + 10 * (f % 2) + k).uniform(-np.pi, np.pi)
            lines.append(f"{cls} {c[0]:.4f} {c[1]:.4f} {c[2]:.4f} {height:.2f} {ext:.2f} {ext:.2f} {yaw:.4f} 0\n")   # :116-124
        lines.insert(1, "1 0.0 0.0 -1.7 1.0 2.0 2.0 0.0 0\n")                    # a class that is not inserted
        frames.append((np.vstack(parts), np.concatenate(labs), lines))
    return frames


def main():
    bridge_scipy()
    tmp = tempfile.mkdtemp(prefix="r3d_cut_")
    try:
        frames = make_sequence()
        seq = os.path.join(tmp, "data", "seq_a")
        for sub in ("lidar", "labels_v3_2", "poses", "bbox"):
            os.makedirs(os.path.join(seq, sub))
        lidar_location = np.array([1.22, 0, 2])
        for f, (xyz_i, label, lines) in enumerate(frames):
            raw = np.zeros((len(xyz_i), 6))
            raw[:, :4] = xyz_i
            raw[:, :3] += lidar_location                                  # __getitem__ subtracts it again (:259)
            np.save(os.path.join(seq, "lidar", f"{f:06d}.npy"), raw)
            np.save(os.path.join(seq, "labels_v3_2", f"{f:06d}.npy"), np.column_stack([np.arange(len(label)) % 9, label]))
            np.save(os.path.join(seq, "poses", f"{f:06d}.npy"), np.eye(4))
            if f != 1:                                                     # frame 1 has no box file (:97-98)
                with open(os.path.join(seq, "bbox", f"{f:06d}.txt"), "w") as fh:
                    fh.writelines(lines)
        with open(os.path.join(REF, "semantic_segmentation", "config", "waymo.yaml")) as fh:
            config = yaml.safe_load(fh)
        save = os.path.join(tmp, "objects")
        config["path"].update(dataset_path=os.path.join(tmp, "data"), annotation_path=os.path.join(tmp, "data"), bbox_path=save)
        os.makedirs(os.path.join(tmp, "semantic_segmentation", "config"))
        run_dir = os.path.join(tmp, "semantic_segmentation", "cut_object")
        os.makedirs(run_dir)
        with open(os.path.join(tmp, "semantic_segmentation", "config", "waymo.yaml"), "w") as fh:
            yaml.safe_dump(config, fh)
        sys.path.insert(0, REF)
        sys.path.insert(0, os.path.join(REF, "semantic_segmentation", "cut_object"))
        cwd, real_input = os.getcwd(), builtins.input
        os.chdir(run_dir)
        keep = {}
        try:
            builtins.input = lambda *a: "2"
            runpy.run_path(os.path.join(REF, "semantic_segmentation", "cut_object", "cut_out.py"), run_name="__main__")
            before = sorted(glob.glob(os.path.join(save, "*", "*.npz")))
            for i, p in enumerate(before):
                d = np.load(p, allow_pickle=True)
                keep[f"name{i}"], keep[f"anno{i}"], keep[f"pcl{i}"] = np.array(os.path.relpath(p, save)), np.array(str(d["anno"])), d["pcl"]
            runpy.run_path(os.path.join(REF, "semantic_segmentation", "cut_object", "filter_objects.py"), run_name="__main__")
            after = sorted(glob.glob(os.path.join(save, "*", "*.npz")))
            # the frames as the reference's dataset class hands them to the script
            from semantic_segmentation.Real3DAug.tools.datasets import Waymo
            ds = Waymo(config)
            for f in range(len(ds)):
                pts, _, anno_file, _, sequence = ds[f]
                keep[f"points{f}"], keep[f"anno_file{f}"] = pts, np.array(os.path.relpath(anno_file, tmp))
            keep["sequence"] = np.array(sequence)
        finally:
            builtins.input = real_input
            os.chdir(cwd)
        keep["n_files"] = np.array(len(before))
        keep["after_filter"] = np.array([os.path.relpath(p, save) for p in after])
        ins = config["insertion"]
        keep["classes"] = np.array(ins["classes"])
        keep["min_points"] = np.array([ins["min_points"][c] for c in ins["classes"]])
        keep["shortcuts"] = np.array([ins["labels_shortcut"][c] for c in ins["classes"]])
        keep["names"] = np.array([config["labels"][c] for c in ins["classes"]])
        for f, (_, _, lines) in enumerate(frames):
            keep[f"bbox{f}"] = np.array("".join(lines))
        np.savez_compressed(os.path.join(HERE, "cut_objects.npz"), **keep)
        print(f"\n{len(before)} samples written by cut_out.py, {len(after)} left by filter_objects.py:")
        for p in before:
            print("  ", os.path.relpath(p, save), "" if p in after else "(removed)")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
