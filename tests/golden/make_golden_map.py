#!/usr/bin/env python3
"""Golden fixture for the rich-map rasterisation, made by running the
REFERENCE's own script semantic_segmentation/rich_map/drivable_area_map.py in this container:

    python tests/golden/make_golden_map.py

The script has no importable function (everything is in its ``__main__`` block, which asks for
its inputs on the terminal and reads a SemanticKITTI tree from disk), so it is run as it is, with
``runpy``, on a small synthetic sequence written to a temporary directory in the dataset's layout
(velodyne/*.bin, labels/*.label, poses.txt), with a configuration derived from the reference's
YAML whose paths point there, and with ``input`` answering its three prompts (dataset 1, sequence
0, order "no").  The reference is not modified and nothing of it is stored; the fixture holds the
frames, the transform matrices its dataset class computed, and the map it saved.
"""
import builtins
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


def make_frames(n_frames=4):
    synth = __import__("importlib").import_module("pcl-augmentation_amd.synth")
    frames, poses = [], []
    for f in range(n_frames):
        xyzi, label = synth.make_scene(300 + f, 24, 360)
        label = label.copy()
        ground = label == 40
        label[ground & (xyzi[:, 1] > 4.0)] = 48                              # sidewalk
        label[ground & (xyzi[:, 0] < -7.0) & (xyzi[:, 1] <= 4.0)] = 44       # parking
        label[ground & (xyzi[:, 0] > 12.0) & (xyzi[:, 1] <= 4.0)] = 60       # lane marking: road
        if f >= 2:                                                           # later frames see another sidewalk
            label[ground & (xyzi[:, 1] < -6.0)] = 48
        label = label | (np.arange(len(label), dtype=np.uint32) % 7 << 16)   # instance bits must be ignored
        a = 0.05 * f                                                         # camera-frame pose: yaw about y, forward along z
        pose = np.array([[np.cos(a), 0, np.sin(a), 0.8 * f], [0, 1, 0, 0.02 * f], [-np.sin(a), 0, np.cos(a), 3.1 * f]])
        frames.append((xyzi, label.astype(np.uint32)))
        poses.append(pose.reshape(-1))
    return frames, np.array(poses)


def main():
    tmp = tempfile.mkdtemp(prefix="r3d_map_")
    try:
        frames, poses = make_frames()
        seq = os.path.join(tmp, "data", "sequences", "00")
        os.makedirs(os.path.join(seq, "velodyne"))
        os.makedirs(os.path.join(seq, "labels"))
        for f, (xyzi, label) in enumerate(frames):
            xyzi.tofile(os.path.join(seq, "velodyne", f"{f:06d}.bin"))
            label.tofile(os.path.join(seq, "labels", f"{f:06d}.label"))
        np.savetxt(os.path.join(seq, "poses.txt"), poses)
        with open(os.path.join(REF, "semantic_segmentation", "config", "semantic-kitti.yaml")) as fh:
            config = yaml.safe_load(fh)
        os.makedirs(os.path.join(tmp, "out"))
        config["path"].update(dataset_path=os.path.join(tmp, "data"), annotation_path=os.path.join(tmp, "data"),
                              maps_path=os.path.join(tmp, "out", "maps", "small", "npz"))
        cfg_dir = os.path.join(tmp, "semantic_segmentation", "config")
        run_dir = os.path.join(tmp, "semantic_segmentation", "rich_map")
        os.makedirs(cfg_dir)
        os.makedirs(run_dir)
        with open(os.path.join(cfg_dir, "semantic-kitti.yaml"), "w") as fh:
            yaml.safe_dump(config, fh)
        answers = iter(["1", "0", "no"])
        real_input, cwd = builtins.input, os.getcwd()
        builtins.input = lambda *a: next(answers)
        sys.path.insert(0, REF)
        os.chdir(run_dir)
        try:
            runpy.run_path(os.path.join(REF, "semantic_segmentation", "rich_map", "drivable_area_map.py"),
                           run_name="__main__")
            # the transform matrices, from the reference's dataset class (tools/datasets.py:62-67)
            from semantic_segmentation.Real3DAug.tools.datasets import SemanticKITTI
            answers = iter(["no"])
            ds = SemanticKITTI(config, "00")
            transforms = np.array([ds[i][1] for i in range(len(ds))])
        finally:
            builtins.input = real_input
            os.chdir(cwd)
        out = np.load(os.path.join(tmp, "out", "maps", "small", "npz", "00.npz"))
        keep = {"transforms": transforms, "map": out["map"].astype(np.uint8), "move": out["move"],
                "labels_road": np.array(config["insertion"]["placement_labels"][1]),
                "labels_sidewalk": np.array(config["insertion"]["placement_labels"][2]),
                "labels_parking": np.array(config["insertion"]["placement_labels"][3])}
        assert np.array_equal(out["map"], keep["map"])
        for f, (xyzi, label) in enumerate(frames):
            keep[f"xyzi{f}"], keep[f"label{f}"] = xyzi, label
        np.savez_compressed(os.path.join(HERE, "rich_map.npz"), **keep)
        print("\nmap", out["map"].shape, "move", out["move"].ravel(), "cells per code",
              [int((out["map"] == c).sum()) for c in range(4)])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
