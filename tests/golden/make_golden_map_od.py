#!/usr/bin/env python3
"""Golden fixture for the object-detection rich maps, made by running the
REFERENCE's own script object_detection/rich_map/single_drivable_area_map.py in this container:

    python tests/golden/make_golden_map_od.py

The script is run as it is (``runpy``) from a temporary object_detection/rich_map directory whose
../config/KITTI.yaml points to three synthetic frames laid out like KITTI (velodyne/*.bin, labels,
train.txt).  scikit-image is not installed: ``img_as_ubyte``, ``disk``, ``closing`` and ``dilation`` are
stood in with scipy.ndimage grey morphology, the routines scikit-image delegates to (so this call is
"parity unpinned" like the 5x3 closing of the hot path).  The fixture holds the frames and,
per frame, the two maps and offsets the script saved.
"""
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


def skimage_stand_in():
    from scipy import ndimage as ndi
    sk, sku, skm = types.ModuleType("skimage"), types.ModuleType("skimage.util"), types.ModuleType("skimage.morphology")
    sku.img_as_ubyte = lambda a: np.round(np.asarray(a) * 255).astype(np.uint8)

    def disk(r):
        y, x = np.mgrid[-r:r + 1, -r:r + 1]
        return (x * x + y * y <= r * r).astype(np.uint8)
    skm.disk = disk
    skm.rectangle = lambda nrows, ncols: np.ones((nrows, ncols), dtype=np.uint8)
    skm.dilation = lambda img, selem: ndi.grey_dilation(img, footprint=selem)
    skm.closing = lambda img, selem: ndi.grey_erosion(ndi.grey_dilation(img, footprint=selem), footprint=selem)
    sk.util, sk.morphology = sku, skm
    sys.modules.update({"skimage": sk, "skimage.util": sku, "skimage.morphology": skm})


def make_frames(n_frames=3):
    synth = __import__("importlib").import_module("pcl-augmentation_amd.synth")
    frames = []
    for f in range(n_frames):
        xyzi, label = synth.make_scene(330 + f, 20, 300)
        label = label.copy()
        ground = label == 40
        r = np.hypot(xyzi[:, 0], xyzi[:, 1])
        label[ground & (np.abs(xyzi[:, 1]) > 5.0 + f)] = 48                # sidewalks beside a road along x
        label[ground & (r > 25.0) & (xyzi[:, 0] < 0)] = 72                  # terrain: holes in the road raster
        label = label | (np.arange(len(label), dtype=np.uint32) % 5 << 16)
        frames.append((xyzi, label.astype(np.uint32)))
    return frames


def main():
    skimage_stand_in()
    tmp = tempfile.mkdtemp(prefix="r3d_map_od_")
    try:
        frames = make_frames()
        data = os.path.join(tmp, "kitti")
        os.makedirs(os.path.join(data, "velodyne"))
        os.makedirs(os.path.join(tmp, "labels"))
        os.makedirs(os.path.join(tmp, "out"))
        with open(os.path.join(tmp, "train.txt"), "w") as fh:
            for f, (xyzi, label) in enumerate(frames):
                xyzi.tofile(os.path.join(data, "velodyne", f"{f:06d}.bin"))
                label.tofile(os.path.join(tmp, "labels", f"{f:06d}.label"))
                fh.write(f"{f:06d}\n")
        with open(os.path.join(REF, "object_detection", "config", "KITTI.yaml")) as fh:
            config = yaml.safe_load(fh)
        config["path"].update(dataset_path=data, label_path=os.path.join(tmp, "labels"), maps_path=os.path.join(tmp, "out"),
                              train_txt_path=os.path.join(tmp, "train.txt"), output_path=os.path.join(tmp, "out"))
        os.makedirs(os.path.join(tmp, "object_detection", "config"))
        run_dir = os.path.join(tmp, "object_detection", "rich_map")
        os.makedirs(run_dir)
        with open(os.path.join(tmp, "object_detection", "config", "KITTI.yaml"), "w") as fh:
            yaml.safe_dump(config, fh)
        sys.path.insert(0, REF)
        cwd = os.getcwd()
        os.chdir(run_dir)
        try:
            runpy.run_path(os.path.join(REF, "object_detection", "rich_map", "single_drivable_area_map.py"), run_name="__main__")
        finally:
            os.chdir(cwd)
        keep = {"road_label": np.array(config["labels"]["Road"])}
        for f, (xyzi, label) in enumerate(frames):
            road = np.load(os.path.join(tmp, "out", "maps", "road_maps", "npz", f"{f:06d}.npz"))
            ped = np.load(os.path.join(tmp, "out", "maps", "pedestrian_area", "npz", f"{f:06d}.npz"))
            assert int(road["min_x"]) == int(ped["min_x"]) and int(road["min_y"]) == int(ped["min_y"])
            keep.update({f"xyzi{f}": xyzi, f"label{f}": label, f"road{f}": road["map"], f"ped{f}": ped["map"],
                         f"min{f}": np.array([int(road["min_x"]), int(road["min_y"])])})
            print(f"\nframe {f}: map {road['map'].shape} {road['map'].dtype}, road cells {int(road['map'].sum())}, "
                  f"pedestrian cells {int(ped['map'].sum())}, offset {keep[f'min{f}']}")
        np.savez_compressed(os.path.join(HERE, "rich_map_od.npz"), **keep)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
