#!/usr/bin/env python3
"""Golden fixture for the float64 (Waymo) flavour of the batched path, made by the REFERENCE's own functions:

    python tests/golden/make_golden_waymo.py

A scene whose coordinates are genuine float64 -- what semantic_segmentation/Real3DAug/tools/datasets.py:240-262
(Waymo.__getitem__) hands the driver after subtracting the LiDAR offset -- goes through the reference's K-insert
chain (make_golden.chain: fill_spherical, geometrical_front_view, smooth_out, the inline block of
insertion.py:463-526), and the reference's own Waymo.save_data (:287-303) writes the three .npy files.
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import importlib                                   # noqa: E402

from make_golden import chain, import_reference, ref_save_bytes, save   # noqa: E402


def main():
    ref = import_reference()
    synth = importlib.import_module("pcl-augmentation_amd.synth")
    xyzi, label = synth.make_scene(41, 24, 500)
    rng = np.random.default_rng(41)
    scene5 = synth.scene5_from_packed(xyzi, label)
    scene5[:, 0:3] += rng.normal(0.0, 1e-3, (len(scene5), 3))            # not float32 values any more
    scene5[:, 0:3] -= np.array([1.22, 0, 2]) * 1e-3                         # (an offset subtracted in float64, :259)
    scene5[:, 4] = np.where(scene5[:, 4] == 40, 18, 14)                     # Waymo's label set
    kinds = ["pedestrian", "car", "cyclist"]
    samples = [synth.make_insert(4100 + k, kind, rng_range=(5.0, 15.0)) for k, kind in enumerate(kinds)]
    samples.insert(2, synth.make_insert(98, "car", centre_range=55.0))      # a rejected candidate (behind the wall)
    need = [20] * len(samples)
    need[2] = 5000
    merged, allvis, acc = chain(ref, scene5, samples, need)
    lidar, labels, check = ref_save_bytes("waymo", merged, allvis)
    save("chain_waymo_f64.npz", scene5=scene5, sample_sizes=np.array([len(s) for s in samples], dtype=np.int32),
         samples=np.vstack(samples), min_points=np.array(need, dtype=np.int32),
         merged=merged[:, [0, 1, 2, 6, 7]], all_visible=allvis[:, [0, 1, 2, 6, 7]], accepted=acc,
         lidar_npy=np.frombuffer(lidar, dtype=np.uint8), labels_npy=np.frombuffer(labels, dtype=np.uint8),
         check_npy=np.frombuffer(check, dtype=np.uint8))
    print("accepted", acc, "merged", merged.shape, "added", allvis.shape)


if __name__ == "__main__":
    main()
