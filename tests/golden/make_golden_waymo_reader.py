#!/usr/bin/env python3
"""Golden fixture for the Waymo frame reader, made by the REFERENCE's own class:

    python tests/golden/make_golden_waymo_reader.py

A small synthetic Waymo tree (sequence/lidar/*.npy float32 N x 6, labels_v3_2/*.npy int32 N x 2 = instance, semantic,
poses/*.npy 4 x 4) is read by semantic_segmentation/Real3DAug/tools/datasets.py Waymo.__getitem__ (:239-270), unmodified;
the fixture holds the files' arrays and what __getitem__ returned (the cloud after the LiDAR offset was subtracted in
float64, the pose times the correction matrix, the instances)."""
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import importlib                                   # noqa: E402

from make_golden import import_reference, save     # noqa: E402


def main():
    import_reference()                             # puts Real3DAug/ on the path (and the scikit-image stand-in)
    from tools import datasets as ref_ds           # the reference module
    synth = importlib.import_module("pcl-augmentation_amd.synth")
    rng = np.random.default_rng(77)
    with tempfile.TemporaryDirectory() as root:
        seq = os.path.join(root, "data", "seq_a")
        for sub in ("lidar", "labels_v3_2", "poses"):
            os.makedirs(os.path.join(seq, sub))
        xyzi, label = synth.make_scene(77, 16, 300)
        lidar = np.zeros((len(xyzi), 6), dtype=np.float32)
        lidar[:, :4] = xyzi
        lidar[:, :3] += np.array([1.22, 0, 2], dtype=np.float32)
        lidar[:, 4:] = rng.random((len(xyzi), 2), dtype=np.float32)                 # elongation etc.: dropped by the reader
        labels = np.stack([rng.integers(0, 500, len(xyzi)), np.where(label == 40, 18, 14)], axis=1).astype(np.int32)
        pose = np.eye(4)
        pose[:3, :3] = [[0.8, -0.6, 0.0], [0.6, 0.8, 0.0], [0.0, 0.0, 1.0]]
        pose[:3, 3] = [12.5, -3.25, 0.75]
        np.save(os.path.join(seq, "lidar", "000007.npy"), lidar)
        np.save(os.path.join(seq, "labels_v3_2", "000007.npy"), labels)
        np.save(os.path.join(seq, "poses", "000007.npy"), pose)
        ds = ref_ds.Waymo({"path": {"dataset_path": os.path.join(root, "data"), "annotation_path": "/anno", "output_path": root}})
        pcl, matrix, anno, instances, sequence = ds[0]
    assert pcl.dtype == np.float64 and sequence == "seq_a" and anno == "/anno/seq_a/bbox/000007.txt"
    save("waymo_reader.npz", lidar=lidar, labels=labels, pose=pose, pcl=pcl, matrix=matrix, instances=instances)
    print("pcl", pcl.shape, pcl.dtype, "float32-exact coordinates:", bool(np.array_equal(pcl[:, :3], pcl[:, :3].astype(np.float32))))


if __name__ == "__main__":
    main()
