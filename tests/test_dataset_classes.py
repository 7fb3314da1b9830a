"""CPU: the dataset classes of the mirror keep the reference's interface (SS tools/datasets.py:20-215, :218-410; OD
tools/datasets.py:40-190): listing, ``__len__``, ``__getitem__`` (arrays as the reference builds them with NumPy),
``delete_item``, ``create_directories`` (the tree the reference makes; the prompts' answers as arguments)."""
import os

import numpy as np
import pytest


def _config(root):
    return {"path": {"dataset_path": str(root / "data"), "annotation_path": str(root / "anno"), "output_path": str(root / "out"),
                     "label_path": str(root / "data" / "pseudo"), "train_txt_path": str(root / "data" / "train.txt")},
            "insertion": {"random": False, "classes": [30, 10], "number_of_classes": [2, 1], "number_of_object": 3},
            "labels": {30: "person", 10: "car", "Road": 40}, "split": {"train": [0, 3]}}


def test_semantic_kitti_class(pkg, tmp_path):
    ds = pkg.Real3DAug.tools.datasets
    cfg = _config(tmp_path)
    seq = tmp_path / "data" / "sequences" / "03"
    (seq / "velodyne").mkdir(parents=True), (seq / "labels").mkdir()
    rng = np.random.default_rng(1)
    frames = {}
    for f in (0, 1, 2, 5):
        x = rng.random((50 + f, 4), dtype=np.float32)
        l = (rng.integers(0, 60, 50 + f).astype(np.uint32) | (rng.integers(0, 900, 50 + f).astype(np.uint32) << 16))
        x.tofile(seq / "velodyne" / f"{f:06d}.bin"), l.tofile(seq / "labels" / f"{f:06d}.label")
        frames[f] = (x, l)
    np.savetxt(seq / "poses.txt", rng.random((6, 12)))
    d = ds.SemanticKITTI(cfg, "03", skip_scenes=1)
    assert len(d) == 3 and d.velodyne_list[0].endswith("000001.bin")
    assert ds.SemanticKITTI(cfg, "03", reverse=True).velodyne_list[0].endswith("000005.bin")
    pcl, T, bbox, inst, s = d[2]
    x, l = frames[5]
    assert pcl.dtype == np.float64 and np.array_equal(pcl[:, :4], x.astype(np.float64)) and np.array_equal(pcl[:, 4], (l & 0xFFFF))
    assert np.array_equal(inst[:, 0], l >> 16) and s == "03" and bbox == f"{cfg['path']['annotation_path']}/sequences/03/bbox/000005.txt"
    pose = np.vstack((d.poses[5].reshape(3, 4), [0, 0, 0, 1]))
    assert np.array_equal(T, np.dot(np.linalg.inv(d.my_calib), np.dot(pose, d.velo_2_cam)))
    d.delete_item(0)
    assert len(d) == 2 and d.velodyne_list[0].endswith("000002.bin")
    os.makedirs(cfg["path"]["output_path"])
    folder, number = d.create_directories("run", 7)
    assert (folder, number) == ("run/07/sequences", 7)
    for sq in ("00", "03"):
        for sub in ("velodyne", "check", "labels", "added_objects"):
            assert (tmp_path / "out" / folder / sq / sub).is_dir()
    assert (tmp_path / "out" / folder / "setting.txt").read_text() == "Inserted classes:\n     2x   person\n     1x   car\n"
    with pytest.raises(ValueError):
        d.create_directories("run", 100)
    # a writer without a list (what the pipelines construct)
    assert len(ds.SemanticKITTI(cfg)) == 0


def test_kitti_class(pkg, tmp_path):
    ds = pkg.Real3DAug.tools.datasets
    cfg = _config(tmp_path)
    data = tmp_path / "data"
    (data / "velodyne").mkdir(parents=True), (data / "pseudo").mkdir()
    rng = np.random.default_rng(2)
    x = rng.random((40, 4), dtype=np.float32)
    l = rng.integers(0, 1 << 20, 40).astype(np.uint32)
    x.tofile(data / "velodyne" / "000007.bin"), l.tofile(data / "pseudo" / "000007.label")
    (data / "train.txt").write_text("3\n7\n")
    d = ds.KITTI(cfg)
    assert len(d) == 2 and d.velodyne_list[1] == f"{data}/velodyne/000007.bin"
    pcl, label_2, inst, calib, img = d[1]
    assert np.array_equal(pcl[:, :4], x.astype(np.float64)) and np.array_equal(pcl[:, 4], l & 0xFFFF) and np.array_equal(inst[:, 0], l >> 16)
    assert (label_2, calib, img) == (f"{data}/label_2/000007.txt", f"{data}/calib/000007.txt", f"{data}/image_2/000007.png")
    d.delete_item(0)
    assert len(d) == 1
    os.makedirs(cfg["path"]["output_path"])
    folder, number = d.create_directories("od")
    assert (folder, number) == ("od/00", 0)
    assert all((tmp_path / "out" / folder / sub).is_dir() for sub in ("velodyne", "check", "label_2", "added_objects"))
    cfg2 = _config(tmp_path)
    del cfg2["path"]["train_txt_path"]
    assert len(ds.KITTI(cfg2)) == 0                                      # (a writer only)


def test_waymo_class(pkg, tmp_path):
    ds = pkg.Real3DAug.tools.datasets
    cfg = _config(tmp_path)
    rng = np.random.default_rng(3)
    for sq in ("segA", "segB"):
        for sub in ("lidar", "labels_v3_2", "poses"):
            (tmp_path / "data" / sq / sub).mkdir(parents=True)
        for f in (0, 1):
            np.save(tmp_path / "data" / sq / "lidar" / f"{f:04d}.npy", rng.random((30, 6)))
            np.save(tmp_path / "data" / sq / "labels_v3_2" / f"{f:04d}.npy", rng.integers(0, 20, (30, 2)))
            np.save(tmp_path / "data" / sq / "poses" / f"{f:04d}.npy", rng.random((4, 4)))
    d = ds.Waymo(cfg)
    assert len(d) == 4
    i = [k for k, f in enumerate(d.velodyne_list) if f.endswith("segB/lidar/0001.npy")][0]
    pcl, T, bbox, inst, sq = d[i]
    raw = np.load(tmp_path / "data" / "segB" / "lidar" / "0001.npy")
    lab = np.load(tmp_path / "data" / "segB" / "labels_v3_2" / "0001.npy")
    assert sq == "segB" and np.array_equal(pcl[:, :3], raw[:, :3] - d.LiDAR_location) and np.array_equal(pcl[:, 3], raw[:, 3])
    assert np.array_equal(pcl[:, 4], lab[:, 1]) and np.array_equal(inst[:, 0], lab[:, 0])
    corr = np.eye(4)
    corr[0:3, 3] = d.LiDAR_location
    assert np.array_equal(T, np.load(tmp_path / "data" / "segB" / "poses" / "0001.npy") @ corr)
    assert bbox == f"{cfg['path']['annotation_path']}/segB/bbox/0001.txt"
    d.delete_item(0)
    assert len(d) == 3
