"""The output-format half of the reference's ``Real3DAug/tools/datasets.py``.

Only what the hot path's drop-in promise needs: reading a frame the way ``__getitem__`` does and
writing ``velodyne/{f}.bin``, ``labels/{f}.label``, ``check/{f}.bin`` and (object detection)
``label_2/{f}.txt`` byte for byte the way ``save_data`` does (SS tools/datasets.py:45-60, 72-106;
OD tools/datasets.py:20-37, 56-109).  Directory creation prompts and pose handling stay with the
reference's driver.  Files are written under a temporary name and renamed, so that a killed run never
leaves a truncated file that a resumed run would take for finished.
"""
from __future__ import annotations

import os

import numpy as np

from ... import _lib
from .._dev import ptr, to_device


def remove_space_for_spherical(point_cloud):
    """SS tools/datasets.py:93-106: (N x 4 xyz+intensity, N x 1 label), float64 like the reference."""
    n = len(point_cloud)
    labels = np.zeros((n, 1))
    pcl = np.ones((n, 4)) * -1
    if n:
        pc = point_cloud.cpu().numpy() if hasattr(point_cloud, "cpu") else point_cloud
        pcl[:, 0:3] = pc[:, 0:3]
        pcl[:, 3] = pc[:, 6]
        labels[:, 0] = pc[:, 7]
    return pcl, labels


def pack_for_save(point_cloud, check_cols=5):
    """Device-side casts of save_data: (xyzi float32 [N,4], label uint32 [N], check float32)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    dev, _ = to_device(point_cloud, torch.float64)
    n = dev.shape[0]
    xyzi = torch.empty((n, 4), dtype=torch.float32, device=dev.device)
    label = torch.empty(n, dtype=torch.int32, device=dev.device)
    check = torch.empty((n, check_cols), dtype=torch.float32, device=dev.device)
    _lib.check(lib.r3d_remove_space_for_spherical(ptr(dev), n, ptr(xyzi), ptr(label), ptr(check), check_cols,
                                                  _lib.stream_ptr()), "remove_space_for_spherical")
    return xyzi.cpu().numpy(), label.cpu().numpy().view(np.uint32), check.cpu().numpy()


def read_frame(velodyne_file, label_file):
    """SS tools/datasets.py:51-56: float32 N x 4 + uint32 labels -> (xyzi, semantic label, instance)."""
    xyzi = np.fromfile(velodyne_file, dtype=np.float32).reshape(-1, 4)
    labels = np.fromfile(label_file, dtype=np.uint32)
    return xyzi, labels & 0xFFFF, labels >> 16


LIDAR_LOCATION = np.array([1.22, 0, 2])               # SS tools/datasets.py:226


def read_frame_waymo(lidar_file, label_file=None, pose_file=None):
    """SS tools/datasets.py:239-270 (Waymo.__getitem__): ``lidar/{f}.npy`` rows of 6 -> x y z intensity, the semantic
    column of ``labels_v3_2/{f}.npy`` (rows: instance, semantic) appended (float64 by NumPy's promotion, :257), the
    LiDAR offset subtracted in float64 (:259).  Returns (pcl N x 5 float64, pose @ correction 4 x 4 or None,
    instances N x 1).  The other two files sit beside the scan (``labels_v3_2/``, ``poses/``) unless given."""
    parts = lidar_file.split("/")
    if label_file is None:
        label_file = "/".join(parts[:-2] + ["labels_v3_2", parts[-1]])
    if pose_file is None:
        pose_file = "/".join(parts[:-2] + ["poses", parts[-1]])
    pcl = np.load(lidar_file).reshape(-1, 6)
    pcl = pcl[:, :4]
    semantic_labels = np.load(label_file).reshape(-1, 2)
    instances = semantic_labels[:, 0].reshape(-1, 1)
    semantic_labels = semantic_labels[:, 1].reshape(-1, 1)
    pcl = np.hstack((pcl, semantic_labels))
    pcl[:, 0:3] -= LIDAR_LOCATION
    matrix = None
    if os.path.exists(pose_file):
        correction_matrix = np.eye(4)
        correction_matrix[0:3, 3] = LIDAR_LOCATION.T
        matrix = np.load(pose_file).reshape(4, 4) @ correction_matrix
    return pcl, matrix, instances


def save_arrays_waymo(merged5, added5):
    """The three arrays Waymo.save_data stores (SS tools/datasets.py:287-301) from N x 5 float64 rows
    [x y z intensity label]: the offset added back in float64, then the casts."""
    m, a = np.array(merged5, dtype=np.float64, copy=True), np.array(added5, dtype=np.float64, copy=True).reshape(-1, 5)
    m[:, 0:3] += LIDAR_LOCATION
    a[:, 0:3] += LIDAR_LOCATION
    # (C order whatever the layout of the arrays that came in: np.save records the order in the file's header)
    return m[:, 0:4].astype(np.float32, order="C"), m[:, 4:5].astype(np.uint32, order="C"), a.astype(np.float32, order="C")


def write_frame_waymo(output_path, folder, name, merged5, added5):
    """``lidar/ labels_v3_2/ check/{name}.npy`` as Waymo.save_data writes them; check/ last (its presence marks the
    frame as done)."""
    import io
    base = os.path.join(output_path, folder)
    for sub, arr in zip(("lidar", "labels_v3_2", "check"), save_arrays_waymo(merged5, added5)):
        os.makedirs(os.path.join(base, sub), exist_ok=True)
        buf = io.BytesIO()
        np.save(buf, arr)
        _commit(os.path.join(base, sub, f"{name}.npy"), buf.getvalue())


def _commit(path, data):
    """Write bytes (or an array's bytes) to path.tmp, then rename: the file is whole or absent."""
    tmp = path + ".tmp"
    if not isinstance(data, (bytes, bytearray, memoryview)):
        data = np.ascontiguousarray(data)
        data = memoryview(data).cast("B") if data.size else b""      # (a view with a zero in its shape cannot be cast)
    with open(tmp, "wb") as fh:
        fh.write(data)
    os.replace(tmp, path)


def create_annotation(old_address, new_address, additional_annotations_lines):
    """OD tools/datasets.py:20-37: the frame's label_2 file followed by the lines of the inserted objects."""
    with open(old_address, "r") as fh:
        text = fh.read()
    _commit(new_address, (text + "".join(additional_annotations_lines)).encode())


def write_frame(output_path, folder, name, xyzi, label, check, write_labels=True, label_2=None):
    """Write the files of save_data from packed arrays (SS :80-89; OD :81-93: no labels, but
    label_2 = (path of the frame's label_2 file, lines of the inserted objects)).  check/ is written
    last: its presence marks the frame as done (``AugmentPipeline`` resumes by it)."""
    base = os.path.join(output_path, folder)
    for sub in ("velodyne", "check") + (("labels",) if write_labels else ()) + (("label_2",) if label_2 else ()):
        os.makedirs(os.path.join(base, sub), exist_ok=True)
    if label_2:
        create_annotation(label_2[0], os.path.join(base, "label_2", f"{name}.txt"), label_2[1])
    _commit(os.path.join(base, "velodyne", f"{name}.bin"), np.ascontiguousarray(xyzi, dtype=np.float32))
    if write_labels:
        _commit(os.path.join(base, "labels", f"{name}.label"), np.ascontiguousarray(label, dtype=np.uint32))
    _commit(os.path.join(base, "check", f"{name}.bin"), np.ascontiguousarray(check, dtype=np.float32))


class SemanticKITTI:
    """``save_data`` / ``remove_space_for_spherical`` of the reference's class (SS :72-106)."""

    def __init__(self, config):
        self.config = config

    def remove_space_for_spherical(self, point_cloud):
        return remove_space_for_spherical(point_cloud)

    def save_data(self, point_cloud, added_points, folder, name, idx=None):
        xyzi, label, _ = pack_for_save(point_cloud)
        _, _, check = pack_for_save(added_points, 5)
        write_frame(self.config["path"]["output_path"], folder, name, xyzi, label, check, True)


class KITTI:
    """Object-detection variant (OD tools/datasets.py:76-109): no label file, 4-column check, and
    ``label_2/{f}.txt`` = the frame's annotation file plus one line per inserted object."""

    def __init__(self, config):
        self.config = config
        self.data_path = config["path"].get("dataset_path")
        self.save_output_folder = config["path"]["output_path"]

    def remove_space_for_spherical(self, point_cloud):
        return remove_space_for_spherical(point_cloud)[0]

    def save_data(self, point_cloud, added_points, folder, name, idx=None, additional_anno_lines=()):
        xyzi, _, _ = pack_for_save(point_cloud)
        _, _, check = pack_for_save(added_points, 4)
        if self.data_path is None:
            raise KeyError("config['path']['dataset_path'] is needed for label_2 (OD tools/datasets.py:82)")
        write_frame(self.save_output_folder, folder, name, xyzi, None, check, False,
                    label_2=(os.path.join(self.data_path, "label_2", f"{name}.txt"), list(additional_anno_lines)))


class Waymo:
    """The Waymo flavour of ``save_data`` (SS tools/datasets.py:287-303): the LiDAR offset that
    ``__getitem__`` subtracted (:259) is added back in float64, then the casts of ``pack_for_save``
    and three ``.npy`` files.  Waymo clouds are genuine float64 after that subtraction, so they go
    through the Level-1 functions (N x 9 float64), not through the float32 slabs of ``SceneBatch``."""

    LiDAR_location = np.array([1.22, 0, 2])           # :226

    def __init__(self, config):
        self.config = config

    def remove_space_for_spherical(self, point_cloud):
        return remove_space_for_spherical(point_cloud)

    def save_data(self, point_cloud, added_points, folder, name, idx=None):
        point_cloud = np.array(point_cloud, dtype=np.float64, copy=True)
        added_points = np.array(added_points, dtype=np.float64, copy=True)
        point_cloud[:, 0:3] += self.LiDAR_location                                  # :288
        added_points[:, 0:3] += self.LiDAR_location                                 # :289
        xyzi, label, _ = pack_for_save(point_cloud)
        _, _, check = pack_for_save(added_points, 5)
        base = os.path.join(self.config["path"]["output_path"], folder)
        for sub in ("lidar", "labels_v3_2", "check"):
            os.makedirs(os.path.join(base, sub), exist_ok=True)
        import io
        for sub, arr in (("lidar", xyzi), ("labels_v3_2", label.reshape(-1, 1)), ("check", check)):   # :297-301
            buf = io.BytesIO()
            np.save(buf, arr)
            _commit(os.path.join(base, sub, f"{name}.npy"), buf.getvalue())
