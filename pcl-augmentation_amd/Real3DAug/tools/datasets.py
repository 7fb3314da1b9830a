"""The output-format half of the reference's ``Real3DAug/tools/datasets.py``.

The dataset classes keep the reference's names and methods -- ``SemanticKITTI``, ``KITTI``, ``Waymo`` with ``__len__``,
``__getitem__``, ``delete_item``, ``save_data``, ``remove_space_for_spherical``, ``create_directories`` (SS
tools/datasets.py:20-215, :218-410; OD tools/datasets.py:40-190) --, so that ``from tools.datasets import *`` of the
reference's driver can be pointed here.  What crosses the drop-in boundary byte for byte: a frame as ``__getitem__`` reads
it and ``velodyne/{f}.bin``, ``labels/{f}.label``, ``check/{f}.bin``, (object detection) ``label_2/{f}.txt``, (Waymo) the
``.npy`` files as ``save_data`` writes them.  The reference asks its questions with ``input()`` (which frames to skip,
whether to reverse, which folder number); here they are arguments with the reference's defaults.  Files are written under a
temporary name and renamed, so that a killed run never leaves a truncated file that a resumed run would take for finished.
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


def create_read_me(save_folder, config):
    """SS tools/datasets.py:6-17: setting.txt of a run."""
    with open(f"{save_folder}/setting.txt", "w") as txt:
        txt.write("Inserted classes:\n")
        if config["insertion"]["random"]:
            for c in config["insertion"]["classes"]:
                txt.write("     " + config["labels"][c] + "\n")
            txt.write("Randomly inserted " + str(config["insertion"]["number_of_object"]) + " objects\n")
        else:
            for c in range(len(config["insertion"]["classes"])):
                txt.write("     " + str(config["insertion"]["number_of_classes"][c]) + "x   " +
                          config["labels"][config["insertion"]["classes"][c]] + "\n")


def _folder(save_output_folder, save_folder, folder_number):
    """The numbered run folder of create_directories (SS :144-190, OD :130-175): the reference asks whether to take another
    number when {folder_number:02d} exists; here the caller passes the number (0 = the reference's default) and an existing
    folder is used as it is -- what answering "no" does there."""
    folder_number = int(folder_number)
    if not 0 <= folder_number <= 99:
        raise ValueError("folder_number must be a number between 0 and 99")
    os.makedirs(f"{save_output_folder}/{save_folder}/{folder_number:02d}", exist_ok=True)
    return folder_number


class SemanticKITTI:
    """SS tools/datasets.py:20-215.  ``sequence`` None: only the writers (``save_data``, ``remove_space_for_spherical``) -- the
    form the pipelines of this package use; with a sequence the class lists and reads its frames as the reference's does."""

    velo_2_cam = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03],
                           [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
                           [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01],
                           [0, 0, 0, 1]])                                            # :22-25
    my_calib = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]])    # :26-29

    def __init__(self, config, sequence=None, skip_scenes=0, reverse=False):
        """skip_scenes / reverse: the answers to the reference's prompt in create_velodyne_list (:108-131)."""
        self.config, self.sequence = config, sequence
        self.velodyne_list = np.array([])
        if sequence is not None:
            self.data_path = config["path"]["dataset_path"]
            self.anno_path = config["path"]["annotation_path"]
            self.poses = np.loadtxt(f"{self.data_path}/sequences/{self.sequence}/poses.txt")      # :36
            self.create_velodyne_list(skip_scenes, reverse)

    def __len__(self):
        return len(self.velodyne_list)

    def __getitem__(self, idx):
        """:45-60: (N x 5 float64 x y z intensity label, transform_matrix, bbox file, instance ids, sequence)."""
        file = self.velodyne_list[idx]
        frame_name = file.split("/")[-1].split(".")[0]
        pcl = np.fromfile(file, dtype=np.float32).reshape(-1, 4)
        labels = np.fromfile(f"{self.data_path}/sequences/{self.sequence}/labels/{frame_name}.label", dtype=np.uint32).reshape(-1, 1)
        instance = labels >> 16
        semantic_labels = labels & 0xFFFF
        pcl = np.hstack((pcl, semantic_labels))
        transform_matrix = self.create_transform_matrix(poses=self.poses, frame_number=int(frame_name))
        return pcl, transform_matrix, f"{self.anno_path}/sequences/{self.sequence}/bbox/{frame_name}.txt", instance, self.sequence

    def delete_item(self, idx):
        self.velodyne_list = np.delete(self.velodyne_list, idx)                        # :62-63

    def create_transform_matrix(self, poses, frame_number):
        """:65-70."""
        pose = np.vstack((poses[frame_number].reshape(3, 4), np.array([0, 0, 0, 1])))
        return np.dot(np.linalg.inv(self.my_calib), np.dot(pose, self.velo_2_cam))

    def create_velodyne_list(self, skip_scenes=0, reverse=False):
        """:108-142 with the prompt's answers as arguments."""
        import glob
        velodyne_address = np.array(glob.glob(f"{self.data_path}/sequences/{self.sequence}/velodyne/*.bin"))
        velodyne_address.sort()
        velodyne_address = velodyne_address[int(skip_scenes):]
        self.velodyne_list = velodyne_address[::-1] if reverse else velodyne_address

    def create_directories(self, save_folder, folder_number=0):
        """:144-215: {output_path}/{save_folder}/{NN}/sequences/{s:02d}/{velodyne,check,labels,added_objects} for the train
        split and setting.txt; returns (folder for save_data, folder_number)."""
        out = self.config["path"]["output_path"]
        folder_number = _folder(out, save_folder, folder_number)
        save_folder = f"{save_folder}/{folder_number:02d}/sequences"
        os.makedirs(f"{out}/{save_folder}", exist_ok=True)
        create_read_me(f"{out}/{save_folder}", self.config)
        for s in self.config["split"]["train"]:
            for sub in ("velodyne", "check", "labels", "added_objects"):
                os.makedirs(f"{out}/{save_folder}/{s:02d}/{sub}", exist_ok=True)
        return f"{save_folder}", folder_number

    def remove_space_for_spherical(self, point_cloud):
        return remove_space_for_spherical(point_cloud)

    def save_data(self, point_cloud, added_points, folder, name, idx=None):
        """:72-91; ``idx``: the frame leaves the list as in the reference (None: a writer without a list)."""
        xyzi, label, _ = pack_for_save(point_cloud)
        _, _, check = pack_for_save(added_points, 5)
        write_frame(self.config["path"]["output_path"], folder, name, xyzi, label, check, True)
        if idx is not None and len(self.velodyne_list):
            self.delete_item(idx)


class KITTI:
    """Object-detection variant (OD tools/datasets.py:40-190): no label file, 4-column check, and
    ``label_2/{f}.txt`` = the frame's annotation file plus one line per inserted object.  The frames are the ones
    ``config['path']['train_txt_path']`` lists (:112-128); without that key the class is a writer only."""

    def __init__(self, config):
        self.config = config
        self.data_path = config["path"].get("dataset_path")
        self.label_path = config["path"].get("label_path")
        self.train_txt_path = config["path"].get("train_txt_path")
        self.save_output_folder = config["path"]["output_path"]
        self.velodyne_list = np.array([])
        if self.train_txt_path is not None:
            self.create_velodyne_list()

    def __len__(self):
        return len(self.velodyne_list)

    def __getitem__(self, idx):
        """OD :56-71: (N x 5 float64, label_2 file, instance ids, calib file, image file)."""
        file = self.velodyne_list[idx]
        frame_name = file.split("/")[-1].split(".")[0]
        pcl = np.fromfile(file, dtype=np.float32).reshape(-1, 4)
        labels = np.fromfile(f"{self.label_path}/{frame_name}.label", dtype=np.uint32).reshape(-1, 1)
        instance = labels >> 16
        semantic_labels = labels & 0xFFFF
        pcl = np.hstack((pcl, semantic_labels))
        return (pcl, f"{self.data_path}/label_2/{frame_name}.txt", instance, f"{self.data_path}/calib/{frame_name}.txt",
                f"{self.data_path}/image_2/{frame_name}.png")

    def delete_item(self, idx):
        self.velodyne_list = np.delete(self.velodyne_list, idx)                        # OD :73-74

    def create_velodyne_list(self):
        """OD :112-128: one frame number per line of train.txt."""
        with open(self.train_txt_path) as fh:
            self.velodyne_list = [f"{self.data_path}/velodyne/{int(line):06d}.bin" for line in fh if line.strip()]

    def create_directories(self, save_folder, folder_number=0):
        """OD :130-190: {output_path}/{save_folder}/{NN}/{velodyne,check,label_2,added_objects}; returns (folder, number)."""
        folder_number = _folder(self.save_output_folder, save_folder, folder_number)
        save_folder = f"{save_folder}/{folder_number:02d}"
        for sub in ("velodyne", "check", "label_2", "added_objects"):
            os.makedirs(f"{self.save_output_folder}/{save_folder}/{sub}", exist_ok=True)
        return f"{save_folder}", folder_number

    def remove_space_for_spherical(self, point_cloud):
        return remove_space_for_spherical(point_cloud)[0]

    def save_data(self, point_cloud, added_points, folder, name, idx=None, additional_anno_lines=()):
        xyzi, _, _ = pack_for_save(point_cloud)
        _, _, check = pack_for_save(added_points, 4)
        if self.data_path is None:
            raise KeyError("config['path']['dataset_path'] is needed for label_2 (OD tools/datasets.py:82)")
        write_frame(self.save_output_folder, folder, name, xyzi, None, check, False,
                    label_2=(os.path.join(self.data_path, "label_2", f"{name}.txt"), list(additional_anno_lines)))
        if idx is not None and len(self.velodyne_list):
            self.delete_item(idx)


class Waymo:
    """The Waymo flavour of ``save_data`` (SS tools/datasets.py:287-303): the LiDAR offset that
    ``__getitem__`` subtracted (:259) is added back in float64, then the casts of ``pack_for_save``
    and three ``.npy`` files.  Waymo clouds are genuine float64 after that subtraction, so they go
    through the Level-1 functions (N x 9 float64), not through the float32 slabs of ``SceneBatch``."""

    LiDAR_location = np.array([1.22, 0, 2])           # :226

    def __init__(self, config):
        self.config = config
        self.data_path = config["path"].get("dataset_path")
        self.anno_path = config["path"].get("annotation_path")
        self.velodyne_list = np.array([])
        self.sequence = None
        if self.data_path is not None:
            self.create_velodyne_list()

    def __len__(self):
        return len(self.velodyne_list)

    def create_velodyne_list(self):
        """:319-336: every sequence's lidar/*.npy, sorted within the sequence."""
        import glob
        velodyne_address = []
        for sequence in glob.glob(f"{self.data_path}/*/", recursive=True):
            velodyne_address = velodyne_address + sorted(glob.glob(f"{sequence}lidar/*.npy"))
        self.velodyne_list = np.array(velodyne_address)
        self.sequence = self.velodyne_list[0].split("/")[-3] if len(self.velodyne_list) else None

    def __getitem__(self, idx):
        """:240-270: (N x 5 float64 with the LiDAR offset subtracted, pose @ correction, bbox file, instance ids, sequence)."""
        pcl_file = self.velodyne_list[idx]
        label_file, matrix_file = pcl_file.split("/"), pcl_file.split("/")
        sequence, frame_name = label_file[-3], label_file[-1].split(".")[0]
        label_file[-2], matrix_file[-2] = "labels_v3_2", "poses"
        pcl = np.load(pcl_file).reshape(-1, 6)[:, :4]
        semantic_labels = np.load("/".join(label_file)).reshape(-1, 2)
        instances = semantic_labels[:, 0].reshape(-1, 1)
        pcl = np.hstack((pcl, semantic_labels[:, 1].reshape(-1, 1)))
        pcl[:, 0:3] -= self.LiDAR_location
        transform_matrix = np.load("/".join(matrix_file)).reshape(4, 4)
        correction_matrix = np.eye(4)
        correction_matrix[0:3, 3] = self.LiDAR_location.T
        return pcl, transform_matrix @ correction_matrix, f"{self.anno_path}/{sequence}/bbox/{frame_name}.txt", instances, sequence

    def delete_item(self, idx, subdirectoties=True):
        """:272-285 (the sequence's sub-directories are made by save_data here, on demand)."""
        self.velodyne_list = np.delete(self.velodyne_list, idx)
        self.sequence = self.velodyne_list[0].split("/")[-3] if len(self.velodyne_list) else None

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
        if idx is not None and len(self.velodyne_list):
            self.delete_item(idx)                                                    # :303
