"""The hot-path functions of the reference's ``Real3DAug/insertion.py``, same names and meaning.

Each function takes what the reference's takes (NumPy float64 arrays in the N x 9 scratch layout)
and returns what it returns; the work is done by the HIP kernels behind the C ABI.  A CUDA/ROCm
``torch.Tensor`` may be passed instead of a NumPy array, in which case results stay on the device.
Paths cited are relative to the reference root, SS = semantic_segmentation/.
"""
from __future__ import annotations


import numpy as np

from .. import _lib
from ._dev import back, ptr, to_device

NUMROW = 112          # SS Real3DAug/insertion.py:22
NUMCOLUMN = 360 * 4   # SS Real3DAug/insertion.py:23.  As in the reference, the pixel ids of column 8 multiply by THIS global
                      # (:116, :127, :470), read at call time, not by the num_column argument: whoever works on another
                      # grid edits the two globals (level1.augment_scene passes its grid to the private helpers
                      # _front_view / _merge instead, so that concurrent callers never see an edited global)


def add_space_for_spherical(point_cloud):
    """SS Real3DAug/insertion.py:54-64: N x 5 -> N x 9, unused columns -1."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    src, was_np = to_device(point_cloud, torch.float64)
    n = src.shape[0]
    out = torch.empty((n, 9), dtype=torch.float64, device=src.device)
    _lib.check(lib.r3d_add_space_for_spherical(ptr(src), n, ptr(out), _lib.stream_ptr()), "add_space_for_spherical")
    return back(out, was_np)


def fill_spherical(point_cloud):
    """SS Real3DAug/insertion.py:67-81: fills columns 3..5 in place, returns (pcl, max_el, min_el)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    if len(point_cloud) == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    dev, was_np = to_device(point_cloud, torch.float64)
    bounds = torch.empty(2, dtype=torch.float64, device=dev.device)
    status = torch.zeros(1, dtype=torch.int32, device=dev.device)
    _lib.check(lib.r3d_fill_spherical(ptr(dev), dev.shape[0], ptr(bounds), ptr(status), _lib.stream_ptr()),
               "fill_spherical")
    b = bounds.cpu().numpy()
    if int(status.item()) & _lib.S_NONFINITE:
        b = np.array([np.nan, np.nan])           # np.min / np.max propagate the NaN (:78-79)
    if was_np:
        point_cloud[:, 3:6] = dev[:, 3:6].cpu().numpy()
        return point_cloud, np.float64(b[0]), np.float64(b[1])
    return dev, np.float64(b[0]), np.float64(b[1])


def geometrical_front_view(point_cloud, num_row, num_column, max_elevation_angle, min_elevation_angle,
                           sample=False):
    """SS Real3DAug/insertion.py:84-129: returns (train, label, point_cloud); column 8 is written
    in place for every binned point; the reference's asserts (:110-112) are raised as
    AssertionError."""
    return _front_view(point_cloud, num_row, num_column, max_elevation_angle, min_elevation_angle, sample, int(NUMCOLUMN))


def _front_view(point_cloud, num_row, num_column, max_elevation_angle, min_elevation_angle, sample, id_columns):
    """geometrical_front_view with the multiplier of the pixel ids (the reference's global NUMCOLUMN) as an argument."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    dev, was_np = to_device(point_cloud, torch.float64)
    n = dev.shape[0]
    train = torch.empty((num_row, num_column), dtype=torch.float64, device=dev.device)
    label = torch.empty((num_row, num_column), dtype=torch.float64, device=dev.device)
    ws_bytes = lib.r3d_front_view_workspace_bytes(num_row, num_column)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev.device)
    status = torch.zeros(1, dtype=torch.int32, device=dev.device)
    _lib.check(lib.r3d_geometrical_front_view_grid(ptr(dev), n, num_row, num_column, int(id_columns),
                                                   float(max_elevation_angle), float(min_elevation_angle),
                                                   1 if sample else 0, ptr(train), ptr(label), ptr(ws), ws_bytes,
                                                   ptr(status), _lib.stream_ptr()), "geometrical_front_view")
    _lib.raise_status(int(status.item()), "geometrical_front_view")
    if was_np:
        if n:
            point_cloud[:, 8] = dev[:, 8].cpu().numpy()
        return train.cpu().numpy(), label.cpu().numpy(), point_cloud
    return train, label, dev


def occlusion_merge(scene_pcl, sample_pcl, scene_train, sample_train):
    """The unnamed inline block SS Real3DAug/insertion.py:463-482.

    Returns (scene_pcl, visible_sample, covered_scene) exactly as those variables stand after
    the loop: scene rows outside the visible pixels in original order; sample rows inside them
    grouped by pixel (row-major) and then sample order; removed scene rows in the same grouping.
    With no visible pixel the two lists are ``np.array([])`` like the reference's initial values.
    """
    return _merge(scene_pcl, sample_pcl, scene_train, sample_train, int(NUMCOLUMN))


def _merge(scene_pcl, sample_pcl, scene_train, sample_train, id_columns):
    """occlusion_merge with the multiplier of the pixel ids (the reference's global NUMCOLUMN) as an argument."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    sc, was_np = to_device(scene_pcl, torch.float64)
    sm, _ = to_device(sample_pcl, torch.float64)
    st, _ = to_device(scene_train, torch.float64)
    mt, _ = to_device(sample_train, torch.float64)
    n, m = sc.shape[0], sm.shape[0]
    rows, cols = st.shape
    out = torch.empty((n, 9), dtype=torch.float64, device=sc.device)
    cov = torch.empty((n, 9), dtype=torch.float64, device=sc.device)
    vis = torch.empty((m, 9), dtype=torch.float64, device=sc.device)
    counts = torch.zeros(3, dtype=torch.int64, device=sc.device)
    ws_bytes = lib.r3d_occlusion_merge_workspace_bytes(n, m, rows, cols)
    if ws_bytes == 0:
        raise _lib.R3DError("occlusion_merge: workspace query failed: " + lib.r3d_last_error().decode())
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=sc.device)
    _lib.check(lib.r3d_occlusion_merge_grid(ptr(sc), n, ptr(sm), m, ptr(st), ptr(mt), rows, cols, int(id_columns), ptr(out),
                                            ptr(vis), ptr(cov), ptr(counts), ptr(ws), ws_bytes, _lib.stream_ptr()),
               "occlusion_merge")
    n_out, n_vis, n_cov = (int(v) for v in counts.cpu().numpy())
    scene_out, visible, covered = out[:n_out], vis[:n_vis], cov[:n_cov]
    if was_np:
        scene_out, visible, covered = scene_out.cpu().numpy(), visible.cpu().numpy(), covered.cpu().numpy()
        if n_vis == 0 and n_cov == 0 and not bool((mt < st).any()):
            return scene_out, np.array([]), np.array([])
    return scene_out, visible, covered


def create_annotation_line(original_string, new_annotation_dict, rotation):
    """OD insertion.py:227-265: the KITTI label_2 line of an inserted object.  The sample's own line
    gives truncation, 2-D box and dimensions; the class and the centre (LiDAR frame) come from the
    placement; location is the centre in the camera frame (-y, -z - 0.08, x - 0.27), rotation_y the
    sample's minus the placement rotation, alpha the viewing angle plus rotation_y, both wrapped once
    into [-pi, pi]; occlusion is always 3."""
    items = (original_string.item() if hasattr(original_string, "item") else str(original_string)).split(" ")
    c = new_annotation_dict["center"]
    cam_x, cam_y, cam_z = c["y"] * -1, (c["z"] * -1) - 0.08, c["x"] - 0.27

    def wrap(a):
        if a < -np.pi:
            a += 2 * np.pi
        elif a > np.pi:
            a -= 2 * np.pi
        return a

    rotation_y = wrap(float(items[14]) - np.deg2rad(rotation))
    assert -np.pi <= rotation_y <= np.pi, f"Error in range of sample_rotation_y. Sample_rotation_y = {rotation_y}"
    alpha = wrap((np.arctan2(cam_x, cam_z) * -1) + rotation_y)
    assert -np.pi <= alpha <= np.pi, f"Error in range of alpha. Alpha = {alpha}"
    fields = [new_annotation_dict["class"], items[1], "3", f"{alpha:.02f}", items[4], items[5], items[6], items[7],
              items[8], items[9], items[10], f"{cam_x:.02f}", f"{cam_y:.02f}", f"{cam_z:.02f}", f"{rotation_y:.02f}"]
    return " ".join(fields) + "\n"
