"""Host-side pieces of the object-detection object database (no GPU): the KITTI label
line -> box conversion, the PNG header reader and the camera field-of-view test of
pcl-augmentation_amd/cut_object.py, against the fixture of tests/golden/make_golden_cut_od.py and an
independent float64 formulation of the pinhole projection."""
import importlib
import struct
import zlib

import numpy as np

from conftest import load_golden

co = importlib.import_module("pcl-augmentation_amd.cut_object")


def _png(path, width, height):
    """A minimal valid PNG (one grey byte per pixel), written without any image library."""
    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + bytes(width) for _ in range(height))
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 0, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_image_shape_reads_the_png_header(tmp_path):
    _png(tmp_path / "a.png", 1242, 375)
    assert tuple(co.image_shape(str(tmp_path / "a.png"))) == (375, 1242)
    _png(tmp_path / "b.png", 7, 3)
    assert tuple(co.image_shape(str(tmp_path / "b.png"))) == (3, 7)


def test_label_line_to_lidar_box():
    """object_cut_out.py:108-137: camera x y z -> LiDAR (z + 0.27, -x, -y - 0.08), yaw = -rotation_y, the box
    0.2 / 0.2 / 0.1 m larger than annotated; the annotation line of every sample the reference saved parses."""
    g = load_golden("cut_objects_od.npz")
    for i in range(int(g["n_files"])):
        items = str(g[f"anno{i}"]).split(" ")
        box = co.kitti_box_from_label_line(items)
        h, w, l = float(items[8]), float(items[9]), float(items[10])
        assert (box["length"], box["width"], box["height"]) == (w + 0.2, l + 0.2, h + 0.1)
        c = box["center"]
        assert (c["x"], c["y"], c["z"]) == (float(items[13]) + 0.27, float(items[11]) * -1, float(items[12]) * -1 - 0.08)
        q = box["rotation"]
        yaw = 2.0 * np.arctan2(q["z"], q["w"])
        assert abs(np.angle(np.exp(1j * (yaw + float(items[14])))) ) < 1e-12 and q["x"] == 0 and q["y"] == 0
        # the reference named the file after the rounded-down ground distance of that centre
        assert str(g[f"name{i}"]).endswith(f"_{int(np.sqrt(c['x'] ** 2 + c['y'] ** 2))}_m.npz")


def test_camera_fov_flags_against_a_plain_pinhole_projection(tmp_path):
    g = load_golden("cut_objects_od.npz")
    (tmp_path / "calib.txt").write_text(str(g["calib"]))
    P2, R0, V2C = co.read_calibration(str(tmp_path / "calib.txt"))
    assert P2.dtype == np.float32 and P2.shape == (3, 4) and R0.shape == (3, 3) and V2C.shape == (3, 4)
    rng = np.random.default_rng(0)
    pts = np.column_stack([rng.uniform(-30, 60, 4000), rng.uniform(-40, 40, 4000), rng.uniform(-3, 3, 4000)])
    shape = np.array([375, 1242])
    got = co.camera_fov_flags(pts, str(tmp_path / "calib.txt"), shape)
    cam = R0.astype(np.float64) @ (V2C.astype(np.float64) @ np.column_stack([pts, np.ones(len(pts))]).T)      # 3 x N
    uvw = P2.astype(np.float64) @ np.vstack([cam, np.ones(len(pts))])
    u, v = uvw[0] / cam[2], uvw[1] / cam[2]
    want = (u >= 0) & (u < 1242) & (v >= 0) & (v < 375) & (uvw[2] - float(P2[2, 3]) >= 0)
    clear = (np.abs(u) > 1e-6) & (np.abs(u - 1242) > 1e-6) & (np.abs(v) > 1e-6) & (np.abs(v - 375) > 1e-6) & (np.abs(cam[2]) > 1e-6)
    assert np.array_equal(got[clear], want[clear]) and 0.05 < got.mean() < 0.6
