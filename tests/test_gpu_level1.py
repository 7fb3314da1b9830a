"""GPU parity, function by function, through the C ABI (-m gpu).

Each mirror of a reference function is checked against the golden vectors captured from the
reference and against the oracle on fresh seeded inputs.  Integer / index / order results are
bit-exact.  Tolerance for the spherical coordinates: 1e-12 rad here (north_star allows 1e-5);
r is exact because sqrt and the squares are IEEE operations.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import real3d_oracle as O

pytestmark = pytest.mark.gpu
ANGLE_TOL = 1e-12


@pytest.fixture(scope="module")
def R(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return pkg.Real3DAug


def _scene5(g):
    return np.hstack((g["in_xyzi"].astype(np.float64), g["in_label"].astype(np.float64)[:, None]))


@pytest.mark.parametrize("name", ["step_s2k.npz", "step_s8k.npz", "step_s20k.npz"])
def test_step_functions_against_golden(R, name):
    g = load_golden(name)
    ins, clo = R.insertion, R.tools.closing
    sc = ins.add_space_for_spherical(_scene5(g))
    assert np.array_equal(sc, O.add_space_for_spherical(_scene5(g)))
    sc, mx, mn = ins.fill_spherical(sc)
    assert np.array_equal(sc[:, 3], g["scene_sph"][:, 0])                       # r exact
    assert np.abs(sc[:, 4:6] - g["scene_sph"][:, 1:3]).max() <= ANGLE_TOL
    assert abs(mx - g["bounds"][0]) <= ANGLE_TOL and abs(mn - g["bounds"][1]) <= ANGLE_TOL
    # a3 on the golden angles: pure IEEE arithmetic, must be bit-exact
    sc_g = sc.copy()
    sc_g[:, 3:6] = g["scene_sph"]
    tr, lb, sc_g = ins.geometrical_front_view(sc_g, 112, 1440, g["bounds"][0], g["bounds"][1])
    assert np.array_equal(sc_g[:, 8].astype(np.int64), g["scene_pix"])
    assert np.array_equal(tr, g["scene_train_raw"]) and np.array_equal(lb, g["scene_label_raw"])
    # a3 on the device's own angles: pixel ids still identical
    tr_d, lb_d, sc = ins.geometrical_front_view(sc, 112, 1440, mx, mn)
    assert np.array_equal(sc[:, 8].astype(np.int64), g["scene_pix"])
    assert np.array_equal(lb_d, g["scene_label_raw"])
    # a4 / a5
    assert np.array_equal(clo.class_closing(lb), g["scene_closed_u8"])
    tr2, lb2 = clo.smooth_out(tr, lb)
    assert np.array_equal(tr2, g["scene_train"]) and np.array_equal(lb2, g["scene_label"])
    # sample side, sample=True
    sm = ins.add_space_for_spherical(g["sample5"])
    sm, _, _ = ins.fill_spherical(sm)
    assert np.array_equal(sm[:, 3], g["sample_sph"][:, 0])
    assert np.abs(sm[:, 4:6] - g["sample_sph"][:, 1:3]).max() <= ANGLE_TOL
    sm[:, 3:6] = g["sample_sph"]
    mtr, mlb, sm = ins.geometrical_front_view(sm, 112, 1440, g["bounds"][0], g["bounds"][1], sample=True)
    assert np.array_equal(sm[:, 8].astype(np.int64), g["sample_pix"])
    assert np.array_equal(mtr, g["sample_train_raw"])
    mtr2, mlb2 = clo.smooth_out(mtr, mlb)
    assert np.array_equal(mtr2, g["sample_train"]) and np.array_equal(mlb2, g["sample_label"])
    # a6-a8
    out, vis, cov = ins.occlusion_merge(sc_g, sm, tr2, mtr2)
    assert np.array_equal(out, g["scene_out"])
    assert np.array_equal(vis.reshape(-1, 9), g["visible_sample"])
    assert np.array_equal(cov.reshape(-1, 9), g["covered_scene"])


@pytest.mark.parametrize("name", ["edge_above.npz", "edge_below.npz", "edge_hidden.npz", "edge_seam.npz"])
def test_edge_cases(R, name):
    g = load_golden(name)
    ins, clo = R.insertion, R.tools.closing
    sc = ins.add_space_for_spherical(_scene5(g))
    sc, mx, mn = ins.fill_spherical(sc)
    tr, lb, sc = ins.geometrical_front_view(sc, 112, 1440, mx, mn)
    assert np.array_equal(sc[:, 8].astype(np.int64), g["scene_pix"])
    tr, lb = clo.smooth_out(tr, lb)
    assert np.array_equal(lb, g["scene_label"])
    assert np.abs(tr - g["scene_train"]).max() == 0
    sm = ins.add_space_for_spherical(g["sample5"])
    sm, _, _ = ins.fill_spherical(sm)
    mtr, mlb, sm = ins.geometrical_front_view(sm, 112, 1440, mx, mn, sample=True)
    assert np.array_equal(sm[:, 8].astype(np.int64), g["sample_pix"])            # -1 kept for skipped rows
    mtr, mlb = clo.smooth_out(mtr, mlb)
    assert np.array_equal(mtr, g["sample_train"]) and np.array_equal(mlb, g["sample_label"])
    out, vis, cov = ins.occlusion_merge(sc, sm, tr, mtr)
    cols = [0, 1, 2, 6, 7, 8]
    assert np.array_equal(out[:, cols], g["scene_out"][:, cols])
    assert np.array_equal(vis.reshape(-1, 9)[:, cols], g["visible_sample"][:, cols])
    assert np.array_equal(cov.reshape(-1, 9)[:, cols], g["covered_scene"][:, cols])
    if name == "edge_hidden.npz":
        assert vis.shape == (0,) and cov.shape == (0,)           # np.array([]) like the reference


def test_far_pixel(R):
    g = load_golden("edge_far.npz")
    ins, clo = R.insertion, R.tools.closing
    sc = ins.add_space_for_spherical(g["scene5"])
    sc, mx, mn = ins.fill_spherical(sc)
    tr, lb, sc = ins.geometrical_front_view(sc, 112, 1440, mx, mn)
    assert (tr > 500).any()
    assert np.array_equal(tr, g["scene_train_raw"]) and np.array_equal(lb, g["scene_label_raw"])
    tr2, lb2 = clo.smooth_out(tr, lb)
    assert np.array_equal(tr2, g["scene_train"]) and np.array_equal(lb2, g["scene_label"])


def test_c1_120k(R, synth):
    g = load_golden("c1_120k.npz")
    ins, clo = R.insertion, R.tools.closing
    xyzi, label = synth.make_scene(int(g["scene_seed"]))
    sc = ins.add_space_for_spherical(synth.scene5_from_packed(xyzi, label))
    sc, mx, mn = ins.fill_spherical(sc)
    assert np.abs(sc[:, 5] - g["scene_el"]).max() <= ANGLE_TOL and np.abs(sc[:, 4] - g["scene_az"]).max() <= ANGLE_TOL
    tr, lb, sc = ins.geometrical_front_view(sc, 112, 1440, mx, mn)
    assert np.array_equal(sc[:, 8].astype(np.int32), g["scene_pix"])
    assert np.array_equal(tr, g["scene_train_raw"])
    tr, lb = clo.smooth_out(tr, lb)
    assert np.array_equal(tr, g["scene_train"]) and np.array_equal(lb.astype(np.int8), g["scene_label"])
    sm = ins.add_space_for_spherical(g["sample5"])
    sm, _, _ = ins.fill_spherical(sm)
    mtr, mlb, sm = ins.geometrical_front_view(sm, 112, 1440, mx, mn, sample=True)
    assert np.array_equal(sm[:, 8].astype(np.int32), g["sample_pix"])
    mtr, mlb = clo.smooth_out(mtr, mlb)
    assert np.array_equal(mtr, g["sample_train"])
    rr, cc = np.nonzero(mtr < tr)
    assert np.array_equal(rr, g["vis_rows"]) and np.array_equal(cc, g["vis_cols"])
    out, vis, _ = ins.occlusion_merge(sc, sm, tr, mtr)
    assert np.array_equal(out[:, :3], sc[g["keep_idx"], :3])
    assert np.array_equal(vis[:, :3], g["sample5"][g["visible_idx"], :3])


def test_random_grids_closing_and_merge_vs_oracle(R):
    rng = np.random.default_rng(17)
    lab = np.where(rng.random((112, 1440)) < 0.3, 1.0, -1.0)
    tr = np.where(lab == 1, rng.uniform(1, 80, lab.shape), 500.0)
    assert np.array_equal(R.tools.closing.class_closing(lab), O.class_closing(lab))
    a, b = R.tools.closing.smooth_out(tr, lab)
    c, d = O.smooth_out(tr, lab)
    assert np.array_equal(a, c) and np.array_equal(b, d)
    # odd shape, borders everywhere
    lab2 = np.where(rng.random((7, 37)) < 0.4, 1.0, -1.0)
    tr2 = np.where(lab2 == 1, rng.uniform(1, 80, lab2.shape), 500.0)
    a, b = R.tools.closing.smooth_out(tr2, lab2)
    c, d = O.smooth_out(tr2, lab2)
    assert np.array_equal(a, c) and np.array_equal(b, d)
    # merge with heavy culling (sample in front of a third of the image)
    n, m = 30000, 5000
    sc = np.full((n, 9), -1.0)
    sc[:, :3] = rng.normal(size=(n, 3))
    sc[:, 6] = rng.random(n)
    sc[:, 8] = rng.integers(0, 112 * 1440, n)
    sm = np.full((m, 9), -1.0)
    sm[:, :3] = rng.normal(size=(m, 3))
    sm[:, 8] = np.where(rng.random(m) < 0.1, -1, rng.integers(0, 112 * 1440, m))
    mt = np.where(rng.random((112, 1440)) < 0.33, 1.0, 500.0)
    got = R.insertion.occlusion_merge(sc, sm, tr, mt)
    want = O.occlusion_merge(sc, sm, tr, mt)
    for x, y in zip(got, want):
        assert np.array_equal(x, y)


def test_error_behaviour(R):
    ins = R.insertion
    with pytest.raises(ValueError):
        ins.fill_spherical(np.zeros((0, 9)))
    pc = ins.add_space_for_spherical(np.array([[1.0, 0, 0, 0, 0], [0, 1.0, 1.0, 0, 0], [5.0, 2, -1, 0, 0]]))
    pc, mx, mn = ins.fill_spherical(pc)
    with pytest.raises(AssertionError):                      # a scene point outside the bounds given
        ins.geometrical_front_view(pc, 112, 1440, mx - 0.2, mn)
    ins.geometrical_front_view(pc, 112, 1440, mx - 0.2, mn, sample=True)   # same call is fine for a sample
    origin = ins.add_space_for_spherical(np.zeros((2, 5)))
    origin, mx, mn = ins.fill_spherical(origin)
    assert np.isnan(mx) and np.isnan(mn)                     # r = 0: NaN bounds like NumPy's


def test_save_bytes(R, tmp_path):
    g = load_golden("chain_c20k.npz")
    ds = R.tools.datasets
    merged = np.full((len(g["merged"]), 9), -1.0)
    merged[:, [0, 1, 2, 6, 7]] = g["merged"]
    added = np.full((len(g["all_visible"]), 9), -1.0)
    added[:, [0, 1, 2, 6, 7]] = g["all_visible"]
    ds.SemanticKITTI({"path": {"output_path": str(tmp_path)}}).save_data(merged, added, "out", "000000", 0)
    assert (tmp_path / "out/velodyne/000000.bin").read_bytes() == g["velodyne_bin"].tobytes()
    assert (tmp_path / "out/labels/000000.label").read_bytes() == g["label_bin"].tobytes()
    assert (tmp_path / "out/check/000000.bin").read_bytes() == g["check_bin"].tobytes()
    g = load_golden("chain_c8k_od.npz")
    merged = np.full((len(g["merged"]), 9), -1.0)
    merged[:, [0, 1, 2, 6, 7]] = g["merged"]
    added = np.full((len(g["all_visible"]), 9), -1.0)
    added[:, [0, 1, 2, 6, 7]] = g["all_visible"]
    (tmp_path / "kitti" / "label_2").mkdir(parents=True)
    (tmp_path / "kitti" / "label_2" / "000001.txt").write_text("Car 0 0 0 1 2 3 4 1.5 1.6 3.9 1 2 30 0.1\n")
    ds.KITTI({"path": {"output_path": str(tmp_path), "dataset_path": str(tmp_path / "kitti")}}).save_data(merged, added, "od", "000001", 0, [])
    assert (tmp_path / "od/velodyne/000001.bin").read_bytes() == g["velodyne_bin"].tobytes()
    assert (tmp_path / "od/check/000001.bin").read_bytes() == g["check_bin"].tobytes()
    assert (tmp_path / "od/label_2/000001.txt").read_text() == "Car 0 0 0 1 2 3 4 1.5 1.6 3.9 1 2 30 0.1\n"
    with pytest.raises(KeyError):                           # the reference's constructor needs dataset_path as well
        ds.KITTI({"path": {"output_path": str(tmp_path)}}).save_data(merged, added, "od", "000002", 0, [])


def test_save_data_waymo(R, tmp_path):
    """SS tools/datasets.py:287-303: LiDAR offset added back in float64, float32 / uint32 casts,
    three .npy files (restated here with NumPy: the casts are plain astype calls)."""
    g = load_golden("chain_c20k.npz")
    merged = np.full((len(g["merged"]), 9), -1.0)
    merged[:, [0, 1, 2, 6, 7]] = g["merged"]
    merged[:, :3] -= np.array([1.22, 0, 2])                     # what __getitem__ leaves (:259): genuine float64
    added = np.full((len(g["all_visible"]), 9), -1.0)
    added[:, [0, 1, 2, 6, 7]] = g["all_visible"]
    added[:, :3] -= np.array([1.22, 0, 2])
    keep_m, keep_a = merged.copy(), added.copy()
    R.tools.datasets.Waymo({"path": {"output_path": str(tmp_path)}}).save_data(merged, added, "w", "f0", 0)
    assert np.array_equal(merged, keep_m) and np.array_equal(added, keep_a)
    want = keep_m.copy()
    want[:, :3] += np.array([1.22, 0, 2])
    wadd = keep_a.copy()
    wadd[:, :3] += np.array([1.22, 0, 2])
    lidar = np.load(tmp_path / "w/lidar/f0.npy")
    labels = np.load(tmp_path / "w/labels_v3_2/f0.npy")
    check = np.load(tmp_path / "w/check/f0.npy")
    assert lidar.dtype == np.float32 and np.array_equal(lidar, want[:, [0, 1, 2, 6]].astype(np.float32))
    assert labels.dtype == np.uint32 and labels.shape == (len(want), 1)
    assert np.array_equal(labels[:, 0], want[:, 7].astype(np.uint32))
    assert check.dtype == np.float32 and np.array_equal(check, wadd[:, [0, 1, 2, 6, 7]].astype(np.float32))


def test_save_data_writes_the_references_bytes(R, tmp_path):
    """Row a9: the three dataset flavours of ``save_data`` write, from the merged N x 9 cloud, the
    bytes the REFERENCE's own ``save_data`` wrote (fixture from tests/golden/make_golden_save.py):
    SemanticKITTI .bin/.label/check, Waymo .npy files, KITTI .bin/check + label_2 with the annotation
    lines of the inserted objects appended."""
    g = load_golden("save_data.npz")
    ds = R.tools.datasets
    merged9, allvis9 = g["merged9"], g["allvis9"]
    out = str(tmp_path)
    ds.SemanticKITTI({"path": {"output_path": out}}).save_data(merged9.copy(), allvis9.copy(), "ss", "000000", 0)
    for sub, ext, key in (("velodyne", "bin", "semantic_0"), ("labels", "label", "semantic_1"), ("check", "bin", "semantic_2")):
        assert (tmp_path / "ss" / sub / f"000000.{ext}").read_bytes() == g[key].tobytes()
    ds.Waymo({"path": {"output_path": out}}).save_data(merged9.copy(), allvis9.copy(), "wy", "000000", 0)
    for sub, key in (("lidar", "waymo_0"), ("labels_v3_2", "waymo_1"), ("check", "waymo_2")):
        assert (tmp_path / "wy" / sub / "000000.npy").read_bytes() == g[key].tobytes()
    (tmp_path / "in" / "label_2").mkdir(parents=True)
    (tmp_path / "in" / "label_2" / "000000.txt").write_bytes(g["label_2_in"].tobytes())
    kitti = ds.KITTI({"path": {"output_path": out, "dataset_path": str(tmp_path / "in")}})
    kitti.save_data(merged9.copy(), allvis9.copy(), "od", "000000", 0, [str(x) for x in g["anno_lines"]])
    assert (tmp_path / "od" / "velodyne" / "000000.bin").read_bytes() == g["kitti_0"].tobytes()
    assert (tmp_path / "od" / "check" / "000000.bin").read_bytes() == g["kitti_1"].tobytes()
    assert (tmp_path / "od" / "label_2" / "000000.txt").read_bytes() == g["kitti_label_2"].tobytes()
    assert not list(tmp_path.rglob("*.tmp"))


@pytest.mark.parametrize("rows,cols", [(64, 2048), (448, 2880)])
def test_another_grid_by_editing_the_two_globals(R, synth, monkeypatch, rows, cols):
    """The reference's grid is its two module globals (insertion.py:22-23): the pixel id of column 8 multiplies by the
    global NUMCOLUMN whatever num_column is passed (:116, :127, :470).  The mirror keeps the globals with that meaning
    (r3d_geometrical_front_view_grid / r3d_occlusion_merge_grid underneath): function by function against the oracle
    with both edited, on grids wider than the 1440 columns the plain C entry points are fixed at."""
    ins, clo = R.insertion, R.tools.closing
    monkeypatch.setattr(O, "NUMROW", rows)
    monkeypatch.setattr(O, "NUMCOLUMN", cols)
    monkeypatch.setattr(ins, "NUMROW", rows)
    monkeypatch.setattr(ins, "NUMCOLUMN", cols)
    xyzi, label = synth.make_scene(300 + rows, 48, 800, shuffle=True)
    s5 = synth.scene5_from_packed(xyzi, label)
    sample5 = synth.make_insert(9, "car", centre_range=6.0, centre_az=1.0)
    sc, osc = ins.add_space_for_spherical(s5), O.add_space_for_spherical(s5)
    sc, mx, mn = ins.fill_spherical(sc)
    osc, omx, omn = O.fill_spherical(osc)
    assert mx == omx and mn == omn
    tr, lb, sc = ins.geometrical_front_view(sc, rows, cols, mx, mn)
    otr, olb, osc = O.geometrical_front_view(osc, rows, cols, omx, omn)
    assert np.array_equal(sc[:, 8], osc[:, 8]) and sc[:, 8].max() >= 1440 * rows   # ids beyond the fixed stride's range
    assert np.array_equal(tr, otr) and np.array_equal(lb, olb)
    tr, lb = clo.smooth_out(tr, lb)
    otr, olb = O.smooth_out(otr, olb)
    assert np.array_equal(tr, otr)
    sm, osm = ins.add_space_for_spherical(sample5), O.add_space_for_spherical(sample5)
    sm, _, _ = ins.fill_spherical(sm)
    osm, _, _ = O.fill_spherical(osm)
    mtr, mlb, sm = ins.geometrical_front_view(sm, rows, cols, mx, mn, sample=True)
    omtr, omlb, osm = O.geometrical_front_view(osm, rows, cols, omx, omn, sample=True)
    assert np.array_equal(sm[:, 8], osm[:, 8])
    mtr, _ = clo.smooth_out(mtr, mlb)
    omtr, _ = O.smooth_out(omtr, omlb)
    out, vis, cov = ins.occlusion_merge(sc, sm, tr, mtr)
    oout, ovis, ocov = O.occlusion_merge(osc, osm, otr, omtr)
    assert len(vis) > 50 and len(cov) > 0
    assert out.shape == oout.shape and vis.shape == ovis.shape and cov.shape == ocov.shape, (out.shape, oout.shape, vis.shape, ovis.shape, cov.shape, ocov.shape)
    for got, want in ((vis, ovis), (cov, ocov), (out, oout)):                 # rows and order exact; the angles to 1e-12 rad
        assert np.array_equal(got[:, [0, 1, 2, 3, 6, 7, 8]], want[:, [0, 1, 2, 3, 6, 7, 8]])
        assert np.abs(got[:, 4:6] - want[:, 4:6]).max() <= ANGLE_TOL
