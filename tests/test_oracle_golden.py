"""The CPU oracle against the golden vectors captured from the reference (not gpu).

Every intermediate of one insert step, the edge cases, the K-insert chains and the written
.bin/.label bytes must be array_equal to what the reference's own functions produced
(tests/golden/make_golden.py).  Spherical coordinates are compared exactly too: oracle and
golden were both evaluated by NumPy; when the host's libm/SVML differs from the generating
host's they may differ by an ULP, so that one comparison falls back to 1e-12.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import real3d_oracle as O

STEP = ["step_s2k.npz", "step_s8k.npz", "step_s20k.npz"]
EDGE = ["edge_above.npz", "edge_below.npz", "edge_hidden.npz", "edge_seam.npz"]


def _scene9(g):
    s5 = np.hstack((g["in_xyzi"].astype(np.float64), g["in_label"].astype(np.float64)[:, None]))
    return O.add_space_for_spherical(s5)


def _close(a, b):
    return np.array_equal(a, b) or np.allclose(a, b, rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", STEP)
def test_step_intermediates(name):
    g = load_golden(name)
    sc = _scene9(g)
    sc, mx, mn = O.fill_spherical(sc)
    assert _close(sc[:, 3:6], g["scene_sph"])
    assert np.array_equal(sc[:, 3], g["scene_sph"][:, 0])          # r: sqrt is exact everywhere
    assert _close(np.array([mx, mn]), g["bounds"])
    # bin with the golden angles so an ULP in the host's atan2/acos cannot move a pixel
    sc[:, 3:6] = g["scene_sph"]
    mx, mn = g["bounds"]
    tr, lb, sc = O.geometrical_front_view(sc, O.NUMROW, O.NUMCOLUMN, mx, mn)
    assert np.array_equal(sc[:, 8].astype(np.int64), g["scene_pix"])
    assert np.array_equal(tr, g["scene_train_raw"])
    assert np.array_equal(lb, g["scene_label_raw"])
    assert np.array_equal(O.class_closing(lb), g["scene_closed_u8"])
    tr2, lb2 = O.smooth_out(tr, lb)
    assert np.array_equal(tr2, g["scene_train"])
    assert np.array_equal(lb2, g["scene_label"])
    sm = O.add_space_for_spherical(g["sample5"])
    sm, _, _ = O.fill_spherical(sm)
    assert _close(sm[:, 3:6], g["sample_sph"])
    sm[:, 3:6] = g["sample_sph"]
    mtr, mlb, sm = O.geometrical_front_view(sm, O.NUMROW, O.NUMCOLUMN, mx, mn, sample=True)
    assert np.array_equal(sm[:, 8].astype(np.int64), g["sample_pix"])
    assert np.array_equal(mtr, g["sample_train_raw"])
    mtr2, mlb2 = O.smooth_out(mtr, mlb)
    assert np.array_equal(mtr2, g["sample_train"])
    assert np.array_equal(mlb2, g["sample_label"])
    rr, cc = np.nonzero(mtr2 < tr2)
    assert np.array_equal(rr, g["vis_rows"]) and np.array_equal(cc, g["vis_cols"])
    out, vis, cov = O.occlusion_merge(sc, sm, tr2, mtr2)
    assert np.array_equal(out, g["scene_out"])
    assert np.array_equal(vis.reshape(-1, 9), g["visible_sample"])
    assert np.array_equal(cov.reshape(-1, 9), g["covered_scene"])


@pytest.mark.parametrize("name", EDGE)
def test_edge_cases(name):
    g = load_golden(name)
    sc = _scene9(g)
    sc, tr, lb, mx, mn = O.scene_field_of_view(sc)
    assert _close(np.array([mx, mn]), g["bounds"])
    assert np.array_equal(sc[:, 8].astype(np.int64), g["scene_pix"])
    assert np.array_equal(tr, g["scene_train"]) and np.array_equal(lb, g["scene_label"])
    out, vis, cov = O.evaluate_candidate(sc, tr, mx, mn, g["sample5"])
    assert np.array_equal(out[:, [0, 1, 2, 6, 7, 8]], g["scene_out"][:, [0, 1, 2, 6, 7, 8]])
    assert np.array_equal(vis.reshape(-1, 9)[:, [0, 1, 2, 6, 7, 8]], g["visible_sample"][:, [0, 1, 2, 6, 7, 8]])
    assert np.array_equal(cov.reshape(-1, 9)[:, [0, 1, 2, 6, 7, 8]], g["covered_scene"][:, [0, 1, 2, 6, 7, 8]])


def test_edge_semantics_documented():
    """What the edge fixtures actually exercise (guards the generator against drifting)."""
    g = load_golden("edge_above.npz")
    assert (g["sample_pix"] == -1).any() and (g["sample_pix"] >= 0).any()      # some rows dropped
    g = load_golden("edge_below.npz")
    assert (g["sample_pix"] == -1).any()
    g = load_golden("edge_hidden.npz")
    assert len(g["visible_sample"]) == 0 and len(g["vis_rows"]) == 0
    g = load_golden("edge_seam.npz")
    cols = g["sample_pix"][g["sample_pix"] >= 0] % O.NUMCOLUMN
    assert (cols == 0).any() and (cols == O.NUMCOLUMN - 1).any()
    assert (g["sample_pix"][-2:] % O.NUMCOLUMN == 0).all()                     # az = 2pi and az = 0


def test_far_pixel_first_hit_overwrites_500():
    g = load_golden("edge_far.npz")
    sc = O.add_space_for_spherical(g["scene5"])
    sc, mx, mn = O.fill_spherical(sc)
    tr, lb, sc = O.geometrical_front_view(sc, O.NUMROW, O.NUMCOLUMN, mx, mn)
    assert (tr > 500).any()
    assert np.array_equal(tr, g["scene_train_raw"]) and np.array_equal(lb, g["scene_label_raw"])
    tr2, lb2 = O.smooth_out(tr, lb)
    assert np.array_equal(tr2, g["scene_train"]) and np.array_equal(lb2, g["scene_label"])


@pytest.mark.parametrize("name,od", [("chain_c20k.npz", False), ("chain_c8k_od.npz", True)])
def test_chain_and_bytes(name, od):
    g = load_golden(name)
    s5 = np.hstack((g["in_xyzi"].astype(np.float64), g["in_label"].astype(np.float64)[:, None]))
    cuts = np.cumsum(g["sample_sizes"])[:-1]
    samples = np.split(g["samples"], cuts)
    merged, allvis, acc = O.augment_scene(s5, [[s] for s in samples], list(g["min_points"]))
    assert np.array_equal(np.array([a >= 0 for a in acc], dtype=np.int32), g["accepted"])
    assert 0 in g["accepted"] and 1 in g["accepted"]
    assert np.array_equal(merged[:, [0, 1, 2, 6, 7]], g["merged"])
    assert np.array_equal(merged[:, 8].astype(np.int64), g["merged_pix"])
    assert np.array_equal(allvis[:, [0, 1, 2, 6, 7]], g["all_visible"])
    if od:
        vb, cb = O.save_bytes_kitti(merged, allvis)
    else:
        vb, lb, cb = O.save_bytes_semantic(merged, allvis)
        assert lb == g["label_bin"].tobytes()
    assert vb == g["velodyne_bin"].tobytes() and cb == g["check_bin"].tobytes()


def test_chain_on_float64_cloud():
    """The oracle on a frame with genuine float64 coordinates (the Waymo flavour) equals the reference's chain
    (tests/golden/make_golden_waymo.py)."""
    g = load_golden("chain_waymo_f64.npz")
    samples = np.split(g["samples"], np.cumsum(g["sample_sizes"])[:-1])
    merged, allvis, acc = O.augment_scene(g["scene5"], [[s] for s in samples], list(g["min_points"]))
    assert np.array_equal(np.array([a >= 0 for a in acc], dtype=np.int32), g["accepted"]) and 0 in g["accepted"]
    assert np.array_equal(merged[:, [0, 1, 2, 6, 7]], g["merged"])
    assert np.array_equal(allvis[:, [0, 1, 2, 6, 7]], g["all_visible"])
    assert not np.array_equal(g["scene5"][:, :3], g["scene5"][:, :3].astype(np.float32))


def test_c1_120k(synth):
    g = load_golden("c1_120k.npz")
    xyzi, label = synth.make_scene(int(g["scene_seed"]))
    sc = O.add_space_for_spherical(synth.scene5_from_packed(xyzi, label))
    sc, tr, lb, mx, mn = O.scene_field_of_view(sc)
    assert _close(np.array([mx, mn]), g["bounds"])
    assert np.array_equal(sc[:, 8].astype(np.int32), g["scene_pix"])
    assert np.array_equal(tr, g["scene_train"]) and np.array_equal(lb.astype(np.int8), g["scene_label"])
    out, vis, _ = O.evaluate_candidate(sc, tr, mx, mn, g["sample5"])
    assert np.array_equal(out[:, :3], sc[g["keep_idx"], :3])
    assert len(out) == len(g["keep_idx"]) and len(vis) == len(g["visible_idx"])
    assert np.array_equal(vis[:, :3], g["sample5"][g["visible_idx"], :3])


def test_loop_forms_pin_vector_forms():
    """Statement-by-statement restatements == vectorised forms on a small random case."""
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(600, 5)) * [8, 8, 1.0, 1, 1]
    pts[:, 4] = rng.integers(0, 60, 600)
    a = O.add_space_for_spherical(pts)
    a, mx, mn = O.fill_spherical(a)
    b = a.copy()
    t1, l1, a = O.geometrical_front_view(a, 24, 90, mx, mn)
    t2, l2, b = O.geometrical_front_view_loop(b, 24, 90, mx, mn)
    assert np.array_equal(t1, t2) and np.array_equal(l1, l2) and np.array_equal(a, b)
    s1 = O.smooth_out(t1, l1)
    s2 = O.smooth_out_loop(t1, l1)
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1])
    # merge forms need NUMROW x NUMCOLUMN grids (pixel ids use the global constant)
    a, mx, mn = O.fill_spherical(O.add_space_for_spherical(pts))
    t, l, a = O.geometrical_front_view(a, O.NUMROW, O.NUMCOLUMN, mx, mn)
    smp = pts[:150].copy()
    smp[:, :3] *= 0.8
    m = O.add_space_for_spherical(smp)
    m, _, _ = O.fill_spherical(m)
    mt, ml, m = O.geometrical_front_view(m, O.NUMROW, O.NUMCOLUMN, mx, mn, sample=True)
    r1 = O.occlusion_merge(a, m, t, mt)
    r2 = O.occlusion_merge_loop(a, m, t, mt)
    assert len(r1[1]) > 0
    for x, y in zip(r1, r2):
        assert np.array_equal(x, y)


def test_closing_matches_scipy_definition():
    from scipy import ndimage as ndi
    rng = np.random.default_rng(3)
    lab = np.where(rng.random((40, 70)) < 0.25, 1.0, -1.0)
    fp = np.ones((5, 3), dtype=np.uint8)
    img = (np.clip(lab, 0, 1) * 255).astype(np.uint8)
    want = ndi.grey_erosion(ndi.grey_dilation(img, footprint=fp), footprint=fp)
    assert np.array_equal(O.class_closing(lab), want)
    want2 = ndi.binary_erosion(ndi.binary_dilation(img > 0, structure=fp), structure=fp, border_value=1)
    assert np.array_equal(O.class_closing(lab) == 255, want2)


def test_annotation_lines_and_label_2_match_the_reference(pkg, tmp_path):
    """Row a9, object detection: ``create_annotation_line`` (OD insertion.py:227-265) and
    ``create_annotation`` (OD tools/datasets.py:20-37) restated on the host give the strings / bytes
    the reference's functions produced (tests/golden/make_golden_save.py)."""
    g = load_golden("save_data.npz")
    ins, ds = pkg.Real3DAug.insertion, pkg.Real3DAug.tools.datasets
    lines = []
    for o, c, r, cl, want in zip(g["anno_originals"], g["anno_centres"], g["anno_rotations"], g["anno_classes"], g["anno_lines"]):
        anno = {"center": {"x": float(c[0]), "y": float(c[1]), "z": float(c[2])}, "class": str(cl)}
        line = ins.create_annotation_line(np.array(str(o)), anno, int(r))
        assert line == str(want)
        lines.append(line)
    old = tmp_path / "000000.txt"
    old.write_bytes(g["label_2_in"].tobytes())
    ds.create_annotation(str(old), str(tmp_path / "new.txt"), lines)
    assert (tmp_path / "new.txt").read_bytes() == g["kitti_label_2"].tobytes()
    # wrap-around of the two angles
    far = ins.create_annotation_line(np.array("Car 0 0 0 1 2 3 4 1.5 1.6 3.9 0 0 0 -3.0"), {"center": {"x": -5.0, "y": 0.1, "z": -1.0}, "class": "Car"}, 90)
    assert far.split(" ")[14].strip() == f"{-3.0 - np.pi / 2 + 2 * np.pi:.02f}"


def test_od_rich_map_oracle_matches_the_reference_script():
    """Row f-4, object detection: the oracle's per-frame road / pedestrian maps equal the maps the reference's
    single_drivable_area_map.py saved (tests/golden/make_golden_map_od.py)."""
    from oracle import rich_map_oracle as M
    g = load_golden("rich_map_od.npz")
    for f in range(3):
        road, ped, mx, my = M.od_maps(g[f"xyzi{f}"], g[f"label{f}"], int(g["road_label"]))
        assert (mx, my) == tuple(g[f"min{f}"])
        assert road.dtype == g[f"road{f}"].dtype and np.array_equal(road, g[f"road{f}"])
        assert np.array_equal(ped, g[f"ped{f}"])
