"""The placement-search oracle (oracle/find_spot_oracle.py) against the fixtures captured from the
reference's own find_possible_places (tests/golden/make_golden_places.py), and its restated
third-party arithmetic against the libraries of this image."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import find_spot_oracle as F
from oracle import real3d_oracle as O

PLACEMENT = {11: [1, 3], 15: [1, 3], 18: [1, 3], 30: [2], 31: [1, 3], 32: [1, 3], 253: [1, 3], 255: [1, 3]}
PLACEMENT_LABELS = {1: [40, 60], 2: [48], 3: [44]}
CASES = ["places_cyclist.npz", "places_pedestrian.npz", "places_car_smallmap.npz", "places_one_point.npz"]


def oracle_inputs(g):
    original = np.hstack((g["xyzi"].astype(np.float64), g["label"].astype(np.float64)[:, None]))
    scene9 = O.add_space_for_spherical(np.vstack([original, g["extra"]]))
    annos = [F.read_label_line(str(l)) for l in g["anno_lines"]]
    return original, scene9, annos


@pytest.mark.parametrize("name", CASES)
def test_oracle_equals_reference_outputs(name):
    g = load_golden(name)
    original, scene9, annos = oracle_inputs(g)
    pcl, anno, rot, not_on_road, collisions = F.find_possible_places(
        scene9, annos, g["sample"], str(g["sample_line"]), g["rich"].astype(np.float64), g["move"], original, g["T"],
        PLACEMENT, PLACEMENT_LABELS)
    assert rot == list(g["out_rot"])
    assert 0 < len(rot) < 360 and not_on_road > 0 and collisions > 0
    assert np.array_equal(np.array(pcl), g["out_pcl"])                     # bit for bit
    assert np.array_equal(np.array([F.anno_center(a) for a in anno]), g["out_centre"])
    assert np.array_equal(np.array([F.anno_quat(a) for a in anno]), g["out_quat"])


def test_emulated_fma_is_exact():
    from fractions import Fraction as Fr
    rng = np.random.default_rng(0)
    n = 20000
    a = rng.normal(0, 1, n) * 10.0 ** rng.integers(-3, 3, n)
    b = rng.normal(0, 1, n)
    c = -a * b * (1 + rng.normal(0, 1e-3, n)) * rng.choice([1, 1, 1e-8, 1e8], n)
    got = F.fma(a, b, c)
    want = np.array([float(Fr(a[i]) * Fr(b[i]) + Fr(c[i])) for i in range(n)])
    assert np.array_equal(got, want)


def test_rotation_restatement_equals_scipy():
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(1)
    for _ in range(500):
        q = rng.normal(size=4)
        r = R.from_quat(q)
        mine = F.quat_normalize(q)
        assert np.array_equal(r.as_quat(), mine)
        assert np.array_equal(r.as_matrix(), F.quat_to_matrix(mine))
        assert np.array_equal(R.from_matrix(r.as_matrix()).as_quat(), F.matrix_to_quat(F.quat_to_matrix(mine)))


def test_mean_along_rows_is_a_sequential_sum():
    rng = np.random.default_rng(2)
    for n in (1, 2, 7, 8, 9, 130, 1000):
        a = rng.normal(-1.7, 0.05, size=(n, 5)).astype(np.float32).astype(np.float64)
        acc = 0.0
        for v in a[:, 2]:
            acc += float(v)
        assert np.mean(a, axis=0)[2] == acc / n


@pytest.mark.parametrize("name", ["places_od_car.npz", "places_od_pedestrian.npz"])
def test_od_oracle_equals_reference_outputs(name):
    """object_detection flavour (OD tools/find_spot.py:227-304)."""
    g = load_golden(name)
    original, scene9, _ = oracle_inputs({**g, "anno_lines": []})
    annos = [F.read_label_line_od(str(l)) for l in g["anno_lines"]]
    pcl, anno, rot, not_on_road, collisions = F.find_possible_places_od(
        scene9, annos, g["sample"], str(g["sample_line"]), g["rich"].astype(np.float64), g["move"], original, 40)
    assert rot == list(g["out_rot"]) and 0 < len(rot) < 360 and collisions > 0
    assert np.array_equal(np.array(pcl), g["out_pcl"])
    assert np.array_equal(np.array([F.anno_center(a) for a in anno]), g["out_centre"])
    assert np.array_equal(np.array([F.anno_quat(a) for a in anno]), g["out_quat"])


def test_rich_map_oracle_equals_the_reference_script():
    """oracle/rich_map_oracle.py against the map the reference's drivable_area_map.py wrote."""
    from oracle import rich_map_oracle as M
    g = load_golden("rich_map.npz")
    frames = [(g[f"xyzi{f}"], g[f"label{f}"], g["transforms"][f]) for f in range(len(g["transforms"]))]
    area, move = M.build_rich_map(frames, list(g["labels_road"]), list(g["labels_sidewalk"]), list(g["labels_parking"]))
    assert np.array_equal(move, g["move"])
    assert np.array_equal(area, g["map"].astype(np.float64))
    assert all((area == c).any() for c in range(4))
