"""The incremental algorithm (tests/incremental_model.py) == the oracle's literal chain (not gpu)."""
import numpy as np
import pytest

from conftest import blob_in_front_of_extreme, load_golden
from incremental_model import IncrementalScene
from oracle import real3d_oracle as O


def _run_both(xyzi, label, samples, need):
    s5 = np.hstack((xyzi.astype(np.float64), label.astype(np.float64)[:, None]))
    merged, allvis, acc = O.augment_scene(s5, [[s] for s in samples], need)
    inc = IncrementalScene(xyzi, label)
    got_acc = [inc.step(s, n) for s, n in zip(samples, need)]
    m2, _, log = inc.finalize()
    assert [a >= 0 for a in acc] == [g > 0 for g in got_acc]
    assert np.array_equal(merged[:, [0, 1, 2, 6, 7]], m2)
    assert np.array_equal(allvis[:, [0, 1, 2, 6, 7]], log[:, [0, 1, 2, 6, 7]])
    return inc


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_chains(synth, seed):
    xyzi, label = synth.make_scene(seed, 48, 700, shuffle=(seed == 2))
    kinds = ["pedestrian", "car", "cyclist", "car", "pedestrian", "car"]
    samples = [synth.make_insert(seed * 77 + k, kind, rng_range=(4.0, 14.0)) for k, kind in enumerate(kinds)]
    # overlapping inserts: later ones cull earlier ones (same azimuth, nearer)
    samples.append(synth.make_insert(5, "car", centre_range=9.0, centre_az=0.3))
    samples.append(synth.make_insert(6, "car", centre_range=6.0, centre_az=0.3))
    samples.append(synth.make_insert(7, "pedestrian", centre_range=4.0, centre_az=0.3))
    inc = _run_both(xyzi, label, samples, [20] * len(samples))
    assert inc.rebases == 0


def test_rebase_paths(synth):
    xyzi, label = synth.make_scene(9, 48, 700)
    tall = synth.make_insert(5, "pedestrian", centre_range=3.0)
    tall[:, 2] = tall[:, 2] * 3.0 + 2.0              # visible points above the top beam: min_el moves
    low = synth.make_insert(6, "car", centre_range=2.5)
    low[:, 2] -= 1.0                                 # reaches below the lowest beam: max_el moves
    cover_bottom = synth.make_insert(8, "car", centre_range=3.2, centre_az=1.0)   # culls last-row points
    later = synth.make_insert(10, "cyclist", centre_range=7.0)
    samples = [blob_in_front_of_extreme(xyzi, "max"), later * [1, -1, 1, 1, 1],
               blob_in_front_of_extreme(xyzi, "min"), synth.make_insert(12, "car", centre_range=8.0),
               synth.make_insert(3, "cyclist", centre_range=9.0), tall, later, low, cover_bottom,
               synth.make_insert(11, "pedestrian", centre_range=5.0, centre_az=1.0)]
    inc = _run_both(xyzi, label, samples, [10] * len(samples))
    assert inc.rebases >= 3


@pytest.mark.parametrize("name", ["chain_c20k.npz", "chain_c8k_od.npz"])
def test_golden_chains(name):
    g = load_golden(name)
    samples = np.split(g["samples"], np.cumsum(g["sample_sizes"])[:-1])
    inc = IncrementalScene(g["in_xyzi"], g["in_label"])
    acc = [inc.step(s, n) > 0 for s, n in zip(samples, g["min_points"])]
    merged, _, log = inc.finalize()
    assert np.array_equal(np.array(acc, dtype=np.int32), g["accepted"])
    assert np.array_equal(merged, g["merged"])
    assert np.array_equal(log[:, [0, 1, 2, 6, 7]], g["all_visible"])
