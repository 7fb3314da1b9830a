"""A fixed slice of the randomised parity campaign (tools/fuzz_parity.py) inside the GPU suite: random scenes, grids,
chains with clustered placements, far points, duplicated points, samples outside the field of view, several candidates
per slot -- HIP path against the oracle, byte for byte, on the default insert launch, the three-kernel launch and
without speculation.  The long campaign is the tool; its log of the round is quoted in DESIGN.md par.5."""
import importlib
import os
import sys

import numpy as np
import pytest

from oracle import real3d_oracle as O

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed0", [1000, 1040])
def test_random_batches_equal_the_oracle(pkg, synth, monkeypatch, seed0):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    F = importlib.import_module("fuzz_parity")
    scenes = 0
    for seed in range(seed0, seed0 + 10):
        rng = np.random.default_rng(seed)
        rows, cols = F.GRIDS[int(rng.integers(len(F.GRIDS)))]
        B = int(rng.choice([1, 2, 3, 6, 9]))
        if rows * cols > 128 * 2048:
            rows, cols = 128, 2048                            # (the oracle needs seconds per scene on 448 x 2880)
        cases = [F.make_case(synth, rng, rows, cols) for _ in range(min(B, 3))]
        debug = int(rng.choice([0, 0, 0, 64, 2]))              # (as the tool drew it when the slice was fixed)
        monkeypatch.setattr(O, "NUMROW", rows)
        monkeypatch.setattr(O, "NUMCOLUMN", cols)
        res, acc = pkg.augment_batch([(c[0], c[1]) for c in cases], [c[2] for c in cases], [c[3] for c in cases],
                                     rows=rows, cols=cols, debug=debug)
        for i, c in enumerate(cases):
            vb, lb, cb, oacc = F.oracle_case((rows, cols) + c)
            assert list(acc[i]) == list(oacc), (seed, i)
            assert res[i][0].tobytes() == vb and res[i][1].tobytes() == lb and res[i][2].tobytes() == cb, (seed, i)
            scenes += 1
    assert scenes >= 10


def _near_car_cases(synth):
    """Seed 5485 of the campaign: two frames on 448 x 2880, one with an object so near that its window exceeds a CU's LDS."""
    F = importlib.import_module("fuzz_parity")
    rng = np.random.default_rng(5485)
    rows, cols = F.GRIDS[int(rng.integers(len(F.GRIDS)))]
    B = int(rng.choice([1, 2, 3, 6, 9]))
    assert (rows, cols, B) == (448, 2880, 2)
    return F, rows, cols, [F.make_case(synth, rng, rows, cols) for _ in range(B)]


def test_window_beyond_the_lds_goes_through_the_level1_kernels(pkg, synth, monkeypatch):
    """R3D_S_WINDOW_TOO_LARGE is not the caller's problem: augment_batch runs such a frame once more through the Level-1
    kernels (level1.augment_scene), the streamed path does the same in collect -- same bytes as the oracle either way."""
    F, rows, cols, cases = _near_car_cases(synth)
    monkeypatch.setattr(O, "NUMROW", rows)
    monkeypatch.setattr(O, "NUMCOLUMN", cols)
    res, acc = pkg.augment_batch([(c[0], c[1]) for c in cases], [c[2] for c in cases], [c[3] for c in cases], rows=rows, cols=cols)
    assert pkg.SceneBatch.last_level1, "the case no longer exceeds the LDS: pick another seed"
    for i, c in enumerate(cases):
        vb, lb, cb, oacc = F.oracle_case((rows, cols) + c)
        assert list(acc[i]) == list(oacc)
        assert res[i][0].tobytes() == vb and res[i][1].tobytes() == lb and res[i][2].tobytes() == cb
    # the streamed path: one candidate per slot
    streaming = importlib.import_module("pcl-augmentation_amd.streaming")
    firsts = [(c[0], c[1], [[slot[0]] for slot in c[2]], c[3]) for c in cases]
    K = max(len(c[2]) for c in firsts)
    n_max = max(len(c[0]) for c in firsts)
    grow = max(sum(len(slot[0]) for slot in c[2]) for c in firsts)
    srows = max(sum(len(c[2][k][0]) for c in firsts if k < len(c[2])) for k in range(K))
    aug = streaming.StreamedAugmenter(len(firsts), n_max, grow, K, srows, lanes=1, rows=rows, cols=cols)
    aug.submit(0, [(c[0], c[1]) for c in firsts], [[slot[0] for slot in c[2]] for c in firsts], [c[3] for c in firsts])
    _, results, accepted = aug.collect(0)
    assert getattr(aug, "level1_frames", 0) >= 1
    for i, c in enumerate(firsts):
        vb, lb, cb, oacc = F.oracle_case((rows, cols) + c)
        assert [a for a in accepted[i][:len(oacc)]] == list(oacc)
        assert results[i][0].tobytes() == vb and results[i][1].tobytes() == lb and results[i][2].tobytes() == cb


def test_random_placement_queries_equal_the_oracle(pkg):
    """A fixed slice of tools/fuzz_places.py: twelve random placement queries, everything the search returns."""
    FP = importlib.import_module("fuzz_places")
    T = importlib.import_module("test_gpu_places")
    fs = pkg.Real3DAug.tools.find_spot
    specs = [FP.spec_of(3000 + i) for i in range(12)]
    queries = []
    for spec in specs:
        c = FP.make_case(spec)
        sa = fs.read_label_line(c["line"])
        ok_map, ok_labels = fs.placement_surfaces(sa, T.CONFIG)
        scene = pkg.PlaceScene(c["scene9"], c["original"], [fs._anno10(fs.read_label_line(l)) for l in c["lines"]], c["rich"],
                               c["move"], c["T"])
        queries.append({"scene": scene, "sample": c["sample"], "anno": fs._anno10(sa), "ok_labels": ok_labels, "ok_map": ok_map})
    placements = 0
    for spec, r in zip(specs, pkg.find_places(queries)):
        rot, clouds, centres, quats, off, hit = FP.oracle_case(spec)
        f = r["flags"]
        assert list(r["rotations"]) == rot, spec
        assert np.ascontiguousarray(r["clouds"]).tobytes() == clouds, spec
        assert np.ascontiguousarray(r["anno"][:, :3]).tobytes() == centres and np.ascontiguousarray(r["anno"][:, 3:]).tobytes() == quats, spec
        assert int(((f & 1) == 0).sum()) == off and int((((f & 3) == 3) & ((f & 16) == 0)).sum()) == hit, spec
        placements += len(rot)
    assert placements > 100


def test_random_rich_maps_equal_the_oracle(pkg, synth):
    """A fixed slice of tools/fuzz_rich_map.py: sequences under random poses and single object-detection frames."""
    FR = importlib.import_module("fuzz_rich_map")
    M = importlib.import_module("oracle.rich_map_oracle")
    labels = {1: [40], 2: [48, 72], 3: [44]}
    for seed in range(9000, 9012):
        rng = np.random.default_rng(seed)
        origin = FR.random_pose(rng)
        frames = []
        for _ in range(int(rng.integers(1, 6))):
            xyzi, label = FR.random_frame(synth, rng)
            T = origin.copy()
            T[:3, 3] += rng.uniform(-15, 15, 3) * np.array([1, 1, 0.02])
            frames.append((xyzi, label, T))
        area, move = pkg.build_rich_map(frames, labels)
        want, wmove = M.build_rich_map(frames, labels[1], labels[2], labels[3])
        assert np.array_equal(move, wmove) and area.shape == want.shape and np.array_equal(area, want), seed
        xyzi, label = FR.random_frame(synth, rng)
        got, wod = pkg.rich_map.build_od_maps(xyzi, label, 40), M.od_maps(xyzi, label, 40)
        assert got[2:] == wod[2:] and np.array_equal(got[0], wod[0]) and np.array_equal(got[1], wod[1]), seed
