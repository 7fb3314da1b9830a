"""Randomised placed batches through both routes of PlacedInserter -- the search on the scenes where they stand in the batch
(R3D_PQ_SCENE_SLAB / R3D_PQ_ORIG_SLAB) and the search on exported float64 rows -- compared with each other: rotations,
counts of possible placements, the scenes' boxes, the merged clouds, labels and check rows, bit for bit.  (Both routes
against the oracle chain: tests/test_gpu_places.py.)  Ragged batches, 16 / 32 / 64 beams, frames in ring order and in a
random point order, 0-8 annotated boxes, 3-6 insert slots of random classes and thresholds, candidate windows of 2 / 8,
with and without the reference's rejected-candidate state.

    python tools/fuzz_placed.py [trials] [first_seed]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")
synth, fs = pkg.synth, pkg.Real3DAug.tools.find_spot
config = {"insertion": {"placement": synth.PLACEMENT, "placement_labels": synth.PLACEMENT_LABELS}}
KINDS = ["car", "pedestrian", "cyclist"]


def one_trial(seed):
    rng = np.random.default_rng(seed)
    B = int(rng.integers(3, 10))
    frames = []
    for s in range(B):
        f = synth.make_place_frame(seed * 100 + s, n_boxes=int(rng.integers(0, 9)), n_beams=int(rng.choice([16, 32, 64])),
                                   n_az=int(rng.integers(300, 1876)))
        if rng.random() < 0.3:                                        # a frame whose points come in no file order
            perm = rng.permutation(len(f["xyzi"]))
            f["xyzi"], f["label"] = np.ascontiguousarray(f["xyzi"][perm]), np.ascontiguousarray(f["label"][perm])
        frames.append(f)
    K = int(rng.integers(3, 7))
    slots, needs = [], []
    for k in range(K):
        smp, annos, okl, okm = [], [], [], []
        for s in range(B):
            if rng.random() < 0.1:                                    # no sample for this scene in this slot
                smp.append(None); annos.append(np.zeros(10)); okl.append([]); okm.append([])
                continue
            pts, line = synth.make_place_sample(seed * 1000 + s * 10 + k, KINDS[int(rng.integers(0, 3))])
            sa = fs.read_label_line(line)
            m, l = fs.placement_surfaces(sa, config)
            smp.append(pts); annos.append(fs._anno10(sa)); okl.append(l); okm.append(m)
        slots.append((smp, annos, okl, okm))
        needs.append([int(rng.choice([5, 20, 60, 10 ** 6])) for _ in range(B)])
    rejected_state, chunk = bool(rng.random() < 0.5), int(rng.choice([2, 8]))
    last_try = [bool(rng.random() < 0.7) for _ in range(K)]
    grow = sum(max((len(x) for x in sl[0] if x is not None), default=0) for sl in slots)
    n = max(len(f["xyzi"]) for f in frames)
    got = {}
    for slab in (True, False):
        batch = pkg.SceneBatch(B, n + grow + 64, grow + 64)
        batch.load([(f["xyzi"], f["label"]) for f in frames])
        batch.begin()
        ins = pkg.PlacedInserter(batch, *[[f[k] for f in frames] for k in ("rich", "move", "pose", "boxes")],
                                 reference_rejected_state=rejected_state, scene_slab=slab)
        assert ins.slab == slab
        seen = [ins.insert_slot(smp, annos, okl, okm, needs[k], chunk=chunk, last_try=last_try[k])
                for k, (smp, annos, okl, okm) in enumerate(slots)]
        batch.finish()
        got[slab] = (seen, [tuple(a.tobytes() for a in r) for r in batch.results()], [b.copy() for b in ins.boxes])
    same = got[True][0] == got[False][0] and got[True][1] == got[False][1] and all(
        np.array_equal(a, b) for a, b in zip(got[True][2], got[False][2]))
    placed = sum(r > 0 for slot in got[True][0] for r in slot[0])
    windows = sum(n_p > chunk for slot in got[True][0] for n_p in slot[1])
    return same, B * K, placed, windows


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    t0, bad, tried, placed, windows = time.time(), [], 0, 0, 0
    for t in range(trials):
        same, n_tried, n_placed, n_windows = one_trial(seed0 + t)
        tried, placed, windows = tried + n_tried, placed + n_placed, windows + n_windows
        if not same:
            bad.append(seed0 + t)
        if t % 5 == 4:
            print(f"{t + 1} batches, {int(time.time() - t0)} s", flush=True)
    print(f"{trials} batches, {tried} (scene, slot) pairs, {placed} objects placed, {windows} pairs with more placements than one "
          f"window of candidates: {len(bad)} batches differ between the routes {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
