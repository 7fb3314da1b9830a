#!/usr/bin/env python3
"""Randomised parity campaign of the placement search (SS Real3DAug/tools/find_spot.py:42-192): random frames, poses,
rich maps, samples of the three classes, scene boxes and a blob in the way -- `find_places` on the GPU against
`oracle.find_spot_oracle.find_possible_places`: accepted rotations, the rotated clouds bit for bit, centre and quaternion
of every candidate, the counts of steps off the surface / in collision.  Test infrastructure (imports oracle/ and the
case generator of tests/test_gpu_places.py).

    python tools/fuzz_places.py [queries] [first_seed] [workers]
"""
import importlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def spec_of(seed):
    rng = np.random.default_rng(seed)
    cls = int(rng.choice([30, 31, 18]))
    n_boxes = int(rng.integers(0, 7))
    beams = int(rng.choice([16, 32, 64]))
    n_az = int(rng.integers(300, 1200))
    dist = None if rng.random() < 0.6 else float(rng.uniform(3.0, 25.0))
    return seed, cls, n_boxes, beams, n_az, dist


def make_case(spec):
    T = importlib.import_module("test_gpu_places")
    synth = importlib.import_module("pcl-augmentation_amd.synth")
    return T._random_query(synth, *spec)


def oracle_case(spec):
    F = importlib.import_module("oracle.find_spot_oracle")
    T = importlib.import_module("test_gpu_places")
    c = make_case(spec)
    annos = [F.read_label_line(l) for l in c["lines"]]
    pcl, anno, rot, not_on_road, collisions = F.find_possible_places(
        c["scene9"], annos, c["sample"], c["line"], c["rich"].astype(np.float64), c["move"], c["original"], c["T"],
        T.PLACEMENT, T.PLACEMENT_LABELS)
    m = len(c["sample"])
    return (list(rot), np.array(pcl).reshape(len(rot), m, 5).tobytes(),
            np.array([F.anno_center(a) for a in anno]).reshape(len(rot), 3).tobytes(),
            np.array([F.anno_quat(a) for a in anno]).reshape(len(rot), 4).tobytes(), int(not_on_road), int(collisions))


def main():
    n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    pool = mp.get_context("spawn").Pool(workers)              # before any GPU call in this process
    specs = [spec_of(seed0 + i) for i in range(n_q)]
    want = pool.map_async(oracle_case, specs, chunksize=4)
    pkg = importlib.import_module("pcl-augmentation_amd")
    T = importlib.import_module("test_gpu_places")
    fs = pkg.Real3DAug.tools.find_spot
    got, t0 = [], time.time()
    for i0 in range(0, n_q, 16):
        queries = []
        for spec in specs[i0:i0 + 16]:
            c = make_case(spec)
            sa = fs.read_label_line(c["line"])
            ok_map, ok_labels = fs.placement_surfaces(sa, T.CONFIG)
            scene = pkg.PlaceScene(c["scene9"], c["original"], [fs._anno10(fs.read_label_line(l)) for l in c["lines"]], c["rich"],
                                   c["move"], c["T"])
            queries.append({"scene": scene, "sample": c["sample"], "anno": fs._anno10(sa), "ok_labels": ok_labels, "ok_map": ok_map})
        for r in pkg.find_places(queries):
            f = r["flags"]
            got.append((list(r["rotations"]), np.ascontiguousarray(r["clouds"]).tobytes(),
                        np.ascontiguousarray(r["anno"][:, :3]).tobytes(), np.ascontiguousarray(r["anno"][:, 3:]).tobytes(),
                        int(((f & 1) == 0).sum()), int((((f & 3) == 3) & ((f & 16) == 0)).sum())))
        print(f"{len(got)} queries on the GPU, {time.time() - t0:.0f} s", flush=True)
    oracle = want.get()
    pool.close()
    bad, placements = 0, 0
    for spec, g, o in zip(specs, got, oracle):
        placements += len(o[0])
        if g != o:
            bad += 1
            what = [n for n, a, b in zip(("rotations", "clouds", "centres", "quaternions", "off the surface", "collisions"), g, o) if a != b]
            print("MISMATCH", spec, what, "rotations", g[0][:8], o[0][:8])
    print(f"{n_q} queries, {placements} possible placements: {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
