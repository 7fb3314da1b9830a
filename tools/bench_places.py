#!/usr/bin/env python3
"""Timing of the placement search on the C2-shaped workload: B synthetic 120k-point scenes, K
samples tried per scene (one query each), all queries in one call.

    python tools/bench_places.py [B] [K] [cap] [boxes per scene]

Prints ms per call (HIP events around r3d_find_possible_places, inputs resident), queries/s and
the equivalent rate of reference steps (360 per query)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")
import torch  # noqa: E402

PLACEMENT = {18: [1, 3], 30: [2], 31: [1, 3]}
PLACEMENT_LABELS = {1: [40, 60], 2: [48], 3: [44]}
CONFIG = {"insertion": {"placement": PLACEMENT, "placement_labels": PLACEMENT_LABELS}}


def make_scene(seed, n_boxes=6):
    synth = pkg.synth
    xyzi, label = synth.make_scene(seed)
    label = label.copy()
    ground = label == 40
    label[ground & (xyzi[:, 1] > 4.0)] = 48
    label[ground & (xyzi[:, 0] < -8.0) & (xyzi[:, 1] <= 4.0)] = 44
    original = synth.scene5_from_packed(xyzi, label)
    T = np.eye(4)
    T[:3, 3] = [500.5 + seed, -200.25, 1.7]
    half = 70
    move = np.array([[int(np.floor(T[0, 3])) - half], [int(np.floor(T[1, 3])) - half], [0], [1]])
    rich = np.zeros((2 * half + 1, 2 * half + 1), dtype=np.uint8)
    world = (T @ np.hstack((original[:, :3], np.ones((len(original), 1)))).T - move).astype(int)
    inside = (world[0] >= 0) & (world[0] < rich.shape[0]) & (world[1] >= 0) & (world[1] < rich.shape[1])
    for value, labels in ((1, (40,)), (2, (48,)), (3, (44,))):
        sel = inside & np.isin(original[:, 4], labels)
        rich[world[0][sel], world[1][sel]] = value
    scene9 = np.full((len(original), 9), -1.0)
    scene9[:, :3], scene9[:, 6], scene9[:, 7] = original[:, :3], original[:, 3], original[:, 4]
    rng = np.random.default_rng(seed)
    boxes = []
    for ang in rng.uniform(-np.pi, np.pi, size=n_boxes):
        d = rng.uniform(6, 25)
        boxes.append([d * np.cos(ang), d * np.sin(ang), -1.73, 0, 0, np.sin(ang / 2), np.cos(ang / 2), 4.2, 1.8, 1.5])
    return pkg.PlaceScene(scene9, original, boxes, rich, move, T)


def make_query(scene, seed, kind):
    synth, fs = pkg.synth, pkg.Real3DAug.tools.find_spot
    cls = {"pedestrian": 30, "cyclist": 31, "car": 18}[kind]
    smp = synth.make_insert(seed, kind)
    length, width, height, _, _ = synth.INSERT_KINDS[kind]
    centre = [smp[:, 0].mean(), smp[:, 1].mean(), smp[:, 2].min()]
    line = " ".join([str(cls)] + [repr(float(v)) for v in (*centre, height, length, width, 0.3)])
    sa = fs.read_label_line(line)
    ok_map, ok_labels = fs.placement_surfaces(sa, CONFIG)
    return {"scene": scene, "sample": smp, "anno": fs._anno10(sa), "ok_labels": ok_labels, "ok_map": ok_map}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cap = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    n_boxes = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    kinds = pkg.synth.CONFIG_INSERTS["C2"]
    t0 = time.time()
    scenes = [make_scene(s, n_boxes) for s in range(B)]
    queries = [make_query(scenes[s], s * 100 + k, kinds[k % len(kinds)]) for s in range(B) for k in range(K)]
    batch = pkg.places.PlaceBatch(queries, cand_cap=cap)
    print(f"setup {time.time() - t0:.1f} s, {len(queries)} queries, workspace {batch.ws_bytes / 2**20:.0f} MiB", flush=True)
    batch.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 5
    e0.record()
    for _ in range(steps):
        batch.run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    res = batch.results()
    npos = np.array([len(r["rotations"]) for r in res])
    print(f"{ms:.3f} ms per call, {len(queries) / ms * 1e3:.0f} queries/s, {len(queries) * 360 / ms * 1e3:.3e} rotation steps/s, "
          f"possible placements per query: mean {npos.mean():.1f} min {npos.min()} max {npos.max()}")


if __name__ == "__main__":
    main()
