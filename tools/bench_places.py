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

CONFIG = {"insertion": {"placement": pkg.synth.PLACEMENT, "placement_labels": pkg.synth.PLACEMENT_LABELS}}


def make_scene(seed, n_boxes=6):
    f = pkg.synth.make_place_frame(seed, n_boxes)
    scene9 = np.full((len(f["original"]), 9), -1.0)
    scene9[:, :3], scene9[:, 6], scene9[:, 7] = f["original"][:, :3], f["original"][:, 3], f["original"][:, 4]
    return pkg.PlaceScene(scene9, f["original"], f["boxes"], f["rich"], f["move"], f["pose"])


def make_query(scene, seed, kind):
    fs = pkg.Real3DAug.tools.find_spot
    smp, line = pkg.synth.make_place_sample(seed, kind)
    sa = fs.read_label_line(line)
    ok_map, ok_labels = fs.placement_surfaces(sa, CONFIG)
    return {"scene": scene, "sample": smp, "anno": fs._anno10(sa), "ok_labels": ok_labels, "ok_map": ok_map}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cap = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    n_boxes = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    kinds = os.environ.get("R3D_KINDS", "").split(",") if os.environ.get("R3D_KINDS") else pkg.synth.CONFIG_INSERTS["C2"]
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
