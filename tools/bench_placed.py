#!/usr/bin/env python3
"""End-to-end timing of placement search + occlusion merge for a batch of frames
(PlacedInserter: export rows -> r3d_find_possible_places -> r3d_batch_insert per candidate).

    python tools/bench_placed.py [B] [slots]
"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")
import torch  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    synth, fs = pkg.synth, pkg.Real3DAug.tools.find_spot
    config = {"insertion": {"placement": synth.PLACEMENT, "placement_labels": synth.PLACEMENT_LABELS}}
    kinds = synth.CONFIG_INSERTS["C2"]
    frames = [synth.make_place_frame(s) for s in range(B)]
    slots = []
    for k in range(K):
        smp, annos, okl, okm = [], [], [], []
        for s in range(B):
            pts, line = synth.make_place_sample(s * 100 + k, kinds[k % len(kinds)])
            sa = fs.read_label_line(line)
            m, l = fs.placement_surfaces(sa, config)
            smp.append(pts); annos.append(fs._anno10(sa)); okl.append(l); okm.append(m)
        slots.append((smp, annos, okl, okm))
    grow = sum(max(len(x) for x in sl[0]) for sl in slots)
    n = max(len(f["xyzi"]) for f in frames)
    batch = pkg.SceneBatch(B, n + grow + 64, grow + 64)
    scenes = [(f["xyzi"], f["label"]) for f in frames]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch.load(scenes)
        batch.begin()
        ins = pkg.PlacedInserter(batch, [f["rich"] for f in frames], [f["move"] for f in frames], [f["pose"] for f in frames],
                                 [f["boxes"] for f in frames])
        t1 = time.perf_counter()
        placed = 0
        for smp, annos, okl, okm in slots:
            rot, _ = ins.insert_slot(smp, annos, okl, okm, [20] * B)
            placed += sum(1 for r in rot if r > 0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        batch.finish()
        res = batch.results()
        t3 = time.perf_counter()
    if os.environ.get("R3D_PROFILE_SLOT"):                     # where the host time of one slot goes
        import cProfile, pstats
        batch.load(scenes)
        batch.begin()
        ins = pkg.PlacedInserter(batch, [f["rich"] for f in frames], [f["move"] for f in frames], [f["pose"] for f in frames],
                                 [f["boxes"] for f in frames])
        pr = cProfile.Profile()
        pr.enable()
        for smp, annos, okl, okm in slots:
            ins.insert_slot(smp, annos, okl, okm, [20] * B)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    print(f"B={B} slots={K}: load+begin+setup {1e3*(t1-t0):.1f} ms, {K} placed slots {1e3*(t2-t1):.1f} ms "
          f"({1e3*(t2-t1)/K:.1f} ms per slot), finish+download {1e3*(t3-t2):.1f} ms; {placed} objects placed; "
          f"{B/(t3-t0):.0f} frames/s end to end")


if __name__ == "__main__":
    main()
