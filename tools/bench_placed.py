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


def measure(pkg, B=256, K=5, reps=3, profile=False, lanes=2):
    """B frames, K insert slots each with the placement search in the loop (PlacedInserter); host staging and the
    download of the results included.  `lanes` batches are in flight at a time, each on its own thread, HIP stream and
    device batch (as AugmentPipeline.run_placed(lanes=...) runs them): the upload and the host work of one batch overlap
    the kernels of the other.  Returns the rate over `reps` batches per lane after one warm-up batch each, and the
    stage times of one batch run alone."""
    import threading
    import torch
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
    scenes = [(f["xyzi"], f["label"]) for f in frames]
    info = [[f[k] for f in frames] for k in ("rich", "move", "pose", "boxes")]

    def one_batch(batch, times=None):
        t0 = time.perf_counter()
        batch.load(scenes)
        batch.begin()
        ins = pkg.PlacedInserter(batch, *info)
        t1 = time.perf_counter()
        placed = 0
        for smp, annos, okl, okm in slots:
            rot, _ = ins.insert_slot(smp, annos, okl, okm, [20] * B)
            placed += sum(1 for r in rot if r > 0)
        torch.cuda.current_stream().synchronize()
        t2 = time.perf_counter()
        if os.environ.get("R3D_PLACED_WHOLE_CLOUDS"):
            batch.finish()
            batch.download_views()                               # merged clouds, labels, check rows in pinned host memory
        else:
            batch.download_delta_views(5)                        # the delta back, merged clouds / labels / check rows put together on the host
        t3 = time.perf_counter()
        if times is not None:
            times.update(load_begin_setup=round(1e3 * (t1 - t0), 1), placed_slots=round(1e3 * (t2 - t1), 1),
                         per_slot=round(1e3 * (t2 - t1) / K, 2), finish_download=round(1e3 * (t3 - t2), 1))
        return placed

    batches = [pkg.SceneBatch(B, n + grow + 64, grow + 64) for _ in range(lanes)]
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    alone, placed = {}, [0]
    for _ in range(2):                                             # warm-up, then the stage times of a batch run alone
        placed[0] = one_batch(batches[0], alone)
    start, errors = threading.Barrier(lanes + 1), []

    def lane(i):
        try:
            with torch.cuda.stream(streams[i]):
                if i:
                    one_batch(batches[i])                          # this lane's own warm-up (allocations, pinned staging)
                start.wait()
                for _ in range(reps):
                    one_batch(batches[i])
        except Exception as e:
            errors.append(e)
            start.abort()

    threads = [threading.Thread(target=lane, args=(i,)) for i in range(lanes)]
    for t in threads:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    if errors:
        raise errors[0]
    if profile:                                                # where the host time of one slot goes
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        one_batch(batches[0])
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    return {"frames_per_s": round(lanes * reps * B / wall, 1), "frames": lanes * reps * B, "batch": B, "slots": K, "lanes": lanes,
            "objects_placed_per_batch": placed[0], "objects_tried_per_batch": B * K, "ms_one_batch_alone": alone,
            "frames_per_s_one_batch_at_a_time": round(B / (sum(v for k, v in alone.items() if k != "per_slot") * 1e-3), 1),
            "what": "the per-frame body of the reference's driver (insertion.py:380-545) for batches of frames: frames from host "
                    "memory -> upload, begin; per insert slot the placement search on the current clouds (find_possible_places: "
                    "360 rotations, map, height, collisions) and its candidates tried in rotation order until one is accepted; "
                    "finish, merged clouds + check rows back in pinned host memory; `lanes` batches in flight"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    for lanes in (1, 2, 3):
        r = measure(pkg, B, K, profile=bool(os.environ.get("R3D_PROFILE_SLOT")) and lanes == 1, lanes=lanes)
        print(f"B={B} slots={K} lanes={lanes}: alone {r['ms_one_batch_alone']}; {r['objects_placed_per_batch']} objects placed; "
              f"{r['frames_per_s']:.0f} frames/s end to end")


if __name__ == "__main__":
    main()
