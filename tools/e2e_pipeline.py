#!/usr/bin/env python3
"""End-to-end rates of the file-to-file driver.

    python tools/e2e_pipeline.py [n_frames] [batch]          # both figures
    bench.py --e2e N                                          # the in-memory figure as the bench line's `e2e` object

``measure``: frames resident in HOST memory (as read from .bin / .label) -> native packer into pinned
staging -> upload, begin / insert_many / finish, download over three lanes (StreamedAugmenter) ->
merged clouds, labels and check rows in pinned host memory.  PCIe-inclusive, disk-exclusive.
``measure_disk``: the same through AugmentPipeline.run_streamed with files on local disk on both sides.
"""
import importlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(pkg, n_frames=1024, B=256, lanes=3, pack_threads=16, delta=True, device="cuda:0", before_timed=None):
    synth = pkg.synth
    kinds = synth.CONFIG_INSERTS["C2"]
    n_distinct = min(n_frames, B)                     # B distinct frames, cycled: the generator is not what is measured
    scenes = [synth.make_scene(s) for s in range(n_distinct)]
    inserts = [synth.make_inserts(s, kinds) for s in range(n_distinct)]
    need = [[20] * len(kinds)] * B
    n_max = max(len(x) for x, _ in scenes)
    grow = max(sum(len(i) for i in ins) for ins in inserts)
    srows = max(sum(len(ins[k]) for ins in inserts) for k in range(len(kinds))) * (B // n_distinct + 1)
    from importlib import import_module
    streaming = import_module("pcl-augmentation_amd.streaming")
    aug = streaming.StreamedAugmenter(B, n_max, grow, len(kinds), srows, lanes=lanes, pack_threads=pack_threads, delta=delta,
                                      device=device)
    batch = [scenes[s % n_distinct] for s in range(B)], [inserts[s % n_distinct] for s in range(B)]
    n_batches = max(2, n_frames // B)
    got = {"frames": 0, "points": 0}

    def consume(tag, results, accepted):
        got["frames"] += len(results)
        got["points"] += sum(len(r[0]) for r in results)

    aug.run([(batch[0], batch[1], need, 0)] * 2, consume)           # warm-up: allocations, kernel load, clocks
    got["frames"] = got["points"] = 0
    aug.bytes_h2d = aug.bytes_d2h = 0
    if before_timed is not None:
        before_timed()                                                 # (a barrier: the ranks of one host start together)
    t0 = time.perf_counter()
    aug.run([(batch[0], batch[1], need, i) for i in range(n_batches)], consume)
    dt = time.perf_counter() - t0
    return {"frames_per_s": round(got["frames"] / dt, 1), "frames": got["frames"], "seconds": dt, "batch": B, "lanes": lanes,
            "h2d_GBps": round(aug.bytes_h2d / dt / 1e9, 2), "d2h_GBps": round(aug.bytes_d2h / dt / 1e9, 2),
            "pack_threads": pack_threads, "delta": delta,
            "what": "config C2 frames (120k points, 5 inserts) resident in host memory -> native packer -> pinned staging -> "
                    "upload / begin / insert_many / " + ("delta export / download (alive bits + inserted points) -> host merge"
                                                         if delta else "finish / download of the whole clouds") +
                    " on three lanes -> merged cloud, labels, check rows in host memory; disk excluded"}


def measure_disk(pkg, n_frames=512, B=64, io_threads=16, where=None):
    """where: directory for the files (default: the system's temporary directory, i.e. the box's local disk;
    /dev/shm shows what the software does when the file system is memory)."""
    synth = pkg.synth
    kinds = synth.CONFIG_INSERTS["C2"]
    root = tempfile.mkdtemp(prefix="r3d_e2e_", dir=where)
    try:
        os.makedirs(f"{root}/in/velodyne"), os.makedirs(f"{root}/in/labels")
        frames, ins = [], {}
        for i in range(n_frames):
            xyzi, label = synth.make_scene(i % 64)
            xyzi.tofile(f"{root}/in/velodyne/{i:06d}.bin")
            label.tofile(f"{root}/in/labels/{i:06d}.label")
            frames.append(pkg.Frame(f"{root}/in/velodyne/{i:06d}.bin", f"{root}/in/labels/{i:06d}.label"))
            ins[i] = (synth.make_inserts(i % 64, kinds), [20] * len(kinds))
        pipe = pkg.AugmentPipeline(f"{root}/out", "run", batch_size=B)
        pipe.run_streamed(frames[:B], lambda i: ins[i], io_threads=io_threads, pack_threads=io_threads)   # warm-up
        shutil.rmtree(f"{root}/out")
        st = pipe.run_streamed(frames, lambda i: ins[i], io_threads=io_threads, pack_threads=io_threads)
        return {"frames_per_s": round(st["frames_per_s"], 1), "frames": st["written"], "batch": B,
                "io_threads": io_threads, "directory": where or tempfile.gettempdir(),
                "what": "the same frames as .bin / .label files, read, processed and written back (velodyne, labels, check)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def measure_run(pkg, n_frames=512, B=64, where=None):
    """``AugmentPipeline.run`` (the candidate loop: here two placement candidates per insert) with files on both sides."""
    synth = pkg.synth
    kinds = synth.CONFIG_INSERTS["C2"]
    root = tempfile.mkdtemp(prefix="r3d_run_", dir=where)
    try:
        os.makedirs(f"{root}/in/velodyne"), os.makedirs(f"{root}/in/labels")
        frames, ins = [], {}
        for i in range(n_frames):
            xyzi, label = synth.make_scene(i % 64)
            xyzi.tofile(f"{root}/in/velodyne/{i:06d}.bin")
            label.tofile(f"{root}/in/labels/{i:06d}.label")
            frames.append(pkg.Frame(f"{root}/in/velodyne/{i:06d}.bin", f"{root}/in/labels/{i:06d}.label"))
            a, b = synth.make_inserts(i % 64, kinds), synth.make_inserts(i % 64 + 64, kinds)
            ins[i] = ([[x, y] for x, y in zip(a, b)], [20] * len(kinds))
        pipe = pkg.AugmentPipeline(f"{root}/out", "run", batch_size=B)
        pipe.run(frames[:B], lambda i: ins[i])                                                           # warm-up
        shutil.rmtree(f"{root}/out")
        st = pipe.run(frames, lambda i: ins[i])
        return {"frames_per_s": round(st["frames_per_s"], 1), "frames": st["written"], "batch": B,
                "t_read": round(st["t_read"], 3), "t_process": round(st["t_process"], 3), "t_write": round(st["t_write"], 3),
                "t_total": round(st["t_total"], 3), "directory": where or tempfile.gettempdir()}
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    pkg = importlib.import_module("pcl-augmentation_amd")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    print("in memory:", measure(pkg, n, bs))
    print("in memory, whole clouds downloaded:", measure(pkg, n, bs, delta=False))
    print("disk     :", measure_disk(pkg, min(n, 512), min(bs, 64)))
    if os.path.isdir("/dev/shm"):
        print("tmpfs    :", measure_disk(pkg, min(n, 2048), 256, where="/dev/shm"))
    print("run, disk:", measure_run(pkg, min(n, 512), min(bs, 64)))
    if os.path.isdir("/dev/shm"):
        print("run, tmpfs:", measure_run(pkg, min(n, 512), min(bs, 64), where="/dev/shm"))
