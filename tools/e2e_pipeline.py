#!/usr/bin/env python3
"""End-to-end rate of the file-to-file driver (row f-2) on synthetic frames written to local disk:
read .bin/.label -> upload -> batched HIP path -> download -> write velodyne/labels/check.
usage: python tools/e2e_pipeline.py [n_frames] [batch]"""
import importlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("pcl-augmentation_amd")
synth = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kinds = synth.CONFIG_INSERTS["C2"]
root = tempfile.mkdtemp(prefix="r3d_e2e_")
try:
    os.makedirs(f"{root}/in/velodyne"), os.makedirs(f"{root}/in/labels")
    frames, cands = [], {}
    for i in range(n):
        xyzi, label = synth.make_scene(i)
        xyzi.tofile(f"{root}/in/velodyne/{i:06d}.bin")
        label.tofile(f"{root}/in/labels/{i:06d}.label")
        frames.append(pkg.Frame(f"{root}/in/velodyne/{i:06d}.bin", f"{root}/in/labels/{i:06d}.label"))
        cands[i] = ([[x] for x in synth.make_inserts(i, kinds)], [20] * len(kinds))
    pipe = pkg.AugmentPipeline(f"{root}/out", "run", batch_size=bs)
    pipe.run(frames[:bs], lambda i: cands[i])              # warm-up: allocations, kernel load
    shutil.rmtree(f"{root}/out")
    t0 = time.perf_counter()
    st = pipe.run(frames, lambda i: cands[i])
    dt = time.perf_counter() - t0
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
    print(f"end to end: {n / dt:.1f} frames/s ({n} frames of 120k points, 5 inserts, batch {bs}), "
          f"read {st['t_read']:.2f}s process {st['t_process']:.2f}s write {st['t_write']:.2f}s (overlapped)")
finally:
    shutil.rmtree(root, ignore_errors=True)
