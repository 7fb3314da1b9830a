"""Where the time before a placed batch's first insert slot goes (tools/bench_placed.py's `load_begin_setup`): SceneBatch.load,
begin, PlacedInserter.__init__ piece by piece, the stream drained after each: python tools/placed_setup_stages.py [B]"""
import cProfile
import importlib
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
synth = pkg.synth
frames = [synth.make_place_frame(s) for s in range(B)]
n = max(len(f["xyzi"]) for f in frames)
batch = pkg.SceneBatch(B, n + 3000 + 64, 3000 + 64)
scenes = [(f["xyzi"], f["label"]) for f in frames]
info = [[f[k] for f in frames] for k in ("rich", "move", "pose", "boxes")]
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch.load(scenes)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    batch.begin()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if rep == 3:
        pr = cProfile.Profile()
        pr.enable()
    ins = pkg.PlacedInserter(batch, *info)
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    if rep == 3:
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(12)
    print({"load (host)": round(1e3 * (t1 - t0), 2), "load (drain)": round(1e3 * (t2 - t1), 2), "begin": round(1e3 * (t3 - t2), 2),
           "inserter (host)": round(1e3 * (t4 - t3), 2), "inserter (drain)": round(1e3 * (t5 - t4), 2)}, flush=True)
