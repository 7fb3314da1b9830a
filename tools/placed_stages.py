"""Where a PlacedInserter slot's time goes (config C2's placed leg): the device calls of PlacedInserter.insert_slot one at a time,
the stream drained around each, the rest of the slot as one figure: python tools/placed_stages.py [B] [slab|rows]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
synth, fs = pkg.synth, pkg.Real3DAug.tools.find_spot
config = {"insertion": {"placement": synth.PLACEMENT, "placement_labels": synth.PLACEMENT_LABELS}}
kinds = synth.CONFIG_INSERTS["C2"]
frames = [synth.make_place_frame(s) for s in range(B)]
slots = []
for k in range(5):
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
info = [[f[k] for f in frames] for k in ("rich", "move", "pose", "boxes")]
placed = importlib.import_module("pcl-augmentation_amd.placed")
places = importlib.import_module("pcl-augmentation_amd.places")
T = {}


def timed(obj, name, label):
    """obj.name runs between two drains of the stream; its time goes to T[label]."""
    f = getattr(obj, name)

    def g(*a, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = f(*a, **kw)
        torch.cuda.synchronize()
        T[label] = T.get(label, 0.0) + time.perf_counter() - t0
        return out

    setattr(obj, name, g)


mode = sys.argv[2] if len(sys.argv) > 2 else "slab"
timed(batch, "export_alive", "export_alive")
timed(batch, "export_rows", "export_rows")
timed(places, "chunk_ranges", "chunk_ranges")
placed.chunk_ranges = places.chunk_ranges
timed(places.PlaceBatch, "run", "search")
timed(batch, "insert_first_device", "candidates (insert_first)")
timed(placed.PlacedInserter, "_fill_descriptors", "descriptors")
for rep in range(3):
    batch.load([(f["xyzi"], f["label"]) for f in frames])
    batch.begin()
    ins = pkg.PlacedInserter(batch, *info, scene_slab=mode == "slab")
    torch.cuda.synchronize()
    T.clear()
    t0 = time.perf_counter()
    for smp, annos, okl, okm in slots:
        ins.insert_slot(smp, annos, okl, okm, [20] * B)
    torch.cuda.synchronize()
    whole = time.perf_counter() - t0
    T["staging, upload, gather, read-back, host"] = whole - sum(T.values())
    print(mode, {k: round(1e3 * v / len(slots), 3) for k, v in T.items()}, "ms per slot; sum", round(1e3 * whole / len(slots), 2), flush=True)
