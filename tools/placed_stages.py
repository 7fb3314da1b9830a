"""Where a PlacedInserter slot's time goes (config C2's placed leg): the steps of PlacedInserter._insert_slot one at a time,
the stream drained after each: python tools/placed_stages.py [B]"""
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


def lap(name, t0):
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


for rep in range(3):
    T.clear()
    batch.load([(f["xyzi"], f["label"]) for f in frames])
    batch.begin()
    ins = pkg.PlacedInserter(batch, *info)
    torch.cuda.synchronize()
    for smp, annos, okl, okm in slots:
        t = time.perf_counter()
        rows, n_rows = batch.export_rows()
        n_rows_h = n_rows.cpu().numpy()
        t = lap("export_rows", t)
        who = list(range(B))
        max_b = max(1, int(ins.n_boxes.max()))
        boxes_d = torch.from_numpy(np.ascontiguousarray(ins.boxes_h[:, :max_b])).to(batch.device)
        t = lap("boxes", t)
        all_ranges = places.chunk_ranges(rows.view(B * batch.cap, 4)).view(B, batch.cap // 64, 2)
        t = lap("chunk_ranges", t)
        smp_rows, smp_off = batch.pack_samples(smp)
        smp_off_h = smp_off.cpu().numpy()
        t = lap("pack_samples", t)
        pb = ins._pack_slot(who, rows, n_rows_h, boxes_d, max_b, all_ranges, smp_rows, smp_off_h, annos, okl, okm, None, 8)
        t = lap("pack_slot (descriptors)", t)
        pb.run(first_cand=0)
        t = lap("search", t)
        orig_run = pb.run
        pb.run = lambda first_cand=0: pb                       # (the search has run: _try_candidates' first window)
        ins._try_candidates(pb, who, [20] * B, 8, annos, [-1] * B, [0] * B)
        pb.run = orig_run
        t = lap("candidates + read-back", t)
    print({k: round(1e3 * v / len(slots), 3) for k, v in T.items()}, "ms per slot; sum", round(1e3 * sum(T.values()) / len(slots), 2), flush=True)
