"""cProfile of the host side of PlacedInserter.insert_slot on config C2's placed batch (256 frames, five slots):
python tools/prof_placed_slot.py"""
import cProfile
import importlib
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")
import torch
B = 256
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
        smp.append(pts)
        annos.append(fs._anno10(sa))
        okl.append(l)
        okm.append(m)
    slots.append((smp, annos, okl, okm))
grow = sum(max(len(x) for x in sl[0]) for sl in slots)
n = max(len(f["xyzi"]) for f in frames)
batch = pkg.SceneBatch(B, n + grow + 64, grow + 64)
info = [[f[k] for f in frames] for k in ("rich", "move", "pose", "boxes")]
for rep in range(3):
    batch.load([(f["xyzi"], f["label"]) for f in frames])
    batch.begin()
    ins = pkg.PlacedInserter(batch, *info)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for smp, annos, okl, okm in slots:
        ins.insert_slot(smp, annos, okl, okm, [20] * B)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
