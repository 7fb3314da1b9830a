#!/usr/bin/env python3
"""One diagnostic soak with the bounds-checked build of the library (hipcc -DR3D_CHECK, csrc/r3d_insert.hip: CHK):

    R3D_LIB=pcl-augmentation_amd/libreal3daug_hip_check.so python tools/check_soak.py [iterations] [C2|C3]

256 full-size frames, two batches in flight; prints the diagnostic counters -- `check_failures` lists, per check code,
how many derived indices lay outside their array (the access was skipped instead of made)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")
synth = pkg.synth
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kinds = synth.CONFIG_INSERTS[sys.argv[2] if len(sys.argv) > 2 else "C3"]
B = 256
scenes = [synth.make_scene(s) for s in range(B)]
inserts = [synth.make_inserts(s, kinds) for s in range(B)]
grow = sum(max(len(inserts[s][k]) for s in range(B)) for k in range(len(kinds)))
lanes = []
for _ in range(2):
    bt = pkg.SceneBatch(B, 120000 + grow, grow)
    bt.load(scenes)
    pk = [bt.pack_samples([inserts[s][k] for s in range(B)]) for k in range(len(kinds))]
    nd = torch.full((B,), 20, dtype=torch.int32, device=bt.device)
    lanes.append((bt, pk, nd, torch.cuda.Stream()))
torch.cuda.synchronize()
ref = None
bad = 0
for it in range(n_iter):
    for bt, pk, nd, st in (lanes if it % 2 else lanes[:1]):          # alone, then two in flight, alternating
        with torch.cuda.stream(st):
            bt.begin()
            bt.insert_many_device(pk, [nd] * len(pk))
            bt.finish(check_cols=5)
    torch.cuda.synchronize()
    for bt, _, _, _ in (lanes if it % 2 else lanes[:1]):
        fp = (bt.n_out.sum().item(), bt.out_xyzi.view(torch.int32).sum(dtype=torch.int64).item(), int(bt.status.sum().item()))
        if ref is None:
            ref = fp
        bad += fp != ref
print(f"{n_iter} iterations: {bad} fingerprints differ from the first; first = {ref}")
for lane, (bt, _, _, _) in enumerate(lanes):
    print("lane", lane, bt.debug_counters())
