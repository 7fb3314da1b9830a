"""Per-phase wall-clock breakdown of the insert kernel on config C2 (256 scenes x 5 slots = 1 280
(scene, slot) pairs), from a diagnostic build of the library (`make STAMPS=1`, see csrc/Makefile):

    R3D_LIB=pcl-augmentation_amd/libreal3daug_hip_stamps.so python tools/stamps_insert.py [out.npz]

Every workgroup leaves a 100 MHz wall-clock stamp per phase in its scene's out_xyzi slab; one launch
per slot (r3d_batch_insert) so that the stamps of a slot can be read before the next one overwrites
them.  Prints a table (mean / p50 / max per phase over all pairs) and saves the raw stamps."""
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
pkg = importlib.import_module("pcl-augmentation_amd")
synth = pkg.synth
B = 256
KINDS = ["pedestrian", "cyclist", "car", "pedestrian", "cyclist"]
scenes = [synth.make_scene(s) for s in range(B)]
inserts = [synth.make_inserts(s, KINDS) for s in range(B)]
grow = sum(max(len(inserts[s][k]) for s in range(B)) for k in range(5))
batch = pkg.SceneBatch(B, 120000 + grow, grow)
batch.load(scenes)
need = torch.full((B,), 20, dtype=torch.int32, device=batch.device)
packed = [batch.pack_samples([inserts[s][k] for s in range(B)]) for k in range(5)]
names = ["project", "window", "re-key + sort", "occ+rank+sdepth", "scene tile", "D bits", "closing", "cands",
         "evaluate", "count", "commit", "cleanup"]
NS = len(names) + 1
raw = np.zeros((5, B, 16), dtype=np.int64)
for rep in range(2):
    batch.begin()
    for k, (s5, off) in enumerate(packed):
        batch.insert_device(s5, off, need)
        torch.cuda.synchronize()
        raw[k] = batch.out_xyzi.view(B, -1)[:, :32].contiguous().view(torch.int64).cpu().numpy()

d = np.diff(raw[:, :, :NS], axis=2) / 100.0          # us
tot = (raw[:, :, NS - 1] - raw[:, :, 0]) / 100.0
ww = raw[:, :, 13] >> 32
ncand = raw[:, :, 13] & 0xFFFFFFFF
nlist = raw[:, :, 14] >> 32
nvalid = raw[:, :, 14] & 0xFFFFFFFF
print(f"| phase | mean us | p50 | p95 | max | share of mean total |")
print("|---|---|---|---|---|---|")
flat = d.reshape(-1, len(names))
for i, n in enumerate(names):
    c = flat[:, i]
    print(f"| {n} | {c.mean():.2f} | {np.percentile(c, 50):.2f} | {np.percentile(c, 95):.2f} | {c.max():.2f} | {100 * c.mean() / tot.mean():.1f} % |")
print(f"| total | {tot.mean():.2f} | {np.percentile(tot, 50):.2f} | {np.percentile(tot, 95):.2f} | {tot.max():.2f} | |")
for k in range(5):
    print(f"slot {k} ({KINDS[k]}): total mean {tot[k].mean():.1f} max {tot[k].max():.1f} us; window words mean {ww[k].mean():.0f} "
          f"max {ww[k].max()}; chunks listed mean {nlist[k].mean():.0f} max {nlist[k].max()}; candidates mean {ncand[k].mean():.0f}; "
          f"valid sample points mean {nvalid[k].mean():.0f}")
print(f"sum over slots of (max over scenes) = {tot.max(1).sum():.1f} us; max over scenes of (sum over slots) = "
      f"{tot.sum(0).max():.1f} us; mean scene sum = {tot.sum(0).mean():.1f} us")
if len(sys.argv) > 1:
    np.savez_compressed(sys.argv[1], raw=raw, names=np.array(names))
