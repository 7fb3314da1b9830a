"""Timeline of one chain launch from the stamps build (tools/build_stamps.sh): when every (slot, scene) pair was taken
off the queue, when its evaluation ended, who committed it and when it was published.

    R3D_LIB=pcl-augmentation_amd/libreal3daug_hip_stamps.so python tools/timeline_insert.py [C2|C5] [scans] [slots]

Prints, per slot: take / publish times (mean, max), how many pairs were committed by another workgroup than their evaluator;
the distribution of the scenes' chain ends; the busy time of the workgroup slots."""
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
pkg = importlib.import_module("pcl-augmentation_amd")
synth = pkg.synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
if cfg == "C5":
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    KINDS = (["car", "pedestrian", "cyclist", "pedestrian", "cyclist"] * 10)[:int(sys.argv[3]) if len(sys.argv) > 3 else 50]
    distinct = [synth.make_scene(s, n_beams=256, n_az=3906) for s in range(min(B, 16))]
    scenes = [distinct[s % len(distinct)] for s in range(B)]
    shape = dict(rows=448, cols=2880)
else:
    B, KINDS = 256, ["pedestrian", "cyclist", "car", "pedestrian", "cyclist"]
    scenes = [synth.make_scene(s) for s in range(B)]
    shape = {}
K = len(KINDS)
inserts = [synth.make_inserts(s, KINDS) for s in range(B)]
grow = sum(max(len(inserts[s][k]) for s in range(B)) for k in range(K))
batch = pkg.SceneBatch(B, max(len(x) for x, _ in scenes) + grow, grow, **shape)
batch.load(scenes)
batch.count_pairs(True)
need = torch.full((B,), 20, dtype=torch.int32, device=batch.device)
packed = [batch.pack_samples([inserts[s][k] for s in range(B)]) for k in range(K)]
for rep in range(3):
    batch.begin()
    batch.out_xyzi.view(B, -1)[:, :K * 64].zero_()
    batch.insert_many_device(packed, [need] * K)
    torch.cuda.synchronize()
    raw = batch.out_xyzi.view(B, -1)[:, :K * 64].contiguous().view(torch.int64).cpu().numpy().reshape(B, K, 32)
print(batch.debug_counters())
t0 = raw[:, :, 21][raw[:, :, 21] > 0].min()
us = lambda a: (a - t0) / 100.0
take, last_take, done, pub = us(raw[:, :, 21]), us(raw[:, :, 28]), us(raw[:, :, 30]), us(raw[:, :, 15])
who, mode = raw[:, :, 16] & 0xFFFFFFFF, raw[:, :, 16] >> 32
ev_end = us(raw[:, :, 11])
print(f"{cfg}: {B} scenes x {K} slots; launch span {pub.max():.1f} us")
print("slot kind        take mean/max      eval-end mean    publish mean/max   taken over   mean busy")
for k in range(K):
    busy = (done[:, k] - last_take[:, k]).mean()
    print(f"{k:3d}  {KINDS[k]:10s} {take[:, k].mean():7.1f} {take[:, k].max():7.1f}   {ev_end[:, k].mean():8.1f}   {pub[:, k].mean():8.1f} {pub[:, k].max():8.1f}   {int((mode[:, k] != 0).sum()):5d}   {busy:7.1f}")
end = pub.max(axis=1)
print("chain ends: p50 %.1f p90 %.1f p99 %.1f max %.1f (scene %d)" % (np.percentile(end, 50), np.percentile(end, 90), np.percentile(end, 99), end.max(), int(end.argmax())))
sw = int(end.argmax())
print("slowest scene, per slot: take, eval end, last take, publish, mode")
for k in range(K):
    print(f"   {k:2d} {KINDS[k]:10s} {take[sw, k]:8.1f} {ev_end[sw, k]:8.1f} {last_take[sw, k]:8.1f} {pub[sw, k]:8.1f}  mode {int(mode[sw, k])}")
np.savez(f"gpurun_out/timeline_{cfg}.npz", raw=raw)
