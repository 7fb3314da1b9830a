"""Where k_commit_chain spends its time on config C2: per-slot ticks by path (tools/trace_chain.py [scenes])."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("pcl-augmentation_amd")
import bench
cfg = bench.CONFIGS[os.environ.get("CFG", "C2")]
B = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["scenes"]
synth = pkg.synth
scenes = [bench.build_scene(synth, cfg, s) for s in range(B)]
inserts = [synth.make_inserts(s, cfg["kinds"]) for s in range(B)]
K = len(cfg["kinds"])
grow = sum(max(len(inserts[s][k]) for s in range(B)) for k in range(K))
bt = pkg.SceneBatch(B, max(len(x) for x, _ in scenes) + grow, grow, rows=cfg["rows"], cols=cfg["cols"])
bt.load(scenes)
pk = [bt.pack_samples([inserts[s][k] for s in range(B)]) for k in range(K)]
nd = torch.full((B,), 20, dtype=torch.int32, device=bt.device)
for _ in range(3):
    bt.begin(); bt.insert_many_device(pk, [nd] * K); bt.finish()
torch.cuda.synchronize()
bt.debug_counters()
K = min(K, 64)
bt.begin(); bt.insert_many_device(pk[:K], [nd] * K); torch.cuda.synchronize()
print(bt.debug_counters())
ticks, path, start = bt.debug_trace()
ticks, path, start = ticks[:, :K] / 100.0, path[:, :K], (start[:, :K] - start[:, :K].min()) / 100.0
names = {0: "none", 4: "stored", 5: "rejected", 6: "conflict", 7: "uneval", 10: "nofit"}
for pid in sorted(set(path.ravel())):
    t = ticks[path == pid]
    print(f"path {names.get(int(pid), pid)}: n {t.size}  mean {t.mean():.1f} us  p50 {np.median(t):.1f}  p95 {np.percentile(t,95):.1f}  max {t.max():.1f}")
per = ticks.sum(axis=1)
print("per scene total us: mean %.1f p95 %.1f max %.1f; first start %.1f last end %.1f" % (per.mean(), np.percentile(per, 95), per.max(), start.min(), (start + ticks).max()))
w = int(per.argmax())
print("worst scene", w, [f"{names.get(int(p), p)}:{t:.0f}" for p, t in zip(path[w], ticks[w])])
