import importlib, sys, numpy as np, torch
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
names = ["project", "window", "re-key + sort", "occ+rank+sdepth", "scene tile", "D bits", "closing", "cands", "evaluate", "count", "commit", "cleanup"]
for rep in range(2):
    batch.begin()
    sums = np.zeros(B)
    maxes = []
    for k, (s5, off) in enumerate(packed):
        batch.insert_device(s5, off, need)
        torch.cuda.synchronize()
        st = batch.out_xyzi.view(B, -1)[:, :32].contiguous().view(torch.int64).cpu().numpy()   # [B,16]
        d = np.diff(st[:, :13], axis=1) / 100.0   # us (100 MHz)
        tot = (st[:, 12] - st[:, 0]) / 100.0
        sums += tot
        maxes.append(tot.max())
        if rep == 1:
            worst = int(np.argmax(tot))
            print(f"insert {k} ({KINDS[k]}): mean total {tot.mean():.1f} us, max {tot.max():.1f} us (scene {worst})")
            print("   mean per phase:", " ".join(f"{n}={v:.1f}" for n, v in zip(names, d.mean(0))))
            print("   worst scene   :", " ".join(f"{n}={v:.1f}" for n, v in zip(names, d[worst])),
                  f"ww={st[worst,13]>>32} ncand={st[worst,13]&0xffffffff} nlist={st[worst,14]>>32} nvalid={st[worst,14]&0xffffffff}")

print(f"sum over inserts of (max over scenes) = {sum(maxes):.1f} us; max over scenes of (sum over inserts) = {sums.max():.1f} us; mean scene sum = {sums.mean():.1f} us")
