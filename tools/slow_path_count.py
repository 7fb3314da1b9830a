import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("pcl-augmentation_amd")
synth = pkg.synth
B = 16
scenes = [synth.make_scene(s) for s in range(B)]
batch = pkg.SceneBatch(B, 120000 + 64, 64)
batch.load(scenes)
batch.begin()
torch.cuda.synchronize()
print("slow-path points per scene:", batch.n_out.cpu().numpy())
