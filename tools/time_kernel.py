"""HIP-event times of the streaming kernels of step 0 / the compaction on config C2's batch, each alone
(r3d_batch_launch_one): python tools/time_kernel.py"""
import ctypes as C
import importlib
import sys

import torch

sys.path.insert(0, '.')
pkg = importlib.import_module("pcl-augmentation_amd")
L, synth = pkg._lib, pkg.synth
B = 256
bt = pkg.SceneBatch(B, 123500, 3500)
bt.load([synth.make_scene(s) for s in range(B)])
bt.begin()
torch.cuda.synchronize()
for which, name in ((L.K_BOUNDS, "k_bounds"), (L.K_PROJECT, "k_project"), (L.K_ALIVE_WRITE, "k_alive_write")):
    launch = lambda: L.check(bt.lib.r3d_batch_launch_one(C.byref(bt.desc), which, L.stream_ptr()), "launch_one")
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        launch()
    z.record()
    torch.cuda.synchronize()
    print(name, f"{a.elapsed_time(z) / 20:.4f} ms")
