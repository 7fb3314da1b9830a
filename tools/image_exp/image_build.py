"""Round 6, premise of the image route: what does a persistent raw range image per scene cost at step 0?
HIP-event times of k_image_clear / k_image_build (r3d_batch_launch_one) beside k_bounds / k_project / k_alive_write on
config C2 (ring order and shuffled) and C5:  python tools/image_exp/image_build.py [C2|C2s|C5] [scenes]"""
import ctypes as C
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
pkg = importlib.import_module("pcl-augmentation_amd")
L, synth = pkg._lib, pkg.synth
which = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which == "C5":
    distinct = 16
    scenes = [synth.make_scene(s, 256, 3906) for s in range(distinct)]
    scenes = [scenes[s % distinct] for s in range(B)]
    bt = pkg.SceneBatch(B, 1000000 + 75000, 75000, rows=448, cols=2880)
else:
    scenes = [synth.make_scene(s, shuffle=(which == "C2s")) for s in range(B)]
    bt = pkg.SceneBatch(B, 123500, 3500)
bt.load(scenes)
bt.begin()
torch.cuda.synchronize()
n_pts = sum(len(x) for x, _ in scenes)
K_IMAGE_CLEAR, K_IMAGE_BUILD, K_IMAGE_BANDS = 6, 7, 8


def image_fields(bt):
    """img / occ of the batch's workspace as tensors (the layout of carve_batch, counted from the end)."""
    al = lambda v: (v + 255) // 256 * 256
    npix = bt.rows * bt.cols
    sizes = [("img", bt.B * npix * 8), ("kstep", bt.B * npix * 2), ("occ", bt.B * (npix // 32) * 4), ("img_valid", bt.B * 4),
             ("img_dirty", bt.B * 4), ("hold_pix", bt.B * 16 * 4), ("n_hold", bt.B * 4), ("band_fall", bt.B * bt.rows), ("n_fall", bt.B * 4),
             ("dbg", 64 * 4)]
    end = bt.ws.numel()
    off = {}
    for name, nbytes in reversed(sizes):
        end -= al(nbytes)
        off[name] = (end, nbytes)
    f = lambda name, dt: bt.ws[off[name][0]:off[name][0] + off[name][1]].view(dt)
    return f("img", torch.int64), f("occ", torch.int32), f("n_hold", torch.int32), f("n_fall", torch.int32), f("kstep", torch.int16)


img, occ, n_hold, n_fall, kstep = image_fields(bt)
one = lambda k: L.check(bt.lib.r3d_batch_launch_one(C.byref(bt.desc), k, L.stream_ptr()), "launch_one")
one(K_IMAGE_CLEAR)
one(K_IMAGE_BUILD)
torch.cuda.synchronize()
ref_img, ref_occ, ref_hold = img.clone(), occ.clone(), n_hold.clone()
img.fill_(12345)
occ.fill_(777)
kstep.fill_(3)
one(K_IMAGE_BANDS)
torch.cuda.synchronize()
print("bands == atomics:", bool((img == ref_img).all()), bool((occ == ref_occ).all()), bool((n_hold == ref_hold).all()),
      "kstep zero:", bool((kstep == 0).all()), "| occupied pixels per scene", int((ref_img != -1).sum()) // B,
      "| bands left to atomics:", int(n_fall.sum()), "| holders/scene", float(ref_hold.float().mean()), flush=True)
for which_k, name in ((L.K_BOUNDS, "k_bounds"), (L.K_PROJECT, "k_project"), (K_IMAGE_CLEAR, "k_image_clear"),
                      (K_IMAGE_BUILD, "k_image_build"), (K_IMAGE_BANDS, "k_image_bands(+prepare, build)"), (L.K_ALIVE_WRITE, "k_alive_write")):
    launch = lambda: L.check(bt.lib.r3d_batch_launch_one(C.byref(bt.desc), which_k, L.stream_ptr()), "launch_one")
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        launch()
    z.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(z) / 10
    print(f"{which} B={B} {name}: {ms:.4f} ms  ({n_pts / ms / 1e6:.1f} G points/s)", flush=True)
