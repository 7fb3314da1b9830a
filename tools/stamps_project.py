"""When does every workgroup of k_project start its first tile and end?  (a library built with tools/build_flavour.sh pstamp -DR3D_EXP_STAMP, chosen with R3D_LIB)"""
import ctypes as C, importlib, sys
import numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("pcl-augmentation_amd")
L, synth = pkg._lib, pkg.synth
B = 256
bt = pkg.SceneBatch(B, 123500, 3500)
bt.load([synth.make_scene(s) for s in range(B)])
bt.begin()
torch.cuda.synchronize()
for rep in range(3):
    L.check(bt.lib.r3d_batch_launch_one(C.byref(bt.desc), L.K_PROJECT, L.stream_ptr()), "launch_one")
    torch.cuda.synchronize()
raw = bt.out_xyzi.view(torch.int64).reshape(-1)[:4 * 4096].cpu().numpy().reshape(-1, 4)
raw = raw[raw[:, 2] > 0]
t0 = raw[:, 0].min()
st, first, en = (raw[:, 0] - t0) / 100.0, (raw[:, 1] - t0) / 100.0, (raw[:, 2] - t0) / 100.0   # 100 MHz -> us
print("workgroups", len(raw))
for name, v in (("start", st), ("first tile", first), ("end", en), ("start->first", first - st), ("life", en - st)):
    print(f"{name:14s} min {v.min():7.1f} p10 {np.percentile(v,10):7.1f} p50 {np.percentile(v,50):7.1f} p90 {np.percentile(v,90):7.1f} max {v.max():7.1f} us")
xcc = (raw[:, 3] >> 32) & 0xF
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"xcc {x}: {m.sum():4d} workgroups, start p50 {np.percentile(st[m],50):6.1f}, end p50 {np.percentile(en[m],50):6.1f} max {en[m].max():6.1f}")
hw = raw[:, 3] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 20 + cu
u, c = np.unique(key, return_counts=True)
print("workgroups per CU: ", np.bincount(c))
# do the four workgroups of a CU end together (a slow CU) or apart (slow tiles)?
spread_in_cu = np.array([en[key == k].max() - en[key == k].min() for k in u])
mean_of_cu = np.array([en[key == k].mean() for k in u])
print(f"end spread inside a CU: p50 {np.percentile(spread_in_cu,50):.1f} p90 {np.percentile(spread_in_cu,90):.1f} us; CU means: p10 {np.percentile(mean_of_cu,10):.1f} "
      f"p50 {np.percentile(mean_of_cu,50):.1f} p90 {np.percentile(mean_of_cu,90):.1f} max {mean_of_cu.max():.1f} us")
order = np.argsort(raw[:, 0] * 0 + np.arange(len(raw)))          # workgroup index = position in the tile sequence
blk = np.arange(len(raw))
for lo in range(0, len(raw), len(raw) // 8):
    m = (blk >= lo) & (blk < lo + len(raw) // 8)
    print(f"workgroups {lo:4d}..{lo + len(raw) // 8 - 1:4d} (consecutive tiles): life mean {(en - st)[m].mean():6.1f} us")
