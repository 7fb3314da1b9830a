"""Per-phase wall-clock breakdown of the insert kernel on config C2 (256 scenes x 5 slots = 1 280
(scene, slot) pairs), from a diagnostic build of the library (`make STAMPS=1`, see csrc/Makefile):

    R3D_LIB=pcl-augmentation_amd/libreal3daug_hip_stamps.so python tools/stamps_insert.py [out.npz]

Every workgroup of k_insert_chain leaves 100 MHz wall-clock stamps at its phase boundaries in its
scene's out_xyzi slab (32 words per slot).  Prints a table (mean / p50 / p95 / max per phase over all
pairs) and saves the raw stamps."""
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
pkg = importlib.import_module("pcl-augmentation_amd")
synth = pkg.synth
import os
if os.environ.get("R3D_STAMPS_CONFIG") == "C5":            # the first launch (32 slots) of config C5's 50 inserts
    B = int(os.environ.get("R3D_STAMPS_SCANS", "32"))        # scans; distinct ones: 8, cycled
    KINDS = (["car", "pedestrian", "cyclist", "pedestrian", "cyclist"] * 10)[:int(os.environ.get("R3D_STAMPS_SLOTS", "32"))]
    distinct = [synth.make_scene(s, n_beams=256, n_az=3906) for s in range(min(B, 8))]
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
need = torch.full((B,), 20, dtype=torch.int32, device=batch.device)
packed = [batch.pack_samples([inserts[s][k] for s in range(B)]) for k in range(K)]
# ("hand-over": round 5 -- nobody waits; what is left between the end of the evaluation and the commit is the look at the
# progress word, the conflict test and, for a pair that parks, the record it leaves)
names = ["project sample", "window + re-key", "occupancy + rank", "counting sort + depths", "sample closing + count",
         "scene set-up", "chunk list", "candidates + gather (first band)", "scene bits + closing", "evaluate (+ further bands)",
         "visible list + kill masks", "hand-over", "commit", "publish"]
edges = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 11), (11, 13), (13, 14), (14, 15)]
for rep in range(3):
    batch.begin()
    batch.out_xyzi.view(B, -1)[:, :K * 64].zero_()
    batch.insert_many_device(packed, [need] * K)
    torch.cuda.synchronize()
    raw = batch.out_xyzi.view(B, -1)[:, :K * 64].contiguous().view(torch.int64).cpu().numpy().reshape(B, K, 32)
raw = raw.transpose(1, 0, 2)                       # [K, B, 16]
info = raw[:, :, 12]
attempts, all_lds, ww, nlist = info & 0xFF, (info >> 8) & 1, (info >> 16) & 0xFFFF, (info >> 32) & 0xFFFF
# a pair that did not have to wait (slot 0, or its predecessor was done) has no stamp 13: use stamp 11
st = raw.astype(np.float64)
st[:, :, 13] = np.where(raw[:, :, 13] == 0, raw[:, :, 11], raw[:, :, 13])
# a redone evaluation overwrites stamps 0..11 with the second attempt's; totals use the kernel's span
tot = (st[:, :, 15] - st[:, :, 0]) / 100.0
print("| phase | mean us | p50 | p95 | max | share of mean total |")
print("|---|---|---|---|---|---|")
once = attempts == 1
for n, (a, z) in zip(names, edges):
    c = ((st[:, :, z] - st[:, :, a]) / 100.0)[once]
    print(f"| {n} | {c.mean():.2f} | {np.percentile(c, 50):.2f} | {np.percentile(c, 95):.2f} | {c.max():.2f} | {100 * c.mean() / tot[once].mean():.1f} % |")
print(f"| total (pairs evaluated once) | {tot[once].mean():.2f} | {np.percentile(tot[once], 50):.2f} | {np.percentile(tot[once], 95):.2f} | {tot[once].max():.2f} | |")
print("per slot, mean us of the phases from the chunk list on (chain kernel; pairs evaluated once):")
for k in range(K):
    sel = once[k]
    if sel.any():
        print(f"  slot {k} ({KINDS[k]}): " + ", ".join(
            f"{n.split(' ')[0]} {((st[k, :, z] - st[k, :, a]) / 100.0)[sel].mean():.1f}" for n, (a, z) in list(zip(names, edges))[6:]))
t0 = raw[:, :, 0].min()
print(f"pairs: {once.sum()} evaluated once, {(attempts == 2).sum()} twice (a predecessor changed their pixels), "
      f"{(attempts == 0).sum()} not at all; {(all_lds == 0).sum()} left to k_insert_big")
print(f"launch span: first start -> last publish = {(raw[:, :, 15].max() - t0) / 100.0:.1f} us; "
      f"slot starts (mean, us after launch): " + ", ".join(f"{(raw[k, :, 0].mean() - t0) / 100.0:.1f}" for k in range(K)))
for k in range(K):
    print(f"slot {k} ({KINDS[k]}): total mean {tot[k].mean():.1f} max {tot[k].max():.1f} us; window words mean {ww[k].mean():.0f} "
          f"max {ww[k].max()}; chunks listed mean {nlist[k].mean():.0f} max {nlist[k].max()}; end (mean, us after launch) "
          f"{(raw[k, :, 15].mean() - t0) / 100.0:.1f}")
g = raw[:, :, 17:21].astype(np.float64) / 100.0                  # thread 0 inside the gather loop (loads drained at every mark)
print("gather loop, thread 0, us per pair (mean / max): " + "; ".join(
    f"{n} {g[:, :, i].mean():.2f} / {g[:, :, i].max():.2f}" for i, n in enumerate(["list + pixel ids", "place in tile", "coordinates", "sqrt + min"])))
c = raw.astype(np.float64)
if (raw[:, :, 22] > 0).any():
    # (pairs committed by the workgroup that evaluated them: a parked pair's commit stamps are its taker's, its stamp 14 is not set)
    m = (raw[:, :, 22] > 0) & (raw[:, :, 14] > raw[:, :, 25]) & (raw[:, :, 22] >= st[:, :, 13].astype(np.int64))
    print("commit (accepted pairs), us: header %.2f; append %.2f; kills %.2f; far list + barrier %.2f; tail %.2f" % (
        ((c[:, :, 22] - st[:, :, 13]) / 100)[m].mean(), ((c[:, :, 23] - c[:, :, 22]) / 100)[m].mean(),
        ((c[:, :, 24] - c[:, :, 23]) / 100)[m].mean(), ((c[:, :, 25] - c[:, :, 24]) / 100)[m].mean(),
        ((c[:, :, 14] - c[:, :, 25]) / 100)[m].mean()))
if len(sys.argv) > 1:
    np.savez_compressed(sys.argv[1], raw=raw, names=np.array(names))
# inside the two longest phases, by size of the depth tile (pooled tiles: the windows that exceed the LDS)
if (raw[:, :, 26] > 0).any():
    npx = (info >> 48) << 4
    cls = [(0, 4000, "tile < 4k px"), (4000, 9000, "4k - 9k"), (9000, 10**9, ">= 9k px")] if shape == {} else \
          [(0, 15000, "tile < 15k px"), (15000, 40000, "15k - 40k"), (40000, 10**9, ">= 40k px")]
    for lo, hi, lab in cls:
        sel = once & (npx >= lo) & (npx < hi) & (raw[:, :, 26] > 0)
        if not sel.any():
            continue
        f = lambda a, z, sl=None: ((st[:, :, z] - st[:, :, a]) / 100.0)[sel if sl is None else sl].mean()
        acc = sel & (raw[:, :, 31] > 0)                            # (stamp 31, "visible list made", exists for accepted pairs only)
        vis = f"visible list {f(10, 31, acc):.1f}, kill masks {f(31, 11, acc):.1f} ({acc.sum()} accepted)" if acc.any() else "no accepted pair"
        print(f"{lab}: {sel.sum()} pairs, total {tot[sel].mean():.1f} us ({100 * tot[sel].sum() / tot[once].sum():.0f} % of all pair time) | "
              f"list candidates + clear {f(7, 26):.1f}, gather {f(26, 27):.1f}, roots + bits {f(27, 8):.1f}, evaluate {f(9, 10):.1f}, {vis}")
# the workgroup's own span (kernel entry -> exit) against the phases' span, and how long a CU stays empty between two workgroups
if raw.shape[2] > 30 and raw[:, :, 28].any():
    ent, ext, retry = raw[:, :, 28].astype(np.float64), raw[:, :, 30].astype(np.float64), raw[:, :, 29]
    t00 = ent[ent > 0].min()
    print(f"workgroup entry -> first stamp: mean {((st[:, :, 0] - ent) / 100).mean():.1f} us; last stamp -> exit: {((ext - st[:, :, 15]) / 100).mean():.1f} us; "
          f"pairs that the LDS flavour turned down (second pass with pooled images): {(retry > 0).sum()}")
    starts, ends = np.sort(((ent - t00) / 100).ravel()), np.sort(((ext - t00) / 100).ravel())
    for cap in (192, 224, 256):
        lag = starts[cap:] - ends[:len(starts) - cap]
        print(f"  entry of workgroup i minus exit of workgroup i-{cap}: median {np.median(lag):.1f} us, negative for {100 * (lag < 0).mean():.0f} %")
    span = (ext.max() - t00) / 100
    print(f"  workgroup-time {((ext - ent) / 100).sum() / 1000:.1f} ms over a span of {span / 1000:.2f} ms = {((ext - ent) / 100).sum() / span:.0f} workgroups on the device on average")
# the chain that ends last: per slot when it started, how long each phase took, when it published
last = np.unravel_index(np.argmax(raw[:, :, 15]), raw[:, :, 15].shape)[1]
print(f"the scene whose chain ends last (scene {last}):")
for k in range(K):
    r = st[k, last]
    # (a stamp that was never set -- a pair committed from its parked record has no evaluation stamps of its own -- prints as "-")
    ph = ", ".join(f"{n.split(' ')[0]} " + (f"{(r[z] - r[a]) / 100.0:.1f}" if raw[k, last, z] > 0 and raw[k, last, a] > 0 and r[z] >= r[a] else "-")
                   for n, (a, z) in zip(names, edges))
    print(f"  slot {k} ({KINDS[k]}): start {(r[0] - t0) / 100.0:.1f} end {(r[15] - t0) / 100.0:.1f} attempts {attempts[k, last]} words {ww[k, last]} chunks {nlist[k, last]} tile px {(info[k, last] >> 48) << 4} | {ph}")
ends = (raw[K - 1, :, 15] - t0) / 100.0
print("end of the scenes' chains, us after launch: p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(ends, [50, 90, 99, 100])))
