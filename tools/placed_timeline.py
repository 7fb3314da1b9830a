"""Per-batch wall-clock intervals of the placed leg's lanes (tools/bench_placed.py): which batch of which lane took how long,
stage by stage -- to see where a slow run of the leg loses its time: python tools/placed_timeline.py [lanes] [reps]"""
import importlib
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")
import torch

lanes_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if os.environ.get("SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["SWITCH_INTERVAL"]))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, K = 256, 5
synth, fs = pkg.synth, pkg.Real3DAug.tools.find_spot
config = {"insertion": {"placement": synth.PLACEMENT, "placement_labels": synth.PLACEMENT_LABELS}}
kinds = synth.CONFIG_INSERTS["C2"]
frames = [synth.make_place_frame(s) for s in range(B)]
slots = []
for k in range(K):
    smp, annos, okl, okm = [], [], [], []
    for s in range(B):
        pts, line = synth.make_place_sample(s * 100 + k, kinds[k % len(kinds)])
        sa = fs.read_label_line(line)
        m, l = fs.placement_surfaces(sa, config)
        smp.append(pts)
        annos.append(fs._anno10(sa))
        okl.append(l)
        okm.append(m)
    slots.append((smp, annos, okl, okm))
grow = sum(max(len(x) for x in sl[0]) for sl in slots)
n = max(len(f["xyzi"]) for f in frames)
scenes = [(f["xyzi"], f["label"]) for f in frames]
info = [[f[k] for f in frames] for k in ("rich", "move", "pose", "boxes")]
log, lock = [], threading.Lock()


def one_batch(batch, lane, it):
    t = [time.perf_counter()]
    batch.load(scenes)
    t.append(time.perf_counter())
    batch.begin()
    ins = pkg.PlacedInserter(batch, *info)
    t.append(time.perf_counter())
    for smp, annos, okl, okm in slots:
        ins.insert_slot(smp, annos, okl, okm, [20] * B)
        t.append(time.perf_counter())
    torch.cuda.current_stream().synchronize()
    batch.download_delta_views(5)
    t.append(time.perf_counter())
    with lock:
        log.append((lane, it, t))


batches = [pkg.SceneBatch(B, n + grow + 64, grow + 64) for _ in range(lanes_n)]
streams = [torch.cuda.Stream() for _ in range(lanes_n)]
start = threading.Barrier(lanes_n + 1)


def lane(i):
    with torch.cuda.stream(streams[i]):
        one_batch(batches[i], i, -1)
        start.wait()
        for it in range(reps):
            one_batch(batches[i], i, it)


for round_ in range(3):
    log.clear()
    start.reset()
    threads = [threading.Thread(target=lane, args=(i,)) for i in range(lanes_n)]
    for th in threads:
        th.start()
    start.wait()
    t0 = time.perf_counter()
    for th in threads:
        th.join()
    wall = time.perf_counter() - t0
    print(f"round {round_}: {lanes_n} lanes x {reps} batches, {lanes_n * reps * B / wall:.0f} frames/s", flush=True)
    timed = sorted((x for x in log if x[1] >= 0), key=lambda x: x[2][0])
    total = sorted(x[2][-1] - x[2][0] for x in timed)
    print("  batch ms: median %.1f, max %.1f" % (1e3 * total[len(total) // 2], 1e3 * total[-1]))
    import statistics
    stages = [[1e3 * (b - a) for a, b in zip(t, t[1:])] for _, _, t in timed]
    med = [statistics.median(col) for col in zip(*stages)]
    print("  median ms per stage: load %.1f  begin+setup %.1f  slots %s  download %.1f" % (med[0], med[1], " ".join("%.1f" % v for v in med[2:-1]), med[-1]))
    for ln, it, t in timed:
        d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
        if sum(d) > 2.5 * 1e3 * total[len(total) // 2]:
            print(f"  slow: lane {ln} batch {it} at {1e3 * (t[0] - t0):.0f} ms:", "load %.1f  begin+setup %.1f  slots %s  download %.1f"
                  % (d[0], d[1], " ".join("%.1f" % v for v in d[2:-1]), d[-1]), flush=True)
