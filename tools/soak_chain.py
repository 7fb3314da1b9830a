#!/usr/bin/env python3
"""Soak test of r3d_batch_insert_many (the one-launch insert with XCD-local synchronisation): the
same batch again and again, alone and with a second batch in flight on another stream; every
iteration must give the same survivors, bytes and status.

    python tools/soak_chain.py [iterations] [C2|C3|C4]          # C3 / C4: ten / eight inserts per frame

SOAK_LANES (default 2) batches in flight; SOAK_DEPTH (default 1) rounds enqueued on every lane between two device-wide
synchronisations (the results of the last round are the ones compared).
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    synth = pkg.synth
    which = sys.argv[2] if len(sys.argv) > 2 else "C2"
    shape = {}
    if which == "C5":                                          # 64 scans of config C5's size (8 distinct), 20 of its 50 slots
        B, kinds, shape = 64, (["car", "pedestrian", "cyclist", "pedestrian", "cyclist"] * 4), dict(rows=448, cols=2880)
        distinct = [synth.make_scene(s, n_beams=256, n_az=3906) for s in range(8)]
        scenes = [distinct[s % 8] for s in range(B)]
    else:
        B, kinds = 256, synth.CONFIG_INSERTS[which]
        scenes = [synth.make_scene(s) for s in range(B)]
    inserts = [synth.make_inserts(s, kinds) for s in range(B)]
    grow = sum(max(len(inserts[s][k]) for s in range(B)) for k in range(len(kinds)))
    lanes = []
    n_lanes, depth = int(os.environ.get("SOAK_LANES", "2")), int(os.environ.get("SOAK_DEPTH", "1"))
    for _ in range(n_lanes):
        bt = pkg.SceneBatch(B, max(len(x) for x, _ in scenes) + grow, grow, **shape)
        bt.load(scenes)
        pk = [bt.pack_samples([inserts[s][k] for s in range(B)]) for k in range(len(kinds))]
        nd = torch.full((B,), 20, dtype=torch.int32, device=bt.device)
        lanes.append((bt, pk, nd, torch.cuda.Stream()))

    def step(lane):
        bt, pk, nd, st = lanes[lane]
        with torch.cuda.stream(st):
            bt.begin()
            _, acc = bt.insert_many_device(pk, [nd] * len(pk))
            bt.finish(check_cols=0)
        return acc

    def fingerprint(lane, acc):                      # on the default stream, after a device-wide synchronize
        bt = lanes[lane][0]
        per_frame = torch.stack((bt.n_out.to(torch.int64), bt.out_xyzi.view(torch.int32).sum(dim=(1, 2), dtype=torch.int64),
                                 bt.out_label.sum(dim=1, dtype=torch.int64), acc.sum(dim=0, dtype=torch.int64)))
        return per_frame.cpu().numpy().tolist() + [int(bt.status.sum(dtype=torch.int64).item())]

    torch.cuda.synchronize()                          # the uploads ran on the default stream
    acc = step(0)
    torch.cuda.synchronize()
    ref = fingerprint(0, acc)
    bad = 0
    import time
    t_last = time.time()
    for it in range(n_iter):
        if time.time() - t_last > 45:                          # (a sign of life for whoever watches the run)
            print(f"... iteration {it} of {n_iter}, {bad} with a mismatch so far", flush=True)
            t_last = time.time()
        for _ in range(depth):
            accs = [step(lane) for lane in range(n_lanes)]    # the batches in flight
        torch.cuda.synchronize()
        fps = [fingerprint(lane, accs[lane]) for lane in range(n_lanes)]
        if os.environ.get("SOAK_WATCH"):                   # a diagnostic build's failed checks, as soon as they show
            for lane in range(n_lanes):
                cnt = lanes[lane][0].debug_counters(reset=False)
                if "check_failures" in cnt or cnt["rebases_in_chain"]:
                    print("iteration", it, "lane", lane, {k: v for k, v in cnt.items() if k.startswith(("check", "rebase"))}, flush=True)
                    lanes[lane][0].debug_counters(reset=True)
        if any(f != ref for f in fps):
            bad += 1
            print("iteration", it, "differs in frames", [[s for s in range(B) if f[0][s] != ref[0][s] or f[1][s] != ref[1][s]] for f in fps],
                  "status sums", [f[-1] for f in fps], flush=True)
            for lane, f in enumerate(fps):                     # what exactly differs, per frame
                bt = lanes[lane][0]
                for s in range(B):
                    if any(f[j][s] != ref[j][s] for j in range(4)):
                        a = accs[lane][:, s].cpu().numpy().tolist()
                        print(f"  lane {lane} frame {s}: n_out {f[0][s]} (want {ref[0][s]}), xyzi sum {f[1][s]} ({ref[1][s]}), labels {f[2][s]} ({ref[2][s]}), "
                              f"accepted {a} (sum want {ref[3][s]}), status {int(bt.status[s])}, rebases {int(bt.rebase[s])}, n_total {int(bt.n_total[s])}, "
                              f"n_log {int(bt.n_log[s])}, bounds {bt.bounds[s].cpu().numpy().tolist()}", flush=True)
                print(f"  lane {lane} counters {bt.debug_counters(reset=False)}", flush=True)
    print(f"{n_iter} iterations x {depth} rounds x {n_lanes} lanes x {B} frames x {len(kinds)} slots: {bad} iterations with a mismatch; status {ref[-1]}")
    # R3D_DEBUG_BITS=64: every speculative evaluation that was about to be committed was done again after its
    # predecessors and compared (csrc/r3d_insert.hip, kDbgVerify)
    for lane, (bt, _, _, _) in enumerate(lanes):
        cnt = bt.debug_counters()
        print(f"lane {lane}: {cnt}")
        if cnt["verify_mismatch"]:
            bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
