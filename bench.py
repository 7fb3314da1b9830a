#!/usr/bin/env python3
"""Headline benchmark: augmented scenes/s on BASELINE.json config C2 -- a batch of 256 synthetic
64-beam ~120k-point scenes, 5 inserts each -- with inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one pass of the hot path over one batch: r3d_batch_begin (elevation bounds,
spherical projection), the five insert slots (r3d_batch_insert_many: sample projection, closing /
hole fill on the candidate pixels, visibility mask, cull, append) and r3d_batch_finish (compaction
into the velodyne/.bin + labels/.label + check/.bin byte layout).  By default two steps are in
flight: consecutive steps alternate between two HBM-resident copies of the batch on two HIP
streams, as consecutive batches of a real run would, so that the streaming kernels of one step fill
the CUs that the latency-bound insert kernel of the other leaves idle (`--overlap 1`: one step at
a time; that rate is also reported, as config.scenes_per_s_one_step_in_flight).  With N > 1 the
driver starts one process per GPU (torchrun); every rank runs its own batch of 256 scenes (weak
scaling, scenes are independent, no collective on the data path), the timed region is bracketed
by a barrier + synchronize and the maximum over ranks is reported.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, measured with HIP events on
the launch stream after the timed region; `cpu_baseline` is the NumPy oracle (a port of the
reference's algorithm, single core) timed on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
B_SCENES = 256
KINDS = ["pedestrian", "cyclist", "car", "pedestrian", "cyclist"]      # config C2
MIN_POINTS = 20


def build_inputs(pkg, rank, B):
    synth = pkg.synth
    scenes = [synth.make_scene(1000 * rank + s) for s in range(B)]
    inserts = [synth.make_inserts(1000 * rank + s, KINDS) for s in range(B)]
    return scenes, inserts


def event_time_ms(torch, fn, reps=5):
    """Average duration of fn() in ms, HIP events on the current (launch) stream."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def cpu_baseline(pkg, budget_s=20.0, max_scenes=16):
    """The oracle's literal K-insert chain on scenes of the same workload, one core.  The synthetic
    generator is outside the timed region."""
    from oracle import real3d_oracle as O
    synth = pkg.synth
    done, spent = 0, 0.0
    while done < max_scenes:
        xyzi, label = synth.make_scene(5000 + done)
        ins = synth.make_inserts(5000 + done, KINDS)
        s5 = synth.scene5_from_packed(xyzi, label)
        t1 = time.perf_counter()
        O.augment_scene(s5, [[x] for x in ins], [MIN_POINTS] * len(ins))
        per = time.perf_counter() - t1
        spent += per
        done += 1
        if spent + per > budget_s:
            break
    return done, spent


def placement_leg(pkg, torch, n_frames, with_cpu):
    """r3d_find_possible_places on n_frames synthetic 120k-point frames x the 5 samples of config C2
    (one query each, all in one call), inputs resident; the CPU figure is the oracle's
    find_possible_places on the first queries."""
    synth, fs = pkg.synth, pkg.Real3DAug.tools.find_spot
    config = {"insertion": {"placement": synth.PLACEMENT, "placement_labels": synth.PLACEMENT_LABELS}}
    frames = [synth.make_place_frame(s) for s in range(n_frames)]
    queries, raw = [], []
    for s, f in enumerate(frames):
        scene9 = np.full((len(f["original"]), 9), -1.0)
        scene9[:, :3], scene9[:, 6], scene9[:, 7] = f["original"][:, :3], f["original"][:, 3], f["original"][:, 4]
        ps = pkg.PlaceScene(scene9, f["original"], f["boxes"], f["rich"], f["move"], f["pose"])
        for k, kind in enumerate(KINDS):
            smp, line = synth.make_place_sample(s * 100 + k, kind)
            sa = fs.read_label_line(line)
            ok_map, ok_labels = fs.placement_surfaces(sa, config)
            queries.append({"scene": ps, "sample": smp, "anno": fs._anno10(sa), "ok_labels": ok_labels, "ok_map": ok_map})
            if s == 0:
                raw.append((scene9, f, smp, line))
    pb = pkg.places.PlaceBatch(queries, cand_cap=4)
    pb.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        pb.run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res = pb.results()
    out = {"queries": len(queries), "ms_per_call": round(ms, 3), "queries_per_s": round(len(queries) / ms * 1e3, 1),
           "rotation_steps_per_s": round(len(queries) * 360 / ms * 1e3, 1),
           "mean_possible_placements": round(float(np.mean([len(r["rotations"]) for r in res])), 1),
           "workload": f"{n_frames} synthetic 120k-point frames x 5 samples (2 pedestrians, 2 cyclists, 1 car), "
                       "6 annotated boxes per frame, 141 x 141 rich map; timing excludes descriptor packing"}
    if with_cpu:
        from oracle import find_spot_oracle as F
        t0, same = time.perf_counter(), True
        for qi, (scene9, f, smp, line) in enumerate(raw):
            annos = [F.make_annotation(b[:3], b[3:7], b[7], b[8], b[9]) for b in f["boxes"]]
            _, _, rot, _, _ = F.find_possible_places(scene9, annos, smp, line, f["rich"].astype(np.float64), f["move"],
                                                     f["original"], f["pose"], synth.PLACEMENT, synth.PLACEMENT_LABELS)
            same = same and rot == list(res[qi]["rotations"])
        secs = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(len(raw) / secs, 3), "unit": "queries/s", "cores": 1, "kind": "port",
                               "sample": f"the {len(raw)} queries of the first frame through oracle.find_possible_places",
                               "same_rotations_as_gpu": same}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scenes", type=int, default=B_SCENES, help="scenes per GPU batch (256 = config C2)")
    ap.add_argument("--streams", type=int, default=1,
                    help="the batch is processed as this many sub-batches on separate HIP streams, so that "
                         "one scene's long insert does not idle the other CUs")
    ap.add_argument("--overlap", type=int, default=2,
                    help="consecutive steps alternate between this many full-size batches, each on its own HIP stream, "
                         "so that the streaming kernels of one step can fill the CUs the insert kernel of the "
                         "previous step leaves idle (1 = every step on the same batch and stream; ignored with "
                         "--streams / --graph / --per-slot-launches)")
    ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph per step instead of launching "
                    "every kernel from Python (measured: same step time at one stream, slower with several)")
    ap.add_argument("--per-slot-launches", action="store_true",
                    help="one r3d_batch_insert call per insert slot instead of one r3d_batch_insert_many call for the five")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--placement", type=int, default=0, metavar="SCENES",
                    help="also time the placement search (SURVEY.md par.8 f-1) on SCENES frames x 5 samples and add "
                         "a `placement_search` object to the JSON line (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torchrun with WORLD_SIZE={args.gpus} (got {world})")
    # R3D_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a
    # one-GPU box); the driver's runs use RCCL ("nccl") with one rank per GPU.
    backend = os.environ.get("R3D_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("pcl-augmentation_amd")
    B = args.scenes
    scenes, inserts = build_inputs(pkg, rank, B)
    n_max = max(len(x) for x, _ in scenes)
    grow = sum(max(len(inserts[s][k]) for s in range(B)) for k in range(len(KINDS)))
    n_pts = float(sum(len(x) for x, _ in scenes))
    m_pts = float(sum(len(i) for ins in inserts for i in ins))

    def make_batch(lo, hi):
        bt = pkg.SceneBatch(hi - lo, n_max + grow, grow, device=f"cuda:{local_rank}")
        bt.load(scenes[lo:hi])                            # inputs resident in HBM from here on
        pk = [bt.pack_samples([inserts[s][k] for s in range(lo, hi)]) for k in range(len(KINDS))]
        nd = torch.full((hi - lo,), MIN_POINTS, dtype=torch.int32, device=bt.device)
        return bt, pk, nd

    # the whole batch as one descriptor (per-kernel timing) ...
    batch, packed, need = make_batch(0, B)
    # ... and as sub-batches on their own streams (the timed pipeline)
    n_sub = max(1, min(args.streams, B))
    cuts = [B * i // n_sub for i in range(n_sub + 1)]
    subs = [(batch, packed, need)] if n_sub == 1 else [make_batch(cuts[i], cuts[i + 1]) for i in range(n_sub)]
    streams = [torch.cuda.Stream() for _ in subs]

    def enqueue_step():
        main = torch.cuda.current_stream()
        for (bt, pk, nd), st in zip(subs, streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                bt.begin()
                if args.per_slot_launches:
                    accs = []
                    for s5, off in pk:
                        accs.append(bt.insert_device(s5, off, nd)[1].clone())
                    bt.last_acc = torch.stack(accs)
                else:
                    _, bt.last_acc = bt.insert_many_device(pk, [nd] * len(pk))
                bt.finish(check_cols=0)
            main.wait_stream(st)

    enqueue_step()                      # first call outside any capture (sets kernel attributes)
    torch.cuda.synchronize()
    if not args.per_slot_launches and int((batch.status & pkg._lib.S_CHAIN_TIMEOUT).sum().item()):
        # the one-launch insert could not order a scene's slots on this device: fall back, and say so
        args.per_slot_launches = True
        batch.status.zero_()
        enqueue_step()
        torch.cuda.synchronize()
    # --overlap D: D whole batches, step i runs on batch / stream i % D and is not joined until the end
    depth = 1 if (args.streams > 1 or args.graph or args.per_slot_launches) else max(1, args.overlap)
    lanes = [(batch, packed, need)] + [make_batch(0, B) for _ in range(depth - 1)] if depth > 1 else []
    lane_streams = [torch.cuda.Stream() for _ in lanes]
    lane_no = [0]

    def enqueue_overlapped():
        lane = lane_no[0] % depth
        lane_no[0] += 1
        bt, pk, nd = lanes[lane]
        with torch.cuda.stream(lane_streams[lane]):
            bt.begin()
            _, bt.last_acc = bt.insert_many_device(pk, [nd] * len(pk))
            bt.finish(check_cols=0)
    graph = None
    if args.graph:
        # the ~25 launches of one step are captured once into a hipGraph and replayed per step
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            enqueue_step()

    def one_step():
        if depth > 1:
            enqueue_overlapped()
        elif graph is not None:
            graph.replay()
        else:
            enqueue_step()

    # setup, not measurement: a freshly started GPU takes a few launches to reach its clocks and to
    # have every page of the batch touched; step until three consecutive steps agree within 10 %
    # (at most 100 steps / 2 s), then do the W warm-up steps and the K timed steps of the contract
    settle, recent, t_settle = 0, [], time.perf_counter()
    while settle < 100 and time.perf_counter() - t_settle < 2.0:
        ts = time.perf_counter()
        one_step()
        torch.cuda.synchronize()
        recent = (recent + [time.perf_counter() - ts])[-3:]
        settle += 1
        if len(recent) == 3 and max(recent) <= 1.1 * min(recent):
            break
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=batch.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for bt, _, _ in subs + lanes[1:]:
        bt.raise_on_status()
    if depth > 1:                                   # every lane worked on the same input: same output sizes
        ref_out = lanes[0][0].n_out.cpu().numpy()
        assert all(np.array_equal(bt.n_out.cpu().numpy(), ref_out) for bt, _, _ in lanes[1:])
        # the same K steps one at a time, for the record (not the headline value)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            enqueue_step()
        torch.cuda.synchronize()
        serial_elapsed = time.perf_counter() - t1
    else:
        serial_elapsed = elapsed
    n_out = np.concatenate([bt.n_out.cpu().numpy() for bt, _, _ in subs])
    n_accepted = int(sum(int(bt.last_acc.sum().item()) for bt, _, _ in subs))
    rebases = sum(int(bt.rebase.sum().item()) for bt, _, _ in subs)   # informational: a rebase leaves the inputs intact

    if rank == 0:
        import ctypes as C
        L = pkg._lib
        lib, desc = batch.lib, batch.desc

        def one(which):
            return lambda: L.check(lib.r3d_batch_launch_one(C.byref(desc), which, L.stream_ptr()), "launch_one")

        # per-kernel timing after the timed region, HIP events on the launch stream
        batch.begin()
        t_bounds = event_time_ms(torch, one(L.K_BOUNDS))
        t_reset = event_time_ms(torch, one(L.K_PREPARE))
        t_project = event_time_ms(torch, one(L.K_PROJECT))
        t_begin = event_time_ms(torch, batch.begin)

        def five_inserts():                         # begin() restores the state the inserts mutate
            batch.begin()
            if args.per_slot_launches:
                for s5, off in packed:
                    batch.insert_device(s5, off, need)
            else:
                batch.insert_many_device(packed, [need] * len(packed))

        # per-slot launches: 5 x (k_insert + idle k_rebase); otherwise one k_insert_chain launch
        t_insert_all = event_time_ms(torch, five_inserts) - t_begin
        batch.finish(check_cols=0)
        t_write = event_time_ms(torch, one(L.K_ALIVE_WRITE))
        t_finish = event_time_ms(torch, lambda: batch.finish(check_cols=0))
        n_out_pts = float(batch.n_out.sum().item())
        # Algorithmic bytes per launch (SURVEY.md par.8d with the conservative rule:
        # a pass that is not made is not claimed).  bounds / project read xyzi once (16 B per point);
        # the compaction reads xyzi + label and writes the survivors (20 B + 20 B per point); an
        # insert launch reads every sample row twice (2 x 40 B per sample point).
        kernels = {
            "k_bounds": {"ms": t_bounds, "launches_per_step": 1, "alg_bytes": 16.0 * n_pts},
            "k_project": {"ms": t_project, "launches_per_step": 1, "alg_bytes": 16.0 * n_pts},
            **({"k_insert": {"ms": t_insert_all / len(KINDS), "launches_per_step": len(KINDS),
                             "alg_bytes": 80.0 * m_pts / len(KINDS)}} if args.per_slot_launches else
               {"k_insert_chain": {"ms": t_insert_all, "launches_per_step": 1, "alg_bytes": 80.0 * m_pts}}),
            "k_alive_write": {"ms": t_write, "launches_per_step": 1, "alg_bytes": 20.0 * n_pts + 20.0 * n_out_pts},
            "k_prepare": {"ms": t_reset, "launches_per_step": 1, "alg_bytes": 0.0},
        }
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))["kernels"]
        except Exception:
            pass
        for name, k in kernels.items():
            k["GBps"] = k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9
            k["frac"] = k["GBps"] / HBM_PEAK_GBS
            k["share_of_step"] = k["ms"] * k["launches_per_step"] / (1e3 * serial_elapsed / args.steps)   # of a step run alone
        dominant = max(kernels, key=lambda k: kernels[k]["ms"] * kernels[k]["launches_per_step"])
        dk = kernels[dominant]
        roofline = {
            "bound": "hbm", "kernel": dominant, "achieved": round(dk["GBps"], 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(dk["frac"], 4),
            "traffic": pmc.get(dominant, {}).get("hbm_bytes_corrected"),
            "traffic_source": "profiles/r01_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                              "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch" if dominant in pmc else None,
            "alg_bytes_per_launch": dk["alg_bytes"], "ms_per_launch": round(dk["ms"], 4),
            "launches_per_step": dk["launches_per_step"],
            "note": "the insert kernel is one workgroup per (scene, slot) working in LDS on the window of the range image "
                    "around the inserted object; it is latency-bound by design and moves almost no HBM bytes (the "
                    "incremental pipeline removed the per-insert streaming passes), so its HBM fraction is low by "
                    "construction. The HBM-bound streaming kernels are listed in all_kernels."
                    if dominant.startswith("k_insert") else "",
            "all_kernels": {n: {"ms_per_launch": round(k["ms"], 4), "launches_per_step": k["launches_per_step"],
                                "alg_GBps": round(k["GBps"], 1), "frac_of_hbm_peak": round(k["frac"], 4),
                                "share_of_step": round(k["share_of_step"], 3),
                                "pmc_hbm_bytes_per_launch": pmc.get(n, {}).get("hbm_bytes_corrected")}
                            for n, k in kernels.items()},
            "api_calls_ms": {"r3d_batch_begin": round(t_begin, 4), ("r3d_batch_insert_x5" if args.per_slot_launches else "r3d_batch_insert_many_5"): round(t_insert_all, 4),
                             "r3d_batch_finish": round(t_finish, 4)},
        }
        scenes_per_s = B * world * args.steps / elapsed
        # whole-step algorithmic bytes actually needed by this pipeline
        step_bytes = 16.0 * n_pts * 2 + 20.0 * n_pts + 20.0 * n_out_pts + 80.0 * m_pts
        out = {
            "metric": "augmented scenes/sec (120k-pt, 64-beam)", "value": round(scenes_per_s, 1),
            "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: batch of 256 synthetic 64-beam 120k-pt scenes, 5 inserts each "
                                   "(2 pedestrians, 2 cyclists, 1 car), per GPU",
                       "scenes_per_gpu": B, "points_per_scene": int(n_pts / B), "inserts_per_scene": len(KINDS),
                       "range_image": [batch.rows, batch.cols], "inserts_accepted": n_accepted, "inserts_tried": B * len(KINDS),
                       "sub_batches_on_streams": n_sub, "hip_graph": graph is not None, "steps_in_flight": depth,
                       "scenes_per_s_one_step_in_flight": round(B * world * args.steps / serial_elapsed, 1),
                       "insert_api": "r3d_batch_insert x5" if args.per_slot_launches else "r3d_batch_insert_many(5)",
                       "rebases_in_timed_steps": rebases, "settle_steps_in_setup": settle,
                       "mean_points_out": float(n_out.mean())},
            "roofline": roofline,
            "pipeline_alg_GBps_per_gpu": round(step_bytes * args.steps / elapsed / 1e9, 1),
        }
        if not args.no_cpu_baseline and world == 1:     # rank 0 at N = 1 only: the other ranks would wait at the barrier
            done, secs = cpu_baseline(pkg)
            out["cpu_baseline"] = {"value": round(done / secs, 3), "unit": "scenes/s", "cores": 1, "kind": "port",
                                   "sample": f"{done} scenes of the same workload (120k points, 5 inserts) through "
                                             "oracle.augment_scene (NumPy port of the reference), one core"}
        if args.placement > 0 and world == 1:
            out["placement_search"] = placement_leg(pkg, torch, args.placement, not args.no_cpu_baseline)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
