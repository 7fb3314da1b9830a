#!/usr/bin/env python3
"""Headline benchmark: augmented scenes/s on BASELINE.json config C2 -- a batch of 256 synthetic
64-beam ~120k-point scenes, 5 inserts each -- with inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C5]

One "step" is one pass of the hot path over one batch: r3d_batch_begin (elevation bounds,
spherical projection), the insert slots (r3d_batch_insert_many: sample projection, closing / hole
fill on the candidate pixels, visibility mask, cull, append) and r3d_batch_finish (compaction into
the velodyne/.bin + labels/.label byte layout, plus the check/.bin rows).  By default three steps are
in flight: consecutive steps rotate over three HBM-resident copies of the batch on three HIP
streams, as consecutive batches of a real run do (streaming.py), so that the streaming kernels of one step fill
the CUs that the latency-bound insert kernel of the other leaves idle (`--overlap 1`: one step at a
time; that rate is also reported, as config.scenes_per_s_one_step_in_flight).

With N > 1 one process per GPU runs its own batch (weak scaling: scenes are independent, there is
no collective on the data path); the timed region is bracketed by a barrier + synchronize and the
maximum over ranks is reported.  Started under torchrun (RANK / WORLD_SIZE in the environment) the
process is one rank; started plainly with --gpus N it starts the N ranks itself (torchrun as a child
process, before anything here touches a GPU) and exits with the child's code.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, measured with HIP events on
the launch stream after the timed region; `cpu_baseline` is the NumPy oracle (a port of the
reference's algorithm) timed on a bounded sample of the same scenes -- one core, and one process
per core -- and its output bytes are compared with the GPU's for those scenes (`parity_checked`).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MIN_POINTS = 20
PROFILE_TAG = "r06"

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "C2": {"scenes": 256, "beams": 64, "az": 1875, "rows": 112, "cols": 1440,
           "kinds": ["pedestrian", "cyclist", "car", "pedestrian", "cyclist"],
           "workload": "C2: batch of 256 synthetic 64-beam 120k-pt scenes, 5 inserts each (2 pedestrians, 2 cyclists, "
                       "1 car), per GPU"},
    # BASELINE.json configs[2] / [3] as shapes of the hot path (10 resp. 8 inserts per 120k-point frame): not bench lines
    # of their own (the line is quoted on C2), selectable for measurements of longer chains on the reference's grid
    "C3": {"scenes": 256, "beams": 64, "az": 1875, "rows": 112, "cols": 1440,
           "kinds": ["car", "pedestrian", "cyclist", "car", "pedestrian", "cyclist", "car", "pedestrian", "cyclist", "pedestrian"],
           "workload": "C3 shape: batch of 256 synthetic 64-beam 120k-pt frames, 10 mixed car / pedestrian / cyclist inserts each, per GPU"},
    "C4": {"scenes": 256, "beams": 64, "az": 1875, "rows": 112, "cols": 1440,
           "kinds": ["pedestrian", "cyclist", "car", "pedestrian", "cyclist", "car", "pedestrian", "cyclist"],
           "workload": "C4 shape: batch of 256 synthetic 64-beam 120k-pt frames, 8 inserts each, per GPU"},
    # BASELINE.json configs[4]: the HBM-bound stress run (range image 448 x 2880 as SURVEY.md par.8d proposes)
    "C5": {"scenes": 32, "beams": 256, "az": 3906, "rows": 448, "cols": 2880,
           "kinds": (["car", "pedestrian", "cyclist", "pedestrian", "cyclist"] * 10),
           "workload": "C5: batch of 32 synthetic 256-beam 1M-pt scans, 50 inserts each, range image 448 x 2880, per GPU"},
}


def scene_seed(rank, s):
    return 1000 * rank + s


def build_scene(synth, cfg, seed):
    # cfg["shuffle"]: SURVEY.md par.8d / BASELINE.md par.4's second point order -- the same scan with its points in a random
    # order (the reference loops over the points in whatever order they come, insertion.py:100-127)
    return synth.make_scene(seed, n_beams=cfg["beams"], n_az=cfg["az"], shuffle=bool(cfg.get("shuffle")))


def oracle_scene(pkg, cfg, seed):
    """The oracle's literal K-insert chain on one scene of the workload: (seconds, velodyne bytes,
    label bytes, check bytes).  The synthetic generator is outside the timed region."""
    from oracle import real3d_oracle as O
    synth = pkg.synth
    if (O.NUMROW, O.NUMCOLUMN) != (cfg["rows"], cfg["cols"]):
        O.NUMROW, O.NUMCOLUMN = cfg["rows"], cfg["cols"]          # the reference's two globals (insertion.py:22-23)
    xyzi, label = build_scene(synth, cfg, seed)
    ins = synth.make_inserts(seed, cfg["kinds"])
    s5 = synth.scene5_from_packed(xyzi, label)
    t1 = time.perf_counter()
    merged, allvis, _ = O.augment_scene(s5, [[x] for x in ins], [MIN_POINTS] * len(ins))
    secs = time.perf_counter() - t1
    vb, lb, cb = O.save_bytes_semantic(merged, allvis)
    return secs, vb, lb, cb


def cpu_worker(args):
    """Child process of the one-process-per-core baseline: scenes [lo, hi) through the oracle; prints
    the seconds spent inside the oracle."""
    pkg = importlib.import_module("pcl-augmentation_amd")
    cfg = dict(CONFIGS[args.config], shuffle=args.order == "shuffled")
    lo, hi = args.cpu_worker
    spent = 0.0
    for seed in range(lo, hi):
        spent += oracle_scene(pkg, cfg, seed)[0]
    print(json.dumps({"scenes": hi - lo, "seconds": spent}))


def cpu_baselines(pkg, cfg, config_name, n_check, budget_s=20.0, all_cores=True):
    """Single core: the first scenes of rank 0's batch (their bytes are kept for the parity check),
    at most budget_s seconds.  All cores: one process per core over disjoint further scenes."""
    done, spent, kept = 0, 0.0, []
    while done < max(n_check, 1) or (done < 16 and spent < budget_s / 2):
        secs, vb, lb, cb = oracle_scene(pkg, cfg, scene_seed(0, done))
        if done < n_check:
            kept.append((vb, lb, cb))
        spent += secs
        done += 1
        if spent > budget_s:
            break
    single = {"value": round(done / spent, 3), "unit": "scenes/s", "cores": 1, "kind": "port",
              "sample": f"the first {done} scenes of the timed batch ({cfg['workload'].split(':')[0]}: "
                        f"{cfg['beams'] * cfg['az']} points, {len(cfg['kinds'])} inserts) through oracle.augment_scene "
                        "(NumPy port of the reference), one core"}
    if not all_cores:
        return single, None, kept
    visible = len(os.sched_getaffinity(0))
    cores = min(visible, 16)            # one GPU's share of the host (the box shows all of its cores to every lease)
    per = max(1, min(8, int(budget_s / max(spent / done, 1e-3) / 2)))
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--config", config_name, "--cpu-worker",
                               str(5000 + c * per), str(5000 + (c + 1) * per)] + (["--order", "shuffled"] if cfg.get("shuffle") else []),
                              stdout=subprocess.PIPE, text=True,
                              env={**os.environ, "OMP_NUM_THREADS": "1", "OPENBLAS_NUM_THREADS": "1", "MKL_NUM_THREADS": "1"})
             for c in range(cores)]
    outs = [p.communicate()[0] for p in procs]
    wall = time.perf_counter() - t0
    ok = all(p.returncode == 0 for p in procs)
    inner = [json.loads(o.strip().splitlines()[-1])["seconds"] for o in outs] if ok else []
    multi = {"value": round(cores * per / max(inner), 3) if ok else None, "unit": "scenes/s", "cores": cores,
             "kind": "port", "sample": f"{per} further scenes per process, one process per core "
                                       f"(os.sched_getaffinity shows {visible}, {cores} used: one GPU's share of the host); slowest process' time inside the oracle; "
                                       f"wall incl. interpreter start {wall:.1f} s"}
    return single, multi, kept


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


def placement_leg(pkg, torch, n_frames, with_cpu, kinds):
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
        for k, kind in enumerate(kinds):
            smp, line = synth.make_place_sample(s * 100 + k, kind)
            sa = fs.read_label_line(line)
            ok_map, ok_labels = fs.placement_surfaces(sa, config)
            queries.append({"scene": ps, "sample": smp, "anno": fs._anno10(sa), "ok_labels": ok_labels, "ok_map": ok_map})
            if s == 0:
                raw.append((scene9, f, smp, line))
    pb = pkg.places.PlaceBatch(queries, cand_cap=4)
    pb.run()
    torch.cuda.synchronize()
    ms = event_time_ms(torch, pb.run)
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


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks with torchrun as a CHILD process (this process has
    not touched a GPU and never will), pass its output through, exit with its code."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    return subprocess.call(cmd, env=env)


def extra_legs(pkg, torch, args, all_of_them):
    """The objects beside the headline that run on this GPU: placement_search, e2e, placed, e2e_files (config C5 has its own
    child process).  `all_of_them`: the default line; else only the ones asked for by flag."""
    out = {}
    only = getattr(args, "only_leg", None)
    want = lambda name, asked: (all_of_them or asked) and only in (None, name)

    def give_back():
        """A leg's batches, lanes and pinned staging go back before the next one starts: with the `placed` leg's three lanes
        (and their pinned slabs) still cached in this process, the first file leg wrote its files at 6.5-8 thousand frames/s
        where it reaches 12 in a fresh process."""
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        if hasattr(torch._C, "_host_emptyCache"):
            torch._C._host_emptyCache()

    if want("e2e_files", args.e2e_files):
        # file to file, the shapes of configs C3 (object detection, label_2) and C4 (SemanticKITTI sweep), files on tmpfs
        e2e = importlib.import_module("tools.e2e_pipeline")
        out["e2e_files"] = {}
        for shape in ("C3", "C4"):
            try:
                # BASELINE.json's sizes: 10 000 frames (C3), 23 201 = SemanticKITTI sequences 00-10 (C4); 16 frames of each
                # run, spread evenly over it, against the oracle's bytes
                full = {"C3": 10000, "C4": 23201}[shape]
                out["e2e_files"][shape] = e2e.measure_files(pkg, shape, n_frames=args.e2e_files or full,
                                                            check=0 if args.no_cpu_baseline else 16)
            except Exception as e:                             # the headline must not depend on this leg
                out["e2e_files"][shape] = {"error": repr(e)[:300]}
            give_back()
    if want("placement_search", args.placement > 0):
        out["placement_search"] = placement_leg(pkg, torch, args.placement or 64, not args.no_cpu_baseline, CONFIGS["C2"]["kinds"])
        give_back()
    if want("e2e", args.e2e > 0):
        e2e = importlib.import_module("tools.e2e_pipeline")
        out["e2e"] = e2e.measure(pkg, n_frames=args.e2e or 4096)
        give_back()
    if want("placed", args.placed):
        # the placement search in the loop (SURVEY.md par.8 f-1 + the hot path): search -> candidates -> merge per insert slot
        try:
            out["placed"] = importlib.import_module("tools.bench_placed").measure(pkg, B=args.placed or 256, K=len(CONFIGS["C2"]["kinds"]),
                                                                               reps=8, lanes=4)
        except Exception as e:
            out["placed"] = {"error": repr(e)[:300]}
    return out


def extra_legs_in_child(args, all_of_them):
    """The same, every leg in a child process of its own, so that nothing a leg does -- a frame flagged by the device, a fault
    of the runtime (DESIGN.md par.9 has two queue aborts of the placed lanes on record) -- can take the headline or another
    leg with it: the child's JSON line is merged, or its failure recorded."""
    wanted = [k for k, on in (("e2e_files", args.e2e_files > 0), ("placement_search", args.placement > 0), ("e2e", args.e2e > 0),
                              ("placed", args.placed > 0)) if on or all_of_them]
    out = {}
    for leg in wanted:
        cmd = [sys.executable, os.path.abspath(__file__), "--legs-child", "--only-leg", leg, "--placement", str(args.placement),
               "--e2e", str(args.e2e), "--placed", str(args.placed), "--e2e-files", str(args.e2e_files)]
        if all_of_them:
            cmd.append("--legs-all")
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                out.update(json.loads(lines[-1]))
                continue
            tail = [ln for ln in (r.stderr or "").strip().splitlines() if "amdgpu.ids" not in ln]
            named = [ln.strip() for ln in tail if "Kernel Name" in ln or "grid=" in ln][:2]
            why = f"the leg's process ended with code {r.returncode}: " + " | ".join(named + tail[-2:])[-600:]
        except Exception as e:
            why = repr(e)[:300]
        out[leg] = {"error": why}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2",
                    help="C2 = the configuration the metric is quoted on (default); C5 = the 1M-point stress run")
    ap.add_argument("--scenes", type=int, default=0, help="scenes per GPU batch (default: the config's)")
    ap.add_argument("--overlap", type=int, default=3,
                    help="consecutive steps alternate between this many full-size batches, each on its own HIP stream "
                         "(1 = every step on the same batch and stream)")
    ap.add_argument("--per-slot-launches", action="store_true",
                    help="one r3d_batch_insert call per insert slot instead of one r3d_batch_insert_many call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-scenes", type=int, default=8, help="scenes compared byte for byte with the oracle")
    ap.add_argument("--placement", type=int, default=0, metavar="SCENES",
                    help="also time the placement search (SURVEY.md par.8 f-1) on SCENES frames x 5 samples")
    ap.add_argument("--e2e", type=int, default=0, metavar="FRAMES",
                    help="also time the file-to-file pipeline (SURVEY.md par.8 f-2) on FRAMES frames: host frames in, "
                         "pinned double-buffered transfers overlapped with the kernels, host files' bytes out")
    ap.add_argument("--placed", type=int, default=0, metavar="FRAMES",
                    help="also time placement search + merge per insert slot (PlacedInserter) on a batch of FRAMES frames")
    ap.add_argument("--e2e-files", type=int, default=0, metavar="FRAMES",
                    help="also time the file-to-file pipelines of configs C3 / C4 (files on tmpfs) on FRAMES frames each")
    ap.add_argument("--distinct", type=int, default=0, metavar="N",
                    help="generate only N distinct scenes and cycle them through the batch (the inserts stay per scene); "
                         "default: every scene of the batch is its own")
    ap.add_argument("--cpu-budget", type=float, default=0.0, metavar="SECONDS",
                    help="seconds of oracle time for the single-core CPU baseline (default 20; 30 for C5); below 20 the "
                         "one-process-per-core leg is skipped")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="only the headline measurement: no c5 / e2e / placement_search objects in the line")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region of K steps is run this many times; `value` is the first, `repeats` in the line "
                         "holds min / median / max of all of them")
    ap.add_argument("--order", choices=["ring", "shuffled"], default="ring",
                    help="point order of the synthetic scans: ring-major then azimuth (the order of KITTI / SemanticKITTI files), "
                         "or the same scans with their points shuffled (SURVEY.md par.8d, BASELINE.md par.4)")
    ap.add_argument("--cpu-worker", type=int, nargs=2, metavar=("LO", "HI"), help=argparse.SUPPRESS)
    ap.add_argument("--legs-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--legs-all", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--only-leg", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args)
    if args.legs_child:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        pkg = importlib.import_module("pcl-augmentation_amd")
        import torch
        torch.cuda.set_device(0)
        print(json.dumps(extra_legs(pkg, torch, args, args.legs_all)), flush=True)
        return 0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # before the runtime starts

    cfg = dict(CONFIGS[args.config], shuffle=args.order == "shuffled")
    B = args.scenes or cfg["scenes"]
    cfg["workload"] = cfg["workload"].replace(f"batch of {cfg['scenes']} ", f"batch of {B} ")
    if cfg["shuffle"]:
        cfg["workload"] += "; points of every scan in a random order"
    if os.environ.get("R3D_BENCH_KINDS"):                             # (experiments: another mix of inserts, same count)
        mix = os.environ["R3D_BENCH_KINDS"].split(",")
        cfg["kinds"] = [mix[k % len(mix)] for k in range(len(cfg["kinds"]))]
    kinds = cfg["kinds"]
    pkg = importlib.import_module("pcl-augmentation_amd")
    synth = pkg.synth

    # CPU legs first, on rank 0, before this process starts the GPU runtime: the worker processes are plain children.
    # N > 1: a shorter single-core sample only (the other ranks wait for rank 0 in init_process_group meanwhile, and the
    # host's cores belong to all of them), so that every --gpus N line carries `cpu_baseline` and the oracle's bytes for
    # the parity check of rank 0's batch
    cpu_single = cpu_multi = None
    oracle_bytes = []
    if rank == 0 and not args.no_cpu_baseline:
        n_check = max(0, min(args.parity_scenes if args.config != "C5" else 1, B))   # a C5 scene takes the oracle ~10 s
        budget = args.cpu_budget or ((20.0 if args.config != "C5" else 30.0) if world == 1 else 10.0)
        cpu_single, cpu_multi, oracle_bytes = cpu_baselines(pkg, cfg, args.config, n_check, budget_s=budget,
                                                            all_cores=budget >= 20.0 and world == 1)

    # N > 1: every rank of the node on its own slice of the host's cores, on its GPU's NUMA node where sysfs says which that is --
    # before the GPU runtime starts, so that its helper threads and the pinned staging slabs' pages follow (affinity.py)
    binding = None
    if world > 1:
        binding = importlib.import_module("pcl-augmentation_amd.affinity").bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    import torch
    import torch.distributed as dist
    # R3D_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a
    # one-GPU box); the driver's runs use RCCL ("nccl") with one rank per GPU.
    backend = os.environ.get("R3D_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # under a launcher (RANK / WORLD_SIZE / MASTER_* in the environment) the process group is set up for ANY world size:
    # `torchrun --nproc-per-node 1 bench.py --gpus 1` runs the same init / barrier / all_reduce / destroy the 8-GPU run does
    # (tests/test_gpu_sharded.py runs it with RCCL, which wants one GPU per rank, on the one-GPU box)
    grouped = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if grouped:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    distinct = min(args.distinct, B) if args.distinct > 0 else B
    # (the generator is NumPy and releases the interpreter lock: a 1M-point scan takes 0.5 s, 64 of them 8 s on 8 threads)
    from concurrent.futures import ThreadPoolExecutor
    gen_threads = len(os.sched_getaffinity(0)) if (binding and binding.get("bound")) else len(os.sched_getaffinity(0)) // max(world, 1)
    with ThreadPoolExecutor(max_workers=max(1, min(16, gen_threads))) as gen:
        scenes = list(gen.map(lambda s: build_scene(synth, cfg, scene_seed(rank, s)), range(distinct)))
    scenes = [scenes[s % distinct] for s in range(B)]
    inserts = [synth.make_inserts(scene_seed(rank, s), kinds) for s in range(B)]
    K = len(kinds)
    n_max = max(len(x) for x, _ in scenes)
    grow = sum(max(len(inserts[s][k]) for s in range(B)) for k in range(K))
    n_pts = float(sum(len(x) for x, _ in scenes))
    m_pts = float(sum(len(i) for ins in inserts for i in ins))

    def make_batch():
        bt = pkg.SceneBatch(B, n_max + grow, grow, rows=cfg["rows"], cols=cfg["cols"], device=f"cuda:{local_rank}")
        bt.load(scenes)                                   # inputs resident in HBM from here on
        pk = [bt.pack_samples([inserts[s][k] for s in range(B)]) for k in range(K)]
        nd = torch.full((B,), MIN_POINTS, dtype=torch.int32, device=bt.device)
        return bt, pk, nd

    depth = 1 if args.per_slot_launches else max(1, args.overlap)
    lanes = [make_batch() for _ in range(depth)]
    lane_streams = [torch.cuda.Stream() for _ in lanes]
    batch, packed, need = lanes[0]

    def run_step(bt, pk, nd):
        bt.begin()
        if args.per_slot_launches:
            accs = [bt.insert_device(s5, off, nd)[1].clone() for s5, off in pk]
            bt.last_acc = torch.stack(accs)
        else:
            bt.last_vis, bt.last_acc = bt.insert_many_device(pk, [nd] * len(pk))
        bt.finish(check_cols=5)

    def enqueue_serial():
        run_step(batch, packed, need)

    lane_no = [0]

    def one_step():
        if depth == 1:
            return enqueue_serial()
        lane = lane_no[0] % depth
        lane_no[0] += 1
        with torch.cuda.stream(lane_streams[lane]):
            run_step(*lanes[lane])

    enqueue_serial()                      # first call (kernel attributes, lazy allocations)
    torch.cuda.synchronize()
    batch.raise_on_status()
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
    for bt, _, _ in lanes:
        bt.raise_on_status()              # (every lane has run at least once; SceneBatch also learns its clouds' point order here)
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # the same K steps again, --repeats - 1 more times (each bracketed like the first): the spread beside `value`
    # (`value` itself is the first, the contract's, timed region)
    region_s = [elapsed]
    for _ in range(max(0, args.repeats - 1)):
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        tr = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
        region_s.append(time.perf_counter() - tr)
    if grouped:
        t = torch.tensor(region_s, dtype=torch.float64, device=batch.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_s = [float(v) for v in t.tolist()]
        elapsed = region_s[0]
    for bt, _, _ in lanes:
        bt.raise_on_status()
    lanes_checked = 0
    if depth > 1:
        # the batches of the timed, overlapped run: every lane worked on the same input, so the lanes must agree byte
        # for byte (lane 0 is compared with the oracle below, BEFORE anything runs on it again)
        ref = lanes[0][0]
        ref_out = ref.n_out.cpu().numpy()
        for bt, _, _ in lanes[1:]:
            assert np.array_equal(bt.n_out.cpu().numpy(), ref_out), "lanes of the overlapped run disagree (n_out)"
            same = bool(torch.equal(bt.out_xyzi.view(torch.int32), ref.out_xyzi.view(torch.int32))) and \
                bool(torch.equal(bt.out_label, ref.out_label)) and bool(torch.equal(bt.n_log, ref.n_log))
            m = int(ref.n_log.max().item())
            same = same and bool(torch.equal(bt.check[:, :m].view(torch.int32), ref.check[:, :m].view(torch.int32)))
            if not same:
                raise SystemExit("parity FAILED: lanes of the three-in-flight run disagree")
            lanes_checked += 1
        overlapped_bytes = None
        if rank == 0 and oracle_bytes:
            n_out_h, n_log_h = ref_out, ref.n_log.cpu().numpy()
            for s, (vb, lb, cb) in enumerate(oracle_bytes):
                xyzi = ref.out_xyzi[s, :n_out_h[s]].cpu().numpy()
                label = ref.out_label[s, :n_out_h[s]].cpu().numpy().view(np.uint32)
                check = ref.check[s, :n_log_h[s]].cpu().numpy()
                if xyzi.tobytes() != vb or label.tobytes() != lb or check.tobytes() != cb:
                    raise SystemExit(f"parity FAILED: scene {s} of the three-in-flight run differs from the oracle")
            overlapped_bytes = len(oracle_bytes)
        # the same K steps one at a time, for the record (not the headline value)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            enqueue_serial()
        torch.cuda.synchronize()
        serial_elapsed = time.perf_counter() - t1
    else:
        serial_elapsed = elapsed
    batch.debug_counters(reset=True)
    batch.count_pairs(True)                               # (the per-pair counters cost atomics on the chain's critical path: this step only)
    enqueue_serial()
    insert_paths = batch.debug_counters(reset=True)       # which way every pair of one step went
    batch.count_pairs(False)
    n_out = batch.n_out.cpu().numpy()
    n_accepted = int(batch.last_acc.sum().item())
    n_appended = int((batch.last_vis * batch.last_acc).sum().item()) if hasattr(batch, "last_vis") else None
    rebases = int(batch.rebase.sum().item())        # informational: a rebase leaves the inputs intact

    # N > 1: the PCIe-inclusive leg on EVERY rank at the same time -- the ranks of a node share the host's cores and memory
    # system, which is what bounds that rate: frames of all ranks / the slowest rank's time
    e2e_all = None
    if world > 1 and not args.no_extra_legs and args.config == "C2":
        e2e = importlib.import_module("tools.e2e_pipeline")
        cores = len(os.sched_getaffinity(0))
        # (bound: the mask already is this rank's slice of the host)
        threads = max(1, min(16, cores if (binding and binding.get("bound")) else cores // world))
        for bt, _, _ in lanes[1:]:
            bt.ws = None                                            # the resident lanes' pools are not needed any more
        del lanes[1:]
        torch.cuda.empty_cache()
        mine = e2e.measure(pkg, n_frames=args.e2e or 2048, pack_threads=threads, device=f"cuda:{local_rank}", before_timed=dist.barrier)
        t = torch.tensor([mine["seconds"]], dtype=torch.float64, device=batch.device if backend == "nccl" else "cpu")
        f = torch.tensor([float(mine["frames"])], dtype=torch.float64, device=t.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(f, op=dist.ReduceOp.SUM)
        e2e_all = {"frames_per_s_all_ranks": round(float(f.item()) / float(t.item()), 1), "frames": int(f.item()),
                   "slowest_rank_seconds": round(float(t.item()), 3), "rank0_frames_per_s": mine["frames_per_s"],
                   "pack_threads_per_rank": threads, "host_cores_visible": cores, "ranks": world, "rank0_core_binding": binding,
                   "what": "every rank streams its own frames host memory -> GPU -> host memory at the same time "
                           "(tools/e2e_pipeline.measure, delta download); the ranks share the host"}

    if rank == 0:
        import ctypes as C
        L = pkg._lib
        lib, desc = batch.lib, batch.desc
        # parity: the timed batch's outputs against the oracle's bytes for its first scenes
        parity_checked = 0
        if oracle_bytes:
            n_log_h = batch.n_log.cpu().numpy()
            for s, (vb, lb, cb) in enumerate(oracle_bytes):
                xyzi = batch.out_xyzi[s, :n_out[s]].cpu().numpy()
                label = batch.out_label[s, :n_out[s]].cpu().numpy().view(np.uint32)
                check = batch.check[s, :n_log_h[s]].cpu().numpy()
                if xyzi.tobytes() != vb or label.tobytes() != lb or check.tobytes() != cb:
                    raise SystemExit(f"parity FAILED: scene {s} of the timed batch differs from the oracle")
                parity_checked += 1

        def one(which):
            return lambda: L.check(lib.r3d_batch_launch_one(C.byref(desc), which, L.stream_ptr()), "launch_one")

        # per-kernel timing after the timed region, HIP events on the launch stream
        batch.begin()
        t_bounds = event_time_ms(torch, one(L.K_BOUNDS))
        t_prepare = event_time_ms(torch, one(L.K_PREPARE))
        t_project = event_time_ms(torch, one(L.K_PROJECT))
        t_begin = event_time_ms(torch, batch.begin)

        def all_inserts():                          # begin() restores the state the inserts mutate
            batch.begin()
            if args.per_slot_launches:
                for s5, off in packed:
                    batch.insert_device(s5, off, need)
            else:
                batch.insert_many_device(packed, [need] * K)

        t_insert_all = event_time_ms(torch, all_inserts) - t_begin
        batch.finish(check_cols=5)
        t_write = event_time_ms(torch, one(L.K_ALIVE_WRITE))
        t_finish = event_time_ms(torch, lambda: batch.finish(check_cols=5))
        n_out_pts = float(batch.n_out.sum().item())
        n_log_pts = float(batch.n_log.sum().item())
        # Algorithmic bytes per launch (SURVEY.md par.8d with the conservative rule:
        # a pass that is not made is not claimed).  bounds / project read xyzi once (16 B per point);
        # the compaction reads xyzi + label and writes the survivors (20 B + 20 B per point); an
        # insert launch reads every sample row twice (2 x 40 B per sample point).
        ins_name = "k_insert_chain" if not args.per_slot_launches else "k_insert_chain(1 slot)"
        kernels = {
            "k_bounds": {"ms": t_bounds, "launches_per_step": 1, "alg_bytes": 16.0 * n_pts},
            "k_project": {"ms": t_project, "launches_per_step": 1, "alg_bytes": 16.0 * n_pts},
            ins_name: ({"ms": t_insert_all, "launches_per_step": 1, "alg_bytes": 80.0 * m_pts} if not args.per_slot_launches
                       else {"ms": t_insert_all / K, "launches_per_step": K, "alg_bytes": 80.0 * m_pts / K}),
            "k_alive_write": {"ms": t_write, "launches_per_step": 1, "alg_bytes": 20.0 * n_pts + 20.0 * n_out_pts},
            "k_prepare": {"ms": t_prepare, "launches_per_step": 1, "alg_bytes": 0.0},
        }
        pmc, pmc_file = {}, f"profiles/{PROFILE_TAG}_pmc.json" if args.config != "C5" else f"profiles/{PROFILE_TAG}_c5_pmc.json"
        try:
            pmc = json.load(open(os.path.join(ROOT, pmc_file)))["kernels"]
        except Exception:
            pass
        step_ms_serial = 1e3 * serial_elapsed / args.steps
        for name, k in kernels.items():
            k["GBps"] = k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9
            k["frac"] = k["GBps"] / HBM_PEAK_GBS
            k["share_of_step"] = k["ms"] * k["launches_per_step"] / step_ms_serial      # of a step run alone
        dominant = max(kernels, key=lambda k: kernels[k]["ms"] * kernels[k]["launches_per_step"])
        dk = kernels[dominant]
        pmc_key = "k_insert_chain" if dominant.startswith("k_insert") else dominant
        roofline = {
            "bound": "hbm", "kernel": dominant, "achieved": round(dk["GBps"], 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(dk["frac"], 4),
            "traffic": pmc.get(pmc_key, {}).get("hbm_bytes_corrected"),
            "traffic_source": f"{pmc_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                              "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch" if pmc_key in pmc else None,
            "alg_bytes_per_launch": dk["alg_bytes"], "ms_per_launch": round(dk["ms"], 4),
            "launches_per_step": dk["launches_per_step"],
            "note": "the insert kernel works in LDS on the window of the range image around the inserted object; it is "
                    "latency-bound by design and moves almost no HBM bytes (the incremental pipeline removed the "
                    "per-insert streaming passes), so its HBM fraction is low by construction; the HBM-bound streaming "
                    "kernels are in all_kernels, the whole step's fraction is pipeline_frac_of_hbm_peak."
                    if dominant.startswith("k_insert") else "",
            "all_kernels": {n: {"ms_per_launch": round(k["ms"], 4), "launches_per_step": k["launches_per_step"],
                                "alg_GBps": round(k["GBps"], 1), "frac_of_hbm_peak": round(k["frac"], 4),
                                "share_of_step": round(k["share_of_step"], 3),
                                "pmc_hbm_bytes_per_launch": pmc.get("k_insert_chain" if n.startswith("k_insert") else n, {}).get("hbm_bytes_corrected")}
                            for n, k in kernels.items()},
            "api_calls_ms": {"r3d_batch_begin": round(t_begin, 4), f"r3d_batch_insert_many_{K}" if not args.per_slot_launches
                             else f"r3d_batch_insert_x{K}": round(t_insert_all, 4), "r3d_batch_finish": round(t_finish, 4)},
        }
        scenes_per_s = B * world * args.steps / elapsed
        # whole-step algorithmic bytes this pipeline needs (per GPU): bounds + project + compaction +
        # inserts + the check rows (40 B read, 20 B written per inserted point)
        step_bytes = 16.0 * n_pts * 2 + 20.0 * n_pts + 20.0 * n_out_pts + 80.0 * m_pts + 60.0 * n_log_pts
        pipe_gbs = step_bytes * args.steps / elapsed / 1e9
        # SURVEY.md par.8d's floor for the same result: every point read once and written once, 40 B per point
        floor_bytes = 20.0 * n_pts + 20.0 * n_out_pts
        floor_gbs = floor_bytes * args.steps / elapsed / 1e9
        out = {
            "metric": "augmented scenes/sec (120k-pt, 64-beam)" if args.config != "C5" else "augmented scenes/sec (1M-pt, 256-beam)",
            "value": round(scenes_per_s, 1),
            "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "repeats": {"regions": len(region_s), "steps_each": args.steps,
                        "ms_per_step": [round(1e3 * r / args.steps, 3) for r in region_s],
                        "scenes_per_s_min": round(B * world * args.steps / max(region_s), 1),
                        "scenes_per_s_median": round(B * world * args.steps / float(np.median(region_s)), 1),
                        "scenes_per_s_max": round(B * world * args.steps / min(region_s), 1)},
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["workload"],
                       "scenes_per_gpu": B, "distinct_scenes": distinct, "point_order": args.order,
                       "process_group": backend if grouped else None,
                       "points_per_scene": int(n_pts / B), "inserts_per_scene": K,
                       "range_image": [batch.rows, batch.cols], "inserts_accepted": n_accepted, "inserts_tried": B * K,
                       "points_appended": n_appended, "points_culled": None if n_appended is None else int(n_pts + n_appended - n_out.sum()),
                       "steps_in_flight": depth, "check_rows_in_timed_region": True,
                       "scenes_per_s_one_step_in_flight": round(B * world * args.steps / serial_elapsed, 1),
                       "ms_per_step_one_step_in_flight": round(step_ms_serial, 3),
                       "insert_api": f"r3d_batch_insert x{K}" if args.per_slot_launches else f"r3d_batch_insert_many({K})",
                       "rebases_in_timed_steps": rebases, "settle_steps_in_setup": settle,
                       "mean_points_out": float(n_out.mean()), "insert_paths_one_step": insert_paths,
                       # SceneBatch(order="auto"): the first begin of a batch object looks at the point order (four small
                       # launches, ~20 us per 256 scenes); once a batch has come through without an unordered scene the begins
                       # carry R3D_B_FILE_ORDER -- the timed begins do -- and every 64th looks again.  `shuffled` pays the look
                       # and the re-numbering in every timed begin.
                       "point_order_look": "outside the timed region" if args.order == "ring" else "inside every timed begin",
                       # points of ONE step of the timed batch (begin + the inserts) whose pixel the reference formula decided
                       # with the fractional row / column position within 1e-12 of an integer: where an ULP of arctan2 /
                       # arccos could move a pixel against the reference (expected 0)
                       "bin_edge_risk_points": {"scene": insert_paths.get("bin_edge_risk_scene_points"),
                                                "sample": insert_paths.get("bin_edge_risk_sample_points"),
                                                "scene_points_projected": int(n_pts)}},
            "roofline": roofline,
            "pipeline_alg_GBps_per_gpu": round(pipe_gbs, 1),
            "pipeline_frac_of_hbm_peak": round(pipe_gbs / HBM_PEAK_GBS, 4),
            "pipeline_alg_bytes_per_scene": round(step_bytes / B, 1),
            "pipeline_frac_floor_model": round(floor_gbs / HBM_PEAK_GBS, 4),
            "pipeline_byte_models": {"passes_made": "16 N (bounds) + 16 N (project) + 20 N + 20 N_out (compaction) + 80 M (inserts) + "
                                                    "60 M_visible (check rows): what this pipeline's kernels have to move",
                                     "floor": "20 N + 20 N_out: every point read once and written once (SURVEY.md par.8d)"},
            "parity_checked": parity_checked,
            "parity_of_overlapped_run": {"scenes_vs_oracle_lane0": overlapped_bytes if depth > 1 else None,
                                         "lanes_byte_equal_to_lane0": lanes_checked},
        }
        if e2e_all is not None:
            out["e2e_all_ranks"] = e2e_all
        if cpu_single is not None:
            out["cpu_baseline"] = cpu_single
            out["cpu_baseline_all_cores"] = cpu_multi
        extra = world == 1 and args.config == "C2" and not args.no_extra_legs
        t_extra = time.perf_counter()
        if world == 1 and (extra or args.placement > 0 or args.e2e > 0 or args.placed or args.e2e_files):
            out.update(extra_legs_in_child(args, extra))
        if extra:
            # BASELINE.json's stress configuration in the same line: a child process (its own batches, freed when it ends);
            # 64 distinct scans (each four times in the batch of 256), 10 timed steps
            cmd = [sys.executable, os.path.abspath(__file__), "--config", "C5", "--scenes", "256", "--distinct", "64", "--steps", "10",
                   "--warmup", "2", "--repeats", "2", "--no-extra-legs", "--parity-scenes", "1", "--cpu-budget", "8"]
            if args.no_cpu_baseline:
                cmd.append("--no-cpu-baseline")
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
                c5 = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                out["c5"] = {"scans_per_s": c5["value"], "ms_per_step": c5["ms_per_step"], "scans_per_batch": c5["config"]["scenes_per_gpu"],
                             "distinct_scans": c5["config"].get("distinct_scenes"), "steps": c5["steps"],
                             "points_per_scan": c5["config"]["points_per_scene"], "inserts_per_scan": c5["config"]["inserts_per_scene"],
                             "range_image": c5["config"]["range_image"], "steps_in_flight": c5["config"]["steps_in_flight"],
                             "ms_per_step_one_step_in_flight": c5["config"]["ms_per_step_one_step_in_flight"],
                             "pipeline_frac_of_hbm_peak": c5["pipeline_frac_of_hbm_peak"],
                             "pipeline_frac_floor_model": c5.get("pipeline_frac_floor_model"),
                             "api_calls_ms": c5["roofline"]["api_calls_ms"], "parity_checked": c5["parity_checked"],
                             "insert_paths_one_step": c5["config"].get("insert_paths_one_step"),
                             "cpu_baseline": c5.get("cpu_baseline"), "workload": c5["config"]["workload"],
                             "command": " ".join(cmd[1:])}
            except Exception as e:                                 # the headline must not depend on this leg
                out["c5"] = {"error": repr(e)[:300]}
            # ... and the headline's configuration with the points of every scan in a random order (SURVEY.md par.8d,
            # the reference is order-agnostic, insertion.py:100-127; the chunk boxes of the incremental
            # design lean on the order of LiDAR files)
            cmd = [sys.executable, os.path.abspath(__file__), "--config", "C2", "--order", "shuffled", "--steps", "10", "--warmup", "3",
                   "--repeats", "2", "--no-extra-legs", "--parity-scenes", "2", "--cpu-budget", "2"]
            if args.no_cpu_baseline:
                cmd.append("--no-cpu-baseline")
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                sh = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                paths = sh["config"].get("insert_paths_one_step") or {}
                out["shuffled"] = {"scenes_per_s": sh["value"], "ms_per_step": sh["ms_per_step"],
                                   "step_time_vs_ring_major": round(sh["ms_per_step"] / out["ms_per_step"], 2),
                                   "ms_per_step_one_step_in_flight": sh["config"]["ms_per_step_one_step_in_flight"],
                                   "api_calls_ms": sh["roofline"]["api_calls_ms"],
                                   "chunks_listed_per_pair": paths.get("chunks_listed_per_pair"),
                                   "scenes_in_sorted_order": paths.get("scenes_in_sorted_order"),
                                   "insert_paths_one_step": paths,
                                   "parity_checked": sh["parity_checked"], "workload": sh["config"]["workload"],
                                   "command": " ".join(cmd[1:])}
            except Exception as e:
                out["shuffled"] = {"error": repr(e)[:300]}
            out["extra_legs_seconds"] = round(time.perf_counter() - t_extra, 1)
        print(json.dumps(out))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
