#!/usr/bin/env python3
"""File-to-file augmentation sharded over the GPUs of one node (BASELINE config C4).

    python tools/run_sharded_pipeline.py --gpus 8 --velodyne IN/velodyne --labels IN/labels \\
        --plan PLAN_DIR --output OUT --folder run0 [--dataset semantic|kitti] [--batch 64]

Rank r of G takes frames r, r + G, ... (sorted by file name) and writes ``OUT/<folder>/velodyne|labels|check``
exactly as the reference's ``save_data`` does; frames whose outputs exist are skipped (resume).  Started
plainly it starts the G ranks itself -- torchrun as a CHILD process, before anything here touches a GPU --
and exits with the child's code; started under torchrun it is one rank.  No collective on the data path:
the ranks only sum their counters at the end (gloo).

The insert plan: ``PLAN_DIR/<frame>.npz`` with ``samples`` (rows x 5 float64: x y z intensity label, the
inserts' points one after the other), ``sizes`` (points per insert) and ``min_points`` (per insert) -- what
the placement step (``find_possible_places`` + the driver's choice) produced for that frame.
``--synthetic-plan K`` makes K synthetic inserts per frame instead (tests, dry runs).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def spawn(args):
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    return subprocess.call(cmd, env=env)


def load_plan(plan_dir, name):
    with np.load(os.path.join(plan_dir, name + ".npz")) as z:
        sizes = z["sizes"].astype(np.int64)
        parts = np.split(z["samples"].astype(np.float64), np.cumsum(sizes)[:-1]) if len(sizes) else []
        return parts, [int(x) for x in z["min_points"]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--velodyne", required=True)
    ap.add_argument("--labels", required=True)
    ap.add_argument("--plan")
    ap.add_argument("--synthetic-plan", type=int, default=0, metavar="K")
    ap.add_argument("--output", required=True)
    ap.add_argument("--folder", default="run0")
    ap.add_argument("--dataset", choices=["semantic", "kitti"], default="semantic")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lanes", type=int, default=3)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn(args))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pkg = importlib.import_module("pcl-augmentation_amd")
    # every rank on its own slice of the host's cores (its GPU's NUMA node where sysfs tells), before the GPU runtime starts
    binding = pkg.affinity.bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else None
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")                       # counters only; the data path has no collective
    names = sorted(os.path.splitext(f)[0] for f in os.listdir(args.velodyne) if f.endswith(".bin"))
    frames = [pkg.Frame(os.path.join(args.velodyne, n + ".bin"), os.path.join(args.labels, n + ".label"), n) for n in names]

    def inserts_for(i):
        if args.synthetic_plan:
            kinds = (["pedestrian", "cyclist", "car"] * args.synthetic_plan)[:args.synthetic_plan]
            return ([pkg.synth.make_insert(1000 * i + k, kind, rng_range=(5.0, 12.0)) for k, kind in enumerate(kinds)],
                    [15] * len(kinds))
        return load_plan(args.plan, names[i])

    import torch
    n_dev = max(torch.cuda.device_count(), 1)
    st = pkg.run_sharded_files(frames, inserts_for, args.output, args.folder, rank, world, device=f"cuda:{local_rank % n_dev}",
                               dataset=args.dataset, batch_size=args.batch, lanes=args.lanes)
    tot = torch.tensor([st["written"], st["skipped_existing"], st["inserted"]], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(tot)
    line = json.dumps({"rank": rank, "world_size": world, "frames": len(frames), "mine": len(st["frame_indices"]),
                       "written": st["written"], "skipped_existing": st["skipped_existing"],
                       "frames_per_s": round(st.get("frames_per_s", 0.0), 1), "core_binding": binding,
                       "all_ranks": {"written": int(tot[0]), "skipped_existing": int(tot[1]), "inserted": int(tot[2])}})
    sys.stdout.flush()
    os.write(1, (line + "\n").encode())                      # one write: the ranks share the launcher's pipe
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
