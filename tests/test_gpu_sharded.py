"""The N > 1 path with the real library (-m gpu): two spawned processes (gloo), both on cuda:0, each run
`run_sharded` -> `augment_batch` on its shard of 2 x 32 scenes -- insert_many chains of both processes on the GPU at
the same time, which is where the chain kernel's ordering of a scene's slots has to hold up against foreign
workgroups on the CUs; the union of the shards equals the oracle byte for byte, no R3D_S_CHAIN_TIMEOUT.  And
tools/run_sharded_pipeline.py with two ranks on the one GPU, files in -> files out."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
N_SCENES, K = 64, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _job(synth, i):
    xyzi, label = synth.make_scene(900 + i, 24, 500)
    ins = [synth.make_insert(9000 + 10 * i + k, kind, rng_range=(5.0, 14.0)) for k, kind in
           enumerate(["pedestrian", "car", "cyclist", "pedestrian"][:K])]
    return xyzi, label, [[x] for x in ins], [15] * K


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("pcl-augmentation_amd")
    from oracle import real3d_oracle as O
    synth = pkg.synth
    reuse = {}

    def process(indices):
        out = []
        for rep in range(3):                                  # several batches per rank: the two processes overlap for sure
            jobs = [_job(synth, i) for i in indices]
            res, acc = pkg.augment_batch([(j[0], j[1]) for j in jobs], [j[2] for j in jobs], [j[3] for j in jobs],
                                         device="cuda:0", reuse=reuse)
            out = [(r[0].tobytes(), r[1].tobytes(), r[2].tobytes(), a) for r, a in zip(res, acc)]
        # this rank's shard against the oracle
        for i, (vb, lb, cb, a) in zip(indices, out):
            j = _job(synth, i)
            merged, allvis, oacc = O.augment_scene(synth.scene5_from_packed(j[0], j[1]), j[2], j[3])
            assert a == oacc and (vb, lb, cb) == O.save_bytes_semantic(merged, allvis), f"scene {i} on rank {rank}"
        return out

    dist.barrier()                                            # start the GPU work together
    merged = pkg.run_sharded(N_SCENES, process)
    if rank == 0:
        assert sorted(merged) == list(range(N_SCENES))
        np.save(os.path.join(out_dir, "accepted.npy"), np.array([merged[i][3] for i in range(N_SCENES)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_share_one_gpu(tmp_path):
    mp.start_processes(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    acc = np.load(tmp_path / "accepted.npy")
    assert acc.shape == (N_SCENES, K) and (acc >= 0).sum() > N_SCENES


def test_sharded_file_driver_two_ranks_one_gpu(synth, tmp_path):
    from test_pipeline import _candidates, _check_outputs, _make_dataset
    frames = _make_dataset(synth, tmp_path / "in", 9)
    os.makedirs(tmp_path / "plan")
    for i in range(9):
        slots, need = _candidates(synth, i)
        np.savez(tmp_path / "plan" / f"{i:06d}.npz", samples=np.vstack([s[0] for s in slots]),
                 sizes=np.array([len(s[0]) for s in slots]), min_points=np.array(need))
    args = ["--gpus", "2", "--velodyne", str(tmp_path / "in" / "velodyne"), "--labels", str(tmp_path / "in" / "labels"),
            "--plan", str(tmp_path / "plan"), "--output", str(tmp_path / "out"), "--folder", "c4", "--batch", "4", "--lanes", "2"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sharded_pipeline.py")] + args, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1] and all(l["all_ranks"]["written"] == 9 for l in lines)
    _check_outputs(synth, frames, tmp_path / "out", "c4", False)


def test_bench_command_with_four_ranks():
    """The command the driver runs on a multi-GPU node -- `bench.py --gpus N`: torchrun child, every rank bound to its slice of
    the host's cores before its GPU runtime starts (affinity.bind_rank), process-group init, barriers around the timed region,
    all_reduce(MAX) of the time, the PCIe-inclusive leg on every rank at the same time (e2e_all_ranks) -- with four ranks on
    this one GPU (R3D_DIST_BACKEND=gloo: RCCL wants one GPU per rank).  One JSON line, n_gpus 4, parity keys present."""
    env = {**os.environ, "R3D_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--scenes", "64",
                        "--repeats", "2", "--e2e", "256", "--cpu-budget", "3", "--parity-scenes", "2"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 4 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["scenes_per_gpu"] == 64 and out["repeats"]["regions"] == 2
    assert out["parity_of_overlapped_run"]["lanes_byte_equal_to_lane0"] == 2 and "parity_checked" in out
    assert out["roofline"]["frac"] > 0 and out["e2e_all_ranks"]["ranks"] == 4 and out["e2e_all_ranks"]["frames_per_s_all_ranks"] > 0
    bound = out["e2e_all_ranks"]["rank0_core_binding"]
    assert bound["bound"] and 1 <= bound["cores"] <= max(1, bound["cores_of_the_host_share"] // 4 + 1)
    assert out["e2e_all_ranks"]["pack_threads_per_rank"] == min(16, bound["cores"])
    # every --gpus N line carries the CPU baseline (rank 0 times the oracle before it starts its GPU runtime) and rank 0's
    # batch is compared with the oracle's bytes
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["kind"] == "port"
    assert out["parity_checked"] == 2 and out["parity_of_overlapped_run"]["scenes_vs_oracle_lane0"] == 2


def test_bench_under_torchrun_with_rccl_world_of_one():
    """The code the 8-GPU driver executes, once, on the one GPU this box has: `torchrun --nproc-per-node 1 bench.py --gpus 1`
    with the default backend ("nccl" = RCCL): init_process_group with device_id, barriers around the timed region, all_reduce
    (MAX) of a GPU tensor, destroy_process_group.  (No scaling is measured here; `n_gpus` is 1.)"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    env.pop("R3D_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--scenes", "64",
           "--repeats", "2", "--no-extra-legs", "--cpu-budget", "2", "--parity-scenes", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["config"]["process_group"] == "nccl"
    assert out["cpu_baseline"]["value"] > 0 and out["roofline"]["frac"] > 0 and out["parity_checked"] == 1
