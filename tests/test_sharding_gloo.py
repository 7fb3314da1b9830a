"""Multi-GPU path on CPU: two ranks over gloo (not gpu).

Scenes are independent, so the N > 1 path is sharding + gathering host logic around the one-GPU
pipeline.  Here the per-shard processing function is the oracle (the HIP library needs a GPU);
what is proven is that every scene is processed exactly once and that the union of the shards
is byte-identical to the unsharded run, plus the bench's max-over-ranks timing reduction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene_job(seed):
    import importlib
    from oracle import real3d_oracle as O
    synth = importlib.import_module("pcl-augmentation_amd.synth")
    xyzi, label = synth.make_scene(seed, 16, 300)
    ins = [synth.make_insert(seed * 10 + k, kind, rng_range=(5.0, 12.0)) for k, kind in enumerate(["pedestrian", "car"])]
    merged, allvis, acc = O.augment_scene(synth.scene5_from_packed(xyzi, label), [[x] for x in ins], [10, 10])
    return O.save_bytes_semantic(merged, allvis), acc


def _worker(rank, world, port, n_scenes, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("pcl-augmentation_amd")
    seen = []

    def process(indices):
        seen.extend(indices)
        return [_scene_job(100 + i) for i in indices]

    merged = pkg.run_sharded(n_scenes, process)
    assert seen == pkg.shard_indices(n_scenes, rank, world)
    # the bench's timing reduction: every rank reports the slowest rank's time
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(world)
    if rank == 0:
        np.save(os.path.join(out_dir, "keys.npy"), np.array(sorted(merged)))
        blob = b"".join(merged[i][0][0] + merged[i][0][1] + merged[i][0][2] for i in sorted(merged))
        open(os.path.join(out_dir, "union.bin"), "wb").write(blob)
    dist.barrier()
    dist.destroy_process_group()


def test_union_of_two_shards_equals_single_run(tmp_path):
    n = 7                                            # odd: ragged shards (4 + 3)
    mp.start_processes(_worker, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    assert list(np.load(tmp_path / "keys.npy")) == list(range(n))
    single = b"".join(b"".join(_scene_job(100 + i)[0]) for i in range(n))
    assert (tmp_path / "union.bin").read_bytes() == single


def test_shard_indices_partition(pkg):
    for n in (0, 1, 8, 23201):
        for g in (1, 2, 4, 8):
            parts = [pkg.shard_indices(n, r, g) for r in range(g)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        pkg.shard_indices(4, 2, 2)
