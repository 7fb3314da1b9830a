"""BASELINE config C4's driver on CPU: tools/run_sharded_pipeline.py with two ranks (torchrun child, gloo),
files in -> files out, every frame written by exactly one rank, bytes equal to the oracle's, a second run
resumes by existence.  The GPU leg is replaced by the oracle here (tests/sharded_files_child.py wraps the driver's
main() with it -- the driver itself has no such switch); tests/test_gpu_sharded.py runs two ranks with the real library
on one GPU."""
import json
import os
import subprocess
import sys

from conftest import ROOT
from test_pipeline import _check_outputs, _make_dataset


def _run(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sharded_files_child.py")] + args, capture_output=True,
                       text=True, timeout=600, env={**os.environ, "OMP_NUM_THREADS": "1"})
    assert p.returncode == 0, p.stdout + p.stderr
    return [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_two_ranks_file_to_file(synth, tmp_path, monkeypatch):
    import test_pipeline
    frames = _make_dataset(synth, tmp_path / "in", 5)
    # the plan the ranks read: the candidates test_pipeline._expected assumes, one file per frame
    import numpy as np
    os.makedirs(tmp_path / "plan")
    for i in range(5):
        slots, need = test_pipeline._candidates(synth, i)
        np.savez(tmp_path / "plan" / f"{i:06d}.npz", samples=np.vstack([s[0] for s in slots]),
                 sizes=np.array([len(s[0]) for s in slots]), min_points=np.array(need))
    args = ["--gpus", "2", "--velodyne", str(tmp_path / "in" / "velodyne"), "--labels", str(tmp_path / "in" / "labels"),
            "--plan", str(tmp_path / "plan"), "--output", str(tmp_path / "out"), "--folder", "c4", "--batch", "2"]
    lines = _run(args)
    assert sorted(l["rank"] for l in lines) == [0, 1]
    assert sorted(l["mine"] for l in lines) == [2, 3] and all(l["all_ranks"]["written"] == 5 for l in lines)
    _check_outputs(synth, frames, tmp_path / "out", "c4", False)
    lines = _run(args)                                            # resume: every output exists
    assert all(l["all_ranks"]["written"] == 0 and l["all_ranks"]["skipped_existing"] == 5 for l in lines)
