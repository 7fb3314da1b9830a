import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("pcl-augmentation_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("pcl-augmentation_amd.synth")


def blob_in_front_of_extreme(xyzi, which="max", n=200, seed=0):
    """A small float64 blob half way to the scene's max- (or min-) elevation point: it covers that
    point's pixel, so the point is culled and the elevation bounds move (forces a rebase)."""
    xyz = xyzi[:, :3].astype(np.float64)
    r = np.sqrt((xyz * xyz).sum(1))
    el = np.arccos(xyz[:, 2] / r)
    i = int(np.argmax(el) if which == "max" else np.argmin(el))
    rng = np.random.default_rng(seed)
    pts = xyz[i] * 0.5 + rng.normal(0.0, 0.01, size=(n, 3))
    return np.column_stack([pts, rng.random(n), np.full(n, 30.0)])


def rows_from_alive_words(batch, s):
    """The float64 rows [x y z label] of scene s put together from what a placement query with R3D_PQ_SCENE_SLAB reads: the
    float32 slab, the label words, r3d_batch_export_alive's words, the log rows of the inserted points."""
    n_total, n_head = int(batch.n_total[s]), int(batch.n_head[s])
    words = batch.export_alive()[s].cpu().numpy().view(np.uint64)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little").astype(bool)
    assert not bits[n_total:].any()                                  # (masked to the scene's count)
    xyz = batch.xyzi[s, :n_total, :3].cpu().numpy().astype(np.float64)
    ref = batch.tail_ref[s].cpu().numpy()[:n_total - n_head]
    xyz[n_head:] = batch.log5[s].cpu().numpy()[ref, :3]
    lab = (batch.label[s, :n_total].cpu().numpy() & 0xFFFF).astype(np.float64)
    return np.c_[xyz, lab][bits[:n_total]]
