"""The Waymo flavour at file level: the frame reader against what the reference's own Waymo.__getitem__ returned
(tests/golden/make_golden_waymo_reader.py), and lidar/*.npy in -> lidar/ labels_v3_2/ check/*.npy out through
AugmentPipeline(dataset="waymo") -- on CPU with the oracle injected, on the GPU through r3d_batch_begin_f64."""
import io
import os

import numpy as np
import pytest

from conftest import load_golden
from oracle import real3d_oracle as O


def _tree(root, lidar, labels, pose, name="000007"):
    seq = os.path.join(root, "seq_a")
    for sub in ("lidar", "labels_v3_2", "poses"):
        os.makedirs(os.path.join(seq, sub), exist_ok=True)
    np.save(os.path.join(seq, "lidar", f"{name}.npy"), lidar)
    np.save(os.path.join(seq, "labels_v3_2", f"{name}.npy"), labels)
    if pose is not None:
        np.save(os.path.join(seq, "poses", f"{name}.npy"), pose)
    return os.path.join(seq, "lidar", f"{name}.npy")


def test_reader_equals_the_reference_class(pkg, tmp_path):
    g = load_golden("waymo_reader.npz")
    ds = pkg.Real3DAug.tools.datasets
    f = _tree(str(tmp_path), g["lidar"], g["labels"], g["pose"])
    pcl, matrix, instances = ds.read_frame_waymo(f)
    assert pcl.dtype == np.float64 and np.array_equal(pcl, g["pcl"])
    assert np.array_equal(matrix, g["matrix"]) and np.array_equal(instances, g["instances"])
    assert not np.array_equal(pcl[:, :3], pcl[:, :3].astype(np.float32))       # genuine float64 after the offset


def _inserts(synth, i):
    kinds = ["pedestrian", "car"]
    return [[synth.make_insert(8800 + 10 * i + k, kind, rng_range=(5.0, 12.0))] for k, kind in enumerate(kinds)], [15, 15]


def _npy_bytes(arr):
    buf = io.BytesIO()
    np.save(buf, arr)
    return buf.getvalue()


def _expected_files(pkg, synth, lidar_file, i):
    """What the reference's driver would store for this frame: its reader, the oracle's chain, Waymo.save_data's casts
    (SS tools/datasets.py:287-301) stated in NumPy."""
    pcl, _, _ = pkg.Real3DAug.tools.datasets.read_frame_waymo(lidar_file)
    slots, need = _inserts(synth, i)
    merged, added, _ = O.augment_scene(pcl, slots, need)
    loc = np.array([1.22, 0, 2])
    m, a = merged.copy(), added.copy()
    m[:, 0:3] += loc
    a[:, 0:3] += loc
    pc, lab = O.remove_space_for_spherical(m)
    ap, al = O.remove_space_for_spherical(a)
    return _npy_bytes(pc.astype(np.float32)), _npy_bytes(lab.astype(np.uint32)), _npy_bytes(np.hstack((ap, al)).astype(np.float32))


def _dataset(synth, root, n):
    files = []
    rng = np.random.default_rng(5)
    for i in range(n):
        xyzi, label = synth.make_scene(880 + i, 16, 300)
        lidar = np.zeros((len(xyzi), 6), dtype=np.float32)
        lidar[:, :4] = xyzi
        lidar[:, :3] += np.array([1.22, 0, 2], dtype=np.float32)
        labels = np.stack([rng.integers(0, 99, len(xyzi)), np.where(label == 40, 18, 14)], axis=1).astype(np.int32)
        files.append(_tree(str(root), lidar, labels, None, name=f"{i:06d}"))
    return files


def _check(pkg, synth, files, out, folder):
    for i, f in enumerate(files):
        exp = _expected_files(pkg, synth, f, i)
        for sub, e in zip(("lidar", "labels_v3_2", "check"), exp):
            assert (out / folder / sub / f"{i:06d}.npy").read_bytes() == e, (i, sub)


def _oracle_process(scenes5, candidates, min_points):
    out, acc = [], []
    for s5, sl, nd in zip(scenes5, candidates, min_points):
        merged, added, a = O.augment_scene(s5, sl, nd)
        out.append((merged[:, [0, 1, 2, 6, 7]], added[:, [0, 1, 2, 6, 7]], None))
        acc.append(a)
    return out, acc


def test_waymo_pipeline_plumbing_with_injected_oracle(pkg, synth, tmp_path):
    files = _dataset(synth, tmp_path / "in", 3)
    fr = [pkg.Frame(f, None, f"{i:06d}") for i, f in enumerate(files)]
    pipe = pkg.AugmentPipeline(str(tmp_path / "out"), "w", dataset="waymo", batch_size=2)
    pipe.process = _oracle_process                          # (the test's stand-in for the device leg)
    st = pipe.run(fr, lambda i: _inserts(synth, i))
    assert st["written"] == 3
    _check(pkg, synth, files, tmp_path / "out", "w")
    assert pipe.run(fr, lambda i: _inserts(synth, i))["skipped_existing"] == 3


@pytest.mark.gpu
def test_waymo_pipeline_on_gpu(pkg, synth, tmp_path):
    files = _dataset(synth, tmp_path / "in", 5)
    fr = [pkg.Frame(f, None, f"{i:06d}") for i, f in enumerate(files)]
    pipe = pkg.AugmentPipeline(str(tmp_path / "out"), "w", dataset="waymo", batch_size=2)
    st = pipe.run(fr, lambda i: _inserts(synth, i))
    assert st["written"] == 5 and st["inserted"] > 0
    _check(pkg, synth, files, tmp_path / "out", "w")


@pytest.mark.gpu
def test_waymo_frames_with_batches_in_flight_and_sharded(pkg, synth, tmp_path):
    """The float64 flavour with several batches on the GPU at a time (``run(lanes=2)``: a worker thread, a HIP stream and a
    device batch per lane) and through the sharded file driver (``run_sharded_files(dataset="waymo")``, this rank's share):
    the three .npy files of every frame equal what the reference's ``Waymo.save_data`` would have written."""
    files = _dataset(synth, tmp_path / "in", 7)
    fr = [pkg.Frame(f, None, f"{i:06d}") for i, f in enumerate(files)]
    pipe = pkg.AugmentPipeline(str(tmp_path / "out"), "lanes", dataset="waymo", batch_size=2)
    st = pipe.run(fr, lambda i: _inserts(synth, i), lanes=2)
    assert st["written"] == 7 and st["inserted"] > 0
    _check(pkg, synth, files, tmp_path / "out", "lanes")
    one_each = lambda i: ([c[0] for c in _inserts(synth, i)[0]], _inserts(synth, i)[1])      # one placement per insert
    st = pkg.run_sharded_files(fr, one_each, str(tmp_path / "out"), "sharded", rank=0, world_size=1, device="cuda:0",
                               dataset="waymo", batch_size=3, lanes=2)
    assert st["written"] == 7
    _check(pkg, synth, files, tmp_path / "out", "sharded")


@pytest.mark.gpu
def test_begin_f64_is_within_three_times_of_begin(pkg, synth):
    """The float64 flavour's step 0 (guess from the float32 rounding, float64 confirmation on the exact coordinates)
    against the float32 one on the same 64 frames of 120k points."""
    import torch
    B = 32
    scenes = [synth.make_scene(60 + s) for s in range(B)]
    n = max(len(x) for x, _ in scenes)
    b32 = pkg.SceneBatch(B, n + 64, 64)
    b32.load(scenes)
    rows = [synth.scene5_from_packed(x, l) + np.array([1e-4, -2e-4, 3e-4, 0, 0]) for x, l in scenes]
    b64 = pkg.SceneBatch(B, n + 64, n + 64)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    import ctypes as C
    L = pkg._lib
    b64.begin_f64(rows)                                           # uploads once; then time the device call alone
    t64 = timed(lambda: L.check(b64.lib.r3d_batch_begin_f64(C.byref(b64.desc), C.c_void_p(b64._rows5.data_ptr()),
                                                             C.c_void_p(b64.n_points.data_ptr()), L.stream_ptr()), "begin_f64"))
    t32 = timed(b32.begin)
    b64.raise_on_status()
    # pixel ids equal the oracle's for one frame
    s9 = O.add_space_for_spherical(rows[0])
    s9, mx, mn = O.fill_spherical(s9)
    _, _, s9 = O.geometrical_front_view(s9, O.NUMROW, O.NUMCOLUMN, mx, mn)
    assert np.array_equal(b64.pixel_ids()[0, :len(rows[0])], s9[:, 8].astype(np.int32))
    print(f"begin_f64 {t64:.3f} ms, begin {t32:.3f} ms, ratio {t64 / t32:.2f}")
    assert t64 < 3.0 * t32 + 0.05
