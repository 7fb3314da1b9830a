"""File-to-file driver (row f-2): reading, batching, resume and the written bytes.

The CPU test injects the oracle as the processing function (the HIP path needs a GPU); the gpu
test runs the real thing.  Both compare every written file with the oracle's bytes."""
import os

import numpy as np
import pytest

from oracle import real3d_oracle as O


def _make_dataset(synth, root, n, od=False):
    frames = []
    for i in range(n):
        xyzi, label = synth.make_scene(700 + i, 24, 400)
        os.makedirs(root / "velodyne", exist_ok=True)
        os.makedirs(root / "labels", exist_ok=True)
        xyzi.tofile(root / "velodyne" / f"{i:06d}.bin")
        (label | (np.uint32(i + 1) << 16)).astype(np.uint32).tofile(root / "labels" / f"{i:06d}.label")   # instance bits set
        frames.append((str(root / "velodyne" / f"{i:06d}.bin"), str(root / "labels" / f"{i:06d}.label")))
    return frames


def _candidates(synth, i):
    slots = [[synth.make_insert(7000 + 10 * i + k, kind, rng_range=(5.0, 12.0))] for k, kind in enumerate(["pedestrian", "car"])]
    return slots, [15, 15]


def _oracle_process(check_cols):
    def process(scenes, candidates, min_points):
        out, acc = [], []
        for (xyzi, label), sl, nd in zip(scenes, candidates, min_points):
            s5 = np.hstack((xyzi.astype(np.float64), label.astype(np.float64)[:, None]))
            merged, allvis, a = O.augment_scene(s5, sl, nd)
            pc, lab = O.remove_space_for_spherical(merged)
            ap, al = O.remove_space_for_spherical(allvis)
            chk = np.hstack((ap, al)) if check_cols == 5 else ap
            out.append((pc.astype(np.float32), lab.astype(np.uint32).reshape(-1), chk.astype(np.float32)))
            acc.append(a)
        return out, acc
    return process


def _expected(synth, frames, i, od):
    xyzi = np.fromfile(frames[i][0], dtype=np.float32).reshape(-1, 4)
    label = np.fromfile(frames[i][1], dtype=np.uint32) & 0xFFFF
    if od:
        label = np.where(label == 40, 40, 1).astype(np.uint32)
    s5 = np.hstack((xyzi.astype(np.float64), label.astype(np.float64)[:, None]))
    sl, nd = _candidates(synth, i)
    merged, allvis, _ = O.augment_scene(s5, sl, nd)
    return O.save_bytes_kitti(merged, allvis) if od else O.save_bytes_semantic(merged, allvis)


def _check_outputs(synth, frames, out_root, folder, od):
    for i in range(len(frames)):
        exp = _expected(synth, frames, i, od)
        name = f"{i:06d}"
        assert (out_root / folder / "velodyne" / f"{name}.bin").read_bytes() == exp[0]
        if od:
            assert not (out_root / folder / "labels").exists()
            assert (out_root / folder / "check" / f"{name}.bin").read_bytes() == exp[1]
        else:
            assert (out_root / folder / "labels" / f"{name}.label").read_bytes() == exp[1]
            assert (out_root / folder / "check" / f"{name}.bin").read_bytes() == exp[2]


@pytest.mark.parametrize("od", [False, True])
def test_pipeline_plumbing_with_injected_oracle(pkg, synth, tmp_path, od):
    frames = _make_dataset(synth, tmp_path / "in", 5)
    fr = [pkg.Frame(v, l) for v, l in frames]
    pipe = pkg.AugmentPipeline(str(tmp_path / "out"), "run0", dataset="kitti" if od else "semantic", batch_size=2)
    pipe.process = _oracle_process(4 if od else 5)          # (the test's stand-in for the device leg: the pipeline has no such parameter)
    stats = pipe.run(fr, lambda i: _candidates(synth, i))
    assert stats["written"] == 5 and stats["skipped_existing"] == 0
    _check_outputs(synth, frames, tmp_path / "out", "run0", od)
    # resume: nothing left to do
    stats = pipe.run(fr, lambda i: _candidates(synth, i))
    assert stats["written"] == 0 and stats["skipped_existing"] == 5
    # a reader error surfaces
    bad = fr + [pkg.Frame(str(tmp_path / "missing.bin"), str(tmp_path / "missing.label"))]
    with pytest.raises(Exception):
        pipe.run(bad, lambda i: _candidates(synth, i))


@pytest.mark.gpu
@pytest.mark.parametrize("od", [False, True])
def test_pipeline_on_gpu(pkg, synth, tmp_path, od):
    frames = _make_dataset(synth, tmp_path / "in", 7)
    fr = [pkg.Frame(v, l) for v, l in frames]
    pipe = pkg.AugmentPipeline(str(tmp_path / "out"), "gpu", dataset="kitti" if od else "semantic", batch_size=3)
    stats = pipe.run(fr, lambda i: _candidates(synth, i))
    assert stats["written"] == 7 and stats["inserted"] > 0
    _check_outputs(synth, frames, tmp_path / "out", "gpu", od)


@pytest.mark.gpu
def test_streamed_pipeline_on_gpu(pkg, synth, tmp_path):
    """run_streamed: pinned lanes, native packer, overlapped transfers -- every written file equals the
    oracle's bytes (SemanticKITTI and KITTI flavour, a last batch that is not full, label_2 files)."""
    frames = _make_dataset(synth, tmp_path / "in", 7)
    for od in (False, True):
        pipe = pkg.AugmentPipeline(str(tmp_path / "out"), "od" if od else "ss", dataset="kitti" if od else "semantic", batch_size=3)
        fr = [pkg.Frame(v, l) for v, l in frames]
        (tmp_path / "in" / "label_2").mkdir(exist_ok=True)
        for f in fr:
            (tmp_path / "in" / "label_2" / f"{f.name}.txt").write_text("Car 0 0 0 1 2 3 4 1.5 1.6 3.9 1 2 30 0.1\n")

        def inserts_for(i):
            slots, need = _candidates(synth, i)
            return [s[0] for s in slots], need

        def label_2_for(i, acc):
            return str(tmp_path / "in" / "label_2" / f"{fr[i].name}.txt"), [f"Pedestrian 0 3 0 0 0 0 0 1 1 1 0 0 {k} 0\n" for k, a in enumerate(acc) if a >= 0]

        st = pipe.run_streamed(fr, inserts_for, lanes=2, label_2_for=label_2_for if od else None)
        assert st["written"] == 7
        _check_outputs(synth, frames, tmp_path / "out", "od" if od else "ss", od)
        # once more with whole clouds downloaded instead of the delta (delta=False), and frames without any insert
        import shutil
        shutil.rmtree(tmp_path / "out" / ("od" if od else "ss"))
        st = pipe.run_streamed(fr, inserts_for, lanes=2, label_2_for=label_2_for if od else None, delta=False, io_threads=3)
        assert st["written"] == 7
        _check_outputs(synth, frames, tmp_path / "out", "od" if od else "ss", od)
        if od:
            txt = (tmp_path / "out" / "od" / "label_2" / "000003.txt").read_text()
            assert txt.startswith("Car 0 0 0") and txt.count("\n") >= 2


@pytest.mark.gpu
def test_streamed_pipeline_without_inserts_and_growing_shapes(pkg, synth, tmp_path):
    """Batches whose frames have no insert at all (K = 0) come back unchanged; a later batch with more slots and
    larger frames makes the driver build larger lanes and carry on."""
    frames = _make_dataset(synth, tmp_path / "in", 6)
    fr = [pkg.Frame(v, l) for v, l in frames]

    def inserts_for(i):
        if i < 3:
            return [], []
        slots, need = _candidates(synth, i)
        return [s[0] for s in slots], need

    pipe = pkg.AugmentPipeline(str(tmp_path / "out"), "k0", batch_size=3)
    st = pipe.run_streamed(fr, inserts_for, lanes=2)
    assert st["written"] == 6
    for i in range(3):
        assert (tmp_path / "out" / "k0" / "velodyne" / f"{i:06d}.bin").read_bytes() == open(frames[i][0], "rb").read()
        assert (tmp_path / "out" / "k0" / "check" / f"{i:06d}.bin").read_bytes() == b""
    for i in range(3, 6):
        exp = _expected(synth, frames, i, False)
        assert (tmp_path / "out" / "k0" / "velodyne" / f"{i:06d}.bin").read_bytes() == exp[0]
        assert (tmp_path / "out" / "k0" / "check" / f"{i:06d}.bin").read_bytes() == exp[2]
