"""CPU: r3d_host_merge_frames (the host half of the delta path of the streamed driver) against NumPy: merged cloud =
surviving frame points in order, then surviving inserted points; check rows = every inserted point."""
import ctypes as C

import numpy as np
import pytest


def _merge_numpy(x, l, n_head, alive_bits, tx, tl, n_total):
    keep = alive_bits[:n_total]
    allx = np.vstack([x[:n_head], tx[:n_total - n_head]])
    alll = np.concatenate([l[:n_head], tl[:n_total - n_head]])
    return allx[keep], alll[keep]


@pytest.mark.parametrize("check_cols", [5, 4, 0])
def test_merge_equals_numpy(pkg, check_cols):
    lib = pkg._lib.load()
    rng = np.random.default_rng(3)
    B, cap, tail_stride = 5, 1000, 96
    chunks = (cap + 63) // 64
    in_x = rng.random((B, cap, 4), dtype=np.float32)
    in_l = rng.integers(0, 60000, (B, cap)).astype(np.uint32)
    tail_x = rng.random((B, tail_stride, 4), dtype=np.float32)
    tail_l = rng.integers(0, 80000, (B, tail_stride)).astype(np.uint32)
    n_head = np.array([900, 64, 0, 129, 640], dtype=np.int32)
    n_tail = np.array([96, 0, 50, 7, 64], dtype=np.int32)
    counts = np.stack([n_head, n_head + n_tail]).astype(np.int32)
    bits = rng.random((B, chunks * 64)) < 0.9
    bits[3, :128] = True                                        # whole chunks alive: the memcpy route
    for s in range(B):
        bits[s, counts[1, s]:] = False
    alive = np.packbits(bits.reshape(B, chunks, 64), axis=2, bitorder="little").view(np.uint64).reshape(B, chunks)
    out_x = np.zeros((B, cap, 4), dtype=np.float32)
    out_l = np.zeros((B, cap), dtype=np.uint32)
    n_out = np.zeros(B, dtype=np.int32)
    check = np.zeros((B, tail_stride, max(check_cols, 4)), dtype=np.float32)
    P = lambda a: a.ctypes.data
    rc = lib.r3d_host_merge_frames(P(in_x), P(in_l), cap, P(alive), chunks, P(tail_x), P(tail_l), tail_stride, P(counts), B,
                                   P(out_x), P(out_l), cap, P(n_out), P(check) if check_cols else None, tail_stride,
                                   max(check_cols, 4), 3)
    assert rc == 0, lib.r3d_last_error()
    for s in range(B):
        ex, el = _merge_numpy(in_x[s], in_l[s], n_head[s], bits[s], tail_x[s], tail_l[s], counts[1, s])
        assert n_out[s] == len(ex)
        assert np.array_equal(out_x[s, :n_out[s]], ex) and np.array_equal(out_l[s, :n_out[s]], el)
        if check_cols:
            exp = tail_x[s, :n_tail[s]] if check_cols == 4 else np.hstack([tail_x[s, :n_tail[s]], tail_l[s, :n_tail[s], None].astype(np.float32)])
            assert np.array_equal(check[s, :n_tail[s], :check_cols], exp)


def test_merge_rejects_bad_counts(pkg):
    lib = pkg._lib.load()
    z = np.zeros(64, dtype=np.float32)
    counts = np.array([[10], [200]], dtype=np.int32)            # 190 inserted points, stride 4
    rc = lib.r3d_host_merge_frames(z.ctypes.data, z.ctypes.data, 256, z.ctypes.data, 4, z.ctypes.data, z.ctypes.data, 4,
                                   counts.ctypes.data, 1, z.ctypes.data, z.ctypes.data, 256, z.ctypes.data, None, 4, 5, 1)
    assert rc == pkg._lib.E_ARG if hasattr(pkg._lib, "E_ARG") else rc < 0


def test_native_file_readers_and_writers(pkg, tmp_path):
    """r3d_host_read_frames / r3d_host_write_frames against NumPy's fromfile / tofile: same arrays in, same bytes out,
    labels masked or collapsed like the packer's, a missing file named in the error."""
    lib = pkg._lib.load()
    rng = np.random.default_rng(8)
    B, cap = 4, 500
    ns = [500, 0, 37, 256]
    vel, lab = [], []
    for s, n in enumerate(ns):
        x = rng.random((n, 4), dtype=np.float32)
        l = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        x.tofile(tmp_path / f"{s}.bin")
        l.tofile(tmp_path / f"{s}.label")
        vel.append(x)
        lab.append(l)
    enc = lambda names: (C.c_char_p * B)(*[None if n is None else str(tmp_path / n).encode() for n in names])
    for keep in (-1, 40):
        dx = np.full((B, cap, 4), -1, np.float32)
        dl = np.full((B, cap), 7, np.uint32)
        dn = np.zeros(B, np.int32)
        rc = lib.r3d_host_read_frames(enc([f"{s}.bin" for s in range(B)]), enc([f"{s}.label" for s in range(B)]), B, cap,
                                      dx.ctypes.data, dl.ctypes.data, dn.ctypes.data, keep, 3)
        assert rc == 0, lib.r3d_last_error()
        assert list(dn) == ns
        for s, n in enumerate(ns):
            assert np.array_equal(dx[s, :n], vel[s])
            exp = lab[s] & 0xFFFF if keep < 0 else np.where((lab[s] & 0xFFFF) == keep, keep, 1)
            assert np.array_equal(dl[s, :n], exp.astype(np.uint32))
    rc = lib.r3d_host_read_frames(enc(["0.bin", "nope.bin", "2.bin", "3.bin"]), None, B, cap, dx.ctypes.data, dl.ctypes.data,
                                  dn.ctypes.data, -1, 2)
    assert rc < 0 and b"nope.bin" in lib.r3d_last_error()
    rc = lib.r3d_host_read_frames(enc([f"{s}.bin" for s in range(B)]), None, B, 100, dx.ctypes.data, dl.ctypes.data,
                                  dn.ctypes.data, -1, 2)
    assert rc < 0                                               # 500 points do not fit a capacity of 100
    # writing: frame 1 is skipped (NULL path), no label file for frame 2
    n_out = np.array([400, 5, 37, 0], np.int32)
    ck = rng.random((B, 16, 5), dtype=np.float32)
    n_ck = np.array([16, 0, 3, 0], np.int32)
    rc = lib.r3d_host_write_frames(enc(["o0.bin", None, "o2.bin", "o3.bin"]), enc(["o0.label", None, None, "o3.label"]),
                                   enc(["c0.bin", None, "c2.bin", "c3.bin"]), B, dx.ctypes.data, dl.ctypes.data, cap,
                                   n_out.ctypes.data, ck.ctypes.data, 16, 5, n_ck.ctypes.data, 3)
    assert rc == 0, lib.r3d_last_error()
    assert (tmp_path / "o0.bin").read_bytes() == dx[0, :400].tobytes() and (tmp_path / "o0.label").read_bytes() == dl[0, :400].tobytes()
    assert (tmp_path / "c0.bin").read_bytes() == ck[0, :16].tobytes() and (tmp_path / "c2.bin").read_bytes() == ck[2, :3].tobytes()
    assert (tmp_path / "o3.bin").read_bytes() == b"" and not (tmp_path / "o2.label").exists() and not (tmp_path / "o1.bin").exists()
    assert not list(tmp_path.glob("*.tmp"))


@pytest.mark.parametrize("check_cols", [5, 4])
def test_delta_writer_writes_what_merge_and_write_do(pkg, tmp_path, check_cols):
    """r3d_host_write_delta_frames (files straight from the staging slab + the delta, run by run of surviving points)
    against r3d_host_merge_frames followed by r3d_host_write_frames: the same bytes in every file, for alive patterns that
    put runs across word boundaries, whole words, single points, empty frames and frames that lose everything."""
    lib = pkg._lib.load()
    rng = np.random.default_rng(11)
    B, cap, tail_stride = 9, 1500, 200
    chunks = (cap + 63) // 64
    in_x = rng.random((B, cap, 4), dtype=np.float32)
    in_l = rng.integers(0, 60000, (B, cap)).astype(np.uint32)
    tail_x = rng.random((B, tail_stride, 4), dtype=np.float32)
    tail_l = rng.integers(0, 80000, (B, tail_stride)).astype(np.uint32)
    n_head = np.array([1300, 64, 0, 129, 640, 1000, 1, 777, 1200], dtype=np.int32)
    n_tail = np.array([200, 0, 50, 7, 64, 0, 0, 199, 13], dtype=np.int32)
    counts = np.stack([n_head, n_head + n_tail]).astype(np.int32)
    bits = rng.random((B, chunks * 64)) < 0.97
    bits[3, :] = True                                            # nobody dies: one run
    bits[4, :] = rng.random(chunks * 64) < 0.5                   # runs of a point or two
    bits[5, :] = False                                           # nobody lives: an empty file
    bits[5, 63:65] = True                                        # ... but a run across a word boundary
    bits[7, :] = True
    bits[7, 64:128] = False                                      # a whole word dead
    bits[8, 1199] = False                                        # the frame's last point dead, the first inserted alive
    for s in range(B):
        bits[s, counts[1, s]:] = rng.random(chunks * 64 - counts[1, s]) < 0.5     # garbage beyond n_total must not matter
    alive = np.packbits(bits.reshape(B, chunks, 64), axis=2, bitorder="little").view(np.uint64).reshape(B, chunks)
    clean = bits.copy()
    for s in range(B):
        clean[s, counts[1, s]:] = False
    alive_clean = np.packbits(clean.reshape(B, chunks, 64), axis=2, bitorder="little").view(np.uint64).reshape(B, chunks)
    P = lambda a: a.ctypes.data
    enc = lambda sub, ext, skip=(): (C.c_char_p * B)(*[None if s in skip else str(tmp_path / sub / f"{s}.{ext}").encode() for s in range(B)])
    for sub in ("a", "b"):
        (tmp_path / sub).mkdir()
    # the two-step way (it wants clean bits beyond n_total: r3d_batch_export_delta delivers them that way)
    out_x, out_l, n_out = np.zeros((B, cap, 4), np.float32), np.zeros((B, cap), np.uint32), np.zeros(B, np.int32)
    check = np.zeros((B, tail_stride, check_cols), np.float32)
    assert lib.r3d_host_merge_frames(P(in_x), P(in_l), cap, P(alive_clean), chunks, P(tail_x), P(tail_l), tail_stride, P(counts), B,
                                     P(out_x), P(out_l), cap, P(n_out), P(check), tail_stride, check_cols, 3) == 0
    n_ck = n_tail.astype(np.int32)
    assert lib.r3d_host_write_frames(enc("a", "bin", (2,)), enc("a", "label", (2, 6)), enc("a", "check", (2,)), B, P(out_x), P(out_l), cap,
                                     P(n_out), P(check), tail_stride, check_cols, P(n_ck), 3) == 0
    n_out2 = np.zeros(B, np.int32)
    rc = lib.r3d_host_write_delta_frames(enc("b", "bin", (2,)), enc("b", "label", (2, 6)), enc("b", "check", (2,)), B, P(in_x), P(in_l), cap,
                                         P(alive), chunks, P(tail_x), P(tail_l), tail_stride, P(counts), check_cols, P(n_out2), 4)
    assert rc == 0, lib.r3d_last_error()
    assert np.array_equal(n_out, n_out2)
    names = sorted(p.name for p in (tmp_path / "a").iterdir())
    assert names == sorted(p.name for p in (tmp_path / "b").iterdir()) and len(names) == 3 * 8 - 1 and not any(n.endswith(".tmp") for n in names)
    for n in names:
        assert (tmp_path / "a" / n).read_bytes() == (tmp_path / "b" / n).read_bytes(), n
    assert (tmp_path / "b" / "5.bin").stat().st_size == 2 * 16 and (tmp_path / "b" / "3.bin").stat().st_size == 136 * 16
    # refused: counts beyond the buffers; a path that cannot be written is named
    bad = counts.copy()
    bad[1, 0] = cap + 1
    assert lib.r3d_host_write_delta_frames(enc("b", "bin"), None, None, B, P(in_x), P(in_l), cap, P(alive), chunks, P(tail_x), P(tail_l),
                                           tail_stride, P(bad), check_cols, None, 2) < 0
    nowhere = (C.c_char_p * B)(*[str(tmp_path / "no" / "dir" / f"{s}.bin").encode() for s in range(B)])
    assert lib.r3d_host_write_delta_frames(nowhere, None, None, B, P(in_x), P(in_l), cap, P(alive), chunks, P(tail_x), P(tail_l),
                                           tail_stride, P(counts), check_cols, None, 2) < 0
    assert b"no/dir" in lib.r3d_last_error()


def test_native_label_2_writer_equals_create_annotation(pkg, tmp_path):
    """r3d_host_append_text_files = OD tools/datasets.py:20-37 for a batch: the frame's label_2 file followed by the lines of
    the inserted objects (the mirror's create_annotation is the reference for the bytes)."""
    import os
    lib = pkg._lib.load()
    from importlib import import_module
    ds = import_module("pcl-augmentation_amd.Real3DAug.tools.datasets")
    n = 37
    srcs, lines = [], []
    for i in range(n):
        p = tmp_path / f"src_{i}.txt"
        # (frames 11 / 12 / 13: CRLF, lone CR and mixed line ends -- the reference opens the file in text mode, where universal
        # newlines turn all of them into "\n"; the mirror's create_annotation does the same, so must the native writer)
        eol = {11: "\r\n", 12: "\r"}.get(i, "\n")
        body = "".join(f"Car 0.00 0 -1.57 {j}.0 156.40 629.75 189.25 2.85 2.63 12.34 0.47 1.49 69.44 -1.56{eol if (i != 13 or j % 2) else chr(10)}"
                       for j in range(max(i % 5, 3 if i in (11, 12, 13) else 0)))
        p.write_bytes(body.encode())
        srcs.append(str(p))
        lines.append([f"Pedestrian 0 0 0 0 0 0 0 1.7 0.6 0.6 {i}.5 1.0 2.0 0.1\n"] * (i % 3))
    (tmp_path / "a").mkdir(), (tmp_path / "b").mkdir()
    for i in range(n):
        ds.create_annotation(srcs[i], str(tmp_path / "a" / f"{i}.txt"), lines[i])
    dst = [None if i == 5 else str(tmp_path / "b" / f"{i}.txt") for i in range(n)]
    enc = lambda xs: (C.c_char_p * n)(*[None if x is None else os.fsencode(x) for x in xs])
    extra = (C.c_char_p * n)(*[None if i == 7 else "".join(lines[i]).encode() for i in range(n)])
    pkg._lib.check(lib.r3d_host_append_text_files(enc(srcs), enc(dst), extra, n, 4), "append")
    for i in range(n):
        if i == 5:
            assert not (tmp_path / "b" / "5.txt").exists()
            continue
        want = (tmp_path / "a" / f"{i}.txt").read_bytes() if i != 7 else open(srcs[i], "rb").read()
        assert (tmp_path / "b" / f"{i}.txt").read_bytes() == want
        if i in (11, 12, 13):
            assert b"\r" not in want and want.count(b"\n") == 3 + len(lines[i])
    assert not list((tmp_path / "b").glob("*.tmp"))
    # a source that does not exist: an I/O error that names the file, not a silent skip
    bad = list(srcs)
    bad[3] = str(tmp_path / "nothing.txt")
    assert lib.r3d_host_append_text_files(enc(bad), enc(dst), extra, n, 4) == pkg._lib.E_IO
    assert b"nothing.txt" in lib.r3d_last_error()
    # a destination that cannot be written (its directory does not exist): R3D_E_IO, the message names THAT file, no .tmp stays
    gone = list(dst)
    gone[2] = str(tmp_path / "no_such_dir" / "2.txt")
    assert lib.r3d_host_append_text_files(enc(srcs), enc(gone), extra, n, 1) == pkg._lib.E_IO
    assert b"no_such_dir" in lib.r3d_last_error() and not list(tmp_path.glob("**/*.tmp"))


def test_xyz_packers_equal_numpy(pkg, tmp_path):
    """r3d_host_pack_frames_xyz / r3d_host_read_frames_xyz: the slabs of r3d_host_pack_frames / r3d_host_read_frames and, beside
    them, x y z of every point as rows of 12 bytes (what r3d_batch_begin_xyz takes); ragged frames, odd counts, no xyz3 wanted."""
    import os
    lib = pkg._lib.load()
    rng = np.random.default_rng(9)
    B, cap = 5, 1024
    ns = [1000, 1, 0, 777, 1024]
    xs = [rng.random((n, 4), dtype=np.float32) for n in ns]
    ls = [rng.integers(0, 1 << 20, n).astype(np.uint32) for n in ns]
    n = np.array(ns, dtype=np.int32)
    px = (C.c_void_p * B)(*[x.ctypes.data for x in xs])
    pl = (C.c_void_p * B)(*[l.ctypes.data for l in ls])
    for want3 in (True, False):
        dx, dl = np.full((B, cap, 4), -1, dtype=np.float32), np.zeros((B, cap), dtype=np.uint32)
        d3 = np.full((B, cap, 3), -7, dtype=np.float32)
        rc = lib.r3d_host_pack_frames_xyz(px, pl, n.ctypes.data, B, cap, dx.ctypes.data, dl.ctypes.data, d3.ctypes.data if want3 else None, -1, 3)
        assert rc == 0, lib.r3d_last_error()
        for s in range(B):
            assert np.array_equal(dx[s, :ns[s]], xs[s]) and np.array_equal(dl[s, :ns[s]], ls[s] & 0xFFFF)
            if want3:
                assert np.array_equal(d3[s, :ns[s]], xs[s][:, :3]) and np.all(d3[s, ns[s]:] == -7)
        if not want3:
            assert np.all(d3 == -7)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        for s in range(B):
            xs[s].tofile(f"{s}.bin")
            ls[s].tofile(f"{s}.label")
        enc = lambda names: (C.c_char_p * len(names))(*[x.encode() for x in names])
        dx, dl, d3 = np.zeros((B, cap, 4), dtype=np.float32), np.zeros((B, cap), dtype=np.uint32), np.zeros((B, cap, 3), dtype=np.float32)
        got = np.zeros(B, dtype=np.int32)
        rc = lib.r3d_host_read_frames_xyz(enc([f"{s}.bin" for s in range(B)]), enc([f"{s}.label" for s in range(B)]), B, cap,
                                          dx.ctypes.data, dl.ctypes.data, d3.ctypes.data, got.ctypes.data, -1, 2)
        assert rc == 0, lib.r3d_last_error()
        assert list(got) == ns
        for s in range(B):
            assert np.array_equal(dx[s, :ns[s]], xs[s]) and np.array_equal(d3[s, :ns[s]], xs[s][:, :3]) and np.array_equal(dl[s, :ns[s]], ls[s] & 0xFFFF)
    finally:
        os.chdir(cwd)
