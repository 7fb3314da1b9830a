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
