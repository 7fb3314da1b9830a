"""CPU: the host half of the streamed driver (r3d_hostpack.cpp: r3d_host_pack_frames, r3d_host_merge_frames, r3d_host_write_delta_frames,
r3d_host_append_text_files) built with
g++ -fsanitize=address,undefined and driven through ctypes the way streaming.py does, edge cases included (empty
frames, counts at chunk boundaries, refused arguments).  GPU sanitizers are not available on the pool; this covers
the native host code of the path."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "pcl-augmentation_amd", "csrc")
HIP_INC = "/opt/rocm/include"

DRIVER = textwrap.dedent("""
    import ctypes as C, sys
    import numpy as np
    lib = C.CDLL(sys.argv[1])
    P = C.c_void_p
    lib.r3d_host_pack_frames.argtypes = [P, P, P, C.c_int32, C.c_int64, P, P, C.c_int32, C.c_int32]
    lib.r3d_host_merge_frames.argtypes = [P, P, C.c_int64, P, C.c_int64, P, P, C.c_int64, P, C.c_int32, P, P, C.c_int64, P, P,
                                          C.c_int64, C.c_int32, C.c_int32]
    lib.r3d_host_write_delta_frames.argtypes = [P, P, P, C.c_int32, P, P, C.c_int64, P, C.c_int64, P, P, C.c_int64, P, C.c_int32, P, C.c_int32]
    rng = np.random.default_rng(1)
    for B, cap, tail in ((1, 64, 1), (3, 130, 17), (7, 1000, 64)):
        chunks = (cap + 63) // 64
        n = rng.integers(0, cap - tail + 1, B).astype(np.int32)
        n[0] = 0 if B > 1 else cap - tail                       # an empty frame / a full one
        xs = [rng.random((k, 4), dtype=np.float32) for k in n]
        ls = [rng.integers(0, 1 << 32, k, dtype=np.uint64).astype(np.uint32) for k in n]
        px = (P * B)(*[x.ctypes.data for x in xs]); pl = (P * B)(*[l.ctypes.data for l in ls])
        sx = np.zeros((B, cap, 4), np.float32); sl = np.zeros((B, cap), np.uint32)
        for keep in (-1, 40):
            assert lib.r3d_host_pack_frames(px, pl, n.ctypes.data, B, cap, sx.ctypes.data, sl.ctypes.data, keep, 3) == 0
        n_tail = rng.integers(0, tail + 1, B).astype(np.int32)
        counts = np.stack([n, n + n_tail]).astype(np.int32)
        bits = rng.random((B, chunks * 64)) < 0.8
        for s in range(B):
            bits[s, counts[1, s]:] = False
        alive = np.packbits(bits.reshape(B, chunks, 64), axis=2, bitorder="little").view(np.uint64).reshape(B, chunks).copy()
        tx = rng.random((B, tail, 4), dtype=np.float32); tl = rng.integers(0, 99, (B, tail)).astype(np.uint32)
        ox = np.zeros((B, cap, 4), np.float32); ol = np.zeros((B, cap), np.uint32); no = np.zeros(B, np.int32)
        ck = np.zeros((B, tail, 5), np.float32)
        for cols, ckp in ((5, ck.ctypes.data), (4, ck.ctypes.data), (5, None)):
            rc = lib.r3d_host_merge_frames(sx.ctypes.data, sl.ctypes.data, cap, alive.ctypes.data, chunks, tx.ctypes.data, tl.ctypes.data,
                                           tail, counts.ctypes.data, B, ox.ctypes.data, ol.ctypes.data, cap, no.ctypes.data, ckp, tail, cols, 2)
            assert rc == 0
        assert [int(bits[s].sum()) for s in range(B)] == list(no)
        # refused, not executed: counts beyond the buffers, null pointers, a frame beyond the capacity
        bad = counts.copy(); bad[1, 0] = cap + 5
        assert lib.r3d_host_merge_frames(sx.ctypes.data, sl.ctypes.data, cap, alive.ctypes.data, chunks, tx.ctypes.data, tl.ctypes.data,
                                         tail, bad.ctypes.data, B, ox.ctypes.data, ol.ctypes.data, cap, no.ctypes.data, None, tail, 5, 2) < 0
        assert lib.r3d_host_merge_frames(None, sl.ctypes.data, cap, alive.ctypes.data, chunks, tx.ctypes.data, tl.ctypes.data,
                                         tail, counts.ctypes.data, B, ox.ctypes.data, ol.ctypes.data, cap, no.ctypes.data, None, tail, 5, 2) < 0
        big = n.copy(); big[-1] = cap + 1
        assert lib.r3d_host_pack_frames(px, pl, big.ctypes.data, B, cap, sx.ctypes.data, sl.ctypes.data, -1, 2) < 0
        # the delta writer (files straight from the staging slab + the delta, run by run): the bytes of merge + the counts above,
        # with garbage in the alive bits beyond n_total (it masks them itself)
        import os, tempfile
        dirty = bits.copy()
        for s in range(B):
            dirty[s, counts[1, s]:] = rng.random(chunks * 64 - counts[1, s]) < 0.5
        alive_d = np.packbits(dirty.reshape(B, chunks, 64), axis=2, bitorder="little").view(np.uint64).reshape(B, chunks).copy()
        with tempfile.TemporaryDirectory() as d:
            enc = lambda ext: (C.c_char_p * B)(*[os.path.join(d, f"{s}.{ext}").encode() for s in range(B)])
            no2 = np.zeros(B, np.int32)
            for cols in (5, 4):
                rc = lib.r3d_host_write_delta_frames(enc("bin"), enc("label"), enc("check"), B, sx.ctypes.data, sl.ctypes.data, cap,
                                                     alive_d.ctypes.data, chunks, tx.ctypes.data, tl.ctypes.data, tail, counts.ctypes.data,
                                                     cols, no2.ctypes.data, 3)
                assert rc == 0 and list(no2) == list(no)
                for s in range(B):
                    assert open(os.path.join(d, f"{s}.bin"), "rb").read() == ox[s, :no[s]].tobytes()
                    assert open(os.path.join(d, f"{s}.label"), "rb").read() == ol[s, :no[s]].tobytes()
                    assert os.path.getsize(os.path.join(d, f"{s}.check")) == int(n_tail[s]) * cols * 4
            assert lib.r3d_host_write_delta_frames(enc("bin"), None, None, B, sx.ctypes.data, sl.ctypes.data, cap, alive_d.ctypes.data, chunks,
                                                   tx.ctypes.data, tl.ctypes.data, tail, bad.ctypes.data, 5, None, 2) < 0
            # the label_2 writer: sources of 0 .. 9 000 bytes (several read buffers), extra lines or none, a skipped frame
            lib.r3d_host_append_text_files.argtypes = [P, P, P, C.c_int32, C.c_int32]
            lib.r3d_host_append_text_files.restype = C.c_int
            nfr = 9
            body = ["Car 0.00 0 -1.57 599.41 156.40 629.75 189.25 2.85 2.63 12.34 0.47 1.49 69.44 -1.56\\n" * (i * 13) for i in range(nfr)]
            for i in range(nfr):
                open(os.path.join(d, f"src{i}.txt"), "w").write(body[i])
            extra = [None if i == 2 else ("Pedestrian 0 0 0 0 0 0 0 1.7 0.6 0.6 1.5 1.0 2.0 0.1\\n" * (i % 3)).encode() for i in range(nfr)]
            srcs = (C.c_char_p * nfr)(*[os.path.join(d, f"src{i}.txt").encode() for i in range(nfr)])
            dsts = (C.c_char_p * nfr)(*[None if i == 4 else os.path.join(d, f"dst{i}.txt").encode() for i in range(nfr)])
            assert lib.r3d_host_append_text_files(srcs, dsts, (C.c_char_p * nfr)(*extra), nfr, 4) == 0
            for i in range(nfr):
                if i != 4:
                    assert open(os.path.join(d, f"dst{i}.txt"), "rb").read() == body[i].encode() + (extra[i] or b"")
            assert not os.path.exists(os.path.join(d, "dst4.txt"))
    print("asan driver ok")
""")


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan for this gcc")
    stub = tmp_path / "err.cpp"
    stub.write_text('#include <string>\nnamespace r3d { std::string &last_error_ref() { static thread_local std::string e; return e; } }\n')
    so = tmp_path / "libr3d_hostpack_asan.so"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", f"-I{HIP_INC}", os.path.join(CSRC, "r3d_hostpack.cpp"), str(stub), "-lpthread", "-o", str(so)]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("sanitizer build of r3d_hostpack.cpp failed here: " + b.stderr[-300:])
    drv = tmp_path / "drive.py"
    drv.write_text(DRIVER)
    env = {**os.environ, "LD_PRELOAD": libasan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"}
    r = subprocess.run([sys.executable, str(drv), str(so)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "asan driver ok" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-1500:]
