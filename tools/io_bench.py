"""Native file readers / writers of the streamed driver alone (r3d_host_read_frames / r3d_host_write_frames): frames/s and GB/s
for 256 frames of 120 000 points, pageable and pinned buffers, 16 and 64 threads.  tools/io_bench.py [directory]"""
import importlib, sys, os, time, ctypes as C, tempfile, shutil
import numpy as np
sys.path.insert(0, ".")
pkg = importlib.import_module("pcl-augmentation_amd")
import torch
lib = pkg._lib.load()
B, cap = 256, 124600
root = tempfile.mkdtemp(dir=sys.argv[1] if len(sys.argv) > 1 else None)
for s in range(B):
    x, l = pkg.synth.make_scene(s % 16)
    x.tofile(f"{root}/{s}.bin"); l.tofile(f"{root}/{s}.label")
enc = lambda names: (C.c_char_p * B)(*[f"{root}/{n}".encode() for n in names])
for pinned in (False, True):
    dx = torch.empty((B, cap, 4), dtype=torch.float32); dl = torch.empty((B, cap), dtype=torch.int32); dn = torch.zeros(B, dtype=torch.int32)
    if pinned:
        dx, dl = dx.pin_memory(), dl.pin_memory()
    for th in (16, 64):
        for rep in range(2):
            t = time.perf_counter()
            rc = lib.r3d_host_read_frames(enc([f"{s}.bin" for s in range(B)]), enc([f"{s}.label" for s in range(B)]), B, cap, dx.data_ptr(), dl.data_ptr(), dn.data_ptr(), -1, th)
            dt = time.perf_counter() - t
        print(f"read  pinned={pinned} threads={th}: {B/dt:8.0f} frames/s  {B*120000*20/dt/1e9:5.1f} GB/s", flush=True)
        ck = torch.zeros((B, 16, 5)); nck = torch.full((B,), 16, dtype=torch.int32)
        for rep in range(2):
            t = time.perf_counter()
            rc = lib.r3d_host_write_frames(enc([f"o{s}.bin" for s in range(B)]), enc([f"o{s}.label" for s in range(B)]), enc([f"c{s}.bin" for s in range(B)]), B, dx.data_ptr(), dl.data_ptr(), cap, dn.data_ptr(), ck.data_ptr(), 16, 5, nck.data_ptr(), th)
            dt = time.perf_counter() - t
        print(f"write pinned={pinned} threads={th}: {B/dt:8.0f} frames/s  {B*120000*20/dt/1e9:5.1f} GB/s", flush=True)
shutil.rmtree(root)
