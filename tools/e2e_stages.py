#!/usr/bin/env python3
"""Where the time of one streamed batch goes (tools/e2e_pipeline.py's `measure`, one lane, stage by stage):
the host packer, the upload, the kernels, the download of the delta, the host merge -- each timed alone, so the
stage that bounds the three-lane pipeline shows.

    python tools/e2e_stages.py [batch] [threads]
"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    pkg = importlib.import_module("pcl-augmentation_amd")
    streaming = importlib.import_module("pcl-augmentation_amd.streaming")
    _lib = pkg._lib
    import torch
    synth = pkg.synth
    kinds = synth.CONFIG_INSERTS["C2"]
    scenes = [synth.make_scene(s) for s in range(B)]
    inserts = [synth.make_inserts(s, kinds) for s in range(B)]
    need = [[20] * len(kinds)] * B
    n_max = max(len(x) for x, _ in scenes)
    grow = max(sum(len(i) for i in ins) for ins in inserts)
    srows = max(sum(len(ins[k]) for ins in inserts) for k in range(len(kinds))) * 2
    aug = streaming.StreamedAugmenter(B, n_max, grow, len(kinds), srows, lanes=1, pack_threads=threads, delta=True)
    ln = aug.lanes[0]
    lib = aug.lib
    for _ in range(2):
        aug.submit(0, scenes, inserts, need)
        aug.collect(0)

    def timed(fn, reps=5):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e3

    xs = [np.ascontiguousarray(x, dtype=np.float32) for x, _ in scenes]
    ls = [np.ascontiguousarray(l, dtype=np.uint32) for _, l in scenes]
    n = np.array([len(x) for x in xs], dtype=np.int32)
    px = (C.c_void_p * B)(*[x.ctypes.data for x in xs])
    pl = (C.c_void_p * B)(*[l.ctypes.data for l in ls])
    bt = ln.bt

    def pack():
        _lib.check(lib.r3d_host_pack_frames(px, pl, n.ctypes.data, B, bt.cap, ln.in_xyzi.data_ptr(), ln.in_label.data_ptr(),
                                            -1, threads), "pack")

    def upload(labels=True):
        def go():
            bt.xyzi.copy_(ln.in_xyzi, non_blocking=True)
            if labels:
                bt.label.copy_(ln.in_label, non_blocking=True)
            torch.cuda.synchronize()
        return go

    def whole():
        aug.submit(0, scenes, inserts, need)
        ln.done.synchronize()
        ln.busy = False

    def submit_only():
        t0 = time.perf_counter()
        aug.submit(0, scenes, inserts, need)
        dt = time.perf_counter() - t0
        ln.done.synchronize()
        ln.busy = False
        return dt

    def merge():
        cc = max(ln.check_cols, 4)
        _lib.check(lib.r3d_host_merge_frames(
            ln.in_xyzi.data_ptr(), ln.in_label.data_ptr(), bt.cap, ln.h_alive.data_ptr(), ln.chunks, ln.h_tail_xyzi.data_ptr(),
            ln.h_tail_label.data_ptr(), bt.log_cap, ln.h_dcounts.data_ptr(), B, ln.out_xyzi.data_ptr(),
            ln.out_label.data_ptr(), bt.cap, ln.h_n_out.data_ptr(), ln.out_check.data_ptr(), bt.log_cap, cc, threads), "merge")

    def kernels():
        bt.begin()
        torch.cuda.synchronize()

    mb = B * bt.cap * 20 / 1e6
    out = {"batch": B, "threads": threads, "MB_per_batch_xyzi_label": round(mb, 1)}
    out["pack_ms"] = round(timed(pack), 2)
    out["upload_xyzi_label_ms"] = round(timed(upload(True)), 2)
    out["upload_xyzi_ms"] = round(timed(upload(False)), 2)
    out["begin_ms"] = round(timed(kernels), 2)
    out["whole_lane_ms"] = round(timed(whole), 2)
    out["submit_call_ms"] = round(np.mean([submit_only() for _ in range(4)]) * 1e3, 2)
    out["merge_ms"] = round(timed(merge), 2)
    # the Python part of submit: the list comprehension over frames and the insert rows
    t0 = time.perf_counter()
    for _ in range(3):
        [np.ascontiguousarray(x, dtype=np.float32) for x, _ in scenes]
        [np.ascontiguousarray(l, dtype=np.uint32) for _, l in scenes]
    out["python_lists_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    out["frames_per_s_if_bound_by"] = {k: round(B / out[k] * 1e3) for k in ("pack_ms", "upload_xyzi_label_ms", "upload_xyzi_ms", "merge_ms", "submit_call_ms")}
    print(out)


if __name__ == "__main__":
    main()
