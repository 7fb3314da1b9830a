#!/usr/bin/env python3
"""Soak of the file-to-file legs (configs C3 / C4 through run_streamed / run_sharded_files on tmpfs, three lanes):

    [R3D_LIB=pcl-augmentation_amd/libreal3daug_hip_check.so] python tools/soak_files.py [repetitions] [frames]

Every repetition runs both shapes; a frame the device flags is reported with its status, rebases and the batch's counters."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pcl-augmentation_amd")
e = importlib.import_module("tools.e2e_pipeline")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
t0, n, bad = time.time(), 0, 0
for rep in range(reps):
    for shape in ("C4", "C3"):
        try:
            n += e.measure_files(pkg, shape, frames, 256, check=0)["frames"]
        except Exception as ex:
            bad += 1
            print(rep, shape, "ERR", repr(ex)[:600], flush=True)
print(f"{n} frames in {time.time() - t0:.1f} s, {bad} failing runs")
streaming = importlib.import_module("pcl-augmentation_amd.streaming")
if streaming._SUM_COUNTERS is not None:
    print("counters over all batches:", streaming._SUM_COUNTERS)
