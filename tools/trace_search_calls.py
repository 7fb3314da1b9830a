"""Per-call durations of the placement kernels of one search call, from a rocprofv3 --kernel-trace CSV:
python tools/trace_search_calls.py <dir of the trace>"""
import csv
import glob
import os
import re
import sys

files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(max(files, key=os.path.getsize))))     # (the process with the kernels, not a helper's)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(re.search(r"k_[a-z_]+", r["Kernel_Name"]).group(0), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
        int(r["Start_Timestamp"])) for r in rows if "k_place" in r["Kernel_Name"] or "k_insert_first" in r["Kernel_Name"]]
start = len(seq) // 2
while start < len(seq) and "centres" not in seq[start][0]:
    start += 1
t0 = seq[start][2]
for n, d, t in seq[start:start + 14]:
    print(f"{n:30s} starts {1e-3 * (t - t0):8.1f} us, runs {d:8.1f} us")
