"""Per-kernel totals of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir or kernel_stats.csv> [top]"""
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':64s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'%':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    name = re.sub(r"\(anonymous namespace\)::|r3d::|void ", "", r["Name"])
    name = re.sub(r"\(.*", "", name) if not name.startswith("at::") else name[:64]
    print(f"{name[:64]:64s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['AverageNs']) / 1e3:9.1f} {100 * float(r['TotalDurationNs']) / tot:6.2f}")
