#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one profiling session into the files kept under profiles/.

On the GPU box (see profiles/README.md for the three rocprofv3 commands):

    python tools/make_profiles.py <stats_dir> <fetch_dir> <write_dir> <round-tag, e.g. r01>

stats_dir holds *kernel_stats.csv of the --kernel-trace --stats run; fetch_dir / write_dir hold the
*counter_collection.csv of the two --pmc passes (FETCH_SIZE, WRITE_SIZE).  Writes
profiles/<tag>_kernel_stats.csv (our kernels only), profiles/<tag>_pmc_fetch_write.csv and
profiles/<tag>_pmc.json (read by bench.py for roofline.traffic).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    if not hits:
        raise SystemExit(f"nothing matches {pattern}")
    return hits[0]


def short(name):
    base = name.split("(")[0].replace("r3d::", "").replace("(anonymous namespace)::", "").replace("void ", "").strip()
    return base.split("<")[0]


def counter_avgs(directory, counter):
    vals = defaultdict(list)
    with open(one(os.path.join(directory, "**", "*counter_collection.csv"))) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") == counter and "k_" in row["Kernel_Name"]:
                vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return vals


def main():
    stats_dir, fetch_dir, write_dir, tag = sys.argv[1:5]
    out = os.path.join(ROOT, "profiles")
    with open(one(os.path.join(stats_dir, "**", "*kernel_stats.csv"))) as fh:
        rows = list(csv.DictReader(fh))
    ours = [r for r in rows if "r3d::" in r["Name"] or "k_place" in r["Name"]]
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(ours)
    fetch, write = counter_avgs(fetch_dir, "FETCH_SIZE"), counter_avgs(write_dir, "WRITE_SIZE")
    kernels = {}
    with open(os.path.join(out, f"{tag}_pmc_fetch_write.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg", "FETCH_SIZE_KB_max", "WRITE_SIZE_KB_avg", "WRITE_SIZE_KB_max",
                    "hbm_bytes_per_launch_corrected"])
        for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
            f, wr = fetch[k], write.get(k, [0.0])
            fa, wa = sum(f) / len(f), sum(wr) / len(wr)
            corrected = (2 * fa + wa) * 1024
            kernels[k] = {"fetch_kb": fa, "write_kb": wa, "hbm_bytes_corrected": corrected}
            w.writerow([k, len(f), round(fa), round(max(f)), round(wa), round(max(wr)), round(corrected)])
    with open(os.path.join(out, f"{tag}_pmc.json"), "w") as fh:
        json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py "
                              "--steps 2 --warmup 1 --no-cpu-baseline",
                   "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 64 B per 128-B "
                                 "request of a wide coalesced read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact "
                                 "for streaming stores",
                   "kernels": kernels}, fh, indent=1)
    print("wrote", len(ours), "kernel rows,", len(kernels), "kernels with counters")


if __name__ == "__main__":
    main()
