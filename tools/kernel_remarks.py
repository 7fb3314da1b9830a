#!/usr/bin/env python3
"""Register / spill / scratch table of a HIP file's kernels from the compiler's own remarks:

    python tools/kernel_remarks.py pcl-augmentation_amd/csrc/r3d_insert.hip [name filter] [-- extra hipcc flags]

Compiles with the flags of csrc/Makefile plus -Rpass-analysis=kernel-resource-usage (object to /tmp) and prints one row
per kernel: SGPRs, VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD), SGPR / VGPR spills, static LDS."""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[:args.index("--")]
src = args[0]
flt = args[1] if len(args) > 1 else ""
flags = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-function", "-Wno-pass-failed"]
if src.endswith("r3d_insert.hip"):
    flags += ["-mllvm", "-disable-machine-licm"]
p = subprocess.run(["hipcc"] + flags + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_remarks.o"],
                   capture_output=True, text=True)
if p.returncode:
    sys.exit(p.stderr[-4000:])
rows, cur = [], None
for ln in p.stderr.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
demangle = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} SGPR VGPR AGPR scratch occ s-spill v-spill  LDS")
for r, d in zip(rows, demangle):
    short = re.sub(r"\(.*", "", d).replace("r3d::", "").replace("void ", "")
    if flt and flt not in short:
        continue
    print(f"{short[:70]:70s} {r.get('TotalSGPRs', r.get('SGPRs', 0)):4d} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('ScratchSize', 0):7d} "
          f"{r.get('Occupancy', 0):3d} {r.get('SGPRs Spill', 0):7d} {r.get('VGPRs Spill', 0):7d} {r.get('LDS Size', 0):5d}")
