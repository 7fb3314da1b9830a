#!/usr/bin/env python3
"""Instructions per source line range of one kernel, from an assembly listing made with -gline-tables-only -S:

    hipcc --offload-arch=gfx950 <flags of csrc/Makefile> --cuda-device-only -gline-tables-only -S X.hip -o X.s
    python tools/isa_lines.py X.s <mangled-name substring> [bucket]

Prints the instruction count per bucket of `bucket` source lines (default 25) of the main file, spill traffic apart."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 25
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
files, cur = {}, (0, 0)
per, spill = collections.Counter(), collections.Counter()
for l in lines[:start]:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', t)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
        continue
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    op = t.split()[0]
    k = (files.get(cur[0], str(cur[0])).split("/")[-1], cur[1] // bucket * bucket)
    per[k] += 1
    if op in ("v_readlane_b32", "v_writelane_b32", "s_nop", "scratch_load_dword", "scratch_store_dword"):
        spill[k] += 1
tot = sum(per.values())
print(f"{lines[start][:-1]}: {tot} instructions, {sum(spill.values())} of them readlane / writelane / s_nop / scratch")
for k in sorted(per):
    if per[k] >= 40:
        print(f"  {k[0]:22s} {k[1]:5d}+  {per[k]:6d}  spill-related {spill[k]:5d}")
