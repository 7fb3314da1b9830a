# The kernels of ONE step (begin, insert_many, finish) on the timeline: start / end / gap to the previous kernel.
#   tools/trace_step.sh <tag> [bench args]      -> gpurun_out/step_<tag>.txt
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ts_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/ts_$tag -- python3 $R/bench.py --no-extra-legs --overlap 1 --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline "$@" > /tmp/ts_$tag.log 2>&1
f=$(find /tmp/ts_$tag -name "*kernel_trace.csv" | head -1)
mkdir -p $R/gpurun_out
python3 - $f > $R/gpurun_out/step_$tag.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "r3d::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("r3d::", "")
# the last complete step: from the last k_begin_init to the k_pack_log / k_alive_write that follows it
starts = [i for i, r in enumerate(rows) if name(r) == "k_begin_init"]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:7.1f}"
    print(f"{name(r):34s} start {(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:8.1f}  gap {gap}")
    busy += e - s
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {(prev_end - t0 - busy) / 1e3:.1f} us")
PY
cat $R/gpurun_out/step_$tag.txt
