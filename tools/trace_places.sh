# timeline of one placement-search call (kernel trace): tools/trace_places.sh [B]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -- python3 $R/tools/bench_places.py ${1:-64} 5 > /tmp/tp.log 2>&1
f=$(find /tmp/tp -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_place" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete call: from the last k_place_centres on
last = max(i for i, r in enumerate(rows) if "k_place_centres" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1]
    print(f"{n:28s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  end {(int(r['End_Timestamp']) - t0) / 1e3:8.1f} us  queue {r.get('Queue_Id', '?')}")
PY
