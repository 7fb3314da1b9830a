# kernel stats of the placement search at B frames x 5 samples: tools/prof_places.sh [B]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/tools/bench_places.py ${1:-64} 5 > /tmp/pp.log 2>&1
grep -v rocprofv3 /tmp/pp.log | tail -3; find /tmp/pp -name "*.csv" | head
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    if 'k_place' in n or 'k_chunk' in n or 'k_alive' in n:
        print(f"{n:34s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e3:9.1f}")
PY
# every call of the point passes, in launch order (the three k_place_road_min passes differ)
t=$(find /tmp/pp -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_place' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-10:]
for r in last:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0]
    print(f"  {n:28s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}")
PY
