cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in a b; do
  if [ $tag = b ]; then export R3D_INSERT_NT=1024 R3D_INSERT_LDS_KB=160; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/q_f$tag -- python3 $R/bench.py --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/q1.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/q_w$tag -- python3 $R/bench.py --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/q2.log 2>&1
  python3 - /tmp/q_f$tag /tmp/q_w$tag $tag <<'PY'
import csv, glob, sys, collections
def avg(d, c):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    v = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "k_insert" in r["Kernel_Name"]:
            v[r["Kernel_Name"].split("(")[0][-30:]].append(float(r["Counter_Value"]))
    return {k: sum(x)/len(x) for k, x in v.items()}
print(sys.argv[3], "fetch KB", avg(sys.argv[1], "FETCH_SIZE"), "write KB", avg(sys.argv[2], "WRITE_SIZE"))
PY
done
