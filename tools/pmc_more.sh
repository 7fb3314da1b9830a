# More SQ counters of the bench's kernels, one pass per quoted list:  tools/pmc_more.sh "CTR CTR ..." "CTR ..." [-- bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
lists=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do lists+=("$1"); shift; done
[ "$1" = "--" ] && shift
A="--overlap 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --repeats 1 $@"
i=0
for l in "${lists[@]}"; do
  i=$((i+1))
  rm -rf /tmp/p_more_$i
  rocprofv3 --kernel-trace --pmc $l --output-format csv -d /tmp/p_more_$i -- python3 $R/bench.py $A > /tmp/pmore_$i.log 2>&1
  python3 - $(find /tmp/p_more_$i -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); first = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void r3d::", "").replace("r3d::", "")
    first = first or r["Counter_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == first: n[k] += 1
for k, c in acc.items():
    if not n[k] or not k.startswith("k_"): continue
    print(f"{k[:44]:44s} launches {n[k]:3d} " + " ".join(f"{name} {v / n[k]:.4g}" for name, v in sorted(c.items())))
PY
done
