# FETCH_SIZE / WRITE_SIZE of the C5 leg under the environment given by the caller: tools/pmc_c5.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1
A="--no-cpu-baseline --no-extra-legs --repeats 1 --config C5 --scenes 256 --distinct 8 --overlap 1 --steps 1 --warmup 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${tag}_fetch -- python3 $R/bench.py $A > /tmp/${tag}_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${tag}_write -- python3 $R/bench.py $A > /tmp/${tag}_w.log 2>&1
python3 - <<PY
import csv, glob, collections
for what in ("fetch", "write"):
    f = glob.glob("/tmp/${tag}_%s/**/*counter_collection.csv" % what, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:4]:
        print(what, k, len(v), "launches, KB per launch avg", round(sum(v) / len(v)))
PY
