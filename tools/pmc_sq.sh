# SQ counters of the bench's kernels: where do the wave cycles go?  tools/pmc_sq.sh [bench args]
# two passes: issue / wait statistics, then the instruction cache
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--overlap 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --repeats 1 $@"
rm -rf /tmp/p_sq /tmp/p_ic
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d /tmp/p_sq -- python3 $R/bench.py $A > /tmp/psq.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES --output-format csv -d /tmp/p_ic -- python3 $R/bench.py $A > /tmp/pic.log 2>&1
python3 - $(find /tmp/p_sq -name "*counter_collection.csv" | head -1) $(find /tmp/p_ic -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); first = None
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void r3d::", "").replace("r3d::", "")
        first = first or r["Counter_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first: n[k] += 1
    for k, c in acc.items():
        if not n[k] or not k.startswith("k_"): continue
        print(f"{k[:44]:44s} launches {n[k]:3d} " + " ".join(f"{name} {v / n[k]:.4g}" for name, v in sorted(c.items())))
PY
