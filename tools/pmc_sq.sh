# SQ counters of the bench's kernels (one pass, 8 slots): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d /tmp/p_sq -- python3 $R/bench.py --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/psq.log 2>&1
f=$(find /tmp/p_sq -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void r3d::", "").replace("r3d::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in acc.items():
    if not n[k]: continue
    w = c["SQ_WAVE_CYCLES"] or 1
    print(f"{k[:40]:40s} launches {n[k]:4d} wave_cyc/launch {w/n[k]:.3e} wait_any {c['SQ_WAIT_ANY']/w:.2f} wait_inst {c['SQ_WAIT_INST_ANY']/w:.2f} active {c['SQ_ACTIVE_INST_ANY']/w:.2f} active_valu {c['SQ_ACTIVE_INST_VALU']/w:.2f} valu/launch {c['SQ_INSTS_VALU']/n[k]:.3e} salu/launch {c['SQ_INSTS_SALU']/n[k]:.3e} busy/launch {c['SQ_BUSY_CYCLES']/n[k]:.3e}")
PY
