# kernel trace of the bench with three steps in flight: which kernels run at the same time, and for how long?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tr -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > /tmp/ptr.log 2>&1
f=$(find /tmp/p_tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "r3d::" in r["Kernel_Name"]]
ev = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("r3d::", "").split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id", r.get("Queue_Id", "?"))))
ev.sort()
# a window inside the pipelined part of the run: from the 6th to the 16th launch of the insert kernel
ic = [e for e in ev if e[2] == "k_insert_chain"]
t0, t1 = ic[5][0], ic[15][0]
ev = [e for e in ev if e[1] > t0 and e[0] < t1]
print(f"window {1e-6*(t1-t0):.3f} ms, {len(ev)} kernels, per step {1e-6*(t1-t0)/10:.3f} ms")
pts = []
for s, e, n, q in ev:
    pts.append((max(s, t0), 1, n)); pts.append((min(e, t1), -1, n))
pts.sort()
active = collections.Counter(); last = t0; hist = collections.Counter(); combo = collections.Counter()
for t, d, n in pts:
    big = ("k_bounds", "k_project", "k_insert_chain", "k_alive_write")
    k = tuple(sorted(f"{active[x]}x{x}" if active[x] > 1 else x for x in active if active[x] > 0 and x in big))
    hist[sum(active[x] for x in active if x in big)] += t - last
    combo[k] += t - last
    last = t
    active[n] += d
tot = sum(hist.values())
print("big kernels running at once: " + ", ".join(f"{k}: {100*v/tot:.1f}%" for k, v in sorted(hist.items())))
for k, v in combo.most_common(12):
    print(f"  {100*v/tot:5.1f}%  {' + '.join(k) if k else '(none)'}")
dur = collections.defaultdict(list)
for s, e, n, q in ev: dur[n].append(e - s)
for n, d in sorted(dur.items(), key=lambda x: -sum(x[1])):
    print(f"  {n:18s} n={len(d):3d} avg {1e-3*sum(d)/len(d):8.1f} us  min {1e-3*min(d):8.1f}  max {1e-3*max(d):8.1f}")
PY
