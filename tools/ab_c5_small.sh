# A/B on small C5 batches: tools/ab_c5_small.sh <scenes> "ENV..." ...
R=$GRAFT_REPO_ROOT; n=$1; shift; cd $R
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  env $e timeout -k 10 300 python bench.py --config C5 --scenes $n --distinct 8 --steps 4 --warmup 1 --no-extra-legs --no-cpu-baseline --repeats 2 > /tmp/c5s.json 2> /tmp/c5s.err || tail -3 /tmp/c5s.err
  python - <<PY
import json
d=json.load(open("/tmp/c5s.json"))
print("[$e] C5 x $n", d["value"], d["repeats"]["ms_per_step"], d["roofline"]["api_calls_ms"]["r3d_batch_insert_many_50"], d["config"]["insert_paths_one_step"]["evaluated_twice"])
PY
done
