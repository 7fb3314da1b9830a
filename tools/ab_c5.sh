# A/B of environment settings on the C5 leg: tools/ab_c5.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   ("-" = no setting)
R=$GRAFT_REPO_ROOT; tag=$1; shift; mkdir -p $R/gpurun_out/$tag; cd $R
i=0
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  env $e timeout -k 10 300 python bench.py --config C5 --scenes 256 --distinct 8 --steps 4 --warmup 1 --no-extra-legs --no-cpu-baseline --repeats 2 > gpurun_out/$tag/c5_$i.json 2> gpurun_out/$tag/c5_$i.err || { tail -5 gpurun_out/$tag/c5_$i.err; }
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/c5_$i.json"))
    print("[$e] C5", d["value"], d["repeats"]["ms_per_step"], d["roofline"]["api_calls_ms"], d["config"]["ms_per_step_one_step_in_flight"], d["config"]["insert_paths_one_step"])
except Exception as ex:
    print("[$e] failed", ex)
PY
  i=$((i+1))
done
