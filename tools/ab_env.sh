# A/B of environment settings on the bench: tools/ab_env.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   ("-" = no setting)
R=$GRAFT_REPO_ROOT; tag=$1; shift; mkdir -p $R/gpurun_out/$tag; cd $R
i=0
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  env $e timeout -k 10 200 python bench.py --no-extra-legs --no-cpu-baseline --repeats 3 > gpurun_out/$tag/$i.json 2> gpurun_out/$tag/$i.err || { tail -5 gpurun_out/$tag/$i.err; }
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/$i.json"))
    print("[$e]", d["value"], d["repeats"]["ms_per_step"], d["roofline"]["api_calls_ms"], d["config"]["ms_per_step_one_step_in_flight"], d["config"]["insert_paths_one_step"]["evaluated_twice"])
except Exception as ex:
    print("[$e] failed", ex)
PY
  i=$((i+1))
done
