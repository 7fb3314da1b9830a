# A/B of insert variants: tools/ab_insert.sh "<bench args>" "ENV=.. ENV=.." "ENV=.." ...   (each further argument: one environment)
args=$1; shift
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python bench.py --no-extra-legs --no-cpu-baseline $args > gpurun_out/r03_ab_$i.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03_ab_$i.log").read().strip().splitlines()[-1])
print("[$e]", "3-in-flight", d["ms_per_step"], "serial", d["config"]["ms_per_step_one_step_in_flight"], d["roofline"]["api_calls_ms"], d["config"].get("insert_paths_one_step"))
PY
done
