#!/bin/bash
# A/B of environment settings on the headline configuration: tools/ab_env.sh OUTDIR "VAR=a VAR2=b" "VAR=c" ...
# one bench.py run (headline only, no CPU legs) per setting; prints step time, the three API calls and the insert paths
out=$1; shift
mkdir -p "$out"
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting timeout -k 10 300 python bench.py --no-extra-legs --no-cpu-baseline ${AB_ARGS} > "$out/ab_$i.json" 2> "$out/ab_$i.err" || { echo "run $i failed"; tail -3 "$out/ab_$i.err"; exit 1; }
  python - "$out/ab_$i.json" "$setting" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
p = d["config"]["insert_paths_one_step"]
print(f"{sys.argv[2][-40:]:40s} step {d['ms_per_step']:.3f} ms  ({min(d['repeats']['ms_per_step']):.3f}-{max(d['repeats']['ms_per_step']):.3f}); alone {d['config']['ms_per_step_one_step_in_flight']:.3f}; "
      f"{d['roofline']['api_calls_ms']}; twice {p['evaluated_twice']}, parked {p.get('parked_with_record')}/{p.get('parked_unevaluated')}, from record {p.get('committed_from_record')}, pooled {p['tiles_pooled']}")
PY
done
