# One GPU session of round 4: tools/r04_session.sh <tag> [what...]   (what: tests quick bench stamps c5 c5stamps)
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
for w in "$@"; do
  case $w in
    tests)  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }; tail -3 $O/tests.log ;;
    quick)  timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q > $O/quick.log 2>&1 || { tail -40 $O/quick.log; exit 1; }; tail -3 $O/quick.log ;;
    bench)  timeout -k 10 300 python bench.py --no-extra-legs --no-cpu-baseline --repeats 3 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
            python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("C2", d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"], d["roofline"]["api_calls_ms"], d["config"]["ms_per_step_one_step_in_flight"], d["config"]["insert_paths_one_step"])
PY
            ;;
    stamps) R3D_LIB=pcl-augmentation_amd/libreal3daug_hip_stamps.so timeout -k 10 300 python tools/stamps_insert.py > $O/stamps.md 2> $O/stamps.err || { tail -20 $O/stamps.err; exit 1; }; head -22 $O/stamps.md ;;
    c5)     timeout -k 10 400 python bench.py --config C5 --scenes 256 --distinct 8 --steps 4 --warmup 1 --no-extra-legs --no-cpu-baseline --repeats 2 > $O/c5.json 2> $O/c5.err || { tail -20 $O/c5.err; exit 1; }
            python - <<PY
import json
d=json.load(open("$O/c5.json"))
print("C5", d["value"], d["ms_per_step"], d["roofline"]["api_calls_ms"], d["config"]["ms_per_step_one_step_in_flight"], d["pipeline_frac_of_hbm_peak"], d["config"]["insert_paths_one_step"])
PY
            ;;
    c5stamps) R3D_STAMPS_CONFIG=C5 R3D_STAMPS_SCANS=256 R3D_STAMPS_SLOTS=50 R3D_LIB=pcl-augmentation_amd/libreal3daug_hip_stamps.so timeout -k 10 400 python tools/stamps_insert.py > $O/c5stamps.md 2> $O/c5stamps.err || { tail -20 $O/c5stamps.err; exit 1; }; head -20 $O/c5stamps.md ;;
  esac
done
