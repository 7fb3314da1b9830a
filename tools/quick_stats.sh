# per-kernel durations of one bench configuration: tools/quick_stats.sh <tag> [bench args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/qs_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs_$tag -- python3 $R/bench.py --no-extra-legs --overlap 1 --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/qs_$tag.log 2>&1
f=$(find /tmp/qs_$tag -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out
(head -1 $f; grep "r3d::" $f) > $R/gpurun_out/qs_$tag.csv
python3 $R/tools/show_stats.py $R/gpurun_out/qs_$tag.csv
