set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --overlap 1 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $R/bench.py --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $R/bench.py --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/p3.log 2>&1
cd $R && python3 tools/make_profiles.py /tmp/p_stats /tmp/p_fetch /tmp/p_write r02 && mkdir -p gpurun_out/prof && cp profiles/r02_* gpurun_out/prof/
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats3 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /tmp/p4.log 2>&1
f=$(find /tmp/p_stats3 -name "*kernel_stats.csv" | head -1); (head -1 $f; grep "r3d::" $f) > $R/gpurun_out/prof/r02_kernel_stats_three_in_flight.csv
# config C5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c_stats -- python3 $R/bench.py --config C5 --overlap 1 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/c1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/c_fetch -- python3 $R/bench.py --config C5 --overlap 1 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/c2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/c_write -- python3 $R/bench.py --config C5 --overlap 1 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/c3.log 2>&1
cd $R && python3 tools/make_profiles.py /tmp/c_stats /tmp/c_fetch /tmp/c_write r02_c5 && cp profiles/r02_c5_* gpurun_out/prof/
ls -la gpurun_out/prof
