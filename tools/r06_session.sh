# One profiling session on the GPU box (round 6): kernel trace + the two PMC passes for config C2 (one step at a time and
# three in flight), the SQ counters of its kernels, config C5 (256 scans, 64 distinct) and the shuffled variant of C2;
# summaries into profiles/r06_* (copied to gpurun_out/prof/ so that they travel back).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-extra-legs --repeats 1"
mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --overlap 1 --steps 5 --warmup 2 $A > /tmp/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $R/bench.py --overlap 1 --steps 2 --warmup 1 $A > /tmp/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $R/bench.py --overlap 1 --steps 2 --warmup 1 $A > /tmp/p3.log 2>&1
cd $R && python3 tools/make_profiles.py /tmp/p_stats /tmp/p_fetch /tmp/p_write r06 && cp profiles/r06_* gpurun_out/prof/
echo "C2 done"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats3 -- python3 $R/bench.py --steps 6 --warmup 3 $A > /tmp/p4.log 2>&1
f=$(find /tmp/p_stats3 -name "*kernel_stats.csv" | head -1); (head -1 $f; grep "r3d::" $f) > $R/gpurun_out/prof/r06_kernel_stats_three_in_flight.csv
# the shuffled variant (virtual point order at begin)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s_stats -- python3 $R/bench.py --order shuffled --overlap 1 --steps 5 --warmup 2 $A > /tmp/s1.log 2>&1
f=$(find /tmp/s_stats -name "*kernel_stats.csv" | head -1); (head -1 $f; grep "r3d::" $f) > $R/gpurun_out/prof/r06_shuffled_kernel_stats.csv
echo "three in flight + shuffled done"
# the placed leg (placement search + candidate loop per insert slot, four lanes): per-kernel totals
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_stats -- python3 $R/tools/bench_placed.py 256 5 > /tmp/pl.log 2>&1
python3 $R/tools/kstats.py /tmp/pl_stats 40 > $R/gpurun_out/prof/r06_placed_kernel_stats.txt 2>&1 || true
echo "placed done"
# config C5, 256 scans per batch, 64 distinct (what the c5 leg of the default bench line runs)
C="--config C5 --scenes 256 --distinct 64"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c_stats -- python3 $R/bench.py $C --overlap 1 --steps 3 --warmup 1 $A > /tmp/c1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/c_fetch -- python3 $R/bench.py $C --overlap 1 --steps 1 --warmup 1 $A > /tmp/c2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/c_write -- python3 $R/bench.py $C --overlap 1 --steps 1 --warmup 1 $A > /tmp/c3.log 2>&1
cd $R && python3 tools/make_profiles.py /tmp/c_stats /tmp/c_fetch /tmp/c_write r06_c5 && cp profiles/r06_c5_* gpurun_out/prof/
ls -la gpurun_out/prof
