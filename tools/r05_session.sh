# One profiling session on the GPU box (round 5): kernel trace + the two PMC passes for config C2 (one step at a time and
# three in flight), the SQ counters of its kernels, config C5 (256 scans, 64 distinct) and the shuffled variant of C2;
# summaries into profiles/r05_* (copied to gpurun_out/prof/ so that they travel back).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-extra-legs --repeats 1"
mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --overlap 1 --steps 5 --warmup 2 $A > /tmp/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $R/bench.py --overlap 1 --steps 2 --warmup 1 $A > /tmp/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $R/bench.py --overlap 1 --steps 2 --warmup 1 $A > /tmp/p3.log 2>&1
cd $R && python3 tools/make_profiles.py /tmp/p_stats /tmp/p_fetch /tmp/p_write r05 && cp profiles/r05_* gpurun_out/prof/
echo "C2 done"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats3 -- python3 $R/bench.py --steps 6 --warmup 3 $A > /tmp/p4.log 2>&1
f=$(find /tmp/p_stats3 -name "*kernel_stats.csv" | head -1); (head -1 $f; grep "r3d::" $f) > $R/gpurun_out/prof/r05_kernel_stats_three_in_flight.csv
# the shuffled variant (virtual point order at begin)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s_stats -- python3 $R/bench.py --order shuffled --overlap 1 --steps 5 --warmup 2 $A > /tmp/s1.log 2>&1
f=$(find /tmp/s_stats -name "*kernel_stats.csv" | head -1); (head -1 $f; grep "r3d::" $f) > $R/gpurun_out/prof/r05_shuffled_kernel_stats.csv
echo "three in flight + shuffled done"
# SQ counters: where do the wave cycles of the streaming kernels go?
bash $R/tools/pmc_sq.sh > $R/gpurun_out/prof/r05_sq.txt 2>&1 || true
echo "SQ done"
# config C5, 256 scans per batch, 64 distinct (what the c5 leg of the default bench line runs)
C="--config C5 --scenes 256 --distinct 64"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c_stats -- python3 $R/bench.py $C --overlap 1 --steps 3 --warmup 1 $A > /tmp/c1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/c_fetch -- python3 $R/bench.py $C --overlap 1 --steps 1 --warmup 1 $A > /tmp/c2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/c_write -- python3 $R/bench.py $C --overlap 1 --steps 1 --warmup 1 $A > /tmp/c3.log 2>&1
cd $R && python3 tools/make_profiles.py /tmp/c_stats /tmp/c_fetch /tmp/c_write r05_c5 && cp profiles/r05_c5_* gpurun_out/prof/
ls -la gpurun_out/prof
