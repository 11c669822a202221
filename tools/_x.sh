export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof2 && rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --cpu-iters 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_under_profiler.json 2>/dev/null
cd $GRAFT_REPO_ROOT
DB2=$(find /tmp/prof2 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB2 2>/dev/null | head -40 > gpurun_out/cfg2_kernel_stats.csv; head -30 gpurun_out/cfg2_kernel_stats.csv
ITER_BACK=3 python tools/rocpd_gaps.py $DB2 0 100000 > gpurun_out/iteration_timeline.txt 2>/dev/null; grep -n "onelaunch" -B3 -A3 gpurun_out/iteration_timeline.txt | head -60
